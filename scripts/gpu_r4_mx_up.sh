#!/bin/bash
# f16mx with block1's up-sampling conv on the fp8 correction MFMA (R3D_MX_UPCONV=1, default) vs on the 3-term fp16 split (=0): parity + SR time
cd "$(dirname "$0")/.." || exit 1
for v in 0 1; do
  echo "== R3D_MX_UPCONV=$v"
  R3D_MX_UPCONV=$v R3D_SR_PRECISION=f16mx python scripts/prof_sr.py 40 2>&1 | grep "SR 128"
done
R3D_MX_UPCONV=1 timeout 900 python -m pytest tests/test_gpu_mx.py tests/test_gpu_pinned_config.py -m gpu -q -s --no-header -p no:cacheprovider -k "mx or benchmarked" 2>&1 | grep -E "sr_full|heavy tail|benchmarked frame|passed|failed|FAILED|Error|error|assert|^E " | tail -40
