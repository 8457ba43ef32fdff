#!/bin/bash
# A/B of experiment builds of libr3d_hip.so on ONE box: bench.py (no extras) with each library given on the command line (paths relative to
# the repo root; "default" = the shipped one).  Prints value / single-stream / dominant-kernel time per library.
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  if [ "$lib" = default ]; then unset R3D_LIB; else export R3D_LIB=$PWD/$lib; fi
  python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-44s value %7.1f  1-stream %7.1f  conv %.4f ms frac %.4f  upconv %.4f ms  repeats %s' % ('$lib', d['value'], d['value_single_stream'] or 0, r['avg_launch_ms'], r['frac'], r['upconv_fir_f16x3_kernel']['avg_launch_ms'], d['repeats']['fps']))"
done
