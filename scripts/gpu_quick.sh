#!/bin/bash
# quick GPU iteration: SR-side parity tests + bench line (no CPU baseline)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x -k "${1:-sr or conv or fusion or warp or synthesis or toplane or stream}" 2>&1 | tail -4
python bench.py --steps 40 --warmup 5 --no-cpu-baseline ${2:-} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'single', d.get('value_single_stream'))
print('conv avg ms', r['avg_launch_ms'], 'frac', r['frac'], '| upconv', r.get('upconv_fir_f16x3_kernel'))
print('breakdown', d.get('breakdown_ms_per_frame'))
print('torso', {k: v for k, v in d.get('torso_frame', {}).items() if k in ('ms_per_frame', 'fps', 'breakdown_ms_per_frame')})
"
