#!/bin/bash
# what the power management reports while the bench loop runs: average socket power, sclk, temperature (rocm-smi samples every 0.5 s)
cd "$(dirname "$0")/.." || exit 1
rocm-smi --showmaxpower --showpowercap 2>/dev/null | grep -v "^=\|^$" | head -8
(python bench.py --steps 2400 --warmup 24 --no-extras --no-cpu-baseline --no-traffic > /tmp/b.json 2>/dev/null) &
BP=$!
sleep 6
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor (junction|edge)" | tr '\n' ' ' | sed 's/GPU\[0\]\s*: //g'; echo
  sleep 0.5
done
wait $BP
python -c "
import json; d=json.loads([l for l in open('/tmp/b.json') if l.startswith('{')][0]); print('bench', d['value'], d['roofline']['shader_clock_ghz'], d['roofline']['shader_clock_ghz_other_kernels'])"
