"""Which host-side calls of the torso frame end in a device copy / fill (the __amd_rocclr_copyBuffer launches of the kernel trace): torch.profiler with stacks."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
G, clip, dec, scene = bench.build_scene(torch, dev, n_frames=8)
frame, flops = bench.build_torso_frame(torch, dev, G)
for t in range(3): frame(t)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for t in range(4): frame(t)
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.name.startswith("aten::") and e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::contiguous", "aten::clone", "aten::to", "aten::_to_copy", "aten::empty", "aten::empty_strided", "aten::full", "aten::ones", "aten::zeros")]
import collections
c = collections.Counter()
for e in ev:
    st = [s for s in (e.stack or []) if "real3dportrait_amd" in s or "bench.py" in s]
    c[(e.name, str(e.input_shapes)[:60], st[0][-90:] if st else "?")] += 1
for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:40]:
    print("%5.1f/frame  %-22s %-60s %s" % (v / 4.0, k[0], k[1], k[2]))
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12)[:3000])
