"""Compressed instruction schedule between the s_barriers of one kernel in a hipcc .s listing (M16/M8 = f16 / fp8 MFMA, R = ds_read_b128,
D = LDS-DMA, W = s_waitcnt, n = s_nop): what the compiler made of a hand-ordered main loop.
usage: isa_sched.py <file.s> <mangled-name-substring> [first-barrier [count]]"""
import re, sys
s = open(sys.argv[1]).read()
sub = sys.argv[2]
name = [m for m in re.findall(r'^(_Z\w+):', s, re.M) if sub in m][0]
meta = re.search(r'\.name:\s+' + name + r'\n(.*?)\.wavefront_size', s, re.S)
print(name, re.findall(r'\.(vgpr_count|vgpr_spill_count|sgpr_spill_count):\s+(\d+)', meta.group(1)))
i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
ins = [l.strip().split()[0] for l in s[i:j].split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
bars = [k for k, x in enumerate(ins) if x == 's_barrier']
short = {'v_mfma_f32_32x32x16_f16': 'M16', 'v_mfma_scale_f32_32x32x64_f8f6f4': 'M8', 'ds_read_b128': 'R', 's_waitcnt': 'W', 'global_load_lds_dwordx4': 'D', 's_nop': 'n'}
def compress(seq):
    out = []; prev = None; c = 0
    for x in seq:
        x = short.get(x, x)
        if x.startswith('s_') and x != 's_barrier': continue
        if x == prev: c += 1
        else:
            if prev: out.append(prev + (str(c) if c > 1 else ''))
            prev = x; c = 1
    out.append(prev + (str(c) if c > 1 else ''))
    return ' '.join(out)
b0 = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n = int(sys.argv[4]) if len(sys.argv) > 4 else 4
print(len(ins), "instructions,", len(bars), "barriers")
for a, b in zip(bars[b0:b0 + n], bars[b0 + 1:b0 + n + 1]): print(compress(ins[a:b]))
