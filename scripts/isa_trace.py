#!/usr/bin/env python3
"""Compact trace of a kernel's device assembly (the .fix.s the Makefile keeps under real3dportrait_amd/lib/obj): runs of loads / stores / waits /
MFMAs / LDS ops / transcendentals in program order with the count of other instructions between them -- enough to see whether a load really
stays in flight across a compute segment.  usage: isa_trace.py <file.s> <symbol substring> [first line] [last line]"""
import re, sys
src, sym = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and sym in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start + 1:end]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else len(body)
kinds = [("global_load", "LOAD"), ("global_store", "STORE"), ("scratch_", "SCRATCH"), ("s_waitcnt", "WAIT"), ("v_mfma", "MFMA"), ("ds_", "LDS"),
         ("v_exp", "TRANS"), ("v_log", "TRANS"), ("v_rcp", "TRANS"), ("s_cbranch", "BR"), ("s_branch", "BR"), ("v_readlane", "RDLANE"), ("v_writelane", "WRLANE")]
prev, cnt, other, first_i = None, 0, 0, 0
def flush():
    if prev: print("%6d  %-8s x%-3d %s" % (first_i, prev, cnt, detail))
nvalu = 0
for i, l in enumerate(body):
    if i < lo or i >= hi: continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if t.startswith(".LBB"):
            flush(); prev = None
            print("%6d  %s" % (i, t))
        continue
    op = t.split()[0]
    k = next((name for pre, name in kinds if op.startswith(pre)), None)
    if k is None:
        other += 1
        continue
    if k == prev and k != "WAIT":
        cnt += 1
    else:
        flush()
        prev, cnt, first_i = k, 1, i
        detail = ("(+%d other before) " % other) + (t if k in ("WAIT", "BR") else op)
        other = 0
flush()
print("total lines", len(body))
