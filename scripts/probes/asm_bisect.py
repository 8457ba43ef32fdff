#!/usr/bin/env python3
"""ISA-level bisection helper: takes the device assembly of scripts/probes/coexec_probe.hip, and writes code objects in
which ONE kernel's body has been mutated -- `s_nop 3` inserted after every VALU instruction whose index (among the body's VALU
instructions) lies in a given range, or after selected instruction indices -- so that the instruction pair whose spacing matters can be
located on the GPU with scripts/probes/coexec_mod.  CPU only (hipcc / clang / ld.lld).

    asm_bisect.py <dev.s> <kernel> <outdir> chunks <n>             # n variants, variant k pads the k-th of n equal VALU ranges
    asm_bisect.py <dev.s> <kernel> <outdir> range <lo> <hi> <n>    # the same inside VALU range [lo, hi)
    asm_bisect.py <dev.s> <kernel> <outdir> list                   # print the body with VALU indices
"""
import os, subprocess, sys
LLVM = "/opt/rocm/lib/llvm/bin"

def load(path, kernel):
    L = open(path).read().split("\n")
    a = next(i for i, l in enumerate(L) if l.startswith(kernel + ":"))
    b = next(i for i in range(a, len(L)) if "s_endpgm" in L[i])
    return L, a, b

def is_valu(l):
    t = l.strip()
    return t.startswith("v_") and not t.startswith("v_mfma")

def build(L, out):
    s = out + ".s"
    open(s, "w").write("\n".join(L))
    subprocess.check_call([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", out + ".o"])
    subprocess.check_call([LLVM + "/ld.lld", "-shared", out + ".o", "-o", out + ".hsaco"])
    os.remove(out + ".o"); os.remove(s)

def mutate(L, a, b, sel, pad="\ts_nop 3"):
    out, k = [], 0
    for i, l in enumerate(L):
        out.append(l)
        if a < i < b and is_valu(l):
            if sel(k):
                out.append(pad)
            k += 1
    return out

def main():
    path, kernel, outdir, mode = sys.argv[1:5]
    os.makedirs(outdir, exist_ok=True)
    L, a, b = load(path, kernel)
    nv = sum(1 for i in range(a + 1, b) if is_valu(L[i]))
    if mode == "list":
        k = 0
        for i in range(a + 1, b):
            tag = ""
            if is_valu(L[i]):
                tag = "%4d" % k; k += 1
            print("%s %s" % (tag.rjust(4), L[i]))
        return
    if mode == "chunks":
        lo, hi, n = 0, nv, int(sys.argv[5])
    else:
        lo, hi, n = int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    build(L, os.path.join(outdir, "base"))
    build(mutate(L, a, b, lambda k: lo <= k < hi), os.path.join(outdir, "all_%d_%d" % (lo, hi)))
    edges = [lo + (hi - lo) * j // n for j in range(n + 1)]
    for j in range(n):
        if edges[j + 1] > edges[j]:
            build(mutate(L, a, b, lambda k, j=j: edges[j] <= k < edges[j + 1]), os.path.join(outdir, "pad_%03d_%03d" % (edges[j], edges[j + 1])))
    print("VALU instructions in body:", nv, "variants in", outdir)

if __name__ == "__main__":
    main()
