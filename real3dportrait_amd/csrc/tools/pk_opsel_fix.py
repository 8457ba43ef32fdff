#!/usr/bin/env python3
"""Rewrite the packed-FP32 instruction forms that compute WRONG results on gfx950 (MI355X) while another wave of the SIMD executes MFMAs.

Found in round 3 (scripts/probes/coexec_probe.hip patterns 34-50): a VOP3P f32 instruction (v_pk_mul_f32 / v_pk_add_f32 /
v_pk_fma_f32) whose LOW half reads the HIGH dword of src1 or src2 (op_sel bit of src1 / src2 set) while src0's op_sel bit is clear returns
a wrong low-half result in lanes 48-63, only while a co-resident wave issues MFMAs (never alone; wait states do not help; VGPR or SGPR
source alike).  The same selection on src0 is exact, so:

  * mul / add (and fma with a clean src2): commutative in src0 / src1 -> swap the two sources together with their op_sel / op_sel_hi /
    neg_lo / neg_hi bits;
  * both sources crossed, or an fma whose src2 is crossed: the two dwords of that VGPR pair are exchanged in place (v_swap_b32) in front
    of the instruction, its op_sel / op_sel_hi bits flipped, and exchanged back behind it unless the pair is the destination.

The v_swap_b32 path has never been needed by a shipped object (0 sites) and has not been validated on hardware next to MFMAs (no hazard /
s_nop analysis around the inserted swaps), so it is REFUSED unless R3D_PK_ALLOW_DWORD_SWAP=1: a build that would take it fails, and whoever
hits that validates the path first.  Every other packed instruction with 64-bit operands (today only v_pk_mov_b32, not
commutative) must not carry a crossed src1 / src2 either: such a line fails the rewrite instead of passing through unchecked -- its probe
(scripts/probes/coexec_probe.hip pattern 47) is inconclusive, so the form is simply not allowed into the library.

Usage: pk_opsel_fix.py in.s out.s   (prints a summary; `--check file.s` only counts and exits 1 if a hazardous form is present;
`--selftest <llvm bin dir>` assembles a file of known hazardous forms, rewrites it, re-assembles the result and checks the encodings --
the build runs it first, so a ROCm whose assembler spells these instructions differently stops the build instead of shipping them)
"""
import os
import re
import subprocess
import sys
import tempfile

PK = re.compile(r"^(\s*)(v_pk_(mul|add|fma)_f32)\s+(.*)$")
OTHER_PK64 = re.compile(r"^\s*(v_pk_mov_b32)\s+(.*)$")        # packed ops with 64-bit operands that the rewriter cannot fix
MODS = ("op_sel", "op_sel_hi", "neg_lo", "neg_hi")


def parse(line):
    m = PK.match(line)
    if not m:
        return None
    indent, op, kind, rest = m.groups()
    rest = rest.split(";")[0].rstrip()
    mods = {}
    for k in MODS:
        mm = re.search(r"\b%s:\[([01,]+)\]" % k, rest)
        if mm:
            mods[k] = [int(x) for x in mm.group(1).split(",")]
            rest = rest.replace(mm.group(0), "")
    ops = [t.strip() for t in rest.strip().rstrip(",").split(",")]
    ops = [t for t in ops if t]
    nsrc = 3 if kind == "fma" else 2
    if len(ops) != 1 + nsrc:
        raise ValueError("cannot parse: " + line)
    mods.setdefault("op_sel", [0] * nsrc)
    mods.setdefault("op_sel_hi", [1] * nsrc)
    mods.setdefault("neg_lo", [0] * nsrc)
    mods.setdefault("neg_hi", [0] * nsrc)
    return indent, op, kind, ops[0], ops[1:], mods


def hazardous(p):
    _, _, kind, _, srcs, mods = p
    sel = mods["op_sel"]
    return any(sel[i] == 1 and not is_const(srcs[i]) for i in range(1, len(sel)))


def is_const(tok):
    return not (tok.startswith("v") or tok.startswith("s[") or re.match(r"s\d", tok) or tok in ("vcc", "exec"))


def fmt_mods(mods, n):
    out = []
    for k, dflt in (("op_sel", 0), ("op_sel_hi", 1), ("neg_lo", 0), ("neg_hi", 0)):
        v = mods[k]
        if any(x != dflt for x in v):
            out.append("%s:[%s]" % (k, ",".join(str(x) for x in v)))
    return (" " + " ".join(out)) if out else ""


def half(tok, hi):
    """The 32-bit register of a 64-bit operand's low / high dword (constants are the same for both)."""
    m = re.match(r"([vs])\[(\d+):(\d+)\]$", tok)
    if m:
        return "%s%d" % (m.group(1), int(m.group(2)) + (1 if hi else 0))
    if tok == "vcc":
        return "vcc_hi" if hi else "vcc_lo"
    return tok


def regs_of(tok):
    m = re.match(r"([vs])\[(\d+):(\d+)\]$", tok)
    if m:
        return {(m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"([vs])(\d+)$", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def swap_fix(p, i, stats):
    """Source i (1 or 2) has its op_sel bit set: swap the two dwords of that VGPR pair in place (v_swap_b32), flip the source's op_sel /
    op_sel_hi bits, and swap back afterwards unless the pair is the destination."""
    indent, op, kind, dst, srcs, mods = p
    c = srcs[i]
    if os.environ.get("R3D_PK_ALLOW_DWORD_SWAP") != "1":
        raise RuntimeError("pk_opsel_fix: the v_swap_b32 rewrite (both sources crossed, or a crossed src2) is not validated on hardware; "
                           "refusing: " + " ".join([op, dst] + srcs) + fmt_mods(mods, len(srcs)))
    if not re.match(r"v\[\d+:\d+\]$", c):
        raise RuntimeError("pk_opsel_fix: cannot dword-swap a non-VGPR source: " + " ".join([op, dst] + srcs))
    for j, t in enumerate(srcs):
        if j != i and regs_of(t) & regs_of(c):
            raise RuntimeError("pk_opsel_fix: crossed source is used twice: " + " ".join([op, dst] + srcs))
    lo, hi = half(c, 0), half(c, 1)
    mods["op_sel"][i] = 0
    mods["op_sel_hi"][i] = 1 - mods["op_sel_hi"][i]
    out = ["%sv_swap_b32 %s, %s" % (indent, lo, hi), "%s%s %s, %s%s" % (indent, op, dst, ", ".join(srcs), fmt_mods(mods, len(srcs)))]
    if c != dst:
        if regs_of(c) & regs_of(dst):
            raise RuntimeError("pk_opsel_fix: destination overlaps the crossed source: " + " ".join([op, dst] + srcs))
        out.append("%sv_swap_b32 %s, %s" % (indent, lo, hi))
    stats["dword_swapped"] += 1
    return out


def other_pk64_hazard(line):
    """A packed op with 64-bit operands other than mul / add / fma whose src1 (or src2) op_sel bit is set."""
    m = OTHER_PK64.match(line)
    if not m:
        return False
    mm = re.search(r"\bop_sel:\[([01,]+)\]", m.group(2).split(";")[0])
    sel = [int(x) for x in mm.group(1).split(",")] if mm else [0, 0]
    return any(sel[1:])


def fix_line(line, stats):
    if other_pk64_hazard(line):
        raise RuntimeError("pk_opsel_fix: packed instruction with a crossed src1 / src2 that cannot be rewritten: " + line.strip())
    p = parse(line)
    if p is None:
        return [line]
    stats["pk"] += 1
    if not hazardous(p):
        return [line]
    indent, op, kind, dst, srcs, mods = p
    sel = mods["op_sel"]
    if len(sel) == 3 and sel[2] == 1 and not is_const(srcs[2]):
        out = swap_fix(p, 2, stats)
        # the rewritten instruction may still have a crossed src1: run it through again
        res = []
        for l in out:
            res.extend(fix_line(l, {"pk": 0, "swapped": 0, "dword_swapped": 0}) if l.strip().startswith("v_pk_") else [l])
        return res
    if sel[0] == 0 or is_const(srcs[0]):               # src1 crossed, src0 clean: the sources of a mul / add / fma commute
        srcs = [srcs[1], srcs[0]] + srcs[2:]
        for k in MODS:
            mods[k] = [mods[k][1], mods[k][0]] + mods[k][2:]
        stats["swapped"] += 1
        return ["%s%s %s, %s%s" % (indent, op, dst, ", ".join(srcs), fmt_mods(mods, len(srcs)))]
    return swap_fix(p, 1, stats)                        # both crossed


SELFTEST_FORMS = [          # (hazardous form, the exact equivalent the rewriter must produce)
    ("v_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_mul_f32 v[2:3], v[6:7], v[4:5] op_sel:[1,0] op_sel_hi:[0,1]"),
    ("v_pk_mul_f32 v[2:3], v[4:5], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_mul_f32 v[2:3], v[2:3], v[4:5] op_sel:[1,0] op_sel_hi:[0,1]"),
    ("v_pk_add_f32 v[8:9], v[4:5], v[6:7] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]", "v_pk_add_f32 v[8:9], v[6:7], v[4:5] op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]"),
    ("v_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9] op_sel:[0,1,0] op_sel_hi:[1,0,1]", "v_pk_fma_f32 v[2:3], v[6:7], v[4:5], v[8:9] op_sel:[1,0,0] op_sel_hi:[0,1,1]"),
    ("v_pk_mul_f32 v[2:3], v[4:5], s[6:7] op_sel:[0,1]", "v_pk_mul_f32 v[2:3], s[6:7], v[4:5] op_sel:[1,0]"),
    ("v_pk_mul_f32 v[2:3], v[4:5], v[6:7]", "v_pk_mul_f32 v[2:3], v[4:5], v[6:7]"),                                      # clean: untouched
    ("v_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[1,0] op_sel_hi:[0,1]", "v_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[1,0] op_sel_hi:[0,1]"),
]


def selftest(llvm_bin):
    """Assembler round trip: hazardous forms -> rewriter -> clang assembles BOTH the input and the output for gfx950 (so the mnemonic /
    modifier spellings this script parses and prints are the ones this ROCm's assembler accepts) -> the rewritten object contains no
    hazardous form and encodes exactly the expected instructions; the refused forms raise."""
    clang, objdump = os.path.join(llvm_bin, "clang"), os.path.join(llvm_bin, "llvm-objdump")
    with tempfile.TemporaryDirectory() as tmp:
        def assemble(lines, name):
            src, obj = os.path.join(tmp, name + ".s"), os.path.join(tmp, name + ".o")
            open(src, "w").write(".text\n" + "\n".join(lines) + "\n")
            subprocess.check_call([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", src, "-o", obj])
            dis = subprocess.check_output([objdump, "-d", "--mcpu=gfx950", obj], text=True)
            return [re.sub(r"\s*//.*$", "", l).strip() for l in dis.splitlines() if "v_pk_" in l]
        stats = {"pk": 0, "swapped": 0, "dword_swapped": 0}
        fixed = []
        for form, _ in SELFTEST_FORMS:
            fixed.extend(fix_line("\t" + form, stats))
        assemble([f for f, _ in SELFTEST_FORMS], "in")                # the input spellings are what this assembler prints for hipcc's code
        got = assemble(fixed, "out")
        want = assemble([w for _, w in SELFTEST_FORMS], "want")
        if got != want:
            raise SystemExit("pk_opsel_fix selftest: rewritten encodings differ from the expected ones:\n  got  %s\n  want %s" % (got, want))
        for l in got:
            pp = parse("\t" + l)
            if pp is None or hazardous(pp):
                raise SystemExit("pk_opsel_fix selftest: hazardous or unparsable form after the rewrite: " + l)
        for refused in ("v_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[1,1] op_sel_hi:[0,0]", "v_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9] op_sel:[0,0,1] op_sel_hi:[1,1,0]",
                        "v_pk_mov_b32 v[2:3], v[4:5], v[6:7] op_sel:[0,1]"):
            allow = os.environ.pop("R3D_PK_ALLOW_DWORD_SWAP", None)      # the refusal is what is checked here: the escape hatch stays usable from make
            try:
                fix_line("\t" + refused, {"pk": 0, "swapped": 0, "dword_swapped": 0})
            except RuntimeError:
                continue
            finally:
                if allow is not None:
                    os.environ["R3D_PK_ALLOW_DWORD_SWAP"] = allow
            raise SystemExit("pk_opsel_fix selftest: a form without a validated rewrite was not refused: " + refused)
    print("pk_opsel_fix selftest: %d forms round-tripped through the gfx950 assembler, %d rewritten, refusals ok" % (len(SELFTEST_FORMS), stats["swapped"]))


def main():
    if sys.argv[1] == "--selftest":
        selftest(sys.argv[2] if len(sys.argv) > 2 else "/opt/rocm/lib/llvm/bin")
        return
    if sys.argv[1] == "--check":
        bad = 0
        for fn in sys.argv[2:]:
            for ln, line in enumerate(open(fn), 1):
                if other_pk64_hazard(line.rstrip("\n")):
                    bad += 1
                    print("%s:%d: %s" % (fn, ln, line.strip()))
                    continue
                p = parse(line.rstrip("\n")) if "v_pk_" in line else None
                if p is not None and hazardous(p):
                    bad += 1
                    if bad <= 10:
                        print("%s:%d: %s" % (fn, ln, line.strip()))
        print("hazardous packed-f32 op_sel forms: %d" % bad)
        sys.exit(1 if bad else 0)
    src, dst = sys.argv[1:3]
    stats = {"pk": 0, "swapped": 0, "dword_swapped": 0}
    out = []
    for line in open(src):
        line = line.rstrip("\n")
        out.extend(fix_line(line, stats) if "v_pk_" in line else [line])
    open(dst, "w").write("\n".join(out) + "\n")
    print("pk_opsel_fix %s: %d packed-f32 instructions, %d rewritten by source swap, %d by v_swap_b32 of the crossed pair"
          % (src.split("/")[-1], stats["pk"], stats["swapped"], stats["dword_swapped"]))


if __name__ == "__main__":
    main()
