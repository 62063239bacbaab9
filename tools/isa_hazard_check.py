"""Static check of the kernels that issue v_mfma_f32_16x16x4_f32 as inline asm (conv3x3_wino16.hip, conv3x3_wgrad_wino16.hip,
conv3x3_wino_k16.hip;
DESIGN.md 4g): the compiler does not know those statements are MFMAs, so it pads neither of the hazards it would otherwise pad.
This walks the gfx950 ISA hipcc emits and reports

  * a VALU write of an MFMA's A / B operand register within the two preceding wait states (gfx950 needs 2), and
  * accumulator copies (v_accvgpr_read / _write / _mov) inside a loop that contains MFMAs -- the register-allocator
    shuffle the inline asm exists to avoid.

usage: python tools/isa_hazard_check.py file.s [file.s ...]      (hipcc ... --cuda-device-only -S -o file.s file.hip)
       python tools/isa_hazard_check.py --build                  (compiles the two kernel files itself)
exit status 1 when anything is found."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ("conv3x3_wino16.hip", "conv3x3_wgrad_wino16.hip", "conv3x3_wino_k16.hip")
# (conv3x3_wino_k16.hip reads its accumulators inside the loop over output-channel tiles -- its epilogue: only writes / moves count there)
EPILOGUE_IN_LOOP = ("conv3x3_wino_k16.hip",)


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def check(path):
    raw = open(path).read().split("\n")
    lines = [l.strip() for l in raw if l.strip() and not l.strip().startswith((";", "."))]
    nmfma = hazards = 0
    for i, l in enumerate(lines):
        if not l.startswith("v_mfma"):
            continue
        nmfma += 1
        ops = l.split(None, 1)[1].split(",")
        use = regs(ops[1]) | regs(ops[2])
        ws, j = 0, i - 1
        while j >= 0 and ws < 2:
            p = lines[j]
            if p.startswith("s_nop"):
                ws += int(p.split()[1]) + 1
            else:
                ws += 1
                if p.startswith("v_") and not p.startswith(("v_mfma", "v_cmp")):
                    if regs(p.split(None, 1)[1].split(",")[0]) & use:
                        hazards += 1
                        print("%s: VALU -> MFMA hazard: `%s` then `%s`" % (path, p, l))
            j -= 1
    # accumulator copies inside MFMA loops: a backward branch to a label, with MFMAs in between
    labels = {l.split(":")[0]: i for i, l in enumerate(raw) if l.startswith(".LBB")}
    loops = []
    for i, l in enumerate(raw):
        m = re.search(r"s_cbranch\w+\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)", l)
        tgt = m and (m.group(1) or m.group(2))
        if tgt in labels and labels[tgt] < i and sum("v_mfma" in b for b in raw[labels[tgt]:i]) >= 32:
            loops.append((labels[tgt], i))
    copies = 0
    for a, b in loops:
        if any((c, d) != (a, b) and a <= c and d <= b for c, d in loops):
            continue                                  # (an enclosing region: only the innermost MFMA loops count)
        pat = r"\s*v_accvgpr_(write|mov)" if os.path.basename(path).startswith(EPILOGUE_IN_LOOP) else r"\s*v_accvgpr_(read|write|mov)"
        copies += sum(bool(re.match(pat, x)) for x in raw[a:b])
    print("%s: %d MFMAs, %d VALU -> MFMA hazards, %d accumulator copies inside MFMA loops" % (os.path.basename(path), nmfma, hazards, copies))
    return nmfma, hazards, copies


def build(tmp):
    out = []
    for f in FILES:
        s = os.path.join(tmp, f + ".s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--cuda-device-only", "-S",
                               "-o", s, os.path.join(ROOT, "scm_gan_amd", "csrc", f)], stderr=subprocess.DEVNULL)
        out.append(s)
    return out


def main(argv):
    if argv[:1] == ["--build"]:
        with tempfile.TemporaryDirectory() as tmp:
            res = [check(p) for p in build(tmp)]
    else:
        res = [check(p) for p in argv]
    bad = any(h or c or not n for n, h, c in res)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))


# ---------------------------------------------------------------- inline-asm LDS stores in front of a barrier
LDS_ASM_FILES = ("conv3x3_wgrad_winoz.hip",)


def check_lds_stores_before_barriers(path):
    """conv3x3_wgrad_winoz.hip issues its operand-image stores as inline `ds_write_b64`: the compiler does not count them,
    so it inserts no `s_waitcnt lgkmcnt(0)` for them in front of an `s_barrier`.  Walks every straight-line stretch that
    ends in a barrier: after the last ds_write there must be an lgkmcnt(0) wait before the barrier.  Returns
    (barriers seen, barriers with an unwaited store in front)."""
    raw = open(path).read().split("\n")
    lines = [l.strip() for l in raw if l.strip() and not l.strip().startswith((";", ".")) or l.strip().startswith(".LBB")]
    seen = bad = 0
    for i, l in enumerate(lines):
        if not l.startswith("s_barrier"):
            continue
        seen += 1
        j = i - 1
        waited = False
        while j >= 0:
            p = lines[j]
            if p.startswith(".LBB") or p.startswith("s_barrier") or p.startswith("s_endpgm") or p.startswith("s_cbranch") \
                    or p.startswith("s_branch"):
                break
            if p.startswith("s_waitcnt") and ("lgkmcnt(0)" in p or p.split()[-1] == "0"):
                waited = True
            if p.startswith("ds_write") and not waited:
                bad += 1
                break
            if p.startswith("ds_write"):
                break
            j -= 1
    return seen, bad


def build_lds(outdir):
    outs = []
    for f in LDS_ASM_FILES:
        out = os.path.join(outdir, f + ".s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--cuda-device-only", "-S",
                               "-w", "-I", os.path.join(ROOT, "scm_gan_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "scm_gan_amd", "csrc", f), "-o", out])
        outs.append(out)
    return outs
