#!/usr/bin/env python3
"""Instruction statistics of the hottest loop of a kernel in hipcc's assembly output.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip -S --cuda-device-only -Iscm_gan_amd/csrc -Iinclude <file.hip> -o k.s
  python tools/isa_loop_stats.py k.s <substring of the mangled kernel name> [--dump]

Prints, for the largest backward-branch span of the kernel (its K loop), the instruction count per opcode and per
class (MFMA / VALU / packed VALU / v_mov + accvgpr copies / LDS / global / scalar / waitcnt), i.e. how many issue slots
sit between the MFMAs: on gfx950 an f32 MFMA wave gets no free VALU slots, so every one of them is paid for
(DESIGN.md 4f / 4g)."""
import collections
import re
import sys


def kernel_body(lines, sub):
    starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and sub in l.split(':')[0]]
    if not starts:
        raise SystemExit("no kernel label containing %r" % sub)
    s = starts[0]
    e = next(i for i in range(s, len(lines)) if lines[i].strip().startswith('s_endpgm'))
    return lines[s].split(':')[0], lines[s:e + 1]


def hottest_loop(body):
    """The K loop: of all backward-branch spans, the shortest one that holds the most MFMAs (out-of-line blocks placed
    behind the loop branch back into it: their spans are longer and hold no more MFMAs)."""
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    mf = [0]
    for l in body:
        mf.append(mf[-1] + (1 if l.strip().startswith('v_mfma') else 0))
    best = None
    for i, l in enumerate(body):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            j = labels[m.group(1)]
            key = (-(mf[i + 1] - mf[j]), i - j)
            if best is None or key < best[0]:
                best = (key, j, i)
    if best is None:
        raise SystemExit("no loop found")
    return body[best[1]:best[2] + 1]


def classify(op):
    if op.startswith('v_mfma'):
        return 'mfma'
    if op.startswith('v_accvgpr') or op in ('v_mov_b32', 'v_mov_b64', 'v_mov_b32_dpp', 'v_mov_b32_e32', 'v_mov_b32_e64'):
        return 'mov/copy'
    if op.startswith('v_pk_'):
        return 'valu packed'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_') or op.startswith('scratch_'):
        return 'vmem' if not op.startswith('scratch_') else 'scratch (spill)'
    if op.startswith('s_waitcnt') or op.startswith('s_nop') or op.startswith('s_barrier'):
        return 'wait/nop/barrier'
    if op.startswith('s_'):
        return 'scalar'
    return 'other'


def main():
    path, sub = sys.argv[1], sys.argv[2]
    lines = open(path).read().split('\n')
    name, body = kernel_body(lines, sub)
    loop = hottest_loop(body)
    ops = collections.Counter()
    for l in loop:
        t = l.strip().split()
        if not t or t[0][0] in '.;' or t[0].endswith(':'):
            continue
        ops[t[0]] += 1
    cls = collections.Counter()
    for k, v in ops.items():
        cls[classify(k)] += v
    total = sum(ops.values())
    print("%s\nhottest loop: %d instructions, %d MFMA, %d others (%.2f per MFMA)" % (
        name, total, cls['mfma'], total - cls['mfma'], (total - cls['mfma']) / max(1, cls['mfma'])))
    for k, v in cls.most_common():
        print("  %-18s %5d" % (k, v))
    print("opcodes:")
    for k, v in ops.most_common(40):
        print("  %5d %s" % (v, k))
    if '--dump' in sys.argv:
        print('\n'.join(loop))


if __name__ == '__main__':
    main()
