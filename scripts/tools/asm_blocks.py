#!/usr/bin/env python3
"""Basic blocks of one kernel's gfx950 assembly (hipcc -S): per block the number of vector, scalar, LDS and
vector-memory instructions, the branch targets, and which blocks close a loop (a branch to an earlier label).
Development aid for counting what a row loop issues per sample.

    python3 scripts/tools/asm_blocks.py kernel.s [--dump LABEL ...]
"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_"):
        if re.match(r"v_(fma|mul|add|sub|max|min|floor|fract|rndne|trunc|ceil|div|ldexp|frexp|cmp|cmpx|cndmask)\w*_f64", op) or op.endswith("_f64"):
            return "v64"
        if op.startswith("v_cvt"):
            return "vcvt"
        if re.match(r"v_(rcp|rsq|sqrt|sin|cos|exp|log)", op):
            return "vtrans"
        if re.match(r"v_(readlane|writelane|readfirstlane)", op):
            return "vlane"
        if re.match(r"v_(mov|accvgpr)", op):
            return "vmov"
        if re.match(r"v_(cmp|cndmask)", op):
            return "vcmp"
        if re.search(r"_(f32|f16)$", op) or re.match(r"v_(fma|mac|fmac|mad)_f32", op) or op in ("v_fmac_f32", "v_fma_f32"):
            return "vf32"
        return "vint"
    if op.startswith("s_"):
        return "s"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path = sys.argv[1]
    dump = set(sys.argv[3:]) if len(sys.argv) > 2 and sys.argv[2] == "--dump" else set()
    blocks, order, cur = {}, [], "entry"
    blocks[cur] = []
    order.append(cur)
    for line in open(path):
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
            continue
        s = line.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        blocks[cur].append(s)
    idx = {b: i for i, b in enumerate(order)}
    for b in order:
        c = Counter(classify(i.split()[0]) for i in blocks[b])
        targets = [i.split()[-1] for i in blocks[b] if i.startswith(("s_cbranch", "s_branch"))]
        back = [t for t in targets if t in idx and idx[t] <= idx[b]]
        nv = sum(v for k, v in c.items() if k.startswith("v") and k != "vmem")
        print(f"{b:14s} n={len(blocks[b]):4d} valu={nv:4d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())) +
              (f"  -> {','.join(targets)}" if targets else "") + (f"  LOOP->{back}" if back else ""))
        if b in dump:
            for i in blocks[b]:
                print("      ", i)


if __name__ == "__main__":
    main()
