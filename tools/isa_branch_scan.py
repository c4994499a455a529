#!/usr/bin/env python3
"""Branches inside the hot region of every kernel (hipcc -save-temps .s files): between a kernel's first and last MFMA (or, for a
kernel without MFMAs, over the whole kernel) count scalar branches and exec-mask branches (s_and_saveexec) per MFMA / per 100
instructions.  hipcc turns a run-time test around ONE element of an unrolled loop into a branch per element (and a select whose
one side is expensive into an exec-mask branch): round 5 found 41 such branches per sub-tile in the attention backward.
usage: tools/isa_branch_scan.py file.s [file.s ...]"""
import re, sys
for path in sys.argv[1:]:
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for st in starts:
        name = lines[st].split(":")[0]
        end = next((i for i in range(st, len(lines)) if ".end_amdhsa_kernel" in lines[i]), len(lines))
        body = [(i, l.strip()) for i, l in enumerate(lines[st:end]) if l.strip() and not l.strip().startswith((";", "."))]
        mf = [k for k, (_, l) in enumerate(body) if l.startswith("v_mfma")]
        lo, hi = (mf[0], mf[-1]) if mf else (0, len(body) - 1)
        hot = [l for _, l in body[lo:hi + 1]]
        br = sum(l.startswith("s_cbranch") for l in hot)
        sx = sum("saveexec" in l for l in hot)
        print(f"{br:4d} br {sx:4d} saveexec {len(mf):4d} mfma {len(hot):6d} instr in the hot region   {name[:100]}")
