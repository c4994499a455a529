#!/usr/bin/env python3
"""Scan the gfx950 ISA of the kernels that carry inline-assembly memory instructions for a hazard hipcc cannot see.

A vector-memory instruction that reads an SGPR needs 5 wait states behind a VALU instruction that wrote it
(v_readfirstlane / v_readlane -- e.g. the reload of a spilled SGPR -- or a VOP3 compare / carry-out).  The compiler's hazard
recogniser inserts them for its own instructions but does not look inside inline assembly, so an `asm("buffer_load ... %soff")`
right behind such a write reads a stale register.  Found the hard way in round 4 (a deep-prefetch variant of gemm_h3 whose
scalar offsets were reloaded from a spill lane in front of the asm loads: wrong rows, fixed by `s_nop 4`, variant not kept).

A second hazard, which hipcc DOES know but exempts a case of: a `buffer_store_dwordx3 / x4` fetches its data
registers some cycles after it issues, so a VALU instruction that overwrites one of them needs two wait states behind the
store (tools/micro/store_war.hip: with none or one, 5 % of the stored dwords come out as the NEW value).  LLVM's hazard
recogniser (GCNHazardRecognizer::createsVALUHazard) inserts them -- except when the store's `soffset` is an SGPR, which it
takes to be exempt; gfx950 is not (round 5's head-image corruption: every damaged store had a scalar row offset).  `scan_store`
reports every wide buffer store whose data registers a VALU instruction writes fewer than two wait states later.

usage: python tools/isa_hazard_scan.py [file.hip ...]     (default: every csrc/*.hip)
exit code 1 and one line per finding if any."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WRITERS = (re.compile(r"\s+(v_readfirstlane_b32|v_readlane_b32)\s+(s\d+)"),
           re.compile(r"\s+(v_cmp\w*_e64|v_add_co_u32_e64|v_sub_co_u32_e64|v_subrev_co_u32_e64)\s+(s\[\d+:\d+\])"))


def sregs(text):
    out = set()
    for a, b, c in re.findall(r"s\[(\d+):(\d+)\]|\b(s\d+)\b", text):
        if c:
            out.add(c)
        else:
            out.update("s%d" % k for k in range(int(a), int(b) + 1))
    return out


def scan(lines):
    def real(t):
        return t and not t.startswith(";") and not t.startswith(".") and not t.endswith(":")
    found = []
    for i, l in enumerate(lines):
        m = WRITERS[0].match(l) or WRITERS[1].match(l)
        if not m:
            continue
        written = sregs(m.group(2))
        states, j, inasm = 0, i + 1, False
        while j < len(lines) and states < 5:
            t = lines[j].strip()
            if t.startswith(";;#ASMSTART"):
                inasm = True
            elif t.startswith(";;#ASMEND"):
                inasm = False
            elif real(t):
                if inasm and ("buffer_" in t or "global_" in t) and (sregs(t) & written):
                    found.append((i + 1, l.strip(), states, t))
                n = re.match(r"s_nop (\d+)", t)
                states += int(n.group(1)) + 1 if n else 1
            j += 1
    return found


def vregs(tok):
    """VGPRs named by one operand token: v12 or v[12:15] (anything else: none)"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


STORE = re.compile(r"\s*buffer_store_dwordx[34]\s+(v\[\d+:\d+\])")
NO_VGPR_DEST = ("v_cmp", "v_readlane", "v_readfirstlane", "v_nop")


def scan_store(lines, need: int = 2):
    """wide buffer stores whose data registers a VALU instruction writes fewer than `need` wait states behind them"""
    def real(t):
        return t and not t.startswith(";") and not t.startswith(".") and not t.endswith(":")
    found = []
    for i, l in enumerate(lines):
        m = STORE.match(l)
        if not m:
            continue
        data = vregs(m.group(1))
        states, j = 0, i + 1
        while j < len(lines) and states < need:
            t = lines[j].strip()
            if real(t):
                if t.startswith("s_cbranch") or t.startswith("s_branch") or t.startswith("s_endpgm") or t.startswith("s_setpc"):
                    break                                  # (a taken branch costs more than the hazard's two states)
                if t.startswith("v_") and not t.startswith(NO_VGPR_DEST):
                    dest = t.split(None, 1)[1].split(",")[0].strip() if " " in t else ""
                    if vregs(dest) & data:
                        found.append((i + 1, l.strip(), states, t))
                n = re.match(r"s_nop (\d+)", t)
                states += int(n.group(1)) + 1 if n else 1
            j += 1
    return found


def main(argv):
    files = argv or sorted(glob.glob(os.path.join(ROOT, "transformertts_amd", "csrc", "*.hip")))
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for f in files:
            out = os.path.join(tmp, os.path.basename(f) + ".s")
            procs.append((f, out, subprocess.Popen([os.environ.get("HIPCC", "hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S",
                                                    "--cuda-device-only", f, "-o", out], stderr=subprocess.DEVNULL)))
        for f, out, pr in procs:
            if pr.wait() != 0:
                print(f"{f}: hipcc failed"); bad += 1; continue
            isa = open(out).read().split("\n")
            hits = scan(isa) if "asm" in open(f).read() else []
            shits = scan_store(isa)
            print(f"{os.path.basename(f)}: {len(hits)} + {len(shits)} finding(s)")
            for ln, w, st, t in hits:
                print(f"  ISA line {ln}: `{w}` and {st} wait state(s) later, inside inline assembly: `{t}`")
            for ln, w, st, t in shits:
                print(f"  ISA line {ln}: `{w}` and {st} wait state(s) later its data is overwritten: `{t}`")
            bad += len(hits) + len(shits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
