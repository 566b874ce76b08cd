#!/usr/bin/env python3
"""Register / LDS / scratch census of every kernel of csrc/*.hip as hipcc compiles it for gfx950 (build container, no GPU): the
numbers that bound occupancy — vector registers (512 per SIMD lane, shared by its waves), LDS (160 KB per CU) and scratch (spills)
— read from the code object metadata.  A kernel whose RARE path inflates the register count pays for it in every wave (round 4:
mean_distance_chunks held 191 registers = two waves per SIMD because of a branch one piece in 150 000 takes).
Usage: python tools/kernel_resources.py [file.hip ...]          (default: every .hip of platymatch_amd/csrc)"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "platymatch_amd", "csrc")
sys.path.insert(0, ROOT)
from platymatch_amd import build as B  # noqa: E402


def census(files=None):
    """-> list of dicts {file, name (demangled where c++filt exists, arguments cut), vgpr (unified total), agpr (its accumulation part), lds, scratch, spills, waves, workgroups}"""
    files = files or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    rows = []
    for f in files:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            # the compiler, target and flags of the product's own build (platymatch_amd/build.py)
            subprocess.run([B._hipcc()] + [x for x in B.FLAGS if x != "-fPIC"] + ["-S", "--cuda-device-only", os.path.join(CSRC, f), "-o", out],
                           check=True, capture_output=True)
            text = open(out).read()
        meta = text[text.find("amdhsa.kernels:"):]
        for blk in re.split(r"\n  - \.agpr_count:", meta)[1:]:
            def g(key):
                m = re.search(r"\.%s:\s+(\d+)" % key, blk)
                return int(m.group(1)) if m else 0
            agpr = int(re.match(r"\s+(\d+)", blk).group(1))
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            if filt:
                name = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(.*", "", name)
            vgpr, lds = g("vgpr_count"), g("group_segment_fixed_size")
            # gfx90a and later have ONE register file per lane: .vgpr_count is the unified total and already contains the
            # accumulation registers .agpr_count reports (adding them counted those twice)
            regs = (vgpr + 7) // 8 * 8
            rows.append(dict(file=f, name=name, vgpr=vgpr, agpr=agpr, lds=lds, scratch=g("private_segment_fixed_size"), spills=g("vgpr_spill_count"),
                             waves=min(8, 512 // max(regs, 8)), workgroups=(160 * 1024 // lds) if lds else None))
    return rows


if __name__ == "__main__":
    print("%-100s %5s %5s %7s %7s %6s  %s" % ("kernel", "vgpr", "agpr", "lds B", "scratch", "spills", "waves/SIMD (registers), workgroups/CU (LDS)"))
    for r in census(sys.argv[1:] or None):
        print("%-100s %5d %5d %7d %7d %6d  %d, %s" % ((r["file"] + ": " + r["name"])[:100], r["vgpr"], r["agpr"], r["lds"], r["scratch"], r["spills"],
                                                       r["waves"], r["workgroups"] if r["workgroups"] is not None else "-"))
