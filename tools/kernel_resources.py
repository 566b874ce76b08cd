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
files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
print("%-100s %5s %5s %7s %7s %6s  %s" % ("kernel", "vgpr", "agpr", "lds B", "scratch", "spills", "waves/SIMD (registers), workgroups/CU (LDS)"))
for f in files:
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-S",
                        "--cuda-device-only", os.path.join(CSRC, f), "-o", out], check=True, capture_output=True)
        text = open(out).read()
    meta = text[text.find("amdhsa.kernels:"):]
    for blk in re.split(r"\n  - \.agpr_count:", meta)[1:]:
        g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, blk).group(1)) if re.search(r"\.%s:\s+(\d+)" % key, blk) else 0
        agpr = int(re.match(r"\s+(\d+)", blk).group(1))
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        if filt:
            name = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name)
        vgpr, lds, scratch, spills, wg = g("vgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("vgpr_spill_count"), g("max_flat_workgroup_size")
        regs = (vgpr + agpr + 7) // 8 * 8
        waves = min(8, 512 // max(regs, 8))
        by_lds = (160 * 1024 // lds) if lds else 99
        print("%-100s %5d %5d %7d %7d %6d  %d, %s" % ((f + ": " + name)[:100], vgpr, agpr, lds, scratch, spills, waves, by_lds if lds else "-"))
