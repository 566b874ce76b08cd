#!/usr/bin/env python3
"""Instruction census of a kernel's gfx950 code (build container, no GPU): hipcc -S on a source of csrc/, the kernel's body cut out
by its mangled-name fragment, instructions counted by mnemonic (optionally between two labels: a loop).  Used for the statement
that the descriptor tile kernel's inner step has almost nothing left to pack (profiles/r04_sc_tile_isa_census.txt).
Usage: python tools/isa_census.py pm_shape_context.hip sc_tile_kernelILi4 [first_label last_label]
       python tools/isa_census.py pm_chi2.hip chi2_sym_kernelILi4ELi2ELin1ELi94ELb0 --blocks      (list the basic blocks: find the loops)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from platymatch_amd import build as B  # noqa: E402
argv = [a for a in sys.argv[1:] if a != "--blocks"]
src, frag = argv[0], argv[1]
lo, hi = (argv[2], argv[3]) if len(argv) > 3 else (None, None)
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    # the compiler, target and flags of the product's own build (platymatch_amd/build.py)
    subprocess.run([B._hipcc()] + [x for x in B.FLAGS if x != "-fPIC"] + ["-S", "--cuda-device-only", os.path.join(B.CSRC, src), "-o", out],
                   check=True, capture_output=True)
    lines = open(out).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(frag), l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end + 1]
if "--blocks" in sys.argv:
    # one line per basic block: where the kernel's loops are (a loop body is a block that branches back to its own label)
    print("%s: %s, basic blocks (instructions, float64 VALU, LDS, self-loop)" % (src, lines[start].split(":")[0][:60]))
    name, ins = "entry", []
    blocks = []
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append((name, ins))
            name, ins = m.group(1), []
        elif re.match(r"^\s+(v|s|ds|global|buffer|flat|scratch)_", l):
            ins.append(l.split())
    blocks.append((name, ins))
    for name, ins in blocks:
        if len(ins) < 16:
            continue
        f64 = sum(1 for x in ins if x[0].startswith("v_") and "f64" in x[0])
        lds = sum(1 for x in ins if x[0].startswith("ds_"))
        loop = any(x[0].startswith("s_cbranch") and x[-1] == name for x in ins)
        print("  %-10s %5d %5d %4d  %s" % (name, len(ins), f64, lds, "loop" if loop else ""))
    sys.exit(0)
if lo:
    a = next(i for i, l in enumerate(body) if l.startswith(lo + ":"))
    b = next(i for i, l in enumerate(body) if l.startswith(hi + ":"))
    body = body[a:b]
count = collections.Counter()
for l in body:
    m = re.match(r"^\s+((?:v|s|ds|global|buffer|flat|scratch)_\w+)", l)
    if m:
        count[re.sub(r"_e32$|_e64$", "", m.group(1))] += 1
kinds = collections.Counter()
for k, c in count.items():
    kind = ("packed float32 (v_pk_*)" if k.startswith("v_pk_") else "float64 VALU" if k.startswith("v_") and "f64" in k else
            "float32 arithmetic (mul / fma / add / min)" if re.match(r"v_(mul|fma|fmac|add|sub|min|min3|max)_f32", k) else
            "compare / select / integer VALU" if k.startswith("v_") else "wait states (s_nop)" if k == "s_nop" else
            "scalar" if k.startswith("s_") else "LDS" if k.startswith("ds_") else "memory")
    kinds[kind] += c
print("%s: %s, %d instructions%s" % (src, lines[start].split(":")[0][:60], sum(count.values()), " between %s and %s" % (lo, hi) if lo else ""))
for kind, c in kinds.most_common():
    print("  %-46s %4d" % (kind, c))
print("  by mnemonic: " + ", ".join("%s x%d" % kc for kc in count.most_common(40)))
