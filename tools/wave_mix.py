#!/usr/bin/env python
"""Static instruction mix of the factor kernel's step loop (the assembly `tools/wave_audit.py --product --keep` leaves behind and names):
instructions between the first and the last MFMA of the kernel by class, per ring phase (the step body exists 8 times).
Usage: python tools/wave_audit.py --product --keep; python tools/wave_mix.py <that .s file> [kernel]"""
import collections
import os
import re
import sys
import tempfile

want = sys.argv[2] if len(sys.argv) > 2 else "sftb_factor_kernel"
txt = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(txt) if re.match(r"^_Z\S*" + re.escape(want) + r"\S*:", l))
end = next(i for i in range(start, len(txt)) if ".end_amdhsa_kernel" in txt[i])
ins = [l.strip() for l in txt[start:end] if l.strip() and l.strip()[0] not in ";." and not l.strip().endswith(":")]
mf = [i for i, t in enumerate(ins) if "v_mfma" in t]
body = ins[mf[0]:mf[-1] + 1]


def cls(t):
    op = t.split()[0]
    if "v_mfma" in op: return "mfma"
    if "accvgpr" in op: return "valu: accvgpr move"
    if op.startswith("v_"):
        if "dpp" in t or "row_" in t or "quad_perm" in t: return "valu: dpp"
        if re.match(r"v_(readlane|readfirstlane|writelane|permlane|swap)", op): return "valu: lane ops"
        if re.match(r"v_(rsq|rcp|sqrt|div_|frexp|ldexp|trig)", op): return "valu: f64 special"
        if op.endswith("_f64") or "_f64_" in op: return "valu: f64 arithmetic"
        if re.match(r"v_(mov|pk_mov)", op): return "valu: mov"
        if re.match(r"v_cndmask", op): return "valu: select"
        if re.match(r"v_cmp", op): return "valu: compare"
        return "valu: integer / address"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): return "vmem " + ("store" if "store" in op else "load")
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_"): return "salu / branch"
    return "other"


c = collections.Counter(cls(t) for t in body)
n = len(body)
print(f"{want}: {n} instructions between the first and the last MFMA, {c['mfma']} MFMAs; per ring phase (/8):")
for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
    print(f"  {k:28s} {v:6d}  {v / 8:8.1f}")
valu = sum(v for k, v in c.items() if k.startswith("valu"))
print(f"  VALU total {valu} = {valu / c['mfma']:.2f} per MFMA")
if "--ops" in sys.argv:
    oc = collections.Counter(t.split()[0] for t in body if cls(t).startswith("valu"))
    for k, v in oc.most_common(40):
        print(f"    {k:32s} {v:6d} {v / 8:8.1f}")
