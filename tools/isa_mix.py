"""Static instruction mix of a device function's loops (gfx950 assembly of sft_kernels.hip): how many MFMA / VALU / LDS / global / SALU
instructions the loop bodies hold and which VALU opcodes lead.  usage: python tools/isa_mix.py FUNCTION-SUBSTRING [asm file]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = sys.argv[1]
asm = sys.argv[2] if len(sys.argv) > 2 else None
if asm is None:
    asm = os.path.join(tempfile.gettempdir(), "sft_kernels_isa_mix.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                    os.path.join(ROOT, "defslam_amd", "csrc", "sft_kernels.hip"), "-o", asm], check=True, capture_output=True)
lines = open(asm).read().splitlines()
start = None
for i, l in enumerate(lines):
    if re.match(r"^_Z\w+:", l) and want in l:
        start = i
    if start is not None and i > start and ".Lfunc_end" in l:
        end = i
        break
def cat(op):
    return ("mfma" if "mfma" in op else "lds" if op.startswith("ds_") else "global" if op.startswith("global_") else "scratch" if op.startswith("scratch")
            else "salu" if op.startswith("s_") else "valu" if op.startswith("v_") else "other")
tot, inl, ops, depth = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter()
loop = False
for l in lines[start:end]:
    if re.match(r"^\.LBB\d+_\d+:", l) or l.startswith("; %bb."):
        loop = ("in Loop:" in l) or ("Loop Header" in l)
    m = re.match(r"^\s+([a-z_0-9]+)\s", l)
    if not m:
        continue
    c = cat(m.group(1))
    tot[c] += 1
    if loop:
        inl[c] += 1
        if c == "valu":
            ops[m.group(1)] += 1
print(lines[start][:100])
print(" whole function:", dict(tot))
print(" inside loops  :", dict(inl))
print(" VALU opcodes inside loops:", ops.most_common(16))
