"""Where the scratch (private memory) of the SfT kernels is touched: compiles sft_kernels.hip to gfx950 assembly and reports, per function,
the scratch instructions inside and outside loops (LLVM annotates every basic block of a loop with '; in Loop:' / 'Loop Header').
The solver phases are non-inlined functions of up to 256 VGPRs; the AMDGPU calling convention makes a callee save the callee-saved
registers it uses (v40-v47, v56-v63, ... ), once in its prologue and once in its epilogue -- that is what 'ScratchSize' of the kernels is.
usage: python tools/scratch_report.py [extra hipcc flags] > profiles/r02/scratch_report.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "defslam_amd", "csrc", "sft_kernels.hip")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "--offload-arch=gfx950", "-S", "--cuda-device-only", src, "-o", out] + sys.argv[1:]
    rem = subprocess.run(cmd + ["-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    lines = open(out).read().splitlines()
print("# hipcc", " ".join(cmd[1:-4]), "(ROCm 7.2, gfx950)")
print("# kernel resource usage remarks")
for ln in rem.stderr.splitlines():
    m = re.search(r"remark: +(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize|Occupancy|SGPRs Spill|VGPRs Spill|LDS Size)(.*?)\[-Rpass", ln)
    if m:
        print(("" if m.group(1) == "Function Name" else "    ") + m.group(1) + m.group(2).rstrip())
print()
print("# scratch instructions per function: total, inside loops, where the in-loop ones are")
fn, in_loop, stats = None, False, {}
for i, ln in enumerate(lines):
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        fn, in_loop = m.group(1), False
        stats[fn] = [0, 0, []]
        continue
    if fn is None:
        continue
    if re.match(r"^\.LBB\d+_\d+:", ln) or re.match(r"^; %bb\.", ln):
        in_loop = ("in Loop:" in ln) or ("Loop Header" in ln)
    if "scratch_load" in ln or "scratch_store" in ln:
        stats[fn][0] += 1
        if in_loop:
            stats[fn][1] += 1
            stats[fn][2].append(ln.strip())
for f, (tot, inl, where) in stats.items():
    name = subprocess.run(["c++filt", f], capture_output=True, text=True).stdout.strip() or f
    print(f"{name}\n    scratch instructions: {tot}, inside loops: {inl}")
    for w in where[:8]:
        print("       ", w)
