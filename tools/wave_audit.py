#!/usr/bin/env python
"""Audit of the hand-pinned accumulator tiles of sft_wave.h in the compiled ISA (hipcc does not model what is inside an asm statement):
  1. no compiler instruction touches an accumulator register in front of the last MFMA of the kernel (behind it -- the back substitution --
     the window is dead and the compiler may use the file as it likes);
  2. no scratch, no VGPR spills;
  3. no VALU instruction writes a register that an MFMA statement reads within the two issue slots in front of that statement.
Usage: python tools/wave_audit.py [kernel-name-substring]   (compiles defslam_amd/csrc/sft_kernels.hip for the device, -DDSH_LAB)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if a != "--product"]
product = "--product" in sys.argv[1:]      # audit the code of libdefslam_hip.so (no -DDSH_LAB) instead of the lab build
want = args[0] if args else ("sftb_factor_kernel" if product else "sft_wave_solve_kernel")
extra = args[1:]
out = os.path.join(tempfile.gettempdir(), "wave_audit.s")
subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-w", "--offload-arch=gfx950"] + ([] if product else ["-DDSH_LAB"]) +
                      ["--cuda-device-only", "-S", os.path.join(ROOT, "defslam_amd", "csrc", "sft_kernels.hip"), "-o", out] + extra)
txt = open(out).read().split("\n")
start = next(i for i, l in enumerate(txt) if re.match(r"^_Z\S*" + re.escape(want) + r"\S*:", l))
end = next(i for i in range(start, len(txt)) if ".end_amdhsa_kernel" in txt[i])
body = txt[start:end]


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


inasm, last_mfma, agpr_outside = False, -1, []
stream = []   # (index, in_asm, text)
for i, l in enumerate(body):
    t = l.strip()
    if t.startswith(";;#ASMSTART"):
        inasm = True
        continue
    if t.startswith(";;#ASMEND"):
        inasm = False
        continue
    if not t or t[0] in ";." or t.endswith(":"):
        continue
    stream.append((i, inasm, t))
    if "v_mfma" in t:
        last_mfma = len(stream) - 1
bad1 = [(i, t) for n, (i, a, t) in enumerate(stream) if n < last_mfma and not a and (re.search(r"\ba\[?\d", t) or "accvgpr" in t)]
bad3 = []
for n, (i, a, t) in enumerate(stream):
    if not (a and "v_mfma" in t):
        continue
    ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
    read = set()
    for o in ops[1:4]:
        read |= regs(o.split()[0])
    states, back = 0, 1
    while states < 2 and n - back >= 0:      # two wait states between a VALU write and the MFMA that reads it; s_nop N counts N + 1
        j, a2, t2 = stream[n - back]
        back += 1
        if t2.startswith("s_nop"):
            states += int(t2.split()[1]) + 1
            continue
        if t2.startswith("v_") and "v_mfma" not in t2:
            dst = t2.split(None, 1)[1].split(",")[0].strip()
            if regs(dst) & read:
                bad3.append((i, t2, t))
        states += 1
meta = {}
for l in txt[start:end + 60]:
    m = re.match(r"\s*\.amdhsa_(next_free_vgpr|accum_offset|private_segment_fixed_size|next_free_sgpr)\s+(\d+)", l)
    if m:
        meta[m.group(1)] = int(m.group(2))
scratch = sum(1 for _, a, t in stream if t.startswith("scratch_"))
in_loop_scratch = sum(1 for n, (_, a, t) in enumerate(stream) if t.startswith("scratch_") and n < last_mfma and any("v_mfma" in stream[j][2] for j in range(max(0, n - 400), n)))
nm = sum(1 for _, _, t in stream if "v_mfma" in t)
print(f"{want}: {len(stream)} instructions, {nm} MFMAs")
print("  kernel descriptor:", meta, f"; scratch instructions: {scratch} ({in_loop_scratch} between MFMAs)")
print(f"  compiler instructions on accumulator registers in front of the last MFMA: {len(bad1)}", bad1[:3])
print(f"  VALU writes within two slots in front of an MFMA statement that reads them: {len(bad3)}")
for b in bad3[:6]:
    print("     ", b)
sys.exit(1 if (bad1 or bad3) else 0)
