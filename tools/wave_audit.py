#!/usr/bin/env python
"""Audit of the hand-pinned accumulator tiles of sft_wave.h in the compiled ISA (hipcc does not model what is inside an asm statement):
  1. no compiler instruction touches an accumulator register in front of the last MFMA of the kernel (behind it -- the back substitution --
     the window is dead and the compiler may use the file as it likes);
  2. no scratch, no VGPR spills;
  3. no VALU instruction writes a register that an MFMA statement reads within the two issue slots in front of that statement;
  4. no instruction other than an MFMA accumulating into the same registers touches the VGPR result of an MFMA before the matrix pipe has
     written it (20 issue slots behind a 16-pass v_mfma_f64_16x16x4, 9 behind a 4-pass v_mfma_f64_4x4x4; an MFMA in between counts its passes,
     except the last one in front of a memory / LDS instruction, which issues in its shadow);
  5. no LDS / global LOAD lands in a VGPR that an MFMA issued within the last 8 issue slots reads as its A or B operand, or that an MFMA is
     still writing (memory instructions issue in the shadow of MFMAs).  The probe of r06 (tools/probes/mfma4x4_probe.hip, part 6) shows the
     matrix pipe has NO dependency tracking of its own -- a dependent 4 x 4 x 4 pair issues back to back and reads a stale accumulator -- so an
     MFMA starts when it issues and has read A / B a few slots later; 8 slots is a margin, not a measured bound;
  6. a v_mfma_f64_4x4x4 that accumulates into the result of another one has at least four issue slots between them (s_nop or another MFMA):
     the hardware does not hold a back-to-back dependent pair back long enough (tools/probes/mfma4x4_probe.hip).
Usage: python tools/wave_audit.py [kernel-name-substring]   (compiles defslam_amd/csrc/sft_kernels.hip for the device, -DDSH_LAB)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if a not in ("--product", "--keep")]
product = "--product" in sys.argv[1:]      # audit the code of libdefslam_hip.so (no -DDSH_LAB) instead of the lab build
want = args[0] if args else ("sftb_factor_kernel" if product else "sft_wave_solve_kernel")
extra = args[1:]
out = os.path.join(tempfile.gettempdir(), f"wave_audit_{os.getpid()}.s")     # (build() runs three audits side by side)
subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-w", "--offload-arch=gfx950"] + ([] if product else ["-DDSH_LAB"]) +
                      ["--cuda-device-only", "-S", os.path.join(ROOT, "defslam_amd", "csrc", "sft_kernels.hip"), "-o", out] + extra)
txt = open(out).read().split("\n")
if "--keep" in sys.argv[1:]:
    print("assembly kept:", out)
else:
    os.remove(out)
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


def allregs(t):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", t):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else {int(m.group(3))}
    return out


def passes(t):
    return 4 if "4x4x4" in t else 16


bad4 = []
for n, (i, a, t) in enumerate(stream):
    if "v_mfma" not in t:
        continue
    ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
    dst = regs(ops[0].split()[0])
    if not dst:
        continue      # accumulator-file destination: only MFMAs touch those (rule 1)
    need = 20 if passes(t) == 16 else 9
    slots, m = 0, n + 1
    while m < len(stream) and slots < need:
        j, a2, t2 = stream[m]
        if t2.startswith("s_nop"):
            slots += int(t2.split()[1]) + 1
        elif "v_mfma" in t2:
            o2 = [o.strip() for o in t2.split(None, 1)[1].split(",")]
            srcab = regs(o2[1].split()[0]) | regs(o2[2].split()[0])
            if srcab & dst:
                bad4.append((j, t, t2))
            d2 = regs(o2[0].split()[0])
            if d2 & dst and d2 != dst:
                bad4.append((j, t, t2))
            # the last MFMA in front of a memory instruction: that one issues in its shadow
            nxt = stream[m + 1][2] if m + 1 < len(stream) else ""
            slots += 1 if re.match(r"(ds_|global_|buffer_|flat_|scratch_)", nxt) else passes(t2)
        elif t2.startswith("s_") and not t2.startswith("s_nop"):
            if re.match(r"s_(cbranch|branch|setpc|swappc|endpgm)", t2):
                break     # control flow: the phase bodies end with a fence by construction (checked where they are entered linearly)
            slots += 1
        else:
            if allregs(t2) & dst:
                bad4.append((j, t, t2))
            slots += 1
        m += 1
bad5 = []
for n, (i, a, t) in enumerate(stream):
    if not re.match(r"(ds_read|ds_load|global_load|buffer_load|flat_load|scratch_load)", t) or "lds" in t.split()[0]:
        continue
    dst = regs(t.split(None, 1)[1].split(",")[0].strip())
    if not dst:
        continue
    slots, m = 0, n - 1
    while m >= 0 and slots < 20:
        j, a2, t2 = stream[m]
        if t2.startswith("s_nop"):
            slots += int(t2.split()[1]) + 1
        elif "v_mfma" in t2:
            o2 = [o.strip() for o in t2.split(None, 1)[1].split(",")]
            srcab = regs(o2[1].split()[0]) | regs(o2[2].split()[0])
            acc = regs(o2[0].split()[0]) | regs(o2[3].split()[0])
            # operands A / B: anywhere in the window (a queued MFMA reads them when its turn comes); the accumulator: while it is being written
            if ((srcab & dst) and slots < 8) or ((acc & dst) and slots < (20 if passes(t2) == 16 else 9)):
                bad5.append((i, t2, t))
            slots += passes(t2)
        elif re.match(r"s_(cbranch|branch|setpc|swappc)", t2):
            break
        else:
            slots += 1
        m -= 1
bad6 = []
for n, (i, a, t) in enumerate(stream):
    if "v_mfma" not in t or "4x4x4" not in t:
        continue
    o = [x.strip() for x in t.split(None, 1)[1].split(",")]
    acc = regs(o[3].split()[0])
    if not acc:
        continue
    slots, m = 0, n - 1
    while m >= 0 and slots < 4:
        j, a2, t2 = stream[m]
        if t2.startswith("s_nop"):
            slots += int(t2.split()[1]) + 1
        elif "v_mfma" in t2:
            if regs(t2.split(None, 1)[1].split(",")[0].strip()) & acc:
                bad6.append((i, t2, t))
                break
            slots += passes(t2)
        elif re.match(r"s_(cbranch|branch|setpc|swappc)", t2):
            break
        else:
            slots += 1
        m -= 1
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
print(f"  instructions that touch the VGPR result of an MFMA before it is written: {len(bad4)}")
for b in bad4[:6]:
    print("     ", b)
print(f"  loads into a VGPR that an MFMA issued within the last 8 slots reads as an operand (or still writes): {len(bad5)}")
for b in bad5[:8]:
    print("     ", b)
print(f"  dependent 4x4x4 MFMAs with fewer than four slots between them: {len(bad6)}")
for b in bad6[:4]:
    print("     ", b)
sys.exit(1 if (bad1 or bad3 or bad4 or bad5 or bad6) else 0)
