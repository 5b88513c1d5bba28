"""Timeline of one API call from a rocprofv3 trace (kernel_trace.csv + memory_copy_trace.csv in a directory): for the last calls of the
run, when the device was busy with kernels / copies and where the gaps are.  usage: python tools/trace_gaps.py DIR [first-kernel-name-substring]"""
import csv
import glob
import sys

d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "swpb_cellid"
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-40:]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "copy"))))
ev.sort()
starts = [i for i, e in enumerate(ev) if first in e[2]]
if len(starts) < 3:
    print("no calls found"); sys.exit(0)
# one call = from the copies right before its first kernel to the event before the next call's first copies
a, b = starts[-2], starts[-1]
while a > 0 and ev[a - 1][2].startswith("C") : a -= 1
while b > 0 and ev[b - 1][2].startswith("C") : b -= 1
call = ev[a:b]
t0 = call[0][0]
busy_k = sum(e[1] - e[0] for e in call if e[2][0] == "K")
busy_c = sum(e[1] - e[0] for e in call if e[2][0] == "C")
print(f"call: {len(call)} events, span {(call[-1][1] - t0) / 1e3:.1f} us, kernels {busy_k / 1e3:.1f} us, copies {busy_c / 1e3:.1f} us; next call starts {(ev[b][0] - t0) / 1e3:.1f} us after this one")
prev = t0
for s, e, n in call:
    gap = (s - prev) / 1e3
    print(f"  +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  {n}")
    prev = max(prev, e)
