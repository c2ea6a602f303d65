#!/usr/bin/env python3
"""tools/trace_timeline.py <rocprofv3 output dir> [steps from the end] -- the kernels of the last steps of a traced
run in launch order with their durations and the idle gaps between them (kernel-trace CSV of rocprofv3)."""
import csv, glob, os, sys

d = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-last:]
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    if name.startswith("_Z"):
        import subprocess
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    name = name.replace("ssym::", "").split("(")[0][:70]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"gap {gap:9.1f} us   run {(e - s) / 1e3:9.1f} us   {name}")
    prev_end = e
