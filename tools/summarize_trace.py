"""Average duration of the kernels whose name contains FILT in a rocprofv3 kernel_trace.csv.
   python tools/summarize_trace.py <kernel_trace.csv> <substring>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
if d:
    print(f"duration_us,{sys.argv[2]},n={len(d)},avg={sum(d)/len(d):.1f},min={min(d):.1f},max={max(d):.1f}")
