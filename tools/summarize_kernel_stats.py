#!/usr/bin/env python3
"""Condense rocprofv3's *_kernel_stats.csv: demangled-ish short names, calls, total/avg ns, percentage."""
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("caco::", "")
    return name if len(name) <= 110 else name[:107] + "..."


rows = list(csv.DictReader(open(sys.argv[1])))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "percent"])
for r in sorted(rows, key=lambda r: -float(r.get("TotalDurationNs", 0) or 0)):
    w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], f'{float(r["AverageNs"]):.0f}', r["MinNs"], r["MaxNs"],
                f'{float(r["Percentage"]):.2f}'])
