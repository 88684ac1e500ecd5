#!/usr/bin/env python3
"""rocprofv3 counter_collection.csv -> per-kernel mean of every counter (one row per kernel), followed by the minimum and
the maximum over the dispatches (rows "<kernel> [min]" / "[max]"): a pass whose values are digit-identical to an older
profile's although the kernel changed (round-2 verdict, weak 9) shows as min = max = mean of another run."""
import csv
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(anonymous namespace\)::|caco::|^void ", "", r["Kernel_Name"])[:70]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "dispatches"] + names)
for k, cs in acc.items():
    n = max(len(v) for v in cs.values())
    w.writerow([k, n] + [f"{sum(cs[c]) / len(cs[c]):.4g}" if c in cs else "" for c in names])
    w.writerow([k + " [min]", n] + [f"{min(cs[c]):.6g}" if c in cs else "" for c in names])
    w.writerow([k + " [max]", n] + [f"{max(cs[c]):.6g}" if c in cs else "" for c in names])
