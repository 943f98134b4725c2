#!/usr/bin/env python3
"""Per-dispatch averages of a rocprofv3 --pmc counter_collection.csv, for kernels matching a substring.
    python tools/pmc_sum.py <dir-with-csv> [kernel-substring]"""
import collections, csv, glob, re, sys
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:30]
        if pat not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k, v in acc.items():
        n = len(disp[k]); print(k, "dispatches", n, {a: int(b / n) for a, b in v.items()})
