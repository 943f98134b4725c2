#!/usr/bin/env python3
"""Print the kernel timeline of the last voting item from a rocprofv3 --kernel-trace CSV directory."""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_seam_count" in r["Kernel_Name"]][-1]
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    n = r["Kernel_Name"]
    n = n[n.find("k_"):][:34] if "k_" in n else n[:34]
    print("%-34s start %8.1f us dur %8.1f us" % (n, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
