"""Summarise rocprofv3 --pmc counter_collection.csv files: mean per dispatch of
each counter for kernels whose name contains a substring."""
import collections
import csv
import glob
import sys

pat, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "ais_half")
for fn in sorted(glob.glob(pat, recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        if sub in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(fn)
    for k, v in sorted(agg.items()):
        print(f"  {k:32s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
