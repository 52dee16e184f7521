"""Condense rocprofv3 output directories (kernel stats + PMC csv) into a short text summary for profiles/."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
def find(pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))
for f in find("*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, root))
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i < 8: print("  ", ", ".join(row))
for f in find("*counter_collection.csv"):
    agg = defaultdict(lambda: defaultdict(list))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            agg[row.get("Kernel_Name", "?")[:60]][row.get("Counter_Name", "?")].append(float(row.get("Counter_Value", 0)))
    print("== counters:", os.path.relpath(f, root))
    for k, cs in agg.items():
        for c, v in sorted(cs.items()):
            print(f"   {k:60s} {c:24s} n={len(v):5d} mean={sum(v)/len(v):.4g} sum={sum(v):.4g}")
