#!/usr/bin/env python3
"""Per-kernel averages of the memory-pipeline counters collected by the rocprofv3 --pmc passes below (diagnostic)."""
import collections, csv, glob, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1] + "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if "edge_kernel" in k or "atom_kernel<true, 0>" in k:
            acc[k][row["Counter_Name"]].append((int(row.get("Grid_Size", 0) or 0), float(row["Counter_Value"])))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        gmax = max(g for g, _ in v)
        v = [x for g, x in v if g >= 0.5 * gmax]  # the fused (large) launches only
        print("   %-40s %14.1f  (n=%d, grid >= %d)" % (c, sum(v) / len(v), len(v), gmax // 2))
