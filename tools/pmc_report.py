#!/usr/bin/env python3
"""Per-kernel, per-launch averages of the rocprofv3 --pmc passes collected by tools/pmc_run.sh <tag> (separate passes: never
a tracing domain next to --pmc), written to profiles/<tag>_counters.txt; with --traffic also the HBM bytes of the dominant
kernel into profiles/edge_kernel.json under the given workload key (what bench.py reports as roofline.traffic).

  python tools/pmc_report.py r02b [--traffic qm9_g16]

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are KiB and on gfx950 FETCH_SIZE counts
half the bytes of 16-B-per-lane reads (MI355X_MICROARCH.md, HBM section); every global read of edge_kernel is such a read
except the 8-B-per-edge index loads."""
import collections
import csv
import glob
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "gpurun_out", "pmc_%s_*" % tag, "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][row["Counter_Name"]].append((int(row.get("Grid_Size", 0) or 0), float(row["Counter_Value"])))
lines = ["# rocprofv3 --pmc, separate passes (tools/pmc_run.sh %s); per-launch averages over the launches with grid >= 0.8 x the largest" % tag]
avg = collections.defaultdict(dict)
for k, d in sorted(acc.items()):
    lines.append(k)
    for c, v in sorted(d.items()):
        gmax = max(g for g, _ in v)
        vv = [x for g, x in v if g >= 0.8 * gmax]
        avg[k][c] = sum(vv) / len(vv)
        lines.append("   %-32s %16.1f  (n=%d, grid >= %d)" % (c, avg[k][c], len(vv), int(0.8 * gmax)))
open(os.path.join(root, "profiles", "%s_counters.txt" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
if "--traffic" in sys.argv:
    key = sys.argv[sys.argv.index("--traffic") + 1]
    # the plain g_update kernel on 64-row tiles (not the first layer's launch with the basis MLP fused in: `<true, 2, true>`)
    # (template arguments: GUPD, RT, FB, EX, KEEP, DEAD since round 5; earlier rounds had fewer)
    names = ("scann::edge_kernel<true, 2>", "scann::edge_kernel<true, 2, false>", "scann::edge_kernel<true, 2, false, false>",
             "scann::edge_kernel<true, 2, false, false, false, false>")
    ek = [v for k, v in avg.items() if k in names and "FETCH_SIZE" in v and "WRITE_SIZE" in v]
    if ek:
        path = os.path.join(root, "profiles", "edge_kernel.json")
        j = json.load(open(path))
        j.setdefault("traffic", {})[key] = {
            "hbm_bytes_per_launch": (2 * ek[0]["FETCH_SIZE"] + ek[0]["WRITE_SIZE"]) * 1024,
            "fetch_KiB_raw": ek[0]["FETCH_SIZE"], "write_KiB": ek[0]["WRITE_SIZE"],
            "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tools/pmc_run.sh %s); (2 x FETCH_SIZE + WRITE_SIZE) KiB" % tag}
        json.dump(j, open(path, "w"), indent=1)
