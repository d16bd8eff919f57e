#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs (one pass per counter) into per-kernel per-launch averages and
write profiles/<tag>_pmc_summary.json + profiles/edge_kernel_traffic.json (read by bench.py for roofline.traffic).
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are in KiB and on gfx950 FETCH_SIZE
counts half the bytes of 16-B-per-lane reads (MI355X_MICROARCH.md, HBM section); every global read of edge_kernel is a
16-B-per-lane (float4) read except the 8-B-per-edge index loads."""
import collections, csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
group = int(sys.argv[2]) if len(sys.argv) > 2 else 4
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(root, "gpurun_out", "pmc_" + c, "*", "*_counter_collection.csv")):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == c:
                acc[row["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            res[k][c + "_KiB_avg"] = sum(v) / len(v)
            res[k][c + "_launches"] = len(v)
for k, v in res.items():
    if "FETCH_SIZE_KiB_avg" in v and "WRITE_SIZE_KiB_avg" in v:
        v["hbm_bytes_per_launch_corrected"] = (2 * v["FETCH_SIZE_KiB_avg"] + v["WRITE_SIZE_KiB_avg"]) * 1024
json.dump(res, open(os.path.join(root, "profiles", tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
ek = [v for k, v in res.items() if k.startswith("scann::edge_kernel")]
if ek:
    json.dump({"hbm_bytes_per_launch": ek[0]["hbm_bytes_per_launch_corrected"], "batches_per_launch": group,
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 64 --pool 16 (default --group), batch 128; "
                         "(2*FETCH_SIZE + WRITE_SIZE) KiB per launch, gfx950 FETCH_SIZE x2 correction for 16-B/lane reads",
               "fetch_KiB_raw": ek[0]["FETCH_SIZE_KiB_avg"], "write_KiB": ek[0]["WRITE_SIZE_KiB_avg"]},
              open(os.path.join(root, "profiles", "edge_kernel_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
