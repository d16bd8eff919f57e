#!/usr/bin/env python3
"""Timeline of ONE steady-state training step from a rocprofv3 kernel trace (tools/train_kstats.sh writes it):
    python3 tools/train_timeline.py gpurun_out/ks_<tag> [step index]
prints every kernel between two adam_kernel launches: start offset, duration, gap to the previous kernel's end on the same queue."""
import csv, glob, sys

f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("scann::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?")))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(adam) // 2
lo, hi = adam[k] + 1, adam[k + 1] + 1
t0 = rows[lo][0]
last_end = {}
queues = sorted({r[3] for r in rows[lo:hi]})
print("step %d: %d kernels, %.1f us from the first start to adam's end; queues %s" % (k, hi - lo, (rows[hi - 1][1] - t0) / 1e3, queues))
busy = {}
for s, e, n, q in rows[lo:hi]:
    gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
    print("%9.1f  %7.1f us  gap %6.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, queues.index(q), n[:60]))
    last_end[q] = e
    busy[q] = busy.get(q, 0) + (e - s) / 1e3
print("busy per queue:", {queues.index(q): round(v, 1) for q, v in busy.items()})
