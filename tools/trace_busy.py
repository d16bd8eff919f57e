#!/usr/bin/env python3
"""Device-busy fraction from a rocprofv3 kernel trace: union of the kernel intervals over the span from the first start to the last end
(optionally only the last `frac` of the span).  python3 tools/trace_busy.py <dir> [frac]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f)))
t0, t1 = iv[0][0], max(e for _, e in iv)
lo = t1 - (t1 - t0) * frac
busy, cur_s, cur_e, gaps = 0, None, None, []
for s, e in iv:
    if e <= lo:
        continue
    s = max(s, lo)
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = t1 - lo
gaps.sort()
print("last %.0f %% of the trace: span %.2f ms, device busy %.1f %%, %d idle gaps, sum %.2f ms, largest %s us"
      % (frac * 100, span / 1e6, 100.0 * busy / span, len(gaps), sum(gaps) / 1e6, [round(g / 1e3, 1) for g in gaps[-5:]]))
