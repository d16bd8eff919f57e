#!/usr/bin/env python3
"""Diagnostic: where the workgroup slots of one edge_kernel launch are idle.  In-kernel stamps (s_memtime at entry / exit, HW_ID,
XCC_ID) of every workgroup of one layer's launch -> per-CU timelines: workgroups in flight, the gap between a workgroup's end and the
start of the next one on its CU, per-XCD spans.
Run with SCANN_HIP_LIB=scann--material_amd/lib/libscann_hip_stamps.so (make -C scann--material_amd/csrc stamps) [SCANN_STAMP_LAYER=l]
  python3 tools/stamp_occ.py [structures=1280] [warm forwards=5]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scann--material_amd")); sys.path.insert(0, ROOT)
from scann.models.scann_model import HipModel, normalize_config
import bench

cfg = normalize_config({"model": dict(bench.QM9_MODEL), "hyper": {"target": "homo"}})
model = HipModel(cfg, device=0, seed=1234, infer=True)
eng = model.engine
rng = np.random.default_rng(0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
rb = eng.upload(bench.synth_packed_batch(rng, nb))
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):
    eng.forward_resident(rb, 0)
eng.sync()
st = eng.debug_stamps(rb).astype(np.int64)
t0, t1, hw = st[:, 0], st[:, 7], st[:, 14]
print("records %d: entry stamp missing %d, exit stamp missing %d, exit <= entry %d" % (len(t0), (t0 == 0).sum(), (t1 == 0).sum(), ((t1 <= t0) & (t1 != 0)).sum()))
ok = (t0 > 0) & (t1 > t0)
t0, t1, hw = t0[ok], t1[ok], hw[ok]
xcc = (hw >> 32) & 15
hwid = hw & 0xffffffff
cu, sh, se = (hwid >> 8) & 15, (hwid >> 12) & 1, (hwid >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
ticks = (st[ok, 13] - st[ok, 12]).astype(np.float64)
mhz = (t1 - t0).sum() / ticks[ticks > 0].sum() * 100.0 if (ticks > 0).any() else float("nan")
print("workgroups %d  CUs seen %d  XCDs %d  shader clock %.0f MHz" % (len(t0), len(np.unique(key)), len(np.unique(xcc)), mhz))
dur = t1 - t0
print("workgroup lifetime: mean %.0f  median %.0f  p95 %.0f  max %.0f cycles" % (dur.mean(), np.median(dur), np.percentile(dur, 95), dur.max()))
# (s_memtime counters are not one clock across the chip -- not even inside an XCD: everything below stays inside one CU)
gaps, conc, spans, per_cu, lead, tail = [], [], [], [], [], []
for k in np.unique(key):
    c = key == k
    s, e = np.sort(t0[c]), np.sort(t1[c])
    per_cu.append(len(s))
    span = e[-1] - s[0]
    spans.append(span)
    conc.append(dur[c].sum() / float(span))
    # three slots per CU: the i-th end frees the slot that the (i + 3)-th start takes
    for i in range(len(s) - 3):
        gaps.append(s[i + 3] - e[i])
    if len(s) >= 3:
        lead.append(s[2] - s[0])          # how long the CU takes to get its first three workgroups
        tail.append(e[-1] - e[-3])        # ... and how long it runs with fewer than three at the end
spans = np.array(spans); gaps = np.array(gaps); conc = np.array(conc)
print("per-CU span (first start -> last end): mean %.0f  min %.0f  max %.0f cycles (%.1f / %.1f / %.1f us at the measured clock)"
      % (spans.mean(), spans.min(), spans.max(), spans.mean() / mhz, spans.min() / mhz, spans.max() / mhz))
print("workgroups per CU: mean %.1f  min %d  max %d" % (np.mean(per_cu), np.min(per_cu), np.max(per_cu)))
print("workgroups in flight per CU over its own span: mean %.2f  min %.2f  max %.2f (3 = every slot busy)" % (conc.mean(), conc.min(), conc.max()))
print("first three workgroups of a CU start within: mean %.0f  max %.0f cycles;  last end - third-last end: mean %.0f  max %.0f cycles"
      % (np.mean(lead), np.max(lead), np.mean(tail), np.max(tail)))
if len(gaps):
    print("end of a workgroup -> start of the workgroup that takes its slot: mean %.0f  median %.0f  p95 %.0f  max %.0f cycles; negative %.1f %%"
          % (gaps.mean(), np.median(gaps), np.percentile(gaps, 95), gaps.max(), 100.0 * (gaps < 0).mean()))
for x in np.unique(xcc):
    m = xcc == x
    print("   XCD %d: %d workgroups on %d CUs, lifetime mean %.0f, per-CU span mean %.0f max %.0f" % (x, m.sum(), len(np.unique(key[m])), dur[m].mean(),
          np.mean([t1[m & (key == k)].max() - t0[m & (key == k)].min() for k in np.unique(key[m])]),
          np.max([t1[m & (key == k)].max() - t0[m & (key == k)].min() for k in np.unique(key[m])])))
