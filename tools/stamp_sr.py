"""Phase clocks of the structure-resident kernel (diagnostic build: make -C scann--material_amd/csrc stamps).
SCANN_HIP_LIB=.../libscann_hip_stamps.so python tools/stamp_sr.py [n_batches]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), os.path.join(ROOT, "oracle")]
import scann_oracle as so  # noqa: E402
from scann import _hip  # noqa: E402
from scann.models.scann_model import HipModel  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = so.default_config("qm9")
w = so.init_weights(cfg, 1234, perturb=True)
m = HipModel(cfg, w, device=0, infer=True)
m.engine.set_resident_limit(6)
parts = []
for i in range(nb):
    de, dn = so.synth_dataset(128, 100 + i)
    parts.append(_hip.pack_inputs(so.pad_batch(de, dn, g_update=True)[0]))
pk = _hip.concat_packed(parts) if nb > 1 else parts[0]
rb = m.engine.upload(pk)
for _ in range(3):
    m.engine.forward_resident(rb)
m.engine.sync()
info = m.engine.batch_info(rb)
st = m.engine.debug_stamps(rb, max_tiles=4 * info["resident_small"]).reshape(-1, 64).astype(np.int64)
plan = _hip.plan_groups(pk)
nt = plan["small"][:, 3]
print(info)
names = {0: "layer start", 1: "atom phase end", 15: "phase A end", 16: "c/q copied", 35: "layer end"}
for k in (1, 2, 3):
    sel = st[nt == k]
    if not len(sel):
        continue
    d = sel - sel[:, :1]
    mean = d.mean(axis=0)
    print("groups with %d tiles: %d; layer-2 total %.0f cycles" % (k, len(sel), mean[35]))
    print("  atom phase %.0f" % mean[1])
    for p in range(k):
        a = mean[3 + 4 * p: 7 + 4 * p]
        prev = mean[1] if p == 0 else mean[6 + 4 * (p - 1)]
        print("  A tile %d: head %.0f | planes+barrier %.0f | mfma+epilogue %.0f | barrier+LN %.0f" % (p, a[0] - prev, a[1] - a[0], a[2] - a[1], a[3] - a[2]))
    print("  A end -> Wk issue %.0f ; c/q copy + barrier %.0f" % (mean[15] - mean[6 + 4 * (k - 1)], mean[16] - mean[15]))
    for p in range(k):
        b = mean[17 + 6 * p: 23 + 6 * p]
        prev = mean[16] if p == 0 else mean[22 + 6 * (p - 1)]
        print("  B tile %d: head %.0f | gate planes+barrier %.0f | mfma+logits %.0f | barrier+K+barrier %.0f | softmax+LN %.0f | barrier %.0f"
              % (p, b[0] - prev, b[1] - b[0], b[2] - b[1], b[3] - b[2], b[4] - b[3], b[5] - b[4]))
