"""Where do the structure-resident kernels and the layer-streamed kernels first differ?  Per-layer centres / geometry / context of
both paths on one batch (SCANN_SR_DEBUG=1 makes the resident kernels keep them)."""
import os
import sys

import numpy as np

os.environ["SCANN_SR_DEBUG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), os.path.join(ROOT, "oracle")]
import scann_oracle as so  # noqa: E402
from scann import _hip  # noqa: E402
from scann.models.scann_model import HipModel  # noqa: E402

cfg = so.default_config("qm9")
w = so.init_weights(cfg, 1234, perturb=True)
L = cfg["model"]["n_attention"]
de, dn = so.synth_dataset(32, 5)
pk = _hip.pack_inputs(so.pad_batch(de, dn, g_update=True)[0])
a = HipModel(cfg, w, device=0, infer=True)
a.engine.set_debug(True)
ra = a.engine.upload(pk)
a.engine.forward_resident(ra)
ya = a.engine.download(ra)
b = HipModel(cfg, w, device=0, infer=True)
b.engine.set_resident_limit(6)
rb = b.engine.upload(pk)
b.engine.forward_resident(rb)
yb = b.engine.download(rb)
print(b.engine.batch_info(rb))
for l in range(L):
    for what, name, lay in ((0, "centres", l), (1, "geometry", l + 1), (2, "context", l + 1)):
        x, y = a.engine.debug_read(ra, what, lay), b.engine.debug_read(rb, what, lay)
        d = np.abs(x - y)
        print("layer %d %-8s equal %-5s max |d| %.3e at %s" % (l, name, np.array_equal(x, y), d.max(), np.unravel_index(d.argmax(), d.shape)))
print("y equal", np.array_equal(ya[0], yb[0]), "ga equal", np.array_equal(ya[1], yb[1]))
