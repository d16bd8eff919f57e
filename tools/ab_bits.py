"""Do two builds of the library give the same bytes on the streamed path?  python tools/ab_bits.py libA.so libB.so (each in a child process)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, hashlib
import numpy as np
ROOT = sys.argv[1]
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), os.path.join(ROOT, "oracle")]
import scann_oracle as so
from scann import _hip
from scann.models.scann_model import HipModel
h = hashlib.sha256()
for name in ("qm9", "mp2018"):
    cfg = so.default_config(name)
    w = so.init_weights(cfg, 1234, perturb=True)
    m = HipModel(cfg, w, device=0, infer=True)
    for n, seed in ((128, 3), (700, 4)):
        de, dn = so.synth_dataset(n, seed)
        pk = _hip.pack_inputs(so.pad_batch(de, dn, g_update=bool(cfg["model"]["g_update"]))[0])
        y, ga = m.engine.forward(pk)
        h.update(y.tobytes()); h.update(ga.tobytes())
print(h.hexdigest())
'''
out = []
for lib in sys.argv[1:]:
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=dict(os.environ, SCANN_HIP_LIB=os.path.abspath(lib)), capture_output=True, text=True)
    print(lib, r.stdout.strip(), r.stderr[-300:])
    out.append(r.stdout.strip())
print("same bytes" if len(set(out)) == 1 else "DIFFERENT")
