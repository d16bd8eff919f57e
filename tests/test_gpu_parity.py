"""GPU parity: the HIP path (through the C ABI) against the NumPy oracle on the same seeded inputs.

Tolerance: BASELINE.json north_star asks <= 1e-4 relative on fp32 outputs.  Elementwise checks use
|gpu - oracle| <= RTOL * max(|oracle|, scale) with scale = RMS of the oracle tensor, so values that
happen to sit near zero are judged against the tensor's own magnitude.
"""
import numpy as np
import pytest

import scann_oracle as so

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def rel_err(got, ref):
    ref = np.asarray(ref, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    scale = max(float(np.sqrt(np.mean(ref * ref))), 1e-30)
    return float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), scale)))


def make(cfg_name="qm9", n=24, seed=0, kind="qm9", perturb=True, **over):
    from scann.models.scann_model import HipModel

    cfg = so.default_config(cfg_name)
    cfg["model"].update(over.pop("model", {}))
    cfg["hyper"].update(over.pop("hyper", {}))
    w = so.init_weights(cfg, 1234, perturb=perturb)
    de, dn = so.synth_dataset(n, seed, kind)
    inputs, _ = so.pad_batch(de, dn, g_update=cfg["model"]["g_update"])
    model = HipModel(cfg, w, device=0, infer=True)
    return cfg, w, inputs, model


def test_forward_matches_oracle_qm9(hip_lib):
    cfg, w, inputs, model = make()
    y, ga = model.predict(inputs)
    y_ref, ga_ref = so.forward(cfg, w, inputs, np.float32)
    assert y.shape == y_ref.shape and ga.shape == ga_ref.shape
    assert rel_err(y, y_ref) <= RTOL
    assert rel_err(ga, ga_ref) <= RTOL


def test_every_layer_intermediate(hip_lib):
    """centres / geometry / context after every LocalAttention iteration (SURVEY.md 8c pin (1))."""
    from scann import _hip

    cfg, w, inputs, model = make(n=12)
    inter = {}
    so.forward(cfg, w, inputs, np.float32, intermediates=inter)
    pk = _hip.pack_inputs(inputs)
    amask = inputs["atom_mask"][..., 0]
    emask = inputs["neighbor_mask"] & amask[:, :, None]
    eng = model.engine
    eng.set_debug(True)
    rb = eng.upload(pk)
    eng.forward_resident(rb, 0)
    eng.sync()
    L = cfg["model"]["n_attention"]
    worst = {}
    for l in range(L + 1):
        worst["centers_%d" % l] = rel_err(eng.debug_read(rb, 0, l), inter["centers_%d" % l][amask])
        worst["geometry_%d" % l] = rel_err(eng.debug_read(rb, 1, l), inter["geometry_%d" % l][emask])
        if l >= 1:
            worst["context_%d" % l] = rel_err(eng.debug_read(rb, 2, l), inter["context_%d" % l][amask])
    eng.set_debug(False)
    rb.free()
    bad = {k: v for k, v in worst.items() if not v <= RTOL}
    assert not bad, bad
