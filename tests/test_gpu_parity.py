"""GPU parity: the HIP path (through the C ABI) against the NumPy oracle on the same seeded inputs.

Tolerance: BASELINE.json north_star asks <= 1e-4 relative on fp32 outputs.  Elementwise checks use
|gpu - oracle| <= RTOL * max(|oracle|, scale) with scale = RMS of the oracle tensor, so values that
happen to sit near zero are judged against the tensor's own magnitude.
"""
import os

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


def strict_rel_err(got, ref):
    """SURVEY.md 8(d) "Parity measurement": max_i |y_gpu - y_oracle32| / max(|y_oracle32|, 1e-6) -- every prediction against its
    OWN magnitude, however small (rel_err above judges values near zero against the tensor's RMS)."""
    ref = np.asarray(ref, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    return float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-6)))


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
    assert strict_rel_err(y, y_ref) <= RTOL, strict_rel_err(y, y_ref)  # the north-star bound in SURVEY 8(d)'s own form


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


@pytest.mark.parametrize("over", [
    dict(model=dict(g_update=False)),                      # base SCANN branch (attention.py:155)
    dict(model=dict(use_attn_norm=False)),                 # no ResidualNorm (scann_model.py:404-408)
    dict(model=dict(use_ga_norm=False)),                   # GlobalAttention(norm=False)
    dict(hyper=dict(target="e_b")),                        # mrelu head (scann_model.py:446)
    dict(model=dict(g_update=False, use_attn_norm=False, use_ga_norm=False)),
    dict(model=dict(n_attention=1)),
    dict(model=dict(n_attention=0)),
], ids=["base", "no_attn_norm", "no_ga_norm", "e_b", "base_plain", "L1", "L0"])
def test_branches(hip_lib, over):
    """Every architecture switch.  Some variants are ill-conditioned in fp32 (use_ga_norm False feeds raw pair
    sums, |a| >> 1, into a softmax), so the bound is max(1e-4, 2 x the fp32 oracle's own error) against the
    fp64 oracle -- the GPU must not be worse than the reference precision by more than that."""
    cfg, w, inputs, model = make(n=10, seed=3, **{k: dict(v) for k, v in over.items()})
    y, ga = model.predict(inputs)
    y32, ga32 = so.forward(cfg, w, inputs, np.float32)
    y64, ga64 = so.forward(cfg, w, inputs, np.float64)
    assert rel_err(y, y64) <= max(RTOL, 2 * rel_err(y32, y64))
    assert rel_err(ga, ga64) <= max(RTOL, 2 * rel_err(ga32, ga64))


@pytest.mark.parametrize("ring,cgcnn", [(True, False), (False, True), (True, True)], ids=["ring", "cgcnn", "ring+cgcnn"])
def test_ring_and_cgcnn_embeddings(hip_lib, ring, cgcnn):
    """scann_model.py:361-374: extra ring/aromatic embedding and the 92-d CGCNN feature variant."""
    from scann.models.scann_model import HipModel

    cfg = so.default_config("qm9")
    cfg["model"].update(n_attention=2, use_ring=ring, feature="cgcnn" if cgcnn else "atomic")
    w = so.init_weights(cfg, 99, perturb=True)
    de, dn = so.synth_dataset(7, 31, use_ring=ring)
    inputs, _ = so.pad_batch(de, dn, True, use_ring=ring)
    if cgcnn:
        table = np.random.default_rng(5).integers(0, 2, size=(101, 92)).astype("float32")  # stand-in for atomic_features
        inputs["atomic"] = table[inputs["atomic"]]
    model = HipModel(cfg, w, device=0, infer=True)
    y, ga = model.predict(inputs)
    y_ref, ga_ref = so.forward(cfg, w, inputs, np.float32)
    assert rel_err(y, y_ref) <= RTOL and rel_err(ga, ga_ref) <= RTOL


@pytest.mark.parametrize("over", [
    dict(local_dim=64, num_head=4, global_dim=96, dense_out=32),
    dict(local_dim=192, num_head=6, global_dim=160, dense_out=200, n_attention=3),
    dict(local_dim=48, num_head=48, global_dim=16, dense_out=8, g_update=False, use_attn_norm=False, use_ga_norm=False),
    dict(local_dim=32, num_head=1, global_dim=300, dense_out=128, use_ring=True, n_attention=2),
    dict(local_dim=128, num_head=16, global_dim=128, dense_out=128, n_attention=2),
    dict(local_dim=96, num_head=8, global_dim=64, dense_out=64, feature="cgcnn", n_attention=2),
], ids=["64x4", "192x6", "48x48_base_plain", "32x1_ring", "128x16", "96x8_cgcnn"])
def test_widths_other_than_128_and_8_heads(hip_lib, over):
    """scann_model.py:330-434 builds the graph for whatever local_dim / num_head / global_dim / dense_out the yaml holds (every
    shipped one: 128 / 8 / 128 / 128, the MFMA kernels).  Any other widths run the plain-fp32 forward of csrc/scann_generic.hip:
    same packed batch, same C ABI, fp32 products and sums; compared with the fp32 / fp64 restatements like the 128-wide kernels
    (QM9-shaped and worst-case molecules, a structure with an atom without neighbours, and -- when GlobalAttention normalises --
    the one-atom structure whose score is the reference's 0 / 0).  (Training such a handle: tests/test_gpu_training.py.)"""
    from scann import _hip
    from scann.models.scann_model import HipModel

    cfg = so.default_config("qm9")
    cfg["model"].update(over)
    cfg["model"]["n_atoms"] = 100
    ring, cg = bool(cfg["model"].get("use_ring")), cfg["model"].get("feature") == "cgcnn"
    w = so.init_weights(cfg, 123, perturb=True)
    de, dn = so.synth_dataset(9, 41, use_ring=ring)
    de2, dn2 = so.synth_dataset(2, 42, "worst", use_ring=ring)
    inputs, _ = so.pad_batch(list(de) + list(de2), list(dn) + list(dn2), True, use_ring=ring)
    if cg:
        table = np.random.default_rng(5).integers(0, 2, size=(101, 92)).astype("float32")
        inputs["atomic"] = table[inputs["atomic"]]
    model = HipModel(cfg, w, device=0, infer=True)
    y, ga = model.predict(inputs)
    y32, ga32 = so.forward(cfg, w, inputs, np.float32)
    y64, ga64 = so.forward(cfg, w, inputs, np.float64)
    assert rel_err(y, y64) <= max(RTOL, 2 * rel_err(y32, y64)), (rel_err(y, y64), rel_err(y32, y64))
    assert rel_err(ga, ga64) <= max(RTOL, 2 * rel_err(ga32, ga64))
    assert strict_rel_err(y, y32) <= 10 * RTOL
    # the same structures through the resident-batch entry points, twice (the per-batch workspace is reused), and split in two
    pk = _hip.pack_inputs(inputs)
    rb = model.engine.upload(pk)
    for slot in (0, 1):
        model.engine.forward_resident(rb, slot)
        y2, _ = model.engine.download(rb)
        assert np.array_equal(np.asarray(y2).ravel(), np.asarray(y).ravel())
    rb.free()
    half = {k: v[:5] for k, v in inputs.items()}
    yh, _ = model.predict(half)
    assert np.array_equal(np.asarray(yh).ravel(), np.asarray(y).ravel()[:5])  # a structure's result does not depend on its batch
    if not cg and not ring:  # packed edge cases: an atom without neighbours, a one-atom structure
        lone = _hip.PackedBatch([6, 1, 8, 1], [0, 1, 4], [0, 0, 1, 2, 2], [2, 1], [1.1, 1.3], [0.9, 1.7])
        rb = model.engine.upload(lone)
        model.engine.forward_resident(rb, 0)
        yl, gal = model.engine.download(rb)
        rb.free()
        if cfg["model"].get("use_ga_norm", True):
            assert np.isnan(yl[0]) and np.isfinite(yl[1])
        else:
            assert np.all(np.isfinite(yl))
        assert abs(float(np.sum(gal[1:])) - 1.0) < 1e-5


@pytest.mark.parametrize("over", [
    dict(model=dict(g_update=False)),
    dict(model=dict(use_attn_norm=False)),
    dict(model=dict(use_ga_norm=False)),
    dict(hyper=dict(target="e_b")),
    dict(model=dict(g_update=False, use_attn_norm=False, use_ga_norm=False)),
    dict(model=dict(n_attention=1)),
    dict(model=dict(n_attention=0)),
    dict(model=dict(use_ring=True)),
], ids=["base", "no_attn_norm", "no_ga_norm", "e_b", "base_plain", "L1", "L0", "ring"])
def test_every_branch_agrees_between_the_two_gpu_implementations(hip_lib, monkeypatch, over):
    """Every architecture switch of create_model (scann_model.py:362-447) through BOTH GPU implementations -- the MFMA kernels and the
    plain-fp32 forward (SCANN_GENERIC=1) -- on the same inputs: each within the fp32 restatement's own distance from the fp64
    restatement, as in test_branches."""
    from scann.models.scann_model import HipModel

    ring = bool(over.get("model", {}).get("use_ring"))
    cfg = so.default_config("qm9")
    for k, v in over.items():
        cfg[k].update(v)
    w = so.init_weights(cfg, 31, perturb=True)
    de, dn = so.synth_dataset(12, 7, use_ring=ring)
    inputs, _ = so.pad_batch(de, dn, cfg["model"].get("g_update", True), use_ring=ring)
    fast = HipModel(cfg, w, device=0, infer=True)
    monkeypatch.setenv("SCANN_GENERIC", "1")
    plain = HipModel(cfg, w, device=0, infer=True)
    monkeypatch.delenv("SCANN_GENERIC")
    y64, ga64 = so.forward(cfg, w, inputs, np.float64)
    y32, ga32 = so.forward(cfg, w, inputs, np.float32)
    for name, model in (("mfma", fast), ("plain", plain)):
        y, ga = model.predict(inputs)
        assert rel_err(y, y64) <= max(RTOL, 2 * rel_err(y32, y64)), name
        assert rel_err(ga, ga64) <= max(RTOL, 2 * rel_err(ga32, ga64)), name


def test_plain_fp32_forward_cross_checks_the_mfma_kernels(hip_lib, monkeypatch):
    """Two independent GPU implementations of the same graph: the split-fp16 MFMA kernels (csrc/scann_kernels.hip) and the plain-fp32
    forward written for other widths (csrc/scann_generic.hip: scalar fmaf loops, no matrix instructions, no shared code beyond swish),
    here forced onto the 128 / 8 QM9 and MP2018 configs with SCANN_GENERIC=1.  They must agree with each other as closely as each
    agrees with the fp32 restatement (attention.py:118-216, :267-318; scann_model.py:362-447)."""
    from scann.models.scann_model import HipModel

    for name, kind, n in (("qm9", "qm9", 24), ("mp2018", "mp2018", 5)):
        cfg, w, inputs, fast = make(name, n=n, seed=17, kind=kind)
        y_fast, ga_fast = fast.predict(inputs)
        monkeypatch.setenv("SCANN_GENERIC", "1")
        plain = HipModel(cfg, w, device=0, infer=True)
        monkeypatch.delenv("SCANN_GENERIC")
        y_plain, ga_plain = plain.predict(inputs)
        y32, ga32 = so.forward(cfg, w, inputs, np.float32)
        assert rel_err(y_plain, y32) <= RTOL and rel_err(ga_plain, ga32) <= RTOL, name
        assert rel_err(y_fast, y_plain) <= RTOL and rel_err(ga_fast, ga_plain) <= RTOL, (name, rel_err(y_fast, y_plain))
        assert not np.array_equal(y_fast, y_plain)  # (they ARE different arithmetic)


@pytest.mark.parametrize("infer", [True, False])
def test_predict_on_a_whole_padded_dataset_runs_as_a_pipeline_of_chunks(hip_lib, monkeypatch, infer):
    """`model.predict(x)` with x the WHOLE padded dataset (Keras batches internally; scann_model.py:266,316 are called that way by
    evaluate / predict_model.py): above HipModel.BIG_PREDICT structures the rows are cut into chunks that a producer thread packs and
    uploads while the device runs the previous one.  A structure's result does not depend on its batch, so the chunked call must
    return the BYTES of the plain call -- y, and in infer mode the padded [B, M, 1] GlobalAttention scores with zeros on padding."""
    from scann.models.scann_model import HipModel

    cfg = so.default_config("qm9")
    w = so.init_weights(cfg, 9, perturb=True)
    de, dn = so.synth_dataset(1100, 21)
    inputs, _ = so.pad_batch(de, dn, True)
    model = HipModel(cfg, w, device=0, infer=infer)
    monkeypatch.setattr(HipModel, "BIG_PREDICT", 1 << 30)
    plain = model.predict(inputs)  # one launch sequence
    monkeypatch.setattr(HipModel, "BIG_PREDICT", 1000)
    monkeypatch.setattr(HipModel, "PREDICT_CHUNK", 300)  # 300 + 300 + 300 + 200
    chunked = model.predict(inputs)
    monkeypatch.setattr(HipModel, "BIG_PREDICT", 1 << 30)
    monkeypatch.setattr(HipModel, "BIG_SLOTS", 60_000)  # ... and by padded size: few structures, many slots
    by_slots = model.predict(inputs)
    assert np.array_equal(by_slots[0] if infer else by_slots, plain[0] if infer else plain)
    if infer:
        assert chunked[0].shape == plain[0].shape == (1100, 1) and chunked[1].shape == plain[1].shape
        assert np.array_equal(chunked[0], plain[0]) and np.array_equal(chunked[1], plain[1])
        assert np.all(chunked[1][np.asarray(inputs["atom_mask"]) == 0] == 0)
    else:
        assert chunked.shape == plain.shape == (1100, 1) and np.array_equal(chunked, plain)


@pytest.mark.parametrize("widths", ["128x8", "64x4"])
def test_repeated_predicts_do_not_eat_device_memory(hip_lib, widths):
    """The drop-in call in a loop (scann_model.py:315-319 as a user runs it): `predict` on batches of changing size, many times, on
    one handle -- the MFMA kernels' and the plain-fp32 path's.  Per-call workspaces are reused or returned: after the first pass over
    the sizes the device's free memory must not fall any further."""
    from scann.models.scann_model import HipModel

    cfg = so.default_config("qm9")
    if widths == "64x4":
        cfg["model"].update(local_dim=64, num_head=4, global_dim=64, dense_out=64)
    w = so.init_weights(cfg, 5, perturb=True)
    model = HipModel(cfg, w, device=0, infer=True)
    batches = []
    for n, seed in ((6, 1), (40, 2), (17, 3), (64, 4)):
        de, dn = so.synth_dataset(n, seed)
        batches.append(so.pad_batch(de, dn, True)[0])
    first = [model.predict(b)[0].copy() for b in batches]
    free0, total = model.engine.device_memory()
    assert 0 < free0 <= total
    for rep in range(25):
        for b, y0 in zip(batches, first):
            assert np.array_equal(model.predict(b)[0], y0)
    free1, _ = model.engine.device_memory()
    assert free0 - free1 <= 32 << 20, (free0, free1)


def test_keras_default_init_and_configs(hip_lib):
    """Keras-default weights (zero biases, unit gamma) and the other shipped architectures."""
    for name, kind, n in (("qm9", "qm9", 16), ("qm9_std", "qm9", 8), ("mp2018", "mp2018", 6)):
        cfg, w, inputs, model = make(name, n=n, seed=5, kind=kind, perturb=False)
        y, ga = model.predict(inputs)
        y_ref, ga_ref = so.forward(cfg, w, inputs, np.float32)
        assert rel_err(y, y_ref) <= RTOL, name
        assert rel_err(ga, ga_ref) <= RTOL, name


def test_worst_case_and_ragged(hip_lib):
    """Swc (29 atoms x 12 neighbours everywhere) and a batch holding a 3-atom next to a 29-atom molecule."""
    cfg, w, inputs, model = make(n=6, kind="worst")
    y, ga = model.predict(inputs)
    y_ref, ga_ref = so.forward(cfg, w, inputs, np.float32)
    assert rel_err(y, y_ref) <= RTOL and rel_err(ga, ga_ref) <= RTOL
    de, dn = so.synth_dataset(64, 11)
    sizes = np.array([len(e[0]) for e in de])
    pick = [int(sizes.argmin()), int(sizes.argmax()), 0, 1]
    inputs, _ = so.pad_batch(de[pick], dn[pick], True)
    y, ga = model.predict(inputs)
    y_ref, ga_ref = so.forward(cfg, w, inputs, np.float32)
    assert rel_err(y, y_ref) <= RTOL and rel_err(ga, ga_ref) <= RTOL


def test_isolated_atom_and_masked_garbage(hip_lib):
    """An atom whose neighbour slots are all masked gives ctx = LN(q) (attention.py:186-212); garbage in masked
    slots changes nothing; extra padding rows/columns leave real outputs unchanged."""
    cfg, w, inputs, model = make(n=5, seed=7)
    inputs = {k: np.array(v) for k, v in inputs.items()}
    inputs["neighbor_mask"][0, 1, :] = False  # isolate atom 1 of molecule 0
    y0, ga0 = model.predict(inputs)
    y_ref, ga_ref = so.forward(cfg, w, inputs, np.float32)
    assert rel_err(y0, y_ref) <= RTOL and rel_err(ga0, ga_ref) <= RTOL
    rng = np.random.default_rng(0)
    g = {k: np.array(v) for k, v in inputs.items()}
    dead = ~g["neighbor_mask"]
    g["neighbor_distance"][dead] = rng.uniform(0, 9, dead.sum()).astype("float32")
    g["neighbor_weight"][dead] = rng.uniform(0, 9, dead.sum()).astype("float32")
    y1, ga1 = model.predict(g)
    assert np.array_equal(y0, y1) and np.array_equal(ga0, ga1)
    B, M, N = g["neighbors"].shape
    pad = {
        "atomic": np.pad(inputs["atomic"], ((0, 0), (0, 3))),
        "atom_mask": np.pad(inputs["atom_mask"], ((0, 0), (0, 3), (0, 0))),
        "neighbors": np.pad(inputs["neighbors"], ((0, 0), (0, 3), (0, 2))),
        "neighbor_mask": np.pad(inputs["neighbor_mask"], ((0, 0), (0, 3), (0, 2))),
        "neighbor_weight": np.pad(inputs["neighbor_weight"], ((0, 0), (0, 3), (0, 2))),
        "neighbor_distance": np.pad(inputs["neighbor_distance"], ((0, 0), (0, 3), (0, 2))),
    }
    y2, ga2 = model.predict(pad)
    assert np.array_equal(y0, y2) and np.array_equal(ga0, ga2[:, :M]) and not ga2[:, M:].any()


def test_batch_composition_and_permutation(hip_lib):
    """Molecule order permutes outputs; a molecule's result does not depend on its batch mates; relabelling the
    atoms of a molecule leaves y unchanged (to rounding) and permutes the GA scores."""
    cfg, w, inputs, model = make(n=9, seed=2)
    de, dn = so.synth_dataset(9, 2)
    y, ga = model.predict(inputs)
    perm = np.random.default_rng(1).permutation(9)
    ip, _ = so.pad_batch(de[perm], dn[perm], True)
    yp, gap = model.predict(ip)
    assert np.array_equal(yp, y[perm])
    single, _ = so.pad_batch(de[2:3], dn[2:3], True)
    ys, gas = model.predict(single)
    assert np.array_equal(ys[0], y[2])
    # atom relabelling of molecule 0
    A = len(de[0][0])
    p = np.random.default_rng(3).permutation(A)          # new position i holds old atom p[i]
    inv = np.argsort(p)
    e0 = [[de[0][0][j] for j in p], de[0][1]]
    n0 = [[[n[0], int(inv[n[1]]), n[2], n[3], n[4]] for n in dn[0][j]] for j in p]
    de2, dn2 = np.empty(1, dtype=object), np.empty(1, dtype=object)
    de2[0], dn2[0] = e0, n0
    i2, _ = so.pad_batch(de2, dn2, True)
    y2, ga2 = model.predict(i2)
    assert rel_err(y2[0], y[0]) <= 1e-5
    assert rel_err(ga2[0, :A, 0], ga[0, :A, 0][p]) <= 1e-5


def test_single_atom_structure_is_nan_like_reference(hip_lib):
    """tf.linalg.normalize has no epsilon: a 1-atom structure gives 0/0 = NaN with use_ga_norm (attention.py:297)."""
    from scann import _hip

    cfg, w, inputs, model = make(n=2)
    pk = _hip.PackedBatch([6, 1, 1, 8], [0, 1, 4], [0, 0, 1, 2, 4], [2, 1, 1, 2], [1.0, 1.1, 1.2, 1.3],
                          [1.0, 2.0, 1.5, 0.7])
    y, ga = model.engine.forward(pk)
    assert np.isnan(y[0]) and np.isfinite(y[1])


def test_resident_pipeline_matches_sync(hip_lib):
    """Batches run asynchronously on different streams give the same bytes as the synchronous call."""
    from scann import _hip

    cfg, w, inputs, model = make(n=8)
    eng = model.engine
    pks = []
    for s in range(6):
        de, dn = so.synth_dataset(8 + s, 20 + s)
        pks.append(_hip.pack_inputs(so.pad_batch(de, dn, True)[0]))
    sync = [eng.forward(pk) for pk in pks]
    rbs = [eng.upload(pk) for pk in pks]
    for rep in range(3):
        for i, rb in enumerate(rbs):
            eng.forward_resident(rb, i)
    eng.sync()
    for rb, (y, ga) in zip(rbs, sync):
        y2, ga2 = eng.download(rb)
        assert np.array_equal(y, y2) and np.array_equal(ga, ga2)
        rb.free()


def test_errors_are_reported(hip_lib):
    from scann import _hip

    cfg, w, inputs, model = make(n=2)
    bad = _hip.PackedBatch([6, 1], [0, 2], [0, 1, 2], [1, 5], [1.0, 1.0], [1.0, 1.0])  # neighbour outside structure
    with pytest.raises(_hip.ScannHipError) as e:
        model.engine.forward(bad)
    assert e.value.code == -1
    bad = _hip.PackedBatch([6, 11], [0, 2], [0, 1, 2], [1, 0], [1.0, 1.0], [1.0, 1.0])  # Z >= n_atoms
    with pytest.raises(_hip.ScannHipError):
        model.engine.forward(bad)
    w2 = dict(w)
    w2.pop("after_Lc/bias")
    with pytest.raises(_hip.ScannHipError):
        model.engine.load_weights(w2)


@pytest.mark.parametrize("name", ["qm9_plus", "qm9_base", "qm9_no_norms", "qm9_e_b", "mp2018", "dense_and_sparse"])
def test_golden_vectors(hip_lib, name):
    """Committed fixtures (tests/golden/*.npz, written by make_golden.py from the oracle)."""
    import importlib.util
    import os

    from scann.models.scann_model import HipModel

    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(here, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    cfg, w, _ = mg.build(name)
    z = np.load(os.path.join(here, "golden", name + ".npz"))
    inputs = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    model = HipModel(cfg, w, device=0, infer=True)
    y, ga = model.predict(inputs)
    tol_y = max(RTOL, 2 * rel_err(z["y32"], z["y64"]))
    tol_g = max(RTOL, 2 * rel_err(z["ga32"], z["ga64"]))
    assert rel_err(y, z["y64"]) <= tol_y and rel_err(ga, z["ga64"]) <= tol_g


def test_full_size_properties(hip_lib):
    """BASELINE config 2 sizes (batch 128, QM9 shapes) through the resident multi-stream pipeline: size-independent
    properties instead of the (slow) oracle -- GA scores of every molecule sum to 1, outputs are finite, a batch split
    in two gives the same bytes (batch-composition independence), and a sampled batch matches the fp32 oracle."""
    from scann import _hip

    cfg, w, _, model = make(n=2)
    eng = model.engine
    de, dn = so.synth_dataset(128 * 12, 77)
    rbs, pks = [], []
    for b in range(12):
        inputs, _ = so.pad_batch(de[128 * b:128 * (b + 1)], dn[128 * b:128 * (b + 1)], True)
        pks.append(_hip.pack_inputs(inputs))
        rbs.append(eng.upload(pks[-1]))
    for rep in range(2):
        for i, rb in enumerate(rbs):
            eng.forward_resident(rb, i)
    eng.sync()
    outs = [eng.download(rb) for rb in rbs]
    for pk, (y, ga) in zip(pks, outs):
        assert np.isfinite(y).all() and np.isfinite(ga).all()
        sums = np.add.reduceat(ga, pk.mol_offset[:-1])
        assert np.allclose(sums, 1.0, atol=2e-6)
    from scann.parallel import concat_outputs, split_packed

    halves = split_packed(pks[3], 2)
    y2, ga2 = concat_outputs([eng.forward(h) for h in halves])
    assert np.array_equal(y2, outs[3][0]) and np.array_equal(ga2, outs[3][1])
    inputs, _ = so.pad_batch(de[128 * 5:128 * 6], dn[128 * 5:128 * 6], True)
    y_ref, ga_ref = so.forward(cfg, w, inputs, np.float32)
    assert rel_err(outs[5][0], y_ref[:, 0]) <= RTOL
    assert rel_err(outs[5][1], ga_ref[..., 0][inputs["atom_mask"][..., 0]]) <= RTOL
    for rb in rbs:
        rb.free()


def test_predict_dataset_pipeline_equals_per_batch_predict(hip_lib):
    """The pipelined / grouped dataset path (SCANN.evaluate, predict_model.py) returns the bytes of per-batch predict."""
    from scann.utils import DataIterator, PackedDataset

    cfg, w, _, model = make(n=2)
    de, dn = so.synth_dataset(70, 41)
    it = DataIterator(de, dn, batch_size=8, g_update=True)
    pd_ = PackedDataset(de, dn, batch_size=8, g_update=True)
    ref_y = np.concatenate([model.predict(it[i][0])[0][:, 0] for i in range(len(it))])
    for data in (it, pd_):
        for group in (1, 3):
            y, ga, t = model.predict_dataset(data, group=group, want_ga=True)
            assert np.array_equal(y, ref_y) and len(t) == 70 and ga.shape[0] == sum(len(e[0]) for e in de)


def test_degenerate_batches(hip_lib):
    """No edges at all (every atom isolated: ctx = LN(q) in every layer), and atoms with the maximum 64 neighbours."""
    from scann import _hip

    cfg, w, _, model = make(n=2)
    # (a) three structures, no edges
    B, M, N = 3, 4, 2
    atomic = np.array([[6, 1, 1, 0], [8, 1, 0, 0], [7, 6, 1, 1]], dtype="int32")
    inputs = {"atomic": atomic, "atom_mask": (atomic != 0)[..., None], "neighbors": np.zeros((B, M, N), "int32"),
              "neighbor_mask": np.zeros((B, M, N), bool), "neighbor_weight": np.zeros((B, M, N), "float32"),
              "neighbor_distance": np.zeros((B, M, N), "float32")}
    y, ga = model.predict(inputs)
    y_ref, ga_ref = so.forward(cfg, w, inputs, np.float32)
    assert rel_err(y, y_ref) <= RTOL and rel_err(ga, ga_ref) <= RTOL
    # (b) one 70-atom structure whose atoms each have 64 neighbours, next to a small molecule
    rng = np.random.default_rng(0)
    A = 70
    big = [[[6, int(j), float(rng.uniform(0.4, 3.5)), 1.0, float(rng.uniform(0.9, 4.0))]
            for j in rng.choice(np.delete(np.arange(A), a), 64, replace=False)] for a in range(A)]
    de, dn = so.synth_dataset(1, 3)
    de2, dn2 = np.empty(2, dtype=object), np.empty(2, dtype=object)
    de2[0], dn2[0] = [[6] * A, 0.0], big
    de2[1], dn2[1] = de[0], dn[0]
    inputs, _ = so.pad_batch(de2, dn2, True)
    y, ga = model.predict(inputs)
    y_ref, ga_ref = so.forward(cfg, w, inputs, np.float32)
    assert rel_err(y, y_ref) <= RTOL and rel_err(ga, ga_ref) <= RTOL
    # (c) 65 neighbours: one more than an edge tile holds -- two chunk tiles merged by edge_merge_kernel
    big[0].append([6, 1, 1.0, 1.0, 1.0])
    de2[0], dn2[0] = [[6] * A, 0.0], big
    inputs, _ = so.pad_batch(de2, dn2, True)
    y, ga = model.predict(inputs)
    y_ref, ga_ref = so.forward(cfg, w, inputs, np.float32)
    assert rel_err(y, y_ref) <= RTOL and rel_err(ga, ga_ref) <= RTOL


@pytest.mark.parametrize("g_update", [True, False], ids=["scann_plus", "base"])
def test_more_than_64_neighbours(hip_lib, g_update):
    """Atoms with 65 ... 219 neighbours (chunk tiles + softmax merge), between ordinary atoms and next to small molecules,
    on BOTH branches of LocalAttention (one edge-kernel family: attention.py:141-155)."""
    from scann import _hip

    cfg, w, _, model = make(n=2, model=dict(g_update=g_update))
    rng = np.random.default_rng(11)
    A = 220
    degs = {0: 219, 1: 65, 7: 128, 8: 129, 9: 64, 100: 200, 219: 70}
    nb = []
    for a in range(A):
        d = degs.get(a, int(rng.integers(0, 9)))
        js = rng.choice(np.delete(np.arange(A), a), d, replace=False)
        ang = rng.uniform(0.4, 3.5, size=d)
        nb.append([[6, int(j), float(ang[k]), float(ang[k] / ang.max()), float(rng.uniform(0.9, 4.0))] for k, j in enumerate(js)])
    de, dn = so.synth_dataset(2, 3)
    de3, dn3 = np.empty(3, dtype=object), np.empty(3, dtype=object)
    de3[0], dn3[0] = de[0], dn[0]
    de3[1], dn3[1] = [[int(z) for z in rng.choice([1, 6, 7, 8], A)], 0.0], nb
    de3[2], dn3[2] = de[1], dn[1]
    inputs, _ = so.pad_batch(de3, dn3, g_update)
    y, ga = model.predict(inputs)
    y_ref, ga_ref = so.forward(cfg, w, inputs, np.float32)
    assert rel_err(y, y_ref) <= RTOL and rel_err(ga, ga_ref) <= RTOL
    # the same batch through the resident / multi-stream path
    pk = _hip.pack_inputs(inputs)
    rb = model.engine.upload(pk)
    model.engine.forward_resident(rb, 1)
    y2, ga2 = model.engine.download(rb)
    assert np.array_equal(y2, y[:, 0]) and np.array_equal(pk.repad_ga(ga2), ga)
    rb.free()


def test_tile_order_and_streams_switches(hip_lib, monkeypatch):
    """The two remaining runtime switches (launch-order tiles instead of one run per XCD; 2 streams) change nothing."""
    cfg, w, inputs, model = make(n=40, seed=13)
    y0, ga0 = model.predict(inputs)
    monkeypatch.setenv("SCANN_XCD_REMAP", "0")
    monkeypatch.setenv("SCANN_STREAMS", "2")
    cfg, w, inputs, model = make(n=40, seed=13)
    y, ga = model.predict(inputs)
    y_ref, ga_ref = so.forward(cfg, w, inputs, np.float32)
    assert np.array_equal(y, y0) and np.array_equal(ga, ga0)
    assert rel_err(y, y_ref) <= RTOL and rel_err(ga, ga_ref) <= RTOL


def test_split_fp16_projections_reach_fp32_accuracy(hip_lib):
    """The edge kernel multiplies hi/lo fp16 operand pairs on the f16 matrix pipe (three products, fp32 accumulate).  Its
    per-layer geometry and context must sit as close to the fp64 oracle as the fp32 oracle does (DESIGN.md numerics): the
    split is an fp32-accuracy scheme, not a reduced-precision one."""
    from scann import _hip

    cfg, w, inputs, model = make(n=24, seed=5)
    eng = model.engine
    eng.set_debug(True)
    rb = eng.upload(_hip.pack_inputs(inputs))
    eng.forward_resident(rb, 0)
    eng.sync()
    amask = inputs["atom_mask"][..., 0]
    nmask = inputs["neighbor_mask"] & amask[:, :, None]
    t64, t32 = {}, {}
    so.forward(cfg, w, inputs, np.float64, intermediates=t64)
    so.forward(cfg, w, inputs, np.float32, intermediates=t32)
    L = cfg["model"]["n_attention"]
    for l in (1, L):
        for what, key, mask in ((1, "geometry_%d" % l, nmask), (2, "context_%d" % l, amask)):
            got = eng.debug_read(rb, what, l)
            e_gpu = rel_err(got, t64[key][mask])
            e_f32 = rel_err(t32[key][mask], t64[key][mask])
            assert e_gpu <= max(2.0 * e_f32, 2e-6), (key, e_gpu, e_f32)
    rb.free()


def test_sparse_graphs_hit_the_atoms_per_tile_limit(hip_lib):
    """Chains and stars: atoms with one or two neighbours, so an edge tile is closed by its atom count (TQ = 24 for
    edge_kernel_lean) long before it holds 64 edges; includes atoms without any neighbour between them."""
    cfg, w, _, model = make(n=2)
    rng = np.random.default_rng(5)
    de, dn = np.empty(4, dtype=object), np.empty(4, dtype=object)
    for s, A in enumerate((29, 27, 29, 3)):
        nb = []
        for a in range(A):
            if s == 0:    # chain: both chain neighbours
                js = [j for j in (a - 1, a + 1) if 0 <= j < A]
            elif s == 1:  # star: everybody sees atom 0 only, atom 0 sees three of them
                js = [0] if a else [1, 2, 3]
            elif s == 2:  # every third atom isolated, the others see one neighbour
                js = [] if a % 3 == 0 else [a - 1]
            else:
                js = [(a + 1) % A]
            nb.append([[6, int(j), float(rng.uniform(0.4, 3.5)), 1.0, float(rng.uniform(0.9, 4.0))] for j in js])
        de[s], dn[s] = [[int(z) for z in rng.choice([1, 6, 7, 8], A)], 0.0], nb
    inputs, _ = so.pad_batch(de, dn, True)
    y, ga = model.predict(inputs)
    y_ref, ga_ref = so.forward(cfg, w, inputs, np.float32)
    assert rel_err(y, y_ref) <= RTOL and rel_err(ga, ga_ref) <= RTOL


def test_multi_gpu_predictor_matches_single_handle(hip_lib):
    """One process, one handle + host thread per device (two handles on the one GPU of this box): same numbers, same order."""
    from scann.parallel import MultiGpuPredictor
    from scann.utils import PackedDataset

    cfg, w, _, model = make(n=2)
    de, dn = so.synth_dataset(90, 21)
    ds = PackedDataset(de, dn, batch_size=8, g_update=True)
    y1, ga1, t1 = model.predict_dataset(ds, group=3, want_ga=True)
    multi = MultiGpuPredictor(cfg, w, devices=[0, 0])
    y2, ga2, t2 = multi.predict_dataset(ds, group=3, want_ga=True)
    assert np.array_equal(t1, t2) and y1.shape == y2.shape == (90,)
    assert rel_err(y2, y1) <= 1e-6 and rel_err(ga2, ga1) <= 1e-6
    runs = MultiGpuPredictor._runs([5, 1, 1, 1, 5, 5], 3)
    assert runs[0][0] == 0 and runs[-1][1] == 6 and all(hi > lo for lo, hi in runs) and len(runs) == 3


def test_s134k_sampled_batches_match_c_oracle(hip_lib):
    """BASELINE configs[1] at FULL size, sampled: S134k = 130,831 synthetic QM9-shaped molecules in 1,023 batches of 128
    (batch k is synth_dataset(n, seed=1000 + 128 k), tests/manual/parity_s134k.py runs all of them).  Eight batches spread
    over the set -- the ragged last one included -- go through the dataset pipeline (predict_dataset) with the full L=7
    model and Keras-default weights and are compared with the C/OpenMP port of the oracle at the north-star bound."""
    import scann_oracle_c as soc
    from scann.models.scann_model import HipModel, normalize_config

    N, B = 130831, 128
    n_batches = (N + B - 1) // B
    assert n_batches == 1023
    cfg = normalize_config(so.default_config("qm9"))
    w = so.init_weights(cfg, 1234)
    model = HipModel(cfg, w, device=0, infer=True)
    rng = np.random.default_rng(2)
    picks = sorted(set(rng.choice(n_batches - 1, size=7, replace=False).tolist()) | {n_batches - 1})
    items, refs = [], []
    for k in picks:
        n = min(B, N - B * k)
        de, dn = so.synth_dataset(n, seed=1000 + B * k)
        inputs, tgt = so.pad_batch(de, dn, True)
        items.append((inputs, tgt))
        refs.append(soc.forward(cfg, w, inputs))
    assert items[-1][0]["atomic"].shape[0] == N - B * (n_batches - 1) == 15
    y, ga, t = model.predict_dataset(items, group=3, want_ga=True)
    y_ref = np.concatenate([r[0][:, 0] for r in refs])
    ga_ref = np.concatenate([r[1][..., 0][it[0]["atom_mask"][..., 0]] for r, it in zip(refs, items)])
    assert y.shape == y_ref.shape == (7 * B + 15,) and np.array_equal(t, np.concatenate([it[1] for it in items]))
    assert rel_err(y, y_ref) <= RTOL, rel_err(y, y_ref)
    assert strict_rel_err(y, y_ref) <= RTOL, strict_rel_err(y, y_ref)  # |dy| / max(|y|, 1e-6), all 911 predictions
    assert float(np.max(np.abs(ga - ga_ref))) <= 1e-5
    # the MAE SCANN.evaluate() would print (scann_model.py:273-280) agrees to the same bound
    mae_gpu, mae_ref = float(np.mean(np.abs(y - t))), float(np.mean(np.abs(y_ref - t)))
    assert abs(mae_gpu - mae_ref) <= RTOL * mae_ref


def test_bench_two_ranks_through_the_self_spawning_launcher(hip_lib):
    """`python bench.py --gpus 2` with no external launcher: the parent spawns the ranks, they meet over the loopback
    rendezvous, and rank 0 prints ONE line with n_gpus 2.  On a one-GPU box both ranks share the device (--oversubscribe:
    a rehearsal of the multi-process path, not a scaling number)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ndev = hip_lib.scann_device_count()
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-extras",
           "--min-time", "0.2", "--prewarm", "0.1", "--pool", "32"] + (["--oversubscribe"] if ndev < 2 else [])
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["value"] > 0
    assert out.get("oversubscribed", False) == (ndev < 2)
    assert out["roofline"]["launches_sampled"] > 0 and 0 < out["roofline"]["frac"] < 1.5
    # what a SCALE record reads off the line: inference has no collective (rccl_ranks null), every rank's own rate, and the N = 1 rate
    # of the same process layout measured inside this run
    assert out["rccl_ranks"] is None and out["ranks_reporting"] == 2 and len(out["rank_values"]) == 2 and min(out["rank_values"]) > 0
    assert out["n1_same_layout"]["value"] > 0
    assert out["value"] <= 2.0 * max(out["rank_values"]) * 1.001  # the job's rate is 2 x the SLOWEST rank's


def test_keras_h5_checkpoint_loads_and_predicts(hip_lib):
    """SCANN(config, pretrained=<Keras .h5>, mode="infer") -- the reference's way of loading a trained model
    (scann_model.py:79-83) -- through the pure-Python HDF5 reader and the Keras-name map, on the COMMITTED file
    tests/golden/keras_layout_qm9_L2.h5 (Keras' ModelCheckpoint layout written by h5py, not by TensorFlow: unavailable here;
    tests/golden/make_keras_fixture.py) -- no h5py needed on the box that runs this.  Same predictions as the same seeded weights
    loaded from the native container, and the oracle's numbers."""
    import importlib.util
    import os

    from scann.models import SCANN
    from scann.models.scann_model import HipModel

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_keras_fixture", os.path.join(root, "tests", "golden", "make_keras_fixture.py"))
    fx = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fx)
    cfg, w = fx.fixture_config(), fx.fixture_weights()
    de, dn = so.synth_dataset(10, 3)
    inputs, _ = so.pad_batch(de, dn, g_update=cfg["model"]["g_update"])
    yaml_cfg = {"model": {k: v for k, v in cfg["model"].items() if k in ("n_atoms", "scale")}, "hyper": {"target": "homo"}}
    scann = SCANN(yaml_cfg, pretrained=os.path.join(root, "tests", "golden", fx.NAME), mode="infer")
    y1, ga1 = scann.model.predict(inputs)
    y0, ga0 = HipModel(cfg, w, device=0, infer=True).predict(inputs)
    assert np.array_equal(y1, y0) and np.array_equal(ga1, ga0)
    assert scann.model.config["model"]["n_attention"] == 2
    y_ref, ga_ref = so.forward(cfg, w, inputs, np.float32)
    assert rel_err(y1, y_ref) <= RTOL and rel_err(ga1, ga_ref) <= RTOL


def test_activation_outside_the_split_fp16_range_is_rerun_in_exact_fp32(hip_lib, monkeypatch):
    """The projections carry every operand as fp16 hi + lo parts: an activation beyond 65504 would become inf, then NaN.  The
    kernels test every LayerNorm variance downstream of a split (and the one activation no LayerNorm follows).  When that guard
    fires, the call that returns the results runs the forward AGAIN on exact-fp32 matrix instructions -- the reference evaluates
    any fp32 checkpoint (attention.py:95-113) -- and returns the oracle's numbers; with SCANN_STRICT_RANGE=1 it fails with
    SCANN_ERR_RANGE and names the layer instead.  Never NaNs, and the handle stays usable."""
    from scann import _hip
    from scann.models.scann_model import HipModel

    cfg = so.default_config("qm9")
    w = so.init_weights(cfg, 1234, perturb=True)
    de, dn = so.synth_dataset(6, 0)
    inputs, _ = so.pad_batch(de, dn, True)
    y_ok = HipModel(cfg, w, device=0).predict(inputs)
    assert np.isfinite(y_ok).all()
    cases = []
    bad = dict(w)
    bad["local_attention_1/layer_norm_g/gamma"] = (w["local_attention_1/layer_norm_g/gamma"] * 3.0e5).astype(np.float32)  # geom' ~ 3e5
    cases.append((bad, "local_attention_"))
    bad = dict(w)
    bad["after_Lc/bias"] = (w["after_Lc/bias"] + 1.0e5).astype(np.float32)  # the activation with no LayerNorm behind it
    cases.append((bad, "after_Lc"))
    for bad, where in cases:
        model = HipModel(cfg, bad, device=0, infer=True)
        assert model.engine.exact_reruns() == 0
        y, ga = model.predict(inputs)
        assert model.engine.exact_reruns() == 1
        y32, ga32 = so.forward(cfg, bad, inputs, np.float32)
        y64, ga64 = so.forward(cfg, bad, inputs, np.float64)
        assert np.isfinite(y).all() and np.isfinite(ga).all()
        assert rel_err(y, y64) <= max(RTOL, 2 * rel_err(y32, y64)), (where, rel_err(y, y64), rel_err(y32, y64))
        assert rel_err(ga, ga64) <= max(RTOL, 2 * rel_err(ga32, ga64)), (where, rel_err(ga, ga64), rel_err(ga32, ga64))
        model.set_weights(w)  # same handle, sane weights again: the fast path, the same bytes as before, no further re-run
        assert np.array_equal(model.predict(inputs)[0], y_ok) and model.engine.exact_reruns() == 1
        # the resident-batch pipeline: the batch whose guard fired is re-run at its download, its neighbour on the other stream is not
        model.set_weights(bad)
        pk = _hip.pack_inputs(inputs)
        rb1, rb2 = model.engine.upload(pk), model.engine.upload(pk)
        model.engine.forward_resident(rb1, 0)
        model.engine.forward_resident(rb2, 1)
        y1, _ = model.engine.download(rb1)
        y2, _ = model.engine.download(rb2)
        assert np.array_equal(y1, y[:, 0]) and np.array_equal(y2, y[:, 0]) and model.engine.exact_reruns() == 3
        rb1.free()
        rb2.free()
    # the exact kernels on an ordinary model: the fp32 oracle's numbers (they ARE the reference's arithmetic)
    monkeypatch.setenv("SCANN_STRICT_RANGE", "1")
    for bad, where in cases:
        model = HipModel(cfg, bad, device=0)
        with pytest.raises(_hip.ScannHipError) as ei:
            model.predict(inputs)
        assert ei.value.code == -7 and where in str(ei.value) and "65504" in str(ei.value), str(ei.value)
        model.set_weights(w)  # the flag was cleared by the failed call
        assert np.array_equal(model.predict(inputs), y_ok)


@pytest.mark.parametrize("name,kind,n,over", [
    ("qm9", "qm9", 40, {}), ("qm9", "qm9", 12, dict(g_update=False)), ("qm9", "qm9", 12, dict(use_attn_norm=False)),
    ("mp2018", "mp2018", 6, {}), ("qm9", "worst", 4, {}),
], ids=["qm9", "base", "no_attn_norm", "mp2018", "worst"])
def test_exact_fp32_kernels_match_the_oracle(hip_lib, monkeypatch, name, kind, n, over):
    """SCANN_EXACT=1 runs every inference forward on the EX instantiations of the atom / edge kernels (exact-fp32 MFMA: the re-run
    path of a forward whose split-fp16 range guard fired), here on ordinary models: the oracle's numbers on every branch, 32- and
    64-row tiles, and the fast path's to the split scheme's accuracy."""
    from scann.models.scann_model import HipModel

    cfg, w, inputs, fast = make(name, n=n, seed=9, kind=kind, model=dict(over))
    y_fast, ga_fast = fast.predict(inputs)
    monkeypatch.setenv("SCANN_EXACT", "1")
    y, ga = HipModel(cfg, w, device=0, infer=True).predict(inputs)
    y32, ga32 = so.forward(cfg, w, inputs, np.float32)
    y64, ga64 = so.forward(cfg, w, inputs, np.float64)
    assert rel_err(y, y64) <= max(RTOL, 2 * rel_err(y32, y64)) and rel_err(ga, ga64) <= max(RTOL, 2 * rel_err(ga32, ga64))
    assert rel_err(y, y_fast) <= RTOL and rel_err(ga, ga_fast) <= RTOL


def test_weights_beyond_the_split_fp16_range_run_on_the_exact_kernels(hip_lib):
    """A 128x128 kernel with |w| >= 255.9 cannot be held as fp16 hi / lo parts of w * 2^8.  The reference loads any fp32 checkpoint
    (scann_model.py:79): such a handle runs its inference forwards on the exact-fp32 kernels (the oracle's numbers), refuses to
    train, and a K = 20 filter -- multiplied in split form only -- or a value that is not finite is still refused at load."""
    from scann import _hip
    from scann.models.scann_model import HipModel

    cfg = so.default_config("qm9")
    w = so.init_weights(cfg, 1234, perturb=True)
    de, dn = so.synth_dataset(9, 3)
    inputs, _ = so.pad_batch(de, dn, True)
    big = dict(w)
    big["local_attention_0/key/kernel"] = (w["local_attention_0/key/kernel"] * 4000.0).astype(np.float32)
    big["residual_norm_2/dense_1/kernel"] = (w["residual_norm_2/dense_1/kernel"] * 3000.0).astype(np.float32)
    assert np.abs(big["local_attention_0/key/kernel"]).max() > 255.9
    model = HipModel(cfg, big, device=0, infer=True)
    y, ga = model.predict(inputs)
    y32, ga32 = so.forward(cfg, big, inputs, np.float32)
    y64, ga64 = so.forward(cfg, big, inputs, np.float64)
    assert np.isfinite(y).all()
    assert rel_err(y, y64) <= max(RTOL, 2 * rel_err(y32, y64)), (rel_err(y, y64), rel_err(y32, y64))
    assert rel_err(ga, ga64) <= max(RTOL, 2 * rel_err(ga32, ga64))
    with pytest.raises(_hip.ScannHipError) as ei:
        model.engine.train_begin()
    assert ei.value.code == -2 and "255.9" in str(ei.value)
    for name, factor in (("neighbor_d/kernel", 4000.0), ("after_Lc/kernel", np.inf)):
        bad = dict(w)
        bad[name] = (w[name] * factor).astype(np.float32)
        with pytest.raises(_hip.ScannHipError) as ei:
            HipModel(cfg, bad, device=0)
        assert ei.value.code == -2 and name in str(ei.value), str(ei.value)
    model.set_weights(w)  # back inside the range: the fast path again
    assert np.array_equal(model.predict(inputs)[0], HipModel(cfg, w, device=0, infer=True).predict(inputs)[0])
    model.engine.train_begin()


def test_process_per_gpu_predictor_equals_single_handle(hip_lib):
    """MultiProcessPredictor: one worker PROCESS per device (spawned: fresh interpreters), the dataset's CSR arrays in shared
    memory, outputs in a shared array -- the same bytes as one handle predicting the whole dataset.  Two workers on the one GPU
    of this box rehearse the plumbing; inference needs no collective."""
    from scann.models.scann_model import HipModel, normalize_config
    from scann.parallel import MultiProcessPredictor
    from scann.utils import PackedDataset

    cfg = normalize_config(so.default_config("qm9"))
    cfg["model"]["n_attention"] = 2
    w = so.init_weights(cfg, 21, perturb=True)
    de, dn = so.synth_dataset(37, 6)
    ds = PackedDataset(data_energy=de, data_neighbor=dn, batch_size=5, use_ring=False, feature="atomic", g_update=True,
                       atomic_features=None, shuffle=False)
    ref_y, ref_ga, ref_t = HipModel(cfg, w, device=0, infer=True).predict_dataset(ds, group=2, want_ga=True)
    with MultiProcessPredictor(cfg, w, devices=[0, 0], infer=True) as mp:
        y, ga, t = mp.predict_dataset(ds, group=2, want_ga=True)
        y2, _, _ = mp.predict_dataset(ds, group=3)  # the shared dataset is reused
        assert np.array_equal(y, ref_y) and np.array_equal(ga, ref_ga) and np.array_equal(t, ref_t)
        assert np.array_equal(y2, ref_y)
        # the caller reshuffles (on_epoch_end) and re-batches between two calls: the workers must slice with the order and the
        # batch size of THIS call, not with the copy they saw first
        ds.shuffle = True
        np.random.seed(4)
        ds.on_epoch_end()
        ds.batch_size = 7
        assert not np.array_equal(ds.indexes, np.arange(37))
        single = HipModel(cfg, w, device=0, infer=True)
        sy, _, st = single.predict_dataset(ds, group=2)
        y3, _, t3 = mp.predict_dataset(ds, group=2)
        assert np.array_equal(t3, st) and np.array_equal(t3, ds.target[ds.indexes])
        assert np.array_equal(y3, sy)
        assert np.array_equal(np.sort(y3), np.sort(ref_y))  # the same structures, another order
        mp.forget(ds)
        # a second dataset object gets a token of its own (never reused), and a collected one is dropped at the next call
        ds_b = PackedDataset(data_energy=de[:9], data_neighbor=dn[:9], batch_size=4, use_ring=False, feature="atomic", g_update=True,
                             atomic_features=None, shuffle=False)
        yb, _, _ = mp.predict_dataset(ds_b)
        assert np.array_equal(yb, ref_y[:9]) and ds_b._scann_mp_token != ds._scann_mp_token
        tok = ds_b._scann_mp_token
        del ds_b
        import gc
        gc.collect()
        mp.predict_dataset(ds)
        assert tok not in mp._shared



def test_first_layer_computes_its_geometry_rows_itself(hip_lib, monkeypatch):
    """Inference launches never write geom0: the first layer's edge kernel runs the basis MLP on its own rows, and with the per-species
    embedding table it takes c, P1, P3, q from per-species tables instead of an atom launch.  Same bytes as the plain path
    (SCANN_FUSE_BASIS=0), on 64- and 32-row tiles, and within the contract of the oracle."""
    from scann.models.scann_model import HipModel

    cfg = so.default_config("qm9")
    w = so.init_weights(cfg, 1234, perturb=True)
    de, dn = so.synth_dataset(384, 21, "qm9")  # > 32 Ki edges: the plan uses 64-row tiles
    inputs, _ = so.pad_batch(de, dn, g_update=True)
    out = {}
    for flag in ("1", "0", "basis only"):  # fused + per-species tables | two-kernel path, atom launch | fused basis, atom launch
        monkeypatch.setenv("SCANN_FUSE_BASIS", "0" if flag == "0" else "1")
        monkeypatch.setenv("SCANN_SPECIES_TABLES", "1" if flag == "1" else "0")
        out[flag] = HipModel(cfg, w, device=0, infer=True).predict(inputs)
    assert int(inputs["neighbor_mask"].sum()) > 32 * 1024
    for flag in ("0", "basis only"):
        assert np.array_equal(out["1"][0], out[flag][0]) and np.array_equal(out["1"][1], out[flag][1])
    small = {k: v[:16] for k, v in inputs.items()}  # one round of 32-row tiles: the other instantiation
    monkeypatch.setenv("SCANN_FUSE_BASIS", "1")
    monkeypatch.setenv("SCANN_SPECIES_TABLES", "1")
    ys = HipModel(cfg, w, device=0, infer=True).predict(small)
    assert np.array_equal(ys[0], out["0"][0][:16])
    sub = {k: v[:48] for k, v in inputs.items()}
    y_ref, _ = so.forward(cfg, w, sub, np.float32)
    assert rel_err(out["1"][0][:48], y_ref) <= RTOL


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["qm9", "worst", "mp2018", "corners"])
def test_device_packing_gives_the_host_packers_arrays(hip_lib, case):
    """Padded -> CSR on the DEVICE (scann_upload_padded: the host reads the masks, pack_padded_kernel compacts the payload arrays that
    crossed the bus as they were) against the native host packer (scann_pack_padded, i.e. what DataIterator.__getitem__ + gather_shape
    amount to: datagenerator.py:69-135, custom_layers.py:18-28): every CSR array byte for byte -- QM9-shaped, worst-case 29 x 12,
    MP2018-shaped crystals (hundreds of atoms, up to 24 neighbours), and the corners: an atom without neighbours, a one-atom structure,
    garbage in masked slots, float32 masks.  Then the forward on the device-packed batch: the bytes of the host-packed one."""
    from scann import _hip
    from scann.models.scann_model import HipModel

    rng = np.random.default_rng(11)
    name = "mp2018" if case == "mp2018" else "qm9"
    cfg = so.default_config(name)
    w = so.init_weights(cfg, 1234, perturb=True)
    if case == "corners":
        de, dn = so.synth_dataset(9, 4)
        de[2] = [[6], 0.5]
        dn[2] = [[]]           # a one-atom structure (no neighbour at all)
        dn[5][1] = []           # an atom without neighbours inside an ordinary molecule
    else:
        de, dn = so.synth_dataset({"qm9": 300, "worst": 64, "mp2018": 24}[case], 7, kind=case)
    inputs, _ = so.pad_batch(de, dn, True)
    inputs["neighbors"] = np.where(inputs["neighbor_mask"], inputs["neighbors"], rng.integers(0, 2**30, inputs["neighbors"].shape)).astype(np.int32)
    inputs["neighbor_distance"] = np.where(inputs["neighbor_mask"], inputs["neighbor_distance"], np.float32(np.nan)).astype(np.float32)
    ref = _hip.pack_inputs(inputs)
    model = HipModel(cfg, w, device=0, infer=True)
    eng = model.engine
    for mask_dtype in (np.bool_, np.float32):
        x = dict(inputs)
        x["atom_mask"] = np.asarray(inputs["atom_mask"]).astype(mask_dtype)
        x["neighbor_mask"] = np.asarray(inputs["neighbor_mask"]).astype(mask_dtype)
        rb = eng.upload_padded(x)
        assert (rb.packed.n_struct, rb.packed.n_atom, rb.packed.n_edge) == (ref.n_struct, ref.n_atom, ref.n_edge)
        got = eng.read_csr(rb)
        for f in ("atomic", "mol_offset", "edge_offset", "edge_col", "edge_dist", "edge_weight"):
            assert np.array_equal(got[f].view(np.int32), np.asarray(getattr(ref, f)).view(np.int32)), (case, mask_dtype, f)
        eng.forward_resident(rb, 0)
        y_dev, ga_dev = eng.download(rb)
        rb.free()
    y_host, ga_host = eng.forward(ref)
    assert np.array_equal(y_dev.view(np.int32), y_host.view(np.int32)) and np.array_equal(ga_dev.view(np.int32), ga_host.view(np.int32))
    # the drop-in call (scann_forward_padded packs on the device too): the same numbers, GlobalAttention scores re-padded with zeros
    y, ga = model.predict(inputs)
    # (bit patterns: the one-atom structure's score is the reference's own NaN, tf.linalg.normalize of a zero vector)
    assert np.array_equal(y[:, 0].view(np.int32), y_host.view(np.int32)) and np.array_equal(ga.view(np.int32), ref.repad_ga(ga_host).view(np.int32))


@pytest.mark.gpu
def test_device_packing_of_a_large_batch_uses_the_threaded_paths(hip_lib):
    """7,200 structures in ONE scann_upload_padded call: the mask pass runs on several threads (>= 2,048 structures) and so does the
    staging copy of the payload arrays (>= 8 MiB each) -- same CSR arrays as the host packer, same outputs as the chunked call."""
    from scann import _hip
    from scann.models.scann_model import HipModel

    cfg = so.default_config("qm9")
    model = HipModel(cfg, so.init_weights(cfg, 5, perturb=True), device=0)
    de, dn = so.synth_dataset(300, 13)
    small, _ = so.pad_batch(de, dn, True)
    inputs = {k: np.concatenate([np.asarray(v)] * 24) for k, v in small.items()}
    assert inputs["neighbors"].nbytes >= 8 << 20
    ref = _hip.pack_inputs(inputs)
    rb = model.engine.upload_padded(inputs)
    got = model.engine.read_csr(rb)
    for f in ("atomic", "mol_offset", "edge_offset", "edge_col", "edge_dist", "edge_weight"):
        assert np.array_equal(got[f].view(np.int32), np.asarray(getattr(ref, f)).view(np.int32)), f
    model.engine.forward_resident(rb, 0)
    y_one, _ = model.engine.download(rb, want_ga=False)
    rb.free()
    y = model.predict(inputs)  # 7,200 >= BIG_PREDICT: the chunked pipeline
    assert np.array_equal(y[:, 0].view(np.int32), y_one.view(np.int32))
    assert np.array_equal(y[:300], y[300:600])  # a structure's result does not depend on its batch


@pytest.mark.gpu
def test_device_packing_reports_what_the_host_packer_refuses(hip_lib):
    """An unmasked neighbour slot that points at a padded atom, an atomic number outside the embedding table: the host packer raises
    when it packs; a batch packed on the device reports the same at its download (no fault, no NaN, the handle stays usable)."""
    from scann import _hip
    from scann.models.scann_model import HipModel

    cfg = so.default_config("qm9")
    model = HipModel(cfg, so.init_weights(cfg, 3), device=0, infer=True)
    de, dn = so.synth_dataset(12, 1)
    inputs, _ = so.pad_batch(de, dn, True)
    na = np.asarray(inputs["atom_mask"]).reshape(len(de), -1).sum(1)
    M = inputs["neighbors"].shape[1]
    b = int(np.argmax(na < M))
    bad = dict(inputs)
    bad["neighbors"] = inputs["neighbors"].copy()
    bad["neighbors"][b, 0, 0] = M - 1  # a padded atom of that structure
    with pytest.raises(_hip.ScannHipError, match="padded atom"):
        model.predict(bad)
    bad = dict(inputs)
    bad["atomic"] = inputs["atomic"].copy()
    bad["atomic"][0, 0] = cfg["model"]["n_atoms"] + 3
    with pytest.raises(_hip.ScannHipError, match="embedding table"):
        model.predict(bad)
    y, _ = model.predict(inputs)
    assert np.isfinite(y).all()
    # the entry points that hand device-side tensors back without a download read the flag too (no sanitised stand-ins as data), and the
    # chunked pipeline names the chunk whose input was bad although the error surfaces chunks later
    eng = model.engine
    bad = dict(inputs)
    bad["neighbors"] = inputs["neighbors"].copy()
    bad["neighbors"][b, 0, 0] = M - 1
    rb = eng.upload_padded(bad)
    with pytest.raises(_hip.ScannHipError, match="scann_batch_read_csr.*padded atom"):
        eng.read_csr(rb)
    rb.free()
    big = {k: np.concatenate([v] * 200) for k, v in inputs.items()}  # 2,400 structures: the chunked path (>= 1,024)
    big["atomic"] = big["atomic"].copy()
    big["atomic"][1500, 0] = cfg["model"]["n_atoms"] + 3
    with pytest.raises(_hip.ScannHipError, match=r"embedding table.*\[structures (\d+)\.\.(\d+) of this call\]") as ei:
        model.predict(big)
    import re
    lo, hi = map(int, re.search(r"structures (\d+)\.\.(\d+)", str(ei.value)).groups())
    assert lo <= 1500 <= hi
    y, _ = model.predict(inputs)
    assert np.isfinite(y).all()
