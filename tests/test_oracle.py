"""CPU tests of the oracle itself: committed golden vectors, an independent torch restatement, hand-computable
micro-cases and the invariances provable from the reference source (SURVEY.md 8c pins 1-3)."""
import importlib.util
import os

import numpy as np
import pytest

import scann_oracle as so

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)


def rel(got, ref):
    ref = np.asarray(ref, dtype=np.float64)
    scale = max(float(np.sqrt(np.mean(ref * ref))), 1e-30)
    return float(np.max(np.abs(np.asarray(got, dtype=np.float64) - ref) / np.maximum(np.abs(ref), scale)))


def test_parameter_count_matches_survey():
    assert so.count_params(so.default_config("qm9")) == 890977


@pytest.mark.parametrize("name", sorted(mg.CASES))
def test_oracle_reproduces_golden(name):
    cfg, w, inputs = mg.build(name)
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    assert mg.weights_digest(w) == str(z["weights_sha256"])
    for k, v in inputs.items():
        assert np.array_equal(v, z["in_" + k]), k
    y64, ga64 = so.forward(cfg, w, inputs, np.float64)
    np.testing.assert_allclose(y64, z["y64"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(ga64, z["ga64"], rtol=1e-12, atol=1e-14)
    y32, ga32 = so.forward(cfg, w, inputs, np.float32)
    assert rel(y32, z["y64"]) < 1e-4 and rel(ga32, z["ga64"]) < 1e-4  # fp32 restatement stays within the parity bar


@pytest.mark.parametrize("name", ["qm9_plus", "qm9_base", "qm9_no_norms", "qm9_e_b"])
def test_oracle_agrees_with_independent_torch_graph(name):
    """Same graph from stock torch ops on the packed layout (tests/torch_ref.py), fp64 both sides."""
    pytest.importorskip("torch")
    import torch_ref
    from scann import _hip

    cfg, w, inputs = mg.build(name)
    pk = _hip.pack_inputs(inputs)
    y_t, ga_t = torch_ref.forward_packed(cfg, w, pk, "float64")
    y, ga = so.forward(cfg, w, inputs, np.float64)
    amask = inputs["atom_mask"][..., 0]
    np.testing.assert_allclose(y, y_t, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(ga[..., 0][amask], ga_t, rtol=1e-9, atol=1e-12)


def test_gaussian_basis_hits_one_at_its_centre():
    c = np.linspace(0, 4.0, 20, dtype="float32")
    g = so.gaussian_expansion(c[None, None, :].astype(np.float64), c, np.dtype(np.float64))
    assert np.allclose(np.diagonal(g[0, 0]), 1.0)
    assert g[0, 0, 0, 1] == pytest.approx(np.exp(-(float(c[1]) ** 2) / 0.25))


def test_two_atom_global_attention_by_hand():
    cfg = {"use_ga_norm": True}
    rng = np.random.default_rng(0)
    w = {"global_attention/query/kernel": rng.normal(size=(4, 4)), "global_attention/query/bias": rng.normal(size=4),
         "global_attention/key/kernel": rng.normal(size=(4, 4)), "global_attention/key/bias": rng.normal(size=4)}
    x = rng.normal(size=(1, 3, 4))
    mask = np.array([[[1.0], [1.0], [0.0]]])
    attn, ctx = so.global_attention(w, cfg, x, mask, np.dtype(np.float64))
    q = x[0] @ w["global_attention/query/kernel"] + w["global_attention/query/bias"]
    k = x[0] @ w["global_attention/key/kernel"] + w["global_attention/key/bias"]
    a = np.array([k[0] @ q[1], k[1] @ q[0]])
    a = a / np.linalg.norm(a)
    p = np.exp(a - a.max())
    p /= p.sum()
    assert np.allclose(attn[0, :2, 0], p) and attn[0, 2, 0] == 0.0
    assert np.allclose(ctx[0], p[0] * k[0] + p[1] * k[1])


def test_padding_batch_and_permutation_invariance():
    cfg = so.default_config("qm9")
    cfg["model"]["n_attention"] = 2
    w = so.init_weights(cfg, 7, perturb=True)
    de, dn = so.synth_dataset(5, 9)
    inputs, _ = so.pad_batch(de, dn, True)
    y, ga = so.forward(cfg, w, inputs, np.float64)
    M = inputs["atomic"].shape[1]
    pad = {k: np.pad(v, [(0, 0), (0, 2)] + [(0, 3 if k.startswith("neighbor") and v.ndim == 3 else 0)] * (v.ndim - 2))
           for k, v in inputs.items()}
    y2, ga2 = so.forward(cfg, w, pad, np.float64)
    assert np.allclose(y, y2, rtol=1e-12) and np.allclose(ga, ga2[:, :M], rtol=1e-12)
    one, _ = so.pad_batch(de[3:4], dn[3:4], True)
    y3, _ = so.forward(cfg, w, one, np.float64)
    assert np.allclose(y3[0], y[3], rtol=1e-12)
    # garbage in masked slots
    g = {k: np.array(v) for k, v in inputs.items()}
    dead = ~g["neighbor_mask"]
    g["neighbor_distance"][dead] = 7.0
    g["neighbor_weight"][dead] = 5.0
    y4, ga4 = so.forward(cfg, w, g, np.float64)
    assert np.allclose(y, y4, rtol=1e-12) and np.allclose(ga, ga4, rtol=1e-12)


def test_fully_masked_atom_gives_layernorm_of_query():
    """attention.py:186-212: all slots masked -> uniform softmax, zeroed by the mask -> ctx = LN(q)."""
    cfg = so.default_config("qm9")["model"]
    w = so.init_weights({"model": cfg, "hyper": {}}, 3, perturb=True)
    rng = np.random.default_rng(1)
    B, M, N, d = 1, 3, 4, 128
    c = rng.normal(size=(B, M, d))
    idx = so.gather_shape(np.zeros((B, M, N), dtype=np.int64))
    geom = rng.normal(size=(B, M, N, d))
    mask = np.ones((B, M, N))
    mask[0, 1] = 0
    p = "local_attention_0/"
    attn, ctx, _ = so.local_attention(w, p, cfg, c, idx, geom, mask, None, np.dtype(np.float64))
    q = so.dense(c, w, p + "query", np.dtype(np.float64))
    want = so.layer_norm(q, w[p + "layer_norm/gamma"].astype(np.float64), w[p + "layer_norm/beta"].astype(np.float64))
    assert np.allclose(ctx[0, 1], want[0, 1], rtol=1e-12)
    # in fp32 the additive -1e9 absorbs the logit completely (ulp(1e9) = 64) -> exactly uniform attention
    attn32, ctx32, _ = so.local_attention(w, p, cfg, c.astype(np.float32), idx, geom.astype(np.float32),
                                          mask.astype(np.float32), None, np.dtype(np.float32))
    assert np.array_equal(attn32[0, :, 1], np.full((8, 4), 0.25, np.float32))
    assert np.allclose(ctx32[0, 1], want[0, 1], rtol=1e-4, atol=1e-5)


def test_single_atom_structure_is_nan_with_ga_norm():
    cfg = so.default_config("qm9")
    cfg["model"]["n_attention"] = 1
    w = so.init_weights(cfg, 1)
    inputs = {"atomic": np.array([[6]]), "atom_mask": np.array([[[True]]]), "neighbors": np.zeros((1, 1, 1), "int32"),
              "neighbor_mask": np.zeros((1, 1, 1), bool), "neighbor_weight": np.zeros((1, 1, 1), "float32"),
              "neighbor_distance": np.zeros((1, 1, 1), "float32")}
    y, _ = so.forward(cfg, w, inputs, np.float32)
    assert np.isnan(y).all()


def test_c_port_matches_numpy_oracle():
    """The C/OpenMP port (cpu_baseline of bench.py) is the same algorithm as the NumPy oracle."""
    import subprocess

    here = os.path.dirname(HERE)
    subprocess.check_call(["make", "-s", "-C", os.path.join(here, "oracle")])
    import scann_oracle_c as soc

    for name, kind, n in (("qm9", "qm9", 9), ("mp2018", "mp2018", 3)):
        cfg = so.default_config(name)
        cfg["model"]["n_attention"] = 3
        w = so.init_weights(cfg, 11, perturb=True)
        de, dn = so.synth_dataset(n, 21, kind)
        inputs, _ = so.pad_batch(de, dn, True)
        y, ga = soc.forward(cfg, w, inputs)
        y64, ga64 = so.forward(cfg, w, inputs, np.float64)
        assert rel(y, y64) < 5e-5 and rel(ga, ga64) < 5e-5
