"""importing the reference's Keras HDF5 checkpoints without h5py / TensorFlow.

Pins: (1) the pure-Python HDF5 reader against a committed file written by h5py / libhdf5 1.10.6; (2) the Keras-name ->
container-name mapping on a nested dict shaped like Keras' ``model_weights`` group; (3) when an interpreter with h5py is
around (this image: /opt/conda/bin/python3.9), a full container -> Keras-layout .h5 -> importer round trip for every
architecture switch.  None of this is a file written by TensorFlow itself (unavailable offline): see keras_import.py."""
import importlib.util
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import scann_oracle as so

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H5PY_PYTHON = next((p for p in ("/opt/conda/bin/python3.9", shutil.which("python3.9") or "") if p and os.path.exists(p) and
                    subprocess.run([p, "-c", "import h5py"], capture_output=True).returncode == 0), None)


def test_hdf5_lite_reads_a_file_written_by_libhdf5():
    from scann.utils.hdf5_lite import Dataset, File, Group

    spec = importlib.util.spec_from_file_location("make_h5_fixture", os.path.join(ROOT, "tests", "golden", "make_h5_fixture.py"))
    # the generator imports h5py at module level: take only its contents() (numpy + json) by executing it with h5py stubbed
    import sys
    import types
    had = sys.modules.get("h5py")
    sys.modules["h5py"] = types.ModuleType("h5py")
    try:
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        if had is None:
            del sys.modules["h5py"]
        else:
            sys.modules["h5py"] = had
    c = mod.contents()
    f = File(os.path.join(ROOT, "tests", "golden", "hdf5_lite_fixture.h5"))
    assert f.attrs["keras_version"] == "2.10.0" and f.attrs["model_config"] == c["config"] and f.attrs["backend"] == b"tensorflow"
    g = f["model_weights"]
    assert isinstance(g, Group) and [n.decode() for n in g.attrs["layer_names"]] == c["names"] and sorted(g.keys()) == sorted(c["names"])
    for n in c["names"]:
        d = g[n]["%s/sub/kernel:0" % n]
        assert isinstance(d, Dataset) and d.shape == (6, 5) and np.array_equal(d.read(), c["kernels"][n])
        assert [w.decode() for w in g[n].attrs["weight_names"]] == ["%s/sub/kernel:0" % n]
    assert np.array_equal(f["vec64"].read(), c["vec64"]) and f["vec64"].read().dtype == np.float64
    assert np.array_equal(f["ints"].read(), c["ints"]) and np.array_equal(f["be"].read(), np.asarray(c["be"], dtype="<f4"))
    assert f["scalar"].read() == 3.5
    assert len(f["empty_attr_holder"].attrs["weight_names"]) == 0
    assert len(list(g.visit_datasets())) == len(c["names"])
    with pytest.raises(KeyError):
        f["model_weights/nope"]


def test_hdf5_lite_chunked_datasets_and_named_refusals():
    """Second libhdf5-written pin (tests/golden/make_h5_fixture2.py): chunked datasets without filters are READ (chunk B-tree with
    several leaves, ragged edge chunks, big-endian elements, a dataset never written); what stays outside the reader -- a filter
    pipeline, a libver='latest' file -- is refused with a message that names the feature and the way out (h5repack)."""
    from scann.utils.hdf5_lite import File, Hdf5Error

    rng = np.random.default_rng(7)
    c = {"a": rng.normal(size=(37, 128)).astype(np.float32), "b": np.arange(1000, dtype=np.int64).reshape(10, 100) * 3 - 7,
         "c": rng.normal(size=(5, 3, 9)), "d": rng.normal(size=(300,)).astype(">f4")}
    f = File(os.path.join(ROOT, "tests", "golden", "hdf5_lite_chunked.h5"))
    for k, ref in c.items():
        got = f[k].read()
        assert got.shape == ref.shape and got.dtype.byteorder in "<=|" and np.array_equal(got, ref.astype(ref.dtype.newbyteorder("<"))), k
    assert np.array_equal(f["never_written"].read(), np.zeros((6, 4), np.float32))
    with pytest.raises(Hdf5Error, match=r"gz: dataset with a filter pipeline \(shuffle, deflate / gzip\).*h5repack"):
        f["gz"]
    with pytest.raises(Hdf5Error, match=r"superblock version 3 .*libver='latest'.*h5repack --low=0 --high=0 -l CONTI -f NONE"):
        File(os.path.join(ROOT, "tests", "golden", "hdf5_lite_latest.h5"))
    with pytest.raises(Hdf5Error, match="not an HDF5 file"):
        File(os.path.join(ROOT, "tests", "golden", "make_h5_fixture2.py"))


def _keras_nested(cfg, w):
    """The container `w` as the nested structure Keras' model_weights group has (layer -> [(weight name, array)]), with the
    auto-generated names a create_model build produces: global LayerNormalization / Dense counters, weightless layers."""
    m = cfg["model"]
    sfx = lambda stem, k: stem if k == 0 else "%s_%d" % (stem, k)  # noqa: E731
    layers = [("atomic", []), ("neighbors", [])]
    if "embed_atom/embeddings" in w:
        layers.append(("embed_atom", [("embed_atom/embeddings:0", w["embed_atom/embeddings"])]))
    else:
        layers.append(("embed_atom", [("embed_atom/kernel:0", w["embed_atom/kernel"]), ("embed_atom/bias:0", w["embed_atom/bias"])]))
    plain = lambda n: (n, [("%s/kernel:0" % n, w[n + "/kernel"]), ("%s/bias:0" % n, w[n + "/bias"])])  # noqa: E731
    if m["use_ring"]:
        layers.append(plain("extra_embed"))
    layers += [plain("dense_embed"), ("dropout", []), ("gaussian_expansion", [])]
    if m["g_update"]:
        layers += [plain("neighbor_d"), plain("neighbor_w"), ("geometry_features", [])]
    ln, dn = 3, 5  # counters need not start at 0: other models built earlier in the same process shift them
    for k in range(m["n_attention"]):
        name, p = sfx("local_attention", k), "local_attention_%d/" % k
        ws = []
        for sub in ("query", "key", "filter_geo"):
            ws += [("%s/%s/kernel:0" % (name, sub), w[p + sub + "/kernel"]), ("%s/%s/bias:0" % (name, sub), w[p + sub + "/bias"])]
        for part in (["layer_norm", "layer_norm_g"] if m["g_update"] else ["layer_norm"]):
            ws += [("%s/%s/gamma:0" % (name, sfx("layer_normalization", ln)), w[p + part + "/gamma"]),
                   ("%s/%s/beta:0" % (name, sfx("layer_normalization", ln)), w[p + part + "/beta"])]
            ln += 1
        layers.append((name, ws))
        if m["use_attn_norm"]:
            name, p = sfx("residual_norm", k), "residual_norm_%d/" % k
            ws = []
            for j in (1, 2):
                ws += [("%s/sequential/%s/kernel:0" % (name, sfx("dense", dn)), w[p + "dense_%d/kernel" % j]),
                       ("%s/sequential/%s/bias:0" % (name, sfx("dense", dn)), w[p + "dense_%d/bias" % j])]
                dn += 1
            ws += [("%s/%s/gamma:0" % (name, sfx("layer_normalization", ln)), w[p + "layer_norm/gamma"]),
                   ("%s/%s/beta:0" % (name, sfx("layer_normalization", ln)), w[p + "layer_norm/beta"])]
            ln += 1
            layers.append((name, ws))
    layers.append(plain("after_Lc"))
    layers.append(("global_attention", [("global_attention/%s/%s:0" % (s_, l_), w["global_attention/%s/%s" % (s_, l_)])
                                        for s_ in ("query", "key") for l_ in ("kernel", "bias")]))
    layers += [plain("bf_property"), plain("predict_property")]
    from collections import OrderedDict
    return OrderedDict(layers)


CASES = {
    "qm9_plus": dict(),
    "base": dict(g_update=False),
    "no_attn_norm": dict(use_attn_norm=False, use_ga_norm=False),
    "ring_cgcnn": dict(use_ring=True, feature="cgcnn"),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_keras_name_map_on_a_model_weights_shaped_dict(name):
    from scann.models.keras_import import infer_model_config, map_keras_weights

    cfg = so.default_config("qm9")
    cfg["model"].update(CASES[name], n_attention=3)
    w = so.init_weights(cfg, 11, perturb=True)
    got = map_keras_weights(_keras_nested(cfg, w))
    assert set(got) == set(w), set(got) ^ set(w)
    for k in w:
        assert got[k].dtype == np.float32 and np.array_equal(got[k], w[k]), k
    m, _ = infer_model_config(got)
    for key in ("n_attention", "g_update", "use_attn_norm", "use_ring", "local_dim", "global_dim", "dense_out", "embedding_dim"):
        assert m[key] == cfg["model"][key], key
    assert m["feature"] == cfg["model"].get("feature", "atomic")


def test_keras_name_map_rejects_foreign_layers():
    from scann.models.keras_import import map_keras_weights

    cfg = so.default_config("qm9")
    cfg["model"]["n_attention"] = 1
    layers = _keras_nested(cfg, so.init_weights(cfg, 1))
    layers["conv1d"] = [("conv1d/kernel:0", np.zeros((3, 3), np.float32))]
    with pytest.raises(ValueError, match="not part of the SCANN graph"):
        map_keras_weights(layers)


def _committed_fixture():
    spec = importlib.util.spec_from_file_location("make_keras_fixture", os.path.join(ROOT, "tests", "golden", "make_keras_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return os.path.join(ROOT, "tests", "golden", mod.NAME), mod.fixture_config(), mod.fixture_weights()


def test_committed_keras_layout_file_loads_without_h5py():
    """tests/golden/keras_layout_qm9_L2.h5 (written once by h5py in Keras' ModelCheckpoint layout, tests/golden/make_keras_fixture.py;
    scann_model.py:166-177) through the pure-Python reader and the Keras-name map: the seeded weights bit for bit, the architecture read
    from the file, every tensor mapped exactly once -- on any box, no h5py, no TensorFlow."""
    from scann.models.keras_import import load_keras_h5, mapping_report

    path, cfg, w = _committed_fixture()
    got_cfg, got = load_keras_h5(path, {"model": {"n_atoms": 10, "scale": 0.5}, "hyper": {"target": "homo"}})
    assert set(got) == set(w)
    for k in w:
        assert got[k].dtype == np.float32 and np.array_equal(got[k], w[k]), k
    for key in ("n_attention", "g_update", "use_attn_norm", "use_ga_norm", "use_ring", "feature", "gaussian_d", "embedding_dim", "num_head",
                "local_dim", "global_dim", "dense_out"):
        assert got_cfg["model"][key] == cfg["model"][key], (key, got_cfg["model"][key], cfg["model"][key])
    rep = mapping_report(path)
    assert rep["tensors"] == len(w) and rep["parameters"] == sum(int(v.size) for v in w.values()) == 310497 and set(rep["names"]) == set(w)
    r = subprocess.run([os.sys.executable, os.path.join(ROOT, "tools", "keras_h5_to_container.py"), "--check", path], capture_output=True, text=True)
    assert r.returncode == 0 and "CHECK OK" in r.stdout and "310497 parameters" in r.stdout, r.stderr[-500:]


@pytest.mark.skipif(H5PY_PYTHON is None, reason="no interpreter with h5py to write the Keras-layout file")
def test_committed_keras_layout_file_is_what_the_generator_writes(tmp_path):
    """(where h5py exists) the committed file is reproducible: regenerating it gives the same datasets."""
    from scann.models.keras_import import load_keras_h5

    path, cfg, w = _committed_fixture()
    npz, h5 = tmp_path / "m.npz", tmp_path / "m.h5"
    np.savez(npz, __config__=np.array(json.dumps(cfg)), **w)
    subprocess.run([H5PY_PYTHON, os.path.join(ROOT, "tools", "make_keras_h5_fixture.py"), str(npz), str(h5)], check=True)
    a, b = load_keras_h5(path, None)[1], load_keras_h5(str(h5), None)[1]
    assert set(a) == set(b) and all(np.array_equal(a[k], b[k]) for k in a)


@pytest.mark.skipif(H5PY_PYTHON is None, reason="no interpreter with h5py to write the Keras-layout file")
@pytest.mark.parametrize("name", sorted(CASES) + ["e_b"])
def test_container_to_keras_h5_and_back(tmp_path, name):
    """container (.npz) -> Keras-layout HDF5 written by h5py (tools/make_keras_h5_fixture.py) -> load_keras_h5: the same
    bits, and the architecture the file determines."""
    from scann.models.keras_import import load_keras_h5
    from scann.models.scann_model import normalize_config

    cfg = normalize_config(so.default_config("qm9"))
    cfg["model"].update(CASES.get(name, {}), n_attention=2, gaussian_d=5.0)
    if name == "e_b":
        cfg["hyper"]["target"] = "e_b"
    w = so.init_weights(cfg, 7, perturb=True)
    npz, h5 = tmp_path / "model.npz", tmp_path / "model_keras.h5"
    np.savez(npz, __config__=np.array(json.dumps(cfg)), **w)
    subprocess.run([H5PY_PYTHON, os.path.join(ROOT, "tools", "make_keras_h5_fixture.py"), str(npz), str(h5)], check=True)
    yaml_like = {"model": {"n_atoms": 10, "scale": 0.5}, "hyper": {"target": "homo", "batch_size": 128}}
    got_cfg, got = load_keras_h5(str(h5), yaml_like)
    assert set(got) == set(w)
    for k in w:
        assert np.array_equal(got[k], w[k]), k
    for key in ("n_attention", "g_update", "use_attn_norm", "use_ga_norm", "use_ring", "feature", "gaussian_d", "embedding_dim", "num_head"):
        assert got_cfg["model"][key] == cfg["model"][key], (key, got_cfg["model"][key], cfg["model"][key])
    assert got_cfg["hyper"]["target"] == ("e_b" if name == "e_b" else "homo") and got_cfg["hyper"]["batch_size"] == 128



def test_expected_param_count_matches_the_survey():
    from scann.models.keras_import import expected_param_count
    from scann.models.scann_model import normalize_config

    cfg = normalize_config(so.default_config("qm9"))
    assert expected_param_count(cfg["model"]) == 890977  # parameter count at the QM9 config
    for over in CASES.values():
        c = normalize_config(so.default_config("qm9"))
        c["model"].update(over, n_attention=3)
        w = so.init_weights(c, 1)
        assert expected_param_count(c["model"]) == sum(int(v.size) for v in w.values()), over


def test_sublayer_order_comes_from_the_name_path_not_the_file_order():
    """Keras numbers auto-named sub-layers globally (layer_normalization_7, dense_3): the importer orders an attention layer's two
    LayerNormalizations by that number, so a file that lists them the other way round is still mapped right (with a warning)."""
    from scann.models.keras_import import map_keras_weights

    cfg = so.default_config("qm9")
    cfg["model"]["n_attention"] = 2
    w = so.init_weights(cfg, 5, perturb=True)
    layers = _keras_nested(cfg, w)
    ws = layers["local_attention"]
    gb = [t for t in ws if t[0].endswith(("gamma:0", "beta:0"))]
    layers["local_attention"] = [t for t in ws if t not in gb] + gb[2:] + gb[:2]  # layer_norm_g's pair first
    with pytest.warns(UserWarning, match="creation order"):
        got = map_keras_weights(layers)
    for k in w:
        assert np.array_equal(got[k], w[k]), k


@pytest.mark.skipif(H5PY_PYTHON is None, reason="no interpreter with h5py to write the Keras-layout file")
def test_mapping_report_and_check_tool(tmp_path):
    """tools/keras_h5_to_container.py --check on a Keras-layout file: every tensor mapped exactly once, none left over, parameter
    count of the architecture; a file with a tensor too many fails."""
    from scann.models.keras_import import mapping_report
    from scann.models.scann_model import normalize_config

    cfg = normalize_config(so.default_config("qm9"))
    w = so.init_weights(cfg, 7, perturb=True)
    npz, h5 = tmp_path / "model.npz", tmp_path / "model_keras.h5"
    np.savez(npz, __config__=np.array(json.dumps(cfg)), **w)
    subprocess.run([H5PY_PYTHON, os.path.join(ROOT, "tools", "make_keras_h5_fixture.py"), str(npz), str(h5)], check=True)
    rep = mapping_report(str(h5))
    assert rep["tensors"] == len(w) and rep["parameters"] == 890977 and set(rep["names"]) == set(w)
    r = subprocess.run([os.sys.executable, os.path.join(ROOT, "tools", "keras_h5_to_container.py"), "--check", str(h5)],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "CHECK OK" in r.stdout and "890977 parameters" in r.stdout, r.stderr[-500:]


# ---- the day a checkpoint written by TensorFlow itself is at hand ---------------------------------------------------------------
# SCANN_REF_H5=<.../model_<target>.h5> [SCANN_REF_CONFIG=<.../config.yaml>] python -m pytest tests/test_keras_import.py -k reference_checkpoint

REF_H5, REF_CFG = os.environ.get("SCANN_REF_H5"), os.environ.get("SCANN_REF_CONFIG")


def _ref_config():
    if not REF_CFG:
        return None
    import yaml

    return yaml.safe_load(open(REF_CFG))


@pytest.mark.skipif(not REF_H5, reason="SCANN_REF_H5 not set: no reference-written checkpoint at hand (README.md:126, figshare)")
def test_reference_checkpoint_maps_completely():
    """A real ModelCheckpoint file (scann_model.py:166-177): every HDF5 tensor mapped exactly once, none left over, parameter
    count = the architecture the file implies; use_ring is read from the file (the published QM9 models take a 7th input
    `ring_aromatic`, qm9_pretrained.ipynb cell 5)."""
    from scann.models.keras_import import load_keras_h5, mapping_report

    rep = mapping_report(REF_H5, _ref_config())
    assert rep["tensors"] == len(rep["names"]) and rep["parameters"] > 0
    cfg, w = load_keras_h5(REF_H5, _ref_config())
    assert cfg["model"]["use_ring"] == ("extra_embed/kernel" in w)
    assert cfg["model"]["local_dim"] % cfg["model"]["num_head"] == 0  # (128 / 8: the MFMA kernels; anything else: csrc/scann_generic.hip)


# (the GPU half of the hook exists only when a file is named: the driver's GPU record then lists ONE skip, the 2-rank RCCL test on a
#  one-GPU box, and the placeholder stays visible as the skipped CPU test above)
def _reference_checkpoint_runs_a_finite_forward():
    from scann.models import SCANN
    from scann.models.keras_import import load_keras_h5

    cfg, _ = load_keras_h5(REF_H5, _ref_config())
    cfg.setdefault("hyper", {}).setdefault("target", "homo")
    sc = SCANN(cfg, pretrained=REF_H5, mode="infer")
    de, dn = so.synth_dataset(8, 0)
    inputs, _ = so.pad_batch(de, dn, cfg["model"]["g_update"])
    if cfg["model"]["use_ring"]:
        inputs["ring_aromatic"] = np.zeros(inputs["atomic"].shape + (2,), np.int32)
    y, ga = sc.model.predict(inputs)
    assert y.shape == (8, 1) and np.isfinite(y).all() and np.isfinite(ga).all()


if REF_H5:
    test_reference_checkpoint_runs_a_finite_forward = pytest.mark.gpu(_reference_checkpoint_runs_a_finite_forward)
