"""GPU tests of the training step: hand-written backward vs torch autograd (fp64, CPU) of the independent torch
graph (tests/torch_ref.py), Adam + l2 regulariser vs a NumPy restatement of tf.keras Adam, dropout statistics."""
import os

import numpy as np
import pytest

import scann_oracle as so

pytestmark = pytest.mark.gpu


def setup(n=6, L=2, seed=1, **model_over):
    from scann import _hip
    from scann.models.scann_model import HipModel

    cfg = so.default_config("qm9")
    cfg["model"]["n_attention"] = L
    cfg["model"].update(model_over)
    w = so.init_weights(cfg, 3, perturb=True)
    de, dn = so.synth_dataset(n, seed)
    inputs, targets = so.pad_batch(de, dn, cfg["model"]["g_update"])
    pk = _hip.pack_inputs(inputs)
    model = HipModel(cfg, w, device=0)
    return cfg, w, pk, targets, model


GRAD_FLOOR, GRAD_SLACK, GRAD_CAP = 2e-5, 4.0, 2e-4


def check_grads(got, cfg, w, pk, targets, attn_scale=None, cap=None, drop=None):
    """Gradient parity rule.  Reference: fp64 autograd of the independent torch graph.  A single-precision step cannot be closer
    to it than the SAME graph run by torch in fp32 is, so every tensor is held to max(GRAD_FLOOR, GRAD_SLACK x that fp32
    error), and to GRAD_CAP overall.  (Measured, tests/manual/grad_floor.py: HIP errors 4e-7 ... 1.5e-5 of the tensor scale,
    1 - 2.2 x the torch-fp32 floor; the 2e-3 bound of round 1 was three orders looser than the implementation needs.)
    Returns (rmse of the fp64 graph, errors)."""
    import torch_ref

    _, rmse, ref, _ = torch_ref.loss_and_grads(cfg, w, pk, targets, attn_scale=attn_scale, drop=drop)
    _, _, g32, _ = torch_ref.loss_and_grads(cfg, w, pk, targets, attn_scale=attn_scale, dtype="float32", drop=drop)
    for k in ref:  # the autograd loss includes the l2 term; the library adds 2*l2*W inside the optimiser step -> remove it here
        if k.endswith(torch_ref.REGULARIZED):
            ref[k] = ref[k] - 2e-4 * w[k].astype(np.float64)
            g32[k] = g32[k] - 2e-4 * w[k].astype(np.float64)
    e_gpu, e_32 = grad_errors(got, ref), grad_errors(g32, ref)
    bad = {k: (e_gpu[k], e_32[k]) for k in ref if not e_gpu[k] <= min(cap or GRAD_CAP, max(GRAD_FLOOR, GRAD_SLACK * e_32[k]))}
    assert not bad, bad
    return rmse, e_gpu


def grad_errors(got, ref):
    out = {}
    for k, r in ref.items():
        scale = max(float(np.sqrt(np.mean(r * r))), 1e-12)
        out[k] = float(np.max(np.abs(got[k].astype(np.float64) - r)) / max(float(np.abs(r).max()), scale))
    return out


def test_mrelu_backward_is_the_identity(hip_lib):
    """target="e_b": predict_property goes through mrelu (scann_model.py:446), max(x, 0) forward with an IDENTITY gradient
    (custom_layers.py:6-15) -- structures whose output is clipped to 0 still push their full error back.  The output bias is
    shifted so that about half of the pre-activations are negative."""
    import torch_ref
    from scann.models.scann_model import HipModel

    cfg, w, pk, targets, model = setup(n=12)
    y_pre, _ = torch_ref.forward_packed(cfg, w, pk)
    w = dict(w)
    w["predict_property/bias"] = (w["predict_property/bias"] - np.median(y_pre)).astype(np.float32)
    cfg["hyper"]["target"] = "e_b"
    model = HipModel(cfg, w, device=0)
    y = model.predict(pk)
    assert (y == 0).sum() >= 3 and (y > 0).sum() >= 3, y.ravel()
    y_ref, _ = torch_ref.forward_packed(cfg, w, pk)
    assert np.allclose(y, y_ref, rtol=1e-4, atol=1e-5)
    eng = model.engine
    eng.train_begin()
    rb = eng.upload(pk)
    sse = eng.train_forward(rb, targets)
    eng.zero_grads()
    eng.train_backward(rb, sse, pk.n_struct)
    got = eng.get_grads()
    rmse, _ = check_grads(got, cfg, w, pk, targets)
    assert abs(np.sqrt(sse / pk.n_struct) - rmse) <= 1e-5 * max(rmse, 1e-6)
    # the clipped structures DO contribute: with torch.relu's zero gradient there the head gradient would differ visibly
    torch = pytest.importorskip("torch")

    W = {k: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=True) for k, v in w.items()}
    cfg_lin = {"model": cfg["model"], "hyper": dict(cfg["hyper"], target="homo")}
    yy, _ = torch_ref.forward_packed(cfg_lin, W, pk, as_tensor=True)
    t = torch.tensor(np.asarray(targets), dtype=torch.float64).reshape(-1, 1)
    torch.sqrt(torch.mean((torch.relu(yy) - t) ** 2)).backward()
    g_zero = W["predict_property/kernel"].grad.numpy()
    scale = float(np.abs(g_zero).max())
    assert float(np.abs(got["predict_property/kernel"].reshape(g_zero.shape) - g_zero).max()) > 1e-2 * scale
    rb.free()


@pytest.mark.parametrize("over", [dict(), dict(use_attn_norm=False), dict(use_ga_norm=False), dict(g_update=False)],
                         ids=["qm9", "no_attn_norm", "no_ga_norm", "base"])
def test_gradients_match_autograd(hip_lib, over):
    import torch_ref

    cfg, w, pk, targets, model = setup(**over)
    eng = model.engine
    eng.train_begin()
    rb = eng.upload(pk)
    sse = eng.train_forward(rb, targets)
    eng.zero_grads()
    eng.train_backward(rb, sse, pk.n_struct)
    got = eng.get_grads()
    rmse, _ = check_grads(got, cfg, w, pk, targets)
    assert abs(np.sqrt(sse / pk.n_struct) - rmse) <= 1e-5 * max(rmse, 1e-6)
    rb.free()


def test_seven_layer_gradients_and_accumulation(hip_lib):
    """Full QM9 depth; two backward calls accumulate (gradient of the sum)."""
    import torch_ref

    cfg, w, pk, targets, model = setup(n=5, L=7, seed=4)
    eng = model.engine
    eng.train_begin()
    rb = eng.upload(pk)
    sse = eng.train_forward(rb, targets)
    eng.zero_grads()
    eng.train_backward(rb, sse, pk.n_struct)
    g1 = eng.get_grads()
    eng.train_backward(rb, sse, pk.n_struct)
    g2 = eng.get_grads()
    check_grads(g1, cfg, w, pk, targets)
    for k in g1:
        assert np.allclose(g2[k], 2 * g1[k], rtol=1e-5, atol=1e-9 + 1e-6 * np.abs(g1[k]).max())  # the reduce adds a second, identical sum
    rb.free()


def test_adam_step_matches_keras_formula(hip_lib):
    """tf.keras Adam (epsilon outside the sqrt) with the l2(1e-4) regulariser gradient; forward afterwards uses the
    updated (re-packed) weights."""
    import torch_ref

    cfg, w, pk, targets, model = setup()
    eng = model.engine
    eng.train_begin()
    rb = eng.upload(pk)
    m = {k: np.zeros_like(v, dtype=np.float64) for k, v in w.items()}
    v = {k: np.zeros_like(x, dtype=np.float64) for k, x in w.items()}
    wref = {k: x.astype(np.float64) for k, x in w.items()}
    lr, b1, b2, eps = 5e-4, 0.9, 0.999, 1e-7
    for step in range(1, 4):
        sse = eng.train_forward(rb, targets)
        eng.zero_grads()
        eng.train_backward(rb, sse, pk.n_struct)
        g = eng.get_grads()
        lr_t = lr / (1 + 1e-5 * (step - 1))  # legacy `decay`
        eng.adam_step(lr_t)
        for k in wref:
            gi = g[k].astype(np.float64) + (2e-4 * wref[k] if k.endswith(torch_ref.REGULARIZED) else 0.0)
            m[k] = b1 * m[k] + (1 - b1) * gi
            v[k] = b2 * v[k] + (1 - b2) * gi * gi
            wref[k] -= lr_t * np.sqrt(1 - b2 ** step) / (1 - b1 ** step) * m[k] / (np.sqrt(v[k]) + eps)
    got = eng.get_weights()
    for k in wref:
        assert np.allclose(got[k], wref[k], rtol=1e-4, atol=2e-6), k
    # the inference path sees the new weights
    y_new, _ = eng.forward(pk)
    y_fresh = type(model)(cfg, got, device=0).predict(pk)  # a fresh model loaded with the updated parameters
    assert np.allclose(y_new, y_fresh[:, 0], rtol=1e-5, atol=1e-6)
    rb.free()


def test_loss_decreases_and_dropout_is_active(hip_lib):
    cfg, w, pk, targets, model = setup(n=16, L=2, seed=9)
    eng = model.engine
    eng.train_begin()
    rb = eng.upload(pk)
    s0 = eng.train_forward(rb, targets)
    s_drop_a = eng.train_forward(rb, targets, dropout=0.1, seed=1)
    s_drop_b = eng.train_forward(rb, targets, dropout=0.1, seed=2)
    s_drop_a2 = eng.train_forward(rb, targets, dropout=0.1, seed=1)
    assert s_drop_a != s0 and s_drop_a != s_drop_b and s_drop_a == s_drop_a2  # masks depend on the seed only
    # a learnable target: standardised carbon count of each molecule
    nC = np.array([np.sum(pk.atomic[pk.mol_offset[i]:pk.mol_offset[i + 1]] == 6) for i in range(pk.n_struct)], dtype=np.float64)
    t2 = ((nC - nC.mean()) / (nC.std() + 1e-9)).astype(np.float32)
    s_start = eng.train_forward(rb, t2)
    for step in range(80):
        sse = eng.train_forward(rb, t2, dropout=0.1, seed=100 + step)
        eng.zero_grads()
        eng.train_backward(rb, sse, pk.n_struct)
        eng.adam_step(2e-3 / (1 + 1e-5 * step))
    final = eng.train_forward(rb, t2)
    assert final < 0.25 * s_start, (s_start, final)
    rb.free()


def _write_dataset(tmp_path, n=96, seed=5):
    """A dataset in the reference's on-disk format (general.py:104-144) with a learnable target (carbon fraction)."""
    de, dn = so.synth_dataset(n, seed)
    full = np.empty(n, dtype=object)
    for i in range(n):
        Z = de[i][0]
        full[i] = {"Atomic": Z, "Properties": {"homo": float(np.mean(np.asarray(Z) == 6) * 5.0 - 2.0)}}
    np.save(tmp_path / "data_energy.npy", full, allow_pickle=True)
    np.save(tmp_path / "data_nei.npy", dn, allow_pickle=True)
    return str(tmp_path / "data_energy.npy"), str(tmp_path / "data_nei.npy")


@pytest.mark.parametrize("widths", ["128x8", "64x4"])
def test_scann_train_evaluate_roundtrip(hip_lib, tmp_path, widths):
    """SCANN.prepare_dataset -> train -> evaluate like train.py does: checkpoint, config.yaml, report.txt, hist_data.npy;
    then infer mode from the checkpoint (predict_model.py path) reproduces the evaluation predictions.  Once at the shipped widths
    (MFMA kernels) and once at local_dim 64 / 4 heads / global_dim 96 / dense_out 32 (the plain-fp32 kernels)."""
    import yaml
    from scann.models import SCANN

    e_path, n_path = _write_dataset(tmp_path)
    cfg = so.default_config("qm9")
    cfg["model"]["n_attention"] = 2
    if widths == "64x4":
        cfg["model"].update(OTHER_WIDTHS["64x4"])
    cfg["hyper"].update(batch_size=16, test_percent=0.125, scaler=True, scheduler="cosine", train_size="", test_size="",
                        data_size=96, data_nei_path=n_path, data_energy_path=e_path, lr=2e-3, min_lr=2e-4,
                        save_path=str(tmp_path / "run"), pretrained="", use_ref=False, target="homo")
    np.random.seed(0)
    scann = SCANN(cfg, "")
    scann.prepare_dataset()
    scann.train(epochs=12)
    h = scann.hist.history
    assert len(h["val_mae"]) == 12 and h["mae"][-1] < 0.6 * h["mae"][0], h["mae"]
    mae, r2 = scann.evaluate()  # reloads the best checkpoint, like the reference after training
    out = str(tmp_path / "run_homo")
    assert os.path.exists(out + "/models/model_homo.h5") and os.path.exists(out + "/config.yaml")
    rep = open(out + "/report.txt").read()
    assert "Training MAE" in rep and "Test MAE" in rep and os.path.exists(out + "/hist_data.npy")
    saved = yaml.safe_load(open(out + "/config.yaml"))
    assert "target_mean" in saved["hyper"] and saved["hyper"]["data_size"] == 96
    infer = SCANN(saved, out + "/models/model_homo.h5", mode="infer")
    inputs, tgt = scann.testIter[0]
    y, ga = infer.predict_data(inputs)
    assert y.shape == (len(tgt), 1) and ga.shape[:2] == inputs["atomic"].shape
    assert abs(np.mean(np.abs((y[:, 0] - infer.mean) / infer.std - tgt)) * infer.std - mae) < 0.5 * mae + 1e-3


def test_rccl_single_rank_communicator(hip_lib):
    """world_size 1 RCCL communicator: the all-reduce entry points run and leave the gradients unchanged."""
    from scann import _hip

    cfg, w, pk, targets, model = setup(n=4)
    eng = model.engine
    eng.train_begin()
    assert eng.comm_ranks() == 0  # no communicator yet
    eng.comm_init(_hip.comm_unique_id(), 0, 1)
    assert eng.comm_ranks() == 1  # ncclCommCount of the live communicator: what bench.py --train prints as rccl_ranks
    rb = eng.upload(pk)
    sse = eng.train_forward(rb, targets)
    s2, c2 = eng.allreduce_sse(sse, pk.n_struct)
    assert (s2, c2) == (sse, pk.n_struct)
    eng.zero_grads()
    eng.train_backward(rb, sse, pk.n_struct)
    g1 = eng.get_grads()
    eng.allreduce_grads()
    g2 = eng.get_grads()
    assert all(np.array_equal(g1[k], g2[k]) for k in g1)
    rb.free()


def test_cli_train_then_predict_model(hip_lib, tmp_path):
    """train.py (12 epochs) then predict_model.py on its output directory, as subprocesses like a user would run them."""
    import subprocess
    import sys

    import yaml

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e_path, n_path = _write_dataset(tmp_path, n=64)
    cfg = so.default_config("qm9")
    cfg["model"]["n_attention"] = 1
    cfg["hyper"].update(batch_size=16, test_percent=0.125, scaler=True, scheduler="sgdr", train_size="", test_size="",
                        data_size=64, data_nei_path=n_path, data_energy_path=e_path, lr=2e-3, min_lr=2e-4,
                        save_path=str(tmp_path / "cli"), pretrained="")
    cfg["model"].pop("feature"); cfg["model"].pop("use_drop"); cfg["hyper"].pop("target")  # the CLI injects these
    ypath = tmp_path / "cfg.yaml"
    yaml.safe_dump(cfg, open(ypath, "w"))
    r = subprocess.run([sys.executable, os.path.join(root, "train.py"), "homo", str(ypath), "--epochs", "3"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = str(tmp_path / "cli_homo")
    assert "Test MAE" in open(out + "/report.txt").read()
    r = subprocess.run([sys.executable, os.path.join(root, "predict_model.py"), out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert os.path.exists(out + "/ga_scores_homo.pickle") and os.path.exists(out + "/energy_pre_homo.pickle")
    # the pickles in the reference's shapes (predict_model.py:60-92): one [M, 1] GA array per structure, M = the largest structure
    # of ITS batch of 16 (zeros behind the structure's own atoms), scores of a structure sum to 1; [targets, predictions] lists
    import pickle

    ga = pickle.load(open(out + "/ga_scores_homo.pickle", "rb"))
    y, pred = pickle.load(open(out + "/energy_pre_homo.pickle", "rb"))
    data = np.load(e_path, allow_pickle=True)
    sizes = [len(d["Atomic"]) for d in data]
    assert len(ga) == len(y) == len(pred) == 64
    for b0 in range(0, 64, 16):
        m = max(sizes[b0:b0 + 16])
        for i in range(b0, b0 + 16):
            assert ga[i].shape == (m, 1) and abs(float(ga[i][:sizes[i]].sum()) - 1.0) < 1e-4 and not ga[i][sizes[i]:].any()
    assert np.isfinite(np.asarray(pred, dtype=np.float64)).all()


def drop_scale_np(seed, tag, idx, p):
    import torch_ref

    return torch_ref.drop_scale_np(seed, tag, idx, p)


@pytest.mark.parametrize("g_update", [True, False], ids=["scann_plus", "base"])
def test_attention_dropout_gradients(hip_lib, g_update):
    """use_drop: the GPU's counter-based attention mask, rebuilt on the host, is fed to the torch graph; gradients match
    (both LocalAttention branches run the same softmax code)."""
    import torch_ref

    cfg, w, pk, targets, model = setup(n=6, L=2, seed=2, g_update=g_update)
    eng = model.engine
    eng.train_begin()
    eng.set_attention_dropout(0.3)  # large rate so that many weights are actually dropped
    rb = eng.upload(pk)
    seed = 12345
    sse = eng.train_forward(rb, targets, dropout=0.0, seed=seed)
    eng.zero_grads()
    eng.train_backward(rb, sse, pk.n_struct)
    got = eng.get_grads()
    idx = np.arange(pk.n_edge * 8, dtype=np.uint64)
    scales = [drop_scale_np(seed, 2000 + l, idx, 0.3).reshape(pk.n_edge, 8) for l in range(2)]
    assert 0.2 < np.mean(scales[0] == 0) < 0.4
    rmse, _ = check_grads(got, cfg, w, pk, targets, attn_scale=scales)
    assert abs(np.sqrt(sse / pk.n_struct) - rmse) <= 2e-5 * max(rmse, 1e-6)
    eng.set_attention_dropout(0.0)
    rb.free()


@pytest.mark.parametrize("ring,cgcnn", [(True, False), (False, True), (True, True)], ids=["ring", "cgcnn", "ring+cgcnn"])
def test_ring_cgcnn_embedding_gradients(hip_lib, ring, cgcnn):
    """Backward of the general embedding path (use_ring / feature="cgcnn", scann_model.py:361-374)."""
    import torch_ref
    from scann import _hip
    from scann.models.scann_model import HipModel

    cfg = so.default_config("qm9")
    cfg["model"].update(n_attention=1, use_ring=ring, feature="cgcnn" if cgcnn else "atomic")
    w = so.init_weights(cfg, 8, perturb=True)
    de, dn = so.synth_dataset(6, 3, use_ring=ring)
    inputs, targets = so.pad_batch(de, dn, True, use_ring=ring)
    if cgcnn:
        table = np.random.default_rng(5).integers(0, 2, size=(101, 92)).astype("float32")
        inputs["atomic"] = table[inputs["atomic"]]
    pk = _hip.pack_inputs(inputs)
    model = HipModel(cfg, w, device=0)
    eng = model.engine
    eng.train_begin()
    rb = eng.upload(pk)
    sse = eng.train_forward(rb, targets)
    eng.zero_grads()
    eng.train_backward(rb, sse, pk.n_struct)
    got = eng.get_grads()
    rmse, _ = check_grads(got, cfg, w, pk, targets)
    assert abs(np.sqrt(sse / pk.n_struct) - rmse) <= 2e-5 * max(rmse, 1e-6)
    eng.adam_step(1e-3)  # also exercises the re-pack without the species LUT
    s2 = eng.train_forward(rb, targets)
    assert np.isfinite(s2)
    rb.free()


def test_gradients_mp2018_shapes(hip_lib):
    """Crystal-shaped batch (up to 24 neighbours per atom): exercises the general attention backward (degree > 16)."""
    import torch_ref
    from scann import _hip
    from scann.models.scann_model import HipModel

    cfg = so.default_config("mp2018")
    cfg["model"]["n_attention"] = 2
    w = so.init_weights(cfg, 21, perturb=True)
    de, dn = so.synth_dataset(4, 6, "mp2018")
    inputs, targets = so.pad_batch(de, dn, True)
    pk = _hip.pack_inputs(inputs)
    assert np.diff(pk.edge_offset).max() > 16
    model = HipModel(cfg, w, device=0)
    eng = model.engine
    eng.train_begin()
    rb = eng.upload(pk)
    sse = eng.train_forward(rb, targets)
    eng.zero_grads()
    eng.train_backward(rb, sse, pk.n_struct)
    got = eng.get_grads()
    rmse, _ = check_grads(got, cfg, w, pk, targets)
    rb.free()


def test_gradients_with_more_than_64_neighbours(hip_lib):
    """Atoms with 70 and 130 neighbours (chunk tiles + softmax merge in the keep-mode forward, the degree-agnostic attention
    backward) next to ordinary molecules: gradients against autograd."""
    import torch_ref
    from scann import _hip

    cfg, w, _, _, model = setup(n=2, L=2, seed=3)
    rng = np.random.default_rng(4)
    A = 140
    deg = {0: 70, 5: 130, 139: 65}
    nb = []
    for a in range(A):
        d = deg.get(a, int(rng.integers(1, 7)))
        js = rng.choice(np.delete(np.arange(A), a), d, replace=False)
        nb.append([[6, int(j), float(rng.uniform(0.4, 3.5)), 1.0, float(rng.uniform(0.9, 4.0))] for j in js])
    de, dn = so.synth_dataset(2, 3)
    de3, dn3 = np.empty(3, dtype=object), np.empty(3, dtype=object)
    de3[0], dn3[0] = de[0], dn[0]
    de3[1], dn3[1] = [[int(z) for z in rng.choice([1, 6, 7, 8], A)], 0.3], nb
    de3[2], dn3[2] = de[1], dn[1]
    inputs, targets = so.pad_batch(de3, dn3, True)
    pk = _hip.pack_inputs(inputs)
    eng = model.engine
    eng.train_begin()
    rb = eng.upload(pk)
    sse = eng.train_forward(rb, targets)
    eng.zero_grads()
    eng.train_backward(rb, sse, pk.n_struct)
    got = eng.get_grads()
    rmse, _ = check_grads(got, cfg, w, pk, targets)
    assert abs(np.sqrt(sse / pk.n_struct) - rmse) <= 1e-5 * max(rmse, 1e-6)
    rb.free()


_RCCL_WORKER = r'''
import os, sys
import numpy as np
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), os.path.join(ROOT, "oracle")]
import scann_oracle as so
from scann import _hip
from scann.models.scann_model import HipModel, normalize_config
from scann.models.trainer import Communicator
from scann.parallel import rank_slice, slice_packed

rank = int(os.environ["RANK"])
cfg = normalize_config(so.default_config("qm9")); cfg["model"]["n_attention"] = 2
model = HipModel(cfg, device=int(os.environ["LOCAL_RANK"]), seed=100 + rank)   # DIFFERENT initialiser draw per rank
eng = model.engine
eng.train_begin()
comm = Communicator(eng)                                                       # ncclCommInitRank(world=2) + weight broadcast
assert comm.world == 2
w = eng.get_weights()
digests = comm.rdzv.allgather(float(sum(np.abs(v).sum() for v in w.values())))
assert digests[0] == digests[1], digests                                        # every replica now holds rank 0's parameters
de, dn = so.synth_dataset(10, 3)
inputs, tgt = so.pad_batch(de, dn, True)
pk = _hip.pack_inputs(inputs)
lo, hi = rank_slice(pk.n_struct, rank, 2)
rb = eng.upload(slice_packed(pk, lo, hi))
sse = eng.train_forward(rb, np.asarray(tgt, np.float32)[lo:hi])
sse_g, cnt_g = comm.sum_pair(sse, hi - lo)                                      # global RMSE (losses.py:5-6)
eng.zero_grads(); eng.train_backward(rb, sse_g, cnt_g); eng.allreduce_grads()
g = eng.get_grads()
if rank == 0:                                                                   # the same step on ONE rank with the whole batch
    ref = HipModel(cfg, w, device=0).engine
    ref.train_begin()
    rb2 = ref.upload(pk)
    sse1 = ref.train_forward(rb2, np.asarray(tgt, np.float32))
    ref.zero_grads(); ref.train_backward(rb2, sse1, pk.n_struct)
    g1 = ref.get_grads()
    assert cnt_g == pk.n_struct and abs(sse_g - sse1) <= 1e-5 * sse1, (sse_g, sse1)
    for k in g1:
        scale = max(float(np.sqrt(np.mean(g1[k].astype(np.float64) ** 2))), 1e-12)
        assert float(np.max(np.abs(g[k] - g1[k]))) <= 2e-5 * scale + 1e-9, k  # same sums, different association: rounding only
# the same optimisation step through scann_train_step (all-reduces on the stream, loss scale formed on the device)
sse_s, cnt_s = eng.train_step(rb, np.asarray(tgt, np.float32)[lo:hi], 1e-3, dropout=0.0, seed=1)
ws = eng.get_weights()
digests = comm.rdzv.allgather(float(sum(np.abs(v.astype(np.float64)).sum() for v in ws.values())))
assert digests[0] == digests[1], digests                                        # the replicas stay one model
if rank == 0:
    assert cnt_s == pk.n_struct and abs(sse_s - sse1) <= 1e-5 * sse1, (sse_s, sse1)
    ref.zero_grads(); ref.train_backward(rb2, sse1, pk.n_struct); ref.adam_step(1e-3)
    w1 = ref.get_weights()
    for k in w1:
        assert np.allclose(ws[k], w1[k], rtol=1e-4, atol=2e-6), k
    print("RCCL2_OK")
comm.rdzv.barrier()
'''


def test_two_rank_rccl_gradients_match_single_rank(hip_lib, tmp_path):
    """configs[2] with a REAL world-size-2 RCCL communicator: two processes, one device each; the ranks start from different
    initialiser draws (the broadcast makes them one model), all-reduced gradients and the global RMSE equal the single-rank
    full-batch step.  Needs two GPUs: skipped on a one-GPU box (RCCL refuses two ranks on one device)."""
    import os

    if hip_lib.scann_device_count() < 2:
        pytest.skip("needs 2 GPUs")
    from scann.parallel import spawn_ranks

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rccl_worker.py"
    script.write_text("ROOT = %r\n" % root + _RCCL_WORKER)
    assert spawn_ranks([str(script)], 2, timeout=600) == 0


def test_validation_forward_has_no_attention_dropout(hip_lib, tmp_path):
    """use_drop: Dropout(0.05) on the attention weights is a TRAINING-mode layer (attention.py:115-116,191).  fit() must run
    its validation passes without it: val_mae drives ModelCheckpoint / EarlyStopping / SGDR."""
    from scann.models import trainer

    cfg, w, pk, targets, model = setup(n=6)
    eng = model.engine
    eng.train_begin()
    calls = []
    orig = eng.set_attention_dropout
    eng.set_attention_dropout = lambda p: (calls.append(p), orig(p))[1]

    class It:
        def __len__(self):
            return 1

        def __getitem__(self, i):
            return pk, targets

    class S:
        pass

    s = S()
    cfg["model"]["use_drop"] = True
    cfg["hyper"].update(save_path=str(tmp_path / "run"), target="homo", lr=1e-4, min_lr=1e-5, scheduler="cosine")
    s.config, s.model, s.trainIter, s.validIter = cfg, model, It(), It()
    hist = trainer.fit(s, epochs=1, verbose=False)
    assert calls == [0.05, 0.0], calls  # training epoch with the rate, validation epoch without
    rb = eng.upload(pk)
    sse_val = eng.train_forward(rb, targets, dropout=0.0, seed=9)   # what the validation pass computed ...
    y_inf, _ = model.engine.forward(pk, want_ga=False)              # ... equals the inference forward
    assert abs(sse_val - float(np.sum((y_inf - targets) ** 2))) <= 1e-4 * max(sse_val, 1e-6)
    assert np.isfinite(hist["val_mae"][0])
    rb.free()


def test_weight_gradients_are_bit_reproducible(hip_lib):
    """Dense-kernel, bias and LayerNorm gradients are two-stage reductions with a fixed order (per-slab partial slots + one
    reduce launch): two backward passes over the same batch give the same BITS.  The small tensors that still end in float
    atomics (embedding table, K = 20 basis filters, the 128-wide output head) are compared to rounding only."""
    cfg, w, pk, targets, model = setup(n=48, L=3, seed=4)
    eng = model.engine
    eng.train_begin()
    rb = eng.upload(pk)
    grads = []
    for _ in range(2):
        sse = eng.train_forward(rb, targets, dropout=0.1, seed=5)
        eng.zero_grads()
        eng.train_backward(rb, sse, pk.n_struct)
        grads.append(eng.get_grads())
    atomic = ("embed_atom/", "dense_embed/", "neighbor_d/", "neighbor_w/", "predict_property/")
    for k in grads[0]:
        if k.startswith(atomic):
            assert np.allclose(grads[0][k], grads[1][k], rtol=1e-4, atol=1e-7), k
        else:
            assert np.array_equal(grads[0][k], grads[1][k]), k
    rb.free()


def test_train_step_equals_the_split_sequence(hip_lib):
    """scann_train_step (one asynchronous sequence, loss scale formed on the device) ends in the same weights as
    train_forward / zero_grads / train_backward / adam_step called one by one, and reports the same SSE."""
    from scann.models.scann_model import HipModel

    cfg, w, pk, targets, _ = setup(n=24, L=3, seed=12)
    models = [HipModel(cfg, w, device=0) for _ in range(2)]
    rbs = []
    for m in models:
        m.engine.train_begin()
        rbs.append(m.engine.upload(pk))
    a, b = models[0].engine, models[1].engine
    for step in range(3):
        lr_t = 1e-3 / (1 + 1e-5 * step)
        sse = a.train_forward(rbs[0], targets, dropout=0.1, seed=step)
        a.zero_grads()
        a.train_backward(rbs[0], sse, pk.n_struct)
        a.adam_step(lr_t)
        sse_b, cnt_b = b.train_step(rbs[1], targets, lr_t, dropout=0.1, seed=step)
        # step 0 starts from identical weights; later steps differ by the rounding of the few float-atomic gradient tensors
        assert cnt_b == pk.n_struct and (sse_b == sse if step == 0 else abs(sse_b - sse) <= 1e-6 * sse), (step, sse, sse_b)
    wa, wb = a.get_weights(), b.get_weights()
    atomic = ("embed_atom/", "dense_embed/", "neighbor_d/", "neighbor_w/", "predict_property/")  # float atomics: equal to rounding
    for k in wa:
        if k.startswith(atomic):
            assert np.allclose(wa[k], wb[k], rtol=1e-4, atol=1e-6), k
        else:
            assert np.allclose(wa[k], wb[k], rtol=1e-5, atol=1e-7), k
    for rb in rbs:
        rb.free()


def test_fused_backward_kernels_match_the_modular_ones(hip_lib, monkeypatch):
    """SCANN_TRAIN_FUSED=0 selects the one-kernel-per-operation backward (exact fp32 MFMA); the fused chains (split-fp16 with
    per-row scaling) must give the same gradients to rounding -- also with a loss scale 2^-20 (tiny gradients), where an
    unscaled fp16 split would lose everything."""
    from scann.models.scann_model import HipModel

    cfg, w, pk, targets, _ = setup(n=32, L=3, seed=21)
    grads = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SCANN_TRAIN_FUSED", mode)
        m = HipModel(cfg, w, device=0)
        eng = m.engine
        eng.train_begin()
        rb = eng.upload(pk)
        for tag, count in (("unit", pk.n_struct), ("tiny", pk.n_struct << 20)):
            sse = eng.train_forward(rb, targets, dropout=0.1, seed=3)
            eng.zero_grads()
            eng.train_backward(rb, sse * (count / pk.n_struct), count)  # same rmse, gradients scaled by n_struct / count
            grads[mode, tag] = eng.get_grads()
        rb.free()
    for tag in ("unit", "tiny"):
        for k, ref in grads["0", tag].items():
            scale = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2))) + 1e-30
            err = float(np.max(np.abs(grads["1", tag][k].astype(np.float64) - ref))) / scale
            assert err < 2e-5, (tag, k, err)
    # and the tiny-scale gradients are the unit ones times 2^-20 (power of two: exact up to the float atomics' order)
    for k, ref in grads["1", "unit"].items():
        scale = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2))) + 1e-30
        assert float(np.max(np.abs(grads["1", "tiny"][k].astype(np.float64) * 2.0 ** 20 - ref))) / scale < 2e-5, k


@pytest.mark.parametrize("widths", ["128x8", "64x4"])
def test_two_steps_in_flight_equal_one_at_a_time(hip_lib, widths):
    """scann_train_step_begin may be called for step k + 1 before scann_train_step_end of step k (the device then never waits for
    the host): same weights as ending every step before the next begins, the reported {sse, count, sum |y - t|} belong to the
    right step, a third begin is refused, and the batches are released without a device-wide synchronisation.  On the MFMA kernels and
    on the plain-fp32 ones (64 / 4), whose per-batch tensors and temporaries are released with the batch as well."""
    from scann import _hip
    from scann.models.scann_model import HipModel

    if widths == "64x4":
        cfg, w, pk, targets, _ = setup_widths(OTHER_WIDTHS["64x4"], n=20, seed=31)
    else:
        cfg, w, pk, targets, _ = setup(n=20, L=2, seed=31)
    de, dn = so.synth_dataset(14, 77)
    inputs2, targets2 = so.pad_batch(de, dn, cfg["model"]["g_update"])
    pk2 = _hip.pack_inputs(inputs2)
    batches = [(pk, targets), (pk2, targets2), (pk, targets), (pk2, targets2), (pk, targets)]
    results = {}
    for mode in ("serial", "pipelined"):
        eng = HipModel(cfg, w, device=0).engine
        eng.train_begin()
        stats, pending = [], []
        for i, (b, t) in enumerate(batches):
            rb = eng.upload(b)
            eng.train_step_begin(rb, t, 1e-3, dropout=0.1, seed=i)
            pending.append(rb)
            if mode == "serial" or len(pending) == 2:
                stats.append(eng.train_step_end())
                pending.pop(0).release()
        while pending:
            stats.append(eng.train_step_end())
            pending.pop(0).release()
        results[mode] = (stats, eng.get_weights())
        if mode == "pipelined":  # a third step in flight is refused (and nothing is left in flight afterwards)
            extra = [eng.upload(pk) for _ in range(3)]
            eng.train_step_begin(extra[0], targets, 1e-3)
            eng.train_step_begin(extra[1], targets, 1e-3)
            with pytest.raises(RuntimeError):
                eng.train_step_begin(extra[2], targets, 1e-3)
            eng.train_step_end()
            eng.train_step_end()
            with pytest.raises(RuntimeError):
                eng.train_step_end()
            for rb in extra:
                rb.release()
    assert [s[1] for s in results["serial"][0]] == [20, 14, 20, 14, 20]
    for a, b in zip(results["serial"][0], results["pipelined"][0]):
        assert a[1] == b[1] and abs(a[0] - b[0]) <= 1e-6 * a[0] and abs(a[2] - b[2]) <= 1e-6 * a[2], (a, b)
    # five Adam steps; the few float-atomic gradient tensors (embedding, basis filters, head) differ by rounding between any two runs and
    # Adam's normalisation carries that into every weight at the 1e-7 level
    for k, v in results["serial"][1].items():
        assert np.allclose(v, results["pipelined"][1][k], rtol=1e-4, atol=2e-6), k
    # sum |y - t| reported by the step = what the forward of the same weights gives
    eng = HipModel(cfg, w, device=0).engine
    eng.train_begin()
    rb = eng.upload(pk)
    sse = eng.train_forward(rb, targets, dropout=0.1, seed=0)
    y, _ = eng.download(rb, want_ga=False)
    first = results["serial"][0][0]
    assert abs(first[0] - sse) <= 1e-6 * sse and abs(first[2] - float(np.abs(y - targets).sum())) <= 1e-5 * first[2]
    rb.free()


def test_weight_pushed_out_of_range_by_the_optimiser_is_an_error(hip_lib):
    """scann_load_weights refuses |w| >= 255.9 (the fp16 hi part of w * 2^8 must be finite); after an optimiser step the device
    re-splits the weights itself -- a step that pushes a weight past the limit must surface as SCANN_ERR_RANGE, not as inf."""
    from scann import _hip

    cfg, w, pk, targets, model = setup()
    eng = model.engine
    eng.train_begin()
    rb = eng.upload(pk)
    eng.train_step(rb, targets, 1e-3, dropout=0.0, seed=1)  # an ordinary step is fine
    with pytest.raises(_hip.ScannHipError) as ei:
        eng.train_step(rb, targets, 1.0e3, dropout=0.0, seed=2)  # Adam moves every weight by ~lr: far past 255.9
    assert ei.value.code == -7 and "weight" in str(ei.value), str(ei.value)
    rb.free()


def test_batch_uploaded_before_training_mode_trains_the_same(hip_lib):
    """scann_batch_upload builds the reverse adjacency (which only the backward pass reads) for handles in training mode; a batch
    that was uploaded before scann_train_begin gets it on its first backward.  Same gradients either way."""
    from scann.models.scann_model import HipModel

    cfg, w, pk, targets, _ = setup(n=24, L=3, seed=31)
    eng = HipModel(cfg, w, device=0).engine
    early = eng.upload(pk)          # inference-mode upload: no reverse adjacency
    y0, _ = eng.forward(pk)         # (and the handle can predict before it trains)
    eng.train_begin()
    late = eng.upload(pk)
    grads = []
    for rb in (late, early):
        sse = eng.train_forward(rb, targets, dropout=0.0, seed=5)
        eng.zero_grads()
        eng.train_backward(rb, sse, pk.n_struct)
        grads.append(eng.get_grads())
    assert np.isfinite(y0).all()
    for k in grads[0]:
        assert np.array_equal(grads[0][k], grads[1][k]) or np.allclose(grads[0][k], grads[1][k], rtol=1e-5, atol=1e-8), k
    late.free()
    early.free()


# ---- widths other than 128 / 8: the plain-fp32 training kernels of csrc/scann_generic_train.hip ----

OTHER_WIDTHS = {
    "64x4": dict(local_dim=64, num_head=4, global_dim=96, dense_out=32),
    "192x6_L3": dict(local_dim=192, num_head=6, global_dim=160, dense_out=200, n_attention=3),
    "48x48_base_plain": dict(local_dim=48, num_head=48, global_dim=16, dense_out=8, g_update=False, use_attn_norm=False, use_ga_norm=False),
    "32x1_ring": dict(local_dim=32, num_head=1, global_dim=300, dense_out=128, use_ring=True),
    "128x16": dict(local_dim=128, num_head=16, global_dim=128, dense_out=128),
    "96x8_cgcnn": dict(local_dim=96, num_head=8, global_dim=64, dense_out=64, feature="cgcnn"),
    "40x5_e_b": dict(local_dim=40, num_head=5, global_dim=24, dense_out=72, n_attention=1),
}


def setup_widths(over, n=6, seed=1, target=None):
    from scann import _hip
    from scann.models.scann_model import HipModel

    cfg = so.default_config("qm9")
    cfg["model"]["n_attention"] = 2
    cfg["model"].update(over)
    cfg["model"]["n_atoms"] = 100
    if target:
        cfg["hyper"]["target"] = target
    ring, cg = bool(cfg["model"].get("use_ring")), cfg["model"].get("feature") == "cgcnn"
    w = so.init_weights(cfg, 3, perturb=True)
    de, dn = so.synth_dataset(n, seed, use_ring=ring)
    inputs, targets = so.pad_batch(de, dn, cfg["model"]["g_update"], use_ring=ring)
    if cg:
        table = np.random.default_rng(5).integers(0, 2, size=(101, 92)).astype("float32")
        inputs["atomic"] = table[inputs["atomic"]]
    pk = _hip.pack_inputs(inputs)
    model = HipModel(cfg, w, device=0)
    return cfg, w, pk, targets, model


@pytest.mark.parametrize("name", list(OTHER_WIDTHS))
def test_gradients_at_other_widths(hip_lib, name):
    """The reference compiles and fits whatever create_model built (scann_model.py:199-241 on :330-434).  A handle whose widths are
    not 128 / 8 trains on the plain-fp32 kernels (csrc/scann_generic_train.hip); its gradients are held to the rule of the
    128-wide path (check_grads: the fp64 autograd of the torch graph, slack relative to the same graph in fp32), on both LocalAttention
    branches, with the ring / cgcnn embeddings, and through mrelu (target e_b).  Two backward calls accumulate."""
    cfg, w, pk, targets, model = setup_widths(OTHER_WIDTHS[name], target="e_b" if name.endswith("e_b") else None)
    eng = model.engine
    eng.train_begin()
    rb = eng.upload(pk)
    sse = eng.train_forward(rb, targets)
    eng.zero_grads()
    eng.train_backward(rb, sse, pk.n_struct)
    got = eng.get_grads()
    rmse, _ = check_grads(got, cfg, w, pk, targets)
    assert abs(np.sqrt(sse / pk.n_struct) - rmse) <= 2e-5 * max(rmse, 1e-6)
    eng.train_backward(rb, sse, pk.n_struct)
    g2 = eng.get_grads()
    for k in got:
        assert np.allclose(g2[k], 2 * got[k], rtol=1e-5, atol=1e-9 + 1e-6 * np.abs(got[k]).max()), k
    # the training forward (Dropout off) is the inference forward: same kernels up to the property head
    y_inf = model.predict(pk)
    assert abs(float(np.sum((np.asarray(y_inf).ravel() - np.asarray(targets).ravel()) ** 2)) - sse) <= 1e-4 * max(sse, 1e-9)
    rb.free()


def test_plain_fp32_training_cross_checks_the_mfma_path(hip_lib, monkeypatch):
    """Two independent GPU implementations of the training step on the 128 / 8 QM9 config: the split-fp16 MFMA kernels and the plain-fp32
    kernels (SCANN_GENERIC=1).  Same gradients to the fp32 floor, and three Adam steps with the Dropout layers active (the masks are a
    function of (seed, layer, element) in both) end in the same weights."""
    from scann.models.scann_model import HipModel

    cfg, w, pk, targets, fast = setup(n=12, L=2, seed=6)
    monkeypatch.setenv("SCANN_GENERIC", "1")
    plain = HipModel(cfg, w, device=0)
    monkeypatch.delenv("SCANN_GENERIC")
    out = {}
    for name, model in (("mfma", fast), ("plain", plain)):
        eng = model.engine
        eng.train_begin()
        rb = eng.upload(pk)
        sse = eng.train_forward(rb, targets)
        eng.zero_grads()
        eng.train_backward(rb, sse, pk.n_struct)
        g = eng.get_grads()
        check_grads(g, cfg, w, pk, targets)
        sses = []
        for step in range(3):
            s_, cnt = eng.train_step(rb, targets, 1e-3, dropout=0.1, seed=40 + step)
            assert cnt == pk.n_struct
            sses.append(s_)
        out[name] = (sse, g, sses, eng.get_weights())
        rb.free()
    (sse_a, ga, sa, wa), (sse_b, gb, sb, wb) = out["mfma"], out["plain"]
    assert abs(sse_a - sse_b) <= 1e-4 * sse_a
    for k in ga:
        scale = max(float(np.abs(ga[k]).max()), 1e-12)
        assert float(np.abs(ga[k] - gb[k]).max()) <= 2e-4 * scale, k
    assert np.allclose(sa, sb, rtol=2e-3), (sa, sb)
    for k in wa:  # Adam moves a weight by ~lr * sign(g): where |g| sits at the rounding floor the two paths may step apart (<= 2 lr a step)
        diff = np.abs(wa[k] - wb[k]).ravel()
        assert float(np.quantile(diff, 0.9)) <= 2e-4 and float(diff.max()) <= 6.5e-3, (k, float(np.quantile(diff, 0.9)), float(diff.max()))


def test_attention_dropout_gradients_at_other_widths(hip_lib):
    """use_drop on a generic-width handle: the mask of gen_attn_kernel (element = edge * num_head + head), rebuilt on the host, is fed to
    the torch graph; gradients match.  And the Dropout(0.1) layers are active, seed-keyed and reproducible to the bit."""
    cfg, w, pk, targets, model = setup_widths(OTHER_WIDTHS["64x4"], seed=2)
    H = cfg["model"]["num_head"]
    eng = model.engine
    eng.train_begin()
    eng.set_attention_dropout(0.3)
    rb = eng.upload(pk)
    seed = 4242
    sse = eng.train_forward(rb, targets, dropout=0.0, seed=seed)
    eng.zero_grads()
    eng.train_backward(rb, sse, pk.n_struct)
    got = eng.get_grads()
    idx = np.arange(pk.n_edge * H, dtype=np.uint64)
    scales = [drop_scale_np(seed, 2000 + l, idx, 0.3).reshape(pk.n_edge, H) for l in range(2)]
    rmse, _ = check_grads(got, cfg, w, pk, targets, attn_scale=scales)
    assert abs(np.sqrt(sse / pk.n_struct) - rmse) <= 2e-5 * max(rmse, 1e-6)
    eng.set_attention_dropout(0.0)
    s0 = eng.train_forward(rb, targets)
    grads = []
    for sd in (1, 2, 1):
        s_ = eng.train_forward(rb, targets, dropout=0.1, seed=sd)
        eng.zero_grads()
        eng.train_backward(rb, s_, pk.n_struct)
        grads.append((s_, eng.get_grads()))
    assert grads[0][0] != s0 and grads[0][0] != grads[1][0] and grads[0][0] == grads[2][0]
    for k in grads[0][1]:  # no atomics anywhere on this path: the same step twice gives the same BITS
        assert np.array_equal(grads[0][1][k], grads[2][1][k]), k
    rb.free()


def test_fit_at_other_widths_lowers_the_loss(hip_lib):
    """train_step on a generic-width handle: Adam (tf.keras formula, l2 regulariser) against the NumPy restatement for one step, then
    the loss of a fixed batch goes down over 30 steps and the inference forward sees the updated weights."""
    import torch_ref
    from scann.models.scann_model import HipModel

    cfg, w, pk, targets, model = setup_widths(OTHER_WIDTHS["64x4"], n=16, seed=9)
    eng = model.engine
    eng.train_begin()
    rb = eng.upload(pk)
    sse0 = eng.train_forward(rb, targets)
    eng.zero_grads()
    eng.train_backward(rb, sse0, pk.n_struct)
    g = eng.get_grads()
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-7
    eng.adam_step(lr)
    got = eng.get_weights()
    for k in w:
        gi = g[k].astype(np.float64) + (2e-4 * w[k].astype(np.float64) if k.endswith(torch_ref.REGULARIZED) else 0.0)
        m, v = (1 - b1) * gi, (1 - b2) * gi * gi
        ref = w[k].astype(np.float64) - lr * np.sqrt(1 - b2) / (1 - b1) * m / (np.sqrt(v) + eps)
        assert np.allclose(got[k], ref, rtol=1e-4, atol=2e-6), k
    t2 = np.asarray(targets) * 0.5 + 0.3
    first = last = None
    for step in range(30):
        sse, cnt = eng.train_step(rb, t2, 2e-3, dropout=0.0, seed=step)
        first = sse if first is None else first
        last = sse
    assert last < 0.5 * first, (first, last)
    y_new, _ = eng.forward(pk)
    y_fresh = HipModel(cfg, eng.get_weights(), device=0).predict(pk)
    assert np.allclose(np.asarray(y_new).ravel(), np.asarray(y_fresh).ravel(), rtol=1e-5, atol=1e-6)
    rb.free()


def test_gradients_at_other_widths_corner_batches(hip_lib):
    """Generic-width training on the batches the 128-wide path has corner tests for: an atom with 130 neighbours and one with 70 among
    ordinary molecules, an atom without neighbours and a one-atom structure (GlobalAttention without normalisation: the reference's
    0 / 0 otherwise), a batch without any edge, and a model without LocalAttention layers."""
    from scann import _hip
    from scann.models.scann_model import HipModel

    over = dict(local_dim=64, num_head=4, global_dim=96, dense_out=32, use_ga_norm=False)
    cfg, w, _, _, model = setup_widths(over, n=2, seed=3)
    cfg_n, w_n, _, _, model_n = setup_widths(dict(over, use_ga_norm=True), n=2, seed=3)  # the 140-atom structure: normalised scores
    rng = np.random.default_rng(4)
    A = 140
    deg = {0: 70, 5: 130, 139: 65}
    nb = []
    for a in range(A):
        dd = deg.get(a, int(rng.integers(1, 7)))
        js = rng.choice(np.delete(np.arange(A), a), dd, replace=False)
        nb.append([[6, int(j), float(rng.uniform(0.4, 3.5)), 1.0, float(rng.uniform(0.9, 4.0))] for j in js])
    de, dn = so.synth_dataset(2, 3)
    de3, dn3 = np.empty(3, dtype=object), np.empty(3, dtype=object)
    de3[0], dn3[0] = de[0], dn[0]
    de3[1], dn3[1] = [[int(z) for z in rng.choice([1, 6, 7, 8], A)], 0.3], nb
    de3[2], dn3[2] = de[1], dn[1]
    inputs, targets = so.pad_batch(de3, dn3, True)
    big = _hip.pack_inputs(inputs)
    lone = _hip.PackedBatch([6, 1, 8, 1], [0, 1, 4], [0, 0, 1, 2, 2], [2, 1], [1.1, 1.3], [0.9, 1.7])
    bare = _hip.PackedBatch([6, 1, 8], [0, 1, 3], [0, 0, 0, 0], [], [], [])
    model.engine.train_begin()
    model_n.engine.train_begin()
    for name, pk, tg in (("big", big, targets), ("lone", lone, np.array([0.3, -0.2], np.float32)), ("bare", bare, np.array([0.1, 0.4], np.float32))):
        eng, cfg_, w_ = (model_n.engine, cfg_n, w_n) if name == "big" else (model.engine, cfg, w)
        rb = eng.upload(pk)
        sse = eng.train_forward(rb, tg)
        eng.zero_grads()
        eng.train_backward(rb, sse, pk.n_struct)
        got = eng.get_grads()
        # (un-normalised GlobalAttention scores, use_ga_norm=False: the query gradient is a difference of nearly equal terms -- d agg sums
        #  to ~0 over a structure's atoms -- and the SAME graph in torch fp32 can sit above the overall cap (2.3e-4 in the L = 0 case
        #  below); the cap is lifted for those models, the 4 x fp32 rule stands)
        rmse, _ = check_grads(got, cfg_, w_, pk, tg, cap=None if name == "big" else 2e-3)
        assert abs(np.sqrt(sse / pk.n_struct) - rmse) <= 2e-5 * max(rmse, 1e-6), name
        rb.free()
    cfg0, w0, pk0, t0, model0 = setup_widths(dict(over, n_attention=0), n=5, seed=8)
    eng0 = model0.engine
    eng0.train_begin()
    rb = eng0.upload(pk0)
    sse = eng0.train_forward(rb, t0)
    eng0.zero_grads()
    eng0.train_backward(rb, sse, pk0.n_struct)
    check_grads(eng0.get_grads(), cfg0, w0, pk0, t0, cap=2e-3)  # (un-normalised scores: the fp32 graph itself sits 2.3e-4 off)
    rb.free()


@pytest.mark.parametrize("path", ["mfma", "plain", "64x4"])
def test_dropout_layer_gradients_match_autograd(hip_lib, monkeypatch, path):
    """The two Dropout(0.1) layers of the training graph (scann_model.py:374 on the centres, attention.py:29 inside ResidualNorm) with the
    library's counter-based masks rebuilt on the host and fed to the torch graph: loss and gradients of a Dropout-active step against
    fp64 autograd, on the MFMA kernels, on the plain-fp32 kernels at the same widths, and at 64 / 4."""
    from scann.models.scann_model import HipModel

    if path == "64x4":
        cfg, w, pk, targets, model = setup_widths(OTHER_WIDTHS["64x4"], n=8, seed=5)
    else:
        cfg, w, pk, targets, model = setup(n=8, L=3, seed=5)
        if path == "plain":
            monkeypatch.setenv("SCANN_GENERIC", "1")
            model = HipModel(cfg, w, device=0)
            monkeypatch.delenv("SCANN_GENERIC")
    eng = model.engine
    eng.train_begin()
    rb = eng.upload(pk)
    for seed, p in ((7, 0.1), (123456789, 0.35)):
        sse = eng.train_forward(rb, targets, dropout=p, seed=seed)
        eng.zero_grads()
        eng.train_backward(rb, sse, pk.n_struct)
        rmse, _ = check_grads(eng.get_grads(), cfg, w, pk, targets, drop=(seed, p))
        assert abs(np.sqrt(sse / pk.n_struct) - rmse) <= 2e-5 * max(rmse, 1e-6), (seed, p)
    rb.free()


def test_plain_backward_on_the_ill_conditioned_two_atom_batch(hip_lib, monkeypatch):
    """Batch 7797 of tests/manual/fuzz_grads.py (committed: tests/golden/fuzz_batch_7797.npz; six structures, one of them a C-O pair whose
    GlobalAttention score k_0 . q_1 = 0.034 is 1.5e-3 of the sum of its terms' magnitudes).  Round 5's plain-fp32 backward sat 17 x further
    from fp64 autograd than the fp32 graph on it, the same factor in every tensor upstream of the readout: gen_pool_bwd_kernel formed the
    scores a second time in fp32 (6e-5 of the score apart from the forward's), the normalisation backward (du - u (u . du)) / |agg|
    cancels ten-fold, and the structure's whole gradient inherits the error.  Now the scores (forward and backward: one function) and the
    per-structure scalar chain are fp64:
    (a) the kernel's d gq / d gk equal the fp64 formula on the kernel's OWN fp32 inputs to 1e-5 of the batch rms (1.8e-3 before);
    (b) what is left is carried in by the fp32 activations: the pair's d gq sits within 12 x what ONE fp32 rounding of gq / gk / d rep does
        to it (8 x measured; 26 x before), every other structure's within 1e-4 of the batch rms;
    (c) every parameter gradient within 16 x the fp32 graph's own distance from fp64 (6 x measured; 17-20 x before).
    tools/debug_plain_grads.py prints the stages; profiles/r06_notes.md section 3 has the before / after and a census over random batches."""
    import importlib.util

    import torch_ref
    from scann import _hip
    from scann.models.scann_model import HipModel, normalize_config

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("debug_plain_grads", os.path.join(root, "tools", "debug_plain_grads.py"))
    dbg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dbg)
    z = np.load(os.path.join(root, "tests", "golden", "fuzz_batch_7797.npz"))
    targets = z["targets"]
    pk = _hip.pack_inputs({k: z[k] for k in z.files if k != "targets"})
    sizes = np.diff(pk.mol_offset).tolist()
    assert sizes == [4, 7, 8, 2, 3, 26] and pk.n_edge == 148
    cfg = normalize_config(so.default_config("qm9"))
    cfg["model"].update(n_attention=3)
    w = so.init_weights(cfg, 77, perturb=True)
    monkeypatch.setenv("SCANN_GENERIC", "1")
    eng = HipModel(cfg, w, device=0).engine
    monkeypatch.delenv("SCANN_GENERIC")
    eng.train_begin()
    rb = eng.upload(pk)
    drop, seed = 0.1, 100 + 7797
    sse = eng.train_forward(rb, targets, dropout=drop, seed=seed)
    eng.zero_grads()
    eng.train_backward(rb, sse, pk.n_struct)
    got = eng.get_grads()
    dg, norm = cfg["model"]["global_dim"], cfg["model"]["use_ga_norm"]
    t = {k: eng.train_debug_read(rb, k, dg).astype(np.float64) for k in ("gq", "gk", "drep", "dgq", "dgk")}
    rb.free()
    # (a) the kernel against the fp64 formula on its own inputs
    own_q, own_k = dbg.pool_bwd64(pk, t["gq"], t["gk"], t["drep"], norm)
    assert max(dbg.per_struct(pk, t["dgq"], own_q)) <= 1e-5 and max(dbg.per_struct(pk, t["dgk"], own_k)) <= 1e-5
    # (b) against fp64 autograd, per structure, in units of what one fp32 rounding of the inputs does
    cap = {}
    _, _, ref, _ = torch_ref.loss_and_grads(cfg, w, pk, targets, drop=(seed, drop), capture=cap)
    ref_q, _ = dbg.pool_bwd64(pk, cap["gq"][0], cap["gk"][0], cap["rep"][1], norm)
    rng = np.random.default_rng(0)
    pert = lambda x: x * (1.0 + rng.uniform(-6e-8, 6e-8, x.shape))  # noqa: E731
    one_q, _ = dbg.pool_bwd64(pk, pert(cap["gq"][0]), pert(cap["gk"][0]), pert(cap["rep"][1]), norm)
    pair = sizes.index(2)
    sens = dbg.per_struct(pk, one_q, ref_q)[pair]
    err = dbg.per_struct(pk, t["dgq"], cap["gq"][1])
    assert 5e-5 < sens < 1e-3, sens                      # (the batch IS ill-conditioned there: 2.3e-4 of the batch rms per rounding)
    assert err[pair] <= 12.0 * sens, (err[pair], sens)
    assert max(e for i, e in enumerate(err) if i != pair) <= 1e-4, err
    # (c) parameter gradients against the fp32 graph's own distance
    _, _, g32, _ = torch_ref.loss_and_grads(cfg, w, pk, targets, dtype="float32", drop=(seed, drop))
    for k in ref:
        if k.endswith(torch_ref.REGULARIZED):
            ref[k] = ref[k] - 2e-4 * w[k].astype(np.float64)
            g32[k] = g32[k] - 2e-4 * w[k].astype(np.float64)
    e_gpu, e_32 = grad_errors(got, ref), grad_errors(g32, ref)
    bad = {k: (e_gpu[k], e_32[k]) for k in ref if not e_gpu[k] <= max(GRAD_FLOOR, 16.0 * e_32[k])}
    assert not bad, bad
