"""Independent cross-check of the NumPy oracle's op semantics: the same graph written a second time with stock
torch CPU ops (F.embedding, F.linear, F.silu, F.layer_norm, F.softmax) on the PACKED/CSR layout, so it shares
neither the layout nor the hand-written LayerNorm/softmax/swish of oracle/scann_oracle.py.  Test-only."""
import math

import numpy as np
import pytest

pytest.importorskip("torch")  # test infrastructure only: a box without torch SKIPS the tests that cross-check against it


REGULARIZED = ("query/kernel", "key/kernel", "filter_geo/kernel", "dense_1/kernel", "dense_2/kernel",
               "after_Lc/kernel", "bf_property/kernel")  # kernel_regularizer=l2(1e-4): attention.py:27-28,95-109,260-265; scann_model.py:428,441


def _mrelu(x):
    """mrelu of the reference (custom_layers.py:6-15): forward max(x, 0), backward the IDENTITY -- a tf.custom_gradient whose
    grad(dy) returns dy, also where x < 0 (torch.relu would pass a zero there)."""
    import torch

    class MRelu(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return torch.clamp(t, min=0)

        @staticmethod
        def backward(ctx, g):
            return g

    return MRelu.apply(x)


def drop_scale_np(seed, tag, idx, p):
    """NumPy twin of drop_scale() in csrc/scann_internal.h (64-bit mix, top 24 bits -> uniform): 0 or 1 / (1 - p) per element index."""
    M = np.uint64(0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (idx.astype(np.uint64) + np.uint64(1)) + (np.uint64(tag) << np.uint64(48))) & M
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float64) / 16777216.0
    return np.where(u < np.float32(p), 0.0, 1.0 / (1.0 - float(np.float32(p))))


def loss_and_grads(config, weights, pk, targets, attn_scale=None, dtype="float64", drop=None, capture=None):
    """Training loss of the reference (scann_model.py:210-214: RMSE + sum of l2(1e-4) kernel regularisers) and its
    gradient w.r.t. every tensor, by torch autograd in fp64 (``dtype="float32"``: the same graph in single precision --
    the rounding floor any fp32 implementation of the step sits on).  Dropout layers are inactive (rate 0) unless ``drop`` =
    (seed, rate) asks for the library's counter-based masks of the two Dropout(0.1) layers (scann_model.py:374, attention.py:29)."""
    import torch

    dt = getattr(torch, dtype)
    W = {k: torch.tensor(np.asarray(v), dtype=dt, requires_grad=True) for k, v in weights.items()}
    y, _ = forward_packed(config, W, pk, dtype, as_tensor=True, attn_scale=attn_scale, drop=drop, capture=capture)
    t = torch.tensor(np.asarray(targets), dtype=dt).reshape(-1, 1)
    rmse = torch.sqrt(torch.mean((y - t) ** 2))  # losses.py:5-6
    reg = sum((W[k] ** 2).sum() for k in W if k.endswith(REGULARIZED)) * 1e-4
    loss = rmse + reg
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).numpy().astype(np.float64) for k, v in W.items()}  # (None: a tensor the graph does not reach)
    if capture is not None:  # readout-stage tensors and their gradients (tools/debug_plain_grads.py): name -> (value, gradient)
        reps = capture.pop("_reps")
        for k in list(capture):
            t = capture[k]
            capture[k] = (t.detach().numpy().astype(np.float64), t.grad.numpy().astype(np.float64))
        capture["rep"] = (np.stack([r.detach().numpy() for r in reps]).astype(np.float64), np.stack([r.grad.numpy() for r in reps]).astype(np.float64))
    return float(loss.detach()), float(rmse.detach()), grads, y.detach().numpy()


def forward_packed(config, weights, pk, dtype="float64", as_tensor=False, attn_scale=None, drop=None, capture=None):
    """attn_scale: optional list (one [E, H] array per layer) of inverted-dropout factors for the attention weights.
    drop: optional (seed, rate): the library's masks on the centres after dense_embed (tag 1000) and on the ResidualNorm branch of
    layer l (tag l), element index = atom * local_dim + column."""
    import torch
    import torch.nn.functional as F

    dt = getattr(torch, dtype)
    cfg = config["model"]
    W = {k: (v if torch.is_tensor(v) else torch.tensor(np.asarray(v), dtype=dt)) for k, v in weights.items()}
    d, H = cfg["local_dim"], cfg["num_head"]
    hd = d // H

    def lin(x, p):
        return F.linear(x, W[p + "/kernel"].T, W[p + "/bias"])

    def ln(x, p):
        return F.layer_norm(x, (x.shape[-1],), W[p + "/gamma"], W[p + "/beta"], eps=1e-6)

    def gauss(x, stop):
        c = torch.tensor(np.linspace(0, stop, 20, dtype="float32"), dtype=dt)
        return torch.exp(-((x[:, None] - c[None, :]) ** 2) / 0.25)

    atomic = torch.tensor(pk.atomic, dtype=torch.long) if pk.atomic is not None else None
    col = torch.tensor(pk.edge_col, dtype=torch.long)
    deg = np.diff(pk.edge_offset)
    row = torch.tensor(np.repeat(np.arange(pk.n_atom), deg), dtype=torch.long)
    dist = torch.tensor(pk.edge_dist, dtype=dt)
    wgt = torch.tensor(pk.edge_weight, dtype=dt)
    A, E = pk.n_atom, pk.n_edge

    if cfg.get("feature", "atomic") == "cgcnn":  # scann_model.py:365
        v = lin(torch.tensor(pk.cgcnn, dtype=dt), "embed_atom")
    else:
        v = F.embedding(atomic, W["embed_atom/embeddings"])
    if cfg.get("use_ring", False):  # scann_model.py:367-371
        v = torch.cat([v, lin(torch.tensor(pk.ring, dtype=dt), "extra_embed")], -1)
    c = F.silu(lin(v, "dense_embed"))

    def mask(tag):
        return torch.tensor(drop_scale_np(drop[0], tag, np.arange(A * d, dtype=np.uint64), drop[1]).reshape(A, d), dtype=dt)

    if drop:
        c = c * mask(1000)
    gd = gauss(dist, cfg["gaussian_d"])
    if cfg["g_update"]:
        geom = F.silu(lin(gd, "neighbor_d")) * F.silu(lin(gauss(wgt, math.pi * 2), "neighbor_w"))
    for i in range(cfg["n_attention"]):
        p = "local_attention_%d" % i
        cn = c[col]
        if cfg["g_update"]:
            upd = F.silu(lin(torch.cat([c[row], geom, cn], -1), p + "/filter_geo"))
            geom = ln(upd + geom, p + "/layer_norm_g")
            g = geom
        else:
            g = F.silu(lin(gd, p + "/filter_geo")) * wgt[:, None]
        q = lin(c, p + "/query")
        k = lin(cn * g, p + "/key")
        e = ((q[row] * hd ** -0.5).view(E, H, hd) * k.view(E, H, hd)).sum(-1)  # [E,H]
        attn = torch.zeros_like(e)
        off = pk.edge_offset
        for a in range(A):  # per-atom softmax over its own edges
            if off[a + 1] > off[a]:
                attn[off[a]:off[a + 1]] = F.softmax(e[off[a]:off[a + 1]], 0)
        if attn_scale is not None:
            attn = attn * torch.tensor(attn_scale[i], dtype=dt)
        ctx = torch.zeros(A, d, dtype=dt).index_add_(0, row, (attn[:, :, None] * k.view(E, H, hd)).reshape(E, d)) + q
        ctx = ln(ctx, p + "/layer_norm")
        if cfg["use_attn_norm"]:
            r = "residual_norm_%d" % i
            ffn = lin(F.silu(lin(ctx, r + "/dense_1")), r + "/dense_2")
            c = ln(ctx + (ffn * mask(i) if drop else ffn), r + "/layer_norm")
        else:
            c = ctx
    z = F.silu(lin(c, "after_Lc"))
    gq, gk = lin(z, "global_attention/query"), lin(z, "global_attention/key")
    reps = []
    ys, gas = [], []
    for s in range(pk.n_struct):
        a0, a1 = pk.mol_offset[s], pk.mol_offset[s + 1]
        en = gk[a0:a1] @ gq[a0:a1].T
        agg = (en - torch.diag(torch.diag(en))).sum(-1)
        if cfg["use_ga_norm"]:
            agg = agg / torch.linalg.vector_norm(agg)
        at = F.softmax(agg, 0)
        rep = (at[:, None] * gk[a0:a1]).sum(0)
        reps.append(rep)
        y = lin(F.silu(lin(rep, "bf_property")), "predict_property")
        if config.get("hyper", {}).get("target") == "e_b":
            y = _mrelu(y)
        ys.append(y)
        gas.append(at)
    if capture is not None:
        capture.update(z=z, gq=gq, gk=gk)
        for t in capture.values():
            t.retain_grad()
        for r in reps:
            r.retain_grad()
        capture["_reps"] = reps
    if as_tensor:
        return torch.stack(ys).reshape(-1, 1), torch.cat(gas)
    return torch.stack(ys).numpy().reshape(-1, 1), torch.cat(gas).numpy()
