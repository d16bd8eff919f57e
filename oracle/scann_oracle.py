"""CPU oracle for the SCANN / SCANN+ forward hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it.  The shipped path
(``scann--material_amd/``) never routes through it and fails loudly without its HIP library.

PARITY UNPINNED: the reference (sinhvt3421/scann--material @ 2024_08_07) ships no tests,
no golden vectors and no weights, and its arithmetic lives in TensorFlow/Keras 2.10 which is
absent from this image.  This oracle is therefore a line-by-line NumPy
restatement of the reference *source*, in the reference's own padded-dense ``[B, M, N, d]``
layout, each function citing the reference ``file:line`` it follows.  TF/Keras op semantics
that are restated from the TF 2.10 documentation/source (third-party, not in the tree):

* ``Dense``:               ``x @ kernel[in, out] + bias`` then activation
* ``swish``:               ``x * sigmoid(x)``
* ``LayerNormalization``:  epsilon=1e-6 takes the non-fused path: ``mean, var = moments(x, -1)``
                           (population variance), ``inv = rsqrt(var + eps) * gamma``,
                           ``y = x * inv + (beta - mean * inv)``
* ``softmax``:             ``exp(x - max) / sum(exp(x - max))``
* ``tf.linalg.normalize``: ``x / sqrt(sum(x * x, axis))`` -- no epsilon
* ``Embedding``:           row lookup, initialiser U(-0.05, 0.05); ``Dense`` initialiser
                           Glorot-uniform, zero bias; LayerNorm gamma=1, beta=0.

All arithmetic is done in ``dtype`` (float32 = the reference precision; float64 is used to
attribute error).
"""
from __future__ import annotations

import numpy as np

N_GAUSS = 20

# ----------------------------------------------------------------------------------------------
# configuration (reference: configs/model_qm9.yaml:1-28, train.py:37-43)
# ----------------------------------------------------------------------------------------------


def default_config(name="qm9"):
    """The ``model:`` section of a reference yaml plus the CLI-injected keys
    (``feature``, ``use_ring``, ``use_drop`` -- train.py:37-39) and ``hyper.target``."""
    if name == "qm9":  # configs/model_qm9.yaml
        model = dict(n_atoms=10, embedding_dim=48, n_attention=7, local_dim=128, num_head=8,
                     global_dim=128, dense_out=128, scale=0.5, use_attn_norm=True,
                     use_ga_norm=True, use_ring=False, g_update=True, gaussian_d=4.0)
        hyper = dict(batch_size=128, target="homo")
    elif name == "mp2018":  # configs/model_mp2018.yaml
        model = dict(n_atoms=95, embedding_dim=128, n_attention=9, local_dim=128, num_head=8,
                     global_dim=128, dense_out=128, scale=0.5, use_attn_norm=True,
                     use_ga_norm=True, use_ring=False, g_update=True, gaussian_d=6.0)
        hyper = dict(batch_size=64, target="e_f")
    elif name == "qm9_std":  # configs/model_qm9_std.yaml
        model = dict(n_atoms=10, embedding_dim=48, n_attention=8, local_dim=128, num_head=8,
                     global_dim=128, dense_out=128, scale=0.5, use_attn_norm=True,
                     use_ga_norm=True, use_ring=False, g_update=True, gaussian_d=4.0)
        hyper = dict(batch_size=128, target="gap")
    else:
        raise KeyError(name)
    model.update(feature="atomic", use_drop=False)
    return {"model": model, "hyper": hyper}


# ----------------------------------------------------------------------------------------------
# weights (Keras default initialisers; canonical tensor names used by the weight container)
# ----------------------------------------------------------------------------------------------


def weight_shapes(config):
    """Ordered ``[(name, shape)]`` of every trainable tensor created by
    ``create_model`` (scann_model.py:329-453) and the layers it instantiates
    (attention.py:25-35, 95-113, 260-262)."""
    m = config["model"]
    d, dg, do = m["local_dim"], m["global_dim"], m["dense_out"]
    emb = m["embedding_dim"]
    out = []
    if m.get("feature", "atomic") == "cgcnn":  # scann_model.py:365
        out += [("embed_atom/kernel", (92, emb)), ("embed_atom/bias", (emb,))]
    else:  # scann_model.py:362
        out += [("embed_atom/embeddings", (m["n_atoms"], emb))]
    cin = emb
    if m.get("use_ring", False):  # scann_model.py:368-371
        out += [("extra_embed/kernel", (2, 10)), ("extra_embed/bias", (10,))]
        cin = emb + 10
    out += [("dense_embed/kernel", (cin, d)), ("dense_embed/bias", (d,))]  # :373
    if m.get("g_update", False):  # :381-388
        out += [("neighbor_d/kernel", (N_GAUSS, d)), ("neighbor_d/bias", (d,)),
                ("neighbor_w/kernel", (N_GAUSS, d)), ("neighbor_w/bias", (d,))]
    for i in range(m["n_attention"]):  # :413-421, fresh weights per iteration
        p = "local_attention_%d/" % i
        fin = 3 * d if m.get("g_update", False) else N_GAUSS  # attention.py:142-155
        out += [(p + "query/kernel", (d, d)), (p + "query/bias", (d,)),
                (p + "key/kernel", (d, d)), (p + "key/bias", (d,)),
                (p + "filter_geo/kernel", (fin, d)), (p + "filter_geo/bias", (d,)),
                (p + "layer_norm/gamma", (d,)), (p + "layer_norm/beta", (d,))]
        if m.get("g_update", False):
            out += [(p + "layer_norm_g/gamma", (d,)), (p + "layer_norm_g/beta", (d,))]
        if m.get("use_attn_norm", True):  # attention.py:25-35
            r = "residual_norm_%d/" % i
            out += [(r + "dense_1/kernel", (d, d)), (r + "dense_1/bias", (d,)),
                    (r + "dense_2/kernel", (d, d)), (r + "dense_2/bias", (d,)),
                    (r + "layer_norm/gamma", (d,)), (r + "layer_norm/beta", (d,))]
    out += [("after_Lc/kernel", (d, dg)), ("after_Lc/bias", (dg,)),  # :424
            ("global_attention/query/kernel", (dg, dg)), ("global_attention/query/bias", (dg,)),
            ("global_attention/key/kernel", (dg, dg)), ("global_attention/key/bias", (dg,)),
            ("bf_property/kernel", (dg, do)), ("bf_property/bias", (do,)),  # :437
            ("predict_property/kernel", (do, 1)), ("predict_property/bias", (1,))]  # :445
    return out


def init_weights(config, seed=1234, perturb=False):
    """Keras-default initialisation, seeded.  ``perturb=True`` additionally randomises biases,
    LayerNorm gamma/beta so that tests exercise them (Keras defaults are 0 / 1 / 0)."""
    rng = np.random.default_rng(seed)
    w = {}
    for name, shape in weight_shapes(config):
        leaf = name.rsplit("/", 1)[1]
        if leaf == "kernel":
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))  # glorot_uniform
            t = rng.uniform(-lim, lim, size=shape)
        elif leaf == "embeddings":
            t = rng.uniform(-0.05, 0.05, size=shape)
        elif leaf == "gamma":
            t = np.ones(shape) + (rng.normal(0, 0.1, size=shape) if perturb else 0.0)
        elif leaf in ("bias", "beta"):
            t = rng.normal(0, 0.1, size=shape) if perturb else np.zeros(shape)
        else:
            raise KeyError(name)
        w[name] = np.ascontiguousarray(t, dtype=np.float32)
    return w


def count_params(config):
    return int(sum(int(np.prod(s)) for _, s in weight_shapes(config)))


# ----------------------------------------------------------------------------------------------
# op restatements
# ----------------------------------------------------------------------------------------------


def swish(x):
    """tf.nn.swish: x * sigmoid(x)."""
    one = x.dtype.type(1.0)
    return x * (one / (one + np.exp(-x)))


def dense(x, w, prefix, dt, act=None):
    """tf.keras.layers.Dense: x @ kernel + bias (+ activation)."""
    y = np.matmul(x, w[prefix + "/kernel"].astype(dt)) + w[prefix + "/bias"].astype(dt)
    return swish(y) if act == "swish" else y


def layer_norm(x, gamma, beta, eps=1e-6):
    """tf.keras.layers.LayerNormalization(epsilon=1e-6), non-fused path
    (attention.py:35,111,113)."""
    dt = x.dtype
    mean = x.mean(-1, keepdims=True, dtype=dt)
    var = np.mean((x - mean) ** 2, -1, keepdims=True, dtype=dt)
    inv = (dt.type(1.0) / np.sqrt(var + dt.type(eps))) * gamma.astype(dt)
    return x * inv + (beta.astype(dt) - mean * inv)


def softmax(x, axis):
    """tf.nn.softmax."""
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return e / e.sum(axis=axis, keepdims=True, dtype=x.dtype)


def gaussian_expansion(x, centers, dt, width=0.5):
    """custom_layers.py:39-65 -- exp(-(x - c)^2 / width^2); ``width=0.5`` default is squared in
    the constructor (custom_layers.py:48-51)."""
    c = centers.astype(dt)
    w2 = dt.type(width) ** 2
    return np.exp(-((x[..., None] - c[None, None, None, :]) ** 2) / w2)


def gather_shape(neighbors):
    """custom_layers.py:18-28 -- [B,M,N] -> [B,M,N,2] (batch id, atom id)."""
    B, M, N = neighbors.shape
    rb = np.broadcast_to(np.arange(B, dtype=neighbors.dtype).reshape(B, 1, 1, 1), (B, M, N, 1))
    return np.concatenate([rb, neighbors[..., None]], -1)


def gather_nd(params, indices):
    """tf.gather_nd with 2-component indices into the leading two axes."""
    return params[indices[..., 0], indices[..., 1]]


# ----------------------------------------------------------------------------------------------
# layers
# ----------------------------------------------------------------------------------------------


def local_attention(w, p, cfg, atom_query, nbr_idx, neighbor_geometry, mask, neighbor_weight, dt):
    """LocalAttention.call (attention.py:118-216) with v_proj=False, kq_proj=True
    (scann_model.py:395-403).  Returns (attn, context, neighbor_geometry)."""
    d = cfg["local_dim"]
    H = cfg["num_head"]
    hd = d // H
    B, M, N = mask.shape
    atom_neighbor = gather_nd(atom_query, nbr_idx)  # :136  [B,M,N,d]
    atom_neighbor = atom_neighbor.reshape(B, M, N, d)  # :139
    if cfg.get("g_update", False):  # :141-153
        cat = np.concatenate(
            [np.repeat(atom_query[:, :, None, :], N, 2), neighbor_geometry, atom_neighbor], -1)
        geometry_update = dense(cat, w, p + "filter_geo", dt, "swish")
        neighbor_geometry = layer_norm(geometry_update + neighbor_geometry,
                                       w[p + "layer_norm_g/gamma"], w[p + "layer_norm_g/beta"])
    else:  # :155
        neighbor_geometry = dense(neighbor_geometry, w, p + "filter_geo", dt, "swish") * neighbor_weight
    atom_neighbor_geometry = atom_neighbor * neighbor_geometry  # :157
    query = dense(atom_query, w, p + "query", dt)  # :160
    key = dense(atom_neighbor_geometry, w, p + "key", dt)  # :163
    query_t = query.reshape(B, -1, H, hd)  # :170
    key = key.reshape(B, -1, N, H, hd)  # :173
    dk = dt.type(float(hd) ** (-0.5))  # :180  (scale default 0.5, attention.py:63; yaml "scale" unread)
    query_t = query_t * dk  # :181
    energy = np.einsum("bchd,bcnhd->bhcn", query_t, key)  # :183
    mask_scaled = (dt.type(1.0) - mask[:, None]) * dt.type(-1e9)  # :186
    energy = energy + mask_scaled  # :187
    attn = softmax(energy, -1)  # :189   (Dropout(0.05) is train-only, :191)
    v, q = key, query  # :198-200
    context = np.einsum("bcn,bcnhd->bcnhd", mask, np.einsum("bhcn,bcnhd->bcnhd", attn, v))  # :206
    context = context.reshape(B, M, N, d)  # :208
    context = context.sum(2, dtype=dt) + q  # :212
    context = layer_norm(context, w[p + "layer_norm/gamma"], w[p + "layer_norm/beta"])  # :214
    return attn, context, neighbor_geometry


def residual_norm(w, r, x, dt):
    """ResidualNorm.call (attention.py:37-40); Dropout(0.1) is train-only."""
    y = dense(dense(x, w, r + "dense_1", dt, "swish"), w, r + "dense_2", dt)
    return layer_norm(x + y, w[r + "layer_norm/gamma"], w[r + "layer_norm/beta"])


def global_attention(w, cfg, atom_query, mask, dt):
    """GlobalAttention.call (attention.py:267-318), v_proj=False, kq_proj=True.
    ``mask`` is [B,M,1].  Returns (attn [B,M,1], context [B,dg])."""
    query = dense(atom_query, w, "global_attention/query", dt)  # :269
    key = dense(atom_query, w, "global_attention/key", dt)  # :272
    energy = np.einsum("bkd,bqd->bkq", mask * key, mask * query)  # :279
    M = energy.shape[1]
    mask_center = (~np.eye(M, dtype=bool)).astype(dt)[None]  # :282-283
    energy = mask_center * energy  # :285
    agg = energy.sum(-1, dtype=dt).reshape(atom_query.shape[0], -1, 1)  # :289-290
    agg = mask * agg  # :292
    if cfg.get("use_ga_norm", True):  # :295-297  (no epsilon: 0/0 -> NaN for 1-atom structures)
        with np.errstate(invalid="ignore", divide="ignore"):
            agg = agg / np.sqrt((agg * agg).sum(1, keepdims=True, dtype=dt))
    agg = agg + (dt.type(1.0) - mask) * dt.type(-1e9)  # :299-300
    attn = softmax(agg, 1)  # :302
    context = (mask * (attn * key)).sum(1, dtype=dt)  # :314-316
    return attn, context


# ----------------------------------------------------------------------------------------------
# whole forward graph
# ----------------------------------------------------------------------------------------------


def forward(config, weights, inputs, dtype=np.float32, intermediates=None):
    """``create_model`` graph (scann_model.py:329-453) evaluated on one padded batch dict
    (keys = Keras Input names, scann_model.py:338-357).  Returns ``(y [B,1], ga_attn [B,M,1])``.
    If ``intermediates`` is a dict it receives per-layer tensors."""
    cfg = config["model"]
    dt = np.dtype(dtype)
    w = {k: v.astype(dt) for k, v in weights.items()}
    atom_mask = np.asarray(inputs["atom_mask"]).astype(dt)
    nbr = np.asarray(inputs["neighbors"]).astype(np.int64)
    nmask = np.asarray(inputs["neighbor_mask"]).astype(dt)
    nweight = np.asarray(inputs["neighbor_weight"]).astype(dt)
    ndist = np.asarray(inputs["neighbor_distance"]).astype(dt)

    if cfg.get("feature", "atomic") == "cgcnn":  # :365
        centers = np.matmul(np.asarray(inputs["atomic"]).astype(dt), w["embed_atom/kernel"]) + w["embed_atom/bias"]
    else:  # :362
        centers = w["embed_atom/embeddings"][np.asarray(inputs["atomic"]).astype(np.int64)]
    if cfg.get("use_ring", False):  # :367-371
        ring = dense(np.asarray(inputs["ring_aromatic"]).astype(dt), w, "extra_embed", dt)
        centers = np.concatenate([centers, ring], -1)
    centers = dense(centers, w, "dense_embed", dt, "swish")  # :373 (Dropout train-only :374)
    idx = gather_shape(nbr)  # :376
    gd = gaussian_expansion(ndist, np.linspace(0, cfg["gaussian_d"], N_GAUSS, dtype="float32"), dt)  # :378
    if cfg.get("g_update", False):  # :380-389
        nd = dense(gd, w, "neighbor_d", dt, "swish")
        gw = gaussian_expansion(nweight, np.linspace(0, np.pi * 2, N_GAUSS, dtype="float32"), dt)
        nw = dense(gw, w, "neighbor_w", dt, "swish")
        geometry = nd * nw
        nweight_e = None
    else:
        geometry = gd
        nweight_e = nweight[..., None]  # :391
    if intermediates is not None:
        intermediates["centers_0"] = centers.copy()
        intermediates["geometry_0"] = geometry.copy()
    for i in range(cfg["n_attention"]):  # :413-421
        p = "local_attention_%d/" % i
        attn_local, context, g_f = local_attention(w, p, cfg, centers, idx, geometry, nmask, nweight_e, dt)
        if cfg.get("use_attn_norm", True):  # :404-408
            centers = residual_norm(w, "residual_norm_%d/" % i, context, dt)
        else:
            centers = context
        if cfg.get("g_update", False):
            geometry = g_f  # :415-417  (base branch discards g_f, :419-421)
        if intermediates is not None:
            intermediates["context_%d" % (i + 1)] = context.copy()
            intermediates["centers_%d" % (i + 1)] = centers.copy()
            intermediates["attn_local_%d" % (i + 1)] = attn_local.copy()
            if cfg.get("g_update", False):
                intermediates["geometry_%d" % (i + 1)] = geometry.copy()
    centers = dense(centers, w, "after_Lc", dt, "swish")  # :424-429
    attn_global, struc_rep = global_attention(w, cfg, centers, atom_mask, dt)  # :432-434
    struc_rep = dense(struc_rep, w, "bf_property", dt, "swish")  # :437-442
    out = dense(struc_rep, w, "predict_property", dt)  # :445-447
    if config.get("hyper", {}).get("target") == "e_b":  # mrelu forward = max(x, 0), custom_layers.py:15
        out = np.maximum(out, dt.type(0))
    if intermediates is not None:
        intermediates["after_Lc"] = centers.copy()
        intermediates["struc_rep"] = struc_rep.copy()
    return out, attn_global


# ----------------------------------------------------------------------------------------------
# synthetic "QM9-shaped" / "MP2018-shaped" data in the reference's on-disk object format
# (voronoi_neighbor.py:38-47: per atom a list of [species, idx, solid_angle, ratio, distance];
#  general.py:127-137: data_energy rows [Atomic, target(, ring features)]) -- SURVEY.md 8(d).
# ----------------------------------------------------------------------------------------------

_QM9_Z = np.array([1, 6, 7, 8, 9])
_QM9_P = np.array([0.51, 0.35, 0.06, 0.08, 0.002])
_QM9_P = _QM9_P / _QM9_P.sum()


def synth_dataset(n, seed=0, kind="qm9", use_ring=False):
    """Returns ``(data_energy, data_neighbor)`` object arrays like ``load_dataset``
    (general.py:104-144) would."""
    rng = np.random.default_rng(seed)
    energy = np.empty(n, dtype=object)
    neigh = np.empty(n, dtype=object)
    for i in range(n):
        if kind == "qm9":
            A = int(np.clip(np.rint(rng.normal(18.0, 2.9)), 3, 29))
            Z = rng.choice(_QM9_Z, size=A, p=_QM9_P)
            lo, hi, dlo, dhi = 3, min(12, A - 1), 0.9, 4.0
        elif kind == "worst":  # every molecule 29 atoms x 12 neighbours
            A = 29
            Z = rng.choice(_QM9_Z, size=A, p=_QM9_P)
            lo, hi, dlo, dhi = 12, 12, 0.9, 4.0
        elif kind == "mp2018":
            A = int(np.clip(np.rint(rng.lognormal(3.0, 0.8)), 2, 300))
            Z = rng.integers(1, 95, size=A)
            lo, hi, dlo, dhi = min(6, A - 1), min(24, A - 1), 1.5, 6.0
        else:
            raise KeyError(kind)
        lo = max(1, min(lo, hi))
        atoms = []
        for a in range(A):
            n_nb = int(rng.integers(lo, hi + 1))
            others = np.delete(np.arange(A), a)
            ids = rng.choice(others, size=n_nb, replace=False)
            ang = rng.uniform(0.4, 3.5, size=n_nb)
            dist = rng.uniform(dlo, dhi, size=n_nb)
            ratio = ang / ang.max()
            atoms.append([[int(Z[j]), int(j), float(ang[k]), float(ratio[k]), float(dist[k])]
                          for k, j in enumerate(ids)])
        neigh[i] = atoms
        row = [[int(z) for z in Z], float(rng.normal(0.0, 1.0))]
        if use_ring:
            row.append(rng.integers(0, 2, size=(A, 2)).astype("int32"))
        energy[i] = row
    return energy, neigh


def pad_batch(batch_energy, batch_nei, g_update=True, use_ring=False):
    """The padded input dict ``DataIterator.__getitem__`` builds (datagenerator.py:69-135):
    per-batch maxima M, N; neighbour sentinel 1000 -> mask, then rewritten to 0 (:82-90);
    weight column 2 (raw solid angle) when g_update else 3 (normalised) (:48-50)."""
    B = len(batch_nei)
    M = max(len(c) for c in batch_nei)
    N = max(len(n) for c in batch_nei for n in c)
    wi = 2 if g_update else 3
    nbr = np.full((B, M, N), 1000, dtype="int32")
    wt = np.zeros((B, M, N), dtype="float32")
    ds = np.zeros((B, M, N), dtype="float32")
    atomic = np.zeros((B, M), dtype="int32")
    for b in range(B):
        Z = batch_energy[b][0]
        atomic[b, : len(Z)] = Z
        for a, lst in enumerate(batch_nei[b]):
            for k, n in enumerate(lst):
                nbr[b, a, k] = n[1]
                wt[b, a, k] = n[wi]
                ds[b, a, k] = n[-1]
    mask_local = nbr != 1000
    nbr[nbr == 1000] = 0
    inputs = {
        "atomic": atomic,
        "atom_mask": (atomic != 0)[..., None],
        "neighbors": nbr,
        "neighbor_mask": mask_local,
        "neighbor_weight": wt,
        "neighbor_distance": ds,
    }
    if use_ring:
        ring = np.zeros((B, M, 2), dtype="int32")
        for b in range(B):
            r = np.asarray(batch_energy[b][2])
            ring[b, : len(r)] = r
        inputs["ring_aromatic"] = ring
    target = np.array([float(e[1]) for e in batch_energy], "float32")
    return inputs, target


# FLOP / byte models ------------------------------------------------------------


def flops_min(A, E, cfg):
    """Minimal algebraic FLOPs per structure (concat-GEMM split into per-atom parts),
    SURVEY.md 8(d): per layer E*(4*d*d + 4*d) + A*(10*d*d); plus embed / basis / readout."""
    d, G, L = cfg["local_dim"], N_GAUSS, cfg["n_attention"]
    f = A * 2 * cfg["embedding_dim"] * d
    if cfg.get("g_update", False):
        f += 2 * E * 2 * G * d
        f += L * (E * (4 * d * d + 4 * d) + A * (10 * d * d))
    else:
        f += L * (E * (2 * G * d + 2 * d * d + 4 * d) + A * (6 * d * d))
    f += A * (2 * d * d + 4 * d * d) + 2 * A * A * d + 2 * d * d + 2 * d
    return int(f)
