"""ctypes wrapper of the C/OpenMP oracle port (oracle/scann_oracle_c.c).  TEST INFRASTRUCTURE ONLY: used by
tests/ and by bench.py's cpu_baseline leg, never by the product path.  Supports the g_update=True, feature="atomic",
use_ring=False configurations (QM9, QM9-std, MP2018 shapes)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "libscann_oracle_c.so")
if not os.path.exists(_PATH):
    raise ImportError("oracle C port not built (make -C oracle)")
_lib = C.CDLL(_PATH)
_F = C.POINTER(C.c_float)


class _Layer(C.Structure):
    _fields_ = [(n, _F) for n in ("q_w", "q_b", "k_w", "k_b", "fg_w", "fg_b", "ln_g", "ln_b", "lng_g", "lng_b",
                                  "f1_w", "f1_b", "f2_w", "f2_b", "lnr_g", "lnr_b")]


class _Model(C.Structure):
    _fields_ = [("n_atoms", C.c_int), ("emb", C.c_int), ("n_attention", C.c_int), ("use_attn_norm", C.c_int),
                ("use_ga_norm", C.c_int), ("relu_out", C.c_int), ("gaussian_d", C.c_float)] + \
               [(n, _F) for n in ("embed", "de_w", "de_b", "nd_w", "nd_b", "nw_w", "nw_b")] + \
               [("layers", C.POINTER(_Layer))] + \
               [(n, _F) for n in ("al_w", "al_b", "gq_w", "gq_b", "gk_w", "gk_b", "bf_w", "bf_b", "pp_w", "pp_b")]


_lib.scann_oracle_forward.restype = C.c_int
_lib.scann_oracle_forward.argtypes = [C.POINTER(_Model), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]


def _p(a):
    return a.ctypes.data_as(_F)


def forward(config, weights, inputs):
    m = config["model"]
    if not m.get("g_update") or m.get("use_ring") or m.get("feature", "atomic") != "atomic" or m["local_dim"] != 128:
        raise NotImplementedError("C oracle port covers the g_update=True atomic-feature configurations")
    w = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in weights.items()}
    L = m["n_attention"]
    layers = (_Layer * max(L, 1))()
    for i in range(L):
        p, r = "local_attention_%d/" % i, "residual_norm_%d/" % i
        lw = layers[i]
        lw.q_w, lw.q_b, lw.k_w, lw.k_b = _p(w[p + "query/kernel"]), _p(w[p + "query/bias"]), _p(w[p + "key/kernel"]), _p(w[p + "key/bias"])
        lw.fg_w, lw.fg_b = _p(w[p + "filter_geo/kernel"]), _p(w[p + "filter_geo/bias"])
        lw.ln_g, lw.ln_b = _p(w[p + "layer_norm/gamma"]), _p(w[p + "layer_norm/beta"])
        lw.lng_g, lw.lng_b = _p(w[p + "layer_norm_g/gamma"]), _p(w[p + "layer_norm_g/beta"])
        if m.get("use_attn_norm", True):
            lw.f1_w, lw.f1_b, lw.f2_w, lw.f2_b = _p(w[r + "dense_1/kernel"]), _p(w[r + "dense_1/bias"]), _p(w[r + "dense_2/kernel"]), _p(w[r + "dense_2/bias"])
            lw.lnr_g, lw.lnr_b = _p(w[r + "layer_norm/gamma"]), _p(w[r + "layer_norm/beta"])
    mw = _Model(m["n_atoms"], m["embedding_dim"], L, int(m.get("use_attn_norm", True)), int(m.get("use_ga_norm", True)),
                int(config.get("hyper", {}).get("target") == "e_b"), float(m["gaussian_d"]),
                _p(w["embed_atom/embeddings"]), _p(w["dense_embed/kernel"]), _p(w["dense_embed/bias"]),
                _p(w["neighbor_d/kernel"]), _p(w["neighbor_d/bias"]), _p(w["neighbor_w/kernel"]), _p(w["neighbor_w/bias"]),
                layers, _p(w["after_Lc/kernel"]), _p(w["after_Lc/bias"]), _p(w["global_attention/query/kernel"]),
                _p(w["global_attention/query/bias"]), _p(w["global_attention/key/kernel"]), _p(w["global_attention/key/bias"]),
                _p(w["bf_property/kernel"]), _p(w["bf_property/bias"]), _p(w["predict_property/kernel"]), _p(w["predict_property/bias"]))
    atomic = np.ascontiguousarray(inputs["atomic"], dtype=np.int32)
    B, M = atomic.shape
    nbr = np.ascontiguousarray(inputs["neighbors"], dtype=np.int32)
    N = nbr.shape[2]
    if N > 256 or M > 1024:
        raise NotImplementedError("C oracle port: N <= 256, M <= 1024")
    am = np.ascontiguousarray(np.asarray(inputs["atom_mask"]).reshape(B, M), dtype=np.float32)
    nm = np.ascontiguousarray(inputs["neighbor_mask"], dtype=np.float32)
    nw = np.ascontiguousarray(inputs["neighbor_weight"], dtype=np.float32)
    nd = np.ascontiguousarray(inputs["neighbor_distance"], dtype=np.float32)
    y = np.empty(B, dtype=np.float32)
    ga = np.empty((B, M), dtype=np.float32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = _lib.scann_oracle_forward(C.byref(mw), B, M, N, vp(atomic), vp(am), vp(nbr), vp(nm), vp(nw), vp(nd), vp(y), vp(ga))
    if rc:
        raise MemoryError("scann_oracle_forward failed")
    return y.reshape(B, 1), ga.reshape(B, M, 1)
