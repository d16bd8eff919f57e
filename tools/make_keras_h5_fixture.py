#!/usr/bin/env python3
"""Write a weight container of this package (.npz, HipModel.save) as a Keras-2.10-style full-model HDF5 file -- the layout the
reference's ModelCheckpoint produces (scann_model.py:166-177) -- with h5py.  Test infrastructure for the importer
(scann/models/keras_import.py): needs an interpreter that has h5py (in this image: /opt/conda/bin/python3.9); the product
path never needs h5py.

  /opt/conda/bin/python3.9 tools/make_keras_h5_fixture.py model.npz model_keras.h5

Layer / weight names follow Keras' auto-naming for create_model (scann_model.py:329-453): unnamed LocalAttention /
ResidualNorm layers become local_attention, local_attention_1, ...; their LayerNormalization / Dense sub-layers are numbered
by global counters in creation order (per block: LocalAttention's two LayerNormalizations, then ResidualNorm's Sequential
Dense pair and its LayerNormalization).
"""
import json
import sys

import h5py
import numpy as np


def main(src, dst):
    z = np.load(src, allow_pickle=False)
    cfg = json.loads(str(z["__config__"]))
    w = {k: z[k] for k in z.files if k != "__config__"}
    m = cfg["model"]
    L = int(m["n_attention"])
    sfx = lambda stem, k: stem if k == 0 else "%s_%d" % (stem, k)  # noqa: E731
    layers = []  # (keras layer name, class name, layer config, [(weight name, array)])
    for nm in ("atomic", "atom_mask", "neighbors", "neighbor_mask", "neighbor_weight", "neighbor_distance"):
        layers.append((nm, "InputLayer", {"name": nm}, []))
    if "embed_atom/embeddings" in w:
        layers.append(("embed_atom", "Embedding", {"name": "embed_atom"}, [("embed_atom/embeddings:0", w["embed_atom/embeddings"])]))
    else:
        layers.append(("embed_atom", "Dense", {"name": "embed_atom"}, [("embed_atom/kernel:0", w["embed_atom/kernel"]), ("embed_atom/bias:0", w["embed_atom/bias"])]))
    plain = lambda n, act=None: (n, "Dense", {"name": n, "activation": act}, [("%s/kernel:0" % n, w[n + "/kernel"]), ("%s/bias:0" % n, w[n + "/bias"])])  # noqa: E731
    if m.get("use_ring"):
        layers.append(plain("extra_embed"))
    layers.append(plain("dense_embed", "swish"))
    layers.append(("dropout", "Dropout", {"name": "dropout", "rate": 0.1}, []))
    layers.append(("get_neighbor", "Lambda", {"name": "get_neighbor"}, []))
    gd = float(m["gaussian_d"])
    layers.append(("gaussian_expansion", "GaussianExpansion", {"name": "gaussian_expansion", "centers": np.linspace(0, gd, 20, dtype="float32").tolist()}, []))
    if m["g_update"]:
        layers.append(plain("neighbor_d", "swish"))
        layers.append(("gaussian_expansion_1", "GaussianExpansion", {"name": "gaussian_expansion_1", "centers": np.linspace(0, np.pi * 2, 20, dtype="float32").tolist()}, []))
        layers.append(plain("neighbor_w", "swish"))
        layers.append(("geometry_features", "Multiply", {"name": "geometry_features"}, []))
    ln = dn = 0  # global auto-name counters of LayerNormalization / Dense sub-layers
    for k in range(L):
        name, p = sfx("local_attention", k), "local_attention_%d/" % k
        ws = []
        for sub in ("query", "key", "filter_geo"):
            ws += [("%s/%s/kernel:0" % (name, sub), w[p + sub + "/kernel"]), ("%s/%s/bias:0" % (name, sub), w[p + sub + "/bias"])]
        ws += [("%s/%s/gamma:0" % (name, sfx("layer_normalization", ln)), w[p + "layer_norm/gamma"]),
               ("%s/%s/beta:0" % (name, sfx("layer_normalization", ln)), w[p + "layer_norm/beta"])]
        ln += 1
        if m["g_update"]:
            ws += [("%s/%s/gamma:0" % (name, sfx("layer_normalization", ln)), w[p + "layer_norm_g/gamma"]),
                   ("%s/%s/beta:0" % (name, sfx("layer_normalization", ln)), w[p + "layer_norm_g/beta"])]
            ln += 1
        layers.append((name, "LocalAttention", {"name": name, "dim": m["local_dim"], "scale": 0.5, "num_head": m["num_head"], "v_proj": False,
                                               "kq_proj": True, "g_update": bool(m["g_update"]), "dropout": bool(m.get("use_drop", False))}, ws))
        if m["use_attn_norm"]:
            name, p = sfx("residual_norm", k), "residual_norm_%d/" % k
            ws = []
            for j in (1, 2):
                ws += [("%s/%s/kernel:0" % (name, sfx("dense", dn)), w[p + "dense_%d/kernel" % j]), ("%s/%s/bias:0" % (name, sfx("dense", dn)), w[p + "dense_%d/bias" % j])]
                dn += 1
            ws += [("%s/%s/gamma:0" % (name, sfx("layer_normalization", ln)), w[p + "layer_norm/gamma"]),
                   ("%s/%s/beta:0" % (name, sfx("layer_normalization", ln)), w[p + "layer_norm/beta"])]
            ln += 1
            layers.append((name, "ResidualNorm", {"name": name, "dim": m["local_dim"], "dropout": 0.1}, ws))
    layers.append(plain("after_Lc", "swish"))
    ws = []
    for sub in ("query", "key"):
        ws += [("global_attention/%s/kernel:0" % sub, w["global_attention/%s/kernel" % sub]), ("global_attention/%s/bias:0" % sub, w["global_attention/%s/bias" % sub])]
    layers.append(("global_attention", "GlobalAttention", {"name": "global_attention", "dim": m["global_dim"], "norm": bool(m["use_ga_norm"]),
                                                            "v_proj": False, "kq_proj": True}, ws))
    layers.append(plain("bf_property", "swish"))
    head = plain("predict_property", "mrelu" if cfg["hyper"].get("target") == "e_b" else "linear")
    layers.append(head)

    f = h5py.File(dst, "w")
    f.attrs["keras_version"] = "2.10.0"
    f.attrs["backend"] = "tensorflow"
    f.attrs["model_config"] = json.dumps({"class_name": "Functional", "config": {
        "name": "model", "layers": [{"class_name": c, "name": n, "config": lc} for n, c, lc, _ in layers]}})
    g = f.create_group("model_weights")
    g.attrs["layer_names"] = np.array([n.encode("utf8") for n, _, _, _ in layers])
    g.attrs["backend"] = "tensorflow".encode("utf8")
    g.attrs["keras_version"] = "2.10.0".encode("utf8")
    for n, _, _, ws in layers:
        lg = g.create_group(n)
        lg.attrs["weight_names"] = np.array([wn.encode("utf8") for wn, _ in ws]) if ws else np.array([], dtype="S1")
        for wn, a in ws:
            d = lg.create_dataset(wn, a.shape, dtype=a.dtype)  # what Keras' save_weights_to_hdf5_group does
            if a.shape:
                d[:] = a
            else:
                d[()] = a
    f.close()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
