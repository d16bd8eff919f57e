"""Importer for the reference's Keras full-model HDF5 checkpoints.

The reference saves ``ModelCheckpoint(filepath=".../models/model_<target>.h5", save_weights_only=False)``
(scann_model.py:166-177) and loads it with ``load_model(path, custom_objects=_CUSTOM_OBJECTS)`` (:79,87,323).  Such a file
holds, under ``/model_weights``, one group per Keras layer (attribute ``layer_names``), each with an attribute
``weight_names`` and one dataset per weight at ``<layer>/<weight name>``; the architecture sits in the root attribute
``model_config`` (JSON).  This module turns that into the weight container of this package (``scann_weight_name`` order,
INTEGRATION.md section 3) plus whatever hyper-parameters the file itself determines.

Name mapping (Keras 2.10 auto-names; the explicit names are given in create_model, scann_model.py:362-447):

====================================  =================================================================================
Keras layer (group)                   container tensors
====================================  =================================================================================
``embed_atom``                        ``embed_atom/embeddings`` (Embedding) or ``embed_atom/{kernel,bias}`` (cgcnn Dense)
``extra_embed``, ``dense_embed``,     same names, ``{kernel,bias}``
``neighbor_d``, ``neighbor_w``,
``after_Lc``, ``bf_property``,
``predict_property``
``local_attention[_k]``               ``local_attention_k/{query,key,filter_geo}/{kernel,bias}``; its two LayerNormalization
                                      sub-layers in creation order: first = ``layer_norm``, second = ``layer_norm_g``
                                      (attention.py:111-113)
``residual_norm[_k]``                 ``residual_norm_k/dense_1``, ``dense_2`` (the Sequential's two Dense layers in order,
                                      attention.py:25-31) and ``layer_norm``
``global_attention``                  ``global_attention/{query,key}/{kernel,bias}``
====================================  =================================================================================

Sub-layer weights are matched by the path component Keras derives from the sub-layer's ``name=`` (``query``, ``key``,
``filter_geo``) and, for the auto-named ones (``layer_normalization_7``, ``dense_3``: global counters whose values depend on
the whole build order), by their ORDER inside the layer, which is the order of attribute assignment in the layer's
``__init__`` and does not depend on those counters.  Every tensor is shape-checked against the architecture.

NOT verified against a file written by TensorFlow itself: TensorFlow / Keras are not installable here and the published
checkpoints (README.md:126) are not reachable offline.  The layout above follows the Keras 2.10 ``hdf5_format`` source and is
exercised on files written by h5py / libhdf5 (tests/test_keras_import.py).
"""
from __future__ import annotations

import json
import re
from collections import OrderedDict

import numpy as np

_PLAIN = ("extra_embed", "dense_embed", "neighbor_d", "neighbor_w", "after_Lc", "bf_property", "predict_property")


def _suffix_index(name, stem):
    """'local_attention' -> 0, 'local_attention_3' -> 3, anything else -> None."""
    if name == stem:
        return 0
    m = re.fullmatch(re.escape(stem) + r"_(\d+)", name)
    return int(m.group(1)) if m else None


def _leaf(wname):
    return wname.rsplit("/", 1)[-1].split(":")[0]


def _sublayer(wname):
    """'local_attention/layer_normalization_7/gamma:0' -> ('layer_normalization', 7); no numeric suffix -> (name, 0)."""
    parts = wname.split("/")
    sub = parts[-2] if len(parts) >= 2 else ""
    m = re.fullmatch(r"(.*?)_(\d+)", sub)
    return (m.group(1), int(m.group(2))) if m else (sub, 0)


def _pairs(weights, leaves, where=""):
    """The (leaves[0], leaves[1]) pairs among ``weights`` -- every (kernel, bias) or (gamma, beta) of the auto-named sub-layers --
    in the sub-layers' CREATION order.  Keras gives those sub-layers a global counter suffix (``layer_normalization_7``,
    ``dense_3``): where the weight names carry it, the pairs are ordered by that number (the name path, not the position in the
    file, identifies the sub-layer); a file whose order disagrees with the numbers is reported with a warning.  Names without
    a path fall back to the file order."""
    import warnings

    ws = [(n, a) for n, a in weights if _leaf(n) in leaves]
    out, i = [], 0
    while i + 1 < len(ws):
        if _leaf(ws[i][0]) == leaves[0] and _leaf(ws[i + 1][0]) == leaves[1]:
            out.append((_sublayer(ws[i][0]), _sublayer(ws[i + 1][0]), ws[i][1], ws[i + 1][1]))
            i += 2
        else:
            i += 1
    for s0, s1, _, _ in out:
        if s0 != s1:
            raise ValueError("%s: %s and %s of one pair come from different sub-layers (%s, %s)" % (where, leaves[0], leaves[1], s0, s1))
    named = all(s0[0] and "/" in n for (s0, _, _, _), (n, _) in zip(out, ws[::2]))
    if named and len({s0[0] for s0, _, _, _ in out}) == 1:
        by_number = sorted(out, key=lambda t: t[0][1])
        if [t[0] for t in by_number] != [t[0] for t in out]:
            warnings.warn("%s: sub-layers appear in the file as %s; using their creation order %s" %
                          (where, [t[0] for t in out], [t[0] for t in by_number]))
        out = by_number
    elif len(out) > 1:
        warnings.warn("%s: sub-layer names carry no creation counter; relying on the order of weight_names" % where)
    return [(a, b) for _, _, a, b in out]


def map_keras_weights(layers):
    """``layers``: ordered mapping Keras layer name -> list of (weight name, array) in the file's ``weight_names`` order.
    Returns the container dict (names of ``scann_weight_name``).  Layers without weights (inputs, Lambda, Dropout,
    GaussianExpansion, Multiply) are ignored; an unexpected weighted layer is an error."""
    out = OrderedDict()
    la, rn = {}, {}
    for lname, weights in layers.items():
        weights = [(n, np.asarray(a, dtype=np.float32)) for n, a in weights]
        if not weights:
            continue
        if lname == "embed_atom":
            for n, a in weights:
                out["embed_atom/" + _leaf(n)] = a  # embeddings | kernel, bias
        elif lname in _PLAIN:
            for n, a in weights:
                out[lname + "/" + _leaf(n)] = a
        elif _suffix_index(lname, "local_attention") is not None:
            la[_suffix_index(lname, "local_attention")] = weights
        elif _suffix_index(lname, "residual_norm") is not None:
            rn[_suffix_index(lname, "residual_norm")] = weights
        elif lname == "global_attention":
            for sub in ("query", "key"):
                for n, a in weights:
                    if "/" + sub + "/" in "/" + n:
                        out["global_attention/%s/%s" % (sub, _leaf(n))] = a
        else:
            raise ValueError("Keras layer %r with weights %s is not part of the SCANN graph" % (lname, [n for n, _ in weights]))
    # auto-numbered layers: Keras numbers them in creation order (k-th LocalAttention of the model = iteration k)
    for k, idx in enumerate(sorted(la)):
        weights = la[idx]
        p = "local_attention_%d/" % k
        for sub in ("query", "key", "filter_geo"):
            for n, a in weights:
                if "/" + sub + "/" in "/" + n:
                    out[p + "%s/%s" % (sub, _leaf(n))] = a
        if any("/value/" in "/" + n for n, _ in weights):
            raise ValueError("LocalAttention with v_proj=True is not what create_model builds (scann_model.py:396)")
        lns = _pairs(weights, ("gamma", "beta"), "local_attention %d" % k)
        if len(lns) not in (1, 2):
            raise ValueError("local_attention %d: expected 1 or 2 LayerNormalization sub-layers, found %d" % (k, len(lns)))
        out[p + "layer_norm/gamma"], out[p + "layer_norm/beta"] = lns[0]
        if len(lns) == 2:
            out[p + "layer_norm_g/gamma"], out[p + "layer_norm_g/beta"] = lns[1]
    for k, idx in enumerate(sorted(rn)):
        weights = rn[idx]
        p = "residual_norm_%d/" % k
        dense, lns = _pairs(weights, ("kernel", "bias"), "residual_norm %d" % k), _pairs(weights, ("gamma", "beta"), "residual_norm %d" % k)
        if len(dense) != 2 or len(lns) != 1:
            raise ValueError("residual_norm %d: expected two Dense layers and one LayerNormalization" % k)
        (out[p + "dense_1/kernel"], out[p + "dense_1/bias"]), (out[p + "dense_2/kernel"], out[p + "dense_2/bias"]) = dense
        out[p + "layer_norm/gamma"], out[p + "layer_norm/beta"] = lns[0]
    return out


def infer_model_config(weights, model_config=None):
    """Architecture keys of ``config['model']`` (and ``hyper.target`` hints) that the checkpoint itself determines: from the
    tensor shapes, and -- for what shapes cannot tell (use_ga_norm, gaussian_d, the mrelu head) -- from ``model_config``."""
    m = {}
    L = len({k.split("/")[0] for k in weights if k.startswith("local_attention_")})
    m["n_attention"] = L
    if "embed_atom/embeddings" in weights:
        m["feature"] = "atomic"
        m["n_atoms"], m["embedding_dim"] = (int(x) for x in weights["embed_atom/embeddings"].shape)
    else:
        m["feature"] = "cgcnn"
        m["embedding_dim"] = int(weights["embed_atom/kernel"].shape[1])
    m["use_ring"] = "extra_embed/kernel" in weights
    m["local_dim"] = int(weights["dense_embed/kernel"].shape[1])
    m["global_dim"] = int(weights["after_Lc/kernel"].shape[1])
    m["dense_out"] = int(weights["bf_property/kernel"].shape[1])
    m["use_attn_norm"] = any(k.startswith("residual_norm_") for k in weights)
    if L:
        m["g_update"] = weights["local_attention_0/filter_geo/kernel"].shape[0] == 3 * m["local_dim"]
    hints = {}
    if model_config:
        def walk(node):
            if isinstance(node, dict):
                cn, cfg = node.get("class_name"), node.get("config") if isinstance(node.get("config"), dict) else None
                if cn == "LocalAttention" and cfg:
                    m.setdefault("num_head", int(cfg.get("num_head", 8)))
                    m["use_drop"] = bool(cfg.get("dropout", False))
                elif cn == "GlobalAttention" and cfg:
                    m["use_ga_norm"] = bool(cfg.get("norm", True))
                elif cn == "GaussianExpansion" and cfg and "gaussian_d" not in m:
                    c = cfg.get("centers")
                    if isinstance(c, (list, tuple)) and c:
                        m["gaussian_d"] = float(c[-1])  # the first expansion is the distance one (scann_model.py:378)
                elif cn == "Dense" and cfg and cfg.get("name") == "predict_property":
                    hints["relu_out"] = "mrelu" in json.dumps(cfg.get("activation"))
                for v in node.values():
                    walk(v)
            elif isinstance(node, list):
                for v in node:
                    walk(v)
        walk(model_config)
    return m, hints


def _chunked_attr(attrs, name):
    """Keras splits a long name list over ``name0``, ``name1``, ... (hdf5_format.save_attributes_to_hdf5_group)."""
    if name in attrs:
        vals = list(np.asarray(attrs[name]).ravel())
    else:
        vals, i = [], 0
        while "%s%d" % (name, i) in attrs:
            vals.extend(np.asarray(attrs["%s%d" % (name, i)]).ravel())
            i += 1
    return [v.decode("utf-8") if isinstance(v, (bytes, np.bytes_)) else str(v) for v in vals]


def read_keras_layers(path):
    """-> (ordered mapping layer -> [(weight name, array)], model_config dict | None) of a Keras HDF5 checkpoint, read
    with the pure-Python HDF5 reader (no h5py needed)."""
    from ..utils.hdf5_lite import File

    f = File(path)
    root = f["model_weights"] if "model_weights" in f else f  # save_weights() files have the layer groups at the root
    layers = OrderedDict()
    for lname in _chunked_attr(root.attrs, "layer_names"):
        g = root[lname]
        layers[lname] = [(w, g[w].read()) for w in _chunked_attr(g.attrs, "weight_names")]
    mc = f.attrs.get("model_config")
    if isinstance(mc, (bytes, np.bytes_)):
        mc = mc.decode("utf-8")
    return layers, (json.loads(mc) if isinstance(mc, str) else None)


def load_keras_h5(path, config=None):
    """Keras checkpoint -> (config, weights) for ``HipModel(config, weights)``.  ``config`` (the training run's yaml, as
    predict_model.py reads ``config.yaml``) supplies what the file does not determine; what the file does determine wins."""
    layers, mc = read_keras_layers(path)
    weights = map_keras_weights(layers)
    inferred, hints = infer_model_config(weights, mc)
    cfg = {"model": dict((config or {}).get("model", {})), "hyper": dict((config or {}).get("hyper", {}))}
    cfg["model"].update(inferred)
    cfg["model"].setdefault("num_head", 8)
    if hints.get("relu_out"):
        cfg["hyper"]["target"] = "e_b"  # the only target with the mrelu head (scann_model.py:446)
    return cfg, dict(weights)


def expected_param_count(model):
    """Parameters of the graph ``create_model`` builds for these architecture keys (scann_model.py:329-453; the tensor list of
    ``scann_weight_name``), without a device."""
    d, dg, do, emb, L = model["local_dim"], model["global_dim"], model["dense_out"], model["embedding_dim"], model["n_attention"]
    n = 92 * emb + emb if model.get("feature") == "cgcnn" else model["n_atoms"] * emb
    cin = emb + (10 if model.get("use_ring") else 0)
    n += (2 * 10 + 10 if model.get("use_ring") else 0) + cin * d + d
    if model.get("g_update"):
        n += 2 * (20 * d + d)
    per = 2 * (d * d + d) + ((3 * d if model.get("g_update") else 20) * d + d) + 2 * d + (2 * d if model.get("g_update") else 0)
    if model.get("use_attn_norm"):
        per += 2 * (d * d + d) + 2 * d
    n += L * per
    return n + (d * dg + dg) + 2 * (dg * dg + dg) + (dg * do + do) + (do + 1)


def mapping_report(path, config=None):
    """What a maintainer wants to see the day a real reference checkpoint is at hand: every tensor of the
    file mapped to exactly one container name, none left over, the parameter count of the architecture the file implies,
    and the hyper-parameters read from it.  Raises on any mismatch; returns the report as a dict."""
    layers, mc = read_keras_layers(path)
    file_tensors = [(lname, wname, np.asarray(a, dtype=np.float32)) for lname, ws in layers.items() for wname, a in ws]
    weights = map_keras_weights(layers)
    inferred, hints = infer_model_config(weights, mc)

    def finger(a):
        a = np.asarray(a, dtype=np.float64)
        return (a.shape, float(a.sum()), float(np.abs(a).sum()), float(a.ravel()[0]) if a.size else 0.0)

    src = sorted(finger(a) for _, _, a in file_tensors)
    dst = sorted(finger(a) for a in weights.values())
    if src != dst:
        raise ValueError("checkpoint %s: %d tensors in the file, %d mapped -- the two sets differ" % (path, len(src), len(dst)))
    model = dict((config or {}).get("model", {}))
    model.update(inferred)
    n_param = int(sum(a.size for a in weights.values()))
    want = expected_param_count(model)
    if n_param != want:
        raise ValueError("checkpoint %s: %d parameters, the architecture it implies has %d" % (path, n_param, want))
    if not all(np.isfinite(a).all() for a in weights.values()):
        raise ValueError("checkpoint %s holds non-finite weights" % path)
    return {"tensors": len(src), "parameters": n_param, "model": inferred, "hints": hints,
            "map": [(lname + "/" + wname if not wname.startswith(lname) else wname, tuple(a.shape)) for lname, wname, a in file_tensors],
            "names": list(weights)}
