#!/usr/bin/env python3
"""Where does the plain-fp32 (generic-width) backward leave the fp32 floor?  (run on the GPU box.)

  python3 tools/debug_plain_grads.py 7797 [1847 1111 ...]

Regenerates batch n of tests/manual/fuzz_grads.py (same generator, same seeds), runs the plain-fp32 training kernels on it and reads the
tensors of the readout stage (scann_train_debug_read) next to fp64 autograd of the torch graph with the library's Dropout masks
(tests/torch_ref.py, capture=...), stage by stage:
  forward   z, gq, gk, rep          -- what the backward is seeded with
  backward  d rep, d gq, d gk, d z  -- per STRUCTURE, distance from fp64 relative to the tensor's rms over the batch
and, to split "the kernel's arithmetic" from "the kernel's inputs": the pooling backward (attention.py:279-316) evaluated in fp64 NumPy on
the plain path's OWN fp32 gq / gk / d rep -- if that reproduces the kernel's d gq / d gk, the error came in with the inputs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), os.path.join(ROOT, "oracle"), ROOT, os.path.join(ROOT, "tests", "manual"), os.path.join(ROOT, "tests")]
import scann_oracle as so
from scann import _hip
from scann.models.scann_model import HipModel, normalize_config
from fuzz_parity_lib import random_batch
import torch_ref


def fuzz_batch(index, g_update):
    """Batch `index` of fuzz_grads.py: the generator's draws are replayed (no GPU work for the batches before it)."""
    rng = np.random.default_rng(5)
    n = 0
    while True:
        inputs, targets = random_batch(rng, g_update, big=(n % 7 == 0), max_struct=1 if n % 5 == 0 else 8)
        pk = _hip.pack_inputs(inputs)
        if np.any(np.diff(pk.mol_offset) == 1):
            continue
        if n == index:
            return pk, targets, (0.1 if n % 2 else 0.0), 100 + n
        n += 1


def pool_bwd64(pk, gq, gk, drep, use_norm=True):
    """GlobalAttention pooling backward in fp64 NumPy on the given (fp32) inputs: d gq, d gk."""
    gq, gk, drep = (np.asarray(x, np.float64) for x in (gq, gk, drep))
    dgq, dgk = np.zeros_like(gq), np.zeros_like(gk)
    for s in range(pk.n_struct):
        a0, a1 = pk.mol_offset[s], pk.mol_offset[s + 1]
        q, k, dr = gq[a0:a1], gk[a0:a1], drep[s]
        S = q.sum(0)
        agg = np.einsum("ik,ik->i", k, S[None, :] - q)
        nrm = np.linalg.norm(agg) if use_norm else 1.0
        u = agg / nrm
        e = np.exp(u - u.max())
        at = e / e.sum()
        dat = k @ dr
        du = at * (dat - (at * dat).sum())
        da = (du - u * (u * du).sum()) / nrm if use_norm else du
        dgk[a0:a1] = at[:, None] * dr[None, :] + da[:, None] * (S[None, :] - q)
        W = (da[:, None] * k).sum(0)
        dgq[a0:a1] = W[None, :] - da[:, None] * k
    return dgq, dgk


def per_struct(pk, got, ref, rows_are_atoms=True):
    scale = float(np.sqrt(np.mean(ref ** 2))) + 1e-300
    out = []
    for s in range(pk.n_struct):
        sl = slice(pk.mol_offset[s], pk.mol_offset[s + 1]) if rows_are_atoms else slice(s, s + 1)
        out.append(float(np.max(np.abs(got[sl] - ref[sl]))) / scale)
    return out


def main():
    cfg = normalize_config(so.default_config("qm9"))
    cfg["model"].update(n_attention=3)
    w = so.init_weights(cfg, 77, perturb=True)
    dg = cfg["model"]["global_dim"]
    os.environ["SCANN_GENERIC"] = "1"
    plain = HipModel(cfg, w, device=0)
    os.environ.pop("SCANN_GENERIC")
    plain.engine.train_begin()
    os.environ["SCANN_TRAIN_FUSED"] = "0"
    modular = HipModel(cfg, w, device=0)
    modular.engine.train_begin()
    for index in [int(a) for a in sys.argv[1:]] or [7797]:
        pk, targets, drop, seed = fuzz_batch(index, cfg["model"]["g_update"])
        sizes = np.diff(pk.mol_offset)
        print("==== batch %d: %d structures, atoms per structure %s, %d edges, dropout %.1f, seed %d" % (index, pk.n_struct, sizes.tolist(), pk.n_edge, drop, seed))
        dr = (seed, drop) if drop else None
        cap64, cap32 = {}, {}
        g64 = torch_ref.loss_and_grads(cfg, w, pk, targets, drop=dr, capture=cap64)[2]
        g32 = torch_ref.loss_and_grads(cfg, w, pk, targets, drop=dr, dtype="float32", capture=cap32)[2]
        grads = {}
        for name, m in (("plain", plain), ("modular", modular)):
            eng = m.engine
            rb = eng.upload(pk)
            sse = eng.train_forward(rb, targets, dropout=drop, seed=seed)
            y = np.asarray(eng.download(rb)[0], np.float64).ravel()
            eng.zero_grads()
            eng.train_backward(rb, sse, pk.n_struct)
            grads[name] = eng.get_grads()
            if name == "plain":
                t = {k: eng.train_debug_read(rb, k, dg).astype(np.float64) for k in ("z", "gq", "gk", "rep", "drep", "dgq", "dgk", "dz")}
            print(" y %-8s %s" % (name, np.array2string(y, precision=7)))
            rb.free()
        print(" -- parameter gradients, max |g - fp64| / rms(fp64): plain / modular / torch fp32")
        for k in ("predict_property/kernel", "bf_property/kernel", "global_attention/key/kernel", "global_attention/query/kernel", "after_Lc/kernel",
                  "residual_norm_2/dense_2/kernel", "local_attention_0/query/kernel", "dense_embed/kernel"):
            reg = 2e-4 * w[k].astype(np.float64) if k.endswith(torch_ref.REGULARIZED) else 0.0
            r = g64[k] - reg
            sc = float(np.sqrt(np.mean(r ** 2))) + 1e-300
            e = lambda g: float(np.max(np.abs(np.asarray(g, np.float64).reshape(r.shape) - r))) / sc
            print("   %-36s %.2e  %.2e  %.2e" % (k, e(grads["plain"][k]), e(grads["modular"][k]), e(g32[k] - reg)))
        print(" -- forward of the readout stage, per structure: plain | torch fp32   (max |x - fp64| / rms)")
        for k in ("z", "gq", "gk"):
            print("   %-5s plain %s" % (k, " ".join("%.1e" % v for v in per_struct(pk, t[k], cap64[k][0]))))
            print("   %-5s fp32  %s" % (k, " ".join("%.1e" % v for v in per_struct(pk, cap32[k][0], cap64[k][0]))))
        print("   rep   plain %s" % " ".join("%.1e" % v for v in per_struct(pk, t["rep"], cap64["rep"][0], False)))
        print("   rep   fp32  %s" % " ".join("%.1e" % v for v in per_struct(pk, cap32["rep"][0], cap64["rep"][0], False)))
        print(" -- backward, per structure: plain | torch fp32 | fp64 pooling backward on the plain path's own fp32 inputs")
        print("   drep  plain %s" % " ".join("%.1e" % v for v in per_struct(pk, t["drep"], cap64["rep"][1], False)))
        print("   drep  fp32  %s" % " ".join("%.1e" % v for v in per_struct(pk, cap32["rep"][1], cap64["rep"][1], False)))
        own_q, own_k = pool_bwd64(pk, t["gq"], t["gk"], t["drep"], cfg["model"]["use_ga_norm"])
        ref_q, ref_k = pool_bwd64(pk, cap64["gq"][0], cap64["gk"][0], cap64["rep"][1], cfg["model"]["use_ga_norm"])
        print("   (check of the NumPy formula against autograd: dgq %.1e dgk %.1e)" % (np.max(np.abs(ref_q - cap64["gq"][1])) / np.sqrt(np.mean(ref_q ** 2)),
                                                                                        np.max(np.abs(ref_k - cap64["gk"][1])) / np.sqrt(np.mean(ref_k ** 2))))
        for k, own in (("dgq", own_q), ("dgk", own_k)):
            ref = cap64[k[1:]][1]
            print("   %-5s plain %s" % (k, " ".join("%.1e" % v for v in per_struct(pk, t[k], ref))))
            print("   %-5s fp32  %s" % (k, " ".join("%.1e" % v for v in per_struct(pk, cap32[k[1:]][1], ref))))
            print("   %-5s own64 %s   <- fp64 arithmetic on the plain path's inputs" % (k, " ".join("%.1e" % v for v in per_struct(pk, own, ref))))
            print("   %-5s kernel vs own64 %s" % (k, " ".join("%.1e" % v for v in per_struct(pk, t[k], own))))
        # which INPUT carries the error: the fp64 formula on the fp64 inputs with ONE of them replaced by the plain path's fp32 tensor
        for who, args in (("gq", (t["gq"], cap64["gk"][0], cap64["rep"][1])), ("gk", (cap64["gq"][0], t["gk"], cap64["rep"][1])),
                          ("drep", (cap64["gq"][0], cap64["gk"][0], t["drep"]))):
            one_q, _ = pool_bwd64(pk, *args, cfg["model"]["use_ga_norm"])
            print("   dgq with only %-4s from the plain path: %s" % (who, " ".join("%.1e" % v for v in per_struct(pk, one_q, ref_q))))
        for who, args in (("gq", (cap32["gq"][0], cap64["gk"][0], cap64["rep"][1])), ("gk", (cap64["gq"][0], cap32["gk"][0], cap64["rep"][1])),
                          ("drep", (cap64["gq"][0], cap64["gk"][0], cap32["rep"][1]))):
            one_q, _ = pool_bwd64(pk, *args, cfg["model"]["use_ga_norm"])
            print("   dgq with only %-4s from the torch fp32 graph: %s" % (who, " ".join("%.1e" % v for v in per_struct(pk, one_q, ref_q))))
        out_dir = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        np.savez(os.path.join(out_dir, "r6_plain_batch_%d.npz" % index), mol_offset=pk.mol_offset, atomic=pk.atomic, edge_offset=pk.edge_offset,
                 **{"plain_" + k: v for k, v in t.items()}, **{"f64_" + k: v[0] for k, v in cap64.items()}, **{"f64_d" + k: v[1] for k, v in cap64.items()},
                 **{"f32_" + k: v[0] for k, v in cap32.items()}, **{"f32_d" + k: v[1] for k, v in cap32.items()})
        # sensitivity of the pooling backward to its inputs: perturb the fp64 inputs by one fp32 rounding each and look at d gq
        rng = np.random.default_rng(0)
        pert = lambda x: x * (1.0 + rng.uniform(-6e-8, 6e-8, x.shape))
        pq, pk_ = pool_bwd64(pk, pert(cap64["gq"][0]), pert(cap64["gk"][0]), pert(cap64["rep"][1]), cfg["model"]["use_ga_norm"])
        print("   condition: one fp32 rounding on gq / gk / d rep moves dgq by %s" % " ".join("%.1e" % v for v in per_struct(pk, pq, ref_q)))


if __name__ == "__main__":
    main()
