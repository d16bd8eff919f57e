#!/usr/bin/env python3
"""QM9-shaped forward throughput of the SCANN+ HIP path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one forward of the whole graph (scann_model.py:329-453) over one batch of 128 synthetic QM9-shaped
molecules (configs/model_qm9.yaml: 7 local-attention layers, d=128, 8 heads, g_update) whose packed inputs are
already resident in HBM.  The engine fuses --group resident batches into one launch sequence (the packed layout has
no per-batch padding, so a group is the concatenation of its batches; EXACTLY --steps batches are processed);
the timed region is bracketed by barrier + device sync; value = molecules of all ranks / max-over-ranks time.
Inference shards by structure with no data-path collective ("weak" scaling: per-GPU work fixed).

Rank 0 prints ONE JSON line with `roofline` (dominant kernel = edge_kernel, fp32 MFMA bound; achieved =
algorithmic FLOPs per launch / HIP-event launch duration) and `cpu_baseline` (the NumPy/C oracle timed on the
host cores -- a reported baseline, not the product path).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "scann--material_amd"))

from scann import _hip  # noqa: E402  (loads libscann_hip.so; no torch / no oracle on the product path)
from scann.models.scann_model import HipModel, normalize_config  # noqa: E402

D = 128
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
QM9_MODEL = dict(n_atoms=10, embedding_dim=48, n_attention=7, local_dim=128, num_head=8, global_dim=128,
                 dense_out=128, scale=0.5, use_attn_norm=True, use_ga_norm=True, use_ring=False, g_update=True,
                 gaussian_d=4.0)  # configs/model_qm9.yaml:1-14


MP2018_MODEL = dict(n_atoms=95, embedding_dim=128, n_attention=9, local_dim=128, num_head=8, global_dim=128,
                    dense_out=128, scale=0.5, use_attn_norm=True, use_ga_norm=True, use_ring=False, g_update=True,
                    gaussian_d=6.0)  # configs/model_mp2018.yaml:1-14


def synth_packed_crystals(rng, n_struct):
    """MP2018-shaped structures: A ~ clip(LogNormal(3.0, 0.8), 2, 300), neighbours per atom U[6, 24],
    Z in [1, 94], distances U(1.5, 6.0)."""
    atomic, mol_off, e_off, cols = [], [0], [0], []
    for _ in range(n_struct):
        A = int(np.clip(np.rint(rng.lognormal(3.0, 0.8)), 2, 300))
        base = mol_off[-1]
        atomic.append(rng.integers(1, 95, size=A))
        lo, hi = min(6, A - 1), min(24, A - 1)
        deg = rng.integers(lo, hi + 1, size=A)
        for a in range(A):
            others = rng.choice(A - 1, size=deg[a], replace=False)
            cols.append(base + others + (others >= a))
            e_off.append(e_off[-1] + int(deg[a]))
        mol_off.append(base + A)
    cols = np.concatenate(cols)
    E = cols.shape[0]
    return _hip.PackedBatch(np.concatenate(atomic), mol_off, e_off, cols, rng.uniform(1.5, 6.0, size=E),
                            rng.uniform(0.4, 3.5, size=E))


def synth_packed_batch(rng, n_mol, worst=False):
    """Synthetic QM9-shaped molecules straight into packed form: atoms ~ clip(round(N(18,2.9)),3,29),
    species {H .51, C .35, N .06, O .08, F .002}, neighbours per atom ~ U[3, min(12, A-1)] without replacement,
    distance ~ U(0.9, 4.0), solid angle ~ U(0.4, 3.5)."""
    zs = np.array([1, 6, 7, 8, 9])
    ps = np.array([0.51, 0.35, 0.06, 0.08, 0.002])
    ps = ps / ps.sum()
    atomic, mol_off, e_off, cols = [], [0], [0], []
    for _ in range(n_mol):
        A = 29 if worst else int(np.clip(np.rint(rng.normal(18.0, 2.9)), 3, 29))
        base = mol_off[-1]
        atomic.append(rng.choice(zs, size=A, p=ps))
        hi = min(12, A - 1)
        deg = np.full(A, 12) if worst else rng.integers(3, hi + 1, size=A)
        keys = rng.random((A, A))
        keys[np.arange(A), np.arange(A)] = 2.0  # never pick self
        order = np.argsort(keys, axis=1)
        for a in range(A):
            cols.append(base + order[a, : deg[a]])
            e_off.append(e_off[-1] + int(deg[a]))
        mol_off.append(base + A)
    cols = np.concatenate(cols)
    E = cols.shape[0]
    return _hip.PackedBatch(np.concatenate(atomic), mol_off, e_off, cols,
                            rng.uniform(0.9, 4.0, size=E), rng.uniform(0.4, 3.5, size=E))


def edge_flops(E):
    """Algorithmic FLOPs of one edge_kernel launch: per edge the geometry third of filter_geo (2 d^2), the key
    projection (2 d^2) and the q.k / attn.k contractions (4 d) -- SURVEY.md 8(d) minimal form, edge part."""
    return E * (4 * D * D + 4 * D)


def total_flops_min(A, E, L=7, emb=48):
    f = A * 2 * emb * D + 2 * E * 2 * 20 * D + L * (E * (4 * D * D + 4 * D) + A * 10 * D * D)
    return f + A * 6 * D * D + 2 * D * D + 2 * D  # + readout (GA pair term is per-structure A^2, omitted: <1 %)


def host_cores():
    """CPU cores this process may actually use: min(affinity mask, cgroup v2/v1 CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(seconds_budget=12.0):
    """The oracle (checker) timed on this host's cores on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import scann_oracle as so

    cfg = so.default_config("qm9")
    w = so.init_weights(cfg, 1234)
    cores = host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)  # before the OpenMP runtime of the C port starts
    try:
        import scann_oracle_c as soc  # C/OpenMP port of the same padded-dense algorithm

        de, dn = so.synth_dataset(128, 0)
        inputs, _ = so.pad_batch(de, dn, True)
        soc.forward(cfg, w, inputs)  # warm
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds_budget:
            soc.forward(cfg, w, inputs)
            n += 128
        dt = time.perf_counter() - t0
        return {"value": n / dt, "unit": "molecules/s", "cores": cores, "kind": "port",
                "sample": "%d molecules (batches of 128, QM9-shaped seed 0), C/OpenMP fp32 restatement of the "
                          "reference's padded-dense graph, %d threads" % (n, cores)}
    except ImportError:
        pass
    de, dn = so.synth_dataset(128, 0)
    inputs, _ = so.pad_batch(de, dn, True)
    n, t0 = 0, time.perf_counter()
    while n == 0 or time.perf_counter() - t0 < seconds_budget:
        so.forward(cfg, w, inputs, np.float32)
        n += 128
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "molecules/s", "cores": cores, "kind": "port",
            "sample": "%d molecules (batches of 128, QM9-shaped seed 0), NumPy fp32 restatement of the reference's "
                      "padded-dense graph (BLAS threads = host cores)" % n}


def train_bench(args, eng, rank, world):
    """Weak-scaling training throughput: every rank trains on its own --batch molecules per step; the ranks exchange the
    scalar SSE/count and one flat fp32 gradient all-reduce per step over RCCL."""
    from scann.models.trainer import Communicator

    eng.train_begin()
    comm = Communicator(eng)  # gloo only carries the 128-byte ncclUniqueId; gradients go over RCCL
    rng = np.random.default_rng(2000 + rank)
    pool = [eng.upload(synth_packed_batch(rng, args.batch)) for _ in range(8)]
    targets = [rng.normal(size=args.batch).astype(np.float32) for _ in pool]

    def step(i):
        rb, t = pool[i % 8], targets[i % 8]
        sse = eng.train_forward(rb, t, dropout=0.1, seed=i)
        sse_g, cnt_g = comm.sum_pair(sse, args.batch)
        eng.zero_grads()
        eng.train_backward(rb, sse_g, cnt_g)
        eng.allreduce_grads()
        eng.adam_step(5e-4 / (1.0 + 1e-5 * i))

    steps, warm = min(args.steps, 400), min(args.warmup, 20)
    for i in range(warm):
        step(i)
    eng.sync()
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warm + i)
    eng.sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch
        import torch.distributed as dist

        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "QM9 molecules/s training (forward + backward + Adam)", "value": world * steps * args.batch / elapsed,
            "unit": "molecules/s", "n_gpus": world, "steps": steps, "warmup": warm, "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[2]: QM9 training, configs/model_qm9.yaml, %d molecules per GPU per step, dropout 0.1, "
                                   "RCCL flat gradient all-reduce (%d floats)" % (args.batch, eng.param_count()),
                       "global_batch": world * args.batch, "parallelism": "dp%d" % world}}))
    for rb in pool:
        rb.free()
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--pool", type=int, default=128, help="distinct resident batches cycled through")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("SCANN_STREAMS", "1")),
                    help="HIP streams the launch sequences are spread over.  Default 1: launches do not overlap, so the "
                         "HIP-event launch durations taken inside the timed region are the kernel's own (what rocprofv3 "
                         "--stats reports); 2 streams x --group 8 is ~6 %% faster end to end but co-schedules kernels")
    ap.add_argument("--group", type=int, default=int(os.environ.get("SCANN_BENCH_GROUP", "16")),
                    help="resident 128-molecule batches the engine fuses into one launch sequence (packed layout: a group is "
                         "the concatenation of its batches; 1 = one batch per launch)")
    ap.add_argument("--worst", action="store_true", help="Swc: every molecule 29 atoms x 12 neighbours")
    ap.add_argument("--config", default="qm9", choices=["qm9", "mp2018"],
                    help="qm9 = BASELINE configs[1] (the metric); mp2018 = configs[3] shapes (crystals, L=9, batch 64), extra")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--one-batch-ref", action="store_true",
                    help="also time single-batch launch sequences (reported as one_batch_per_launch)")
    ap.add_argument("--train", action="store_true",
                    help="extra (BASELINE configs[2]): time data-parallel TRAINING steps instead of the forward metric -- "
                         "forward(train, dropout 0.1) + SSE all-reduce + backward + flat RCCL gradient all-reduce + Adam; "
                         "--batch molecules per GPU per step")
    ap.add_argument("--profile-reps", type=int, default=20)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

    os.environ["SCANN_STREAMS"] = str(max(1, args.streams))  # read by scann_create
    if args.config == "mp2018" and args.batch == 128:
        args.batch = 64  # configs/model_mp2018.yaml:16
    model_cfg = dict(QM9_MODEL) if args.config == "qm9" else dict(MP2018_MODEL)
    cfg = normalize_config({"model": model_cfg, "hyper": {"target": "homo", "batch_size": args.batch}})
    ndev = _hip.load_library().scann_device_count()
    if ndev <= 0:
        raise SystemExit("bench.py needs a GPU: libscann_hip has no CPU fallback")
    model = HipModel(cfg, device=local % ndev, seed=1234)  # random-init weights of the QM9 architecture
    eng = model.engine
    if args.train:
        return train_bench(args, eng, rank, world)
    nstream = eng.num_streams()
    G = max(1, args.group)
    n_groups = max(nstream, (max(args.pool // G, 1) + nstream - 1) // nstream * nstream)
    rng = np.random.default_rng(1000 + rank)
    if args.config == "mp2018":
        batches = [synth_packed_crystals(rng, args.batch) for _ in range(n_groups * G)]
    else:
        batches = [synth_packed_batch(rng, args.batch, args.worst) for _ in range(n_groups * G)]
    # every step is one 128-molecule batch; the engine runs G of them per launch sequence
    pool = [eng.upload(_hip.concat_packed(batches[i * G:(i + 1) * G]) if G > 1 else batches[i]) for i in range(n_groups)]
    pool_n = n_groups
    tails = {}  # remainder groups so that EXACTLY the requested number of steps is executed
    for k in {args.steps % G, args.warmup % G} - {0}:
        tails[k] = eng.upload(_hip.concat_packed(batches[:k]))
    mols_per_step = args.batch

    dist = None
    if world > 1:  # coordination only (barrier + max of the timings); never touches the GPU through torch
        import torch
        import torch.distributed as dist_mod

        dist_mod.init_process_group("gloo", rank=rank, world_size=world)
        dist = dist_mod

    def barrier():
        if dist is not None:
            dist.barrier()

    def run(nsteps):
        for i in range(nsteps // G):
            eng.forward_resident(pool[i % pool_n], i % nstream)
        if nsteps % G:
            eng.forward_resident(tails[nsteps % G], 0)

    run(args.warmup)
    eng.sync()
    barrier()
    if rank == 0:
        eng.edge_timing(4)  # HIP events around the edge-kernel launches of every 4th forward, on their own streams
    t0 = time.perf_counter()
    run(args.steps)
    t_issue = time.perf_counter() - t0  # host time to enqueue every launch (diagnostic: host- vs device-bound)
    eng.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel launch durations: (a) sampled live inside the timed region (co-scheduled with the other streams: the
    # figure rocprofv3 --stats reports too), (b) sequential launches alone on the chip (scann_forward_profile)
    roof = None
    if rank == 0:
        live_us, live_n, live_edges = eng.edge_timing_read()
        eng.edge_timing(0)
        ms_edge = n_edge = fl = 0.0
        prof_tot = []
        for i in range(args.profile_reps):
            rb = pool[i % pool_n]
            p = eng.profile(rb)
            if i >= 2:  # first reps warm the caches
                ms_edge += p["ms_edge"]
                n_edge += p["n_edge_launch"]
                fl += edge_flops(rb.packed.n_edge) * p["n_edge_launch"]
                prof_tot.append(p)
        avg_ms = ms_edge / max(n_edge, 1)
        exclusive = fl / max(n_edge, 1) / (avg_ms * 1e-3) / 1e12
        achieved = edge_flops(live_edges) / (live_us * 1e-6) / 1e12 if live_n else exclusive
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "edge_kernel_traffic.json")
        if os.path.exists(tfile):
            tj = json.load(open(tfile))
            if tj.get("batches_per_launch") == G and not args.worst:  # PMC passes were taken at this launch size
                traffic = tj.get("hbm_bytes_per_launch")
        roof = {"bound": "mfma", "kernel": "%s (scann_kernels.hip), %d batches per launch" % ("edge_kernel_lean" if os.environ.get("SCANN_EDGE_LEAN", "1") != "0" else "edge_kernel_w8", G), "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                "avg_launch_us": live_us if live_n else avg_ms * 1e3, "launches_sampled": live_n,
                "note": "achieved = HIP events around sampled edge-kernel launches inside the timed region (with 1 stream "
                        "launches never overlap); exclusive = the same kernel in the separate profiling pass",
                "achieved_exclusive": exclusive, "exclusive_launch_us": avg_ms * 1e3,
                "per_forward_ms": {k: float(np.mean([p[k] for p in prof_tot])) for k in
                                   ("ms_basis", "ms_atom", "ms_edge", "ms_readout", "ms_total")}}

    if rank == 0:
        A = float(np.mean([b.n_atom for b in batches]))
        E = float(np.mean([b.n_edge for b in batches]))
        value = world * args.steps * mols_per_step / elapsed
        L_cfg, emb_cfg = model_cfg["n_attention"], model_cfg["embedding_dim"]
        out = {
            "metric": "QM9 molecules/s forward" if args.config == "qm9" else "MP2018-shaped structures/s forward", "value": value, "unit": "molecules/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("configs[1]: QM9-shaped%s, configs/model_qm9.yaml (SCANN+, L=7, d=128, H=8), "
                                    "batch=128 per step, forward" % (" worst-case 29x12" if args.worst else ""))
                       if args.config == "qm9" else
                       "configs[3] shapes: MP2018-shaped crystals, configs/model_mp2018.yaml (SCANN+, L=9), batch=64 per step, forward",
                       "batch": args.batch, "atoms_per_batch": A, "edges_per_batch": E,
                       "streams": nstream, "batches_fused_per_launch": G, "parallelism": "dp%d (independent shards, no collective)" % world},
            "host_issue_ms_per_step": t_issue / args.steps * 1e3,
            "whole_path_tflops_min": world * args.steps * total_flops_min(A, E, L_cfg, emb_cfg) / elapsed / 1e12,
            "roofline": roof,
        }
        if args.one_batch_ref and world == 1 and G > 1 and not args.worst:
            # the same engine with exactly ONE 128-molecule batch per launch sequence (no fusing), for reference; opt-in so that
            # the default command's rocprofv3 per-kernel averages cover the fused launches only
            singles = [eng.upload(b) for b in batches[:min(len(batches), 4 * nstream, 32)]]
            n1 = 400
            for i in range(40):
                eng.forward_resident(singles[i % len(singles)], i % nstream)
            eng.sync()
            t1 = time.perf_counter()
            for i in range(n1):
                eng.forward_resident(singles[i % len(singles)], i % nstream)
            eng.sync()
            out["one_batch_per_launch"] = {"value": n1 * mols_per_step / (time.perf_counter() - t1), "unit": "molecules/s",
                                           "steps": n1, "streams": nstream}
            for rb in singles:
                rb.free()
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    for rb in list(pool) + list(tails.values()):
        rb.free()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
