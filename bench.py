#!/usr/bin/env python3
"""QM9-shaped forward throughput of the SCANN+ HIP path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N > 1: this process spawns the N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (ranks made by the launcher)

A step = one forward of the whole graph (scann_model.py:329-453) over one batch of 128 synthetic QM9-shaped
molecules (configs/model_qm9.yaml: 7 local-attention layers, d=128, 8 heads, g_update) whose packed inputs are
already resident in HBM.  The packed layout has no per-batch padding, so the engine runs the K steps as
ceil(K/16) launch sequences over groups of at most 16 concatenated batches (equal sizes +-1): EXACTLY K batches
of 128 molecules per timed region.  The timed region is bracketed by device sync + rank barrier on both sides and is
REPEATED until >= 1 s has been measured (after an untimed pre-warm that does not depend on --warmup: the chip needs
a few hundred ms under load to reach its sustained clock); the reported time is the median over repeats of the
max-over-ranks region time.  value = molecules of all ranks / that time.  Inference shards by structure with no
data-path collective ("weak" scaling: per-GPU work fixed); ranks meet through a loopback TCP rendezvous (no torch).

Rank 0 prints ONE JSON line with `roofline` (dominant kernel = the fused edge kernel: both bounds are reported,
`bound` names the tighter one), `one_batch_per_launch`, the host-inclusive `end_to_end` rate and `cpu_baseline`
(the C/OpenMP oracle timed on the host cores -- a reported baseline, never the product path).
"""
import argparse
import importlib.util
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "scann--material_amd")

D = 128
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2516.6   # MI355X_MICROARCH.md: ~2.5 PF dense f16/bf16 = 16 x the f32 rate (1024 FLOP/clk/SIMD)
PEAK_HBM_TBS = 8.0              # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable)
QM9_MODEL = dict(n_atoms=10, embedding_dim=48, n_attention=7, local_dim=128, num_head=8, global_dim=128,
                 dense_out=128, scale=0.5, use_attn_norm=True, use_ga_norm=True, use_ring=False, g_update=True,
                 gaussian_d=4.0)  # configs/model_qm9.yaml:1-14
QM9_STD_MODEL = dict(QM9_MODEL, n_attention=8)  # configs/model_qm9_std.yaml:1-14 (BASELINE configs[4]: SCANN+ on QM9 Gap)
MP2018_MODEL = dict(n_atoms=95, embedding_dim=128, n_attention=9, local_dim=128, num_head=8, global_dim=128,
                    dense_out=128, scale=0.5, use_attn_norm=True, use_ga_norm=True, use_ring=False, g_update=True,
                    gaussian_d=6.0)  # configs/model_mp2018.yaml:1-14


def synth_packed_crystals(rng, n_struct):
    """MP2018-shaped structures: A ~ clip(LogNormal(3.0, 0.8), 2, 300), neighbours per atom U[6, 24],
    Z in [1, 94], distances U(1.5, 6.0)."""
    import numpy as np
    from scann import _hip

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
    import numpy as np
    from scann import _hip

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
    """Algorithmic FLOPs of one edge-kernel launch: per edge the geometry third of filter_geo (2 d^2), the key
    projection (2 d^2) and the q.k / attn.k contractions (4 d) -- SURVEY.md 8(d) minimal form, edge part."""
    return E * (4 * D * D + 4 * D)


def edge_bytes(A, E):
    """ALGORITHMIC HBM bytes of one edge-kernel launch = SURVEY.md 8(d)(ii), the layer-streamed model, per layer: geometry row
    in and out + index and mask per edge, centres in and out per atom (gathered neighbour rows are served on-chip and counted
    once per atom): E (2 d 4 + 8) + A (2 d 4).  This is what roofline.frac is computed from."""
    return E * (2 * D * 4 + 8) + A * (2 * D * 4)


def edge_bytes_design(A, E):
    """What THIS design's edge kernel has to move per launch: the same per-edge bytes, but FIVE atom rows
    (c, P1, P3, q in; context out) because the per-atom projections live in atom_kernel.  Reported beside the algorithmic
    figure, never as the roofline fraction."""
    return E * (2 * D * 4 + 8) + A * (5 * D * 4)


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
    import numpy as np

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


def group_sizes(steps, target=16):
    """K steps -> sizes of the launch groups: ceil(K / target) groups whose sizes differ by at most one, so no group exceeds
    `target` batches (a 16-batch group is what the on-die caches hold of one layer's working set: 141 MB of geometry + 95 MB
    of atom rows against the 256 MB Infinity Cache; 20 batches already fall off that edge) and a short run (the driver's
    --steps 20 -> 10 + 10) does not end in a small, chip-underfilling tail group."""
    if steps <= 0:
        return []
    n = (steps + target - 1) // target
    base, rem = divmod(steps, n)
    return [base + 1] * rem + [base] * (n - rem)


def scaling_fields(world, units_per_rank_region, rank_region_s, solo_region_s=None, rccl_ranks=None):
    """What a SCALE record can be checked against, from one N-rank run: `rccl_ranks` = the size RCCL itself reports for the
    communicator (ncclCommCount after scann_comm_init; None on the collective-free inference path), `rank_values` = every rank's OWN
    rate over the timed region (the headline divides the job by the slowest rank's time), `n1_same_layout` = rank 0 alone in the same
    process layout while the other ranks wait (what N = 1 gives inside this very run: must agree with the driver's N = 1 line)."""
    out = {"rccl_ranks": rccl_ranks, "ranks_reporting": len(rank_region_s),
           "rank_values": [units_per_rank_region / t if t > 0 else None for t in rank_region_s]}
    if world > 1:
        out["n1_same_layout"] = ({"value": units_per_rank_region / solo_region_s, "unit": "molecules/s",
                                  "what": "rank 0 alone (the other ranks idle at a barrier), same process layout and workload"}
                                 if solo_region_s else None)
    return out


def train_bench(args, eng, rdzv):
    """Weak-scaling training throughput: every rank trains on its own --batch molecules per step; the ranks exchange the
    scalar SSE/count and one flat fp32 gradient all-reduce per step over RCCL."""
    import numpy as np
    from scann.models.trainer import Communicator

    rank, world = rdzv.rank, rdzv.world
    eng.train_begin()
    rng = np.random.default_rng(2000 + rank)
    pool = [eng.upload(synth_packed_batch(rng, args.batch)) for _ in range(8)]
    targets = [rng.normal(size=args.batch).astype(np.float32) for _ in pool]
    solo = None
    if world > 1:  # N = 1 inside this run: rank 0 trains alone, BEFORE the communicator exists (afterwards every step is a collective)
        if rank == 0:
            for i in range(10):
                eng.train_step_begin(pool[i % 8], targets[i % 8], 5e-4, dropout=0.1, seed=i)
                eng.train_step_end()
            eng.sync()
            t0 = time.perf_counter()
            for i in range(100):
                eng.train_step_begin(pool[i % 8], targets[i % 8], 5e-4, dropout=0.1, seed=i)
                if i:
                    eng.train_step_end()
            eng.train_step_end()
            eng.sync()
            solo = (time.perf_counter() - t0) / 100
        rdzv.barrier()
    comm = Communicator(eng, rdzv)  # the rendezvous only carries the 128-byte ncclUniqueId; gradients go over RCCL (broadcasts rank 0's weights)
    rccl_ranks = eng.comm_ranks() if world > 1 else None

    def step(i):
        rb, t = pool[i % 8], targets[i % 8]
        if args.split_step:  # the step as six calls with host round trips between them (the round-1 / early round-2 sequence)
            sse = eng.train_forward(rb, t, dropout=0.1, seed=i)
            sse_g, cnt_g = comm.sum_pair(sse, args.batch)
            eng.zero_grads()
            eng.train_backward(rb, sse_g, cnt_g)
            eng.allreduce_grads()
            eng.adam_step(5e-4 / (1.0 + 1e-5 * i))
        else:  # two steps in flight: step i + 1 is enqueued before step i is waited for (what trainer.fit does)
            eng.train_step_begin(rb, t, 5e-4 / (1.0 + 1e-5 * i), dropout=0.1, seed=i)
            inflight[0] += 1
            if inflight[0] == 2:
                eng.train_step_end()
                inflight[0] -= 1

    inflight = [0]

    def drain():
        while inflight[0]:
            eng.train_step_end()
            inflight[0] -= 1
        eng.sync()

    steps, warm = min(args.steps, 400), max(min(args.warmup, 20), 5)
    for i in range(warm):
        step(i)
    drain()
    rdzv.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warm + i)
    drain()
    elapsed = time.perf_counter() - t0
    rdzv.barrier()
    per_rank = rdzv.gather([elapsed])  # every rank's own time for the K steps (rank 0 gets the list)
    elapsed = rdzv.allreduce_max(elapsed)
    if rank == 0:
        print(json.dumps({
            "metric": "QM9 molecules/s training (forward + backward + Adam)", "value": world * steps * args.batch / elapsed,
            "unit": "molecules/s", "n_gpus": world, "steps": steps, "warmup": warm, "ms_per_step": elapsed / steps * 1e3,
            "rank_ms_per_step": [float(t[0]) / steps * 1e3 for t in per_rank],
            **scaling_fields(world, steps * args.batch, [float(t[0]) for t in per_rank], solo * steps if solo else None, rccl_ranks),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (projections: split-fp16 hi/lo operands, 3 v_mfma_f32_32x32x16_f16 per product, fp32 accumulate; everything else fp32)",
            "data": "synthetic",
            "config": {"workload": "configs[2]: QM9 training, configs/model_qm9.yaml, %d molecules per GPU per step, dropout 0.1, "
                                   "RCCL flat gradient all-reduce (%d floats)" % (args.batch, eng.param_count()),
                       "global_batch": world * args.batch, "parallelism": "dp%d" % world}}), flush=True)
    for rb in pool:
        rb.free()
    rdzv.barrier()


def training_leg(cfg, batches, batch_size, seconds=1.0):
    """configs[2] beside the headline: one optimisation step (forward + backward + Adam, dropout 0.1) per resident batch, two steps
    in flight as trainer.fit runs them.  `bench.py --train` is the full (multi-rank) version of this leg."""
    import numpy as np
    from scann.models.scann_model import HipModel

    model = HipModel(cfg, device=int(os.environ.get("LOCAL_RANK", "0")), seed=1234)
    eng = model.engine
    eng.train_begin()
    rng = np.random.default_rng(7)
    pool = [eng.upload(b) for b in batches[:8]]
    targets = [rng.normal(size=batch_size).astype(np.float32) for _ in pool]
    inflight, done = 0, 0

    def run(n, i0):
        nonlocal inflight
        for i in range(i0, i0 + n):
            eng.train_step_begin(pool[i % 8], targets[i % 8], 5e-4 / (1.0 + 1e-5 * i), dropout=0.1, seed=i)
            inflight += 1
            if inflight == 2:
                eng.train_step_end()
                inflight -= 1
        while inflight:
            eng.train_step_end()
            inflight -= 1

    run(20, 0)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        run(100, 20 + done)
        done += 100
    dt = time.perf_counter() - t0
    for rb in pool:
        rb.free()
    eng.close()
    return {"value": done * batch_size / dt, "unit": "molecules/s", "ms_per_step": dt / done * 1e3, "steps": done, "batch": batch_size,
            "what": "forward + backward + Adam on resident batches (configs[2], 1 rank), two steps in flight"}


def end_to_end(cfg, batches, batch_size, seconds=1.5):
    """Host-inclusive rate of the dataset path behind SCANN.evaluate / predict_model.py: a host PackedDataset (flat CSR in
    host memory) -> native slicing -> H2D -> forward -> D2H, pipelined over the handle's default 2 streams and two host threads
    (HipModel.predict_dataset)."""
    import numpy as np
    from scann.models.scann_model import HipModel
    from scann.utils import PackedDataset

    mol, eoff, atomic, local, dist, wgt = [0], [0], [], [], [], []
    batches = list(batches) * max(1, 65536 // max(1, len(batches) * batch_size))  # >= 65,536 molecules: the pipeline's steady state, not its ramp
    for b in batches:
        base = np.repeat(b.mol_offset[:-1], np.diff(b.mol_offset))            # first atom row of every atom's structure
        deg = np.diff(b.edge_offset)
        local.append(b.edge_col - np.repeat(base, deg))
        mol.extend((b.mol_offset[1:].astype(np.int64) + mol[-1]).tolist())
        eoff.extend((b.edge_offset[1:].astype(np.int64) + eoff[-1]).tolist())
        atomic.append(b.atomic); dist.append(b.edge_dist); wgt.append(b.edge_weight)
    n = len(mol) - 1
    ds = PackedDataset.from_arrays(mol, np.concatenate(atomic), eoff, np.concatenate(local), np.concatenate(dist),
                                   np.concatenate(wgt), np.zeros(n, np.float32), batch_size=batch_size)
    os.environ.pop("SCANN_STREAMS", None)  # the library's default
    model = HipModel(cfg, device=int(os.environ.get("LOCAL_RANK", "0")), seed=1234)
    model.predict_dataset(ds, group=8)  # warm (allocator cache, clocks)
    reps, t0 = 0, time.perf_counter()
    while reps == 0 or time.perf_counter() - t0 < seconds:
        model.predict_dataset(ds, group=8)
        reps += 1
    dt = time.perf_counter() - t0
    ns = model.engine.num_streams()
    model.engine.close()
    return {"value": reps * n / dt, "unit": "molecules/s", "molecules": n, "passes": reps, "streams": ns, "group": 8,
            "path": "host PackedDataset -> scann_slice_batch -> upload -> forward -> download (HipModel.predict_dataset), PCIe-inclusive"}


def two_stream_leg(cfg, batches, batch_size, seconds=0.8, group=8, streams=2):
    """The resident-input forward with the launch groups of TWO half-size groups in flight on two streams (what
    HipModel.predict_dataset does with its four): the first layer's launch (basis MLP: VALU) and the atom launches of one group fill the
    units the edge launches of the other leave idle.  Kernel durations overlap in this mode, so the headline (and its roofline, which
    needs the duration of a launch that has the device to itself) stays on one stream; this is the same engine's throughput."""
    from scann import _hip
    from scann.models.scann_model import HipModel

    os.environ["SCANN_STREAMS"] = str(streams)
    model = HipModel(cfg, device=int(os.environ.get("LOCAL_RANK", "0")), seed=1234)
    eng = model.engine
    n_g = max(2, min(len(batches) // group, 8))
    groups = [eng.upload(_hip.concat_packed([batches[(i * group + j) % len(batches)] for j in range(group)])) for i in range(n_g)]
    for i in range(4 * n_g):
        eng.forward_resident(groups[i % n_g], i % streams)
    eng.sync()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for i in range(8 * n_g):
            eng.forward_resident(groups[i % n_g], i % streams)
        eng.sync()
        n += 8 * n_g
    dt = time.perf_counter() - t0
    for rb in groups:
        rb.free()
    eng.close()
    return {"value": n * group * batch_size / dt, "unit": "molecules/s", "streams": streams, "batches_fused_per_launch": group,
            "launch_sequences": n}


def one_batch_leg(cfg, batches, batch_size, seconds=0.5):
    """The same engine with exactly ONE 128-molecule batch per launch sequence (no fusing of batches), one stream."""
    from scann.models.scann_model import HipModel

    os.environ["SCANN_STREAMS"] = "1"
    eng = HipModel(cfg, device=int(os.environ.get("LOCAL_RANK", "0")), seed=1234).engine
    singles = [eng.upload(b) for b in batches[:32]]
    for i in range(64):
        eng.forward_resident(singles[i % len(singles)], 0)
    eng.sync()
    n1, t1 = 0, time.perf_counter()
    while time.perf_counter() - t1 < seconds:
        for i in range(64):
            eng.forward_resident(singles[i % len(singles)], 0)
        eng.sync()
        n1 += 64
    dt = time.perf_counter() - t1
    for rb in singles:
        rb.free()
    eng.close()
    return {"value": n1 * batch_size / dt, "unit": "molecules/s", "steps": n1, "streams": 1}


def padded_predict_leg(cfg, batches, batch_size, seconds=1.0):
    """The reference's literal call on a whole dataset: `model.predict(x)` with x the padded Keras input dict of ALL the leg's molecules
    (128 batches: 16,384 structures, [B, M, N] neighbour slots) -- host packing (padded -> CSR), uploads, forwards, downloads and Python
    included; HipModel.predict software-pipelines chunks of it."""
    import numpy as np
    from scann import _hip
    from scann.models.scann_model import HipModel

    pk = _hip.concat_packed(batches)
    mol, eoff = np.asarray(pk.mol_offset, np.int64), np.asarray(pk.edge_offset, np.int64)
    B, A, E = pk.n_struct, pk.n_atom, pk.n_edge
    s_of_a = np.repeat(np.arange(B), np.diff(mol))
    a_loc = np.arange(A) - mol[s_of_a]
    a_of_e = np.repeat(np.arange(A), np.diff(eoff))
    n_loc = np.arange(E) - eoff[a_of_e]
    M, N = int(np.diff(mol).max()), int(max(1, np.diff(eoff).max()))
    atomic = np.zeros((B, M), np.int32)
    atomic[s_of_a, a_loc] = pk.atomic
    amask = np.zeros((B, M, 1), np.float32)
    amask[s_of_a, a_loc, 0] = 1.0
    nbr, nmask = np.zeros((B, M, N), np.int32), np.zeros((B, M, N), np.float32)
    dist, wgt = np.zeros((B, M, N), np.float32), np.zeros((B, M, N), np.float32)
    idx = (s_of_a[a_of_e], a_loc[a_of_e], n_loc)
    nbr[idx] = np.asarray(pk.edge_col, np.int64) - mol[s_of_a[a_of_e]]
    nmask[idx] = 1.0
    dist[idx], wgt[idx] = pk.edge_dist, pk.edge_weight
    inputs = {"atomic": atomic, "atom_mask": amask, "neighbors": nbr, "neighbor_mask": nmask, "neighbor_weight": wgt, "neighbor_distance": dist}
    model = HipModel(cfg, device=int(os.environ.get("LOCAL_RANK", "0")), seed=1234)
    y0 = model.predict(inputs)
    t_all, t0 = [], time.perf_counter()
    while time.perf_counter() - t0 < seconds or len(t_all) < 3:
        t1 = time.perf_counter()
        y = model.predict(inputs)
        t_all.append(time.perf_counter() - t1)
    assert y.shape == (B, 1) and np.array_equal(y, y0)
    dt = float(np.median(t_all))
    model.engine.close()
    return {"value": B / dt, "unit": "molecules/s", "molecules": B, "ms_per_call": 1e3 * dt, "calls": len(t_all), "padded_shape": [B, M, N],
            "what": "model.predict(padded input dict of the whole set): padded -> CSR packing, uploads, forwards, downloads, Python inclusive"}


def exact_fp32_leg(cfg, batches, batch_size, seconds=0.8, group=8):
    """The same forward on the bitwise-fp32 kernels (SCANN_EXACT=1: every 128x128 product on v_mfma_f32_32x32x2_f32, the arithmetic of
    the reference's fp32 Dense layers, attention.py:95-113; no hi / lo split anywhere): what this path does at the reference's own
    arithmetic, beside the split-fp16 headline.  Resident 8-batch groups, one stream."""
    from scann import _hip
    from scann.models.scann_model import HipModel

    os.environ["SCANN_STREAMS"] = "1"
    os.environ["SCANN_EXACT"] = "1"
    eng = HipModel(cfg, device=int(os.environ.get("LOCAL_RANK", "0")), seed=1234).engine
    groups = [eng.upload(_hip.concat_packed(batches[i * group:(i + 1) * group])) for i in range(4)]
    for i in range(8):
        eng.forward_resident(groups[i % 4], 0)
    eng.sync()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for i in range(4):
            eng.forward_resident(groups[i], 0)
        eng.sync()
        n += 4
    dt = time.perf_counter() - t0
    for rb in groups:
        rb.free()
    eng.close()
    return {"value": n * group * batch_size / dt, "unit": "molecules/s", "batches_fused_per_launch": group, "streams": 1,
            "dtype": "f32 (every projection on v_mfma_f32_32x32x2_f32: bitwise fp32 products and accumulation)",
            "what": "SCANN_EXACT=1: the forward on the exact-fp32 instantiations of the atom / edge kernels (the fallback a forward takes when "
                    "an operand leaves the split-fp16 range)"}


def training_b16_leg(cfg, batches, batch_size, seconds=1.0):
    """configs[2] as the reference would run it on 8 GPUs: GLOBAL batch 128 (model_qm9.yaml:16; the reference's optimisation trajectory,
    SURVEY.md 7 'Training DP semantics') = 16 molecules per rank and step.  This is ONE rank's step at that shape, without the two
    collectives (a scalar and 3.56 MB over xGMI): the device-side floor of a rank's step, the number DESIGN.md 6's <= 1.35 x prediction
    for the 8-way split rests on."""
    import numpy as np

    rng = np.random.default_rng(1016)
    small = [synth_packed_batch(rng, 16) for _ in range(8)]
    out = training_leg(cfg, small, 16, seconds)
    out["what"] = ("forward + backward + Adam on resident 16-molecule batches: one rank's share of configs[2] at the reference's global batch "
                   "of 128 over 8 GPUs (no collective in this leg), two steps in flight")
    return out


def roofline_shapes(eng, batches, batch_size, A, E, groups=(10, 12, 14, 16), forwards=96):
    """The dominant kernel's HBM-roofline fraction at launch shapes other than the timed one, measured in this run with the same HIP
    events (scann_edge_timing on every third forward, like the timed region's sparse sample; one stream): 10 / 12 / 14 / 16 batches per
    launch, and the least-squares line through them -- microseconds per 64-row edge tile in steady state (the slope) and per launch (the
    intercept: ramp, drain and the last, partly filled round of workgroups).  SURVEY.md 8(d)(ii) bytes throughout."""
    import numpy as np
    from scann import _hip

    pts = []
    for g in groups:
        n_g = max(1, min(len(batches) // g, 4))
        res = [eng.upload(_hip.concat_packed([batches[(i * g + j) % len(batches)] for j in range(g)])) for i in range(n_g)]
        for i in range(32):
            eng.forward_resident(res[i % n_g], 0)
        eng.sync()
        eng.edge_timing(3)
        for i in range(forwards):
            eng.forward_resident(res[i % n_g], 0)
        eng.sync()
        us, n, edges = eng.edge_timing_read()
        eng.edge_timing(0)
        for rb in res:
            rb.free()
        if n:
            a_l = edges * A / E
            pts.append({"batches_per_launch": g, "avg_launch_us": us, "launches_sampled": n, "edges_per_launch": edges,
                        "frac": edge_bytes(a_l, edges) / (us * 1e-6) / 1e12 / PEAK_HBM_TBS})
    out = {"points": pts}
    if len(pts) >= 2:
        x = np.array([p["edges_per_launch"] / 64.0 for p in pts])
        y = np.array([p["avg_launch_us"] for p in pts])
        slope, icpt = np.polyfit(x, y, 1)
        tile_bytes = 64 * (2 * D * 4 + 8) + 64 * (A / E) * (2 * D * 4)
        out["per_tile"] = {"ns_per_64_edge_tile": slope * 1e3, "intercept_us": icpt, "bytes_per_tile": tile_bytes,
                           "frac_steady_state": tile_bytes / (slope * 1e-6) / 1e12 / PEAK_HBM_TBS if slope > 0 else None,
                           "what": "least-squares line through the points: slope = a tile's cost with the chip full, intercept = per-launch ramp / drain / partial last round"}
    return out


LEGS = {"two_streams": two_stream_leg, "training_step_b16": training_b16_leg, "exact_fp32": exact_fp32_leg, "one_batch_per_launch": one_batch_leg, "end_to_end": end_to_end, "training_step": training_leg,
        "padded_predict": padded_predict_leg}


def run_leg_here(name, batch_size, config_name):
    """`bench.py --leg NAME`: one extra leg in a process of its own, its dict on stdout."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "scann--material_amd"))
    from scann.models.scann_model import normalize_config

    model_cfg = dict(QM9_MODEL)
    cfg = normalize_config({"model": model_cfg, "hyper": {"target": "homo"}})
    rng = np.random.default_rng(1000)
    batches = [synth_packed_batch(rng, batch_size) for _ in range(128)]
    print(json.dumps(LEGS[name](cfg, batches, batch_size)), flush=True)


def leg_in_subprocess(name, batch_size):
    """The extra legs run in fresh processes: HIP deals a process's streams onto a few hardware queues in an order that depends on
    every stream the process has ever made, and a leg run after the headline's engine read 4-14 % low (two of its "concurrent"
    streams on one queue) -- an artefact of sharing the process, not of the leg."""
    import subprocess

    env = dict(os.environ)
    env.pop("SCANN_STREAMS", None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg", name, "--batch", str(batch_size)], env=env, capture_output=True,
                       text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": "leg %s failed (rc %d): %s" % (name, r.returncode, (r.stderr or "")[-300:])}
    return json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--pool", type=int, default=128, help="distinct resident batches the groups are built from")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("SCANN_STREAMS", "1")),
                    help="HIP streams the launch sequences are spread over.  Default 1: launches do not overlap, so the "
                         "HIP-event launch durations taken inside the timed region are the kernel's own (what rocprofv3 "
                         "--stats reports)")
    ap.add_argument("--group", type=int, default=int(os.environ.get("SCANN_BENCH_GROUP", "16")),
                    help="target number of resident 128-molecule batches per launch sequence (see group_sizes; 1 = one batch per launch)")
    ap.add_argument("--worst", action="store_true", help="Swc: every molecule 29 atoms x 12 neighbours")
    ap.add_argument("--config", default="qm9", choices=["qm9", "qm9_std", "mp2018"],
                    help="qm9 = BASELINE configs[1] (the metric); qm9_std = configs[4] (SCANN+ L=8, QM9 shapes), mp2018 = configs[3] "
                         "shapes (crystals, L=9, batch 64): extras")
    ap.add_argument("--min-time", type=float, default=1.0, help="seconds of timed regions to collect (median over repeats)")
    ap.add_argument("--prewarm", type=float, default=0.6, help="seconds of untimed load before the first timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip one_batch_per_launch / end_to_end / cpu_baseline (rocprofv3 runs: only the timed launch shape then "
                         "contributes to the per-kernel averages)")
    ap.add_argument("--train", action="store_true",
                    help="extra (BASELINE configs[2]): time data-parallel TRAINING steps instead of the forward metric -- "
                         "forward(train, dropout 0.1) + SSE all-reduce + backward + flat RCCL gradient all-reduce + Adam; "
                         "--batch molecules per GPU per step")
    ap.add_argument("--split-step", action="store_true", help="--train: forward / backward / Adam as separate calls instead of scann_train_step")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="rehearsal only: allow more ranks than visible devices (rank r runs on device r %% n_devices); the line "
                         "is then marked \"oversubscribed\" and is not a scaling measurement")
    ap.add_argument("--profile-reps", type=int, default=12)
    ap.add_argument("--leg", default=None, choices=sorted(LEGS), help="internal: run one extra leg in this process and print its dict")
    args = ap.parse_args()
    if args.leg:
        return run_leg_here(args.leg, args.batch, args.config)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: THIS process becomes the launcher.  It spawns one fresh process per GPU before anything here has
        # touched HIP (the package, ctypes binding and library are not even imported) and exits with their code.
        spec = importlib.util.spec_from_file_location("_scann_launch", os.path.join(PKG, "scann", "parallel", "launch.py"))
        launch = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(launch)
        sys.exit(launch.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus))

    sys.path.insert(0, PKG)
    import numpy as np
    from scann import _hip  # loads libscann_hip.so on first use; no torch / no oracle on the product path
    from scann.models.scann_model import HipModel, normalize_config
    from scann.parallel.rendezvous import Rendezvous

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    rdzv = Rendezvous(rank, world)

    os.environ["SCANN_STREAMS"] = str(max(1, args.streams))  # read by scann_create
    if args.config == "mp2018" and args.batch == 128:
        args.batch = 64  # configs/model_mp2018.yaml:16
    model_cfg = dict({"qm9": QM9_MODEL, "qm9_std": QM9_STD_MODEL, "mp2018": MP2018_MODEL}[args.config])
    cfg = normalize_config({"model": model_cfg, "hyper": {"target": "homo", "batch_size": args.batch}})
    ndev = _hip.load_library().scann_device_count()
    if ndev <= 0:
        raise SystemExit("bench.py needs a GPU: libscann_hip has no CPU fallback")
    if world > ndev and not args.oversubscribe:
        raise SystemExit("--gpus %d but only %d device(s) visible" % (world, ndev))
    model = HipModel(cfg, device=local % ndev, seed=1234)  # random-init weights of the QM9 architecture
    eng = model.engine
    if args.train:
        return train_bench(args, eng, rdzv)
    nstream = eng.num_streams()
    rng = np.random.default_rng(1000 + rank)
    n_pool = max(args.pool, 32)
    if args.config == "mp2018":
        batches = [synth_packed_crystals(rng, args.batch) for _ in range(n_pool)]
    else:
        batches = [synth_packed_batch(rng, args.batch, args.worst) for _ in range(n_pool)]

    # resident groups: for every group size the schedules need, as many distinct groups as the pool yields (>= 1)
    sizes_t, sizes_w = group_sizes(args.steps, args.group), group_sizes(args.warmup, args.group)
    resident = {}
    for g in sorted(set(sizes_t + sizes_w)):
        n_g = max(1, min(n_pool // g, 8))
        resident[g] = [eng.upload(_hip.concat_packed([batches[(i * g + j) % n_pool] for j in range(g)]) if g > 1 else batches[i % n_pool])
                       for i in range(n_g)]

    def run(sizes):
        for i, g in enumerate(sizes):
            eng.forward_resident(resident[g][i % len(resident[g])], i % nstream)

    run(sizes_w)
    eng.sync()
    # untimed pre-warm, independent of --warmup: clocks and caches reach their loaded state (a cold process runs the first
    # few hundred launches at ~2.15 GHz instead of ~2.4)
    t_pw = time.perf_counter()
    while time.perf_counter() - t_pw < args.prewarm:
        run(sizes_t)
        eng.sync()
    t0 = time.perf_counter()
    run(sizes_t)
    eng.sync()
    est = rdzv.allreduce_max(time.perf_counter() - t0)
    repeats = int(min(4000, max(3, math.ceil(args.min_time / max(est, 1e-6))))) | 1  # odd: the median is a measured region
    if rank == 0:  # HIP events around the edge-kernel launches of sampled forwards, on their own streams (<= ~64 forwards)
        eng.edge_timing(max(1, repeats * len(sizes_t) // 64))
    times, t_issue = [], 0.0
    for _ in range(repeats):
        eng.sync()
        rdzv.barrier()
        t0 = time.perf_counter()
        run(sizes_t)
        t1 = time.perf_counter()
        eng.sync()
        times.append(time.perf_counter() - t0)
        t_issue += t1 - t0  # host time to enqueue every launch (diagnostic: host- vs device-bound)
        rdzv.barrier()
    all_times = rdzv.gather(times)
    solo = None
    if world > 1:  # N = 1 inside this run: rank 0 repeats the timed region alone while the other ranks wait at the barrier
        if rank == 0:
            ts = []
            for _ in range(min(repeats, 101)):
                eng.sync()
                t0 = time.perf_counter()
                run(sizes_t)
                eng.sync()
                ts.append(time.perf_counter() - t0)
            solo = float(np.median(ts))
        rdzv.barrier()

    if rank == 0:
        per_repeat = np.max(np.asarray(all_times, dtype=np.float64), axis=0)  # max over ranks of every timed region
        elapsed = float(np.median(per_repeat))
        live_us, live_n, live_edges = eng.edge_timing_read()
        eng.edge_timing(0)
        # sequential per-kernel durations of one forward (scann_forward_profile), largest timed group
        gmax = max(sizes_t)
        prof = []
        for i in range(args.profile_reps):
            p = eng.profile(resident[gmax][i % len(resident[gmax])])
            if i >= 2:
                prof.append(p)
        A = float(np.mean([b.n_atom for b in batches]))
        E = float(np.mean([b.n_edge for b in batches]))
        G_eff = args.steps / len(sizes_t)
        if live_n:
            e_launch, us = live_edges, live_us
        else:
            e_launch, us = E * gmax, 1e3 * float(np.mean([p["ms_edge"] / max(p["n_edge_launch"], 1) for p in prof]))
        a_launch = e_launch * A / E
        alg_flops, alg_bytes = edge_flops(e_launch), edge_bytes(a_launch, e_launch)
        des_bytes = edge_bytes_design(a_launch, e_launch)
        tfl = alg_flops / (us * 1e-6) / 1e12
        tbs = alg_bytes / (us * 1e-6) / 1e12
        kinfo = {}
        kfile = os.path.join(ROOT, "profiles", "edge_kernel.json")  # written with the kernel: pipe, passes, traffic per workload
        if os.path.exists(kfile):
            kinfo = json.load(open(kfile))
        passes = int(kinfo.get("mfma_passes", 1))      # MFMA products issued per algorithmic product (split operands)
        pipe_peak = PEAK_F16_MFMA_TFLOPS if kinfo.get("mfma_pipe", "f32") == "f16" else PEAK_FP32_MFMA_TFLOPS
        frac_mfma = tfl * passes / pipe_peak
        frac_hbm = tbs / PEAK_HBM_TBS
        wkey = "%s%s_g%d" % (args.config, "_worst" if args.worst else "", gmax)
        traffic = (kinfo.get("traffic") or {}).get(wkey)  # PMC bytes per launch, measured at THIS workload and launch size or null
        hbm_bound = frac_hbm >= frac_mfma
        roof = {"bound": "hbm" if hbm_bound else "mfma",
                "kernel": "%s (scann_kernels.hip), %d batches per launch" % (kinfo.get("name", "edge_kernel"), gmax),
                "achieved": tbs * 1e3 if hbm_bound else tfl * passes, "peak": PEAK_HBM_TBS * 1e3 if hbm_bound else pipe_peak,
                "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": frac_hbm if hbm_bound else frac_mfma,
                "traffic": traffic["hbm_bytes_per_launch"] if traffic else None,
                "traffic_over_algorithmic": traffic["hbm_bytes_per_launch"] / alg_bytes if traffic else None,
                "traffic_source": ("profiles/edge_kernel.json[%s]: STATIC rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload "
                                   "and launch size (tools/pmc_run.sh), not measured in this run; fabric-side counters that include "
                                   "Infinity-Cache hits (a 16-batch layer's working set fits the 256 MB cache)" % wkey) if traffic else None,
                "avg_launch_us": us, "launches_sampled": live_n,
                "algorithmic": {"model": "SURVEY.md 8(d)(ii): E (2 d 4 + 8) + A (2 d 4) bytes, E (4 d^2 + 4 d) FLOPs per launch",
                                "flops_per_launch": alg_flops, "bytes_per_launch": alg_bytes, "edges_per_launch": e_launch,
                                "tflops": tfl, "tbytes_per_s": tbs},
                "design_bytes": {"model": "this design's edge kernel: E (2 d 4 + 8) + A (5 d 4) -- c, P1, P3, q in and the context out per atom",
                                 "bytes_per_launch": des_bytes, "frac_of_hbm_peak": des_bytes / (us * 1e-6) / 1e12 / PEAK_HBM_TBS},
                "mfma": {"pipe": kinfo.get("mfma_pipe", "f32"), "passes": passes, "executed_tflops": tfl * passes, "peak": pipe_peak,
                         "frac": frac_mfma, "frac_of_fp32_mfma_peak_algorithmic": tfl / PEAK_FP32_MFMA_TFLOPS},
                "hbm": {"achieved_gbs": tbs * 1e3, "peak_gbs": PEAK_HBM_TBS * 1e3, "frac": frac_hbm},
                "note": "avg_launch_us = HIP events around sampled edge_kernel<true, 2, false, false, false, false> launches (template arguments "
                        "GUPD, RT, FB, EX, KEEP, DEAD) inside the timed regions, on the launch stream (1 stream: launches never overlap): the "
                        "layers between the first and the last.  Not sampled: the first layer's launch (a different kernel: basis MLP and "
                        "per-species atom rows fused in, FB = true) and the last layer's (DEAD = true: geom' is not stored, ~10 % shorter) -- "
                        "profiles/r05_kernel_stats_g10.csv lists all three; achieved = ALGORITHMIC bytes or FLOPs of a launch / that time",
                "per_forward_ms": {k: float(np.mean([p[k] for p in prof])) for k in
                                   ("ms_basis", "ms_atom", "ms_edge", "ms_readout", "ms_total")} if prof else None}
        if not args.no_extras and world == 1 and args.config == "qm9" and not args.worst:
            # the same kernel at 10 / 12 / 14 / 16 batches per launch and its per-tile slope, beside the timed shape's figure (read "0.37 here,
            # 0.39 there, 0.41 in steady state" off ONE record); one stream, kernel sampling as above
            roof["shapes"] = {"timed": {"batches_per_launch": gmax, "avg_launch_us": us, "frac": frac_hbm}, **roofline_shapes(eng, batches, args.batch, A, E)}
        value = world * args.steps * args.batch / elapsed
        L_cfg, emb_cfg = model_cfg["n_attention"], model_cfg["embedding_dim"]
        out = {
            "metric": "QM9 molecules/s forward" if args.config != "mp2018" else "MP2018-shaped structures/s forward",
            "value": value, "unit": "molecules/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (projections: split-fp16 hi/lo operands, 3 v_mfma_f32_32x32x16_f16 per product, fp32 accumulate; everything else fp32)",
            "data": "synthetic",
            "config": {"workload": ("configs[1]: QM9-shaped%s, configs/model_qm9.yaml (SCANN+, L=7, d=128, H=8), "
                                    "batch=128 per step, forward" % (" worst-case 29x12" if args.worst else ""))
                       if args.config == "qm9" else
                       "configs[4]: QM9-shaped, configs/model_qm9_std.yaml (SCANN+, L=8, d=128, H=8), batch=128 per step, forward"
                       if args.config == "qm9_std" else
                       "configs[3] shapes: MP2018-shaped crystals, configs/model_mp2018.yaml (SCANN+, L=9), batch=64 per step, forward",
                       "batch": args.batch, "atoms_per_batch": A, "edges_per_batch": E,
                       "streams": nstream, "batches_fused_per_launch": G_eff, "launch_groups": sorted(set(sizes_t)),
                       "parallelism": "dp%d (independent shards, no collective)" % world},
            "timing": {"repeats": repeats, "timed_region_ms_median": elapsed * 1e3, "timed_region_ms_min": float(per_repeat.min()) * 1e3,
                       "timed_region_ms_max": float(per_repeat.max()) * 1e3, "prewarm_s": args.prewarm,
                       "rank_median_ms": [float(np.median(t)) * 1e3 for t in all_times],
                       "rule": "each repeat = EXACTLY --steps batches between sync+barrier pairs; value from the median repeat (max over ranks)"},
            "host_issue_ms_per_step": t_issue / repeats / args.steps * 1e3,
            "whole_path_tflops_min": world * args.steps * total_flops_min(A, E, L_cfg, emb_cfg) / elapsed / 1e12,
            "roofline": roof,
            **scaling_fields(world, args.steps * args.batch, [float(np.median(t)) for t in all_times], solo, None),
        }
        if world > ndev:
            out["oversubscribed"] = True
    for g in list(resident):
        for rb in resident.pop(g):
            rb.free()
    if rank == 0:
        if not args.no_extras and world == 1 and args.config == "qm9" and not args.worst:
            # each in a process of its own (leg_in_subprocess); this process's engine goes first, so that the legs have the device
            eng.close()
            for name in ("two_streams", "one_batch_per_launch", "end_to_end", "padded_predict", "training_step", "training_step_b16", "exact_fp32"):
                out[name] = leg_in_subprocess(name, args.batch)
        if not args.no_cpu_baseline and not args.no_extras and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    rdzv.barrier()
    rdzv.close()


if __name__ == "__main__":
    main()
