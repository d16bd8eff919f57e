"""Threaded host code under ThreadSanitizer (CPU build only -- the GPU pool runs no sanitizers).

`make -C scann--material_amd/csrc tsan` rebuilds the host-only C++ (scann_pack.cpp: the packers' threaded passes, the threaded staging copy
scann_host_copy, the slicer; scann_listwalk.cpp) with -fsanitize=thread; a child interpreter preloads the TSan runtime and drives
  * scann_pack_padded / scann_count_padded on 2 .. 7 worker threads (SCANN_PACK_THREADS) against their one-thread results,
  * scann_host_copy on 2 .. 8 threads (SCANN_COPY_THREADS), also from two Python threads at once on disjoint buffers (the documented
    use: scann_batch_upload may be called from a second thread while the first enqueues launches),
  * the two-thread dataset pipeline (HipModel.predict_dataset with SCANN_DATASET_THREAD=1: the producer thread slices groups with the
    native slicer and "uploads" them -- a stand-in engine whose upload is the staging copy -- while the consumer fetches results),
  * the training loader (trainer._Prefetch: batch k + 1 sliced on a worker thread while the consumer holds batch k).
No suppressions: any report makes the child exit non-zero (halt_on_error, exitcode 66)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TSAN_DIR = os.path.join(ROOT, "scann--material_amd", "lib", "tsan")

_CHILD = r'''
import ctypes as C, importlib.util, os, sys, threading
import numpy as np
ROOT, TSAN_DIR = sys.argv[1], sys.argv[2]
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), os.path.join(ROOT, "oracle")]
from scann import _hip

lib = C.CDLL(os.path.join(TSAN_DIR, "libscann_pack_tsan.so"))
for name, res, args in _hip.SYMBOLS:
    if name.startswith(("scann_pack", "scann_slice", "scann_plan", "scann_count", "scann_host_copy")):
        fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
_hip._lib = lib
spec = importlib.util.spec_from_file_location("scann._listwalk", os.path.join(TSAN_DIR, "_listwalk.so"))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
import scann
sys.modules["scann._listwalk"] = mod; scann._listwalk = mod

import scann_oracle as so
from scann.models.scann_model import HipModel
from scann.models import trainer
from scann.utils import PackedDataset

# ---- 1. the packers' threaded passes: same arrays on 1, 2, 3, 7 threads --------------------------------------------------------
de, dn = so.synth_dataset(300, 5)
inputs, _ = so.pad_batch(de, dn, True)
rep = 8  # 2,400 structures: above the packers' own 2,048-structure threshold
big = {k: np.concatenate([v] * rep) for k, v in inputs.items()}
ref = None
for n_thr in ("1", "2", "3", "7"):
    os.environ["SCANN_PACK_THREADS"] = n_thr
    for cast in (np.bool_, np.float32):
        x = dict(big, atom_mask=np.asarray(big["atom_mask"]).astype(cast), neighbor_mask=np.asarray(big["neighbor_mask"]).astype(cast))
        pk = _hip.pack_inputs(x)
        mol, eoff, row_of = _hip.count_padded(x)
        got = (pk.atomic, pk.mol_offset, pk.edge_offset, pk.edge_col, pk.edge_dist, pk.edge_weight, mol, eoff, row_of)
        if ref is None:
            ref = got
        assert all(np.array_equal(a, b) for a, b in zip(got, ref)), n_thr
os.environ.pop("SCANN_PACK_THREADS")

# ---- 2. the staging copy: threads inside one call, and two calls at once from two Python threads ---------------------------------
rng = np.random.default_rng(1)
src = rng.integers(0, 255, size=(3 << 20) + 77, dtype=np.uint8)
for n_thr in ("2", "5", "8"):
    os.environ["SCANN_COPY_THREADS"] = n_thr
    dst = np.zeros_like(src)
    assert lib.scann_host_copy(dst.ctypes.data, src.ctypes.data, src.nbytes) == 0 and np.array_equal(dst, src)
d1, d2 = np.zeros_like(src), np.zeros_like(src)
ts = [threading.Thread(target=lambda d=d: lib.scann_host_copy(d.ctypes.data, src.ctypes.data, src.nbytes)) for d in (d1, d2)]
[t.start() for t in ts]; [t.join() for t in ts]
assert np.array_equal(d1, src) and np.array_equal(d2, src)
os.environ["SCANN_COPY_THREADS"] = "3"

# ---- 3. the two-thread dataset pipeline on a stand-in engine ---------------------------------------------------------------------
class _RB:
    def __init__(self, pk, staged):
        self.packed, self.staged, self.y = pk, staged, None
    def free(self): pass
    release = free

class StandIn:  # upload = the staging copy of the batch's edge arrays (native, threaded); forward / download = sums over the staged copy
    training = False
    def num_streams(self): return 2
    def upload(self, pk):
        st = np.empty(pk.n_edge, np.float32)
        assert lib.scann_host_copy(st.ctypes.data, pk.edge_dist.ctypes.data, st.nbytes) == 0
        return _RB(pk, st)
    def forward_resident(self, rb, slot=0):
        rb.y = np.add.reduceat(rb.staged, rb.packed.edge_offset[rb.packed.mol_offset[:-1]].astype(np.int64)).astype(np.float32) if rb.packed.n_edge else None
    def download(self, rb, want_ga=True):
        return rb.y, (np.zeros(rb.packed.n_atom, np.float32) if want_ga else None)

ds = PackedDataset(data_energy=de, data_neighbor=dn, batch_size=16, use_ring=False, feature="atomic", g_update=True, atomic_features=None, shuffle=False)
model = HipModel.__new__(HipModel)
model.engine, model.infer = StandIn(), False
outs = {}
for flag in ("0", "1"):
    os.environ["SCANN_DATASET_THREAD"] = flag
    y, _, t = model.predict_dataset(ds, group=3)
    outs[flag] = (y, t)
assert np.array_equal(outs["0"][0], outs["1"][0]) and np.array_equal(outs["0"][1], outs["1"][1]) and len(outs["1"][0]) == 300
os.environ.pop("SCANN_DATASET_THREAD")

# ---- 4. the training loader thread (batch k + 1 sliced while batch k is consumed), one rank and rank 1 of 2 ----------------------
class Comm:
    def __init__(self, rank, world): self.rank, self.world = rank, world
    def shard(self, packed):
        from scann.parallel import rank_slice, slice_packed
        lo, hi = rank_slice(packed.n_struct, self.rank, self.world)
        return slice_packed(packed, lo, hi), slice(lo, hi)
for rank, world in ((0, 1), (1, 2)):
    n = 0
    for shard, target in trainer._Prefetch(ds, Comm(rank, world)):
        assert shard.n_struct == len(target) and shard.edge_offset[-1] == shard.n_edge
        n += shard.n_struct
    assert n == (300 if world == 1 else 300 // 2), n
# a consumer that gives up mid-epoch: the worker is stopped and joined
pf = trainer._Prefetch(ds, Comm(0, 1))
it = iter(pf); next(it); it.close()
assert not pf.t.is_alive()
print("TSAN_OK")
'''


# negative control: two Python threads copy DIFFERENT sources into the SAME destination -- a real race on the destination's bytes
_RACY = r'''
import ctypes as C, os, sys, threading
import numpy as np
lib = C.CDLL(os.path.join(sys.argv[2], "libscann_pack_tsan.so"))
lib.scann_host_copy.restype = C.c_int; lib.scann_host_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
a = np.arange(1 << 20, dtype=np.uint8); b = a[::-1].copy(); d = np.zeros_like(a)
ts = [threading.Thread(target=lambda s=s: [lib.scann_host_copy(d.ctypes.data, s.ctypes.data, s.nbytes) for _ in range(20)]) for s in (a, b)]
[t.start() for t in ts]; [t.join() for t in ts]
print("RACY_DONE")
'''


def _tsan_env():
    csrc = os.path.join(ROOT, "scann--material_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "tsan"], capture_output=True, text=True)
    libs = [os.path.join(TSAN_DIR, n) for n in ("libscann_pack_tsan.so", "_listwalk.so")]
    if r.returncode != 0 or not all(os.path.exists(p) for p in libs):
        pytest.skip("sanitizer build unavailable: " + r.stderr[-300:])
    runtime = subprocess.run(["make", "-s", "--no-print-directory", "-C", csrc, "tsan-runtime"], capture_output=True, text=True).stdout.strip()
    if not runtime or not os.path.exists(runtime):
        pytest.skip("sanitizer runtime not found: " + runtime)
    probe = subprocess.run([sys.executable, "-c", "print('up')"], env=dict(os.environ, LD_PRELOAD=runtime), capture_output=True, text=True)
    if probe.returncode != 0 or "up" not in probe.stdout:
        pytest.skip("this interpreter does not start under the ThreadSanitizer runtime: " + probe.stderr[-300:])
    return runtime


def test_the_harness_reports_a_deliberate_race(tmp_path):
    """What a green run of the test below is worth: the same set-up (preloaded runtime, ctypes into the instrumented library, Python
    threads) DOES report a race when there is one."""
    runtime = _tsan_env()
    script = tmp_path / "tsan_racy.py"
    script.write_text(_RACY)
    env = dict(os.environ, LD_PRELOAD=runtime, TSAN_OPTIONS="halt_on_error=1:exitcode=66", SCANN_COPY_THREADS="2")
    r = subprocess.run([sys.executable, str(script), ROOT, TSAN_DIR], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 66 and "ThreadSanitizer: data race" in r.stderr and "scann_host_copy" in r.stderr, (r.returncode, r.stderr[-2000:])


def test_threaded_host_code_under_thread_sanitizer(tmp_path):
    runtime = _tsan_env()
    script = tmp_path / "tsan_child.py"
    script.write_text(_CHILD)
    env = dict(os.environ, LD_PRELOAD=runtime, TSAN_OPTIONS="halt_on_error=1:exitcode=66:report_signal_unsafe=0:second_deadlock_stack=1", OMP_NUM_THREADS="2")
    env.pop("SCANN_PACK_THREADS", None)
    r = subprocess.run([sys.executable, str(script), ROOT, TSAN_DIR], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "TSAN_OK" in r.stdout and "ThreadSanitizer" not in r.stderr, (r.returncode, r.stdout[-1500:], r.stderr[-4000:])
