"""ctypes binding of libscann_hip.so (include/scann_hip.h) and the padded-dict <-> packed-CSR shim.

There is no CPU fallback: if the HIP library is missing or no GPU is visible the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SCANN_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libscann_hip.so")

SCANN_OK = 0
STATUS = {0: "OK", -1: "INVALID", -2: "UNSUPPORTED", -3: "NO_DEVICE", -4: "HIP", -5: "WEIGHTS", -6: "OOM", -7: "RANGE"}


class ScannHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libscann_hip: %s (%d): %s" % (STATUS.get(code, "?"), code, msg))
        self.code, self.detail = code, msg


class Config(C.Structure):
    _fields_ = [
        ("n_atoms", C.c_int32), ("embedding_dim", C.c_int32), ("local_dim", C.c_int32), ("num_head", C.c_int32),
        ("n_attention", C.c_int32), ("global_dim", C.c_int32), ("dense_out", C.c_int32), ("n_gauss", C.c_int32),
        ("gaussian_d", C.c_float), ("g_update", C.c_int32), ("use_attn_norm", C.c_int32), ("use_ga_norm", C.c_int32),
        ("use_ring", C.c_int32), ("feature_cgcnn", C.c_int32), ("relu_out", C.c_int32),
    ]


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("offset", C.c_int64), ("numel", C.c_int64)]


class Batch(C.Structure):
    _fields_ = [
        ("n_struct", C.c_int32), ("n_atom", C.c_int32), ("n_edge", C.c_int32),
        ("atomic", C.c_void_p), ("mol_offset", C.c_void_p), ("edge_offset", C.c_void_p), ("edge_col", C.c_void_p),
        ("edge_dist", C.c_void_p), ("edge_weight", C.c_void_p), ("ring", C.c_void_p), ("cgcnn", C.c_void_p),
    ]


class Profile(C.Structure):
    _fields_ = [
        ("ms_basis", C.c_float), ("ms_atom", C.c_float), ("ms_edge", C.c_float), ("ms_readout", C.c_float),
        ("ms_total", C.c_float), ("n_edge_launch", C.c_int32), ("n_atom_launch", C.c_int32), ("reserved", C.c_int32),
    ]


# every symbol include/scann_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("scann_abi_version", C.c_int, []),
    ("scann_device_count", C.c_int, []),
    ("scann_create", C.c_int, [C.POINTER(Config), C.c_int, C.POINTER(_P)]),
    ("scann_destroy", None, [_P]),
    ("scann_last_error", C.c_char_p, [_P]),
    ("scann_weight_count", C.c_int, [_P]),
    ("scann_weight_name", C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("scann_load_weights", C.c_int, [_P, _P, C.POINTER(TensorDesc), C.c_int]),
    ("scann_forward", C.c_int, [_P, C.POINTER(Batch), _P, _P]),
    ("scann_forward_padded", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P]),
    ("scann_batch_upload", C.c_int, [_P, C.POINTER(Batch), C.POINTER(_P)]),
    ("scann_batch_free", None, [_P, _P]),
    ("scann_batch_release", None, [_P, _P]),
    ("scann_forward_resident", C.c_int, [_P, _P, C.c_int]),
    ("scann_batch_download", C.c_int, [_P, _P, _P, _P]),
    ("scann_sync", C.c_int, [_P]),
    ("scann_num_streams", C.c_int, [_P]),
    ("scann_forward_profile", C.c_int, [_P, _P, C.POINTER(Profile)]),
    ("scann_edge_timing", C.c_int, [_P, C.c_int]),
    ("scann_edge_timing_read", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    ("scann_set_debug", C.c_int, [_P, C.c_int]),
    ("scann_debug_read", C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    ("scann_train_debug_read", C.c_int64, [_P, _P, C.c_char_p, _P, C.c_int64]),
    ("scann_debug_stamps", C.c_int, [_P, _P, _P, C.c_int]),
    ("scann_param_count", C.c_int64, [_P]),
    ("scann_train_begin", C.c_int, [_P]),
    ("scann_train_forward", C.c_int, [_P, _P, _P, C.c_float, C.c_uint64, C.POINTER(C.c_double)]),
    ("scann_train_backward", C.c_int, [_P, _P, C.c_double, C.c_int64]),
    ("scann_set_attention_dropout", C.c_int, [_P, C.c_float]),
    ("scann_zero_grads", C.c_int, [_P]),
    ("scann_allreduce_grads", C.c_int, [_P]),
    ("scann_allreduce_sse", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    ("scann_adam_step", C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]),
    ("scann_train_step", C.c_int, [_P, _P, _P, C.c_float, C.c_uint64] + [C.c_float] * 5 + [C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    ("scann_train_step_begin", C.c_int, [_P, _P, _P, C.c_float, C.c_uint64] + [C.c_float] * 5),
    ("scann_train_step_end", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    ("scann_get_grads", C.c_int, [_P, _P]),
    ("scann_get_weights", C.c_int, [_P, _P]),
    ("scann_comm_unique_id", C.c_int, [C.c_char_p]),
    ("scann_comm_init", C.c_int, [_P, C.c_char_p, C.c_int, C.c_int]),
    ("scann_comm_ranks", C.c_int, [_P]),
    ("scann_broadcast_weights", C.c_int, [_P, C.c_int]),
    ("scann_pack_last_error", C.c_char_p, []),
    ("scann_pack_padded", C.c_int, [C.c_int32, C.c_int32, C.c_int32] + [_P] * 17 + [C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("scann_slice_count", C.c_int, [_P, _P, _P, C.c_int32, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("scann_slice_batch", C.c_int, [_P] * 8 + [C.c_int32, C.c_int64] + [_P] * 7),
    ("scann_plan_tiles", C.c_int, [C.POINTER(Batch), C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("scann_exact_reruns", C.c_int64, [_P]),
    ("scann_device_memory", C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("scann_batch_info", C.c_int, [_P, _P, _P]),
    ("scann_count_padded", C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, C.c_int32, _P, _P, _P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("scann_upload_padded", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_int32, _P, _P, C.c_int32, _P, _P, C.POINTER(_P),
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("scann_batch_read_csr", C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    ("scann_host_copy", C.c_int, [_P, _P, C.c_int64]),
]

_lib = None
_pinned = False  # the process has pinned itself to its device's cores (Engine.__init__, multi-rank runs)


def _check_hw_queues():
    """``GPU_MAX_HW_QUEUES`` above ROCm's default of 4 makes a stream that waits for an event of a later-created stream (the upload
    copy stream, the basis-gradient stream of the training step) stall ~2 ms per wait: two queues time-sliced on one hardware pipe,
    the waiter holding it.  Measured (profiles/r03_notes.md, tools/queue_matrix.sh): ``trainer.fit`` 0.93 -> 2.85 ms per step with
    6-16 queues and 2-3 streams per handle; inference is unaffected.  The variable is the PROCESS's (torch and RCCL in the same process
    read it too), so importing this package only SAYS so; ``SCANN_FIX_HW_QUEUES=1`` asks for the value to be put back to 4 -- which works
    only because the HIP runtime reads it at its first call, i.e. if nothing in the process has touched HIP yet."""
    v = os.environ.get("GPU_MAX_HW_QUEUES")
    if not v:
        return
    try:
        n = int(v)
    except ValueError:
        return
    if n > 4:
        import warnings

        fix = os.environ.get("SCANN_FIX_HW_QUEUES") == "1"
        warnings.warn("GPU_MAX_HW_QUEUES=%d: with more than 4 hardware queues a stream waiting on a later-created stream stalls ~2 ms "
                      "per wait (trainer.fit: 3x slower, profiles/r03_notes.md); %s" % (n, "SCANN_FIX_HW_QUEUES=1: using 4 for this process "
                      "(effective only if HIP has not started yet)" if fix else "left as set -- SCANN_FIX_HW_QUEUES=1 puts it back to 4"),
                      RuntimeWarning, stacklevel=3)
        if fix:
            os.environ["GPU_MAX_HW_QUEUES"] = "4"


def _check_runtime_env():
    """Two more HIP-runtime variables whose non-default values cost this library 4-40 % (profiles/r04_notes.md, one box, one call):
    ``HIP_FORCE_DEV_KERNARG=0`` (kernel arguments fetched from host memory by every workgroup: forward -12 %, one batch per launch -16 %,
    training step +10 %), ``AMD_OPT_FLUSH`` other than 1 (-3 % / -11 % / +6 %), ``GPU_FLUSH_ON_EXECUTION=1`` (-12 % / -39 % / +93 %).
    They are left as set -- they may be deliberate, for another library in the process -- but said out loud."""
    bad = []
    if os.environ.get("HIP_FORCE_DEV_KERNARG") == "0":
        bad.append("HIP_FORCE_DEV_KERNARG=0")
    if os.environ.get("AMD_OPT_FLUSH") not in (None, "", "1"):
        bad.append("AMD_OPT_FLUSH=%s" % os.environ["AMD_OPT_FLUSH"])
    if os.environ.get("GPU_FLUSH_ON_EXECUTION") not in (None, "", "0"):
        bad.append("GPU_FLUSH_ON_EXECUTION=%s" % os.environ["GPU_FLUSH_ON_EXECUTION"])
    if bad:
        import warnings

        warnings.warn("%s in the environment: measured 4-40 %% slower forwards and up to 2x slower training steps on MI355X than the "
                      "HIP runtime's defaults (profiles/r04_notes.md)" % ", ".join(bad), RuntimeWarning, stacklevel=3)


def load_library(path=None):
    """dlopen the in-tree library and type every entry point.  Raises if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise OSError(
            "libscann_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C scann--material_amd/csrc`; this package has no CPU fallback" % p)
    if int(os.environ.get("WORLD_SIZE", "1") or 1) > 1:
        # one process per GPU: RCCL shares buffers between the ranks through dmabuf IPC, which this driver only offers with
        # the legacy mode off (else hipIpcGetMemHandle: invalid argument).  The HSA runtime reads the variable when it starts,
        # i.e. at the first HIP call of the process -- so it is set HERE, before the library is even loaded.  spawn_ranks sets
        # it for its children; ranks made by torch.distributed.run get it this way.
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    _check_hw_queues()
    _check_runtime_env()
    lib = C.CDLL(p)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.scann_abi_version() != 1:
        raise OSError("libscann_hip.so ABI version mismatch")
    if path is None:
        _lib = lib
    return lib


def comm_unique_id():
    """ncclGetUniqueId (rank 0); the 128 bytes are handed to the other ranks by the caller (e.g. over gloo)."""
    buf = C.create_string_buffer(128)
    rc = load_library().scann_comm_unique_id(buf)
    if rc != SCANN_OK:
        raise ScannHipError(rc, "scann_comm_unique_id failed")
    return buf.raw


def _ptr(a):
    return a.ctypes.data if a is not None else None  # plain address: c_void_p argtypes / fields take ints


class PackedBatch:
    """Host-side packed (CSR) batch: the arrays scann_batch_t points at."""

    def __init__(self, atomic, mol_offset, edge_offset, edge_col, edge_dist, edge_weight, pad_shape=None, gidx=None,
                 ring=None, cgcnn=None):
        self.atomic = np.ascontiguousarray(atomic, dtype=np.int32) if atomic is not None else None
        self.ring = np.ascontiguousarray(ring, dtype=np.float32) if ring is not None else None      # [n_atom, 2]
        self.cgcnn = np.ascontiguousarray(cgcnn, dtype=np.float32) if cgcnn is not None else None  # [n_atom, 92]
        self.mol_offset = np.ascontiguousarray(mol_offset, dtype=np.int32)
        self.edge_offset = np.ascontiguousarray(edge_offset, dtype=np.int32)
        self.edge_col = np.ascontiguousarray(edge_col, dtype=np.int32)
        self.edge_dist = np.ascontiguousarray(edge_dist, dtype=np.float32)
        self.edge_weight = np.ascontiguousarray(edge_weight, dtype=np.float32)
        self.pad_shape = pad_shape  # (B, M) of the padded dict it came from
        self.atom_mask = gidx       # bool [B, M] or None

    @property
    def n_struct(self):
        return int(self.mol_offset.shape[0] - 1)

    @property
    def n_atom(self):
        return int(self.edge_offset.shape[0] - 1)

    @property
    def n_edge(self):
        return int(self.edge_col.shape[0])

    def as_struct(self):
        return Batch(self.n_struct, self.n_atom, self.n_edge, _ptr(self.atomic), _ptr(self.mol_offset),
                     _ptr(self.edge_offset), _ptr(self.edge_col), _ptr(self.edge_dist), _ptr(self.edge_weight),
                     _ptr(self.ring), _ptr(self.cgcnn))

    def repad_ga(self, ga_packed):
        """Packed GlobalAttention scores -> the reference's [B, M, 1] (padded atoms score exactly 0:
        softmax of -1e9, attention.py:299-302)."""
        if self.pad_shape is None:
            return ga_packed
        out = np.zeros(self.pad_shape + (1,), dtype=np.float32)
        out[self.atom_mask, 0] = ga_packed
        return out


def concat_packed(parts):
    """Several packed batches -> one (structures are independent and the packed layout carries no per-batch padding,
    so a group of batches is just their concatenation with rebased offsets).  Used to fuse resident batches into one
    launch sequence; outputs come back in the same order."""
    parts = list(parts)
    a_off = np.cumsum([0] + [p.n_atom for p in parts])
    e_off = np.cumsum([0] + [p.n_edge for p in parts])
    cat = lambda xs: np.concatenate(xs) if all(x is not None for x in xs) else None  # noqa: E731
    return PackedBatch(
        cat([p.atomic for p in parts]),
        np.concatenate([[0]] + [p.mol_offset[1:].astype(np.int64) + a_off[i] for i, p in enumerate(parts)]),
        np.concatenate([[0]] + [p.edge_offset[1:].astype(np.int64) + e_off[i] for i, p in enumerate(parts)]),
        np.concatenate([p.edge_col.astype(np.int64) + a_off[i] for i, p in enumerate(parts)]),
        np.concatenate([p.edge_dist for p in parts]), np.concatenate([p.edge_weight for p in parts]),
        ring=cat([p.ring for p in parts]), cgcnn=cat([p.cgcnn for p in parts]))


def pack_inputs(inputs):
    """Keras input dict (scann_model.py:338-357; DataIterator.__getitem__, datagenerator.py:123-133)
    -> PackedBatch.  Real atoms are those with atom_mask set; real edges the unmasked neighbour
    slots of real atoms, kept in slot order; neighbour ids become global atom rows (what
    gather_shape + tf.gather_nd do in the reference, custom_layers.py:18-28, attention.py:136)."""
    lib = load_library()
    atomic = np.asarray(inputs["atomic"])
    cgcnn = None
    if atomic.ndim == 3:  # feature="cgcnn": [B, M, 92] float features instead of atomic numbers (scann_model.py:334)
        cgcnn, atomic = np.ascontiguousarray(atomic, dtype=np.float32), None
        if cgcnn.shape[2] != 92:
            raise ValueError("cgcnn features must be [B, M, 92]")
    else:
        atomic = np.ascontiguousarray(atomic, dtype=np.int32)
    amask = np.asarray(inputs["atom_mask"])
    if amask.ndim == 3:
        amask = amask[..., 0]
    amask = np.ascontiguousarray(amask != 0)
    nbr = np.ascontiguousarray(inputs["neighbors"], dtype=np.int32)
    nmask = np.ascontiguousarray(np.asarray(inputs["neighbor_mask"]) != 0)
    B, M = amask.shape
    if nbr.ndim != 3 or nbr.shape[:2] != (B, M) or nmask.shape != nbr.shape or \
            (atomic is not None and atomic.shape != (B, M)) or (cgcnn is not None and cgcnn.shape[:2] != (B, M)):
        raise ValueError("inconsistent input shapes")
    N = nbr.shape[2]
    dist = np.ascontiguousarray(inputs["neighbor_distance"], dtype=np.float32)
    wgt = np.ascontiguousarray(inputs["neighbor_weight"], dtype=np.float32)
    ring = np.ascontiguousarray(inputs["ring_aromatic"], dtype=np.float32) if "ring_aromatic" in inputs else None
    if dist.shape != nbr.shape or wgt.shape != nbr.shape or (ring is not None and ring.shape != (B, M, 2)):
        raise ValueError("inconsistent input shapes")
    o_atomic = np.empty(B * M, np.int32) if atomic is not None else None
    o_cgcnn = np.empty((B * M, 92), np.float32) if cgcnn is not None else None
    o_ring = np.empty((B * M, 2), np.float32) if ring is not None else None
    o_mol, o_eoff = np.empty(B + 1, np.int32), np.empty(B * M + 1, np.int32)
    o_col, o_dist, o_wgt = np.empty(B * M * N, np.int32), np.empty(B * M * N, np.float32), np.empty(B * M * N, np.float32)
    row_of = np.empty(B * M, np.int32)
    na, ne = C.c_int32(0), C.c_int32(0)
    rc = lib.scann_pack_padded(B, M, N, _ptr(atomic), _ptr(cgcnn), _ptr(amask), _ptr(nbr), _ptr(nmask), _ptr(wgt), _ptr(dist),
                               _ptr(ring), _ptr(o_atomic), _ptr(o_cgcnn), _ptr(o_ring), _ptr(o_mol), _ptr(o_eoff), _ptr(o_col),
                               _ptr(o_dist), _ptr(o_wgt), _ptr(row_of), C.byref(na), C.byref(ne))
    if rc != SCANN_OK:
        raise ValueError((lib.scann_pack_last_error() or b"").decode())
    na, ne = na.value, ne.value
    return PackedBatch(o_atomic[:na] if o_atomic is not None else None, o_mol, o_eoff[:na + 1], o_col[:ne], o_dist[:ne],
                       o_wgt[:ne], pad_shape=(B, M), gidx=amask, ring=o_ring[:na] if o_ring is not None else None,
                       cgcnn=o_cgcnn[:na] if o_cgcnn is not None else None)


def plan_tiles(packed, tile_rows=64, tile_atoms=24, allow_chunks=True):
    """The edge-tile plan scann_batch_upload would build for `packed` (host only): (rows_per_tile, tiles[n,4], part[n], n_slots)."""
    lib = load_library()
    cap = packed.n_atom + packed.n_edge // 32 + 2
    tiles, part = np.empty((cap, 4), np.int32), np.empty(cap, np.int32)
    nt, ns = C.c_int32(0), C.c_int32(0)
    st = packed.as_struct()
    rc = lib.scann_plan_tiles(C.byref(st), tile_rows, tile_atoms, int(allow_chunks), cap, _ptr(tiles), _ptr(part), C.byref(nt), C.byref(ns))
    if rc < 0:
        raise ScannHipError(rc, (lib.scann_pack_last_error() or b"").decode())
    return rc, tiles[:nt.value].copy(), part[:nt.value].copy(), ns.value


def slice_dataset(ds_mol_offset, ds_edge_offset, ds_atomic, ds_ring, ds_edge_local, ds_edge_dist, ds_edge_weight, sel):
    """Structures `sel` of a dataset kept in CSR form -> PackedBatch (scann_slice_batch; the batch that
    DataIterator.__getitem__, datagenerator.py:69-135, would assemble from nested lists)."""
    lib = load_library()
    sel = np.ascontiguousarray(sel, dtype=np.int64)
    n_total = int(ds_mol_offset.shape[0] - 1)
    na, ne = C.c_int64(0), C.c_int64(0)
    if lib.scann_slice_count(_ptr(ds_mol_offset), _ptr(ds_edge_offset), _ptr(sel), len(sel), n_total, C.byref(na), C.byref(ne)) != SCANN_OK:
        raise ValueError((lib.scann_pack_last_error() or b"").decode())
    na, ne = na.value, ne.value
    o_atomic, o_mol, o_eoff = np.empty(na, np.int32), np.empty(len(sel) + 1, np.int32), np.empty(na + 1, np.int32)
    o_ring = np.empty((na, 2), np.float32) if ds_ring is not None else None
    o_col, o_dist, o_wgt = np.empty(ne, np.int32), np.empty(ne, np.float32), np.empty(ne, np.float32)
    rc = lib.scann_slice_batch(_ptr(ds_mol_offset), _ptr(ds_edge_offset), _ptr(ds_atomic), _ptr(ds_ring), _ptr(ds_edge_local),
                               _ptr(ds_edge_dist), _ptr(ds_edge_weight), _ptr(sel), len(sel), n_total, _ptr(o_atomic),
                               _ptr(o_ring), _ptr(o_mol), _ptr(o_eoff), _ptr(o_col), _ptr(o_dist), _ptr(o_wgt))
    if rc != SCANN_OK:
        raise ValueError((lib.scann_pack_last_error() or b"").decode())
    return PackedBatch(o_atomic, o_mol, o_eoff, o_col, o_dist, o_wgt, ring=o_ring)


def _mask_arg(m):
    """A mask of the Keras input dict as the C ABI takes it: (contiguous array, element size 1 | 4).  ONE truth rule on every path --
    NumPy's `m != 0` (what the reference's bool(...) / cast-to-float32 masks mean, datagenerator.py:123-133): bool / uint8 / int8 and
    float32 masks go through AS THEY ARE (no `!= 0` pass over a whole dataset's neighbour slots: a byte is set iff non-zero; a float32
    word is set iff any bit but the sign is, i.e. -0.0 is unset and NaN is set, exactly `!= 0`); anything else -- int32 included, whose
    0x80000000 the 4-byte rule would read as unset -- is compared once."""
    m = np.asarray(m)
    if m.dtype in (np.bool_, np.uint8, np.int8):
        return np.ascontiguousarray(m), 1
    if m.dtype == np.float32:
        return np.ascontiguousarray(m), 4
    return np.ascontiguousarray(m != 0), 1


def _mask_bytes(m):
    """The same truth rule as one byte per element (scann_forward_padded takes uint8 masks): a cast to uint8 would turn 0.5 into 0 and
    wrap 256.0 to 0."""
    m, size = _mask_arg(m)
    return (m if size == 1 else np.ascontiguousarray(m != 0)).view(np.uint8)


def count_padded(inputs):
    """The host half of the device packing (scann_count_padded; host only): masks -> (mol_offset, edge_offset, row_of [B, M])."""
    lib = load_library()
    amask, asz = _mask_arg(inputs["atom_mask"])
    nmask, nsz = _mask_arg(inputs["neighbor_mask"])
    B, M, N = nmask.shape
    mol, eoff, row_of = np.empty(B + 1, np.int32), np.empty(B * M + 1, np.int32), np.empty(B * M, np.int32)
    na, ne = C.c_int32(0), C.c_int32(0)
    if lib.scann_count_padded(B, M, N, _ptr(amask), asz, _ptr(nmask), nsz, _ptr(mol), _ptr(eoff), _ptr(row_of), C.byref(na), C.byref(ne)) != SCANN_OK:
        raise ValueError((lib.scann_pack_last_error() or b"").decode())
    return mol, eoff[:na.value + 1], row_of.reshape(B, M)


class PaddedInfo:
    """What the host knows of a batch that was packed on the DEVICE (Engine.upload_padded): counts and where the real atoms sit."""

    def __init__(self, n_struct, n_atom, n_edge, atom_mask):
        self.n_struct, self.n_atom, self.n_edge = n_struct, n_atom, n_edge
        self.atom_mask = atom_mask  # [B, M] bool
        self.pad_shape = atom_mask.shape

    def repad_ga(self, ga_packed):
        out = np.zeros(self.pad_shape + (1,), dtype=np.float32)
        out[self.atom_mask, 0] = ga_packed
        return out


class ResidentBatch:
    """A batch uploaded to HBM (scann_dbatch_t) together with its workspace."""

    def __init__(self, engine, packed, handle):
        self.engine, self.packed, self._h = engine, packed, handle

    def free(self):
        if self._h is not None and self.engine._h is not None:
            self.engine.lib.scann_batch_free(self.engine._h, self._h)
        self._h = None

    def release(self):
        """free() without the device-wide synchronisation: after train_step_end() of the step that used the batch"""
        if self._h is not None and self.engine._h is not None:
            self.engine.lib.scann_batch_release(self.engine._h, self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    """One scann_handle_t: the forward graph of create_model on one GPU."""

    def __init__(self, cfg_struct, device=0):
        self.lib = load_library()
        self._h = None
        global _pinned
        if not _pinned and int(os.environ.get("WORLD_SIZE", "1") or 1) > 1:
            # one rank per GPU means one rank per NUMA neighbourhood: the rank's FIRST engine pins the calling thread (and the threads it
            # starts afterwards; threads that already exist keep their mask) to the cores next to the device it actually opens
            # (scann/parallel/affinity.py; SCANN_NO_AFFINITY=1, or both *_VISIBLE_DEVICES set, leave the affinity alone)
            from .parallel.affinity import pin_to_device

            pin_to_device(int(device))
            _pinned = True
        h = _P()
        rc = self.lib.scann_create(C.byref(cfg_struct), int(device), C.byref(h))
        if rc != SCANN_OK:
            raise ScannHipError(rc, (self.lib.scann_last_error(None) or b"").decode())
        self._h = h
        self.cfg = cfg_struct
        self.device = device
        self.training = False

    def _check(self, rc):
        if rc != SCANN_OK:
            raise ScannHipError(rc, (self.lib.scann_last_error(self._h) or b"").decode())

    def close(self):
        if self._h is not None:
            self.lib.scann_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def weight_specs(self):
        out = []
        for i in range(self.lib.scann_weight_count(self._h)):
            name, r, c = C.c_char_p(), C.c_int64(), C.c_int64()
            self._check(self.lib.scann_weight_name(self._h, i, C.byref(name), C.byref(r), C.byref(c)))
            out.append((name.value.decode(), (r.value, c.value) if c.value else (r.value,)))
        return out

    def load_weights(self, weights):
        specs = self.weight_specs()
        chunks, descs, off = [], [], 0
        names = []
        for name, shape in specs:
            if name not in weights:
                raise ScannHipError(-5, "missing tensor " + name)
            t = np.ascontiguousarray(weights[name], dtype=np.float32)
            if tuple(t.shape) != tuple(shape):
                raise ScannHipError(-5, "tensor %s has shape %s, expected %s" % (name, t.shape, shape))
            chunks.append(t.ravel())
            names.append(name.encode())
            descs.append((names[-1], off, t.size))
            off += t.size
        blob = np.concatenate(chunks).astype(np.float32)
        arr = (TensorDesc * len(descs))(*[TensorDesc(n, o, s) for n, o, s in descs])
        self._check(self.lib.scann_load_weights(self._h, _ptr(blob), arr, len(descs)))

    def forward(self, packed, want_ga=True):
        y = np.empty(packed.n_struct, dtype=np.float32)
        ga = np.empty(packed.n_atom, dtype=np.float32) if want_ga else None
        st = packed.as_struct()
        self._check(self.lib.scann_forward(self._h, C.byref(st), _ptr(y), _ptr(ga)))
        return y, ga

    def forward_padded(self, inputs, want_ga=True):
        """The padded Keras dict straight through the C ABI (native CSR packing)."""
        atomic = np.ascontiguousarray(inputs["atomic"], dtype=np.int32)
        B, M = atomic.shape
        amask = _mask_bytes(np.asarray(inputs["atom_mask"]).reshape(B, M))
        nbr = np.ascontiguousarray(inputs["neighbors"], dtype=np.int32)
        N = nbr.shape[2]
        nmask = _mask_bytes(inputs["neighbor_mask"])
        wgt = np.ascontiguousarray(inputs["neighbor_weight"], dtype=np.float32)
        dst = np.ascontiguousarray(inputs["neighbor_distance"], dtype=np.float32)
        if nbr.shape != (B, M, N) or nmask.shape != nbr.shape or wgt.shape != nbr.shape or dst.shape != nbr.shape:
            raise ValueError("inconsistent input shapes")
        y = np.empty(B, dtype=np.float32)
        ga = np.empty((B, M, 1), dtype=np.float32) if want_ga else None
        self._check(self.lib.scann_forward_padded(self._h, B, M, N, _ptr(atomic), _ptr(amask), _ptr(nbr), _ptr(nmask),
                                                  _ptr(wgt), _ptr(dst), _ptr(y), _ptr(ga)))
        return y, ga

    def upload(self, packed):
        st = packed.as_struct()
        db = _P()
        self._check(self.lib.scann_batch_upload(self._h, C.byref(st), C.byref(db)))
        return ResidentBatch(self, packed, db)

    def upload_padded(self, inputs):
        """The padded Keras input dict (or row views of one) -> a resident batch, packed to CSR ON THE DEVICE (scann_upload_padded): the
        host reads the masks, the payload arrays cross the bus as they are.  feature = "atomic" without ring, inference handles."""
        atomic = np.ascontiguousarray(inputs["atomic"], dtype=np.int32)
        B, M = atomic.shape
        amask, asz = _mask_arg(np.asarray(inputs["atom_mask"]).reshape(B, M))
        nbr = np.ascontiguousarray(inputs["neighbors"], dtype=np.int32)
        N = nbr.shape[2]
        nmask, nsz = _mask_arg(inputs["neighbor_mask"])
        wgt = np.ascontiguousarray(inputs["neighbor_weight"], dtype=np.float32)
        dst = np.ascontiguousarray(inputs["neighbor_distance"], dtype=np.float32)
        if nbr.shape != (B, M, N) or nmask.shape != nbr.shape or wgt.shape != nbr.shape or dst.shape != nbr.shape or amask.shape != (B, M):
            raise ValueError("inconsistent input shapes")
        db, na, ne = _P(), C.c_int32(0), C.c_int32(0)
        self._check(self.lib.scann_upload_padded(self._h, B, M, N, _ptr(atomic), _ptr(amask), asz, _ptr(nbr), _ptr(nmask), nsz, _ptr(wgt),
                                                 _ptr(dst), C.byref(db), C.byref(na), C.byref(ne)))
        return ResidentBatch(self, PaddedInfo(B, na.value, ne.value, amask != 0 if amask.dtype != np.bool_ else amask), db)

    def read_csr(self, rb):
        """The packed arrays of a resident batch, copied back (test hook: device packing against the host packer)."""
        p = rb.packed
        atomic, mol, eoff = np.empty(p.n_atom, np.int32), np.empty(p.n_struct + 1, np.int32), np.empty(p.n_atom + 1, np.int32)
        col, dist, wgt = np.empty(p.n_edge, np.int32), np.empty(p.n_edge, np.float32), np.empty(p.n_edge, np.float32)
        self._check(self.lib.scann_batch_read_csr(self._h, rb._h, _ptr(atomic), _ptr(mol), _ptr(eoff), _ptr(col), _ptr(dist), _ptr(wgt)))
        return {"atomic": atomic, "mol_offset": mol, "edge_offset": eoff, "edge_col": col, "edge_dist": dist, "edge_weight": wgt}

    def forward_resident(self, rb, slot=0):
        self._check(self.lib.scann_forward_resident(self._h, rb._h, int(slot)))

    def download(self, rb, want_ga=True):
        y = np.empty(rb.packed.n_struct, dtype=np.float32)
        ga = np.empty(rb.packed.n_atom, dtype=np.float32) if want_ga else None
        self._check(self.lib.scann_batch_download(self._h, rb._h, _ptr(y), _ptr(ga)))
        return y, ga

    def sync(self):
        self._check(self.lib.scann_sync(self._h))

    def exact_reruns(self):
        """forwards this handle has re-run on the exact-fp32 kernels because an activation left the split-fp16 range"""
        return int(self.lib.scann_exact_reruns(self._h))

    def device_memory(self):
        """(free, total) bytes of the handle's device (hipMemGetInfo)"""
        f, t = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.scann_device_memory(self._h, C.byref(f), C.byref(t)))
        return int(f.value), int(t.value)

    def batch_info(self, rb):
        out = np.zeros(8, dtype=np.int32)
        self._check(self.lib.scann_batch_info(self._h, rb._h, _ptr(out)))
        keys = ("structs", "atoms", "edges", "big_atoms", "merge_slots", "max_degree", "tiles", "tile_rows")
        return dict(zip(keys, (int(v) for v in out)))

    def num_streams(self):
        return self.lib.scann_num_streams(self._h)

    def profile(self, rb):
        p = Profile()
        self._check(self.lib.scann_forward_profile(self._h, rb._h, C.byref(p)))
        return {k: getattr(p, k) for k, _ in Profile._fields_ if k != "reserved"}

    # -- training (scann_model.py:199-241) --------------------------------------------------------------------
    def _unflatten(self, flat):
        out, off = {}, 0
        for name, shape in self.weight_specs():
            n = int(np.prod(shape))
            out[name] = flat[off:off + n].reshape(shape).copy()
            off += n
        return out

    def param_count(self):
        return int(self.lib.scann_param_count(self._h))

    def train_begin(self):
        self._check(self.lib.scann_train_begin(self._h))
        self.training = True  # uploads of a training handle carry the reverse adjacency: packed on the host

    def train_forward(self, rb, targets, dropout=0.0, seed=0):
        t = np.ascontiguousarray(targets, dtype=np.float32)
        sse = C.c_double()
        self._check(self.lib.scann_train_forward(self._h, rb._h, _ptr(t), float(dropout), int(seed), C.byref(sse)))
        return sse.value

    def train_backward(self, rb, sse_global, count_global):
        self._check(self.lib.scann_train_backward(self._h, rb._h, float(sse_global), int(count_global)))

    def set_attention_dropout(self, p):
        self._check(self.lib.scann_set_attention_dropout(self._h, float(p)))

    def zero_grads(self):
        self._check(self.lib.scann_zero_grads(self._h))

    def allreduce_grads(self):
        self._check(self.lib.scann_allreduce_grads(self._h))

    def allreduce_sse(self, sse, count):
        a, b = C.c_double(sse), C.c_int64(count)
        self._check(self.lib.scann_allreduce_sse(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def adam_step(self, lr_t, beta1=0.9, beta2=0.999, eps=1e-7, l2=1e-4):
        self._check(self.lib.scann_adam_step(self._h, float(lr_t), float(beta1), float(beta2), float(eps), float(l2)))

    def train_step(self, rb, targets, lr_t, dropout=0.0, seed=0, beta1=0.9, beta2=0.999, eps=1e-7, l2=1e-4):
        """forward + backward + (all-reduces) + Adam in one asynchronous sequence; returns the global (sse, count)"""
        t = np.ascontiguousarray(targets, dtype=np.float32)
        sse, cnt = C.c_double(), C.c_int64()
        self._check(self.lib.scann_train_step(self._h, rb._h, _ptr(t), float(dropout), int(seed), float(lr_t), float(beta1), float(beta2),
                                              float(eps), float(l2), C.byref(sse), C.byref(cnt)))
        return sse.value, cnt.value

    def train_step_begin(self, rb, targets, lr_t, dropout=0.0, seed=0, beta1=0.9, beta2=0.999, eps=1e-7, l2=1e-4):
        """enqueue one step and return; the next batch may be uploaded before train_step_end()"""
        t = np.ascontiguousarray(targets, dtype=np.float32)
        self._check(self.lib.scann_train_step_begin(self._h, rb._h, _ptr(t), float(dropout), int(seed), float(lr_t), float(beta1),
                                                    float(beta2), float(eps), float(l2)))

    def train_step_end(self):
        """wait for the OLDEST step in flight (up to two may be); returns its global (sse, count, sum |y - target|)"""
        sse, cnt, sabs = C.c_double(), C.c_int64(), C.c_double()
        self._check(self.lib.scann_train_step_end(self._h, C.byref(sse), C.byref(cnt), C.byref(sabs)))
        return sse.value, cnt.value, sabs.value

    def get_grads(self):
        flat = np.empty(self.param_count(), dtype=np.float32)
        self._check(self.lib.scann_get_grads(self._h, _ptr(flat)))
        return self._unflatten(flat)

    def get_weights(self):
        flat = np.empty(self.param_count(), dtype=np.float32)
        self._check(self.lib.scann_get_weights(self._h, _ptr(flat)))
        return self._unflatten(flat)

    def comm_init(self, unique_id, rank, world):
        self._check(self.lib.scann_comm_init(self._h, unique_id, int(rank), int(world)))

    def comm_ranks(self):
        """ranks of the RCCL communicator as RCCL counts them (ncclCommCount); 0 without a communicator"""
        n = self.lib.scann_comm_ranks(self._h)
        if n < 0:
            self._check(n)
        return int(n)

    def broadcast_weights(self, root=0):
        self._check(self.lib.scann_broadcast_weights(self._h, int(root)))

    def edge_timing(self, every):
        self._check(self.lib.scann_edge_timing(self._h, int(every)))

    def edge_timing_read(self):
        us, n, ed = C.c_double(), C.c_int64(), C.c_double()
        self._check(self.lib.scann_edge_timing_read(self._h, C.byref(us), C.byref(n), C.byref(ed)))
        return us.value, n.value, ed.value

    def set_debug(self, on):
        self._check(self.lib.scann_set_debug(self._h, int(bool(on))))

    def debug_stamps(self, rb, max_tiles=4096):
        out = np.zeros((max_tiles, 16), dtype=np.uint64)
        n = self.lib.scann_debug_stamps(self._h, rb._h, _ptr(out), int(max_tiles))
        if n < 0:
            self._check(n)
        return out[:n]

    def train_debug_read(self, rb, name, width):
        """A tensor of the plain-fp32 backward's readout stage (scann_train_debug_read): rows x `width`."""
        rows = rb.packed.n_struct if name in ("rep", "drep") else rb.packed.n_atom
        out = np.empty((rows, int(width)), dtype=np.float32)
        n = self.lib.scann_train_debug_read(self._h, rb._h, name.encode(), _ptr(out), out.size)
        if n < 0:
            self._check(int(n))
        assert n == out.size, (n, out.shape)
        return out

    def debug_read(self, rb, what, layer):
        n = rb.packed.n_edge if what in (1, 3, 4, 5, 6) else rb.packed.n_atom
        out = np.empty((n, 128), dtype=np.float32)
        self._check(self.lib.scann_debug_read(self._h, rb._h, int(what), int(layer), _ptr(out)))
        return out
