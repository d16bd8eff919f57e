"""CPU stand-in for ``HipModel`` / ``Engine`` in multi-rank HOST tests: the training entry points the data-parallel scripts call,
computed by the independent torch graph (tests/torch_ref.py, fp64) with the step's collectives done over a ``Rendezvous``.

Purpose: the 2-rank RCCL worker of tests/test_gpu_training.py needs two GPUs and is skipped on the one-GPU test boxes; run under
this stand-in (tests/test_host.py) its script -- the call sequence of ``Communicator``, the sharding, the global-RMSE rule, the
replica-consistency checks -- is executed on every CPU run and cannot rot.  Test infrastructure only; nothing here is shipped."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import scann_oracle as so  # noqa: E402
import torch_ref  # noqa: E402


class _Batch:
    def __init__(self, packed):
        self.packed = packed

    def free(self):
        pass

    release = free


class CpuEngine:
    def __init__(self, cfg, weights):
        self.cfg = cfg
        self.w = {k: np.asarray(v, dtype=np.float64).copy() for k, v in weights.items()}
        self.g = {k: np.zeros_like(v) for k, v in self.w.items()}
        self.m = {k: np.zeros_like(v) for k, v in self.w.items()}
        self.v = {k: np.zeros_like(v) for k, v in self.w.items()}
        self.t, self.rdzv, self.world = 0, None, 1

    # -- the data-parallel plumbing --------------------------------------------------------------------------------
    def train_begin(self):
        pass

    def comm_init(self, unique_id, rank, world):
        from scann.parallel import Rendezvous

        assert len(unique_id) == 128
        self.world = int(world)
        self.rdzv = Rendezvous(rank=rank, world=world, port=int(os.environ.get("MASTER_PORT", "29500")) + 1)  # a channel of its own

    def comm_ranks(self):
        return self.world if self.rdzv is not None else 0

    def _allreduce(self, flat):
        if self.world <= 1:
            return flat
        return np.sum(np.asarray(self.rdzv.allgather(flat.tolist()), dtype=np.float64), axis=0)

    def broadcast_weights(self, root=0):
        if self.world > 1:
            names = sorted(self.w)
            flat = self.rdzv.broadcast(np.concatenate([self.w[k].ravel() for k in names]).tolist() if self.rdzv.rank == root else None)
            off = 0
            for k in names:
                n = self.w[k].size
                self.w[k] = np.asarray(flat[off:off + n], dtype=np.float64).reshape(self.w[k].shape)
                off += n

    def allreduce_sse(self, sse, count):
        if self.world <= 1:
            return sse, count
        parts = self.rdzv.allgather([float(sse), int(count)])
        return float(sum(p[0] for p in parts)), int(sum(p[1] for p in parts))

    def allreduce_grads(self):
        names = sorted(self.g)
        flat = self._allreduce(np.concatenate([self.g[k].ravel() for k in names]))
        off = 0
        for k in names:
            n = self.g[k].size
            self.g[k] = flat[off:off + n].reshape(self.g[k].shape)
            off += n

    # -- one step, piece by piece (scann_train_forward / _backward / _adam_step) ---------------------------------------
    def upload(self, packed):
        assert packed.n_struct > 0
        return _Batch(packed)

    def _sse(self, rb, targets, grad):
        W = {k: torch.tensor(v, dtype=torch.float64, requires_grad=grad) for k, v in self.w.items()}
        y, _ = torch_ref.forward_packed(self.cfg, W, rb.packed, as_tensor=True)
        sse = ((y.reshape(-1) - torch.tensor(np.asarray(targets, dtype=np.float64))) ** 2).sum()
        return W, sse

    def train_forward(self, rb, targets, dropout=0.0, seed=0):
        rb.targets = np.asarray(targets, dtype=np.float64)
        return float(self._sse(rb, rb.targets, False)[1])

    def zero_grads(self):
        for k in self.g:
            self.g[k][...] = 0.0

    def train_backward(self, rb, sse_global, count_global):
        """accumulates d rmse / d params of THIS shard: rmse = sqrt(SSE_global / N_global) (losses.py:5-6)"""
        W, sse = self._sse(rb, rb.targets, True)
        rmse = float(np.sqrt(sse_global / count_global))
        (sse / (2.0 * rmse * count_global)).backward()
        for k, t in W.items():
            if t.grad is not None:
                self.g[k] += t.grad.numpy()

    def adam_step(self, lr_t, beta1=0.9, beta2=0.999, eps=1e-7, l2=1e-4):
        self.t += 1
        for k in self.w:
            g = self.g[k] + (2.0 * l2 * self.w[k] if k.endswith(torch_ref.REGULARIZED) else 0.0)
            self.m[k] = beta1 * self.m[k] + (1 - beta1) * g
            self.v[k] = beta2 * self.v[k] + (1 - beta2) * g * g
            step = lr_t * np.sqrt(1 - beta2 ** self.t) / (1 - beta1 ** self.t)
            self.w[k] = self.w[k] - step * self.m[k] / (np.sqrt(self.v[k]) + eps)
        self.zero_grads()

    def train_step(self, rb, targets, lr_t, dropout=0.0, seed=0, beta1=0.9, beta2=0.999, eps=1e-7, l2=1e-4):
        sse, cnt = self.allreduce_sse(self.train_forward(rb, targets), rb.packed.n_struct)
        self.zero_grads()
        self.train_backward(rb, sse, cnt)
        self.allreduce_grads()
        self.adam_step(lr_t, beta1, beta2, eps, l2)
        return sse, cnt

    def get_grads(self):
        return {k: v.astype(np.float32) for k, v in self.g.items()}

    def get_weights(self):
        return {k: v.astype(np.float32) for k, v in self.w.items()}


class CpuHipModel:
    """``HipModel(config, weights=None, device=..., seed=...)`` with the oracle's initialiser in place of keras_default_init"""

    def __init__(self, config, weights=None, device=None, infer=False, seed=None):
        self.config = config
        self.engine = CpuEngine(config, weights if weights is not None else so.init_weights(config, 0 if seed is None else seed, perturb=True))


def install():
    """Make ``scann.models.scann_model.HipModel`` the stand-in and the communicator id a constant (RCCL is never called)."""
    from scann import _hip
    from scann.models import scann_model

    scann_model.HipModel = CpuHipModel
    _hip.comm_unique_id = lambda: bytes(range(128))
