"""Training loop behind ``SCANN.train`` -- the counterpart of ``model.compile(...)`` + ``model.fit(...)`` +
callbacks in the reference (scann_model.py:163-241, custom_layers.py:78-179), driving the HIP training step
(``scann_train_forward`` / ``scann_train_backward`` / ``scann_allreduce_grads`` / ``scann_adam_step``).

Data parallelism: every rank sees the same global batch (same shuffle seed), keeps its contiguous
slice of structures, and the ranks exchange exactly two things per step over RCCL: the scalar SSE/count (the loss is
``sqrt(mean_batch((y_hat - y)^2))`` over the GLOBAL batch, losses.py:5-6) and one flat fp32 gradient all-reduce.
"""
from __future__ import annotations

import math
import os
import queue
import threading
import time

import numpy as np

from .. import _hip
from ..parallel import rank_slice, slice_packed


def cosine_decay(step, initial_lr, decay_steps, alpha):
    """tf.keras.optimizers.schedules.CosineDecay as configured at scann_model.py:203-208."""
    s = min(float(step), float(decay_steps))
    cos = 0.5 * (1.0 + math.cos(math.pi * s / float(decay_steps)))
    return initial_lr * ((1.0 - alpha) * cos + alpha)


class SGDRC:
    """Warm-restart schedule with validation-triggered decay, behaviour of the reference callback
    (custom_layers.py:78-179): the rate stays at ``lr_max`` until ``val_mae <= trigger_val_mae``; from then on every
    epoch advances a cosine cycle of ``ti`` epochs (``ti *= tmult`` at each restart) between ``lr_min`` and the current
    warm-up peak; whenever validation improves, the NEXT peak is set to max(peak / lr_max_compression, current lr)."""

    def __init__(self, lr_max, lr_min, lr_max_compression=5, t0=10, tmult=1, trigger_val_mae=9999, show_lr=True):
        self.lr_max, self.lr_min = lr_max, lr_min
        self.lr_max_compression = lr_max_compression
        self.t0, self.tmult = t0, tmult
        self.trigger_val_mae = trigger_val_mae
        self.show_lr = show_lr
        self.reset()

    def reset(self):
        self.triggered = False
        self.peak_next = self.peak = self.lr = self.lr_max
        self.ti, self.tcur = self.t0, 1
        self.best_val_mae = 9999

    on_train_begin = lambda self, logs=None: self.reset()  # noqa: E731

    def on_epoch_end(self, epoch, logs):
        val = logs["val_mae"]
        if not self.triggered and val <= self.trigger_val_mae:
            self.triggered = True
        if self.triggered and val < self.best_val_mae:
            self.best_val_mae = val
            self.peak_next = max(self.peak / self.lr_max_compression, self.lr) if self.lr_max_compression > 0 else self.lr
        if self.show_lr:
            print("sgdr_triggered = %s, current_lr = %f, next_warmup_lr = %f, next_warmup = %d"
                  % (self.triggered, self.lr, self.peak_next, self.ti - self.tcur))

    def lr_scheduler(self, epoch):
        if not self.triggered:
            return self.lr
        self.tcur += 1
        if self.tcur > self.ti:
            self.ti = int(self.tmult * self.ti)
            self.tcur = 1
            self.peak = self.peak_next
        self.lr = float(self.lr_min + (self.peak - self.lr_min) * (1 + np.cos(self.tcur / self.ti * np.pi)) / 2.0)
        return self.lr


class Communicator:
    """RCCL communicator of the training ranks (one process per GPU; launched by ``torch.distributed.run`` or by
    ``scann.parallel.spawn_ranks`` -- RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment).  The 128-byte
    ncclUniqueId travels from rank 0 over the loopback rendezvous (``scann.parallel.Rendezvous``: plain TCP, no torch);
    after ``scann_comm_init`` rank 0's parameters are broadcast so that every replica starts from the SAME model (the
    reference builds one model in one process, scann_model.py:77; each rank here drew its own initialiser)."""

    def __init__(self, engine, rendezvous=None):
        from ..parallel import Rendezvous

        self.engine = engine
        self.rdzv = rendezvous if rendezvous is not None else Rendezvous()
        self.rank, self.world = self.rdzv.rank, self.rdzv.world
        if self.world > 1:
            # (HSA_ENABLE_IPC_MODE_LEGACY=0, which RCCL across processes needs on this driver, is set by _hip.load_library
            # before the first HIP call of a rank -- here it would be too late)
            uid = self.rdzv.broadcast(_hip.comm_unique_id() if self.rank == 0 else None)
            engine.comm_init(uid, self.rank, self.world)
            engine.broadcast_weights(0)

    def shard(self, packed):
        if self.world == 1:
            return packed, slice(0, packed.n_struct)
        lo, hi = rank_slice(packed.n_struct, self.rank, self.world)
        return slice_packed(packed, lo, hi), slice(lo, hi)

    def sum_pair(self, a, b):
        return self.engine.allreduce_sse(a, b) if self.world > 1 else (a, b)


def dp_batches(iterator, world):
    """-> (number of steps, fold_tail).  With ``world`` ranks every rank needs at least one structure of every global batch
    (an empty shard cannot run a step, and the other ranks would wait in the step's all-reduce for ever).  A final batch with
    fewer structures than ranks is therefore folded into the batch before it -- the same decision on every rank, because it
    depends on the dataset size, the batch size and the world size only."""
    n = len(iterator)
    if world <= 1 or n <= 1:
        return n, False
    total, bs = getattr(iterator, "indexes", None), getattr(iterator, "batch_size", None)
    if total is None or not bs:
        return n, False
    tail = len(total) - (n - 1) * int(bs)
    return (n - 1, True) if 0 < tail < world else (n, False)


class _Prefetch:
    """Host pipeline of ``fit``: batch k+1 is assembled (``iterator[k+1]`` + CSR packing + this rank's slice) on a worker
    thread while the GPU runs step k.  The reference gets the same overlap from Keras' ``workers=4,
    use_multiprocessing=True`` (scann_model.py:239-240)."""

    def __init__(self, iterator, comm):
        self.it, self.comm = iterator, comm
        self.n, self.fold_tail = dp_batches(iterator, comm.world)
        total = getattr(iterator, "indexes", None)
        if comm.world > 1 and total is not None and len(total) < comm.world:
            raise ValueError("data-parallel run with %d ranks over a dataset of %d structures: every rank needs at least one "
                             "structure per step" % (comm.world, len(total)))
        self.q = queue.Queue(maxsize=2)
        self.stop = threading.Event()
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def _one(self, b):
        fold = self.fold_tail and b == self.n - 1  # the last step also takes the short final batch
        part = getattr(self.it, "batch_part", None)
        if part is not None and self.comm.world > 1:  # PackedDataset: only this rank's structures are ever packed
            shard, target = part(b, self.comm.rank, self.comm.world, to_end=True) if fold else part(b, self.comm.rank, self.comm.world)
            return shard, np.asarray(target, dtype=np.float32)
        inputs, target = self.it[b]
        packed = inputs if isinstance(inputs, _hip.PackedBatch) else _hip.pack_inputs(inputs)
        target = np.asarray(target, dtype=np.float32)
        if fold:
            inputs2, target2 = self.it[b + 1]
            packed = _hip.concat_packed([packed, inputs2 if isinstance(inputs2, _hip.PackedBatch) else _hip.pack_inputs(inputs2)])
            target = np.concatenate([target, np.asarray(target2, dtype=np.float32)])
        shard, sl = self.comm.shard(packed)
        return shard, target[sl]

    def _put(self, item):
        while not self.stop.is_set():
            try:
                self.q.put(item, timeout=0.1)
                return True
            except queue.Full:
                pass
        return False

    def _run(self):
        try:
            for b in range(self.n):
                if not self._put(self._one(b)):
                    return
        except BaseException as e:  # surfaced in the training thread
            self._put(e)

    def close(self):
        """Stop the worker (the consumer gave up mid-epoch): nothing stays blocked on the queue holding packed batches."""
        self.stop.set()
        while True:
            try:
                self.q.get_nowait()
            except queue.Empty:
                break
        self.t.join(timeout=5.0)

    def __iter__(self):
        try:
            for _ in range(self.n):
                item = self.q.get()
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            self.close()


def fit(scann, epochs=1000, dropout=0.1, verbose=True):
    """``SCANN.train`` body.  Returns the history dict (keys as in Keras: loss, mae, r2_square, val_*, lr)."""
    cfg = scann.config
    hy = cfg["hyper"]
    model = scann.model
    eng = model.engine
    eng.train_begin()
    attn_drop = 0.05 if cfg["model"].get("use_drop") else 0.0  # attention.py:115-116
    comm = Communicator(eng)
    train_it, valid_it = scann.trainIter, scann.validIter
    steps_per_epoch = len(train_it)
    out_dir = "{}_{}".format(hy["save_path"], hy["target"])
    ckpt = "{}/models/model_{}.h5".format(out_dir, hy["target"])
    sgdr = None
    if hy["scheduler"] == "sgdr":  # scann_model.py:181-193
        sgdr = SGDRC(lr_min=hy["min_lr"], lr_max=hy["lr"], t0=50, tmult=2, lr_max_compression=1.2, trigger_val_mae=300)
        sgdr.on_train_begin()
    decay_steps = 0.5 * steps_per_epoch * epochs
    alpha = hy["min_lr"] / hy["lr"]
    hist = {k: [] for k in ("loss", "mae", "r2_square", "val_loss", "val_mae", "val_r2_square", "lr")}
    best, wait, it = float("inf"), 0, 0
    l2 = 1e-4

    def run_epoch(iterator, training, epoch_lr):
        nonlocal it
        sse_t = sabs_t = sy = syy = 0.0
        n_t = 0
        loss_sum = 0.0
        # Keras Dropout layers are the identity outside training (attention.py:115-116,191): validation runs without them
        eng.set_attention_dropout(attn_drop if training else 0.0)
        def finish(rb, tgt, pair):
            # end of a step: wait for the device, account.  Training steps report their own global {sse, count, sum |y - t|};
            # validation batches were run synchronously and hand over (sse, count) plus the downloaded predictions.
            nonlocal sse_t, sabs_t, sy, syy, n_t, loss_sum
            if pair is None:
                sse_g, cnt_g, sabs_g = eng.train_step_end()
                rb.release()  # no device-wide synchronisation: the next step may already be running
                sabs_t += sabs_g if comm.rank == 0 else 0.0  # already summed over the ranks
            else:
                sse_g, cnt_g = pair
                y, _ = eng.download(rb, want_ga=False)
                rb.free()
                sabs_t += float(np.abs(y - tgt).sum())
            loss_sum += math.sqrt(sse_g / cnt_g) * cnt_g
            # this rank's partial sums; they are linear, so ONE reduction over the ranks at the end of the epoch is enough
            sy += float(tgt.sum()); syy += float((tgt.astype(np.float64) ** 2).sum())
            sse_t += sse_g; n_t += cnt_g

        pending = []
        try:
            for shard, tgt in _Prefetch(iterator, comm):
                rb = eng.upload(shard)  # H2D of batch k + 1 while step k (if any) is still running on the device
                seed = (it * 7919 + 17) & 0xFFFFFFFF
                if training:
                    # one asynchronous sequence on the device: forward, global {sse, count}, backward, gradient all-reduce, Adam;
                    # step k + 1 is enqueued BEFORE step k is waited for, so the device never idles while the host works
                    lr_t = (epoch_lr if sgdr is not None else cosine_decay(it, hy["lr"], decay_steps, alpha)) / (1.0 + 1e-5 * it)
                    pending.append((rb, tgt, None))
                    eng.train_step_begin(rb, tgt, lr_t, dropout=dropout, seed=seed, l2=l2)
                    it += 1
                    if len(pending) == 2:
                        finish(*pending.pop(0))
                else:
                    pending.append((rb, tgt, False))
                    sse = eng.train_forward(rb, tgt, dropout=0.0, seed=seed)
                    pending.pop()
                    finish(rb, tgt, comm.sum_pair(sse, shard.n_struct))
            while pending:
                finish(*pending.pop(0))
        except BaseException:
            # a step in flight holds its resident batch: end what can be ended and release the batches before the error travels
            # on (a failed peer may never complete the collective of a step, so every call here is guarded)
            for rb, _, pair in pending:
                try:
                    if pair is None:
                        eng.train_step_end()
                except Exception:
                    pass
                try:
                    rb.free()
                except Exception:
                    pass
            raise
        sabs_t, _ = comm.sum_pair(sabs_t, 0)
        sy, _ = comm.sum_pair(sy, 0)
        syy, _ = comm.sum_pair(syy, 0)
        ss_tot = syy - sy * sy / max(n_t, 1)
        return loss_sum / max(n_t, 1), sabs_t / max(n_t, 1), 1.0 - sse_t / (ss_tot + 1e-7)

    for epoch in range(epochs):
        t0 = time.time()
        epoch_lr = sgdr.lr_scheduler(epoch) if sgdr is not None else None
        loss, mae, r2 = run_epoch(train_it, True, epoch_lr)
        if hasattr(train_it, "on_epoch_end"):
            train_it.on_epoch_end()
        vloss, vmae, vr2 = run_epoch(valid_it, False, None)
        cur_lr = epoch_lr if sgdr is not None else cosine_decay(it, hy["lr"], decay_steps, alpha)
        for k, v in zip(hist, (loss, mae, r2, vloss, vmae, vr2, cur_lr)):
            hist[k].append(float(v))
        if verbose and comm.rank == 0:
            print("Epoch %d/%d - %.0fs - loss: %.4f - mae: %.4f - r2_square: %.4f - val_loss: %.4f - val_mae: %.4f - "
                  "val_r2_square: %.4f" % (epoch + 1, epochs, time.time() - t0, loss, mae, r2, vloss, vmae, vr2))
            if sgdr is None:
                print("current_lr=", cur_lr)
        if sgdr is not None:
            sgdr.on_epoch_end(epoch, {"val_mae": vmae})
        if vmae < best:  # ModelCheckpoint(monitor="val_mae", save_best_only=True), scann_model.py:166-177
            best, wait = vmae, 0
            if comm.rank == 0:
                model._weights = eng.get_weights()
                model.save(ckpt)
        else:
            wait += 1
            if wait >= 200:  # EarlyStopping(monitor="val_mae", patience=200), scann_model.py:179
                break
    model._weights = eng.get_weights()
    comm.rdzv.barrier()  # every rank is done (and rank 0's last checkpoint is on disk) before anybody goes on to evaluate or exit
    return hist
