"""SCANN facade on the MI355X HIP path.

Mirrors the public surface of the reference's ``scann/models/scann_model.py`` -- ``SCANN(config,
pretrained, mode)``, ``.model.predict(inputs)``, ``.prepare_dataset``, ``.evaluate``,
``.predict_data``, ``create_model`` -- with the Keras graph execution replaced by
``libscann_hip.so`` (include/scann_hip.h) through ctypes.  No TensorFlow, no PyTorch.
"""
from __future__ import annotations

import io
import json
import os

import numpy as np

from .. import _hip

INPUT_NAMES = ["atomic", "atom_mask", "neighbors", "neighbor_mask", "neighbor_weight", "neighbor_distance"]

# keys some shipped yaml files omit (model_qm9_std.yaml / model_ptgp.yaml; train.py:37-43 injects the CLI ones)
_MODEL_DEFAULTS = dict(feature="atomic", use_ring=False, use_drop=False, g_update=False, gaussian_d=4.0,
                       use_attn_norm=True, use_ga_norm=True)
_HYPER_DEFAULTS = dict(scaler=False, scheduler="cosine", use_ref=False, target="", pretrained="")


def normalize_config(config):
    """Fill the keys the reference reads but some of its yaml files lack (deliberate deviation: the
    reference raises KeyError there)."""
    config.setdefault("model", {})
    config.setdefault("hyper", {})
    for k, v in _MODEL_DEFAULTS.items():
        config["model"].setdefault(k, v)
    for k, v in _HYPER_DEFAULTS.items():
        config["hyper"].setdefault(k, v)
    return config


def config_struct(config):
    m = config["model"]
    if m["feature"] not in ("atomic", "cgcnn"):
        raise ValueError("model.feature must be 'atomic' or 'cgcnn'")
    return _hip.Config(
        n_atoms=int(m["n_atoms"]), embedding_dim=int(m["embedding_dim"]), local_dim=int(m["local_dim"]),
        num_head=int(m["num_head"]), n_attention=int(m["n_attention"]), global_dim=int(m["global_dim"]),
        dense_out=int(m["dense_out"]), n_gauss=20, gaussian_d=float(m["gaussian_d"]),
        g_update=int(bool(m["g_update"])), use_attn_norm=int(bool(m["use_attn_norm"])),
        use_ga_norm=int(bool(m["use_ga_norm"])), use_ring=int(bool(m["use_ring"])),
        feature_cgcnn=int(m["feature"] == "cgcnn"),
        relu_out=int(config["hyper"].get("target") == "e_b"),  # scann_model.py:446
    )


def keras_default_init(specs, seed=None):
    """Keras default initialisers for a fresh model (what create_model yields before training):
    Dense kernels Glorot-uniform, biases 0, Embedding U(-0.05, 0.05), LayerNorm gamma 1 / beta 0."""
    rng = np.random.default_rng(seed)
    w = {}
    for name, shape in specs:
        leaf = name.rsplit("/", 1)[1]
        if leaf == "kernel":
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            t = rng.uniform(-lim, lim, size=shape)
        elif leaf == "embeddings":
            t = rng.uniform(-0.05, 0.05, size=shape)
        elif leaf == "gamma":
            t = np.ones(shape)
        else:
            t = np.zeros(shape)
        w[name] = np.ascontiguousarray(t, dtype=np.float32)
    return w


def save_container(path, config, weights):
    """Write the weight container ATOMICALLY: the bytes go to a temporary file in the same directory, which then replaces
    ``path`` (os.replace).  A reader -- another rank loading the best checkpoint, a monitoring script -- sees the previous
    complete file or the new complete file, never a partial one."""
    buf = io.BytesIO()
    np.savez(buf, __config__=np.array(json.dumps(config)), **weights)
    folder = os.path.dirname(os.path.abspath(path))
    os.makedirs(folder, exist_ok=True)
    tmp = os.path.join(folder, ".%s.%d.tmp" % (os.path.basename(path), os.getpid()))
    try:
        with open(tmp, "wb") as f:
            f.write(buf.getvalue())
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.unlink(tmp)


class HipModel:
    """Stand-in for the ``tf.keras.Model`` that ``create_model`` returns (scann_model.py:449):
    ``predict`` runs the whole forward graph on the GPU."""

    def __init__(self, config, weights=None, device=None, infer=False, seed=None):
        self.config = normalize_config(config)
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("SCANN_DEVICE") is None \
                else int(os.environ["SCANN_DEVICE"])
        self.engine = _hip.Engine(config_struct(self.config), device)
        self.infer = infer  # True: outputs [y, global_attention scores] (scann_model.py:81-83)
        self.input_names = list(INPUT_NAMES) + (["ring_aromatic"] if self.config["model"]["use_ring"] else [])
        self.output_names = ["predict_property"] + (["global_attention"] if infer else [])
        self._weights = None
        self.set_weights(weights if weights is not None else keras_default_init(self.engine.weight_specs(), seed))

    # -- weights ---------------------------------------------------------------------------------
    def set_weights(self, weights):
        self.engine.load_weights(weights)
        self._weights = {n: np.array(weights[n], dtype=np.float32) for n, _ in self.engine.weight_specs()}

    def get_weights(self):
        return dict(self._weights)

    def count_params(self):
        return int(sum(v.size for v in self._weights.values()))

    def save(self, path):
        """Weight container: a zip (npz) of the named fp32 tensors plus the yaml config as JSON.
        Takes the place of the Keras full-model HDF5 (scann_model.py:166-177)."""
        save_container(path, self.config, self._weights)

    # -- inference ---------------------------------------------------------------------------------
    def predict(self, inputs, batch_size=None, verbose=0, **_):
        """``model.predict(inputs)`` (scann_model.py:266,316): ``[B,1]`` or, in infer mode,
        ``[[B,1], [B,M,1]]``."""
        m = self.config["model"]
        if not isinstance(inputs, _hip.PackedBatch):
            nb = np.shape(inputs["neighbors"])
            slots = int(nb[0]) * int(nb[1]) * max(1, int(nb[2]))  # padded neighbour slots >= edges; >= atoms
            if nb[0] >= self.BIG_PREDICT or (slots > self.BIG_SLOTS and nb[0] > 1):
                return self._predict_chunked(inputs)
        if not isinstance(inputs, _hip.PackedBatch) and m["feature"] == "atomic" and not m["use_ring"]:
            y, ga = self.engine.forward_padded(inputs, want_ga=self.infer)  # native CSR packing
            return [y.reshape(-1, 1), ga] if self.infer else y.reshape(-1, 1)
        packed = inputs if isinstance(inputs, _hip.PackedBatch) else _hip.pack_inputs(inputs)
        y, ga = self.engine.forward(packed, want_ga=self.infer)
        y = y.reshape(-1, 1)
        if self.infer:
            return [y, packed.repad_ga(ga)]
        return y

    __call__ = predict

    BIG_PREDICT = 1024   # structures from which `predict(padded arrays)` runs as a pipeline of chunks
    PREDICT_CHUNK = 2048  # structures per chunk: one launch sequence each (16 batches of the reference's 128)
    BIG_SLOTS = 6_000_000  # ... or padded neighbour slots (a launch sequence takes < 8,388,608 atoms / edges: few but large crystals)

    def _predict_chunked(self, inputs):
        """`model.predict(x)` on a WHOLE padded dataset (what the reference's evaluate / predict scripts do with Keras, which batches
        internally: scann_model.py:266,316): rows are cut into chunks -- views, nothing is copied -- and software-pipelined on this
        thread: chunk k + 1 is packed and uploaded (host, native code) while the device runs chunk k (launches are asynchronous),
        a rolling window of chunks stays in flight over the handle's streams, results are fetched oldest first.  One launch sequence
        for everything would leave the device idle while the host packs 45 M neighbour slots, and the host idle afterwards."""
        eng = self.engine
        B = len(inputs["atom_mask"])
        nb = np.shape(inputs["neighbors"])
        # at least four chunks (the first chunk's packing is the only host work the device waits for), at most PREDICT_CHUNK
        # structures and BIG_SLOTS * 2 / 3 padded slots each
        C = min(self.PREDICT_CHUNK, max(512, -(-B // 4 // 128) * 128))
        C = max(1, min(C, (self.BIG_SLOTS * 2 // 3) // max(1, int(nb[1]) * max(1, int(nb[2])))))
        ns = eng.num_streams()
        window = 2 * ns
        pending, ys, gas = [], [], []

        def fetch_oldest():
            rb, first = pending.pop(0)
            try:
                y, ga = eng.download(rb, want_ga=self.infer)
            except _hip.ScannHipError as e:
                rb.free()
                # (a device-packed chunk reports bad input only here, up to `window` chunks after it was uploaded: say WHICH chunk)
                raise _hip.ScannHipError(e.code, "%s [structures %d..%d of this call]" % (e.detail, first, min(first + C, B) - 1)) from e
            except BaseException:
                rb.free()
                raise
            ys.append(y)
            if self.infer:
                gas.append(ga)
            rb.release()

        # feature = "atomic" without ring: the chunk is packed to CSR on the DEVICE (the host reads its masks only); else the native host packer
        m = self.config["model"]
        device_pack = m["feature"] == "atomic" and not m["use_ring"] and not eng.training
        k = 0
        try:
            for i in range(0, B, C):
                chunk = {key: v[i:i + C] for key, v in inputs.items()}
                rb = eng.upload_padded(chunk) if device_pack else eng.upload(_hip.pack_inputs(chunk))
                try:
                    if len(pending) >= window:
                        fetch_oldest()
                    eng.forward_resident(rb, k % ns)
                except BaseException:
                    rb.free()
                    raise
                k += 1
                pending.append((rb, i))
            while pending:
                fetch_oldest()
        finally:
            for rb, _ in pending:
                rb.free()
        y = np.concatenate(ys).reshape(-1, 1)
        if not self.infer:
            return y
        amask = np.asarray(inputs["atom_mask"]).reshape(B, -1) != 0
        ga_pad = np.zeros(amask.shape, dtype=np.float32)  # softmax of -1e9 -> 0 on padded atoms
        ga_pad[amask] = np.concatenate(gas)  # packed rows are the real atoms in (structure, atom) order
        return [y, ga_pad[..., None]]

    @staticmethod
    def default_group(dataset):
        """Batches fused per launch sequence when the caller does not say: about 1,024 structures (8 batches of 128) -- two such groups
        in flight on the handle's two streams measured best (tools/e2e_size.py: 1.78 M molecules/s; 1.64 M at 4, 1.69 M at 12)."""
        bs = int(getattr(dataset, "batch_size", 0) or 0)
        return max(1, 1024 // bs) if bs > 0 else 8

    def predict_dataset(self, dataset, group=None, want_ga=False):
        """Pipelined inference over a whole ``PackedDataset`` (or any sequence of ``(PackedBatch | inputs dict, target)``):
        batches are fused ``group`` at a time into one launch sequence, spread over the handle's streams, and fetched at
        the end -- the throughput path behind ``SCANN.evaluate`` / ``predict_model.py``.  Returns ``(y [N], ga list | None,
        targets [N])`` in dataset order.

        A software pipeline on the calling thread: group k + 1 is sliced and uploaded (native calls; the upload returns when its copy
        is enqueued) right after group k's launches are enqueued, a rolling window of 2 x streams groups stays in flight, results are
        fetched oldest first.  (Rounds 3-4 used a producer thread for slicing + upload; it is still there behind
        SCANN_DATASET_THREAD=1.)"""
        import queue
        import threading

        eng = self.engine
        ns = eng.num_streams()
        pending, ys, gas, ts = [], [], [], []
        group = int(group) if group else self.default_group(dataset)

        def fetch_oldest():
            # a rolling window of `ns` groups in flight (one per stream): only the OLDEST is waited for, and its batch is released
            # without a device-wide synchronisation, so the device keeps running the younger groups
            rb = pending.pop(0)
            try:
                y, ga = eng.download(rb, want_ga=want_ga)
            except BaseException:
                rb.free()
                raise
            ys.append(y)
            if want_ga:
                gas.append(ga)
            rb.release()

        n = len(dataset)
        grouped = getattr(dataset, "batches", None)  # PackedDataset: a whole group with one native slice call

        def make(g0):
            if grouped is not None:
                pk, tgt = grouped(g0, min(n, g0 + group))
                return pk, [np.asarray(tgt, dtype=np.float32)]
            parts, tg = [], []
            for i in range(g0, min(n, g0 + group)):
                item, tgt = dataset[i]
                parts.append(item if isinstance(item, _hip.PackedBatch) else _hip.pack_inputs(item))
                tg.append(np.asarray(tgt, dtype=np.float32))
            return (_hip.concat_packed(parts) if len(parts) > 1 else parts[0]), tg

        # One thread is enough, and faster: launches and uploads are asynchronous, so slicing + uploading group k + 1 right after
        # enqueueing group k overlaps host and device by itself; the producer thread of rounds 3-4 (below, SCANN_DATASET_THREAD=1)
        # cost more in Python thread hand-offs than it hid (bench.py end_to_end: 1.70-1.72 M -> 1.81 M molecules/s, one box).
        if os.environ.get("SCANN_DATASET_THREAD", "0") != "1":
            k = 0
            try:
                for g0 in range(0, n, group):
                    pk, tg = make(g0)
                    rb = eng.upload(pk)
                    ts.extend(tg)
                    try:
                        if len(pending) >= 2 * ns:
                            fetch_oldest()
                        eng.forward_resident(rb, k % ns)
                    except BaseException:
                        rb.free()
                        raise
                    k += 1
                    pending.append(rb)
                while pending:
                    fetch_oldest()
            finally:
                for rb in pending:
                    rb.free()
            if not ys:
                return np.zeros(0, np.float32), (np.zeros(0, np.float32) if want_ga else None), np.zeros(0, np.float32)
            return np.concatenate(ys), (np.concatenate(gas) if want_ga else None), np.concatenate(ts)
        ready = queue.Queue(maxsize=max(2, ns))  # uploaded groups waiting for their launches
        stop = threading.Event()

        def put(item):
            while not stop.is_set():
                try:
                    ready.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False

        def producer():
            try:
                for g0 in range(0, n, group):
                    if stop.is_set():
                        return
                    pk, tg = make(g0)
                    rb = eng.upload(pk)
                    if not put((rb, tg)):
                        rb.free()
                        return
                put(None)
            except BaseException as e:  # noqa: BLE001 -- handed to the consumer, which re-raises it
                put(e)

        worker = threading.Thread(target=producer, name="scann-upload", daemon=True)
        worker.start()
        k = 0
        taken = None  # a group taken off the queue and not yet in flight
        try:
            while True:
                item = ready.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                taken, tg = item
                ts.extend(tg)
                if len(pending) >= ns:
                    fetch_oldest()  # frees the stream slot the new group is about to use
                eng.forward_resident(taken, k)
                k += 1
                pending.append(taken)
                taken = None
            while pending:
                fetch_oldest()
        finally:
            stop.set()
            worker.join()
            if taken is not None:
                taken.free()
            while True:  # groups uploaded but never launched (an error on either side)
                try:
                    item = ready.get_nowait()
                except queue.Empty:
                    break
                if isinstance(item, tuple):
                    item[0].free()
            for rb in pending:
                rb.free()
        if not ys:
            return np.zeros(0, np.float32), (np.zeros(0, np.float32) if want_ga else None), np.zeros(0, np.float32)
        return np.concatenate(ys), (np.concatenate(gas) if want_ga else None), np.concatenate(ts)

    def summary(self):
        print("SCANN HIP model: %d parameters, %d local-attention layers, g_update=%s" % (
            self.count_params(), self.config["model"]["n_attention"], self.config["model"]["g_update"]))


def _read_container(path, config=None):
    with open(path, "rb") as f:
        magic = f.read(8)
    if magic.startswith(b"\x89HDF"):  # a Keras full-model HDF5 checkpoint of the reference (scann_model.py:166-177)
        from .keras_import import load_keras_h5

        return load_keras_h5(path, config)
    z = np.load(path, allow_pickle=False)
    cfg = json.loads(str(z["__config__"]))
    return cfg, {k: z[k] for k in z.files if k != "__config__"}


def load_model(path, custom_objects=None, infer=False, config=None):
    """``tf.keras.models.load_model`` counterpart (scann_model.py:79,87,323): this package's weight container, or a Keras
    HDF5 checkpoint written by the reference (imported by ``keras_import``; pass the run's ``config`` for the keys the file
    does not determine)."""
    cfg, weights = _read_container(path, config)
    if config is not None:  # the caller's yaml wins for everything but the architecture (hyper.target selects the mrelu head)
        cfg["hyper"].update({k: v for k, v in config.get("hyper", {}).items() if k != "target" or "target" not in cfg["hyper"]})
    return HipModel(cfg, weights, infer=infer)


def create_model_pretrained(pretrained):
    model = load_model(pretrained)
    model.summary()
    return model


def create_model(config, seed=None):
    """The graph builder (scann_model.py:329-453): returns a model with fresh Keras-default weights."""
    model = HipModel(config, seed=seed)
    model.summary()
    return model


class SCANN:
    """API facade, same constructor and methods as the reference class (scann_model.py:42-319)."""

    def __init__(self, config=None, pretrained="", mode="train"):
        self.config = normalize_config(config)
        self.model = None
        self.mean, self.std = 0, 1
        if "target_mean" in self.config["hyper"]:
            self.mean = float(self.config["hyper"]["target_mean"])
            self.std = float(self.config["hyper"]["target_std"])
        if mode == "train" or mode == "eval":
            if pretrained:
                print("load pretrained model from ", pretrained, "\n")
                self.model = create_model_pretrained(pretrained)
                self.config["hyper"]["pretrained"] = pretrained
            else:
                self.model = create_model(self.config)
        else:
            self.model = load_model(pretrained, infer=True, config=self.config)

    @classmethod
    def load_model_infer(cls, path):
        return load_model(path, infer=True)

    @classmethod
    def load_model(cls, path):
        return create_model_pretrained(path)

    def prepare_dataset(self, split=True, packed=False):
        """Reference behaviour (scann_model.py:98-161).  ``packed=True`` (extension) builds ``PackedDataset`` iterators:
        the object arrays are converted once to flat CSR and batches become slices; they yield
        ``(PackedBatch, target)`` instead of ``(inputs dict, target)``."""
        from ..utils.datagenerator import DataIterator
        from ..utils.general import load_dataset, split_data
        from ..utils.packed_dataset import PackedDataset

        if packed:
            DataIterator = PackedDataset  # noqa: F811  (same constructor signature)

        hy, mo = self.config["hyper"], self.config["model"]
        data_energy, data_neighbor = load_dataset(
            use_ref=hy["use_ref"], use_ring=mo["use_ring"], dataset=hy["data_energy_path"],
            dataset_neighbor=hy["data_nei_path"], target_prop=hy["target"])
        if hy["scaler"]:
            target = [d[1] for d in data_energy]
            self.mean, self.std = np.mean(target, dtype="float32"), np.std(target, dtype="float32")
            print("Normalize dataset property with mean: ", self.mean, " , std: ", self.std, "\n")
            data_energy[:, 1] = (data_energy[:, 1] - self.mean) / self.std
        hy["target_mean"] = str(self.mean)
        hy["target_std"] = str(self.std)
        hy["data_size"] = len(data_energy)
        kw = dict(batch_size=hy["batch_size"], use_ring=mo["use_ring"], feature=mo["feature"], g_update=mo["g_update"],
                  atomic_features=hy.get("cgcnn_table"))  # cgcnn: path of the element table (else SCANN_CGCNN_TABLE)
        if split:
            train, valid, test, extra = split_data(len_data=len(data_energy), test_percent=hy["test_percent"],
                                                   train_size=hy["train_size"], test_size=hy["test_size"])
            assert len(extra) == 0, "Split was inexact {} {} {} {}".format(len(train), len(valid), len(test), len(extra))
            print("Number of train data : ", len(train), " , Number of valid data: ", len(valid),
                  " , Number of test data: ", len(test), "\n")
            self.trainIter, self.validIter, self.testIter = [
                DataIterator(data_neighbor=data_neighbor[idx], data_energy=data_energy[idx],
                             shuffle=(len(idx) == len(train)), **kw)
                for idx in (train, valid, test)]
            return train, valid, test
        self.dataIter = DataIterator(data_neighbor=data_neighbor, data_energy=data_energy, **kw)

    def _out_dir(self):
        return "{}_{}".format(self.config["hyper"]["save_path"], self.config["hyper"]["target"])

    def train(self, epochs=1000):
        """``compile`` + ``fit`` of the reference (scann_model.py:199-245): RMSE loss + l2 regularisers, Adam with the
        legacy decay, CosineDecay or SGDR schedule, best-val_mae checkpoint, early stopping; afterwards the model is
        dropped so that ``evaluate`` reloads the best checkpoint, exactly like the reference."""
        import yaml

        from .trainer import fit

        os.makedirs("{}/models/".format(self._out_dir()), exist_ok=True)
        if int(os.environ.get("RANK", "0")) == 0:
            yaml.safe_dump(self.config, open("{}/config.yaml".format(self._out_dir()), "w"), default_flow_style=False)

        class _Hist:
            pass

        self.hist = _Hist()
        self.hist.history = fit(self, epochs)
        del self.model

    def evaluate(self, gpus=None):
        """Test-set loop of the reference (scann_model.py:247-313): predict every batch, report
        R2 and MAE * std, write report.txt.  ``gpus`` (or ``hyper.gpus``) > 1 spreads the batches over that many
        devices of the node from this process (scann.parallel.MultiGpuPredictor; no collective)."""
        from sklearn.metrics import mean_absolute_error, r2_score

        if int(os.environ.get("WORLD_SIZE", "1")) > 1 and int(os.environ.get("RANK", "0")) != 0:
            # data-parallel job: the test set, report.txt and hist_data.npy belong to rank 0 alone (the ranks hold the same
            # model after training; trainer.fit ends with a barrier, so the best checkpoint is complete before rank 0 loads it)
            return None, None
        if not hasattr(self, "model") or self.model is None:
            print("Load best validation weight for predicting testset", "\n")
            t = self.config["hyper"]["target"]
            self.model = load_model("{}/models/model_{}.h5".format(self._out_dir(), t))
        data = self.dataIter if hasattr(self, "dataIter") else self.testIter
        runner = self.model
        n_gpu = int(gpus if gpus is not None else self.config["hyper"].get("gpus", 1))
        if n_gpu > 1:
            from ..parallel import MultiGpuPredictor, MultiProcessPredictor

            # hyper.gpu_processes: one worker PROCESS per device instead of one thread (no shared interpreter; PackedDataset only)
            if self.config["hyper"].get("gpu_processes") and hasattr(data, "batches"):
                runner = MultiProcessPredictor(self.config, self.model.get_weights(), devices=list(range(n_gpu)))
            else:
                runner = MultiGpuPredictor(self.config, self.model.get_weights(), devices=list(range(n_gpu)))
        yp, _, yt = runner.predict_dataset(data)  # same per-batch results as the reference's predict loop (:264-271)
        if hasattr(runner, "close"):
            runner.close()
        y_predict, y = list(yp), list(yt)
        mae = mean_absolute_error(y, y_predict) * self.std
        r2 = r2_score(y, y_predict)
        print("Result for testset ", self.config["hyper"]["target"], " : R2 score: ", r2, " and MAE: ", mae)
        os.makedirs(self._out_dir(), exist_ok=True)
        with open("{}/report.txt".format(self._out_dir()), "w") as f:
            if hasattr(self, "hist"):  # scann_model.py:292-311
                f.write("Training MAE: " + str(min(self.hist.history["mae"]) * self.std) + "\n")
                f.write("Val MAE: " + str(min(self.hist.history["val_mae"]) * self.std) + "\n")
            f.write("Test MAE: " + str(mae) + ", Test R2: " + str(r2))
        if hasattr(self, "hist"):
            np.save("{}/hist_data.npy".format(self._out_dir()), np.array([y_predict, y, self.hist.history], dtype=object))
            print("Saved model record for dataset")
        return mae, r2

    def predict_data(self, ip):
        out = self.model.predict(ip)
        if isinstance(out, list):  # infer mode (the reference tests len(out) == 2, scann_model.py:317)
            return out[0] * self.std + self.mean, out[1]
        return out * self.std + self.mean
