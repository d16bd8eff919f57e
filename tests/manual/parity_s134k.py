#!/usr/bin/env python3
"""BASELINE config 2 at full size: 130,831 synthetic QM9-shaped molecules (batch 128) through the HIP path and through
the C/OpenMP restatement of the reference graph (the checker), Keras-default random weights.  Reports the max relative
error of the predictions, of the GA scores, and the "HOMO MAE" the reference's evaluate() would print
(scann_model.py:273-280) for both against the same synthetic targets.  ~2-3 minutes, dominated by the CPU side."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), os.path.join(ROOT, "oracle"), ROOT]
import bench
os.environ.setdefault("OMP_NUM_THREADS", str(bench.host_cores()))
import scann_oracle as so
import scann_oracle_c as soc
from scann import _hip
from scann.models.scann_model import HipModel, normalize_config

N = int(sys.argv[1]) if len(sys.argv) > 1 else 130831
cfg = normalize_config(so.default_config("qm9"))
w = so.init_weights(cfg, 1234)
model = HipModel(cfg, w, device=0, infer=True)
rng = np.random.default_rng(0)
B = 128
y_gpu, y_cpu, tgt = [], [], []
worst_y = worst_ga = 0.0
t_gpu = t_cpu = 0.0
done = 0
while done < N:
    n = min(B, N - done)
    de, dn = so.synth_dataset(n, seed=1000 + done)
    inputs, t = so.pad_batch(de, dn, True)
    t0 = time.perf_counter(); yg, gg = model.predict(inputs); t_gpu += time.perf_counter() - t0
    t0 = time.perf_counter(); yc, gc = soc.forward(cfg, w, inputs); t_cpu += time.perf_counter() - t0
    y_gpu.append(yg[:, 0]); y_cpu.append(yc[:, 0]); tgt.append(t)
    worst_ga = max(worst_ga, float(np.abs(gg - gc).max()))
    done += n
    if (done // B) % 100 == 0:
        print("%d / %d" % (done, N), flush=True)
y_gpu, y_cpu, tgt = map(np.concatenate, (y_gpu, y_cpu, tgt))
scale = float(np.sqrt(np.mean(y_cpu.astype(np.float64) ** 2)))
rel = np.abs(y_gpu - y_cpu) / np.maximum(np.abs(y_cpu), scale)
mae_gpu, mae_cpu = float(np.mean(np.abs(y_gpu - tgt))), float(np.mean(np.abs(y_cpu - tgt)))
print("molecules %d | max rel err y (gpu vs CPU restatement) %.3e | strict max |dy|/|y| %.3e | max abs GA err %.3e"
      % (N, rel.max(), float(np.max(np.abs(y_gpu - y_cpu) / np.maximum(np.abs(y_cpu), 1e-6))), worst_ga))
print("MAE vs synthetic targets: gpu %.7f  cpu %.7f  relative difference %.3e" % (mae_gpu, mae_cpu, abs(mae_gpu - mae_cpu) / mae_cpu))
print("predict() wall %.1f s (%.0f mol/s, PCIe + Python inclusive) | CPU restatement %.1f s (%.0f mol/s, %s threads)"
      % (t_gpu, N / t_gpu, t_cpu, N / t_cpu, os.environ["OMP_NUM_THREADS"]))
