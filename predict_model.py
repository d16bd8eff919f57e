#!/usr/bin/env python3
"""CLI with the reference's interface (predict_model.py:95-100): load ``<trained_model>/config.yaml`` and
``<trained_model>/models/model_<target>.h5`` in infer mode, predict the whole dataset, pickle GA scores and
predictions next to the model."""
import argparse
import os
import pickle
import sys

import numpy as np
import yaml
from sklearn.metrics import mean_absolute_error, r2_score

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "scann--material_amd"))
from scann.models import SCANN  # noqa: E402


def main(args):
    config = yaml.safe_load(open(os.path.join(args.trained_model, "config.yaml")))
    target = config["hyper"]["target"]
    print("Load pretrained weight for target ", target)
    scann = SCANN(config, os.path.join(args.trained_model, "models", "model_{}.h5".format(target)), mode="infer")
    print("Load data for trained model: ", config["hyper"]["data_energy_path"])
    scann.prepare_dataset(split=False)
    # the reference loops predict_data over the batches (predict_model.py:49-62); the same per-batch results come out of the
    # pipelined dataset path (batches fused per launch sequence, uploads / launches / downloads overlapped over the streams)
    data = scann.dataIter
    yp, ga, yt = scann.model.predict_dataset(data, want_ga=True)
    struct_energy = list(np.asarray(yp) * scann.std + scann.mean)   # predict_data's de-normalisation (scann_model.py:315-319)
    y = list(yt)
    # GA scores in the reference's shape: one [M, 1] array per structure, M = largest structure of ITS batch (zeros behind the
    # structure's own atoms: the softmax of the -1e9 mask)
    counts = [len(data.data_energy[i][0]) for i in data.indexes]
    off = np.concatenate([[0], np.cumsum(counts)])
    ga_scores = []
    for b0 in range(0, len(counts), data.batch_size):
        sel = range(b0, min(len(counts), b0 + data.batch_size))
        m = max(counts[i] for i in sel)
        for i in sel:
            a = np.zeros((m, 1), dtype=np.float32)
            a[:counts[i], 0] = ga[off[i]:off[i + 1]]
            ga_scores.append(a)
    print(len(y))
    print(r2_score(struct_energy, y), mean_absolute_error(struct_energy, y))
    print("Save prediction and GA score")
    pickle.dump(ga_scores, open(os.path.join(args.trained_model, "ga_scores_{}.pickle".format(target)), "wb"))
    pickle.dump([y, struct_energy], open(os.path.join(args.trained_model, "energy_pre_{}.pickle".format(target)), "wb"))


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("trained_model", type=str, help="Target trained model path for loading")
    main(p.parse_args())
