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
    ga_scores, struct_energy, y = [], [], []
    data = scann.dataIter
    for i in range(len(data)):
        inputs, t = data[i]
        energy, attn_global = scann.predict_data(inputs)
        ga_scores.extend(attn_global)
        struct_energy.extend(list(np.squeeze(energy, -1)))
        y.extend(list(t))
        if i % 10 == 0:
            print((i + 1) * data.batch_size)
    print(r2_score(struct_energy, y), mean_absolute_error(struct_energy, y))
    print("Save prediction and GA score")
    pickle.dump(ga_scores, open(os.path.join(args.trained_model, "ga_scores_{}.pickle".format(target)), "wb"))
    pickle.dump([y, struct_energy], open(os.path.join(args.trained_model, "energy_pre_{}.pickle".format(target)), "wb"))


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("trained_model", type=str, help="Target trained model path for loading")
    main(p.parse_args())
