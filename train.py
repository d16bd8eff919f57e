#!/usr/bin/env python3
"""Training / evaluation CLI with the reference's command line (train.py:62-107): positional ``target`` and ``dataset``
(a yaml path), flags --use_ring --use_ref --use_drop --feature --pretrained --mode; the console messages of the reference are
kept because downstream scripts grep them.

Extensions: ``--epochs`` (the reference hard-codes 1000, train.py:53), ``--packed`` (flat-CSR ``PackedDataset`` iterators instead
of the nested-list ``DataIterator``), ``--gpus N`` (test-set prediction spread over N devices), ``--seed``.

Data-parallel training is one process per GPU: ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1 train.py ...`` or ``scann.parallel.spawn_ranks(["train.py", ...], N)``; every rank runs this file, rank 0 reports."""
import argparse
import os
import random
import sys
import time

import numpy as np
import yaml

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "scann--material_amd"))
from scann.models import SCANN  # noqa: E402

# command-line flag -> (config section, key): what the reference's main() copies into the yaml dict before building the model
OVERRIDES = {
    "feature": ("model", "feature"),
    "use_ring": ("model", "use_ring"),
    "use_drop": ("model", "use_drop"),
    "use_ref": ("hyper", "use_ref"),
    "target": ("hyper", "target"),
    "pretrained": ("hyper", "pretrained"),
}
# name, type, default, help -- `type=bool` like the reference: ANY non-empty string is True (train.py:69-88)
FLAGS = (
    ("use_ring", bool, False, "ring / aromatic features as an extra embedding"),
    ("use_ref", bool, False, "train on the reference-energy-corrected target"),
    ("use_drop", bool, False, "dropout on the attention weights while training"),
    ("feature", str, "atomic", "atom input: 'atomic' (embedding of Z) or 'cgcnn' (92-d element descriptors)"),
    ("pretrained", str, "", "checkpoint to start from / to evaluate (container or the reference's Keras .h5)"),
    ("mode", str, "train", "'train' (train, then evaluate the best checkpoint) or anything else (evaluate only)"),
)


def seed_everything(seed):
    """Python / NumPy generators (dataset split and shuffling, weight initialisation of the package draw from NumPy)."""
    random.seed(seed)
    np.random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)


def configured(args):
    config = yaml.safe_load(open(args.dataset))
    for flag, (section, key) in OVERRIDES.items():
        config[section][key] = getattr(args, flag)
    if args.gpus:
        config["hyper"]["gpus"] = args.gpus
    return config


def run(args):
    seed_everything(args.seed)
    chatty = int(os.environ.get("RANK", "0")) == 0
    say = print if chatty else (lambda *a, **k: None)
    config = configured(args)
    say("Create model use Ring Information: ", args.use_ring, "\n")
    scann = SCANN(config, args.pretrained)
    say("Load data for dataset: ", args.dataset, " with target: ", args.target, "\n")
    scann.prepare_dataset(packed=args.packed) if args.packed else scann.prepare_dataset()
    if args.mode == "train":
        say("Start Model training", "\n")
        t0 = time.time()
        scann.train(args.epochs)
        say("Training time: ", time.time() - t0, "\n")
    say("Start Model evaluation:")
    return scann.evaluate()  # rank 0 writes report.txt; the other ranks return at once


def parser():
    p = argparse.ArgumentParser(description="Train / evaluate SCANN on the MI355X HIP path")
    p.add_argument("target", type=str, help="Target energy for training")
    p.add_argument("dataset", type=str, help="Path to dataset configs")
    for name, kind, default, text in FLAGS:
        p.add_argument("--" + name, type=kind, default=default, help=text)
    p.add_argument("--epochs", type=int, default=1000)
    p.add_argument("--packed", action="store_true", help="PackedDataset iterators (flat CSR, native slicing)")
    p.add_argument("--gpus", type=int, default=0, help="devices for the test-set prediction of evaluate()")
    p.add_argument("--seed", type=int, default=0)
    return p


if __name__ == "__main__":
    run(parser().parse_args())
