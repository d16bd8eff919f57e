#!/usr/bin/env python3
"""CLI with the reference's interface (train.py:62-107): positional ``target`` and ``dataset`` (yaml path), flags
--use_ring --use_ref --use_drop --feature --pretrained --mode.  Data-parallel training: launch under
``python -m torch.distributed.run --nproc-per-node N train.py ...`` (one process per GPU, RCCL gradient all-reduce)."""
import argparse
import os
import random
import sys
import time

import numpy as np
import yaml

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "scann--material_amd"))
from scann.models import SCANN  # noqa: E402


def set_seed(seed=2134):
    random.seed(seed)
    np.random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)


def main(args):
    set_seed(0)
    config = yaml.safe_load(open(args.dataset))
    print("Create model use Ring Information: ", args.use_ring, "\n")
    config["model"]["feature"] = args.feature
    config["model"]["use_ring"] = args.use_ring
    config["model"]["use_drop"] = args.use_drop
    config["hyper"]["use_ref"] = args.use_ref
    config["hyper"]["target"] = args.target
    config["hyper"]["pretrained"] = args.pretrained
    scann = SCANN(config, args.pretrained)
    print("Load data for dataset: ", args.dataset, " with target: ", args.target, "\n")
    scann.prepare_dataset()
    if args.mode == "train":
        print("Start Model training", "\n")
        start = time.time()
        scann.train(args.epochs)
        print("Training time: ", time.time() - start, "\n")
    print("Start Model evaluation:")
    scann.evaluate()


if __name__ == "__main__":
    p = argparse.ArgumentParser(description="Train / evaluate SCANN on the MI355X HIP path")
    p.add_argument("target", type=str, help="Target energy for training")
    p.add_argument("dataset", type=str, help="Path to dataset configs")
    # type=bool like the reference: any non-empty string is True (train.py:69-88)
    p.add_argument("--use_ring", type=bool, default=False)
    p.add_argument("--use_ref", type=bool, default=False)
    p.add_argument("--use_drop", type=bool, default=False)
    p.add_argument("--feature", type=str, default="atomic")
    p.add_argument("--pretrained", type=str, default="")
    p.add_argument("--mode", type=str, default="train")
    p.add_argument("--epochs", type=int, default=1000, help="(extension) the reference hard-codes 1000 (train.py:53)")
    main(p.parse_args())
