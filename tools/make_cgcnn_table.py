#!/usr/bin/env python3
"""Writes scann--material_amd/scann/utils/data/cgcnn_atom_init.json: the 92-d CGCNN element descriptors (the public
`atom_init.json` of the CGCNN code base) that the reference keeps as a literal in scann/utils/dataset/atomic_data.py:27-531 and
looks up per atom at datagenerator.py:109-110.  The reference file is READ AS TEXT (the dict literal is parsed with
ast.literal_eval, nothing is imported or executed); every vector is binary, so the data file stores, per atomic number, the
indices of the ones.  Run in the build container only (the reference tree does not travel)."""
import ast
import json
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/scann/utils/dataset/atomic_data.py"
text = open(src).read()
start = text.index("atomic_features = ") + len("atomic_features = ")
depth, end = 0, None
for i in range(start, len(text)):
    if text[i] == "{":
        depth += 1
    elif text[i] == "}":
        depth -= 1
        if depth == 0:
            end = i + 1
            break
table = ast.literal_eval(text[start:end])
out = {}
for k, v in sorted(table.items(), key=lambda kv: int(kv[0])):
    assert len(v) == 92 and all(x in (0, 1) for x in v), k
    out[str(int(k))] = [i for i, x in enumerate(v) if x]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scann--material_amd", "scann", "utils", "data",
                   "cgcnn_atom_init.json")
json.dump({"format": "per atomic number: indices of the ones of the 92-d binary CGCNN descriptor (atom_init.json)", "dim": 92, "ones": out},
          open(dst, "w"), separators=(",", ":"))
print(len(out), "elements ->", dst, os.path.getsize(dst), "bytes")
