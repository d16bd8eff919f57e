"""Adversarial random batches in the reference's on-disk object format (shared by the manual fuzz scripts)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "oracle")]
import scann_oracle as so


def random_batch(rng, g_update, big=True, max_struct=8):
    n = int(rng.integers(1, max_struct + 1))
    de, dn = np.empty(n, dtype=object), np.empty(n, dtype=object)
    for s in range(n):
        mode = int(rng.choice(6, p=[.2, .25, .2, .2, .05, .1]))
        if not big and mode == 4:
            mode = 3
        A = int([rng.integers(2, 5), rng.integers(3, 30), rng.integers(3, 30), rng.integers(60, 110), rng.integers(150, 230), rng.integers(2, 40)][mode])
        nb = []
        for a in range(A):
            kind = rng.integers(0, 8)
            if kind == 0:
                d = 0
            elif kind == 1:
                d = 1
            elif kind == 2 and A > 66:
                d = int(rng.integers(60, min(A - 1, 70) + 1))
            elif kind == 3 and A > 130:
                d = int(rng.integers(100, min(A - 1, 200) + 1))
            else:
                d = int(rng.integers(1, min(12, A - 1) + 1))
            d = min(d, A - 1)
            js = rng.choice(np.delete(np.arange(A), a), d, replace=False)
            ang = rng.uniform(0.4, 3.5, size=max(d, 1))
            nb.append([[6, int(j), float(ang[k]), float(ang[k] / ang.max()), float(rng.uniform(0.9, 4.0))] for k, j in enumerate(js)])
        if A >= 2 and all(len(x) == 0 for x in nb):
            nb[0] = [[6, 1, 1.0, 1.0, 1.5]]
        de[s] = [[int(z) for z in rng.choice([1, 6, 7, 8, 9], A)], float(rng.normal())]
        dn[s] = nb
    return so.pad_batch(de, dn, g_update)

