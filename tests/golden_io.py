"""Reader for tests/golden/*.npz (written by tests/golden/make_golden.py)."""
import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(HERE, name), allow_pickle=False)
    n = int(z["count"][0])
    rows = [dict() for _ in range(n)]
    for k in z.files:
        if k == "count":
            continue
        i, key = int(k[:3]), k[4:]
        v = z[k]
        rows[i][key] = v.item() if v.dtype.kind in "US" and v.shape == () else v
    return rows
