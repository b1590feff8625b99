"""Synthetic MMBertDataset items shared by make_golden.py (fed to the REAL reference class) and the tests (fed to the
build's class): pure numpy, no reference code."""
import numpy as np


def synthetic_features(n_items=5, L=6, seed=11, dataset="mosei"):
    """Items shaped like REF:train.py:191-195: ((input_ids, visual, speech, input_mask), label, segment, words).
    Pure data (numpy PCG64): the test rebuilds exactly these to feed the build's MMBertDataset."""
    rng = np.random.Generator(np.random.PCG64(seed))
    vd, sd = {"mosi": (47, 74), "mosei": (35, 74), "ur_funny": (371, 81)}[dataset]
    feats = []
    for k in range(n_items):
        n = 2 + int(rng.integers(0, L - 3))
        ids = [101] + [int(x) for x in rng.integers(1000, 2000, n)] + [102] + [0] * (L - n - 2)
        vis = rng.standard_normal((L, vd)); vis[n + 1:] = 0
        sp = rng.standard_normal((L, sd)); sp[n + 1:] = 0
        mask = [1] * (n + 2) + [0] * (L - n - 2)
        if dataset == "mosei":
            label = np.array([np.concatenate(([rng.uniform(-3, 3)], (rng.random(6) > 0.6) * rng.uniform(0, 3, 6)))])
        elif dataset == "mosi":
            label = np.array([rng.uniform(-3, 3)])
        else:
            label = np.array([int(rng.integers(0, 2))])
        feats.append(((ids, vis, sp, mask), label, "seg%d" % k, ["w%d" % k]))
    return feats
