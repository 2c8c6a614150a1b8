"""Shared helpers of the test-suite (fixture loading, synthetic problems)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def triu_unpack(t, n):
    P = np.zeros((n, n))
    iu = np.triu_indices(n)
    P[iu] = t
    return P + P.T - np.diag(np.diag(P))


def load_sr4000():
    g = np.load(os.path.join(GOLDEN, "sr4000_step3.npz"))
    d = {k: g[k] for k in g.files}
    n = d["x_k_km1"].shape[0]
    d["n"] = n
    d["N"] = d["h"].shape[0]
    d["p_k_km1"] = triu_unpack(d["p_k_km1_triu"], n)
    d["p_k_k"] = triu_unpack(d["p_k_k_triu"], n)
    d["std_z"] = float(d["std_z"])
    d["ic_idx"] = np.nonzero(d["individually_compatible"])[0].astype(np.int32)
    d["meas_idx"] = np.nonzero(d["has_z"])[0].astype(np.int32)
    d["li_idx"] = np.nonzero(d["low_innovation_inlier"])[0].astype(np.int32)
    d["hi_idx"] = np.nonzero(d["high_innovation_inlier"])[0].astype(np.int32)
    return d


def rel_err(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
