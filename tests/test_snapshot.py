"""CPU: the snapshot%d.mat checkpoint format (mono_slam.m:251-270).  The golden file is the reference's own
snapshot3 (RANSAC_SR4000_result.mat) cut to its first 24 landmarks by tests/golden/make_fixtures.py with scipy's writer;
the expected numbers come from the independently extracted sr4000_step3.npz."""
import importlib
import os

import numpy as np
import pytest
import scipy.sparse as sp

from util import GOLDEN

snapshot = importlib.import_module("3pre_amd.snapshot")
SUB = os.path.join(GOLDEN, "snapshot3_sub.mat")
K = 24


def _same(u, v):
    if sp.issparse(u) or sp.issparse(v):
        return sp.issparse(u) and sp.issparse(v) and u.shape == v.shape and (u != v).nnz == 0
    if isinstance(u, str) or isinstance(v, str):
        return u == v
    return np.array_equal(np.asarray(u, float).reshape(-1), np.asarray(v, float).reshape(-1))


def test_reader_against_the_npz_extraction(sr4000):
    s = snapshot.load_snapshot(SUB)
    g, n = sr4000, 13 + 6 * K
    assert s["step"] == 3 and len(s["features_info"]) == K
    assert tuple(s["filter"].keys()) == snapshot.FILTER_FIELDS and tuple(s["features_info"][0].keys()) == snapshot.FEATURE_FIELDS
    flt = s["filter"]
    assert flt["type"] == "constant_velocity" and float(flt["std_z"]) == g["std_z"]
    assert np.array_equal(flt["x_k_km1"], g["x_k_km1"][:n]) and np.array_equal(flt["x_k_k"], g["x_k_k"][:n])
    assert np.array_equal(np.triu(flt["p_k_km1"]), np.triu(g["p_k_km1"][:n, :n])) and np.array_equal(np.triu(flt["p_k_k"]), np.triu(g["p_k_k"][:n, :n]))
    for i, a in enumerate(s["features_info"]):
        assert a["type"] == "inversedepth" and a["H"].shape == (2, n) and sp.issparse(a["H"])
        assert np.array_equal(a["h"], g["h"][i]) and np.array_equal(a["S"], g["S"][i])
        Hd = a["H"].toarray()
        assert np.array_equal(Hd[:, 0:7], g["Hcam"][i]) and np.array_equal(Hd[:, 13 + 6 * i:19 + 6 * i], g["Hlm"][i])
        assert np.array_equal(a["Descriptor"], g["descriptors"][i]) if "descriptors" in g else a["Descriptor"].shape == (128,)
        assert (np.size(a["z"]) == 2) == bool(g["has_z"][i])
        assert int(a["low_innovation_inlier"]) == int(g["low_innovation_inlier"][i])
    assert np.array_equal(snapshot.map_types(s["features_info"]), np.zeros(K, np.int32))


def test_writer_round_trip(tmp_path):
    s = snapshot.load_snapshot(SUB)
    s["features_info"][3]["type"] = "cartesian"
    p = str(tmp_path / "snapshot3.mat")
    snapshot.save_snapshot(p, s)
    b = snapshot.load_snapshot(p, step=3)
    assert b["step"] == 3 and len(b["features_info"]) == K
    assert all(_same(s["filter"][k], b["filter"][k]) for k in snapshot.FILTER_FIELDS)
    assert all(_same(x[k], y[k]) for x, y in zip(s["features_info"], b["features_info"]) for k in snapshot.FEATURE_FIELDS)
    assert snapshot.map_types(b["features_info"])[3] == 1
    with pytest.raises(KeyError):
        snapshot.load_snapshot(p, step=4)
    assert snapshot.snapshot_path("/data/", 7) == "/data/DataSnapshots/snapshot7.mat"


@pytest.mark.skipif(not os.path.exists("/root/reference/matlab_code/RANSAC_SR4000_result.mat"), reason="reference tree not present")
def test_reader_on_the_reference_file(sr4000):
    s = snapshot.load_snapshot("/root/reference/matlab_code/RANSAC_SR4000_result.mat")
    assert s["step"] == 3 and len(s["features_info"]) == sr4000["N"] == 185
    assert np.array_equal(s["filter"]["x_k_k"], sr4000["x_k_k"]) and s["filter"]["p_k_k"].shape == (1123, 1123)
