"""GPU: a reference checkpoint (snapshot%d.mat, mono_slam.m:251-270) loaded straight onto the device reproduces the
per-landmark S the reference stored in it (search_IC_matches.m:36-43), and writes back a loadable checkpoint."""
import importlib
import os

import numpy as np
import pytest

from util import GOLDEN

pytestmark = pytest.mark.gpu
snapshot = importlib.import_module("3pre_amd.snapshot")


@pytest.mark.parametrize("dtype,tol", [("f64", 1e-10), ("f32", 5e-4)])
def test_snapshot_to_device_and_back(pre3, sr4000, tmp_path, dtype, tol):
    s = snapshot.load_snapshot(os.path.join(GOLDEN, "snapshot3_sub.mat"))
    f = snapshot.filter_from_snapshot(s, sr4000["cam"], which="k_km1", dtype=dtype, max_hyp=8)
    assert f.N == 24 and f.n == 157 and f.std_z == 2.0
    f.search_IC_matches()
    fld = f.landmark_fields()
    S_ref = np.stack([a["S"] for a in s["features_info"]])
    assert fld["has_h"].all() and np.abs(fld["S"] - S_ref).max() < tol
    assert np.array_equal(f.get_descriptors(), np.stack([a["Descriptor"] for a in s["features_info"]], 1))
    f.close()
    # posterior side: upload (x_k_k, p_k_k), write a checkpoint from the device state, read it back
    f = snapshot.filter_from_snapshot(s, sr4000["cam"], which="k_k", dtype="f64", max_hyp=8)
    out = snapshot.update_snapshot_from_filter(s, f, step=4)
    p = str(tmp_path / "snapshot4.mat")
    snapshot.save_snapshot(p, out)
    b = snapshot.load_snapshot(p)
    assert b["step"] == 4 and np.array_equal(b["filter"]["x_k_k"], s["filter"]["x_k_k"])
    assert np.array_equal(b["filter"]["p_k_k"], np.asarray(s["filter"]["p_k_k"]))
    f.close()
