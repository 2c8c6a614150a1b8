"""Checkpoint files of the reference: `snapshot%d.mat`, one struct `snapshot%d` = {features_info, filter, step}.

    mono_slam.m:251-262   writer:  snapshotK.features_info / .filter / .step, save([DATA_FOLDER 'DataSnapshots/snapshotK'])
    mono_slam.m:266-270   reader
    @ekf_filter/ekf_filter.m:27-52                       the 20 fields of the filter object
    add_feature_to_info_vector_my_version_sift.m         the 24 fields of a features_info entry

Host-side IO through scipy.io (MATLAB v5 MAT-files).  `filter` is an old-style MATLAB object in the reference; MAT readers
outside MATLAB see it as a struct with the same fields, and that is also what `save_snapshot` writes -- in MATLAB,
`ekf_filter(s.x_k_k, s.p_k_k, s.std_a, s.std_alpha, s.std_z, s.type)` (ekf_filter.m:61-86) turns it back into an object.
The numerical work never happens here: `filter_from_snapshot` uploads the state to the device-resident EkfFilter.
"""
import re

import numpy as np
import scipy.io as sio
import scipy.sparse as sp

from . import _lib

FILTER_FIELDS = ("type", "x_k_k", "p_k_k", "std_a", "std_alpha", "std_z", "x_k_km1", "p_k_km1", "predicted_measurements",
                 "H_predicted", "R_predicted", "S_predicted", "S_matching", "z", "h", "H_matching", "measurements", "R_matching",
                 "x_k_k_mixing_estimate", "p_k_k_mixing_covariance")
FEATURE_FIELDS = ("r_wc_when_initialized", "R_wc_when_initialized", "uv_when_initialized", "half_patch_size_when_initialized",
                  "half_patch_size_when_matching", "times_predicted", "times_measured", "init_frame", "init_measurement", "type",
                  "yi", "individually_compatible", "low_innovation_inlier", "high_innovation_inlier", "z", "h", "H", "S",
                  "state_size", "measurement_size", "R", "Feature3d_in_code_coordinate", "Descriptor", "last_visible")
_COLUMN = ("x_k_k", "x_k_km1", "yi", "z", "Descriptor", "uv_when_initialized", "init_measurement", "r_wc_when_initialized")
_ROW = ("h", "Feature3d_in_code_coordinate")


def snapshot_path(data_folder, step):
    """mono_slam.m:256-262."""
    return "%sDataSnapshots/snapshot%d.mat" % (data_folder, step)


def _plain(v):
    if sp.issparse(v):
        return v
    if isinstance(v, np.ndarray) and v.dtype.kind in "US" and v.ndim == 0:
        return str(v)
    return v


def _struct_to_dict(s, fields):
    return {k: _plain(getattr(s, k)) for k in fields if hasattr(s, k)}


def load_snapshot(path, step=None):
    """-> dict(step, features_info=[dict, ...], filter=dict).  Values keep MATLAB's meaning: empty arrays stay empty
    (a landmark without z has z.size == 0), H is a scipy sparse matrix as MATLAB stores it (2 x n)."""
    m = sio.loadmat(path, squeeze_me=True, struct_as_record=False)
    names = [k for k in m if re.fullmatch(r"snapshot\d+", k)]
    if step is not None:
        names = [k for k in names if k == "snapshot%d" % step]
    if not names:
        raise KeyError("%s holds no snapshot%s struct" % (path, "" if step is None else str(step)))
    s = m[sorted(names, key=lambda k: int(k[8:]))[0]]
    fi = np.atleast_1d(s.features_info) if np.size(s.features_info) else []
    return dict(step=int(s.step), features_info=[_struct_to_dict(a, FEATURE_FIELDS) for a in fi],
                filter=_struct_to_dict(s.filter, FILTER_FIELDS))


def _shape_for_matlab(k, v):
    if sp.issparse(v):
        return v.tocsc()
    if isinstance(v, str):
        return v
    a = np.asarray(v)
    if a.size == 0:
        return np.zeros((0, 0))
    if a.dtype.kind in "iub":
        a = a.astype(np.float64)                      # MATLAB doubles throughout, as in the reference's own files
    if k in _COLUMN:
        return a.reshape(-1, 1)
    if k in _ROW:
        return a.reshape(1, -1)
    return a


def save_snapshot(path, snap, compress=True):
    fi = snap["features_info"]
    arr = np.zeros((1, len(fi)), dtype=[(k, object) for k in FEATURE_FIELDS])
    for i, a in enumerate(fi):
        for k in FEATURE_FIELDS:
            arr[0, i][k] = _shape_for_matlab(k, a.get(k, np.zeros((0, 0))))
    flt = {k: _shape_for_matlab(k, snap["filter"].get(k, np.zeros((0, 0)))) for k in FILTER_FIELDS}
    name = "snapshot%d" % int(snap["step"])
    sio.savemat(path, {name: {"features_info": arr, "filter": flt, "step": float(snap["step"])}}, format="5",
                do_compression=compress, oned_as="column", long_field_names=True)


def map_types(features_info):
    """features_info(i).type -> the ABI's landmark codes (map order = state order)."""
    t = []
    for a in features_info:
        if a["type"] not in ("inversedepth", "cartesian"):
            raise ValueError("unknown landmark type %r" % (a["type"],))
        t.append(_lib.INVDEPTH if a["type"] == "inversedepth" else _lib.CARTESIAN)
    return np.asarray(t, np.int32)


def filter_from_snapshot(snap, cam, which="k_k", dtype="f32", device=0, max_hyp=1000, max_landmarks=None):
    """Device-resident filter holding the snapshot's (x_k_k, p_k_k) or (x_k_km1, p_k_km1), descriptors included."""
    from .ekf import EkfFilter
    types = map_types(snap["features_info"])
    f = EkfFilter(cam, types, dtype=dtype, device=device, max_hyp=max_hyp, max_landmarks=max_landmarks,
                  std_z=float(np.asarray(snap["filter"].get("std_z", 1.0)).reshape(-1)[0]) if np.size(snap["filter"].get("std_z", 1.0)) else 1.0)
    x, P = snap["filter"]["x_" + which], snap["filter"]["p_" + which]
    if np.size(x) != f.n or np.shape(P) != (f.n, f.n):
        f.close()
        raise ValueError("snapshot state size %s does not match its features_info (n=%d)" % (np.shape(x), f.n))
    (f.set_x_p_k_k if which == "k_k" else f.set_x_p_k_km1)(np.asarray(x, float).reshape(-1), np.asarray(P, float))
    desc = [np.asarray(a.get("Descriptor", []), float).reshape(-1) for a in snap["features_info"]]
    if desc and all(d.size == 128 for d in desc):
        f.set_descriptors(np.stack(desc, 1))
    return f


def update_snapshot_from_filter(snap, f, step=None):
    """The writer side of mono_slam.m:251-254: x_k_k / p_k_k (and the landmark types) from the device."""
    out = dict(step=int(snap["step"] if step is None else step), features_info=[dict(a) for a in snap["features_info"]],
               filter=dict(snap["filter"]))
    if len(out["features_info"]) != f.N:
        raise ValueError("features_info has %d entries, the device map %d" % (len(out["features_info"]), f.N))
    out["filter"]["x_k_k"], out["filter"]["p_k_k"] = f.get_x_k_k(), f.get_p_k_k()
    for a, t in zip(out["features_info"], f.lm_type):
        a["type"] = "inversedepth" if t == _lib.INVDEPTH else "cartesian"
    return out
