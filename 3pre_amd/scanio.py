"""The reference's per-frame SIFT wire format: `SIFT_result%04d.mat`, one struct `SCAN_SIFT`.

    SIFT_extract_save.m:44-45,68-69,86-88,106   writer in the reference
    matching_sift_based.m:55,78                 `%sFeatureExtractionMatching/SIFT_result%04d.mat`, load(..., 'SCAN_SIFT')
    SIFT_match_save.m:8-14                      field list

Fields: idxScan (scalar), Image (144x176 uint8), Descriptor_RAW (128xN double), SCALE_ORIENT_POS_RAW (4xN double; rows 1:2 =
pixel u,v, 1-based after the +1 of SIFT_extract_save.m:55-56), Descriptor (128xM), SCALE_ORIENT_POS (4xM), XYZ_DATA (3xM),
M <= N = the keypoints with a valid range pixel.  MATLAB v5 MAT-files through scipy.io (host-side IO only).
"""
import os

import numpy as np
import scipy.io as sio

FIELDS = ("idxScan", "Image", "Descriptor_RAW", "SCALE_ORIENT_POS_RAW", "Descriptor", "SCALE_ORIENT_POS", "XYZ_DATA")
_SHAPES = {"Descriptor_RAW": 128, "SCALE_ORIENT_POS_RAW": 4, "Descriptor": 128, "SCALE_ORIENT_POS": 4, "XYZ_DATA": 3}


def sift_result_path(data_folder, step):
    """matching_sift_based.m:55 (DATA_FOLDER ends with a separator in the reference's config)."""
    return "%sFeatureExtractionMatching/SIFT_result%04d.mat" % (data_folder, step)


def load_sift_result(path):
    m = sio.loadmat(path, squeeze_me=False, struct_as_record=False)
    if "SCAN_SIFT" not in m:
        raise KeyError("%s holds no SCAN_SIFT struct" % path)
    s = m["SCAN_SIFT"][0, 0]
    out = {}
    for k in FIELDS:
        if not hasattr(s, k):
            continue
        v = np.asarray(getattr(s, k))
        if k in _SHAPES:
            v = np.asarray(v, dtype=np.float64).reshape(_SHAPES[k], -1) if v.size else np.zeros((_SHAPES[k], 0))
        elif k == "idxScan":
            v = int(v.reshape(-1)[0]) if v.size else 0
        out[k] = v
    for k in ("Descriptor_RAW", "SCALE_ORIENT_POS_RAW"):
        if k not in out:
            raise KeyError("%s: SCAN_SIFT.%s is missing" % (path, k))
    if out["Descriptor_RAW"].shape[1] != out["SCALE_ORIENT_POS_RAW"].shape[1]:
        raise ValueError("%s: Descriptor_RAW and SCALE_ORIENT_POS_RAW disagree on the keypoint count" % path)
    return out


def save_sift_result(path, scan):
    s = {}
    for k in FIELDS:
        if k not in scan:
            continue
        v = scan[k]
        if k in _SHAPES:
            v = np.asarray(v, dtype=np.float64).reshape(_SHAPES[k], -1)
        elif k == "idxScan":
            v = float(v)
        s[k] = v
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    sio.savemat(path, {"SCAN_SIFT": s}, format="5", do_compression=False, oned_as="column")
