#!/usr/bin/env python3
"""Regenerate the golden fixtures under tests/golden/ from the reference's own DATA files.

Run in a container that has /root/reference (it does not exist on the GPU box; the
outputs of this script are committed, the inputs are not):

    python tests/golden/make_fixtures.py

Only data is read -- no reference source text is copied or executed:

* matlab_code/RANSAC_SR4000_result.mat  (a MATLAB `save` of the workspace at step 3 of a real
  SR4000 run; variable `snapshot3` written by mono_slam.m:252-254, plus `cam`)
    -> sr4000_step3.npz : x/P before and after the LI+HI updates, per-landmark h/H/S/z and the
       IC/LI/HI flags, the 128-D descriptors and the camera intrinsics.
* matlab_code/sift/data/box.sift, circle.sift (Lowe ASCII keypoint files used by
  sift/sift_demo3.m) -> sift_box.npz, sift_circle.npz : descriptors as uint8 (K x 128) and
  frames (K x 4, as in the file: row col scale orientation).

siftmatch known-answer values (siftmatch_kat.json) were obtained in the survey session by
running the reference's sift/siftmatch.c on box.sift; they are recorded
there as numbers and are copied here as data.  They cannot be regenerated in this image
(siftmatch.c needs MATLAB's mex.h, which the image lacks, and stand-in headers are not allowed).
"""
import json
import os
import sys

import numpy as np
import scipy.io as sio

REF = "/root/reference/matlab_code"
OUT = os.path.dirname(os.path.abspath(__file__))


def triu_pack(P):
    iu = np.triu_indices(P.shape[0])
    return P[iu]


def make_snapshot_sub(K=24):
    """snapshot3 of the reference's RANSAC_SR4000_result.mat cut down to its first K landmarks, re-saved as a MATLAB v5 file
    with scipy's writer directly (NOT through 3pre_amd/snapshot.py, so the reader test is independent of our writer)."""
    m = sio.loadmat(os.path.join(REF, "RANSAC_SR4000_result.mat"), struct_as_record=True, squeeze_me=False)
    snap = m["snapshot3"][0, 0]
    fi = snap["features_info"][:, :K].copy()
    n = 13 + 6 * K
    for i in range(K):
        fi[0, i]["H"] = fi[0, i]["H"][:, :n]
    flt = snap["filter"][0, 0]
    out = {}
    for k in flt.dtype.names:
        v = flt[k]
        if k in ("x_k_k", "x_k_km1"):
            v = v[:n]
        elif k in ("p_k_k", "p_k_km1"):
            v = v[:n, :n]
        out[k] = v
    sio.savemat(os.path.join(OUT, "snapshot3_sub.mat"), {"snapshot3": {"features_info": fi, "filter": out, "step": snap["step"]}},
                format="5", do_compression=True, long_field_names=True)
    print("snapshot3_sub.mat", os.path.getsize(os.path.join(OUT, "snapshot3_sub.mat")), "bytes")


def make_sr4000():
    m = sio.loadmat(os.path.join(REF, "RANSAC_SR4000_result.mat"), struct_as_record=False, squeeze_me=True)
    cam = m["cam"]
    snap = m["snapshot3"]
    f = snap.filter
    fi = snap.features_info
    N = fi.shape[0]
    n = f.x_k_km1.shape[0]
    assert all(a.type == "inversedepth" for a in fi)
    assert n == 13 + 6 * N

    h = np.zeros((N, 2))
    Hcam = np.zeros((N, 2, 7))
    Hlm = np.zeros((N, 2, 6))
    S = np.zeros((N, 2, 2))
    R = np.zeros((N, 2, 2))
    z = np.full((N, 2), np.nan)
    has_z = np.zeros(N, np.uint8)
    ic = np.zeros(N, np.uint8)
    li = np.zeros(N, np.uint8)
    hi = np.zeros(N, np.uint8)
    desc = np.zeros((N, 128))
    for i, a in enumerate(fi):
        h[i] = a.h
        Hd = a.H.toarray()
        # the reference's H_i has non-zeros only in the 7 pose columns and the landmark's 6 columns
        mask = np.ones(n, bool)
        mask[0:7] = False
        mask[13 + 6 * i:19 + 6 * i] = False
        assert not Hd[:, mask].any()
        Hcam[i] = Hd[:, 0:7]
        Hlm[i] = Hd[:, 13 + 6 * i:19 + 6 * i]
        S[i] = a.S
        R[i] = a.R
        if a.z.size:
            z[i] = a.z
            has_z[i] = 1
        ic[i] = a.individually_compatible
        li[i] = a.low_innovation_inlier
        hi[i] = a.high_innovation_inlier
        desc[i] = a.Descriptor

    Pm, Pp = f.p_k_km1, f.p_k_k
    # asymmetry of the stored matrices is < 1e-23 (P scale 6e-4); keep the upper triangle
    asym = max(np.abs(Pm - Pm.T).max(), np.abs(Pp - Pp.T).max())
    assert asym < 1e-22, asym
    np.savez_compressed(
        os.path.join(OUT, "sr4000_step3.npz"),
        cam=np.array([cam.f, cam.Cx, cam.Cy, cam.k1, cam.k2, cam.nRows, cam.nCols], float),
        cam_fields=np.array(["f", "Cx", "Cy", "k1", "k2", "nRows", "nCols"]),
        std_z=float(f.std_z), step=int(snap.step),
        x_k_km1=f.x_k_km1, p_k_km1_triu=triu_pack(Pm),
        x_k_k=f.x_k_k, p_k_k_triu=triu_pack(Pp),
        h=h, Hcam=Hcam, Hlm=Hlm, S=S, R=R, z=z, has_z=has_z,
        individually_compatible=ic, low_innovation_inlier=li, high_innovation_inlier=hi,
        descriptor=desc,
    )
    print("sr4000_step3.npz  N=%d n=%d IC=%d LI=%d HI=%d asym=%.2e" % (N, n, ic.sum(), li.sum(), hi.sum(), asym))


def read_lowe_sift(path):
    with open(path) as fh:
        tok = fh.read().split()
    K, D = int(tok[0]), int(tok[1])
    vals = np.array(tok[2:], float).reshape(K, 4 + D)
    frames = vals[:, :4]
    d = vals[:, 4:]
    assert (d == np.round(d)).all() and d.min() >= 0 and d.max() <= 255
    return frames, d.astype(np.uint8)


def make_sift():
    for name in ("box", "circle"):
        fr, d = read_lowe_sift(os.path.join(REF, "sift", "data", name + ".sift"))
        np.savez_compressed(os.path.join(OUT, "sift_%s.npz" % name), frames=fr, descriptors=d)
        print("sift_%s.npz" % name, d.shape)


def make_siftmatch_kat():
    # outputs of the reference's own sift/siftmatch.c on box.sift.
    # Ranges are 0-based half-open column ranges of the 638 descriptors; pairs are the 1-based
    # (k1,k2) the MEX returns; checksum = sum_{i=1..M} i*(1000*k1_i + k2_i) in output order.
    kat = {
        "source": "reference sift/siftmatch.c run on sift/data/box.sift (SURVEY.md sec. 10); identical for double and uint8 classes",
        "cases": [
            {"L1": [0, 638], "L2": [0, 638], "thresh": 1.5, "M": 638, "checksum": 86855087319,
             "sum_best_d2": 0, "first": [1, 1], "last": [638, 638]},
            {"L1": [0, 400], "L2": [200, 638], "thresh": 1.5, "M": 207, "checksum": 7134569312,
             "sum_best_d2": 347527, "first": [59, 347], "last": [400, 200]},
            {"L1": [0, 200], "L2": [200, 638], "thresh": 1.2, "M": 36, "checksum": 91333558,
             "sum_best_d2": 2901211, "first": [4, 306], "last": [190, 328]},
        ],
        "knn_docstring_example": {
            "source": "kNearestNeighbors.m:13-27 (usage example in the function header)",
            "data": [[1, 1], [2, 2], [3, 2], [4, 4], [5, 6]],
            "query": [[1, 1], [2, 1], [6, 2]],
            "k": 2,
            "neighbors": [[1, 2], [1, 2], [4, 3]],
            "distances": [[0, 1.4142], [1.0, 1.0], [2.8284, 3.0]],
            "distances_decimals": 4,
        },
    }
    with open(os.path.join(OUT, "siftmatch_kat.json"), "w") as fh:
        json.dump(kat, fh, indent=1)
    print("siftmatch_kat.json")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; fixtures can only be regenerated where /root/reference exists")
    make_snapshot_sub()
    make_sr4000()
    make_sift()
    make_siftmatch_kat()
