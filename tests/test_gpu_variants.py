"""The launch structure of pre3_step has switches (what rides in which launch, which tails are launches of their own,
the speculative first panel).  They change WHERE work runs, never what is computed: every variant must give bit-identical states,
flags and statistics."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (round 5: the launch structures below are those of the step WITHOUT the in-launch tail, PRE3_TAIL=0 -- with it P_LI is never rounded to fp32, so
#  it agrees with them to rounding, not to the bit: test_the_step_tail_agrees_with_the_launch_per_stage_form)
VARIANTS = [
    {},
    {"PRE3_SELECT_FUSE": "1", "PRE3_HI_FUSE": "1"},
    {"PRE3_RIDE_INNOV": "0", "PRE3_RIDE_PROJ": "0", "PRE3_RIDE_RESCUE": "0"},
    {"PRE3_CHOL_SPEC0": "0"},
    {"PRE3_FUSE_JN": "0", "PRE3_FUSE_ROWS": "0"},           # the Jnorm pass / the row build as launches of their own
    {"PRE3_HI_FUSED": "0"},                                 # the rescue stage's collection + HI update as four launches sized by a host poll instead of k_hi_fused
    {"PRE3_DD_MAX": "3"},                                   # only three tile groups down-dated inside the persistent launch, the rest (and the rescue's projection) in the launch behind it
    {"PRE3_K9_OVERLAP": "0"},                               # the down-date as a launch behind the persistent factorisation instead of consumers inside it
    {"PRE3_INLINE_G": "0"},                                 # H*P*H' of all measured rows built by a launch of its own instead of entry by entry in the scorer / the LI gather
    {"PRE3_SELECT_GATHER": "0"},
    {"PRE3_SELECT_GATHER_LDS": "0"},
    {"PRE3_MAP_ONE_PASS": "0"},                             # map management's congruence as two gather passes (T = A P, P = T A') instead of one                        # k_select_gather reading H*P from global memory instead of staging its rows in LDS                            # the selection stage and the LI gather as two launches instead of k_select_gather
    {"PRE3_CHOL_EARLY": "0"},                               # the padded last panel runs all ten chain steps (the skipped ones change nothing)
    {"PRE3_CHOLP_PROJ": "0"},                               # the rescue stage's projection in the gate's blocks instead of on the persistent launch's strips behind their x-update
    {"PRE3_RIDE_POSE": "0"},                                # the prediction's projection riders wait for block 0 to publish the predicted pose instead of computing it themselves
    {"PRE3_HF_DEAL": "0"},                                  # k_hi_fused: every workgroup builds all of S = H P H' + I instead of dealing its rows out from 18 rescued landmarks on (round 6)
    {"PRE3_HP_MB": "0"},                                    # H*P of the measured rows one measurement per workgroup (k_ell_HP_build) instead of four sharing the pose rows' loads (round 6)
    {"PRE3_GATE_RIDE": "0"},                                # the rescue stage's chi2 gate as a launch of its own behind the Jnorm pass instead of riding in it (round 5: GateRide recomputes the
                                                            # normalised rows / columns 3..6 it needs from the un-normalised rows the consumers leave behind -- the same bits)
]      # (not here: PRE3_CHOL_PRO_B3 / PRE3_K9_B3 change the ARITHMETIC of the fp32 path -- f32 MFMA instead of the bf16 split -- not just the launches)


def _run(env_extra, tail="0", dump=None):
    # (PRE3_HI_FUSED_TWO=0: 33 .. 64 rescued landmarks take the host's general path in every variant -- k_hi_fused's two-panel path is another fp32
    #  evaluation of the same update (explicit inverse of the first diagonal block), compared to rounding in tests/test_gpu_hi_fused.py)
    env = dict(os.environ, PRE3_TAIL=tail, PRE3_HI_FUSED_TWO="0", **env_extra)
    if dump:
        env["VARIANT_DUMP"] = dump
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "variant_worker.py")], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = {l.split()[0]: l.split()[1] for l in r.stdout.splitlines() if l.startswith("DIGEST ") or l.startswith("FLAGS ")}
    assert "DIGEST" in lines and "FLAGS" in lines, r.stdout[-2000:]
    return lines


def test_launch_structure_variants_are_bit_identical():
    ref = _run(VARIANTS[0])["DIGEST"]
    for v in VARIANTS[1:]:
        assert _run(v)["DIGEST"] == ref, v


def test_the_step_tail_agrees_with_the_launch_per_stage_form(tmp_path):
    """PRE3_OPT_STEP_TAIL (default): rescue stage + HI update inside the LI update's persistent launch.  Against the launch-per-stage form on the
    same sequences: every step's statistics and inlier flags identical, the states after the last step equal to fp32 rounding (fp64 sequences
    do not take the tail: bit-identical)."""
    import numpy as np
    a, b = str(tmp_path / "tail.npz"), str(tmp_path / "stages.npz")
    ra, rb = _run({}, tail="1", dump=a), _run({}, tail="0", dump=b)
    assert ra["FLAGS"] == rb["FLAGS"]
    da, db = np.load(a), np.load(b)
    for k in da.files:
        if k.endswith("f64"):
            assert np.array_equal(da[k], db[k]), k
        else:
            tol = 3e-4 * np.abs(db[k]).max() if k.startswith("P") else 2e-5
            assert np.abs(da[k] - db[k]).max() < tol, (k, np.abs(da[k] - db[k]).max())
