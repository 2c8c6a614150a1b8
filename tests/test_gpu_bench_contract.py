"""GPU: bench.py's output contract, end to end in a child process: EXACTLY one line on stdout (whatever libraries print on the way -- RCCL's
version banner comes from C when libpre3's communicator is created), it parses as JSON, and it carries the keys the round driver and the
judge read, with the roofline and the auxiliary legs filled in."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, expect_rc=0):
    env = dict(os.environ, **(env_extra or {}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"] + extra,
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == expect_rc, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must hold exactly one line, got %d:\n%s" % (len(lines), r.stdout[:1500])
    return json.loads(lines[0])


def test_one_json_line_with_the_contract_s_keys():
    d = _run([])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "steps/s" and d["value"] > 0 and abs(d["value"] * d["ms_per_step"] * 1e-3 - 1.0) < 1e-6
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert d["checked"] is True and d["li_set_equal"] and d["hi_set_equal"]
    # the headline is SURVEY 8(d)'s step: the reference's threshold, an HI update in (nearly) every step; the two legs keep rounds 1-3 comparable
    assert d["config"]["ransac_threshold_px"] == 1.0 and d["config"]["mean_hi_rows"] > 0
    assert d["thr05"]["threshold_px"] == 0.5 and d["thr05"]["mean_hi_rows"] > 0 and d["no_hi"]["threshold_px"] == 1.0
    assert "cholp" in rf["kernel"] and rf["launches"] >= 1 and rf["avg_launch_us"] > 0          # the fused launch is what the bracket times
    assert d["fp64_n200"]["k9"]["frac"] > 0 and d["fp64_n200"]["max_abs_err_vs_c_oracle"]["P"] < 1e-10 * d["fp64_n200"]["max_abs_err_vs_c_oracle"]["P_scale"] + 1e-18
    assert d["frame"]["frames_per_s"] > 0 and d["frame"]["mean_ic_matches"] > 100
    # the auxiliary legs, through libpre3's own communicator (one rank)
    assert d["rccl"]["world"] == 1 and "ncclAllReduce enqueued by libpre3" in d["ransac_shard"]["collective"]
    assert d["ransac_shard"]["value"] > 0 and d["matcher_shard"]["matches"] > 0 and d["matcher"]["ms_per_match"] > 0


def test_a_hung_leg_still_leaves_one_line_and_a_non_zero_exit():
    d = _run([], {"PRE3_BENCH_HANG_LEG": "matcher_shard", "PRE3_BENCH_LEG_TIMEOUT": "6"}, expect_rc=3)
    assert d["value"] > 0 and "timed out" in d["legs"] and "ransac_shard" in d and "matcher_shard" not in d
