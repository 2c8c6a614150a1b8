"""Round 5: the committed summaries under profiles/r5_* from the rocprofv3 outputs of tools/collect_profiles_r5.sh (gpurun_out/r5_*).  Every caption carries
the workload it was taken on, read from the bench line of the same session (gpurun_out/r5_bench_headline.json); LI and HI launches are told apart by
their duration class and labelled as such; a clock is only derived from counters and durations of ONE pass.

usage: python tools/make_profiles_r5.py [tag]"""
import collections, csv, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r5"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def find(d, name):
    for dp, _, fs in os.walk(os.path.join(G, d)):
        if name in fs:
            return os.path.join(dp, name)
    raise FileNotFoundError("%s/%s" % (d, name))


def rows(path):
    return list(csv.DictReader(open(path)))


def dur(r):
    return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3


def med(v):
    v = sorted(v)
    return v[len(v) // 2] if v else float("nan")


line = {}
try:
    line = json.loads([l for l in open(os.path.join(G, "%s_bench_headline.json" % tag)) if l.startswith("{")][-1])
except Exception as e:
    print("no bench line:", e)
cfg = line.get("config", {})
WL = "headline workload: N=500 (n=3013), 200 hypotheses, f32, RANSAC threshold %.1f px, motion noise %.1f: LI update ~%.0f rows, HI update ~%.0f rows per step" % (
    cfg.get("ransac_threshold_px", 1.0), cfg.get("motion_noise", 2.5), cfg.get("mean_li_rows", float("nan")), cfg.get("mean_hi_rows", float("nan")))
CMD = "python3 bench.py --no-cpu-baseline --no-extra-legs --no-check --legacy-steps 0"

# ---- headline: kernel stats + one-step timeline (a step whose rescue stage found work)
shutil.copy(find("%s_trace" % tag, "%s_kernel_stats.csv" % tag), os.path.join(P, "%s_bench_kernel_stats.csv" % tag))
tr = find("%s_trace" % tag, "%s_kernel_trace.csv" % tag)
tl = None; tl_any = None
for back in range(3, 24):
    cand = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_timeline.py"), tr, str(back)], capture_output=True, text=True, check=True).stdout
    tl = tl or cand
    hf = [l for l in cand.splitlines() if l.startswith("k_hi_fused")]
    if hf and float(hf[0].split()[3]) > 8.0:
        # (one LI launch in eight is bracketed by bench.py's HIP events, whose marker shows as a gap of ~4 us: a step without one is preferred)
        import re
        m = re.search(r"step wall[^:]*: ([0-9.]+) us, kernel busy ([0-9.]+) us", cand)
        if tl_any is None: tl_any = cand
        if m and float(m.group(1)) - float(m.group(2)) < 0.5:
            tl = cand
            break
        tl = tl_any
with open(os.path.join(P, "%s_bench_one_step_timeline.txt" % tag), "w") as fh:
    fh.write("# one filter step (%s) from rocprofv3 --kernel-trace of `%s --steps 40 --warmup 5`\n# (the profiler adds ~10 %% to the step; unprofiled numbers: DESIGN.md section 8)\n" % (WL, CMD) + tl)
try:
    trt = find("%s_trace_tail" % tag, "%s_kernel_trace.csv" % tag)
    cand = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_timeline.py"), trt, "7"], capture_output=True, text=True, check=True).stdout
    with open(os.path.join(P, "%s_bench_one_step_timeline_tail.txt" % tag), "w") as fh:
        fh.write("# the same step with PRE3_TAIL=1 (PRE3_OPT_STEP_TAIL: rescue stage + HI update inside k_cholp, one sweep of P; off by default) -- %s\n" % WL + cand)
    shutil.copy(find("%s_trace_tail" % tag, "%s_kernel_stats.csv" % tag), os.path.join(P, "%s_bench_kernel_stats_tail.csv" % tag))
except Exception as e:
    print("no tail trace:", e)

# ---- k_cholp launches: LI updates (the long class) and the general-path HI updates of more than 64 landmarks (the short class: step 0 of a sequence)
# (told apart by position: the first k_cholp behind a k_predict is that step's LI update, a second one the HI update's general path; the first
#  five steps are the warm-up, whose LI updates are still growing -- 216 .. 584 rows -- and are listed apart)
trs = sorted(rows(tr), key=lambda r: int(r["Start_Timestamp"]))
li, li_warm, hi2, step_no, seen = [], [], [], -1, 0
for r in trs:
    if "k_predict" in r["Kernel_Name"]:
        step_no += 1; seen = 0
    elif "k_cholp" in r["Kernel_Name"]:
        (hi2 if seen else (li_warm if step_no < 5 else li)).append(dur(r)); seen += 1
li.sort(); li_warm.sort(); hi2.sort()
out = ["# k_cholp in `%s --steps 40 --warmup 5` (%s), rocprofv3 --kernel-trace" % (CMD, WL),
       "# LI updates (factorisation + solve + x-update + down-date in one launch), timed steps: %d launches, median %.2f us, mean %.2f us, min %.2f us" % (len(li), med(li), sum(li) / len(li), li[0]),
       "# LI updates of the five warm-up steps (the map's first frames: fewer inliers): %s us" % ", ".join("%.1f" % d for d in li_warm)]
if hi2:
    out.append("# HI updates of more than 64 rescued landmarks (the host-polled general path; step 0 of the sequence, inside the warm-up): %d launches, median %.2f us" % (len(hi2), med(hi2)))
hf = sorted(dur(r) for r in rows(tr) if "k_hi_fused" in r["Kernel_Name"])
hd = sorted(dur(r) for r in rows(tr) if "k_downdate_b3" in r["Kernel_Name"])
out.append("# k_hi_fused (collection + HI update of <= 64 landmarks: one panel up to 32, two panels inside the launch up to 64): %d launches, median %.2f us (min %.2f: nothing rescued, max %.2f)" % (len(hf), med(hf), hf[0], hf[-1]))
out.append("# k_downdate_b3 (the HI updates' down-date; device-gated, ~4 us when there is nothing to do; with PRE3_OPT_PEND_HI -- bench.py's headline from round 6 on -- only where "
           "pending rows are flushed: the warm-up's steps, which complete their HI update inside the call): %d launches, median %.2f us, max %.2f us" % (len(hd), med(hd) if hd else 0.0, hd[-1] if hd else 0.0))
open(os.path.join(P, "%s_cholp_launches.txt" % tag), "w").write("\n".join(out) + "\n")
print("\n".join(out))


# ---- counters: per kernel and duration class, durations from the SAME pass's kernel trace
def pmc_summary(dirname, kern, cls):
    res = {}
    base = os.path.join(G, dirname)
    if not os.path.isdir(base):
        return res
    for dp, _, fs in sorted(os.walk(base)):
        if "p_counter_collection.csv" not in fs:
            continue
        cc = [r for r in rows(os.path.join(dp, "p_counter_collection.csv")) if kern in r["Kernel_Name"]]
        kt = {r["Dispatch_Id"]: dur(r) for r in rows(os.path.join(dp, "p_kernel_trace.csv")) if kern in r["Kernel_Name"]} if "p_kernel_trace.csv" in fs else {}
        byc = collections.defaultdict(list)
        for r in cc:
            byc[r["Counter_Name"]].append((float(r["Counter_Value"]), kt.get(r.get("Dispatch_Id", ""), float("nan"))))
        for name, v in byc.items():
            dmax = max(d for _, d in v)
            sel = [x for x in v if cls(x[1], dmax)]
            if sel:
                res[name] = {"launches": len(sel), "mean": sum(x[0] for x in sel) / len(sel), "mean_duration_us_same_pass": sum(x[1] for x in sel) / len(sel)}
    return res


LONG = lambda d, dmax: d > 0.5 * dmax
ALLW = lambda d, dmax: d > 8.0          # launches that did work (the device-gated ones with nothing to do take ~4 us)
cholp = pmc_summary("%s_pmc" % tag, "k_cholp", LONG)
if cholp:
    o = {"command": "rocprofv3 --pmc <one set per pass> --kernel-trace --output-format csv -- %s --steps 10 --warmup 2" % CMD, "workload": WL,
         "kernel": "k_cholp, the LI launches (the long duration class of each pass)", "counters_per_li_launch": cholp,
         "mean_rows": cfg.get("mean_li_rows")}
    if "FETCH_SIZE" in cholp and "WRITE_SIZE" in cholp:
        f_, w_ = cholp["FETCH_SIZE"]["mean"] * 1024, cholp["WRITE_SIZE"]["mean"] * 1024
        o["hbm_bytes_per_li_launch"] = {"raw": f_ + w_, "fetch_doubled": 2 * f_ + w_,
                                        "note": "FETCH_SIZE on gfx950 reports half the bytes of 16 B/lane streams (MI355X_MICROARCH.md): the read side lies between the raw figure and twice it"}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cholp and "GRBM_GUI_ACTIVE" in cholp:
        # GRBM_GUI_ACTIVE arrives summed over its instances: 8 (one per XCD; rounds 4-5) or 128 (round 6's session): the divisor that gives a plausible clock
        d_us = cholp["GRBM_GUI_ACTIVE"]["mean_duration_us_same_pass"]
        div = min((8.0, 16.0, 64.0, 128.0), key=lambda dv: abs(cholp["GRBM_GUI_ACTIVE"]["mean"] / dv / (d_us * 1e3) - 2.4))
        cyc = cholp["GRBM_GUI_ACTIVE"]["mean"] / div
        o["mfma_busy_fraction_of_chip"] = cholp["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (cyc * 1024.0)
        o["clock_GHz_of_the_GRBM_pass"] = cyc / (d_us * 1e3)
        o["grbm_instances_assumed"] = div
    json.dump(o, open(os.path.join(P, "%s_pmc_cholp.json" % tag), "w"), indent=1)
    print("k_cholp LI:", {k: round(v["mean"], 1) for k, v in cholp.items() if k in ("FETCH_SIZE", "WRITE_SIZE")}, o.get("mfma_busy_fraction_of_chip"))
hidd = pmc_summary("%s_pmc" % tag, "k_downdate_b3", ALLW)
if hidd:
    o = {"command": "the same passes", "workload": WL, "kernel": "k_downdate_b3: the HI updates' down-dates that had work (one panel, <= 64 rows: HBM-bound read-modify-write of P)",
         "counters_per_hi_launch": hidd}
    if "FETCH_SIZE" in hidd and "WRITE_SIZE" in hidd:
        f_, w_, d_ = hidd["FETCH_SIZE"]["mean"] * 1024, hidd["WRITE_SIZE"]["mean"] * 1024, hidd["WRITE_SIZE"]["mean_duration_us_same_pass"]
        o["hbm"] = {"raw_bytes": f_ + w_, "fetch_doubled_bytes": 2 * f_ + w_, "TBps_raw": (f_ + w_) / d_ / 1e6, "TBps_fetch_doubled": (2 * f_ + w_) / d_ / 1e6}
    json.dump(o, open(os.path.join(P, "%s_pmc_hi_downdate.json" % tag), "w"), indent=1)
    print("HI down-date:", o.get("hbm"))
k9s = pmc_summary("%s_pmc_k9s" % tag, "k_downdate_b3", LONG)
if k9s:
    n, r = 3013, 554
    o = {"command": "K9_ROWS=554 rocprofv3 --pmc <one set per pass> --kernel-trace -- python3 tools/k9ab.py", "workload": "stand-alone K9: k_downdate_b3 at n = 3013, r = 554 through pre3_bench_downdate (planes already split, back-to-back launches)",
         "counters_per_launch": k9s}
    d_ = k9s.get("WRITE_SIZE", k9s.get("FETCH_SIZE", {})).get("mean_duration_us_same_pass")
    if d_:
        o["f32_equivalent_TFLOPs"] = n * (n + 1.0) * r / d_ / 1e6
        o["of_f32_mfma_peak_157.3"] = o["f32_equivalent_TFLOPs"] / 157.3
        o["executed_bf16_of_2500"] = 6 * o["f32_equivalent_TFLOPs"] / 2500.0
    if "FETCH_SIZE" in k9s and "WRITE_SIZE" in k9s:
        f_, w_ = k9s["FETCH_SIZE"]["mean"] * 1024, k9s["WRITE_SIZE"]["mean"] * 1024
        o["hbm_bytes_per_launch"] = {"raw": f_ + w_, "fetch_doubled": 2 * f_ + w_, "algorithmic": "P upper 18.2 MB read + P 36.3 MB written + planes 10 MB read = ~65 MB"}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in k9s and "GRBM_GUI_ACTIVE" in k9s:
        d_us = k9s["GRBM_GUI_ACTIVE"]["mean_duration_us_same_pass"]
        cyc = k9s["GRBM_GUI_ACTIVE"]["mean"] / min((8.0, 16.0, 64.0, 128.0), key=lambda dv: abs(k9s["GRBM_GUI_ACTIVE"]["mean"] / dv / (d_us * 1e3) - 2.4))
        o["mfma_busy_fraction_of_chip"] = k9s["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (cyc * 1024.0)
    json.dump(o, open(os.path.join(P, "%s_pmc_k9_standalone.json" % tag), "w"), indent=1)
    print("K9 stand-alone:", {k: o[k] for k in o if k.startswith("of_") or k.startswith("mfma") or k.startswith("f32")})

# ---- frame and fp64_n200 legs: per-kernel stats and one timeline each
for leg, first, cap in (("frame", "k_map_", "one mono_slam.m frame at N=500, K2=600: map management (delete + add), prediction, scan upload, IC search, RANSAC + updates (bench.py frame leg; tools/frame_trace.py 24)"),
                        ("fp64", "k_predict", "BASELINE configs[1]: N=200 (n=1213), fp64, pre3_step_all (predict + update of all 160 measured landmarks, r = 320) (bench.py fp64_n200 leg; tools/fp64_trace.py)")):
    try:
        shutil.copy(find("%s_%s" % (tag, leg), "t_kernel_stats.csv"), os.path.join(P, "%s_%s_kernel_stats.csv" % (tag, "frame" if leg == "frame" else "fp64_n200")))
        kt = sorted(rows(find("%s_%s" % (tag, leg), "t_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
        starts = [i for i, r in enumerate(kt) if first in r["Kernel_Name"] and (i == 0 or first not in kt[i - 1]["Kernel_Name"] or leg == "fp64")]
        # the iteration with the MEDIAN wall time (under the profiler some iterations of the frame leg wait for the host between launches)
        walls = sorted((int(kt[starts[q + 1]]["Start_Timestamp"]) - int(kt[starts[q]]["Start_Timestamp"]), q) for q in range(2, len(starts) - 1))
        qm = walls[len(walls) // 2][1]
        a, b = starts[qm], starts[qm + 1]
        t0 = int(kt[a]["Start_Timestamp"]); prev = None
        lines = ["# %s -- one iteration from rocprofv3 --kernel-trace" % cap, "# kernel | start us | duration us | gap to previous us"]
        for r in kt[a:b]:
            st = (int(r["Start_Timestamp"]) - t0) / 1e3
            lines.append("%-44s %8.1f %8.1f %6.1f" % (r["Kernel_Name"].split("(")[0].replace("void pre3::", "").replace("pre3::", "")[:44], st, dur(r), 0.0 if prev is None else st - prev))
            prev = st + dur(r)
        lines.append("# iteration wall: %.1f us, kernel busy %.1f us, %d launches" % ((int(kt[b]["Start_Timestamp"]) - t0) / 1e3, sum(dur(r) for r in kt[a:b]), b - a))
        if leg == "frame":
            lines.append("# (under rocprofv3 the host side of this leg -- descriptor building for map management, the caller's hypothesis draws, ~19 launches per frame --")
            lines.append("#  does not keep the device busy: the gaps above are the profiler's; unprofiled the leg runs at %s frames/s, bench.py `frame`)" % ("%.0f" % line.get("frame", {}).get("frames_per_s", float("nan")) if isinstance(line.get("frame"), dict) else "3.1-3.2 k"))
        open(os.path.join(P, "%s_%s_timeline.txt" % (tag, "frame" if leg == "frame" else "fp64_n200")), "w").write("\n".join(lines) + "\n")
        print(lines[-1])
    except Exception as e:
        print("no %s summary:" % leg, repr(e))

# ---- matchers, tail probe
for d, pre, dst in (("%s_match" % tag, "m", "%s_match_kernel_stats.csv" % tag), ("%s_matchf" % tag, "mf", "%s_match_float_kernel_stats.csv" % tag)):
    try:
        shutil.copy(find(d, "%s_kernel_stats.csv" % pre), os.path.join(P, dst))
    except FileNotFoundError as e:
        print("no matcher stats:", e)
try:
    txt = open(os.path.join(G, "%s_probe_tail.txt" % tag)).read()
    open(os.path.join(P, "%s_tail_timeline.txt" % tag), "w").write(
        "# device-side wall-clock stamps (s_memrealtime, probe build) of ONE persistent launch with the rescue stage + HI update inside (PRE3_OPT_STEP_TAIL = 1;\n"
        "# tools/probe_tail.py 30: N=500, motion noise 2.5, step 31 of the sequence).  See DESIGN.md section 5d for the reading.\n" + "\n".join(l for l in txt.splitlines() if "amdgpu.ids" not in l) + "\n")
except Exception as e:
    print("no tail probe:", e)
