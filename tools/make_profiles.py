"""Turn the rocprofv3 outputs of one measurement session (tools/collect_profiles.sh <tag>) into the committed summaries under profiles/.

    gpurun_out/r1_trace/      rocprofv3 --kernel-trace --stats  -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs
    gpurun_out/r1_pmc_fetch/  rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs
    gpurun_out/r1_pmc_write/  the same with WRITE_SIZE            (separate passes: the two counters do not fit one, MI355X_MICROARCH.md)

usage: python tools/make_profiles.py [tag]      (tag defaults to r1)"""
import csv, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r4"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
K9 = "k_downdate_b3"
FUSED = "k_cholp"      # round 4: the LI update's down-date runs inside the persistent factorisation's launch (pre3_cholp.hip)

def find(d, name):
    for dp, _, fs in os.walk(os.path.join(G, d)):
        if name in fs:
            return os.path.join(dp, name)
    raise FileNotFoundError("%s/%s" % (d, name))


shutil.copy(find("%s_trace" % tag, "%s_kernel_stats.csv" % tag), os.path.join(P, "%s_bench_kernel_stats.csv" % tag))
tl = None
for back in range(3, 24):          # a step whose rescue stage found work (k_hi_fused did an update: > 8 us)
    cand = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_timeline.py"), find("%s_trace" % tag, "%s_kernel_trace.csv" % tag), str(back)],
                          capture_output=True, text=True, check=True).stdout
    tl = tl or cand
    hf = [l for l in cand.splitlines() if l.startswith("k_hi_fused")]
    if hf and float(hf[0].split()[3]) > 8.0:
        tl = cand
        break
with open(os.path.join(P, "%s_bench_one_step_timeline.txt" % tag), "w") as fh:
    fh.write("# one filter step of the headline workload (N=500, n=3013, 200 hypotheses, f32, RANSAC threshold 1.0 px, motion noise 2.5: LI update ~550 rows + an HI update) from rocprofv3 "
             "--kernel-trace of `python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs --no-check --legacy-steps 0`\n# (the profiler adds ~10 % to the step; unprofiled numbers are in DESIGN.md section 8)\n" + tl)


def counter(dirname, prefix, name):
    rows = [r for r in csv.DictReader(open(find(dirname, "%s_counter_collection.csv" % prefix))) if K9 in r["Kernel_Name"] and r["Counter_Name"] == name]
    v = [float(r["Counter_Value"]) for r in rows]
    big = [x for x in v if x > 0.8 * max(v)]                 # the r ~ 640 launches (LI updates); HI updates are far smaller
    return dict(launches=len(v), avg_KB=sum(v) / len(v), max_KB=max(v), li_launch_avg_KB=sum(big) / len(big))


fs, ws = counter("%s_pmc_fetch" % tag, "f", "FETCH_SIZE"), counter("%s_pmc_write" % tag, "w", "WRITE_SIZE")      # (round 4: the K9 launches left are the HI updates')
tr = [r for r in csv.DictReader(open(find("%s_trace" % tag, "%s_kernel_trace.csv" % tag))) if K9 in r["Kernel_Name"]]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr]
big = [d for d in dur if d > 0.6 * max(dur)]
out = {
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs (two separate passes)",
    "kernel": "k_downdate_b3 (K9 as a three-way bf16 split on the bf16 MFMA, 128x128 + 64x64 tiles; the x-update and rescue-projection workgroups ride in the same launch)",
    "counters_KB": {"FETCH_SIZE": fs, "WRITE_SIZE": ws},
    "k9_trace_durations_us": {"launches": len(dur), "avg": sum(dur) / len(dur), "li_launch_avg": sum(big) / len(big)},
    "notes": [
        "WRITE_SIZE of an LI launch = ld^2*4 B (ld = 3072): the whole padded P is written once per launch (upper tile + mirrored tile).",
        "FETCH_SIZE on gfx950 reports half the bytes of 16 B/lane streams (MI355X_MICROARCH.md, HBM section); the plane staging loads are 16 B/lane LDS-DMA, "
        "the P-tile prefetch is 4 B/lane (uncalibrated), so the read side lies between the raw figure and twice it.",
        "algorithmic per LI launch (r=640, n=3013): P upper triangle read 18.2 MB + P written 36.3 MB + bf16 planes of W 11.6 MB = 66 MB",
    ],
    "hbm_bytes_per_li_launch": {"raw": 1024.0 * (fs["li_launch_avg_KB"] + ws["li_launch_avg_KB"]),
                                "fetch_doubled": 1024.0 * (2 * fs["li_launch_avg_KB"] + ws["li_launch_avg_KB"])},
}
# SQ counters of the K9 launches (the LI updates: the upper half by SQ_BUSY_CYCLES-independent ranking on each counter), from the separate --pmc passes
sqk = {}
sqd0 = os.path.join(G, "%s_pmc_sq" % tag)
if os.path.isdir(sqd0):
    for dp, _, fs_ in sorted(os.walk(sqd0)):
        for fn in fs_:
            if fn.endswith("counter_collection.csv"):
                acc = {}
                for r in csv.DictReader(open(os.path.join(dp, fn))):
                    if K9 in r["Kernel_Name"]:
                        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                for k_, v in acc.items():
                    top = [x for x in v if x > 0.6 * max(v)] if max(v) > 0 else v
                    sqk[k_] = {"li_launch_avg": sum(top) / len(top), "launches": len(top)}
if sqk:
    out["sq_counters_per_li_launch"] = sqk
    if "SQ_VALU_MFMA_BUSY_CYCLES" in sqk and "GRBM_GUI_ACTIVE" in sqk:
        # GRBM_GUI_ACTIVE comes back accumulated over the 8 XCDs (its value is ~8 x launch duration x shader clock: check below), so the
        # chip's matrix-pipe capacity during the launch is (GRBM / 8) cycles x 1024 SIMDs
        cyc = sqk["GRBM_GUI_ACTIVE"]["li_launch_avg"] / 8.0
        out["mfma_busy_fraction_of_chip"] = sqk["SQ_VALU_MFMA_BUSY_CYCLES"]["li_launch_avg"] / (cyc * 1024.0)
        out["implied_clock_GHz"] = cyc / (out["k9_trace_durations_us"]["li_launch_avg"] * 1e3)
        out["notes"].append("mfma_busy_fraction_of_chip = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) x 1024 SIMDs): the share of the chip's "
                            "matrix-pipe cycles the LI launch kept busy; SQ_VALU_MFMA_BUSY_CYCLES = 32 x SQ_INSTS_MFMA (v_mfma_f32_32x32x16_bf16: 8 passes); "
                            "implied_clock_GHz = (GRBM_GUI_ACTIVE / 8) / launch duration -- the clock the chip held under this launch")
with open(os.path.join(P, "%s_pmc_k9.json" % tag), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out["counters_KB"], indent=1), out["k9_trace_durations_us"], out["hbm_bytes_per_li_launch"])

# ---- round 4: the fused launch (k_cholp with the down-date consumers inside): HBM bytes and SQ counters of its LI launches
def counter_k(dirname, prefix, name, kern):
    rows = [r for r in csv.DictReader(open(find(dirname, "%s_counter_collection.csv" % prefix))) if kern in r["Kernel_Name"] and r["Counter_Name"] == name]
    v = [float(r["Counter_Value"]) for r in rows]
    big = [x for x in v if x > 0.6 * max(v)]
    return dict(launches=len(v), li_launches=len(big), li_launch_avg_KB=sum(big) / len(big))
try:
    ff, wf = counter_k("%s_pmc_fetch" % tag, "f", "FETCH_SIZE", FUSED), counter_k("%s_pmc_write" % tag, "w", "WRITE_SIZE", FUSED)
    trf = [r for r in csv.DictReader(open(find("%s_trace" % tag, "%s_kernel_trace.csv" % tag))) if FUSED in r["Kernel_Name"]]
    durf = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in trf]
    bigf = [d for d in durf if d > 0.6 * max(durf)]
    fo = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE|<SQ sets> --kernel-trace --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs --no-check --legacy-steps 0 (separate passes)",
          "kernel": "k_cholp with the down-date consumers inside (update.m:32-38 of an LI update in one launch)",
          "counters_KB": {"FETCH_SIZE": ff, "WRITE_SIZE": wf},
          "trace_durations_us": {"launches": len(durf), "li_launches": len(bigf), "li_launch_avg": sum(bigf) / len(bigf), "li_launch_median": sorted(bigf)[len(bigf) // 2]},
          "mean_rows": "~550 (headline workload, 9 panels)",
          "hbm_bytes_per_li_launch": {"raw": 1024.0 * (ff["li_launch_avg_KB"] + wf["li_launch_avg_KB"]), "fetch_doubled": 1024.0 * (2 * ff["li_launch_avg_KB"] + wf["li_launch_avg_KB"])},
          "notes": ["algorithmic per LI launch (r ~ 550, n = 3013): P upper triangle read 18.2 MB + P written 36.3 MB (tile + mirror) + W (f32) written 6.6 MB + its bf16 planes written 10 MB and read by the consumers + S / HP read 7.8 MB",
                    "FETCH_SIZE on gfx950 reports half the bytes of 16 B/lane streams (MI355X_MICROARCH.md): the read side lies between the raw figure and twice it"]}
    sqf = {}
    sqd1 = os.path.join(G, "%s_pmc_sq" % tag)
    if os.path.isdir(sqd1):
        for dp, _, fs_ in sorted(os.walk(sqd1)):
            for fn in fs_:
                if fn.endswith("counter_collection.csv"):
                    acc = {}
                    for r in csv.DictReader(open(os.path.join(dp, fn))):
                        if FUSED in r["Kernel_Name"]:
                            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                    for k_, v in acc.items():
                        top = [x for x in v if x > 0.6 * max(v)] if max(v) > 0 else v
                        sqf[k_] = {"li_launch_avg": sum(top) / len(top), "launches": len(top)}
    if sqf:
        fo["sq_counters_per_li_launch"] = sqf
        if "SQ_VALU_MFMA_BUSY_CYCLES" in sqf and "GRBM_GUI_ACTIVE" in sqf:
            cyc = sqf["GRBM_GUI_ACTIVE"]["li_launch_avg"] / 8.0
            fo["mfma_busy_fraction_of_chip"] = sqf["SQ_VALU_MFMA_BUSY_CYCLES"]["li_launch_avg"] / (cyc * 1024.0)
            fo["implied_clock_GHz"] = cyc / (fo["trace_durations_us"]["li_launch_avg"] * 1e3)
    with open(os.path.join(P, "%s_pmc_cholp.json" % tag), "w") as fh:
        json.dump(fo, fh, indent=1)
    print("fused launch:", json.dumps(fo["trace_durations_us"]), fo["hbm_bytes_per_li_launch"], fo.get("mfma_busy_fraction_of_chip"))
except Exception as e:
    print("no fused-launch PMC summary:", repr(e))

# ---- the factorisation + solve as one persistent launch (k_cholp, round 3): durations of the LI launches (r ~ 640: ten panels) and of the HI
#      launches (one panel), and the SQ counters of its launches
import collections
cp = [((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in csv.DictReader(open(find("%s_trace" % tag, "%s_kernel_trace.csv" % tag))) if "k_cholp" in r["Kernel_Name"]]
if cp:
    big_ = sorted(d for d in cp if d > 0.5 * max(cp)); small_ = sorted(d for d in cp if d <= 0.5 * max(cp))
    cl = ["# k_cholp (K8, round 3: S = L L' and W = L^-1 [HP | nu] in ONE persistent launch) in `bench.py --steps 40` (N=500, n=3013), rocprofv3 --kernel-trace",
          "# LI updates (r ~ 540 rows at the bench's default RANSAC threshold of 0.5 px: nine 64-row panels; ten at r ~ 640): %d launches, median %.2f us, mean %.2f us, min %.2f us" % (len(big_), big_[len(big_) // 2], sum(big_) / len(big_), big_[0])]
    if small_:
        cl.append("# HI updates of two and more panels (one-panel updates take k_chol_step: one launch in either form): %d launches, median %.2f us, mean %.2f us" % (len(small_), small_[len(small_) // 2], sum(small_) / len(small_)))
    sqd = os.path.join(G, "%s_pmc_sq" % tag)
    if os.path.isdir(sqd):
        cl.append("# SQ counters per launch (separate --pmc passes, means over the k_cholp launches above the median of 10 steps: the LI updates)")
        for dp, _, fs_ in sorted(os.walk(sqd)):
            for fn in fs_:
                if fn.endswith("counter_collection.csv"):
                    acc = collections.defaultdict(list)
                    for r in csv.DictReader(open(os.path.join(dp, fn))):
                        if "k_cholp" in r["Kernel_Name"]:
                            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    for k_, v in sorted(acc.items()):
                        top = sorted(v)[len(v) // 2:]
                        cl.append("%-28s %12.5g  (%d launches)" % (k_, sum(top) / len(top), len(top)))
    with open(os.path.join(P, "%s_cholp_launches.txt" % tag), "w") as fh:
        fh.write("\n".join(cl) + "\n")
    print("\n".join(cl))

# ---- the factorisation (k_chol_step): per-launch durations by panel and the SQ counters of its launches
by = collections.defaultdict(list)
for r in csv.DictReader(open(find("%s_trace" % tag, "%s_kernel_trace.csv" % tag))):
    if "k_chol_step" in r["Kernel_Name"]:
        by[int(r.get("Grid_Size", r.get("Grid_Size_X", 0)))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
lines = ["# k_chol_step (K8) in `bench.py --steps 40` (N=500, n=3013, r ~ 640: ten 64-column panels per LI update), rocprofv3 --kernel-trace",
         "# grid threads | launches | median us | mean us"]
tot = 0.0
for g in sorted(by, reverse=True):
    v = sorted(by[g])
    if len(v) > 20:
        lines.append("%8d %6d %8.2f %8.2f" % (g, len(v), v[len(v) // 2], sum(v) / len(v)))
        tot += v[len(v) // 2]
lines.append("# sum of the medians of one LI update's panels: %.1f us" % tot)
sq = os.path.join(G, "%s_pmc_sq" % tag)
if os.path.isdir(sq):
    lines.append("# SQ counters per launch (separate --pmc passes, means over the k_chol_step launches of 10 steps)")
    for dp, _, fs in sorted(os.walk(sq)):
        for fn in fs:
            if fn.endswith("counter_collection.csv"):
                acc = collections.defaultdict(list)
                for r in csv.DictReader(open(os.path.join(dp, fn))):
                    if "k_chol_step" in r["Kernel_Name"]:
                        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                for k_, v in sorted(acc.items()):
                    lines.append("%-28s %12.5g  (%d launches)" % (k_, sum(v) / len(v), len(v)))
if by:
    with open(os.path.join(P, "%s_chol_launches.txt" % tag), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines[-14:]))

# ---- the matcher (k_match_i8_q)
try:
    shutil.copy(find("%s_match" % tag, "m_kernel_stats.csv"), os.path.join(P, "%s_match_kernel_stats.csv" % tag))
    rows = [r for r in csv.DictReader(open(find("%s_match" % tag, "m_kernel_trace.csv"))) if "k_match_i8_q" in r["Kernel_Name"]]
    gkey = "Grid_Size" if "Grid_Size" in rows[0] else "Grid_Size_X"
    gmax = max(int(r[gkey]) for r in rows)                          # the 4096 x 4096 launches (the ragged shapes of the same script have smaller grids)
    big = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if int(r[gkey]) == gmax)
    big = big[:max(1, int(0.9 * len(big)))]                         # drop the slowest tenth (first launches)
    with open(os.path.join(P, "%s_match_kernel_stats.csv" % tag), "a") as fh:
        fh.write("# k_match_i8_q at 4096 x 4096 x 128 uint8 (the %d largest launches of tools/match_ab.py child): mean %.2f us, min %.2f us\n" % (len(big), sum(big) / len(big), min(big)))
    print("matcher 4096^2 launches: mean %.2f us" % (sum(big) / len(big)))
except FileNotFoundError as e:
    print("no matcher trace:", e)

# ---- the float / double classes on the matrix cores (k_rank_pack, k_match_rank; tools/match_float.py)
try:
    shutil.copy(find("%s_matchf" % tag, "mf_kernel_stats.csv"), os.path.join(P, "%s_match_float_kernel_stats.csv" % tag))
    by = {}
    for r in csv.DictReader(open(find("%s_matchf" % tag, "mf_kernel_trace.csv"))):
        for k_ in ("k_match_rank<double>", "k_match_rank<float>", "k_match_i8_q", "k_match_exact_tiled<double>", "k_match_exact_tiled<float>"):
            if k_ in r["Kernel_Name"]:
                by.setdefault(k_, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    with open(os.path.join(P, "%s_match_float_kernel_stats.csv" % tag), "a") as fh:
        for k_, v in sorted(by.items()):
            fh.write("# %s at 4096 x 4096 x 128 (tools/match_float.py): %d launches, mean %.2f us, min %.2f us\n" % (k_, len(v), sum(v) / len(v), min(v)))
            print("%s: mean %.2f us over %d launches" % (k_, sum(v) / len(v), len(v)))
except FileNotFoundError as e:
    print("no float-class matcher trace:", e)
