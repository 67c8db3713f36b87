#!/usr/bin/env python3
"""Does the driver-run bench line agree with the rocprofv3 kernel trace of the same command?
  python tools/compare_stages_to_trace.py <bench.json> <dir with *_kernel_trace.csv> [<untraced bench.json>]  ->  markdown table
For every leg of the line: the in-run time (HIP events) against the sum of the trace's median durations of the kernels the leg
launches (matched by kernel name and grid), and the ratio.  Legs made of many launches per pass (kNN) use launches / passes."""
import collections
import csv
import glob
import json
import statistics as st
import sys


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    for cut in ("(float", "(double", "(unsigned", "(HIP_vector", "(long", "(int", "(Roi", "(char", "(void", "(Gemm", "(Proj", "(runia"):
        name = name.split(cut)[0]
    return name.strip()


bench = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
other = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1]) if len(sys.argv) > 3 else None
trace = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
groups = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    k = r["Kernel_Name"]
    if "at::native" in k or "rocclr" in k or "rocprim" in k:
        continue
    wg = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    groups[(short(k), wg, int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)


def med(name, gx=None, gy=None):
    hits = [v for (n, x, y, z), v in groups.items() if name in n and (gx is None or x == gx) and (gy is None or y == gy)]
    if not hits:
        return None, 0
    allv = [d for v in hits for d in v]
    return st.median(allv), len(allv)


def total(name_parts):
    return sum(sum(v) for (n, x, y, z), v in groups.items() if any(p in n for p in name_parts))


stg = bench["stages"]
passes3 = 3  # cfg3 leg of the default line: 1 warm-up + 2 timed passes
rows = []


def add(label, in_run, parts, note=""):
    t, missing = 0.0, []
    for p in parts:
        m, n = med(*p)
        if m is None:
            missing.append(p[0])
        else:
            t += m
    rows.append((label, in_run, t if not missing else None, note + (" MISSING " + ",".join(missing) if missing else "")))


add("roofline K1 `mc_entropy_kernel<4,4,16,5>` (10 000 images)", bench["roofline"]["avg_launch_ms"], [("mc_entropy_kernel<4, 4, 16, 5, true, true, 0>", 40000)],
    "events attached to the dispatch")
k2, _ = med("proj_sq_kernel<1, 2, true, true, true>", 1264)
k0, _ = med("mc_mask_bits_kernel<4, 4, 16, true>", 2500)
k1, _ = med("mc_entropy_kernel<4, 4, 16, 5, true, true, 0>", 40000)
rows.append(("step = K0 + K1 + K2' (kernel time only)", bench["ms_per_step_stats"]["median"], (k0 or 0) + (k1 or 0) + (k2 or 0), "median step (events) vs sum of the three kernels: the difference is dispatch gaps"))
add("stages.mahalanobis (1 M x 2048)", stg["mahalanobis"]["ms"], [("gemm_rows_kernel<float, double, 5, 2, 4>", 31250)])
add("stages.energy_c1000", stg["energy_c1000"]["ms"], [("lse_wave_kernel<4>", 125000)])
# cfg3's kNN pass = one l2-normalisation of the 1 M queries + the same launches for every query chunk: the groups of kNN kernels that
# were launched (chunks x passes) times - other legs (the harness sweep, tests of small shapes) launch the same kernels on other grids
knn_names = ("knn_dist_bf16_kernel", "kth_select_lists_kernel", "knn_tau_kernel", "knn_gather_rows_kernel", "split_bf16_kernel",
             "kth_select_range_kernel", "row_sqnorm_kernel")
by_count = collections.Counter(len(v) for (n, x, y, z), v in groups.items() if "knn_dist_bf16_kernel<true>" in n)
chunk_launches = max(by_count, default=0)  # the main pass of the filter: one launch per chunk and pass
knn_ms = sum(sum(v) for (n, x, y, z), v in groups.items() if any(p in n for p in knn_names) and len(v) == chunk_launches)
knn_ms += sum(sum(v) for (n, x, y, z), v in groups.items() if "l2_normalize_kernel<8>" in n and x == 125000)
# (the sample pass, the list selection and the threshold kernel share their grids with the harness sweep's kNN: per-launch medians)
for name, gx in (("knn_dist_bf16_kernel<false>", 512), ("kth_select_lists_kernel", 4096), ("knn_tau_kernel", 4096)):
    m, n = med(name, gx)
    if m is not None and n != chunk_launches:
        knn_ms += m * chunk_launches
rows.append(("stages.knn (1 M queries; all kNN kernels of the cfg3 passes / 3)", stg["knn"]["ms"], None, "see below"))
L = stg["cfg4_lared"]
add("cfg4_lared.entropy", L["entropy"]["ms"], [("entropy_per_dim_kernel<16, 5, 4>", 100000)])
add("cfg4_lared.pca", L["pca"]["ms"], [("gemm_rows_kernel<double, double, 0, 2, 4>", 3072), ("gemm_rows_kernel<double, double, 0, 1, 4>", 106)])
add("cfg4_lared.kde", L["kde"]["ms"], [("gemm_rows_kernel<double, double, 4, 2, 4>", 3072), ("gemm_rows_kernel<double, double, 4, 1, 4>", 1696), ("kde_replay_kernel", 106)])
M = stg["cfg4_from_feature_maps"]
add("cfg4_from_feature_maps.channels_last_copy", M["channels_last_copy"]["ms"], [("nchw_to_nhwc_kernel", 57)])
add("cfg4_from_feature_maps.roi_sampler_entropy", M["roi_sampler_entropy"]["ms"],
    [("mc_entropy_kernel<7, 7, 16, 5, true, true, 2>", 524288), ("mc_entropy_kernel<7, 7, 16, 5, true, true, 2>", 275776),
     ("mc_mask_bits_kernel<7, 7, 16, false>", 16384), ("mc_mask_bits_kernel<7, 7, 16, false>", 8617), ("roi_sample_table_kernel", None)],
    "two slices of <= 65 535 proposals: sampler launches + keep-flag tables + one sample table (median of both slices)")
F = stg["fits"]
add("fits.covariance (50 000 x 2048)", F["covariance"]["ms"], [("gram_kernel<float, true>", 136, 15), ("col_sum_kernel<float>", 8), ("gram_finish_kernel", 2080), ("col_mean_finish_kernel", 32)])
E = stg["entropy_joint"]
add("entropy_joint.joint", E["joint"]["ms"], [("entropy_joint_reg_kernel<16, 0>", 10000)])
add("entropy_joint.per_dim", E["per_dim"]["ms"], [("entropy_per_dim_kernel<16, 5, 4>", 5000)])
if "both_one_read" in E:
    add("entropy_joint.both_one_read", E["both_one_read"]["ms"], [("entropy_joint_reg_kernel<16, 5>", 10000)])
f4 = stg["f4"]
add("f4.vim", f4["vim"]["ms"], [("gemm_rows_kernel<float, float, 3, 2, 4>", 31232), ("gemm_rows_kernel<float, float, 3, 1, 4>", None), ("lse_wave_kernel<4>", 125000)])
add("f4.react", f4["react"]["ms"], [("knn_dist_kernel<1>", 62976), ("lse_wave_kernel<4>", 125000)])
add("f4.ash_s", f4["ash_s"]["ms"], [("ash_s_kernel<32>", 125000)])
add("f4.gen", f4["gen"]["ms"], [("gen_kernel<16>", 125000)])
add("f4.pred_h_mi", f4["pred_h_mi"]["ms"], [("mcd_uncertainty_kernel<16>", 7813)])
g1, _ = med("gemm_rows_kernel<float, float, 1, 2, 4>", 8192)
rows.append(("f4.gmm_ddu (10 components)", f4["gmm_ddu"]["ms"], None if g1 is None else 10 * g1, "10 x the per-component launch"))
mt = stg["metrics"]
msum = sum((med(k, g)[0] or 0) for k, g in (("msd_probe_kernel<double>", 256), ("msd_lin_hist_kernel<double>", 245), ("msd_keys_kernel<double>", 489),
                                            ("msd_scatter_kernel<double>", 245), ("msd_bucket_sort_kernel<double>", 1024), ("tile_summary_kernel", 489),
                                            ("tile_prefix_raw_kernel", 489), ("curve_terms_finalize_kernel", 512)))
# the bucket sort has the same grid (1 024 workgroups) for 20 000 and for 2 M scores: take its launches that belong to 2 M-score calls
n_big = med("msd_scatter_kernel<double>", 245)[1]
sorts = sorted(d for (n, x, y, z), v in groups.items() if "msd_bucket_sort_kernel<double>" in n for d in v)
if n_big and len(sorts) >= n_big:
    msum += st.median(sorts[-n_big:]) - (med("msd_bucket_sort_kernel<double>", 1024)[0] or 0)
rows.append(("metrics.larem_f64_2m (8 launches)", mt["larem_f64_2m"]["ms"], msum, "sum of the eight kernels of a 2 M-score call"))

print("| leg | in-run ms (events) | trace ms (sum of kernel medians) | in-run / trace | note |")
print("|---|---|---|---|---|")
for label, a, b, note in rows:
    if label.startswith("stages.knn"):
        b = knn_ms / passes3
        note = "sum of every kNN kernel in the trace / 3 passes (1 warm-up + 2 timed)"
    ratio = "" if not b else f"{a / b:.3f}"
    print(f"| {label} | {a:.4f} | {'' if b is None else f'{b:.4f}'} | {ratio} | {note} |")
if other is not None:
    print()
    print(f"untraced run of the same command on the same box: value {other['value']:.4g} images/s, ms_per_step {other['ms_per_step']}, K1 "
          f"{other['roofline']['avg_launch_ms']} ms, clocks {other['clock_ghz_observed']['before']['ghz']} / {other['clock_ghz_observed']['after']['ghz']} GHz; "
          f"traced: value {bench['value']:.4g}, ms_per_step {bench['ms_per_step']}, K1 {bench['roofline']['avg_launch_ms']} ms")
