#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/profile_round.sh (gpurun_out/prof_<tag>/) into the summaries kept under profiles/.
usage: summarize_profiles.py <tag> <round-prefix>   e.g.  summarize_profiles.py r2b r2"""
import collections, csv, glob, json, os, shutil, sys

tag, pre = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", "prof_" + tag)
dst = "profiles"


def one(pattern):
    f = glob.glob(os.path.join(src, pattern))
    return f[0] if f else None


def counters(d):
    """{(dispatch, kernel): {counter: sum over the rows of that dispatch}} plus duration [ms] and launch shape."""
    f = one(d + "/*/*counter_collection.csv")
    out = collections.OrderedDict()
    if not f:
        return out
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("k_"):
            continue
        e = out.setdefault((int(r["Dispatch_Id"]), k), {"ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, "grid": int(r["Grid_Size"]),
                                                         "workgroup": int(r["Workgroup_Size"]), "vgpr": int(r["VGPR_Count"]), "accum_vgpr": int(r["Accum_VGPR_Count"]),
                                                         "sgpr": int(r["SGPR_Count"]), "lds": int(r["LDS_Block_Size"]), "scratch": int(r["Scratch_Size"])})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return out


# ---- kernel stats / trace of the default bench
f = one("stats/*/*kernel_stats.csv")
if f:
    rows = list(csv.reader(open(f)))
    with open(os.path.join(dst, pre + "_kernel_stats_default_bench.csv"), "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(rows[0])
        for r in rows[1:]:
            if r and r[0].startswith("k_"):
                w.writerow(r)
f = one("stats/*/*kernel_trace.csv")
if f:
    rd = csv.DictReader(open(f))
    keep = ["Kernel_Name", "Workgroup_Size_X", "Grid_Size_X", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Start_Timestamp", "End_Timestamp"]
    with open(os.path.join(dst, pre + "_kernel_trace_default_bench.csv"), "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(keep + ["Duration_ms"])
        for r in rd:
            if r["Kernel_Name"].startswith("k_"):
                w.writerow([r[k].split("(")[0] if k == "Kernel_Name" else r[k] for k in keep] + ["%.3f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)])
for mode in ("evolT", "sweep"):  # kernel stats of the two other bench lines
    f = one("stats_%s/*/*kernel_stats.csv" % mode)
    if f:
        rows = list(csv.reader(open(f)))
        with open(os.path.join(dst, pre + "_kernel_stats_%s_bench.csv" % mode), "w", newline="") as o:
            w = csv.writer(o)
            w.writerow(rows[0])
            for r in rows[1:]:
                if r and r[0].startswith("k_"):
                    w.writerow(r)
    p = os.path.join(src, "bench_stats_%s.json" % mode)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, "%s_bench_%s_under_rocprof_stats.json" % (pre, mode)))
for name in ("bench_stats", "bench_pmc_FETCH_SIZE", "bench_pmc_WRITE_SIZE", "bench_pmc_sq1"):
    p = os.path.join(src, name + ".json")
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, "%s_%s.json" % (pre, name.replace("bench_", "bench_under_rocprof_"))))

# ---- PMC passes of one launch
allc = collections.OrderedDict()
for d in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_sq1", "pmc_sq2", "pmc_tcc"):
    for (disp, k), e in counters(d).items():
        a = allc.setdefault(k, {})
        for c, v in e.items():
            if c in ("grid", "workgroup", "vgpr", "accum_vgpr", "sgpr", "lds", "scratch"):
                a[c] = v
            elif c == "ms":
                a.setdefault("ms_per_pass", []).append(v)
            else:
                a[c] = v
with open(os.path.join(dst, pre + "_pmc_k_solve_default_workload.csv"), "w", newline="") as o:
    w = csv.writer(o)
    w.writerow(["kernel", "counter", "value"])
    for k, a in allc.items():
        for c, v in a.items():
            w.writerow([k, c, " ".join("%.3f" % x for x in v) if isinstance(v, list) else v])

# ---- calibration of FETCH_SIZE / WRITE_SIZE with known byte counts
cal = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    known = json.load(open(os.path.join(src, "calib_%s.json" % c)))
    rows = list(counters("calib_" + c).items())
    names = ["k_calib_write", "k_calib_read_stream", "k_calib_read_reread", "k_calib_read_nt_stream", "k_calib_read_nt_reread"]
    cal[c] = {n: {"counter_KB": e.get(c, 0.0), "counter_bytes": e.get(c, 0.0) * 1024, "ms": e["ms"]} for n, ((d, k), e) in zip(names, rows)}
    cal["known_bytes"] = known
fetch_factor = cal["known_bytes"]["k_calib_read_stream_bytes"] / cal["FETCH_SIZE"]["k_calib_read_stream"]["counter_bytes"]
write_factor = cal["known_bytes"]["k_calib_write_bytes"] / cal["WRITE_SIZE"]["k_calib_write"]["counter_bytes"]
reread_factor = cal["known_bytes"]["k_calib_read_reread_bytes"] / cal["FETCH_SIZE"]["k_calib_read_reread"]["counter_bytes"]
nt_factor = (cal["known_bytes"]["k_calib_read_nt_stream_bytes"] / cal["FETCH_SIZE"]["k_calib_read_nt_stream"]["counter_bytes"]
             if "k_calib_read_nt_stream" in cal["FETCH_SIZE"] else None)
ks = {}  # the pass's solver kernels together: k_solve, then k_solve_team_resume (k_solve_team alongside when cost hints are in place)
for k, a in allc.items():
    if k.startswith("k_solve"):
        for c, v in a.items():
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                ks[c] = ks.get(c, 0.0) + v
bench = json.load(open(os.path.join(src, "bench_pmc_FETCH_SIZE.json")))
steps = bench["cell_steps_per_pass_rank0"]
hbm = ks.get("FETCH_SIZE", 0.0) * 1024 * fetch_factor + ks.get("WRITE_SIZE", 0.0) * 1024 * write_factor
out = {
    "launch": "bench.py --warmup 0 --steps 1 --no-cpu-baseline (default workload, one k_solve launch in queue order); rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes",
    "calibration": {"method": "tools/pmc_calib: 3072 waves, 8 B per lane raw buffer loads/stores (the solver's access pattern), known byte counts",
                    "fetch_factor_streamed_once": fetch_factor, "fetch_factor_reread_8x_1.2GB_footprint": reread_factor, "fetch_factor_nt_loads": nt_factor, "write_factor": write_factor, "raw": cal},
    "what_the_counters_see": "FETCH_SIZE / WRITE_SIZE count what crosses L2 towards the fabric: Infinity-Cache (MALL) hits as well as HBM -- fabric traffic, an upper bound of the HBM bytes",
    "FETCH_SIZE_KB": ks.get("FETCH_SIZE"), "WRITE_SIZE_KB": ks.get("WRITE_SIZE"), "cell_steps": steps,
    "fabric_bytes_corrected": hbm, "hbm_bytes_corrected": hbm, "algorithmic_bytes": bench["roofline"]["algorithmic_bytes_per_launch"],
    "bytes_per_cell_step_corrected": hbm / steps, "ratio_to_algorithmic": hbm / bench["roofline"]["algorithmic_bytes_per_launch"],
    "kernel_ms_under_profiler": ks.get("ms_per_pass"),
    "workload": {"name": "grid", "network": "rate06_dipole_reformated_again_withgrain_lowH2Bind.dat", "cells_per_gpu": bench["config"]["cells_per_gpu"]},
}
json.dump(out, open(os.path.join(dst, pre + "_pmc_calibration.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("bytes_per_cell_step_corrected", "ratio_to_algorithmic", "hbm_bytes_corrected")}))
if "TCC_HIT_sum" in ks:
    print("L2 hit rate", ks["TCC_HIT_sum"] / (ks["TCC_HIT_sum"] + ks["TCC_MISS_sum"]))
if "SQ_WAVE_CYCLES" in ks:
    wc = ks["SQ_WAVE_CYCLES"]
    print("wait %.3f active %.3f issue-stall %.3f; instr per cell-step: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.0f VMEM_WR %.0f SMEM %.0f" % (
        ks["SQ_WAIT_ANY"] / wc, ks["SQ_ACTIVE_INST_ANY"] / wc, ks["SQ_WAIT_INST_ANY"] / wc, ks["SQ_INSTS_VALU"] / steps, ks["SQ_INSTS_SALU"] / steps,
        ks["SQ_INSTS_LDS"] / steps, ks.get("SQ_INSTS_VMEM_RD", 0) / steps, ks.get("SQ_INSTS_VMEM_WR", 0) / steps, ks.get("SQ_INSTS_SMEM", 0) / steps))
