#!/usr/bin/env python3
"""bench.py -- cell-steps/s of the batched chemistry solve on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 1 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (rate coefficients + tolerance policy + the whole chem_evol_solve
integration of every cell to its t_max) over one batch of synthetic cell records.

Workloads (--workload):
  grid      (default) BASELINE.json configs[2]: the full synthetic Andrews-2009 grid, 200 columns x 100 cells = 20 000 cell
            records (rac-2d_amd/cells.py::andrews_grid: n_H 1e3..6e12 cm^-3, T 8..5000 K, Av 1e-4..1e5, per-cell t_max by the
            reference's orbit rule), network rate06_dipole_reformated_again_withgrain_lowH2Bind.dat (467 species, 4801
            reactions), ini_abund_waterice_loMetal.dat, template solver settings (RTOL 1e-4, ATOL 1e-30, t_max0 1e6 yr,
            dt_first_step 1e-8, ratio 1.1, steps_reset_solver 50).  --network default switches to the README-default
            5830-reaction file.
  synth10k  BASELINE.json configs[1]: 10 000 log-uniform random cells, rate06 "withoutgrain" network (round 1's line).

Multi-GPU (--scaling):
  weak      (default) every rank solves a whole grid of its own (rank r: the same disk with the gas mass scaled by 1 + r/16,
            so no two ranks hold the same cells); per-GPU work is fixed as N grows.
  strong    BASELINE.json configs[3]: ONE grid, cells dealt round-robin by expected cost over the ranks
            (rac-2d_amd/sweep.py), total work fixed.
Either way the only exchange is ONE RCCL all-gather of the end-state abundances, t_final, quality and counters, inside the
timed region.  Inputs are resident in HBM before the clock starts.

Scheduling: as between two global iterations of the disk model, every pass hands the per-cell cycle counts it measured to
the next one (racgpu_set_cost_hints), which then starts the costliest cells first; the hand-over is inside the timed region.
The first warm-up pass has no history and runs in queue order: its rate is reported as config.queue_order_first_pass.

metric value = (accepted integrator steps summed over all cells and ranks) / (max over ranks of wall time).
The line also carries: roofline (algorithmic bytes / k_solve time, HIP events on the launch stream), cpu_baseline (the
reference's own Fortran binary on a bounded sample of the same cells, all host cores) and parity (the GPU's end state of
those sample cells against the reference's: the second half of BASELINE.json's metric).
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
DATA = os.path.join(ROOT, "data")
NETWORKS = {
    "grain": "rate06_dipole_reformated_again_withgrain_lowH2Bind.dat",
    "default": "rate06_withgrain_lowH2Bind_hiOBind_lowCObind.dat",
    "nograin": "rate06_dipole_reformated_again_withoutgrain.dat",
    "rate12": "rate12_withGrain_lowH2Bind_hiObind.dat",
}
INITIAL = {"default": "ini_abund_waterice_loMetal_CO.dat"}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes(nS, nR, nnzJ, nzl, nzu, nst, nfe, nje, nlu, qsum):
    """B_step = 8*[a(nnzLU+NEQ) + a(nR+2NEQ) + b(nnzJ+nnzLU) + c(nR+nnzJ) + 4(q+1)NEQ + 6NEQ]
    summed over steps, with a, b, c, q from this run's own counters (a*NST = NFE etc.)."""
    neq = nS + 1
    nnzlu = nzl + nzu + neq
    return 8.0 * (nfe * (nnzlu + neq) + nfe * (nR + 2 * neq) + nlu * (nnzJ + nnzlu) + nje * (nR + nnzJ)
                  + 4.0 * (qsum + nst) * neq + 6.0 * nst * neq)


def algorithmic_bytes_evolT(nS, nR, nnzJ, nzl, nzu, nst, nfe, nje, nlu, qsum):
    """The same with the gas temperature co-evolving: every f(y) also WRITES the rate vector it recomputes at the
    iterate's T (+nR per f); a Jacobian costs two more f(y, T + dT) for the T column (+2(nR + NEQ)) and writes the T column of P;
    a factorisation rescales/reads the T column and writes z = A^-1 b (+3 NEQ); a solve reads z (+NEQ)."""
    neq = nS + 1
    nnzlu = nzl + nzu + neq
    return 8.0 * (nfe * (nnzlu + 2 * neq) + nfe * (2 * nR + 2 * neq) + nlu * (nnzJ + nnzlu + 3 * neq) + nje * (3 * nR + nnzJ + 4 * neq)
                  + 4.0 * (qsum + nst) * neq + 6.0 * nst * neq)


def _read_sections(fn):
    d, cur = {}, None
    for line in open(fn):
        if line.startswith("#"):
            cur = line.split()[1]
            d[cur] = []
        elif cur is not None:
            d[cur].append(float(line))
    return {k: np.array(v) for k, v in d.items()}


def _error_codes(logfile, ncell):
    """[ncell, 4]: how often the reference's error handler (ode_solver_error_handling) saw ISTATE -1, -4, -5, anything else, per cell
    (ref_driver writes '# cell k' into its log before each cell, the handler '!Error: <ISTATE>' per error return)."""
    out = np.zeros((ncell, 4), dtype=np.int64)
    cur = -1
    try:
        for line in open(logfile, errors="replace"):
            if line.startswith("# cell"):
                cur = int(line.split()[2]) - 1
            elif line.startswith("!Error:") and 0 <= cur < ncell:
                p = line.split()
                if len(p) == 2 and p[1].lstrip("-").isdigit():
                    out[cur, {-1: 0, -4: 1, -5: 2}.get(int(p[1]), 3)] += 1
    except OSError:
        pass
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def run_reference(sample, network, initial, params, rtol=None, hc=None):
    """oracle/_ref/ref_driver on the rows of `sample`, one process per host core; returns (list of section dicts, seconds, cores)
    or (None, 0, cores) when the binary is absent or fails.  Every dict also carries "errcodes": the cell's error returns by ISTATE
    code (-1, -4, -5, other)."""
    cores = min(os.cpu_count() or 1, 16)
    driver = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if not os.path.exists(driver):
        return None, 0.0, cores
    with tempfile.TemporaryDirectory() as td:
        procs, dirs = [], []
        t0 = time.perf_counter()
        for w in range(cores):
            part = sample[w::cores]
            if len(part) == 0:
                continue
            d = os.path.join(td, "w%d" % w)
            os.makedirs(d)
            dirs.append((w, d, len(part)))
            np.savetxt(os.path.join(d, "cells.txt"), part, fmt="%.17e")
            extra = ""
            if hc is not None:  # gas temperature co-evolving: the cells' heating/cooling records; tables and enthalpies from data/
                np.savetxt(os.path.join(d, "hc.txt"), hc[w::cores], fmt="%.17e")
                extra = " evolT=1\n hc_file='%s'\n enthalpy='Species_enthalpy.dat'\n transitions_dir='%s/'\n" % (os.path.join(d, "hc.txt"), DATA)
            with open(os.path.join(d, "run.nml"), "w") as f:
                f.write("&ref_run\n chem_dir='%s/'\n network='%s'\n initial='%s'\n out_dir='%s'\n cell_file='%s'\n ncell=%d\n"
                        " rtol=%.17e\n atol=%.17e\n dt_first_step=%.17e\n ratio_tstep=%.17e\n t_max=%.17e\n mxstep=%d\n"
                        " steps_reset=%d\n dump_jac=0\n solve=1\n" % (
                            DATA, network, initial, d, os.path.join(d, "cells.txt"), len(part), rtol or params.RTOL, params.ATOL,
                            params.dt_first_step, params.ratio_tstep, params.t_max, params.mxstep_per_interval,
                            params.steps_reset_solver) + extra + "/\n")
            procs.append(subprocess.Popen([driver, os.path.join(d, "run.nml")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
        ok = all(p.wait() == 0 for p in procs)
        dt = time.perf_counter() - t0
        if not ok:
            return None, dt, cores
        ref = {}
        for w, d, m in dirs:
            ec = _error_codes(os.path.join(d, "ref_log.txt"), m)
            for k in range(m):
                ref[w + k * cores] = _read_sections(os.path.join(d, "cell_%04d.txt" % (k + 1)))
                ref[w + k * cores]["errcodes"] = ec[k]
    return [ref[k] for k in range(len(sample))], dt, cores


def cpu_baseline_and_parity(cells, sample_idx, network, initial, params, gpu, nS, tight=None, hc=None):
    """The reference's own Fortran path (oracle/_ref/ref_driver, built from the unmodified sources in the build container)
    on a bounded sample of the same cells, one process per host core.  Returns (cpu_baseline, parity).  Falls back to the
    C restatement (kind "port", single thread).  Baseline and checker only: nothing here is on the product path."""
    sample = cells[sample_idx]
    nsample = len(sample)
    hcs = None if hc is None else hc[sample_idx]
    ref, dt, cores = run_reference(sample, network, initial, params, hc=hcs)
    if ref is not None:
        steps = float(np.sum(gpu["nst"][sample_idx]))
        base = {"value": steps / dt, "unit": "cell-steps/s", "cores": cores, "kind": "reference", "cpu_model": cpu_model(),
                "sample": "%d cells of the same batch (every %d-th), reference Fortran/DLSODES binary, %d processes, %.1f s wall; "
                          "steps counted with the GPU run's NST for the same cells (DLSODES zeroes its own counter at every solver reset)"
                          % (nsample, max(1, len(cells) // nsample), min(cores, nsample), dt)}
        # The reference's own rounding-noise floor on THESE cells: the same binary on the same cells with n_gas moved by one ulp
        # (the recipe of tests/golden/make_golden.py).  A cell's floor = how far that moves the reference's own end state.
        # Three twins (n_gas up and down by one ulp, Tgas up by one ulp): one sample of a chaotic quantity underestimates it.
        twins = []
        for col, up in ((2, True), (2, False), (0, True)):
            twin_cells = sample.copy()
            twin_cells[:, col] = np.nextafter(twin_cells[:, col], np.inf if up else 0.0)
            twin_cells[:, 5] = twin_cells[:, 2] * twin_cells[:, 6]
            thc = None
            if hcs is not None:
                thc = hcs.copy(); thc[:, 16] = twin_cells[:, 5]  # (n_dusts of the single dust component = ndust_tot)
            tw, _, _ = run_reference(twin_cells, network, initial, params, hc=thc)
            if tw is not None:
                twins.append(tw)
        twin = twins[0] if twins else None

        def worst(yg, yr):  # (with T evolving the last entry of the reference's y is the temperature: it counts like a major species)
            n = nS + 1 if hcs is not None else nS
            yg, yr = np.asarray(yg)[:n], np.asarray(yr)[:n]
            m = yr >= 1e-6
            e = np.zeros_like(yr)
            e[m] = np.abs(yg[m] - yr[m]) / yr[m]
            return float(e.max()), int(e.argmax())

        def ref_freeze_rec(r):  # the record after which the reference's T stays put (0: T evolved to the end): RACGPU_O_TFREEZE_REC's twin
            if "Trecord" not in r or int(r["evolTend"][0]) != 0:
                return 0
            T = np.asarray(r["Trecord"])
            k = len(T) - 1
            while k > 0 and T[k - 1] == T[k]:
                k -= 1
            return k + 1

        frz = []
        errs, floors, spec, tf_eq, q_eq, ne_eq = [], [], [], 0, 0, 0
        codes_ref = np.zeros(4, dtype=np.int64); codes_twin = np.zeros(4, dtype=np.int64); codes_gpu = np.zeros(4, dtype=np.int64)
        ne_ref_twin_eq = 0
        for k in range(nsample):
            r = ref[k]
            yr = r["yend"]
            yg = gpu["y"][sample_idx[k]] if hcs is None else np.r_[gpu["y"][sample_idx[k]], gpu["tgas"][sample_idx[k]]]
            e, sp = worst(yg, yr)
            errs.append(e); spec.append(sp)
            floors.append(max([worst(tw[k]["yend"], yr)[0] for tw in twins]) if twins else 0.0)
            if hcs is not None:
                frz.append((int(gpu["freeze_rec"][sample_idx[k]]), ref_freeze_rec(r)))
            tf_eq += int(r["scalars"][0] == gpu["t_final"][sample_idx[k]])
            q_eq += int(int(r["scalars"][1]) == int(gpu["quality"][sample_idx[k]]))
            ne_eq += int(int(r["scalars"][2]) == int(gpu["nerr"][sample_idx[k]]))
            codes_ref += r["errcodes"]
            ec = int(gpu["errcodes"][sample_idx[k]])
            codes_gpu += np.array([(ec >> s) & 0xffff for s in (0, 16, 32, 48)])
            if twin is not None:
                codes_twin += twin[k]["errcodes"]
                ne_ref_twin_eq += int(int(r["scalars"][2]) == int(twin[k]["scalars"][2]))
        errs = np.array(errs); floors = np.array(floors)
        # cells the engine's modelled run-time guard ended before t_max while the reference (guard off) went on: nothing to compare
        guard = np.array([(int(gpu["quality"][sample_idx[k]]) & 2) != 0 and ref[k]["scalars"][0] > gpu["t_final"][sample_idx[k]] for k in range(nsample)])
        errs = np.where(guard, 0.0, errs)
        bound = np.maximum(1e-4, 3.0 * floors)
        excess = errs / bound
        kw = int(np.argmax(excess))
        names = ("ISTATE -1", "ISTATE -4", "ISTATE -5", "other")
        parity = {"against": "reference Fortran/DLSODES end states of the cpu_baseline sample, same RTOL (%g); species with X >= 1e-6" % params.RTOL,
                  "cells": nsample, "max_rel_err": float(errs.max()), "median_rel_err": float(np.median(errs)),
                  "p90_rel_err": float(np.percentile(errs, 90)), "cells_within_1e-4": int((errs <= 1e-4).sum()),
                  # per-cell floor = the largest move of the reference's own end state under three one-ulp changes of its inputs (n_gas up,
                  # n_gas down, Tgas up), same cells, same settings
                  "floor": {"median": float(np.median(floors)), "p90": float(np.percentile(floors, 90)), "max": float(floors.max()),
                            "cells_reference_moves_more_than_1e-4": int((floors > 1e-4).sum())},
                  "cells_within_max(1e-4,3*floor)": int((errs <= bound).sum()),
                  "worst_cell": {"cell": int(sample_idx[kw]), "species_index": spec[kw], "err": float(errs[kw]), "floor": float(floors[kw]),
                                 "err_over_bound": float(excess[kw])},
                  # (a cell where either side had an error return runs on with RTOL of the offending species raised tenfold, up to 1e-3, by the
                  # reference's own policy: the error returns of the GPU, the reference and its twins are listed with each exception)
                  "exceptions": [dict({"cell": int(sample_idx[k]), "species_index": spec[k], "err": float(errs[k]), "floor": float(floors[k]),
                                       "error_returns": {"gpu": int(gpu["nerr"][sample_idx[k]]), "reference": int(ref[k]["scalars"][2]),
                                                         "reference_twins": [int(tw[k]["scalars"][2]) for tw in twins]}},
                                      **({"T_freeze_record": {"gpu": frz[k][0], "reference": frz[k][1]}} if frz else {}))
                                 for k in np.nonzero(errs > bound)[0][:16]],
                  "ended_early_by_the_modelled_run_time_guard": {"cells": [int(sample_idx[k]) for k in np.nonzero(guard)[0]],
                                                                 "note": "GPU quality bit 2 and t_final below the reference's, whose guard is off; left out of the error statistics"},
                  "t_final_equal": tf_eq, "quality_equal": q_eq,
                  # error returns (ISTATE < 0) of the integrator, by code, on the three sides: GPU, reference, reference's 1-ulp twin
                  "nerr_equal": ne_eq, "nerr_equal_reference_vs_its_twin": ne_ref_twin_eq,
                  "nerr_by_code": {"gpu": dict(zip(names, map(int, codes_gpu))), "reference": dict(zip(names, map(int, codes_ref))),
                                   "reference_twin": dict(zip(names, map(int, codes_twin)))},
                  "nerr_total_reference": int(codes_ref.sum()), "nerr_total_reference_twin": int(codes_twin.sum()), "nerr_total_gpu": int(codes_gpu.sum())}
        if frz:  # evolT: the T-freeze test (src/chemistry.f90:532-546) is a threshold inside the integrator's noise; see RACGPU_O_TFREEZE_REC
            same = np.array([a == b for a, b in frz])
            twin_same = int(sum(ref_freeze_rec(ref[k]) == ref_freeze_rec(twin[k]) for k in range(nsample))) if twin is not None else None
            parity["T_freeze"] = {"cells_frozen_at_the_same_record": int(same.sum()), "reference_vs_its_twin_same_record": twin_same, "cells_frozen_at_different_records": int((~same).sum()),
                                  "max_rel_err_where_same": float(errs[same].max()) if same.any() else None,
                                  "cells_within_bound_where_same": int((errs[same] <= bound[same]).sum()),
                                  "max_rel_err_where_different": float(errs[~same].max()) if (~same).any() else None}
        if tight is not None and hcs is None:  # the same comparison where trajectory noise does not limit it: RTOL 1e-8 on both sides
            tidx, gy = tight
            tref, tdt, _ = run_reference(cells[tidx], network, initial, params, rtol=1e-8)
            tref10, _, _ = run_reference(cells[tidx], network, initial, params, rtol=1e-10)  # how converged is the reference's own 1e-8 answer
            if tref is not None:
                te, ts, tf = [], [], []
                for k in range(len(tidx)):
                    e, sp = worst(gy[k], tref[k]["yend"][:nS])
                    te.append(e); ts.append(sp)
                    tf.append(worst(tref[k]["yend"][:nS], tref10[k]["yend"][:nS])[0] if tref10 is not None else 0.0)
                kt = int(np.argmax(te))
                parity["tight"] = {"rtol": 1e-8, "cells": len(tidx), "max_rel_err": float(np.max(te)), "median_rel_err": float(np.median(te)),
                                   "worst_cell": {"cell": int(tidx[kt]), "species_index": ts[kt], "err": float(te[kt]),
                                                  "reference_1e-8_vs_1e-10": float(tf[kt])},
                                   "reference_1e-8_vs_1e-10_max": float(np.max(tf)), "reference_seconds": tdt}
        return base, parity
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_ctypes as O
    onet = O.Network(os.path.join(DATA, network))
    y0 = onet.initial_abundances(os.path.join(DATA, initial))
    op = O.default_params()
    for f in ("RTOL", "ATOL", "t_max", "dt_first_step", "ratio_tstep", "mxstep_per_interval", "steps_reset_solver"):
        setattr(op, f, getattr(params, f))
    t0 = time.perf_counter()
    steps, n, errs = 0, 0, []
    for k in range(nsample):
        o = onet.solve_cell(op, sample[k], y0)
        steps += o["nst"]
        n += 1
        yr = o["y"][:nS]; m = yr >= 1e-6
        errs.append(float(np.max(np.abs(gpu["y"][sample_idx[k]][m] - yr[m]) / yr[m])))
        if time.perf_counter() - t0 > 30.0:
            break
    dt = time.perf_counter() - t0
    base = {"value": steps / dt, "unit": "cell-steps/s", "cores": 1, "kind": "port",
            "sample": "%d cells of the same batch, C restatement (oracle/), single thread, %.1f s" % (n, dt)}
    parity = {"against": "C restatement (oracle/) end states, species with X >= 1e-6", "cells": n, "max_rel_err": float(np.max(errs)),
              "median_rel_err": float(np.median(errs))}
    return base, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=("grid", "synth10k"), default="grid")
    ap.add_argument("--network", choices=sorted(NETWORKS), default=None, help="default: grain for grid, nograin for synth10k")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--cells", type=int, default=0, help="synth10k: cells per GPU (default 10000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hints", action="store_true", help="take cells in queue order in every pass (no cost feedback)")
    ap.add_argument("--evolT", action="store_true", help="gas temperature co-evolving with the chemistry (chemsol_params%evolT, the reference's "
                    "production default): every cell gets a heating/cooling record (cells.andrews_grid_hc); one wave per cell (k_solve_T)")
    ap.add_argument("--sweep", choices=("batch", "columns"), default="batch",
                    help="batch (default): every cell with its record as given, all cells at once -- frozen shielding, a Jacobi relaxation over "
                    "the global iterations.  columns: the reference's dependency order (src/disk.f90:885-936) on the device (racgpu_column_sweep): "
                    "a cell after the cell above it and the cell on its ray to the star, its toISM and toStar shielding slots rewritten from "
                    "what those ended with; one four-wave team per column")
    ap.add_argument("--nlocal-iter", type=int, default=1, help="> 1: the caller's local-iteration loop (racgpu_calc_cells) instead of one chem_evol_solve pass")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the racgpu path has no CPU fallback")
    # RACGPU_BENCH_REHEARSAL=1 (developer aid): rehearse the N > 1 code path on a box with ONE GPU -- every rank computes on cuda:0
    # and the exchange goes through gloo on host copies.  Never a measurement.
    rehearsal = bool(os.environ.get("RACGPU_BENCH_REHEARSAL"))
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    R = importlib.import_module("rac-2d_amd")
    sweep = importlib.import_module("rac-2d_amd.sweep")
    R.set_device(local_rank)
    netkey = args.network or ("grain" if args.workload == "grid" else "nograin")
    network = NETWORKS[netkey]
    initial = INITIAL.get(netkey, "ini_abund_waterice_loMetal.dat")
    net = R.Network(os.path.join(DATA, network))
    nS = net.nSpecies
    y0 = net.load_initial_abundances(os.path.join(DATA, initial))
    params = R.default_params()
    # chemsol_params%max_runtime_allowed (60 s in the reference's template) is a guard on the CPU WALL CLOCK of a cell in the reference
    # (src/chemistry.f90:480-491); the engine models it from its counters (racgpu_params rt_cost_*).  It stays ON: with T evolving a
    # few cells chatter on a jump of a rate coefficient for minutes (evolT), which is what the guard is for.  The reference
    # runs of cpu_baseline/parity have it off (their clock is this host's), so cells it ended on the GPU are listed, not compared.
    if netkey == "rate12":
        params.RTOL = 1e-6; params.t_max = 1e7  # BASELINE.json configs[4]
    hc_h = None
    if args.workload == "grid":
        if args.scaling == "strong" and world > 1:
            full = R.cells.andrews_grid()
            # expected cost: denser cells take more steps; deal round-robin by it so every rank gets the same mix
            order = sweep.interleaved_order(full[:, R.cells.P_NGAS], world)
            lo, hi = sweep.partition(len(full), world, rank)
            cells_h = np.ascontiguousarray(full[order[lo:hi]])
        else:
            cells_h = R.cells.andrews_grid(Md=2e-2 * (1.0 + rank / 16.0))
        if args.evolT:
            if args.scaling == "strong" and world > 1:
                raise SystemExit("--evolT: weak scaling only")
            cells_h, r_au, z_au = R.cells.andrews_grid(Md=2e-2 * (1.0 + rank / 16.0), return_geometry=True)
            hc_h = R.cells.andrews_grid_hc(cells_h, r_au, z_au)
            net.load_heating_cooling(DATA)
        wl = ("configs[2]: full synthetic Andrews-2009 grid, 200 columns x 100 cells = 20000 cell records (n_H 1e3..6e12 cm^-3, "
              "T 8..5000 K, per-cell t_max by the orbit rule), %s network (%s: %d species, %d reactions), %s, t_max0=%g yr, RTOL=%g, "
              "steps_reset_solver=50; every cell with the shielding factors of its record as given, i.e. frozen shielding (Jacobi relaxation "
              "over global iterations), all cells in one batch" % (netkey, network, nS, net.nReactions, initial, params.t_max, params.RTOL))
    else:
        ncell0 = args.cells or 10000
        cells_h = R.cells.synth_batch(ncell0, seed=20240601 + rank)
        wl = ("configs[1]: %d synthetic cells per GPU (log-uniform T in [10,3000] K, n_H in [1e3,1e12] cm^-3), %s network (%s: %d species, "
              "%d reactions), %s, t_max=%g yr, RTOL=%g, steps_reset_solver=50" % (ncell0, netkey, network, nS, net.nReactions, initial, params.t_max, params.RTOL))
    if args.sweep == "columns" and (args.workload != "grid" or args.evolT or args.nlocal_iter > 1 or world > 1):
        raise SystemExit("--sweep columns: grid workload, fixed T, one pass per step, one GPU (the wavefront runs across all columns)")
    colgrid = None
    if args.sweep == "columns":
        colgrid = R.cells.andrews_columns()
        net.set_co_shielding_table(R.cells.load_co_shielding_table(os.path.join(DATA, "visser2009_co_shielding.dat")))
        net.set_star_rays(colgrid["inner"], colgrid["ds"])
        wl = wl.replace("every cell with the shielding factors of its record as given, i.e. frozen shielding (Jacobi relaxation over global iterations), "
                        "all cells in one batch",
                        "DEPENDENCY-ORDER SWEEP (the reference's own order, src/disk.f90:885-936, as a wavefront on the device: "
                        "racgpu_column_sweep + racgpu_set_star_rays): a cell is solved after the cell above it and the cell on its ray to the "
                        "star, its toISM and toStar self-shielding slots (H2, CO on the Visser 2009 table, H2O, OH) rewritten from what they ended "
                        "with (update_params_above_alt, :1823-1883); one four-wave team per column, 200 columns")
        assert "DEPENDENCY-ORDER" in wl
    if args.evolT and args.workload != "grid":
        raise SystemExit("--evolT goes with the grid workload")
    if args.evolT:
        wl += ("; GAS TEMPERATURE CO-EVOLVING (evolT) in every cell: heating/cooling records by rac-2d_amd/cells.py::andrews_grid_hc, switches of the "
               "reference's template (README.md:135-156)")
    ncell = len(cells_h)
    yinit_h = net.init_abundances(y0, cells_h)

    dev = torch.device("cuda", local_rank)
    cells_d = torch.from_numpy(cells_h).to(dev)
    yinit_d = torch.from_numpy(yinit_h).to(dev)
    # one flat result block per rank: [y | t_final | quality | stats] as f64, so that ONE collective moves everything
    ncol = nS + 2 + R.NSTAT
    y_d = torch.empty_like(yinit_d)
    tfin_d = torch.zeros(ncell, dtype=torch.float64, device=dev)
    qual_d = torch.zeros(ncell, dtype=torch.int32, device=dev)
    stats_d = torch.zeros((ncell, R.NSTAT), dtype=torch.int64, device=dev)
    hc_d = torch.from_numpy(hc_h).to(dev) if args.evolT else None
    cout_d = torch.zeros((ncell, R.NOUT), dtype=torch.float64, device=dev)
    maxn = ncell
    if world > 1:
        t = torch.tensor([ncell], dtype=torch.int64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        maxn = int(t.item())
    block_d = torch.zeros((maxn, ncol), dtype=torch.float64, device=dev)
    gathered = torch.empty((world * maxn, ncol), dtype=torch.float64, device=dev) if world > 1 else None
    stream = torch.cuda.current_stream(dev)
    net.set_stream(stream.cuda_stream)

    kernel_ms = []
    teams = [0, 0]  # cells of the last pass solved by four-wave teams from the start / handed over to teams at the end of the pass

    sweep_cells = [None]

    def one_pass():
        y_d.copy_(yinit_d)
        if colgrid is not None:  # host-buffer entry (the column order is checked on the host); the copies are part of the pass
            o = net.column_sweep(params, cells_h, yinit_h, colgrid["col_ptr"], colgrid["col_cells"], colgrid["dz"], dv_turb=1e5)
            y_d.copy_(torch.from_numpy(o["y"])); tfin_d.copy_(torch.from_numpy(o["t_final"])); qual_d.copy_(torch.from_numpy(o["quality"]))
            stats_d.copy_(torch.from_numpy(o["stats"])); cout_d.copy_(torch.from_numpy(o["cell_out"]))
            sweep_cells[0] = o["cells"]
        elif args.evolT:
            net.evolT_solve_batch_device(params, ncell, cells_d.data_ptr(), hc_d.data_ptr(), y_d.data_ptr(), tfin_d.data_ptr(), qual_d.data_ptr(),
                                         stats_d.data_ptr(), cout_d.data_ptr())
        elif args.nlocal_iter > 1:
            net.calc_cells_device(params, args.nlocal_iter, ncell, cells_d.data_ptr(), y_d.data_ptr(), tfin_d.data_ptr(), qual_d.data_ptr(), stats_d.data_ptr())
        else:
            net.evol_solve_batch_device(params, ncell, cells_d.data_ptr(), y_d.data_ptr(), tfin_d.data_ptr(), qual_d.data_ptr(), stats_d.data_ptr())
        if world > 1:
            block_d[:ncell, :nS] = y_d
            block_d[:ncell, nS] = tfin_d
            block_d[:ncell, nS + 1] = qual_d.to(torch.float64)
            block_d[:ncell, nS + 2:] = stats_d.to(torch.float64)
            if rehearsal:
                gc = torch.empty(gathered.shape, dtype=torch.float64)
                dist.all_gather_into_tensor(gc, block_d.cpu())
                gathered.copy_(gc)
            else:
                dist.all_gather_into_tensor(gathered, block_d)  # the path's single exchange: RCCL over xGMI
        torch.cuda.synchronize(dev)
        kernel_ms.append(net.last_kernel_ms())
        teams[0], teams[1] = net.last_team_cells(), net.last_parked_cells()
        if not args.no_hints and colgrid is None:
            # cost feedback, as between two global iterations of the disk model: the cycles each cell took in this pass
            # order the next pass (costliest first).  Part of the pass, so it is inside the timed region.
            net.set_cost_hints(stats_d[:, R.S_CYC_TOTAL].cpu().numpy().astype(np.float64))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    first_pass_s = None
    for w in range(args.warmup):
        tw = time.perf_counter()
        one_pass()
        if w == 0:
            first_pass_s = time.perf_counter() - tw  # the only pass that runs without cost hints
    hinted = (not args.no_hints) and args.warmup > 0
    kernel_ms.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    barrier()
    dt = time.perf_counter() - t0

    stats = stats_d.cpu().numpy()
    qual = qual_d.cpu().numpy()
    local = torch.tensor([dt, float(stats[:, 0].sum()) * args.steps], dtype=torch.float64, device="cpu" if rehearsal else dev)
    if world > 1:
        tmax = local[0:1].clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        ssum = local[1:2].clone(); dist.all_reduce(ssum, op=dist.ReduceOp.SUM)
        t_all, steps_all = float(tmax.item()), float(ssum.item())
    else:
        t_all, steps_all = dt, float(local[1].item())

    if rank == 0:
        nst, nfe, nje, nlu, qsum = [float(stats[:, k].sum()) for k in (0, 1, 2, 3, 6)]
        abytes = (algorithmic_bytes_evolT if args.evolT else algorithmic_bytes)(nS, net.nReactions, net.nnzJ, net.nzl, net.nzu, nst, nfe, nje, nlu, qsum)
        kms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
        achieved = abytes / (kms * 1e-3) / 1e9
        # HBM bytes per launch.  PMC counters cannot be read from inside this process; the figure is the per-cell-step traffic
        # of the committed rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this same workload (profiles/r2_pmc_calibration.json,
        # FETCH_SIZE corrected by the factor measured with a micro-kernel of known bytes and the same 8 B/lane buffer loads)
        # times this launch's cell-steps: "calibrated, not measured".
        traffic, traffic_src = None, None
        for calname in ("r3_pmc_calibration.json", "r2_pmc_calibration.json"):
            try:
                cal = json.load(open(os.path.join(ROOT, "profiles", calname)))
                if cal["workload"]["name"] == args.workload and cal["workload"]["network"] == network and not args.evolT and colgrid is None:
                    traffic = cal["bytes_per_cell_step_corrected"] * nst
                    traffic_src = ("calibrated, not measured in this run: profiles/%s, %.0f B per cell-step x %d cell-steps; TCC FETCH/WRITE "
                                   "counters see what crosses L2, i.e. Infinity-Cache (MALL) hits as well as HBM" % (calname, cal["bytes_per_cell_step_corrected"], int(nst)))
                    break
            except Exception:
                pass
        cyc = stats[:, R.S_CYC_TOTAL].astype(np.float64)
        out = {
            "metric": "cell-steps/s (whole node)", "value": steps_all / t_all, "unit": "cell-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * t_all / args.steps,
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" if not rehearsal else "synthetic; REHEARSAL of the multi-rank path on one GPU over gloo: not a measurement",
            "config": {"workload": wl, "cells_per_gpu": ncell,
                       "parallelism": ("one GPU: the wavefront of the sweep runs across all columns (200 four-wave teams, at most 100 of them busy)" if colgrid is not None else
                                       "cells sharded over %d GPU(s) (%s), one RCCL all-gather of abundances + t_final + quality + counters at output" % (world, args.scaling)),
                       "local_iterations": args.nlocal_iter,
                       # cells the last pass gave to four-wave teams: from the start (cost hints; racgpu_set_team_threshold) / between
                       # two integrator steps once the queue was empty and at most two waves per CU were left
                       "cells_in_teams": {"from_start": teams[0], "handed_over": teams[1]},
                       "scheduling": ("costliest-first from the previous pass's per-cell cycle counts (racgpu_set_cost_hints)" if hinted
                                      else "queue order (no previous pass to take cost hints from)"),
                       # rank 0's first warm-up pass runs in queue order: its rate is the no-feedback figure
                       "queue_order_first_pass": ({"ms": 1e3 * first_pass_s, "cell_steps_per_s_rank0": float(stats[:, 0].sum()) / first_pass_s}
                                                  if (first_pass_s and hinted) else None)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         # achieved / frac are ALGORITHMIC bytes over kernel time; traffic is what the counters saw crossing L2 for the same
                         # launch, traffic_rate / traffic_frac_of_peak the same divided by the same kernel time
                         "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_rate": (traffic / (kms * 1e-3) / 1e9) if traffic else None,
                         "traffic_frac_of_peak": (traffic / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         # one pass = k_solve (one wave per cell) followed by k_solve_team_resume (the cells handed over at its end),
                         # with k_solve_team (cells in teams from the start) alongside: timed as a whole between two HIP events
                         "kernel": "k_solve_T (+ k_solve_team_T)" if args.evolT else "k_solve_columns" if colgrid is not None else "k_solve (+ k_solve_team, k_solve_team_resume)", "kernel_ms": kms,
                         "algorithmic_bytes_per_launch": abytes,
                         "bytes_per_cell_step": abytes / max(nst, 1.0)},
            "cell_steps_per_pass_rank0": nst, "mean_steps_per_cell": nst / ncell,
            "nfe_per_step": nfe / max(nst, 1), "nlu_per_step": nlu / max(nst, 1), "nje_per_step": nje / max(nst, 1),
            "mean_order": qsum / max(nst, 1), "cells_with_quality_flags": int((qual != 0).sum()),
            # share of each wave's shader-clock cycles per phase (in-kernel s_memtime brackets, summed over cells)
            "phase_cycle_share": {k: float(stats[:, i].sum()) / max(float(cyc.sum()), 1.0)
                                  for k, i in (("rhs", 9), ("jacobian", 10), ("lu", 11), ("tri_solve", 12),
                                               ("lu_scatter", 13), ("lu_lds_pivots", 14), ("lu_reg_pivots", 15))},
            "wave_cycles_per_cell_step": float(cyc.sum()) / max(nst, 1.0),
            "costliest_cell_over_mean": float(cyc.max() / max(cyc.mean(), 1.0)),
        }
        if world == 1 and not args.no_cpu_baseline:
            cores = min(os.cpu_count() or 1, 16)
            nsample = min(ncell, 16 * cores)
            sample_idx = np.arange(nsample) * (ncell // nsample) + (ncell // nsample) // 2  # spread over the whole batch
            gpu = {"y": y_d.cpu().numpy(), "t_final": tfin_d.cpu().numpy(), "quality": qual, "nst": stats[:, 0], "nerr": stats[:, R.S_NERR],
                   "errcodes": stats[:, R.S_ERRCODES], "tgas": cout_d[:, R.O_TGAS].cpu().numpy(),
                   "freeze_rec": cout_d[:, R.O_TFREEZE_REC].cpu().numpy()}
            tidx = sample_idx[::max(1, nsample // 32)][:32]
            p8 = R.default_params()
            for f in ("ATOL", "t_max", "dt_first_step", "ratio_tstep", "mxstep_per_interval", "steps_reset_solver"):
                setattr(p8, f, getattr(params, f))
            p8.RTOL = 1e-8
            net.set_cost_hints(None)
            gy = None if (args.evolT or colgrid is not None) else net.evol_solve_batch(p8, cells_h[tidx], yinit_h[tidx])["y"]
            # (dependency-order sweep: the reference runs on the records as the sweep left them, i.e. with the shielding slots the device
            # wrote; the functions behind the slots are pinned to the compiled reference by tests/test_shielding_sweep.py)
            cells_cmp = sweep_cells[0] if colgrid is not None else cells_h
            out["cpu_baseline"], out["parity"] = cpu_baseline_and_parity(cells_cmp, sample_idx, network, initial, params, gpu, nS,
                                                                         tight=None if (args.evolT or colgrid is not None) else (tidx, gy), hc=hc_h)
            if args.evolT:
                T0, T1 = cells_h[:, R.cells.P_TGAS], gpu["tgas"]
                out["evolT"] = {"cells_T_still_evolving_at_end": int(cout_d[:, R.O_EVOLT_END].sum().item()),
                                "median_abs_change_of_T_K": float(np.median(np.abs(T1 - T0))), "max_T_ratio": float(np.max(T1 / T0)),
                                "min_T_ratio": float(np.min(T1 / T0))}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
