#!/usr/bin/env python3
"""bench.py -- cell-steps/s of the batched chemistry solve on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 1 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (rate coefficients + tolerance policy + the whole chem_evol_solve
integration to t_max) over one batch of synthetic cells.  Workload (BASELINE.json configs[1]): 10 000
synthetic cells per GPU, log-uniform T in [10,3000] K and n_H in [1e3,1e12] cm^-3 (rac-2d_amd/cells.py,
seed 20240601 + rank), network rate06 "withoutgrain" (464 species, 4767 reactions), initial abundances
ini_abund_waterice_loMetal.dat, template solver settings (RTOL 1e-4, ATOL 1e-30, t_max 1e6 yr,
dt_first_step 1e-8, ratio 1.1, steps_reset_solver 50).  Cells shard embarrassingly: every rank solves its
own 10 000 cells (weak scaling); the only exchange is ONE RCCL all-gather of the end-state abundances,
inside the timed region.  Inputs are resident in HBM before the clock starts.

Scheduling: a few cells in 10^4 need 10-20x the median work.  As between two global
iterations of the disk model, every pass hands the per-cell cycle counts it measured to the next one
(racgpu_set_cost_hints), which then starts the costliest cells first; the hand-over is inside the timed region.
The first warm-up pass has no history and runs in queue order: its rate is reported as
config.queue_order_first_pass.  --no-hints keeps queue order throughout.  Results do not depend on the order.

metric value = (accepted integrator steps summed over all cells and ranks) / (max over ranks of wall time).
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
NETWORK = "rate06_dipole_reformated_again_withoutgrain.dat"
INITIAL = "ini_abund_waterice_loMetal.dat"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes(nS, nR, nnzJ, nzl, nzu, nst, nfe, nje, nlu, qsum):
    """B_step = 8*[a(nnzLU+NEQ) + a(nR+2NEQ) + b(nnzJ+nnzLU) + c(nR+nnzJ) + 4(q+1)NEQ + 6NEQ]
    summed over steps, with a, b, c, q from this run's own counters (a*NST = NFE etc.)."""
    neq = nS + 1
    nnzlu = nzl + nzu + neq
    return 8.0 * (nfe * (nnzlu + neq) + nfe * (nR + 2 * neq) + nlu * (nnzJ + nnzlu) + nje * (nR + nnzJ)
                  + 4.0 * (qsum + nst) * neq + 6.0 * nst * neq)


def cpu_baseline(cells, y0_path, nst_gpu, max_seconds=30.0):
    """The reference's own Fortran path (oracle/_ref/ref_driver, built from the unmodified sources in the
    build container) on a bounded sample of the same cells, one process per host core.  Falls back to the C
    restatement (kind "port").  Baseline only."""
    cores = min(os.cpu_count() or 1, 16)
    nsample = min(len(cells), 16 * cores)  # ~1 s per cell per core: 15-25 s of wall time
    sample = cells[:nsample]
    driver = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if os.path.exists(driver):
        with tempfile.TemporaryDirectory() as td:
            procs = []
            t0 = time.perf_counter()
            for w in range(cores):
                part = sample[w::cores]
                if len(part) == 0:
                    continue
                d = os.path.join(td, "w%d" % w)
                os.makedirs(d)
                np.savetxt(os.path.join(d, "cells.txt"), part, fmt="%.17e")
                with open(os.path.join(d, "run.nml"), "w") as f:
                    f.write("&ref_run\n chem_dir='%s/'\n network='%s'\n initial='%s'\n out_dir='%s'\n cell_file='%s'\n ncell=%d\n"
                            " rtol=1D-4\n atol=1D-30\n dt_first_step=1D-8\n ratio_tstep=1.1D0\n t_max=1D6\n mxstep=6000\n"
                            " steps_reset=50\n dump_jac=0\n solve=1\n/\n" % (DATA, NETWORK, INITIAL, d, os.path.join(d, "cells.txt"), len(part)))
                procs.append(subprocess.Popen([driver, os.path.join(d, "run.nml")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
            ok = all(p.wait() == 0 for p in procs)
            dt = time.perf_counter() - t0
        if ok:
            steps = float(np.sum(nst_gpu[:nsample]))
            return {"value": steps / dt, "unit": "cell-steps/s", "cores": cores, "kind": "reference",
                    "sample": "%d cells of the same batch (first %d), reference Fortran/DLSODES binary, %d processes, %.1f s wall; "
                              "steps counted with the GPU run's NST for the same cells (DLSODES zeroes its own counter at every solver reset)"
                              % (nsample, nsample, min(cores, nsample), dt)}
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_ctypes as O
    onet = O.Network(os.path.join(DATA, NETWORK))
    y0 = onet.initial_abundances(y0_path)
    op = O.default_params()
    t0 = time.perf_counter()
    steps = 0
    n = 0
    for c in sample:
        steps += onet.solve_cell(op, c, y0)["nst"]
        n += 1
        if time.perf_counter() - t0 > max_seconds:
            break
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "cell-steps/s", "cores": 1, "kind": "port",
            "sample": "%d cells of the same batch, C restatement (oracle/), single thread, %.1f s" % (n, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cells", type=int, default=10000, help="cells per GPU (default: BASELINE configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hints", action="store_true", help="take cells in queue order in every pass (no cost feedback)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the racgpu path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    R = importlib.import_module("rac-2d_amd")
    R.set_device(local_rank)
    net = R.Network(os.path.join(DATA, NETWORK))
    nS = net.nSpecies
    y0 = net.load_initial_abundances(os.path.join(DATA, INITIAL))
    params = R.default_params()
    ncell = args.cells
    cells_h = R.cells.synth_batch(ncell, seed=20240601 + rank)
    yinit_h = net.init_abundances(y0, cells_h)

    dev = torch.device("cuda", local_rank)
    cells_d = torch.from_numpy(cells_h).to(dev)
    yinit_d = torch.from_numpy(yinit_h).to(dev)
    y_d = torch.empty_like(yinit_d)
    tfin_d = torch.zeros(ncell, dtype=torch.float64, device=dev)
    qual_d = torch.zeros(ncell, dtype=torch.int32, device=dev)
    stats_d = torch.zeros((ncell, R.NSTAT), dtype=torch.int64, device=dev)
    gathered = torch.empty((world * ncell, nS), dtype=torch.float64, device=dev) if world > 1 else None
    stream = torch.cuda.current_stream(dev)
    net.set_stream(stream.cuda_stream)

    kernel_ms = []

    def one_pass():
        y_d.copy_(yinit_d)
        net.evol_solve_batch_device(params, ncell, cells_d.data_ptr(), y_d.data_ptr(), tfin_d.data_ptr(), qual_d.data_ptr(), stats_d.data_ptr())
        if world > 1:
            dist.all_gather_into_tensor(gathered, y_d)  # the path's single exchange: RCCL over xGMI
        torch.cuda.synchronize(dev)
        kernel_ms.append(net.last_kernel_ms())
        if not args.no_hints:
            # cost feedback, as between two global iterations of the disk model: the cycles each cell took in this pass
            # order the next pass (costliest first).  Part of the pass, so it is inside the timed region.
            net.set_cost_hints(stats_d[:, 8].cpu().numpy().astype(np.float64))

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
    local = torch.tensor([dt, float(stats[:, 0].sum()) * args.steps], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = local[0:1].clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        ssum = local[1:2].clone(); dist.all_reduce(ssum, op=dist.ReduceOp.SUM)
        t_all, steps_all = float(tmax.item()), float(ssum.item())
    else:
        t_all, steps_all = dt, float(local[1].item())

    if rank == 0:
        nst, nfe, nje, nlu, qsum = [float(stats[:, k].sum()) for k in (0, 1, 2, 3, 6)]
        abytes = algorithmic_bytes(nS, net.nReactions, net.nnzJ, net.nzl, net.nzu, nst, nfe, nje, nlu, qsum)
        kms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
        achieved = abytes / (kms * 1e-3) / 1e9
        # HBM bytes per launch.  PMC counters cannot be read from inside this process; the figure is the per-cell-step
        # traffic of the committed rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this same workload
        # (profiles/r1_pmc_calibration.json, corrected as MI355X_MICROARCH.md prescribes) times this launch's cell-steps.
        traffic, traffic_src = None, None
        try:
            cal = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_calibration.json")))
            if cal["workload"]["cells_per_gpu"] == ncell and cal["workload"]["network"] == NETWORK:
                traffic = cal["bytes_per_cell_step_corrected"] * nst
                traffic_src = "profiles/r1_pmc_calibration.json: %.0f B per cell-step x %d cell-steps" % (cal["bytes_per_cell_step_corrected"], int(nst))
        except Exception:
            pass
        out = {
            "metric": "cell-steps/s (whole node)", "value": steps_all / t_all, "unit": "cell-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * t_all / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: %d synthetic cells per GPU (log-uniform T in [10,3000] K, n_H in [1e3,1e12] cm^-3), "
                                   "rate06 no-grain network (%s: %d species, %d reactions), %s, t_max=1e6 yr, RTOL=1e-4, "
                                   "steps_reset_solver=50" % (ncell, NETWORK, nS, net.nReactions, INITIAL),
                       "cells_per_gpu": ncell, "parallelism": "cells sharded over %d GPU(s), one RCCL all-gather at output" % world,
                       "scheduling": ("costliest-first from the previous pass's per-cell cycle counts (racgpu_set_cost_hints)" if hinted
                                      else "queue order (no previous pass to take cost hints from)"),
                       # rank 0's first warm-up pass runs in queue order: its rate is the no-feedback figure
                       "queue_order_first_pass": ({"ms": 1e3 * first_pass_s, "cell_steps_per_s_rank0": float(stats[:, 0].sum()) / first_pass_s}
                                                  if (first_pass_s and hinted) else None)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "k_solve", "kernel_ms": kms,
                         "algorithmic_bytes_per_launch": abytes,
                         "bytes_per_cell_step": abytes / max(nst, 1.0)},
            "cell_steps_per_pass_rank0": nst, "mean_steps_per_cell": nst / ncell,
            "nfe_per_step": nfe / max(nst, 1), "nlu_per_step": nlu / max(nst, 1), "nje_per_step": nje / max(nst, 1),
            "mean_order": qsum / max(nst, 1), "cells_with_quality_flags": int((qual != 0).sum()),
            # share of each wave's shader-clock cycles per phase (in-kernel s_memtime brackets, summed over cells)
            "phase_cycle_share": {k: float(stats[:, i].sum()) / max(float(stats[:, 8].sum()), 1.0)
                                  for k, i in (("rhs", 9), ("jacobian", 10), ("lu", 11), ("tri_solve", 12),
                                               ("lu_scatter", 13), ("lu_lds_pivots", 14), ("lu_reg_pivots", 15))},
            "wave_cycles_per_cell_step": float(stats[:, 8].sum()) / max(nst, 1.0),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cells_h, os.path.join(DATA, INITIAL), stats[:, 0])
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
