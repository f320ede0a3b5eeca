#!/usr/bin/env python3
"""Headline benchmark: Newton-Raphson iterations/s of the harmonic power flow on the synthetic 1 000-bus x
25-harmonic radial feeder (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload (per GPU, fixed as N grows -> weak scaling): `--scenarios` Monte-Carlo load scenarios (default 128 =
1 024 / 8) of `gen(1000, seed=0)`, harmonics 1..51 (K = 25), coupled Norton equivalents; scenario ids are dealt
round-robin over ranks (rank + world*i).  Setup — CSV ingest, admittance build, Norton import, fundamental power flow
(on the device), initial mismatch — is outside the timed region and leaves everything resident in HBM.

A *step* is one full NR iteration of every scenario on the GPU: Jacobian assembly + block-tree elimination +
back-substitution, state update, mismatch + inf-norm (HG:537-540).  The W+K timed steps are the first W+K iterations
of the real solves from the reference's start point (no scenario has converged yet: the reference needs 27), run
with `hpf_iterate`, i.e. without host synchronisation.  value = scenarios_total * K / t  [NR iterations / s].

After the timed region every rank finishes its solves with the reference's stop rule (`hpf_solve`), the per-scenario
records (n_iter, flags, err, thd_max; 24 B) are all-gathered with RCCL and rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
INPUTS = os.path.join(REPO, "tests", "golden", "inputs")      # smps_NE.csv (data fixture)

FP64_PEAK_TFLOPS = 78.6      # MI355X FP64 vector = matrix peak (spec)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scenarios", type=int, default=128, help="Monte-Carlo scenarios per GPU")
    ap.add_argument("--buses", type=int, default=1000)
    ap.add_argument("--hmax", type=int, default=51)
    ap.add_argument("--solver", default="block_tree", choices=["block_tree", "dense"])
    ap.add_argument("--cpu-iters", type=int, default=12, help="NR iterations of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-finish", action="store_true", help="skip the untimed solve-to-convergence + stats gather")
    ap.add_argument("--single", action="store_true", help="(default on rank 0 at N=1) also time a single-scenario solve: BASELINE config 3")
    ap.add_argument("--no-single", action="store_true", help="skip the single-scenario latency leg")
    return ap.parse_args()


def build_inputs(args, hp):
    from harmonic_power_flow_amd import ingest, synth
    tmp = tempfile.mkdtemp(prefix="hpf_bench_")
    fb, fl = synth.gen(args.buses, seed=0, outdir=tmp)
    st = hp.Settings(H_MAX=args.hmax)
    buses, lines, m, n, c = hp.init_network(fb, fl, settings=st)
    Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS)
    NE = hp.import_Norton_Equivalents(buses, True, st, INPUTS)
    dev, Y_N, I_N, n_dev = ingest.norton_arrays(buses, NE, True, len(st.HARMONICS))
    return dict(st=st, buses=buses, n=n, m=m, c=c, Y=Y, dev=dev, Y_N=Y_N, I_N=I_N, n_dev=n_dev, files=(fb, fl))


def cpu_baseline(inp, iters):
    """The oracle (CPU restatement, bit-identical to the reference on the golden cases) on the same feeder, ONE scenario,
    `iters` NR iterations from the same start; timer placed like the reference's (HG:535,543)."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import hpf_oracle as o
    fb, fl = inp["files"]
    net = o.init_network(fb, fl)
    H = inp["st"].HARMONICS
    rowptr, col, Yval = o.build_admittance_matrices(net, H)
    Vm, Va, _, _ = o.pf(net, rowptr, col, Yval)
    NE = o.import_Norton_Equivalents(net, H, True, INPUTS)
    mdl = o.Model(net, H, rowptr, col, Yval, NE, True)
    r = o.hpf_from_model(mdl, Vm, Va, thresh_h=0.0, max_iter_h=iters)
    return {"value": r["n_iter_h"] / r["loop_s"], "unit": "NR iterations/s", "cores": 1, "kind": "port",
            "sample": "%d NR iterations of one scenario of the same %d-bus x %d-harmonic feeder (oracle: NumPy/SciPy "
                      "SuperLU restatement, bit-identical to the reference on its golden cases); the reference itself "
                      "measured 0.0257 it/s (38.9 s/it) on this feeder in the build container"
                      % (r["n_iter_h"], inp["n"], len(H) - 1),
            "ms_per_iter": 1e3 * r["loop_s"] / max(r["n_iter_h"], 1)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # one rank per GPU; HPF_BENCH_BACKEND=gloo + fewer GPUs than ranks is a rehearsal mode for a 1-GPU box only
    backend = os.environ.get("HPF_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    import harmonic_power_flow_amd as hp

    inp = build_inputs(args, hp)
    n, S = inp["n"], args.scenarios
    from harmonic_power_flow_amd import synth
    from harmonic_power_flow_amd.sweep import gather_stats, summarize
    scen_ids = rank + world * np.arange(S)
    P0 = inp["buses"]["P"].to_numpy(float)
    Q0 = inp["buses"]["Q"].to_numpy(float)
    scale = np.stack([synth.scenario_scale(n, int(s)) for s in scen_ids])
    dm = hp.DeviceModel(n, inp["m"], inp["c"], inp["st"].HARMONICS, inp["Y"].rowptr, inp["Y"].col, inp["Y"].Yval,
                        inp["dev"], inp["Y_N"], inp["I_N"], inp["n_dev"], True, solver=args.solver,
                        device=dev_index, max_scenarios=S)
    dm.set_loads(P0 * scale, Q0 * scale)
    dm.set_state(None, None, n_scen=S)
    nf, _, _ = dm.fund_pf(inp["st"].thresh_f, inp["st"].max_iter_f)
    seed = dm.get_state()
    dm.mismatch(want_f=False)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    dm.iterate(args.warmup)
    dm.sync()
    barrier()
    t0 = time.perf_counter()
    dm.iterate(args.steps)
    dm.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # per-phase HIP-event timing (on the streams the kernels run on) over the same K steps, continued from the state the
    # timed region left; kept out of the headline region because every span costs two event records on the host
    dm.timing(True)
    dm.timing_reset()
    dm.iterate(args.steps)
    dm.sync()
    tim = dm.timing_get()
    dm.timing(False)

    # ---- untimed: finish the solves with the reference's stop rule, gather convergence statistics (RCCL) ----------
    sweep = None
    if not args.no_finish:
        dm.set_state(seed[0], seed[1])
        torch.cuda.synchronize()
        t_sw = time.perf_counter()
        n_iter, err, _ = dm.solve(inp["st"].thresh_h, inp["st"].max_iter_h)
        t_sw = time.perf_counter() - t_sw
        rec = torch.empty((S, 24), dtype=torch.uint8, device="cuda")
        dm.stats_to_device(rec.data_ptr())
        allrec = gather_stats(rec if backend == "nccl" else rec.cpu(), world)
        sweep = summarize(allrec.cpu().numpy())
        sweep["solve_wall_s_rank0"] = t_sw        # hpf_solve of this rank's scenarios with the reference's stop rule (untimed leg)
        sweep["iters_per_s_rank0"] = float(n_iter.sum()) / t_sw

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    K = args.steps
    Hn = dm.Hn
    total_scen = S * world
    value = total_scen * K / elapsed
    ms_step = 1e3 * elapsed / K
    solve_ms, solve_n = tim["solve"]
    # One HIP-event span per k_factor_w launch (tree level x scenario group), on the stream it runs on.  Scenario groups are
    # independent pipelines on separate streams, so launches of different groups overlap; like rocprofv3 --stats, `avg_ms`
    # averages the launch durations as if they were alone.  achieved = algorithmic flops of all launches / sum of their
    # durations = (flops per launch) / (average launch duration).
    flops_step = dm.solve_flops() * S                  # all factor launches of one NR step on this GPU
    bytes_sweep = dm.solve_bytes() * S                 # algorithmic HBM bytes of the same launches (hpf_solve_bytes)
    launches_per_step = solve_n / max(K, 1)
    achieved_tf = flops_step * K / (solve_ms * 1e-3) / 1e12 if solve_ms > 0 else None
    achieved_gbs = bytes_sweep * K / (solve_ms * 1e-3) / 1e9 if solve_ms > 0 else None
    G = max(1, round(launches_per_step / max(dm.n_levels, 1)))
    b = 2 * Hn
    nnz = len(inp["Y"].col)
    # algorithmic HBM bytes of the other kernels of a step, one scenario (per-scenario arrays only; Y, Y_N, the tree records and
    # the leaf images are shared by all scenarios and served by L2 / Infinity Cache):
    #   mismatch: U in, f out;  back sweep: the inverses of the Gauss-Jordan buses in, w and A(k,parent) in, x out;
    #   2x2 kernels: per bus and harmonic 2x2 inverse + w out and in, voltages in;  update: x, Vm, Va in, Vm, Va, U, E out
    bytes_mismatch = 16 * Hn * n + 8 * dm.N
    bytes_back = dm.back_bytes()
    bytes_2x2 = (32 + 16) * 2 * Hn * n + 2 * 16 * Hn * n
    bytes_update = 16 * Hn * n + 2 * 8 * Hn * n + 2 * 8 * Hn * n + 2 * 16 * Hn * n
    step_bytes = S * (dm.solve_bytes() + bytes_mismatch + bytes_back + bytes_2x2 + bytes_update)
    traffic, traffic_note = pmc_traffic(args, S)
    step_traffic = pmc_step_traffic(args, S)
    out = {
        "metric": "NR iterations/sec + ms/iter, 1 000-bus x 25-harmonic feeder; |dV| vs reference",
        "value": value, "unit": "NR iterations/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "syn%d radial feeder (gen seed 0), harmonics 1..%d odd (K=%d), coupled Norton (smps), "
                               "%d Monte-Carlo load scenarios per GPU (%d total), block-tree Newton step"
                               % (n, args.hmax, Hn - 1, S, total_scen) if args.solver == "block_tree" else
                               "syn%d, K=%d, coupled, %d scenarios per GPU, dense rocSOLVER" % (n, Hn - 1, S),
                   "buses": n, "harmonics": Hn - 1, "unknowns_per_scenario": dm.N, "scenarios_per_gpu": S,
                   "solver": dm.solver, "step": "one NR iteration of every scenario (HG:537-540)",
                   "pf_iterations": int(nf.max())},
        "ms_per_iter_per_scenario": ms_step / S,
        "roofline": {"bound": "hbm",
                     "kernel": "k_factor_q<%d> (multi-wave block-tree factor kernel; one sweep = %d launches per scenario "
                               "group, one per tree level; levels 0 / 1 are mostly k_leaf_batch / k_sleaf_batch: lazy leaves and super-leaves, 16 scenarios per workgroup)" % (52 if b > 28 else (28 if b > 12 else 12), dm.n_levels),
                     "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS if achieved_gbs else None,
                     "traffic": traffic / max(launches_per_step, 1) if traffic else None, "traffic_note": traffic_note,
                     "bytes_per_launch": bytes_sweep / max(launches_per_step, 1),
                     "avg_ms": solve_ms / max(solve_n, 1), "launches_per_step": launches_per_step,
                     "concurrent_groups": G,
                     "aggregate_over_step_wall": bytes_sweep / (ms_step * 1e-3) / 1e9,
                     "note": "launches of the %d scenario groups overlap on separate streams, so a launch's duration includes the "
                             "share of the GPU the other groups take; aggregate_over_step_wall = algorithmic bytes of all factor "
                             "launches of a step / step wall time (GB/s). " % G +
                             "arithmetic intensity of the sweep %.2f flop/B < ridge %.1f: HBM-bound by the roofline; achieved = "
                             "algorithmic bytes per launch / average launch duration (HIP events on the launch streams)"
                             % (flops_step / bytes_sweep, FP64_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS)},
        "roofline_mfma": {"bound": "mfma", "kernel": "same launches", "achieved": achieved_tf, "peak": FP64_PEAK_TFLOPS,
                          "unit": "TFLOP/s", "frac": achieved_tf / FP64_PEAK_TFLOPS if achieved_tf else None,
                          "flop_per_launch": flops_step / max(launches_per_step, 1)},
        "roofline_hbm_step": {"bound": "hbm", "achieved": step_bytes / (ms_step * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": step_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "bytes_per_step": step_bytes,
                              "traffic": step_traffic,
                              "traffic_gbs_over_step_wall": step_traffic / (ms_step * 1e-3) / 1e9 if step_traffic else None,
                              "note": "algorithmic bytes of a whole NR step (factor sweep + back sweep + 2x2 kernels + mismatch "
                                      "+ update) over the step wall time"},
        "phase_ms_per_step": {k: (v[0] / max(v[1], 1)) for k, v in tim.items() if v[1]},
        "phase_note": "solve: per k_factor_w launch; others: per scenario group and step (%d groups overlap on separate "
                      "streams)" % G,
        "vs_reference_measured": value / 0.0257,
    }
    if sweep is not None:
        out["sweep"] = sweep
    if (args.single or world == 1) and not args.no_single:
        out["single_scenario"] = single_scenario(hp, inp, args)
    if args.cpu_iters > 0 and world == 1:
        out["cpu_baseline"] = cpu_baseline(inp, args.cpu_iters)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def pmc_traffic(args, S):
    """HBM bytes of one factor sweep from the committed PMC passes (tools/pmc_traffic.py); only valid for the default
    workload they were collected on."""
    path = os.path.join(REPO, "profiles", "pmc_traffic_latest.json")
    if not os.path.exists(path) or (args.buses, args.hmax, S, args.solver) != (1000, 51, 128, "block_tree"):
        return None, "no PMC pass for this workload"
    per = json.load(open(path))["per_step_bytes"]
    d = per.get("k_factor_q")
    if not d:
        return None, "no PMC pass for this kernel"
    tot = d["fetch_raw"] + d.get("write_calibrated", d["write"])
    for name in ("k_leaf_batch", "k_sleaf_batch"):    # levels 0 / 1 of the sweep: the scenario-batched kernels, same timing spans
        lb = per.get(name)
        if lb:
            tot += lb["fetch_raw"] + lb.get("write_calibrated", lb["write"])
    return tot, ("per launch: (FETCH_SIZE*1024 raw + WRITE_SIZE*1024 x store calibration) of a factor sweep / launches; "
                                         "tile-image loads calibrate at 1.0 (k_back_q, known bytes), stores at ~0.55 (k_update, known "
                                         "bytes); includes the shared leaf images served by the Infinity Cache; separate "
                                         "rocprofv3 --pmc passes, see profiles/pmc_traffic_latest.json")


def pmc_step_traffic(args, S):
    """Counter bytes of a whole NR step (all kernels, fetch raw + calibrated stores) from the committed PMC passes."""
    path = os.path.join(REPO, "profiles", "pmc_traffic_latest.json")
    if not os.path.exists(path) or (args.buses, args.hmax, S, args.solver) != (1000, 51, 128, "block_tree"):
        return None
    t = json.load(open(path))["per_step_bytes"]
    return sum(v["fetch_raw"] + v.get("write_calibrated", v["write"]) for v in t.values())


def K_steps(args):
    return args.steps


def dm_levels(dm):
    return getattr(dm, "n_levels", 0) or 0


def single_scenario(hp, inp, args):
    """Latency of ONE scenario (BASELINE config 3): full solve with the reference's stop rule."""
    n = inp["n"]
    dm = hp.DeviceModel(n, inp["m"], inp["c"], inp["st"].HARMONICS, inp["Y"].rowptr, inp["Y"].col, inp["Y"].Yval,
                        inp["dev"], inp["Y_N"], inp["I_N"], inp["n_dev"], True, solver=args.solver,
                        device=int(os.environ.get("LOCAL_RANK", "0")), max_scenarios=1)
    dm.set_loads(inp["buses"]["P"].to_numpy(float), inp["buses"]["Q"].to_numpy(float))
    dm.set_state(None, None, n_scen=1)
    dm.fund_pf(1e-6, 30)
    seed = dm.get_state()
    dm.solve(1e-4, 50)                                     # warm
    dm.set_state(seed[0], seed[1])
    t0 = time.perf_counter()
    n_iter, err, _ = dm.solve(1e-4, 50)
    t = time.perf_counter() - t0
    dm.close()
    return {"n_iter_h": int(n_iter[0]), "err_h": float(err[0]), "ms_per_iter": 1e3 * t / max(int(n_iter[0]), 1),
            "solve_ms": 1e3 * t}


if __name__ == "__main__":
    main()
