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

Untimed legs after the timed region (rank 0 prints ONE JSON line with all of them):
  roofline        the dominant kernel ALONE -- k_level<52>, the factor sweep of the block tree, one launch per elimination level
                  (Gauss-Jordan workgroups + the level's scenario-batched workgroups in one grid; k_factor_q<B,false> for other
                  block sizes): device-clock span of every one of its launches over K more steps against its own algorithmic
                  bytes (hpf_kernel_model) — compare profiles/*kernel_stats.csv of the same command;
  sweep           every rank finishes its solves with the reference's stop rule (`hpf_solve`: per-scenario freeze, compaction
                  of the running scenarios, pipelined polling); per-scenario records (24 B) all-gathered with RCCL;
  sweep_1gpu      (N = 1 only) the whole 1 024-scenario sweep of BASELINE config 4 on ONE GPU, all scenarios live (74 GB of
                  the 288 GB): lock-step rate and solve-to-convergence rate;
  single_scenario BASELINE config 3 latency;
  cpu_baseline    the oracle (CPU restatement of the reference) on one core and on all host cores (one scenario per core).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
INPUTS = os.path.join(REPO, "tests", "golden", "inputs")      # smps_NE.csv (data fixture)

FP64_PEAK_TFLOPS = 78.6      # MI355X FP64 vector = matrix peak (spec)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scenarios", type=int, default=128, help="Monte-Carlo scenarios per GPU")
    ap.add_argument("--buses", type=int, default=1000)
    ap.add_argument("--hmax", type=int, default=51)
    ap.add_argument("--solver", default="block_tree", choices=["block_tree", "dense"])
    ap.add_argument("--cpu-iters", type=int, default=12, help="NR iterations of the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-workers", type=int, default=0, help="worker processes of the all-cores CPU leg (0 = host cores, at most 16)")
    ap.add_argument("--no-finish", action="store_true", help="skip the untimed solve-to-convergence + stats gather")
    ap.add_argument("--no-one-group", action="store_true", help="skip the one-group timing legs (profiling runs: every k_level launch then has the step's own shape)")
    ap.add_argument("--no-probe", action="store_true", help="skip the untimed probe of the stream configuration (profiling runs: a fixed number of steps)")
    ap.add_argument("--sweep-1gpu", type=int, default=1024, help="(N = 1) scenarios of the single-GPU sweep leg (0 = skip)")
    ap.add_argument("--single", action="store_true", help="(default on rank 0 at N=1) also time a single-scenario solve: BASELINE config 3")
    ap.add_argument("--no-single", action="store_true", help="skip the single-scenario latency leg")
    ap.add_argument("--repeats", type=int, default=5, help="the timed block of `steps` iterations is run this many times (each from the pf "
                                                           "seed, each bracketed by barrier + synchronize); value = the median block")
    return ap.parse_args(argv)


def launch_plan(args, env):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): the command that starts N ranks of this
    script, one per GPU -- or None when this process is itself a rank (or N = 1).  The children are FRESH processes started before
    this one imports torch or touches the GPU; nothing is re-executed in place."""
    if args.gpus <= 1 or "WORLD_SIZE" in env:
        return None
    # --standalone: the launcher's own rendezvous store picks a free port itself (a port found by bind-and-close here could be taken by
    # another process before the ranks connect); --local-addr 127.0.0.1: the container's hostname may not resolve
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
            "--standalone", "--local-addr", "127.0.0.1", os.path.abspath(__file__)] + sys.argv[1:]


def run_launcher(cmd):
    """Start the ranks, relay rank 0's JSON line (the only line a rank prints on stdout) and the launcher's exit code."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    for line in p.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return p.wait()


def build_inputs(args, hp):
    from harmonic_power_flow_amd import ingest, synth
    tmp = tempfile.mkdtemp(prefix="hpf_bench_")
    fb, fl = synth.gen(args.buses, seed=0, outdir=tmp)
    st = hp.Settings(H_MAX=args.hmax)
    t0 = time.perf_counter()
    buses, lines, m, n, c = hp.init_network(fb, fl, settings=st)
    t1 = time.perf_counter()
    Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS)
    t2 = time.perf_counter()
    NE = hp.import_Norton_Equivalents(buses, True, st, INPUTS)
    dev, Y_N, I_N, n_dev = ingest.norton_arrays(buses, NE, True, len(st.HARMONICS))
    t3 = time.perf_counter()
    setup = {"ingest_csv_ms": 1e3 * (t1 - t0), "admittance_ms": 1e3 * (t2 - t1), "norton_ms": 1e3 * (t3 - t2)}
    return dict(st=st, buses=buses, n=n, m=m, c=c, Y=Y, dev=dev, Y_N=Y_N, I_N=I_N, n_dev=n_dev, files=(fb, fl), setup=setup)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_start(args):
    """Start the CPU legs as child processes BEFORE this process touches the GPU (oracle/cpu_worker.py: the oracle on the same
    feeder, `cpu-iters` NR iterations per scenario from the reference's start, timer placed like the reference's HG:535,543):
    first one worker alone (1 core), then one worker per host core at once (the natural CPU parallelisation of the scenario
    sweep: one scenario per core)."""
    if args.cpu_iters <= 0:
        return None
    worker = os.path.join(REPO, "oracle", "cpu_worker.py")
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    nw = args.cpu_workers or min(cores, 16)
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")

    def run(ids):
        t0 = time.perf_counter()
        ps = [subprocess.Popen([sys.executable, worker, str(args.buses), str(args.hmax), str(s), str(args.cpu_iters)],
                               stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True) for s in ids]
        outs = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in ps]
        return outs, time.perf_counter() - t0
    one, _ = run([0])
    many, wall = run(list(range(nw)))
    it1 = one[0]["n_iter"] / one[0]["loop_s"]
    loop_max = max(o["loop_s"] for o in many)
    itn = sum(o["n_iter"] for o in many) / loop_max
    return {"value": itn, "unit": "NR iterations/s", "cores": nw, "cores_note": "%d of %d host cores" % (nw, cores), "kind": "port",
            "sample": "%d NR iterations of each of %d Monte-Carlo scenarios of the same %d-bus x %d-harmonic feeder, one oracle process "
                      "per core (NumPy/SciPy SuperLU restatement, bit-identical to the reference on its golden cases; BLAS pinned to 1 "
                      "thread per process); the reference itself measured 0.0257 it/s (38.9 s/it) on this feeder in the build container"
                      % (args.cpu_iters, nw, args.buses, (args.hmax + 1) // 2 - 1),
            "ms_per_iter_per_core": 1e3 * loop_max / max(many[0]["n_iter"], 1),
            "single_core": {"value": it1, "cores": 1, "ms_per_iter": 1e3 / it1},
            "host_cores": cores, "cpu_model": cpu_model()}


def main():
    args = parse()
    cmd = launch_plan(args, os.environ)
    if cmd is not None:
        raise SystemExit(run_launcher(cmd))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (one rank per GPU: start it as `python bench.py --gpus N` or under "
                         "torch.distributed.run with --nproc-per-node N)" % (args.gpus, world))
    cpu = cpu_baseline_start(args) if (rank == 0 and world == 1) else None      # N = 1 only (the other ranks of a larger job would wait for it); child processes, before any GPU initialisation
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
    from harmonic_power_flow_amd.sweep import gather_stats, scenario_ids, summarize
    scen_ids = scenario_ids(rank, world, S)                  # rank r: r, r + world, r + 2 world, ... (tests/test_sweep_gloo.py)
    P0 = inp["buses"]["P"].to_numpy(float)
    Q0 = inp["buses"]["Q"].to_numpy(float)
    scale = np.stack([synth.scenario_scale(n, int(s)) for s in scen_ids])
    dm = hp.DeviceModel(n, inp["m"], inp["c"], inp["st"].HARMONICS, inp["Y"].rowptr, inp["Y"].col, inp["Y"].Yval,
                        inp["dev"], inp["Y_N"], inp["I_N"], inp["n_dev"], True, solver=args.solver,
                        device=dev_index, max_scenarios=S)
    setup = dict(inp["setup"])
    setup.update(dm.setup_times())                       # hpf_create: tree planning on the host, uploads, per-scenario allocation
    setup["model_setup_ms"] = setup["ingest_csv_ms"] + setup["admittance_ms"] + setup["norton_ms"] + setup["create_ms"]
    dm.set_loads(P0 * scale, Q0 * scale)
    dm.set_state(None, None, n_scen=S)
    nf, _, _ = dm.fund_pf(inp["st"].thresh_f, inp["st"].max_iter_f)
    seed = dm.get_state()
    dm.mismatch(want_f=False)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed probe of the stream configuration: four scenario groups use all four hardware queues of the runtime; if anything else in this
    # process keeps a queue busy (another library's stream) two groups share one and the step is ~30 % slower -- then three groups are the
    # better configuration for THIS process.  A few iterations of each, the default stays unless it is clearly on that cliff.
    groups_probe = None
    if args.solver == "block_tree" and dm.scenario_groups(S) >= 4 and not args.no_probe:
        def per_iter(k=4):
            dm.sync()
            t = time.perf_counter()
            dm.iterate(k)
            dm.sync()
            return 1e3 * (time.perf_counter() - t) / k
        dm.iterate(2)
        t4 = min(per_iter(), per_iter())
        dm.set_option("scenario_groups", 3)
        dm.iterate(2)
        t3 = min(per_iter(), per_iter())
        keep4 = t4 <= 1.1 * t3
        if keep4:
            dm.set_option("scenario_groups", 4)
        groups_probe = {"ms_per_step_4_groups": t4, "ms_per_step_3_groups": t3, "chosen": 4 if keep4 else 3}
    # R timed blocks of exactly K steps, each from the pf seed (W warm-up steps first), each bracketed by barrier + synchronize on both
    # sides and reduced with MAX over the ranks; the reported block is the median one
    blocks, blocks_ranks = [], []
    for rep in range(max(args.repeats, 1)):
        if rep or groups_probe:
            dm.set_state(seed[0], seed[1])
            dm.mismatch(want_f=False)
        dm.iterate(args.warmup)
        dm.sync()
        barrier()
        t0 = time.perf_counter()
        dm.iterate(args.steps)
        dm.sync()
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            # every rank's own time of the block (a straggler shows in the line), then MAX over the ranks
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            blocks_ranks.append([float(p.item()) for p in parts])
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        else:
            blocks_ranks.append([el])
        blocks.append(el)
    elapsed = float(np.median(blocks))
    median_block = int(np.argsort(blocks)[len(blocks) // 2])
    # per-kernel HIP-event timing (on the streams the kernels run on) over K more steps of the same configuration, continued
    # from the state the timed region left; kept out of the headline region because every span costs two event records
    Kt = min(args.steps, 20)                     # steps of each timing leg (bounded: every span is a pair of event records)
    dm.timing(True)
    dm.timing_reset()
    dm.iterate(Kt)
    dm.sync()
    tim = dm.timing_get()
    dm.timing(False)
    # ... and the dominant kernel's own duration on the device clock over K more steps WITHOUT event packets between the kernels
    # (a HIP event record is a barrier packet with cache maintenance: it perturbs the ~25 us kernels it brackets)
    dm.timing(2)
    dm.timing_reset()
    dm.iterate(Kt)
    dm.sync()
    tim["gj_dev"] = dm.timing_get()["gj_dev"]
    dm.timing(False)
    # ... and the same kernel with the chip to itself: the scenarios as ONE group (one launch per level, no other stream busy), K more steps
    tim["gj_dev_one_group"] = None
    if args.solver == "block_tree" and dm.scenario_groups(S) > 1 and not args.no_one_group:
        g_now = dm.scenario_groups(S)
        dm.set_option("scenario_groups", 1)
        dm.iterate(2)
        dm.timing(2)
        dm.timing_reset()
        dm.iterate(Kt)
        dm.sync()
        tim["gj_dev_one_group"] = dm.timing_get()["gj_dev"]
        dm.timing(False)
        dm.timing(True)                              # (HIP-event spans: the mismatch kernel of the one-group configuration)
        dm.timing_reset()
        dm.iterate(Kt)
        dm.sync()
        tim["mismatch_one_group"] = dm.timing_get()["mismatch"]
        dm.timing(False)
        dm.set_option("scenario_groups", g_now)

    # ---- untimed: finish the solves with the reference's stop rule, gather convergence statistics (RCCL) ----------
    sweep = None
    if not args.no_finish:
        dm.set_state(seed[0], seed[1])
        dm.solve(inp["st"].thresh_h, inp["st"].max_iter_h)                   # warm (pinned buffers, repeat-pass state)
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

    # The gather above was the last collective: the communicator is torn down HERE, by all ranks together, before rank 0 starts its
    # single-GPU legs -- no rank waits inside a collective (an RCCL barrier spins on the GPU) while rank 0 works for a minute.
    rccl_ranks = dist.get_world_size() if (world > 1 and backend == "nccl") else None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return

    K = args.steps
    Hn = dm.Hn
    total_scen = S * world
    value = total_scen * K / elapsed
    ms_step = 1e3 * elapsed / K
    b = 2 * Hn
    bt = args.solver == "block_tree"
    G = dm.scenario_groups(S) if bt else 1
    # ---- roofline of the dominant kernel, alone: k_factor_q<B,false> (block_tree) / the rocSOLVER LU (dense) ----------------------
    gj_ms, gj_n = tim["gj_dev"] if (bt and tim.get("gj_dev", (0, 0))[1]) else (tim["gj"] if bt else tim["solve"])
    ev_ms, ev_n = tim["gj"] if bt else tim["solve"]
    by_gj, fl_gj, ln_gj = dm.kernel_model("gj") if bt else dm.kernel_model("solve")
    launches_per_step = gj_n / max(Kt, 1)                                    # all scenario groups
    bytes_per_launch = by_gj * S / max(launches_per_step, 1)                 # average over its launches (tree levels x groups)
    flops_per_launch = fl_gj * S / max(launches_per_step, 1)
    avg_ms = gj_ms / max(gj_n, 1)
    achieved_gbs = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else None
    achieved_tf = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else None
    one_group = None                              # the same kernel, all scenarios in one launch per level, nothing else on the chip
    if bt and tim.get("gj_dev_one_group") and tim["gj_dev_one_group"][1]:
        o_ms, o_n = tim["gj_dev_one_group"]
        o_bpl = by_gj * S / max(o_n / max(Kt, 1), 1)
        o_avg = o_ms / o_n
        one_group = {"avg_ms": o_avg, "bytes_per_launch": o_bpl, "achieved": o_bpl / (o_avg * 1e-3) / 1e9,
                     "frac": o_bpl / (o_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, "launches_per_step": o_n / max(Kt, 1),
                     "mfma_frac": (fl_gj * S / max(o_n / max(Kt, 1), 1)) / (o_avg * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                     "assembly_frac": None,
                     "note": "the same kernel with the scenarios as ONE group (one launch per elimination level, no other stream busy): the kernel's own "
                             "figure, without the chip sharing of the %d concurrent groups the step is run with (which is faster as a whole)" % G}
    # the whole factor sweep (all factor kernels) and the whole step, algorithmic bytes over wall time
    sweep_bytes = dm.solve_bytes() * S
    nnz = len(inp["Y"].col)
    bytes_mismatch = 16 * Hn * n + 8 * dm.N
    bytes_back = dm.back_bytes()
    bytes_2x2 = (32 + 16) * 2 * Hn * n + 2 * 16 * Hn * n
    bytes_update = 16 * Hn * n + 2 * 8 * Hn * n + 2 * 8 * Hn * n + 2 * 16 * Hn * n
    step_bytes = S * (dm.solve_bytes() + bytes_mismatch + bytes_back + bytes_2x2 + bytes_update)
    traffic, traffic_note, traffic_source = pmc_traffic(args, S)
    step_traffic = pmc_step_traffic(args, S)
    census = dm.tree_census() if bt else {}
    fused = bool(census.get("fused_levels"))
    # k_mismatch (harmonic_mismatch HG:360-390, the assembly kernel that reaches HBM): algorithmic bytes of a launch / its HIP-event span
    asm_gbs = (bytes_mismatch * S / max(tim["mismatch"][1] / max(Kt, 1), 1) / (tim["mismatch"][0] / max(tim["mismatch"][1], 1) * 1e-3) / 1e9
               if tim["mismatch"][1] else None)
    if one_group and tim.get("mismatch_one_group") and tim["mismatch_one_group"][1]:
        mm_ms, mm_n = tim["mismatch_one_group"]
        one_group["assembly_frac"] = bytes_mismatch * S / max(mm_n / max(Kt, 1), 1) / (mm_ms / mm_n * 1e-3) / 1e9 / HBM_PEAK_GBS
        one_group["assembly_ms_per_launch"] = mm_ms / mm_n
    kname = "k_factor_q<%d,false>" % (100 if b > 52 else (52 if b > 28 else (28 if b > 12 else 12))) if bt and b <= 100 else ("k_tree_factor (generic)" if bt else "rocsolver_dgetrf/dgetrs")
    kdesc = kname + (" alone (general multi-wave block-tree factor kernel: Gauss-Jordan buses and non-batched super-leaves; %d launches per "
                     "Newton step and scenario group, one per tree level)" % ln_gj)
    if fused:
        kname = "k_level<%d>" % (52 if b > 28 else (28 if b > 12 else 12))
        kdesc = (kname + ": the factor sweep of the block tree, one launch per elimination level (%d per Newton step and scenario group) -- "
                 "Gauss-Jordan workgroups (one per bus and scenario: %d buses) and the scenario-batched workgroups of the constant-inverse "
                 "leaves (%d) and bordered buses (%d) of the level in one grid" % (ln_gj, census["gauss_jordan"], census["const_leaves"],
                                                                                  census["bordered"]))
    out = {
        "metric": "NR iterations/sec + ms/iter, 1 000-bus x 25-harmonic feeder; |dV| vs reference",
        "value": value, "unit": "NR iterations/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "syn%d radial feeder (gen seed 0), harmonics 1..%d odd (K=%d), coupled Norton (smps), "
                               "%d Monte-Carlo load scenarios per GPU (%d total), block-tree Newton step"
                               % (n, args.hmax, Hn - 1, S, total_scen) if bt else
                               "syn%d, K=%d, coupled, %d scenarios per GPU, dense rocSOLVER" % (n, Hn - 1, S),
                   "buses": n, "harmonics": Hn - 1, "unknowns_per_scenario": dm.N, "scenarios_per_gpu": S,
                   "solver": dm.solver, "step": "one NR iteration of every scenario (HG:537-540)",
                   "pf_iterations": int(nf.max())},
        "repeats": len(blocks), "repeat_ms_per_step": [1e3 * b / K for b in blocks],
        "repeat_note": "R timed blocks of exactly `steps` iterations, each from the pf seed after `warmup` iterations, each bracketed by "
                       "barrier + synchronize and MAX-reduced over the ranks; ms_per_step / value are the MEDIAN block's",
        "scenario_groups_probe": groups_probe,      # untimed: four groups (default) against three in this process; the default stays unless > 10 % slower
        # build switches of the handle: none in a product run; under HPF_ENV_SWITCHES=1 (tools/) the HPF_* names found in the environment
        "env_switches": (sorted(k for k in os.environ if k.startswith("HPF_") and k not in ("HPF_ENV_SWITCHES", "HPF_BENCH_BACKEND", "HPF_LIB_PATH"))
                         if os.environ.get("HPF_ENV_SWITCHES", "0") not in ("", "0") else []),
        "backend": backend if world > 1 else None,
        "rccl_ranks": rccl_ranks,
        "ms_per_step_ranks": [1e3 * t / K for t in blocks_ranks[median_block]],   # each rank's own time of the median block (value uses their MAX)
        "scenario_ids_rank0": [int(scen_ids[0]), int(scen_ids[1]) if S > 1 else None, "... + %d" % world],
        "ms_per_iter_per_scenario": ms_step / S,
        "setup_ms": setup["model_setup_ms"], "setup": setup,
        "roofline": {"bound": "hbm",
                     "kernel": kdesc, "tree_census": census,
                     "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS if achieved_gbs else None,
                     "traffic": traffic, "traffic_note": traffic_note, "traffic_source": traffic_source,
                     "mfma_frac": achieved_tf / FP64_PEAK_TFLOPS if achieved_tf else None,
                     "mfma_tflops": achieved_tf,
                     "assembly_gbs": asm_gbs, "assembly_frac": asm_gbs / HBM_PEAK_GBS if asm_gbs else None,
                     "step_hbm_frac": step_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "bytes_per_launch": bytes_per_launch, "flop_per_launch": flops_per_launch,
                     "avg_ms": avg_ms, "avg_ms_hip_event_spans": ev_ms / max(ev_n, 1), "launches_timed": gj_n,
                     "launches_per_step": launches_per_step,
                     "concurrent_groups": G,
                     "one_group": one_group,
                     "mfma": {"achieved": achieved_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": achieved_tf / FP64_PEAK_TFLOPS if achieved_tf else None},
                     "note": "achieved = algorithmic bytes of this kernel's buses (hpf_kernel_model: Schur complements in and out, "
                             "inverses out, per-scenario bus operands) per launch / average launch duration.  Duration of a launch = last "
                             "workgroup end - first workgroup start on the device's constant-rate clock (wall_clock64 stamps written by "
                             "the kernel during the timing leg, %d launches) = what rocprofv3 --kernel-trace --stats averages for this "
                             "kernel (profiles/); avg_ms_hip_event_spans is the same launches bracketed by HIP events on their streams, "
                             "which adds the event packets and queue gaps around a ~25 us kernel.  The launches of the %d scenario groups "
                             "overlap on separate streams, so a launch shares the chip with the other groups' kernels: avg_ms, achieved and frac are "
                             "SHARED-CHIP figures per launch (the sum of the launches of a step exceeds the step's wall time).  arithmetic intensity %.2f flop/B "
                             "< ridge %.1f: HBM-bound by the roofline, in practice bound by workgroup latency (DESIGN.md §5)"
                             % (gj_n, G, fl_gj / max(by_gj, 1.0), FP64_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS)},
        "roofline_factor_sweep": {"bound": "hbm", "kernels": "all factor kernels of a step (k_level, or k_leaf_batch + k_sleaf_batch + k_factor_q)",
                                  "achieved": sweep_bytes / (ms_step * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": sweep_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_step": sweep_bytes,
                                  "note": "algorithmic bytes of the factor sweep of all scenarios / step wall time"},
        "roofline_hbm_step": {"bound": "hbm", "achieved": step_bytes / (ms_step * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": step_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "bytes_per_step": step_bytes,
                              "traffic": step_traffic,
                              "traffic_gbs_over_step_wall": step_traffic / (ms_step * 1e-3) / 1e9 if step_traffic else None,
                              "note": "algorithmic bytes of a whole NR step (factor sweep + back sweep + 2x2 kernels + mismatch "
                                      "+ update) over the step wall time"},
        "roofline_assembly": {"bound": "hbm", "kernel": "k_mismatch<false> (harmonic_mismatch HG:360-390: the mismatch half of the assembly; the "
                                                         "Jacobian half is fused into the factor kernels and never reaches HBM)",
                              "achieved": bytes_mismatch * S / max(tim["mismatch"][1] / max(Kt, 1), 1) / (tim["mismatch"][0] / max(tim["mismatch"][1], 1) * 1e-3) / 1e9
                              if tim["mismatch"][1] else None,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "bytes_per_launch": bytes_mismatch * S / max(tim["mismatch"][1] / max(Kt, 1), 1),
                              "avg_ms_hip_event_span": tim["mismatch"][0] / max(tim["mismatch"][1], 1),
                              "note": "algorithmic bytes (voltages in, mismatch image out) of one launch (one scenario group) / its HIP-event span; "
                                      "rocprofv3 of the same command: profiles/*kernel_stats.csv, counters: profiles/pmc_traffic_latest.json"},
        "phase_ms_per_launch": {k: (v[0] / max(v[1], 1)) for k, v in tim.items() if v and v[1]},
        "phase_launches_per_step": {k: v[1] / max(Kt, 1) for k, v in tim.items() if v and v[1]},
        "phase_note": "HIP-event spans: gj = one per k_level / k_factor_q<B,false> launch; solve = one per launch of the other factor kernels "
                      "(k_leaf_batch, k_sleaf_batch, leaf-only k_factor_q: none with k_level); mismatch / update: per launch; back: per scenario "
                      "group and step (%d groups overlap on separate streams)" % G,
        "vs_reference_single_run_note": "not a like-for-like ratio and not reported as a number: the reference (one single-threaded pandas / SuperLU "
                                        "process, ONE scenario) measured 38.9 s per NR iteration on this feeder in the build container; `value` is "
                                        "the aggregate of %d scenarios in flight per GPU.  cpu_baseline is the comparator timed on this box." % S,
    }
    if sweep is not None:
        out["sweep"] = sweep
        out["sweep_iters_per_s"] = sweep["iters_per_s_rank0"] * world
    if args.sweep_1gpu > 0 and bt and world == 1:
        # the same 1 024-scenario sweep through THIS handle's 128 slots: hpf_solve_queue keeps the slots full (finished scenarios are
        # harvested between chunks of iterations, their slots refilled from the queue); end to end incl. uploads and the pf of all scenarios
        from harmonic_power_flow_amd.sweep import solve_scenarios
        scale_q = np.stack([synth.scenario_scale(n, s) for s in range(args.sweep_1gpu)])
        solve_scenarios(dm, P0 * scale_q[:2 * S], Q0 * scale_q[:2 * S])                         # warm (pinned buffers)
        t0 = time.perf_counter()
        rec_q = solve_scenarios(dm, P0 * scale_q, Q0 * scale_q)
        t_q = time.perf_counter() - t0
        out["sweep_queue_1gpu"] = {"scenarios": int(args.sweep_1gpu), "slots": S, "converged": int(((rec_q["flags"] & 1) != 0).sum()),
                                   "iters_total": int(rec_q["n_iter"].sum()), "wall_s": t_q, "iters_per_s": float(rec_q["n_iter"].sum()) / t_q,
                                   "note": "hpf_solve_queue (sweep.solve_scenarios): more scenarios than slots, the handle stays full until the "
                                           "queue drains; wall time includes the upload of all loads and the fundamental pf of every scenario"}
    if args.sweep_1gpu > 0 and bt:               # rank 0 of every world size: the whole sweep on ONE GPU, the strong-scaling comparator
        dm.close()
        dm = None
        out["sweep_1gpu"] = sweep_one_gpu(hp, inp, args, dev_index)
        # N GPUs with `scenarios` each against ONE GPU that holds all of them ("scaling": "weak" above is per-GPU work held fixed)
        out["sweep_1gpu_lockstep_iters_per_s"] = out["sweep_1gpu"]["lockstep_iters_per_s"]
        out["scaling_strong_vs_1gpu"] = value / out["sweep_1gpu"]["lockstep_iters_per_s"]
        out["scaling_strong_8gpu_projection"] = 8.0 * (value / world) / out["sweep_1gpu"]["lockstep_iters_per_s"]
    if (args.single or world == 1) and not args.no_single:
        out["single_scenario"] = single_scenario(hp, inp, args)
        out["single_ms_per_iter"] = out["single_scenario"]["ms_per_iter"]
        # model set-up without the one-time start of the HIP runtime (which sits inside the FIRST hpf_create of a process)
        out["setup"]["create_warm_ms"] = out["single_scenario"]["create_warm"]["create_ms"]
        out["setup_warm_ms"] = (out["setup"]["ingest_csv_ms"] + out["setup"]["admittance_ms"] + out["setup"]["norton_ms"] +
                                out["setup"]["create_warm_ms"])
    if world == 1 and bt and not args.no_single:
        out["reference_call_shapes"] = reference_call_shapes(hp, inp, args)
        out["meshed"] = meshed_feeder(hp, args)
    out["cpu_baseline"] = cpu
    print(json.dumps(out), flush=True)


def lib_sha16():
    import hashlib
    lib = os.path.join(REPO, "harmonic-power-flow_amd", "libhpf.so")
    return hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16] if os.path.exists(lib) else None


def pmc_file(args, S):
    """The committed PMC passes (tools/pmc_traffic.py) -- only if they were collected on this workload AND with the very libhpf.so
    that is running now (the JSON records the library's hash)."""
    path = os.path.join(REPO, "profiles", "pmc_traffic_latest.json")
    if not os.path.exists(path) or (args.buses, args.hmax, S, args.solver) != (1000, 51, 128, "block_tree"):
        return None, "no PMC pass for this workload"
    j = json.load(open(path))
    if j.get("format") != 2:
        return None, "no PMC pass in the current format"
    if j.get("lib_sha16") != lib_sha16():
        return None, "profiles/pmc_traffic_latest.json was collected with another build of libhpf.so (%s, running %s): not reported" % (
            j.get("lib_sha16"), lib_sha16())
    return j, ""


def pmc_traffic(args, S):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes; None unless they belong to the running library."""
    j, why = pmc_file(args, S)
    if j is None:
        return None, why, None
    d = j.get("per_launch_bytes", {}).get("k_factor_q_general")
    if not d:
        return None, "no PMC pass for this kernel", None
    return d["fetch"] + d["write"], j.get("note", ""), {"file": "profiles/pmc_traffic_latest.json", "lib_sha16": j.get("lib_sha16"),
                                                         "command": j.get("command")}


def pmc_step_traffic(args, S):
    """Counter bytes of a whole NR step (all kernels) from the committed PMC passes (same rule)."""
    j, _ = pmc_file(args, S)
    t = j.get("per_step_bytes") if j else None
    if not t:
        return None
    return sum(v["fetch"] + v["write"] for v in t.values())


def sweep_one_gpu(hp, inp, args, dev_index):
    """BASELINE config 4 on ONE GPU: all `--sweep-1gpu` (1 024) Monte-Carlo scenarios live at once (72 MB of solver state per
    scenario): lock-step rate over `steps` iterations, then the solve with the reference's stop rule from the pf seed."""
    import torch
    from harmonic_power_flow_amd import synth
    from harmonic_power_flow_amd.sweep import summarize
    n, S = inp["n"], args.sweep_1gpu
    P0 = inp["buses"]["P"].to_numpy(float)
    Q0 = inp["buses"]["Q"].to_numpy(float)
    scale = np.stack([synth.scenario_scale(n, s) for s in range(S)])
    # the best single-GPU configuration for ~1 000 live scenarios: the leaf-first tree (HPF_COMPRESS=0: 5 - 8 % faster per step once the
    # elimination levels fill the chip); rounding-level different Newton steps from the default build, same fixed points
    dm = hp.DeviceModel(n, inp["m"], inp["c"], inp["st"].HARMONICS, inp["Y"].rowptr, inp["Y"].col, inp["Y"].Yval,
                        inp["dev"], inp["Y_N"], inp["I_N"], inp["n_dev"], True, solver=args.solver,
                        device=dev_index, max_scenarios=S, options="HPF_COMPRESS=0" if args.solver == "block_tree" else None)
    dm_Hn, N_unk, bytes_solve, bytes_back = dm.Hn, dm.N, dm.solve_bytes(), dm.back_bytes()
    try:
        dm.set_loads(P0 * scale, Q0 * scale)
        dm.set_state(None, None, n_scen=S)
        t0 = time.perf_counter()
        dm.fund_pf(inp["st"].thresh_f, inp["st"].max_iter_f)
        t_pf = time.perf_counter() - t0
        seed = dm.get_state()
        dm.mismatch(want_f=False)
        dm.iterate(2)
        dm.sync()
        Kl = max(4, args.steps // 2)
        t0 = time.perf_counter()
        dm.iterate(Kl)
        dm.sync()
        t_lock = time.perf_counter() - t0
        dm.set_state(seed[0], seed[1])
        dm.solve(inp["st"].thresh_h, 2)                                       # warm the solve path (pinned buffers)
        dm.set_state(seed[0], seed[1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_iter, err, _ = dm.solve(inp["st"].thresh_h, inp["st"].max_iter_h)
        t_solve = time.perf_counter() - t0
        st = dm.stats()
        rec = np.zeros((S, 24), dtype=np.uint8)
        rec.view([("n_iter", "<i4"), ("flags", "<i4"), ("err", "<f8"), ("thd_max", "<f8")])[:, 0] = st
        out = summarize(rec)
    finally:
        dm.close()
    Hn = dm_Hn
    step_bytes = S * (bytes_solve + (16 * Hn * n + 8 * N_unk) + bytes_back + ((32 + 16) * 2 * Hn * n + 2 * 16 * Hn * n) +
                      (16 * Hn * n + 2 * 8 * Hn * n + 2 * 8 * Hn * n + 2 * 16 * Hn * n))
    out.update({"scenarios": S, "lockstep_ms_per_step": 1e3 * t_lock / Kl, "lockstep_iters_per_s": S * Kl / t_lock,
                "roofline_hbm_step": {"bound": "hbm", "achieved": step_bytes / (t_lock / Kl) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": step_bytes / (t_lock / Kl) / 1e9 / HBM_PEAK_GBS, "bytes_per_step": step_bytes,
                                      "note": "algorithmic bytes of a whole lock-step NR iteration of all live scenarios over its wall time: with "
                                              "1 024 live scenarios the latency-bound tree levels amortise and the step runs against the HBM "
                                              "roof (counters of this configuration: profiles/r02/pmc_traffic_s1024.json, 26.2 GB per step)"},
                "solve_wall_s": t_solve, "iters_per_s": float(n_iter.sum()) / t_solve, "pf_wall_s": t_pf,
                "build_options": "HPF_COMPRESS=0",
                "note": "one GPU, all scenarios live, leaf-first tree (options HPF_COMPRESS=0); solve = hpf_solve with the reference's stop rule (per-scenario freeze, "
                        "compaction of the running scenarios between chunks of 4 iterations, pipelined polling)"})
    return out


def reference_call_shapes(hp, inp, args):
    """The reference's own call shapes end to end on this feeder (untimed leg, N = 1): a repeated hp.hpf() call (HG:511: the handle comes from the
    cache, the Norton CSV from the parse memo) and update_harmonic_state_vec on the CSR Jacobian (HG:476-479: hpf_sparse_solve)."""
    st, buses = inp["st"], inp["buses"]
    from harmonic_power_flow_amd import ingest
    lines = ingest.init_lines_from_csv(inp["files"][1], st)
    t0 = time.perf_counter()
    V, err_h, n_it, J = hp.hpf(buses, lines, True, settings=st, ne_dir=INPUTS, verbose=False)
    t1 = time.perf_counter()
    V, err_h, n_it, J = hp.hpf(buses, lines, True, settings=st, ne_dir=INPUTS, verbose=False)
    t2 = time.perf_counter()
    V2, _, _, _ = hp.hpf(buses, lines, True, settings=st, ne_dir=INPUTS, verbose=False, return_jacobian=False)
    t3 = time.perf_counter()
    f = np.cos(0.37 * np.arange(J.shape[0]))
    x0 = np.zeros(J.shape[0])
    hp.update_harmonic_state_vec(J, x0, f)
    t4 = time.perf_counter()
    x1 = hp.update_harmonic_state_vec(J, x0, f)
    t5 = time.perf_counter()
    res = float(np.abs(J @ (x0 - x1) - f).max() / (np.abs(J).dot(np.abs(x0 - x1)).max() + np.abs(f).max()))
    hp.close_all()
    return {"hpf_first_call_ms": 1e3 * (t1 - t0), "hpf_second_call_ms": 1e3 * (t2 - t1), "hpf_second_call_without_jacobian_ms": 1e3 * (t3 - t2),
            "n_iter_h": int(n_it), "jacobian_nnz": int(J.nnz), "unknowns": int(J.shape[0]),
            "update_harmonic_state_vec_ms": 1e3 * (t5 - t4), "update_harmonic_state_vec_rel_residual": res,
            "note": "hp.hpf() of ONE scenario of this feeder end to end (admittance build, Norton import, pf, harmonic NR, post-processing, the CSR "
                    "Jacobian of the last iteration; second call: device handle from the cache) and x - J^-1 f for that CSR Jacobian through "
                    "hpf_sparse_solve (no N x N array; the reference's spsolve takes ~1 s here, its hpf() 1 050 s)"}


def meshed_feeder(hp, args):
    """The feeder with loop-closing lines (untimed leg, N = 1): ms per Newton iteration of ONE scenario through the bordered block-tree step
    (factor-once form: one sweep + selected inversion over the tie endpoints' root paths + block Gauss-Jordan of the border system)."""
    from harmonic_power_flow_amd import api, synth
    out = {}
    for k in (5, 20):
        tmp = tempfile.mkdtemp(prefix="hpf_bench_mesh_")
        fb, fl = synth.gen(args.buses, seed=0, outdir=tmp)
        synth.add_ties(fl, args.buses, k)
        st = hp.Settings(H_MAX=args.hmax)
        buses, lines, m, n, c = hp.init_network(fb, fl, settings=st)
        Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS)
        NE = hp.import_Norton_Equivalents(buses, True, st, INPUTS)
        dm = api._device_model(buses, Y, NE, True, st.HARMONICS, solver="block_tree")
        try:
            dm.set_loads(buses["P"].to_numpy(float), buses["Q"].to_numpy(float))
            dm.set_state(None, None, n_scen=1)
            dm.fund_pf(1e-6, 30)
            seed = dm.get_state()
            dm.solve(1e-4, 3)                              # warm
            dm.set_state(*seed)
            t0 = time.perf_counter()
            it, err, _ = dm.solve(1e-4, 50)
            t = time.perf_counter() - t0
            cs = dm.tree_census()
            out["ties_%d" % k] = {"ms_per_iter": 1e3 * t / max(int(it[0]), 1), "n_iter_h": int(it[0]), "err_h": float(err[0]),
                                  "border_unknowns": int(cs["border_unknowns"]), "root_path_buses": int(cs["root_path_buses"]),
                                  "levels": int(cs["levels"]), "pivoted_border_systems": int(cs["border_repivots"])}
        finally:
            dm.close()
    out["note"] = ("one scenario, reference stop rule; the virtual-sweep form of rounds 2 - 4 (HPF_MESH_SEL=0) takes 4.6 / 16.2 ms per iteration at 5 / 20 ties, "
                   "DESIGN.md 3.5")
    return out


def single_scenario(hp, inp, args):
    """Latency of ONE scenario (BASELINE config 3): full solve with the reference's stop rule."""
    n = inp["n"]
    dm = hp.DeviceModel(n, inp["m"], inp["c"], inp["st"].HARMONICS, inp["Y"].rowptr, inp["Y"].col, inp["Y"].Yval,
                        inp["dev"], inp["Y_N"], inp["I_N"], inp["n_dev"], True, solver=args.solver,
                        device=int(os.environ.get("LOCAL_RANK", "0")), max_scenarios=1)
    setup_warm = dm.setup_times()                          # hpf_create of a further handle in this process: no HIP runtime start in it
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
            "solve_ms": 1e3 * t, "create_warm": setup_warm}


if __name__ == "__main__":
    main()
