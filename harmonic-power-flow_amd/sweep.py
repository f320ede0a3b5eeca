"""Scenario sweeps across GPUs: one process per GPU, scenarios dealt round-robin, NO collective on the data path —
independent feeders / Monte-Carlo load cases share nothing during the NR loop.  The only exchange
is one all-gather of the 24-byte per-scenario records (`hpf_stat`: n_iter, flags, err, thd_max) at the end of a
sweep (`torch.distributed`, backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests)."""
import numpy as np

STAT_DTYPE = np.dtype([("n_iter", "<i4"), ("flags", "<i4"), ("err", "<f8"), ("thd_max", "<f8")])
assert STAT_DTYPE.itemsize == 24


def scenario_ids(rank, world, per_rank):
    """Round-robin deal: iteration counts differ per scenario, interleaving balances the ranks."""
    return rank + world * np.arange(per_rank)


def gather_stats(rec, world):
    """rec: uint8 tensor [S_local, 24] on this rank's device -> uint8 tensor [S_local*world, 24] ordered by global
    scenario id (id = rank + world*i)."""
    if world == 1:
        return rec
    import torch
    import torch.distributed as dist
    parts = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(parts, rec.contiguous())
    return torch.stack(parts, dim=1).reshape(-1, rec.shape[1])


def solve_scenarios(dm, P, Q, thresh_f=1e-6, max_iter_f=30, thresh_h=1e-4, max_iter_h=50, want_voltages=False, refill=True):
    """Monte-Carlo / what-if sweep on ONE GPU: every row of P, Q [n_scen][n] (p.u. loads, HG:197,372) is one scenario of the
    network `dm` (a DeviceModel) holds -- the reference's counterpart is one hpf() call per load case (HG:511).  Per scenario:
    reference start (HG:174-184), fundamental pf (HG:244), harmonic NR with the reference's stop rule (HG:536).
    refill=True (default): `hpf_solve_queue` -- the `dm.S_max` slots of the model stay full: scenarios that meet the stop rule are
    harvested between chunks of Newton iterations and their slots take the next pending scenarios, so a sweep of many more scenarios
    than slots runs at the lock-step rate of a full handle instead of paying every wave's convergence tail (size the model for as
    many live scenarios as fit: 72 MB of solver state per scenario of the 1 000-bus x 25-harmonic feeder; larger batches amortise the
    latency-bound upper tree levels).  A scenario the static-pivot monitor flags (flags bit 3), or whose mismatch turned non-finite in the
    static-pivot kernels (bit 2), is solved again on its own through `hpf_solve`, which repeats exactly those with partial pivoting.
    refill=False: fixed waves of up to S_max scenarios (each through fund_pf + solve).
    -> structured array of per-scenario records (n_iter, flags, err, thd_max: the 24-byte record of the multi-GPU gather)
    [+ raw Vm, Va [n_scen][Hn*n]]; every record and voltage is bit-identical to the scenario solved alone."""
    P = np.ascontiguousarray(np.atleast_2d(P), dtype=np.float64)
    Q = np.ascontiguousarray(np.atleast_2d(Q), dtype=np.float64)
    n_scen = P.shape[0]
    out = np.zeros(n_scen, dtype=STAT_DTYPE)
    Vm = Va = None
    if want_voltages:
        Vm = np.empty((n_scen, dm.n * dm.Hn))
        Va = np.empty_like(Vm)

    def wave(a, b):
        dm.set_loads(P[a:b], Q[a:b])
        dm.set_state(None, None, n_scen=b - a)
        dm.fund_pf(thresh_f, max_iter_f)
        dm.solve(thresh_h, max_iter_h)
        st = dm.stats()
        for k in STAT_DTYPE.names:
            out[k][a:b] = st[k]
        if want_voltages:
            Vm[a:b], Va[a:b] = dm.get_state()

    if not refill:
        for a in range(0, n_scen, dm.S_max):
            wave(a, min(a + dm.S_max, n_scen))
        return (out, Vm, Va) if want_voltages else out
    # the device keeps the voltages of a whole call: bound a call by ~8 GB of result buffers when they are asked for
    per_call = n_scen if not want_voltages else max(dm.S_max, int(8e9 // (16 * dm.n * dm.Hn)))
    for a in range(0, n_scen, per_call):
        b = min(a + per_call, n_scen)
        res = dm.solve_queue(P[a:b], Q[a:b], thresh_f, max_iter_f, thresh_h, max_iter_h, want_voltages=want_voltages)
        if want_voltages:
            rec, Vm[a:b], Va[a:b] = res
        else:
            rec = res
        for k in STAT_DTYPE.names:
            out[k][a:b] = rec[k]
    # static pivot order flagged (bit 3) or non-finite mismatch (bit 2) -- the two conditions hpf_solve's repeat pass looks at (k_mark_repeat):
    # the scenario alone, hpf_solve repeats it pivoted
    for s in np.nonzero((out["flags"] & (8 | 4)) != 0)[0]:
        wave(int(s), int(s) + 1)
    return (out, Vm, Va) if want_voltages else out


def summarize(raw):
    """raw: uint8 array [n, 24] -> convergence statistics of the sweep."""
    st = np.ascontiguousarray(raw).view(STAT_DTYPE).reshape(-1)
    conv = (st["flags"] & 1) != 0
    out = {"scenarios": int(len(st)), "converged": int(conv.sum()), "hit_max_iter": int(((st["flags"] & 2) != 0).sum()),
           "non_finite": int(((st["flags"] & 4) != 0).sum()),
           "iters_min": int(st["n_iter"].min()), "iters_max": int(st["n_iter"].max()),
           "iters_mean": float(st["n_iter"].mean()), "iters_total": int(st["n_iter"].sum())}
    if conv.any():
        out["err_max_converged"] = float(st["err"][conv].max())
        thd = st["thd_max"][conv]
        thd = thd[np.isfinite(thd)]
        if len(thd):
            out["thd_max"] = float(thd.max())
            # distribution over scenarios of the worst-bus THD_F (each record carries the max over buses, HG:563-572)
            for q in (50, 95, 99):
                out["thd_p%d" % q] = float(np.percentile(thd, q))
    return out


def pack_stats(n_iter, flags, err, thd):
    """Host-side packing of the record layout (tests)."""
    st = np.zeros(len(n_iter), dtype=STAT_DTYPE)
    st["n_iter"], st["flags"], st["err"], st["thd_max"] = n_iter, flags, err, thd
    return st.view(np.uint8).reshape(len(n_iter), 24)
