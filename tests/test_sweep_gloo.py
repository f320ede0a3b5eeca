"""Multi-process path of the scenario sweep on CPU: world_size 2, gloo backend (the GPU run uses RCCL through the same
code).  Checks the round-robin deal, the all-gather ordering and the summary; no GPU compute involved."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, per_rank, q):
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    import torch
    import torch.distributed as dist
    from harmonic_power_flow_amd import sweep
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = sweep.scenario_ids(rank, world, per_rank)
    rec = sweep.pack_stats(15 + ids % 7, np.where(ids % 5 == 0, 2, 1), 1e-6 * (ids + 1), 0.1 + 0.01 * ids)
    allrec = sweep.gather_stats(torch.from_numpy(rec.copy()), world)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        q.put((allrec.numpy().copy(), float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_orders_by_global_scenario_id():
    import torch.multiprocessing as mp
    from harmonic_power_flow_amd import sweep
    world, per_rank = 2, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    raw, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    st = raw.view(sweep.STAT_DTYPE).reshape(-1)
    ids = np.arange(world * per_rank)
    assert np.array_equal(st["n_iter"], 15 + ids % 7)
    np.testing.assert_allclose(st["err"], 1e-6 * (ids + 1))
    assert tmax == 2.0
    summ = sweep.summarize(raw)
    assert summ["scenarios"] == 12 and summ["converged"] == int((ids % 5 != 0).sum())
    assert summ["hit_max_iter"] == int((ids % 5 == 0).sum())
    assert summ["iters_total"] == int((15 + ids % 7).sum())


def test_round_robin_partition_covers_all_scenarios():
    from harmonic_power_flow_amd import sweep
    world, per = 8, 128
    all_ids = np.sort(np.concatenate([sweep.scenario_ids(r, world, per) for r in range(world)]))
    assert np.array_equal(all_ids, np.arange(1024))


def test_scenario_scale_is_seeded_and_bounded():
    from harmonic_power_flow_amd import synth
    u = synth.scenario_scale(1000, 3)
    assert np.array_equal(u, synth.scenario_scale(1000, 3))
    assert u.min() >= 0.5 and u.max() <= 1.5 and not np.array_equal(u, synth.scenario_scale(1000, 4))


def test_bench_deals_the_scenarios_with_the_tested_function():
    """bench.py's rank r takes scenario ids r + world * i: it calls sweep.scenario_ids (no second formula of its own), and that function gives
    exactly those ids."""
    import re
    from harmonic_power_flow_amd import sweep
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(repo, "bench.py")).read()
    assert re.search(r"scen_ids\s*=\s*scenario_ids\(rank,\s*world,\s*S\)", src)
    assert "rank + world * np.arange" not in src
    for world in (1, 2, 8):
        for r in range(world):
            assert np.array_equal(sweep.scenario_ids(r, world, 5), r + world * np.arange(5))
