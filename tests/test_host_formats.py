"""Host-side file formats either side of the hot path: Norton-parameter CSV writer/reader
round trip in the layout of the reference's fitting script (NE_from_sim.py:195-209) and the sweep summary."""
import os

import numpy as np
import pandas as pd

from conftest import INPUTS

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import harmonic_power_flow_amd as hp
from harmonic_power_flow_amd import ingest, sweep
from harmonic_power_flow_amd.settings import Settings


def _bits(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.complex128)).view(np.uint64)


def test_norton_file_round_trip_is_bit_exact(tmp_path):
    src = os.path.join(INPUTS, "smps_NE.csv")
    raw = ingest.read_Norton_file(src)
    assert len(raw[0]) == 50 and raw[1].shape == (50, 50)
    out = ingest.export_Norton_Equivalents(str(tmp_path / "copy_NE.csv"), *raw)
    back = ingest.read_Norton_file(out)
    assert back[0] == raw[0]
    for a, b in zip(raw[1:], back[1:]):
        assert np.array_equal(_bits(a), _bits(b))


def test_written_norton_file_imports_like_the_reference_file(tmp_path):
    """A file produced by the writer goes through import_Norton_Equivalents (HG:278-310) to the same p.u. objects
    as the reference's own smps_NE.csv, coupled and uncoupled, for a harmonic subset."""
    raw = ingest.read_Norton_file(os.path.join(INPUTS, "smps_NE.csv"))
    ingest.export_Norton_Equivalents(str(tmp_path / "smps_NE.csv"), *raw)
    st = Settings(H_MAX=11)
    buses = pd.DataFrame({"type": ["slack", "nonlinear"], "component": [None, "smps"]})
    for coupled in (True, False):
        a = ingest.import_Norton_Equivalents(buses, coupled, settings=st, ne_dir=INPUTS)["smps"]
        b = ingest.import_Norton_Equivalents(buses, coupled, settings=st, ne_dir=str(tmp_path))["smps"]
        for x, y in zip(a, b):
            assert np.array_equal(_bits(x.to_numpy()), _bits(y.to_numpy()))
            assert list(x.index) == list(y.index) and list(x.columns) == list(y.columns)


def test_writer_handles_signed_zero_and_extremes(tmp_path):
    K = 3
    Y = np.array([[complex(0.0, -0.0), 1e-300 - 3.3e200j, complex(-0.0, 5)],
                  [1 + 2j, -1 - 2j, 0.1 + 0.2j],
                  [np.pi * 1j, -np.e, 1 / 3 - 2j / 3]])
    v = np.array([1e-17 + 1j, -2.5e-8j, 7.0])
    out = ingest.export_Norton_Equivalents(str(tmp_path / "d_NE.csv"), [50, 150, 250], Y, v, 2 * v, -v)
    f, Yc, Ic, Yu, Iu = ingest.read_Norton_file(out)
    assert f == [50, 150, 250]
    assert np.array_equal(_bits(Yc), _bits(Y)) and np.array_equal(_bits(Ic), _bits(v))
    assert np.array_equal(_bits(Yu), _bits(2 * v)) and np.array_equal(_bits(Iu), _bits(-v))


def test_sweep_summary_reports_thd_percentiles():
    n = 200
    thd = np.linspace(1.0, 40.0, n)
    flags = np.ones(n, dtype=np.int32)
    flags[:3] = 2                                   # three scenarios hit max_iter: excluded from THD statistics
    raw = sweep.pack_stats(np.full(n, 20, dtype=np.int32), flags, np.full(n, 1e-9), thd)
    s = sweep.summarize(raw)
    ok = thd[3:]
    assert s["converged"] == n - 3 and s["hit_max_iter"] == 3
    assert s["thd_max"] == ok.max()
    for q in (50, 95, 99):
        assert s["thd_p%d" % q] == np.percentile(ok, q)


def test_bench_reports_counter_traffic_only_for_the_running_library(tmp_path, monkeypatch):
    """bench.py's `roofline.traffic` comes from committed PMC passes (profiles/pmc_traffic_latest.json); it must be `null` unless that file
    was collected with the very libhpf.so that is running (sha256 recorded by tools/pmc_traffic.py) and on the default workload."""
    import argparse
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    args = argparse.Namespace(buses=1000, hmax=51, solver="block_tree")
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    lib_dir = tmp_path / "harmonic-power-flow_amd"
    lib_dir.mkdir()
    (lib_dir / "libhpf.so").write_bytes(b"not really a library")
    sha = bench.lib_sha16()
    good = {"format": 2, "lib_sha16": sha, "note": "n", "command": "c", "per_launch_bytes": {"k_factor_q_general": {"fetch": 3.0, "write": 4.0}},
            "per_step_bytes": {"k": {"fetch": 10.0, "write": 5.0}}}
    (prof / "pmc_traffic_latest.json").write_text(json.dumps(good))
    t, note, src = bench.pmc_traffic(args, 128)
    assert t == 7.0 and src["lib_sha16"] == sha and bench.pmc_step_traffic(args, 128) == 15.0
    assert bench.pmc_traffic(args, 64)[0] is None                      # another workload
    (prof / "pmc_traffic_latest.json").write_text(json.dumps(dict(good, lib_sha16="0" * 16)))
    t, note, src = bench.pmc_traffic(args, 128)
    assert t is None and src is None and "another build" in note and bench.pmc_step_traffic(args, 128) is None


def test_bench_gpus_flag_starts_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` without a launcher must start N ranks itself (fresh child processes through torch.distributed.run on
    127.0.0.1, before anything touches the GPU); under a launcher (WORLD_SIZE set) or at N = 1 the process is a rank and runs the bench."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    args = bench.parse(["--gpus", "4", "--steps", "7", "--warmup", "2"])
    cmd = bench.launch_plan(args, {})
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    assert bench.launch_plan(args, {"WORLD_SIZE": "4", "RANK": "0"}) is None           # already a rank
    assert bench.launch_plan(bench.parse(["--gpus", "1"]), {}) is None
    assert bench.parse([]).gpus == 1 and bench.parse([]).repeats >= 5


def test_get_THD_equals_the_reference_to_the_last_bit():
    """api.get_THD (HG:563-572, vectorised over buses) on the reference's own final voltages of every golden case = the reference's THD table,
    bit for bit (the per-bus sums keep the reference's order of additions)."""
    import glob
    import pandas as pd
    import harmonic_power_flow_amd as hp
    from conftest import GOLD
    n_checked = 0
    for path in sorted(glob.glob(os.path.join(GOLD, "*_H*_*c.npz"))):
        g = np.load(path, allow_pickle=True)
        if "THD" not in g.files or "V_final" not in g.files or "harmonics" not in g.files:
            continue
        harmonics, n = [int(h) for h in g["harmonics"]], int(g["n"])
        idx = pd.MultiIndex.from_product([harmonics, list(range(n))], names=["harmonic", "bus"])
        V = pd.DataFrame(g["V_final"], index=idx, columns=["V_m", "V_a"])
        thd = hp.get_THD(V).to_numpy()
        ref = np.asarray(g["THD"], dtype=float)
        assert thd.shape == ref.shape
        assert np.array_equal(thd, ref, equal_nan=True), os.path.basename(path)
        n_checked += 1
    assert n_checked >= 16
