"""Host logic of the BLOCK_TREE elimination plan without a GPU: hpf_tree_plan runs the tree planning of hpf_create on the host and dumps one
line per dense bus (bus, dense parent, elimination level, back-sweep depth, kind, ..., compress role).  Checked here: the dependencies
the device kernels rely on -- every bus is eliminated after what it waits for, the back sweep visits a bus after the buses whose step it
needs -- with and without the compress steps, and that the compress steps shorten the chain."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))


def _plan(n, hmax, monkeypatch, compress):
    import tree_plan
    if compress is None:
        monkeypatch.delenv("HPF_COMPRESS", raising=False)
    else:
        monkeypatch.setenv("HPF_COMPRESS", compress)
    rows = tree_plan.plan(n, hmax)
    return {r[0]: dict(par=r[1], level=r[2], depth=r[3], kind=r[4], vector_only=r[5], hbm_children=r[6], via_chain=r[7], role=r[8]) for r in rows}


@pytest.mark.parametrize("n,hmax", [(1000, 51), (300, 51), (400, 27)])
def test_compress_plan_dependencies(n, hmax, monkeypatch):
    flat = _plan(n, hmax, monkeypatch, "0")
    comp = _plan(n, hmax, monkeypatch, None)
    assert set(flat) == set(comp)
    assert all(v["role"] == 0 for v in flat.values())
    # leaf-first order: a bus one level above its tallest dense child, one depth below its parent
    for k, v in flat.items():
        kids = [c for c, w in flat.items() if w["par"] == k]
        assert v["level"] == (max(flat[c]["level"] for c in kids) + 1 if kids else 0) or (not kids and v["kind"] == 0)
        assert v["depth"] == (flat[v["par"]]["depth"] + 1 if v["par"] >= 0 else 0)
    lv_f, lv_c = max(v["level"] for v in flat.values()) + 1, max(v["level"] for v in comp.values()) + 1
    assert lv_c < lv_f
    if (n, hmax) == (1000, 51):
        assert (lv_f, lv_c) == (15, 10)
    vs = [k for k, v in comp.items() if v["role"] == 1]
    cs = [k for k, v in comp.items() if v["role"] == 2]
    assert len(vs) == len(cs) > 0
    for c in cs:                                    # a pending child: Gauss-Jordan bus, re-linked to its grandparent, never compressed itself
        v = flat[c]["par"]
        assert comp[v]["role"] == 1 and comp[c]["par"] == flat[v]["par"] == comp[v]["par"]
        assert comp[c]["kind"] == 0 and comp[v]["kind"] == 0 and comp[comp[v]["par"]]["kind"] == 0
        assert comp[v]["level"] < comp[c]["level"] < comp[comp[c]["par"]]["level"]          # v -> c -> p in the factor sweep
        assert comp[comp[c]["par"]]["depth"] < comp[c]["depth"] < comp[v]["depth"]          # p -> c -> v in the back sweep
    for k, v in comp.items():                       # everybody else: after its dense children, below its parent
        for c, w in comp.items():
            if w["par"] == k:
                assert w["level"] < v["level"] and w["depth"] > v["depth"]
    # the kinds (which kernel takes which bus) do not depend on the compress steps
    assert all(flat[k]["kind"] == comp[k]["kind"] and flat[k]["vector_only"] == comp[k]["vector_only"] for k in flat)


def test_tree_plan_of_a_meshed_model_keeps_the_tie_endpoints_root_paths_plain(tmp_path):
    """Round 5 (factor-once bordered step): the planner keeps every bus on a root path of a tie endpoint as a PLAIN Gauss-Jordan bus -- dense, not a
    constant-inverse / lazy leaf, not a bordered bus, not linked through a contracted chain, no compress role -- so that its inverse sits in its
    inverse slot after a sweep.  hpf_tree_plan (host only) marks them in its last column."""
    import tree_plan
    radial = tree_plan.plan(300, 51)
    meshed = tree_plan.plan(300, 51, ties=4)
    assert all(r[9] == 0 for r in radial)
    marked = {r[0]: r for r in meshed if r[9] != 0}
    assert len(marked) >= 8 and all(r[9] == 1 for r in marked.values())
    for k, r in marked.items():
        assert r[4] == 0 and r[5] == 0 and r[7] == 0 and r[8] == 0           # kind, vector_only, via_chain, compress_role
        assert r[1] == -1 or r[1] in marked                               # closed under "dense parent of"
    assert 0 in marked                                                     # the root is on every root path
    # the marked buses cost levels: more Gauss-Jordan buses, a chain at least as long
    assert sum(1 for r in meshed if r[4] == 0) > sum(1 for r in radial if r[4] == 0)


def test_tree_plan_bad_arguments(tmp_path):
    from harmonic_power_flow_amd import _lib
    lib = _lib.load()
    assert lib.hpf_tree_plan(None, b"/tmp/x") == -1
    d = _lib.hpf_desc()
    assert lib.hpf_tree_plan(C.byref(d), None) == -1


def test_tree_plan_does_not_depend_on_the_handle_capacity_and_leaves_the_environment_alone(monkeypatch, tmp_path):
    """hpf_tree_plan plans exactly what hpf_create would: the same tree for every capacity d->max_scenarios (round 5: the compress steps are the
    default everywhere; HPF_COMPRESS=0 builds the leaf-first tree), does not touch the process environment, and reports a path it cannot write
    instead of passing a stale file."""
    import tree_plan
    from harmonic_power_flow_amd import _lib
    monkeypatch.delenv("HPF_COMPRESS", raising=False)
    monkeypatch.setenv("HPF_TREE_DUMP", "/nonexistent-dir/user-value")
    small = tree_plan.plan(300, 51, max_scenarios=256)
    big = tree_plan.plan(300, 51, max_scenarios=257)
    assert os.environ["HPF_TREE_DUMP"] == "/nonexistent-dir/user-value"
    assert small == big and any(r[8] for r in small)
    monkeypatch.setenv("HPF_COMPRESS", "0")
    flat = tree_plan.plan(300, 51, max_scenarios=257)
    assert not any(r[8] for r in flat) and max(r[2] for r in small) < max(r[2] for r in flat)
