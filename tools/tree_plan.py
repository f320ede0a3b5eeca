"""Host-only look at the BLOCK_TREE elimination plan of a synthetic feeder (no GPU): hpf_tree_plan writes one line per dense bus;
this script prints the level profile and what COMPRESS steps on the Gauss-Jordan skeleton would make of it.

    python tools/tree_plan.py [buses] [H_MAX]
"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import harmonic_power_flow_amd as hp                      # noqa: E402
from harmonic_power_flow_amd import _lib, ingest, synth   # noqa: E402

INPUTS = os.path.join(REPO, "tests", "golden", "inputs")


def plan(nb, hmax, seed=0, max_scenarios=1, ties=0):
    tmp = tempfile.mkdtemp(prefix="hpf_plan_")
    fb, fl = synth.gen(nb, seed=seed, outdir=tmp)
    if ties:
        synth.add_ties(fl, nb, ties)
    st = hp.Settings(H_MAX=hmax)
    buses, lines, m, n, c = hp.init_network(fb, fl, settings=st)
    Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS)
    NE = hp.import_Norton_Equivalents(buses, True, st, INPUTS)
    dev, Y_N, I_N, n_dev = ingest.norton_arrays(buses, NE, True, len(st.HARMONICS))
    d = _lib.hpf_desc()
    rowptr = np.ascontiguousarray(Y.rowptr, dtype=np.int32)
    col = np.ascontiguousarray(Y.col, dtype=np.int32)
    Yv = np.ascontiguousarray(Y.Yval, dtype=np.complex128)
    dev = np.ascontiguousarray(dev, dtype=np.int32)
    Y_N = np.ascontiguousarray(Y_N, dtype=np.complex128)
    I_N = np.ascontiguousarray(I_N, dtype=np.complex128)
    d.n, d.m, d.c, d.Hn, d.nnz = n, m, c, len(st.HARMONICS), len(col)
    d.n_dev, d.coupled, d.solver, d.device, d.max_scenarios = int(n_dev), 1, 1, 0, int(max_scenarios)
    d.rowptr, d.col = rowptr.ctypes.data_as(_lib.c_int_p), col.ctypes.data_as(_lib.c_int_p)
    d.Yval = Yv.view(np.float64).ctypes.data_as(_lib.c_dbl_p)
    d.dev_of_bus = dev.ctypes.data_as(_lib.c_int_p)
    d.Y_N = Y_N.view(np.float64).ctypes.data_as(_lib.c_dbl_p)
    d.I_N = I_N.view(np.float64).ctypes.data_as(_lib.c_dbl_p)
    path = os.path.join(tmp, "plan.txt")
    rc = _lib.load().hpf_tree_plan(C.byref(d), path.encode())
    assert rc == 0, rc
    rows = [tuple(int(x) for x in ln.split()) for ln in open(path) if not ln.startswith("#")]
    return rows


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    hmax = int(sys.argv[2]) if len(sys.argv) > 2 else 51
    rows = plan(nb, hmax)
    info = {r[0]: r for r in rows}
    kids = {}
    for k, par, *_ in rows:
        kids.setdefault(par, []).append(k)
    nlev = max(r[2] for r in rows) + 1
    print("dense buses %d, levels %d, depths %d" % (len(rows), nlev, max(r[3] for r in rows) + 1))
    for l in range(nlev):
        lv = [r for r in rows if r[2] == l]
        print("  level %2d: %3d buses  (gj %d, leaf %d, bordered %d)" % (l, len(lv), sum(r[4] == 0 for r in lv), sum(r[4] == 1 for r in lv), sum(r[4] == 2 for r in lv)))
    # compress simulation: a Gauss-Jordan bus v (not the root) with a Gauss-Jordan parent p and exactly one "pending" child c that is a
    # Gauss-Jordan bus on the critical path; greedy top-down pairing along the critical chains: v compressed => c not compressed
    gj = {k for k, r in info.items() if r[4] == 0}
    height = {k: info[k][2] for k in info}

    def schedule(comp):
        """level of every dense bus under the compress set `comp` (v -> c)"""
        lev = {}
        order = sorted(info, key=lambda k: height[k])
        # new parent / dependencies: c waits for v; p waits for c and v; v waits for its children except c
        deps = {k: [ch for ch in kids.get(k, [])] for k in info}
        for v, c in comp.items():
            p = info[v][1]
            deps[v] = [ch for ch in kids.get(v, []) if ch != c]
            deps[c] = deps[c] + [v]
            deps[p] = deps[p] + [c]
        done = {}

        def lv(k, stack=()):
            if k in done:
                return done[k]
            assert k not in stack, "cycle"
            r = 0
            for dd in deps[k]:
                r = max(r, lv(dd, stack + (k,)) + 1)
            done[k] = r
            return r
        for k in order:
            lev[k] = lv(k)
        return lev

    base = schedule({})
    print("baseline levels:", max(base.values()) + 1)
    # top-down alternation: compress v when its parent is neither compressed nor the child of a compress step ... and v's tallest dense
    # child c is a Gauss-Jordan bus and alone on v's critical path
    comp = {}
    for v in sorted(gj, key=lambda k: -height[k]):
        p = info[v][1]
        if p < 0 or p not in gj or v in comp.values():
            continue
        ch = sorted(kids.get(v, []), key=lambda k: -height[k])
        if not ch or ch[0] not in gj:
            continue
        if len(ch) > 1 and height[ch[1]] == height[ch[0]]:
            continue
        comp[v] = ch[0]
    # drop the steps that do not shorten the schedule
    for v in sorted(comp, key=lambda k: height[k]):
        trial = dict(comp)
        del trial[v]
        if max(schedule(trial).values()) <= max(schedule(comp).values()):
            comp = trial
    for v, c in comp.items():
        print("  compress %4d (height %2d; child %4d, parent %4d)" % (v, height[v], c, info[v][1]))
    lev = schedule(comp)
    print("with %d compress steps: %d levels" % (len(comp), max(lev.values()) + 1))
    for l in range(max(lev.values()) + 1):
        ks = [k for k in lev if lev[k] == l]
        print("  level %2d: %3d buses (gj %d)" % (l, len(ks), sum(k in gj for k in ks)))


if __name__ == "__main__":
    main()
