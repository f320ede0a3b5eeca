#!/usr/bin/env python3
"""First Newton step of the block-tree path with compress steps against the strictly leaf-first order (HPF_COMPRESS=0), bus by bus:
where along p -> c -> v the two part.   python tools/compress_check.py [buses] [hmax] [seed]   (GPU)"""
import os
os.environ.setdefault("HPF_ENV_SWITCHES", "1")      # A/B tooling: HPF_* switches of the environment reach hpf_create (include/hpf.h)
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import harmonic_power_flow_amd as hp             # noqa: E402
from harmonic_power_flow_amd import api, synth   # noqa: E402
import tree_plan                                 # noqa: E402

INPUTS = os.path.join(REPO, "tests", "golden", "inputs")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
hmax = int(sys.argv[2]) if len(sys.argv) > 2 else 51
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
tmp = tempfile.mkdtemp()
fb, fl = synth.gen(n, seed=seed, outdir=tmp)
st = hp.Settings(H_MAX=hmax)
buses, lines, m, nn, c = hp.init_network(fb, fl, settings=st)
Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS)
NE = hp.import_Norton_Equivalents(buses, True, st, INPUTS)
Hn = len(st.HARMONICS)


def first_step():
    dm = api._device_model(buses, Y, NE, True, st.HARMONICS, solver="block_tree", max_scenarios=1)
    dm.set_loads(buses["P"].to_numpy(float), buses["Q"].to_numpy(float))
    dm.set_state(None, None, n_scen=1)
    dm.fund_pf(1e-6, 30)
    s0 = dm.get_state()
    dm.mismatch(want_f=False)
    dm.iterate(1)
    Vm, Va = dm.get_state()
    cen = dm.tree_census()
    dm.close()
    return (Vm[0] - s0[0][0]).reshape(Hn, n), (Va[0] - s0[1][0]).reshape(Hn, n), cen


os.environ.pop("HPF_COMPRESS", None)
rows = tree_plan.plan(n, hmax, seed)
a = first_step()
os.environ["HPF_COMPRESS"] = "0"
b = first_step()
print("census", a[2], "\nflat  ", b[2])
dm_ = np.abs(a[0] - b[0]).max(axis=0)
da_ = np.abs(a[1] - b[1]).max(axis=0)
sc = max(np.abs(b[0]).max(), np.abs(b[1]).max())
print("step scale %.3e; max |d step| Vm %.3e Va %.3e" % (sc, dm_.max(), da_.max()))
info = {r[0]: r for r in rows}
for r in sorted(rows, key=lambda r: r[3]):
    k = r[0]
    if r[8] or k == 0 or any(info[x][8] for x in [r[1]] if x in info):
        print("bus %4d parent %4d level %2d depth %2d role %d : |d| Vm %.2e Va %.2e" % (k, r[1], r[2], r[3], r[8], dm_[k], da_[k]))
