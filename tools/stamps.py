#!/usr/bin/env python3
"""Phase breakdown of the block-tree factor kernel from in-kernel cycle stamps (diagnostic build path, HPF_DEBUG_ABLATE=16).
Run on the GPU box: HPF_DEBUG_ABLATE=16 HPF_GROUPS=1 python tools/stamps.py [scenarios]"""
import ctypes as C
import os
os.environ.setdefault("HPF_ENV_SWITCHES", "1")      # A/B tooling: HPF_* switches of the environment reach hpf_create (include/hpf.h)
import sys
import tempfile

import numpy as np

os.environ.setdefault("HPF_DEBUG_ABLATE", "16")
os.environ.setdefault("HPF_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                   "harmonic-power-flow_amd", "libhpf_stamps.so"))   # build.py --stamps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import harmonic_power_flow_amd as hp
from harmonic_power_flow_amd import ingest, synth

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
NBUS = int(sys.argv[2]) if len(sys.argv) > 2 else 1000          # python tools/stamps.py 1 10000 99: config 5
HMAX = int(sys.argv[3]) if len(sys.argv) > 3 else 51
INPUTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "inputs")
tmp = tempfile.mkdtemp()
fb, fl = synth.gen(NBUS, seed=0, outdir=tmp)
st = hp.Settings(H_MAX=HMAX)
buses, lines, m, n, c = hp.init_network(fb, fl, settings=st)
Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS)
NE = hp.import_Norton_Equivalents(buses, True, st, INPUTS)
dev, Y_N, I_N, n_dev = ingest.norton_arrays(buses, NE, True, len(st.HARMONICS))
dm = hp.DeviceModel(n, m, c, st.HARMONICS, Y.rowptr, Y.col, Y.Yval, dev, Y_N, I_N, n_dev, True, solver="block_tree", max_scenarios=S)
P0, Q0 = buses["P"].to_numpy(float), buses["Q"].to_numpy(float)
scale = np.stack([synth.scenario_scale(n, s) for s in range(S)])
dm.set_loads(P0 * scale, Q0 * scale)
dm.set_state(None, None, n_scen=S)
dm.fund_pf()
dm.mismatch(want_f=False)
dm.iterate(3)
dm.sync()
buf = np.zeros(S * n * 8, dtype=np.int64)
dm._chk(dm.lib.hpf_debug_stamps(dm._h, buf.ctypes.data_as(C.POINTER(C.c_longlong)), buf.size), "hpf_debug_stamps")
d = buf.reshape(S, n, 8)
dense = d[:, :, 3].sum(axis=0) > 0
# bordered buses of the scenario-batched kernel (k_sleaf_batch): o[6] = 1000 + m, phases in o[0..5]
sl = d[0, :, 6] >= 1000
if sl.any():
    for mval in sorted(set(d[0, sl, 6] - 1000)):
        sel = d[0, :, 6] == 1000 + mval
        x = d[:, sel, :]
        t = x[:, :, 3].reshape(-1)
        ph = [np.median(x[:, :, 0]), np.median(x[:, :, 1]), np.median(x[:, :, 2]), np.median(t & 0xfffff), np.median((t >> 20) & 0xfffff),
              np.median((t >> 40) & 0xfffff), np.median(x[:, :, 4]), np.median(x[:, :, 5])]
        print("k_sleaf_batch m=%2d n=%3d (cycles of the 100 MHz*? s_memtime clock): rows %6.0f  harmonics %6.0f  barrier wait %6.0f  T %6.0f  inversion %6.0f  MFMA+Qb v %6.0f  tail %6.0f  total %7.0f"
              % ((mval, sel.sum()) + tuple(ph)))
    d[:, sl, :] = 0
    dense = d[:, :, 3].sum(axis=0) > 0
names = ["assembly", "rows->tiles", "child sums", "MFMA GJ", "store", "push"]
print("dense buses: %d; cycles per block (median over scenarios and buses)" % dense.sum())
for label, sel in (("nonlinear leaf (0 dense children)", dense & (d[0, :, 6] == 0) & ((d[0, :, 7] & 1) == 1)),
                   ("super-leaf (HPF_SLEAF=1)", dense & (d[0, :, 6] >= 100)),
                   ("linear dense bus, lazy children only", dense & (d[0, :, 6] == 0) & ((d[0, :, 7] & 1) == 0)),
                   ("linear dense bus, 1-2 dense children", dense & (d[0, :, 6] >= 1) & (d[0, :, 6] <= 2) & ((d[0, :, 7] & 1) == 0)),
                   ("nonlinear bus, 1-2 dense children", dense & (d[0, :, 6] >= 1) & (d[0, :, 6] <= 2) & ((d[0, :, 7] & 1) == 1)),
                   (">=4 dense children", dense & (d[0, :, 6] >= 4) & (d[0, :, 6] < 100))):
    if sel.sum() == 0:
        continue
    x = d[:, sel, :6].reshape(-1, 6)
    med = np.median(x, axis=0)
    sub = d[:, sel, 1].reshape(-1)
    packed = sub.max() > (1 << 20)      # multi-wave kernel: slot 1 holds packed sub-phases (units of 16 cycles)
    gsp = d[:, sel, 2].reshape(-1)
    osp = d[:, sel, 4].reshape(-1)
    if packed:
        med[1] = 0
        med[2] = 0
        med[4] = 0

    print("%-38s n=%4d  " % (label, sel.sum()) + "  ".join("%s %7.0f" % (nm, v) for nm, v in zip(names, med)) + "   total %8.0f" % med.sum())
    if packed:
        print("      assembly split: roles up to barrier 1 %6.0f   reduction + Y_N %6.0f"
              % tuple(np.median((sub >> sh) & 0xffff) * 16 for sh in (0, 16)))
        print("      wave 0 up to barrier 1: node record %6.0f   base diagonal %6.0f   its linear children %6.0f   wait at barrier %6.0f"
              % tuple(np.median((osp >> sh) & 0xffff) * 16 for sh in (0, 16, 32, 48)))
        print("      GJ split (wave 0, sums over the block steps): owner work of its 4 steps %6.0f   barrier waits %6.0f   post-barrier sections %6.0f"
              % tuple(np.median((gsp >> sh) & 0xfffff) * 16 for sh in (0, 20, 40)))


# wave 0's pre-barrier phases by the number of 2x2-algebra children of the bus (o[7] >> 1)
gj = dense & (d[0, :, 6] < 100)
if gj.any() and (d[:, gj, 1].max() > (1 << 20)):
    nl_ = (d[0, :, 7] >> 1) & 0x7fff
    print("Gauss-Jordan buses by number of 2x2-algebra children: wave 0 base diagonal / children section / wait at barrier 1 (cycles, median)")
    for v in sorted(set(nl_[gj])):
        sel = gj & (nl_ == v)
        osp = d[:, sel, 4].reshape(-1)
        print("   %2d children  n=%3d   %6.0f  %6.0f  %6.0f" % ((v, sel.sum()) + tuple(np.median((osp >> sh) & 0xffff) * 16 for sh in (16, 32, 48))))
    print("... by lazy children (lazy leaves | lazy bordered children): children loop / B0 issue + lazy loads + MFMA / rest up to barrier 1 (lazy bordered children)")
    o7 = d[:, :, 7]
    for lz in (0, 1):
        for nz in (0, 1, 2):
            sel = gj & (((d[0, :, 7] >> 50) & 1) == lz) & ((((d[0, :, 7] >> 48) & 1) + ((d[0, :, 7] >> 49) & 1)) == nz)
            if not sel.any():
                continue
            a = np.median((o7[:, sel] >> 16) & 0xffff) * 16
            bb = np.median((o7[:, sel] >> 32) & 0xffff) * 16
            tot = np.median((d[:, sel, 4] >> 32) & 0xffff) * 16
            print("   lazy %d  bordered children %d  n=%3d   %6.0f  %6.0f  %6.0f   (section %6.0f)" % (lz, nz, sel.sum(), a, bb, tot - a - bb, tot))
