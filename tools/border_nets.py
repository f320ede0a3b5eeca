import os, sys
os.environ.setdefault("HPF_ENV_SWITCHES", "1")
sys.path.insert(0, os.getcwd())
import numpy as np
import harmonic_power_flow_amd as hp
from harmonic_power_flow_amd import api
INP = "tests/golden/inputs"
for net, hmax in (("net1", 51), ("net2", 51), ("net3", 51)):
    st = hp.Settings(H_MAX=hmax)
    buses, lines, m, n, c = hp.init_network(os.path.join(INP, net + "_buses.csv"), os.path.join(INP, net + "_lines.csv"), settings=st)
    Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS); NE = hp.import_Norton_Equivalents(buses, True, st, INP)
    dm = api._device_model(buses, Y, NE, True, st.HARMONICS, solver="block_tree")
    dm.set_loads(buses["P"].to_numpy(float), buses["Q"].to_numpy(float)); dm.set_state(None, None, n_scen=1); dm.fund_pf(1e-6, 30)
    it, err, hist = dm.solve(1e-4, 50)
    cs = dm.tree_census()
    print(net, "iterations", it[0], "err %.2e" % err[0], "ties", cs["ties"], "pivoted border systems", cs["border_repivots"])
    dm.close()
