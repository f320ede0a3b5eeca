#!/usr/bin/env python3
"""HBM traffic of the bench kernels from rocprofv3 PMC passes.

On the GPU box (two separate passes, counters only with --kernel-trace, as MI355X_MICROARCH.md / the pool rules require):
    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
    for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- \
        python3 bench.py --steps 3 --warmup 1 --cpu-iters 0 --no-finish; done
Then here:  python tools/pmc_traffic.py profiles/r01/pmc_traffic.json [steps_profiled=4]
"""
import collections
import csv
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("k_sleaf_batch", "k_sleaf_back_batch", "k_leaf_back_batch", "k_leaf_batch", "k_factor_q", "k_back_q", "k_lin_level", "k_chain", "k_mismatch", "k_update", "k_factor_w", "k_back_w", "k_tree_factor",
        "k_tree_back")


def main():
    out_path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    res = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = max(glob.glob(os.path.join(REPO, "gpurun_out", f"pmc_{c}", "*", "*counter_collection.csv")), key=os.path.getmtime)
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            for key in KEYS:
                if key in r["Kernel_Name"]:
                    agg[key][0] += 1
                    agg[key][1] += float(r["Counter_Value"])
        res[c] = {k: {"dispatches": v[0], "sum_counter_KB": v[1]} for k, v in agg.items()}
    out = {"command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python3 bench.py --steps 3 --warmup 1 "
                      "--cpu-iters 0 --no-finish (two separate passes)",
           "steps_profiled": steps, "raw": res, "per_step_bytes": {}}
    for k in res["FETCH_SIZE"]:
        fe = res["FETCH_SIZE"][k]["sum_counter_KB"] / steps * 1024
        wr = res["WRITE_SIZE"].get(k, {"sum_counter_KB": 0.0})["sum_counter_KB"] / steps * 1024
        out["per_step_bytes"][k] = {"fetch_raw": fe, "fetch_x2_gfx950_wide_stream_correction": 2 * fe, "write": wr}
    # calibration of the store counter on THIS access pattern (MI355X_MICROARCH.md: widths other than 16 B/lane are
    # uncalibrated): k_update writes exactly Vm, Va (8 B/lane) and U, E (16 B/lane) of every (bus, harmonic, scenario)
    n_b, n_h, n_s = (int(a) for a in (sys.argv[3:6] if len(sys.argv) >= 6 else (1000, 26, 128)))
    known = 48.0 * n_b * n_h * n_s
    upd = out["per_step_bytes"].get("k_update")
    if upd and upd["write"] > 0:
        cal = known / upd["write"]
        out["write_calibration"] = {"kernel": "k_update", "known_bytes": known, "counter_bytes": upd["write"], "factor": cal,
                                    "note": "WRITE_SIZE over-counts the stores of these kernels; the tile-image LOADS calibrate at 1.0 "
                                            "on k_back_q (raw FETCH_SIZE = inverse tiles + operands), with 8-byte accesses per "
                                            "lane (up to v14) and with 16-byte accesses (v15: 0.83 -> 0.81 GB) alike"}
        for k in out["per_step_bytes"]:
            out["per_step_bytes"][k]["write_calibrated"] = out["per_step_bytes"][k]["write"] * cal
    out["note"] = ("FETCH_SIZE/WRITE_SIZE are KB. MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads exactly 1/2 of the bytes of a "
                   "wide coalesced 16 B/lane stream and other widths are uncalibrated; on the tile images of the factor / back kernels "
                   "(16 B/lane pairs + one 8 B/lane row group per lane, 1 KB / 512 B rows) the RAW value matches the known bytes "
                   "(see write_calibration.note), so raw and doubled values are both given and the raw one is used.")
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out["per_step_bytes"], indent=1))


if __name__ == "__main__":
    main()
