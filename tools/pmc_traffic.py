#!/usr/bin/env python3
"""HBM-side traffic of the bench kernels from rocprofv3 PMC passes (tools/measure.sh <tag> pmc: FETCH_SIZE and WRITE_SIZE in two
separate --pmc passes with --kernel-trace only), corrected as MI355X_MICROARCH.md (HBM) prescribes and as the known-bytes
calibration of THIS access pattern confirms (tools/pmc_calib.hip -> profiles/r02/pmc_calib.json: FETCH_SIZE reads 0.500 of the bytes
of 16 B/lane, 8 B/lane and tile-image streaming loads alike; WRITE_SIZE reads 1.000 of the stores of all three patterns):
    bytes fetched = FETCH_SIZE [KB] x 1024 x 2,    bytes written = WRITE_SIZE [KB] x 1024.
Infinity-Cache hits are counted (fabric-side requests of the L2), so per-model images served on-die are in `fetch` too.

    python tools/pmc_traffic.py profiles/r02/pmc_traffic_<tag>.json [steps_profiled=7] [calib.json]
"""
import collections
import csv
import glob
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"(k_\w+)(<[^>]*>)?", name)
    if not m:
        return None
    return m.group(1) + (m.group(2) or "").replace(" ", "")


def main():
    out_path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 17
    calib = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else os.path.join(REPO, "profiles", "r02", "pmc_calib.json")
    prefix = sys.argv[4] if len(sys.argv) > 4 else "pmc_"           # gpurun_out/<prefix>FETCH_SIZE, <prefix>WRITE_SIZE
    command = sys.argv[5] if len(sys.argv) > 5 else None
    ff, wf = 2.0, 1.0
    if os.path.exists(calib):
        ck = json.load(open(calib))["kernels"]
        ff = 1.0 / ck["k_tile_load"]["FETCH_SIZE_over_known"]
        wf = 1.0 / ck["k_tile_store"]["WRITE_SIZE_over_known"]
    res = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = max(glob.glob(os.path.join(REPO, "gpurun_out", f"{prefix}{c}", "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                agg[k][0] += 1
                agg[k][1] += float(r["Counter_Value"])
        res[c] = agg
    out = {"format": 2,
           "command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace --output-format csv -- python3 bench.py --steps 5 --warmup 2 --repeats 1 "
                      "--cpu-iters 0 --no-finish --no-probe --no-one-group --no-single --sweep-1gpu 0 (two separate passes; tools/measure.sh <tag> pmc; 17 steps of four scenario groups: 2 warm-up + 5 timed + 2 x 5 timing legs)",
           "steps_profiled": steps, "fetch_factor": ff, "write_factor": wf, "calibration": os.path.relpath(calib, REPO),
           "per_step_bytes": {}, "per_launch_bytes": {}}
    for k, (nd, kb) in sorted(res["FETCH_SIZE"].items()):
        wkb = res["WRITE_SIZE"].get(k, [0, 0.0])[1]
        fe, wr = kb * 1024 * ff, wkb * 1024 * wf
        if "<true>" in k or k in ("k_polar<false>", "k_polar<true>", "k_init_voltages", "k_fill", "k_finalize", "k_compact"):
            continue                                            # set-up kernels (pf, initial state), not part of a timed step
        out["per_step_bytes"][k] = {"fetch": fe / steps, "write": wr / steps, "dispatches_per_step": nd / steps}
        out["per_launch_bytes"][k] = {"fetch": fe / nd, "write": wr / nd}
    if command:
        out["command"] = command
    import hashlib
    lib = os.path.join(REPO, "harmonic-power-flow_amd", "libhpf.so")
    out["lib_sha16"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16] if os.path.exists(lib) else None
    g = [k for k in out["per_launch_bytes"] if k.startswith("k_level<")] or \
        [k for k in out["per_launch_bytes"] if k.startswith("k_factor_q<") and k.endswith("false>")]    # (k_level: one launch per level)
    if g:
        out["per_launch_bytes"]["k_factor_q_general"] = out["per_launch_bytes"][g[0]]
    tot = sum(v["fetch"] + v["write"] for v in out["per_step_bytes"].values())
    out["step_total_bytes"] = tot
    out["note"] = ("per launch / per step: FETCH_SIZE x 1024 x %.3f + WRITE_SIZE x 1024 x %.3f (factors from the known-bytes calibration, = the "
                   "guide's gfx950 correction); fabric-side counters: reads served by the Infinity Cache are included" % (ff, wf))
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(out, open(out_path, "w"), indent=1)
    for k, v in sorted(out["per_step_bytes"].items(), key=lambda kv: -(kv[1]["fetch"] + kv[1]["write"])):
        print("%-28s fetch %8.1f MB  write %8.1f MB  per step (%5.1f launches)" % (k, v["fetch"] / 1e6, v["write"] / 1e6, v["dispatches_per_step"]))
    print("step total %.2f GB" % (tot / 1e9))


if __name__ == "__main__":
    main()
