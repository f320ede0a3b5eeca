#!/usr/bin/env python3
"""Timing-only ablation of the block-tree factor kernel phases (HPF_DEBUG_ABLATE bitmask: 1 skip Gauss-Jordan loop,
2 skip children pull, 4 skip block assembly, 8 skip the A^-1 store, 16 phase stamps, 32 split assembly kernel).  Results of ablated runs are numerically invalid;
only the phase times are read.  Usage: python tools/ablate.py [masks...]"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
masks = [int(x) for x in sys.argv[1:]] or [0, 1, 2, 4, 8, 15]
for m in masks:
    env = dict(os.environ, HPF_DEBUG_ABLATE=str(m))
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "10", "--warmup", "2", "--cpu-iters", "0",
                          "--no-finish"], env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        ph = d["phase_ms_per_step"]
        print("ablate %2d groups %s: factor %.3f ms  back %.3f ms  step %.3f ms  -> %.0f it/s" % (
            m, d["roofline"].get("concurrent_groups"), ph["solve"], ph.get("back", 0), d["ms_per_step"], d["value"]), flush=True)
    except Exception as e:
        print("ablate", m, "failed", e, out.stderr[-500:], flush=True)
