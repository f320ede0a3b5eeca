#!/bin/bash
# per-launch timeline of one scenario group (rocprofv3 --kernel-trace, HPF_GROUPS=1):  bash tools/timeline_run.sh <tag> <scenarios> [ENV=VAL ...]
export HPF_ENV_SWITCHES=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; S=$2; shift; shift
for kv in "$@"; do export "$kv"; done
export HPF_GROUPS=1
rm -rf gpurun_out/${TAG}_trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${TAG}_trace -- python3 bench.py --steps 5 --warmup 2 --repeats 1 --cpu-iters 0 --no-finish --no-probe --no-single --sweep-1gpu 0 --scenarios $S > gpurun_out/${TAG}_trace.log 2>&1
python3 tools/timeline.py gpurun_out/${TAG}_trace > gpurun_out/${TAG}_timeline.txt 2>&1
rm -rf gpurun_out/${TAG}_trace
tail -25 gpurun_out/${TAG}_timeline.txt
