#!/bin/bash
# BASELINE config 5 (10 000 buses, K = 49) under rocprofv3: kernel stats + per-launch timeline of one scenario group, S = 1 and S = 16; PMC traffic of
# the b = 100 kernels (two separate --pmc passes).     bash tools/config5_profile.sh <tag> [stats|pmc ...]
export HPF_ENV_SWITCHES=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-c5}; shift
WHAT=${@:-stats}
ARGS="--buses 10000 --hmax 99 --steps 5 --warmup 2 --repeats 1 --cpu-iters 0 --no-finish --no-probe --no-single --sweep-1gpu 0"
export HPF_GROUPS=1
for w in $WHAT; do
  case $w in
    stats) for S in 1 16; do rm -rf gpurun_out/${TAG}_S${S}_trace; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_S${S}_trace -- python3 bench.py $ARGS --scenarios $S > gpurun_out/${TAG}_S${S}_trace.log 2>&1
             python3 tools/timeline.py gpurun_out/${TAG}_S${S}_trace > gpurun_out/${TAG}_S${S}_timeline.txt 2>&1
             cp gpurun_out/${TAG}_S${S}_trace/*/*_kernel_stats.csv gpurun_out/${TAG}_S${S}_kernel_stats.csv 2>/dev/null
             rm -rf gpurun_out/${TAG}_S${S}_trace; tail -n 12 gpurun_out/${TAG}_S${S}_timeline.txt; done;;
    pmc) for c in FETCH_SIZE WRITE_SIZE; do rm -rf gpurun_out/${TAG}_pmc_$c; timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/${TAG}_pmc_$c -- python3 bench.py $ARGS --scenarios 16 > gpurun_out/${TAG}_pmc_$c.log 2>&1; done;;
  esac
done
