#!/bin/bash
# Measurement set of one code version on the GPU box (run through gpurun):   bash tools/measure.sh <tag> [bench|stats|pmc|calib|mfma ...]
#   bench  : python bench.py (default K/W)                        -> gpurun_out/<tag>_bench.json
#   stats  : rocprofv3 --kernel-trace --stats of the same command -> gpurun_out/<tag>_stats/   (copy *kernel_stats.csv to profiles/)
#   pmc    : FETCH_SIZE / WRITE_SIZE, two separate --pmc passes   -> gpurun_out/pmc_FETCH_SIZE, pmc_WRITE_SIZE  (tools/pmc_traffic.py)
#   calib  : the same two counters on tools/pmc_calib.hip          -> gpurun_out/calib_*                          (tools/pmc_calib_report.py)
#   mfma   : MFMA / LDS counters                                  -> gpurun_out/pmc_SQ_*
# Counters are collected in their own runs with --kernel-trace only (pool rule); the program itself follows `--`.
export HPF_ENV_SWITCHES=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-run}; shift
WHAT=${@:-bench stats}
PROF_ARGS="--steps 5 --warmup 2 --repeats 1 --cpu-iters 0 --no-finish --no-probe --no-one-group --no-single --sweep-1gpu 0"
for w in $WHAT; do
  case $w in
    bench) timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 400 gpurun_out/${TAG}_bench.json; echo;;
    stats) rm -rf gpurun_out/${TAG}_stats; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- python3 bench.py $PROF_ARGS > gpurun_out/${TAG}_stats.log 2>&1; ls gpurun_out/${TAG}_stats/*/ | head -5;;
    pmc) for c in FETCH_SIZE WRITE_SIZE; do rm -rf gpurun_out/pmc_$c; timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py $PROF_ARGS > gpurun_out/pmc_$c.log 2>&1; done;;
    calib) for c in FETCH_SIZE WRITE_SIZE; do rm -rf gpurun_out/calib_$c; timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/calib_$c -- tools/bin/pmc_calib > gpurun_out/calib_$c.log 2>&1; done; tail -1 gpurun_out/calib_WRITE_SIZE.log;;
    mfma) for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_F64 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do tag=$(echo $c | cut -d' ' -f1); rm -rf gpurun_out/pmc_$tag; timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py $PROF_ARGS > gpurun_out/pmc_$tag.log 2>&1; done;;
  esac
done
