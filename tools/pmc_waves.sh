#!/bin/bash
# Wave-level instruction / wait counters of the bench kernels, one rocprofv3 --pmc pass per counter group (kernel-trace only, pool rule):
#   bash tools/pmc_waves.sh <tag> [bench.py args...]   -> gpurun_out/<tag>_pmcw/<group>/ ; tools/pmc_waves.py sums them per kernel
export HPF_ENV_SWITCHES=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
ARGS="--steps 5 --warmup 2 --repeats 1 --cpu-iters 0 --no-finish --no-probe --no-one-group --no-single --sweep-1gpu 0 $@"
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM_RD"; do
  d=gpurun_out/${TAG}_pmcw/g$i; rm -rf $d; mkdir -p $d
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 bench.py $ARGS > $d/run.log 2>&1 || echo "group $i failed: $c"
  find $d -name "*agent_info*" -delete; find $d -name "*kernel_trace*" -delete
  i=$((i+1))
done
python3 tools/pmc_waves.py gpurun_out/${TAG}_pmcw > gpurun_out/${TAG}_pmcw.txt 2>&1
rm -rf gpurun_out/${TAG}_pmcw
cat gpurun_out/${TAG}_pmcw.txt
