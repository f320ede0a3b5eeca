#!/bin/bash
# A/B of library builds on BASELINE config 5 (10 000 buses, K = 49) in ONE gpurun call:  bash tools/ab_config5.sh "<lib.so> ..." <S...>
export HPF_ENV_SWITCHES=1
cd "$GRAFT_REPO_ROOT"
LIBS="$1"; shift
for rep in 1 2; do
  for L in $LIBS; do
    for S in "$@"; do
      HPF_LIB_PATH=$L timeout -k 10 200 python bench.py --buses 10000 --hmax 99 --scenarios $S --steps 10 --warmup 3 --repeats 3 --cpu-iters 0 --no-finish --no-probe --no-one-group --no-single --sweep-1gpu 0 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$L S=$S: %.3f ms/step' % j['ms_per_step'])"
    done
  done
done
