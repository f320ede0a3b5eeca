#!/bin/bash
# A/B of library builds in ONE gpurun call (same box), interleaved twice:  bash tools/ab2.sh "<lib.so> [<lib.so> ...]" <S...>     (groups: HPF_SCALE_GROUPS, default 4)
export HPF_ENV_SWITCHES=1
cd "$GRAFT_REPO_ROOT"
LIBS="$1"; shift
export HPF_SCALE_GROUPS=${HPF_SCALE_GROUPS:-4}
for rep in 1 2 3; do
  for L in $LIBS; do echo "== $L"; HPF_LIB_PATH=$L timeout -k 10 150 python tools/scale_S.py "$@" | grep "S="; done
done
