#!/bin/bash
# A/B of switches (or of tools/bin/libhpf_prev.so via HPF_LIB_PATH) in ONE gpurun call (same box): bash tools/ab.sh "<ENV=VAL> [<ENV=VAL> ...]" <S...>
export HPF_ENV_SWITCHES=1
cd "$GRAFT_REPO_ROOT"
VS="$1"; shift
for rep in 1 2; do
  for V in $VS; do echo "== $V"; env $V timeout -k 10 150 python tools/scale_S.py "$@" | grep "groups=3\|S=    1"; done
  echo "== default";  timeout -k 10 150 python tools/scale_S.py "$@" | grep "groups=3\|S=    1"
done
