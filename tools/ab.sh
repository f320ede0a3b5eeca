#!/bin/bash
# A/B of builds / switches in ONE gpurun call (same box): bash tools/ab.sh <S...>   (-> stdout)
# tools/bin/libhpf_prev.so = the library built from the previous commit (git stash; build.py; cp; git stash pop)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  echo "== prev lib"; HPF_LIB_PATH=$PWD/tools/bin/libhpf_prev.so timeout -k 10 150 python tools/scale_S.py "$@" | grep "groups=3\|S=    1"
  echo "== new";  timeout -k 10 150 python tools/scale_S.py "$@" | grep "groups=3\|S=    1"
done
