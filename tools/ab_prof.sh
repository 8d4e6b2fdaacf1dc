#!/bin/bash
# usage: tools/ab_prof.sh <script> v1 v2 ...  -> per-kernel rocprof summary of each variant
script=$1; shift
for v in "$@"; do
  if [ "$v" = default ]; then unset SHACIRA_HIP_LIB; else export SHACIRA_HIP_LIB=$GRAFT_REPO_ROOT/shacira_amd/lib/variants/$v.so; fi
  echo "=== $v"
  timeout 300 bash tools/prof.sh ab_$v $script 2>&1 | grep -v "^$" | head -12
done
