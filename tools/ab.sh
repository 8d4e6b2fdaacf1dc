#!/bin/bash
# usage: tools/ab.sh <script> v1 v2 ...   (variants under shacira_amd/lib/variants; "default" = the shipped build)
script=$1; shift
for v in "$@"; do
  if [ "$v" = default ]; then unset SHACIRA_HIP_LIB; else export SHACIRA_HIP_LIB=$GRAFT_REPO_ROOT/shacira_amd/lib/variants/$v.so; fi
  timeout 300 python3 $script 2>&1 | tail -3
done
