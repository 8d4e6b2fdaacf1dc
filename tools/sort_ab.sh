#!/bin/bash
# usage: tools/sort_ab.sh <variant>...  -- rocprof kernel times of the forward's sort passes, shipped library and variants
for v in shipped "$@"; do
  if [ "$v" = shipped ]; then unset SHACIRA_HIP_LIB; else export SHACIRA_HIP_LIB=$GRAFT_REPO_ROOT/shacira_amd/lib/variants/$v.so; fi
  echo "== $v"
  bash tools/prof.sh ab_$v tools/fwd_only.py fwd -1 3 10 | grep -E "psort|rows|level_pair" | sed -E 's/\(.*calls/ calls/'
done
