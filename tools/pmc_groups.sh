#!/bin/bash
# usage: tools/pmc_groups.sh <script> "<group1>" "<group2>" ...   (each group = space-separated counters, its own pass)
script=$1; shift
k=0
for g in "$@"; do
  k=$((k+1))
  echo "--- group $k: $g"
  timeout 240 bash tools/pmc.sh pmcg_$k "$g" $script 3 2>&1 | grep -A12 "bin_scatter\|bin_consume" | head -40
done
