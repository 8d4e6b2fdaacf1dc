#!/bin/bash
# usage: tools/brick_prof.sh <optset>...   -- rocprof kernel times of planned fwd+bwd under SHACIRA_OPTIONS=<optset> ("-" = none)
for o in "$@"; do
  if [ "$o" = "-" ]; then unset SHACIRA_OPTIONS; else export SHACIRA_OPTIONS="$o"; fi
  echo "== $o"
  bash tools/prof.sh bp_$(echo $o | tr -c 'a-zA-Z0-9\n' '_') tools/plan_prof.py 10 ${WORKLOAD:-S1} | grep -E "scatter|consume|brick|front16|psort|rows|level_pair|zero|scan" | sed -E 's/\(.*calls/ calls/'
done
