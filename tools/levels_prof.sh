#!/bin/bash
for r in "$@"; do
  lo=${r%%:*}; hi=${r##*:}
  echo "== levels $r"
  bash tools/prof.sh lv_${lo}_${hi} tools/levels_prof.py $lo $hi | grep -E "scatter|consume|front16|transpose|count" | sed -E 's/\(.*calls/ calls/'
done
