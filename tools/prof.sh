#!/bin/bash
# usage: tools/prof.sh <name> <script> [env...]   -> gpurun_out/<name>_stats.csv (kernel summary)
name=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$name
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/$name -o $name --output-format csv -- python3 "$@" > gpurun_out/$name/log.txt 2>&1
f=$(find gpurun_out/$name -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:25]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']}")
PY
