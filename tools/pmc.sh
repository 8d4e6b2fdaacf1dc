#!/bin/bash
# usage: tools/pmc.sh <name> "<counter list>" <script> -> prints per-kernel average counter values (separate --pmc pass)
name=$1; ctrs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$name
rocprofv3 --pmc $ctrs -d gpurun_out/$name -o $name --output-format csv -- python3 "$@" > gpurun_out/$name/log.txt 2>&1
f=$(find gpurun_out/$name -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if not any(t in k for t in ('tiled_', 'level_pair', 'bin_scatter', 'bin_consume', 'ctx_scatter')): continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:28s} avg {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
