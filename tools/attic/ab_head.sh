for w in "$@"; do
  echo "== $w"; SHACIRA_HIP_LIB=$GRAFT_REPO_ROOT/shacira_amd/lib/variants/head.so timeout 200 python tools/r3_ab.py $w - 2>&1 | tail -1 | sed 's/^/head: /'
  timeout 200 python tools/r3_ab.py $w - 2>&1 | tail -1 | sed 's/^/new:  /'
done
