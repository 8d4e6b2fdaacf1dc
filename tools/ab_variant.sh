# usage: ab_variant.sh <variant name> <workload>...   -- timings of the shipped library and of lib/variants/<name>.so, same box
v=$1; shift
for w in "$@"; do
  echo "== $w"; timeout 200 python tools/r3_ab.py $w - 2>&1 | tail -1 | sed 's/^/shipped: /'
  SHACIRA_HIP_LIB=$GRAFT_REPO_ROOT/shacira_amd/lib/variants/$v.so timeout 200 python tools/r3_ab.py $w - 2>&1 | tail -1 | sed "s/^/$v: /"
done
