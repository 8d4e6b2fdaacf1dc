# usage: ab_multi.sh "<variant> <variant> ..." <workload>...   -- backward times of several variant libraries, each twice, same box
vs=$1; shift
for w in "$@"; do
  echo "== $w ${R3_DTYPE:-f32}"
  for rep in 1 2; do
    for v in $vs; do
      SHACIRA_HIP_LIB=$GRAFT_REPO_ROOT/shacira_amd/lib/variants/$v.so timeout 200 python tools/r3_ab.py $w - 2>&1 | tail -1 | sed "s/^/$v: /" | cut -c1-75
    done
  done
done
