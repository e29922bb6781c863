#!/bin/bash
# usage (GPU box): tools/ab_env.sh <out file> <steps> "ENV=a" "ENV=b" ...     per setting: the per-layer micro-benchmark (MICRO_MODE, default fwd) once and the
# bench line's ms/step twice, alternating
cd $GRAFT_REPO_ROOT
out=$1; steps=$2; shift 2
: > $out
for rep in 1 2; do
for n in "$@"; do
  echo "== $n (rep $rep)" >> $out
  [ $rep = 1 ] && env $n MODE=${MICRO_MODE:-fwd} timeout 300 python3 tools/spconv_micro.py 2>&1 | grep -E "^(subm|spconv|down|sum)" | sed 's/rulebook.*plan *[0-9.]* us |//' >> $out
  env $n python3 bench.py --steps $steps --warmup 20 --no-cpu-baseline --no-side-modes 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms/step', d['ms_per_step'], 'roofline', d['roofline']['frac'])" >> $out
done; done
cat $out
