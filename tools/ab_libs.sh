#!/bin/bash
# usage (GPU box): tools/ab_libs.sh <out file> <steps> name1 name2 ...   ("default" = the in-tree library, others = see-vcn_amd/lib/variants/libseevcn_hip_<name>.so)
# per library: the per-layer micro-benchmark (forward) and the bench line's ms/step, alternating twice
cd $GRAFT_REPO_ROOT
out=$1; steps=$2; shift 2
: > $out
for rep in 1 2; do
for n in "$@"; do
  lib=""; [ "$n" != "default" ] && lib="$PWD/see-vcn_amd/lib/variants/libseevcn_hip_$n.so"
  echo "== $n (rep $rep)" >> $out
  [ $rep = 1 ] && SEEVCN_LIB=$lib MODE=${MICRO_MODE:-fwd} timeout 300 python3 tools/spconv_micro.py 2>&1 | grep -E "^(subm|spconv|down|sum)" >> $out
  SEEVCN_LIB=$lib python3 bench.py --steps $steps --warmup 20 --no-cpu-baseline --no-side-modes 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms/step', d['ms_per_step'], 'roofline', d['roofline']['frac'])" >> $out
done; done
cat $out
