#!/bin/bash
# usage (GPU box, repo root): tools/wgrad_ab.sh <tag>
# A/B of the weight gradient -- (row chunk, offset) workgroups against equal pieces -- per layer of the benchmarked backbone, and the per-wave traces of
# both forms on the 64 -> 64 layers (tools/wgrad_trace.py).  Leaves gpurun_out/<tag>_wgrad_ab.txt.
TAG=${1:-wgrad}
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/${TAG}_wgrad_ab.txt
: > $OUT
for planned in 0 1; do
  echo "== SEEVCN_WGRAD_PLANNED=$planned" >> $OUT
  SEEVCN_WGRAD_PLANNED=$planned MODE=wgrad python3 tools/spconv_micro.py 2>&1 | grep -E "wgrad" | grep -v "^sum" | sed -e 's/rulebook.*| wgrad/| wgrad/' >> $OUT
done
for layer in subm3 subm4; do
  for planned in 0 1; do
    echo >> $OUT
    echo "== tools/wgrad_trace.py LAYER=$layer SEEVCN_WGRAD_PLANNED=$planned" >> $OUT
    SEEVCN_WGRAD_PLANNED=$planned LAYER=$layer python3 tools/wgrad_trace.py 2>&1 | grep -v amdgpu.ids >> $OUT
  done
done
tail -5 $OUT | cut -c1-200
