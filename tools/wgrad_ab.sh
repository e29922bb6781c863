#!/bin/bash
# A/B of the weight gradient: (row chunk, offset) workgroups against equal pieces; per-wave traces
cd "${GRAFT_REPO_ROOT:-.}"
for planned in 0 1; do
  echo "== SEEVCN_WGRAD_PLANNED=$planned"
  SEEVCN_WGRAD_PLANNED=$planned MODE=wgrad python tools/spconv_micro.py 2>&1 | grep -E "wgrad" | sed -e 's/rulebook.*| wgrad/| wgrad/'
done
for layer in subm3 subm4; do
  LAYER=$layer python tools/wgrad_trace.py 2>&1 | grep -v amdgpu.ids > gpurun_out/wgrad_trace_${layer}_eq3.txt
done
