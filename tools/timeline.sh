#!/bin/bash
# usage (on the GPU box, from the repo root): tools/timeline.sh <tag> [extra bench.py arguments]
# rocprofv3 --kernel-trace of a short bench run; per-queue idle gaps of the last steps -> gpurun_out/<tag>_timeline.txt
TAG=$1
shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
rm -rf gpurun_out/tl_$TAG
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$TAG -o $TAG -- python3 bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-side-modes --no-kernel-rooflines "$@" > gpurun_out/tl_$TAG.log 2>&1
T=$(find gpurun_out/tl_$TAG -name "*kernel_trace.csv" | head -1)
python3 tools/stream_timeline.py "$T" 3 25 > gpurun_out/${TAG}_timeline.txt
rm -rf gpurun_out/tl_$TAG
cat gpurun_out/${TAG}_timeline.txt
