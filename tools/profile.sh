#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile.sh <tag> [steps] [extra bench.py arguments, e.g. --config pvrcnn]
# rocprofv3 --kernel-trace --stats over the bench command (no PMC in this pass); leaves gpurun_out/<tag>_kernel_stats.csv
TAG=$1
STEPS=${2:-20}
shift; shift
EXTRA="$@"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
rm -rf gpurun_out/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python3 bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-side-modes $EXTRA > gpurun_out/prof_$TAG.log 2>&1
F=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp "$F" gpurun_out/${TAG}_kernel_stats.csv
T=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
python3 tools/steady_stats.py "$T" $((STEPS + 5)) 5 "FusedSgd|fused_sgd" gpurun_out/${TAG}_step_sequence.txt > gpurun_out/${TAG}_steady_kernels.csv
head -1 gpurun_out/${TAG}_steady_kernels.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(l for l in open("gpurun_out/${TAG}_steady_kernels.csv") if not l.startswith("#")))
for r in rows[:int("${TOP:-28}")]:
    print("%-78s %7.1f calls/step %8.1f us -> %.3f ms/step" % (r["Name"][:78], float(r["CallsPerStep"]), float(r["AverageNs"]) / 1e3, float(r["NsPerStep"]) / 1e6))
PY
echo "---- whole process (rocprofv3 --stats) ----"
tail -1 gpurun_out/prof_$TAG.log | cut -c1-200
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/${TAG}_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
steps = $STEPS + 5
print("GPU busy ms/step (incl. warmup steps): %.3f" % (tot / steps / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print("%-78s %7.1f calls/step %8.1f us -> %.3f ms/step" % (r["Name"][:78], int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / steps / 1e6))
PY
rm -rf gpurun_out/prof_$TAG
