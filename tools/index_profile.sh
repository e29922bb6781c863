#!/bin/bash
# usage (on the GPU box, from the repo root): tools/index_profile.sh <tag>   -> gpurun_out/<tag>_index_micro.txt: wall time per build + the
# index kernels' own durations (rocprofv3 --kernel-trace --stats of tools/index_micro.py: nothing else runs on the GPU)
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
rm -rf gpurun_out/ix_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ix_$TAG -o ix -- python3 tools/index_micro.py > gpurun_out/${TAG}_index_micro.txt 2>&1
F=$(find gpurun_out/ix_$TAG -name "*kernel_stats.csv" | head -1)
python3 - >> gpurun_out/${TAG}_index_micro.txt <<PY
import csv
rows = [r for r in csv.DictReader(open("$F")) if any(k in r["Name"] for k in ("k_chain_", "k_batch_", "k_plan_", "k_sparse_", "k_subm_", "k_cellmap", "k_invert", "k_index_", "k_scan_"))]
print("kernel, calls, average us, total us  (46 builds: 23 batched + 23 layer by layer)")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    print(f'{r["Name"][:70]:70s} {int(r["Calls"]):5d} {float(r["AverageNs"]) / 1e3:8.1f} {float(r["TotalDurationNs"]) / 1e3:10.1f}')
b = sum(float(r["TotalDurationNs"]) for r in rows if any(k in r["Name"] for k in ("k_chain_", "k_batch_", "k_plan_region_batch")))
print(f"batched build: {b / 23 / 1e3:.1f} us of index kernels per build")
l = sum(float(r["TotalDurationNs"]) for r in rows if not any(k in r["Name"] for k in ("k_chain_", "k_batch_", "k_plan_region_batch")))
print(f"layer-by-layer build: {l / 23 / 1e3:.1f} us of index kernels per build")
PY
rm -rf gpurun_out/ix_$TAG
tail -30 gpurun_out/${TAG}_index_micro.txt
