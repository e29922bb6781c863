#!/bin/bash
# usage (GPU box, repo root): tools/roofline_pass.sh <tag>
# rocprofv3 --kernel-trace --stats over `bench.py --roofline-only 10`: the launches the bench line's `roofline` block times (the step's own launch
# lists on one stream, weight gradients in line), and only those plus warm-up.  Leaves gpurun_out/<tag>_roofline_pass_kernel_stats.csv and prints the
# time-weighted average duration of the dominant family beside the line's own avg_launch_ms: the two must agree.
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
rm -rf gpurun_out/prof_rp_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rp_$TAG -o rp -- python3 bench.py --roofline-only 10 --warmup 5 > gpurun_out/${TAG}_roofline_pass.json 2> gpurun_out/prof_rp_$TAG.log
T=$(find gpurun_out/prof_rp_$TAG -name "*kernel_trace.csv" | head -1)
F=$(find gpurun_out/prof_rp_$TAG -name "*kernel_stats.csv" | head -1)
cp "$F" gpurun_out/${TAG}_roofline_pass_kernel_stats.csv
python3 - "$T" gpurun_out/${TAG}_roofline_pass.json <<'PY' | tee gpurun_out/${TAG}_roofline_pass.txt
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
line = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
reps, per_step = line["reps"], line["launches_per_step"]
fam = [r for r in rows if r["Kernel_Name"].startswith("void k_spconv_rs3<4, 4,")]
fam.sort(key=lambda r: int(r["Start_Timestamp"]))
last = fam[-reps * per_step:]                          # the measurement passes are the last `reps` steps of the process
avg = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / len(last) / 1e3
print(f"k_spconv_rs3<4, 4, *>: {len(last)} launches of the {reps} measurement passes ({per_step} per step): average {avg:.1f} us by rocprofv3 kernel trace; "
      f"the line's event-timed avg_launch_ms = {line['avg_launch_ms'] * 1e3:.1f} us; frac {line['frac']}")
by = {}
for r in last:
    by.setdefault(r["Kernel_Name"].split("(")[0], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(by.items()):
    print(f"  {k:70s} {len(v) // reps:2d} per step  {sum(v) / len(v):7.1f} us")
PY
rm -rf gpurun_out/prof_rp_$TAG
