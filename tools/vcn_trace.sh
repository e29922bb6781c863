#!/bin/bash
# usage (GPU box, repo root): tools/vcn_trace.sh  -> per-dispatch kernel durations of one VCN_VC eval forward (64 objects)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cat > /tmp/vcn_run.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests", "golden"))
import numpy as np, torch
import seevcn_amd.synth as synth, seevcn_amd.vcn as V
objs, _ = synth.make_object_batch(64, seed=1000)
net = V.MODELS.build({"NAME": "VCN_VC"}).cuda().eval()
x = torch.from_numpy(objs).cuda()
for _ in range(6):
    out = net({"input": x})
torch.cuda.synchronize()
PY
rm -rf gpurun_out/vcn_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/vcn_trace -o vt -- python3 /tmp/vcn_run.py > gpurun_out/vcn_trace.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/vcn_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 6
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print("%-70s grid %9s wg %5s  %8.1f us  (start +%.1f us)" % (r["Kernel_Name"][:70], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size",""), r.get("Workgroup_Size_X", r.get("Workgroup_Size","")), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, (int(r["Start_Timestamp"]) - t0) / 1e3))
print("span %.1f us" % ((int(last[-1]["End_Timestamp"]) - t0) / 1e3))
PY
rm -rf gpurun_out/vcn_trace
