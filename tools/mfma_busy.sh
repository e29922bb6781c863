#!/bin/bash
# usage (on the GPU box, from the repo root): tools/mfma_busy.sh <tag> [bench args...]
# One rocprofv3 --pmc pass (SQ + GRBM counters only, no trace domain) over the bench command -> gpurun_out/<tag>_mfma_busy.json:
# per kernel, matrix-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) x 1024 SIMDs).
# (SQ_VALU_MFMA_BUSY_CYCLES counts pipe cycles summed over SIMDs: 32 per v_mfma_f32_16x16x4_f32, 64 per v_mfma_f32_32x32x2_f32 -- checked
#  against SQ_INSTS_MFMA; rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs, MI355X_MICROARCH.md "DVFS give-back".)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
rm -rf gpurun_out/${TAG}_pmc_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv \
  -d gpurun_out/${TAG}_pmc_mfma -o pmc -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-side-modes "$@" > gpurun_out/${TAG}_pmc_mfma.log 2>&1
python3 - <<PY
import csv, glob, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(collections.Counter)
for f in glob.glob("gpurun_out/${TAG}_pmc_mfma/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); calls[k][row["Counter_Name"]] += 1
out = {}
for k, d in agg.items():
    if d.get("SQ_INSTS_MFMA", 0) <= 0: continue
    n = max(calls[k].values())
    gui = d["GRBM_GUI_ACTIVE"] / 8.0
    out[k] = {"dispatches": n, "mfma_busy_frac": d["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024.0) if gui else None,
              "mfma_insts_per_dispatch": d["SQ_INSTS_MFMA"] / n, "mfma_busy_cycles_per_dispatch": d["SQ_VALU_MFMA_BUSY_CYCLES"] / n,
              "gui_active_cycles_per_xcd_per_dispatch": gui / n, "wave_cycles_per_dispatch": d.get("SQ_WAVE_CYCLES", 0) / n,
              "wait_inst_any_frac_of_wave_cycles": d.get("SQ_WAIT_INST_ANY", 0) / max(d.get("SQ_WAVE_CYCLES", 0), 1),
              "wait_any_frac_of_wave_cycles": d.get("SQ_WAIT_ANY", 0) / max(d.get("SQ_WAVE_CYCLES", 0), 1)}
json.dump({"command": "python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-side-modes $*",
           "formula": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) * 1024 SIMDs)", "kernels": out},
          open("gpurun_out/${TAG}_mfma_busy.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_dispatch"] * kv[1]["dispatches"])[:14]:
    print(f'{k[:80]:80s} n={v["dispatches"]:4d} mfma busy {100 * (v["mfma_busy_frac"] or 0):5.1f} %')
PY
rm -rf gpurun_out/${TAG}_pmc_mfma
