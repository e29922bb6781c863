#!/bin/bash
# usage (on the GPU box, from the repo root): tools/traffic.sh <tag>
# Two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with trace domains) over the bench command,
# then per-kernel per-dispatch HBM bytes -> gpurun_out/<tag>_traffic.json (copy into profiles/ to have bench.py report it).
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
for CNT in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/${TAG}_pmc_$CNT
  rocprofv3 --pmc $CNT --output-format csv -d gpurun_out/${TAG}_pmc_$CNT -o pmc -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-side-modes > gpurun_out/${TAG}_pmc_$CNT.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(lambda: collections.Counter())
for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/${TAG}_pmc_%s/**/*counter_collection.csv" % cnt, recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != cnt: continue
            k = row["Kernel_Name"]
            agg[k][cnt] += float(row["Counter_Value"]); calls[k][cnt] += 1
out = {}
for k, d in agg.items():
    if not k.startswith(("void k_", "k_")): continue
    n = max(calls[k].values())
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; gfx950 tallies 128-B read requests at 64 B (MI355X_MICROARCH.md "HBM"): x2 on reads
    fetch = d.get("FETCH_SIZE", 0.0) / max(calls[k]["FETCH_SIZE"], 1) * 1024 * 2
    write = d.get("WRITE_SIZE", 0.0) / max(calls[k]["WRITE_SIZE"], 1) * 1024
    out[k] = {"dispatches": n, "fetch_bytes_per_dispatch": fetch, "write_bytes_per_dispatch": write, "hbm_bytes_per_dispatch": fetch + write}
json.dump({"command": "python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-side-modes", "units": "bytes; FETCH_SIZE KiB x1024 x2 (gfx950), WRITE_SIZE KiB x1024",
           "kernels": out}, open("gpurun_out/${TAG}_traffic.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_dispatch"] * kv[1]["dispatches"])[:12]:
    print(f'{k[:70]:70s} n={v["dispatches"]:4d} fetch {v["fetch_bytes_per_dispatch"]/1e6:9.2f} MB write {v["write_bytes_per_dispatch"]/1e6:9.2f} MB')
PY
