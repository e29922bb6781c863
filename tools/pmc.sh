#!/bin/bash
# usage: tools/pmc.sh <outdir under gpurun_out> <kernel-name-regex> -- <python script args...>
# Runs separate rocprofv3 --pmc passes (counters never combined with trace domains) and prints per-kernel sums.
OUT=$1; REGEX=$2; shift 3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for CNT in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAVES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE GRBM_TA_BUSY" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $CNT --output-format csv -d $R/gpurun_out/$OUT/p$i -- python3 "$@" > $R/gpurun_out/$OUT.p$i.log 2>&1
done
python3 - <<PY
import csv, glob, re, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob("$R/gpurun_out/$OUT/p*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if not re.search(r"$REGEX", k): continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[(k, row["Counter_Name"])] += 1
for k, d in agg.items():
    print(k[:90])
    for c, v in sorted(d.items()):
        n = calls[(k, c)]
        print(f"   {c:36s} total {v:16.0f}  per-dispatch {v / n:14.1f}  (n={n})")
PY
