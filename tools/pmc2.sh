#!/bin/bash
# usage: tools/pmc2.sh <outdir under gpurun_out> <kernel-regex> "<counters pass 1>" ["<counters pass 2>" ...] -- script args   (env passes through)
OUT=$1; REGEX=$2; shift 2
PASSES=()
while [ "$1" != "--" ]; do PASSES+=("$1"); shift; done
shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for CNT in "${PASSES[@]}"; do
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
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); calls[(k, row["Counter_Name"])] += 1
for k, d in agg.items():
    print(k[:90])
    for c, v in sorted(d.items()):
        print(f"   {c:36s} per-dispatch {v / calls[(k, c)]:16.1f}  (n={calls[(k, c)]})")
PY
