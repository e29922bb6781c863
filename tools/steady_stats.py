#!/usr/bin/env python3
"""Per-kernel statistics of the STEADY steps of a rocprofv3 kernel trace of bench.py.

  tools/steady_stats.py <kernel_trace.csv> <total steps incl. warm-up> <warm-up steps to drop> [marker regex [sequence file]] > table

rocprofv3's own *_kernel_stats.csv sums the whole process: the first step of a detector carries MIOpen's solver search (its naive reference
convolutions, 30-100 ms each, dozens of them) and one-off workspace set-up, which swamp the per-step picture.  Here the timeline is cut into
steps at the optimiser kernel (one multi-tensor SGD launch group per step; marker regex, default 'FusedSgd|fused_sgd') and only the steps
after the warm-up are kept.  Without a marker in the trace (inference modes) everything is kept.  Output: CSV
Name,CallsPerStep,AverageNs,NsPerStep,Percent sorted by time, first line '# steady steps N, GPU busy ms/step X'."""
import csv
import re
import sys
from collections import defaultdict


def main():
    path, total_steps, drop = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    marker = re.compile(sys.argv[4] if len(sys.argv) > 4 else "FusedSgd|fused_sgd")
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker.search(r[2])]
    kept, steps = rows, total_steps
    if marks and len(marks) >= total_steps:
        per = len(marks) // total_steps                          # marker launches per step (parameter groups)
        ends = [marks[(k + 1) * per - 1] for k in range(total_steps)]
        first = ends[drop - 1] + 1 if drop > 0 else 0
        kept, steps = rows[first:ends[-1] + 1], total_steps - drop
    if len(sys.argv) > 5:                                        # launch sequence of the last kept step: name, duration, gap to the previous end
        last = rows[ends[-2] + 1:ends[-1] + 1] if marks and len(marks) >= total_steps and total_steps > 1 else kept
        with open(sys.argv[5], "w") as fh:
            prev = last[0][0]
            for s0, e0, n in last:
                fh.write(f"{(e0 - s0) / 1e3:9.1f} us  gap {(s0 - prev) / 1e3:7.1f} us  {n[:140]}\n")
                prev = e0
    agg = defaultdict(lambda: [0, 0])
    for s, e, n in kept:
        agg[n][0] += 1
        agg[n][1] += e - s
    busy = sum(v[1] for v in agg.values())
    print(f"# steady steps {steps}, GPU busy ms/step {busy / steps / 1e6:.3f}, launches/step {sum(v[0] for v in agg.values()) / steps:.1f}")
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "CallsPerStep", "AverageNs", "NsPerStep", "Percent"])
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([n, round(c / steps, 2), round(t / c), round(t / steps), round(100.0 * t / busy, 2)])


if __name__ == "__main__":
    main()
