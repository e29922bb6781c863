#!/usr/bin/env python3
"""Per-queue timeline of the last steady steps of a rocprofv3 kernel trace of bench.py: for every HIP stream (HSA queue) the busy time, the idle
gaps above a threshold and what ran around them -- where do the main stream's bubbles come from?

  tools/stream_timeline.py <kernel_trace.csv> [steps to show = 2] [gap threshold us = 30]
"""
import csv
import re
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    thr = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if re.search("FusedSgd|fused_sgd", r[2])]
    if len(marks) < nshow + 2:
        print("not enough steps in the trace")
        return
    lo, hi = marks[-nshow - 1] + 1, marks[-1] + 1
    seg = rows[lo:hi]
    t0 = seg[0][0]
    print(f"# {nshow} steps, {len(seg)} launches, wall {(seg[-1][1] - t0) / 1e3:.1f} us = {(seg[-1][1] - t0) / 1e3 / nshow:.1f} us / step")
    byq = defaultdict(list)
    for s, e, n, q in seg:
        byq[q].append((s, e, n))
    for q, ks in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
        busy = sum(e - s for s, e, _ in ks)
        print(f"## queue {q}: {len(ks)} launches, busy {busy / 1e3:.1f} us ({busy / 1e3 / nshow:.1f} / step)")
        prev_e, prev_n = ks[0][0], "(start)"
        for s, e, n in ks:
            gap = (s - prev_e) / 1e3
            if gap > thr:
                print(f"   t={(prev_e - t0) / 1e3:9.1f}  idle {gap:7.1f} us   after {prev_n[:60]:60s} before {n[:60]}")
            if e > prev_e:
                prev_e, prev_n = e, n
    # marker positions of the optimiser kernel
    for i in marks[-nshow - 1:]:
        print(f"# optimiser kernel at t={(rows[i][0] - t0) / 1e3:9.1f} us on queue {rows[i][3]}")


if __name__ == "__main__":
    main()
