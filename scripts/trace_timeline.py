#!/usr/bin/env python
"""Timeline of ONE replayed step out of a rocprofv3 --kernel-trace CSV (graph replay): per kernel start / duration / queue, gaps between
consecutive kernels on the busiest queue (the main lane), busy and idle totals.  usage: trace_timeline.py <kernel_trace.csv> [first-kernel-substring]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "pad_reflect"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows), key=lambda e: e[0])
# a step starts at the first of the two pad_reflect launches: find starts whose predecessor pad is > 0.5 ms earlier
starts = [i for i, e in enumerate(ev) if key in e[2]]
firsts = [i for k, i in enumerate(starts) if k == 0 or ev[i][0] - ev[starts[k - 1]][0] > 500000]
if len(firsts) < 3:
    sys.exit("not enough steps in trace")
a, b = firsts[-3], firsts[-2]            # the last-but-one complete step
step = ev[a:b]
t0 = step[0][0]
qs = collections.Counter(e[3] for e in step)
mainq = qs.most_common(1)[0][0]
print("step: %d kernels, %.1f us first start -> last end, queues %s" % (len(step), (max(e[1] for e in step) - t0) / 1e3, dict(qs)))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:70]
prev_end = None; busy = 0; gaps = []
for s, e, n, q in step:
    g = ""
    if q == mainq:
        if prev_end is not None:
            gaps.append((s - prev_end) / 1e3); g = "gap %6.2f" % gaps[-1]
        prev_end = e; busy += e - s
    print("%9.2f %7.2f  q%-3s %-10s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, g, short(n)))
print("main queue %s: busy %.1f us, gaps total %.1f us over %d edges (median %.2f, mean %.2f, max %.2f)" % (
    mainq, busy / 1e3, sum(gaps), len(gaps), sorted(gaps)[len(gaps) // 2], sum(gaps) / max(1, len(gaps)), max(gaps)))
big = sorted(gaps, reverse=True)[:12]
print("largest gaps:", ["%.1f" % x for x in big])
