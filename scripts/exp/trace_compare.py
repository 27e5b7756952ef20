"""Two rocprofv3 --kernel-trace CSVs of the same replayed step (e.g. replays queued back to back / host wait per step): per kernel name the mean duration per step in both,
the sum over the step, the step period, and the kernels that differ most.   usage: trace_compare.py a_kernel_trace.csv b_kernel_trace.csv [first-kernel-substring]"""
import csv, sys, collections, re


def load(path, key):
    rows = list(csv.DictReader(open(path)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
    starts = [i for i, e in enumerate(ev) if key in e[2]]
    firsts = [i for k, i in enumerate(starts) if k == 0 or ev[i][0] - ev[starts[k - 1]][0] > 500000]
    steps = [ev[firsts[k]:firsts[k + 1]] for k in range(max(0, len(firsts) - 7), len(firsts) - 1)]       # the last six complete steps
    per = collections.defaultdict(float)
    cnt = collections.defaultdict(int)
    for st in steps:
        for s, e, n in st:
            n = re.sub(r"\(anonymous namespace\)::|void ", "", n)[:64]
            per[n] += (e - s) / 1e3; cnt[n] += 1
    ns = len(steps)
    period = (ev[firsts[-1]][0] - ev[firsts[-1 - ns]][0]) / 1e3 / ns
    return {k: v / ns for k, v in per.items()}, {k: v / ns for k, v in cnt.items()}, period, ns


key = sys.argv[3] if len(sys.argv) > 3 else "pack_weights"
a, ca, pa, na = load(sys.argv[1], key)
b, cb, pb, nb = load(sys.argv[2], key)
print("A: %s   %d steps, period %.1f us, sum of kernel durations %.1f us/step" % (sys.argv[1].split("/")[-1], na, pa, sum(a.values())))
print("B: %s   %d steps, period %.1f us, sum of kernel durations %.1f us/step" % (sys.argv[2].split("/")[-1], nb, pb, sum(b.values())))
d = sorted(((b.get(k, 0) - a.get(k, 0), k) for k in set(a) | set(b)), key=lambda x: -abs(x[0]))
print("%-64s %5s %9s %9s %8s" % ("kernel (us per step)", "n", "A", "B", "B - A"))
for dv, k in d[:24]:
    print("%-64s %5.1f %9.1f %9.1f %+8.1f" % (k, ca.get(k, cb.get(k, 0)), a.get(k, 0), b.get(k, 0), dv))
