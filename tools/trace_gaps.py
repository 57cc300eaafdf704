"""Where the GPU idles inside a steady-state train step: reads a rocprofv3 kernel trace (csv), takes the last N optimizer steps (delimited
by adamw_kernel), and reports kernel time, the UNION of busy intervals (launches on side streams overlap), idle = wall - union, the idle
time summed by the kernel that PRECEDES each gap, by the kernel that FOLLOWS it, and the largest single gaps with their neighbours.
    python tools/trace_gaps.py <kernel_trace.csv> [n_steps=3]"""
import collections
import csv
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]


def main():
    f = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
    idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
    steps = rows[idx[-1 - n] + 1: idx[-1] + 1]
    t_begin = rows[idx[-1 - n]][1]
    wall = steps[-1][1] - t_begin
    ktime = sum(e - s for s, e, _ in steps)
    # union of busy intervals + gaps
    gaps = []
    cur_end, prev_name = t_begin, "adamw_kernel (previous step)"
    busy = 0
    for s, e, name in steps:
        if s > cur_end:
            gaps.append((s - cur_end, prev_name, name, s))
            busy += e - s
            cur_end, prev_name = e, name
        else:
            if e > cur_end:
                busy += e - cur_end
                cur_end, prev_name = e, name
    idle = wall - busy
    print("last %d steps: wall %.3f ms/step, kernel time %.3f ms/step, busy (union) %.3f ms/step, idle %.3f ms/step in %.0f gaps/step (mean %.2f us)" % (
        n, wall / 1e6 / n, ktime / 1e6 / n, busy / 1e6 / n, idle / 1e6 / n, len(gaps) / n, idle / 1e3 / max(len(gaps), 1)))
    for title, key in (("idle by PRECEDING kernel", 1), ("idle by FOLLOWING kernel", 2)):
        by = collections.defaultdict(lambda: [0, 0])
        for g in gaps:
            by[short(g[key])][0] += g[0]
            by[short(g[key])][1] += 1
        print("--- " + title)
        for k, v in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]:
            print("%-72s %7.1f gaps/step %7.2f us each %7.3f ms/step" % (k, v[1] / n, v[0] / 1e3 / v[1], v[0] / 1e6 / n))
    print("--- largest single gaps (us): preceding -> following, time since the step's start")
    step_starts = [rows[i][1] for i in idx[-1 - n:-1]]
    for g in sorted(gaps, key=lambda x: -x[0])[:16]:
        st = max(t for t in step_starts if t <= g[3])
        print("%8.1f us  %-50s -> %-50s at +%.2f ms" % (g[0] / 1e3, short(g[1])[:50], short(g[2])[:50], (g[3] - st) / 1e6))
    hist = collections.Counter(min(int(g[0] / 1e3), 50) // 2 * 2 for g in gaps)
    print("--- gap histogram (us bucket: gaps/step): " + ", ".join("%d-%d: %.0f" % (k, k + 2, v / n) for k, v in sorted(hist.items())))


if __name__ == "__main__":
    main()
