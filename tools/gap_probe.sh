# Idle time between consecutive kernels of the train step (rocprofv3 kernel trace of bench.py, last full step): sum and distribution of
# start(i+1) - end(i), by the kernel that precedes the gap.      gpurun -- 'bash tools/gap_probe.sh'
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gap_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap_trace -o t -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/gap_bench.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/gap_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# last step = from the last adamw back to the previous adamw
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
a, b = idx[-2] + 1, idx[-1] + 1
step = rows[a:b]
wall = step[-1][1] - rows[a - 1][1]
busy = 0; gaps = []; prev_end = rows[a - 1][1]; overlap = 0
by = collections.defaultdict(lambda: [0, 0])
for s, e, n in step:
    g = s - prev_end
    gaps.append(g)
    key = n[n.find("scl_"):][:40] if "scl_" in n else n[:40]
    by[key][0] += max(g, 0); by[key][1] += 1
    busy += e - max(s, prev_end) if e > prev_end else 0
    prev_end = max(prev_end, e)
pos = [g for g in gaps if g > 0]
print("last step: %d kernels, wall %.2f ms, kernel-busy %.2f ms, idle between kernels %.2f ms (%d gaps > 0, median %.2f us, p90 %.2f us)" % (
    len(step), wall / 1e6, busy / 1e6, sum(pos) / 1e6, len(pos), sorted(pos)[len(pos) // 2] / 1e3, sorted(pos)[int(0.9 * len(pos))] / 1e3))
print("idle in FRONT of a kernel, by kernel (ms per step, launches, us per launch):")
for k, v in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  %-42s %.3f ms  %4d  %.2f us" % (k, v[0] / 1e6, v[1], v[0] / 1e3 / v[1]))
PY
rm -rf gpurun_out/gap_trace
