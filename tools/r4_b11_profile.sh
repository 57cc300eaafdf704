# pack-sized step (one 11-view pack, M = 2189 rows): kernel summary + GEMM launches by (kernel, grid)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_b11
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b11 -o bench -- python3 bench.py --no-cpu-baseline --batch 11 --rawboost 0 --steps 8 --warmup 2 > gpurun_out/prof_b11.log 2>&1
python3 - <<'PY' > gpurun_out/r4_b11_kernel_summary.txt
import csv, glob, collections
f = glob.glob('gpurun_out/prof_b11/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', ''), r.get('Grid_Size_Z', '')) for r in csv.DictReader(open(f))))
idx = [i for i, r in enumerate(rows) if 'adamw_kernel' in r[2]]
steps = rows[idx[-4] + 1: idx[-1] + 1]
n = 3
by = collections.defaultdict(lambda: [0, 0])
byg = collections.defaultdict(lambda: [0, 0])
for s, e, name, gx, gz in steps:
    k = name.replace('(anonymous namespace)::', '').replace('void ', '')[:70]
    by[k][0] += e - s; by[k][1] += 1
    if 'gemm' in name:
        byg[(k[:46], gx, gz)][0] += e - s; byg[(k[:46], gx, gz)][1] += 1
tot = sum(v[0] for v in by.values())
print("pack-sized step, last %d steps: kernel time %.2f ms/step, %d launches/step, wall %.2f ms/step" % (n, tot / 1e6 / n, len(steps) / n, (steps[-1][1] - rows[idx[-4]][1]) / 1e6 / n))
for k, v in sorted(by.items(), key=lambda kv: -kv[1][0])[:24]:
    print("%-72s %6.1f calls %8.1f us %7.3f ms/step %5.1f%%" % (k, v[1] / n, v[0] / 1e3 / v[1], v[0] / 1e6 / n, 100 * v[0] / tot))
print("--- GEMM launches by (kernel, grid x, grid z)")
for k, v in sorted(byg.items(), key=lambda kv: -kv[1][0])[:24]:
    print("%-48s gx=%-8s gz=%-3s %6.1f calls %8.1f us %7.3f ms/step" % (k[0], k[1], k[2], v[1] / n, v[0] / 1e3 / v[1], v[0] / 1e6 / n))
PY
cat gpurun_out/r4_b11_kernel_summary.txt
rm -rf gpurun_out/prof_b11
