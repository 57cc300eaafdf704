# Kernel summary of the STEADY-STATE steps of a bench.py run (the last 3 optimizer steps of the rocprofv3 kernel trace: warm-up, model
# construction and plan building are left out — rocprofv3's own --stats table sums the whole process).
#   bash tools/steady_state_profile.sh <tag> <bench.py arguments...>     ->  gpurun_out/${ROUND:-r6}_<tag>_steady_state.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
rm -rf gpurun_out/prof_ss
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ss -o bench -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 "$@" > gpurun_out/prof_ss_$tag.log 2>&1
python3 - "$tag" "$*" <<'PY' > gpurun_out/${ROUND:-r6}_${tag}_steady_state.txt
import csv, glob, collections, sys
tag, args = sys.argv[1], sys.argv[2]
f = glob.glob('gpurun_out/prof_ss/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)))
idx = [i for i, r in enumerate(rows) if 'adamw_kernel' in r[2]]
n = 3
steps = rows[idx[-1 - n] + 1: idx[-1] + 1]
by = collections.defaultdict(lambda: [0, 0])
for s, e, name in steps:
    k = name.replace('(anonymous namespace)::', '').replace('void ', '')[:96]
    by[k][0] += e - s; by[k][1] += 1
tot = sum(v[0] for v in by.values())
nat = [(k, v) for k, v in by.items() if k.startswith('at::') or 'rocclr' in k]
print("python bench.py --no-cpu-baseline --steps 6 --warmup 2 %s   (last %d optimizer steps of the kernel trace)" % (args, n))
print("kernel time %.2f ms/step, %.0f launches/step, wall %.2f ms/step (under the profiler); at::native + rocclr: %.1f launches/step, %.3f ms/step = %.2f %% of the kernel time, largest row %.3f %%" % (
    tot / 1e6 / n, len(steps) / n, (steps[-1][1] - rows[idx[-1 - n]][1]) / 1e6 / n, sum(v[1] for _, v in nat) / n, sum(v[0] for _, v in nat) / 1e6 / n,
    100 * sum(v[0] for _, v in nat) / tot, max([100 * v[0] / tot for _, v in nat] or [0])))
for k, v in sorted(by.items(), key=lambda kv: -kv[1][0])[:40]:
    print("%-98s %6.1f calls %8.1f us %7.3f ms/step %5.2f%%" % (k, v[1] / n, v[0] / 1e3 / v[1], v[0] / 1e6 / n, 100 * v[0] / tot))
print("--- every at::native / rocclr row")
for k, v in sorted(nat, key=lambda kv: -kv[1][0]):
    print("%-98s %6.1f calls %8.1f us %7.3f ms/step %5.3f%%" % (k, v[1] / n, v[0] / 1e3 / v[1], v[0] / 1e6 / n, 100 * v[0] / tot))
PY
head -3 gpurun_out/${ROUND:-r6}_${tag}_steady_state.txt | cut -c1-300
sed -n '/--- every/,$p' gpurun_out/${ROUND:-r6}_${tag}_steady_state.txt | cut -c1-180
rm -rf gpurun_out/prof_ss
