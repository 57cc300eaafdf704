cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1b -o bench -- python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 > gpurun_out/prof_r1b_bench.log 2>&1
ls gpurun_out/prof_r1b | head
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_r1b/**/*kernel_stats.csv',recursive=True)
print(f)
rows=list(csv.DictReader(open(f[0])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print("%-80s %6s %9.1f us %8.2f ms %5.1f%%"%(r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6/10, 100*float(r['TotalDurationNs'])/tot))
print("total per step ms", tot/1e6/10)
PY
tail -1 gpurun_out/prof_r1b_bench.log | cut -c1-200
