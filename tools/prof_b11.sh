cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b11 -o bench -- python3 bench.py --no-cpu-baseline --batch 11 --steps 8 --warmup 2 > gpurun_out/prof_b11.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_b11/**/*kernel_stats.csv',recursive=True)
rows=list(csv.DictReader(open(f[0])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:18]:
    print("%-80s %6s %9.1f us %8.2f ms %5.1f%%"%(r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6/10, 100*float(r['TotalDurationNs'])/tot))
print("total per step ms", tot/1e6/10)
PY
rm -rf gpurun_out/prof_b11
