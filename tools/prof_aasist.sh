cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_aasist -o bench -- python3 bench.py --model wav2vec2_aasist --steps 4 --warmup 2 > gpurun_out/prof_aasist.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_aasist/**/*kernel_stats.csv',recursive=True)
rows=list(csv.DictReader(open(f[0])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
n=0
for r in rows:
    if 'scl_gemm' in r['Name']: continue
    print("%-110s %6s %9.1f us %8.2f ms"%(r['Name'][:110], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6/6))
    n+=1
    if n>28: break
print("total per step ms", tot/1e6/6)
PY
