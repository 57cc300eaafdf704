# One-GPU breakdown of the data-parallel choreography (SCL_BENCH_FORCE_DP=1: the bucketed exchange through RCCL with one rank):
# plain step vs forced exchange (all-reduce / shard mode), and the kernel trace of the forced run (which RCCL kernels run, for how long).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
J='{"ms":d["ms_per_step"],"rccl":d.get("rccl")}'
for i in 1 2; do
python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain', json.dumps($J))"
SCL_BENCH_FORCE_DP=1 python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('force_dp allreduce', json.dumps($J)[:900])"
SCL_BENCH_FORCE_DP=1 SCL_DP_MODE=shard python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('force_dp shard', json.dumps($J)[:900])"
done
rm -rf gpurun_out/prof_dp
SCL_BENCH_FORCE_DP=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dp -o bench -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/prof_dp.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_dp/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
n = 6
print("kernels of the forced exchange that are not ours (per step):")
for r in rows:
    nm = r['Name']
    if 'scl_' in nm or 'anonymous namespace' in nm:
        continue
    print("  %-100s %7.1f/step %9.1f us avg %8.3f ms/step" % (nm[:100], int(r['Calls']) / n, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6 / n))
PY
rm -rf gpurun_out/prof_dp
