# configs[3] per-GPU workload: wav2vec2_aasist + SupCon at batch 64 (and batch 32): bench lines + rocprofv3 kernel stats
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for b in 64; do
python3 bench.py --no-cpu-baseline --model wav2vec2_aasist --batch $b --steps 8 --warmup 3 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/r4_bench_wav2vec2_aasist_b$b.json
cut -c1-260 gpurun_out/r4_bench_wav2vec2_aasist_b$b.json
done
rm -rf gpurun_out/prof_aasist
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_aasist -o bench -- python3 bench.py --no-cpu-baseline --model wav2vec2_aasist --batch 64 --steps 4 --warmup 2 > gpurun_out/prof_aasist.log 2>&1
find gpurun_out/prof_aasist -name "*kernel_stats.csv" -exec cp {} gpurun_out/r4_bench_wav2vec2_aasist_b64_kernel_stats.csv \;
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/r4_bench_wav2vec2_aasist_b64_kernel_stats.csv')))
n = 6
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("aasist batch 64: kernel time per step %.2f ms, launches per step %.0f, at::native launches per step %.0f" % (
    tot / 1e6 / n, sum(int(r['Calls']) for r in rows) / n, sum(int(r['Calls']) for r in rows if 'at::native' in r['Name']) / n))
for r in rows[:45]:
    print("%-90s %7.1f/step %9.1f us %7.3f ms/step" % (r['Name'][:90], int(r['Calls']) / n, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6 / n))
PY
rm -rf gpurun_out/prof_aasist
