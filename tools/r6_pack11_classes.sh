# the 11-view pack step (what 02_train.sh runs per pack): per-class GEMM table (live durations only) and the steady-state kernel table
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 bench.py --no-cpu-baseline --batch 11 --rawboost 0 --dump-gemm-launches gpurun_out/gl_p11.json 2>/dev/null | grep '^{"metric"' | tail -1 | cut -c1-300
python3 tools/gemm_classes.py gpurun_out/gl_p11.json > gpurun_out/r6_pack11_gemm_classes.txt
cat gpurun_out/r6_pack11_gemm_classes.txt
rm -rf gpurun_out/prof_p11
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_p11 -o p11 -- python3 bench.py --no-cpu-baseline --batch 11 --rawboost 0 --steps 10 --warmup 5 > gpurun_out/prof_p11.log 2>&1
f=$(find gpurun_out/prof_p11 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' > gpurun_out/r6_pack11_kernel_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("# pack-11 step under rocprofv3 --kernel-trace --stats: 15 steps + set-up; share of kernel time")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    print("%-90s calls %6s  avg %9.1f us  total %8.2f ms  %5.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
cat gpurun_out/r6_pack11_kernel_stats.txt
rm -rf gpurun_out/prof_p11
