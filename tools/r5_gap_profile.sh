# steady-state kernel trace of the default bench line + gap attribution -> gpurun_out/r5_bench_default_gaps.txt (+ the trace itself)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_ss
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ss -o bench -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 "$@" > gpurun_out/prof_ss.log 2>&1
f=$(find gpurun_out/prof_ss -name "*kernel_trace.csv" | head -1)
python3 tools/trace_gaps.py $f 3 > gpurun_out/r5_bench_default_gaps.txt
cp $f gpurun_out/r5_default_kernel_trace.csv
rm -rf gpurun_out/prof_ss
cat gpurun_out/r5_bench_default_gaps.txt
