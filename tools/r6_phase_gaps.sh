# wall - kernel time of the default step, phase by phase: un-profiled HIP-event spans next to the kernel trace of the same loop
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 tools/phase_gaps.py events 2>/dev/null | grep '^{' | tail -1 > gpurun_out/phase_events.json
cat gpurun_out/phase_events.json
rm -rf gpurun_out/prof_ph
( export PHASE_NO_EVENTS=1; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ph -o ph -- python3 tools/phase_gaps.py events > gpurun_out/prof_ph.log 2>&1 )
f=$(find gpurun_out/prof_ph -name "*kernel_trace.csv" | head -1)
python3 tools/phase_gaps.py trace $f gpurun_out/phase_events.json > gpurun_out/r6_bench_default_phase_gaps.txt
python3 tools/trace_gaps.py $f 3 > gpurun_out/r6_bench_default_gaps.txt
rm -rf gpurun_out/prof_ph
cat gpurun_out/r6_bench_default_phase_gaps.txt
head -30 gpurun_out/r6_bench_default_gaps.txt
