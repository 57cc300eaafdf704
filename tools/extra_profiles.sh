# rocprofv3 kernel-stats summaries of the non-default workloads (batch 64 + RawBoost on the GPU; AASIST plugin)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b64rb -o bench -- python3 bench.py --no-cpu-baseline --batch 64 --rawboost 5 --steps 6 > gpurun_out/prof_b64rb.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_aasist -o bench -- python3 bench.py --no-cpu-baseline --model wav2vec2_aasist --steps 6 > gpurun_out/prof_aasist2.log 2>&1
find gpurun_out/prof_b64rb -name "*kernel_stats.csv" -exec cp {} gpurun_out/r1_bench_b64_rawboost5_kernel_stats.csv \;
find gpurun_out/prof_aasist -name "*kernel_stats.csv" -exec cp {} gpurun_out/r1_bench_aasist_kernel_stats.csv \;
grep '^{"metric"' gpurun_out/prof_b64rb.log > gpurun_out/r1_bench_b64_rawboost5.json
grep '^{"metric"' gpurun_out/prof_aasist2.log > gpurun_out/r1_bench_aasist.json
rm -rf gpurun_out/prof_b64rb gpurun_out/prof_aasist
head -c 300 gpurun_out/r1_bench_b64_rawboost5.json; echo; head -c 300 gpurun_out/r1_bench_aasist.json; echo
head -8 gpurun_out/r1_bench_b64_rawboost5_kernel_stats.csv | cut -c1-150
