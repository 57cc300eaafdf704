# Round 6, verdict item 1(a): the per-class table of the GEMM family — live launch durations of the bench step joined with FETCH_SIZE /
# WRITE_SIZE of two separate --pmc passes of the same command (by dispatch order) -> gpurun_out/r6_gemm_classes.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 bench.py --no-cpu-baseline --dump-gemm-launches gpurun_out/gl_live.json 2>gpurun_out/gl_live.err | grep '^{"metric"' | tail -1 > gpurun_out/r6_bench_live.json
cut -c1-400 gpurun_out/r6_bench_live.json
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -o pmc -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --dump-gemm-launches gpurun_out/gl_$c.json > gpurun_out/pmc_$c.log 2>&1
done
python3 tools/gemm_classes.py gpurun_out/gl_live.json --fetch gpurun_out/pmc_FETCH_SIZE --log-fetch gpurun_out/gl_FETCH_SIZE.json \
    --write gpurun_out/pmc_WRITE_SIZE --log-write gpurun_out/gl_WRITE_SIZE.json > gpurun_out/r6_gemm_classes.txt 2> gpurun_out/r6_gemm_classes.err
cat gpurun_out/r6_gemm_classes.err | head -5
cat gpurun_out/r6_gemm_classes.txt
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
