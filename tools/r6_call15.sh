cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python3 -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r6_full_gpu_tests.log
cat gpurun_out/r6_full_gpu_tests.log
timeout 1500 python3 tools/data_path_probe.py 2>&1 | grep -v "amdgpu.ids\|Scores saved\|vocoders" > gpurun_out/r6_pack_builder.txt
cat gpurun_out/r6_pack_builder.txt
python3 bench.py 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/r6_bench_default.json
cut -c1-300 gpurun_out/r6_bench_default.json
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
