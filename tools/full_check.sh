mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/pt_all.log 2>&1; echo pytest rc=$?; tail -4 gpurun_out/pt_all.log
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo bench rc=$?; cut -c1-1500 gpurun_out/bench_default.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
