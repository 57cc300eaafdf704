cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r6_call2.txt; : > $o
echo "== new / changed tests" >> $o
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -k "grouped" 2>&1 | tail -3 >> $o
SCL_WGRAD_XCD_MAJOR=0 timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -k "grouped" 2>&1 | tail -3 >> $o
timeout 900 python -m pytest tests/test_hipnn_gpu.py tests/test_btse_gpu.py -x -q 2>&1 | tail -5 >> $o
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -s -k "trajectory or carried_over" 2>&1 | grep -v amdgpu.ids | tail -40 >> $o
echo "== grouped launch alone (tools/group_fill_probe.py), per-member XCD runs (0) vs XCD-major over the launch (1)" >> $o
for v in 0 1 0 1; do echo "SCL_WGRAD_XCD_MAJOR=$v" >> $o; SCL_WGRAD_XCD_MAJOR=$v python tools/group_fill_probe.py 2>&1 | grep -v amdgpu >> $o; done
echo "== FETCH_SIZE x 2 per grouped launch" >> $o
for v in 0 1; do
  rm -rf gpurun_out/pmcg
  ( export SCL_WGRAD_XCD_MAJOR=$v; timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmcg -o p -- python3 tools/group_fill_probe.py > gpurun_out/pmcg.log 2>&1 )
  python3 - $v >> $o <<'PY'
import csv, glob, sys, collections
f = glob.glob("gpurun_out/pmcg/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] == "FETCH_SIZE" and "group_kernel" in r["Kernel_Name"]:
        agg[r["Grid_Size"] if "Grid_Size" in r else "?"].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("SCL_WGRAD_XCD_MAJOR=%s grid %s: FETCH %7.1f MB per launch (n %d)" % (sys.argv[1], k, 2 * 1024 * sum(v) / len(v) / 1e6, len(v)))
PY
done
rm -rf gpurun_out/pmcg
echo "== whole step, interleaved" >> $o
bash tools/ab_env.sh "SCL_WGRAD_XCD_MAJOR=0" "SCL_WGRAD_XCD_MAJOR=1" 3 >> $o 2>&1
cat $o
