# Round 6, verdict item 1(b): A/B of the XCD ownership axis of the wide-tile GEMM, class by class, time AND fabric traffic.
#   orders: shipped (one grouped walk, 8 rows per group; XCD x owns the x-th eighth of it) | group size 4 / 16 / all rows (= column
#   bands: every XCD streams all of A) | region grids R x 8/R: 8x1 (M bands), 4x2, 2x4, 1x8 (N bands)
# needs the experiment build:  SCL_BUILD_TAG=exp SCL_BUILD_DEFINES=-DSCL_EXPERIMENTS python scl-deepfake-audio-detection_amd/build.py
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_xcd_order_ab.txt; : > $out
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/scl-deepfake-audio-detection_amd/build_exp:$LD_LIBRARY_PATH
export GEMM_BENCH_W8_ONLY=1
orders=("SCL_GEMM_GROUP_M=8 SCL_GEMM_XCD_ROWS=0" "SCL_GEMM_GROUP_M=4 SCL_GEMM_XCD_ROWS=0" "SCL_GEMM_GROUP_M=16 SCL_GEMM_XCD_ROWS=0" "SCL_GEMM_GROUP_M=64 SCL_GEMM_XCD_ROWS=0" \
        "SCL_GEMM_GROUP_M=8 SCL_GEMM_XCD_ROWS=8" "SCL_GEMM_GROUP_M=4 SCL_GEMM_XCD_ROWS=8" "SCL_GEMM_GROUP_M=8 SCL_GEMM_XCD_ROWS=4" "SCL_GEMM_GROUP_M=4 SCL_GEMM_XCD_ROWS=4" \
        "SCL_GEMM_GROUP_M=8 SCL_GEMM_XCD_ROWS=2" "SCL_GEMM_GROUP_M=8 SCL_GEMM_XCD_ROWS=1" "SCL_GEMM_GROUP_M=2 SCL_GEMM_XCD_ROWS=1")
echo "== time: tools/gemm_bench 64 20 <case>, median of 5 interleaved rounds, operands rotated over 3 buffer sets; two passes over the orders" >> $out
for pass in 1 2; do
for o in "${orders[@]}"; do
  for case_ in "fc1 fwd" "fc2 dgrad" "qkv fwd" "out fwd" "fc2 fwd" "fc1 dgrad" "qkv dgrad"; do
    printf "%-48s " "$o" >> $out
    env $o timeout 120 tools/gemm_bench 64 20 "$case_" 2>&1 | grep -v "^case" | cut -c1-80 >> $out
  done
done
done
echo "== fabric traffic: FETCH_SIZE x 2 (MB per launch, gfx950 note) of one gemm_bench case per rocprofv3 --pmc pass" >> $out
for o in "${orders[@]}"; do
  for case_ in "fc1 fwd" "qkv fwd" "fc2 fwd" "fc1 dgrad"; do
    rm -rf gpurun_out/pmcx
    # the environment is exported (not `env ...` behind rocprofv3: the program itself must follow `--`)
    ( export $o; timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmcx -o p -- tools/gemm_bench 64 2 "$case_" > gpurun_out/pmcx.log 2>&1 )
    python3 - "$o" "$case_" >> $out <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/pmcx/**/*counter_collection.csv", recursive=True)
if not f:
    print("%-48s %-10s no counter output" % (sys.argv[1], sys.argv[2])); sys.exit(0)
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == "FETCH_SIZE" and "scl_gemm_w8" in r["Kernel_Name"]]
print("%-48s %-10s FETCH %7.1f MB per launch (n %d)" % (sys.argv[1], sys.argv[2], 2 * 1024 * sum(v) / max(len(v), 1) / 1e6, len(v)))
PY
  done
done
rm -rf gpurun_out/pmcx
tail -50 $out
