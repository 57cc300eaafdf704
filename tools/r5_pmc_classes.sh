# HBM-side traffic of the wide-tile GEMM, class by class (verdict r4 #2b): FETCH_SIZE / WRITE_SIZE of one gemm_bench case per pass (separate
# --pmc runs, kernel trace only), mean per wide-kernel launch next to the algorithmic operand bytes -> gpurun_out/r5_pmc_gemm_classes.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_pmc_gemm_classes.txt; : > $out
[ -x tools/gemm_bench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/gemm_bench.cpp -Iinclude -L scl-deepfake-audio-detection_amd -lscl_hip -Wl,-rpath,'$ORIGIN/../scl-deepfake-audio-detection_amd' -o tools/gemm_bench
for case_ in "fc1 fwd" "fc2 dgrad" "qkv fwd" "out fwd" "fc2 fwd" "fc1 dgrad" "fc1 wgrad sk4"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmcc
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmcc -o p -- tools/gemm_bench 64 2 "$case_" > gpurun_out/pmcc.log 2>&1
    python3 - "$case_" "$ctr" >> $out <<'PY'
import csv, glob, sys, collections
case, ctr = sys.argv[1], sys.argv[2]
f = glob.glob("gpurun_out/pmcc/**/*counter_collection.csv", recursive=True)
if not f:
    print(case, ctr, "no output"); sys.exit(0)
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if r["Counter_Name"] != ctr or "scl_gemm_w8" not in k: continue
    a = agg[k[k.find("scl_gemm"):][:40]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, v in agg.items():
    kib = v[0] / v[1]
    print("%-16s %-10s %-42s %9.1f MB per launch (n %d)%s" % (case, ctr, k, kib * 1024 * (2 if ctr == "FETCH_SIZE" else 1) / 1e6, v[1], "   [FETCH_SIZE doubled: gfx950 tallies 128-B requests at 64 B]" if ctr == "FETCH_SIZE" else ""))
PY
  done
done
rm -rf gpurun_out/pmcc
cat $out
