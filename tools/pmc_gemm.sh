# SQ counter breakdown of one wide-GEMM case (tools/gemm_bench through rocprofv3 --pmc, kernel trace only; one pass per counter set).
#     gpurun -- 'bash tools/pmc_gemm.sh "fc1 dgrad"'
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
case_="${1:-fc1 dgrad}"
mkdir -p gpurun_out; out=gpurun_out/pmc_gemm.txt; : > $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_SCA SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
    i=$((i+1)); rm -rf gpurun_out/pmcg_$i
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcg_$i -o p -- tools/gemm_bench 64 3 "$case_" > gpurun_out/pmcg_$i.log 2>&1
    python3 - "$i" "$set" >> $out <<'PY'
import csv, glob, sys, collections
i, names = sys.argv[1], sys.argv[2].split()
f = glob.glob("gpurun_out/pmcg_%s/**/*counter_collection.csv" % i, recursive=True)
if not f:
    print("pass", i, "no output:", open("gpurun_out/pmcg_%s.log" % i).read()[-400:]); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "w8" not in k: continue
    k = k[k.find("scl_gemm"):][:44]
    a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in agg.items():
    print(k, " ".join("%s=%.4g(n%d)" % (n, v[0] / max(v[1], 1), v[1]) for n, v in d.items()))
PY
    rm -rf gpurun_out/pmcg_$i
done
cat $out
