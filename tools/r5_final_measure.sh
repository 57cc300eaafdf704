# Round-5 measurement set: default bench line (configs[2]) with cpu_baseline, rocprofv3 kernel stats + per-grid GEMM table of the same
# command, PMC HBM traffic passes (separate --pmc runs, kernel-trace only) stamped with the GEMM-source fingerprint bench.py checks,
# configs[1] line, AASIST (configs[3] per-GPU batch 64, and batch 32) / ResNet workloads.  Everything lands in gpurun_out/r5_*.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 bench.py > gpurun_out/bench_default.log 2>&1
grep '^{"metric"' gpurun_out/bench_default.log | tail -1 > gpurun_out/r5_bench_default.json
python3 bench.py --batch 32 --rawboost 0 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/r5_bench_b32_norawboost.json
bash tools/prof_bench.sh r5 > gpurun_out/prof_r5.log 2>&1
cp gpurun_out/prof_r5_kernel_stats.csv gpurun_out/r5_bench_default_kernel_stats.csv
grep -A40 "GEMM launches by" gpurun_out/prof_r5_summary.txt > gpurun_out/r5_gemm_launches_by_grid.txt
head -40 gpurun_out/prof_r5_summary.txt > gpurun_out/r5_bench_default_kernel_summary.txt
grep '^{"metric"' gpurun_out/prof_r5_bench.log | tail -1 > gpurun_out/r5_bench_under_rocprof.json
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/prof_r5
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o pmc -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o pmc -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/pmc_write.log 2>&1
python3 - <<'PY'
import csv, glob, json, collections, sys
sys.path.insert(0, ".")
import bench
out = {"batch": 64, "gemm_src_sha": bench.gemm_source_sha()}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    d = "gpurun_out/pmc_fetch" if name == "FETCH_SIZE" else "gpurun_out/pmc_write"
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f:
        print("no counter file for", name, glob.glob(d + "/**/*", recursive=True)[:5]); continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != name: continue
        k = r["Kernel_Name"]
        fam = "gemm" if ("scl_gemm" in k or "posconv_mfma" in k) else ("adamw" if "adamw" in k else ("ln_bwd" if "ln_bwd" in k else ("attn_bwd" if "attn_bwd" in k else ("fir" if "fir_kernel" in k else None))))
        if fam is None: continue
        agg[fam][0] += float(r["Counter_Value"]); agg[fam][1] += 1
    out[name] = {k: [v[0] / max(v[1], 1), v[1]] for k, v in agg.items()}   # mean KiB per launch, launches
out["note"] = "mean per launch, in KiB as reported by rocprofv3; FETCH_SIZE is doubled by the consumer (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md); gemm_src_sha = fingerprint of csrc/gemm* at collection time (bench.py quotes the traffic only while it matches)"
json.dump(out, open("gpurun_out/r5_pmc_hbm_traffic.json", "w"), indent=1)
print(json.dumps(out)[:700])
PY
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
for spec in "wav2vec2_aasist 64" "wav2vec2_aasist 32" "wav2vec2_resnet_nll 32" "wav2vec2_btse 64" "wav2vec2_btse 128"; do
set -- $spec; m=$1; b=$2
python3 bench.py --no-cpu-baseline --model $m --batch $b --rawboost 0 --steps 6 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/r5_bench_${m}_b$b.json
done
rm -rf gpurun_out/prof_aasist
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_aasist -o bench -- python3 bench.py --no-cpu-baseline --model wav2vec2_aasist --batch 64 --rawboost 0 --steps 4 --warmup 2 > gpurun_out/prof_aasist.log 2>&1
find gpurun_out/prof_aasist -name "*kernel_stats.csv" -exec cp {} gpurun_out/r5_bench_wav2vec2_aasist_b64_kernel_stats.csv \;
rm -rf gpurun_out/prof_aasist
python3 bench.py --no-cpu-baseline --batch 11 --rawboost 0 --steps 10 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/r5_bench_pack11.json
python3 bench.py --eval --steps 5 --warmup 2 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/r5_bench_eval_b64.json
python3 tools/btse_bio_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r5_btse_bio_probe_final.txt
python3 tools/attn_probe.py 64 32 2>&1 | grep -v amdgpu > gpurun_out/r5_attn_probe.txt
python3 tools/posconv_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r5_posconv_probe.txt
for f in gpurun_out/r5_bench_wav2vec2_btse_b64.json gpurun_out/r5_bench_wav2vec2_btse_b128.json gpurun_out/r5_bench_pack11.json gpurun_out/r5_bench_eval_b64.json gpurun_out/r5_bench_default.json gpurun_out/r5_bench_b32_norawboost.json gpurun_out/r5_bench_wav2vec2_aasist_b64.json gpurun_out/r5_bench_wav2vec2_aasist_b32.json gpurun_out/r5_bench_wav2vec2_resnet_nll_b32.json; do echo $f; cut -c1-330 $f; echo; done
