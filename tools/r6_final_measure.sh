# Round-6 measurement set: default bench line (configs[2]) with cpu_baseline, rocprofv3 kernel stats + steady-state table of the same
# command, the PMC passes (separate --pmc runs, kernel-trace only) that feed BOTH the family traffic file bench.py quotes (stamped with
# the GEMM-source fingerprint) and the per-class table, configs[1], the other model plugins, pack 11, scoring, probes.  -> gpurun_out/r6_*
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export ROUND=r6
python3 bench.py --dump-gemm-launches gpurun_out/gl_live.json > gpurun_out/bench_default.log 2>&1
grep '^{"metric"' gpurun_out/bench_default.log | tail -1 > gpurun_out/r6_bench_default.json
python3 bench.py --batch 32 --rawboost 0 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/r6_bench_b32_norawboost.json
bash tools/prof_bench.sh r6 > gpurun_out/prof_r6.log 2>&1
cp gpurun_out/prof_r6_kernel_stats.csv gpurun_out/r6_bench_default_kernel_stats.csv
grep -A40 "GEMM launches by" gpurun_out/prof_r6_summary.txt > gpurun_out/r6_gemm_launches_by_grid.txt
head -40 gpurun_out/prof_r6_summary.txt > gpurun_out/r6_bench_default_kernel_summary.txt
grep '^{"metric"' gpurun_out/prof_r6_bench.log | tail -1 > gpurun_out/r6_bench_under_rocprof.json
bash tools/steady_state_profile.sh bench_default > /dev/null 2>&1
rm -rf gpurun_out/prof_r6
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -o pmc -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --dump-gemm-launches gpurun_out/gl_$c.json > gpurun_out/pmc_$c.log 2>&1
done
python3 tools/gemm_classes.py gpurun_out/gl_live.json --fetch gpurun_out/pmc_FETCH_SIZE --log-fetch gpurun_out/gl_FETCH_SIZE.json \
    --write gpurun_out/pmc_WRITE_SIZE --log-write gpurun_out/gl_WRITE_SIZE.json > gpurun_out/r6_gemm_classes.txt 2> gpurun_out/r6_gemm_classes.err
python3 - <<'PY'
import csv, glob, json, collections, sys
sys.path.insert(0, ".")
import bench
out = {"batch": 64, "gemm_src_sha": bench.gemm_source_sha()}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % name, recursive=True)
    if not f:
        print("no counter file for", name); continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != name: continue
        k = r["Kernel_Name"]
        fam = "gemm" if (("scl_gemm" in k and "f32" not in k.split("(")[0]) or "posconv_mfma" in k or "posconv_wgrad" in k) else ("adamw" if "adamw" in k else ("ln_bwd" if "ln_bwd" in k else ("attn_bwd" if "attn_bwd" in k else ("fir" if "fir_kernel" in k else None))))
        if fam is None: continue
        agg[fam][0] += float(r["Counter_Value"]); agg[fam][1] += 1
    out[name] = {k: [v[0] / max(v[1], 1), v[1]] for k, v in agg.items()}   # mean KiB per launch, launches
out["note"] = "mean per launch, in KiB as reported by rocprofv3; FETCH_SIZE is doubled by the consumer (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md); gemm_src_sha = fingerprint of csrc/gemm* + posconv* at collection time (bench.py quotes the traffic only while it matches); family = the launches bench.py's roofline counts (bf16 GEMM kernels + pos-conv kernels)"
json.dump(out, open("gpurun_out/r6_pmc_hbm_traffic.json", "w"), indent=1)
print(json.dumps(out)[:700])
PY
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
for spec in "wav2vec2_aasist 64" "wav2vec2_aasist 32" "wav2vec2_resnet_nll 32" "wav2vec2_btse 64" "wav2vec2_btse 128"; do
set -- $spec; m=$1; b=$2
python3 bench.py --no-cpu-baseline --model $m --batch $b --rawboost 0 --steps 6 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/r6_bench_${m}_b$b.json
done
python3 bench.py --no-cpu-baseline --batch 11 --rawboost 0 --steps 10 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/r6_bench_pack11.json
python3 bench.py --eval --steps 5 --warmup 2 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/r6_bench_eval_b64.json
python3 tools/attn_probe.py 64 32 2>&1 | grep -v amdgpu > gpurun_out/r6_attn_probe.txt
python3 tools/posconv_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r6_posconv_probe.txt
python3 tools/group_fill_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r6_group_fill_probe.txt
timeout 1500 python3 tools/data_path_probe.py 2>&1 | grep -v "amdgpu.ids\|Scores saved\|vocoders" > gpurun_out/r6_pack_builder.txt
for f in gpurun_out/r6_bench_default.json gpurun_out/r6_bench_b32_norawboost.json gpurun_out/r6_bench_pack11.json gpurun_out/r6_bench_eval_b64.json gpurun_out/r6_bench_wav2vec2_aasist_b64.json gpurun_out/r6_bench_wav2vec2_aasist_b32.json gpurun_out/r6_bench_wav2vec2_resnet_nll_b32.json gpurun_out/r6_bench_wav2vec2_btse_b64.json gpurun_out/r6_bench_wav2vec2_btse_b128.json; do echo $f; cut -c1-330 $f; echo; done
head -4 gpurun_out/r6_bench_default_steady_state.txt | cut -c1-250
tail -4 gpurun_out/r6_gemm_classes.txt
timeout 2400 python3 -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r6_full_gpu_tests.log
cat gpurun_out/r6_full_gpu_tests.log
