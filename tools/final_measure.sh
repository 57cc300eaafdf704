# Round-2 measurement: default bench line (configs[2]), rocprofv3 kernel stats of the same command, PMC HBM traffic passes (separate
# --pmc runs, kernel-trace only), configs[1] line, AASIST / ResNet workloads.  Everything lands in gpurun_out/r2_*.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 bench.py > gpurun_out/bench_default.log 2>&1
grep '^{"metric"' gpurun_out/bench_default.log | tail -1 > gpurun_out/r2_bench_default.json
python3 bench.py --batch 32 --rawboost 0 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/r2_bench_b32_norawboost.json
rm -rf gpurun_out/prof_final
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o bench -- python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 > gpurun_out/prof_final_bench.log 2>&1
grep '^{"metric"' gpurun_out/prof_final_bench.log | tail -1 > gpurun_out/r2_bench_under_rocprof.json
find gpurun_out/prof_final -name "*kernel_stats.csv" -exec cp {} gpurun_out/r2_bench_default_kernel_stats.csv \;
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o pmc -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o pmc -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/pmc_write.log 2>&1
python3 - <<'PY'
import csv, glob, json, collections
out = {"batch": 64}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    d = "gpurun_out/pmc_fetch" if name == "FETCH_SIZE" else "gpurun_out/pmc_write"
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f:
        print("no counter file for", name, glob.glob(d + "/**/*", recursive=True)[:5]); continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != name: continue
        k = r["Kernel_Name"]
        fam = "gemm" if "scl_gemm" in k else ("adamw" if "adamw" in k else ("ln_bwd" if "ln_bwd" in k else ("attn_bwd" if "attn_bwd" in k else ("fir" if "fir_kernel" in k else None))))
        if fam is None: continue
        agg[fam][0] += float(r["Counter_Value"]); agg[fam][1] += 1
    out[name] = {k: [v[0] / max(v[1], 1), v[1]] for k, v in agg.items()}   # mean KiB per launch, launches
out["note"] = "mean per launch, in KiB as reported by rocprofv3; FETCH_SIZE is doubled by the consumer (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md)"
json.dump(out, open("gpurun_out/r2_pmc_hbm_traffic.json", "w"), indent=1)
print(json.dumps(out)[:700])
PY
for m in wav2vec2_aasist wav2vec2_resnet_nll; do
python3 bench.py --no-cpu-baseline --model $m --batch 32 --rawboost 0 --steps 6 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/r2_bench_$m.json
rm -rf gpurun_out/prof_$m
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$m -o bench -- python3 bench.py --no-cpu-baseline --model $m --batch 32 --rawboost 0 --steps 6 --warmup 2 > gpurun_out/prof_$m.log 2>&1
find gpurun_out/prof_$m -name "*kernel_stats.csv" -exec cp {} gpurun_out/r2_bench_${m}_kernel_stats.csv \;
grep '^{"metric"' gpurun_out/prof_$m.log | tail -1 > gpurun_out/r2_bench_${m}_under_rocprof.json
rm -rf gpurun_out/prof_$m
done
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r2_bench_default_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows); n=10
print("default bench under rocprof: kernel time per step %.2f ms"%(tot/1e6/n))
for r in rows[:24]:
    print("%-88s %6s %9.1f us %7.3f ms/step %5.1f%%"%(r['Name'][:88], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6/n, 100*float(r['TotalDurationNs'])/tot))
PY
rm -rf gpurun_out/prof_final gpurun_out/pmc_fetch gpurun_out/pmc_write
for f in gpurun_out/r2_bench_default.json gpurun_out/r2_bench_b32_norawboost.json gpurun_out/r2_bench_wav2vec2_aasist.json gpurun_out/r2_bench_wav2vec2_resnet_nll.json; do echo $f; cut -c1-330 $f; echo; done
