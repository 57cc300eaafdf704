# Round-end measurement: default bench line, rocprofv3 kernel stats of the same command, PMC HBM traffic pass.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/bench_default.log 2>&1
tail -1 gpurun_out/bench_default.log > gpurun_out/r1_bench_default.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o bench -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_final_bench.log 2>&1
tail -1 gpurun_out/prof_final_bench.log > gpurun_out/r1_bench_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o pmc -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o pmc -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/pmc_write.log 2>&1
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    d = "gpurun_out/pmc_fetch" if name == "FETCH_SIZE" else "gpurun_out/pmc_write"
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f:
        print("no counter file for", name, glob.glob(d + "/**/*", recursive=True)[:5]); continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != name: continue
        k = r["Kernel_Name"]
        fam = "gemm" if "scl_gemm" in k else ("adamw" if "adamw" in k else ("ln_bwd" if "ln_bwd" in k else ("attn_bwd" if "attn_bwd" in k else None)))
        if fam is None: continue
        agg[fam][0] += float(r["Counter_Value"]); agg[fam][1] += 1
    out[name] = {k: [v[0] / max(v[1], 1), v[1]] for k, v in agg.items()}   # mean KiB per launch, launches
out["note"] = "mean per launch, in KiB as reported by rocprofv3; FETCH_SIZE is doubled by the consumer (gfx950: 64-B units reported as 32-B, MI355X_MICROARCH.md)"
json.dump(out, open("gpurun_out/r1_pmc_hbm_traffic.json", "w"), indent=1)
print(json.dumps(out)[:600])
PY
cp gpurun_out/prof_final/*kernel_stats.csv gpurun_out/r1_bench_default_kernel_stats.csv 2>/dev/null || find gpurun_out/prof_final -name "*kernel_stats.csv" -exec cp {} gpurun_out/r1_bench_default_kernel_stats.csv \;
rm -rf gpurun_out/prof_final/*kernel_trace.csv gpurun_out/pmc_fetch gpurun_out/pmc_write
cut -c1-400 gpurun_out/r1_bench_default.json
