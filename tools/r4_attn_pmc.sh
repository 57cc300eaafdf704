# L2 -> fabric traffic of the fused attention kernels (separate --pmc passes, kernel-trace only), B = 64
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_attn_f gpurun_out/pmc_attn_w
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_attn_f -o pmc -- python3 tools/attn_probe.py 64 > gpurun_out/pmc_attn_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_attn_w -o pmc -- python3 tools/attn_probe.py 64 > gpurun_out/pmc_attn_w.log 2>&1
python3 - <<'PY' > gpurun_out/r4_attn_pmc.txt
import csv, glob, collections
print("B = 64, T = 199, 16 heads x 64: algorithmic bytes  attn_fwd: qkv 78.2 MB read + ctx 26.1 MB + lse 0.8 MB written;  attn_bwd8: qkv 78.2 + ctx 26.1 + dctx 26.1 + lse 0.8 MB read, dqkv 78.2 MB (+ bias partials) written")
for name, d in (("FETCH_SIZE", "gpurun_out/pmc_attn_f"), ("WRITE_SIZE", "gpurun_out/pmc_attn_w")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name: continue
        k = r["Kernel_Name"]
        fam = "attn_fwd" if "attn_fwd" in k else ("attn_bwd8" if "attn_bwd" in k else None)
        if fam:
            agg[fam][0] += float(r["Counter_Value"]); agg[fam][1] += 1
    for k, v in agg.items():
        kib = v[0] / v[1]
        mb = kib * 1024 / 1e6 * (2 if name == "FETCH_SIZE" else 1)
        print("%-10s %-10s %10.0f KiB per launch as reported (%d launches)  -> %.1f MB%s" % (name, k, kib, v[1], mb, " (x2: gfx950 tallies 128-B requests at 64 B)" if name == "FETCH_SIZE" else ""))
PY
cat gpurun_out/r4_attn_pmc.txt
rm -rf gpurun_out/pmc_attn_f gpurun_out/pmc_attn_w
