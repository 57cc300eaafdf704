#!/bin/bash
# conformer slice: timing at the BTSE-sized shape + per-kernel breakdown
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/conformer_probe.py > gpurun_out/r4_conformer_probe.txt 2>&1
python tools/conformer_probe.py 32 199 256 4 64 >> gpurun_out/r4_conformer_probe.txt 2>&1
SCL_PROBE_CPU=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_conf -o c -- python3 tools/conformer_probe.py > gpurun_out/prof_conf.log 2>&1
python - <<'PY' >> gpurun_out/r4_conformer_probe.txt
import csv, glob
f = glob.glob('gpurun_out/prof_conf/**/c_kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("--- kernels (23 train steps + 23 eval forwards), share of GPU time")
for r in rows[:22]:
    print("%-90s %6s calls %9.1f us avg %5.1f %%" % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
cat gpurun_out/r4_conformer_probe.txt
