# Which launches surround the __amd_rocclr_copyBuffer / fill kernels of a bench step?  (kernel trace of 2 steps; prints, for every copy,
# the kernel before and after it, aggregated)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace_nb
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_nb -o t -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > gpurun_out/trace_nb.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/trace_nb/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
short = lambda n: n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:70]
# last step only: find the last adamw launch and go back to the previous one
idx = [i for i, r in enumerate(rows) if 'adamw' in r['Kernel_Name']]
lo, hi = (idx[-2] + 1, idx[-1] + 1) if len(idx) >= 2 else (0, len(rows))
step = rows[lo:hi]
agg = collections.Counter()
for i, r in enumerate(step):
    if 'copyBuffer' in r['Kernel_Name'] or 'fillBuffer' in r['Kernel_Name']:
        prev = short(step[i - 1]['Kernel_Name']) if i else '-'
        nxt = short(step[i + 1]['Kernel_Name']) if i + 1 < len(step) else '-'
        agg[(prev, short(r['Kernel_Name']), nxt, r.get('Grid_Size_X', r.get('Grid_Size', '')))] += 1
print('launches in the last step:', len(step), ' span ms:', (int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e6)
for k, c in agg.most_common(40):
    print(c, k)
gaps = []
for a, b in zip(step[:-1], step[1:]):
    g = int(b['Start_Timestamp']) - int(a['End_Timestamp'])
    if g > 3000: gaps.append((g, short(a['Kernel_Name']), short(b['Kernel_Name'])))
print('idle gaps > 3 us in the step: n=%d total=%.3f ms' % (len(gaps), sum(g for g, _, _ in gaps) / 1e6))
for g, a, b in sorted(gaps, reverse=True)[:25]:
    print('%8.1f us  %s -> %s' % (g / 1e3, a, b))
PY
rm -rf gpurun_out/trace_nb
