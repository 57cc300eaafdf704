# per kernel: calls per step, average duration, workgroups per launch and threads per workgroup (from rocprofv3's kernel trace of bench.py):
#   bash tools/grid_sizes.sh <tag> <bench.py arguments...>   ->  gpurun_out/r5_<tag>_grids.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
rm -rf gpurun_out/prof_gs
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_gs -o bench -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 "$@" > /dev/null 2>&1
python3 - "$tag" <<'PY' > gpurun_out/r5_${tag}_grids.txt
import csv, glob, collections, sys
f = glob.glob('gpurun_out/prof_gs/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adamw_kernel' in r['Kernel_Name']]
steps = rows[idx[-3] + 1: idx[-1] + 1]
by = collections.defaultdict(lambda: [0, 0, collections.Counter()])
for r in steps:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70]
    wg = int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])
    nwg = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // max(wg, 1)
    by[k][0] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); by[k][1] += 1; by[k][2][(nwg, wg)] += 1
for k, v in sorted(by.items(), key=lambda kv: -kv[1][0])[:45]:
    print("%-72s %5.1f calls %8.1f us  %7.3f ms/step  grids %s" % (k, v[1] / 2, v[0] / 1e3 / v[1], v[0] / 2e6, dict(v[2].most_common(3))))
PY
rm -rf gpurun_out/prof_gs
