# rocprofv3 kernel trace of the default bench (configs[2]); per-kernel and per-(kernel, grid) time per step
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r2}
STEPS=4; WARM=1
shift; EXTRA="$@"      # e.g. tools/prof_bench.sh aasist --model wav2vec2_aasist --batch 32 --rawboost 0
rm -rf gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o bench -- python3 bench.py --no-cpu-baseline --steps $STEPS --warmup $WARM $EXTRA > gpurun_out/prof_${TAG}_bench.log 2>&1
python3 - $TAG $STEPS $WARM <<'PY'
import csv,glob,sys,collections
tag,steps,warm=sys.argv[1],int(sys.argv[2]),int(sys.argv[3])
n=steps+warm
f=glob.glob('gpurun_out/prof_%s/**/*kernel_stats.csv'%tag,recursive=True)
rows=list(csv.DictReader(open(f[0])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
out=open('gpurun_out/prof_%s_summary.txt'%tag,'w')
def P(*a):
    s=" ".join(str(x) for x in a); print(s); out.write(s+"\n")
for r in rows[:int(__import__('os').environ.get('PROF_TOP','28'))]:
    P("%-90s %6s %9.1f us %8.3f ms/step %5.1f%%"%(r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6/n, 100*float(r['TotalDurationNs'])/tot))
P("total kernel time per step ms", tot/1e6/n)
t=glob.glob('gpurun_out/prof_%s/**/*kernel_trace.csv'%tag,recursive=True)
agg=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(t[0])):
    nm=r['Kernel_Name']
    if 'gemm' not in nm and 'posconv_mfma' not in nm: continue
    key=(nm.split('(sclg::GemmK')[0].split('(unsigned short const*')[0].replace('void (anonymous namespace)::','')[-60:], r.get('Grid_Size_X', r.get('Grid_Size','')), r.get('Grid_Size_Z',''))
    d=float(r['End_Timestamp'])-float(r['Start_Timestamp'])
    agg[key][0]+=1; agg[key][1]+=d
P("--- GEMM launches by (kernel, grid x, grid z): calls/step, avg us, ms/step")
for k,(c,d) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:40]:
    P("%-62s gx=%-8s gz=%-4s %6.1f %9.1f us %8.3f ms/step"%(k[0],k[1],k[2],c/n,d/c/1e3,d/1e6/n))
PY
tail -1 gpurun_out/prof_${TAG}_bench.log | cut -c1-600
cp $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/prof_${TAG}_kernel_stats.csv
