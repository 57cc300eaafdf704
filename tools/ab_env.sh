# interleaved A/B of bench.py under two environments in one gpurun call:  bash tools/ab_env.sh "<ENV_A>" "<ENV_B>" [rounds] [bench args...]
A="$1"; B="$2"; R=${3:-3}; shift 3 || true
for i in $(seq 1 $R); do
  for E in "$A" "$B"; do
    env $E python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %.3f ms  %.1f utt/s  frac %.4f  gemm launches/step %.0f  avg %.1f us' % (sys.argv[1], d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['launches_per_step'], d['roofline']['avg_launch_us']))" "$E"
  done
done
