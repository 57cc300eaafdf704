for spec in "wav2vec2_btse 64" "wav2vec2_btse 128"; do
set -- $spec; m=$1; b=$2
python3 bench.py --no-cpu-baseline --model $m --batch $b --rawboost 0 --steps 10 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/r5_bench_${m}_b$b.json
done
bash tools/steady_state_profile.sh bench_wav2vec2_btse_b128 --model wav2vec2_btse --batch 128 --rawboost 0
timeout 2300 python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/r5_full_gpu_tests.log
