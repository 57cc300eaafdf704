# the 11-view pack step of 02_train.sh (batch 11 x 64000, RawBoost off): plain vs AdamW under the backward, interleaved in one call
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for ov in 0 1; do
SCL_ADAMW_OVERLAP=$ov python3 bench.py --no-cpu-baseline --batch 11 --rawboost 0 --steps 40 --warmup 8 2>/dev/null | grep '^{"metric"' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('overlap=$ov  ms/step %.2f  utt/s %.0f  gemm frac %.3f  launches/step %.0f' % (d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['launches_per_step']))"
done
done
for b in 22 33; do
python3 bench.py --no-cpu-baseline --batch $b --rawboost 0 --steps 30 --warmup 6 2>/dev/null | grep '^{"metric"' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('batch $b  ms/step %.2f  utt/s %.0f  gemm frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac']))"
done
