#!/bin/bash
# Train on the MI355X hot path.  Same positional interface as the reference's 02_train.sh:
#   bash 02_train.sh <seed> <config.yaml> <data_path> <comment> [n_gpus]
# One anchor pack per optimizer step (--batch_size 1, the reference's recipe: /root/reference 02_train.sh:50-57), 80 epochs max, repeat padding.
#
# Throughput note (one MI355X, round 6: profiles/r6_upload_stream_final.txt, r6_pack11_gemm_classes.txt): an 11-view pack is M = 2189 encoder rows —
# 14.3 ms per step = 770 utterances/s (14.4 ms end to end from FLAC files), half of what the chip does on a full batch: the step is kernel-bound
# and its GEMM launches are 21 - 36 us each.  PACKS=k puts k anchor packs into one optimizer step (main.py --batch_size k: SupCon positives /
# negatives stay inside their pack, the CE / SupCon terms are averaged over the packs; the reference's reshape only works for k = 1):
# k = 3 -> 1240 utterances/s end to end (26.7 ms per step), k = 6 -> 1495 (44.1 ms).  k = 3 .. 6 is the recommended setting when the
# learning-rate schedule is re-tuned for the larger step; the default stays 1 so that the recipe is the reference's own.
set -e
if [ "$#" -lt 4 ]; then
    echo "usage: bash 02_train.sh <seed> <config> <data_path> <comment> [n_gpus]"; exit 1
fi
SEED=$1; CONFIG=$2; DATA=$3; CMT=$4; NGPU=${5:-1}
ARGS="--seed ${SEED} --config ${CONFIG} --database_path ${DATA} --batch_size ${PACKS:-1} --comment ${CMT} --num_epochs 80 --padding_type repeat"
echo "logs: $PWD/logs/model_weighted_CCE_80_1_1e-08_${CMT}   checkpoints: $PWD/out/model_weighted_CCE_80_1_1e-08_${CMT}"
if [ "${NGPU}" -gt 1 ]; then
    python -m torch.distributed.run --nnodes=1 --nproc-per-node ${NGPU} --master-addr 127.0.0.1 main.py ${ARGS}
else
    python main.py ${ARGS}
fi
