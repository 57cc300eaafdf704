#!/bin/bash
# Train on the MI355X hot path.  Same positional interface as the reference's 02_train.sh:
#   bash 02_train.sh <seed> <config.yaml> <data_path> <comment> [n_gpus]
# One anchor pack per optimizer step (--batch_size 1), 80 epochs max, repeat padding.
set -e
if [ "$#" -lt 4 ]; then
    echo "usage: bash 02_train.sh <seed> <config> <data_path> <comment> [n_gpus]"; exit 1
fi
SEED=$1; CONFIG=$2; DATA=$3; CMT=$4; NGPU=${5:-1}
ARGS="--seed ${SEED} --config ${CONFIG} --database_path ${DATA} --batch_size 1 --comment ${CMT} --num_epochs 80 --padding_type repeat"
echo "logs: $PWD/logs/model_weighted_CCE_80_1_1e-08_${CMT}   checkpoints: $PWD/out/model_weighted_CCE_80_1_1e-08_${CMT}"
if [ "${NGPU}" -gt 1 ]; then
    python -m torch.distributed.run --nnodes=1 --nproc-per-node ${NGPU} --master-addr 127.0.0.1 main.py ${ARGS}
else
    python main.py ${ARGS}
fi
