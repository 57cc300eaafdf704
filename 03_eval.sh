#!/bin/bash
# Score an evaluation set.  Same positional interface as the reference's 03_eval.sh:
#   bash 03_eval.sh <config.yaml> <data_path> <batch_size> <model_path> <eval_output>
# Writes "<utt> <logp_spoof> <logp_bonafide>" lines; compute the EER with evaluate_metrics.compute_eer.
set -e
if [ "$#" -lt 5 ]; then
    echo "usage: bash 03_eval.sh <config> <data_path> <batch_size> <model_path> <eval_output>"; exit 1
fi
python main.py --config $1 --database_path $2 --batch_size $3 --eval --model_path $4 --eval_output $5
