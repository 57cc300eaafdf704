"""Data plugin `eval_only` — arbitrary evaluation sets described by `protocol.txt` lines
"<relative path> <subset> <label>" (reference: datautils/eval_only.py:24-89)."""
from scl_amd.pack import EvalDataset, pad_eval  # noqa: F401

pad = pad_eval


def genList(dir_meta, is_train=False, is_eval=True, is_dev=False):
    files = []
    with open(dir_meta) as f:
        for line in f:
            parts = line.strip().split()
            if len(parts) == 3:
                files.append(parts[0])
    return [], files


class Dataset_for_eval(EvalDataset):
    def __init__(self, list_IDs, base_dir, padding_type="zero"):
        super().__init__(list_IDs, base_dir, padding_type, subdir="")


class Dataset_for(Dataset_for_eval):  # the reference's training-side class of this plugin is never used by main.py
    def __init__(self, args, list_IDs, labels, base_dir, algo=5, **kwargs):
        super().__init__(list_IDs, base_dir)
