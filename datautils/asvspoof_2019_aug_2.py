"""Data plugin `asvspoof_2019_aug_2` — resolved by `config['data']['name']` exactly as the reference's main.py:328-330 does.

Same exports as the reference module of this name (genList, Dataset_for, Dataset_for_eval, the augmenter
functions looked up by string), same pack order / labels / RNG order (recipe 'aug_2' of scl_amd.pack);
the waveform work runs on the MI355X through scl_amd.augment instead of numpy/scipy/pydub in workers.
"""
from scl_amd.pack import (EvalDataset, PackDataset, RawBoost12, background_noise_wrapper, gen_list_scp, pad_eval,  # noqa: F401
                          pitch_wrapper, reverb_wrapper, speed_wrapper)

genList = gen_list_scp
pad = pad_eval


class Dataset_for(PackDataset):
    def __init__(self, args, list_IDs, labels, base_dir, **kwargs):
        super().__init__("aug_2", args, list_IDs, labels, base_dir, **kwargs)


class Dataset_for_eval(EvalDataset):
    def __init__(self, list_IDs, base_dir, padding_type="zero"):
        super().__init__(list_IDs, base_dir, padding_type, subdir="eval")
