"""ORACLE (test infrastructure).  core_scripts/data_io/wav_augmentation.py:209-282 restated."""
import numpy as np


def _ad_length(x, length, repeat_pad):
    """wav_augmentation.py:229-241"""
    if length > x.shape[0]:
        if repeat_pad:
            rt = int(length / x.shape[0]) + 1
            return np.tile(x, (rt, 1))[0:length]
        tmp = np.zeros([length, 1])
        tmp[0:x.shape[0]] = x
        return tmp
    return x[0:length]


def batch_pad_for_multiview(views, wav_samp_rate, length, random_trim_nosil=False, repeat_pad=False, rng=np.random):
    """Every view is cut / tiled / zero-padded to the first view's length, then all views share
    one crop window [start, start+length).  Draws exactly one rng.rand() iff the first view is at
    least `length` long and random_trim_nosil is set (wav_augmentation.py:255-256,272-274)."""
    firstlen = views[0].shape[0]
    batch = [_ad_length(x, firstlen, repeat_pad) for x in views]
    new_len = batch[0].shape[0]
    if repeat_pad is False:
        if new_len < length:
            start, end = 0, new_len
        elif random_trim_nosil:
            start = int(rng.rand() * (new_len - length))
            end = start + length
        else:
            start, end = 0, length
    else:
        if new_len < length:
            start, end = 0, length
            rt = int(length / new_len) + 1
            batch = [np.tile(x, (rt, 1)) for x in batch]
        elif random_trim_nosil:
            start = int(rng.rand() * (new_len - length))
            end = start + length
        else:
            start, end = 0, length
    return [x[start:end] for x in batch]


def pad_eval(x, padding_type, max_len=64600):
    """datautils/asvspoof_2019_augall_3.py:49-60"""
    x_len = x.shape[0]
    if x_len >= max_len:
        return x[:max_len]
    if padding_type == "repeat":
        num_repeats = int(max_len / x_len) + 1
        return np.tile(x, (1, num_repeats))[:, :max_len][0]
    padded = np.zeros(max_len)
    padded[:x_len] = x
    return padded
