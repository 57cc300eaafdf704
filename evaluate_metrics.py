"""DET curve / EER on the host (numpy) — same results as the reference's evaluate_metrics.py:3-40
(offline metric over score files; not part of the GPU path)."""
import numpy as np


def compute_det_curve(target_scores, nontarget_scores):
    target_scores, nontarget_scores = np.asarray(target_scores), np.asarray(nontarget_scores)
    scores = np.concatenate((target_scores, nontarget_scores))
    is_target = np.concatenate((np.ones(target_scores.size), np.zeros(nontarget_scores.size)))
    order = np.argsort(scores, kind="mergesort")          # stable: ties keep targets first, as the reference
    is_target = is_target[order]
    n_tar_below = np.cumsum(is_target)
    n_non_above = nontarget_scores.size - (np.arange(1, scores.size + 1) - n_tar_below)
    frr = np.concatenate(([0.0], n_tar_below / target_scores.size))
    far = np.concatenate(([1.0], n_non_above / nontarget_scores.size))
    thresholds = np.concatenate(([scores[order[0]] - 0.001], scores[order]))
    return frr, far, thresholds


def calculate_confusion_matrix(target_scores, nontarget_scores, threshold):
    tp = np.sum(target_scores > threshold)
    tn = np.sum(nontarget_scores <= threshold)
    fn = np.sum(target_scores <= threshold)
    fp = np.sum(nontarget_scores > threshold)
    return tp, tn, fp, fn


def compute_eer(target_scores, nontarget_scores):
    frr, far, thresholds = compute_det_curve(target_scores, nontarget_scores)
    i = int(np.argmin(np.abs(frr - far)))
    return float(np.mean((frr[i], far[i]))), thresholds[i]
