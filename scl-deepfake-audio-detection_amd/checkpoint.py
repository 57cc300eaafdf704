"""Reading the pre-trained XLS-R checkpoint the reference starts from.

The reference builds its encoder with `fairseq.checkpoint_utils.load_model_ensemble_and_task(['pretrained/xlsr2_300m.pt'])`
(model/xlsr.py:14-16): the file is a fairseq checkpoint, a pickled dict {'model': OrderedDict of tensors under fairseq's
own parameter names, 'cfg' (fairseq >= 0.10.2: nested dict / omegaconf DictConfig) or 'args' (older: argparse.Namespace)}.
fairseq is not a dependency of this framework, so the file is read with a tolerant unpickler (classes of packages that are
not installed — omegaconf, fairseq dataclasses — become inert stubs) and only two things are taken from it: the 'model'
tensors, which go under `ssl_model.model.` into the flat parameter buffer, and the encoder's dropout / layerdrop
probabilities, which SURVEY.md 8(a) M2 says must come from the checkpoint's cfg.
"""
import os
import pickle
import warnings

import torch

DEFAULT_PRETRAINED = os.path.join("pretrained", "xlsr2_300m.pt")     # model/xlsr.py:14 (relative to the working directory)
PREFIX = "ssl_model.model."
DROPOUT_KEYS = ("dropout", "attention_dropout", "activation_dropout", "dropout_input", "dropout_features")


class _Stub:
    """Stand-in for a class whose package is not installed: keeps whatever state the pickle hands it."""

    def __init__(self, *a, **k):
        self._args, self._kwargs = a, k

    def __setstate__(self, state):
        self.__dict__["_state"] = state
        if isinstance(state, dict):
            self.__dict__.update(state)

    def __call__(self, *a, **k):
        return _Stub(*a, **k)


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError, ModuleNotFoundError):
            return type(name, (_Stub,), {"__module__": module})


class _tolerant_pickle:          # the `pickle_module` protocol torch.load expects
    __name__ = "scl_tolerant_pickle"
    Unpickler = _TolerantUnpickler
    load = staticmethod(lambda f, **kw: _TolerantUnpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump, dumps = staticmethod(pickle.dump), staticmethod(pickle.dumps)
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL


def _as_dict(obj):
    """cfg['model'] as a plain dict, whatever it was pickled as (dict, Namespace, omegaconf stub)."""
    if obj is None:
        return {}
    if isinstance(obj, dict):
        return obj
    for attr in ("_content", "_state"):
        inner = getattr(obj, attr, None) if not isinstance(obj, dict) else None
        if isinstance(inner, dict):
            return {k: getattr(v, "_val", v) for k, v in inner.items()}
    return {k: v for k, v in vars(obj).items() if not k.startswith("_")}


def read_fairseq_checkpoint(path):
    """-> (tensors: {fairseq name: Tensor}, probs: {dropout..., encoder_layerdrop}) of a fairseq wav2vec2 checkpoint."""
    ck = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_tolerant_pickle)
    if not isinstance(ck, dict) or "model" not in ck:
        raise ValueError("%s is not a fairseq checkpoint (no 'model' entry)" % path)
    tensors = {k: v for k, v in ck["model"].items() if torch.is_tensor(v)}
    cfg = ck.get("cfg")
    mcfg = _as_dict(_as_dict(cfg).get("model")) if cfg is not None else {}
    if not mcfg and ck.get("args") is not None:
        mcfg = _as_dict(ck["args"])
    probs = {}
    for k in DROPOUT_KEYS + ("encoder_layerdrop",):
        v = mcfg.get(k)
        v = getattr(v, "_val", v)
        if isinstance(v, (int, float)):
            probs[k] = float(v)
    return tensors, probs


def load_pretrained_into(model, path):
    """Copy the checkpoint's encoder tensors into `model` (names prefixed with ssl_model.model.), set the encoder's element-dropout
    probabilities from its cfg and return all of them (LayerDrop included; the caller applies that one)."""
    tensors, probs = read_fairseq_checkpoint(path)
    own = model.state_dict()
    sd, skipped = {}, []
    prefix = getattr(model, "front_prefix", "") + PREFIX        # wav2vec2_btse keeps its encoder under backend.ssl_model.model.
    for k, v in tensors.items():
        name = prefix + k
        if name in own and tuple(own[name].shape) == tuple(v.shape):
            sd[name] = v.float()
        else:
            skipped.append(k)
    missing = [k for k in own if k.startswith(prefix) and k not in sd]
    if missing:
        raise ValueError("pre-trained checkpoint %s lacks %d encoder tensors of this architecture (first: %s)" % (path, len(missing), missing[:3]))
    model.load_state_dict(sd, strict=False)
    # element dropout inside the encoder: the probabilities the checkpoint was trained with are the ones fairseq's model object carries
    # into fine-tuning (model/xlsr.py:14-16 builds the model from the checkpoint's own cfg and runs it in train mode, :33-41).
    # dropout_features acts on a tensor the features_only output never reads.
    cfg = model.cfg
    for key in ("dropout", "attention_dropout", "activation_dropout", "dropout_input"):
        if key in probs:
            setattr(cfg, key, float(probs[key]))
    nz = {k: v for k, v in probs.items() if k in DROPOUT_KEYS and k != "dropout_features" and v != 0.0}
    if nz:
        print("[scl] encoder element dropout from the checkpoint cfg: %s" % nz)
    if skipped:
        print("[scl] pre-trained checkpoint: %d tensors without a counterpart ignored (first: %s)" % (len(skipped), skipped[:3]))
    return probs
