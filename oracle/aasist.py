"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) — deterministic parameter fill for the AASIST golden.

The AASIST back-end has 447 k parameters; instead of shipping the reference's random initial state in the fixture, both
the generator (oracle/gen_golden.py::gen_aasist, which runs the reference's own model/wav2vec2_aasist.py::Model) and the
parity test fill the state dict by this rule, so the fixture only holds the input, the outputs and the gradients.
"""
import numpy as np


def fill_state(shapes, seed=0):
    """shapes: {state-dict key: shape}.  Returns {key: float32 array} (int64 for num_batches_tracked), visiting the keys in
    sorted order so the values do not depend on module construction order."""
    rng = np.random.RandomState(seed)
    out = {}
    for k in sorted(shapes):
        shape = tuple(shapes[k])
        if k.endswith("num_batches_tracked"):
            out[k] = np.zeros(shape, dtype=np.int64)
            continue
        n = int(np.prod(shape)) if shape else 1
        v = rng.standard_normal(n).reshape(shape)
        leaf = k.split(".")[-1]
        if leaf == "running_var":
            v = 0.5 + np.abs(v)
        elif leaf == "running_mean":
            v = 0.1 * v
        elif leaf == "weight" and len(shape) == 1:          # BatchNorm scale
            v = 1.0 + 0.1 * v
        elif leaf == "bias":
            v = 0.05 * v
        elif len(shape) >= 2:                                # conv / linear / attention vectors: ~ 1/sqrt(fan_in)
            fan_in = int(np.prod(shape[1:])) if len(shape) > 2 or shape[1] > 1 else shape[0]
            v = v / np.sqrt(max(fan_in, 1))
        out[k] = v.astype(np.float32)
    return out
