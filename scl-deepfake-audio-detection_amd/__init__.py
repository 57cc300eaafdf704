"""scl-deepfake-audio-detection_amd — MI355X-native training hot path of SCL-Deepfake-audio-detection.

Only what the hot path needs lives here: `csrc/` (HIP kernels + the C ABI of libscl_hip.so),
`lib.py` (ctypes binding), `ops.py` (tensor-level wrappers) and the host-side mirror of the
reference's model / data plugin interface.  Import it as `scl_amd` (the alias package at the
repository root), since this directory's name is not a Python identifier.
"""
__version__ = "0.1.0"
