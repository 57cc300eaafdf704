"""model/wav2vec2_btse/model.py::Model of the reference: XLS-R + MLP back-end + bio transformer, on HIP kernels (scl_amd/model_btse.py)."""
from scl_amd.model_btse import Model  # noqa: F401
