"""`wav2vec2_linear_nll` — the reference's model/wav2vec2_linear_nll.py::Model surface on the MI355X HIP path."""
from scl_amd.model_linear import Model, loss_custom  # noqa: F401
