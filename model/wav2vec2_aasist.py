"""`wav2vec2_aasist` — the reference's model/wav2vec2_aasist.py::Model surface: HIP encoder + LL and the hand-written HIP AASIST back-end (residual stack, graph attention, pooling, readout), one autograd boundary each."""
from scl_amd.model_aasist import Model  # noqa: F401
