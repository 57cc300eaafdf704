"""`wav2vec2_aasist` — the reference's model/wav2vec2_aasist.py::Model surface: HIP encoder + LL, torch-composed graph back-end."""
from scl_amd.model_aasist import Model  # noqa: F401
