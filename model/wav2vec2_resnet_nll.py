"""`wav2vec2_resnet_nll` — the reference's model/wav2vec2_resnet_nll.py::Model surface: HIP encoder + LL, torch-composed ResNet back-end."""
from scl_amd.model_resnet import Model  # noqa: F401
