"""`wav2vec2_resnet_nll` — the reference's model/wav2vec2_resnet_nll.py::Model surface: HIP encoder + LL and the HIP ResNet back-end (scl_amd/resnet_head.py: implicit-GEMM convolutions, fused BatchNorm / pooling kernels)."""
from scl_amd.model_resnet import Model  # noqa: F401
