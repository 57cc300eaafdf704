"""`model.conformer` — the reference's model/conformer.py surface (ConformerBlock :180, Conformer :219) on the HIP kernels."""
from scl_amd.conformer import Conformer, ConformerBlock  # noqa: F401
