"""`wav2vec2_btse` — the reference's model/wav2vec2_btse package surface (model/wav2vec2_btse/__init__.py:1-4)."""
from model.wav2vec2_btse.model import Model as wav2vec2_btse  # noqa: F401
