"""Importable alias of the package directory `scl-deepfake-audio-detection_amd/` (its name is not a
Python identifier).  `import scl_amd.ops` resolves to `scl-deepfake-audio-detection_amd/ops.py`."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "scl-deepfake-audio-detection_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
