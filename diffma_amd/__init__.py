"""Importable alias for the package directory `diffma-diffusion-mamba_amd/` (a hyphen is not a legal
module name).  `import diffma_amd` / `from diffma_amd.model import DiffMa_models` resolve there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "diffma-diffusion-mamba_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
