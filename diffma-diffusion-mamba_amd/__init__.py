"""diffma-diffusion-mamba_amd -- MI355X-native DiffMa denoiser hot path.

Import name: `diffma_amd` (the directory name carries a hyphen; the top-level `diffma_amd/` shim maps
the importable name onto this directory).  Layout:

  csrc/                       hand-written gfx950 HIP kernels + the C ABI (include/diffma_hip.h)
  _lib.py, hip_ops.py         ctypes binding and allocation-explicit launch wrappers
  selective_scan_interface.py reference-facing operators: selective_scan_fn, mamba_inner_fn, ...
  mamba.py, mamba_block.py    Mamba mixer ('spiral' + the baseline orders) and the blocks  (block/mamba.py, block/mamba_block.py)
  model.py, tools.py          DiffMa, DiffMa_models, spiral(), zig(), vmamba_()            (model.py, tools.py)
  diffusion/                  create_diffusion / GaussianDiffusion           (diffusion/*)
"""
__version__ = "0.1.0"

import os as _os

if _os.environ.get("DIFFMA_OVERLAP_MIXERS") == "1":
    # The opt-in two-stream mode must never meet a persistent stream-K GEMM (mamba_block.py): ask hipBLASLt's Tensile for its
    # data-parallel form before the library is initialised.  (Measured: with this setting the configuration that hung 3 of 3
    # runs completed 9 of 9, at the full two-stream speed.)
    _os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")

