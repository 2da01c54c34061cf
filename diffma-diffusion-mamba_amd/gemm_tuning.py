"""hipBLASLt / rocBLAS solution selection for the GEMMs around the scan (in_proj, x_proj, dt_proj, out_proj,
adaLN, fusion MLP and their backward).

The libraries' default heuristics pick poor kernels for DiffMa's tall-skinny shapes (M = batch*196 rows against
K, N in {32, 64, 512, 1024, 2048}): e.g. `in_proj` 182 us by default vs 93 us for the best solution, and
`ddelta @ W_dt` 187 us vs 59 us.  PyTorch's TunableOp times every solution of both libraries once per shape
and records the winner; `tuned/gemm_gfx950.csv` holds those records for the bench / training shapes
(DiffMa-L/2, 256 samples per GPU; made by `tools/tune_gemm.sh`).  The file is keyed by the library build
(validator lines), so on any other ROCm build PyTorch ignores it and falls back to the default heuristics.

This only configures the vendor BLAS; nothing here touches the reference's API surface.
"""
from __future__ import annotations

import os

import torch

TUNED_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "gemm_gfx950.csv")


def enable_tuned_gemms(results_file: str | None = None, tune_missing: bool = True, write_file: str | None = None) -> bool:
    """Use the recorded GEMM solutions.  tune_missing=True (default) additionally times unseen shapes on first use
    (~1 s per new shape, once) and saves what it learnt to write_file (default: a file in the temp directory).
    That is not optional polish: for a shape without a record the libraries' default path queries the hipBLASLt
    heuristics on EVERY batched-GEMM call (~1-5 ms of host time each) and the training step becomes host-bound
    (DiffMa-L/2 at batch 256: 264 ms/step instead of 155).  Returns False (and changes nothing) without a GPU."""
    if not torch.cuda.is_available():
        return False
    import torch.cuda.tunable as tunable
    path = results_file or os.environ.get("DIFFMA_GEMM_TUNING_FILE") or TUNED_FILE
    tunable.enable(True)
    tunable.tuning_enable(bool(tune_missing))
    if tune_missing:
        # DIFFMA_TUNE_MS: time budget per candidate solution (default 20 ms; 100+ gives a steadier ranking);
        # DIFFMA_TUNE_FRESH=1: ignore the recorded table and re-time every shape (tools/tune_gemm.sh --fresh)
        tunable.set_max_tuning_duration(int(os.environ.get("DIFFMA_TUNE_MS", "20")))
        if os.environ.get("DIFFMA_TUNE_NUMCHECK", "1") == "1":
            # a candidate whose result differs from the default solution's is rejected (16-bit GEMMs differ in summation order only)
            tunable.set_numerical_check_tolerances(True, 2e-2, 2e-2)
        import tempfile
        tunable.set_filename(write_file or os.path.join(tempfile.gettempdir(), f"diffma_gemm_tuning_{os.getpid()}.csv"), False)
    else:
        tunable.set_filename(path, False)
    if os.path.exists(path) and not (tune_missing and os.environ.get("DIFFMA_TUNE_FRESH") == "1"):
        tunable.read_file(path)
    return True
