"""torchrun entry point: same command line as the reference's train.py (see diffma-diffusion-mamba_amd/train.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from diffma_amd.train import cli, main  # noqa: E402

if __name__ == "__main__":
    main(cli())
