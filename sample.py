"""torchrun entry point: same command line as the reference's sample.py (see diffma-diffusion-mamba_amd/sample.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from diffma_amd.sample import cli, main  # noqa: E402

if __name__ == "__main__":
    main(cli())
