"""Sharded sampling driver (reference sample.py:29-131): every rank samples its own shard, no collective.

    torchrun --nnodes=1 --nproc_per_node=N sample.py --config config/brain.yaml [--synthetic] [--ddim] [--num-batches K]

`create_diffusion(str(sample_num_steps))` + `p_sample_loop(model.forward, shape, z, clip_denoised=False, ...)` as
in the reference; `--ddim` switches to `create_diffusion("ddim<steps>")` + `ddim_sample_loop` (the API exists in
the reference, gaussian_diffusion.py:600, but no script calls it; BASELINE config 5 needs it).  The frozen VAE /
CLIP / CT encoders need network weights, so offline only `--synthetic` conditioning is possible and the result is
the sampled LATENT batch (saved as .pt), not decoded PNGs.
"""
from __future__ import annotations

import argparse
import os

import torch
import torch.distributed as dist

from .config import load_config
from .diffusion import create_diffusion
from .model import DiffMa_models


def find_model(path, key="ema"):
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    return ckpt[key] if key in ckpt else ckpt


def main(args):
    torch.manual_seed(args.seed)
    torch.set_grad_enabled(False)
    if "RANK" in os.environ and not dist.is_initialized():
        dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo")
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    device = torch.device("cuda", rank % torch.cuda.device_count()) if torch.cuda.is_available() else torch.device("cpu")
    if device.type == "cuda":
        torch.cuda.set_device(device)      # streams, graph capture and the C-ABI launches all follow the current device
        from .gemm_tuning import enable_tuned_gemms
        enable_tuned_gemms()
    latent = args.image_size // 8
    model = DiffMa_models[args.model](input_size=latent, dt_rank=args.dt_rank, d_state=args.d_state,
                                      use_mamba2=bool(args.get("use_mamba2", False))).to(device)
    if args.get("ckpt") and os.path.isfile(args.ckpt):
        model.load_state_dict(find_model(args.ckpt, args.load_ckpt_type))
    elif not args.get("synthetic", False):
        raise FileNotFoundError(f"Could not find checkpoint at {args.ckpt}")
    model.eval()
    steps = int(args.sample_num_steps)
    ddim = bool(args.get("ddim", False))
    diffusion = create_diffusion(f"ddim{steps}" if ddim else str(steps))
    n = int(args.sample_global_batch_size // world) or 1
    tokens = model.x_embedder.num_patches
    from .train import build_ct_encoder
    ct_encoder = build_ct_encoder(args, latent, device)
    os.makedirs(args.save_dir, exist_ok=True)
    g = torch.Generator(device=device).manual_seed(args.seed * world + rank)
    mk = lambda *s: torch.randn(*s, generator=g, device=device)
    out = []
    graphed = None                          # shapes are identical for every batch: one capture serves the whole run
    real = None
    if not args.get("synthetic", False) and args.get("ct_image_folder_val", None):
        # the reference's validation path (sample.py:71-110) through the seam of data.py: conditioning from the CT slice, the
        # samples decoded by the VAE.  Shard = every world-th item (DistributedSampler without shuffle would pad; here the last batch is simply shorter)
        from .data import NpyDataset, build_encoders, prepare_batch, transform_test, VAE_SCALE
        if ct_encoder is None:
            raise FileNotFoundError(f"sampling from data needs the CT_Encoder checkpoint `ct_ckpt` ({args.get('ct_ckpt', None)!r} not found)")
        ds = NpyDataset(args.ct_image_folder_val, args.mask_image_folder_val, args.mir_image_folder_val, transform=transform_test)
        real = dict(ds=ds, enc=build_encoders(args, device), order=list(range(rank, len(ds), world)), decoded=[])
    # Sampling from data walks the WHOLE validation shard, ragged tail included, like the reference's loader (drop_last=False,
    # sample.py:84-110); `num_batches` caps it when given.  Synthetic runs default to one batch.
    nb_default = -(-len(real["order"]) // n) if real is not None else 1
    nb = args.get("num_batches", None)
    for b in range(int(nb) if nb is not None else nb_default):
        nb_cur = n
        if real is not None:
            ids = real["order"][b * n:(b + 1) * n]
            if not ids:
                break
            nb_cur = len(ids)
            items = [real["ds"][j] for j in ids]
            _, y_, y2_, w_, _, _ = prepare_batch(torch.stack([it[0] for it in items]), torch.stack([it[2] for it in items]), real["enc"],
                                                 ct_encoder, device, encode_target=False)
            kw = dict(y=y_, y2=y2_, w=w_)
        z = mk(nb_cur, 4, latent, latent)
        if real is not None:
            pass                                       # conditioning prepared above
        elif ct_encoder is not None:                   # soft mask + token conditioning from the CT latent (reference sample.py:104)
            with torch.no_grad():
                ct_w, ct_y2 = ct_encoder(mk(n, 4, latent, latent))
            kw = dict(y=mk(n, 512), y2=ct_y2, w=ct_w)
        else:
            kw = dict(y=mk(n, 512), y2=mk(n, tokens, 512), w=torch.sigmoid(mk(n, tokens, 1)))
        loop = diffusion.ddim_sample_loop if ddim else diffusion.p_sample_loop
        denoiser = model.forward
        if device.type == "cuda" and not args.get("no_graph", False) and nb_cur == n:      # the ragged tail batch runs eagerly
            if graphed is None:
                from .graphed import GraphedDenoiser  # shapes are static over all steps: capture once, replay per step
                graphed = GraphedDenoiser(model, z, torch.zeros(n, device=device, dtype=torch.long), kw["y"], kw["y2"], kw["w"])
            graphed.set_condition(kw["y"], kw["y2"], kw["w"])
            denoiser = graphed
        samples = loop(denoiser, z.shape, z, clip_denoised=False, model_kwargs=kw, progress=False, device=device)
        out.append(samples.cpu())
        if real is not None:
            with torch.no_grad():
                real["decoded"].append(real["enc"].vae_decode(samples / VAE_SCALE).cpu())      # reference sample.py:108
    # (an empty shard -- more ranks than validation items -- writes an empty tensor instead of failing in torch.cat)
    torch.save(torch.cat(out) if out else torch.empty(0, 4, latent, latent), os.path.join(args.save_dir, f"latents_rank{rank}.pt"))
    if real is not None and real["decoded"]:
        torch.save(torch.cat(real["decoded"]), os.path.join(args.save_dir, f"images_rank{rank}.pt"))
    if dist.is_initialized():
        dist.destroy_process_group()
    return out


def cli(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--config", type=str, required=True)
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--ddim", action="store_true")
    p.add_argument("--use-mamba2", action="store_true")
    p.add_argument("--num-batches", type=int, default=None)
    p.add_argument("--no-graph", action="store_true", help="call the model eagerly instead of replaying a captured hipGraph")
    a = p.parse_args(argv)
    return load_config(a.config, {k: v for k, v in vars(a).items() if v is not None and k != "config"})


if __name__ == "__main__":
    main(cli())
