"""DDP training driver with the reference's config surface (reference train.py:90-325).

    torchrun --nnodes=1 --nproc_per_node=N train.py --config config/brain.yaml [--autocast] [--use-mamba2] [--wandb]
                                                   [--synthetic] [--max-steps K]

Same YAML keys, same step (t ~ U{0..T-1}, training_losses, AdamW lr 1e-4 wd 0, EMA 0.999), same checkpoint
dict {"model","ema","opt","args"} at results_dir/<idx>-<model>/checkpoints/<step:07d>.pt, same log line.
Differences, all deliberate (SURVEY.md A.4-5,10):
  * --autocast means bf16 (no GradScaler needed) by default; `--amp-dtype fp16` gives the reference's own mode, fp16
    autocast with a GradScaler (train.py:95,247-263), on the same kernels (fp16 I/O, fp32 scan state).
  * a non-finite loss is detected COLLECTIVELY (all-reduce of a flag) and the step is skipped on every rank;
    the reference `continue`s on one rank only, which dead-locks DDP.
  * frozen encoders (SD-VAE, BiomedCLIP) need network weights: the real-data path runs through the seam of data.py (NpyDataset +
    an Encoders bundle: `encoders: pretrained | fake`, or your own callables); with --synthetic (the only mode
    that can run offline) latents / embeddings / soft masks are drawn as in BASELINE.md section 4.
"""
from __future__ import annotations

import argparse
import logging
import math
import os
from copy import deepcopy
from glob import glob
from time import time

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP

from .config import load_config
from .diffusion import create_diffusion
from .model import DiffMa_models


@torch.no_grad()
def update_ema(ema_model, model, decay=0.999):
    ep, mp = list(ema_model.parameters()), list(model.parameters())
    torch._foreach_mul_(ep, decay)
    torch._foreach_add_(ep, mp, alpha=1 - decay)


def requires_grad(model, flag=True):
    for p in model.parameters():
        p.requires_grad = flag


def create_logger(logging_dir, rank):
    logger = logging.getLogger("diffma")
    logger.setLevel(logging.INFO)
    logger.handlers.clear()
    if rank == 0:
        fmt = logging.Formatter("%(asctime)s | %(levelname)s | %(message)s", "%Y-%m-%d at %H:%M:%S")
        sh = logging.StreamHandler()
        sh.setFormatter(fmt)
        logger.addHandler(sh)
        if logging_dir:
            fh = logging.FileHandler(f"{logging_dir}/log_0.txt")
            fh.setFormatter(fmt)
            logger.addHandler(fh)
    else:
        logger.addHandler(logging.NullHandler())
    return logger


class SyntheticLatents:
    """Stand-in for NpyDataset + frozen encoders: yields (z_mri, y, y2, w) already in latent space.  With a CT_Encoder the
    soft mask and the token conditioning come from it, applied to a synthetic CT latent (reference train.py:239-240);
    otherwise they are drawn directly."""

    def __init__(self, n, latent, tokens, seed, ct_encoder=None):
        self.n, self.latent, self.tokens, self.seed, self.ct_encoder = n, latent, tokens, seed, ct_encoder

    def batches(self, batch, device, epoch, rank, world):
        g = torch.Generator(device=device).manual_seed(self.seed * 1000003 + epoch * 1009 + rank)
        per_rank = self.n // world
        for _ in range(per_rank // batch):
            mk = lambda *s: torch.randn(*s, generator=g, device=device)
            z, y = mk(batch, 4, self.latent, self.latent), mk(batch, 512)
            if self.ct_encoder is not None:
                with torch.no_grad():
                    w, y2 = self.ct_encoder(mk(batch, 4, self.latent, self.latent))
            else:
                y2, w = mk(batch, self.tokens, 512), torch.sigmoid(mk(batch, self.tokens, 1))
            yield z, y, y2, w


GRAPH_AUTO_MAX_BATCH = 32      # `graph_train: auto`: per-GPU batches up to this are replayed from a hipGraph


def graph_train_decision(setting, device_type, accumulation_steps, local_batch, fp16):
    """graph_train = true / false / "auto".  The reference's own configuration (config/brain.yaml: global batch 8 on 8 GPUs) runs ONE
    sample per GPU, where an eager step is bound by the host's launch rate (~50 ms) and the replayed step takes 6.5 ms: "auto" turns the
    graphed step on for per-GPU batches up to GRAPH_AUTO_MAX_BATCH when nothing rules it out (a ROCm device, no gradient accumulation,
    not fp16 -- the GradScaler's skip decision is a host branch); "true" insists (and train() raises for fp16)."""
    if isinstance(setting, str):
        v = setting.strip().lower()
        if v == "auto":
            return device_type == "cuda" and accumulation_steps == 1 and not fp16 and local_batch <= GRAPH_AUTO_MAX_BATCH
        setting = v in ("1", "true", "yes", "on")
    return bool(setting) and device_type == "cuda" and accumulation_steps == 1


def build_ct_encoder(args, latent, device):
    """The frozen CT_Encoder of train.py:158-169 when its checkpoint is there (or `synthetic_ct_encoder: true` asks for a
    randomly initialised one); None otherwise."""
    from .ct_encoder import CT_Encoder
    path = args.get("ct_ckpt", None)
    have = bool(path) and os.path.isfile(path)
    if not have and not args.get("synthetic_ct_encoder", False):
        return None
    ct = CT_Encoder(img_size=latent, patch_size=int(args.model[-1]), in_channels=4, embed_dim=512, contain_mask_token=True).to(device)
    if have:
        from .sample import find_model
        ct.load_state_dict(find_model(path))
    return ct.eval().requires_grad_(False)


DDP_BUCKET_MB = 64           # 357 MB of fp32 gradients for DiffMa-L/2 -> 6 buckets, each large enough to run at xGMI link speed


def wrap_ddp(model, device, grad_compression="none"):
    """The one place DDP is configured -- train.py and bench.py both call it, so a scaling number measures what training uses.
    Gradients only (reference train.py:153); buckets are views of the gradients, the graph is static (same parameters used
    every step), and the communication hook (a) joins the two mixer streams when the opt-in two-stream mode is on and
    (b) optionally all-reduces bf16 / fp16 copies of the buckets (`grad_compression`: "none" | "bf16" | "fp16"; halves the
    bytes on the xGMI links, the sum is still accumulated by RCCL in that dtype -- opt-in, off by default)."""
    if grad_compression not in ("none", "bf16", "fp16"):
        raise ValueError(f"grad_compression={grad_compression!r}")
    ddp = DDP(model, device_ids=[device.index] if device.type == "cuda" else None, gradient_as_bucket_view=True,
              bucket_cap_mb=DDP_BUCKET_MB, static_graph=True)
    from .mamba_block import Spiral_MambaBlock, make_ddp_comm_hook
    join = device.type == "cuda" and Spiral_MambaBlock.overlap_mixers      # both streams write gradients
    if join or grad_compression != "none":
        ddp.register_comm_hook(None, make_ddp_comm_hook(grad_compression, join))
    return ddp


def main(args):
    backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        dist.init_process_group(backend)            # torchrun env; "nccl" is RCCL on ROCm (reference load_data.py:86)
    world, rank = dist.get_world_size(), dist.get_rank()
    assert args.global_batch_size % world == 0, "Batch size must be divisible by world size."
    if torch.cuda.is_available():
        device = torch.device("cuda", rank % torch.cuda.device_count())
        torch.cuda.set_device(device)
    else:
        device = torch.device("cpu")
    torch.manual_seed(args.global_seed * world + rank)
    if device.type == "cuda":
        from .gemm_tuning import enable_tuned_gemms
        enable_tuned_gemms()          # recorded table; unseen GEMM shapes are timed once on first use

    experiment_dir = checkpoint_dir = None
    if rank == 0:
        os.makedirs(args.results_dir, exist_ok=True)
        idx = len(glob(f"{args.results_dir}/*"))
        experiment_dir = f"{args.results_dir}/{idx:03d}-{args.model.replace('/', '-')}"
        checkpoint_dir = f"{experiment_dir}/checkpoints"
        os.makedirs(checkpoint_dir, exist_ok=True)
    logger = create_logger(experiment_dir, rank)
    logger.info(f"Experiment directory created at {experiment_dir}")

    assert args.image_size % 8 == 0, "Image size must be divisible by 8 (for the VAE encoder)."
    latent = args.image_size // 8
    model = DiffMa_models[args.model](input_size=latent, dt_rank=args.dt_rank, d_state=args.d_state,
                                      use_mamba2=bool(args.get("use_mamba2", False)))
    train_steps = 0
    if args.init_from_pretrain_ckpt:
        ckpt = torch.load(args.pretrain_ckpt_path, map_location="cpu", weights_only=False)
        model.load_state_dict(ckpt["model"])
        ema = deepcopy(model).to(device)
        ema.load_state_dict(ckpt["ema"] if "ema" in ckpt else ckpt["model"])
        train_steps = args.init_train_steps
        logger.info(f"Loaded pretrain model from {args.pretrain_ckpt_path}")
    else:
        ema = deepcopy(model).to(device)
    requires_grad(ema, False)
    model = model.to(device)
    # graph_train: replay the whole optimisation step from a hipGraph (pays off below ~100 samples per GPU,
    # where the eager step is bound by the host's launch rate -- e.g. the reference's own global_batch_size of 8)
    use_graph = graph_train_decision(args.get("graph_train", False), device.type, int(args.accumulation_steps), args.global_batch_size // world,
                                     bool(args.autocast) and args.get("amp_dtype", "bf16") == "fp16")
    if use_graph:
        ddp = model                                  # no reducer hooks inside the captured backward: with several ranks the graphed
        if world > 1:                                # step all-reduces the flattened gradients itself (graphed.GraphedTrainStep)
            with torch.no_grad():
                for t_ in list(model.parameters()) + list(model.buffers()):
                    dist.broadcast(t_.data, 0)       # what the DDP constructor would have done: every rank starts from rank 0's weights
    else:
        ddp = wrap_ddp(model, device, grad_compression=str(args.get("grad_compression", "none")))
    diffusion = create_diffusion(timestep_respacing="")
    logger.info(f"DiffMa Parameters: {sum(p.numel() for p in model.parameters()):,}")
    logger.info(f"Use half-precision training? {args.autocast}")
    lr = args.lr_ if args.init_from_pretrain_ckpt else args.lr
    opt = torch.optim.AdamW(ddp.parameters(), lr=lr, weight_decay=0, fused=device.type == "cuda", capturable=use_graph)
    graphed = None

    tokens = model.x_embedder.num_patches
    if not args.get("synthetic", False):
        # the reference's real-data path (train.py:186-243): NpyDataset -> SD-VAE / BiomedCLIP / CT_Encoder -> denoiser, through the
        # seam of data.py.  `encoders: pretrained` (default) needs diffusers + open_clip + hub weights and says so when they are
        # missing; `encoders: fake` runs the whole path on deterministic stand-ins; a data.Encoders bundle can be passed in `args`.
        from .data import EncodedDataset, NpyDataset, build_encoders, transform_train
        ct_enc = build_ct_encoder(args, latent, device)
        if ct_enc is None:
            raise FileNotFoundError(f"real-data training needs the CT_Encoder checkpoint `ct_ckpt` ({args.get('ct_ckpt', None)!r} not found)")
        ds = NpyDataset(args.ct_image_folder_train, args.mask_image_folder_train, args.mir_image_folder_train, transform=transform_train)
        data = EncodedDataset(ds, build_encoders(args, device), ct_enc, seed=0)
    else:
        data = SyntheticLatents(int(args.get("synthetic_samples", 1024)), latent, tokens, args.global_seed,
                                ct_encoder=build_ct_encoder(args, latent, device))
    local_batch = args.global_batch_size // world
    logger.info(f"Dataset contains {data.n}.")

    update_ema(ema, model, decay=0)                      # EMA starts as a copy of the synced weights
    ddp.train()
    ema.eval()
    log_steps, running_loss, start_time = 0, 0.0, time()
    skipped_seen = 0                     # graphed steps dropped on the device that the step counter has already been corrected for
    max_steps = args.get("max_steps", None)
    amp = (torch.float16 if args.get("amp_dtype", "bf16") == "fp16" else torch.bfloat16) if args.autocast else None
    scaler = torch.amp.GradScaler(device.type, enabled=amp == torch.float16)     # no-op unless fp16 (reference train.py:95)
    if amp == torch.float16 and use_graph:
        raise NotImplementedError("--graph-train with fp16: the GradScaler's skip decision is a host branch; use bf16")
    logger.info(f"Training for {args.epochs} epochs...")
    for epoch in range(args.epochs):
        logger.info(f"Beginning epoch {epoch}...")
        for item, (z, y, y2, w) in enumerate(data.batches(local_batch, device, epoch, rank, world), 1):
            t = torch.randint(0, diffusion.num_timesteps, (z.shape[0],), device=device)
            if use_graph:
                if graphed is None:
                    from .graphed import GraphedTrainStep
                    graphed = GraphedTrainStep(model, ema, opt, diffusion, z, t, y, y2, w, autocast_dtype=amp, ema_decay=0.999,
                                               split=True if os.environ.get("DIFFMA_GRAPH_SPLIT") == "1" else None)
                loss = graphed.step(z, t, y, y2, w)             # forward, backward, AdamW and EMA in one replay
                train_steps += 1
                log_steps += 1
                if train_steps % args.log_every == 0:           # the only host sync of the graphed loop
                    # non-finite steps were dropped ON THE DEVICE (found_inf from the all-reduced gradients, the same decision on
                    # every rank: graphed.GraphedTrainStep._guarded_update); the host only reports them
                    lv, nskip = loss.item(), int(graphed.skipped.item())
                    if nskip > skipped_seen:                    # dropped steps do not count as optimisation steps (reference `continue`)
                        logger.info(f"nan......      ignore losses......   ({nskip} graphed steps skipped so far)")
                        train_steps -= nskip - skipped_seen
                        skipped_seen = nskip
                    running_loss = (lv if math.isfinite(lv) else 0.0) * log_steps   # the log line shows the latest loss instead of a running mean
            else:
                with torch.autocast(device.type, dtype=amp, enabled=amp is not None):
                    loss = diffusion.training_losses(ddp, z, t, dict(y=y, y2=y2, w=w))["loss"].mean()
                # [bad flag, loss value] travel in ONE all-reduce and ONE host read per step (the reference reads loss.item() too)
                lv = loss.detach().float()
                flag = torch.stack([(~torch.isfinite(lv)).float(), torch.nan_to_num(lv, nan=0.0, posinf=0.0, neginf=0.0)])
                if world > 1:
                    dist.all_reduce(flag[:1], op=dist.ReduceOp.MAX)      # every rank takes the same decision
                bad_host, loss_host = flag.tolist()
                bad = bad_host > 0
                if bad:
                    logger.info("nan......      ignore losses......")
                    # backward still runs (it keeps DDP's bucket all-reduces matched across ranks); what it adds is thrown
                    # away, what earlier good micro-batches accumulated is put back
                    kept = [None if p_.grad is None else p_.grad.detach().clone() for p_ in model.parameters()]
                    scaler.scale(loss).backward()
                    for p_, g_ in zip(model.parameters(), kept):
                        if g_ is None:
                            p_.grad = None
                        else:
                            p_.grad.copy_(g_)
                    continue                                 # like the reference (train.py:254-256): the step is not counted, no log / ckpt check
                else:
                    scaler.scale(loss).backward()
                    if train_steps % args.accumulation_steps == 0:
                        scaler.step(opt)                     # fp16: unscales, skips the update on inf/nan gradients
                        scaler.update()
                        update_ema(ema, model)
                        opt.zero_grad(set_to_none=True)
                    running_loss += loss_host
                    log_steps += 1
                train_steps += 1
            if train_steps % args.log_every == 0:
                if device.type == "cuda":
                    torch.cuda.synchronize()
                steps_per_sec = max(log_steps, 1) / (time() - start_time)
                avg = torch.tensor(running_loss / max(log_steps, 1), device=device)
                dist.all_reduce(avg, op=dist.ReduceOp.SUM)
                pct = local_batch * item / data.n * 100
                logger.info(f"({pct:.1f}%) (step={train_steps:07d}) Train Loss: {avg.item() / world:.4f}, Train Steps/Sec: {steps_per_sec:.2f}")
                running_loss, log_steps, start_time = 0.0, 0, time()
            if train_steps % args.ckpt_every == 0 and train_steps > 0:
                if rank == 0:
                    path = f"{checkpoint_dir}/{train_steps:07d}.pt"
                    torch.save({"model": model.state_dict(), "ema": ema.state_dict(), "opt": opt.state_dict(), "args": dict(args)}, path)
                    logger.info(f"Saved checkpoint to {path}")
                dist.barrier()
            if max_steps is not None and train_steps >= max_steps:
                break
        if max_steps is not None and train_steps >= max_steps:
            break
    model.eval()
    logger.info("Done!")
    dist.destroy_process_group()
    return train_steps


def cli(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--wandb", action="store_true", help="accepted for compatibility; wandb is not installed offline")
    p.add_argument("--autocast", action="store_true", help="autocast (bf16 unless --amp-dtype fp16)")
    p.add_argument("--amp-dtype", default="bf16", choices=["bf16", "fp16"], help="fp16 = the reference's mode: fp16 autocast + GradScaler")
    p.add_argument("--use-mamba2", action="store_true")
    p.add_argument("--synthetic", action="store_true", help="synthetic latents/conditioning instead of datasets + frozen encoders")
    p.add_argument("--max-steps", type=int, default=None)
    p.add_argument("--grad-compression", default=None, choices=["none", "bf16", "fp16"],
                   help="all-reduce 16-bit copies of the DDP gradient buckets (opt-in; default none = fp32 like the reference)")
    p.add_argument("--graph-train", nargs="?", const="true", default=None, choices=["true", "false", "auto"],
                   help="replay the whole optimisation step from a hipGraph (for small batches; with several ranks: two graphs around one "
                        "gradient all-reduce); `auto`: on for per-GPU batches up to 32.  Also the YAML key graph_train")
    p.add_argument("--config", type=str, required=True)
    a = p.parse_args(argv)
    over = {k: v for k, v in vars(a).items() if v is not None and k != "config"}
    return load_config(a.config, over)


if __name__ == "__main__":
    main(cli())
