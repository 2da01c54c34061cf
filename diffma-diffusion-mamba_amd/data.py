"""The data path either side of the denoiser: paired CT / mask / MRI slices on disk, and the three frozen
networks the reference runs on them before every step -- SD-VAE encode, BiomedCLIP image tower, (plus the CT_Encoder of
ct_encoder.py) -- and SD-VAE decode after sampling.

What the reference does (study only, nothing copied):
  load_data.py:14-38       NpyDataset(image_folder, mask_folder, mri_folder, transform): one .npy per slice, the SAME file name in
                           the three folders; returns (ct, (mask + 1) / 2, mri)
  load_data.py:41-86       transform_train / transform_test: resize to 224 x 224 (image bilinear, mask / MRI nearest), to_tensor
  train.py:228-243         per batch: 1 -> 3 channels, MRI rescaled to [-1, 1] when out of range, z = vae.encode(mri) * 0.18215,
                           x_ = vae.encode(ct) * 0.18215, (w, y2) = ct_encoder(x_), y = clip.visual(ct)
  sample.py:86-110         the same conditioning, then vae.decode(samples / 0.18215)

The pretrained nets (`stabilityai/sd-vae-ft-*`, `microsoft/BiomedCLIP-*`) need `diffusers` / `open_clip` and hub weights; neither is
available offline.  So this module builds the SEAM: the dataset, the per-batch preparation, and an `Encoders` bundle of three
callables with the reference's shapes and scale conventions.  `pretrained_encoders()` binds the real nets when their packages and
weights are present (and says exactly what is missing otherwise); `FakeEncoders` is a deterministic stand-in with the same
interface for tests and dry runs -- it is NOT a model of the real encoders.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Callable

import numpy as np
import torch
import torch.nn.functional as F

VAE_SCALE = 0.18215                       # reference train.py:238, sample.py:108


class NpyDataset(torch.utils.data.Dataset):
    """One .npy per slice with the same file name in the CT, mask and MRI folders (reference load_data.py:14-38).
    Returns (ct, mask, mri); the mask is mapped from [-1, 1] to [0, 1] like the reference does.  The listing is sorted (the
    reference relies on os.listdir order, which is file-system dependent)."""

    def __init__(self, image_folder, mask_folder, mri_folder, transform=None):
        self.image_folder, self.mask_folder, self.mri_folder, self.transform = image_folder, mask_folder, mri_folder, transform
        self.images = sorted(f for f in os.listdir(image_folder) if f.endswith(".npy"))

    def __len__(self):
        return len(self.images)

    def __getitem__(self, index):
        name = self.images[index]
        image = np.load(os.path.join(self.image_folder, name))
        mask = np.load(os.path.join(self.mask_folder, name))
        mri = np.load(os.path.join(self.mri_folder, name))
        if self.transform:
            image, mask, mri = self.transform(image, mask, mri)
        mask = (mask + 1) / 2
        return image, mask, mri


def _to_tensor(a):
    """torchvision's to_tensor for a 2-D array: [1, H, W] float32, uint8 scaled to [0, 1], everything else as is."""
    t = torch.from_numpy(np.ascontiguousarray(a))
    t = t.float().div(255.0) if t.dtype == torch.uint8 else t.float()
    return t[None] if t.dim() == 2 else t.permute(2, 0, 1)


def _resize(t, size, mode):
    if tuple(t.shape[-2:]) == tuple(size):
        return t                                                    # the reference's data is stored at 224 x 224: identity
    kw = dict(mode="bilinear", antialias=True, align_corners=False) if mode == "bilinear" else dict(mode="nearest")
    return F.interpolate(t[None], size=size, **kw)[0]


def transform_test(image, mask, mri, size=(224, 224)):
    """Resize (CT bilinear, mask / MRI nearest) and convert to [1, H, W] tensors (reference load_data.py:69-84; the train
    transform is the same function -- its augmentations are commented out in the reference, load_data.py:50-60).  Resampling
    uses torch's interpolate instead of PIL's: identical when the stored slices already have the target size."""
    return (_resize(_to_tensor(image), size, "bilinear"), _resize(_to_tensor(mask), size, "nearest"), _resize(_to_tensor(mri), size, "nearest"))


transform_train = transform_test


@dataclass
class Encoders:
    """The frozen networks around the denoiser as three callables (all under no_grad, on the batch's device):
    vae_encode(img [B, 3, H, W] in [-1, 1]) -> latent [B, 4, H/8, W/8], ALREADY multiplied by 0.18215 and sampled
    vae_decode(latent [B, 4, h, w])          -> image [B, 3, 8h, 8w]    (the caller divides by 0.18215 first, like sample.py:108)
    clip_embed(img [B, 3, H, W])             -> [B, 512]                (BiomedCLIP's visual tower)"""
    vae_encode: Callable
    vae_decode: Callable
    clip_embed: Callable
    name: str = "custom"


class FakeEncoders:
    """Deterministic stand-ins with the real interfaces: a fixed random 8 x 8 patch projection as the "VAE" (and its pseudo-inverse
    as the decoder), a fixed random projection of pooled patches as the "CLIP" tower.  For tests and dry runs of the data path."""

    def __init__(self, seed=0, embed_dim=512):
        g = torch.Generator().manual_seed(seed)
        self.w_enc = torch.randn(4, 3 * 64, generator=g) / 8.0                    # 8 x 8 x 3 patch -> 4 latent channels
        self.w_dec = torch.linalg.pinv(self.w_enc)                                # [192, 4]
        self.w_clip = torch.randn(embed_dim, 3 * 16 * 16, generator=g) / 16.0     # 16 x 16 pooled image -> embedding

    def vae_encode(self, img):
        B, C, H, W = img.shape
        p = F.unfold(img.float(), kernel_size=8, stride=8)                         # [B, 192, (H/8)*(W/8)]
        z = torch.einsum("ok,bkn->bon", self.w_enc.to(img.device), p).view(B, 4, H // 8, W // 8)
        return z * VAE_SCALE

    def vae_decode(self, z):
        B, _, h, w = z.shape
        p = torch.einsum("ko,bon->bkn", self.w_dec.to(z.device), z.float().view(B, 4, h * w))
        return F.fold(p, output_size=(8 * h, 8 * w), kernel_size=8, stride=8)

    def clip_embed(self, img):
        pooled = F.adaptive_avg_pool2d(img.float(), 16).flatten(1)                 # [B, 768]
        return pooled @ self.w_clip.to(img.device).t()

    def bundle(self):
        return Encoders(self.vae_encode, self.vae_decode, self.clip_embed, name="fake")


def pretrained_encoders(vae="ema", device="cuda"):
    """Bind the reference's pretrained nets (train.py:156,176-177): needs `diffusers`, `open_clip` and their hub weights."""
    missing = []
    try:
        from diffusers.models import AutoencoderKL
    except ImportError:
        AutoencoderKL = None
        missing.append("diffusers (AutoencoderKL, stabilityai/sd-vae-ft-%s)" % vae)
    try:
        from open_clip import create_model_from_pretrained
    except ImportError:
        create_model_from_pretrained = None
        missing.append("open_clip (hf-hub:microsoft/BiomedCLIP-PubMedBERT_256-vit_base_patch16_224)")
    if missing:
        raise RuntimeError("the pretrained encoders need " + " and ".join(missing) + "; neither the packages nor the weights are "
                           "available offline -- pass your own data.Encoders bundle, or `encoders: fake` for a dry run of the data path")
    vae_net = AutoencoderKL.from_pretrained(f"stabilityai/sd-vae-ft-{vae}").to(device).eval()
    clip_model, _ = create_model_from_pretrained("hf-hub:microsoft/BiomedCLIP-PubMedBERT_256-vit_base_patch16_224")
    tower = clip_model.visual.to(device).eval()
    return Encoders(lambda img: vae_net.encode(img).latent_dist.sample().mul_(VAE_SCALE), lambda z: vae_net.decode(z).sample, tower, name="pretrained")


def build_encoders(args, device):
    kind = args.get("encoders", "pretrained")
    if isinstance(kind, Encoders):
        return kind
    if kind == "fake":
        return FakeEncoders(int(args.get("global_seed", 0))).bundle()
    return pretrained_encoders(args.get("vae", "ema"), device)


@torch.no_grad()
def prepare_batch(x_ct, z_mri, encoders: Encoders, ct_encoder, device, encode_target=True):
    """What the reference does between the loader and the denoiser (train.py:228-243, sample.py:86-106):
    returns (z latent of the MRI target, y CLIP embedding of the CT, y2 CT tokens, w soft mask, and the 3-channel CT / MRI images).
    encode_target=False (sampling: only the conditioning is needed) skips the VAE encode of the MRI target; z is then None."""
    x_ct = torch.cat([x_ct] * 3, dim=1).to(device) if x_ct.shape[1] == 1 else x_ct.to(device)
    z_mri = torch.cat([z_mri] * 3, dim=1).to(device) if z_mri.shape[1] == 1 else z_mri.to(device)
    z = None
    if encode_target:         # (sampling never reads the target: no renormalisation pass, no host-synchronising range check)
        if not torch.all((z_mri >= -1) & (z_mri <= 1)):
            z_mri = ((z_mri - z_mri.min()) * 1.0 / (z_mri.max() - z_mri.min())) * 2.0 - 1.0
        z = encoders.vae_encode(z_mri)
    x_lat = encoders.vae_encode(x_ct)
    w, y2 = ct_encoder(x_lat)
    y = encoders.clip_embed(x_ct)
    return z, y, y2, w, x_ct, z_mri


class EncodedDataset:
    """The train loop's view of the real-data path: the same `.batches(batch, device, epoch, rank, world)` generator as the
    synthetic source (train.SyntheticLatents), fed by NpyDataset through a DistributedSampler-style shard (shuffle with the
    reference's seed 0 + epoch, load_data.py:89-91) and `prepare_batch`."""

    def __init__(self, dataset, encoders, ct_encoder, seed=0):
        self.ds, self.enc, self.ct, self.seed = dataset, encoders, ct_encoder, seed
        self.n = len(dataset)

    def batches(self, batch, device, epoch, rank, world):
        g = torch.Generator().manual_seed(self.seed + epoch)
        order = torch.randperm(self.n, generator=g).tolist()
        order = order[rank::world][: (self.n // world)]
        for i in range(0, len(order) - batch + 1, batch):
            items = [self.ds[j] for j in order[i:i + batch]]
            ct = torch.stack([it[0] for it in items])
            mri = torch.stack([it[2] for it in items])
            z, y, y2, w, _, _ = prepare_batch(ct, mri, self.enc, self.ct, device)
            yield z, y, y2, w
