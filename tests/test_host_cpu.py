"""Host-side logic on CPU: config loader, and the N > 1 data-parallel path (world_size 2, gloo).

The product operators have no CPU implementation; for these plumbing tests the TEST injects the CPU oracle
behind the mixer's operator (monkeypatch), exactly the substitution BASELINE config 1 describes."""
import os
import socket
import sys
import textwrap

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_loader_resolves_omegaconf_scalars(tmp_path):
    from diffma_amd.config import load_config

    p = tmp_path / "brain.yaml"
    p.write_text(textwrap.dedent("""\
        epochs: 50
        log_every: 10
        ckpt_every: 50_000
        lr: 1e-4
        lr_: 1e-4
        model: "DiffMa-L/2"
        init_train_steps: 0_800_000
        init_from_pretrain_ckpt: False
        results_dir: "./results/brain"
    """))
    cfg = load_config(str(p), {"autocast": True, "use_mamba2": None})
    assert cfg.lr == 1e-4 and isinstance(cfg.lr, float)
    assert cfg.init_train_steps == 800000 and cfg.ckpt_every == 50000
    assert cfg.model == "DiffMa-L/2" and cfg.init_from_pretrain_ckpt is False
    assert cfg.autocast is True and "use_mamba2" not in cfg
    shipped = load_config(os.path.join(ROOT, "config", "diffma_l2_synthetic.yaml"))
    assert shipped.lr == 1e-4 and shipped.sample_num_steps == 250


def _oracle_spiral_ssm(xz, conv_w, conv_b, x_proj_w, dt_proj_w, dt_proj_b, A, Dskip, scan_index):
    """CPU stand-in with the contract of selective_scan_interface.spiral_ssm (pre-out_proj merged output)."""
    from oracle.mamba_ref import mamba_inner_ref

    Bsz, L, D2 = xz.shape
    Din = D2 // 2
    xz_cm = xz.transpose(1, 2)                                   # (B, 2Din, L)
    y = 0
    for k in range(scan_index.shape[0]):
        idx = scan_index[k].long()
        yk = mamba_inner_ref(xz_cm[:, :, idx], conv_w, conv_b, x_proj_w, dt_proj_w, None, None, A, None, None, Dskip,
                             delta_bias=dt_proj_b, delta_softplus=True, dtype=torch.float32, return_pre_proj=True)  # (B, Din, L)
        y = y + torch.zeros_like(yk).index_add(2, idx, yk)        # step l belongs to token idx[l]
    return y.transpose(1, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _ddp_worker(rank, world, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import diffma_amd.mamba as mamba_mod
    from diffma_amd import train as train_mod
    from diffma_amd.config import Config

    mamba_mod.spiral_ssm = _oracle_spiral_ssm                     # test-only substitution (see module docstring)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = Config(model="DiffMa-S/7", image_size=224, dt_rank=16, d_state=16, global_batch_size=4, global_seed=0, lr=1e-4, lr_=1e-4,
                 epochs=1, accumulation_steps=1, log_every=1, ckpt_every=2, results_dir=os.path.join(tmpdir, "res"),
                 init_from_pretrain_ckpt=False, pretrain_ckpt_path="", init_train_steps=0, autocast=False, synthetic=True,
                 synthetic_samples=64, max_steps=2)
    steps = train_mod.main(cfg)
    assert steps == 2


def test_ddp_training_two_ranks_gloo(tmp_path):
    """world_size-2 data-parallel training of DiffMa-S/7 (16 tokens) on CPU: DDP wrap, collective NaN guard, EMA,
    loss all-reduce, rank-0 checkpoint + barrier."""
    port = _free_port()
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ck = [os.path.join(dp, f) for dp, _, fs in os.walk(tmp_path) for f in fs if f.endswith(".pt")]
    assert len(ck) == 1 and ck[0].endswith("0000002.pt")
    sd = torch.load(ck[0], map_location="cpu", weights_only=False)
    assert set(sd) == {"model", "ema", "opt", "args"}
    assert len(sd["model"]) == len(sd["ema"]) and "pos_embed" in sd["model"]
    # after 2 optimiser steps the EMA lags the model but is no longer the init
    some = "blocks.0.mamba1.in_proj.weight"
    assert not torch.equal(sd["model"][some], sd["ema"][some])


def _hook_worker(rank, world, port, compression, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from diffma_amd.train import wrap_ddp

    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(24, 40), torch.nn.SiLU(), torch.nn.Linear(40, 8))
    ddp = wrap_ddp(net, torch.device("cpu"), grad_compression=compression)
    x = torch.randn(5, 24, generator=torch.Generator().manual_seed(100 + rank))
    for _ in range(2):                                   # static_graph: the second iteration runs the rebuilt buckets
        net.zero_grad(set_to_none=True)
        ddp(x).square().sum().backward()
    if rank == 0:
        torch.save({k: p.grad.clone() for k, p in net.named_parameters()}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("compression", ["none", "bf16", "fp16"])
def test_ddp_wrapper_and_gradient_compression_hook_two_ranks_gloo(compression, tmp_path):
    """train.wrap_ddp (the wrapper bench.py shares): the all-reduced gradient equals the mean of the per-rank gradients --
    exactly for fp32 buckets, within the 16-bit rounding for the opt-in compressed all-reduce."""
    out_path = str(tmp_path / "grads.pt")
    mp.spawn(_hook_worker, args=(2, _free_port(), compression, out_path), nprocs=2, join=True)
    got = torch.load(out_path)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(24, 40), torch.nn.SiLU(), torch.nn.Linear(40, 8))
    want = None
    for r in range(2):
        net.zero_grad(set_to_none=True)
        net(torch.randn(5, 24, generator=torch.Generator().manual_seed(100 + r))).square().sum().backward()
        gr = {k: p.grad.clone() for k, p in net.named_parameters()}
        want = gr if want is None else {k: (want[k] + gr[k]) / 2 for k in gr}
    tol = {"none": dict(rtol=1e-6, atol=1e-6), "bf16": dict(rtol=2e-2, atol=2e-2), "fp16": dict(rtol=2e-3, atol=2e-3)}[compression]
    for k in want:
        torch.testing.assert_close(got[k], want[k], **tol, msg=lambda m, k=k: f"{k}: {m}")


def test_synthetic_batches_through_ct_encoder():
    """`synthetic_ct_encoder: true`: the soft mask and token conditioning of the synthetic stream come from a CT_Encoder
    (reference train.py:239-240) instead of being drawn directly."""
    from diffma_amd.config import load_config  # noqa: F401  (package import check)
    from diffma_amd.ct_encoder import CT_Encoder
    from diffma_amd.train import SyntheticLatents

    ct = CT_Encoder(img_size=28, patch_size=7, in_channels=4, embed_dim=512).eval()
    data = SyntheticLatents(8, 28, 16, seed=0, ct_encoder=ct)
    z, y, y2, w = next(iter(data.batches(4, torch.device("cpu"), 0, 0, 1)))
    assert z.shape == (4, 4, 28, 28) and y.shape == (4, 512) and y2.shape == (4, 16, 512) and w.shape == (4, 16, 1)
    assert float(w.min()) > 0.0 and float(w.max()) < 1.0 and not y2.requires_grad


def _resume_worker(rank, world, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import diffma_amd.mamba as mamba_mod
    from diffma_amd import sample as sample_mod
    from diffma_amd import train as train_mod
    from diffma_amd.config import Config
    from diffma_amd.model import DiffMa_models

    mamba_mod.spiral_ssm = _oracle_spiral_ssm                     # test-only substitution (see module docstring)
    torch.set_num_threads(2)
    base = dict(model="DiffMa-S/7", image_size=224, dt_rank=16, d_state=16, global_batch_size=2, global_seed=0, lr=1e-4, lr_=1e-4,
                epochs=1, accumulation_steps=1, log_every=1, ckpt_every=1, results_dir=os.path.join(tmpdir, "res"),
                autocast=False, synthetic=True, synthetic_samples=16)
    dist.init_process_group("gloo", rank=0, world_size=1)
    assert train_mod.main(Config(init_from_pretrain_ckpt=False, pretrain_ckpt_path="", init_train_steps=0, max_steps=1, **base)) == 1
    ck = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(tmpdir) for f in fs if f.endswith(".pt"))
    assert len(ck) == 1
    # resume: weights + EMA come from the checkpoint, the step counter from init_train_steps (reference train.py:137-147)
    dist.init_process_group("gloo", rank=0, world_size=1)
    assert train_mod.main(Config(init_from_pretrain_ckpt=True, pretrain_ckpt_path=ck[0], init_train_steps=1, max_steps=2, **base)) == 2
    ck2 = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(tmpdir) for f in fs if f.endswith("0000002.pt"))
    assert len(ck2) == 1
    # what sample.py loads is the EMA of that file, strict
    net = DiffMa_models["DiffMa-S/7"](input_size=28, dt_rank=16, d_state=16)
    net.load_state_dict(sample_mod.find_model(ck2[0]), strict=True)
    first, second = torch.load(ck[0], weights_only=False), torch.load(ck2[0], weights_only=False)
    # (zero-initialised gates keep the mixers' gradients at zero for the first steps: compare a tensor that does move)
    assert not torch.equal(first["model"]["final_layer.linear.weight"], second["model"]["final_layer.linear.weight"])


def test_checkpoint_resume_and_sampler_load(tmp_path):
    """Checkpoint dict {"model","ema","opt","args"} (reference train.py:291-303): written, resumed from, and its EMA
    loaded the way sample.py does."""
    mp.spawn(_resume_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)


def test_kernel_timer_union_of_launch_intervals():
    """bench.py's roofline accounting: overlapping launches of one kernel (two-stream mode) count once in the busy time."""
    from diffma_amd.hip_ops import union_length

    assert union_length([]) == 0.0
    assert union_length([(0.0, 1.0), (2.0, 3.5)]) == 2.5                       # disjoint: the plain sum
    assert union_length([(0.0, 2.0), (1.0, 3.0)]) == 3.0                       # two launches sharing the GPU
    assert union_length([(5.0, 6.0), (0.0, 10.0), (2.0, 3.0)]) == 10.0         # nested, unsorted input
    assert union_length([(0.0, 1.0), (1.0, 2.0)]) == 2.0                       # back to back


def test_bench_reads_the_newest_committed_profiles():
    """bench.py's roofline.traffic / roofline.valu come from profiles/: the NEWEST round's file that holds the kernel must be the
    one cited (the reader silently fell back to round 1 because round 2's file had another schema)."""
    import glob
    import json
    import re

    import bench

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rounds = lambda suffix: sorted(int(re.search(r"r(\d\d)_", os.path.basename(f)).group(1)) for f in glob.glob(os.path.join(root, "profiles", f"r[0-9][0-9]_{suffix}")))
    nbytes, src = bench.load_profile_traffic("dm_selective_scan_bwd", "bf16", 1536)
    newest = max(rounds("traffic.json"))
    assert nbytes and nbytes > 1e9 and src == f"profiles/r{newest:02d}_traffic.json", (nbytes, src)
    # both schemas parse: the `kernels` table (rounds 1 and 3) and round 2's raw counter dump
    for r in rounds("traffic.json"):
        tj = json.load(open(os.path.join(root, "profiles", f"r{r:02d}_traffic.json")))
        assert ("kernels" in tj and "dm_selective_scan_bwd:bf16" in tj["kernels"]) or "bf16" in tj
    if rounds("valu.json"):
        v = bench.load_profile_valu("dm_selective_scan_bwd", "bf16")
        assert v and v["valu_insts_per_wave_step"] > 50 and 0 < v["valu_ceiling_frac_of_8TBps"] < 1 and f"r{max(rounds('valu.json')):02d}_valu.json" in v["source"]


# ---- the data path around the denoiser (NpyDataset + frozen-encoder seam), with deterministic fakes ----------
def _write_slices(root, n, size=224, float_mri=True):
    import numpy as np

    rng = np.random.default_rng(0)
    for sub in ("B", "C", "A"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    for i in range(n):
        name = f"s{i:03d}.npy"
        np.save(os.path.join(root, "B", name), rng.integers(0, 256, (size, size), dtype=np.uint8))               # CT
        np.save(os.path.join(root, "C", name), rng.choice([-1.0, 1.0], (size, size)).astype(np.float32))         # mask in {-1, 1}
        np.save(os.path.join(root, "A", name), (rng.random((size, size)) * (3.0 if float_mri else 1.0)).astype(np.float32))   # MRI (out of [-1, 1])


def test_npy_dataset_and_encoder_seam(tmp_path):
    """NpyDataset (reference load_data.py:14-38: same file name in three folders, mask -> (mask + 1) / 2) and prepare_batch
    (train.py:228-243) with the deterministic stand-in encoders: shapes, scale conventions, the MRI range fix-up."""
    from diffma_amd import data
    from diffma_amd.ct_encoder import CT_Encoder

    _write_slices(str(tmp_path), 5)
    ds = data.NpyDataset(str(tmp_path / "B"), str(tmp_path / "C"), str(tmp_path / "A"), transform=data.transform_test)
    assert len(ds) == 5 and ds.images == sorted(ds.images)
    ct, mask, mri = ds[3]
    assert ct.shape == mask.shape == mri.shape == (1, 224, 224)
    assert 0.0 <= float(ct.min()) and float(ct.max()) <= 1.0                            # uint8 -> [0, 1] like to_tensor
    assert set(torch.unique(mask).tolist()) <= {0.0, 1.0}                               # (mask + 1) / 2
    small = data.transform_test(*[t.numpy()[0] for t in (ct, mask, mri)], size=(112, 112))
    assert all(t.shape == (1, 112, 112) for t in small)                                 # the resize branch
    enc = data.FakeEncoders(seed=0).bundle()
    torch.manual_seed(0)
    ct_enc = CT_Encoder(img_size=28, patch_size=2, in_channels=4, embed_dim=512, contain_mask_token=True).eval()
    items = [ds[i] for i in range(4)]
    z, y, y2, w, ct3, mri3 = data.prepare_batch(torch.stack([it[0] for it in items]), torch.stack([it[2] for it in items]), enc, ct_enc, "cpu")
    assert z.shape == (4, 4, 28, 28) and y.shape == (4, 512) and y2.shape == (4, 196, 512) and w.shape == (4, 196, 1)
    assert ct3.shape == (4, 3, 224, 224) and float(mri3.min()) == -1.0 and float(mri3.max()) == 1.0      # range fix-up of train.py:236-237
    assert float(w.min()) > 0.0 and float(w.max()) < 1.0
    # the stand-in VAE: decode is a right inverse of encode on latents, with the reference's 0.18215 convention
    back = enc.vae_encode(enc.vae_decode(z / data.VAE_SCALE))
    torch.testing.assert_close(back, z, rtol=1e-3, atol=1e-4)
    with pytest.raises(RuntimeError, match="diffusers"):
        data.pretrained_encoders(device="cpu")                                          # offline: says what is missing


def _real_data_worker(rank, world, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import diffma_amd.mamba as mamba_mod
    from diffma_amd import sample as sample_mod
    from diffma_amd import train as train_mod
    from diffma_amd.config import Config

    mamba_mod.spiral_ssm = _oracle_spiral_ssm                     # test-only substitution (see module docstring)
    torch.set_num_threads(2)
    _write_slices(os.path.join(tmpdir, "data"), 4)
    d = os.path.join(tmpdir, "data")
    base = dict(model="DiffMa-S/7", image_size=224, dt_rank=16, d_state=16, global_seed=0, synthetic=False, synthetic_ct_encoder=True,
                encoders="fake", ct_image_folder_train=f"{d}/B", mask_image_folder_train=f"{d}/C", mir_image_folder_train=f"{d}/A",
                ct_image_folder_val=f"{d}/B", mask_image_folder_val=f"{d}/C", mir_image_folder_val=f"{d}/A")
    dist.init_process_group("gloo", rank=0, world_size=1)
    steps = train_mod.main(Config(global_batch_size=2, lr=1e-4, lr_=1e-4, epochs=1, accumulation_steps=1, log_every=1, ckpt_every=2,
                                  results_dir=os.path.join(tmpdir, "res"), init_from_pretrain_ckpt=False, pretrain_ckpt_path="",
                                  init_train_steps=0, autocast=False, max_steps=2, **base))
    assert steps == 2
    ck = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(tmpdir) for f in fs if f.endswith("0000002.pt"))
    out = sample_mod.main(Config(ckpt=ck[0], load_ckpt_type="ema", save_dir=os.path.join(tmpdir, "samples"), seed=0,
                                 sample_global_batch_size=3, sample_num_steps=2, **base))
    # no num_batches: the whole validation shard (4 slices) is sampled, the ragged last batch included (reference drop_last=False)
    assert [tuple(o.shape) for o in out] == [(3, 4, 28, 28), (1, 4, 28, 28)]
    img = torch.load(os.path.join(tmpdir, "samples", "images_rank0.pt"))
    assert img.shape == (4, 3, 224, 224) and torch.isfinite(img).all()


def test_train_and_sample_on_the_real_data_path_with_fake_encoders(tmp_path):
    """train.main / sample.main with `synthetic: false`: .npy slices -> NpyDataset -> (stand-in) VAE / CLIP + CT_Encoder -> denoiser,
    and the sampler's VAE decode at the end (reference train.py:186-243, sample.py:71-110)."""
    mp.spawn(_real_data_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)


def test_step_prep_shadows_survive_two_forwards_and_a_retained_graph():
    """the 16-bit weight copies of step_prep are saved for backward by the projections' autograd nodes.  A second
    prepare() must not write them in place while a graph still holds them: unchanged masters -> no copy at all; changed masters
    with a live graph -> new buffers; changed masters with no live graph -> in place (the steady state of a training loop)."""
    from diffma_amd import step_prep
    from diffma_amd.mamba import Mamba
    from diffma_amd.selective_scan_interface import linear_splitk

    torch.manual_seed(0)
    m = torch.nn.Sequential(Mamba(32, d_state=16))      # prepare() takes the MODEL (its plan holds the mixers, not the model itself)
    x = torch.randn(2, 5, 32)
    W = m[0].in_proj.weight
    with torch.autocast("cpu", dtype=torch.bfloat16):
        step_prep.prepare(m, dtype=torch.bfloat16)
        s0 = step_prep.shadow_of(W, torch.bfloat16)
        assert s0 is not None and s0.dtype == torch.bfloat16
        y1 = linear_splitk(x, W)
        step_prep.prepare(m, dtype=torch.bfloat16)                      # second forward of the same step: nothing is rewritten
        assert step_prep.shadow_of(W, torch.bfloat16) is s0
        y2 = linear_splitk(x, W)
        (y1.float().sum() + y2.float().sum()).backward()                # used to raise "modified by an inplace operation"
        g_two = W.grad.clone()
        W.grad = None
        # a retained graph across a weight update: the old copy must stay what the graph saw
        y3 = linear_splitk(x, W)
        with torch.no_grad():
            W.add_(1.0)
        step_prep.prepare(m, dtype=torch.bfloat16)
        s1 = step_prep.shadow_of(W, torch.bfloat16)
        assert s1 is not None and s1 is not s0                          # new buffers, because y3's graph holds s0
        torch.testing.assert_close(s1.float(), W.detach().to(torch.bfloat16).float())
        y3.float().sum().backward()
        torch.testing.assert_close(W.grad, g_two / 2, rtol=1e-2, atol=1e-2)
        # steady state: no live graph, masters written -> refreshed in place
        del y1, y2, y3
        with torch.no_grad():
            W.mul_(0.5)
        step_prep.prepare(m, dtype=torch.bfloat16)
        assert step_prep.shadow_of(W, torch.bfloat16) is s1
        torch.testing.assert_close(s1.float(), W.detach().to(torch.bfloat16).float())
        # a writer that does not bump `_version` (a replayed hipGraph of the optimizer; here `.data`): invalidate() is what the
        # graphed step calls -- the copies are stale until the next prepare(), which re-casts them
        W.data.mul_(2.0)
        assert step_prep.shadow_of(W, torch.bfloat16) is s1              # the hazard: the version did not move
        step_prep.invalidate()
        assert step_prep.shadow_of(W, torch.bfloat16) is None
        torch.testing.assert_close(step_prep.cast_weight(W, torch.bfloat16).float(), W.detach().to(torch.bfloat16).float())
        step_prep.prepare(m, dtype=torch.bfloat16)
        s2 = step_prep.shadow_of(W, torch.bfloat16)
        assert s2 is not None
        torch.testing.assert_close(s2.float(), W.detach().to(torch.bfloat16).float())
    # a dead master takes its entry (and the shadow's memory) with it
    key = id(W)
    assert key in step_prep._SHADOWS
    del m, W
    import gc
    gc.collect()
    assert key not in step_prep._SHADOWS


def test_staged_backward_equals_one_backward():
    """graphed.StagedBackward: the loss's backward in 4 stages of blocks (last first, `torch.autograd.grad` with the chain / long-skip
    activations and the conditioning vector as cut tensors) must give every parameter the gradient one `backward()` gives it --
    depth 8 (two stages of 4, skips across the cut) and depth 12 (three stages)."""
    import diffma_amd.mamba as mamba_mod
    from diffma_amd.diffusion import create_diffusion
    from diffma_amd.graphed import StagedBackward
    from diffma_amd.model import DiffMa

    real = mamba_mod.spiral_ssm
    mamba_mod.spiral_ssm = _oracle_spiral_ssm                     # test-only substitution (see module docstring)
    try:
        for depth in (8, 12):
            torch.manual_seed(depth)
            net = DiffMa(input_size=8, patch_size=2, strip_size=2, hidden_size=32, depth=depth, d_state=16).train()
            with torch.no_grad():
                for p in net.parameters():
                    if p.requires_grad and float(p.abs().max()) == 0.0:
                        p.copy_(torch.randn_like(p) * 0.05)
            B, T = 2, 16
            z, y, y2 = torch.randn(B, 4, 8, 8), torch.randn(B, 32), torch.randn(B, T, 32)
            w = torch.sigmoid(torch.randn(B, T, 1))
            t = torch.tensor([3, 700])
            nz = torch.randn(B, 4, 8, 8)
            d = create_diffusion("")
            loss = d.training_losses(net, z, t, dict(y=y, y2=y2, w=w), noise=nz)["loss"].mean()
            loss.backward()
            ref = {id(p): p.grad.clone() for p in net.parameters() if p.grad is not None}
            net.zero_grad(set_to_none=True)
            sb = StagedBackward(net, per=4)
            with sb:
                loss = d.training_losses(net, z, t, dict(y=y, y2=y2, w=w), noise=nz)["loss"].mean()
            seen = []
            sb.run(loss, on_stage=lambda k: seen.append(k))
            assert seen == list(range(depth // 4))
            n = 0
            for ps in sb.params:
                for p in ps:
                    torch.testing.assert_close(p.grad, ref[id(p)], rtol=1e-4, atol=1e-6)
                    n += 1
            assert n == len(ref)
            # the hooks are gone: an ordinary forward + backward works again
            net.zero_grad(set_to_none=True)
            d.training_losses(net, z, t, dict(y=y, y2=y2, w=w), noise=nz)["loss"].mean().backward()
            for p in net.parameters():
                if p.requires_grad:
                    torch.testing.assert_close(p.grad, ref[id(p)], rtol=1e-5, atol=1e-7)
    finally:
        mamba_mod.spiral_ssm = real


def test_slab_kernel_broadcasts_from_the_low_dword_only(tmp_path):
    """K4x slab form, DESIGN.md section 3: `float2 * float` must broadcast the LOW dword of the register pair the row-table read
    returns (op_sel_hi).  With the table entry laid out {acc_off, own} hipcc emitted `v_pk_mul_f32 ... op_sel:[1,0]` (low result
    from the HIGH dword) and the kernel miscounted single rows on MI355X now and then.  The built object must not contain that form
    in the slab kernel (a compiler or source change that brings it back fails here, on the CPU, instead of sporadically on a GPU)."""
    import re, shutil, subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    obj = os.path.join(ROOT, "diffma-diffusion-mamba_amd", "csrc", "conv_xproj.o")
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not os.path.isfile(obj) or not all(os.path.isfile(t) for t in tools):
        pytest.skip("needs the built conv_xproj.o and the ROCm llvm tools")
    fat, co = str(tmp_path / "x.fat"), str(tmp_path / "x.co")
    subprocess.run([tools[0], f"--dump-section=.hip_fatbin={fat}", obj], check=True)
    subprocess.run([tools[1], "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
    dis = subprocess.run([tools[2], "-d", co], check=True, capture_output=True, text=True).stdout
    name, seen, bad = None, 0, []
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            name = m.group(1)
            continue
        if name and "conv_xproj_bwd_slab_kernel" in name and re.search(r"v_pk_(mul|fma|add)_f32", line):
            seen += 1
            if re.search(r"op_sel:\[", line):
                bad.append(line.strip())
    assert seen > 1000, "the slab kernel's packed arithmetic was not found in the object"
    assert not bad, f"low-from-high op_sel forms in the slab kernel: {bad[:3]}"


def test_no_matrix_pipe_kernel_takes_a_packed_low_result_from_src1_high(tmp_path):
    """Round 5 bisect of the K4x miscount (profiles/r05_k4x_repro.txt): inside a kernel that also issues MFMAs,
    `v_pk_mul_f32 d, s0, s1 op_sel:[0,1]` -- the LOW result taking the HIGH dword of SRC1 -- gave lanes 48..63 a wrong low result in
    every launch of the soak, whatever stood around it (s_nop 7 before or behind, s_waitcnt lgkmcnt(0), a fresh register copy of the
    pair, an early-clobber destination); the same selection on SRC0, the low-dword broadcast and two plain multiplies are clean.
    Kernels without MFMAs carry hundreds of these forms and pass every test.  So: no object of the library that contains v_mfma may
    contain a packed fp32 instruction whose op_sel sets the src1 bit."""
    import glob, re, subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    objs = sorted(glob.glob(os.path.join(ROOT, "diffma-diffusion-mamba_amd", "csrc", "*.o")))
    if not objs or not all(os.path.isfile(t) for t in tools):
        pytest.skip("needs the built objects and the ROCm llvm tools")
    risky = re.compile(r"v_pk_(mul|add)_f32.*op_sel:\[[01],1\]|v_pk_fma_f32.*op_sel:\[[01],1,[01]\]")
    checked, bad = 0, {}
    for obj in objs:
        fat, co = str(tmp_path / "x.fat"), str(tmp_path / "x.co")
        if subprocess.run([tools[0], f"--dump-section=.hip_fatbin={fat}", obj], capture_output=True).returncode != 0 or not os.path.isfile(fat):
            continue                                                    # host-only object
        subprocess.run([tools[1], "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        dis = subprocess.run([tools[2], "-d", co], check=True, capture_output=True, text=True).stdout
        os.remove(fat)
        name, per = None, {}
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
            if m:
                name = m.group(1)
                per[name] = [0, []]
                continue
            if name is None:
                continue
            if "v_mfma" in line:
                per[name][0] += 1
            elif risky.search(line):
                per[name][1].append(line.strip()[:90])
        for k, (nm, lines) in per.items():
            if nm:
                checked += 1
                if lines:
                    bad[k[:80]] = lines[:2]
    assert checked >= 20, f"only {checked} MFMA kernels found in the objects"
    assert not bad, bad


def test_fused_optimizer_predicate_and_state_layout():
    """optim.supported: only a plain one-group torch.optim.AdamW on fp32 ROCm tensors is taken (anything else keeps torch's own step);
    the argument structs of the C ABI carry the scalars as doubles, as torch keeps them (include/diffma_hip.h)."""
    import ctypes

    from diffma_amd import _lib, optim

    p = [torch.nn.Parameter(torch.randn(8, 4))]
    assert not optim.supported(torch.optim.AdamW(p, lr=1e-4), p)                       # CPU tensors: no product path on the CPU
    assert not optim.supported(torch.optim.SGD(p, lr=1e-4), p)
    assert not optim.supported(torch.optim.AdamW(p, lr=1e-4, amsgrad=True), p)
    fields = dict(_lib.STRUCT_FIELDS["dm_adamw_args"])
    for k in ("lr", "beta1", "beta2", "eps", "weight_decay", "ema_decay"):
        assert fields[k] is ctypes.c_double, k
    assert [f for f, _ in _lib.STRUCT_FIELDS["dm_adamw_tensor"]] == ["p", "m", "v", "g", "ema", "step", "n"]
    assert ctypes.sizeof(_lib.dm_adamw_tensor) == 56                                   # 7 x 8 bytes: the rows FusedAdamWEMA writes as int64


def test_graph_train_setting_true_false_auto():
    """train.graph_train_decision: `graph_train: auto` replays the step from a hipGraph exactly where the reference's own configuration
    sits (config/brain.yaml: one sample per GPU) and leaves large batches, gradient accumulation, fp16 and the CPU alone; booleans and
    the CLI's strings keep their meaning."""
    from diffma_amd.train import GRAPH_AUTO_MAX_BATCH, cli, graph_train_decision as dec

    assert dec("auto", "cuda", 1, 1, False) and dec("auto", "cuda", 1, GRAPH_AUTO_MAX_BATCH, False)
    assert not dec("auto", "cuda", 1, GRAPH_AUTO_MAX_BATCH + 1, False)
    assert not dec("auto", "cuda", 2, 1, False) and not dec("auto", "cuda", 1, 1, True) and not dec("auto", "cpu", 1, 1, False)
    assert dec(True, "cuda", 1, 512, False) and dec("true", "cuda", 1, 512, True) and not dec(True, "cpu", 1, 1, False)
    assert not dec(False, "cuda", 1, 1, False) and not dec("false", "cuda", 1, 1, False) and not dec(True, "cuda", 4, 1, False)
    cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config", "diffma_l2_synthetic.yaml")
    assert cli(["--config", cfg]).get("graph_train", False) in (False, None)          # absent on the command line: the YAML decides
    assert cli(["--config", cfg, "--graph-train"]).graph_train == "true"
    assert cli(["--config", cfg, "--graph-train", "auto"]).graph_train == "auto"


def test_adaln_stack_is_the_blocks_shadows_and_one_product_equals_sixteen():
    """step_prep keeps the blocks' 16-bit adaLN weights / biases as the rows of ONE buffer (adaln_stack) and mamba_block._AdaLNAllFn
    multiplies that buffer once (reference: every block applies its adaLN_modulation to the same c, block/mamba_block.py:82-85, 101).
    On the CPU: the stack's rows ARE the per-weight shadows, the stack-wide product equals the per-block Linear layers (outputs and the
    gradients of SiLU(c), every weight and every bias), a weight update is picked up by the next prepare(), a live autograd graph keeps
    the buffer it saved, and a stale row makes adaln_stack return None."""
    import torch.nn.functional as F

    from diffma_amd import mamba_block, step_prep
    from diffma_amd.model import DiffMa

    torch.manual_seed(2)
    net = DiffMa(input_size=8, patch_size=2, strip_size=2, hidden_size=32, depth=4, d_state=16)
    with torch.no_grad():
        for b in net.blocks:
            b.adaLN_modulation[1].weight.normal_(0, 0.05)
            b.adaLN_modulation[1].bias.normal_(0, 0.05)
    dt = torch.bfloat16
    step_prep.prepare(net, dtype=dt)
    st = step_prep.adaln_stack(net, dt)
    ws, bs = step_prep.adaln_params(net)
    assert st is not None and len(ws) == 4 and st[0].shape == (4,) + tuple(ws[0].shape) and st[1].shape == (4,) + tuple(bs[0].shape)
    for i, (w, b) in enumerate(zip(ws, bs)):
        assert step_prep.shadow_of(w, dt).data_ptr() == st[0][i].data_ptr() and step_prep.shadow_of(b, dt).data_ptr() == st[1][i].data_ptr()
        torch.testing.assert_close(st[0][i].float(), w.detach().to(dt).float())
    sc = torch.randn(3, 64).to(dt).requires_grad_(True)
    outs = mamba_block._AdaLNAllFn.apply(sc, st[0], st[1], *ws, *bs)
    gos = [torch.randn(3, ws[0].shape[0]).to(dt) for _ in range(4)]
    torch.autograd.backward(outs, gos)
    g_sc, g_w, g_b = sc.grad.clone(), [w.grad.clone() for w in ws], [b.grad.clone() for b in bs]
    sc.grad = None
    for p in ws + bs:
        p.grad = None
    ref = [F.linear(sc, w.to(dt), b.to(dt)) for w, b in zip(ws, bs)]
    torch.autograd.backward(ref, gos)
    for a, r in zip(outs, ref):
        torch.testing.assert_close(a.float(), r.float(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(g_sc.float(), sc.grad.float(), rtol=3e-2, atol=3e-2)
    for i in range(4):
        torch.testing.assert_close(g_w[i], ws[i].grad, rtol=2e-2, atol=2e-2)
        torch.testing.assert_close(g_b[i], bs[i].grad, rtol=2e-2, atol=2e-2)
    # a live graph keeps the buffer it saved; the update lands in a new one
    held = mamba_block._AdaLNAllFn.apply(sc, st[0], st[1], *ws, *bs)
    with torch.no_grad():
        ws[1].add_(1.0)
    assert step_prep.adaln_stack(net, dt) is None                       # row 1 is stale until the next prepare()
    step_prep.prepare(net, dtype=dt)
    st2 = step_prep.adaln_stack(net, dt)
    assert st2 is not None and st2[0].data_ptr() != st[0].data_ptr()
    torch.testing.assert_close(st2[0][1].float(), ws[1].detach().to(dt).float())
    torch.testing.assert_close(st[0][1].float(), (ws[1].detach() - 1.0).to(dt).float(), rtol=1e-2, atol=1e-2)     # what `held` saw
    del held


def test_bench_launch_plan_self_launches_for_several_gpus():
    """`python bench.py --gpus N` outside torch.distributed.run must re-execute itself under it (reference train.py:153,190 is
    started by torchrun), and say what is wrong -- not assert -- when the node has fewer GPUs or the launcher disagrees."""
    import bench

    argv = ["--gpus", "8", "--steps", "3", "--warmup", "1"]
    assert bench.launch_plan(1, {}, 0, []) == ("run", None)
    assert bench.launch_plan(1, {}, 8, ["--gpus", "1"]) == ("run", None)
    assert bench.launch_plan(8, {"WORLD_SIZE": "8", "RANK": "3"}, 8, argv) == ("run", None)          # already a rank: the driver's launch
    act, msg = bench.launch_plan(2, {}, 1, ["--gpus", "2"])
    assert act == "error" and "2 GPUs requested, 1 visible" in msg
    act, msg = bench.launch_plan(8, {"WORLD_SIZE": "4"}, 8, argv)
    assert act == "error" and "WORLD_SIZE=4" in msg
    act, cmd = bench.launch_plan(8, {}, 8, argv)
    assert act == "relaunch"
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv                                        # the user's own arguments travel unchanged


def test_bench_without_enough_gpus_fails_with_a_message_not_an_assert():
    """On this GPU-less container `--gpus 2` has to end with exit code 2 and the message on stderr (the first 8-GPU run of the
    driver must produce either a number or a readable reason)."""
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the GPUs")
    assert r.returncode == 2, r.stderr[-500:]
    assert f"2 GPUs requested, {torch.cuda.device_count()} visible" in r.stderr and "AssertionError" not in r.stderr
    assert r.stdout.strip() == ""                                     # no half-written JSON line


def test_bench_leg_specs_cover_the_baseline_configs():
    import bench

    assert set(bench.LEG_SPECS) == {"c2", "c4", "c5", "c3_one_sample_graph"}
    a = bench.parse([])
    for name, spec in bench.LEG_SPECS.items():
        for k in spec:
            assert k == "what" or hasattr(a, k), (name, k)            # every override names a real bench argument
    assert bench.LEG_SPECS["c4"]["use_mamba2"] and bench.LEG_SPECS["c5"]["sampler"] == "ddim50" and bench.LEG_SPECS["c2"]["model"] == "DiffMa-B/4"
