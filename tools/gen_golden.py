"""Generate tests/golden/*.npz by IMPORTING THE REFERENCE in the build container.

Run here only (needs /root/reference; the GPU box never sees it):   python tools/gen_golden.py

What is captured: everything AROUND the third-party operator comes from the
reference's own code; the operator itself (mamba_inner_fn, absent wheel mamba-ssm==2.0.4) is stubbed
with the oracle restatement, so G5 pins the block/model wiring, not the operator arithmetic.
No reference source is copied: the fixtures are inputs and outputs only.
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def _install_stubs():
    from oracle import mamba_ref

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def inner(xz, cw, cb, xw, dw, ow, ob, A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None,
              C_proj_bias=None, delta_softplus=True):
        return mamba_ref.mamba_inner_ref(xz, cw, cb, xw, dw, ow, ob, A, B, C, D, delta_bias=delta_bias,
                                         delta_softplus=delta_softplus, dtype=torch.float64)

    class _Dummy(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    class _RMSNormGated(torch.nn.Module):       # only .weight / .eps are read by the reference (block/mamba2.py:402-403)
        def __init__(self, hidden_size, eps=1e-5, norm_before_gate=False, group_size=None, device=None, dtype=None):
            super().__init__()
            self.eps = eps
            self.weight = torch.nn.Parameter(torch.ones(hidden_size))

    from oracle import mamba2_ref

    def combined(zxbcdt, *a, **k):
        return mamba2_ref.mamba_split_conv1d_scan_combined_ref(zxbcdt, *a, **k, dtype=torch.float64)

    mod("mamba_ssm")
    mod("mamba_ssm.ops")
    mod("mamba_ssm.ops.selective_scan_interface", selective_scan_fn=mamba_ref.selective_scan_ref, mamba_inner_fn=inner)
    mod("mamba_ssm.ops.triton")
    mod("mamba_ssm.ops.triton.selective_state_update", selective_state_update=None)
    mod("mamba_ssm.ops.triton.layernorm", RMSNorm=None, layer_norm_fn=None, rms_norm_fn=None)
    mod("mamba_ssm.ops.triton.layernorm_gated", RMSNorm=_RMSNormGated)
    mod("mamba_ssm.ops.triton.ssd_combined", mamba_chunk_scan_combined=None, mamba_split_conv1d_scan_combined=combined)
    mod("mamba_ssm.distributed")
    mod("mamba_ssm.distributed.tensor_parallel", ColumnParallelLinear=None, RowParallelLinear=None)
    mod("mamba_ssm.distributed.distributed_utils", all_reduce=None, reduce_scatter=None)
    mod("causal_conv1d", causal_conv1d_fn=None, causal_conv1d_update=None)
    mod("timm")
    mod("timm.models")
    mod("timm.models.vision_transformer", Attention=_Dummy, Mlp=_Dummy)
    mod("timm.models.layers", DropPath=_Dummy, to_2tuple=lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v))


def fake_model(x, t, **kw):
    """Deterministic stand-in denoiser with 2*C output channels (learn_sigma=True)."""
    return torch.cat([torch.sin(x) + t.view(-1, 1, 1, 1).float() / 1000.0, torch.cos(x)], dim=1)


def main():
    os.makedirs(OUT, exist_ok=True)
    _install_stubs()
    sys.path.insert(0, REF)
    import tools as ref_tools                      # /root/reference/tools.py
    from diffusion import create_diffusion         # /root/reference/diffusion
    import diffusion.gaussian_diffusion as ref_gd
    import model as ref_model                      # /root/reference/model.py

    # ---- G9 baseline scan orders and blocks: ZigMa / ViM / VMamba / EfficientVMamba ----------------------
    def g9():
        g = {}
        for n in (4, 7, 14):
            for i in range(9):
                a, b = ref_tools.zig(n, i)
                g[f"zig_{n}_{i}.order"] = np.asarray(a, dtype=np.int32)
                g[f"zig_{n}_{i}.inverse"] = np.asarray(b, dtype=np.int32)
            a, b = ref_tools.vmamba_(n)
            g[f"vmamba_{n}.orders"] = np.asarray(a, dtype=np.int32)
            g[f"vmamba_{n}.inverses"] = np.asarray(b, dtype=np.int32)
        # tiny models through the reference classes (operator = oracle stub), one per baseline block type; depth 5 for "zig"
        # (variants 8, 1, 2, 3, 4; all nine tables are pinned above)
        # the "m2." entries are the Mamba-2 twins (use_mamba2=True; the EfficientVMamba twin raises TypeError in the reference)
        for tag, bt, depth, m2 in (("zig", "zig", 5, False), ("vim", "vim", 4, False), ("vmamba", "vmamba", 4, False),
                                   ("efficientVMamba", "efficientVMamba", 4, False), ("m2.zig", "zig", 2, True),
                                   ("m2.vim", "vim", 2, True), ("m2.vmamba", "vmamba", 2, True)):
            torch.manual_seed(3000 + depth)
            net = ref_model.DiffMa(input_size=8, patch_size=2, strip_size=2, hidden_size=64, depth=depth, d_state=16, block_type=bt,
                                   use_mamba2=m2)
            gen = torch.Generator().manual_seed(len(tag))
            with torch.no_grad():
                for name, p in net.named_parameters():
                    if p.requires_grad and float(p.abs().max()) == 0.0:
                        p.copy_(torch.randn(p.shape, generator=gen) * 0.02)
                    if name.endswith("dt_proj.bias"):
                        dt = torch.exp(torch.rand(p.shape, generator=gen) * (np.log(0.1) - np.log(0.001)) + np.log(0.001))
                        p.copy_(dt + torch.log(-torch.expm1(-dt)))
                    if name.endswith("A_log") or name.endswith(".D") or name.endswith("norm.weight"):
                        p.add_(torch.randn(p.shape, generator=gen) * 0.1)
            net.eval()
            bt = tag
            N = 2
            x = torch.randn(N, 4, 8, 8, generator=gen)
            t = torch.tensor([11, 640])
            y = torch.randn(N, 64, generator=gen)
            y2 = torch.randn(N, 16, 64, generator=gen)
            w = torch.sigmoid(torch.randn(N, 16, 1, generator=gen))
            acts = {}
            hooks = [blk.register_forward_hook(lambda m, i, o, k=k: acts.__setitem__(k, o.detach().numpy())) for k, blk in enumerate(net.blocks)]
            with torch.no_grad():
                out = net(x, t, y=y, y2=y2, w=w)
            for h in hooks:
                h.remove()
            g.update({f"{bt}.sd.{k}": v.numpy() for k, v in net.state_dict().items()})
            g.update({f"{bt}.x": x.numpy(), f"{bt}.t": t.numpy(), f"{bt}.y": y.numpy(), f"{bt}.y2": y2.numpy(), f"{bt}.w": w.numpy(),
                      f"{bt}.out": out.numpy(), f"{bt}.depth": np.asarray(depth)})
            g.update({f"{bt}.act.block{k}": v for k, v in acts.items()})
            print("G9", bt, "params", sum(p.numel() for p in net.parameters()), "out abs mean", float(out.abs().mean()))
        np.savez_compressed(os.path.join(OUT, "g9_baseline_blocks.npz"), **g)

    if "--only-g9" in sys.argv:
        g9()
        return
    g9()

    # ---- G1 spiral -------------------------------------------------------------------------------------
    g1 = {}
    for n in (4, 7, 14):
        a, b = ref_tools.spiral(n)
        g1[f"orders_{n}"] = np.asarray(a, dtype=np.int32)
        g1[f"inverses_{n}"] = np.asarray(b, dtype=np.int32)
        print("G1", n, hashlib.sha256(g1[f"orders_{n}"].tobytes()).hexdigest()[:16])
    np.savez_compressed(os.path.join(OUT, "g1_spiral.npz"), **g1)

    # ---- G2 schedule tables -----------------------------------------------------------------------------
    g2 = {}
    for tag, spec in (("full", ""), ("s250", "250"), ("s50", "50"), ("ddim50", "ddim50"), ("s10", "10")):
        d = create_diffusion(spec)
        for name in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next", "sqrt_alphas_cumprod",
                     "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
                     "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                     "posterior_mean_coef1", "posterior_mean_coef2"):
            g2[f"{tag}.{name}"] = np.asarray(getattr(d, name), dtype=np.float64)
        g2[f"{tag}.timestep_map"] = np.asarray(d.timestep_map, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "g2_tables.npz"), **g2)

    # ---- G3 diffusion steps with the fake model -----------------------------------------------------------
    g3 = {}
    gen = torch.Generator().manual_seed(1234)
    x0 = torch.randn(3, 4, 8, 8, generator=gen)
    noise = torch.randn(3, 4, 8, 8, generator=gen)
    step_noise = torch.randn(3, 4, 8, 8, generator=gen)
    g3.update(x0=x0.numpy(), noise=noise.numpy(), step_noise=step_noise.numpy())
    for tag, spec, ts in (("full", "", [0, 500, 999]), ("s250", "250", [0, 17, 249])):
        d = create_diffusion(spec)
        t = torch.tensor(ts)
        g3[f"{tag}.t"] = t.numpy()
        x_t = d.q_sample(x0, t, noise=noise)
        g3[f"{tag}.q_sample"] = x_t.numpy()
        pmv = d.p_mean_variance(fake_model, x_t, t, clip_denoised=False)
        for k in ("mean", "variance", "log_variance", "pred_xstart"):
            g3[f"{tag}.pmv.{k}"] = pmv[k].numpy()
        pmv_c = d.p_mean_variance(fake_model, x_t, t, clip_denoised=True)
        g3[f"{tag}.pmv_clip.mean"] = pmv_c["mean"].numpy()
        vb = d._vb_terms_bpd(fake_model, x0, x_t, t, clip_denoised=False)
        g3[f"{tag}.vb.output"] = vb["output"].numpy()
        tl = d.training_losses(fake_model, x0, t, noise=noise)
        for k, v in tl.items():
            g3[f"{tag}.loss.{k}"] = v.numpy()
        orig = ref_gd.th.randn_like
        ref_gd.th.randn_like = lambda x: step_noise
        try:
            g3[f"{tag}.p_sample"] = d.p_sample(fake_model, x_t, t, clip_denoised=False)["sample"].numpy()
            g3[f"{tag}.ddim_sample_eta0"] = d.ddim_sample(fake_model, x_t, t, clip_denoised=False, eta=0.0)["sample"].numpy()
            g3[f"{tag}.ddim_sample_eta1"] = d.ddim_sample(fake_model, x_t, t, clip_denoised=False, eta=1.0)["sample"].numpy()
        finally:
            ref_gd.th.randn_like = orig
    d10 = create_diffusion("10")
    torch.manual_seed(77)
    g3["loop10.p_sample_loop"] = d10.p_sample_loop(fake_model, (3, 4, 8, 8), noise=x0, clip_denoised=False, device="cpu").numpy()
    torch.manual_seed(77)
    g3["loop10.ddim_sample_loop"] = d10.ddim_sample_loop(fake_model, (3, 4, 8, 8), noise=x0, clip_denoised=False, device="cpu").numpy()
    np.savez_compressed(os.path.join(OUT, "g3_diffusion_steps.npz"), **g3)

    # ---- G4 embeddings ---------------------------------------------------------------------------------------
    pe = ref_model.get_2d_sincos_pos_embed(512, 14).astype(np.float32)
    print("G4 pos_embed", hashlib.sha256(pe.tobytes()).hexdigest()[:16])
    te = ref_model.TimestepEmbed.timestep_embedding(torch.tensor([0, 1, 999]), 256).numpy()
    np.savez_compressed(os.path.join(OUT, "g4_embeddings.npz"), pos_embed_512_14=pe,
                        pos_embed_64_4=ref_model.get_2d_sincos_pos_embed(64, 4).astype(np.float32), timestep_embedding=te)

    # ---- G5 tiny DiffMa through the reference classes (operator = oracle stub) -----------------------------------
    torch.manual_seed(2024)
    net = ref_model.DiffMa(input_size=8, patch_size=2, strip_size=2, hidden_size=64, depth=4, d_state=16)
    gen = torch.Generator().manual_seed(99)
    with torch.no_grad():                    # stock init makes the output exactly 0: re-randomise
        for name, p in net.named_parameters():
            if p.requires_grad and float(p.abs().max()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.02)
            if name.endswith("dt_proj.bias"):
                dt = torch.exp(torch.rand(p.shape, generator=gen) * (np.log(0.1) - np.log(0.001)) + np.log(0.001))
                p.copy_(dt + torch.log(-torch.expm1(-dt)))
            if name.endswith("A_log"):
                p.add_(torch.randn(p.shape, generator=gen) * 0.1)
            if name.endswith(".D"):
                p.add_(torch.randn(p.shape, generator=gen) * 0.1)
    net.eval()
    N = 2
    x = torch.randn(N, 4, 8, 8, generator=gen)
    t = torch.tensor([3, 977])
    y = torch.randn(N, 64, generator=gen)
    y2 = torch.randn(N, 16, 64, generator=gen)
    w = torch.sigmoid(torch.randn(N, 16, 1, generator=gen))
    acts = {}
    hooks = [blk.register_forward_hook(lambda m, i, o, k=k: acts.__setitem__(f"block{k}", o.detach().numpy())) for k, blk in enumerate(net.blocks)]
    with torch.no_grad():
        out = net(x, t, y=y, y2=y2, w=w)
    for h in hooks:
        h.remove()
    g5 = {f"sd.{k}": v.numpy() for k, v in net.state_dict().items()}
    g5.update(x=x.numpy(), t=t.numpy(), y=y.numpy(), y2=y2.numpy(), w=w.numpy(), out=out.numpy())
    g5.update({f"act.{k}": v for k, v in acts.items()})
    # one diffusion training loss on this net (pins training_losses + model together)
    d = create_diffusion("")
    z = torch.randn(N, 4, 8, 8, generator=gen)
    nz = torch.randn(N, 4, 8, 8, generator=gen)
    tt = torch.tensor([10, 900])
    with torch.no_grad():
        tl = d.training_losses(net, z, tt, dict(y=y, y2=y2, w=w), noise=nz)
    g5.update(loss_z=z.numpy(), loss_noise=nz.numpy(), loss_t=tt.numpy(), **{f"loss.{k}": v.numpy() for k, v in tl.items()})
    np.savez_compressed(os.path.join(OUT, "g5_tiny_diffma.npz"), **g5)
    print("G5 params", sum(p.numel() for p in net.parameters()), "out abs mean", float(out.abs().mean()))

    # ---- G7 tiny DiffMa with use_mamba2=True through the reference classes (operator = oracle stub) ---------------------
    torch.manual_seed(2025)
    net2 = ref_model.DiffMa(input_size=8, patch_size=2, strip_size=2, hidden_size=64, depth=4, d_state=16, use_mamba2=True)
    gen = torch.Generator().manual_seed(199)
    with torch.no_grad():
        for name, p in net2.named_parameters():
            if p.requires_grad and float(p.abs().max()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.02)
            if name.endswith("norm.weight") or name.endswith(".D"):
                p.add_(torch.randn(p.shape, generator=gen) * 0.1)
    net2.eval()
    x = torch.randn(N, 4, 8, 8, generator=gen)
    t = torch.tensor([7, 850])
    y = torch.randn(N, 64, generator=gen)
    y2 = torch.randn(N, 16, 64, generator=gen)
    w = torch.sigmoid(torch.randn(N, 16, 1, generator=gen))
    acts = {}
    hooks = [blk.register_forward_hook(lambda m, i, o, k=k: acts.__setitem__(f"block{k}", o.detach().numpy())) for k, blk in enumerate(net2.blocks)]
    with torch.no_grad():
        out = net2(x, t, y=y, y2=y2, w=w)
    for h in hooks:
        h.remove()
    g7 = {f"sd.{k}": v.numpy() for k, v in net2.state_dict().items()}
    g7.update(x=x.numpy(), t=t.numpy(), y=y.numpy(), y2=y2.numpy(), w=w.numpy(), out=out.numpy())
    g7.update({f"act.{k}": v for k, v in acts.items()})
    np.savez_compressed(os.path.join(OUT, "g7_tiny_diffma_mamba2.npz"), **g7)
    print("G7 params", sum(p.numel() for p in net2.parameters()), "out abs mean", float(out.abs().mean()))

    # ---- G8 CT_Encoder (soft mask w + token conditioning y2): the reference module itself, seeded random weights ----------
    from block.CT_encoder import CT_Encoder as RefCT
    g8 = {}
    for tag, (img, patch, emb) in {"p2": (28, 2, 512), "p4": (28, 4, 512), "p7": (28, 7, 64)}.items():
        torch.manual_seed(17)
        ct = RefCT(img_size=img, patch_size=patch, in_channels=4, embed_dim=emb, contain_mask_token=True).eval()
        with torch.no_grad():
            ct.vision_embedding.mask_token.normal_(std=0.02)
            xin = torch.randn(3, 4, img, img)
            wgt, y2o = ct(xin)
        g8.update({f"{tag}.sd.{k}": v.numpy() for k, v in ct.state_dict().items()})
        g8.update({f"{tag}.x": xin.numpy(), f"{tag}.w": wgt.numpy(), f"{tag}.y2": y2o.numpy()})
    np.savez_compressed(os.path.join(OUT, "g8_ct_encoder.npz"), **g8)
    print("G8 CT_Encoder", {k: v.shape for k, v in g8.items() if k.endswith(".w") or k.endswith(".y2")})

    # ---- G6 operator vectors from the ORACLE (regression guard for the restatement itself) ---------------------------
    from oracle import mamba_ref
    gen = torch.Generator().manual_seed(5)
    Bsz, Din, L, Nst, R, dm = 2, 64, 16, 16, 4, 32
    xz = torch.randn(Bsz, 2 * Din, L, generator=gen, dtype=torch.float64)
    P = dict(cw=torch.randn(Din, 1, 4, generator=gen, dtype=torch.float64) * 0.5, cb=torch.randn(Din, generator=gen, dtype=torch.float64) * 0.1,
             xw=torch.randn(R + 2 * Nst, Din, generator=gen, dtype=torch.float64) * 0.2, dw=torch.randn(Din, R, generator=gen, dtype=torch.float64) * 0.5,
             ow=torch.randn(dm, Din, generator=gen, dtype=torch.float64) * 0.2, A=-(torch.rand(Din, Nst, generator=gen, dtype=torch.float64) * 4 + 0.2),
             D=torch.randn(Din, generator=gen, dtype=torch.float64), bias=torch.randn(Din, generator=gen, dtype=torch.float64) * 0.5)
    out = mamba_ref.mamba_inner_ref(xz, P["cw"], P["cb"], P["xw"], P["dw"], P["ow"], None, P["A"], None, None, P["D"],
                                    delta_bias=P["bias"], delta_softplus=True)
    np.savez_compressed(os.path.join(OUT, "g6_oracle_operator.npz"), xz=xz.numpy(), out=out.numpy(), **{k: v.numpy() for k, v in P.items()})
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
