"""Generate tests/golden/*.npz by IMPORTING THE REFERENCE in the build container.

Run here only (needs /root/reference; the GPU box never sees it):   python tools/gen_golden.py

What is captured: everything AROUND the third-party operator comes from the
reference's own code; the operator itself (mamba_inner_fn, absent wheel mamba-ssm==2.0.4) is stubbed
with the oracle restatement, so G5 pins the block/model wiring, not the operator arithmetic.
The OPERATOR arithmetic is pinned by G10: the reference's own pure-PyTorch Mamba.step() / Mamba2.step()
recurrences (block/mamba.py:405-448, block/mamba2.py:715-775) run token by token.
No reference source is copied: the fixtures are inputs and outputs only.
"""
import copy
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

    # ---- G10 the operator arithmetic the REFERENCE ITSELF holds: Mamba.step() / Mamba2.step() ---------------------------
    # block/mamba.py:405-448 and block/mamba2.py:715-775 are pure-PyTorch single-token recurrences (conv step, x_proj, dt_proj,
    # softplus, exp(dt*A) state update, C.h, D skip, SiLU(z) gate, out_proj), reached when `causal_conv1d_update` and
    # `selective_state_update` are None -- which is what the stubs above install.  Run token by token from zero states in fp64,
    # the stacked outputs are what mamba_inner_fn / mamba_split_conv1d_scan_combined must produce on the whole sequence: this
    # pins the oracle's OPERATOR arithmetic to reference-held code (no oracle function is involved in producing G10).
    # The BACKWARD pinned to reference-held arithmetic (round-6 addition): step() cannot be back-propagated (in-place copy_ on the
    # states), but it can be differenced.  For every G10 case: a fixed cotangent dy, the scalar f = <out, dy> of the reference's own
    # token-by-token loop, and its fp64 CENTRAL finite differences (eps = 1e-6) along 8 random directions in the joint space of
    # (hidden, every parameter except A_log), entries in {-1, 0, +1} (stored as int8).  A_log goes through `.float()` inside step()
    # (block/mamba.py:431, block/mamba2.py:741), so a 1e-6 step would be rounded away: its two directions use steps that fp32
    # holds exactly -- A_log is first rounded to a multiple of 2^-16 (`fd.A_log`, used for ALL fd values of the case), steps 2^-5 and
    # 2^-6, Richardson-extrapolated ((4 D(h/2) - D(h)) / 3, error O(h^4)); what bounds their resolution is the fp32 exp inside step()
    # (relative rounding 6e-8 of A, amplified by 1 / h: ~1e-5 of the derivative), hence the large steps and the looser bound in the test.  tests/test_golden_cpu.py contracts the oracle's autograd
    # gradient with each direction and compares.
    def fd_pin(m, run, hidden, tag, g, seed):
        fgen = torch.Generator().manual_seed(seed)
        with torch.no_grad():         # `m` is the caller's private deep copy: the G10 tensors proper are never touched
            m.A_log.copy_(torch.round(m.A_log * 65536.0) / 65536.0)
            out0 = run(hidden)
        dy = torch.randn(out0.shape, generator=fgen, dtype=torch.float64)
        params = {k: v for k, v in m.named_parameters() if k != "A_log"}
        tri = lambda shape: (torch.randint(0, 3, tuple(shape), generator=fgen) - 1)
        f = lambda h: float((run(h) * dy).sum())
        g[f"{tag}.fd.A_log"] = m.A_log.detach().numpy().copy()
        g[f"{tag}.fd.dy"] = dy.numpy()
        eps = 1e-6
        for j in range(8):
            dirs = {k: tri(v.shape) for k, v in params.items()}
            dh = tri(hidden.shape)
            vals = []
            for sgn in (+1.0, -1.0):
                with torch.no_grad():
                    for k, v in params.items():
                        v.add_(dirs[k].double(), alpha=sgn * eps)
                    vals.append(f(hidden + sgn * eps * dh.double()))
                    for k, v in params.items():
                        v.add_(dirs[k].double(), alpha=-sgn * eps)
            g[f"{tag}.fd.val{j}"] = np.asarray((vals[0] - vals[1]) / (2 * eps))
            g[f"{tag}.fd.dir{j}.hidden"] = dh.numpy().astype(np.int8)
            for k, d_ in dirs.items():
                g[f"{tag}.fd.dir{j}.{k}"] = d_.numpy().astype(np.int8)
        for j in range(2):
            dA = tri(m.A_log.shape)
            if not bool(dA.any()):
                dA = torch.ones_like(dA)
            D = []
            for h in (2.0 ** -5, 2.0 ** -6):
                vals = []
                for sgn in (+1.0, -1.0):
                    with torch.no_grad():
                        m.A_log.add_(dA.double(), alpha=sgn * h)
                        assert torch.equal(m.A_log.float().double(), m.A_log)      # the step survives step()'s fp32 rounding exactly
                        vals.append(f(hidden))
                        m.A_log.add_(dA.double(), alpha=-sgn * h)
                D.append((vals[0] - vals[1]) / (2 * h))
            g[f"{tag}.fd.valA{j}"] = np.asarray((4.0 * D[1] - D[0]) / 3.0)
            g[f"{tag}.fd.valA{j}.coarse"] = np.asarray(D[0])
            g[f"{tag}.fd.dirA{j}"] = dA.numpy().astype(np.int8)
        print("G10 fd", tag, "f", f(hidden), "vals", [float(g[f"{tag}.fd.val{j}"]) for j in range(3)], "valA", float(g[f"{tag}.fd.valA0"]),
              "(coarse", float(g[f"{tag}.fd.valA0.coarse"]), ")")

    def g10():
        from block.mamba import Mamba as RefMamba
        from block.mamba2 import Mamba2 as RefMamba2
        g = {}
        gen = torch.Generator().manual_seed(1010)
        rnd = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float64)
        for tag, dm, L in (("m1.a", 32, 16), ("m1.b", 64, 49), ("m1.c", 32, 196)):
            torch.manual_seed(10 + L)
            m = RefMamba(d_model=dm, d_state=16, d_conv=4, expand=2).double()
            with torch.no_grad():
                dt = torch.exp(torch.rand(m.d_inner, generator=gen, dtype=torch.float64) * (np.log(0.1) - np.log(0.001)) + np.log(0.001))
                m.dt_proj.bias.copy_(dt + torch.log(-torch.expm1(-dt)))
                m.A_log.add_(rnd(*m.A_log.shape) * 0.2)
                m.A_log.copy_(m.A_log.float().double())        # step() rounds A_log to fp32 (block/mamba.py:431): keep it exact
                m.D.add_(rnd(*m.D.shape) * 0.3)
                m.conv1d.bias.add_(rnd(*m.conv1d.bias.shape) * 0.1)
            Bsz = 2
            hidden = rnd(Bsz, L, dm)
            conv_state = torch.zeros(Bsz, m.d_inner, m.d_conv, dtype=torch.float64)
            ssm_state = torch.zeros(Bsz, m.d_inner, m.d_state, dtype=torch.float64)
            outs = []
            with torch.no_grad():
                for l in range(L):
                    o, conv_state, ssm_state = m.step(hidden[:, l:l + 1], conv_state, ssm_state)
                    outs.append(o)
            out = torch.cat(outs, dim=1)
            g.update({f"{tag}.sd.{k}": v.numpy() for k, v in m.state_dict().items()})
            g.update({f"{tag}.hidden": hidden.numpy(), f"{tag}.out": out.numpy(), f"{tag}.last_state": ssm_state.numpy(),
                      f"{tag}.A": (-torch.exp(m.A_log.detach().float())).numpy()})
            print("G10", tag, "out abs mean", float(out.abs().mean()))

            mfd = copy.deepcopy(m)

            def run1(h, m=mfd, Bsz=Bsz, L=L):
                cs = torch.zeros(Bsz, m.d_inner, m.d_conv, dtype=torch.float64)
                ss = torch.zeros(Bsz, m.d_inner, m.d_state, dtype=torch.float64)
                o_ = []
                with torch.no_grad():
                    for l in range(L):
                        o, cs, ss = m.step(h[:, l:l + 1], cs, ss)
                        o_.append(o)
                return torch.cat(o_, dim=1)
            fd_pin(mfd, run1, hidden, tag, g, 7000 + L + dm)
        # Mamba-2: rmsnorm=False is held by the reference end to end (gate = y * silu(z), block/mamba2.py:758-759); with
        # rmsnorm=True the reference calls the absent wheel's RMSNormGated (block/mamba2.py:771), here given the documented
        # forward of that class for norm_before_gate=False: rmsnorm(y * silu(z)) * weight -- conv, recurrence and D skip are
        # still the reference's own lines.
        import mamba_ssm.ops.triton.layernorm_gated as lg

        def _norm_forward(self, x, z=None):
            x = x * torch.nn.functional.silu(z)
            return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + self.eps) * self.weight
        lg.RMSNorm.forward = _norm_forward
        for tag, dm, L, hd, rms in (("m2.a", 32, 16, 16, False), ("m2.b", 64, 49, 64, False), ("m2.c", 64, 196, 32, True),
                                    ("m2.d", 32, 49, 16, True)):
            torch.manual_seed(20 + L)
            m = RefMamba2(d_model=dm, d_state=16, d_conv=4, expand=2, headdim=hd, rmsnorm=rms).double()
            with torch.no_grad():
                m.A_log.copy_(m.A_log.float().double())
                m.D.add_(rnd(*m.D.shape) * 0.3)
                m.conv1d.bias.add_(rnd(*m.conv1d.bias.shape) * 0.1)
                if rms:
                    m.norm.weight.add_(rnd(*m.norm.weight.shape) * 0.2)
            Bsz = 2
            hidden = rnd(Bsz, L, dm)
            conv_dim = m.d_ssm + 2 * m.ngroups * m.d_state
            conv_state = torch.zeros(Bsz, conv_dim, m.d_conv, dtype=torch.float64)
            ssm_state = torch.zeros(Bsz, m.nheads, m.headdim, m.d_state, dtype=torch.float64)
            outs = []
            with torch.no_grad():
                for l in range(L):
                    o, conv_state, ssm_state = m.step(hidden[:, l:l + 1], conv_state, ssm_state)
                    outs.append(o)
            out = torch.cat(outs, dim=1)
            g.update({f"{tag}.sd.{k}": v.numpy() for k, v in m.state_dict().items()})
            g.update({f"{tag}.hidden": hidden.numpy(), f"{tag}.out": out.numpy(), f"{tag}.headdim": np.asarray(hd),
                      f"{tag}.rmsnorm": np.asarray(int(rms)), f"{tag}.A": (-torch.exp(m.A_log.detach().float())).numpy()})
            print("G10", tag, "out abs mean", float(out.abs().mean()))

            mfd = copy.deepcopy(m)

            def run2(h, m=mfd, Bsz=Bsz, L=L, conv_dim=conv_dim):
                cs = torch.zeros(Bsz, conv_dim, m.d_conv, dtype=torch.float64)
                ss = torch.zeros(Bsz, m.nheads, m.headdim, m.d_state, dtype=torch.float64)
                o_ = []
                with torch.no_grad():
                    for l in range(L):
                        o, cs, ss = m.step(h[:, l:l + 1], cs, ss)
                        o_.append(o)
                return torch.cat(o_, dim=1)
            fd_pin(mfd, run2, hidden, tag, g, 8000 + L + dm + hd)
        np.savez_compressed(os.path.join(OUT, "g10_reference_step.npz"), **g)

    # ---- G8b CT_Encoder with the reference's SHIPPED weights (pretrain_ct_vision_embedder/*.pt, loaded the way train.py:166-168
    #      does: the "ema" entry) on a seeded latent; the 9 tensors travel in the fixture (they are data, 263 KB per file) -------
    def g8b():
        from block.CT_encoder import CT_Encoder as RefCT
        g = {}
        for name in ("brain", "pelvis"):
            ck = torch.load(os.path.join(REF, "pretrain_ct_vision_embedder", f"{name}_patch_size_2.pt"), map_location="cpu", weights_only=False)
            for which in ("ema",):
                sd = ck[which]
                ct = RefCT(img_size=28, patch_size=2, in_channels=4, embed_dim=512, contain_mask_token=True).eval()
                ct.load_state_dict(sd)                                  # strict, like train.py:168
                xin = torch.randn(2, 4, 28, 28, generator=torch.Generator().manual_seed(88)) * 0.18215 * 5
                with torch.no_grad():
                    wgt, y2o = ct(xin)
                tag = f"{name}.{which}"
                g.update({f"{tag}.sd.{k}": v.numpy() for k, v in sd.items()})
                g.update({f"{tag}.x": xin.numpy(), f"{tag}.w": wgt.numpy(), f"{tag}.y2": y2o.numpy()})
                print("G8b", tag, "w range", float(wgt.min()), float(wgt.max()), "y2 abs mean", float(y2o.abs().mean()))
        np.savez_compressed(os.path.join(OUT, "g8b_ct_encoder_pretrained.npz"), **g)

    # ---- G11 one optimisation step of the reference's training loop on the G5 tiny model:
    #      loss = training_losses(...)["loss"].mean(); backward; AdamW(lr 1e-4, wd 0).step(); update_ema(ema, model) with the
    #      reference's own update_ema (train.py:33-43; train.py itself cannot be imported -- torchvision/diffusers/omegaconf are
    #      absent -- so that one function is compiled from its source text here, in the build container, and run; nothing of it is
    #      written to the repo), after the initial update_ema(ema, model, decay=0) copy of train.py:201. ---------------------------
    def g11():
        import ast
        from collections import OrderedDict
        from copy import deepcopy
        src = open(os.path.join(REF, "train.py")).read()
        fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "update_ema"][0]
        ns = {"torch": torch, "OrderedDict": OrderedDict}
        exec(compile(ast.Module(body=[fn], type_ignores=[]), "reference_train_update_ema", "exec"), ns)
        ref_update_ema = ns["update_ema"]
        g5 = np.load(os.path.join(OUT, "g5_tiny_diffma.npz"))
        net = ref_model.DiffMa(input_size=8, patch_size=2, strip_size=2, hidden_size=64, depth=4, d_state=16)
        net.load_state_dict({k[3:]: torch.from_numpy(g5[k]) for k in g5.files if k.startswith("sd.")})
        net.train()
        ema = deepcopy(net)
        for p_ in ema.parameters():
            p_.requires_grad_(False)
        ref_update_ema(ema, net, decay=0)
        opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=0)
        d = create_diffusion("")
        z, nz, tt = (torch.from_numpy(g5[k]) for k in ("loss_z", "loss_noise", "loss_t"))
        kw = {k: torch.from_numpy(g5[k]) for k in ("y", "y2", "w")}
        g = {}
        for step in range(2):
            loss = d.training_losses(net, z, tt, kw, noise=nz)["loss"].mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
            ref_update_ema(ema, net)
            g[f"step{step}.loss"] = np.asarray(float(loss.detach()))
            if step == 0:
                g.update({f"step0.grad.{k}": p_.grad.numpy().copy() for k, p_ in net.named_parameters() if p_.grad is not None and
                          (k.startswith("blocks.1.mamba1.") or not k.startswith("blocks."))})
            sel = lambda k: step == 0 or k.startswith("blocks.1.mamba1.") or not k.startswith("blocks.")     # step 1: a subset
            g.update({f"step{step}.model.{k}": v.detach().numpy().copy() for k, v in net.named_parameters() if v.requires_grad and sel(k)})
            g.update({f"step{step}.ema.{k}": v.detach().numpy().copy() for k, v in ema.named_parameters()
                      if net.get_parameter(k).requires_grad and (k.startswith("blocks.1.mamba1.") or not k.startswith("blocks."))})
            print("G11 step", step, "loss", float(loss.detach()))
        np.savez_compressed(os.path.join(OUT, "g11_train_step.npz"), **g)


    # ---- G12 deep tiny denoisers (hidden 32) through the reference class: spiral lists 8..15, the wrap of the list index at
    # block 8 (model.py:147-150) and the long-skip pairs of odd and > 8 depths (model.py:286-295) -- depth 4 / 5 (G5, G9) reach
    # neither.  Operator = oracle stub, as in G5.
    def g12():
        g = {}
        for depth in (9, 13):
            torch.manual_seed(4000 + depth)
            net = ref_model.DiffMa(input_size=8, patch_size=2, strip_size=2, hidden_size=32, depth=depth, d_state=16)
            gen = torch.Generator().manual_seed(500 + depth)
            with torch.no_grad():
                for name, p in net.named_parameters():
                    if p.requires_grad and float(p.abs().max()) == 0.0:
                        p.copy_(torch.randn(p.shape, generator=gen) * 0.05)
                    if name.endswith("dt_proj.bias"):
                        dt = torch.exp(torch.rand(p.shape, generator=gen) * (np.log(0.1) - np.log(0.001)) + np.log(0.001))
                        p.copy_(dt + torch.log(-torch.expm1(-dt)))
                    if name.endswith("A_log") or name.endswith(".D"):
                        p.add_(torch.randn(p.shape, generator=gen) * 0.1)
            net.eval()
            N = 2
            x = torch.randn(N, 4, 8, 8, generator=gen)
            t = torch.tensor([5, 731])
            y = torch.randn(N, 32, generator=gen)
            y2 = torch.randn(N, 16, 32, generator=gen)
            w = torch.sigmoid(torch.randn(N, 16, 1, generator=gen))
            acts = {}
            hooks = [blk.register_forward_hook(lambda m, i, o, k=k: acts.__setitem__(k, o.detach().numpy())) for k, blk in enumerate(net.blocks)]
            with torch.no_grad():
                out = net(x, t, y=y, y2=y2, w=w)
            for h in hooks:
                h.remove()
            tag = f"d{depth}"
            g.update({f"{tag}.sd.{k}": v.numpy() for k, v in net.state_dict().items()})
            g.update({f"{tag}.x": x.numpy(), f"{tag}.t": t.numpy(), f"{tag}.y": y.numpy(), f"{tag}.y2": y2.numpy(), f"{tag}.w": w.numpy(),
                      f"{tag}.out": out.numpy()})
            g.update({f"{tag}.act.block{k}": v for k, v in acts.items()})
            print("G12 depth", depth, "params", sum(p.numel() for p in net.parameters()), "out abs mean", float(out.abs().mean()))
        np.savez_compressed(os.path.join(OUT, "g12_deep_tiny_diffma.npz"), **g)

    only = [a for a in sys.argv[1:] if a.startswith("--only-")]
    if only:
        for a in only:
            {"--only-g9": g9, "--only-g10": g10, "--only-g8b": g8b, "--only-g11": g11, "--only-g12": g12}[a]()
        return
    g9()
    g10()
    g8b()
    g12()

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
    g11()
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
