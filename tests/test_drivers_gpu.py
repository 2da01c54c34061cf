"""GPU runs of the harness rows (a12 sampling loops, a13 train.py:main, a14 sample.py:main) and of the
"next" row f1 (CT_Encoder with the reference's shipped weights feeding the denoiser), each against the CPU oracle or a
golden fixture -- never the HIP path against itself.

Sampling-loop parity: the SAME product loop is run twice with the SAME supplied per-step noise -- on the GPU with the HIP
denoiser, on the CPU with the fp64 oracle denoiser (oracle/model_ref.py, pinned by G5/G10) -- tolerance rel-L2 <= 1e-3 in
fp32 (the diffusion arithmetic itself is pinned by G3 on CPU)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_l2(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


class NoiseTape:
    """Replacement for th.randn_like inside diffusion/gaussian_diffusion.py: call i returns tape[i] (generated once on the
    CPU), on the device / dtype of the argument, so a GPU loop and a CPU loop see identical noise."""

    def __init__(self, shape, n, seed):
        g = torch.Generator().manual_seed(seed)
        self.tape = [torch.randn(*shape, generator=g) for _ in range(n)]
        self.i = 0

    def rewind(self):
        self.i = 0

    def __call__(self, x):
        t = self.tape[self.i].to(device=x.device, dtype=x.dtype)
        self.i += 1
        return t


def _g5(gpu):
    from diffma_amd.model import DiffMa

    g = np.load(os.path.join(G, "g5_tiny_diffma.npz"))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    net = DiffMa(input_size=8, patch_size=2, strip_size=2, hidden_size=64, depth=4, d_state=16)
    net.load_state_dict(sd)
    net = net.to(gpu).eval()
    inp = {k: torch.from_numpy(g[k]) for k in ("x", "t", "y", "y2", "w")}
    return g, sd, net, inp


def _oracle_denoiser(sd, patch, depth, compute_device=None):
    """The fp64 oracle denoiser as a `model(x, t, **kw)` callable.  compute_device: run the oracle's torch code there (ATen's fp64
    kernels on the GPU instead of seconds of host time per call) -- inputs are moved over and the result comes back on x's device, so
    the sampling LOOP around it still runs on the host, through the generic (non-fused) diffusion arithmetic."""
    from oracle.model_ref import diffma_forward_ref

    if compute_device is not None and os.environ.get("DIFFMA_TEST_ORACLE_ON_HOST") != "1":
        sd = {k: v.to(compute_device) for k, v in sd.items()}
    else:
        compute_device = None

    def model(x, t, y=None, y2=None, w=None, **kw):
        mv = (lambda v: v.to(compute_device)) if compute_device is not None else (lambda v: v)
        out = diffma_forward_ref(sd, mv(x.double()), mv(t), mv(y.double()), mv(y2.double()), mv(w.double()), patch_size=patch, depth=depth,
                                 dtype=torch.float64).float()
        return out.to(x.device)
    return model


@pytest.mark.parametrize("sampler", ["ddpm", "ddim", "ddim_eta1"])
def test_sampling_loops_match_oracle_driven_loop(gpu, monkeypatch, sampler):
    """p_sample_loop("10") / ddim_sample_loop (reference gaussian_diffusion.py:419-511, 513-680) on the G5 net: HIP denoiser on
    the GPU vs the fp64 oracle denoiser on the CPU, same start latent, same per-step noise."""
    import diffma_amd.diffusion.gaussian_diffusion as gd
    from diffma_amd.diffusion import create_diffusion

    g, sd, net, inp = _g5(gpu)
    z = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(5))
    tape = NoiseTape(z.shape, 16, seed=6)
    monkeypatch.setattr(gd.th, "randn_like", tape)
    d = create_diffusion("10" if sampler == "ddpm" else "ddim10")
    kw_cpu = dict(y=inp["y"], y2=inp["y2"], w=inp["w"])
    kw_gpu = {k: v.to(gpu) for k, v in kw_cpu.items()}
    extra = {} if sampler == "ddpm" else {"eta": 1.0 if sampler == "ddim_eta1" else 0.0}
    loop = (lambda dd: dd.p_sample_loop) if sampler == "ddpm" else (lambda dd: dd.ddim_sample_loop)
    with torch.no_grad():
        got = loop(d)(net.forward, z.shape, z.to(gpu), clip_denoised=False, model_kwargs=kw_gpu, device=gpu, **extra)
        used = tape.i
        tape.rewind()
        ref = loop(d)(_oracle_denoiser(sd, 2, 4), z.shape, z.clone(), clip_denoised=False, model_kwargs=kw_cpu, device="cpu", **extra)
    assert used == tape.i == 10
    assert torch.isfinite(got).all()
    assert rel_l2(got, ref) <= 1e-3, rel_l2(got, ref)
    assert rel_l2(got, z) > 0.05                                  # the loop did move the latent


def _write_s7_checkpoint(path, seed=0):
    """A DiffMa-S/7 checkpoint in the reference's format ({"model","ema",...}, train.py:291-303) whose zero-initialised tensors
    are re-randomised (the stock init makes the denoiser output exactly 0)."""
    from diffma_amd.model import DiffMa_models

    torch.manual_seed(seed)
    net = DiffMa_models["DiffMa-S/7"](input_size=28, dt_rank=16, d_state=16)
    gen = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if p.requires_grad and float(p.abs().max()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.02)
            if name.endswith("A_log") or name.endswith(".D"):
                p.add_(torch.randn(p.shape, generator=gen) * 0.1)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    torch.save({"model": sd, "ema": sd, "args": {}}, path)
    return sd, net.depth, net.patch_size


@pytest.mark.parametrize("ddim,no_graph", [(False, True), (False, False), (True, True), (True, False)])
def test_sample_main_matches_oracle_driven_loop(gpu, monkeypatch, tmp_path, ddim, no_graph):
    """sample.py:main (reference sample.py:29-115) end to end on the GPU: checkpoint load (EMA entry), conditioning, 10-step
    respaced DDPM or DDIM loop, eager and hipGraph-replayed, two batches -- against the same loop on the CPU with the oracle
    denoiser, same latents / conditioning / per-step noise."""
    import diffma_amd.diffusion.gaussian_diffusion as gd
    from diffma_amd import sample as sample_mod
    from diffma_amd.config import Config
    from diffma_amd.diffusion import create_diffusion

    ck = str(tmp_path / "s7.pt")
    sd, depth, patch = _write_s7_checkpoint(ck)
    n, nb, steps, seed = 2, 2, 10, 3
    tape = NoiseTape((n, 4, 28, 28), nb * steps, seed=11)
    monkeypatch.setattr(gd.th, "randn_like", tape)
    args = Config(model="DiffMa-S/7", image_size=224, dt_rank=16, d_state=16, ckpt=ck, load_ckpt_type="ema", save_dir=str(tmp_path / "out"),
                  seed=seed, sample_global_batch_size=n, sample_num_steps=steps, num_batches=nb, ddim=ddim, no_graph=no_graph, synthetic=True)
    try:
        out = sample_mod.main(args)
    finally:
        torch.set_grad_enabled(True)            # sample.main disables grad globally, like the reference script (sample.py:31)
    assert len(out) == nb and tape.i == nb * steps
    saved = torch.load(os.path.join(args.save_dir, "latents_rank0.pt"))
    assert torch.equal(saved, torch.cat(out))
    # the same draws sample.main made (generator seeded seed*world+rank on the device, z then y, y2, w per batch)
    gen = torch.Generator(device=gpu).manual_seed(seed)
    mk = lambda *s: torch.randn(*s, generator=gen, device=gpu)
    d = create_diffusion(f"ddim{steps}" if ddim else str(steps))
    model = _oracle_denoiser(sd, patch, depth, compute_device=gpu)       # the loop below stays on the host
    tape.rewind()
    for b in range(nb):
        z = mk(n, 4, 28, 28)
        kw = dict(y=mk(n, 512).cpu(), y2=mk(n, 16, 512).cpu(), w=torch.sigmoid(mk(n, 16, 1)).cpu())
        loop = d.ddim_sample_loop if ddim else d.p_sample_loop
        with torch.no_grad():
            ref = loop(model, z.shape, z.cpu(), clip_denoised=False, model_kwargs=kw, device="cpu")
        assert torch.isfinite(out[b]).all()
        assert rel_l2(out[b], ref) <= 1e-3, (b, rel_l2(out[b], ref))


def _dist_env(monkeypatch, port):
    for k, v in dict(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0").items():
        monkeypatch.setenv(k, v)


@pytest.mark.parametrize("mode", ["bf16", "fp32", "fp16", "bf16-graph", "bf16-graph-split", "bf16-gradcomp", "bf16-mamba2"])
def test_train_main_runs_on_gpu(gpu, monkeypatch, tmp_path, mode):
    """train.py:main (reference train.py:90-311) on one GPU through RCCL + DDP: DiffMa-S/2 (196 tokens) on synthetic latents,
    bf16 autocast (default), fp32, the reference's fp16 + GradScaler mode, the graphed step (one graph, and the data-parallel
    form: two graphs around one RCCL all-reduce of the flattened gradients), the opt-in bf16 gradient all-reduce, and
    --use-mamba2 (the SSD core on the matrix pipe: the step must launch dm_ssd_fwd / dm_ssd_bwd and no scan).  Checks: step count, the reference-format checkpoint, finite weights that moved, EMA = its recurrence."""
    import socket

    from diffma_amd import train as train_mod
    from diffma_amd.config import Config
    from diffma_amd.model import DiffMa_models

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    _dist_env(monkeypatch, port)
    if mode == "bf16-graph-split":
        monkeypatch.setenv("DIFFMA_GRAPH_SPLIT", "1")
    m2 = mode == "bf16-mamba2"
    calls = {"ssd_fwd": 0, "ssd_bwd": 0, "scan_fwd": 0, "scan_bwd": 0}
    if m2:
        from diffma_amd import hip_ops
        for name in calls:
            real = getattr(hip_ops, name)
            monkeypatch.setattr(hip_ops, name, (lambda real, name: lambda *a, **k: (calls.__setitem__(name, calls[name] + 1), real(*a, **k))[1])(real, name))
    cfg = Config(model="DiffMa-S/2", image_size=224, dt_rank=16, d_state=16, global_batch_size=4, global_seed=0, lr=1e-4, lr_=1e-4,
                 epochs=1, accumulation_steps=1, log_every=1, ckpt_every=3, results_dir=str(tmp_path / "res"),
                 init_from_pretrain_ckpt=False, pretrain_ckpt_path="", init_train_steps=0, synthetic=True, synthetic_samples=64,
                 max_steps=3, autocast=mode != "fp32", amp_dtype="fp16" if mode == "fp16" else "bf16",
                 graph_train=mode.startswith("bf16-graph"), grad_compression="bf16" if mode == "bf16-gradcomp" else "none", use_mamba2=m2)
    assert train_mod.main(cfg) == 3
    if m2:                                                                 # 4 blocks x 2 mixers x 3 steps, nothing on the scan pair
        assert calls == {"ssd_fwd": 24, "ssd_bwd": 24, "scan_fwd": 0, "scan_bwd": 0}, calls
    ck = [os.path.join(dp, f) for dp, _, fs in os.walk(tmp_path) for f in fs if f.endswith("0000003.pt")]
    assert len(ck) == 1
    sd = torch.load(ck[0], map_location="cpu", weights_only=False)
    assert set(sd) == {"model", "ema", "opt", "args"}
    torch.manual_seed(0)                                                   # the seed rule of train.py:99 at world 1, rank 0
    init = DiffMa_models["DiffMa-S/2"](input_size=28, dt_rank=16, d_state=16, use_mamba2=m2).state_dict()
    k = "final_layer.linear.weight"                                        # zero-initialised: every step moves it by ~lr
    assert all(torch.isfinite(v).all() for v in sd["model"].values())
    moved = (sd["model"][k] - init[k]).abs().max().item()
    assert 0.5e-4 < moved < 4e-4, moved
    # EMA after 3 steps from a decay-0 copy of the init: 0.999^3 w0 + sum_i 0.001 * 0.999^(3-i) w_i; bounded by the model's travel
    e = (sd["ema"][k] - init[k]).abs().max().item()
    assert 0 < e < moved * 0.01, (e, moved)
    net = DiffMa_models["DiffMa-S/2"](input_size=28, dt_rank=16, d_state=16, use_mamba2=m2)
    net.load_state_dict(sd["ema"], strict=True)


def test_ct_encoder_pretrained_feeds_denoiser_on_device(gpu):
    """the CT_Encoder with the reference's shipped weights (G8b) runs on the device and its (w, y2) drive the G5
    denoiser -- HIP output vs the oracle denoiser fed the fixture's (w, y2)."""
    import torch.nn.functional as F

    from diffma_amd.ct_encoder import CT_Encoder

    g8 = np.load(os.path.join(G, "g8b_ct_encoder_pretrained.npz"))
    tag = "brain.ema"
    ct = CT_Encoder(img_size=28, patch_size=2, in_channels=4, embed_dim=512, contain_mask_token=True).eval()
    ct.load_state_dict({k[len(tag) + 4:]: torch.from_numpy(g8[k]) for k in g8.files if k.startswith(tag + ".sd.")}, strict=True)
    ct = ct.to(gpu)
    with torch.no_grad():
        w, y2 = ct(torch.from_numpy(g8[f"{tag}.x"]).to(gpu))
    np.testing.assert_allclose(w.cpu().numpy(), g8[f"{tag}.w"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(y2.cpu().numpy(), g8[f"{tag}.y2"], rtol=1e-3, atol=1e-4)
    # the G5 net has 16 tokens x 64 features: feed it a 16-token, 64-feature window of the encoder outputs
    g, sd, net, inp = _g5(gpu)
    w16, y216 = w[:, :16].contiguous(), F.layer_norm(y2[:, :16, :64], (64,)).contiguous()
    with torch.no_grad():
        out = net(inp["x"].to(gpu), inp["t"].to(gpu), y=inp["y"].to(gpu), y2=y216, w=w16)
    w_ref = torch.from_numpy(g8[f"{tag}.w"])[:, :16]
    y2_ref = F.layer_norm(torch.from_numpy(g8[f"{tag}.y2"])[:, :16, :64].double(), (64,))
    ref = _oracle_denoiser(sd, 2, 4)(inp["x"], inp["t"], y=inp["y"], y2=y2_ref, w=w_ref)
    assert rel_l2(out, ref) <= 1e-3, rel_l2(out, ref)


def test_train_step_adamw_ema_matches_reference_on_device(gpu):
    """G11 on the HIP path (fp32): two steps of training_losses -> AdamW(1e-4, wd 0) -> update_ema from the G5 state; gradients,
    weights and EMA against the reference loop's (operator = fp64 oracle)."""
    import copy

    from diffma_amd.diffusion import create_diffusion
    from diffma_amd.train import update_ema

    g5, sd, net, inp = _g5(gpu)
    g = np.load(os.path.join(G, "g11_train_step.npz"))
    net.train()
    ema = copy.deepcopy(net).requires_grad_(False)
    update_ema(ema, net, decay=0)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=0, fused=True)
    d = create_diffusion("")
    z, nz, tt = (torch.from_numpy(g5[k]).to(gpu) for k in ("loss_z", "loss_noise", "loss_t"))
    kw = {k: inp[k].to(gpu) for k in ("y", "y2", "w")}
    for step in range(2):
        loss = d.training_losses(net, z, tt, kw, noise=nz)["loss"].mean()
        opt.zero_grad()
        loss.backward()
        if step == 0:
            for k in (f[len("step0.grad."):] for f in g.files if f.startswith("step0.grad.")):
                assert rel_l2(net.get_parameter(k).grad, torch.from_numpy(g[f"step0.grad.{k}"])) <= 5e-3, k
        opt.step()
        update_ema(ema, net)
        assert abs(float(loss.detach()) - float(g[f"step{step}.loss"])) <= 2e-3 * abs(float(g[f"step{step}.loss"]))
        for f in g.files:
            if f.startswith(f"step{step}.model."):
                k = f[len(f"step{step}.model."):]
                ref, got, w0 = torch.from_numpy(g[f]), net.get_parameter(k).detach().cpu(), sd[k]
                # Adam's step is -lr * m/(sqrt(v)+eps): elements whose gradient is far above eps moved by a full lr
                gk = f"step0.grad.{k}"
                solid = torch.from_numpy(np.abs(g[gk]) > 1e-4) if (step == 0 and gk in g.files) else (ref - w0).abs() > 0.9e-4 * (step + 1)
                if int(solid.sum()) == 0:
                    continue
                bad = ((got[solid] - ref[solid]).abs() > 2e-5).float().mean().item()
                assert bad <= 0.01, (k, bad)
            elif f.startswith(f"step{step}.ema."):
                k = f[len(f"step{step}.ema."):]
                torch.testing.assert_close(ema.get_parameter(k).detach().cpu(), torch.from_numpy(g[f]), rtol=0, atol=3e-7, msg=lambda m, k=k: f"ema {k}: {m}")


@pytest.fixture(autouse=True)
def _isolate_process_wide_gemm_tuning():
    """train.main / sample.main switch PyTorch's TunableOp on for the whole process (recorded table + online tuning of unseen
    shapes).  Tests that run later in the same process must not inherit online tuning: restore the library defaults."""
    yield
    if torch.cuda.is_available():
        import torch.cuda.tunable as tunable
        tunable.tuning_enable(False)
        tunable.enable(False)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_diffusion_step_matches_golden_and_generic_path(gpu, monkeypatch, dtype):
    """K9 (csrc/diffusion_step.hip): p_sample / ddim_sample (eta 0 and 1, clipped and not, mixed timesteps incl. t = 0) through
    the fused kernel against the reference's own outputs (G3, fake model, supplied noise) and against the generic ATen path."""
    import diffma_amd.diffusion.gaussian_diffusion as gd
    from diffma_amd.diffusion import create_diffusion

    g = np.load(os.path.join(G, "g3_diffusion_steps.npz"))
    fake = lambda x, t, **kw: torch.cat([torch.sin(x) + t.view(-1, 1, 1, 1).float() / 1000.0, torch.cos(x)], dim=1).to(dtype)
    step_noise = torch.from_numpy(g["step_noise"]).to(gpu)
    monkeypatch.setattr(gd.th, "randn_like", lambda x: step_noise.to(x.dtype))
    for tag, spec in (("full", ""), ("s250", "250")):
        d = create_diffusion(spec)
        t = torch.from_numpy(g[f"{tag}.t"]).to(gpu)
        x_t = torch.from_numpy(g[f"{tag}.q_sample"]).to(gpu)
        tol = dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
        for name, call in (("p_sample", lambda dd: dd.p_sample(fake, x_t, t, clip_denoised=False)),
                           ("ddim_sample_eta0", lambda dd: dd.ddim_sample(fake, x_t, t, clip_denoised=False, eta=0.0)),
                           ("ddim_sample_eta1", lambda dd: dd.ddim_sample(fake, x_t, t, clip_denoised=False, eta=1.0))):
            d.fused_step = True
            got = call(d)
            np.testing.assert_allclose(got["sample"].cpu().numpy(), g[f"{tag}.{name}"], err_msg=f"{tag}.{name}", **tol)
            d.fused_step = False
            ref = call(d)
            torch.testing.assert_close(got["sample"], ref["sample"].float(), **tol)
            torch.testing.assert_close(got["pred_xstart"], ref["pred_xstart"].float(), **tol)
        for clip in (True, False):                                      # clipping and a t = 0 element (no noise term)
            t0 = torch.tensor([0, 5, d.num_timesteps - 1], device=gpu)
            d.fused_step = True
            a = d.p_sample(fake, x_t * 3, t0, clip_denoised=clip)
            d.fused_step = False
            b = d.p_sample(fake, x_t * 3, t0, clip_denoised=clip)
            torch.testing.assert_close(a["sample"], b["sample"].float(), **tol)
            torch.testing.assert_close(a["pred_xstart"], b["pred_xstart"].float(), **tol)


@pytest.mark.parametrize("compression", ["none", "bf16", "graph"])
def test_bench_under_torchrun_over_rccl_one_rank(gpu, compression, tmp_path):
    """The driver's multi-GPU launch contract on the one GPU a test box has: `python -m torch.distributed.run --nproc-per-node 1
    bench.py --gpus 1 ...` with the DDP / RCCL path forced on (BENCH_FORCE_DDP=1: process group over nccl = RCCL, the shared
    `train.wrap_ddp` settings, optionally 16-bit gradient buckets; "graph": `--graph`, the step replayed from two hipGraphs around one
    all-reduce instead of the DDP wrapper), barrier-bracketed timing, one JSON line on rank 0."""
    import json
    import subprocess
    import sys

    import socket

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]                    # a free port per case: a fixed one can still be in TIME_WAIT from the previous case
    sk.close()
    graph = compression == "graph"
    env = dict(os.environ, BENCH_FORCE_DDP="1", DIFFMA_GRAD_COMPRESSION="none" if graph else compression, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch-per-gpu", "4",
           "--cpu-steps", "0", "--no-extras"] + (["--graph"] if graph else [])
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["config"]["world_size_seen_by_rccl"] == 1 and d["config"]["parallelism"].startswith("dp1")
    assert d["value"] > 0 and d["ms_per_step"] > 0 and 0 < d["roofline"]["frac"] < 1
    assert d["roofline"]["kernel"].startswith("dm_")        # (at this test's 4 samples the projections' dm_gemm can outweigh the scans)
    assert ("hipGraphs around the gradient all-reduce" in d["config"]["workload"]) == graph
    assert d["comm"]["world"] == 1 and d["comm"]["bytes_all_reduced_per_step"] > 0
    if not graph:
        # a data-parallel run of the default model also times the reference's own regime as a leg of the SAME record: one sample per GPU,
        # two hipGraphs around one gradient all-reduce, with its own comm block (what an 8-GPU run of the driver will report)
        leg = d["configs"]["c3_one_sample_graph"]
        assert leg["ms_per_step"] > 0 and leg["global_batch"] == 1 and "hipGraphs around the gradient all-reduce" in leg["workload"]
        assert leg["comm"]["collectives_per_step"] == 1 and d["roofline"]["configs"]["c3_one_sample_graph"]["ms_per_step"] == leg["ms_per_step"]
    else:
        assert "configs" not in d


def test_graphed_train_step_two_graphs_equal_one_graph(gpu, monkeypatch):
    """graphed.GraphedTrainStep in its data-parallel form (graph 1: forward + backward + flattened gradients | eager
    all_reduce(AVG) over RCCL | graph 2: AdamW + EMA) on a one-rank process group must reproduce the single-graph step bit for
    bit: same loss sequence, same weights, same EMA after 6 steps."""
    import copy
    import socket

    import torch.distributed as dist

    from diffma_amd.diffusion import create_diffusion
    from diffma_amd.graphed import GraphedTrainStep
    from diffma_amd.model import DiffMa

    if not dist.is_initialized():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        _dist_env(monkeypatch, port)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu)
    torch.manual_seed(21)
    net0 = DiffMa(input_size=8, patch_size=2, strip_size=2, hidden_size=64, depth=2, d_state=16).to(gpu)
    with torch.no_grad():
        for p in net0.parameters():
            if float(p.abs().max()) == 0.0:
                p.normal_(0, 0.02)
    B = 4
    mk = lambda *sh: torch.randn(*sh, device=gpu)
    x, y, y2, w = mk(B, 4, 8, 8), mk(B, 64), mk(B, 16, 64), torch.sigmoid(mk(B, 16, 1))
    d = create_diffusion("")

    def run(split):
        torch.manual_seed(11)
        net = copy.deepcopy(net0).train()
        ema = copy.deepcopy(net).requires_grad_(False)
        opt = torch.optim.AdamW(net.parameters(), lr=2e-3, weight_decay=0, fused=True, capturable=True)
        gs = GraphedTrainStep(net, ema, opt, d, x, torch.zeros(B, device=gpu, dtype=torch.long), y, y2, w, ema_decay=0.9, warmup=2, split=split)
        assert gs.split == split
        tg = torch.Generator(device=gpu).manual_seed(5)
        losses = [float(gs.step(x, torch.randint(0, d.num_timesteps, (B,), device=gpu, generator=tg), y, y2, w)) for _ in range(6)]
        return losses, [p.detach().clone() for p in net.parameters()], [p.detach().clone() for p in ema.parameters()]

    l1, p1, e1 = run(False)
    l2, p2, e2 = run(True)
    assert all(l == l for l in l1) and l1 == l2, (l1, l2)
    for a, b in zip(p1 + e1, p2 + e2):
        torch.testing.assert_close(a, b, rtol=0, atol=0)


def test_graphed_train_step_in_stages_matches_one_all_reduce(gpu, monkeypatch):
    """The data-parallel graphed step with the backward cut into block groups (graphed.StagedBackward: one hipGraph per group, each
    group's all-reduce launched asynchronously between replays) against the two-graph form with ONE all-reduce, on a one-rank
    process group: same losses, weights and EMA to rounding (the contributions to the conditioning vector's gradient are added in
    another order) after 6 steps, fp32 and bf16 autocast."""
    import copy
    import socket

    import torch.distributed as dist

    from diffma_amd.diffusion import create_diffusion
    from diffma_amd.graphed import GraphedTrainStep
    from diffma_amd.model import DiffMa

    if not dist.is_initialized():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        _dist_env(monkeypatch, port)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu)
    torch.manual_seed(23)
    net0 = DiffMa(input_size=8, patch_size=2, strip_size=2, hidden_size=64, depth=8, d_state=16).to(gpu)
    with torch.no_grad():
        for p in net0.parameters():
            if float(p.abs().max()) == 0.0:
                p.normal_(0, 0.02)
    B = 4
    mk = lambda *sh: torch.randn(*sh, device=gpu)
    x, y, y2, w = mk(B, 4, 8, 8), mk(B, 64), mk(B, 16, 64), torch.sigmoid(mk(B, 16, 1))
    d = create_diffusion("")

    for amp, tol in ((None, 1e-4), (torch.bfloat16, 3e-2)):
        def run(stages):
            torch.manual_seed(11)
            net = copy.deepcopy(net0).train()
            ema = copy.deepcopy(net).requires_grad_(False)
            opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=0, fused=True, capturable=True)
            gs = GraphedTrainStep(net, ema, opt, d, x, torch.zeros(B, device=gpu, dtype=torch.long), y, y2, w, autocast_dtype=amp,
                                  ema_decay=0.9, warmup=2, split=True, stages=stages)
            assert (gs.staged is not None) == bool(stages)
            if stages:
                assert gs.staged.nstage == 2 and len(gs.stage_graphs) == 2
            tg = torch.Generator(device=gpu).manual_seed(5)
            losses = [float(gs.step(x, torch.randint(0, d.num_timesteps, (B,), device=gpu, generator=tg), y, y2, w)) for _ in range(6)]
            return losses, [p.detach().clone() for p in net.parameters()], [p.detach().clone() for p in ema.parameters()]

        l1, p1, e1 = run(0)
        l2, p2, e2 = run(4)
        assert all(l == l for l in l1 + l2)
        for a, b in zip(l1, l2):
            assert abs(a - b) <= tol * max(1.0, abs(a)), (l1, l2)
        num = sum(float((a - b).float().norm() ** 2) for a, b in zip(p1 + e1, p2 + e2)) ** 0.5
        den = sum(float(a.float().norm() ** 2) for a in p1 + e1) ** 0.5
        assert num / den <= tol, num / den
