"""bench.py -- DiffMa-L/2 @224x224 (4x28x28 latents, 196 tokens) data-parallel TRAINING step on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is the reference's training step (train.py:243-265): t ~ U{0..999}, q_sample, denoiser forward,
loss = mse + vb, backward (RCCL gradient all-reduce through DDP when N > 1), AdamW, EMA update -- on a
synthetic batch of BASELINE.md section 4 (no datasets / encoders exist offline).  Per-GPU batch is fixed, so
scaling is weak; `value` = samples*steps per second summed over all ranks ("diffusion-steps/sec").

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     : the dominant C-ABI kernel of the timed region, timed with events on the launch stream
  kernels      : the same numbers for every C-ABI kernel
  gemm         : FLOPs and device time of the GEMMs of one step, as a fraction of the dense bf16 MFMA peak
  selective_scan_fn : the scan kernels on their own at the DiffMa-L/2 operator shape (north-star metric, N = 1 only)
  cpu_baseline : the oracle port of the same training step on the host cores (rank 0, N = 1 only)
"""
import argparse
import copy
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
HBM_COPY_GBPS = 6290.0   # MI355X_MICROARCH.md: measured float4 copy ceiling
MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="DiffMa-L/2")
    ap.add_argument("--batch-per-gpu", type=int, default=512)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--mode", default="train", choices=["train", "sample"])
    ap.add_argument("--cpu-steps", type=int, default=5, help="timed training steps of the CPU oracle baseline, median reported (>= 5; 0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip the scan micro-benchmark and the GEMM accounting")
    ap.add_argument("--global-seed", type=int, default=0)
    ap.add_argument("--gemm-tuning", default="file", choices=["file", "frozen", "off", "tune"],
                    help="hipBLASLt/rocBLAS solution selection (diffma_amd.gemm_tuning): file = recorded table, shapes it does not know "
                         "are timed once during the warm-up steps; frozen = recorded table only; off = library defaults; "
                         "tune = like file and the learnt records are written to gpurun_out/ (tools/tune_gemm.sh)")
    ap.add_argument("--torch-profile", default="", help="developer aid: write a torch.profiler op table of one extra step to this file")
    ap.add_argument("--use-mamba2", action="store_true", help="Mamba-2 (SSD) mixers, BASELINE config 4")
    ap.add_argument("--sampler", default="ddpm250", choices=["ddpm250", "ddim50"], help="sample mode: 250-step respaced DDPM (p_sample) or 50-step DDIM")
    ap.add_argument("--route-a", action="store_true", help="the mixers call mamba_inner_fn three times per mixer exactly as the reference's Mamba.forward does (INTEGRATION.md route A) instead of the fused 3-direction operator: what the plain import swap delivers")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph (sample mode: the denoiser call; train mode: the whole optimisation step, with several ranks as two graphs around one gradient all-reduce -- for small batches where the eager step is host-bound)")
    ap.add_argument("--no-config-legs", action="store_true",
                    help="skip the short legs after the headline region (BASELINE configs 2 / 4 / 5 and the one-sample-per-GPU graphed step)")
    ap.add_argument("--legs", default="", help="comma list of the legs to run after the headline region (c2,c4,c5,c3_one_sample_graph); "
                                               "default: all four on one GPU, c3_one_sample_graph only on several")
    ap.add_argument("--leg-steps", type=int, default=10, help="timed steps of each leg")
    return ap.parse_args(argv)


_T0 = time.perf_counter()


def _log(msg):
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def rerandomize_zero_init(model, gen_seed):
    """reference init, then the zero-initialised tensors get N(0, 0.02^2) so that the
    network is not the identity."""
    g = torch.Generator().manual_seed(gen_seed)
    with torch.no_grad():
        for p in model.parameters():
            if p.requires_grad and float(p.abs().max()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)


def synthetic_batch(B, tokens, dev, gen):
    mk = lambda *s: torch.randn(*s, generator=gen, device=dev)
    return dict(z=mk(B, 4, 28, 28), y=mk(B, 512), y2=mk(B, tokens, 512), w=torch.sigmoid(mk(B, tokens, 1)))


@torch.no_grad()
def update_ema(ema, model, decay=0.999):
    ep = list(ema.parameters())
    mp = list(model.parameters())
    torch._foreach_mul_(ep, decay)
    torch._foreach_add_(ep, mp, alpha=1 - decay)


def _cpu_train_steps(model_name, timed, threads):
    """`timed` + 1 training steps (fwd + bwd + AdamW, batch 1, fp32) of the oracle port; the first is an untimed warm-up.
    Returns the per-step wall times."""
    from diffma_amd.diffusion import create_diffusion
    from diffma_amd.model import DiffMa_models
    from oracle.model_ref import diffma_forward_ref

    torch.set_num_threads(threads)
    torch.manual_seed(0)
    net = DiffMa_models[model_name](input_size=28, dt_rank=16, d_state=16, use_mamba2=False)
    rerandomize_zero_init(net, 1)
    tokens = net.x_embedder.num_patches
    sd = {k: v.detach().clone().requires_grad_(k != "pos_embed") for k, v in net.state_dict().items()}
    params = [v for k, v in sd.items() if k != "pos_embed"]
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0)
    d = create_diffusion("")
    gen = torch.Generator().manual_seed(0)
    b = synthetic_batch(1, tokens, "cpu", gen)
    depth, patch = net.depth, net.patch_size
    model = lambda x, t, **kw: diffma_forward_ref(sd, x, t, kw["y"], kw["y2"], kw["w"], patch_size=patch, depth=depth, dtype=torch.float32,
                                                  block_type=net.block_type)
    times = []
    for _ in range(timed + 1):
        t0 = time.perf_counter()
        t = torch.randint(0, d.num_timesteps, (1,))
        loss = d.training_losses(model, b["z"], t, dict(y=b["y"], y2=b["y2"], w=b["w"]))["loss"].mean()
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        times.append(time.perf_counter() - t0)
    return times[1:]


def cpu_baseline(args, tokens):
    """SURVEY.md 8(d) / BASELINE.md section 5: the oracle ("port") of the same training step on the host cores -- batch 1, fp32,
    torch.set_num_threads(os.cpu_count()), median of >= 5 steps after one warm-up step, for the bench model and for
    DiffMa-S/7 (BASELINE config 1)."""
    import statistics

    ncpu, quota = os.cpu_count() or 1, cgroup_cpu_quota()
    threads = max(1, min(ncpu, len(os.sched_getaffinity(0)), quota or ncpu))
    n = max(5, args.cpu_steps)
    s7_t = _cpu_train_steps("DiffMa-S/7", n, threads)
    _log(f"cpu_baseline: DiffMa-S/7 at {threads} threads: median {statistics.median(s7_t):.3f} s/step")
    main_t = _cpu_train_steps(args.model, n, threads)
    _log(f"cpu_baseline: {args.model} at {threads} threads: median {statistics.median(main_t):.2f} s/step")
    med, med7 = statistics.median(main_t), statistics.median(s7_t)
    return {"value": 1.0 / med, "unit": "samples*steps/s", "cores": threads, "kind": "port", "os_cpu_count": ncpu, "cgroup_cpu_quota": quota,
            "sample": f"median of {n} training steps (fwd+bwd+AdamW) after 1 warm-up step, {args.model} at batch 1, fp32, pure-PyTorch "
                      f"oracle (sequential scan), torch.set_num_threads({threads}) = the host cores this container may use "
                      f"(os.cpu_count() {ncpu}, cgroup cpu.max quota {quota}); {sum(main_t):.1f} s of timed CPU work",
            "step_seconds": [round(x, 3) for x in main_t],
            "s7": {"model": "DiffMa-S/7", "value": 1.0 / med7, "unit": "samples*steps/s", "step_seconds": [round(x, 4) for x in s7_t]}}


def cgroup_cpu_quota():
    """CPUs this container may actually use (cgroup v2 cpu.max or v1 cfs quota); None when unlimited.  The GPU boxes show 256
    logical CPUs with a quota of 16: 256 threads on 16 CPUs' worth of time is ~100x slower than 16 threads."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, q // per)
    except (OSError, ValueError):
        pass
    return None


def _profile_files(suffix):
    """profiles/rNN_<suffix> files, newest round first."""
    import glob
    import re
    files = glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{suffix}"))
    return sorted(files, key=lambda f: int(re.search(r"r(\d\d)_", os.path.basename(f)).group(1)), reverse=True)


def load_profile_traffic(kernel, dtype, nseq):
    """HBM bytes per launch of `kernel` (C-ABI name) at this shape from the NEWEST committed traffic profile that holds it
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/prof_r03.sh; cannot be read live).  Two schemas are accepted: the
    `kernels` table (rounds 1, 3: "<kernel>:<dtype>" -> hbm_bytes_per_launch) and round 2's raw per-kernel-name counter dump."""
    for f in _profile_files("traffic.json"):
        try:
            tj = json.load(open(f))
        except (OSError, ValueError):
            continue
        ent = (tj.get("kernels") or {}).get(f"{kernel}:{dtype}")
        if ent and ent.get("nseq") == nseq and ent.get("hbm_bytes_per_launch"):
            return int(ent["hbm_bytes_per_launch"]), f"profiles/{os.path.basename(f)}"
        raw = tj.get(dtype)
        if isinstance(raw, dict) and nseq == 1536:                      # round 2: KiB counters keyed by the mangled kernel name
            want = "scan_bwd_kernel" if kernel.endswith("bwd") else "scan_fwd_kernel"
            for name, c in raw.items():
                if want in name and ", true, true, true" in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:   # the row-index instantiation
                    return int((2 * c["FETCH_SIZE"]["avg"] + c["WRITE_SIZE"]["avg"]) * 1024), f"profiles/{os.path.basename(f)}"
    return None, None


def load_profile_valu(kernel, dtype):
    """The VALU side of the roofline for a VALU-bound kernel, from the newest committed PMC profile (tools/prof_r03.sh):
    executed VALU instructions and busy SIMD-cycles per wave-step, the clock under that instruction mix, and the ceiling they
    imply as a fraction of the 8 TB/s HBM peak (time at 100 % pipe utilisation -> algorithmic bytes / that time)."""
    for f in _profile_files("valu.json"):
        try:
            ent = json.load(open(f)).get(f"{kernel}:{dtype}")
        except (OSError, ValueError):
            continue
        if ent and "valu_insts_per_wave_step" in ent:
            return dict(ent, source=f"profiles/{os.path.basename(f)} (rocprofv3 --pmc SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / GRBM_GUI_ACTIVE passes at nseq {ent.get('nseq')}; not re-measured in this run)")
    return None


def scan_microbench(dev, nseq=768, L=196, Dm=1024, N=16, iters=20):
    """The north-star metric proper: `selective_scan_fn` on its own -- forward without checkpoints and the
    training pair (forward with checkpoints + backward) -- at the DiffMa-L/2 operator shape, fp32 and bf16 I/O, timed with
    events on the launch stream.  GB/s = SURVEY.md 8(d) algorithmic bytes / time."""
    from diffma_amd import hip_ops

    out = {}
    g = torch.Generator(device=dev).manual_seed(0)
    mk = lambda *s: torch.randn(*s, generator=g, device=dev)
    for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        es = 4 if dt == torch.float32 else 2
        u, delta, z, dout = mk(nseq, L, Dm).to(dt), (mk(nseq, L, Dm) * 0.5).to(dt), mk(nseq, L, Dm).to(dt), mk(nseq, L, Dm).to(dt)
        A = -(torch.rand(Dm, N, generator=g, device=dev) * 4 + 0.2)
        Bm, Cm, Dp, bias = mk(nseq, L, N).to(dt), mk(nseq, L, N).to(dt), mk(Dm), mk(Dm) * 0.5
        y = torch.empty_like(u)
        ckpt = hip_ops.alloc_scan_ckpt(nseq, L, N, Dm, dt, dev)

        def timeit(fn):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters * 1e-3

        t_f = timeit(lambda: hip_ops.scan_fwd(u, delta, A, Bm, Cm, Dp, z, bias, True, out=y))
        t_fc = timeit(lambda: hip_ops.scan_fwd(u, delta, A, Bm, Cm, Dp, z, bias, True, out=y, ckpt=ckpt))
        t_b = timeit(lambda: hip_ops.scan_bwd(u, delta, A, Bm, Cm, Dp, z, bias, dout, ckpt, True))     # incl. the partial-row sums
        bf = hip_ops.scan_fwd_algorithmic_bytes(nseq, Dm, L, N, es, es)
        bb = hip_ops.scan_bwd_algorithmic_bytes(nseq, Dm, L, N, es)
        row = lambda t, nb: {"us": round(t * 1e6, 1), "GBps": round(nb / t / 1e9, 1), "frac": round(nb / t / 1e9 / HBM_PEAK_GBPS, 4),
                             "frac_of_measured_copy_ceiling": round(nb / t / 1e9 / HBM_COPY_GBPS, 4), "algorithmic_bytes": nb}
        out[name] = {"fwd": row(t_f, bf), "fwd_with_checkpoints": row(t_fc, bf), "bwd_incl_partial_sums": row(t_b, bb)}
        if dt == torch.bfloat16 and nseq % 3 == 0:
            # the launches the DiffMa mixer issues since round 3: 3 directions through row tables, NO z (the gate is applied once
            # per token by the merge), delta already activated by dm_dtproj_softplus_fwd, the pre-gated gradient shared
            Bd = nseq // 3
            idx = torch.stack([torch.arange(L), torch.randperm(L), torch.randperm(L)]).to(torch.int32).to(dev)
            act = torch.nn.functional.softplus(delta.float() + bias).to(dt)
            kw = dict(z_row_index=idx, out_row_index=idx, batch_per_dir=Bd, delta_activated=True)
            t_mf = timeit(lambda: hip_ops.scan_fwd(u, act, A, Bm, Cm, Dp, None, bias, True, out=y, ckpt=ckpt, **kw))
            t_mb = timeit(lambda: hip_ops.scan_bwd(u, act, A, Bm, Cm, Dp, None, bias, dout[:Bd], ckpt, True, **kw))
            out["bf16_mixer_call_pattern"] = {
                "fwd_with_checkpoints": row(t_mf, hip_ops.scan_fwd_algorithmic_bytes(nseq, Dm, L, N, es, es, has_z=False)),
                "bwd_incl_partial_sums": row(t_mb, hip_ops.scan_bwd_algorithmic_bytes(nseq, Dm, L, N, es, has_z=False)),
                "note": "no z / dz in these launches: 3*s (fwd) and 5*s (bwd) bytes per element instead of 4*s / 7*s"}
        del u, delta, z, dout, y, ckpt
    out["shape"] = {"nseq": nseq, "L": L, "D": Dm, "N": N, "iters": iters}
    return out


def gemm_accounting(step, steps_time_ms):
    """GEMM FLOPs of one step (torch FlopCounterMode: every aten mm / addmm / bmm / convolution actually executed, forward and
    backward) and the device time of the GEMM kernels of one step (torch.profiler kernel records whose name is a Tensile /
    rocBLAS / hipBLASLt GEMM), both taken on extra steps AFTER the timed region."""
    import re

    from torch.profiler import ProfilerActivity, profile
    from torch.utils.flop_counter import FlopCounterMode

    from diffma_amd import hip_ops
    own = hip_ops.KernelTimer()                    # the products on this repository's own MFMA kernels (dm_gemm / dm_gemm_large) are C-ABI
    prev = hip_ops.set_timer(own)                  # launches, invisible to the aten-level FLOP counter: their FLOPs come from the launch log
    with FlopCounterMode(display=False) as fc:
        step()
    hip_ops.set_timer(prev)
    torch.cuda.synchronize()
    own_flops = int(sum(r["flops_per_launch"] * r["launches"] for n, r in own.summary().items() if n.startswith("dm_gemm")))
    flops = int(fc.get_total_flops()) + own_flops
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    pat = re.compile(r"Cijk_|gemm|Gemm|GEMM|gemv")
    gemm_us = other_us = own_us = 0.0
    n_gemm = n_own = 0
    for e in prof.key_averages():
        t = float(getattr(e, "self_device_time_total", 0.0) or 0.0)
        if t <= 0:
            continue
        if pat.search(e.key):
            gemm_us += t
            n_gemm += e.count
            if "dm::gemm" in e.key:                 # K11 gemm_kernel, K12 gemm_large_kernel
                own_us += t
                n_own += e.count
        else:
            other_us += t
    ms = gemm_us * 1e-3
    return {"flops_per_step": flops, "ms_per_step": round(ms, 3), "launches_per_step": n_gemm,
            "own_kernels": {"launches_per_step": n_own, "ms_per_step": round(own_us * 1e-3, 3), "flops_per_step": own_flops,
                            "TFLOPs": round(own_flops / (own_us * 1e-6) / 1e12, 1) if own_us > 0 else None,
                            "what": "dm_gemm (K11, csrc/gemm.hip) and dm_gemm_large (K12, csrc/gemm_large.hip); the rest is hipBLASLt / rocBLAS"},
            "TFLOPs": round(flops / (ms * 1e-3) / 1e12, 1) if ms > 0 else None, "peak_TFLOPs": MFMA_BF16_PEAK_TFLOPS,
            "frac_mfma_peak": round(flops / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4) if ms > 0 else None,
            "share_of_step": round(ms / steps_time_ms, 4), "all_kernels_ms_in_profiled_step": round((gemm_us + other_us) * 1e-3, 3),
            "source": "one step under FlopCounterMode (FLOPs) and one under torch.profiler (GEMM kernel time), both after the timed region"}


def install_route_a():
    """--route-a: the mixers call the operator the way the REFERENCE's Mamba.forward does after the three-line import swap of
    INTEGRATION.md section A (reference block/mamba.py:333-355): channel-major xz (B, 2*Din, L) from in_proj (a permuted VIEW of the
    (2*Din, B*L) product), CrossScan into one (B, 3, 2*Din, L) buffer (two gathers along L; its backward gathers with the inverse
    lists, block/mamba.py:31-57), THREE mamba_inner_fn calls on the buffer's strided slices (each with its own conv / x_proj /
    dt_proj / scan / out_proj), the three outputs copied into one (B, 3, L, d_model) buffer, CrossMerge (row gathers with the
    inverse lists; backward = row gathers with the forward lists, block/mamba.py:59-82).  Everything outside mamba_inner_fn is the
    reference's own torch glue and is restated here as torch code; what the drop-in controls is what happens INSIDE the operator.
    The native path (one fused 3-direction operator, token-major, merge before out_proj) is `Mamba.forward`."""
    from diffma_amd import selective_scan_interface as ssi
    from diffma_amd.mamba import Mamba
    from diffma_amd.selective_scan_interface import mamba_inner_fn

    ssi.PAIR_MIXERS = False            # the paired path is part of the native mixer, not of the reference's call pattern

    class _Scan3(torch.autograd.Function):       # the reference's CrossScan: gathers along the LAST axis of (B, C, L)
        @staticmethod
        def forward(ctx, x, order, order_rev, orig, orig_rev):
            ctx.inv = (orig, orig_rev)
            xs = x.new_empty((x.shape[0], 3) + tuple(x.shape[1:]))
            xs[:, 0] = x
            xs[:, 1] = x[:, :, order]
            xs[:, 2] = x[:, :, order_rev]
            return xs

        @staticmethod
        def backward(ctx, g):
            orig, orig_rev = ctx.inv
            return g[:, 0] + g[:, 1][:, :, orig].contiguous() + g[:, 2][:, :, orig_rev].contiguous(), None, None, None, None

    class _Merge3(torch.autograd.Function):      # the reference's CrossMerge: gathers along the ROW axis of (B, L, C)
        @staticmethod
        def forward(ctx, ys, order, order_rev, orig, orig_rev):
            ctx.fwd = (order, order_rev)
            return ys[:, 0] + ys[:, 1][:, orig, :].contiguous() + ys[:, 2][:, orig_rev, :].contiguous()

        @staticmethod
        def backward(ctx, g):
            order, order_rev = ctx.fwd
            gs = g.new_empty((g.shape[0], 3) + tuple(g.shape[1:]))
            gs[:, 0] = g
            gs[:, 1] = g[:, order, :]
            gs[:, 2] = g[:, order_rev, :]
            return gs, None, None, None, None

    def forward(self, hidden_states, scan_type="spiral", inference_params=None):
        assert scan_type == "spiral" and inference_params is None
        Bsz, L, _ = hidden_states.shape
        idx = getattr(self, "_route_a_idx", None)
        if idx is None or idx[0].device != hidden_states.device:
            mk = lambda l: torch.tensor(list(l), dtype=torch.long, device=hidden_states.device)
            idx = self._route_a_idx = (mk(self.token_list), mk(self.token_list_reversal), mk(self.origina_list), mk(self.origina_list_reversal))
        xz = (self.in_proj.weight @ hidden_states.reshape(Bsz * L, -1).t()).reshape(-1, Bsz, L).permute(1, 0, 2)     # (B, 2*Din, L) view, L contiguous
        A = -torch.exp(self.A_log.float())
        xz_list = _Scan3.apply(xz, *idx)
        outs = [mamba_inner_fn(xz_list[:, k], self.conv1d.weight, self.conv1d.bias, self.x_proj.weight, self.dt_proj.weight, self.out_proj.weight,
                               self.out_proj.bias, A, None, None, self.D.float(), delta_bias=self.dt_proj.bias.float(), delta_softplus=True)
                for k in range(3)]
        out_m = outs[0].new_empty((Bsz, 3, L, outs[0].shape[-1]))
        for k in range(3):
            out_m[:, k] = outs[k]
        return _Merge3.apply(out_m, *idx)

    Mamba.forward = forward


def comm_accounting(step, model, ddp_net, graph_train, rank, world, grad_compression, gstep=None):
    """What the first real multi-GPU run needs to be read: the gradient bytes all-reduced per step, the number
    of collectives they travel in, and -- from ONE profiled step after the timed region, rank 0's own trace -- the device time of
    the RCCL kernels and how much of it was EXPOSED, i.e. not covered by any compute kernel running at the same time.
    Every rank runs the extra step (it contains the collective); only rank 0 profiles."""
    import torch.distributed as dist
    from torch.profiler import ProfilerActivity, profile

    nparam = sum(p.numel() for p in model.parameters() if p.requires_grad)
    es = 2 if (grad_compression in ("bf16", "fp16") and not graph_train) else 4
    info = {"gradient_elements": nparam, "bytes_all_reduced_per_step": nparam * es,
            "wire_dtype": {2: grad_compression, 4: "fp32"}[es], "world": world,
            # ring all-reduce = reduce-scatter + all-gather: every rank sends and receives 2 (N - 1) / N of the payload
            "bytes_on_the_wire_per_rank": int(2 * (world - 1) / max(world, 1) * nparam * es)}
    if graph_train and gstep is not None and getattr(gstep, "staged", None) is not None:
        info["collectives_per_step"] = gstep.staged.nstage
        info["bytes_per_collective"] = [int(f.numel() * 4) for f in gstep.flats]
        info["schedule"] = (f"{gstep.staged.nstage} hipGraphs, one per group of {gstep.staged.per} blocks of the backward (the first also holds the forward); "
                            "each group's all_reduce(AVG) is launched asynchronously right after its graph and runs on RCCL's stream under the next "
                            "group's graph; the last one is exposed; then the AdamW + EMA graph")
    elif graph_train:
        info["collectives_per_step"] = 1
        info["schedule"] = "graph 1 (forward + backward + flatten) | ONE all_reduce(AVG) of the flat fp32 gradient | graph 2 (AdamW + EMA): not overlapped"
    else:
        nb = None
        try:
            ld = ddp_net._get_ddp_logging_data()
            nb = ld.get("num_buckets") or (len(str(ld.get("bucket_sizes", "")).split(",")) if ld.get("bucket_sizes") else None)
            info["bucket_cap_MB"] = ld.get("bucket_cap_bytes", 0) / 2 ** 20 if ld.get("bucket_cap_bytes") else None
        except Exception:
            pass
        info["collectives_per_step"] = int(nb) if nb else None
        info["schedule"] = "DistributedDataParallel buckets (gradient_as_bucket_view, static_graph), all-reduced on RCCL's stream while the backward continues"
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
    else:
        step()
        torch.cuda.synchronize()
    dist.barrier()
    if rank != 0:
        return None
    comm, comp = [], []
    for e in prof.events():
        if e.device_type != torch.autograd.DeviceType.CUDA:
            continue
        iv = (e.time_range.start, e.time_range.end)
        name = e.name.lower()
        (comm if ("nccl" in name or "rccl" in name) else comp).append(iv)
    from diffma_amd.hip_ops import union_length
    busy = union_length(comm)
    # exposed = the part of the collectives' busy time during which no compute kernel was running
    covered = 0.0
    comp_sorted = sorted(comp)
    for s0, e0 in _merge(comm):
        covered += union_length((max(s0, a), min(e0, b)) for a, b in comp_sorted if b > s0 and a < e0)
    span = (max(e for _, e in comm + comp) - min(s for s, _ in comm + comp)) if (comm or comp) else 0.0
    info.update({"rccl_kernel_launches": len(comm), "rccl_kernel_ms": round(sum(e - s for s, e in comm) / 1e3, 3),
                 "rccl_busy_ms": round(busy / 1e3, 3), "rccl_exposed_ms": round((busy - covered) / 1e3, 3),
                 "profiled_step_ms": round(span / 1e3, 3),
                 "note": "one profiled step after the timed region (rank 0's trace): rccl_busy = union of the RCCL kernels' intervals, "
                         "exposed = the part of it with no compute kernel running; on one GPU (BENCH_FORCE_DDP=1) the collective is a local copy"})
    return info


def _merge(intervals):
    out = []
    for s0, e0 in sorted(intervals):
        if out and s0 <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e0)
        else:
            out.append([s0, e0])
    return out


def launch_plan(gpus, environ, visible_gpus, argv):
    """What `python bench.py --gpus N ...` does when it is NOT already a rank of a torch.distributed.run job (reference
    train.py:153,190 is started by torchrun; load_data.py:86 shards by rank).  Pure function (CPU-tested):
      ("run", None)       : N == 1, or the process already is a rank (WORLD_SIZE set): run in this process;
      ("error", message)  : more GPUs requested than visible, or WORLD_SIZE disagrees with --gpus;
      ("relaunch", cmd)   : re-execute under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on 127.0.0.1."""
    ws = environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != gpus:
            return "error", f"bench.py: --gpus {gpus} but WORLD_SIZE={ws}: the launcher's --nproc-per-node and --gpus must agree"
        return "run", None
    if gpus <= 1:
        return "run", None
    if visible_gpus < gpus:
        return "error", f"bench.py: {gpus} GPUs requested, {visible_gpus} visible: run on a node with >= {gpus} MI355X (or lower --gpus)"
    import socket
    with socket.socket() as so:                       # a free port for the rendezvous
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    return "relaunch", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def main():
    args = parse()
    action, what = launch_plan(args.gpus, os.environ, torch.cuda.device_count(), sys.argv[1:])
    if action == "error":
        print(what, file=sys.stderr, flush=True)
        sys.exit(2)
    if action == "relaunch":
        import subprocess
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")      # the host driver only supports dmabuf IPC (RCCL needs it across processes)
        _log("not under torch.distributed.run: re-executing as " + " ".join(what))
        sys.exit(subprocess.run(what, env=env).returncode)           # rank 0 of the children prints the ONE JSON line on the shared stdout
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist

    if not torch.cuda.is_available() or local >= torch.cuda.device_count():
        print(f"bench.py: rank {rank} needs cuda:{local}, {torch.cuda.device_count()} GPUs visible", file=sys.stderr, flush=True)
        sys.exit(2)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    force_ddp = os.environ.get("BENCH_FORCE_DDP") == "1"          # exercise the DDP/RCCL path on a single GPU (testing only)
    if world > 1 or force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)          # RCCL over xGMI

    from diffma_amd import _lib

    _lib.load()                                                  # fail loudly if the HIP library is missing
    if args.gemm_tuning != "off":
        from diffma_amd import gemm_tuning
        wf = None
        if args.gemm_tuning == "tune":
            os.makedirs(os.path.join(os.getcwd(), "gpurun_out"), exist_ok=True)
            wf = os.path.join(os.getcwd(), "gpurun_out", f"gemm_tuning_rank{rank}.csv")
        gemm_tuning.enable_tuned_gemms(tune_missing=args.gemm_tuning != "frozen", write_file=wf)
    if args.route_a:
        install_route_a()
    res = run_workload(args, dev, rank, world, force_ddp, headline=True)
    legs = config_legs(args, dev, rank, world, force_ddp)
    if rank == 0 and legs:
        res["configs"] = legs
        res["roofline"]["configs"] = legs       # the driver's record keeps `roofline` whole and only the NAMES of other extra keys
    finish(res, rank, world, force_ddp)


LEG_SPECS = {
    # BASELINE.json configs[1], [3], [4] and the regime the reference's own brain.yaml runs config [2] in (global batch 8 on 8 GPUs)
    "c2": dict(model="DiffMa-B/4", mode="sample", sampler="ddpm250", batch_per_gpu=64, graph=True, use_mamba2=False,
               what="BASELINE config 2: DiffMa-B/4 250-step respaced DDPM p_sample step (reference sample.py:29-115)"),
    "c4": dict(model="DiffMa-XL/2", mode="train", sampler="ddpm250", batch_per_gpu=256, graph=False, use_mamba2=True,
               what="BASELINE config 4: DiffMa-XL/2 --use-mamba2 training step, bf16 autocast"),
    "c5": dict(model="DiffMa-XXL/2", mode="sample", sampler="ddim50", batch_per_gpu=64, graph=True, use_mamba2=False,
               what="BASELINE config 5: DiffMa-XXL/2 50-step DDIM ddim_sample step"),
    "c3_one_sample_graph": dict(model="DiffMa-L/2", mode="train", sampler="ddpm250", batch_per_gpu=1, graph=True, use_mamba2=False,
                                what="BASELINE config 3 at the reference's own batch (brain.yaml: global batch 8 on 8 GPUs = one sample per GPU): "
                                     "the step replayed from hipGraphs (one graph on one GPU, two around ONE gradient all-reduce on several)"),
}


def config_legs(args, dev, rank, world, force_ddp):
    """Short legs AFTER the headline timed region, so that every BASELINE configuration has a driver-timed number: same timing
    rule as the headline (barrier + synchronize on both sides, max over ranks), `--leg-steps` steps each.  Sampling configs are
    replicas (no collective): one GPU only.  Returns {name: summary} on rank 0, None elsewhere."""
    import gc
    default_run = (args.model == "DiffMa-L/2" and args.mode == "train" and not args.use_mamba2 and not args.route_a and not args.graph
                   and args.dtype == "bf16")
    if args.no_config_legs or not default_run:
        return None
    names = [n for n in args.legs.split(",") if n] or (list(LEG_SPECS) if world == 1 and not force_ddp else ["c3_one_sample_graph"])
    out = {}
    for name in names:
        spec = dict(LEG_SPECS[name])
        what = spec.pop("what")
        la = copy.copy(args)
        for k, v in spec.items():
            setattr(la, k, v)
        la.steps, la.warmup, la.cpu_steps, la.no_extras, la.torch_profile = args.leg_steps, 3, 0, True, ""
        gc.collect()
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        try:
            r = run_workload(la, dev, rank, world, force_ddp, headline=False)
        except Exception as e:                       # a leg must never take the headline line down with it
            if world > 1:
                raise                                # ... but with peers in a collective there is no safe way on
            _log(f"leg {name} FAILED: {type(e).__name__}: {e}")
            out[name] = {"what": what, "error": f"{type(e).__name__}: {str(e)[:300]}"}
            continue
        if rank == 0:
            rf = r["roofline"]
            out[name] = {"what": what, "workload": r["config"]["workload"], "ms_per_step": r["ms_per_step"], "samples_steps_per_s": r["value"],
                         "steps": r["steps"], "warmup": r["warmup"], "n_gpus": world, "global_batch": r["config"]["global_batch"],
                         "dominant_kernel": rf["kernel"], "bound": rf["bound"], "frac": rf["frac"], "achieved": rf["achieved"], "unit": rf["unit"],
                         "avg_us": rf["avg_us"], "leg_wall_s": round(time.perf_counter() - t0, 1)}
            if r.get("comm"):
                out[name]["comm"] = r["comm"]
            _log(f"leg {name}: {r['ms_per_step']:.2f} ms/step ({time.perf_counter() - t0:.0f} s)")
    gc.collect()
    torch.cuda.empty_cache()
    return out if rank == 0 else None


def run_workload(args, dev, rank, world, force_ddp, headline):
    """One workload: build the model, warm up, time `args.steps` steps (barrier + synchronize on both sides, max over ranks) and,
    on rank 0, return the JSON record (None on the other ranks)."""
    import torch.distributed as dist

    from diffma_amd import hip_ops
    from diffma_amd.diffusion import create_diffusion
    from diffma_amd.model import DiffMa_models

    torch.manual_seed(args.global_seed * world + rank)           # reference seed rule (train.py:99)
    model = DiffMa_models[args.model](input_size=28, dt_rank=16, d_state=16, use_mamba2=args.use_mamba2)
    rerandomize_zero_init(model, 1)
    model = model.to(dev)
    tokens = model.x_embedder.num_patches
    diffusion = create_diffusion("")
    B = args.batch_per_gpu
    gen = torch.Generator(device=dev).manual_seed(args.global_seed * world + rank)
    batch = synthetic_batch(B, tokens, dev, gen)
    kw = dict(y=batch["y"], y2=batch["y2"], w=batch["w"])
    amp = torch.bfloat16 if args.dtype == "bf16" else None

    if args.mode == "train":
        ema = copy.deepcopy(model).requires_grad_(False)
        net = model
        graph_train = args.graph
        data_parallel = world > 1 or force_ddp
        if data_parallel and not graph_train:
            from diffma_amd.train import wrap_ddp                        # the SAME wrapper (buckets, static_graph, hooks) as train.py
            net = wrap_ddp(model, dev, grad_compression=os.environ.get("DIFFMA_GRAD_COMPRESSION", "none"))
        elif data_parallel:                                              # graphed step: two graphs around ONE all-reduce (graphed.py), no DDP wrapper
            with torch.no_grad():
                for t_ in list(model.parameters()) + list(model.buffers()):
                    dist.broadcast(t_.data, 0)
                for pe, pm in zip(ema.parameters(), model.parameters()):
                    pe.copy_(pm)
        opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=0, fused=True, capturable=graph_train)
        net.train()
        if graph_train:
            from diffma_amd.graphed import GraphedTrainStep
            gstep = GraphedTrainStep(model, ema, opt, diffusion, batch["z"], torch.zeros(B, device=dev, dtype=torch.long),
                                     kw["y"], kw["y2"], kw["w"], autocast_dtype=amp, ema_decay=0.999, split=True if data_parallel else None)

            def step():
                t = torch.randint(0, diffusion.num_timesteps, (B,), device=dev)
                return gstep.step(batch["z"], t, kw["y"], kw["y2"], kw["w"])
        else:
            def step():
                t = torch.randint(0, diffusion.num_timesteps, (B,), device=dev)
                with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
                    loss = diffusion.training_losses(net, batch["z"], t, kw)["loss"].mean()
                loss.backward()
                opt.step()
                update_ema(ema, model)
                opt.zero_grad(set_to_none=True)
                return loss
    else:
        model.eval()
        sdiff = create_diffusion("250" if args.sampler == "ddpm250" else "ddim50")
        state = {"x": batch["z"].clone(), "i": sdiff.num_timesteps - 1}
        denoiser = model.forward
        if args.graph:
            from diffma_amd.graphed import GraphedDenoiser
            denoiser = GraphedDenoiser(model, state["x"], torch.zeros(B, device=dev, dtype=torch.long), kw["y"], kw["y2"], kw["w"],
                                       autocast_dtype=amp)

        def step():
            t = torch.full((B,), state["i"], device=dev, dtype=torch.long)
            with torch.no_grad(), torch.autocast("cuda", dtype=amp, enabled=amp is not None and not args.graph):
                out = (sdiff.p_sample if args.sampler == "ddpm250" else sdiff.ddim_sample)(denoiser, state["x"], t, clip_denoised=False, model_kwargs=kw)
            state["x"] = out["sample"].float()
            state["i"] = state["i"] - 1 if state["i"] > 0 else sdiff.num_timesteps - 1
            return out["sample"]

    for _ in range(args.warmup):
        step()
    if args.torch_profile and rank == 0:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            step()
            torch.cuda.synchronize()
        with open(args.torch_profile, "w") as f:
            f.write(prof.key_averages().table(sort_by="cuda_time_total", row_limit=70, max_name_column_width=60))
            f.write("\n\n== copies / reductions / elementwise by input shape ==\n")
            rows = [e for e in prof.key_averages(group_by_input_shape=True)
                    if e.key in ("aten::copy_", "aten::sum", "aten::add_", "aten::add", "aten::mul", "aten::fill_", "aten::zero_", "aten::cat")]
            rows.sort(key=lambda e: -e.device_time_total)
            for e in rows[:40]:
                f.write(f"{e.key:14s} n={e.count:4d} dev_us={e.device_time_total:10.1f}  {str(e.input_shapes)[:150]}\n")
    timer = hip_ops.KernelTimer()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    hip_ops.set_timer(timer)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    hip_ops.set_timer(None)
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(last.detach().float()).all(), "non-finite loss/sample in the timed region"
    comm = None
    if (world > 1 or force_ddp) and args.mode == "train":
        comm = comm_accounting(step, model, net, args.graph, rank, world, os.environ.get("DIFFMA_GRAD_COMPRESSION", "none"),
                               gstep if (args.graph and args.mode == "train") else None)

    if rank == 0:
        ksum = timer.summary()
        kernel_source = "events on the launch stream over the timed region"
        if args.mode == "sample" and args.graph:         # hipGraph replay: the denoiser's launches are not visible to the host-side timer
            timer = hip_ops.KernelTimer()                 # (only the fused sampler kernel outside the graph was): time 2 eager steps instead
            hip_ops.set_timer(timer)
            for _ in range(2):
                t_ = torch.full((B,), 5, device=dev, dtype=torch.long)
                with torch.no_grad(), torch.autocast("cuda", dtype=amp, enabled=amp is not None):
                    (sdiff.p_sample if args.sampler == "ddpm250" else sdiff.ddim_sample)(model.forward, state["x"], t_, clip_denoised=False, model_kwargs=kw)
            torch.cuda.synchronize()
            hip_ops.set_timer(None)
            ksum = timer.summary()
            kernel_source = "2 eager sampler steps after the timed region (the timed steps replay a hipGraph)"
        elif not ksum:                                   # graphed training step: time the kernels of 2 eager steps instead
            hip_ops.set_timer(timer)
            for _ in range(2):
                t_ = torch.randint(0, diffusion.num_timesteps, (B,), device=dev)
                with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
                    l_ = diffusion.training_losses(net, batch["z"], t_, kw)["loss"].mean()
                l_.backward()
                opt.zero_grad(set_to_none=True)
            torch.cuda.synchronize()
            hip_ops.set_timer(None)
            ksum = timer.summary()
            kernel_source = "2 eager forward+backward passes after the timed region (the timed steps replay a hipGraph)"
        kernels = {}
        for name, r in ksum.items():
            # rate while at least one launch of the kernel is running (= bytes / avg_us when launches never overlap)
            gbps = r["bytes_per_launch"] * r["launches"] / (r["busy_ms"] * 1e-3) / 1e9
            gbps_d = r["design_bytes_per_launch"] * r["launches"] / (r["busy_ms"] * 1e-3) / 1e9
            kernels[name] = dict(launches_per_step=r["launches"] / args.steps, avg_us=round(r["avg_us"], 2),
                                 ms_per_step=round(r["total_ms"] / args.steps, 3), algorithmic_MB_per_launch=round(r["bytes_per_launch"] / 1e6, 3),
                                 design_MB_per_launch=round(r["design_bytes_per_launch"] / 1e6, 3),
                                 concurrent_launches=round(r["total_ms"] / r["busy_ms"], 3), GBps=round(gbps, 1),
                                 frac_hbm_peak=round(gbps / HBM_PEAK_GBPS, 4), frac_design=round(gbps_d / HBM_PEAK_GBPS, 4))
        nsteps_k = args.steps if "timed region" in kernel_source and "after" not in kernel_source else 2
        for v in kernels.values():
            v["launches_per_step"] = v["launches_per_step"] * args.steps / nsteps_k
            v["ms_per_step"] = round(v["ms_per_step"] * args.steps / nsteps_k, 3)
        dom = max(ksum, key=lambda n: ksum[n]["total_ms"])
        r = ksum[dom]
        per_launch = r["bytes_per_launch"] / (r["avg_us"] * 1e-6) / 1e9
        conc = r["total_ms"] / r["busy_ms"]              # the block's two mixers run on two streams: launches of this kernel overlap
        achieved = per_launch * conc                     # = all bytes of the kernel / time during which it was running
        # HBM traffic of the same kernel at the same shape from the committed PMC passes (cannot be read live)
        traffic = traffic_src = None
        if args.model == "DiffMa-L/2" and not args.use_mamba2:
            traffic, tsrc = load_profile_traffic(dom, args.dtype, 3 * B)
            if traffic is not None:
                traffic_src = f"{tsrc} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes at this shape; not re-measured in this run)"
        valu = load_profile_valu(dom, args.dtype) if (args.model == "DiffMa-L/2" and not args.use_mamba2) else None
        design_per_launch = r["design_bytes_per_launch"] / (r["avg_us"] * 1e-6) / 1e9 * conc
        mfma_roof = None
        if r.get("flops_per_launch", 0) > 0:             # a matrix-pipe kernel dominates (dm_gemm at small batches): its roof is the MFMA peak
            tf = r["flops_per_launch"] / (r["avg_us"] * 1e-6) / 1e12 * conc
            mfma_roof = {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4),
                         "flops_per_launch": int(r["flops_per_launch"]),
                         "note": "launch-bound regime: a few hundred workgroups of a 64 x 64 tile per launch, both mixers of a block in one launch"}
        res = {
            "metric": f"diffusion-steps/sec ({args.model}{' mamba2' if args.use_mamba2 else ''}, 224x224, {'training' if args.mode == 'train' else ('250-step DDPM sampling' if args.sampler == 'ddpm250' else '50-step DDIM sampling')}; samples*steps/s)",
            "value": round(args.steps * B * world / elapsed, 3),
            "unit": "samples*steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 autocast (fp32 master weights, fp32 scan state)" if amp else "f32",
            "data": "synthetic (BASELINE.md section 4), random-init weights with zero-init tensors re-randomised",
            "config": {"workload": f"{args.model} DDP training step, 4x28x28 latents (196 tokens), batch {B}/GPU" + (", whole step replayed from a hipGraph" if (args.graph and world == 1 and not force_ddp) else (", step replayed from hipGraphs around the gradient all-reduce(s)" if args.graph else "")) if args.mode == "train"
                       else f"{args.model} {'p_sample step (250-step respaced DDPM)' if args.sampler == 'ddpm250' else 'ddim_sample step (50-step DDIM)'}, batch {B}/GPU" + (", hipGraph replay" if args.graph else ""),
                       "global_batch": B * world, "seq_len": tokens, "parallelism": f"dp{world}",
                       "optimizer_steps_per_sec": round(args.steps / elapsed, 4), "gemm_tuning": args.gemm_tuning,
                       **({"route": "A: three mamba_inner_fn calls per mixer, reference call pattern (block/mamba.py:346-348)"} if args.route_a else {})},
            "roofline": {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "frac_design": round(design_per_launch / HBM_PEAK_GBPS, 4),
                         "design_bytes_per_launch": int(r["design_bytes_per_launch"]),
                         "bytes_note": "algorithmic = SURVEY.md 8(d) for what the launch computes: the mixer's scans run WITHOUT z since round 3 "
                                       "(gate hoisted into the merge), i.e. 5*s B per (seq, channel, step) element + fp32 dB/dC for the "
                                       "backward and 3*s + B/C rows for the forward (7*s / 4*s with z: the stand-alone lines under "
                                       "selective_scan_fn); design adds what this implementation moves on top (state checkpoints every 4 "
                                       "steps, per-workgroup dB/dC partial rows, dA/dD/dbias partials)",
                         "avg_us": round(r["avg_us"], 2), "algorithmic_bytes_per_launch": int(r["bytes_per_launch"]),
                         "concurrent_launches": round(conc, 3), "achieved_per_launch": round(per_launch, 1),
                         # the round-over-round yardstick: SURVEY.md 8(d)'s figure for the operator WITH z (7*s backward / 4*s forward per
                         # element) over this launch's time -- the z / dz bytes themselves are moved by dm_token_merge / dm_gate_bwd now
                         "frac_8d_with_z": (round((hip_ops.scan_bwd_algorithmic_bytes(3 * B, 1024, tokens, 16, 2, True) if dom.endswith("bwd")
                                                   else hip_ops.scan_fwd_algorithmic_bytes(3 * B, 1024, tokens, 16, 2, 2, True))
                                                  / (r["avg_us"] * 1e-6) / 1e9 * conc / HBM_PEAK_GBPS, 4)
                                            if (dom.startswith("dm_selective_scan") and args.model.startswith("DiffMa-") and amp and not args.use_mamba2) else None),
                         "limiter": "valu" if valu else "hbm",      # `bound` names the roof `peak` belongs to; the scans sit on the VALU pipe
                         "valu": valu,
                         "timing": kernel_source + "; achieved = algorithmic bytes per launch / avg_us x concurrent_launches "
                                   "(concurrent_launches = sum of launch durations / union of launch intervals: 1.0 unless the "
                                   "opt-in two-stream mode DIFFMA_OVERLAP_MIXERS=1 lets launches of the two mixers share the GPU)"},
            "kernels": kernels,
        }
        if mfma_roof is not None:
            res["roofline"].update(mfma_roof, limiter="launch / latency", hbm_view={k: res["roofline"][k] for k in ("achieved", "peak", "unit", "frac")})
        _log(f"timed region done: {1e3 * elapsed / args.steps:.1f} ms/step")
        res["config"]["world_size_seen_by_rccl"] = dist.get_world_size() if dist.is_initialized() else 1
        if comm is not None:
            res["comm"] = comm
        if not args.no_extras:
            if args.mode == "train" and world == 1 and not args.graph:      # extra steps on one rank only: never with peers waiting in an all-reduce
                res["gemm"] = gemm_accounting(step, 1e3 * elapsed / args.steps)
                _log("gemm accounting done")
            if world == 1:
                torch.cuda.empty_cache()
                res["selective_scan_fn"] = scan_microbench(dev)
                _log("scan micro-benchmark done")
            # the north-star metric and the GEMM share ALSO inside `roofline` (the driver's record keeps that object and drops
            # unknown top-level keys).  valu_ceiling_frac: the fraction of 8 TB/s the kernel's own instruction stream allows at 100 %
            # VALU-pipe occupancy and the clock it runs at (counters in profiles/r0x_pmc_scan_kernels.txt)
            ss = res.get("selective_scan_fn")
            if ss:
                pick = lambda d, ceil: {"us": d["us"], "frac": d["frac"], "valu_ceiling_frac": ceil}
                res["roofline"]["selective_scan_fn"] = {
                    "shape": ss["shape"], "target_frac": 0.70,
                    "fp32_fwd": pick(ss["fp32"]["fwd"], 0.59), "bf16_fwd": pick(ss["bf16"]["fwd"], 0.30),
                    "fp32_bwd": pick(ss["fp32"]["bwd_incl_partial_sums"], None), "bf16_bwd": pick(ss["bf16"]["bwd_incl_partial_sums"], 0.19)}
            if res.get("gemm"):
                res["roofline"]["gemm_frac_mfma_peak"] = res["gemm"]["frac_mfma_peak"]
                res["roofline"]["gemm_ms_per_step"] = res["gemm"]["ms_per_step"]
                res["roofline"]["gemm_own_kernels_ms_per_step"] = res["gemm"]["own_kernels"]["ms_per_step"]
        if world == 1 and args.cpu_steps > 0 and args.mode == "train":
            res["cpu_baseline"] = cpu_baseline(args, tokens)
    return res if rank == 0 else None


def finish(res, rank, world, force_ddp):
    import torch.distributed as dist

    # RCCL writes its banner / warnings through C stdio (block-buffered on a pipe, NCCL_DEBUG=VERSION is set on the GPU boxes):
    # every rank flushes that before the group goes away, and rank 0 prints afterwards, so that the JSON line is the LAST line
    # of the merged stdout
    import ctypes

    def flush_all():
        try:
            ctypes.CDLL(None).fflush(None)
        except (OSError, AttributeError):
            pass
        sys.stdout.flush()

    if world > 1 or force_ddp:
        flush_all()
        dist.barrier()
        dist.destroy_process_group()
    flush_all()
    if rank == 0:
        if world > 1:
            time.sleep(1.0)                                      # let the other ranks' exit-time output drain first
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
