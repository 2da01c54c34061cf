"""Which host-side operations launch the small fill / copy / add kernels of one training step (run on the GPU box):
    python tools/fill_sources.py [--batch 512] [--pattern FillFunctor]
One step of the bench's model under torch.profiler (with stacks); every CPU operator that owns a device kernel whose name matches
the pattern is listed with its input shapes and the innermost frames of this repository on its Python stack."""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--pattern", default="FillFunctor")
    args = ap.parse_args()
    from torch.profiler import ProfilerActivity, profile

    import bench
    from diffma_amd import gemm_tuning
    from diffma_amd.diffusion import create_diffusion
    from diffma_amd.model import DiffMa_models

    gemm_tuning.enable_tuned_gemms(tune_missing=False)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = DiffMa_models["DiffMa-L/2"](input_size=28, dt_rank=16, d_state=16)
    bench.rerandomize_zero_init(model, 1)
    model = model.to(dev).train()
    d = create_diffusion("")
    gen = torch.Generator(device=dev).manual_seed(0)
    b = bench.synthetic_batch(args.batch, model.x_embedder.num_patches, dev, gen)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0, fused=True)

    def step():
        t = torch.randint(0, d.num_timesteps, (args.batch,), device=dev)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = d.training_losses(model, b["z"], t, dict(y=b["y"], y2=b["y2"], w=b["w"]))["loss"].mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    hits = collections.Counter()
    for e in prof.events():
        ks = [k for k in getattr(e, "kernels", []) if args.pattern in k.name]
        if not ks:
            continue
        frames = [f for f in (e.stack or []) if "diffma" in f or "bench.py" in f][:3]
        hits[(e.name, str(e.input_shapes)[:80], " <- ".join(f.split("/")[-1][:70] for f in frames))] += len(ks)
    for (name, shapes, where), n in hits.most_common(40):
        print(f"{n:5d}  {name:32s} {shapes:82s} {where}")
    print("total", sum(hits.values()))


if __name__ == "__main__":
    main()
