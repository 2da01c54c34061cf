"""Launch census of one DiffMa-L/2 training step (run on the GPU box): which ATen ops / kernels make up the ~2 700 launches.
   python tools/count_ops.py [batch]"""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from diffma_amd.model import DiffMa_models  # noqa: E402
from diffma_amd.diffusion import create_diffusion  # noqa: E402
from diffma_amd.gemm_tuning import enable_tuned_gemms  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
enable_tuned_gemms()
torch.manual_seed(0)
net = DiffMa_models["DiffMa-L/2"](input_size=28, dt_rank=16, d_state=16).to(dev).train()
bench.rerandomize_zero_init(net, 1)
opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=0, fused=True)
d = create_diffusion("")
g = torch.Generator(device=dev).manual_seed(0)
batch = bench.synthetic_batch(B, 196, dev, g)
kw = dict(y=batch["y"], y2=batch["y2"], w=batch["w"])


def step():
    t = torch.randint(0, 1000, (B,), device=dev)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = d.training_losses(net, batch["z"], t, kw)["loss"].mean()
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
kern = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        kern[e.name[:70]] += 1
print("kernel launches in one step:", sum(kern.values()))
ops = [e for e in prof.key_averages(group_by_input_shape=True) if e.device_time_total > 0 and e.key.startswith("aten::")]
ops.sort(key=lambda e: -e.count)
print("== aten ops with device time, by call count ==")
for e in ops[:70]:
    print(f"{e.key:28s} n={e.count:4d} dev_us={e.device_time_total:9.1f}  {str(e.input_shapes)[:140]}")
print("== kernels by launch count ==")
for k, c in kern.most_common(45):
    print(f"{c:5d}  {k}")
