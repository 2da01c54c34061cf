"""developer aid: run train.main (one mode) and then the bf16 Mamba-2 forward of G7 in the same process; report where NaN first appears."""
import os, sys, socket, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
from diffma_amd import train as train_mod
from diffma_amd.config import Config
from diffma_amd.model import DiffMa
gpu = torch.device("cuda", 0)

def m2(label):
    g = np.load(os.path.join(ROOT, "tests/golden/g7_tiny_diffma_mamba2.npz"))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    net = DiffMa(input_size=8, patch_size=2, strip_size=2, hidden_size=64, depth=4, d_state=16, use_mamba2=True)
    net.load_state_dict(sd); net = net.to(gpu).eval()
    inp = {k: torch.from_numpy(g[k]).to(gpu) for k in ("x", "t", "y", "y2", "w")}
    bad = []
    def hook(name):
        def f(m, i, o):
            if torch.is_tensor(o) and not torch.isfinite(o.float()).all():
                bad.append((name, "in_finite=%s" % all(torch.isfinite(t.float()).all().item() for t in i if torch.is_tensor(t))))
        return f
    hs = [m.register_forward_hook(hook(n)) for n, m in net.named_modules()]
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        out = net(inp["x"], inp["t"], y=inp["y"], y2=inp["y2"], w=inp["w"]).float()
    ref = torch.from_numpy(g["out"]).to(gpu)
    print(label, "finite:", bool(torch.isfinite(out).all()), "rel", float((out - ref).norm() / ref.norm()), "first bad:", bad[:4], flush=True)

m2("before")
if mode != "none":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    tmp = tempfile.mkdtemp()
    cfg = Config(model="DiffMa-S/2", image_size=224, dt_rank=16, d_state=16, global_batch_size=4, global_seed=0, lr=1e-4, lr_=1e-4,
                 epochs=1, accumulation_steps=1, log_every=1, ckpt_every=3, results_dir=tmp + "/res",
                 init_from_pretrain_ckpt=False, pretrain_ckpt_path="", init_train_steps=0, synthetic=True, synthetic_samples=64,
                 max_steps=3, autocast=mode != "fp32", amp_dtype="fp16" if mode == "fp16" else "bf16",
                 graph_train=mode == "bf16-graph", grad_compression="bf16" if mode == "bf16-gradcomp" else "none")
    train_mod.main(cfg)
    for i in range(4):
        m2("after train.main[%s] #%d" % (mode, i))
    import torch.cuda.tunable as tn
    res = [r for r in tn.get_results() if ("_290_" in str(r) or "_32_" in str(r))]
    print("tuned entries with 290/32:", len(res))
    for r in res[:12]:
        print("   ", r)
    tn.enable(False)
    m2("tunable off")
