import sys, copy, torch, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import pmf_torch as O, losses_ref
from pmf_amd.utils.detinit import deterministic_init, synthetic_batch
torch.set_num_threads(8)
n, h, w = 2, 64, 1024
ref = deterministic_init(O.PMFNet(5, 3, 20, 32, False, "resnet34")).train()
g = torch.Generator().manual_seed(3)
masks = {nm: (torch.rand(n, c, generator=g) > 0.2).float() / 0.8 for nm, _, c in O.dropout_sites(ref)}
pcd, rgb, label, _ = synthetic_batch(n, h, w, 20, seed=21, fill=0.25)
alpha = torch.linspace(0.2, 1.0, 20); alpha[0] = 0
def run(dt, perturb=0.0):
    m = copy.deepcopy(ref).to(dt)
    O.set_dropout_masks(m, {k: v.to(dt) for k, v in masks.items()})
    if perturb:
        with torch.no_grad():
            wt = m.camera_stream_encoder.conv1.weight
            gg = torch.Generator().manual_seed(9)
            wt.mul_(1 + perturb * (torch.rand(wt.shape, generator=gg, dtype=wt.dtype) - 0.5))
    a, b = m(pcd.to(dt), rgb.to(dt))
    tot, _ = losses_ref.pmf_total_loss(a, b, label, alpha.to(dt))
    tot.backward()
    return {k: p.grad.detach().double().clone() for k, p in m.named_parameters()}
g64 = run(torch.float64)
g32 = run(torch.float32)
g32p = run(torch.float32, 2e-6)
g64p = run(torch.float64, 2e-6)
for k in g64:
    if k.startswith("camera_stream_decoder") or k in ("camera_stream_encoder.conv1.weight", "lidar_stream.logits.weight", "camera_stream_encoder.layer3.2.conv1.weight"):
        d = g64[k].norm().clamp_min(1e-30)
        print("%-45s f32 %.2e  f32+pert %.2e  f64+pert %.2e" % (k, (g32[k]-g64[k]).norm()/d, (g32p[k]-g64[k]).norm()/d, (g64p[k]-g64[k]).norm()/d), flush=True)
