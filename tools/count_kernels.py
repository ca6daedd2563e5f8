"""Kernel-launch census of one step-1 / step-3 style iteration by region (torch.profiler)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from torch.profiler import profile, ProfilerActivity, record_function
import bench, gan2shape_amd
from gan2shape_amd.model import GAN2Shape
from gan2shape_amd.trainer import Trainer
dev = torch.device("cuda:0")
torch.manual_seed(0)
tr = Trainer(GAN2Shape, bench.face_config(8), device=dev)
m = tr.model
image, latent = bench.synthetic_sample(m, 1234, dev)

def step1_regions():
    h = w = 128
    with record_function("R:depth_net"):
        with torch.no_grad():
            depth_raw = m.depth_net(image)
    with record_function("R:clamp_depth"):
        depth = m.get_clamped_depth(depth_raw.squeeze(1), h, w)
    with record_function("R:view_net"):
        with torch.no_grad():
            view = m.viewpoint_net(image)
    with record_function("R:view_transform"):
        view = view + m.view_light_sampler.view_mean.unsqueeze(0)
        m.renderer.set_transform_matrices(m.get_view_transformation(view))
    with record_function("R:albedo_net"):
        albedo = m.albedo_net(image)
    with record_function("R:light_net"):
        with torch.no_grad():
            lighting = m.lighting_net(image)
        lighting = lighting + m.view_light_sampler.light_mean.unsqueeze(0)
        la, lb, ld = m.get_lighting_directions(lighting)
    with record_function("R:normal"):
        normal = m.renderer.get_normal_from_depth(depth)
    with record_function("R:shading"):
        diffuse, texture = m.get_shading(normal, la, lb, ld, albedo)
    with record_function("R:warp_canon_depth"):
        recon_depth = m.renderer.warp_canon_depth(depth)
    with record_function("R:inv_warp_grid"):
        grid = m.renderer.get_inv_warped_2d_grid(recon_depth)
    with record_function("R:grid_sample_mask"):
        mask = (recon_depth < 1.2).float().unsqueeze(1).detach()
        recon_im = F.grid_sample(texture, grid, mode='bilinear', align_corners=True).clamp(min=-1, max=1)
    with record_function("R:loss_photo"):
        l1 = m.photometric_loss(recon_im[:1], image, mask=mask[:1])
    with record_function("R:loss_lpips"):
        lp = torch.mean(m.perceptual_loss(recon_im[:1] * mask[:1], image * mask[:1]))
    with record_function("R:loss_smooth"):
        ls = m.smooth_loss(depth) + m.smooth_loss(diffuse)
    loss = l1 + lp + 0.01 * ls
    with record_function("R:backward"):
        loss.backward()
    with record_function("R:optim"):
        tr.optim_step1.step()
        tr.optim_step1.zero_grad()

for _ in range(3):
    step1_regions()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step1_regions()
    torch.cuda.synchronize()
ev = prof.events()
regions = [e for e in ev if e.name.startswith("R:")]
kern = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA]
print("total device events", len(kern))
# attribute kernels to regions by CPU launch time (correlation via time range of the region on CPU)
cpu_launch = [e for e in ev if e.device_type == torch.autograd.DeviceType.CPU and e.kernels]
cnt = collections.Counter(); tim = collections.Counter()
for e in cpu_launch:
    t = e.time_range.start
    reg = "other"
    for r in regions:
        if r.time_range.start <= t <= r.time_range.end:
            reg = r.name; break
    for k in e.kernels:
        cnt[reg] += 1; tim[reg] += k.duration
for r, c in cnt.most_common():
    print(f"{r:24s} kernels={c:5d}  device_us={tim[r]:9.1f}")
