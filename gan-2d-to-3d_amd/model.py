"""GAN2Shape model: the three training steps on the MI355X-native kernels.

Host-side mirror of GAN2Shape/model.py:18-470 — same constructor `GAN2Shape(config, debug)`, the
step API the trainer calls through getattr (trainer.py:103-104)
`forward_step{1,2,3}(images, latents, collected, n_proj_samples=...) -> (loss, collected)`,
`depth_net_forward`, `evaluate_results`, checkpoint helpers, and the attributes the optimisers are
built from (`albedo_net`, `offset_encoder_net`, `lighting_net`, `viewpoint_net`, `depth_net`).

Conscious differences (none changes a returned value):
  * `device` argument instead of hard-coded .cuda() (model.py:33-46,81);
  * step 2's `collected` stays on the device (the reference round-trips it through the CPU every
    iteration, model.py:222,234);
  * dead work is dropped: `recon_normal` (model.py:142), the `gan_im` generator pass when
    `relative_encoding` is False (:193-200), generator / discriminator weight gradients (G and D
    are frozen; their parameters get requires_grad=False);
  * `center_w` / `center_h` (:201-203) are constants of the frozen generator and computed once;
  * every grid_sample passes align_corners=True (torch-1.2 semantics the reference relies on);
  * offline: a missing `gan_ckpt_path` / `view_mvn_path` / `light_mvn_path` falls back to
    random-init G/D and to the `view_mvn` / `light_mvn` entries of the config (mean, cov lists).
"""
import datetime
import logging
import math
import os
from glob import glob

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.distributions.multivariate_normal import MultivariateNormal

from . import networks, utils, zeropool
from .op import grid_sample
from .losses import DiscriminatorLoss, PhotometricLoss, SmoothLoss
from .lpips import PerceptualLoss
from .renderer import Renderer
from .stylegan2 import Discriminator, Generator


_WSUM_CONST = {}


def weighted_total(terms):
    """sum_i w_i * mean(t_i) over `terms` = [(w_i, t_i)] (t_i a scalar loss or a vector of per-sample
    losses) as TWO launches forward and ONE backward: one concatenation and one dot product with a
    constant weight vector, instead of a mean / multiply / add (and their backward nodes) per term.
    Same value up to the order of a handful of fp32 additions."""
    flat = [t.reshape(-1) for _, t in terms]
    # a weight is a number (weight of the term's MEAN) or a tuple of per-element weights
    per_elem = [tuple(float(v) for v in w) if isinstance(w, (tuple, list)) else (float(w) / f.numel(),) * f.numel()
                for (w, _), f in zip(terms, flat)]
    assert all(len(pe) == f.numel() for pe, f in zip(per_elem, flat))
    key = (tuple(per_elem), flat[0].device)
    wv = _WSUM_CONST.get(key)
    if wv is None:
        if flat[0].is_cuda and torch.cuda.is_current_stream_capturing():
            # a host-to-device upload cannot be captured: the eager warm-up iterations build every constant
            raise RuntimeError("weighted_total: first use of this weight vector under graph capture; run the step eagerly once")
        if len(_WSUM_CONST) > 64:
            _WSUM_CONST.clear()
        wv = _WSUM_CONST[key] = torch.tensor([v for pe in per_elem for v in pe], dtype=torch.float32).to(flat[0].device)
    return torch.dot(torch.cat(flat), wv)


class GAN2Shape(nn.Module):
    NETS = ['lighting', 'viewpoint', 'depth', 'albedo', 'offset_encoder']

    def __init__(self, config, debug=False, device="cuda"):
        super().__init__()
        self.device = torch.device(device)
        self.z_dim = config.get('z_dim')
        self.debug = debug
        # frozen StyleGAN2 pair (model.py:26-37)
        self.generator = Generator(config.get('gan_size'), self.z_dim, 8,
                                   channel_multiplier=config.get('channel_multiplier'))
        self.discriminator = Discriminator(config.get('gan_size'),
                                           channel_multiplier=config.get('channel_multiplier'))
        ckpt_path = config.get('gan_ckpt_path')
        self.gan_pretrained = bool(ckpt_path) and os.path.exists(ckpt_path)
        if self.gan_pretrained:
            gan_ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=True)
            self.generator.load_state_dict(gan_ckpt['g_ema'], strict=False)
            self.discriminator.load_state_dict(gan_ckpt['d'], strict=False)
        else:
            logging.warning("gan_ckpt_path %r not found: random-initialised StyleGAN2", ckpt_path)
        self.generator = self.generator.to(self.device).eval().requires_grad_(False)
        self.discriminator = self.discriminator.to(self.device).eval().requires_grad_(False)

        self.image_size = config.get('image_size')
        self.collected = None
        # `deterministic: true` (added; no reference counterpart): reproducible launch partitions for
        # the whole process (include/g2s.h g2s_set_deterministic) — bit-identical network forwards run
        # to run, slower.  Absent = leave the process setting alone.
        if config.get('deterministic') is not None and self.device.type == "cuda":
            from . import lib as _lib
            _lib.set_deterministic(bool(config['deterministic']))
        # BASELINE config 5: fp16 operands / fp32 accumulation for the frozen G / D / VGG GEMMs (a
        # process-wide switch of modconv.py; absent = the reference's fp32 arithmetic)
        if config.get('mfma_operands') is not None:
            from . import modconv as _mc
            if config['mfma_operands'] not in ('f32', 'f16'):
                raise ValueError("mfma_operands must be 'f32' or 'f16'")
            _mc.OPERANDS = config['mfma_operands']

        self.lighting_net = networks.LightingNet(self.image_size, self.debug).to(self.device)
        self.viewpoint_net = networks.ViewpointNet(self.image_size, self.debug).to(self.device)
        self.depth_net = networks.DepthNet(self.image_size, self.debug).to(self.device)
        self.albedo_net = networks.AlbedoNet(self.image_size, self.debug).to(self.device)
        self.offset_encoder_net = networks.OffsetEncoder(self.image_size, debug=self.debug).to(self.device)

        # depth + albedo and viewpoint + lighting are pairs of the same architecture on the same input
        # (networks.py:53-167): on the GPU each pair runs as ONE pass with grouped convolutions —
        # half the launches of these launch-latency-bound nets; per net the arithmetic is unchanged
        self.paired_nets = bool(config.get('paired_nets', True)) and self.device.type == 'cuda'
        if self.paired_nets:
            networks.pair_parameters(self.depth_net, self.albedo_net)
            networks.pair_parameters(self.viewpoint_net, self.lighting_net)

        # Misc (model.py:48-66)
        self.max_depth = 1.1
        self.min_depth = 0.9
        self.border_depth = 0.7 * self.max_depth + 0.3 * self.min_depth
        self.lam_perc = 1
        self.lam_smooth = 0.01
        self.lam_regular = 0.01
        self.xyz_rotation_range = config.get('xyz_rotation_range', 60)
        self.xy_translation_range = config.get('xy_translation_range', 0.1)
        self.z_translation_range = config.get('z_translation_range', 0.1)
        self.use_mask = config.get('use_mask', True)
        self.relative_encoding = config.get('relative_encoding', False)
        self.rand_light = config.get('rand_light', [-1, 1, -0.2, 0.8, -0.1, 0.6, -0.6])
        self.truncation = config.get('truncation', 1)
        if self.truncation < 1:
            with torch.no_grad():
                self.mean_latent = self.generator.mean_latent(4096)
        else:
            self.mean_latent = None
        self.renderer = Renderer(config, self.image_size, self.min_depth, self.max_depth,
                                 device=self.device)

        view_mvn_path = config.get('view_mvn_path', 'checkpoints/view_light/view_mvn.pth')
        light_mvn_path = config.get('light_mvn_path', 'checkpoints/view_light/light_mvn.pth')
        self.view_light_sampler = ViewLightSampler(view_mvn_path, light_mvn_path,
                                                   config.get('view_scale', 1), self.device,
                                                   config.get('view_mvn'), config.get('light_mvn'))
        self.ckpt_paths = config.get('our_nets_ckpts')

        # losses (model.py:77-83)
        self.smooth_loss = SmoothLoss()
        self.perceptual_loss = PerceptualLoss(
            model='net-lin', net='vgg', lin_weights_path=config.get('lpips_lin_weights'),
            vgg_weights_path=config.get('lpips_vgg_weights')).to(self.device)
        self.discriminator_loss = DiscriminatorLoss()
        self.photometric_loss = PhotometricLoss()

        self._depth_border = None
        self._centers = None
        self.batch_mean = None  # see get_clamped_depth
        self._relight_consts = None
        # run the independent D / A / V / L nets as concurrent stream branches (see _fork); off by
        # default: measured slower on MI355X (step 1: 3.45 -> 5.66 ms) — a forked HIP graph pays
        # more for its cross-queue joins than the overlapped launch latencies give back
        self.parallel_nets = bool(config.get('parallel_nets', False))
        self._side_streams = []

    # ------------------------------------------------------------------ small helpers
    def rescale_depth(self, depth):
        return (1 + depth) / 2 * self.max_depth + (1 - depth) / 2 * self.min_depth

    def depth_net_forward(self, inputs, prior):
        """model.py:88-93, including the (B,H,W) - (1,1,1,1) broadcast to a 4-D depth."""
        depth_raw = self.depth_net(inputs).squeeze(1)
        mean = depth_raw.view(1, 1, -1).mean(2) if self.batch_mean is None else self.batch_mean(depth_raw)
        depth = depth_raw - mean.view(1, 1, 1, 1)
        depth = depth.tanh()
        depth = self.rescale_depth(depth)
        return F.mse_loss(depth[0], prior.detach().expand(depth.size(1), -1, -1)), depth

    def get_view_transformation(self, view):
        return torch.cat([view[:, :3] * math.pi / 180 * self.xyz_rotation_range,
                          view[:, 3:5] * self.xy_translation_range,
                          view[:, 5:] * self.z_translation_range], 1)

    def _set_view(self, view):
        """renderer.set_transform_matrices(get_view_transformation(view)) (model.py:119-120)."""
        self.renderer.set_view(view, math.pi / 180 * self.xyz_rotation_range,
                               self.xy_translation_range, self.z_translation_range)

    def get_clamped_depth(self, depth_raw, h, w, clamp_border=True):
        """model.py:337-345: centre over the WHOLE batch, tanh, rescale; the two left/right border
        columns blend towards border_depth with weight 1.02."""
        # batch_mean: None = this process holds the whole batch; the joint trainer under data
        # parallelism sets sharding.global_mean so that the centre stays the WHOLE-batch mean
        if self.batch_mean is None and depth_raw.is_cuda and self.renderer.fused and depth_raw.dtype == torch.float32 \
                and depth_raw.shape[-1] == w and w >= 4:
            from .fused_geometry import depth_head     # the whole chain below as one kernel each way
            return depth_head(depth_raw, w, self.min_depth, self.max_depth, clamp_border, self.border_depth)
        mean = depth_raw.view(1, -1).mean(1) if self.batch_mean is None else self.batch_mean(depth_raw)
        depth_centered = depth_raw - mean.view(1, 1, 1)
        depth = self.rescale_depth(torch.tanh(depth_centered))
        if clamp_border:
            if self._depth_border is None or self._depth_border.shape[-2:] != (h, w):
                border = torch.zeros(1, h, w - 4, device=depth.device)
                self._depth_border = F.pad(border, (2, 2), mode='constant', value=1.02)
            depth = depth * (1 - self._depth_border) + self._depth_border * self.border_depth
        return depth

    def get_lighting_directions(self, lighting):
        lighting_a = lighting[:, :1] / 2 + 0.5  # ambience term
        lighting_b = lighting[:, 1:2] / 2 + 0.5  # diffuse term
        lighting_d = torch.cat([lighting[:, 2:], lighting.new_ones(lighting.size(0), 1)], 1)
        lighting_d = lighting_d / ((lighting_d ** 2).sum(1, keepdim=True)) ** 0.5
        return lighting_a, lighting_b, lighting_d

    def get_shading(self, normal, lighting_a, lighting_b, lighting_d, albedo):
        diffuse_shading = (normal * lighting_d.view(-1, 1, 1, 3)).sum(3).clamp(min=0).unsqueeze(1)
        shading = lighting_a.view(-1, 1, 1, 1) + lighting_b.view(-1, 1, 1, 1) * diffuse_shading
        texture = (albedo / 2 + 0.5) * shading * 2 - 1
        return diffuse_shading, texture

    def _shade(self, normal, lighting, albedo):
        """get_lighting_directions + get_shading (model.py:347-360) -> (a, b, diffuse, texture);
        one fused kernel on the GPU."""
        if lighting.is_cuda and self.renderer.fused:
            from .fused_geometry import shading
            diffuse_shading, texture = shading(normal, lighting, albedo)
            ab = lighting[:, :2] / 2 + 0.5          # ambience and diffuse terms, one pass for both
            return ab[:, :1], ab[:, 1:2], diffuse_shading, texture
        lighting_a, lighting_b, lighting_d = self.get_lighting_directions(lighting)
        diffuse_shading, texture = self.get_shading(normal, lighting_a, lighting_b, lighting_d, albedo)
        return lighting_a, lighting_b, diffuse_shading, texture

    @staticmethod
    def _head(x, n):
        """x[:n]; the tensor itself when it has exactly n rows (an identity slice would still cost a
        zero-fill + copy in backward)."""
        return x if x.shape[0] == n else x[:n]

    def _no_grad_if(self, cond):
        return torch.no_grad() if cond else torch.enable_grad()

    def _fork(self, jobs):
        """Run independent sub-networks concurrently: job 0 on the current stream, the others on side
        streams forked from it and joined before returning.  The D / A / V / L nets are chains of
        small launch-latency-bound kernels; as parallel branches (of the captured HIP graph, or of
        the eager stream set) their latencies overlap.  autograd replays each branch's backward on
        the stream of its forward, so the backward passes overlap too.  Results are identical:
        only the order in which independent kernels reach the GPU changes."""
        if len(jobs) < 2 or not self.parallel_nets or self.device.type != 'cuda':
            return [job() for job in jobs]
        main = torch.cuda.current_stream(self.device)
        while len(self._side_streams) < len(jobs) - 1:
            self._side_streams.append(torch.cuda.Stream(self.device))
        outs = [None] * len(jobs)
        for i, (job, side) in enumerate(zip(jobs[1:], self._side_streams), 1):
            side.wait_stream(main)
            with torch.cuda.stream(side):
                outs[i] = job()
        outs[0] = jobs[0]()
        for out, side in zip(outs[1:], self._side_streams):
            main.wait_stream(side)
            for t in (out if isinstance(out, (tuple, list)) else (out,)):
                t.record_stream(main)
        return outs

    # ------------------------------------------------------------------ step 1
    def forward_step1(self, images, latents, collected, step1=True, eval=False, **kwargs):
        """model.py:95-173: optimise the albedo net (step1=True) / everything (step1=False).

        `_nets`, `_defer_perc` are internal (forward_step3 batches the image with its projected
        samples through V / L and through LPIPS; the arithmetic per sample is unchanged)."""
        b = 1
        h, w = self.image_size, self.image_size
        if eval:
            zeropool.end()      # evaluation is no training step: nothing of it is served from (or left in) a step's pool
        elif step1 and images.is_cuda:   # a step of its own (forward_step3 begins the pool for its inner call)
            zeropool.begin(1, images.device)

        def frozen_if_step1(net, x):  # model.py:99-122: only the albedo net learns in step 1
            def job():
                with self._no_grad_if(step1):
                    return net(x)
            return job

        pre = kwargs.get('_nets')
        if pre is None and self.paired_nets:
            def vl_pair():
                with self._no_grad_if(step1):
                    return networks.forward_pair(self.viewpoint_net, self.lighting_net, images)
            # the two paired passes read the same image and nothing else: two branches when parallel_nets is on
            (depth_raw, albedo), (view, lighting) = self._fork([
                lambda: networks.forward_pair(self.depth_net, self.albedo_net, images, train_a=not step1), vl_pair])
            if step1:
                depth_raw = depth_raw.detach()
        elif pre is None:  # the four nets read the same image and nothing else: independent chains
            depth_raw, albedo, view, lighting = self._fork([
                frozen_if_step1(self.depth_net, images), lambda: self.albedo_net(images),
                frozen_if_step1(self.viewpoint_net, images), frozen_if_step1(self.lighting_net, images)])
        else:
            depth_raw, albedo, view, lighting = pre
        depth = self.get_clamped_depth(depth_raw.squeeze(1), h, w)
        view = view + self.view_light_sampler.view_mean.unsqueeze(0)
        self._set_view(view)
        lighting = lighting + self.view_light_sampler.light_mean.unsqueeze(0)

        normal = self.renderer.get_normal_from_depth(depth)
        lighting_a, lighting_b, diffuse_shading, texture = self._shade(normal, lighting, albedo)

        recon_depth = self.renderer.warp_canon_depth(depth)
        grid_2d_from_canon = self.renderer.get_inv_warped_2d_grid(recon_depth)
        margin = (self.max_depth - self.min_depth) / 2
        # invalid border pixels have been clamped at max_depth+margin
        recon_im_mask = (recon_depth < self.max_depth + margin).float().unsqueeze(1).detach()
        recon_im = grid_sample(texture, grid_2d_from_canon, -1, 1)   # F.grid_sample(..).clamp(-1, 1), one launch
        if eval:
            return recon_im, recon_depth

        recon_b, mask_b = self._head(recon_im, b), self._head(recon_im_mask, b)   # model.py:150: b = 1
        loss_l1_im = self.photometric_loss(recon_b, images, mask=mask_b)
        perc_pair = (recon_b * mask_b, images * mask_b)
        smooth_terms = [(self.lam_smooth, self.smooth_loss(depth)), (self.lam_smooth, self.smooth_loss(diffuse_shading))]
        canon_mask = None if len(images) == 1 else [None] * len(images)
        collected = (normal, lighting_a, lighting_b, albedo, depth, canon_mask)
        if kwargs.get('_defer_perc'):  # caller adds lam_perc * mean(LPIPS(*perc_pair)) and sums the terms
            return [(1.0, loss_l1_im)] + smooth_terms, collected, perc_pair
        # loss_total = loss_l1_im + lam_perc * mean(perc) + lam_smooth * (smooth(depth) + smooth(shading))
        loss_total = weighted_total([(1.0, loss_l1_im), (self.lam_perc, self.perceptual_loss(*perc_pair))] + smooth_terms)
        return loss_total, collected

    # ------------------------------------------------------------------ step 2
    def _latent_centers(self):
        if self._centers is None:
            with torch.no_grad():
                zero = torch.zeros(1, self.z_dim, device=self.device)
                self._centers = (self.generator.style_forward(zero),
                                 self.generator.style_forward(zero, depth=8 - 2))
        return self._centers

    def forward_step2(self, image, latent, collected, n_proj_samples=8, **kwargs):
        """model.py:175-223: optimise the offset encoder so that G reproduces the pseudo samples."""
        F1_d = 2  # number of mapping network layers used to regularize the latent offset
        if image.is_cuda:
            zeropool.begin(2, image.device)
        *tensors, canon_mask = collected
        normal, light_a, light_b, albedo, depth = [t.detach() for t in tensors]

        with torch.no_grad():
            pseudo_im, mask = self.sample_pseudo_imgs(n_proj_samples, normal, light_a, light_b,
                                                      albedo, depth, canon_mask,
                                                      _draws=kwargs.get('_draws'))
            gan_im = None
            if self.relative_encoding:
                gan_im, _ = self.generator([latent], input_is_w=True,
                                           truncation_latent=self.mean_latent,
                                           truncation=self.truncation, randomize_noise=False)
                gan_im = utils.resize(gan_im.clamp(min=-1, max=1), [self.image_size, self.image_size])
            center_w, center_h = self._latent_centers()

        latent_projection = self.latent_projection(pseudo_im, gan_im, latent, center_w, center_h, F1_d)
        projected_image, offset = self.generator.invert(pseudo_im, latent_projection,
                                                        self.truncation, self.mean_latent)
        projected_image = utils.resize(projected_image, [self.image_size, self.image_size])
        self.loss_l1 = self.photometric_loss(projected_image, pseudo_im, mask=mask)
        self.loss_rec = self.discriminator_loss(self.discriminator, projected_image, pseudo_im,
                                                mask=mask)
        self.loss_latent_norm = torch.mean(offset ** 2)
        loss_total = weighted_total([(1.0, self.loss_l1), (1.0, self.loss_rec), (self.lam_regular, self.loss_latent_norm)])
        return loss_total, (projected_image.detach(), mask.detach())

    def latent_projection(self, image, gan_im, latent, center_w, center_h, F1_d):
        """model.py:282-289."""
        if self.relative_encoding:
            # one pass of E over the samples and the GAN image together (the encoder has no batch
            # statistics: per-sample results equal the reference's two calls, launches halve)
            both = self.offset_encoder_net(torch.cat([image, gan_im], 0))
            offset = both[:len(image)] - both[len(image):]
        else:
            offset = self.offset_encoder_net(image)
        hidden = offset + center_h
        offset = self.generator.style_forward(hidden, skip=8 - F1_d) - center_w
        return offset, latent + offset

    def sample_pseudo_imgs(self, n_images, normal, light_a, light_b, albedo, depth, canon_mask=None,
                           _draws=None):
        """model.py:291-328: random relighting (3 uniform draws) + n random views.
        `_draws` = (light dxy [n,2], diffuse rand [n,1,1,1], views [n,6]) replaces the three random
        draws (parity tests feed the draws of a recorded reference run; the device generator cannot
        reproduce a CPU generator's stream)."""
        h, w = self.image_size, self.image_size
        dev = self.device
        x_min, x_max, y_min, y_max, diffuse_min, diffuse_max, alpha = self.rand_light
        rand_light_dxy = torch.empty(n_images, 2, device=dev)
        rand_light_dxy[:, 0].uniform_(x_min, x_max)
        rand_light_dxy[:, 1].uniform_(y_min, y_max)
        if _draws is not None:
            rand_light_dxy = _draws[0].to(dev)
        rand = torch.empty(n_images, 1, 1, 1, device=dev).uniform_(diffuse_min, diffuse_max)
        if _draws is not None:
            rand = _draws[1].to(dev)
        if rand.is_cuda and self.renderer.fused and normal.dtype == torch.float32:
            # the same relighting through the fused shading kernel (csrc/geometry.hip shading_fwd computes
            # a = l0/2 + .5, b = l1/2 + .5, direction = normalize(l2, l3, 1)): hand it the light vector
            # whose a and b are light_a + alpha * rand and light_b + rand
            from .fused_geometry import shading
            if self._relight_consts is None or self._relight_consts[0] != alpha:    # keyed on alpha: rand_light may change
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("sample_pseudo_imgs: constants for this rand_light are built eagerly; run step 2 once before capture")
                self._relight_consts = (alpha, torch.tensor([[2.0 * alpha, 2.0]], device=dev))
            ab = torch.cat([light_a[:1].reshape(1, 1), light_b[:1].reshape(1, 1)], 1) * 2 - 1
            light = torch.cat([ab + rand.view(-1, 1) * self._relight_consts[1], rand_light_dxy], 1)
            _, rand_light_im = shading(normal[:1], light, albedo[:1])
        else:
            rand_light_d = torch.cat([rand_light_dxy, torch.ones(n_images, 1, device=dev)], 1)
            rand_light_d = rand_light_d / ((rand_light_d ** 2).sum(1, keepdim=True)) ** 0.5
            rand_diffuse_shading = (normal[0, None] * rand_light_d.view(-1, 1, 1, 3)).sum(3)\
                .clamp(min=0).unsqueeze(1)
            rand_diffuse = (light_b[0, None].view(-1, 1, 1, 1) + rand) * rand_diffuse_shading
            rand_shading = light_a[0, None].view(-1, 1, 1, 1) + alpha * rand + rand_diffuse
            rand_light_im = (albedo[0, None] / 2 + 0.5) * rand_shading * 2 - 1

        depth = depth[0, None]
        if canon_mask is not None:
            mask = canon_mask.expand(n_images, 3, h, w)
        else:
            mask = torch.ones(n_images, 3, h, w, device=dev)

        if _draws is not None:
            rand_views_trans = self.get_view_transformation(_draws[2].to(dev))
        else:       # get_view_transformation of the draws, its three factors folded into the sampler's affine map
            rot = math.pi / 180 * self.xyz_rotation_range
            rand_views_trans = self.view_light_sampler.sample(
                n_images, 'view', scale=(rot, rot, rot, self.xy_translation_range, self.xy_translation_range,
                                         self.z_translation_range))
        pseudo_im, mask = self.renderer.render_given_view(rand_light_im, depth.expand(n_images, h, w),
                                                          view=rand_views_trans, mask=mask,
                                                          grid_sample=True)
        mask = mask[:, 0, None, ...]
        return pseudo_im.clamp(min=-1, max=1), mask.contiguous()

    # ------------------------------------------------------------------ step 3
    def forward_step3(self, images, latents, collected, **kwargs):
        """model.py:225-280: optimise V, L, D, A on the image and its projected samples."""
        projected_samples, masks = collected
        if images.is_cuda:
            zeropool.begin(3, images.device)
        # the reference draws (and discards) a permutation here (model.py:231-233): keep the draw so
        # that a seeded run consumes the CPU generator identically
        torch.randperm(len(projected_samples))
        projected_samples, masks = projected_samples.to(self.device), masks.to(self.device)

        # V and L see the image and its projected samples in ONE batch (1 + n), and so does LPIPS
        # below: the reference runs them as two calls each (model.py:113-131,243-250 and :159,275);
        # per-sample results are unchanged (no batch statistics in these nets), launches halve.
        b = len(projected_samples)
        both = torch.cat([images[:1], projected_samples], 0) if len(images) == 1 else None
        if both is not None:
            if self.paired_nets:
                (depth_raw, albedo1), (view_all, light_all) = self._fork([
                    lambda: networks.forward_pair(self.depth_net, self.albedo_net, images),
                    lambda: networks.forward_pair(self.viewpoint_net, self.lighting_net, both)])
            else:
                depth_raw, albedo1, view_all, light_all = self._fork([
                    lambda: self.depth_net(images), lambda: self.albedo_net(images),
                    lambda: self.viewpoint_net(both), lambda: self.lighting_net(both)])
            # one split each (a single concatenation in backward) instead of two slices (a zero-fill
            # + copy each)
            view1, view = view_all.split([1, b])
            light1, light = light_all.split([1, b])
            step1_loss, collected, perc1 = self.forward_step1(
                images, None, None, step1=False, _nets=(depth_raw, albedo1, view1, light1),
                _defer_perc=True)
        else:
            step1_loss, collected = self.forward_step1(images, None, None, step1=False)
            perc1 = None
            view, light = self.viewpoint_net(projected_samples), self.lighting_net(projected_samples)
        normal, _, _, albedo, depth, _ = collected

        view = view + self.view_light_sampler.view_mean.unsqueeze(0)
        self._set_view(view)
        light = light + self.view_light_sampler.light_mean.unsqueeze(0)
        _, _, diffuse_shading, texture = self._shade(normal, light, albedo)

        depth = depth.expand(b, self.image_size, self.image_size)
        recon_depth = self.renderer.warp_canon_depth(depth)
        grid_2d_from_canon = self.renderer.get_inv_warped_2d_grid(recon_depth)
        margin = (self.max_depth - self.min_depth) / 2
        recon_im_mask = (recon_depth < self.max_depth + margin).float().unsqueeze(1).detach() * masks
        recon_im = grid_sample(texture, grid_2d_from_canon, -1, 1)   # F.grid_sample(..).clamp(-1, 1), one launch

        recon_b, mask_b = self._head(recon_im, b), self._head(recon_im_mask, b)
        loss_l1_im = self.photometric_loss(recon_b, projected_samples, mask=mask_b)
        pred, target = recon_b * mask_b, projected_samples * mask_b
        if perc1 is not None:
            # step1_loss arrives as its list of terms; LPIPS ran once on (image, n samples): its first entry
            # is the inner step-1 term (mean over 1), the other b the projected samples' (mean over b)
            perc = self.perceptual_loss(torch.cat([perc1[0], pred], 0), torch.cat([perc1[1], target], 0)).reshape(-1)
            terms = step1_loss + [(1.0, loss_l1_im), ((self.lam_perc,) + (self.lam_perc / b,) * b, perc)]
            return weighted_total(terms), None
        loss_perc_im = torch.mean(self.perceptual_loss(pred, target))
        return step1_loss + loss_l1_im + self.lam_perc * loss_perc_im, None

    # ------------------------------------------------------------------ evaluation / checkpoints
    def evaluate_results(self, image):
        """model.py:362-368."""
        with torch.no_grad():
            recon_im, recon_depth = self.forward_step1(image, None, None, eval=True)
            depth_raw = self.depth_net(image).squeeze(1)
            recon_depth = self.get_clamped_depth(depth_raw, self.image_size, self.image_size,
                                                 clamp_border=False)
        return recon_im, recon_depth

    def reinitialize_model(self):
        """model.py:370-383 (never called by the reference's trainer)."""
        for name in GAN2Shape.NETS:
            for layer in getattr(self, f'{name}_net').modules():
                if hasattr(layer, 'reset_parameters'):
                    layer.reset_parameters()

    def save_checkpoint(self, img_idx, stage, total_it, category='car'):
        """model.py:385-408: one .pth per net, {total_it, dataset, model_state_dict}.  Like the
        reference, a failure (no `our_nets_ckpts` in the config, I/O error) is logged and training
        continues."""
        try:
            now = datetime.datetime.now().strftime("%Y_%m_%d_%H_%M")
            for net in GAN2Shape.NETS:
                save_dict = {'total_it': total_it, 'dataset': category,
                             'model_state_dict': getattr(self, f'{net}_net').state_dict()}
                filename = self.build_checkpoint_path(self.ckpt_paths['VLADE_nets'], category, net,
                                                      img_idx, stage, total_it, now)
                os.makedirs(os.path.dirname(filename), exist_ok=True)
                with open(filename, 'wb') as f:
                    torch.save(save_dict, f)
        except Exception as e:  # noqa: BLE001  (model.py:406-408 swallows everything)
            logging.error("Error: %s", e)
            logging.error(">>>Saving failed... continuing training<<<")

    def load_from_checkpoints(self, path_base, category):
        paths, indices = self.build_checkpoint_path(path_base, category)
        for path, img_idx in zip(paths, indices):
            self.load_from_checkpoint(path)
            yield img_idx

    def load_from_checkpoint(self, filename_path):
        """model.py:416-423: weights only (no optimiser state, no iteration resume)."""
        for net in GAN2Shape.NETS:
            with open(filename_path(net), 'rb') as f:
                checkpoint = torch.load(f, map_location=self.device, weights_only=True)
            getattr(self, f'{net}_net').load_state_dict(checkpoint['model_state_dict'])

    def build_checkpoint_path(self, base, category, net=None, img_idx="*", stage="*",
                              total_it="*", time="*", general=False):
        """model.py:425-445."""
        if net is not None:
            return f'{base}/{category}/{net}_image_{img_idx}_stage_{stage}_{total_it}_it_{time}.pth'
        net = GAN2Shape.NETS[0]
        possible_paths = glob(f'{base}/{category}/{net}_image_*_stage_*_*_it_*.pth')
        assert possible_paths
        paths, img_ids = [], []
        for path in sorted(possible_paths):
            beginning, end = path.split(net)[:2]
            paths.append(lambda x, b=beginning, e=end: f'{b}{x}{e}')
            words = path.split('_')
            if not general:
                img_ids.append(int(words[words.index('image') + 1]))
        return paths, img_ids


class ViewLightSampler():
    """model.py:448-470.  One standard-normal draw per sample, in sequence — n MultivariateNormal.sample()
    calls consume the device generator exactly like this, so a seeded run sees the reference's draws —
    then the affine map loc + L eps (and the view's yaw scale, folded into loc and L) for all n at
    once: n + 1 launches instead of 5 n."""

    def __init__(self, view_mvn_path, light_mvn_path, view_scale, device="cuda",
                 view_mvn=None, light_mvn=None):
        def load(path, fallback, dim):
            if path and os.path.exists(path):
                d = torch.load(path, map_location="cpu", weights_only=True)
                return d['mean'].float(), d['cov'].float()
            if fallback is not None:
                return (torch.tensor(fallback['mean'], dtype=torch.float32),
                        torch.tensor(fallback['cov'], dtype=torch.float32))
            raise FileNotFoundError(f"{path} not found and no inline mean/cov given in the config")

        vm, vc = load(view_mvn_path, view_mvn, 6)
        lm, lc = load(light_mvn_path, light_mvn, 4)
        self.view_mean = vm.to(device)
        self.light_mean = lm.to(device)
        self.view_scale = view_scale
        self.view_dist = MultivariateNormal(vm.to(device), vc.to(device))
        self.light_dist = MultivariateNormal(lm.to(device), lc.to(device))
        self._affine = {}
        for name, dist in (('view', self.view_dist), ('light', self.light_dist)):
            scale = torch.ones_like(dist.loc)
            if name == 'view':
                scale[1] = view_scale            # sample[0, 1] *= view_scale (model.py:462-463)
            tril = dist._unbroadcasted_scale_tril
            self._affine[name] = ((dist.loc * scale)[None].contiguous(), (tril * scale[:, None]).t().contiguous())

    def sample(self, n=1, sample_type='view', scale=None):
        """`scale` (a tuple, one factor per component): the samples times these factors — folded into the
        affine map, so get_view_transformation (model.py:330-335) of the draws costs no extra launch."""
        loc, tril_t = self._affine[sample_type]
        if scale is not None:
            key = (sample_type, tuple(scale))
            if key not in self._affine:
                sc = torch.tensor(scale, dtype=loc.dtype, device=loc.device)
                self._affine[key] = ((loc * sc).contiguous(), (tril_t * sc[None, :]).contiguous())
            loc, tril_t = self._affine[key]
        eps = torch.empty(n, loc.shape[1], dtype=loc.dtype, device=loc.device)
        for row in eps:
            row.normal_()                        # _standard_normal(event_shape) of one .sample() call
        return torch.addmm(loc.expand(n, -1), eps, tril_t)
