"""Instance-wise training loop: mirror of GAN2Shape/trainer.py:13-171 (`Trainer`).

Same constructor and `fit` signature, so `Trainer(model=GAN2Shape, model_config=cfg, ...)` works as
in the reference's main.py:130-153: per image, optional prior pre-training of the depth net, then
stages x steps {1,2,3} x N iterations of zero_grad / forward_stepK / backward / Adam.step, with the
last `collected` of step K handed to step K+1.  Three persistent Adam optimisers over {A}, {E},
{L,V,D,A} (lr 1e-4, classic L2 weight decay 5e-4), a fresh Adam for every image's prior
pre-training (trainer.py:40-48,131,163-171).

Added (not in the reference): `rank` / `world_size` image sharding — rank r trains images
r, r+W, r+2W, ... with no collective (the per-image optimisation is independent; note that nets
are carried from image to image, so a shard equals the reference run on that rank's subset).
Plotting / wandb hooks are out of scope.
"""
import logging

import torch
from torch.utils.data import DataLoader, Subset

from . import zeropool
from .priors import PriorGenerator
from .sharding import shard_indices  # noqa: F401  (re-exported)


class Trainer():
    def __init__(self, model, model_config, debug=False, plot_intermediate=False, log_wandb=False,
                 save_ckpts=False, load_dict=None, masking_model=None, device="cuda",
                 capturable=False):
        try:
            self.model = model(model_config, debug, device=device)
        except TypeError:  # a model class with the reference's exact (config, debug) signature
            self.model = model(model_config, debug)
        self.device = torch.device(device)
        self.image_size = model_config.get('image_size')
        self.category = model_config.get('category')
        self.n_proj_samples = model_config.get('n_proj_samples', 8)
        self.n_epochs_prior = model_config.get('n_epochs_prior', 1000)
        self.n_workers = model_config.get('n_workers', 0)
        self.learning_rate = model_config.get('learning_rate', 1e-4)
        self.save_ckpts = save_ckpts
        self.debug = debug
        self.capturable = capturable  # Adam(capturable=True): needed to record steps in HIP graphs
        self.prior_generator = PriorGenerator(self.image_size, self.category,
                                              model_config.get('prior_name', 'ellipsoid'),
                                              masking_model=masking_model)
        self.optim_step1 = Trainer.default_optimizer([self.model.albedo_net], lr=self.learning_rate,
                                                     capturable=capturable)
        self.optim_step2 = Trainer.default_optimizer([self.model.offset_encoder_net],
                                                     lr=self.learning_rate, capturable=capturable)
        self.optim_step3 = Trainer.default_optimizer([self.model.lighting_net,
                                                      self.model.viewpoint_net,
                                                      self.model.depth_net,
                                                      self.model.albedo_net], lr=self.learning_rate,
                                                     capturable=capturable)
        self.load_dict = load_dict
        if load_dict is not None:
            paths, _ = self.model.build_checkpoint_path(load_dict['base_path'],
                                                        load_dict['category'], general=True)
            self.model.load_from_checkpoint(paths[-1])
        self.history = []  # (image index, stage, step, last loss) per finished step block

    def fit(self, images_latents, plot_depth_map=False,
            stages=[{'step1': 1, 'step2': 1, 'step3': 1}] * 2, shuffle=False,
            rank=0, world_size=1, graphs=False, **_):
        """graphs=True replays each step kind as a HIP graph (graphs.py): same iterations, same
        order; needs Trainer(..., capturable=True) and a non-default current stream."""
        if graphs:
            return self._fit_graphed(images_latents, stages, shuffle, rank, world_size)
        total_it = 0
        n_stages = len(stages)
        data = images_latents
        if world_size > 1:
            data = Subset(images_latents, shard_indices(len(images_latents), rank, world_size))
        # the original training is instance-based => batch size = 1
        dataloader = DataLoader(data, batch_size=1, shuffle=shuffle, num_workers=self.n_workers)
        for batch in dataloader:
            image, latent, data_index = batch
            image, latent = image.to(self.device), latent.to(self.device)
            data_index = int(data_index[0])
            logging.info(f'Training on image {data_index}')
            if not self.debug and self.load_dict is None:
                self.pretrain_on_prior(image, data_index, plot_depth_map)
            stage = 0
            for stage in range(n_stages):
                old_collected = None
                for step in [1, 2, 3]:
                    optim = getattr(self, f'optim_step{step}')
                    forward = getattr(self.model, f'forward_step{step}')
                    collected, loss = None, None
                    for _ in range(stages[stage][f'step{step}']):
                        optim.zero_grad()
                        loss, collected = forward(image, latent, old_collected,
                                                  n_proj_samples=self.n_proj_samples)
                        loss.backward()
                        optim.step()
                        zeropool.end()       # the step's cleared pool serves nothing beyond the step
                        total_it += 1
                    self.history.append((data_index, stage, step,
                                         None if loss is None else float(loss.detach())))
                    old_collected = collected
            if self.save_ckpts:
                self.model.save_checkpoint(data_index, stage, total_it, self.category)
        logging.info('Finished Training')
        return total_it

    def _fit_graphed(self, images_latents, stages, shuffle, rank, world_size):
        from .graphs import GraphedSteps
        if not self.capturable:
            raise RuntimeError("graphs=True needs Trainer(..., capturable=True)")
        data = images_latents
        if world_size > 1:
            data = Subset(images_latents, shard_indices(len(images_latents), rank, world_size))
        dataloader = DataLoader(data, batch_size=1, shuffle=shuffle, num_workers=self.n_workers)
        total_it = 0
        graphed = None
        for image, latent, data_index in dataloader:
            image, latent = image.to(self.device), latent.to(self.device)
            data_index = int(data_index[0])
            if not self.debug and self.load_dict is None:
                self.pretrain_on_prior(image, data_index)
            if graphed is None:
                graphed = GraphedSteps(self, image, latent)
            else:
                graphed.set_sample(image, latent)   # graphs read the static image / latent buffers
            for stage in range(len(stages)):
                for step in [1, 2, 3]:
                    n = stages[stage][f'step{step}']
                    done = 0
                    if n > 0 and step not in graphed.graphs:
                        if step > 1 and graphed.collected[step - 1] is None:
                            raise RuntimeError(f"step {step} needs at least one step-{step - 1} iteration first")
                        # the warm-up iterations are real iterations: never more than requested
                        done = graphed.capture(step, warmup=min(graphed.warmup, n))
                    for _ in range(n - done):
                        graphed.run(step)
                    total_it += n
                    if n > 0:
                        self.history.append((data_index, stage, step, float(graphed.loss[step])))
            if self.save_ckpts:
                self.model.save_checkpoint(data_index, len(stages) - 1, total_it, self.category)
        return total_it

    def pretrain_on_prior(self, image, i_batch, plot_depth_map=False):
        """trainer.py:130-161: n_epochs_prior Adam steps of mse(depth, prior) on a fresh optimiser."""
        optim = Trainer.default_optimizer([self.model.depth_net])
        prior = self.prior_generator(image, device=self.device)
        losses = []
        for _ in range(self.n_epochs_prior):
            optim.zero_grad()
            loss, depth = self.model.depth_net_forward(image, prior)
            loss.backward()
            optim.step()
            losses.append(loss.detach())
        return [float(v) for v in torch.stack(losses).cpu()] if losses else []

    def _sync_and_step(self, optim):
        """optim.step(), preceded under data parallelism by ONE flat-bucket gradient all-reduce
        (mean) over the parameters this optimiser owns (sharding.allreduce_mean_gradients)."""
        from .sharding import allreduce_mean_gradients
        allreduce_mean_gradients([p for g in optim.param_groups for p in g['params']])
        optim.step()
        zeropool.end()

    @staticmethod
    def default_optimizer(model_list, lr=1e-4, betas=(0.9, 0.999), weight_decay=5e-4,
                          capturable=False):
        param_list = []
        for model in model_list:
            param_list += [p for p in model.parameters() if p.requires_grad]
        # GPU: the whole step as one launch of libg2s (optim.Adam on g2s_adam_step: same update rule,
        # trainer.py:163-171; device-side step counter, so it can be recorded in HIP graphs)
        if param_list and all(p.is_cuda for p in param_list):
            from gan2shape_amd.optim import Adam
            return Adam(param_list, lr=lr, betas=betas, weight_decay=weight_decay)
        return torch.optim.Adam(param_list, lr=lr, betas=betas, weight_decay=weight_decay, capturable=capturable)


class GeneralizingTrainer2(Trainer):
    """Joint training over a set of images (trainer.py:338-479, on GeneralizingTrainer :174-335):
    pre-train the depth net on every image's prior, then per epoch and per batch of images run
    step 1 on the batch and steps 2 / 3 image by image — one shared model.

    Data parallelism (added; the reference is single-process): with world_size W every batch of
    `batch_size` images is dealt round-robin to the ranks (batch_size must be a multiple of W), each
    rank runs the reference's iteration on its images, and every optimiser step is preceded by one
    flat-bucket gradient all-reduce (mean) over RCCL (`sharding.allreduce_mean_gradients`: step 1
    A, step 2 E, step 3 L+V+D+A).  Where the reference centres the depth over the whole image
    batch — prior pre-training (model.py:90) and the batched step 1 (model.py:338) —
    `get_clamped_depth` / `depth_net_forward` use the mean over ALL ranks' images
    (`sharding.global_mean`); steps 2 / 3 run image by image in the reference (trainer.py:436-452),
    so their inner step-1 pass centres each image by its own mean, on every rank as at W = 1.
    W = 1 is the reference's loop exactly.  W > 1 equals it up to the reference's own `b = 1`
    slicing in forward_step1 (model.py:96,150: only the first reconstruction of a call enters the
    losses), which then applies per rank."""

    def __init__(self, model, model_config, **kwargs):
        super().__init__(model, model_config, **kwargs)
        self.n_epochs = model_config.get('n_epochs_generalized', 1)

    def _local_batches(self, images_latents, batch_size, shuffle, rank, world_size):
        if batch_size % world_size:
            raise ValueError(f"batch_size {batch_size} must be a multiple of world_size {world_size}")
        loader = DataLoader(images_latents, batch_size=batch_size, shuffle=shuffle,
                            num_workers=self.n_workers)
        for images, latents, indices in loader:
            if len(images) % world_size:
                raise ValueError(f"{len(images_latents)} images do not fill batches of {batch_size} "
                                 f"evenly over {world_size} ranks")
            sel = slice(rank, None, world_size)
            yield images[sel].to(self.device), latents[sel].to(self.device), [int(i) for i in indices[sel]]

    def fit(self, images_latents, plot_depth_map=False,
            stages=[{'step1': 1, 'step2': 1, 'step3': 1}] * 2, batch_size=2, shuffle=False,
            rank=0, world_size=1, graphs=False, **_):
        """graphs=True: the per-image steps 2 / 3 replay as HIP graphs in two segments with the gradient
        all-reduce between them (graphs.GraphedJointSteps; needs capturable=True and a non-default current
        stream); the batched step 1 stays eager (ragged batches, and under W > 1 a collective in its forward)."""
        from . import sharding
        # centre of the depth over the WHOLE image batch (all ranks) — only where the reference
        # itself runs a batch of images through the depth net: prior pre-training and step 1
        whole_batch = sharding.global_mean if world_size > 1 else None
        total_it = 0
        graphed = None
        if graphs and not self.capturable:
            raise RuntimeError("graphs=True needs GeneralizingTrainer2(..., capturable=True)")
        try:
            if self.load_dict is None:
                self.model.batch_mean = whole_batch
                self.pretrain_on_prior_all(images_latents, batch_size, rank, world_size)
            for epoch in range(self.n_epochs):
                for images, latents, indices in self._local_batches(images_latents, batch_size, shuffle,
                                                                    rank, world_size):
                    loss = collected = None
                    self.model.batch_mean = whole_batch
                    for _ in range(stages[0]['step1']):
                        self.optim_step1.zero_grad()
                        loss, collected = self.model.forward_step1(images, latents, None)
                        loss.backward()
                        self._sync_and_step(self.optim_step1)
                        total_it += 1
                    self.model.batch_mean = None   # steps 2 / 3: one image at a time, its own mean
                    self.history.append((tuple(indices), epoch, 1, None if loss is None else float(loss.detach())))
                    if collected is None:
                        continue
                    normals, lights_a, lights_b, albedos, depths, canon_masks = collected
                    if not isinstance(canon_masks, list):
                        canon_masks = [canon_masks]
                    for bi, index in enumerate(indices):
                        image, latent = images[bi:bi + 1], latents[bi:bi + 1]
                        step1_collected = (normals[bi:bi + 1].detach(), lights_a[bi:bi + 1].detach(),
                                           lights_b[bi:bi + 1].detach(), albedos[bi:bi + 1].detach(),
                                           depths[bi:bi + 1].detach(), canon_masks[bi])
                        loss2 = loss3 = step2_collected = None
                        if graphs:
                            from .graphs import GraphedJointSteps
                            if graphed is None:
                                graphed = GraphedJointSteps(self, image, latent)
                            else:
                                graphed.set_sample(image, latent)
                            graphed.set_source(2, step1_collected)
                            n2, n3 = stages[0]['step2'], stages[0]['step3']
                            n3 = n3 if n2 > 0 else 0        # no step-2 hand-off, no step 3 (as in the eager loop below)
                            for step, n in ((2, n2), (3, n3)):
                                done = 0
                                if n > 0 and step not in graphed.graphs:     # warm-up iterations are real iterations
                                    done = graphed.capture(step, warmup=min(graphed.warmup, n))
                                for _ in range(n - done):
                                    graphed.run(step)
                                total_it += n
                            loss2 = graphed.loss.get(2) if n2 > 0 else None
                            loss3 = graphed.loss.get(3) if n3 > 0 else None
                            self.history.append((index, epoch, 2, None if loss2 is None else float(loss2)))
                            self.history.append((index, epoch, 3, None if loss3 is None else float(loss3)))
                            continue
                        for _ in range(stages[0]['step2']):
                            self.optim_step2.zero_grad()
                            loss2, step2_collected = self.model.forward_step2(
                                image, latent, step1_collected, n_proj_samples=self.n_proj_samples)
                            loss2.backward()
                            self._sync_and_step(self.optim_step2)
                            total_it += 1
                        for _ in range(stages[0]['step3'] if step2_collected is not None else 0):
                            self.optim_step3.zero_grad()
                            loss3, _c = self.model.forward_step3(image, latent, step2_collected)
                            loss3.backward()
                            self._sync_and_step(self.optim_step3)
                            total_it += 1
                        self.history.append((index, epoch, 2, None if loss2 is None else float(loss2.detach())))
                        self.history.append((index, epoch, 3, None if loss3 is None else float(loss3.detach())))
                if epoch % 20 == 0 and self.save_ckpts and rank == 0:
                    self.model.save_checkpoint("", epoch, total_it, self.category)
        finally:
            self.model.batch_mean = None
        logging.info('Finished Training')
        return total_it

    def pretrain_on_prior_all(self, images_latents, batch_size, rank=0, world_size=1):
        """GeneralizingTrainer.pretrain_on_prior (trainer.py:296-335): one prior per image, then
        n_epochs_prior passes over the batches with a fresh Adam on the depth net."""
        optim = Trainer.default_optimizer([self.model.depth_net])
        priors = {}
        for image, _, index in DataLoader(images_latents, batch_size=1, shuffle=False):
            if (int(index[0]) % batch_size) % world_size == rank % world_size or world_size == 1:
                priors[int(index[0])] = self.prior_generator(image.to(self.device), device=self.device)
        loss = None
        for _ in range(self.n_epochs_prior):
            for images, _latents, indices in self._local_batches(images_latents, batch_size, False,
                                                                 rank, world_size):
                for i in indices:
                    if i not in priors:
                        item = images_latents[i]
                        priors[i] = self.prior_generator(item[0].unsqueeze(0).to(self.device), device=self.device)
                prior = torch.stack([priors[i].reshape(self.image_size, self.image_size) for i in indices])
                optim.zero_grad()
                loss, _depth = self.model.depth_net_forward(images, prior)
                loss.backward()
                self._sync_and_step(optim)
        return None if loss is None else float(loss.detach())
