"""The mapper's optimisation step (reference src/mapping.py:60-148, `Mapper.optimize_map`):
render a ray batch, colour / depth / SDF / eikonal losses, backward, clip_grad_norm_(35), AdamW --
with the ray batch optionally sharded over the GPUs of a node (distributed.py)."""
import torch

from .distributed import FlatGradReducer, mapping_loss_sharded, shard_rays


def make_optimizer(model, net_lr=1e-3, grid_lr=1e-2, fused=None):
    """reference src/mapping.py:55-58 (same hyper-parameters).  On the GPU the single-kernel (`fused`) AdamW is
    used: the step is identical, but the 6 parameter tensors are updated by one launch instead of ~20."""
    groups = [{"params": model.get_training_parameters(), "lr": net_lr},
              {"params": model.get_volume_parameters(), "lr": grid_lr}]
    if fused is None:
        fused = all(p.is_cuda for g in groups for p in g["params"])
    kw = dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    if fused:
        try:
            return torch.optim.AdamW(groups, fused=True, **kw)
        except (RuntimeError, TypeError, ValueError):
            pass
    return torch.optim.AdamW(groups, **kw)


class MapTrainer:
    def __init__(self, model, renderer, net_lr=1e-3, grid_lr=1e-2, w_color=2.0, w_sdf=2.0, w_eikonal=0.1,
                 uncertainty=True, group=None, rank=0, world=1):
        self.model, self.renderer = model, renderer
        self.train_params = model.get_training_parameters() + model.get_volume_parameters()
        self.optimizer = make_optimizer(model, net_lr, grid_lr)
        self.w = dict(w_color=w_color, w_sdf=w_sdf, w_eikonal=w_eikonal, uncertainty=uncertainty)
        self.group, self.rank, self.world = group, rank, world
        self.reducer = FlatGradReducer(self.train_params) if world > 1 else None

    def step(self, rays_o, rays_d, rays_color, rays_depth, perturb_rand=None):
        """One joint iteration on the GLOBAL batch (every rank passes the same tensors; each renders
        its contiguous shard).  Returns the global loss as a 0-dim tensor (no host sync in the step)."""
        if self.world > 1:
            rays_o, rays_d, rays_color, rays_depth = shard_rays([rays_o, rays_d, rays_color, rays_depth],
                                                                self.rank, self.world)
        self.optimizer.zero_grad(set_to_none=False)
        z_vals, dists = self.renderer.sample(rays_o, rays_d, self.model.bound, rays_depth, perturb_rand)
        ret = self.renderer.eval_points(rays_o, rays_d, z_vals, dists, self.model, None)
        loss, loss_value = mapping_loss_sharded(ret, rays_color, rays_depth, self.model.compute_sdf_error,
                                                self.group, **self.w)
        loss.backward()
        if self.reducer is not None:
            self.reducer.reduce(self.group)          # sum of shard gradients == single-GPU gradient
        torch.nn.utils.clip_grad_norm_(self.train_params, max_norm=35.0, foreach=True if rays_o.is_cuda else None)
        self.optimizer.step()
        return loss_value
