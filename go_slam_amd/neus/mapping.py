"""Mapping driver (mirrors src/mapping.py:11-302): picks the keyframes to visit, draws their ray batches and runs the
joint iterations on the NeuS hot path (Renderer.sample -> InstantNeuS forward/backward -> fused loss -> fused AdamW).

The keyframe schedule and the ray draws are the reference's (same NumPy / torch RNG consumption, so a seeded run picks
the same pixels).  `optimize_map` differs in formulation only: the reference gathers the rays with a depth measurement
by boolean indexing (a nonzero + host sync per use); here every term is a masked sum over the full batch (sync-free,
`neus/distributed.mapping_loss_sharded`), which gives the same loss and gradients.
"""
import numpy as np
import torch

from .distributed import mapping_loss_sharded
from .mapper import MapTrainer
from .rays import RayBank, build_rays


def random_select(l, k, start=0):
    """k stratified-random indices from start..l-1, zeros dropped (src/nerf_func.py:28-40)"""
    m = (l - start) / k
    idx = np.linspace(start, l - 1 - m, k) + np.random.rand(k) * m
    idx = idx.clip(start, l - 1)
    return [int(i) for i in list(idx) if i > 0]


class Mapper:
    use_ray_bank = True         # False: the reference's per-frame build_rays calls in every iteration (tools/mapper_call_bench.py)

    def __init__(self, cfg, args, slam):
        self.cfg, self.args = cfg, args
        self.verbose = getattr(slam, "verbose", False)
        self.bound = slam.bound
        self.video = slam.video
        self.mapping_net = slam.mapping_net
        self.renderer = slam.renderer
        self.reload_map = slam.reload_map
        m = cfg["mapping"]
        self.device = m["device"]
        self.num_joint_iters = m["iters"]
        self.decay = float(m["decay"])
        self.w_color_loss, self.w_sdf_loss, self.w_eikonal_loss = m["w_color_loss"], m["w_sdf_loss"], m["w_eikonal_loss"]
        self.uncertainty_based = m["uncertainty_weight_loss"]
        if m.get("BA", False):
            raise NotImplementedError("mapping-side camera refinement (mapping.BA) is off in every reference config; "
                                      "the HIP renderer does not return ray-origin / direction gradients")
        self.mapping_pixels = m["pixels"]
        self.mapping_window_size = m["mapping_window_size"]
        self.H, self.W, self.fx, self.fy, self.cx, self.cy = slam.H, slam.W, slam.fx, slam.fy, slam.cx, slam.cy
        self.local_step = self.global_step = self.last_visit = 0
        self.init = True
        net_param = self.mapping_net.get_training_parameters(ignore_keys=())
        grid_param = self.mapping_net.get_volume_parameters()
        self.train_params = list(net_param) + list(grid_param)
        # the joint iteration itself: MapTrainer (fused HIP step captured in a hipGraph when the model is on the GPU with
        # tiny-cuda-nn's fp16 table gradients -- the default --, else autograd + torch AdamW); `self.optimizer` is what
        # the reference's optimize_map is handed: torch's AdamW, or the fused step's param_groups facade
        self.trainer = MapTrainer(self.mapping_net, self.renderer, m["net_lr"], m["grid_lr"], w_color=self.w_color_loss,
                                  w_sdf=self.w_sdf_loss, w_eikonal=self.w_eikonal_loss,
                                  uncertainty=self.uncertainty_based)
        self.optimizer = self.trainer.optimizer

    def optimize_map(self, rays_o, rays_d, rays_color, rays_depth, optimizer, num_joint_iters):
        """mapping iterations on one ray batch (src/mapping.py:59-148)"""
        net = self.mapping_net
        if self.trainer.fused and optimizer is self.trainer.optimizer:
            for _ in range(num_joint_iters):            # forward + loss + backward + clip + AdamW without autograd
                self.local_step += 1
                self.global_step += 1
                self.trainer.step(rays_o, rays_d, rays_color, rays_depth)
            return
        for _ in range(num_joint_iters):
            self.local_step += 1
            self.global_step += 1
            optimizer.zero_grad(set_to_none=False)
            with torch.enable_grad():
                ret = self.renderer.render_batch_ray(rays_o=rays_o, rays_d=rays_d, net=net,
                                                     render_params={"global_step": self.global_step},
                                                     device=self.device, gt_depth=rays_depth)
                loss, _ = mapping_loss_sharded(ret, rays_color, rays_depth, net.compute_sdf_error, None,
                                               w_color=self.w_color_loss, w_sdf=self.w_sdf_loss,
                                               w_eikonal=self.w_eikonal_loss, uncertainty=self.uncertainty_based)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(self.train_params, max_norm=35.0,
                                           foreach=True if rays_o.is_cuda else None)
            optimizer.step()
        optimizer.zero_grad(set_to_none=False)

    def _ray_batch(self, frames, items, n_rays):
        """n_rays random (mask-aware) rays from each frame, concatenated (src/mapping.py:222-240, 262-283).  `items` is a
        RayBank over the frames of this call (the per-frame glue of build_rays done once per call, rays.py) or the
        reference's dict frame -> (color, depth, c2w, gt_c2w, mask)."""
        if isinstance(items, RayBank):
            return items.sample(frames, n_rays)
        H, W = self.H, self.W
        parts = [[], [], [], []]
        for frame in frames:
            color, depth, c2w, _, mask = items[frame]
            out = build_rays(0, H, 0, W, n_rays, H, W, self.fx, self.fy, self.cx, self.cy, c2w, depth, color,
                             self.device, nerf_coordinate=False, dir_normalize=False, mask=mask)
            for acc, x in zip(parts, out):
                acc.append(x.float())
        rays_o, rays_d, depth, color = (torch.cat(p, dim=0) for p in parts)
        return rays_o, rays_d, color, depth

    def __call__(self, the_end=False):
        v = self.video
        cur_idx = int(v.filtered_id.item())                 # keyframes [0, cur_idx) have been filtered
        if cur_idx <= 1:
            return
        num_joint_iters = self.num_joint_iters * (10 if the_end else 1)
        self.local_step = 0
        unvisit_list = list(range(self.last_visit, cur_idx))
        visit_list = [cur_idx - 1, cur_idx - 2]
        if self.last_visit > 0:                             # 10 highest-priority + stratified-random old keyframes
            priority = v.update_priority[:self.last_visit].detach()
            order = torch.sort(priority, dim=0, descending=True).indices
            visit_list += list(order.cpu().numpy())[:10]
            visit_list += random_select(self.last_visit, self.mapping_window_size - 12)
        if hasattr(v, "get_mapping_items"):                 # all hand-outs of the call in one pass (depth_video.py)
            visit_frame = v.get_mapping_items(visit_list, self.device, decay=self.decay)
            unvisit_frame = v.get_mapping_items(unvisit_list, self.device, decay=self.decay)
        else:
            visit_frame = {f: v.get_mapping_item(f, self.device, decay=self.decay) for f in visit_list}
            unvisit_frame = {f: v.get_mapping_item(f, self.device, decay=self.decay) for f in unvisit_list}
        self.mapping_net.update_bound(v.get_bound())
        if self.use_ray_bank:
            bank = lambda items: RayBank(items, self.H, self.W, self.fx, self.fy, self.cx, self.cy, self.device)
            visit_frame, unvisit_frame = bank(visit_frame), bank(unvisit_frame)
        # new keyframes first: window-sized random subsets of them, 10x the iterations on the very first call
        unvisit_factor = num_joint_iters * 10 if self.init else num_joint_iters
        if len(unvisit_list) > 2:
            self.last_visit = cur_idx
            for _ in range(unvisit_factor):
                sub = list(np.random.choice(unvisit_list, self.mapping_window_size))
                rays_o, rays_d, color, depth = self._ray_batch(sub, unvisit_frame, self.mapping_pixels // len(sub))
                if len(rays_o) < 100:
                    continue
                self.optimize_map(rays_o, rays_d, color, depth, self.optimizer, 1)
        for _ in range(num_joint_iters):
            rays_o, rays_d, color, depth = self._ray_batch(visit_list, visit_frame,
                                                           self.mapping_pixels // len(visit_list))
            if len(rays_o) < 100:
                continue
            self.optimize_map(rays_o, rays_d, color, depth, self.optimizer, 1)
        self.reload_map += 1                                # tells the mesher the map changed
        self.init = False
