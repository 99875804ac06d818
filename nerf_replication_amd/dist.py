"""Multi-GPU frame rendering: rays shard embarrassingly (no cross-ray op anywhere in
Renderer.render, volume_renderer.py:321/:386 already treat ray blocks independently), one
all_gather of packed [n_local,4] (rgb+depth) per frame over RCCL/xGMI reassembles the image.

New functionality of the build (the reference has no multi-GPU inference path).
One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm, "gloo" for CPU tests.
"""
import os

import torch
import torch.distributed as dist


def _force_collective():
    """NERF_DIST_FORCE_COLLECTIVE=1: take the collective path even in a one-rank group, so that a 1-GPU box can drive the very
    RCCL calls an N-rank run makes (tests/test_gpu_parity.py::test_rccl_single_rank_collectives); results are unchanged."""
    return dist.is_initialized() and os.environ.get("NERF_DIST_FORCE_COLLECTIVE") == "1"


def shard_bounds(n_rays: int, rank: int, world: int):
    """Contiguous ray range [lo, hi) of `rank`: horizontal image tiles, sizes differ by at most 1 row
    of padding; every rank's slot in the gather buffer is ceil(n/world) rays."""
    per = (n_rays + world - 1) // world
    lo = min(rank * per, n_rays)
    return lo, min(lo + per, n_rays), per


def render_shard(renderer, rays_o_local, rays_d_local, n_total, group=None, events=None):
    """This rank's contiguous shard ONLY: rays_*_local [hi-lo,3] are rows [lo,hi) = shard_bounds(n_total, rank, world)
    of the frame (e.g. from generate_rays(pixel_begin=lo, n_pixels=hi-lo)); no rank ever materialises the whole
    frame's rays.  Renders them, then one all_gather returns the full (rgb [n_total,3], depth [n_total]) on every
    rank.  `events`, if a list, receives a (start, end) pair of device events around the local render (CUDA only)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi, per = shard_bounds(n_total, rank, world)
    if rays_o_local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank}/{world}: expected {hi - lo} local rays (rows [{lo},{hi}) of {n_total}), "
                         f"got {rays_o_local.shape[0]}")
    timed = events is not None and rays_o_local.is_cuda
    if timed:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    rgb, depth = renderer.render({"rays_o": rays_o_local[None], "rays_d": rays_d_local[None]})
    if timed:
        ev[1].record()
        events.append(ev)
    if world == 1 and not _force_collective():
        return rgb, depth
    packed = torch.zeros((per, 4), dtype=torch.float32, device=rgb.device)
    packed[: hi - lo, :3] = rgb
    packed[: hi - lo, 3] = depth
    full = torch.empty((world * per, 4), dtype=torch.float32, device=rgb.device)
    dist.all_gather_into_tensor(full, packed, group=group)
    return full[:n_total, :3].contiguous(), full[:n_total, 3].contiguous()


def render_sharded(renderer, rays_o, rays_d, group=None):
    """Convenience over render_shard for a caller that already holds the FULL frame's rays [N,3] on every rank:
    slices this rank's rows and gathers.  With world_size 1 this is exactly renderer.render."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = rays_o.shape[0]
    lo, hi, _ = shard_bounds(n, rank, world)
    return render_shard(renderer, rays_o[lo:hi], rays_d[lo:hi], n, group)


def _shared_flat_view(grads):
    """If the gradient tensors are back-to-back contiguous views of ONE storage (in this order), return a 1-D
    tensor over exactly that range (no copy); otherwise None."""
    g0 = grads[0]
    if g0.dtype != torch.float32 or not g0.is_contiguous():
        return None
    expect = g0.storage_offset()
    for g in grads:
        if (g.dtype != torch.float32 or not g.is_contiguous() or g.storage_offset() != expect
                or g.untyped_storage().data_ptr() != g0.untyped_storage().data_ptr()):
            return None
        expect += g.numel()
    total = expect - g0.storage_offset()
    return g0.new_empty(0).set_(g0.untyped_storage(), g0.storage_offset(), (total,))


def allreduce_gradients(params, group=None):
    """Data-parallel training (what DDP does for the reference in trainer.py:16-21): average the 48 parameter
    gradients over the ranks with ONE all_reduce of a flat 4.77 MB buffer (2 x 595 844 fp32), then scatter
    the averages back into the .grad tensors.  No-op for a single process."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not _force_collective()):
        return
    params = [p for p in params if p.grad is not None]
    if not params:
        return
    flat = _shared_flat_view([p.grad for p in params])
    if flat is not None:          # training.RenderFunction hands out views of one buffer: reduce it in place
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(dist.get_world_size(group))
        return
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(dist.get_world_size(group))
    off = 0
    for p in params:
        n = p.grad.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n
