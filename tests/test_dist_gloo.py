"""World-size-2 gloo test of the ray-shard + all_gather frame path (runs on CPU).  The renderer
is replaced by a stub that calls the CPU oracle (test-only): what is under test is the sharding,
padding and gather logic of nerf_replication_amd.dist, which is device-agnostic."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO, GOLDEN


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _OracleRenderer:
    def __init__(self, sd):
        import nerf_oracle
        self.orc, self.sd = nerf_oracle, sd

    def render(self, batch):
        with torch.no_grad():
            return self.orc.render(self.sd, batch["rays_o"], batch["rays_d"], n_importance=0)


def _worker(rank, world, port, n_rays, out_dir):
    import sys
    for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerf_replication_amd.dist import render_sharded, shard_bounds
        import nerf_oracle as orc
        sd = torch.load(os.path.join(GOLDEN, "synthetic_ckpt.pth"), weights_only=True)["net"]
        ids = torch.randperm(800 * 800, generator=torch.Generator().manual_seed(4))[:n_rays]
        o, d = orc.pinhole_rays(800, 800, orc.camera_pose(10.0), pixel_ids=ids)
        ren = _OracleRenderer(sd)
        rgb, dep = render_sharded(ren, o, d)
        assert rgb.shape == (n_rays, 3) and dep.shape == (n_rays,)
        lo, hi, per = shard_bounds(n_rays, rank, world)
        assert 0 <= lo <= hi <= n_rays and hi - lo <= per
        torch.save({"rgb": rgb, "dep": dep, "lo": lo, "hi": hi}, os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _run(world, n_rays, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_rays, str(tmp_path)), nprocs=world, join=True)
    return [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]


def test_two_rank_frame_equals_single_process(tmp_path, oracle, synthetic_sd):
    n_rays = 101                                   # odd: shards of 51 and 50, one padded slot
    outs = _run(2, n_rays, tmp_path)
    ids = torch.randperm(800 * 800, generator=torch.Generator().manual_seed(4))[:n_rays]
    o, d = oracle.pinhole_rays(800, 800, oracle.camera_pose(10.0), pixel_ids=ids)
    with torch.no_grad():
        ref_rgb, ref_dep = oracle.render(synthetic_sd, o[None], d[None], n_importance=0)
    assert (outs[0]["lo"], outs[0]["hi"], outs[1]["lo"], outs[1]["hi"]) == (0, 51, 51, 101)
    for r in range(2):                             # every rank holds the whole, identical frame
        assert torch.equal(outs[r]["rgb"], outs[0]["rgb"]) and torch.equal(outs[r]["dep"], outs[0]["dep"])
        assert torch.allclose(outs[r]["rgb"], ref_rgb, atol=1e-6) and torch.allclose(outs[r]["dep"], ref_dep, atol=1e-5)


def test_shard_bounds_cover_every_ray_once():
    from nerf_replication_amd.dist import shard_bounds
    for n in (0, 1, 7, 8, 9, 640000, 2560000 + 3):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            assert all(hi - lo <= per for lo, hi, per in spans)


def _grad_worker(rank, world, port, out_dir, shared):
    import sys
    for p in (REPO,):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerf_replication_amd.dist import allreduce_gradients
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.zeros(s)) for s in ((256, 63), (256,), (3, 128), (1,))]
        if shared:          # the layout training.RenderFunction produces: views of one zeroed buffer, in order
            flat = torch.zeros(sum(p.numel() for p in params) + 7)
            off = 3                                                  # a storage offset, and slack at the end
            for i, p in enumerate(params):
                p.grad = flat[off:off + p.numel()].view(p.shape)
                p.grad.fill_(float(rank + 1) * (i + 1))
                off += p.numel()
        else:
            for i, p in enumerate(params):
                p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        params.append(torch.nn.Parameter(torch.zeros(5)))            # no gradient: skipped
        from nerf_replication_amd.dist import _shared_flat_view
        assert (_shared_flat_view([p.grad for p in params[:-1]]) is not None) == shared
        allreduce_gradients(params)
        if shared:
            assert torch.all(flat[:3] == 0) and torch.all(flat[-4:] == 0)      # nothing outside the range touched
        torch.save([p.grad for p in params[:-1]], os.path.join(out_dir, f"g{rank}.pt"))
    finally:
        dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("shared", [False, True])
def test_gradient_allreduce_averages_over_ranks(tmp_path, shared):
    port = _free_port()
    mp.spawn(_grad_worker, args=(2, port, str(tmp_path), shared), nprocs=2, join=True)
    for r in range(2):
        grads = torch.load(os.path.join(tmp_path, f"g{r}.pt"))
        for i, g in enumerate(grads):
            assert torch.all(g == 1.5 * (i + 1))                     # mean of (1, 2) * (i + 1)


def _ws8_worker(rank, world, port, n_rays, out_dir):
    import sys
    for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerf_replication_amd.dist import render_shard, shard_bounds, allreduce_gradients
        import nerf_oracle as orc
        sd = torch.load(os.path.join(GOLDEN, "synthetic_ckpt.pth"), weights_only=True)["net"]
        # every rank builds ONLY its own tile of the frame's rays (what bench.py does with generate_rays(pixel_begin, n_pixels))
        lo, hi, per = shard_bounds(n_rays, rank, world)
        ids = torch.randperm(800 * 800, generator=torch.Generator().manual_seed(6))[:n_rays][lo:hi]
        o, d = orc.pinhole_rays(800, 800, orc.camera_pose(25.0), pixel_ids=ids)
        assert o.shape[0] == hi - lo
        rgb, dep = render_shard(_OracleRenderer(sd), o, d, n_rays)
        assert rgb.shape == (n_rays, 3) and dep.shape == (n_rays,)
        try:                                             # a wrong local ray count is refused, not silently mis-gathered
            render_shard(_OracleRenderer(sd), o[:-1], d[:-1], n_rays)
            raise AssertionError("expected ValueError")
        except ValueError:
            pass
        # data-parallel gradient averaging over 8 ranks, the shared-buffer layout of training.RenderFunction
        params = [torch.nn.Parameter(torch.zeros(s)) for s in ((256, 63), (256,), (3, 128), (1,))]
        flat = torch.zeros(sum(p.numel() for p in params))
        off = 0
        for i, p in enumerate(params):
            p.grad = flat[off:off + p.numel()].view(p.shape)
            p.grad.fill_(float(rank) + 0.25 * i)
            off += p.numel()
        allreduce_gradients(params)
        torch.save({"rgb": rgb, "dep": dep, "lo": lo, "hi": hi, "grads": [p.grad.clone() for p in params]},
                   os.path.join(out_dir, f"w{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_eight_ranks_uneven_frame_and_gradient_allreduce(tmp_path, oracle, synthetic_sd):
    """SURVEY 8(e) at the node's real rank count: 8 ranks, a ray count NOT divisible by 8 (1003 = 7 x 126 + 121: the
    last slot is padded), every rank holding only its own rays; plus the 8-way gradient average."""
    n_rays, world = 1003, 8
    port = _free_port()
    mp.spawn(_ws8_worker, args=(world, port, n_rays, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"w{r}.pt")) for r in range(world)]
    assert [(o["lo"], o["hi"]) for o in outs] == [(126 * r, min(126 * (r + 1), n_rays)) for r in range(world)]
    ids = torch.randperm(800 * 800, generator=torch.Generator().manual_seed(6))[:n_rays]
    o, d = oracle.pinhole_rays(800, 800, oracle.camera_pose(25.0), pixel_ids=ids)
    with torch.no_grad():
        ref_rgb, ref_dep = oracle.render(synthetic_sd, o[None], d[None], n_importance=0)
    for r in range(world):
        assert torch.equal(outs[r]["rgb"], outs[0]["rgb"]) and torch.equal(outs[r]["dep"], outs[0]["dep"])
        for i, g in enumerate(outs[r]["grads"]):
            assert torch.all(g == 3.5 + 0.25 * i)                     # mean of rank 0..7
    assert torch.allclose(outs[0]["rgb"], ref_rgb, atol=1e-6) and torch.allclose(outs[0]["dep"], ref_dep, atol=1e-5)
