"""Child process of tests/test_gpu_parity.py::test_rccl_single_rank_collectives (test infrastructure).

ONE rank, backend "nccl" (= RCCL on ROCm), NERF_DIST_FORCE_COLLECTIVE=1: render_shard's all_gather_into_tensor and
allreduce_gradients' in-place flat all-reduce run through librccl.so on the one GPU of the test box -- the calls an N-rank
run makes, which gloo rehearsals never touch.  Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))


def main():
    import nerf_oracle as orc          # (rays only)
    import nerf_replication_amd as pkg
    from nerf_replication_amd.dist import allreduce_gradients, render_shard
    from nerf_replication_amd.training import render_with_grad
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    ck = torch.load(os.path.join(REPO, "tests", "golden", "synthetic_ckpt.pth"), weights_only=True)["net"]
    net = pkg.Network()
    net.load_state_dict(ck, strict=True)
    net = net.to(dev).eval()
    ren = pkg.Renderer(net)
    n = 4099                                                  # not a multiple of anything
    ids = torch.randperm(800 * 800, generator=torch.Generator().manual_seed(4))[:n]
    o, d = orc.pinhole_rays(800, 800, orc.camera_pose(40.0), pixel_ids=ids)
    o, d = o.to(dev), d.to(dev)
    out = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
    with torch.no_grad():
        os.environ["NERF_DIST_FORCE_COLLECTIVE"] = "0"
        rgb0, dep0 = render_shard(ren, o, d, n)
        os.environ["NERF_DIST_FORCE_COLLECTIVE"] = "1"
        rgb1, dep1 = render_shard(ren, o, d, n)               # packed [n,4] -> all_gather_into_tensor -> unpack
    out["render_bit_equal"] = bool(torch.equal(rgb0, rgb1) and torch.equal(dep0, dep1))
    # gradients: flat in-place all-reduce of the shared 4.77 MB buffer (sum over one rank / 1 = identity, bit for bit)
    net.train()
    target = torch.rand(256, 3, generator=torch.Generator().manual_seed(5)).to(dev)
    rgb, _ = render_with_grad(ren, o[:256].contiguous(), d[:256].contiguous())
    torch.nn.functional.mse_loss(rgb, target).backward()
    before = [p.grad.clone() for p in net.parameters()]
    allreduce_gradients(net.parameters())
    torch.cuda.synchronize()
    out["grads_bit_equal"] = all(torch.equal(a, p.grad) for a, p in zip(before, net.parameters()))
    # the copy path too (gradients that are NOT views of one buffer)
    for p in net.parameters():
        p.grad = p.grad.clone()
    allreduce_gradients(net.parameters())
    torch.cuda.synchronize()
    out["grads_copy_path_bit_equal"] = all(torch.equal(a, p.grad) for a, p in zip(before, net.parameters()))
    maps = open("/proc/self/maps").read()
    out["librccl_mapped"] = "librccl" in maps
    out["libnerf_mapped"] = "libnerf_mi355x.so" in maps
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
