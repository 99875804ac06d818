#!/usr/bin/env python3
"""Render one frame end to end with the drop-in classes and write it as a PNG (what run.py --type evaluate does
per test view: rays from the camera pose, Renderer.render, evaluator image dump).

    python examples/render_frame.py --ckpt tests/golden/synthetic_ckpt.pth --angle 40 --res 400 --out /tmp/nerf_out
    python examples/render_frame.py --precision f16      # BASELINE config 5 arithmetic

Needs an MI355X and the built library (python -c "import __graft_entry__ as g; g.build()")."""
import argparse
import math
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import nerf_replication_amd as nerf  # noqa: E402


def camera_pose(angle_deg, radius=4.0311, elevation_deg=30.0):
    """Camera on the upper hemisphere looking at the origin (Blender convention: -z forward, y up), [4,4] c2w."""
    th, ph = math.radians(angle_deg), math.radians(elevation_deg)
    eye = torch.tensor([radius * math.cos(ph) * math.cos(th), radius * math.cos(ph) * math.sin(th), radius * math.sin(ph)])
    fwd = -eye / eye.norm()
    right = torch.linalg.cross(fwd, torch.tensor([0.0, 0.0, 1.0])); right = right / right.norm()
    up = torch.linalg.cross(right, fwd)
    c2w = torch.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, up, -fwd, eye
    return c2w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", default=os.path.join(REPO, "tests", "golden", "synthetic_ckpt.pth"))
    ap.add_argument("--angle", type=float, default=40.0)
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--precision", default="f32", choices=["f32", "f16", "f32x"])
    ap.add_argument("--out", default="nerf_out")
    args = ap.parse_args()

    net = nerf.Network()
    nerf.load_network(net, args.ckpt)
    net = net.cuda().eval()
    net.precision = args.precision
    renderer = nerf.Renderer(net)
    rays_o, rays_d = nerf.generate_rays(camera_pose(args.angle), args.res, args.res, 0.6911112070083618, "cuda")
    with torch.no_grad():
        renderer.render({"rays_o": rays_o[None, :1024], "rays_d": rays_d[None, :1024]})       # warm-up (weight packing)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rgb, depth = renderer.render({"rays_o": rays_o[None], "rays_d": rays_d[None]})
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("{}x{} frame, {}: {:.1f} ms, {:.0f} rays/s".format(args.res, args.res, args.precision, dt * 1e3, rays_o.shape[0] / dt))
    ev = nerf.Evaluator(result_dir=args.out)                    # writes <out>/images/view000_{pred,gt}.png
    ev.evaluate((rgb, depth), {"colors": rgb[None], "H": torch.tensor(args.res), "W": torch.tensor(args.res), "id": torch.tensor(0)})
    print("wrote", os.path.join(args.out, "images", "view000_pred.png"))


if __name__ == "__main__":
    main()
