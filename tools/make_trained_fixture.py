#!/usr/bin/env python3
"""Produce tests/golden/trained_ckpt.pth: a network TRAINED by this build (developer tool, GPU box).

    python tools/make_trained_fixture.py --steps 3000 --out gpurun_out/trained_ckpt.pth

The parity scenes of oracle.WEIGHT_FAMILIES are synthetic fields.  A trained NeRF has Adam-shaped weights; with no dataset
or latest.pth offline, the closest thing is to train one: a freshly initialised network (nn.Linear default init) is trained with
examples/train_synthetic.py's loop (the reference's step: 4096 random rays, MSE on the fine RGB, clip 40, Adam 5e-4,
exponential decay; exact-fp32 path) against renders of the "sharp" family scene, and its state_dict is stored in the
reference's {"net": ...} layout.  oracle/gen_golden.py then renders it with the REAL reference (render_family_trained.npz)."""
import argparse
import os
import subprocess
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import nerf_oracle as orc  # noqa: E402  (developer tool: builds the teacher scene)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "trained_ckpt.pth"))
    args = ap.parse_args()
    base = torch.load(os.path.join(REPO, "tests", "golden", "synthetic_ckpt.pth"), weights_only=True)["net"]
    work = os.path.join(REPO, "gpurun_out", "trained_work")
    os.makedirs(work, exist_ok=True)
    torch.save({"net": orc.weight_family(base, "sharp"), "epoch": 0}, os.path.join(work, "teacher.pth"))
    subprocess.run([sys.executable, os.path.join(REPO, "examples", "train_synthetic.py"), "--teacher", os.path.join(work, "teacher.pth"),
                    "--steps", str(args.steps), "--precision", "f32", "--views", "24", "--out", os.path.join(work, "run")], check=True)
    ck = torch.load(os.path.join(work, "run", "latest.pth"), weights_only=True)
    torch.save({"net": {k: v.cpu().contiguous() for k, v in ck["net"].items()}, "epoch": int(ck.get("epoch", 0))}, args.out)
    print("wrote", args.out, os.path.getsize(args.out), "bytes")


if __name__ == "__main__":
    main()
