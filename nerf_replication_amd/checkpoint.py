"""Checkpoint files in the reference's layout, so a `latest.pth` written by the reference
loads here and the other way round.  Behaviour follows src/utils/net_utils.py: `load_network` :346-379
(model_dir may be a directory -- `latest.pth` wins, else the highest epoch number -- or a file; returns the epoch
to resume from), `load_model` :288-320 and `save_model` :323-343 ({"net","optim","scheduler","recorder","epoch"},
`latest.pth` or `<epoch>.pth`, at most five numbered files kept).

Files are read with `torch.load(..., weights_only=True)`: nothing in a checkpoint is executed.
"""
import os

import torch


def _numbered(model_dir):
    out = []
    for name in os.listdir(model_dir):
        stem, ext = os.path.splitext(name)
        if ext == ".pth" and stem.isdigit():
            out.append(int(stem))
    return out


def _resolve(model_dir, epoch):
    """Path of the checkpoint `epoch` (-1: latest.pth if present, else the highest number), or None."""
    if not os.path.exists(model_dir):
        return None
    if not os.path.isdir(model_dir):
        return model_dir
    names = os.listdir(model_dir)
    nums = _numbered(model_dir)
    if not nums and "latest.pth" not in names:
        return None
    if epoch == -1:
        stem = "latest" if "latest.pth" in names else str(max(nums))
    else:
        stem = str(epoch)
    return os.path.join(model_dir, stem + ".pth")


def load_network(net, model_dir, resume=True, epoch=-1, strict=True):
    """Weights only (run.py / evaluation).  Returns the epoch to resume from (0 if nothing was loaded)."""
    if not resume:
        return 0
    path = _resolve(model_dir, epoch)
    if path is None:
        print("pretrained model does not exist")
        return 0
    print("load model: {}".format(path))
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    net.load_state_dict(ckpt["net"], strict=strict)
    return ckpt["epoch"] + 1 if "epoch" in ckpt else 0


def load_model(net, optim, scheduler, recorder, model_dir, resume=True, epoch=-1):
    """Weights + optimizer / scheduler / recorder state (train.py).  `scheduler` and `recorder` may be None."""
    if not resume and os.path.isdir(model_dir):
        for n in _numbered(model_dir):
            os.remove(os.path.join(model_dir, "{}.pth".format(n)))
        if os.path.exists(os.path.join(model_dir, "latest.pth")):
            os.remove(os.path.join(model_dir, "latest.pth"))
    path = _resolve(model_dir, epoch) if os.path.isdir(model_dir) else None
    if path is None:
        return 0
    print("load model: {}".format(path))
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    net.load_state_dict(ckpt["net"])
    if "optim" not in ckpt:
        return 0
    optim.load_state_dict(ckpt["optim"])
    if scheduler is not None and "scheduler" in ckpt:
        scheduler.load_state_dict(ckpt["scheduler"])
    if recorder is not None and "recorder" in ckpt:
        recorder.load_state_dict(ckpt["recorder"])
    return ckpt["epoch"] + 1


def save_model(net, optim, scheduler, recorder, model_dir, epoch, last=False):
    os.makedirs(model_dir, exist_ok=True)
    model = {
        "net": net.state_dict(),
        "optim": optim.state_dict(),
        "scheduler": scheduler.state_dict() if scheduler is not None else {},
        "recorder": recorder.state_dict() if recorder is not None else {},
        "epoch": epoch,
    }
    torch.save(model, os.path.join(model_dir, "latest.pth" if last else "{}.pth".format(epoch)))
    nums = _numbered(model_dir)             # keep at most five numbered checkpoints
    if len(nums) > 5:
        os.remove(os.path.join(model_dir, "{}.pth".format(min(nums))))
