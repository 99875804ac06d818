"""Checkpoint compatibility with the reference's file layout (src/utils/net_utils.py:288-379): CPU-only."""
import os

import pytest
import torch

from conftest import GOLDEN


@pytest.fixture(scope="module")
def amd_cpu():
    """The package itself imports without a GPU (only the kernels need one)."""
    import nerf_replication_amd
    return nerf_replication_amd


def test_load_network_reads_reference_layout(amd_cpu, tmp_path, synthetic_sd):
    net = amd_cpu.Network()
    # a file path, as run.py passes cfg.trained_model_dir when it is a file
    assert amd_cpu.load_network(net, os.path.join(GOLDEN, "synthetic_ckpt.pth")) == 1        # fixture has epoch 0
    for k, v in net.state_dict().items():
        assert torch.equal(v, synthetic_sd[k])
    assert amd_cpu.load_network(net, str(tmp_path / "missing")) == 0
    assert amd_cpu.load_network(net, os.path.join(GOLDEN, "synthetic_ckpt.pth"), resume=False) == 0


def test_save_model_rotation_and_resume(amd_cpu, tmp_path, synthetic_sd):
    from nerf_replication_amd.training import FusedAdam
    net = amd_cpu.Network()
    net.load_state_dict(synthetic_sd)
    opt = FusedAdam(net.parameters(), lr=3e-4)
    opt.step_count = 7
    opt.exp_avg[0].fill_(0.25); opt.exp_avg_sq[3].fill_(2.0)
    d = str(tmp_path / "ckpt")
    for ep in range(7):
        amd_cpu.save_model(net, opt, None, None, d, ep)
    assert sorted(os.listdir(d)) == ["2.pth", "3.pth", "4.pth", "5.pth", "6.pth"]             # five numbered files kept
    amd_cpu.save_model(net, opt, None, None, d, 9, last=True)
    assert "latest.pth" in os.listdir(d)

    net2 = amd_cpu.Network()
    opt2 = FusedAdam(net2.parameters())
    assert amd_cpu.load_model(net2, opt2, None, None, d) == 10                                # latest.pth wins: epoch 9 + 1
    assert opt2.step_count == 7 and opt2.lr == 3e-4
    assert torch.all(opt2.exp_avg[0] == 0.25) and torch.all(opt2.exp_avg_sq[3] == 2.0)
    assert amd_cpu.load_model(net2, opt2, None, None, d, epoch=4) == 5
    assert amd_cpu.load_network(net2, d, epoch=-1) == 10


def test_fused_adam_state_dict_interoperates_with_torch_adam(amd_cpu, synthetic_sd):
    from nerf_replication_amd.training import FusedAdam
    net = amd_cpu.Network()
    net.load_state_dict(synthetic_sd)
    ref = torch.optim.Adam(net.parameters(), lr=5e-4, eps=1e-8)
    for p in net.parameters():
        p.grad = torch.full_like(p, 1e-3)
    ref.step(); ref.step()
    fused = FusedAdam(net.parameters())
    fused.load_state_dict(ref.state_dict())                       # torch -> fused
    assert fused.step_count == 2
    for i, p in enumerate(net.parameters()):
        assert torch.equal(fused.exp_avg[i], ref.state[p]["exp_avg"])
    ref2 = torch.optim.Adam(net.parameters(), lr=1.0)
    ref2.load_state_dict(fused.state_dict())                      # fused -> torch
    assert ref2.param_groups[0]["lr"] == 5e-4
    for p in net.parameters():
        assert torch.equal(ref2.state[p]["exp_avg_sq"], ref.state[p]["exp_avg_sq"])
        assert float(ref2.state[p]["step"]) == 2.0


def test_fused_adam_is_driven_by_reference_style_schedulers(amd_cpu, synthetic_sd):
    """The reference builds ExponentialLR / MultiStepLR as torch _LRScheduler subclasses over its optimizer
    (src/utils/optimizer/lr_scheduler.py:52-79, src/train/scheduler.py): that requires a real torch Optimizer
    with param_groups.  Same formula restated here: lr = base_lr * gamma ** (epoch / decay_epochs)."""
    import warnings
    from nerf_replication_amd.training import FusedAdam

    class ExponentialLR(torch.optim.lr_scheduler._LRScheduler):
        def __init__(self, optimizer, decay_epochs, gamma=0.1, last_epoch=-1):
            self.decay_epochs, self.gamma = decay_epochs, gamma
            super().__init__(optimizer, last_epoch)

        def get_lr(self):
            return [base_lr * self.gamma ** (self.last_epoch / self.decay_epochs) for base_lr in self.base_lrs]

    net = amd_cpu.Network()
    opt = FusedAdam(net.parameters(), lr=5e-4)
    assert isinstance(opt, torch.optim.Optimizer) and len(opt.param_groups) == 1 and len(opt.params) == 48
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                 # "scheduler.step() before optimizer.step()": no GPU here
        sched = ExponentialLR(opt, decay_epochs=500, gamma=0.1)
        for _ in range(250):
            sched.step()
    assert abs(opt.lr - 5e-4 * 0.1 ** 0.5) < 1e-12
    assert abs(FusedAdam.exponential_lr(5e-4, 250) - opt.lr) < 1e-15
    sd = sched.state_dict()
    assert sd["last_epoch"] == 250


def test_png_writer_round_trip(amd_cpu, tmp_path):
    """evaluator.write_png (the evaluator's image dump, evaluators/nerf.py:50-61): decode the file again."""
    import struct, zlib
    from nerf_replication_amd.evaluator import write_png
    img = (torch.rand(5, 7, 3, generator=torch.Generator().manual_seed(1)) * 255).to(torch.uint8)
    path = str(tmp_path / "x.png")
    write_png(path, img)
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, {}
    while pos < len(data):
        n, tag = struct.unpack(">I", data[pos:pos + 4])[0], data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xffffffff
        chunks[tag] = body
        pos += 12 + n
    W, H, depth, ctype = struct.unpack(">IIBB", chunks[b"IHDR"][:10])
    assert (W, H, depth, ctype) == (7, 5, 8, 2)
    raw = zlib.decompress(chunks[b"IDAT"])
    rows = torch.frombuffer(bytearray(raw), dtype=torch.uint8).reshape(5, 1 + 7 * 3)
    assert torch.all(rows[:, 0] == 0) and torch.equal(rows[:, 1:].reshape(5, 7, 3), img)
