"""Renderer with the reference's plugin surface (src/models/nerf/renderer/volume_renderer.py:8-24,
:290-432): ``Renderer(net).render(batch) -> (rgb [B*N,3], depth [B*N])``, executed by the HIP
kernels behind include/nerf_mi355x.h.

Loadable through the reference's loader (src/models/nerf/renderer/make_renderer.py:4-8), see
INTEGRATION.md.  Hyper-parameters follow the reference's *effective* behaviour:
N_samples 64, N_importance 128, near/far 2/6, deterministic sampling, white background.
"""
import os
import sys

import torch


def _sibling(name):
    """Import a sibling module of this package by its absolute name.  The reference loads this file by PATH
    (imp.load_source(cfg.*_module, cfg.*_path), make_network.py:4-8 / make_renderer.py:4-8), under whatever dotted name
    the YAML gives and with the CWD -- not necessarily sys.path -- holding the package directory."""
    import importlib
    try:
        return importlib.import_module("nerf_replication_amd." + name)
    except ModuleNotFoundError as exc:
        if exc.name != "nerf_replication_amd":
            raise
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        return importlib.import_module("nerf_replication_amd." + name)


_lib = _sibling("_lib")


def _reference_cfg():
    mod = sys.modules.get("src.config")
    return getattr(mod, "cfg", None) if mod is not None else None


class Renderer:
    def __init__(self, net):
        cfg = _reference_cfg()
        self.net = net
        # same getattr-on-top-level-cfg reads as volume_renderer.py:14-24 (they normally miss -> defaults)
        self.N_samples = getattr(cfg, "N_samples", 64)
        self.chunk_size = getattr(cfg, "chunk_size", 1024)
        self.white_bkgd = getattr(cfg, "white_bkgd", True)
        self.N_importance = getattr(cfg, "N_importance", 128)
        self.sample_size = 64
        self.rays_size = 160000
        self.task = getattr(cfg, "task", "test")
        self.perturb = bool(getattr(cfg, "perturb", True)) if self.task == "train" else False
        self.fast_sampling = getattr(cfg, "fast_sampling", False)
        self.weights_threshold = getattr(cfg, "weights_threshold", 0.25)
        self.t_near, self.t_far = 2.0, 6.0            # default args of the stratified sampler (:27)
        self.device = None
        self._tables = {}
        self._workspace = None
        if self.N_samples != _lib.N_SAMPLES or self.N_importance not in (0, _lib.N_IMPORTANCE):
            raise ValueError("HIP renderer is built for N_samples=64 and N_importance in {0,128}")
        if self.perturb or self.task == "train":
            raise NotImplementedError("stochastic (task=='train') sampling is unreachable from the reference's "
                                      "configs (SURVEY F2) and is not built")

    # host-built, bit-sensitive tables: torch.linspace on the CPU, then copied
    def _get_tables(self, dev):
        tabs = self._tables.get(dev)
        if tabs is None:
            t_c = torch.linspace(self.t_near, self.t_far, self.N_samples).to(dev)
            u = torch.linspace(0.0, 1.0, steps=_lib.N_IMPORTANCE).to(dev)
            tabs = self._tables[dev] = (t_c, u)
        return tabs

    def _get_workspace(self, nbytes, dev):
        ws = self._workspace
        if ws is None or ws.device != dev or ws.numel() < nbytes:
            self._workspace = None
            ws = self._workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return ws

    def render(self, batch):
        rays_o, rays_d = batch["rays_o"], batch["rays_d"]
        self.device = dev = rays_o.device
        if dev.type != "cuda":
            raise _lib.NerfLibraryError("Renderer.render needs rays on a GPU: the render path is HIP-only")
        lib = _lib.load()
        B, N, _ = rays_o.shape
        n = B * N
        o = rays_o.detach().reshape(n, 3).to(torch.float32).contiguous()
        d = rays_d.detach().reshape(n, 3).to(torch.float32).contiguous()
        if torch.is_grad_enabled() and getattr(self.net, "training", False) and \
                any(p.requires_grad for p in self.net.parameters()):
            # training call (trainers/nerf.py:27 under trainer.py:53-60): forward with activation save,
            # backward through the adjoint HIP kernels (training.py)
            render_with_grad = _sibling("training").render_with_grad
            if n == 0:
                return torch.empty((0, 3), device=dev), torch.empty((0,), device=dev)
            return render_with_grad(self, o, d)
        t_c, u = self._get_tables(dev)
        pk_c = self.net.packed("")
        pk_f = self.net.packed("fine") if self.N_importance > 0 else None
        rgb = torch.empty((n, 3), dtype=torch.float32, device=dev)
        depth = torch.empty((n,), dtype=torch.float32, device=dev)
        if n == 0:
            return rgb, depth
        fast = int(bool(self.fast_sampling) and self.N_importance > 0)
        nbytes = int(lib.nerf_render_workspace_bytes(n, self.N_importance, fast))
        ws = self._get_workspace(nbytes, dev)
        prec = _lib.PRECISIONS[getattr(self.net, "precision", "f32")]
        with torch.cuda.device(dev):
            _lib.check(lib.nerf_render_forward(
                _lib.ptr(o), _lib.ptr(d), n, pk_c.data_ptr(), pk_f.data_ptr() if pk_f is not None else None,
                _lib.ptr(t_c), _lib.ptr(u),
                int(self.N_importance), int(bool(self.white_bkgd)), prec, fast, float(self.weights_threshold),
                ws.data_ptr(), ws.numel(),
                _lib.ptr(rgb), _lib.ptr(depth), _lib.stream_of(dev)), "nerf_render_forward")
        return rgb, depth
