"""Network / NeRF modules with the reference's constructor, attributes, forward signature and
state_dict keys (src/models/nerf/network.py:9-74, :126-258), backed by the fused HIP MLP.

Loadable through the reference's plugin loader (src/models/make_network.py:4-8:
``imp.load_source(cfg.network_module, cfg.network_path).Network()``), see INTEGRATION.md.
"""
import ctypes
import os
import sys

import torch
import torch.nn as nn


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
    """The reference's global yacs cfg when running inside it (network.py:6), else None."""
    mod = sys.modules.get("src.config")
    return getattr(mod, "cfg", None) if mod is not None else None


class NeRF(nn.Module):
    """Parameter container with the layer names/shapes of network.py:22-47.  The arithmetic of
    network.py:49-74 lives in csrc/nerf_mlp_f32.hip.inc; this module only owns the tensors."""

    def __init__(self, D=8, W=256, input_ch=63, input_ch_views=27, skips=(4,), use_viewdirs=True):
        super().__init__()
        if (D, W, input_ch, input_ch_views, tuple(skips), bool(use_viewdirs)) != (8, 256, 63, 27, (4,), True):
            raise ValueError("the HIP kernels are built for the lego.yaml architecture: D=8, W=256, skips=[4], "
                             "xyz freq 10, dir freq 4, use_viewdirs=True")
        self.D, self.W, self.input_ch, self.input_ch_views = D, W, input_ch, input_ch_views
        self.skips, self.use_viewdirs = list(skips), True
        self.pts_linears = nn.ModuleList(
            [nn.Linear(input_ch, W)]
            + [nn.Linear(W + input_ch, W) if i in self.skips else nn.Linear(W, W) for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
        self.feature_linear = nn.Linear(W, W)
        self.alpha_linear = nn.Linear(W, 1)
        self.rgb_linear = nn.Linear(W // 2, 3)

    def ordered_params(self):
        """The 24 tensors in the order nerf_pack_model expects (= state_dict order)."""
        mods = list(self.pts_linears) + [self.views_linears[0], self.feature_linear, self.alpha_linear, self.rgb_linear]
        out = []
        for m in mods:
            out += [m.weight, m.bias]
        return out

    def forward(self, x):
        raise RuntimeError("NeRF.forward on an embedded chunk is fused into Network.forward (HIP); "
                           "call Network.forward(inputs, viewdirs, valid_mask, model)")


class Network(nn.Module):
    def __init__(self):
        super().__init__()
        cfg = _reference_cfg()
        ta = getattr(cfg, "task_arg", None) if cfg is not None else None
        self.N_samples = getattr(ta, "N_samples", 64)
        self.N_importance = getattr(ta, "N_importance", 128)
        self.chunk = getattr(ta, "chunk_size", 512)          # kept for API parity; the fused kernel has no chunk loop
        self.batch_size = getattr(ta, "N_rays", 1024)
        self.white_bkgd = getattr(ta, "white_bkgd", 1)
        self.use_viewdirs = bool(getattr(ta, "use_viewdirs", True))
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.sample_size = 64
        self.rays_size = 160000
        net = getattr(cfg, "network", None) if cfg is not None else None
        xyz_f = getattr(getattr(net, "xyz_encoder", None), "freq", 10)
        dir_f = getattr(getattr(net, "dir_encoder", None), "freq", 4)
        nerf_cfg = getattr(net, "nerf", None)
        D, W = getattr(nerf_cfg, "D", 8), getattr(nerf_cfg, "W", 256)
        skips = tuple(getattr(nerf_cfg, "skips", [4]))
        self.input_ch, self.input_ch_views = 3 + 6 * xyz_f, 3 + 6 * dir_f
        self.embed_fn = lambda x: positional_encoding(x, xyz_f)
        self.embeddirs_fn = lambda x: positional_encoding(x, dir_f)
        mk = lambda: NeRF(D=D, W=W, input_ch=self.input_ch, input_ch_views=self.input_ch_views,
                          skips=skips, use_viewdirs=self.use_viewdirs)
        self.model = mk()
        self.model_fine = mk()
        self.precision = "f32"
        self._packed = {}       # tag -> (key, tensor)

    # ---- packed weight stream (csrc/nerf_layout.h) -------------------------------------------
    def packed(self, model=""):
        """Device byte tensor with the kernel's weight stream (for self.precision: "f32" exact fp32 MFMA,
        "f16" fp16 activations / fp32 accumulate) of the coarse ("") or fine model; repacked
        on the device whenever a parameter's storage or version changed (load_state_dict, .to(),
        optimizer.step())."""
        tag = "fine" if model == "fine" else ""
        sub = self.model_fine if tag == "fine" else self.model
        params = sub.ordered_params()
        dev = params[0].device
        if dev.type != "cuda":
            raise _lib.NerfLibraryError("Network parameters are on the CPU: call .cuda() first; the render path is "
                                        "HIP-only (no CPU fallback)")
        prec = _lib.PRECISIONS[self.precision]
        key = (prec,) + tuple((p.data_ptr(), p._version) for p in params)
        hit = self._packed.get(tag)
        if hit is not None and hit[0] == key:
            return hit[1]
        lib = _lib.load()
        out = torch.empty(_lib.packed_model_bytes(prec), dtype=torch.uint8, device=dev)
        srcs = [p.detach().contiguous() for p in params]
        arr = (ctypes.c_void_p * 24)(*[_lib.ptr(t) for t in srcs])
        with torch.cuda.device(dev):
            _lib.check(lib.nerf_pack_model(arr, out.data_ptr(), prec, _lib.stream_of(dev)), "nerf_pack_model")
        self._packed[tag] = (key, out)
        return out

    def forward(self, inputs, viewdirs, valid_mask, model=""):
        """inputs [n,s,3], viewdirs [n,3], valid_mask BoolTensor[n,s] | None, model "" | "fine"
        -> raw [n,s,4] = (r,g,b,sigma) pre-activation (network.py:199-258).

        Differentiable like the reference's: with autograd on and either `inputs.requires_grad` or a `.train()` network
        whose parameters require grad (or `viewdirs.requires_grad`), the forward runs the SAVE-mode fused kernel and backward() the adjoint HIP
        kernels (nerf_mlp_backward_points, nerf_viewdirs_backward): gradients w.r.t. the 24 tensors of the selected
        sub-model, `inputs` and `viewdirs`."""
        sub = self.model_fine if model == "fine" else self.model
        if torch.is_grad_enabled() and (inputs.requires_grad or viewdirs.requires_grad or
                                        (self.training and any(p.requires_grad for p in sub.parameters()))):
            return self._forward_with_grad(inputs, viewdirs, valid_mask, model, sub)
        lib = _lib.load()
        dev = inputs.device
        n, s = inputs.shape[0], inputs.shape[1]
        packed = self.packed(model)
        prec = _lib.PRECISIONS[self.precision]
        if valid_mask is None:
            pts = inputs.detach().to(torch.float32).contiguous()
            dirs = viewdirs.detach().to(torch.float32).contiguous()
            raw = torch.empty((n, s, 4), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.nerf_mlp_forward(_lib.ptr(pts), _lib.ptr(dirs), n, s, packed.data_ptr(),
                                                _lib.ptr(raw), prec, _lib.stream_of(dev)), "nerf_mlp_forward")
            return raw
        # ESS/ERT path (network.py:207-214, :238-253): run only the valid points, zeros elsewhere
        flat = valid_mask.reshape(-1)
        pts = inputs.detach().reshape(-1, 3)[flat].to(torch.float32).contiguous()
        dirs = viewdirs.detach()[:, None].expand(n, s, 3).reshape(-1, 3)[flat].to(torch.float32).contiguous()
        m = pts.shape[0]
        raw_valid = torch.empty((m, 1, 4), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.nerf_mlp_forward(_lib.ptr(pts), _lib.ptr(dirs), m, 1, packed.data_ptr(),
                                            _lib.ptr(raw_valid), prec, _lib.stream_of(dev)), "nerf_mlp_forward")
        out = torch.zeros((n * s, 4), dtype=torch.float32, device=dev)
        out[flat] = raw_valid.reshape(m, 4)
        return out.reshape(n, s, 4)

    def _forward_with_grad(self, inputs, viewdirs, valid_mask, model, sub):
        if self.precision not in ("f32", "f32x"):
            raise NotImplementedError("training runs on the fp32-accurate paths: precision 'f32' or 'f32x'")
        n, s = inputs.shape[0], inputs.shape[1]
        params = tuple(sub.ordered_params())
        if valid_mask is None:
            return _MlpFunction.apply(self, model, inputs.to(torch.float32), viewdirs.to(torch.float32), *params)
        # masked (network.py:207-214, :238-253): the valid points as m one-sample rays; the scatter back into zeros is
        # index plumbing whose adjoint (a gather) torch provides
        flat = valid_mask.reshape(-1)
        pts = inputs.to(torch.float32).reshape(-1, 3)[flat][:, None, :]
        dirs = viewdirs.to(torch.float32)[:, None].expand(n, s, 3).reshape(-1, 3)[flat]      # (its adjoint sums over samples)
        out = torch.zeros((n * s, 4), dtype=torch.float32, device=inputs.device)
        if pts.shape[0] > 0:
            out = out.index_put((flat.nonzero(as_tuple=True)[0],), _MlpFunction.apply(self, model, pts, dirs, *params)[:, 0])
        return out.reshape(n, s, 4)


class _MlpFunction(torch.autograd.Function):
    """Network.forward (network.py:199-258) under autograd: SAVE-mode fused forward, adjoint HIP kernels backward.
    torch.autograd only routes the gradients; every number comes out of a HIP kernel."""

    @staticmethod
    def forward(ctx, net, model, inputs, viewdirs, *params):
        lib = _lib.load()
        dev = inputs.device
        n, s = inputs.shape[0], inputs.shape[1]
        prec = _lib.PRECISIONS[net.precision]
        pts = inputs.detach().contiguous()
        dirs = viewdirs.detach().contiguous()
        raw = torch.empty((n, s, 4), dtype=torch.float32, device=dev)
        save = torch.empty(max(1, int(lib.nerf_train_save_floats(n * s))), dtype=torch.float32, device=dev)
        if n * s > 0:
            packed = net.packed(model)
            with torch.cuda.device(dev):
                _lib.check(lib.nerf_mlp_forward_points_save(_lib.ptr(pts), _lib.ptr(dirs), n, s, packed.data_ptr(),
                                                            _lib.ptr(raw), _lib.ptr(save), prec, _lib.stream_of(dev)),
                           "nerf_mlp_forward_points_save")
        ctx.prec, ctx.shape = prec, (n, s)
        # the parameters go through save_for_backward: an in-place update between forward and backward (optimizer.step(),
        # load_state_dict) then raises autograd's version-counter error instead of pairing new weights with old activations
        ctx.save_for_backward(pts, save, dirs, *params)
        return raw

    @staticmethod
    def backward(ctx, g_raw):
        lib = _lib.load()
        pts, save, dirs, *params = ctx.saved_tensors
        (n, s), prec = ctx.shape, ctx.prec
        dev = pts.device
        st = _lib.stream_of(dev)
        flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
        grads, off = [], 0
        for p in params:
            grads.append(flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
        g_pts = torch.zeros((n, s, 3), dtype=torch.float32, device=dev) if ctx.needs_input_grad[2] else None
        if n * s > 0:
            g_raw = g_raw.contiguous().to(torch.float32)
            srcs = [p.detach().contiguous() for p in params]
            arr = (ctypes.c_void_p * 24)(*[t.data_ptr() for t in srcs])
            garr = (ctypes.c_void_p * 24)(*[g.data_ptr() for g in grads])
            with torch.cuda.device(dev):
                pk_b = torch.empty(int(lib.nerf_packed_bwd_bytes(prec)), dtype=torch.uint8, device=dev)
                _lib.check(lib.nerf_pack_model_bwd(arr, pk_b.data_ptr(), prec, st), "nerf_pack_model_bwd")
                gsave = torch.empty(int(lib.nerf_train_grad_floats(n * s)), dtype=torch.float32, device=dev)
                _lib.check(lib.nerf_mlp_backward_points(_lib.ptr(pts), n, s, pk_b.data_ptr(), _lib.ptr(g_raw), _lib.ptr(save),
                                                        _lib.ptr(gsave), None if g_pts is None else _lib.ptr(g_pts), garr,
                                                        prec, st), "nerf_mlp_backward_points")
        g_dirs = None
        if ctx.needs_input_grad[3]:
            g_dirs = torch.zeros((n, 3), dtype=torch.float32, device=dev)
            if n * s > 0:
                w_views = params[16].detach().contiguous()          # views_linears.0.weight [128, 283]
                with torch.cuda.device(dev):
                    _lib.check(lib.nerf_viewdirs_backward(_lib.ptr(gsave), n, s, _lib.ptr(w_views), _lib.ptr(dirs),
                                                          _lib.ptr(g_dirs), st), "nerf_viewdirs_backward")
        return (None, None, g_pts, g_dirs) + tuple(g.to(p.dtype) if p.requires_grad else None for g, p in zip(grads, params))


def positional_encoding(x, n_freqs):
    """Encoder.embed (src/models/encoding/freq.py:31-32) on the device: [..., 3] -> [..., 3+6*n_freqs]."""
    lib = _lib.load()
    flat = x.detach().reshape(-1, 3).to(torch.float32).contiguous()
    out = torch.empty((flat.shape[0], 3 + 6 * n_freqs), dtype=torch.float32, device=flat.device)
    with torch.cuda.device(flat.device):
        _lib.check(lib.nerf_positional_encoding(_lib.ptr(flat), flat.shape[0], n_freqs, _lib.ptr(out),
                                                _lib.stream_of(flat.device)), "nerf_positional_encoding")
    return out.reshape(*x.shape[:-1], 3 + 6 * n_freqs)
