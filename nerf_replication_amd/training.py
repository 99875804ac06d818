"""Training path (BASELINE config 3): Renderer.render under autograd, fp32.

The reference trains by calling the same Renderer.render with autograd on (src/train/trainers/nerf.py:27,
trainer.py:53-60): MSE on the fine RGB only, and -- because fine_sample_points does not detach the
coarse weights -- the coarse network learns through the sample positions.  Here the
forward runs the SAVE-mode fused kernels and the backward is the chain of adjoint kernels behind the
C ABI (nerf_composite_backward, nerf_mlp_backward, nerf_sample_fine_backward); torch.autograd only
carries the 48 parameter gradients back to the optimizer.  No ATen op computes on this path.
"""
import ctypes

import torch

from . import _lib


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class RenderFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, renderer, rays_o, rays_d, *params):
        lib = _lib.load()
        net = renderer.net
        dev = rays_o.device
        n = rays_o.shape[0]
        st = _lib.stream_of(dev)
        t_c, u = renderer._get_tables(dev)
        S_c, S_f = _lib.N_SAMPLES, _lib.N_SAMPLES + _lib.N_IMPORTANCE
        pk_c, pk_f = net.packed(""), net.packed("fine")
        prec = _lib.PRECISIONS[getattr(net, "precision", "f32")]
        f32 = dict(dtype=torch.float32, device=dev)
        raw_c = torch.empty((n, S_c, 4), **f32)
        save_c = torch.empty(int(lib.nerf_train_save_floats(n * S_c)), **f32)
        t_sorted = torch.empty((n, S_f), **f32)
        raw_f = torch.empty((n, S_f, 4), **f32)
        save_f = torch.empty(int(lib.nerf_train_save_floats(n * S_f)), **f32)
        rgb, depth = torch.empty((n, 3), **f32), torch.empty((n,), **f32)
        with torch.cuda.device(dev):
            # coarse pass: only its sigma is ever used (it places the fine samples; the coarse colour is never
            # composited) -> the density-only forward / backward pair
            _lib.check(lib.nerf_mlp_forward_rays_save_density(_lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(t_c), 0, n, S_c,
                                                              pk_c.data_ptr(), _lib.ptr(raw_c), _lib.ptr(save_c), prec, st), "forward(coarse)")
            _lib.check(lib.nerf_sample_fine(_lib.ptr(raw_c), _lib.ptr(t_c), _lib.ptr(u), n, _lib.ptr(t_sorted), None, None,
                                            0.0, 0.0, st), "nerf_sample_fine")
            # fine pass: its raw goes to compositing only, and its gradient will come from compositing's adjoint (zero
            # wherever sigma <= 0): tiles without density skip the colour branch and its stores (exact, see the header)
            _lib.check(lib.nerf_mlp_forward_rays_save_for_compositing(_lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(t_sorted), S_f, n, S_f,
                                                      pk_f.data_ptr(), _lib.ptr(raw_f), _lib.ptr(save_f), prec, st), "forward(fine)")
            _lib.check(lib.nerf_composite(_lib.ptr(raw_f), _lib.ptr(t_sorted), S_f, n, S_f, int(bool(renderer.white_bkgd)),
                                          _lib.ptr(rgb), _lib.ptr(depth), None, st), "nerf_composite")
        ctx.renderer = renderer
        ctx.prec = prec
        ctx.n = n
        ctx.params = params
        ctx.save_for_backward(rays_o, rays_d, raw_c, save_c, t_sorted, raw_f, save_f)
        return rgb, depth

    @staticmethod
    def backward(ctx, g_rgb, g_depth):
        lib = _lib.load()
        renderer, n, params = ctx.renderer, ctx.n, ctx.params
        rays_o, rays_d, raw_c, save_c, t_sorted, raw_f, save_f = ctx.saved_tensors
        dev = rays_o.device
        st = _lib.stream_of(dev)
        t_c, u = renderer._get_tables(dev)
        S_c, S_f = _lib.N_SAMPLES, _lib.N_SAMPLES + _lib.N_IMPORTANCE
        f32 = dict(dtype=torch.float32, device=dev)
        g_rgb = g_rgb.contiguous().to(torch.float32)
        g_depth = None if g_depth is None else g_depth.contiguous().to(torch.float32)
        # 24 coarse + 24 fine gradient tensors as views of one zeroed buffer (one memset instead of 48)
        flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
        grads, off = [], 0
        for p in params:
            grads.append(flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
        prec = ctx.prec
        nbwd = int(lib.nerf_packed_bwd_bytes(prec))
        with torch.cuda.device(dev):
            # fine pass: image -> raw_fine and depths; MLP backward; points -> depths
            g_raw_f = torch.empty((n, S_f, 4), **f32)
            g_t = torch.empty((n, S_f), **f32)
            _lib.check(lib.nerf_composite_backward(_lib.ptr(raw_f), _lib.ptr(t_sorted), S_f, n, S_f,
                                                   int(bool(renderer.white_bkgd)), _lib.ptr(g_rgb),
                                                   None if g_depth is None else _lib.ptr(g_depth),
                                                   _lib.ptr(g_raw_f), _lib.ptr(g_t), st), "nerf_composite_backward")
            pk_b = torch.empty(nbwd, dtype=torch.uint8, device=dev)
            _lib.check(lib.nerf_pack_model_bwd(_ptr_array([p.detach().contiguous() for p in params[24:]]), pk_b.data_ptr(), prec, st))
            gsave = torch.empty(int(lib.nerf_train_grad_floats(n * S_f)), **f32)
            g_t_pts = torch.empty((n, S_f), **f32)
            _lib.check(lib.nerf_mlp_backward(_lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(t_sorted), S_f, n, S_f,
                                             pk_b.data_ptr(), _lib.ptr(g_raw_f), _lib.ptr(save_f), _lib.ptr(gsave),
                                             _lib.ptr(g_t_pts), _ptr_array(grads[24:]), prec, st), "nerf_mlp_backward(fine)")
            g_t.add_(g_t_pts)                     # plumbing: one elementwise add of two [n,192] buffers
            cnt_f = None
            if getattr(renderer, "live_tile_stats", None) is not None:
                cnt_f = gsave[int(lib.nerf_train_live_count_offset(n * S_f))].view(torch.int32).clone()
            # coarse pass: depths -> coarse density -> coarse MLP parameters
            g_raw_c = torch.empty((n, S_c, 4), **f32)
            _lib.check(lib.nerf_sample_fine_backward(_lib.ptr(raw_c), _lib.ptr(t_c), _lib.ptr(u), n, _lib.ptr(t_sorted),
                                                     _lib.ptr(g_t), _lib.ptr(g_raw_c), st), "nerf_sample_fine_backward")
            cap = getattr(renderer, "capture_adjoints", None)
            if cap is not None:       # tests: the per-ray sampler adjoint d loss / d raw_coarse and d loss / d t_sorted (parity attribution)
                cap["g_raw_coarse"], cap["g_t_sorted"], cap["raw_coarse"] = g_raw_c.clone(), g_t.clone(), raw_c.clone()
                cap["t_sorted"] = t_sorted.clone()
            _lib.check(lib.nerf_pack_model_bwd(_ptr_array([p.detach().contiguous() for p in params[:24]]), pk_b.data_ptr(), prec, st))
            gsave_c = gsave[: int(lib.nerf_train_grad_floats(n * S_c))]
            _lib.check(lib.nerf_mlp_backward_density(_lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(t_c), 0, n, S_c,
                                                     pk_b.data_ptr(), _lib.ptr(g_raw_c), _lib.ptr(save_c), _lib.ptr(gsave_c),
                                                     None, _ptr_array(grads[:24]), prec, st), "nerf_mlp_backward(coarse)")
            stats = getattr(renderer, "live_tile_stats", None)
            if stats is not None:
                # (live tiles, tiles) of the coarse pass as 1-element device tensors: no host sync here; the fine pass's
                # count was cloned above, before its gsave was reused
                cnt_c = gsave_c[int(lib.nerf_train_live_count_offset(n * S_c))].view(torch.int32).clone()
                stats.append((cnt_f, n * S_f // 32, cnt_c, n * S_c // 32))
        return (None, None, None) + tuple(g.to(p.dtype) for g, p in zip(grads, params))


def render_with_grad(renderer, rays_o, rays_d):
    """rays [n,3] (contiguous fp32, on the GPU) -> (rgb [n,3], depth [n]) attached to the autograd graph of
    the 48 network parameters (coarse sub-model first, then fine, state_dict order)."""
    net = renderer.net
    if getattr(net, "precision", "f32") not in ("f32", "f32x"):
        raise NotImplementedError("training runs on the fp32-accurate paths: precision 'f32' (exact fp32 MFMA) or 'f32x' "
                                  "(forward on split-fp16 MFMA; the backward kernels are fp32 MFMA either way)")
    if renderer.N_importance != _lib.N_IMPORTANCE or renderer.fast_sampling:
        raise NotImplementedError("training path is built for N_importance=128 without fast_sampling")
    params = tuple(net.model.ordered_params()) + tuple(net.model_fine.ordered_params())
    return RenderFunction.apply(renderer, rays_o, rays_d, *params)


class FusedAdam(torch.optim.Optimizer):
    """clip_grad_value_ + Adam in one HIP launch over all parameter tensors of a group (nerf_adam_step).  Same update
    as torch.optim.Adam(lr, eps, weight_decay) of src/train/optimizer.py:21-24 preceded by trainer.py:59's
    clip_grad_value_(40).  It IS a torch.optim.Optimizer (param_groups, state, state_dict in torch.optim.Adam's layout),
    so the reference's schedulers (ExponentialLR / MultiStepLR of src/utils/optimizer/lr_scheduler.py, built by
    make_lr_scheduler) and its save_model / load_model (net_utils.py:288-343) drive it unchanged, and a checkpoint
    moves freely between this optimizer and torch.optim.Adam."""

    def __init__(self, params, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_value=40.0):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, clip_value=clip_value,
                        amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None)
        super().__init__([p for p in params if not isinstance(p, torch.Tensor) or p.requires_grad], defaults)
        for g in self.param_groups:
            if len(g["params"]) > 48:
                raise ValueError("FusedAdam handles at most 48 tensors per parameter group (one launch per group)")

    def _state_of(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32)
            st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32)
        return st

    # conveniences over param_groups[0] / state (tests, bench)
    @property
    def params(self):
        return [p for g in self.param_groups for p in g["params"]]

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    @lr.setter
    def lr(self, value):
        for g in self.param_groups:
            g["lr"] = value

    @property
    def clip_value(self):
        return self.param_groups[0]["clip_value"]

    @property
    def exp_avg(self):
        return [self._state_of(p)["exp_avg"] for p in self.params]

    @property
    def exp_avg_sq(self):
        return [self._state_of(p)["exp_avg_sq"] for p in self.params]

    @property
    def step_count(self):
        return max([int(float(self._state_of(p)["step"])) for p in self.params] or [0])

    @step_count.setter
    def step_count(self, value):
        for p in self.params:
            self._state_of(p)["step"] = torch.tensor(float(value))

    @staticmethod
    def exponential_lr(lr0, epoch, gamma=0.1, decay_epochs=500):
        return lr0 * gamma ** (epoch / decay_epochs)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        bump = getattr(torch.autograd.graph, "increment_version", None)
        for g in self.param_groups:
            if g.get("amsgrad", False) or g.get("maximize", False):
                raise ValueError("FusedAdam does not implement amsgrad / maximize")
            live = [p for p in g["params"] if p.grad is not None]
            if not live:
                continue
            states = [self._state_of(p) for p in live]
            steps = {int(float(st["step"])) for st in states}
            if len(steps) != 1:
                raise ValueError("FusedAdam: tensors of one group are at different steps {}".format(sorted(steps)))
            step = steps.pop() + 1
            dev = live[0].device
            for st in states:                     # state loaded from a CPU checkpoint follows its parameter
                for k in ("exp_avg", "exp_avg_sq"):
                    if st[k].device != dev:
                        st[k] = st[k].to(dev)
            grads = [p.grad.contiguous() for p in live]
            numel = (ctypes.c_int64 * len(live))(*[p.numel() for p in live])
            with torch.cuda.device(dev):
                _lib.check(lib.nerf_adam_step(len(live), arr(live), arr(grads), arr([st["exp_avg"] for st in states]),
                                              arr([st["exp_avg_sq"] for st in states]), numel, float(g["lr"]),
                                              float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                              float(g["weight_decay"]), float(g.get("clip_value", 40.0)), step,
                                              _lib.stream_of(dev)), "nerf_adam_step")
            for p, st in zip(live, states):
                st["step"] = torch.tensor(float(step))
                # the packed weight streams are keyed on (data_ptr, _version): tell autograd the HIP kernel wrote in
                # place, so Network.packed() repacks (cache invalidation only, no kernel launched)
                if bump is not None:
                    bump(p)
                else:
                    p.add_(0)
        return loss


def train_step(renderer, optimizer, rays_o, rays_d, colors, clip_value=40.0, group=None):
    """One step of the reference's intended loop (trainer.py:53-60 with trainers/nerf.py:27-33): render,
    MSE on the fine RGB, backward, [data-parallel: one gradient all-reduce], clip_grad_value_(40),
    optimizer.step().  Returns the loss tensor."""
    from .dist import allreduce_gradients
    optimizer.zero_grad(set_to_none=True)
    rgb, _ = renderer.render({"rays_o": rays_o[None], "rays_d": rays_d[None]})
    loss = torch.nn.functional.mse_loss(rgb, colors)
    loss.backward()
    allreduce_gradients(renderer.net.parameters(), group)
    if not isinstance(optimizer, FusedAdam):          # FusedAdam clips inside its kernel
        torch.nn.utils.clip_grad_value_(renderer.net.parameters(), clip_value)
    optimizer.step()
    return loss.detach()
