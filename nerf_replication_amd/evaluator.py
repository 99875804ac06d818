"""Device-side evaluator metrics, mirroring src/evaluators/nerf.py:
`evaluate(output, batch)` accumulates the float MSE of the clipped images (:96-100) and the PSNR of
`psnr_metric` (:23-30) -- both the value that function really prints (its uint8 subtraction and
squaring wrap modulo 256) and the float PSNR it was meant to compute -- and, when the batch
is a whole H x W image, the SSIM of ssim_metric (:49-77, skimage win 7).  The PNG dump is not built."""
import math

import torch

from . import _lib


def image_sums(pred, gt):
    """-> (sum of squared float differences of the clipped images, sum of the uint8-wrapped integrand, n)."""
    lib = _lib.load()
    pred = pred.detach().reshape(-1).to(torch.float32).contiguous()
    gt = gt.detach().reshape(-1).to(device=pred.device, dtype=torch.float32).contiguous()
    if pred.numel() != gt.numel():
        raise ValueError("pred and gt must have the same number of values")
    sums = torch.empty(2, dtype=torch.float64, device=pred.device)
    with torch.cuda.device(pred.device):
        _lib.check(lib.nerf_image_metrics(_lib.ptr(pred), _lib.ptr(gt), pred.numel(), sums.data_ptr(),
                                          _lib.stream_of(pred.device)), "nerf_image_metrics")
    s = sums.cpu().tolist()
    return s[0], s[1], pred.numel()


def image_ssim(pred_hw3, gt_hw3):
    """SSIM as ssim_metric computes it (evaluators/nerf.py:49-77), on [H,W,3] float images in [0,1]."""
    lib = _lib.load()
    H, W = int(pred_hw3.shape[0]), int(pred_hw3.shape[1])
    pred = pred_hw3.detach().to(torch.float32).contiguous()
    gt = gt_hw3.detach().to(device=pred.device, dtype=torch.float32).contiguous()
    out = torch.empty(1, dtype=torch.float64, device=pred.device)
    with torch.cuda.device(pred.device):
        _lib.check(lib.nerf_image_ssim(_lib.ptr(pred), _lib.ptr(gt), H, W, out.data_ptr(), _lib.stream_of(pred.device)),
                   "nerf_image_ssim")
    return out.item() / ((H - 6) * (W - 6) * 3)


class Evaluator:
    def __init__(self):
        self.mse, self.psnr, self.psnr_float, self.ssim, self.imgs = [], [], [], [], []

    def evaluate(self, output, batch):
        rgb_pred = output[0]
        rgb_gt = batch["colors"][..., :3].reshape(-1, 3)
        s_f, s_u, n = image_sums(rgb_pred, rgb_gt)
        mse = s_f / n
        mse_u8 = s_u / n
        self.mse.append(mse)
        self.psnr.append(100.0 if mse_u8 < 1e-10 else 10.0 * math.log10(255.0 ** 2 / mse_u8))   # what psnr_metric prints
        self.psnr_float.append(100.0 if mse < 1e-20 else 10.0 * math.log10(1.0 / mse))            # float PSNR, data_range 1
        if "H" in batch and "W" in batch:                                                         # whole image: SSIM (:115-120)
            H, W = int(batch["H"]), int(batch["W"])
            if H * W * 3 == n:
                self.ssim.append(image_ssim(rgb_pred.reshape(H, W, 3), rgb_gt.reshape(H, W, 3)))

    def summarize(self):
        mean = lambda v: float(sum(v) / len(v)) if v else 0.0
        return {"mse": mean(self.mse), "psnr": mean(self.psnr), "psnr_float": mean(self.psnr_float),
                "ssim": mean(self.ssim) if self.ssim else None}
