"""Device-side evaluator metrics, mirroring src/evaluators/nerf.py:
`evaluate(output, batch)` accumulates the float MSE of the clipped images (:96-100) and the PSNR of
`psnr_metric` (:23-30) -- both the value that function really prints (its uint8 subtraction and
squaring wrap modulo 256) and the float PSNR it was meant to compute -- and, when the batch
is a whole H x W image, the SSIM of ssim_metric (:49-77, skimage win 7).  With a `result_dir` the predicted and
ground-truth images are also written as `images/view{id:03d}_pred.png` / `_gt.png` like ssim_metric does (:50-61;
`(img * 255).astype(uint8)`, RGB), by a small zlib PNG writer (cv2 / imageio are not available here)."""
import math
import os
import struct
import zlib

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


def write_png(path, img_hw3_u8):
    """8-bit RGB PNG (filter 0 on every row); `img_hw3_u8`: uint8 tensor or array [H,W,3]."""
    img = torch.as_tensor(img_hw3_u8).detach().cpu().to(torch.uint8).contiguous()
    H, W, C = img.shape
    if C != 3:
        raise ValueError("write_png expects [H,W,3]")
    rows = torch.cat([torch.zeros(H, 1, dtype=torch.uint8), img.reshape(H, W * 3)], dim=1)      # filter byte 0 per row
    raw = rows.numpy().tobytes()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


class Evaluator:
    def __init__(self, result_dir=None):
        self.mse, self.psnr, self.psnr_float, self.ssim, self.imgs = [], [], [], [], []
        self.result_dir = result_dir           # cfg.result_dir of the reference: images go to <result_dir>/images

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
                if self.result_dir is not None:
                    d = os.path.join(self.result_dir, "images")
                    os.makedirs(d, exist_ok=True)
                    view = int(batch["id"]) if "id" in batch else len(self.ssim) - 1
                    to_u8 = lambda x: (x.detach().to(torch.float32).clamp(0, 1) * 255).to(torch.uint8).reshape(H, W, 3)
                    write_png(os.path.join(d, "view{:03d}_pred.png".format(view)), to_u8(rgb_pred))
                    write_png(os.path.join(d, "view{:03d}_gt.png".format(view)), to_u8(rgb_gt))

    def summarize(self):
        mean = lambda v: float(sum(v) / len(v)) if v else 0.0
        return {"mse": mean(self.mse), "psnr": mean(self.psnr), "psnr_float": mean(self.psnr_float),
                "ssim": mean(self.ssim) if self.ssim else None}
