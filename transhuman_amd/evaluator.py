"""Evaluator of the rendered images -- the consumer of `renderer.render_fast` in `run.py --type evaluate`
("the evaluator's PSNR / PNG path").

Mirrors /root/reference/lib/evaluators/if_nerf.py: ``evaluate(output, batch)`` appends the frame's MSE and PSNR
(:34-37, :121-130), rebuilds the H x W image from the rays inside the body box (`mask_at_box`, :41-57), crops it to the
mask's bounding rectangle (:60-62) and writes `pred/frame{i}_view{v}.png` and `gt/..._gt.png` under
`<result_dir>/<human>/` (:64-99); ``summarize()`` stores `mse.npy` / `psnr.npy` and returns the means (:146-170).
PNG files are written with PIL (cv2 is absent; `cv2.imwrite` of a float image = round-to-nearest, saturate to uint8,
and the reference's RGB -> BGR swap followed by cv2's BGR file order is the identity on the stored RGB).  SSIM
(skimage) and LPIPS (a VGG network download) are third-party, absent here and not part of the rendering path:
``ssim`` / ``lpips`` stay empty lists.
"""
import os

import numpy as np
import torch

from .config import get_cfg
from .mesh import psnr_metric


def _np(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def bounding_rect(mask):
    """cv2.boundingRect of a binary mask -> (x, y, w, h)"""
    ys, xs = np.nonzero(mask)
    if ys.size == 0:
        return 0, 0, 0, 0
    return int(xs.min()), int(ys.min()), int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1)


def to_uint8(img):
    """what cv2.imwrite does to a float image: saturate_cast<uchar>(x) = round to nearest, clamp to [0, 255]"""
    return np.clip(np.rint(img * 255.0), 0, 255).astype(np.uint8)


class Evaluator:
    def __init__(self, result_dir=None):
        cfg = get_cfg()
        self.result_dir = result_dir if result_dir is not None else os.path.join(
            getattr(cfg, "result_dir", "data/result"), "epoch_" + str(getattr(getattr(cfg, "test", None), "epoch", -1)),
            str(getattr(getattr(cfg, "test", None), "exp_folder_name", "debug")))
        self.mse, self.psnr, self.ssim, self.lpips = [], [], [], []

    def psnr_metric(self, img_pred, img_gt):
        return psnr_metric(img_pred, img_gt)

    def images(self, rgb_pred, rgb_gt, batch, H=None, W=None):
        """(:41-62) full-frame prediction / ground truth from the masked ray list, cropped to the box's rectangle"""
        cfg = get_cfg()
        if H is None:
            H, W = int(cfg.H * cfg.ratio), int(cfg.W * cfg.ratio)
        m = _np(batch["mask_at_box"][0]).reshape(H, W).astype(bool)
        fill = 1.0 if cfg.white_bkgd else 0.0
        pred = np.full((H, W, 3), fill)
        gt = np.full((H, W, 3), fill)
        pred[m] = rgb_pred
        gt[m] = rgb_gt
        x, y, w, h = bounding_rect(m)
        return pred[y:y + h, x:x + w], gt[y:y + h, x:x + w]

    def evaluate(self, output, batch, H=None, W=None, save=True):
        from PIL import Image
        rgb_pred = _np(output["rgb_map"][0])
        rgb_gt = _np(batch["rgb"][0])
        mse = float(np.mean((rgb_pred - rgb_gt) ** 2))                     # :124
        self.mse.append(mse)
        self.psnr.append(self.psnr_metric(rgb_pred, rgb_gt))               # :127
        if save and "mask_at_box" in batch:
            pred, gt = self.images(rgb_pred, rgb_gt, batch, H, W)
            human = batch["human_name"][0] if "human_name" in batch else "human"
            frame = int(_np(batch["frame_index"]).reshape(-1)[0]) if "frame_index" in batch else len(self.mse) - 1
            view = int(_np(batch["cam_ind"]).reshape(-1)[0]) if "cam_ind" in batch else 0
            for sub, img, suffix in (("pred", pred, ""), ("gt", gt, "_gt")):
                d = os.path.join(self.result_dir, human, sub)
                os.makedirs(d, exist_ok=True)
                Image.fromarray(to_uint8(img)).save(os.path.join(d, f"frame{frame}_view{view}{suffix}.png"))
        return {"mse": mse, "psnr": self.psnr[-1]}

    def summarize(self):
        os.makedirs(self.result_dir, exist_ok=True)
        np.save(os.path.join(self.result_dir, "mse.npy"), self.mse)
        np.save(os.path.join(self.result_dir, "psnr.npy"), self.psnr)
        out = {"mse": float(np.mean(self.mse)) if self.mse else float("nan"),
               "psnr": float(np.mean(self.psnr)) if self.psnr else float("nan")}
        self.mse, self.psnr, self.ssim, self.lpips = [], [], [], []
        return out
