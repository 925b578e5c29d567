"""TEST INFRASTRUCTURE (CPU oracle) -- restatement of the reference's per-frame input preprocessing.

Follows dataset/stereo_dataset.py:12-16 (mask_specularities), dataset/transforms.py:20-39 (ResizeStereo) and the
uint8 HWC -> float CHW conversion at dataset/stereo_dataset.py:35-37 / dataset/video_dataset.py:60-62.
PARITY UNPINNED: the reference functions need cv2 and torchvision, neither of which is installed here, so they cannot
be imported to generate golden vectors.  The restatement uses the arithmetic of the versions the reference pins
(requirements.txt: torch==1.13.0, torchvision==0.14.0):
  * torchvision.transforms.functional.resize on a float tensor = torch.nn.functional.interpolate(size=..., mode=
    'bilinear', align_corners=False) with antialias off (antialias=None means False for tensors in 0.14); NEAREST on a
    bool mask goes through float32 and back;
  * center_crop: top = int(round((H - h) / 2.0)), left likewise (Python round);
  * cv2.erode(mask, np.ones((11, 11))) = 11x11 minimum filter whose border pixels never erode (default border value);
    here -max_pool2d(-m, 11, 1, 5), whose implicit padding is -inf.
"""
import numpy as np
import torch
import torch.nn.functional as F


def mask_specularities(img, mask=None, spec_thr=0.96):
    """img: (H,W,3) uint8 array; mask: (H,W) bool or None -> (H,W) uint8 (stereo_dataset.py:12-16)."""
    img = np.asarray(img)
    spec_mask = img.sum(axis=-1) < (3 * 255 * spec_thr)
    m = (np.asarray(mask).astype(bool) & spec_mask) if mask is not None else spec_mask
    t = torch.from_numpy(m.astype(np.float32))[None, None]
    eroded = -F.max_pool2d(-t, kernel_size=11, stride=1, padding=5)
    return eroded[0, 0].numpy().astype(np.uint8)


def resized_size(h, w, size):
    """transforms.py:25-29 with self.size = [int(size[1]), int(size[0])] (size is given as [W, H])."""
    th, tw = int(size[1]), int(size[0])
    scale = max(th / h, tw / w)
    return [int(scale * h), int(scale * w)], (th, tw)


def center_crop_offsets(h, w, th, tw):
    return int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))


class ResizeStereo:
    def __init__(self, size):
        self.size = [int(size[1]), int(size[0])]

    def __call__(self, left, right, mask=None):
        h, w = left.shape[-2:]
        scale = max(self.size[0] / h, self.size[1] / w)
        size = [int(scale * h), int(scale * w)]
        return self._rc(left, size), self._rc(right, size), self._rc(mask, size, nearest=True)

    def _rc(self, img, size, nearest=False):
        if img is None:
            return None
        if nearest:
            out = F.interpolate(img[None].float(), size=size, mode='nearest')[0].to(img.dtype)
        else:
            out = F.interpolate(img[None], size=size, mode='bilinear', align_corners=False)[0]
        th, tw = self.size
        if size[0] < th or size[1] < tw:
            raise ValueError('resized image smaller than the crop (torchvision would zero-pad)')
        top, left = center_crop_offsets(size[0], size[1], th, tw)
        return out[..., top:top + th, left:left + tw]


def to_float_chw(img_hwc_u8):
    return torch.from_numpy(np.asarray(img_hwc_u8)).permute(2, 0, 1).float()
