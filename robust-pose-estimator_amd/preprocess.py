"""Input side on the GPU: counterparts of the reference's per-frame CPU preprocessing, same names and arguments.

  mask_specularities(img, mask=None, spec_thr=0.96)   dataset/stereo_dataset.py:12-16
  ResizeStereo(size)(left, right, mask)               dataset/transforms.py:20-39

Images may be passed as the reference passes them (float32 (3,H,W) tensors) or as decoded (uint8 (H,W,3)), which fuses
the `.permute(2,0,1).float()` of dataset/stereo_dataset.py:35-37 into the resize.  Everything runs in librpe_hip.so
(rpe_mask_specularities, rpe_resize_crop, rpe_resize_crop_mask); tensors must be on the GPU.
"""
import math

import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr


def _gpu(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise _lib.RpeError(f'{name}: expected a tensor on the GPU (the HIP path has no CPU fallback)')
    return t.contiguous()


def mask_specularities(img, mask=None, spec_thr=0.96):
    """img: (H,W,3) uint8 RGB as decoded; mask: (H,W) bool/uint8 or None.  Returns the (H,W) uint8 mask cv2.erode gives."""
    img = _gpu(img, 'img')
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
        raise _lib.RpeError('mask_specularities: img must be (H,W,3) uint8')
    h, w, _ = img.shape
    m = None
    if mask is not None:
        m = _gpu(mask, 'mask')
        m = m.view(torch.uint8) if m.dtype == torch.bool else m
        if m.dtype != torch.uint8 or tuple(m.shape) != (h, w):
            raise _lib.RpeError('mask_specularities: mask must be (H,W) bool/uint8')
    out = torch.empty(h, w, dtype=torch.uint8, device=img.device)
    thr = math.ceil(3 * 255 * spec_thr)              # integer sum < float threshold  <=>  sum < ceil(threshold)
    check(lib().rpe_mask_specularities(ptr(img), ptr(m), h, w, thr, ptr(out), stream_ptr()), 'rpe_mask_specularities')
    return out


class ResizeStereo:
    def __init__(self, size):
        self.size = [int(size[1]), int(size[0])]

    def __call__(self, left, right, mask=None):
        # resize with cropping to conserve aspect ratio (transforms.py:25-33)
        h, w = (left.shape[0], left.shape[1]) if left.dtype == torch.uint8 else left.shape[-2:]
        scale = max(self.size[0] / h, self.size[1] / w)
        size = [int(scale * h), int(scale * w)]
        return self._resize_with_crop(left, size), self._resize_with_crop(right, size), self._resize_with_crop(mask, size, nearest=True)

    def _resize_with_crop(self, img, size, nearest=False):
        if img is None:
            return None
        th, tw = self.size
        if size[0] < th or size[1] < tw:
            raise _lib.RpeError('ResizeStereo: resized image smaller than the crop (torchvision would zero-pad; not supported)')
        top, left = int(round((size[0] - th) / 2.0)), int(round((size[1] - tw) / 2.0))
        img = _gpu(img, 'image')
        if nearest:
            m = img.view(torch.uint8) if img.dtype == torch.bool else img
            if m.dtype != torch.uint8 or m.dim() != 3 or m.shape[0] != 1:
                raise _lib.RpeError('ResizeStereo: mask must be (1,H,W) bool/uint8')
            out = torch.empty(1, th, tw, dtype=torch.uint8, device=m.device)
            check(lib().rpe_resize_crop_mask(ptr(m), m.shape[1], m.shape[2], size[0], size[1], top, left, th, tw, ptr(out), stream_ptr()),
                  'rpe_resize_crop_mask')
            return out.view(torch.bool) if img.dtype == torch.bool else out
        if img.dtype == torch.uint8:                                   # decoded (H,W,C)
            h, w, c = img.shape
            u8 = 1
        elif img.dtype == torch.float32 and img.dim() == 3:            # (C,H,W) as the reference passes it
            c, h, w = img.shape
            u8 = 0
        else:
            raise _lib.RpeError('ResizeStereo: image must be (C,H,W) float32 or (H,W,C) uint8')
        out = torch.empty(c, th, tw, dtype=torch.float32, device=img.device)
        check(lib().rpe_resize_crop(ptr(img), u8, c, h, w, size[0], size[1], top, left, th, tw, ptr(out), stream_ptr()), 'rpe_resize_crop')
        return out
