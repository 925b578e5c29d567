"""Input-side preprocessing.  CPU: known answers for the oracle's restatement (the reference
functions need cv2 / torchvision, absent here: parity unpinned, see oracle/preprocess.py).  GPU: the HIP kernels against
the oracle -- masks and nearest indices bit-exact, bilinear within 2e-5 of 255."""
import numpy as np
import pytest
import torch

from oracle import preprocess as opre


def test_oracle_erosion_known_answers():
    img = np.full((40, 50, 3), 100, np.uint8)
    assert opre.mask_specularities(img).all()                                  # nothing specular: borders do not erode
    img[20, 25] = 255                                                          # one specular pixel (765 >= 734.4)
    m = opre.mask_specularities(img)
    assert not m[15:26, 20:31].any() and m.sum() == 40 * 50 - 121
    img[20, 25] = (245, 245, 244)                                              # 734 < 734.4: not specular
    assert opre.mask_specularities(img).all()
    img[20, 25] = (245, 245, 245)                                              # 735: specular
    assert not opre.mask_specularities(img).all()
    tool = np.ones((40, 50), bool); tool[0, 0] = False                         # AND with the tool mask, eroded at the corner
    m = opre.mask_specularities(np.full((40, 50, 3), 100, np.uint8), tool)
    assert not m[:6, :6].any() and m[6, 6] and m[0, 6]


def test_oracle_resize_known_answers():
    rs = opre.ResizeStereo((640, 512))                                         # size is [W, H]
    img = torch.arange(3 * 1024 * 1280, dtype=torch.float32).reshape(3, 1024, 1280) % 251
    l, r, m = rs(img, img + 1, torch.ones(1, 1024, 1280, dtype=torch.bool))
    assert l.shape == (3, 512, 640) and m.shape == (1, 512, 640) and m.dtype == torch.bool
    blocks = img.reshape(3, 512, 2, 640, 2).mean((2, 4))                       # exact 2x: mean of each 2x2 block
    assert torch.allclose(l, blocks, atol=1e-4) and torch.allclose(r, blocks + 1, atol=1e-4)
    same = opre.ResizeStereo((1280, 1024))(img, img, None)
    assert torch.equal(same[0], img) and same[2] is None
    assert opre.resized_size(1080, 1920, (640, 512)) == ([512, 910], (512, 640))
    assert opre.center_crop_offsets(512, 910, 512, 640) == (0, 135) and opre.center_crop_offsets(5, 5, 2, 2) == (2, 2)


@pytest.mark.gpu
@pytest.mark.parametrize('h,w', [(1024, 1280), (37, 129), (540, 960)])
def test_mask_specularities_bit_exact(rpe, h, w):
    from rpe_amd import preprocess
    rng = np.random.default_rng(h)
    img = rng.integers(150, 256, size=(h, w, 3), dtype=np.uint8)               # ~0.4 % of the pixels above the threshold
    img[rng.random((h, w)) < 0.97] //= 2
    tool = rng.random((h, w)) > 0.001
    for mask in (None, tool):
        ref = opre.mask_specularities(img, mask)
        got = preprocess.mask_specularities(torch.from_numpy(img).cuda(), None if mask is None else torch.from_numpy(mask).cuda())
        assert got.dtype == torch.uint8 and np.array_equal(got.cpu().numpy(), ref)
        assert 0 < ref.mean() < 1


@pytest.mark.gpu
@pytest.mark.parametrize('h,w,size', [(1024, 1280, (640, 512)), (1080, 1920, (640, 512)), (480, 640, (640, 512)), (600, 500, (320, 256))])
def test_resize_stereo_matches_oracle(rpe, h, w, size):
    from rpe_amd import preprocess
    rng = np.random.default_rng(w)
    left = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8); right = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    mask = rng.random((1, h, w)) > 0.3
    ol, orr, om = opre.ResizeStereo(size)(opre.to_float_chw(left), opre.to_float_chw(right), torch.from_numpy(mask))
    rs = preprocess.ResizeStereo(size)
    gl, gr, gm = rs(opre.to_float_chw(left).cuda(), opre.to_float_chw(right).cuda(), torch.from_numpy(mask).cuda())
    assert gl.shape == ol.shape and (gl.cpu() - ol).abs().max() < 2e-5 * 255 and (gr.cpu() - orr).abs().max() < 2e-5 * 255
    assert gm.dtype == torch.bool and torch.equal(gm.cpu(), om)
    ul, ur, _ = rs(torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda(), None)       # decoded uint8 HWC in, fused conversion
    assert torch.equal(ul, gl) and torch.equal(ur, gr)


@pytest.mark.gpu
def test_resize_smaller_than_crop_is_refused(rpe):
    from rpe_amd import preprocess
    with pytest.raises(rpe.RpeError):
        preprocess.ResizeStereo((640, 512))._resize_with_crop(torch.zeros(3, 8, 8, device='cuda'), [100, 100])
