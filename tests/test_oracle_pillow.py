"""The oracle's scaler against an independent implementation of the same filter.

x/image/draw's BiLinear is the tent kernel with its support widened by the downscale ratio (draw/scale.go newDistrib); Pillow's
Image.resize(..., BILINEAR) is the same filter (Resample.c precompute_coeffs: support * max(scale, 1), weights 1 - |x| renormalised,
horizontal pass first), computed with 8-bit intermediates and fixed-point coefficients.  So the two agree to about one grey level per
pass -- while the 2-tap ApproxBiLinear, which rounds 1-2 restated here by mistake, is far away on any real downscale.  This does not
pin Go's last bit (PARITY UNPINNED stays), it pins the ALGORITHM: tap ranges, weights, normalisation, the crop rectangle.
"""
import numpy as np
import pytest

import oracle

PIL = pytest.importorskip("PIL.Image")


def _frames():
    rng = np.random.default_rng(0x7E47)
    yy, xx = np.mgrid[0:270, 0:480]
    smooth = np.stack([(xx * 255 // 479), (yy * 255 // 269), ((xx + yy) * 255 // 748)], -1).astype(np.uint8)
    noise = rng.integers(0, 256, (270, 480, 3), dtype=np.uint8)
    checker = (((xx // 3 + yy // 5) & 1) * 255).astype(np.uint8)[..., None].repeat(3, -1)
    return {"smooth": smooth, "noise": noise, "checker": checker}


def _rgba(rgb):
    return np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 255, np.uint8)], -1)


@pytest.mark.parametrize("name", ["smooth", "noise", "checker"])
@pytest.mark.parametrize("dw,dh,box", [(256, 192, None),          # 480x270 -> 256x192: the headline's 1.875 / 1.40625 ratios
                                        (256, 144, None),          # keep_aspect
                                        (50, 50, (105, 0, 375, 270)),   # the crop thumbnail's 5.4
                                        (768, 432, None),          # upscale 1.6
                                        (37, 23, None)])           # 12.97 / 11.7: 25 taps per axis
def test_tent_scaler_matches_pillow_bilinear(name, dw, dh, box):
    rgb = _frames()[name]
    ours = oracle.scale_bilinear(_rgba(rgb), dw, dh, sr=box)
    img = PIL.fromarray(rgb, "RGB")
    if box:                        # crop FIRST, like cropAndResize (thumbnail.go:128-131): Pillow's own box= reads beyond the box
        img = img.crop(box)
    theirs = np.asarray(img.resize((dw, dh), PIL.BILINEAR, reducing_gap=None))
    assert (ours[..., 3] == 255).all()
    d = np.abs(ours[..., :3].astype(int) - theirs.astype(int))
    assert d.max() <= 2, (name, dw, dh, int(d.max()))
    assert d.mean() < 0.35, (name, float(d.mean()))


def test_two_tap_interpolation_is_not_this_filter():
    """what the earlier rounds computed (sample 2x2 taps at the scaled centre) differs grossly on a 5.4x downscale of noise"""
    rgb = _frames()["noise"]
    ours = oracle.scale_bilinear(_rgba(rgb), 50, 50, sr=(105, 0, 375, 270))[..., :3].astype(float)
    crop = rgb[:, 105:375].astype(float)
    sy = (np.arange(50) + 0.5) * 5.4 - 0.5
    y0 = np.floor(sy).astype(int); fy = (sy - y0)[:, None, None]
    x0 = y0; fx = (sy - x0)[None, :, None]
    two_tap = ((crop[y0][:, x0] * (1 - fx) + crop[y0][:, x0 + 1] * fx) * (1 - fy) +
               (crop[y0 + 1][:, x0] * (1 - fx) + crop[y0 + 1][:, x0 + 1] * fx) * fy)
    assert np.abs(ours - two_tap).mean() > 20      # noise averaged over 11 x 11 taps vs over 2 x 2
    assert ours.std() < 0.5 * two_tap.std()
