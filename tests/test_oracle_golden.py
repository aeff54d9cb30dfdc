"""The CPU oracle against the committed known answers (tests/golden/kats.json).

PARITY UNPINNED: the fixtures are hand/model-derived (tests/golden/make_kats.py), not output of
the reference -- the reference holds no tests or vectors for this path.
"""
import json
import os

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
import glob
CASES = []   # kats.json (hand / model derived) plus, when a maintainer has generated it, kats_go.json (tools/gen_go_vectors: Go's own output)
for _p in sorted(glob.glob(os.path.join(HERE, "golden", "kats*.json"))):
    with open(_p) as f:
        CASES += json.load(f)["cases"]


def _by(kind):
    return [c for c in CASES if c["kind"] == kind]


def _frame(flat, w, h):
    return np.array(flat, np.uint8).reshape(h, w, 4)


@pytest.mark.parametrize("c", _by("scale"), ids=lambda c: c["name"])
def test_scale(c):
    src = _frame(c["src"], c["sw"], c["sh"])
    dst = _frame(c["dst"], c["dw"], c["dh"]).copy()
    oracle.scale_bilinear(src, c["dw"], c["dh"], sr=c["sr"], dr=c["dr"], op=c["op"], dst=dst)
    np.testing.assert_array_equal(dst, _frame(c["expect"], c["dw"], c["dh"]))


@pytest.mark.parametrize("c", _by("draw"), ids=lambda c: c["name"])
def test_draw(c):
    src = _frame(c["src"], c["sw"], c["sh"])
    dst = _frame(c["dst"], c["dw"], c["dh"]).copy()
    oracle.draw(dst, c["r"], src, c["sp"], c["op"])
    np.testing.assert_array_equal(dst, _frame(c["expect"], c["dw"], c["dh"]))


@pytest.mark.parametrize("c", _by("glyphs"), ids=lambda c: c["name"])
def test_glyphs(c):
    dst = _frame(c["dst"], c["dw"], c["dh"]).copy()
    glyphs = [{"mask": np.array(g["mask"], np.uint8).reshape(g["mh"], g["mw"]), "dr": g["dr"],
               "mp": g["mp"]} for g in c["glyphs"]]
    oracle.composite_glyphs(dst, glyphs, c["col"])
    np.testing.assert_array_equal(dst, _frame(c["expect"], c["dw"], c["dh"]))


def _planes(img):
    w, h, ratio = img["w"], img["h"], img["ratio"]
    chh, cw = oracle.chroma_shape(w, h, ratio)
    return (np.array(img["y"], np.uint8).reshape(h, w), np.array(img["cb"], np.uint8).reshape(chh, cw),
            np.array(img["cr"], np.uint8).reshape(chh, cw), ratio)


@pytest.mark.parametrize("c", _by("draw_nrgba") + _by("scale_nrgba"), ids=lambda c: c["name"])
def test_nrgba_sources(c):
    src = _frame(c["src"], c["sw"], c["sh"])
    dst = _frame(c["dst"], c["dw"], c["dh"]).copy()
    if c["kind"] == "draw_nrgba":
        oracle.draw_nrgba(dst, c["r"], src, c["sp"], c["op"])
    else:
        oracle.scale_bilinear_nrgba(src, c["dw"], c["dh"], sr=c["sr"], dr=c["dr"], op=c["op"], dst=dst)
    np.testing.assert_array_equal(dst, _frame(c["expect"], c["dw"], c["dh"]))


@pytest.mark.parametrize("c", _by("draw_ycbcr") + _by("scale_ycbcr"), ids=lambda c: c["name"])
def test_ycbcr_sources(c):
    y, cb, cr, ratio = _planes(c["img"])
    dst = _frame(c["dst"], c["dw"], c["dh"]).copy()
    if c["kind"] == "draw_ycbcr":
        oracle.draw_ycbcr(dst, c["r"], y, cb, cr, ratio, c["sp"])
    else:
        oracle.scale_bilinear_ycbcr(y, cb, cr, ratio, c["dw"], c["dh"], sr=c["sr"], dr=c["dr"], dst=dst)
    np.testing.assert_array_equal(dst, _frame(c["expect"], c["dw"], c["dh"]))


def test_geometry():
    for c in _by("resize_dims"):
        assert list(oracle.resize_dims(c["ow"], c["oh"], c["w"], c["h"], c["keep_aspect"])) == c["expect"]
    for c in _by("thumb_geometry"):
        crop, nw, nh = oracle.thumb_geometry(c["ow"], c["oh"], c["size"], c["crop_to_fit"])
        assert list(crop) == c["expect"]["crop"] and (nw, nh) == (c["expect"]["nw"], c["expect"]["nh"])
    for c in _by("text_height"):
        assert oracle.text_height_px(c["font_size"]) == c["expect"]
    for c in _by("anchor"):
        assert list(oracle.watermark_anchor(c["position"], c["w"], c["h"], c["width_px"],
                                            c["height_px"])) == c["expect"]
    for c in _by("parse_color"):
        rgba, err = oracle.parse_color(c["s"], c["opacity"])
        assert list(rgba) == c["expect"]["rgba"] and err == c["expect"]["error"], c


def test_pipeline_composition():
    """process() == the three operators run separately on the ORIGINAL frame."""
    rng = np.random.default_rng(7)
    src = rng.integers(0, 256, (45, 80, 4), dtype=np.uint8)
    src[..., 3] = 255
    mask = rng.integers(0, 256, (9, 14), dtype=np.uint8)
    glyphs = [{"mask": mask, "dr": (60, 30, 74, 39), "mp": (0, 0)}]
    out = oracle.process(src, resize=(32, 24, True), thumb=(10, True), glyphs=glyphs)
    nw, nh = oracle.resize_dims(80, 45, 32, 24, True)
    np.testing.assert_array_equal(out["resize"], oracle.scale_bilinear(src, nw, nh))
    crop, tw, th = oracle.thumb_geometry(80, 45, 10, True)
    np.testing.assert_array_equal(out["thumbnail"], oracle.scale_bilinear(src, tw, th, sr=crop))
    wm = src.copy()
    oracle.composite_glyphs(wm, glyphs, (255, 255, 255, 127))
    np.testing.assert_array_equal(out["watermark"], wm)
