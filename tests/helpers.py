"""Shared synthetic inputs for the tests (seeded, shape-realistic, no reference data)."""
import numpy as np


def rgba_frames(n, w, h, seed=0x1F00D, opaque=True, premul=True):
    """n x h x w x 4 uint8; opaque frames mimic decoded photos (A = 255)."""
    rng = np.random.default_rng(seed)
    f = rng.integers(0, 256, (n, h, w, 4), dtype=np.uint8)
    if opaque:
        f[..., 3] = 255
    elif premul:
        a = f[..., 3:4].astype(np.uint16)
        f[..., :3] = (f[..., :3].astype(np.uint16) * a // 255).astype(np.uint8)
    return f


def text_glyphs(w, h, n=16, seed=0xA8, width_px=300, height_px=44, position="bottom-right",
                margin=20):
    """A run of n glyph-like A8 masks laid out like freetype.DrawString would for the default
    36 pt watermark (watermark.go:116-151): boxes about 19 x 44 px walking right from the
    anchor, neighbouring boxes overlapping by a pixel or two, coverage in {0, 1..254, 255}.
    NOT rasterised text: the glyph producer is outside the path."""
    rng = np.random.default_rng(seed)
    if position == "bottom-right":
        px, py = w - width_px - margin, h - margin
    elif position == "top-left":
        px, py = margin, margin + height_px
    else:
        px, py = (w - width_px) // 2, (h + height_px) // 2
    adv = width_px / n
    glyphs = []
    for i in range(n):
        mw = int(adv) + int(rng.integers(1, 4))
        mh = int(rng.integers(height_px // 2, height_px + 1))
        m = rng.integers(0, 256, (mh, mw), dtype=np.uint8)
        sel = rng.random((mh, mw))
        m[sel < 0.35] = 0
        m[sel > 0.75] = 255
        x0 = px + int(i * adv) - 1
        y0 = py - mh + int(rng.integers(0, 8))
        glyphs.append({"mask": m, "dr": (x0, y0, x0 + mw, y0 + mh), "mp": (0, 0)})
    return glyphs


DEFAULT_COL = (255, 255, 255, 127)  # parseColor("255,255,255", 0.5): NOT premultiplied
