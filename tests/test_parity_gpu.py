"""HIP path vs the CPU oracle, through the C ABI (include/ipx.h).  Bit-exact everywhere: the
kernels do the kernel scaler's float64 operations in the reference's order with contraction off, and the
uint32 composite as it is (the +-1 LSB allowance of the north star for the scaler is not needed and not used).

PARITY UNPINNED against the Go reference itself: see oracle/ipx_oracle.h.
"""
import json
import os

import numpy as np
import pytest

import oracle
from helpers import DEFAULT_COL, rgba_frames, text_glyphs

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
import glob
CASES = []   # kats.json (hand / model derived) plus, when a maintainer has generated it, kats_go.json (tools/gen_go_vectors: Go's own output)
for _p in sorted(glob.glob(os.path.join(HERE, "golden", "kats*.json"))):
    with open(_p) as f:
        CASES += json.load(f)["cases"]


@pytest.fixture(scope="module")
def ipx():
    import imageprocessor_amd as m
    return m


@pytest.fixture(scope="module")
def ctx(ipx):
    c = ipx.Context(lanes=2)
    yield c
    c.close()


def _frame(flat, w, h):
    return np.array(flat, np.uint8).reshape(h, w, 4)


def _glyphs_from_case(c):
    return [{"mask": np.array(g["mask"], np.uint8).reshape(g["mh"], g["mw"]), "dr": g["dr"],
             "mp": g["mp"]} for g in c["glyphs"]]


# ---- committed known answers through the per-operation seam ---------------------------------------

def test_golden_scale(ctx):
    for c in [c for c in CASES if c["kind"] == "scale"]:
        dst = _frame(c["dst"], c["dw"], c["dh"]).copy()
        ctx.scale_bilinear(_frame(c["src"], c["sw"], c["sh"]), c["dw"], c["dh"], sr=c["sr"], dr=c["dr"],
                           op=c["op"], dst=dst)
        np.testing.assert_array_equal(dst, _frame(c["expect"], c["dw"], c["dh"]), err_msg=c["name"])


def test_golden_draw(ctx):
    for c in [c for c in CASES if c["kind"] == "draw"]:
        dst = _frame(c["dst"], c["dw"], c["dh"]).copy()
        ctx.draw(dst, c["r"], _frame(c["src"], c["sw"], c["sh"]), c["sp"], c["op"])
        np.testing.assert_array_equal(dst, _frame(c["expect"], c["dw"], c["dh"]), err_msg=c["name"])


def test_golden_glyphs(ctx):
    for c in [c for c in CASES if c["kind"] == "glyphs"]:
        dst = _frame(c["dst"], c["dw"], c["dh"]).copy()
        ctx.composite_glyphs(dst, _glyphs_from_case(c), c["col"])
        np.testing.assert_array_equal(dst, _frame(c["expect"], c["dw"], c["dh"]), err_msg=c["name"])


# ---- per-operation seam vs the oracle on seeded inputs ----------------------------------------------

SCALE_GEOMS = [
    # sw, sh, dw, dh, sr, dr, op, opaque, nonzero dst
    (1920, 1080, 1024, 768, None, None, 0, True, False),    # BASELINE resize: 3-4 x 2-3 taps
    (1920, 1080, 1024, 576, None, None, 0, True, False),    # product default (keep_aspect)
    (1920, 1080, 200, 200, (420, 0, 1500, 1080), None, 0, True, False),  # thumbnail: x5.4, 11 x 11 taps
    (640, 480, 1024, 768, None, None, 0, True, False),      # upscale x1.6
    (854, 480, 1024, 575, None, None, 0, False, False),     # odd width, translucent
    (333, 500, 511, 768, None, None, 1, False, False),
    (97, 61, 31, 200, None, None, 0, False, True),          # Over onto a used frame, translucent
    (97, 61, 31, 200, None, None, 0, True, True),           # Over + opaque source => Src
    (97, 61, 40, 40, (5, 7, 90, 55), (3, 2, 36, 39), 0, False, True),
    (97, 61, 40, 40, None, (-7, -5, 50, 47), 1, False, True),  # dr clipped by dst
    (2, 2, 9, 7, None, None, 0, False, False),
    (1, 5, 4, 9, None, None, 0, False, False),              # 1-wide source: a single tap per column
    (64, 64, 64, 64, None, None, 0, False, True),           # equal size: one tap of weight 1 per axis, Over a used frame
    (3840, 2160, 200, 200, (840, 0, 3000, 2160), None, 0, True, False),   # 4K thumbnail: x10.8, 22-23 taps per axis
    (200, 150, 1024, 768, None, None, 0, False, False),     # upscale x5.12: ten destination rows per source row
]


@pytest.mark.parametrize("g", SCALE_GEOMS, ids=lambda g: "%dx%d->%dx%d" % g[:4])
def test_scale_vs_oracle(ctx, g):
    sw, sh, dw, dh, sr, dr, op, opaque, used = g
    src = rgba_frames(1, sw, sh, seed=sw * 31 + dh, opaque=opaque)[0]
    dst0 = rgba_frames(1, dw, dh, seed=7, opaque=False)[0] if used else np.zeros((dh, dw, 4), np.uint8)
    want = oracle.scale_bilinear(src, dw, dh, sr=sr, dr=dr, op=op, dst=dst0.copy())
    got = ctx.scale_bilinear(src, dw, dh, sr=sr, dr=dr, op=op, dst=dst0.copy())
    np.testing.assert_array_equal(got, want)


def test_scale_rejects_source_rect_outside(ctx, ipx):
    src = rgba_frames(1, 8, 8)[0]
    with pytest.raises(ipx.IpxError) as e:
        ctx.scale_bilinear(src, 4, 4, sr=(-1, 0, 7, 8))
    assert e.value.status == -4


def test_draw_and_glyphs_vs_oracle(ctx):
    src = rgba_frames(1, 300, 120, seed=3, opaque=False)[0]
    for op in (0, 1):
        for r, sp in [((0, 0, 300, 120), (0, 0)), ((10, 5, 280, 100), (3, 9)), ((-5, -5, 400, 400), (0, 0))]:
            d0 = rgba_frames(1, 300, 120, seed=4, opaque=False)[0]
            want = oracle.draw(d0.copy(), r, src, sp, op)
            got = ctx.draw(d0.copy(), r, src, sp, op)
            np.testing.assert_array_equal(got, want)
    frame = rgba_frames(1, 640, 360, seed=5)[0]
    for pos in ("bottom-right", "top-left", "center"):
        gl = text_glyphs(640, 360, position=pos)
        gl.append({"mask": gl[0]["mask"], "dr": (630, 350, 660, 380), "mp": (2, 1)})   # clipped by the frame
        gl.append({"mask": gl[1]["mask"], "dr": (-4, -6, 12, 20), "mp": (0, 0)})
        for col in (DEFAULT_COL, (0, 0, 0, 127), (200, 10, 90, 255)):
            want = oracle.composite_glyphs(frame.copy(), gl, col)
            got = ctx.composite_glyphs(frame.copy(), gl, col)
            np.testing.assert_array_equal(got, want)


# ---- the batched path (one-pass kernel, per-output fallback) vs the oracle -----------------------------------------------

PLAN_CASES = [
    # sw, sh, n, resize, thumbnail, env
    (1920, 1080, 2, (1024, 768, False), (200, True), {}),
    (1920, 1080, 1, (1024, 768, True), (200, True), {}),
    (1920, 1080, 1, (1024, 768, True), (200, False), {"IPX_KS_STRIPS": "4"}),
    (640, 480, 3, (1024, 768, True), (200, True), {}),
    (854, 480, 2, (1024, 768, True), (200, True), {}),                  # rows not 16-byte aligned
    (1080, 1920, 1, (1024, 768, True), (200, True), {"IPX_KS_SPLIT_ROWS": "37"}),
    (3840, 2160, 1, (1024, 768, False), (200, True), {}),
    (332, 500, 2, (1024, 768, True), (64, True), {"IPX_KS_STRIPS": "5", "IPX_KS_SPLIT_ROWS": "13"}),
    (333, 500, 2, (1024, 768, True), (64, True), {}),                   # odd width: per-output kernels
    (200, 200, 1, (200, 200, False), (200, True), {}),                  # scale 1: one tap of weight 1 per axis
    (1280, 720, 2, (100, 30, False), (200, False), {"IPX_KS_SPLIT": "0"}),
    (37, 23, 2, (64, 64, True), (10, True), {}),
    (7680, 4320, 1, (1024, 768, True), (200, True), {}),
]


@pytest.mark.parametrize("case", PLAN_CASES, ids=lambda c: "%dx%dx%d" % c[:3])
def test_plan_vs_oracle(ctx, case, monkeypatch):
    sw, sh, n, resize, thumb, env = case
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    frames = rgba_frames(n, sw, sh, seed=sw + sh)
    glyphs = text_glyphs(sw, sh)
    gs = ctx.glyphset(glyphs, DEFAULT_COL)
    plan = ctx.plan(sw, sh, resize=resize, thumbnail=thumb, watermark=gs)
    got = plan.run_host(frames)
    for i in range(n):
        want = oracle.process(frames[i], resize=resize, thumb=thumb, glyphs=glyphs, col=DEFAULT_COL)
        for k in ("resize", "thumbnail", "watermark"):
            np.testing.assert_array_equal(got[k][i], want[k], err_msg="%s frame %d" % (k, i))
    plan.close()
    gs.close()


def test_plan_subsets_and_unfused(ctx, monkeypatch):
    frames = rgba_frames(2, 320, 200, seed=9, opaque=False)
    glyphs = text_glyphs(320, 200, n=6, width_px=120, height_px=30)
    gs = ctx.glyphset(glyphs, (10, 20, 30, 200))
    want = [oracle.process(f, resize=(100, 100, True), thumb=(50, True), glyphs=glyphs, col=(10, 20, 30, 200))
            for f in frames]
    for env in ({}, {"IPX_NO_FUSE": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for kw in (dict(resize=(100, 100, True), thumbnail=None, watermark=None),
                   dict(resize=None, thumbnail=(50, True), watermark=None),
                   dict(resize=None, thumbnail=None, watermark=gs),
                   dict(resize=(100, 100, True), thumbnail=(50, True), watermark=gs),
                   dict(resize=None, thumbnail=None, watermark=True)):
            plan = ctx.plan(320, 200, **kw)
            got = plan.run_host(frames)
            for i in range(2):
                for k, v in got.items():
                    exp = want[i][k] if not (k == "watermark" and kw["watermark"] is True) else frames[i]
                    np.testing.assert_array_equal(v[i], exp, err_msg="%s %s" % (k, kw))
            plan.close()
    gs.close()


PATH_ENVS = [{"IPX_FUSED": "0"},                                 # per-output kernels (ks_generic_kernel), no one-pass kernel
             {"IPX_KS_FAST": "0"},                               # float64 throughout: no float pass
             {"IPX_KS_FAST": "0", "IPX_KS_STRIPS": "3"},
             {"IPX_KS_FIX_CAP": "7"},                            # the float pass with lists of 7 pixels: frames fill them, their items are redone in float64
             {"IPX_KS_FIX_CAP": "300", "IPX_KS_SPLIT": "1", "IPX_KS_SPLIT_ROWS": "50"},   # some items fit their frame's list, some do not
             {"IPX_KS_SPEC": "0"},                               # the general four-channel kernel alone, no speculative opaque pass
             {"IPX_KS_SPLIT": "0"},                              # one segment per frame even for a small batch
             {"IPX_KS_SPLIT": "1", "IPX_KS_SPLIT_ROWS": "50"},   # many short segments: every destination row near a seam re-stages its rows
             {"IPX_KS_STRIPS": "3"},                             # several column strips (the thumbnail's columns sit in the middle ones)
             {"IPX_KS_STRIPS": "7", "IPX_KS_SPEC": "0"},
             {"IPX_KS_STRIPS": "2", "IPX_KS_SPLIT": "0"}]


@pytest.mark.parametrize("env", PATH_ENVS, ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_every_kernel_path_is_bit_exact(ctx, env, monkeypatch):
    """The shipped default is the one-pass kernel with the float pass on frames taken as opaque; float64 throughout, the
    per-output kernels, the general kernel and other tilings (strips, segments) must give the same bytes.  Frame 1 of every batch is translucent, so the
    speculative pass gives it up and the general kernel redoes its items."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for sw, sh, n, resize in ((1920, 1080, 3, (1024, 768, False)), (1280, 720, 2, (1024, 768, True)),
                              (854, 480, 2, (1024, 768, True))):
        frames = rgba_frames(n, sw, sh, seed=sw)
        frames[1] = rgba_frames(1, sw, sh, seed=sw + 1, opaque=False)[0]
        glyphs = text_glyphs(sw, sh)
        gs = ctx.glyphset(glyphs, DEFAULT_COL)
        plan = ctx.plan(sw, sh, resize=resize, thumbnail=(200, True), watermark=gs)
        got = plan.run_host(frames)
        for i in range(n):
            want = oracle.process(frames[i], resize=resize, thumb=(200, True), glyphs=glyphs, col=DEFAULT_COL)
            for k in ("resize", "thumbnail", "watermark"):
                np.testing.assert_array_equal(got[k][i], want[k], err_msg="%s frame %d %s" % (k, i, env))
        plan.close()
        gs.close()


def test_random_geometries_vs_oracle(ctx, monkeypatch):
    """Seeded sweep over frame sizes (odd widths included), operator parameters (down- and upscales, crop and non-crop
    thumbnails) and tilings: the ownership tables (which strip owns a destination column, which segment a destination row),
    the row tables and both kernels must agree with the oracle on every byte."""
    rng = np.random.default_rng(20261004)
    for trial in range(36):
        sw = int(rng.choice([rng.integers(2, 90), rng.integers(90, 700), 4 * rng.integers(60, 500)]))
        sh = int(rng.choice([rng.integers(2, 60), rng.integers(60, 500)]))
        resize = (int(rng.integers(1, 1300)), int(rng.integers(1, 900)), bool(rng.integers(0, 2)))
        thumb = (int(rng.integers(1, 300)), bool(rng.integers(0, 2)))
        for k in ("IPX_KS_STRIPS", "IPX_KS_SPLIT_ROWS", "IPX_KS_SPLIT", "IPX_KS_SPEC", "IPX_KS_FIX_CAP"):
            monkeypatch.delenv(k, raising=False)
        monkeypatch.delenv("IPX_KS_FAST", raising=False)            # the float pass (the default)
        if rng.random() < 0.5:
            monkeypatch.setenv("IPX_KS_STRIPS", str(int(rng.choice([1, 2, 3, 5, 9]))))
        if rng.random() < 0.5:
            monkeypatch.setenv("IPX_KS_SPLIT_ROWS", str(int(rng.choice([4, 9, 17, 64, 200]))))
        if rng.random() < 0.5:
            monkeypatch.setenv("IPX_KS_SPLIT", str(int(rng.integers(0, 2))))
        if rng.random() < 0.25:
            monkeypatch.setenv("IPX_KS_SPEC", "0")
        pick = rng.random()
        if pick < 0.25:
            monkeypatch.setenv("IPX_KS_FAST", "0")                  # float64 throughout
        elif pick < 0.5:
            monkeypatch.setenv("IPX_KS_FIX_CAP", str(int(rng.choice([1, 5, 40, 400]))))
        n = int(rng.integers(1, 4))
        frames = rgba_frames(n, sw, sh, seed=trial, opaque=bool(rng.integers(0, 2)))
        glyphs = text_glyphs(sw, sh, n=5, width_px=min(60, sw), height_px=min(20, sh))
        gs = ctx.glyphset(glyphs, DEFAULT_COL)
        plan = ctx.plan(sw, sh, resize=resize, thumbnail=thumb, watermark=gs)
        got = plan.run_host(frames)
        for i in range(n):
            want = oracle.process(frames[i], resize=resize, thumb=thumb, glyphs=glyphs, col=DEFAULT_COL)
            for k in ("resize", "thumbnail", "watermark"):
                if k in got:
                    np.testing.assert_array_equal(got[k][i], want[k], err_msg="trial %d %s %dx%d resize=%s thumb=%s env=%s" % (
                        trial, k, sw, sh, resize, thumb, {e: os.environ.get(e) for e in ("IPX_KS_STRIPS", "IPX_KS_SPLIT_ROWS", "IPX_KS_SPLIT", "IPX_KS_SPEC", "IPX_KS_FAST", "IPX_KS_FIX_CAP")}))
        plan.close()
        gs.close()


def test_config1_plumbing_640x480_thumbnail(ctx):
    """BASELINE config 1: one 640x480 frame -> 200x200 thumbnail (crop 480^2 at x=80), GPU vs oracle."""
    frame = rgba_frames(1, 640, 480, seed=1)
    plan = ctx.plan(640, 480, resize=None, thumbnail=(200, True), watermark=None)
    assert (plan.info.thumb_w, plan.info.thumb_h) == (200, 200)
    c = plan.info.thumb_crop
    assert (c.x0, c.y0, c.x1, c.y1) == (80, 0, 560, 480)
    got = plan.run_host(frame)["thumbnail"][0]
    np.testing.assert_array_equal(got, oracle.process(frame[0], thumb=(200, True), want=("thumbnail",))["thumbnail"])
    plan.close()


def test_one_pixel_wide_source_takes_unfused_path(ctx):
    frames = rgba_frames(2, 1, 40, seed=11, opaque=False)
    plan = ctx.plan(1, 40, resize=(8, 8, False), thumbnail=(4, True), watermark=True)
    got = plan.run_host(frames)
    for i in range(2):
        want = oracle.process(frames[i], resize=(8, 8, False), thumb=(4, True))
        for k in ("resize", "thumbnail", "watermark"):
            np.testing.assert_array_equal(got[k][i], want[k])


# ---- BASELINE-size batch: size-independent properties + sampled frames vs the oracle ----------------

def test_full_size_batch_properties(ctx):
    n, sw, sh = 64, 1920, 1080
    rng = np.random.default_rng(1)
    frames = rgba_frames(n, sw, sh, seed=21)
    frames[0][...] = 173            # constant frame: every scaled output equals the constant (K1)
    glyphs = text_glyphs(sw, sh)
    gs = ctx.glyphset(glyphs, DEFAULT_COL)
    plan = ctx.plan(sw, sh, resize=(1024, 768, False), thumbnail=(200, True), watermark=gs)
    assert plan.info.algorithmic_bytes == 19894528
    src = ctx.alloc(frames.nbytes).upload(frames)
    i = plan.info
    res, th, wm = ctx.alloc(n * i.resize_bytes), ctx.alloc(n * i.thumb_bytes), ctx.alloc(n * i.wm_bytes)
    plan.run_dev(n, src.ptr, res.ptr, th.ptr, wm.ptr)
    ctx.sync()
    r = res.download((n, 768, 1024, 4))
    t = th.download((n, 200, 200, 4))
    w = wm.download((n, sh, sw, 4))
    assert (r[0] == 173).all() and (t[0] == 173).all()
    # the watermark frame is the source outside the glyph boxes, in every frame
    box = np.zeros((sh, sw), bool)
    for g in glyphs:
        x0, y0, x1, y1 = g["dr"]
        box[max(y0, 0):y1, max(x0, 0):x1] = True
    assert (w[:, ~box] == frames[:, ~box]).all()
    # opaque sources give opaque outputs
    assert (r[1:, ..., 3] == 255).all() and (t[1:, ..., 3] == 255).all()
    # running the batch again is idempotent (no state carried between launches)
    plan.run_dev(n, src.ptr, res.ptr, th.ptr, wm.ptr)
    ctx.sync()
    assert (res.download((n, 768, 1024, 4)) == r).all()
    # sampled frames, bit for bit against the oracle
    for k in (0, 1, int(rng.integers(2, n)), n - 1):
        want = oracle.process(frames[k], resize=(1024, 768, False), thumb=(200, True), glyphs=glyphs,
                              col=DEFAULT_COL)
        np.testing.assert_array_equal(r[k], want["resize"])
        np.testing.assert_array_equal(t[k], want["thumbnail"])
        np.testing.assert_array_equal(w[k], want["watermark"])
    for b in (src, res, th, wm):
        b.free()


# ---- the headline shapes of BASELINE.json, each on the path the bench times -------------------------------

@pytest.mark.parametrize("resize", [(1024, 768, False), (1024, 768, True)], ids=["1024x768", "keep_aspect-1024x576"])
@pytest.mark.parametrize("env", [{}, {"IPX_FUSED": "0"}, {"IPX_KS_SPEC": "0"}, {"IPX_KS_SPLIT": "0"}, {"IPX_KS_FAST": "0"}],
                         ids=["default", "per-output", "general-kernel", "one-segment", "float64"])
def test_config2_resize_only_1080p_plan(ctx, resize, env, monkeypatch):
    """BASELINE config 2: a resize-only plan on 1920x1080 frames through the batched path (resize.go:61-75,121-125)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n = 3
    frames = rgba_frames(n, 1920, 1080, seed=0xC2)
    plan = ctx.plan(1920, 1080, resize=resize, thumbnail=None, watermark=None)
    assert (plan.info.resize_w, plan.info.resize_h) == ((1024, 768) if not resize[2] else (1024, 576))
    src = ctx.alloc(frames.nbytes).upload(frames)
    out = ctx.alloc(n * plan.info.resize_bytes)
    plan.run_dev(n, src.ptr, out.ptr, None, None)
    ctx.sync()
    got = out.download((n, plan.info.resize_h, plan.info.resize_w, 4))
    for i in range(n):
        np.testing.assert_array_equal(got[i], oracle.process(frames[i], resize=resize, want=("resize",))["resize"], err_msg="frame %d" % i)
    for b in (src, out):
        b.free()
    plan.close()


@pytest.mark.parametrize("workload", ["full", "resize"])
def test_bench_batch_shape_1024_frames(ctx, workload):
    """The batch bench.py times (BASELINE configs 2 and 3): 1024 slots of 1920x1080, 32 seeded frames tiled over them, one launch.  The 32
    distinct frames are compared with the oracle, and every other slot with the slot that holds the same frame: every workgroup of
    the launch, wherever it ran and whatever its neighbours were, produced the bytes of its frame."""
    F, P, sw, sh = 1024, 32, 1920, 1080
    pool = rgba_frames(P, sw, sh, seed=0xB3)
    glyphs = text_glyphs(sw, sh)
    gs = ctx.glyphset(glyphs, DEFAULT_COL)
    full = workload == "full"
    plan = ctx.plan(sw, sh, resize=(1024, 768, False), thumbnail=(200, True) if full else None, watermark=gs if full else None)
    info = plan.info
    assert info.algorithmic_bytes == (19894528 if full else 11440128)
    fbytes = sw * sh * 4
    src = ctx.alloc(F * fbytes).upload(pool)
    for i in range(P, F, P):
        ctx.copy_d2d(src.ptr + i * fbytes, src.ptr, P * fbytes)
    outs = {"resize": ((info.resize_h, info.resize_w, 4), info.resize_bytes)}
    if full:
        outs["thumbnail"] = ((info.thumb_h, info.thumb_w, 4), info.thumb_bytes)
        outs["watermark"] = ((sh, sw, 4), info.wm_bytes)
    bufs = {k: ctx.alloc(F * nb) for k, (_, nb) in outs.items()}
    plan.run_dev(F, src.ptr, bufs["resize"].ptr, bufs["thumbnail"].ptr if full else None, bufs["watermark"].ptr if full else None)
    ctx.sync()
    for k, (shape, nb) in outs.items():
        first = bufs[k].download((P,) + shape)
        for i in range(P):
            want = oracle.process(pool[i], resize=(1024, 768, False), thumb=(200, True), glyphs=glyphs if full else (), col=DEFAULT_COL, want=(k,))[k]
            np.testing.assert_array_equal(first[i], want, err_msg="%s of pool frame %d" % (k, i))
        for j in range(1, F // P):
            chunk = bufs[k].download((P,) + shape, offset=j * P * nb)
            assert np.array_equal(chunk, first), "%s: slots %d..%d differ from slots 0..%d" % (k, j * P, j * P + P - 1, P - 1)
    for b in [src] + list(bufs.values()):
        b.free()
    plan.close()
    gs.close()


def test_config4_4k_batch(ctx):
    """BASELINE config 4's per-GPU shape: a batch of 3840x2160 frames through the full pipeline (64 slots, 8 distinct frames): properties
    on every slot, four frames against the oracle."""
    F, P, sw, sh = 64, 8, 3840, 2160
    pool = rgba_frames(P, sw, sh, seed=0x4C)
    pool[0][...] = 91
    pool[0][..., 3] = 255
    glyphs = text_glyphs(sw, sh)
    gs = ctx.glyphset(glyphs, DEFAULT_COL)
    plan = ctx.plan(sw, sh, resize=(1024, 768, False), thumbnail=(200, True), watermark=gs)
    info = plan.info
    assert info.algorithmic_bytes == 69660928
    fbytes = sw * sh * 4
    src = ctx.alloc(F * fbytes).upload(pool)
    for i in range(P, F, P):
        ctx.copy_d2d(src.ptr + i * fbytes, src.ptr, P * fbytes)
    res, th, wm = ctx.alloc(F * info.resize_bytes), ctx.alloc(F * info.thumb_bytes), ctx.alloc(F * info.wm_bytes)
    plan.run_dev(F, src.ptr, res.ptr, th.ptr, wm.ptr)
    ctx.sync()
    r = res.download((F, 768, 1024, 4))
    t = th.download((F, 200, 200, 4))
    assert (r[0, ..., :3] == 91).all() and (t[0, ..., :3] == 91).all()          # a constant frame scales to the constant (K1)
    assert (r[..., 3] == 255).all() and (t[..., 3] == 255).all()                # opaque in, opaque out
    for j in range(1, F // P):                                                  # a slot equals the slot that holds the same frame
        assert np.array_equal(r[j * P:(j + 1) * P], r[:P]) and np.array_equal(t[j * P:(j + 1) * P], t[:P])
    box = np.zeros((sh, sw), bool)
    for g in glyphs:
        x0, y0, x1, y1 = g["dr"]
        box[max(y0, 0):y1, max(x0, 0):x1] = True
    for slot in (0, 5, 26, F - 1):
        w = wm.download((sh, sw, 4), offset=slot * info.wm_bytes)
        assert (w[~box] == pool[slot % P][~box]).all()                          # the watermark frame is the source outside the glyph boxes
        want = oracle.process(pool[slot % P], resize=(1024, 768, False), thumb=(200, True), glyphs=glyphs, col=DEFAULT_COL)
        np.testing.assert_array_equal(r[slot], want["resize"])
        np.testing.assert_array_equal(t[slot], want["thumbnail"])
        np.testing.assert_array_equal(w, want["watermark"])
    for b in (src, res, th, wm):
        b.free()
    plan.close()
    gs.close()


def test_concurrent_callers(ctx):
    """worker.go:90-96: several goroutines share one processor; lanes=2 forces waiting."""
    import threading
    src = rgba_frames(1, 400, 300, seed=2)[0]
    want = oracle.scale_bilinear(src, 123, 77)
    errs = []

    def work():
        try:
            for _ in range(5):
                np.testing.assert_array_equal(ctx.scale_bilinear(src, 123, 77), want)
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=work) for _ in range(6)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs


def test_mixed_entries_from_many_threads(ctx):
    """Goroutines of one worker use different entries of one context at once: compressed-in / compressed-out (one lane per part, decode
    scratch bumped out of the lane's buffer), the batched host entries (every lane), decode alone (one lane), per-operation calls.
    Every result must be the bytes the same call gives when it runs alone."""
    import io
    import threading
    from PIL import Image
    w, h, n = 320, 200, 20
    frames = rgba_frames(n, w, h, seed=5)
    files = []
    for i in range(n):
        buf = io.BytesIO()
        Image.fromarray(frames[i][..., :3]).save(buf, "JPEG", quality=80 + i % 10, **({"restart_marker_rows": 2} if i % 3 == 0 else {}))
        files.append(buf.getvalue())
    glyphs = text_glyphs(w, h, n=5, width_px=120, height_px=24)
    gs = ctx.glyphset(glyphs, DEFAULT_COL)
    plan = ctx.plan(w, h, resize=(256, 160, False), thumbnail=(64, True), watermark=gs)
    nrgba = np.random.default_rng(1).integers(0, 256, (n, h, w, 4), dtype=np.uint8)
    jobs = {
        "jpeg->jpeg": lambda: plan.run_jpeg_jpeg(files),
        "host rgba": lambda: plan.run_host(frames),
        "host nrgba": lambda: plan.run_host_nrgba(nrgba),
        "host jpeg": lambda: plan.run_host_jpeg(frames, 85),
        "decode": lambda: ctx.jpeg_decode_batch(files),
        "scale": lambda: ctx.scale_bilinear(frames[0], 123, 77),
    }

    def flat(x):
        if isinstance(x, dict):
            return {k: flat(v) for k, v in x.items() if k != "free"}
        if isinstance(x, (list, tuple)):
            return [flat(v) for v in x]
        return x.tobytes() if isinstance(x, np.ndarray) else x
    alone = {k: flat(f()) for k, f in jobs.items()}
    errs = []

    def work(name):
        try:
            for _ in range(4):
                assert flat(jobs[name]()) == alone[name], name
        except Exception as e:  # noqa: BLE001
            errs.append((name, repr(e)[:300]))
    ts = [threading.Thread(target=work, args=(k,)) for k in list(jobs) * 2]
    [t.start() for t in ts]
    [t.join() for t in ts]
    plan.close()
    gs.close()
    assert not errs, errs


@pytest.mark.parametrize("env", [{}, {"IPX_KS_FIX_CAP": "64"}, {"IPX_KS_FAST": "0"}], ids=["float-pass", "float-pass-short-lists", "float64"])
def test_values_exactly_on_a_rounding_boundary(ctx, env, monkeypatch):
    """The float pass (ipx_ks_fused.hip) decides a byte from float sums only when the value is provably clear of a multiple of 256 and
    leaves the rest to float64 (ks_fix_kernel).  Exact 2:1 and 4:1 downscales make the hardest case plentiful: the weights are
    multiples of 1/8 (1/32) per axis, so sum(tap * weight) + 0.5 lands EXACTLY on a multiple of 256 for about one value in 3000 of a
    random frame -- there the byte is whatever the reference's float64 roundings make of it.  All of them must come out as the
    oracle's."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    frames = rgba_frames(3, 1920, 1080, seed=0xB0DA)
    # frame 2: EVERY interior value of the 2:1 output on a boundary.  With a source of period 2 in x and y the 16 taps weigh each of the
    # four tile values by 16 / 64, so sum(tap * weight) = 257 * (a + b + c + d) / 4, and tiles that sum to 510 give 32767.5: + 0.5 is
    # 128 * 256 exactly.  All 64 lanes of a wave are undecided at once, every frame's list overflows, every item is redone in float64.
    rng = np.random.default_rng(0xA11)
    for ch in range(3):
        a = rng.integers(100, 156, (540, 960)); b = rng.integers(100, 156, (540, 960)); c = rng.integers(100, 156, (540, 960))
        d = 510 - a - b - c
        tile = np.empty((1080, 1920), np.int64)
        tile[0::2, 0::2] = a[0, 0]; tile[0::2, 1::2] = b[0, 0]; tile[1::2, 0::2] = c[0, 0]; tile[1::2, 1::2] = d[0, 0]   # one tile, repeated
        frames[2, :, :, ch] = tile.astype(np.uint8)
    plan = ctx.plan(1920, 1080, resize=(960, 540, False), thumbnail=(270, True), watermark=None)
    got = plan.run_host(frames)
    on_boundary = 0
    for i in range(3):
        want = oracle.process(frames[i], resize=(960, 540, False), thumb=(270, True), want=("resize", "thumbnail"))
        for k in ("resize", "thumbnail"):
            np.testing.assert_array_equal(got[k][i], want[k], err_msg="%s frame %d %s" % (k, i, env))
        # how many values of the 2:1 output sit exactly on a boundary (integer arithmetic: weights (1, 3, 3, 1) / 8 per axis)
        f = frames[i][..., :3].astype(np.int64)
        wy = np.array([1, 3, 3, 1])
        rows = sum(wy[k] * np.pad(f, ((1, 1), (0, 0), (0, 0)), mode="edge")[k:k + 1080:2] for k in range(4))[:540]
        cols = sum(wy[k] * np.pad(rows, ((0, 0), (1, 1), (0, 0)), mode="edge")[:, k:k + 1920:2] for k in range(4))[:, :960]
        inner = cols[1:-1, 1:-1]                                     # (edge rows and columns have their own, renormalised weights)
        on_boundary += int(np.count_nonzero((257 * inner + 32) % (256 * 64) == 0))
    assert on_boundary > 100 + 3 * 900 * 500, on_boundary            # the case is really there: hundreds in the random frames, all of frame 2
    plan.close()


def test_the_float_pass_is_what_runs_by_default(ctx, monkeypatch, capfd):
    """IPX_KS_STATS=1 makes the runtime report the float pass's lists after a launch: by default opaque RGBA frames take the float pass
    (a few undecided pixels in ten thousand, nothing redone in float64), with IPX_KS_FAST=0 there is no such report."""
    import re
    frames = rgba_frames(2, 1280, 720, seed=5)
    plan = ctx.plan(1280, 720, resize=(1024, 768, True), thumbnail=(200, True), watermark=None)
    monkeypatch.setenv("IPX_KS_STATS", "1")
    plan.run_host(frames)
    err = capfd.readouterr().err
    m = re.search(r"\[ipx ks stats\] \d+ frames: undecided pixels per frame resize mean ([0-9.]+) max (\d+) \(room (\d+)\).*; (\d+) of (\d+) items redone", err)
    assert m, err[-400:]
    assert 0 < float(m.group(1)) < 0.004 * 1024 * 576 and int(m.group(2)) < int(m.group(3)) and int(m.group(4)) == 0
    monkeypatch.setenv("IPX_KS_FAST", "0")
    plan.run_host(frames)
    assert "ks stats" not in capfd.readouterr().err
    plan.close()


@pytest.mark.parametrize("split", ["1", "0"], ids=["two-lanes-per-column", "one-lane-per-column"])
def test_many_taps_per_column(ctx, split, monkeypatch):
    """Large reductions: 39 and 43 horizontal taps per column (odd counts: the second lane's half ends in a zero-weight tap).  The float
    pass may give such a column to two adjacent lanes (IPX_KS_TAPSPLIT=1: wherever 16 taps and more allow it; 0: never); the bytes are
    the oracle's either way, for RGBA and for decoded-JPEG planes."""
    monkeypatch.setenv("IPX_KS_TAPSPLIT", split)
    for sw, sh, resize, thumb in ((1920, 1080, (100, 60, False), (50, True)), (1600, 900, (96, 54, True), (40, False))):
        frames = rgba_frames(2, sw, sh, seed=sw + int(split))
        plan = ctx.plan(sw, sh, resize=resize, thumbnail=thumb, watermark=None)
        got = plan.run_host(frames)
        for i in range(2):
            want = oracle.process(frames[i], resize=resize, thumb=thumb, want=("resize", "thumbnail"))
            for k in ("resize", "thumbnail"):
                np.testing.assert_array_equal(got[k][i], want[k], err_msg="%s frame %d %dx%d" % (k, i, sw, sh))
        plan.close()
