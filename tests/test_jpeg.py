"""jpeg.Encode, the last step of every operator (resize.go:80, thumbnail.go:70, watermark.go:68; Quality 85).

Go's image/jpeg writer is restated twice: oracle/ipx_jpeg_oracle.c (scalar, the checker) and the product
(transform on the GPU in csrc/ipx_jpeg.hip, tables / headers / entropy coder on the host in
csrc/ipx_jpeg_host.cpp).  PARITY UNPINNED against Go itself (no toolchain); what pins the oracle:
  * libjpeg (through Pillow) writes the SAME scan bytes as the Gray path of the oracle for every image and
    quality tried, and the same DQT / DHT payloads: that covers fdct, the quantiser's rounding, the Huffman
    tables, run lengths, bit packing, 0xff stuffing, edge replication and the final padding;
  * colour conversion: known answers of color.RGBToYCbCr (primaries, greys) and a decode through Pillow.
The product is then held to the oracle byte for byte (host half here; the GPU half in the gpu-marked tests).
"""
import io

import numpy as np
import pytest

import oracle


def segments(b):
    """[(marker, payload)] plus ('scan', entropy-coded bytes)"""
    assert b[:2] == b"\xff\xd8" and b[-2:] == b"\xff\xd9"
    i, out = 2, []
    while True:
        assert b[i] == 0xFF
        m, n = b[i + 1], (b[i + 2] << 8) | b[i + 3]
        out.append((m, b[i + 4:i + 2 + n]))
        i += 2 + n
        if m == 0xDA:
            out.append(("scan", b[i:-2]))
            return out


def _images():
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:120, 0:200]
    return {"noise 64x48": rng.integers(0, 256, (48, 64), dtype=np.uint8),
            "noise 37x29": rng.integers(0, 256, (29, 37), dtype=np.uint8),
            "smooth 200x120": ((np.sin(xx / 9.0) + np.cos(yy / 7.0)) * 60 + 128).clip(0, 255).astype(np.uint8),
            "flat 16x16": np.full((16, 16), 77, np.uint8),
            "one pixel": np.array([[200]], np.uint8),
            "extremes 24x8": np.tile(np.array([0, 255], np.uint8), (8, 12))}


@pytest.mark.parametrize("quality", [85, 50, 20, 95, 100, 1])
def test_gray_scan_is_libjpegs(quality):
    from PIL import Image
    for name, g in _images().items():
        mine = segments(oracle.jpeg_encode_gray(g, quality))
        buf = io.BytesIO()
        Image.fromarray(g, "L").save(buf, "JPEG", quality=quality, optimize=False)
        ref = segments(buf.getvalue())
        assert dict(mine)["scan"] == dict(ref)["scan"], (name, quality)
        assert dict(mine)[0xDB][1:65] == dict(ref)[0xDB][1:65], "luminance DQT"
        assert b"".join(p for m, p in mine if m == 0xC4) == b"".join(p for m, p in ref if m == 0xC4), "DHT"


def test_colour_quant_tables_are_libjpegs():
    from PIL import Image
    for q in (85, 75, 30, 98):
        buf = io.BytesIO()
        Image.new("RGB", (16, 16)).save(buf, "JPEG", quality=q, optimize=False)
        qt = Image.open(io.BytesIO(buf.getvalue())).quantization   # natural order in recent Pillow
        mine = oracle.jpeg_quant(q)
        zig = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
               35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
        for i in (0, 1):
            ref = list(qt[i])
            assert list(mine[i]) in ([ref[z] for z in zig], ref), q


def test_rgba_stream_layout_and_known_colours():
    """Stream layout as Go writes it (SOI DQT SOF0 DHT SOS ... EOI, no APPn) and color.RGBToYCbCr known answers read
    back from the DC terms: a flat frame has DC = div(64*(v - 128), 8*q0) per block and no AC."""
    q = oracle.jpeg_quant(85)
    for rgb, ycc in (((255, 0, 0), (76, 85, 255)), ((0, 255, 0), (150, 44, 21)), ((0, 0, 255), (29, 255, 107)),
                     ((255, 255, 255), (255, 128, 128)), ((0, 0, 0), (0, 128, 128)), ((128, 128, 128), (128, 128, 128))):
        f = np.zeros((16, 16, 4), np.uint8)
        f[..., :3] = rgb
        f[..., 3] = 255
        data, coefs = oracle.jpeg_encode_rgba(f, 85, want_coefs=True)
        assert [m for m, _ in segments(data)] == [0xDB, 0xC0, 0xC4, 0xDA, "scan"]
        sof = dict(segments(data))[0xC0]
        assert sof == bytes([8, 0, 16, 0, 16, 3, 1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1])

        def dc(v, q0):
            a, b = 64 * (v - 128), 8 * int(q0)
            return (a + b // 2) // b if a >= 0 else -((-a + b // 2) // b)
        assert coefs.shape == (1, 6, 64) and not coefs[0, :, 1:].any()
        assert [int(c) for c in coefs[0, :, 0]] == [dc(ycc[0], q[0][0])] * 4 + [dc(ycc[1], q[1][0]), dc(ycc[2], q[1][0])], rgb


def test_rgba_decodes_to_the_source():
    """Pillow decodes the oracle's stream to within the quantisation error of the source (PSNR > 30 dB on a smooth frame)."""
    from PIL import Image
    yy, xx = np.mgrid[0:180, 0:250]
    f = np.stack([(np.sin(xx / 23.0) * 100 + 128), (np.cos(yy / 17.0) * 100 + 128), ((xx + yy) % 256), np.full(xx.shape, 255)], -1)
    f = f.clip(0, 255).astype(np.uint8)
    data = oracle.jpeg_encode_rgba(f, 85)
    back = np.asarray(Image.open(io.BytesIO(data)).convert("RGB")).astype(np.float64)
    mse = ((back - f[..., :3]) ** 2).mean()
    assert back.shape == (180, 250, 3) and 10 * np.log10(255 ** 2 / mse) > 30


def test_host_entropy_coder_matches_oracle():
    """csrc/ipx_jpeg_host.cpp on the oracle's own coefficients: identical bytes (headers, Huffman coding, stuffing, padding)."""
    import imageprocessor_amd as ipx
    rng = np.random.default_rng(11)
    for (w, h), q in (((64, 48), 85), ((37, 29), 85), ((1, 1), 85), ((200, 33), 40), ((48, 48), 100), ((31, 17), 1)):
        f = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        if q == 100:
            f[..., :3] = rng.integers(0, 2, (h, w, 3), dtype=np.uint8) * 255   # large coefficients, long codes, many 0xff
        want, coefs = oracle.jpeg_encode_rgba(f, q, want_coefs=True)
        assert ipx.jpeg_entropy_encode(coefs, w, h, q) == want, (w, h, q)
        np.testing.assert_array_equal(ipx.jpeg_quant_tables(q), oracle.jpeg_quant(q))


def test_quantiser_reciprocal_is_exact():
    """The kernel divides with mulhi(n, ceil(2^32 / d)): exact whenever n * (d - 1) < 2^32, i.e. for every divisor 8*q <= 2040 and every
    n < 2^20; checked exhaustively over n < 2^17 (all the transform can produce) for a spread of divisors."""
    n = np.arange(0, 1 << 17, dtype=np.uint64)
    for d in list(range(8, 2041, 136)) + [16, 24, 1016, 2032, 2040]:
        m = ((1 << 32) + d - 1) // d
        assert np.array_equal((n * np.uint64(m)) >> np.uint64(32), n // np.uint64(d)), d


def test_too_large_and_bad_arguments():
    """jpeg.Encode refuses 65536 pixels on a side ("jpeg: image is too large to encode"); so does the host half."""
    import ctypes as C
    import imageprocessor_amd as ipx
    L = ipx.lib()
    coefs = np.zeros(384, np.int16)
    out, n = C.c_void_p(), C.c_size_t()
    assert L.ipx_jpeg_entropy_encode(coefs.ctypes.data, 1 << 16, 16, 85, C.byref(out), C.byref(n)) == -1
    assert b"too large" in L.ipx_last_error()
    assert L.ipx_jpeg_entropy_encode(None, 16, 16, 85, C.byref(out), C.byref(n)) == -1
    assert L.ipx_jpeg_coef_count(1920, 1080) == 120 * 68 * 384 and L.ipx_jpeg_coef_count(0, 5) == 0


# ---- GPU half ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ctx():
    import imageprocessor_amd as ipx
    c = ipx.Context()
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(16, 16), (128, 16), (129, 17), (640, 360), (1024, 576), (200, 200), (1, 1), (15, 33), (1920, 1080),
                                   (257, 40)], ids=lambda s: "%dx%d" % s)
def test_gpu_encode_is_byte_exact(ctx, shape):
    from helpers import rgba_frames
    w, h = shape
    for seed, opaque in ((1, True), (2, False)):
        f = rgba_frames(1, w, h, seed=seed, opaque=opaque)[0]
        for q in ((85, 30) if w * h < 100000 else (85,)):
            assert ctx.jpeg_encode(f, q) == oracle.jpeg_encode_rgba(f, q), (shape, seed, q)


@pytest.mark.gpu
def test_gpu_encode_structured_frames(ctx):
    """Smooth gradients (long zero runs, ZRL), flat frames (EOB only), black / white checker (largest coefficients)."""
    yy, xx = np.mgrid[0:96, 0:160]
    frames = [np.stack([xx * 255 // 159, yy * 255 // 95, (xx + yy) % 256, np.full(xx.shape, 255)], -1).astype(np.uint8),
              np.full((96, 160, 4), 200, np.uint8),
              np.repeat(np.repeat(((xx // 1 + yy // 1) % 2 * 255)[..., None], 4, -1), 1, 0).astype(np.uint8)]
    for f in frames:
        for q in (85, 100, 5):
            assert ctx.jpeg_encode(np.ascontiguousarray(f), q) == oracle.jpeg_encode_rgba(f, q)


@pytest.mark.gpu
def test_gpu_batch_encode_from_hbm(ctx):
    """The operator outputs stay in HBM and are encoded from there: resize + thumbnail + watermark, then three jpeg.Encode."""
    from PIL import Image
    from helpers import DEFAULT_COL, rgba_frames, text_glyphs
    w, h, n = 640, 360, 5
    frames = rgba_frames(n, w, h, seed=5)
    glyphs = text_glyphs(w, h)
    gs = ctx.glyphset(glyphs, DEFAULT_COL)
    plan = ctx.plan(w, h, resize=(1024, 768, True), thumbnail=(200, True), watermark=gs)
    i = plan.info
    src = ctx.alloc(frames.nbytes).upload(frames)
    res, th, wm = ctx.alloc(n * i.resize_bytes), ctx.alloc(n * i.thumb_bytes), ctx.alloc(n * i.wm_bytes)
    plan.run_dev(n, src.ptr, res.ptr, th.ptr, wm.ptr)
    ctx.sync()
    for buf, (ow, oh), key in ((res, (i.resize_w, i.resize_h), "resize"), (th, (i.thumb_w, i.thumb_h), "thumbnail"), (wm, (w, h), "watermark")):
        got = ctx.jpeg_encode_batch_dev(buf.ptr, ow, oh, n, 85)
        for k in range(n):
            want = oracle.process(frames[k], resize=(1024, 768, True), thumb=(200, True), glyphs=glyphs, col=DEFAULT_COL)[key]
            assert got[k] == oracle.jpeg_encode_rgba(want, 85), (key, k)
        assert Image.open(io.BytesIO(got[0])).size == (ow, oh)
    plan.close()
    gs.close()


@pytest.mark.gpu
@pytest.mark.parametrize("host_entropy,fused_len", [("0", "1"), ("0", "0"), ("1", "1")], ids=["gpu-entropy", "gpu-entropy-separate-sizing-pass", "host-entropy"])
def test_gpu_batch_entropy_paths(ctx, host_entropy, fused_len, monkeypatch):
    """The entropy coders behind ipx_jpeg_encode_batch_dev: sizing (inside the transform kernel by default, or as the earlier separate pass
    over the coefficients) + scan + placement + stuffing on the GPU, and the host loop over downloaded coefficients.  Frames of one batch
    differ wildly in stream length; q=100 on black / white noise gives the longest codes and the most 0xff bytes; 1x1 and 17x9 frames
    are a single (partial) MCU."""
    from helpers import rgba_frames
    monkeypatch.setenv("IPX_JPEG_HOST_ENTROPY", host_entropy)
    monkeypatch.setenv("IPX_JPEG_FUSED_LEN", fused_len)
    rng = np.random.default_rng(21)
    for (w, h), q in (((320, 200), 85), ((64, 64), 100), ((17, 9), 85), ((1, 1), 50), ((640, 360), 20), ((1024, 64), 95)):
        frames = rgba_frames(4, w, h, seed=w + q)
        frames[1][...] = 128                                                     # flat: EOB only
        frames[2][..., :3] = rng.integers(0, 2, (h, w, 3), dtype=np.uint8) * 255   # binary noise
        yy, xx = np.mgrid[0:h, 0:w]
        frames[3][..., 0] = (xx * 3) % 256
        frames[3][..., 1] = (yy * 5) % 256
        frames[3][..., 2] = ((xx + yy) // 2) % 256
        src = ctx.alloc(frames.nbytes).upload(frames)
        got = ctx.jpeg_encode_batch_dev(src.ptr, w, h, 4, q)
        for k in range(4):
            want = oracle.jpeg_encode_rgba(frames[k], q)
            assert got[k] == want, ((w, h), q, k, len(got[k]), len(want))
        src.free()


@pytest.mark.gpu
def test_host_frames_to_jpeg_streams(ctx):
    """ipx_plan_run_host_jpeg: decoded frames in host memory -> the three encoded objects the worker stores
    (image_processor.go:64-77), chunked over the lanes; byte-exact against oracle operators + oracle encoder."""
    from helpers import DEFAULT_COL, rgba_frames, text_glyphs
    w, h, n = 640, 360, 11
    frames = rgba_frames(n, w, h, seed=9)
    glyphs = text_glyphs(w, h)
    gs = ctx.glyphset(glyphs, DEFAULT_COL)
    for kw, chunk in ((dict(resize=(1024, 768, True), thumbnail=(200, True), watermark=gs), "4"), (dict(resize=None, thumbnail=(64, False)), "32")):
        import os
        os.environ["IPX_HOST_CHUNK_JPEG"] = chunk
        plan = ctx.plan(w, h, **kw)
        got = plan.run_host_jpeg(frames, 85)
        for k in range(n):
            want = oracle.process(frames[k], resize=kw.get("resize") or (1, 1, False), thumb=kw.get("thumbnail") or (1, False),
                                  glyphs=glyphs if "watermark" in kw else [], col=DEFAULT_COL)
            for key in got:
                assert got[key][k] == oracle.jpeg_encode_rgba(want[key], 85), (key, k)
        assert set(got) == {x for x in kw if kw[x] is not None}
        plan.close()
    os.environ.pop("IPX_HOST_CHUNK_JPEG", None)
    gs.close()


@pytest.mark.gpu
def test_decoded_jpeg_planes_to_jpeg_streams(ctx):
    """ipx_plan_run_host_ycbcr_jpeg: *image.YCbCr planes in, encoded objects out -- the reference's per-operator conversion rules
     followed by jpeg.Encode, against the oracle's composition of the same helpers."""
    from helpers import DEFAULT_COL, text_glyphs
    from test_sources_gpu import _expect_ycbcr_ops, _rand_ycbcr
    for (w, h, ratio) in ((640, 360, 2), (320, 240, 0), (333, 251, 1)):   # the last one takes the three-kernel operator path
        n = 5
        planes = [_rand_ycbcr(w, h, ratio, 70 + i) for i in range(n)]
        y = np.stack([p[0] for p in planes]); cb = np.stack([p[1] for p in planes]); cr = np.stack([p[2] for p in planes])
        glyphs = text_glyphs(w, h, n=6, width_px=min(150, w), height_px=min(30, h))
        gs = ctx.glyphset(glyphs, DEFAULT_COL)
        plan = ctx.plan(w, h, resize=(512, 384, True), thumbnail=(100, True), watermark=gs)
        got = plan.run_host_ycbcr_jpeg(y, cb, cr, ratio, 85)
        for k in range(n):
            want = _expect_ycbcr_ops(y[k], cb[k], cr[k], ratio, (512, 384, True), (100, True), glyphs, DEFAULT_COL)
            for key in ("resize", "thumbnail", "watermark"):
                assert got[key][k] == oracle.jpeg_encode_rgba(want[key], 85), (w, h, ratio, key, k)
        plan.close()
        gs.close()
