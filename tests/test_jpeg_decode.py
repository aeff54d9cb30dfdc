"""image.Decode for JPEGs (image_processor.go:47 -> Go's image/jpeg): baseline decoder.

oracle/ipx_jpeg_dec_oracle.c restates reader.go / scan.go / huffman.go / idct.go; the product parses on the host and
decodes on the GPU (csrc/ipx_jpeg_dec_host.cpp, csrc/ipx_jpeg_dec.hip).  PARITY UNPINNED against Go itself.  Pins of
the oracle: decode(encode(x)) returns the pinned encoder's coefficients exactly (Huffman decoding, de-zig-zag, DC
prediction), and libjpeg (Pillow) decodes the same files to within +-1..2 of its different IDCT.
"""
import io

import numpy as np
import pytest

import oracle

ZIG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
       35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def picture(w, h, seed=0, noise=8.0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([np.sin(xx / 9.0 + seed) * 100 + 128, np.cos(yy / 7.0) * 100 + 128, (xx * 2 + yy + 31 * seed) % 256], -1)
    return (img + rng.normal(0, noise, img.shape)).clip(0, 255).astype(np.uint8)


def pil_jpeg(img, **kw):
    from PIL import Image, ImageFile
    ImageFile.MAXBLOCK = 1 << 25    # libjpeg's optimize pass needs the whole file in one buffer ("Suspension not allowed here")
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, "JPEG", **kw)
    return buf.getvalue()


def test_decode_of_our_encoder_returns_its_coefficients():
    for (w, h), q in (((150, 97), 85), ((16, 16), 50), ((33, 70), 100), ((1, 1), 85)):
        rgb = picture(w, h, seed=w)
        rgba = np.concatenate([rgb, np.full((h, w, 1), 255, np.uint8)], -1)
        data, coefs = oracle.jpeg_encode_rgba(rgba, q, want_coefs=True)
        d = oracle.jpeg_decode(data, want_coefs=True)
        enc = coefs.reshape(-1, 64)
        nat = np.zeros_like(enc)
        nat[:, ZIG] = enc
        np.testing.assert_array_equal(d["coefs"][:enc.size].reshape(-1, 64), nat)
        assert (d["w"], d["h"], d["ratio"]) == (w, h, 2)
        assert d["y"].shape == (16 * ((h + 15) // 16), 16 * ((w + 15) // 16)) and d["cb"].shape == (8 * ((h + 15) // 16), 8 * ((w + 15) // 16))


@pytest.mark.parametrize("sub", [0, 1, 2], ids=["444", "422", "420"])
def test_close_to_libjpeg(sub):
    """Same file through libjpeg (Pillow, luma of the YCbCr draft mode): the two integer IDCTs differ by at most 2."""
    from PIL import Image
    img = picture(150, 97, seed=4)
    for kw in ({}, {"restart_marker_blocks": 3}, {"optimize": True}, {"quality": 30}):
        b = pil_jpeg(img, subsampling=sub, **{"quality": 85, **kw})
        d = oracle.jpeg_decode(b)
        p = Image.open(io.BytesIO(b))
        p.draft("YCbCr", (150, 97))
        p.load()
        assert p.mode == "YCbCr" and d["ratio"] == sub
        ref = np.asarray(p).astype(int)
        diff = np.abs(d["y"][:97, :150].astype(int) - ref[..., 0])
        assert diff.max() <= 2 and diff.mean() < 0.1, (sub, kw, diff.max(), diff.mean())


def test_unsupported_and_malformed():
    from PIL import Image
    img = picture(64, 48)
    for blob, what in ((pil_jpeg(img)[:200], "malformed"),
                       (b"not a jpeg at all", "malformed")):
        with pytest.raises(ValueError, match=what):
            oracle.jpeg_decode(blob)
    buf = io.BytesIO()
    Image.fromarray(np.dstack([img, img[..., :1]]), "CMYK").save(buf, "JPEG")
    with pytest.raises(ValueError, match="unsupported"):
        oracle.jpeg_decode(buf.getvalue())
    ok = pil_jpeg(img)
    with pytest.raises(ValueError, match="malformed"):
        oracle.jpeg_decode(ok[:len(ok) // 2])                                       # scan data runs out
    dup = bytearray(ok)
    sof = dup.index(b"\xff\xc0")
    assert dup[sof + 10] == 1 and dup[sof + 13] == 2
    dup[sof + 13] = 1                                                               # processSOF: "repeated component identifier"
    with pytest.raises(ValueError, match="malformed"):
        oracle.jpeg_decode(bytes(dup))


# ---- GPU --------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ctx():
    import imageprocessor_amd as ipx
    c = ipx.Context()
    yield c
    c.close()


def _check_batch(ctx, files, expect_status=None):
    info, st = ctx.jpeg_decode_batch(files)
    for i, f in enumerate(files):
        try:
            want = oracle.jpeg_decode(f)
        except ValueError as e:
            assert st[i] == (-1 if "malformed" in str(e) else -4), (i, st[i], str(e))
            continue
        if expect_status and expect_status[i]:
            assert st[i] == expect_status[i]
            continue
        if want["dc_wide"]:       # a DC value beyond int16: Go (int32) decodes on, the GPU pipeline hands the file back
            assert st[i] == -4, (i, st[i])
            continue
        assert st[i] == 0, (i, st[i])
        assert (info["w"], info["h"], info["ratio"]) == (want["w"], want["h"], want["ratio"])
        for k in ("y", "cb", "cr") if want["ratio"] != 4 else ("y",):
            np.testing.assert_array_equal(info[k][i], want[k], err_msg="%s of file %d" % (k, i))
    return info, st


@pytest.mark.gpu
@pytest.mark.parametrize("shared,piece", [("1", "1"), ("1", "0"), ("0", "0")], ids=["piece-kernels", "bytewise-shared-tables-kernel", "bytewise-per-lane-tables-kernel"])
def test_gpu_decode_same_tables_batch(ctx, shared, piece, monkeypatch):
    """A batch whose files all carry the Annex K tables: restart intervals and short scans go through the piece kernels (unstuffed copy,
    word-wise reader, one pass -- the default); IPX_JPEG_PIECE=0 keeps the byte-wise kernel with one shared table copy per workgroup,
    IPX_JPEG_SHARED_TABLES=0 forces the byte-wise per-lane-table kernel on the same batch."""
    monkeypatch.setenv("IPX_JPEG_SHARED_TABLES", shared)
    monkeypatch.setenv("IPX_JPEG_PIECE", piece)
    for (w, h, n) in ((320, 200, 300), (1920, 1080, 3), (17, 9, 5)):
        files = [pil_jpeg(picture(w, h, seed=i, noise=3.0 + 9 * (i % 5)), quality=[85, 60, 95][i % 3], **({"restart_marker_blocks": 7} if i % 4 == 0 else {}))
                 for i in range(n)]
        _check_batch(ctx, files)


@pytest.mark.gpu
@pytest.mark.parametrize("sub", [0, 1, 2], ids=["444", "422", "420"])
def test_gpu_decode_matches_oracle(ctx, sub):
    for (w, h) in ((150, 97), (640, 360), (16, 8), (1920, 1080)):
        n = 3 if w * h > 500000 else 70   # more than one wave of images
        files = []
        for i in range(n):
            kw = [{}, {"restart_marker_blocks": 5}, {"optimize": True}, {"quality": 100}, {"quality": 25, "restart_marker_rows": 1}][i % 5]
            files.append(pil_jpeg(picture(w, h, seed=i, noise=4.0 + 10 * (i % 4)), subsampling=sub, **{"quality": 85, **kw}))
        _check_batch(ctx, files)


@pytest.mark.gpu
@pytest.mark.parametrize("stage", ["0", "1"], ids=["scan-through-l2", "scan-rows-in-lds"])
@pytest.mark.parametrize("sub_bytes", ["128", "256", "512", "1024"])
def test_parallel_huffman_passes_at_every_sub_sequence_size(ctx, monkeypatch, sub_bytes, stage):
    """The decoder that is parallel inside a scan, with each sub-sequence length it may choose for a batch and with the scan bytes of a
    wave read through L1 / L2 or staged as LDS rows (what small batches get): the same planes as the oracle either way."""
    monkeypatch.setenv("IPX_JPEG_PAR_SUB", sub_bytes)
    monkeypatch.setenv("IPX_JPEG_PAR_STAGE", stage)
    files = [pil_jpeg(picture(1920, 1080, seed=20 + i, noise=3.0 + 9 * i), subsampling=2, quality=(85, 100, 40)[i], optimize=(i == 1)) for i in range(3)]
    files += [pil_jpeg(picture(1920, 1080, seed=29, noise=1.0), subsampling=2, quality=10)]     # a short scan: few sub-sequences, the last one ragged
    _check_batch(ctx, files)


@pytest.mark.gpu
def test_gpu_decode_of_go_style_streams_and_440(ctx):
    """Streams as Go's own encoder writes them (no JFIF, both tables, 4:2:0), and 4:4:0 built by transposing a 4:2:2 file's role:
    Pillow cannot write 4:4:0, so the sampling bytes of a 4:2:2 file are swapped -- a valid 4:4:0 stream of other content."""
    files = []
    for i in range(5):
        rgb = picture(200, 120, seed=10 + i)
        files.append(oracle.jpeg_encode_rgba(np.concatenate([rgb, np.full((120, 200, 1), 255, np.uint8)], -1), 85))
    _check_batch(ctx, files)
    b = bytearray(pil_jpeg(picture(64, 64, seed=1), subsampling=1, quality=90))
    i = b.index(b"\xff\xc0")
    assert b[i + 11] == 0x21
    b[i + 11] = 0x12                      # Y sampling 1 x 2: 4:4:0.  64x64: 4 x 8 MCUs of 16x8 become 8 x 4 MCUs of 8x16, same block count
    _check_batch(ctx, [bytes(b)] * 2)


@pytest.mark.gpu
def test_a_sampling_factor_of_three_is_refused_for_a_single_component_too(ctx):
    """processSOF returns errUnsupportedSubsamplingRatio for h == 3 or v == 3 BEFORE it sets a single component's (h, v) to (1, 1): a Gray
    file whose V_1 is 3 fails in Go although its data would decode (tools/fuzz_corrupt.py found the host parser accepting it)."""
    good = pil_jpeg(picture(96, 64)[..., 0])
    i = good.index(b"\xff\xc0")
    assert good[i + 9] == 1 and good[i + 11] == 0x11
    for hv in (0x13, 0x31, 0x33):
        f = bytearray(good)
        f[i + 11] = hv
        with pytest.raises(ValueError, match="unsupported"):
            oracle.jpeg_decode(bytes(f))
        info, st = ctx.jpeg_decode_batch([good, bytes(f)])
        assert st == [0, -4]
    f = bytearray(good)
    f[i + 11] = 0x42                                      # any other factor of a single component is as good as (1, 1)
    _check_batch(ctx, [good, bytes(f)])


@pytest.mark.gpu
def test_gpu_decode_statuses(ctx):
    """A batch is one size and one sampling; the odd ones out, Gray files and broken files get a status and the rest still decode."""
    good = [pil_jpeg(picture(96, 64, seed=i), quality=85) for i in range(4)]
    other_size = pil_jpeg(picture(64, 64), quality=85)
    other_sampling = pil_jpeg(picture(96, 64), quality=85, subsampling=0)
    progressive = pil_jpeg(picture(96, 64), progressive=True)
    gray = pil_jpeg(picture(96, 64)[..., 0])          # decodable, but not in a batch of colour files
    truncated = good[0][:len(good[0]) // 2]
    garbage = b"\xff\xd8" + bytes(100)
    files = [good[0], other_size, good[1], other_sampling, progressive, gray, truncated, garbage, good[2], good[3]]
    info, st = ctx.jpeg_decode_batch(files)
    assert st == [0, -4, 0, -4, 0, -4, -1, -1, 0, 0]       # the progressive file decodes too (its scans on the host, the rest on the GPU)
    for i in (0, 2, 4, 8, 9):
        want = oracle.jpeg_decode(files[i])
        for k in ("y", "cb", "cr"):
            np.testing.assert_array_equal(info[k][i], want[k], err_msg="file %d %s" % (i, k))
    info, st = ctx.jpeg_decode_batch([progressive, progressive])
    assert st == [0, 0] and np.array_equal(info["y"][1], oracle.jpeg_decode(progressive)["y"])
    # asking for a size: everything else is refused
    info, st = ctx.jpeg_decode_batch([other_size, good[0]], w=96, h=64)
    assert st == [-4, 0]


@pytest.mark.gpu
def test_decode_operators_encode_entirely_on_the_gpu(ctx):
    """Compressed bytes up, planes stay in HBM, ipx_plan_run_dev_ycbcr on them, streams down: against the oracle's decoder +
    operators (per-operator YCbCr rules) + encoder."""
    from helpers import DEFAULT_COL, text_glyphs
    from test_sources_gpu import _expect_ycbcr_ops
    w, h, n = 640, 360, 6
    files = [pil_jpeg(picture(w, h, seed=20 + i), quality=90) for i in range(n)]
    info, st = ctx.jpeg_decode_batch(files, download=False)
    assert st == [0] * n
    b = info["batch"]
    glyphs = text_glyphs(w, h, n=6, width_px=150, height_px=30)
    gs = ctx.glyphset(glyphs, DEFAULT_COL)
    plan = ctx.plan(w, h, resize=(512, 384, True), thumbnail=(100, True), watermark=gs)
    i_ = plan.info
    res, th, wm = ctx.alloc(n * i_.resize_bytes), ctx.alloc(n * i_.thumb_bytes), ctx.alloc(n * i_.wm_bytes)
    plan.run_dev_ycbcr(n, b.y, b.cb, b.cr, b.ratio, b.ystride, b.cstride, b.y_frame_stride, b.c_frame_stride, res.ptr, th.ptr, wm.ptr)
    ctx.sync()
    streams = {"resize": ctx.jpeg_encode_batch_dev(res.ptr, i_.resize_w, i_.resize_h, n), "thumbnail": ctx.jpeg_encode_batch_dev(th.ptr, i_.thumb_w, i_.thumb_h, n),
               "watermark": ctx.jpeg_encode_batch_dev(wm.ptr, w, h, n)}
    info["free"]()
    for k in range(n):
        d = oracle.jpeg_decode(files[k])
        # the operators see the *image.YCbCr with its MCU-padded strides; the oracle helpers take tight planes
        ch, cw = (h + 1) // 2, (w + 1) // 2
        want = _expect_ycbcr_ops(np.ascontiguousarray(d["y"][:h, :w]), np.ascontiguousarray(d["cb"][:ch, :cw]), np.ascontiguousarray(d["cr"][:ch, :cw]),
                                 2, (512, 384, True), (100, True), glyphs, DEFAULT_COL)
        for key in streams:
            assert streams[key][k] == oracle.jpeg_encode_rgba(want[key], 85), (key, k)
    plan.close()
    gs.close()


@pytest.mark.gpu
def test_compressed_in_compressed_out(ctx):
    """ipx_plan_run_jpeg_jpeg: what the worker does per message (image_processor.go:41-77) for a batch, without the pixels ever
    leaving HBM; files the GPU decoder refuses come back with a status and no outputs."""
    from helpers import DEFAULT_COL, text_glyphs
    from test_sources_gpu import _expect_ycbcr_ops
    w, h = 320, 200
    files = [pil_jpeg(picture(w, h, seed=40 + i), quality=80 + i) for i in range(7)]
    files.insert(3, pil_jpeg(picture(w, h), progressive=True))
    files.insert(5, pil_jpeg(picture(64, 64)))
    glyphs = text_glyphs(w, h, n=6, width_px=150, height_px=30)
    gs = ctx.glyphset(glyphs, DEFAULT_COL)
    import os
    for chunk, part in (("256", "384"), ("4", "384"), ("256", "3")):   # the last one splits the nine files over three lanes
        os.environ["IPX_JPEG_JPEG_CHUNK"] = chunk
        os.environ["IPX_JPEG_JPEG_PART"] = part
        plan = ctx.plan(w, h, resize=(512, 384, True), thumbnail=(100, True), watermark=gs)
        got, st = plan.run_jpeg_jpeg(files)
        assert st == [0, 0, 0, 0, 0, -4, 0, 0, 0]           # the progressive file (index 3) goes through as well
        for k, f in enumerate(files):
            if st[k]:
                assert all(got[key][k] is None for key in got)
                continue
            d = oracle.jpeg_decode(f)
            ch, cw = (h + 1) // 2, (w + 1) // 2
            want = _expect_ycbcr_ops(np.ascontiguousarray(d["y"][:h, :w]), np.ascontiguousarray(d["cb"][:ch, :cw]), np.ascontiguousarray(d["cr"][:ch, :cw]),
                                     2, (512, 384, True), (100, True), glyphs, DEFAULT_COL)
            for key in ("resize", "thumbnail", "watermark"):
                assert got[key][k] == oracle.jpeg_encode_rgba(want[key], 85), (key, k)
        plan.close()
    os.environ.pop("IPX_JPEG_JPEG_CHUNK", None)
    os.environ.pop("IPX_JPEG_JPEG_PART", None)
    gs.close()


@pytest.mark.gpu
def test_randomised_codec_sweep(ctx):
    """Seeded sweep over sizes (1..200 px, partial MCUs in both directions), samplings, qualities 1..100, restart intervals and
    optimised tables: decode on the GPU == oracle decoder, and re-encoding the decoded picture on the GPU == oracle encoder."""
    rng = np.random.default_rng(20261004)
    by_shape = {}
    for t in range(60):
        w, h = int(rng.integers(1, 200)), int(rng.integers(1, 200))
        sub = int(rng.integers(0, 3))
        kw = {"quality": int(rng.integers(1, 101)), "subsampling": sub}
        r = rng.random()
        if r < 0.3:
            kw["restart_marker_blocks"] = int(rng.integers(1, 9))
        elif r < 0.5:
            kw["restart_marker_rows"] = int(rng.integers(1, 3))
        if rng.random() < 0.3:
            kw["optimize"] = True
        noise = float(rng.choice([0.0, 3.0, 25.0, 90.0]))
        by_shape.setdefault((w, h, sub), []).append(pil_jpeg(picture(w, h, seed=t, noise=noise), **kw))
        if rng.random() < 0.5:   # a second file of the same shape with other tables / quality: exercises the per-lane-table kernel
            kw2 = dict(kw, quality=int(rng.integers(1, 101)), optimize=True)
            by_shape[(w, h, sub)].append(pil_jpeg(picture(w, h, seed=t + 100, noise=noise), **kw2))
    for (w, h, sub), files in by_shape.items():
        info, st = _check_batch(ctx, files)
        # re-encode the first decoded picture (as RGBA through the oracle's DrawYCbCr) on the GPU
        d = oracle.jpeg_decode(files[0])
        chh, cww = {0: (h, w), 1: (h, (w + 1) // 2), 2: ((h + 1) // 2, (w + 1) // 2)}[sub]
        rgba = oracle.draw_ycbcr(np.zeros((h, w, 4), np.uint8), (0, 0, w, h), np.ascontiguousarray(d["y"][:h, :w]),
                                 np.ascontiguousarray(d["cb"][:chh, :cww]), np.ascontiguousarray(d["cr"][:chh, :cww]), sub)
        q = int(rng.integers(1, 101))
        assert ctx.jpeg_encode(rgba, q) == oracle.jpeg_encode_rgba(rgba, q), (w, h, sub, q)


def test_gray_close_to_libjpeg():
    from PIL import Image
    g = picture(150, 97, seed=2)[..., 0]
    for kw in ({}, {"restart_marker_blocks": 4}, {"optimize": True, "quality": 40}):
        b = pil_jpeg(g, **{"quality": 85, **kw})
        d = oracle.jpeg_decode(b)
        assert d["ratio"] == 4 and d["y"].shape == (104, 152)     # *image.Gray with Stride 8 * mxx
        diff = np.abs(d["y"][:97, :150].astype(int) - np.asarray(Image.open(io.BytesIO(b))).astype(int))
        assert diff.max() <= 2 and diff.mean() < 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("flat", ["gray", "flat", "0"], ids=["planar-pass-gray-source", "planar-pass-flat-chroma", "expanded-rgba-pass"])
def test_gray_jpegs_on_the_gpu(ctx, flat, monkeypatch):
    """One-component files: decoded on the GPU into an *image.Gray plane and read as (y, y, y, 0xff) -- what image/draw's drawGray and
    x/image's scale_RGBA_Gray_Src read -- by the converted-tile kernel's Gray source (the Y plane alone, the default), by its YCbCr
    source with a stride-0 row of 128s as chroma (IPX_GRAY_SRC=0), or expanded to RGBA8 and run through the RGBA pass (IPX_GRAY_FLAT=0);
    against the oracle on the expanded frame."""
    from helpers import DEFAULT_COL, text_glyphs
    monkeypatch.setenv("IPX_GRAY_FLAT", "0" if flat == "0" else "1")
    monkeypatch.setenv("IPX_GRAY_SRC", "1" if flat == "gray" else "0")
    w, h = 320, 200
    files = [pil_jpeg(picture(w, h, seed=60 + i)[..., 0], quality=70 + 5 * i, **({"restart_marker_rows": 2} if i == 1 else {})) for i in range(5)]
    files.append(pil_jpeg(picture(w, h, seed=3)))                       # a colour file in a Gray batch: refused
    info, st = _check_batch(ctx, files, expect_status=[0, 0, 0, 0, 0, -4])
    assert st == [0, 0, 0, 0, 0, -4] and info["ratio"] == 4
    for big in ((1920, 1080), (17, 9)):
        _check_batch(ctx, [pil_jpeg(picture(*big, seed=5)[..., 0], quality=85)] * 2)
    glyphs = text_glyphs(w, h, n=6, width_px=150, height_px=30)
    gs = ctx.glyphset(glyphs, DEFAULT_COL)
    for resize, thumb in (((512, 384, True), (100, True)), ((500, 301, False), (90, False)), ((160, 100, False), (50, True))):
        plan = ctx.plan(w, h, resize=resize, thumbnail=thumb, watermark=gs)
        got, st = plan.run_jpeg_jpeg(files)
        assert st == [0, 0, 0, 0, 0, -4]
        for k in range(5):
            y = oracle.jpeg_decode(files[k])["y"][:h, :w]
            rgba = np.dstack([y, y, y, np.full_like(y, 255)])
            want = oracle.process(rgba, resize=resize, thumb=thumb, glyphs=glyphs, col=DEFAULT_COL)
            for key in ("resize", "thumbnail", "watermark"):
                assert got[key][k] == oracle.jpeg_encode_rgba(want[key], 85), (key, k, resize, thumb)
        plan.close()
    gs.close()


@pytest.mark.gpu
def test_baseline_config_1_single_jpeg_to_thumbnail(ctx):
    """BASELINE.json configs[0]: "Single 640x480 JPEG -> 200x200 thumbnail" as the worker does it (image.Decode, Thumbnailer with
    crop_to_fit, jpeg.Encode at 85), here entirely on the GPU: file in, file out, against the oracle's decoder, crop-copy + scale and
    encoder.  (The crop thumbnail of a YCbCr source converts the crop to RGBA8 first: thumbnail.go:128-131.)"""
    from PIL import Image
    w, h = 640, 480
    f = pil_jpeg(picture(w, h, seed=99, noise=5.0), quality=90)
    plan = ctx.plan(w, h, resize=None, thumbnail=(200, True))
    got, st = plan.run_jpeg_jpeg([f])
    assert st == [0] and set(got) == {"thumbnail"}
    d = oracle.jpeg_decode(f)
    crop, tw, th = oracle.thumb_geometry(w, h, 200, True)
    assert crop == (80, 0, 560, 480) and (tw, th) == (200, 200)
    cs = crop[2] - crop[0]
    cropped = oracle.scale_bilinear_ycbcr(np.ascontiguousarray(d["y"][:h, :w]), np.ascontiguousarray(d["cb"][:h // 2, :w // 2]),
                                          np.ascontiguousarray(d["cr"][:h // 2, :w // 2]), 2, cs, cs, sr=crop)
    want = oracle.jpeg_encode_rgba(oracle.scale_bilinear(cropped, tw, th), 85)
    assert got["thumbnail"][0] == want
    assert Image.open(io.BytesIO(got["thumbnail"][0])).size == (200, 200)
    plan.close()


@pytest.mark.gpu
def test_damaged_files_never_disagree(ctx):
    """Bit flips, random bytes, truncation and missing chunks (tools/fuzz_corrupt.py holds the long version: 800 cases, no mismatch):
    the GPU decoder agrees with the oracle's verdict -- or hands the file back to Go's decoder (-4) where the oracle says malformed --
    and where a damaged file still decodes, every byte matches (the self-synchronising decoder converges to the serial decoding of
    whatever bits there are)."""
    rng = np.random.default_rng(5)
    img = picture(333, 250, seed=8, noise=10.0)
    clean = [pil_jpeg(img, quality=85), pil_jpeg(img, quality=85, restart_marker_rows=1), pil_jpeg(img, quality=90, subsampling=0, optimize=True),
             pil_jpeg(img[..., 0], quality=80)]
    for t in range(40):
        f = bytearray(clean[t % 4])
        sos = f.index(b"\xff\xda")
        kind = t % 5
        if kind == 0:
            for _ in range(3):
                f[int(rng.integers(sos + 14, len(f) - 2))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            f[int(rng.integers(sos + 14, len(f) - 2))] = int(rng.integers(0, 256))
        elif kind == 2:
            f[int(rng.integers(2, sos + 14))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 3:
            f = f[:int(rng.integers(sos, len(f)))]
        else:
            a = int(rng.integers(sos + 14, len(f) - 10))
            del f[a:a + int(rng.integers(1, 1500))]
        f = bytes(f)
        try:
            want, exp = oracle.jpeg_decode(f), 0
        except ValueError as e:
            want, exp = None, -1 if "malformed" in str(e) else -4
        info, st = ctx.jpeg_decode_batch([f, clean[t % 4]])
        if want is not None and info is not None and (want["w"], want["h"], want["ratio"]) != (info["w"], info["h"], info["ratio"]):
            continue                      # the damage changed the size or kind: the one-size-per-batch rule decides, not the decoder
        if want is not None and want["dc_wide"]:
            exp = -4                      # decodable by Go's int32 arithmetic only
        assert st[0] == exp or (exp != 0 and st[0] in (-1, -4)), (t, kind, exp, st)   # which of Go's two error kinds a broken file earns is not part of the contract
        assert st[1] == 0
        if exp == 0:
            for k in ("y", "cb", "cr") if want["ratio"] != 4 else ("y",):
                np.testing.assert_array_equal(info[k][0], want[k], err_msg="case %d plane %s" % (t, k))


@pytest.mark.gpu
def test_first_failing_restart_interval_decides(ctx):
    """Two damaged restart intervals in one file: a sequential decoder (Go's, the oracle) stops at the first, so the first one's verdict is
    the file's -- whichever of the independently decoded pieces reports last on the GPU (found by tools/fuzz_corrupt.py seed 29)."""
    rng = np.random.default_rng(2)
    img = picture(200, 128, seed=4, noise=10.0)
    clean = pil_jpeg(img, quality=85, restart_marker_rows=1)
    sos = clean.index(b"\xff\xda") + 14
    marks = [i for i in range(sos, len(clean) - 2) if clean[i] == 0xff and 0xd0 <= clean[i + 1] <= 0xd7]
    assert len(marks) == 7

    def verdict(f):
        try:
            oracle.jpeg_decode(f)
            return 0
        except ValueError as e:
            return -1 if "malformed" in str(e) else -4

    def damage(lo, hi, want):      # one changed byte between lo and hi that gives the verdict `want` on its own
        for _ in range(4000):
            f = bytearray(clean)
            p = int(rng.integers(lo, hi))
            f[p] = int(rng.integers(0, 255))
            if f[p] == 0xff or f[p] == clean[p] or f[p - 1] == 0xff:
                continue
            if verdict(bytes(f)) == want:
                return p, f[p]
        raise AssertionError("no such damage found")

    for first, second in ((-4, -1), (-1, -4)):
        p0, v0 = damage(marks[1] + 2, marks[2], first)
        p1, v1 = damage(marks[4] + 2, marks[5], second)
        f = bytearray(clean)
        f[p0], f[p1] = v0, v1
        f = bytes(f)
        assert verdict(f) == first
        _, st = ctx.jpeg_decode_batch([f, clean])
        assert list(st) == [first, 0], (first, second, st)


@pytest.mark.gpu
def test_large_files(ctx):
    """Thousands of sub-sequences per scan (several workgroups per image in the parallel decoder), 16-bit-wide block counts, high-entropy
    scans: 4K 4:2:0, 8K 4:4:4, and a 1080p file at quality 100 over noise (a 4 MB scan); next to a small file in its own batch."""
    for (w, h), kw, noise in (((3840, 2160), dict(quality=90, subsampling=2), 10.0), ((7680, 4320), dict(quality=50, subsampling=0), 4.0),
                              ((1920, 1080), dict(quality=100, subsampling=2), 90.0), ((1920, 1080), dict(quality=97, subsampling=1, optimize=True), 30.0)):
        f = pil_jpeg(picture(w, h, seed=w + h, noise=noise), **kw)
        info, st = _check_batch(ctx, [f, f])
        assert list(st) == [0, 0], (w, h, kw)


@pytest.mark.gpu
def test_batch_with_many_table_sets(ctx):
    """Files with their own (optimised) Huffman tables next to files with the Annex K ones, with and without restart intervals: the piece
    kernels share one table set per workgroup, so the host groups the pieces by table set and pads the groups."""
    files = []
    for i in range(40):
        kw = [{}, {"optimize": True}, {"restart_marker_rows": 1}, {"optimize": True, "restart_marker_blocks": 3}][i % 4]
        files.append(pil_jpeg(picture(96, 80, seed=i, noise=2.0 + 5 * (i % 7)), quality=50 + i, **kw))
    _check_batch(ctx, files)
    big = [pil_jpeg(picture(640, 480, seed=i, noise=6.0), quality=80 + i, optimize=bool(i & 1), restart_marker_rows=1) for i in range(6)]
    _check_batch(ctx, big)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"IPX_JPEG_PAR": "0"}, {"IPX_JPEG_PAR": "0", "IPX_JPEG_PIECE": "0"}], ids=["parallel", "pieces", "bytewise"])
def test_dc_beyond_int16_goes_back_to_go(ctx, env, monkeypatch):
    """tests/golden/damaged_gray_scan.jpg (found by tools/fuzz_corrupt.py, seed 101): bit flips push the running DC value of a Gray file
    past 32767.  Go keeps DC predictions in int32 and decodes the file (the oracle does the same and says so in dc_wide); the GPU keeps
    coefficients in int16, so every decoder path has to report IPX_ERR_UNSUPPORTED -- not "malformed", and never wrapped pixels."""
    import os
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    f = open(os.path.join(os.path.dirname(__file__), "golden", "damaged_gray_scan.jpg"), "rb").read()
    assert oracle.jpeg_decode(f)["dc_wide"]
    ok = pil_jpeg(picture(500, 333, seed=1)[..., 0], quality=80)
    _, st = ctx.jpeg_decode_batch([f, ok, f])
    assert list(st) == [-4, 0, -4]
