"""image.Decode for progressive and multi-scan JPEGs (image_processor.go:47 = Go's image/jpeg, scan.go: spectral selection,
successive approximation, EOB runs, refinement passes; reader.go: the marker loop up to EOI).

CPU: the oracle's restatement (oracle/ipx_jpeg_dec_oracle.c, ipxo_jpeg_decode_full) is pinned two ways -- (a) libjpeg writes the SAME
quantised coefficients into the baseline and the progressive file of one image, so the progressive decode must reproduce the
(already pinned) baseline decode EXACTLY wherever the image has pixels; (b) libjpeg itself decodes the progressive files to
within the +-2 of its different IDCT.  PARITY UNPINNED against Go itself (no Go toolchain; tools/gen_go_vectors closes that).
GPU: ipx_jpeg_decode_batch / ipx_plan_run_jpeg_jpeg against that oracle, bit for bit (scans on host threads, transform onwards on
the GPU)."""
import io

import numpy as np
import pytest

import oracle
from test_jpeg_decode import picture, pil_jpeg

SUBS = {0: (1, 1), 1: (2, 1), 2: (2, 2)}      # Pillow's subsampling argument -> (h0, v0)


def _visible(d, w, h, sub):
    h0, v0 = SUBS[sub]
    cw, ch = (w + h0 - 1) // h0, (h + v0 - 1) // v0
    return d["y"][:h, :w], d["cb"][:ch, :cw], d["cr"][:ch, :cw]


@pytest.mark.parametrize("sub", [0, 1, 2])
@pytest.mark.parametrize("size", [(333, 211), (64, 48), (17, 9), (1, 1), (200, 8)])
def test_progressive_decode_equals_the_baseline_decode_of_the_same_coefficients(sub, size):
    w, h = size
    img = picture(w, h, seed=w + sub)
    for q in (85, 30, 97):
        base = oracle.jpeg_decode(pil_jpeg(img, quality=q, subsampling=sub), want_coefs=True)     # the single-scan restatement
        prog = oracle.jpeg_decode(pil_jpeg(img, quality=q, subsampling=sub, progressive=True))
        assert (prog["w"], prog["h"], prog["ratio"]) == (w, h, base["ratio"])
        for a, b in zip(_visible(prog, w, h, sub), _visible(base, w, h, sub)):
            np.testing.assert_array_equal(a, b)


def test_progressive_close_to_libjpeg():
    from PIL import Image
    img = picture(150, 97, seed=3)
    for sub in (0, 1, 2):
        b = pil_jpeg(img, quality=85, subsampling=sub, progressive=True)
        d = oracle.jpeg_decode(b)
        p = Image.open(io.BytesIO(b))
        p.draft("YCbCr", (150, 97))
        p.load()
        ref = np.asarray(p).astype(int)
        diff = np.abs(d["y"][:97, :150].astype(int) - ref[..., 0])
        assert diff.max() <= 2 and diff.mean() < 0.1, (sub, diff.max(), diff.mean())


def test_gray_progressive_and_optimised_tables():
    img = picture(90, 70, seed=5)
    g = img[..., 0]
    base = oracle.jpeg_decode(pil_jpeg(g, quality=80), want_coefs=True)
    prog = oracle.jpeg_decode(pil_jpeg(g, quality=80, progressive=True, optimize=True))
    assert prog["ratio"] == 4
    np.testing.assert_array_equal(prog["y"][:70, :90], base["y"][:70, :90])


def test_marker_loop_rules_of_go():
    """decode() reads segments until EOI: a file that ends after its last scan is io.ErrUnexpectedEOF (the reference marks the task
    failed); bytes that belong to no segment are skipped; a stray RSTn and "\\xff\\x00" between segments are ignored; a second SOF is
    an error; what follows EOI is never read."""
    img = picture(40, 24, seed=9)
    for kw in ({}, {"progressive": True}):
        ok = pil_jpeg(img, quality=85, **kw)
        want = oracle.jpeg_decode(ok)
        assert ok.endswith(b"\xff\xd9")
        with pytest.raises(ValueError, match="malformed"):
            oracle.jpeg_decode(ok[:-2])                                            # no EOI
        same = oracle.jpeg_decode(ok + b"trailing bytes are not read")
        np.testing.assert_array_equal(same["y"], want["y"])
        eoi = len(ok) - 2
        for extra in (b"\x00\x01\x02garbage", b"\xff\x00", b"\xff\xd3", b"\xff\xff\xff"):   # before EOI: skipped / ignored / fill bytes
            got = oracle.jpeg_decode(ok[:eoi] + extra + ok[eoi:])
            np.testing.assert_array_equal(got["y"], want["y"])
        sof = ok.index(b"\xff\xc2" if kw else b"\xff\xc0")
        seg = ok[sof:sof + 2 + int.from_bytes(ok[sof + 2:sof + 4], "big")]
        with pytest.raises(ValueError, match="malformed"):
            oracle.jpeg_decode(ok[:eoi] + seg + ok[eoi:])                          # "multiple SOF markers"


def _without_dqt(f):
    """The file with its DQT segments cut out (what a damaged APP0 length does when Go's liberal marker loop skips over them)."""
    out, i = bytearray(f[:2]), 2
    while f[i + 1] != 0xda:
        n = 2 + int.from_bytes(f[i + 2:i + 4], "big")
        if f[i + 1] != 0xdb:
            out += f[i:i + n]
        i += n
    return bytes(out + f[i:])


def test_undefined_quantisation_table_is_all_zero():
    """Go never checks that a DQT defined the table a component names: it is all zero, every coefficient dequantises to 0 and the picture
    is flat grey (found by tools/fuzz_corrupt.py seed 17, case 345)."""
    img = picture(40, 24, seed=9)
    for kw in ({}, {"progressive": True}):
        d = oracle.jpeg_decode(_without_dqt(pil_jpeg(img, quality=85, **kw)))
        assert (d["y"][:24, :40] == 128).all() and (d["cb"][:12, :20] == 128).all() and (d["cr"][:12, :20] == 128).all()


def _eob_becomes_a_run(f, sym=0x10):
    """The AC tables' end-of-block symbol 0x00 replaced by an end-of-band-run symbol (r = 1..14, s = 0)."""
    out, i = bytearray(f), 2
    while out[i + 1] != 0xda:
        n = 2 + int.from_bytes(out[i + 2:i + 4], "big")
        if out[i + 1] == 0xc4:
            k = i + 4
            while k < i + n:
                total = sum(out[k + 1:k + 17])
                if out[k] >> 4 == 1:
                    for v in range(k + 17, k + 17 + total):
                        if out[v] == 0:
                            out[v] = sym
                k += 17 + total
        i += n
    return bytes(out)


def test_end_of_band_run_in_a_sequential_file():
    """scan.go runs one block loop for every kind of scan: in a baseline file too an AC symbol (r < 15, s = 0) opens an end-of-band run
    and the following blocks lose their AC part (found by tools/fuzz_corrupt.py seed 23: a bit flip in a DHT).  The single-scan
    restatement used for the encoder pin reads it as a plain EOB; the full decoder is the one that follows Go."""
    img = picture(64, 48, seed=2)
    f = _eob_becomes_a_run(pil_jpeg(img, quality=85))
    full = oracle.jpeg_decode(f)
    plain = oracle.jpeg_decode(f, want_coefs=True)
    assert not np.array_equal(full["y"], plain["y"])
    assert np.array_equal(full["y"][:8, :8], plain["y"][:8, :8])          # the first block is still whole


# ---- GPU ------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ctx():
    import imageprocessor_amd as ipx
    c = ipx.Context()
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sub", [0, 1, 2])
def test_gpu_progressive_batch_vs_oracle(ctx, sub):
    w, h = 333, 211
    files = [pil_jpeg(picture(w, h, seed=60 + i), quality=70 + 3 * i, subsampling=sub, progressive=True, optimize=bool(i & 1)) for i in range(6)]
    files.insert(2, pil_jpeg(picture(w, h, seed=99), quality=85, subsampling=sub))                      # a baseline file among them
    files.insert(5, files[1][:len(files[1]) - 2])                                                       # a progressive file without EOI
    files.insert(6, files[0][:len(files[0]) // 2])                                                      # one whose scans run out
    info, st = ctx.jpeg_decode_batch(files)
    assert st == [0, 0, 0, 0, 0, -1, -1, 0, 0], st
    for i, f in enumerate(files):
        if st[i]:
            with pytest.raises(ValueError, match="malformed"):
                oracle.jpeg_decode(f)
            continue
        want = oracle.jpeg_decode(f)
        for k in ("y", "cb", "cr"):
            np.testing.assert_array_equal(info[k][i], want[k], err_msg="file %d %s" % (i, k))


@pytest.mark.gpu
def test_gpu_progressive_gray_and_marker_rules(ctx):
    g = picture(120, 80, seed=4)[..., 0]
    files = [pil_jpeg(g, quality=85, progressive=True), pil_jpeg(g, quality=60, progressive=True, optimize=True), pil_jpeg(g, quality=85)]
    eoi = len(files[2]) - 2
    files.append(files[2][:eoi] + b"\x01\x02skipped" + files[2][eoi:])          # bytes that belong to no segment before EOI
    files.append(files[2][:-2])                                                 # a baseline file without EOI: malformed for Go, malformed here
    files.append(_without_dqt(files[2]))                                        # flat grey, not an error
    files.append(_without_dqt(files[0]))
    files.append(_eob_becomes_a_run(files[2]))                                  # an end-of-band run in a baseline file: Go's one block loop
    files.append(files[2][:2] + b"\xff\xa0\x00\x02" + files[2][2:])             # "unknown marker": FormatError below 0xc0 ...
    files.append(files[2][:2] + b"\xff\xc9\x00\x02" + files[2][2:])             # ... UnsupportedError above
    info, st = ctx.jpeg_decode_batch(files)
    assert st == [0, 0, 0, 0, -1, 0, 0, 0, -1, -4], st
    assert (info["y"][5][:80, :120] == 128).all()
    for f, v in ((files[8], "malformed"), (files[9], "unsupported")):
        with pytest.raises(ValueError, match=v):
            oracle.jpeg_decode(f)
    for i in (0, 1, 2, 3, 5, 6, 7):
        np.testing.assert_array_equal(info["y"][i], oracle.jpeg_decode(files[i])["y"], err_msg="file %d" % i)


@pytest.mark.gpu
def test_progressive_uploads_through_compressed_in_compressed_out(ctx):
    """The worker's whole job for progressive uploads -- the common web case: decode (host scans + GPU transform), operators, three
    jpeg.Encode, against the oracle's decoder + operators + encoder."""
    from helpers import DEFAULT_COL, text_glyphs
    from test_sources_gpu import _expect_ycbcr_ops
    w, h = 320, 200
    files = [pil_jpeg(picture(w, h, seed=70 + i), quality=80 + i, progressive=bool(i % 3)) for i in range(8)]
    glyphs = text_glyphs(w, h, n=6, width_px=150, height_px=30)
    gs = ctx.glyphset(glyphs, DEFAULT_COL)
    plan = ctx.plan(w, h, resize=(512, 384, True), thumbnail=(100, True), watermark=gs)
    got, st = plan.run_jpeg_jpeg(files)
    assert st == [0] * 8
    for k, f in enumerate(files):
        d = oracle.jpeg_decode(f)
        ch, cw = (h + 1) // 2, (w + 1) // 2
        want = _expect_ycbcr_ops(np.ascontiguousarray(d["y"][:h, :w]), np.ascontiguousarray(d["cb"][:ch, :cw]), np.ascontiguousarray(d["cr"][:ch, :cw]),
                                 2, (512, 384, True), (100, True), glyphs, DEFAULT_COL)
        for key in got:
            assert got[key][k] == oracle.jpeg_encode_rgba(want[key], 85), (key, k)
    plan.close()
    gs.close()


@pytest.mark.gpu
def test_damaged_progressive_files_never_disagree(ctx):
    """tools/fuzz_corrupt.py's rule for the host scan decoder: same verdict as the oracle (or the file handed back, -4, where the oracle
    says malformed), and where a damaged file still decodes every byte is the same."""
    rng = np.random.default_rng(11)
    img = picture(333, 250, seed=8, noise=10.0)
    clean = [pil_jpeg(img, quality=85, progressive=True), pil_jpeg(img, quality=90, subsampling=0, optimize=True, progressive=True),
             pil_jpeg(img[..., 0], quality=80, progressive=True)]
    seen = {0: 0, -1: 0, -4: 0}
    for t in range(45):
        f = bytearray(clean[t % 3])
        sos = f.index(b"\xff\xda")
        kind = (t // 3) % 5
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
        info, st = ctx.jpeg_decode_batch([f, clean[t % 3]])
        if want is not None and info is not None and (want["w"], want["h"], want["ratio"]) != (info["w"], info["h"], info["ratio"]):
            continue
        if want is not None and want.get("dc_wide"):
            exp = -4
        assert st[0] == exp or (exp != 0 and st[0] in (-1, -4)), (t, kind, exp, st)   # which of Go's two error kinds a broken file earns is not part of the contract
        assert st[1] == 0
        seen[st[0]] += 1
        if exp == 0:
            for k in ("y", "cb", "cr") if want["ratio"] != 4 else ("y",):
                np.testing.assert_array_equal(info[k][0], want[k], err_msg="case %d plane %s" % (t, k))
    assert seen[0] >= 5 and seen[-1] + seen[-4] >= 5, seen
