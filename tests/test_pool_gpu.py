"""The multi-device pool and the asynchronous job entries (include/ipx.h, ipx_pool_* / ipx_job_*) on the one GPU of the box: a pool
over devices {0, 0} is two contexts, six feeder threads and one largest-first queue -- what a one-process, N-goroutine worker
(worker.go:88-96) would bind on an 8-GPU node.  Every output is compared with the oracle, bit for bit."""
import threading

import numpy as np
import pytest

import oracle
from helpers import DEFAULT_COL, rgba_frames, text_glyphs

pytestmark = pytest.mark.gpu

RESIZE, THUMB = (1024, 768, True), (200, True)


@pytest.fixture(scope="module")
def ipx():
    import imageprocessor_amd as m
    return m


@pytest.fixture(scope="module")
def pool(ipx):
    p = ipx.Pool(devices=(0, 0), lanes_per_device=2, lane_bytes=64 << 20)
    yield p
    p.close()


def _check(frames, got, glyphs, resize=RESIZE, thumb=THUMB, every=1):
    for i in range(0, frames.shape[0], every):
        want = oracle.process(frames[i], resize=resize, thumb=thumb, glyphs=glyphs, col=DEFAULT_COL)
        for k in ("resize", "thumbnail", "watermark"):
            np.testing.assert_array_equal(got[k][i], want[k], err_msg="frame %d %s" % (i, k))


def test_pool_mixed_sizes_vs_oracle(pool):
    """Several jobs of different frame sizes in flight at once: the queue runs the expensive chunks first, both slots pull."""
    assert pool.slots() == 2
    before = [pool.frames_done(s) for s in range(2)]
    jobs = []
    for sw, sh, n in ((1920, 1080, 12), (640, 480, 9), (3840, 2160, 3), (854, 480, 5)):
        frames = rgba_frames(n, sw, sh, seed=sw + n)
        glyphs = text_glyphs(sw, sh)
        jobs.append((frames, glyphs, pool.submit(frames, resize=RESIZE, thumbnail=THUMB, glyphs=glyphs, col=DEFAULT_COL)))
    total = 0
    for frames, glyphs, job in jobs:
        _check(frames, job.wait(), glyphs, every=2)
        total += frames.shape[0]
    done = [pool.frames_done(s) - b for s, b in zip(range(2), before)]
    assert sum(done) == total and min(done) > 0, done          # every frame once, and both slots took part


def test_pool_async_tickets_from_many_threads(pool):
    """Goroutine-style use: callers submit, do something else, then wait; tickets complete independently."""
    frames = rgba_frames(6, 800, 600, seed=3)
    glyphs = text_glyphs(800, 600, n=6, width_px=160, height_px=30)
    want = [oracle.process(f, resize=(400, 300, False), thumb=(100, True), glyphs=glyphs, col=DEFAULT_COL) for f in frames]
    errs = []

    def caller():
        try:
            for _ in range(3):
                job = pool.submit(frames, resize=(400, 300, False), thumbnail=(100, True), glyphs=glyphs, col=DEFAULT_COL)
                job.done()                     # polling never blocks
                got = job.wait()
                for i in range(frames.shape[0]):
                    for k in ("resize", "thumbnail", "watermark"):
                        assert np.array_equal(got[k][i], want[i][k]), (i, k)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e)[:300])
    ts = [threading.Thread(target=caller) for _ in range(5)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs


def test_pool_pinned_staging_and_operator_subsets(pool):
    src = pool.host_alloc(1, (4, 360, 640, 4))
    src[:] = rgba_frames(4, 640, 360, seed=8)
    got = pool.submit(src, resize=(320, 180, False), thumbnail=None, watermark=True).wait()
    assert set(got) == {"resize", "watermark"}
    for i in range(4):
        want = oracle.process(src[i], resize=(320, 180, False), thumb=(1, False), want=("resize",))
        np.testing.assert_array_equal(got["resize"][i], want["resize"])
        np.testing.assert_array_equal(got["watermark"][i], src[i])       # no glyphs: draw.Draw's copy
    pool.host_free(src)


def test_pool_jpeg_jobs_match_the_single_context_entry(ipx, pool):
    """IPX_JOB_JPEG: files in, three streams per file out, through whichever slot pulls the chunk -- the bytes ipx_plan_run_jpeg_jpeg
    of one context gives for the same files (which tests/test_jpeg_decode.py holds against the oracle)."""
    w, h, n = 640, 480, 24
    frames = rgba_frames(n, w, h, seed=11)
    glyphs = text_glyphs(w, h, n=6, width_px=160, height_px=30)
    with ipx.Context(device=0) as ctx:
        files = [ctx.jpeg_encode(frames[i], 85) for i in range(n)]
        files[5] = files[5][:200]                      # a truncated upload: its status says so, the others are unaffected
        gs = ctx.glyphset(glyphs, DEFAULT_COL)
        plan = ctx.plan(w, h, resize=(320, 240, True), thumbnail=(100, True), watermark=gs)
        want, want_st = plan.run_jpeg_jpeg(files, 85)
        plan.close()
        gs.close()
    got, st = pool.submit_jpeg(files, w, h, 85, resize=(320, 240, True), thumbnail=(100, True), glyphs=glyphs, col=DEFAULT_COL).wait()
    assert st == want_st and st[5] != 0 and sum(1 for v in st if v) == 1
    for k in ("resize", "thumbnail", "watermark"):
        assert got[k] == want[k], k


def test_pool_errors_come_back_as_status(ipx, pool):
    frames = rgba_frames(1, 64, 64, seed=1)
    with pytest.raises(ipx.IpxError) as e:           # an output beyond the addressable span: the worker keeps its CPU path
        pool.submit(frames, resize=(40000, 40000, False), thumbnail=None, out={"resize": np.empty((1, 1, 1, 4), np.uint8)}).wait()
    assert e.value.status == -4
    # the pool is fine afterwards
    got = pool.submit(frames, resize=(32, 32, False), thumbnail=None).wait()
    np.testing.assert_array_equal(got["resize"][0], oracle.process(frames[0], resize=(32, 32, False), thumb=(1, False), want=("resize",))["resize"])


def test_pool_jobs_of_the_other_decoded_types(pool):
    """IPX_JOB_NRGBA8 / GRAY8 / NRGBA64 / RGBA64 / GRAY16 / CMYK: frames of every packed image type image.Decode returns, Pix as Go
    holds it, through the queue; the outputs are the oracle's for that type (the per-type rules of tests/test_sources_gpu.py and
    tests/test_deep_gpu.py)."""
    sw, sh, n = 640, 360, 5
    resize, thumb = (320, 200, False), (64, True)
    glyphs = text_glyphs(sw, sh, n=6, width_px=150, height_px=30)
    rng = np.random.default_rng(77)
    nrgba = rng.integers(0, 256, (n, sh, sw, 4), dtype=np.uint8)
    gray = rng.integers(0, 256, (n, sh, sw), dtype=np.uint8)
    v64 = rng.integers(0, 65536, (n, sh, sw, 4), dtype=np.uint16)
    p64 = np.minimum(v64, v64[..., 3:4])
    g16 = rng.integers(0, 65536, (n, sh, sw), dtype=np.uint16)
    cmyk = rng.integers(0, 256, (n, sh, sw, 4), dtype=np.uint8)
    deep = {"nrgba64": (v64, oracle.DEEP_NRGBA64), "rgba64": (p64, oracle.DEEP_RGBA64), "gray16": (g16, oracle.DEEP_GRAY16), "cmyk": (cmyk, oracle.DEEP_CMYK)}
    jobs = {"nrgba": pool.submit(nrgba, resize=resize, thumbnail=thumb, glyphs=glyphs, col=DEFAULT_COL, kind="nrgba"),
            "gray": pool.submit(gray, resize=resize, thumbnail=thumb, glyphs=glyphs, col=DEFAULT_COL, kind="gray")}
    pix = {}
    for name, (vals, dk) in deep.items():
        pix[name] = np.stack([oracle.deep_pix(vals[i], dk) for i in range(n)])
        jobs[name] = pool.submit(pix[name], resize=resize, thumbnail=thumb, glyphs=glyphs, col=DEFAULT_COL, kind=name)
    nw, nh = oracle.resize_dims(sw, sh, *resize)
    crop, tw, th = oracle.thumb_geometry(sw, sh, *thumb)
    cs = crop[2] - crop[0]
    zeros = lambda: np.zeros((sh, sw, 4), np.uint8)      # noqa: E731
    for name, job in jobs.items():
        got = job.wait()
        for i in (0, n - 1):
            if name == "nrgba":
                want = (oracle.scale_bilinear_nrgba(nrgba[i], nw, nh), oracle.scale_bilinear(oracle.scale_bilinear_nrgba(nrgba[i], cs, cs, sr=crop), tw, th),
                        oracle.draw_nrgba(zeros(), (0, 0, sw, sh), nrgba[i]))
            elif name == "gray":
                rgba = np.dstack([gray[i]] * 3 + [np.full_like(gray[i], 255)])
                o = oracle.process(rgba, resize=resize, thumb=thumb, glyphs=[], col=DEFAULT_COL)
                want = (o["resize"], o["thumbnail"], rgba)
            else:
                dk = deep[name][1]
                want = (oracle.scale_bilinear_deep(pix[name][i], dk, nw, nh),
                        oracle.scale_bilinear(oracle.scale_bilinear_deep(pix[name][i], dk, cs, cs, sr=crop), tw, th),
                        oracle.draw_deep(zeros(), (0, 0, sw, sh), pix[name][i], dk))
            np.testing.assert_array_equal(got["resize"][i], want[0], err_msg="%s resize %d" % (name, i))
            np.testing.assert_array_equal(got["thumbnail"][i], want[1], err_msg="%s thumbnail %d" % (name, i))
            np.testing.assert_array_equal(got["watermark"][i], oracle.composite_glyphs(want[2].copy(), glyphs, DEFAULT_COL), err_msg="%s watermark %d" % (name, i))
