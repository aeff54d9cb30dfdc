"""BASELINE config 5 on the GPU: a seeded mixed-size batch (all six sizes from 854x480 to 7680x4320, incl. 2560x1440) through
shard.WorkQueue + MixedBatch -- the exact code path of `bench.py --mixed` (pull scheduling over the largest-first order, two
streams, the next chunk claimed while the previous one runs) -- with frames of every size compared with the oracle, bit for bit.
Reference semantics: every message is independent (worker.go:112-149), every operator reads the original frame
(image_processor.go:64-65)."""
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


def _make_frames(si, w, h, k):
    return rgba_frames(k, w, h, seed=0x1F00D + 16 * si)


def test_mixed_batch_every_size_vs_oracle(ipx):
    from imageprocessor_amd import shard
    rng = np.random.default_rng(0x51)
    draw = [int(v) for v in rng.integers(0, len(shard.MIXED_SIZES), 42)]
    for si in range(len(shard.MIXED_SIZES)):          # the seeded draw covers all six sizes; keep that true if the sizes ever change
        if si not in draw:
            draw.append(si)
    with ipx.Context(device=0) as ctx:
        # 96 MiB chunks: several chunks per size, so both streams carry every size and chunks of one size follow each other
        mb = shard.MixedBatch(ctx, _make_frames, text_glyphs, DEFAULT_COL, resize=RESIZE, thumbnail=THUMB, chunk_bytes=96 << 20,
                              max_chunk=8, distinct=3)
        items = mb.items_for(draw)
        assert sum(m for _, m in items) == len(draw)
        costs = [shard.frame_cost(*shard.MIXED_SIZES[si]) * m for si, m in items]
        assert costs == sorted(costs, reverse=True)            # largest first
        q = shard.WorkQueue(len(items), chunk=1)
        checked = {si: 0 for si in range(len(shard.MIXED_SIZES))}
        want_cache = {}

        def on_done(si, m, s):
            if checked[si] >= 2:
                return
            checked[si] += 1
            w, h = shard.MIXED_SIZES[si]
            for i in sorted({0, m - 1}):
                got = mb.download(si, s, i)
                key = (si, i % len(mb.pool[si]))
                if key not in want_cache:
                    want_cache[key] = oracle.process(mb.pool[si][key[1]], resize=RESIZE, thumb=THUMB, glyphs=mb.glyphs[si], col=DEFAULT_COL)
                for k in ("resize", "thumbnail", "watermark"):
                    np.testing.assert_array_equal(got[k], want_cache[key][k], err_msg="%dx%d chunk frame %d %s" % (w, h, i, k))

        frames, alg = mb.run(items, q, on_done)
        assert frames == len(draw)
        assert alg == sum(mb.algorithmic_bytes(si) for si in draw)
        assert all(v >= 1 for v in checked.values()), checked
        # 2560x1440 -> 1024x576 and the product-default keep_aspect geometry of every size
        for si, (w, h) in enumerate(shard.MIXED_SIZES):
            info = mb.plans[si].info
            assert (info.resize_w, info.resize_h) == oracle.resize_dims(w, h, 1024, 768, True)
        mb.close()


def test_mixed_queue_is_drained_once_by_two_pullers(ipx):
    """Two MixedBatch pullers on one GPU share one queue (what two ranks do through the store counter): every chunk runs exactly once."""
    import threading
    from imageprocessor_amd import shard
    sizes = shard.MIXED_SIZES[:4]
    rng = np.random.default_rng(7)
    draw = [int(v) for v in rng.integers(0, len(sizes), 60)]

    class SharedQueue(shard.WorkQueue):
        lock = threading.Lock()

        def claim(self):
            with self.lock:
                return super().claim()

    with ipx.Context(device=0) as ctx:
        pullers = [shard.MixedBatch(ctx, _make_frames, text_glyphs, DEFAULT_COL, sizes=sizes, resize=RESIZE, thumbnail=THUMB,
                                    chunk_bytes=32 << 20, max_chunk=4) for _ in range(2)]
        items = pullers[0].items_for(draw)
        q = SharedQueue(len(items), chunk=1)
        done, errs, ran = [0, 0], [], [set(), set()]

        def work(r):
            try:
                done[r], _ = pullers[r].run(items, q, lambda si, m, s: ran[r].add((si, s)))
            except Exception as e:  # noqa: BLE001
                errs.append(repr(e))
        ts = [threading.Thread(target=work, args=(r,)) for r in range(2)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errs, errs
        assert sum(done) == len(draw) and min(done) > 0
        # whatever chunk ran last on a (size, stream) of either puller is still the oracle's
        for r, p in enumerate(pullers):
            for si, s_ in sorted(ran[r])[:3]:
                got = p.download(si, s_, 0)
                want = oracle.process(p.pool[si][0], resize=RESIZE, thumb=THUMB, glyphs=p.glyphs[si], col=DEFAULT_COL)
                for k in ("resize", "thumbnail", "watermark"):
                    np.testing.assert_array_equal(got[k], want[k])
            p.close()


def test_frame_beyond_the_addressable_span_is_refused(ipx):
    """include/ipx.h, ipx_frame_supported: a frame of 2 GiB or more gets IPX_ERR_UNSUPPORTED from the plan (and from a run with an
    enormous row stride) -- the worker keeps its CPU path -- instead of wrapped 32-bit offsets."""
    with ipx.Context(device=0) as ctx:
        for w, h in ((23172, 23172), (32768, 16384), (65536, 64)):
            with pytest.raises(ipx.IpxError) as e:
                ctx.plan(w, h, resize=(1024, 768, True), thumbnail=(200, True), watermark=True)
            assert e.value.status == -4 and "span" in e.value.text
        with pytest.raises(ipx.IpxError) as e:     # an output beyond the span
            ctx.plan(640, 480, resize=(40000, 40000, False), thumbnail=None, watermark=None)
        assert e.value.status == -4
        plan = ctx.plan(1920, 1080, resize=(1024, 768, True), thumbnail=(200, True), watermark=True)
        buf = ctx.alloc(1 << 20)
        with pytest.raises(ipx.IpxError) as e:     # 1080 rows at a 2 MiB pitch: 2.2 GiB
            plan.run_dev(1, buf.ptr, None, None, None, sstride=2 << 20, src_frame_stride=0)
        assert e.value.status == -4
        src = np.zeros((4, 4, 4), np.uint8)
        with pytest.raises(ipx.IpxError) as e:
            ctx.scale_bilinear(src, 70000, 2)
        assert e.value.status == -4
        plan.close()
        buf.free()


def test_largest_accepted_frame_class_vs_oracle(ipx):
    """The other side of the guard: a 32764 x 16380 frame (2.147 GB, 720 KB short of the limit) runs through the fused kernel with the
    text box in its bottom-right corner (the highest offsets of the frame) and matches the oracle on every byte."""
    w, h = 32764, 16380
    assert ipx.lib().ipx_frame_supported(w, h, w * 4, 4) == 0
    rng = np.random.default_rng(99)
    # a 2 GB frame of cheap structured noise: a random 512 x 1024 tile repeated with a per-tile offset, opaque
    tile = rng.integers(0, 256, (512, 1024, 4), dtype=np.uint8)
    frame = np.empty((h, w, 4), np.uint8)
    for y0 in range(0, h, 512):
        for x0 in range(0, w, 1024):
            t = tile[:min(512, h - y0), :min(1024, w - x0)]
            frame[y0:y0 + t.shape[0], x0:x0 + t.shape[1]] = t + np.uint8((y0 // 512 * 7 + x0 // 1024 * 13) & 0xff)
    frame[..., 3] = 255
    glyphs = text_glyphs(w, h)
    with ipx.Context(device=0) as ctx:
        gs = ctx.glyphset(glyphs, DEFAULT_COL)
        plan = ctx.plan(w, h, resize=RESIZE, thumbnail=THUMB, watermark=gs)
        got = plan.run_host(frame[None])
        want = oracle.process(frame, resize=RESIZE, thumb=THUMB, glyphs=glyphs, col=DEFAULT_COL)
        for k in ("resize", "thumbnail", "watermark"):
            assert np.array_equal(got[k][0], want[k]), k
        plan.close()
        gs.close()
