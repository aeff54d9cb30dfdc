"""The micro-batcher below the ABI (ipx_batcher_*) on the GPU: single files from many threads, grouped by frame size and operator content,
every file's three streams equal what ipx_plan_run_jpeg_jpeg returns for the same file.  Reference: internal/worker/worker.go:88-96,
112-149 (one message per goroutine), 165-234 (a message is done when ITS objects are saved)."""
import io
import threading

import numpy as np
import pytest

from helpers import DEFAULT_COL, rgba_frames, text_glyphs

pytestmark = pytest.mark.gpu
PIL = pytest.importorskip("PIL.Image")


def _jpeg(rgb, **kw):
    buf = io.BytesIO()
    PIL.fromarray(rgb).save(buf, "JPEG", **kw)
    return buf.getvalue()


def test_single_files_from_many_threads_equal_the_batch_entry():
    import imageprocessor_amd as ipx
    sizes = [(320, 200), (256, 144), (128, 96)]
    rng = np.random.default_rng(11)
    files = []
    for i in range(500):
        w, h = sizes[int(rng.integers(0, 3))]
        rgb = rgba_frames(1, w, h, seed=1000 + i)[0][..., :3]
        kw = {"quality": int(rng.integers(60, 95))}
        if i % 17 == 0:
            kw["progressive"] = True                       # its scans are walked on the host, inside the same batch
        files.append((w, h, _jpeg(rgb, **kw)))
    files[7] = (320, 200, files[7][2][:200])               # a truncated upload: its own status, the neighbours unharmed
    files[8] = (320, 200, _jpeg(rgba_frames(1, 320, 200, seed=5)[0][..., 0]))   # a Gray file among colour files: not in this batch's shape
    ops = {s: dict(resize=(64, 48, False), thumbnail=(32, True), glyphs=text_glyphs(s[0], s[1], n=4, width_px=60, height_px=16), col=DEFAULT_COL) for s in sizes}
    # what the batch entry gives for each file (one call per size and JPEG shape -- the batcher groups by both; statuses per file)
    want = {}
    with ipx.Context(device=0) as ctx:
        for s, gray in [(s, g) for s in sizes for g in (False, True)]:
            idx = [i for i, f in enumerate(files) if f[:2] == s and (i == 8) == gray]
            if not idx:
                continue
            gs = ctx.glyphset(ops[s]["glyphs"], DEFAULT_COL)
            plan = ctx.plan(s[0], s[1], resize=ops[s]["resize"], thumbnail=ops[s]["thumbnail"], watermark=gs)
            out, st = plan.run_jpeg_jpeg([files[i][2] for i in idx], 85)
            for j, i in enumerate(idx):
                want[i] = (st[j], {k: out[k][j] for k in ("resize", "thumbnail", "watermark")})
            plan.close(); gs.close()
    got, errs = {}, []
    with ipx.Pool(devices=(0,)) as pool, ipx.Batcher(pool, max_batch=48, max_wait_us=3000, quality=85) as b:
        order = rng.permutation(len(files))

        def work(part):
            try:
                tickets = []
                for i in part:
                    w, h, data = files[i]
                    tickets.append((i, b.submit(data, w, h, **ops[(w, h)])))
                    if len(tickets) >= 6:                  # a goroutine of the reference holds one message; a few in flight here
                        i0, t0 = tickets.pop(0)
                        got[i0] = b.wait(t0)
                for i0, t0 in tickets:
                    got[i0] = b.wait(t0)
            except Exception as e:  # noqa: BLE001
                errs.append(repr(e)[:300])
        ts = [threading.Thread(target=work, args=(order[k::8],)) for k in range(8)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        st = b.stats()
    assert not errs, errs
    assert st["files"] == 500 and st["largest_batch"] <= 48 and st["batches"] >= 500 // 48
    assert len(got) == 500
    for i in range(500):
        assert got[i][0] == want[i][0], (i, got[i][0], want[i][0])
        if want[i][0] == 0:
            assert got[i][1] == want[i][1], "file %d: streams differ from ipx_plan_run_jpeg_jpeg's" % i
        else:
            assert got[i][1] == {"resize": None, "thumbnail": None, "watermark": None}
    # the truncated file has its own status; the Gray file, in a batch of its own shape whenever it arrives, is processed like any other
    assert want[7][0] != 0 and want[8][0] == 0 and sum(1 for i in want if want[i][0] == 0) >= 490
