"""Sharding of a batch of independent frames over one process per GPU.

The reference scales by running more Kafka consumers in one group (kafka/consumer.go:23,
docker-compose.yaml:86-96): messages are independent, nothing is exchanged.  The same holds here:
frames are independent units, so there is NO data-path collective.  torch.distributed is used only
for (a) a barrier and the max-over-ranks of the timed region and (b) an atomic counter in the
rendezvous store that implements pull scheduling ("work stealing") for mixed-size batches.
"""
import os
import time


def round_robin(n_items, rank, world):
    """Static partition for uniform batches: image_index mod nGPU."""
    return list(range(rank, n_items, world))


def lpt_order(costs):
    """Largest-first order (longest processing time first) for pull scheduling of mixed sizes."""
    return sorted(range(len(costs)), key=lambda i: (-costs[i], i))


def frame_cost(w, h):
    """Bytes moved for one frame by the full pipeline: 4WH read + 4WH watermark + outputs."""
    return 8 * w * h + 4 * 1024 * 768 + 4 * 200 * 200


class WorkQueue:
    """Pull scheduling over a shared counter.  Every rank walks the same largest-first order and
    claims the next unclaimed chunk with an atomic add in the torch.distributed store, so a rank that
    finishes early simply claims more: work stealing falls out of pull scheduling.  With no process
    group (single process) the counter is local."""

    def __init__(self, n_items, chunk=1, key="ipx_queue", store=None):
        self.n, self.chunk, self.key, self.store = n_items, max(1, chunk), key, store
        self._local = 0

    def claim(self):
        """-> range of item positions (in the shared order), or None when the queue is empty."""
        if self.store is not None:
            hi = self.store.add(self.key, self.chunk)   # atomic fetch-add across ranks
            lo = hi - self.chunk
        else:
            lo = self._local
            self._local += self.chunk
        if lo >= self.n:
            return None
        return range(lo, min(self.n, lo + self.chunk))


def default_store():
    """The store of the default process group (TCPStore under torchrun), or None."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.distributed_c10d._get_default_store()
    except Exception:
        pass
    return None


def aggregate(units, seconds):
    """Whole-job numbers: units summed over ranks, seconds = max over ranks."""
    try:
        import torch
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            u = torch.tensor([float(units)], dtype=torch.float64)
            t = torch.tensor([float(seconds)], dtype=torch.float64)
            dist.all_reduce(u, op=dist.ReduceOp.SUM)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(u.item()), float(t.item())
    except ImportError:
        pass
    return float(units), float(seconds)


def init_from_env():
    """One process per GPU under torch.distributed.run; gloo carries barrier / reduce / store only."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("gloo", rank=rank, world_size=world)
    return rank, local_rank, world


# ---- BASELINE config 5: mixed-size batch (480p - 8K), full pipeline, pull scheduling ------------------------------------------

MIXED_SIZES = [(854, 480), (1280, 720), (1920, 1080), (2560, 1440), (3840, 2160), (7680, 4320)]


class MixedBatch:
    """Frames of several sizes through resize + thumbnail + watermark on one GPU of the job (worker.go:112-149: every message is
    independent, so a rank simply pulls the next chunk when it has room).

    Per size: one plan, a pool of `chunk` source frames resident in HBM, and one set of output buffers per stream.  A work item is a
    chunk of equal-size frames (about `chunk_bytes` of source, at most `max_chunk` frames).  run() walks a shared WorkQueue over the
    largest-first order with two streams: chunk k goes to stream k % 2 as soon as chunk k - 2 (the previous one on that stream) is done,
    so the next claim and launch happen WHILE the previous chunk runs and the GPU always has a launch queued behind the running one."""

    def __init__(self, ctx, make_frames, make_glyphs, col, sizes=MIXED_SIZES, resize=(1024, 768, True), thumbnail=(200, True),
                 chunk_bytes=256 << 20, max_chunk=64, nstreams=2, distinct=1):
        self.ctx, self.sizes, self.resize, self.thumbnail, self.col = ctx, list(sizes), resize, thumbnail, col
        self.streams = [ctx.stream() for _ in range(max(1, nstreams))]
        self.plans, self.glyphs, self.gsets, self.src, self.outs, self.chunk_of, self.pool = [], [], [], [], [], [], []
        for si, (w, h) in enumerate(self.sizes):
            fb = w * h * 4
            n = max(1, min(max_chunk, chunk_bytes // fb))
            gl = make_glyphs(w, h)
            gs = ctx.glyphset(gl, col)
            pl = ctx.plan(w, h, resize=resize, thumbnail=thumbnail, watermark=gs)
            src = ctx.alloc(n * fb)
            pool = make_frames(si, w, h, min(n, max(1, distinct)))          # k x h x w x 4, seeded
            for i in range(n):
                src.upload(pool[i % len(pool)], offset=i * fb)
            i_ = pl.info
            self.outs.append([(ctx.alloc(n * i_.resize_bytes), ctx.alloc(n * i_.thumb_bytes), ctx.alloc(n * i_.wm_bytes))
                              for _ in self.streams])
            self.plans.append(pl); self.glyphs.append(gl); self.gsets.append(gs); self.src.append(src)
            self.chunk_of.append(n); self.pool.append(pool)

    def items_for(self, draw):
        """draw: a size index per frame of the batch -> [(size index, frame count)] in largest-first order"""
        items = []
        for si in range(len(self.sizes)):
            cnt = int(sum(1 for d in draw if d == si))
            while cnt > 0:
                m = min(cnt, self.chunk_of[si])
                items.append((si, m))
                cnt -= m
        order = lpt_order([frame_cost(*self.sizes[si]) * m for si, m in items])
        return [items[i] for i in order]

    def algorithmic_bytes(self, si):
        return self.plans[si].info.algorithmic_bytes

    def launch(self, si, m, k):
        """chunk (si, m) on stream k % nstreams; asynchronous"""
        s = k % len(self.streams)
        res, th, wm = self.outs[si][s]
        self.plans[si].run_dev(m, self.src[si].ptr, res.ptr, th.ptr, wm.ptr, stream=self.streams[s])

    def run(self, items, queue, on_done=None):
        """Pull chunks from `queue` (positions into `items`) until it is empty.  -> (frames, algorithmic bytes) this rank processed.
        on_done(si, m, stream index) is called once a chunk is known to be complete (tests download and compare there)."""
        inflight = [None] * len(self.streams)
        k = frames = alg = 0
        while True:
            c = queue.claim()
            if c is None:
                break
            for pos in c:
                si, m = items[pos]
                s = k % len(self.streams)
                if inflight[s] is not None:                 # room on this stream only once its previous chunk is done
                    self.ctx.sync(self.streams[s])
                    if on_done:
                        on_done(*inflight[s], s)
                self.launch(si, m, k)
                inflight[s] = (si, m)
                frames += m
                alg += m * self.algorithmic_bytes(si)
                k += 1
        for s, it in enumerate(inflight):
            if it is not None:
                self.ctx.sync(self.streams[s])
                if on_done:
                    on_done(*it, s)
        return frames, alg

    def download(self, si, s, i):
        """outputs of frame i of the last chunk of size si that ran on stream s"""
        info = self.plans[si].info
        w, h = self.sizes[si]
        res, th, wm = self.outs[si][s]
        return {"resize": res.download((info.resize_h, info.resize_w, 4), offset=i * info.resize_bytes),
                "thumbnail": th.download((info.thumb_h, info.thumb_w, 4), offset=i * info.thumb_bytes),
                "watermark": wm.download((h, w, 4), offset=i * info.wm_bytes)}

    def close(self):
        for st in self.streams:
            self.ctx.stream_destroy(st)
        self.streams = []
        for pl in self.plans:
            pl.close()
        for gs in self.gsets:
            gs.close()
        for b in self.src + [x for o in self.outs for t in o for x in t]:
            b.free()
