"""imageprocessor_amd -- MI355X-native pixel worker for ImageProcessor's resize / thumbnail /
watermark path.  This module is a thin numpy/ctypes front end of the C ABI in include/ipx.h;
all pixel work runs in the hand-written HIP kernels of csrc/ (there is no CPU fallback)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Config, Glyph, PlanInfo, PlanParams, Rect

OP_OVER = 0
DEEP_NRGBA64, DEEP_RGBA64, DEEP_GRAY16, DEEP_CMYK = 0, 1, 2, 3   # ipx.h IPX_DEEP_*
OP_SRC = 1


class IpxError(RuntimeError):
    def __init__(self, status, text):
        super().__init__("ipx status %d: %s" % (status, text))
        self.status = status
        self.text = text


def lib():
    return _lib.load()


def _check(rc):
    if rc < 0:
        raise IpxError(rc, lib().ipx_last_error().decode(errors="replace"))
    return rc


def _rect(r):
    return r if isinstance(r, Rect) else Rect(*[int(v) for v in r])


def _frame(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 4:
        raise ValueError("RGBA8 frame must be H x W x 4")
    return a


# ---- host-only rules (no GPU needed) ------------------------------------------------------------

def resize_dims(ow, oh, w, h, keep_aspect):
    nw, nh = C.c_int(), C.c_int()
    _check(lib().ipx_resize_dims(ow, oh, w, h, int(bool(keep_aspect)), C.byref(nw), C.byref(nh)))
    return nw.value, nh.value


def thumb_geometry(ow, oh, size, crop_to_fit):
    r, nw, nh = Rect(), C.c_int(), C.c_int()
    _check(lib().ipx_thumb_geometry(ow, oh, size, int(bool(crop_to_fit)), C.byref(r), C.byref(nw),
                                    C.byref(nh)))
    return (r.x0, r.y0, r.x1, r.y1), nw.value, nh.value


def text_height_px(font_size):
    return lib().ipx_text_height_px(float(font_size))


def watermark_anchor(position, w, h, width_px, height_px):
    px, py = C.c_int(), C.c_int()
    _check(lib().ipx_watermark_anchor(position.encode(), w, h, width_px, height_px, C.byref(px),
                                      C.byref(py)))
    return px.value, py.value


def parse_color(s, opacity):
    out = (C.c_uint8 * 4)()
    rc = _check(lib().ipx_parse_color(s.encode(), float(opacity), out))
    return tuple(out), rc == 1


def device_count():
    return lib().ipx_device_count()


def _glyph_array(glyphs):
    keep = []
    arr = (Glyph * max(1, len(glyphs)))()
    for i, g in enumerate(glyphs):
        m = np.ascontiguousarray(g["mask"], dtype=np.uint8)
        if m.ndim != 2:
            raise ValueError("glyph mask must be mh x mw")
        keep.append(m)
        mp = g.get("mp", (0, 0))
        arr[i] = Glyph(m.ctypes.data, m.shape[1], m.shape[0], m.shape[1], _rect(g["dr"]), int(mp[0]),
                       int(mp[1]))
    return arr, keep


# ---- GPU objects -------------------------------------------------------------------------------------

class DevBuffer:
    """hipMalloc'd bytes owned by a Context."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        self.ptr = lib().ipx_dev_alloc(ctx.handle, max(1, self.nbytes))
        if not self.ptr:
            raise IpxError(-2, lib().ipx_last_error().decode())

    def upload(self, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        _check(lib().ipx_memcpy_h2d(self.ctx.handle, self.ptr + offset, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, shape, dtype=np.uint8, offset=0):
        out = np.empty(shape, dtype)
        assert offset + out.nbytes <= self.nbytes
        _check(lib().ipx_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr + offset, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().ipx_dev_free(self.ctx.handle, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class GlyphSet:
    """Rasterised watermark text (A8 masks + DrawMask rectangles) resident in HBM."""

    def __init__(self, ctx, glyphs, col):
        self.ctx = ctx
        arr, keep = _glyph_array(list(glyphs))
        c = (C.c_uint8 * 4)(*[int(v) for v in col])
        h = C.c_void_p()
        _check(lib().ipx_glyphset_create(ctx.handle, arr, len(glyphs), c, C.byref(h)))
        self.handle = h.value

    def close(self):
        if self.handle:
            lib().ipx_glyphset_destroy(self.ctx.handle, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Plan:
    """Fused resize + thumbnail + watermark for frames of one size (ipx_plan_*)."""

    def __init__(self, ctx, sw, sh, resize=(1024, 768, True), thumbnail=(200, True), watermark=None):
        """resize=(w, h, keep_aspect) | None; thumbnail=(size, crop_to_fit) | None;
        watermark=None (operator absent) | False/GlyphSet-less copy (True) | GlyphSet."""
        self.ctx = ctx
        self._sw, self._sh = sw, sh
        p = PlanParams()
        p.sw, p.sh = sw, sh
        if resize:
            p.do_resize, p.resize_w, p.resize_h, p.keep_aspect = 1, resize[0], resize[1], int(bool(resize[2]))
        if thumbnail:
            p.do_thumbnail, p.thumb_size, p.crop_to_fit = 1, thumbnail[0], int(bool(thumbnail[1]))
        self._gs = None
        if watermark is not None and watermark is not False:
            p.do_watermark = 1
            if isinstance(watermark, GlyphSet):
                self._gs = watermark
                p.glyphs = watermark.handle
        h = C.c_void_p()
        _check(lib().ipx_plan_create(ctx.handle, C.byref(p), C.byref(h)))
        self.handle = h.value
        self.info = PlanInfo()
        _check(lib().ipx_plan_query(self.handle, C.byref(self.info)))

    def run_dev(self, n, src_ptr, resize_ptr=None, thumb_ptr=None, wm_ptr=None, stream=None,
                sstride=None, src_frame_stride=None, resize_frame_stride=None,
                thumb_frame_stride=None, wm_frame_stride=None):
        i = self.info
        _check(lib().ipx_plan_run_dev(
            self.ctx.handle, stream, self.handle, n, src_ptr,
            sstride if sstride is not None else self._sw * 4,
            src_frame_stride if src_frame_stride is not None else self._sw * self._sh * 4,
            resize_ptr, resize_frame_stride if resize_frame_stride is not None else i.resize_bytes,
            thumb_ptr, thumb_frame_stride if thumb_frame_stride is not None else i.thumb_bytes,
            wm_ptr, wm_frame_stride if wm_frame_stride is not None else i.wm_bytes))

    def run_host(self, frames, want=("resize", "thumbnail", "watermark"), out=None):
        """frames: n x H x W x 4 uint8 (host).  Returns dict of output batches (`out` may supply
        preallocated, e.g. pinned, arrays)."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        n = frames.shape[0]
        i = self.info
        out = dict(out) if out else {}
        if "resize" in want and i.resize_bytes and "resize" not in out:
            out["resize"] = np.empty((n, i.resize_h, i.resize_w, 4), np.uint8)
        if "thumbnail" in want and i.thumb_bytes and "thumbnail" not in out:
            out["thumbnail"] = np.empty((n, i.thumb_h, i.thumb_w, 4), np.uint8)
        if "watermark" in want and i.wm_bytes and "watermark" not in out:
            out["watermark"] = np.empty((n, i.wm_h, i.wm_w, 4), np.uint8)

        def p(k):
            return out[k].ctypes.data if k in out else None
        _check(lib().ipx_plan_run_host(self.ctx.handle, self.handle, n, frames.ctypes.data,
                                       self._sw * 4, self._sw * self._sh * 4,
                                       p("resize"), i.resize_bytes, p("thumbnail"), i.thumb_bytes,
                                       p("watermark"), i.wm_bytes))
        return out

    def _host_outs(self, n, want):
        i = self.info
        out = {}
        if "resize" in want and i.resize_bytes:
            out["resize"] = np.empty((n, i.resize_h, i.resize_w, 4), np.uint8)
        if "thumbnail" in want and i.thumb_bytes:
            out["thumbnail"] = np.empty((n, i.thumb_h, i.thumb_w, 4), np.uint8)
        if "watermark" in want and i.wm_bytes:
            out["watermark"] = np.empty((n, i.wm_h, i.wm_w, 4), np.uint8)
        return out, lambda k: out[k].ctypes.data if k in out else None

    def run_host_nrgba(self, frames, want=("resize", "thumbnail", "watermark")):
        """frames: n x H x W x 4 uint8, non-premultiplied (*image.NRGBA, host) -> dict of output batches"""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        n, i = frames.shape[0], self.info
        out, p = self._host_outs(n, want)
        _check(lib().ipx_plan_run_host_nrgba(self.ctx.handle, self.handle, n, frames.ctypes.data, self._sw * 4, self._sw * self._sh * 4,
                                             p("resize"), i.resize_bytes, p("thumbnail"), i.thumb_bytes, p("watermark"), i.wm_bytes))
        return out

    def run_host_deep(self, pix, kind, want=("resize", "thumbnail", "watermark")):
        """pix: n x H x (W * bpp) uint8, Go's Pix rows of *image.NRGBA64 / RGBA64 / Gray16 / CMYK frames (kind: DEEP_*, host)"""
        pix = np.ascontiguousarray(pix, dtype=np.uint8)
        n, i = pix.shape[0], self.info
        out, p = self._host_outs(n, want)
        _check(lib().ipx_plan_run_host_deep(self.ctx.handle, self.handle, n, kind, pix.ctypes.data, pix.shape[2], pix.shape[1] * pix.shape[2],
                                            p("resize"), i.resize_bytes, p("thumbnail"), i.thumb_bytes, p("watermark"), i.wm_bytes))
        return out

    def run_host_gray(self, frames, want=("resize", "thumbnail", "watermark")):
        """frames: n x H x W uint8 (*image.Gray, host) -> dict of output batches"""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        n, i = frames.shape[0], self.info
        out, p = self._host_outs(n, want)
        _check(lib().ipx_plan_run_host_gray(self.ctx.handle, self.handle, n, frames.ctypes.data, self._sw, self._sw * self._sh,
                                            p("resize"), i.resize_bytes, p("thumbnail"), i.thumb_bytes, p("watermark"), i.wm_bytes))
        return out

    def run_host_paletted(self, index, palettes, want=("resize", "thumbnail", "watermark")):
        """index: n x H x W uint8, palettes: n x 256 x 4 uint8 (R, G, B, A) non-premultiplied (*image.Paletted, host)"""
        index = np.ascontiguousarray(index, dtype=np.uint8)
        palettes = np.ascontiguousarray(palettes, dtype=np.uint8)
        n, i = index.shape[0], self.info
        assert palettes.shape == (n, 256, 4)
        out, p = self._host_outs(n, want)
        _check(lib().ipx_plan_run_host_paletted(self.ctx.handle, self.handle, n, index.ctypes.data, self._sw, self._sw * self._sh,
                                                palettes.ctypes.data, p("resize"), i.resize_bytes, p("thumbnail"), i.thumb_bytes,
                                                p("watermark"), i.wm_bytes))
        return out

    def run_host_jpeg(self, frames, quality=85, want=("resize", "thumbnail", "watermark"), copy=True):
        """frames: n x H x W x 4 uint8 (host, ideally pinned) -> {operator: [jpeg bytes] * n}: operators and jpeg.Encode on
        the GPU, only the streams come back.  copy=False: lengths only (the streams are released unread; for timing)."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        n = frames.shape[0]
        i = self.info
        arrs = {}
        for k, present in (("resize", i.resize_bytes), ("thumbnail", i.thumb_bytes), ("watermark", i.wm_bytes)):
            if k in want and present:
                arrs[k] = (_lib.Bytes * n)()
        res = C.c_void_p()
        _check(lib().ipx_plan_run_host_jpeg(self.ctx.handle, self.handle, n, frames.ctypes.data, self._sw * 4, self._sw * self._sh * 4,
                                            int(quality), arrs.get("resize"), arrs.get("thumbnail"), arrs.get("watermark"), C.byref(res)))
        out = {}
        for k, a in arrs.items():
            out[k] = [C.string_at(a[j].data, a[j].len) if copy else a[j].len for j in range(n)]
        lib().ipx_jpeg_result_free(self.ctx.handle, res)
        return out

    def run_host_ycbcr_jpeg(self, y, cb, cr, ratio, quality=85, want=("resize", "thumbnail", "watermark"), copy=True):
        """Decoded JPEG planes (host) -> {operator: [jpeg bytes] * n}; see run_host_jpeg."""
        y, cb, cr = (np.ascontiguousarray(a, dtype=np.uint8) for a in (y, cb, cr))
        n, h, w = y.shape
        assert (w, h) == (self._sw, self._sh) and cb.shape == cr.shape and cb.shape[0] == n
        b = _lib.YCbCrBatch(y.ctypes.data, cb.ctypes.data, cr.ctypes.data, w, cb.shape[2], h * w, cb.shape[1] * cb.shape[2], int(ratio))
        i = self.info
        arrs = {}
        for k, present in (("resize", i.resize_bytes), ("thumbnail", i.thumb_bytes), ("watermark", i.wm_bytes)):
            if k in want and present:
                arrs[k] = (_lib.Bytes * n)()
        res = C.c_void_p()
        _check(lib().ipx_plan_run_host_ycbcr_jpeg(self.ctx.handle, self.handle, n, C.byref(b), int(quality), arrs.get("resize"),
                                                  arrs.get("thumbnail"), arrs.get("watermark"), C.byref(res)))
        out = {k: [C.string_at(a[j].data, a[j].len) if copy else a[j].len for j in range(n)] for k, a in arrs.items()}
        lib().ipx_jpeg_result_free(self.ctx.handle, res)
        return out

    def run_jpeg_jpeg(self, files, quality=85, want=("resize", "thumbnail", "watermark"), copy=True):
        """JPEG byte strings in -> ({operator: [jpeg bytes | None] * n}, status list): decode, operators, encode on the GPU."""
        n = len(files)
        keep = [bytes(f) for f in files]
        arr = (_lib.Bytes * n)()
        for j, f in enumerate(keep):
            arr[j].data = C.cast(C.c_char_p(f), C.c_void_p)
            arr[j].len = len(f)
        i = self.info
        outs = {}
        for k, present in (("resize", i.resize_bytes), ("thumbnail", i.thumb_bytes), ("watermark", i.wm_bytes)):
            if k in want and present:
                outs[k] = (_lib.Bytes * n)()
        status = (C.c_int * n)()
        res = C.c_void_p()
        _check(lib().ipx_plan_run_jpeg_jpeg(self.ctx.handle, self.handle, n, arr, int(quality), outs.get("resize"), outs.get("thumbnail"),
                                            outs.get("watermark"), status, C.byref(res)))
        out = {k: [(C.string_at(a[j].data, a[j].len) if copy else a[j].len) if a[j].data else None for j in range(n)] for k, a in outs.items()}
        if res:
            lib().ipx_jpeg_result_free(self.ctx.handle, res)
        return out, list(status)

    def run_dev_nrgba(self, n, src_ptr, resize_ptr=None, thumb_ptr=None, wm_ptr=None, stream=None):
        """*image.NRGBA frames (tightly packed) resident in HBM (ipx_plan_run_dev_nrgba)"""
        i = self.info
        _check(lib().ipx_plan_run_dev_nrgba(self.ctx.handle, stream, self.handle, n, src_ptr, self._sw * 4, self._sw * self._sh * 4,
                                            resize_ptr, i.resize_bytes, thumb_ptr, i.thumb_bytes, wm_ptr, i.wm_bytes))

    def run_dev_deep(self, n, kind, src_ptr, stride, frame_stride, resize_ptr=None, thumb_ptr=None, wm_ptr=None, stream=None):
        """*image.NRGBA64 / RGBA64 / Gray16 / CMYK frames (Go's Pix) resident in HBM (ipx_plan_run_dev_deep)"""
        i = self.info
        _check(lib().ipx_plan_run_dev_deep(self.ctx.handle, stream, self.handle, n, kind, src_ptr, stride, frame_stride, resize_ptr,
                                           i.resize_bytes, thumb_ptr, i.thumb_bytes, wm_ptr, i.wm_bytes))

    def run_dev_gray(self, n, gray_ptr, stride, frame_stride, resize_ptr=None, thumb_ptr=None, wm_ptr=None, stream=None):
        """*image.Gray frames resident in HBM (ipx_plan_run_dev_gray)"""
        i = self.info
        _check(lib().ipx_plan_run_dev_gray(self.ctx.handle, stream, self.handle, n, gray_ptr, stride, frame_stride, resize_ptr,
                                           i.resize_bytes, thumb_ptr, i.thumb_bytes, wm_ptr, i.wm_bytes))

    def run_dev_paletted(self, n, index_ptr, stride, frame_stride, palettes_ptr, resize_ptr=None, thumb_ptr=None, wm_ptr=None, stream=None):
        """*image.Paletted frames resident in HBM: index bytes plus 256 x (R, G, B, A) per frame (ipx_plan_run_dev_paletted)"""
        i = self.info
        _check(lib().ipx_plan_run_dev_paletted(self.ctx.handle, stream, self.handle, n, index_ptr, stride, frame_stride, palettes_ptr,
                                               resize_ptr, i.resize_bytes, thumb_ptr, i.thumb_bytes, wm_ptr, i.wm_bytes))

    def run_dev_ycbcr(self, n, y_ptr, cb_ptr, cr_ptr, ratio, ystride, cstride, y_frame_stride, c_frame_stride,
                      resize_ptr=None, thumb_ptr=None, wm_ptr=None, stream=None):
        i = self.info
        b = _lib.YCbCrBatch(y_ptr, cb_ptr, cr_ptr, ystride, cstride, y_frame_stride, c_frame_stride, int(ratio))
        _check(lib().ipx_plan_run_dev_ycbcr(self.ctx.handle, stream, self.handle, n, C.byref(b), resize_ptr,
                                            i.resize_bytes, thumb_ptr, i.thumb_bytes, wm_ptr, i.wm_bytes))

    def run_host_ycbcr(self, y, cb, cr, ratio, want=("resize", "thumbnail", "watermark")):
        """A batch of decoded JPEG frames: y n x H x W, cb / cr n x CH x CW uint8 (image.YCbCr planes)."""
        y, cb, cr = (np.ascontiguousarray(a, dtype=np.uint8) for a in (y, cb, cr))
        n, h, w = y.shape
        assert (w, h) == (self._sw, self._sh) and cb.shape == cr.shape and cb.shape[0] == n
        b = _lib.YCbCrBatch(y.ctypes.data, cb.ctypes.data, cr.ctypes.data, w, cb.shape[2], h * w,
                            cb.shape[1] * cb.shape[2], int(ratio))
        i = self.info
        out = {}
        if "resize" in want and i.resize_bytes:
            out["resize"] = np.empty((n, i.resize_h, i.resize_w, 4), np.uint8)
        if "thumbnail" in want and i.thumb_bytes:
            out["thumbnail"] = np.empty((n, i.thumb_h, i.thumb_w, 4), np.uint8)
        if "watermark" in want and i.wm_bytes:
            out["watermark"] = np.empty((n, i.wm_h, i.wm_w, 4), np.uint8)

        def p(k):
            return out[k].ctypes.data if k in out else None
        _check(lib().ipx_plan_run_host_ycbcr(self.ctx.handle, self.handle, n, C.byref(b), p("resize"), i.resize_bytes,
                                             p("thumbnail"), i.thumb_bytes, p("watermark"), i.wm_bytes))
        return out

    def close(self):
        if self.handle:
            lib().ipx_plan_destroy(self.ctx.handle, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One per process and GPU (ipx_create)."""

    def __init__(self, device=-1, lanes=0, lane_bytes=0):
        cfg = Config(device, lanes, lane_bytes)
        h = C.c_void_p()
        _check(lib().ipx_create(C.byref(cfg), C.byref(h)))
        self.handle = h.value

    def close(self):
        if self.handle:
            lib().ipx_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def alloc(self, nbytes):
        return DevBuffer(self, nbytes)

    def host_alloc(self, shape, dtype=np.uint8):
        """Pinned (hipHostMalloc) staging as a numpy array; free with host_free(arr)."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = lib().ipx_host_alloc(self.handle, max(1, n))
        if not p:
            raise IpxError(-2, lib().ipx_last_error().decode())
        arr = np.frombuffer((C.c_uint8 * n).from_address(p), dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p
        return arr

    def host_free(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p:
            lib().ipx_host_free(self.handle, p)

    def stream(self):
        """A stream of the caller's own for the asynchronous device-pointer entries (free with stream_destroy)."""
        st = lib().ipx_stream_create(self.handle)
        if not st:
            raise IpxError(-3, lib().ipx_last_error().decode())
        return st

    def stream_destroy(self, st):
        _check(lib().ipx_stream_destroy(self.handle, st))

    def sync(self, stream=None):
        _check(lib().ipx_stream_sync(self.handle, stream))

    def device_sync(self):
        _check(lib().ipx_device_sync(self.handle))

    def copy_d2d(self, dst_ptr, src_ptr, nbytes):
        _check(lib().ipx_memcpy_d2d(self.handle, dst_ptr, src_ptr, nbytes))

    def link_probe(self, up_bytes, down_bytes, reps=3):
        """-> {"up", "down", "up_while_down", "down_while_up"} in GB/s: what pinned copies get from the host link on this box"""
        out = (C.c_double * 4)()
        _check(lib().ipx_link_probe(self.handle, int(up_bytes), int(down_bytes), int(reps), C.byref(out)))
        return {"up": round(out[0], 2), "down": round(out[1], 2), "up_while_down": round(out[2], 2), "down_while_up": round(out[3], 2)}

    def stream_copy(self, dst_ptr, src_ptr, nbytes, stream=None):
        """A plain streaming copy kernel: what this box's HBM gives a copy (bench.py's copy_ceiling)."""
        _check(lib().ipx_stream_copy(self.handle, stream, dst_ptr, src_ptr, nbytes))

    def timed(self, fn, stream=None):
        """Runs fn() bracketed by HIP events on `stream`; returns milliseconds (syncs)."""
        L = lib()
        e0, e1 = L.ipx_event_create(self.handle), L.ipx_event_create(self.handle)
        try:
            _check(L.ipx_event_record(self.handle, e0, stream))
            fn()
            _check(L.ipx_event_record(self.handle, e1, stream))
            ms = C.c_float()
            _check(L.ipx_event_elapsed_ms(self.handle, e0, e1, C.byref(ms)))
            return ms.value
        finally:
            L.ipx_event_destroy(self.handle, e0)
            L.ipx_event_destroy(self.handle, e1)

    def glyphset(self, glyphs, col):
        return GlyphSet(self, glyphs, col)

    def plan(self, sw, sh, **kw):
        return Plan(self, sw, sh, **kw)

    # ---- per-operation seam on host arrays (synchronous) ----------------------------------------
    def scale_bilinear(self, src, dw, dh, sr=None, dr=None, op=OP_OVER, dst=None):
        """xdraw.BiLinear.Scale(dst, dr, src, sr, op, nil); dst defaults to a zeroed frame."""
        src = _frame(src)
        sh, sw = src.shape[:2]
        if dst is None:
            dst = np.zeros((dh, dw, 4), np.uint8)
        assert dst.dtype == np.uint8 and dst.flags.c_contiguous and dst.shape == (dh, dw, 4)
        _check(lib().ipx_scale_bilinear_rgba8(
            self.handle, dst.ctypes.data, dw, dh, dw * 4, _rect(dr if dr is not None else (0, 0, dw, dh)),
            src.ctypes.data, sw, sh, sw * 4, _rect(sr if sr is not None else (0, 0, sw, sh)), op))
        return dst

    def draw(self, dst, r, src, sp=(0, 0), op=OP_SRC):
        src = _frame(src)
        assert dst.dtype == np.uint8 and dst.flags.c_contiguous
        dh, dw = dst.shape[:2]
        sh, sw = src.shape[:2]
        _check(lib().ipx_draw_rgba8(self.handle, dst.ctypes.data, dw, dh, dw * 4, _rect(r),
                                    src.ctypes.data, sw, sh, sw * 4, int(sp[0]), int(sp[1]), op))
        return dst

    # ---- the deep source types: Go's Pix rows (H x W*bpp uint8) of *image.NRGBA64 / RGBA64 / Gray16 / CMYK frames ----------
    def scale_bilinear_deep(self, pix, kind, dw, dh, sr=None, dr=None, op=OP_OVER, dst=None):
        pix = np.ascontiguousarray(pix, np.uint8)
        sh, row = pix.shape
        sw = row // {DEEP_GRAY16: 2, DEEP_CMYK: 4}.get(kind, 8)
        if dst is None:
            dst = np.zeros((dh, dw, 4), np.uint8)
        assert dst.dtype == np.uint8 and dst.flags.c_contiguous and dst.shape == (dh, dw, 4)
        _check(lib().ipx_scale_bilinear_deep(self.handle, dst.ctypes.data, dw, dh, dw * 4, _rect(dr if dr is not None else (0, 0, dw, dh)),
                                             pix.ctypes.data, sw, sh, row, kind, _rect(sr if sr is not None else (0, 0, sw, sh)), op))
        return dst

    def draw_deep(self, dst, r, pix, kind, sp=(0, 0), op=OP_SRC):
        pix = np.ascontiguousarray(pix, np.uint8)
        assert dst.dtype == np.uint8 and dst.flags.c_contiguous
        dh, dw = dst.shape[:2]
        sh, row = pix.shape
        sw = row // {DEEP_GRAY16: 2, DEEP_CMYK: 4}.get(kind, 8)
        _check(lib().ipx_draw_deep(self.handle, dst.ctypes.data, dw, dh, dw * 4, _rect(r), pix.ctypes.data, sw, sh, row, kind,
                                   int(sp[0]), int(sp[1]), op))
        return dst

    # ---- source-type variants: *image.NRGBA and *image.YCbCr sources -------------
    def scale_bilinear_nrgba(self, src, dw, dh, sr=None, dr=None, op=OP_OVER, dst=None):
        src = _frame(src)
        sh, sw = src.shape[:2]
        if dst is None:
            dst = np.zeros((dh, dw, 4), np.uint8)
        assert dst.dtype == np.uint8 and dst.flags.c_contiguous and dst.shape == (dh, dw, 4)
        _check(lib().ipx_scale_bilinear_nrgba8(
            self.handle, dst.ctypes.data, dw, dh, dw * 4, _rect(dr if dr is not None else (0, 0, dw, dh)),
            src.ctypes.data, sw, sh, sw * 4, _rect(sr if sr is not None else (0, 0, sw, sh)), op))
        return dst

    def draw_nrgba(self, dst, r, src, sp=(0, 0), op=OP_SRC):
        src = _frame(src)
        assert dst.dtype == np.uint8 and dst.flags.c_contiguous
        dh, dw = dst.shape[:2]
        sh, sw = src.shape[:2]
        _check(lib().ipx_draw_nrgba8(self.handle, dst.ctypes.data, dw, dh, dw * 4, _rect(r), src.ctypes.data, sw, sh,
                                     sw * 4, int(sp[0]), int(sp[1]), op))
        return dst

    @staticmethod
    def _ycbcr(y, cb, cr, ratio):
        y, cb, cr = (np.ascontiguousarray(a, dtype=np.uint8) for a in (y, cb, cr))
        h, w = y.shape
        return _lib.YCbCr(y.ctypes.data, cb.ctypes.data, cr.ctypes.data, w, cb.shape[1], w, h, int(ratio)), (y, cb, cr)

    def scale_bilinear_ycbcr(self, y, cb, cr, ratio, dw, dh, sr=None, dr=None, dst=None):
        st, keep = self._ycbcr(y, cb, cr, ratio)
        if dst is None:
            dst = np.zeros((dh, dw, 4), np.uint8)
        assert dst.dtype == np.uint8 and dst.flags.c_contiguous and dst.shape == (dh, dw, 4)
        _check(lib().ipx_scale_bilinear_ycbcr(self.handle, dst.ctypes.data, dw, dh, dw * 4,
                                              _rect(dr if dr is not None else (0, 0, dw, dh)), C.byref(st),
                                              _rect(sr if sr is not None else (0, 0, st.w, st.h))))
        return dst

    def draw_ycbcr(self, dst, r, y, cb, cr, ratio, sp=(0, 0)):
        st, keep = self._ycbcr(y, cb, cr, ratio)
        assert dst.dtype == np.uint8 and dst.flags.c_contiguous
        dh, dw = dst.shape[:2]
        _check(lib().ipx_draw_ycbcr(self.handle, dst.ctypes.data, dw, dh, dw * 4, _rect(r), C.byref(st), int(sp[0]),
                                    int(sp[1])))
        return dst

    # ---- jpeg.Encode (include/ipx.h, "jpeg.Encode") -------------------------------------------------------
    def jpeg_encode(self, frame, quality=85):
        """jpeg.Encode(w, *image.RGBA, &jpeg.Options{Quality}) of one host frame -> bytes"""
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        h, w = frame.shape[:2]
        out, n = C.c_void_p(), C.c_size_t()
        _check(lib().ipx_jpeg_encode_rgba8(self.handle, frame.ctypes.data, w, h, w * 4, int(quality), C.byref(out), C.byref(n)))
        data = C.string_at(out, n.value)
        lib().ipx_buffer_free(out)
        return data

    def jpeg_fdct_dev(self, src_ptr, w, h, n, coefs_ptr, quality=85, stride=None, frame_stride=None, stream=None):
        _check(lib().ipx_dev_jpeg_fdct_rgba8(self.handle, stream, src_ptr, w, h, stride or w * 4,
                                             frame_stride if frame_stride is not None else w * h * 4, n, int(quality), coefs_ptr))

    def jpeg_encode_batch_dev(self, src_ptr, w, h, n, quality=85, stride=None, frame_stride=None, copy=True):
        """n frames in HBM -> n streams.  copy=False returns (memoryviews into the pinned block, release()) instead of bytes."""
        blob, offs, lens = C.c_void_p(), (C.c_size_t * n)(), (C.c_size_t * n)()
        _check(lib().ipx_jpeg_encode_batch_dev(self.handle, src_ptr, w, h, stride or w * 4,
                                               frame_stride if frame_stride is not None else w * h * 4, n, int(quality),
                                               C.byref(blob), offs, lens))
        if not copy:
            total = offs[n - 1] + lens[n - 1]
            buf = (C.c_uint8 * total).from_address(blob.value)
            mv = memoryview(buf)
            return [mv[offs[i]:offs[i] + lens[i]] for i in range(n)], (lambda: lib().ipx_host_free(self.handle, blob))
        res = [C.string_at(blob.value + offs[i], lens[i]) for i in range(n)]
        lib().ipx_host_free(self.handle, blob)
        return res

    def jpeg_decode_batch(self, files, w=0, h=0, download=True):
        """image.Decode of a batch of JPEG byte strings on the GPU.  -> (info, status list); info = dict(w, h, ratio, ystride,
        cstride, y, cb, cr) with the planes as n x rows x stride arrays (download=True) or device pointers + `free()`."""
        n = len(files)
        keep = [bytes(f) for f in files]
        arr = (_lib.Bytes * n)()
        for i, f in enumerate(keep):
            arr[i].data = C.cast(C.c_char_p(f), C.c_void_p)
            arr[i].len = len(f)
        cw, chh = C.c_int(w), C.c_int(h)
        b = _lib.YCbCrBatch()
        status = (C.c_int * n)()
        owner = C.c_void_p()
        _check(lib().ipx_jpeg_decode_batch(self.handle, None, arr, n, C.byref(cw), C.byref(chh), C.byref(b), status, C.byref(owner)))
        st = list(status)
        if not b.y:
            return None, st
        info = {"w": cw.value, "h": chh.value, "ratio": b.ratio, "ystride": b.ystride, "cstride": b.cstride}
        v0 = 2 if b.ratio in (2, 3) else 1          # 4:2:0 and 4:4:0 halve the chroma rows (Gray: 8 x 8 MCUs)
        myy = (chh.value + 8 * v0 - 1) // (8 * v0)
        yrows, crows = 8 * v0 * myy, 8 * myy            # image.NewYCbCr(Rect(0, 0, 8*h0*mxx, 8*v0*myy), ratio)
        if download:
            def grab(ptr, fs, rows, stride):
                out = np.empty((n, fs), np.uint8)
                _check(lib().ipx_memcpy_d2h(self.handle, out.ctypes.data, ptr, out.nbytes))
                return out[:, :rows * stride].reshape(n, rows, stride)
            info["y"] = grab(b.y, b.y_frame_stride, yrows, b.ystride)
            if b.cb:   # *image.Gray has no chroma planes
                info["cb"] = grab(b.cb, b.c_frame_stride, crows, b.cstride)
                info["cr"] = grab(b.cr, b.c_frame_stride, crows, b.cstride)
            lib().ipx_jpeg_planes_free(self.handle, owner)
        else:
            info.update(batch=b, free=lambda: lib().ipx_jpeg_planes_free(self.handle, owner))
        return info, st

    def composite_glyphs(self, dst, glyphs, col):
        assert dst.dtype == np.uint8 and dst.flags.c_contiguous
        dh, dw = dst.shape[:2]
        arr, keep = _glyph_array(list(glyphs))
        c = (C.c_uint8 * 4)(*[int(v) for v in col])
        _check(lib().ipx_composite_glyphs_rgba8(self.handle, dst.ctypes.data, dw, dh, dw * 4, arr,
                                                len(glyphs), c))
        return dst


class PoolJob:
    """One submitted job of a Pool: keeps the arrays the library writes into alive until wait()."""

    def __init__(self, pool, job, keep, outs, n):
        self.pool, self.job, self.keep, self.outs, self.n = pool, job, keep, outs, n
        t = C.c_uint64()
        _check(lib().ipx_job_submit(pool.handle, C.byref(job), C.byref(t)))
        self.ticket = t.value
        self.released = False

    def done(self):
        d = C.c_int()
        _check(lib().ipx_job_poll(self.pool.handle, self.ticket, C.byref(d)))
        return bool(d.value)

    def wait(self):
        """-> dict of outputs (arrays for a pixel job; ({operator: [bytes | None]}, status list) for a JPEG job); releases the job"""
        fd = C.c_int()
        try:
            _check(lib().ipx_job_wait(self.pool.handle, self.ticket, C.byref(fd)))
            if self.job.kind != 1:          # every kind but IPX_JOB_JPEG hands pixels back
                return self.outs
            out = {k: [C.string_at(a[j].data, a[j].len) if a[j].data else None for j in range(self.n)] for k, a in self.outs.items()}
            return out, list(self.keep["status"])[:self.n]
        finally:
            self.release()

    def release(self):
        if not self.released:
            self.released = True
            lib().ipx_job_release(self.pool.handle, self.ticket)


class Pool:
    """One process, several GPUs (ipx_pool_*): a context per listed device, feeder threads, one largest-first queue."""

    def __init__(self, devices=(0,), lanes_per_device=0, lane_bytes=0):
        arr = (C.c_int * len(devices))(*devices)
        cfg = _lib.PoolConfig(lanes_per_device, lane_bytes)
        h = C.c_void_p()
        _check(lib().ipx_pool_create(arr, len(devices), C.byref(cfg), C.byref(h)))
        self.handle = h.value

    def close(self):
        if self.handle:
            lib().ipx_pool_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def slots(self):
        return lib().ipx_pool_slots(self.handle)

    def frames_done(self, slot):
        return lib().ipx_pool_frames_done(self.handle, slot)

    def host_alloc(self, slot, shape, dtype=np.uint8):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = lib().ipx_pool_host_alloc(self.handle, slot, max(1, n))
        if not p:
            raise IpxError(-2, lib().ipx_last_error().decode())
        arr = np.frombuffer((C.c_uint8 * n).from_address(p), dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = (slot, p)
        return arr

    def host_free(self, arr):
        slot, p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, (None, None))
        if p:
            lib().ipx_pool_host_free(self.handle, slot, p)

    @staticmethod
    def _ops(sw, sh, resize, thumbnail, glyphs, col, watermark):
        o = _lib.PoolOps()
        o.sw, o.sh = sw, sh
        if resize:
            o.do_resize, o.resize_w, o.resize_h, o.keep_aspect = 1, resize[0], resize[1], int(bool(resize[2]))
        if thumbnail:
            o.do_thumbnail, o.thumb_size, o.crop_to_fit = 1, thumbnail[0], int(bool(thumbnail[1]))
        keep = None
        if watermark or glyphs:
            o.do_watermark = 1
            if glyphs:
                arr, keep = _glyph_array(list(glyphs))
                o.glyphs, o.n_glyphs = arr, len(glyphs)
                keep = (arr, keep)
                for i in range(4):
                    o.col[i] = int(col[i])
        return o, keep

    JOB_KINDS = {"rgba": (0, 4), "nrgba": (2, 4), "gray": (3, 1), "nrgba64": (4, 8), "rgba64": (5, 8), "gray16": (6, 2), "cmyk": (7, 4)}   # IPX_JOB_*: (kind, bytes per pixel)

    def submit(self, frames, resize=(1024, 768, True), thumbnail=(200, True), glyphs=None, col=(0, 0, 0, 0), watermark=False, out=None,
               kind="rgba"):
        """frames: n x H x W x 4 uint8 (host); for another `kind` (JOB_KINDS) n x H x (W * bpp) uint8, the type's Pix rows.
        -> PoolJob; wait() gives {"resize", "thumbnail", "watermark"} arrays."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        jk, bpp = self.JOB_KINDS[kind]
        if kind != "rgba":
            frames = frames.reshape(frames.shape[0], frames.shape[1], -1)
        n, sh = frames.shape[:2]
        sw = frames.shape[2] if kind == "rgba" else frames.shape[2] // bpp
        ops, keep = self._ops(sw, sh, resize, thumbnail, glyphs, col, watermark)
        outs = dict(out) if out else {}
        if resize and "resize" not in outs:
            outs["resize"] = np.empty((n,) + tuple(reversed(resize_dims(sw, sh, resize[0], resize[1], resize[2]))) + (4,), np.uint8)
        if thumbnail and "thumbnail" not in outs:
            _, tw, th = thumb_geometry(sw, sh, thumbnail[0] or 200, thumbnail[1])
            outs["thumbnail"] = np.empty((n, th, tw, 4), np.uint8)
        if ops.do_watermark and "watermark" not in outs:
            outs["watermark"] = np.empty((n, sh, sw, 4), np.uint8)
        j = _lib.Job()
        j.kind, j.ops, j.n = jk, ops, n
        j.src, j.sstride, j.src_frame_stride = frames.ctypes.data, sw * bpp, sw * sh * bpp

        def fs(a):
            return int(np.prod(a.shape[1:]))
        if "resize" in outs:
            j.resize_out, j.resize_frame_stride = outs["resize"].ctypes.data, fs(outs["resize"])
        if "thumbnail" in outs:
            j.thumb_out, j.thumb_frame_stride = outs["thumbnail"].ctypes.data, fs(outs["thumbnail"])
        if "watermark" in outs:
            j.wm_out, j.wm_frame_stride = outs["watermark"].ctypes.data, fs(outs["watermark"])
        return PoolJob(self, j, {"frames": frames, "glyphs": keep}, outs, n)

    def submit_jpeg(self, files, sw, sh, quality=85, resize=(1024, 768, True), thumbnail=(200, True), glyphs=None, col=(0, 0, 0, 0),
                    watermark=False):
        """JPEG byte strings of sw x sh images in -> PoolJob; wait() gives ({operator: [bytes | None]}, status list)."""
        n = len(files)
        ops, keep = self._ops(sw, sh, resize, thumbnail, glyphs, col, watermark)
        blobs = [bytes(f) for f in files]
        arr = (_lib.Bytes * max(1, n))()
        for i, f in enumerate(blobs):
            arr[i].data = C.cast(C.c_char_p(f), C.c_void_p)
            arr[i].len = len(f)
        status = (C.c_int32 * max(1, n))()
        outs = {}
        j = _lib.Job()
        j.kind, j.ops, j.n, j.files, j.quality, j.status = 1, ops, n, arr, int(quality), status
        if resize:
            outs["resize"] = (_lib.Bytes * max(1, n))()
            j.resize_jpeg = outs["resize"]
        if thumbnail:
            outs["thumbnail"] = (_lib.Bytes * max(1, n))()
            j.thumb_jpeg = outs["thumbnail"]
        if ops.do_watermark:
            outs["watermark"] = (_lib.Bytes * max(1, n))()
            j.wm_jpeg = outs["watermark"]
        return PoolJob(self, j, {"blobs": blobs, "files": arr, "status": status, "glyphs": keep}, outs, n)


def jpeg_entropy_encode(coefs, w, h, quality=85):
    """Host half of jpeg.Encode: quantised coefficients (int16, 6 x 64 per MCU, zig-zag) -> the byte stream."""
    coefs = np.ascontiguousarray(coefs, dtype=np.int16)
    assert coefs.size == lib().ipx_jpeg_coef_count(w, h)
    out, n = C.c_void_p(), C.c_size_t()
    _check(lib().ipx_jpeg_entropy_encode(coefs.ctypes.data, w, h, int(quality), C.byref(out), C.byref(n)))
    data = C.string_at(out, n.value)
    lib().ipx_buffer_free(out)
    return data


def jpeg_quant_tables(quality):
    out = (C.c_uint8 * 128)()
    _check(lib().ipx_jpeg_quant_tables(int(quality), out))
    return np.frombuffer(out, np.uint8).reshape(2, 64).copy()


class Batcher:
    """Micro-batching of single uploads (ipx_batcher_*): what the goroutines of internal/worker/worker.go:112-149 would call, one file each.

    submit(file bytes, frame size, operators) -> ticket; wait(ticket) -> (status, {operator: bytes | None}); the ticket is released by wait."""

    def __init__(self, pool, max_batch=0, max_wait_us=0, quality=0):
        self.pool = pool
        cfg = _lib.BatcherConfig(max_batch, max_wait_us, quality)
        h = C.c_void_p()
        _check(lib().ipx_batcher_create(pool.handle, C.byref(cfg), C.byref(h)))
        self.handle = h.value
        self._keep = {}
        self._mu = __import__("threading").Lock()

    def close(self):
        if self.handle:
            lib().ipx_batcher_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def submit(self, data, sw, sh, resize=None, thumbnail=None, glyphs=None, col=(255, 255, 255, 127), watermark=False):
        ops_, keep = Pool._ops(sw, sh, resize, thumbnail, glyphs, col, watermark)
        buf = C.create_string_buffer(bytes(data), len(data))
        fb = _lib.Bytes(C.cast(buf, C.c_void_p), len(data))
        t = C.c_uint64()
        _check(lib().ipx_batcher_submit(self.handle, C.byref(fb), C.byref(ops_), C.byref(t)))
        with self._mu:
            self._keep[t.value] = buf          # the file's bytes stay where they are until the ticket is released
        return t.value

    def wait(self, ticket):
        res = _lib.BatchResult()
        try:
            _check(lib().ipx_batcher_wait(self.handle, ticket, C.byref(res)))
            out = {k: (C.string_at(getattr(res, f).data, getattr(res, f).len) if getattr(res, f).data else None)
                   for k, f in (("resize", "resize"), ("thumbnail", "thumb"), ("watermark", "wm"))}
            return res.status, out
        finally:
            lib().ipx_batcher_release(self.handle, ticket)
            with self._mu:
                self._keep.pop(ticket, None)

    def stats(self):
        st = _lib.BatcherStats()
        _check(lib().ipx_batcher_get_stats(self.handle, C.byref(st)))
        return {k: getattr(st, k) for k, _ in st._fields_}
