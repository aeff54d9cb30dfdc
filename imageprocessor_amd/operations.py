"""The reference's operator interface on decoded frames: Resizer / Thumbnailer / Watermarker with
Process(ctx, img, format, params) and ImageProcessor.Process(task, decoded) -- same names, argument
meaning, defaults and error texts as internal/usecase/processor (operations/*.go,
image_processor.go), with the codecs cut off: frames in and out are H x W x 4 uint8 arrays (Go's
*image.RGBA).  Thin ctypes front end of the ipx_*_process entry points of include/ipx.h."""
import ctypes as C

import numpy as np

from . import IpxError, _frame, lib
from ._lib import (GLYPHS_FN, MEASURE_FN, RELEASE_FN, Glyph, Image, Operation, Param, Processed, Rect, Task,
                   TextRasterizer)

PT_FLOAT64, PT_INT, PT_INT64, PT_INT32, PT_BOOL, PT_STRING = 1, 2, 3, 4, 5, 6


class Int64(int):
    """A Go int64 parameter value (plain Python ints map to Go int, floats to float64)."""


class Int32(int):
    """A Go int32 parameter value."""


def _params(d):
    keep = []
    arr = (Param * max(1, len(d)))()
    for i, (k, v) in enumerate(d.items()):
        kb = k.encode()
        keep.append(kb)
        p = Param(kb, 0, 0.0, 0, None)
        if isinstance(v, bool):
            p.type, p.i64 = PT_BOOL, int(v)
        elif isinstance(v, Int64):
            p.type, p.i64 = PT_INT64, int(v)
        elif isinstance(v, Int32):
            p.type, p.i64 = PT_INT32, int(v)
        elif isinstance(v, int):
            p.type, p.i64 = PT_INT, v
        elif isinstance(v, float):
            p.type, p.f64 = PT_FLOAT64, v
        elif isinstance(v, str):
            sb = v.encode()
            keep.append(sb)
            p.type, p.str = PT_STRING, sb
        else:
            p.type = 0  # any other Go type: every type assertion of the reference fails
        arr[i] = p
    return arr, len(d), keep


def _image(a):
    a = _frame(a)
    return Image(a.ctypes.data, a.shape[1], a.shape[0], a.shape[1] * 4), a


def _take(img):
    """Copies a library-owned ipx_image into numpy and frees it."""
    if img.w <= 0 or img.h <= 0 or not img.pix:
        out = np.zeros((max(img.h, 0), max(img.w, 0), 4), np.uint8)
    else:
        buf = (C.c_uint8 * (img.h * img.stride)).from_address(img.pix)
        out = np.frombuffer(buf, np.uint8).reshape(img.h, img.stride // 4, 4)[:, :img.w].copy()
    lib().ipx_image_free(C.byref(img))
    return out


class Font:
    """Host-side text rasteriser handed across the boundary (ipx_text_rasterizer).  `measure(text,
    font_size) -> width_px` and `glyphs(text, font_size, px, py, w, h) -> [{"mask", "dr", "mp"}]` are the
    two things the reference asks of golang/freetype (watermark.go:105-117 and :151)."""

    def __init__(self, measure, glyphs):
        self._measure, self._glyphs = measure, glyphs
        self._keep = None

        def c_measure(user, text, size, out):
            try:
                out[0] = int(self._measure(text.decode(), size))
                return 0
            except Exception:
                return 1

        def c_glyphs(user, text, size, px, py, w, h, out, n):
            try:
                gl = list(self._glyphs(text.decode(), size, px, py, w, h))
                arr = (Glyph * max(1, len(gl)))()
                masks = []
                for i, g in enumerate(gl):
                    m = np.ascontiguousarray(g["mask"], dtype=np.uint8)
                    masks.append(m)
                    mp = g.get("mp", (0, 0))
                    arr[i] = Glyph(m.ctypes.data, m.shape[1], m.shape[0], m.shape[1], Rect(*[int(v) for v in g["dr"]]),
                                   int(mp[0]), int(mp[1]))
                self._keep = (arr, masks)
                out[0] = C.cast(arr, C.POINTER(Glyph))
                n[0] = len(gl)
                return 0
            except Exception:
                return 1

        def c_release(user):
            self._keep = None

        self._cb = (MEASURE_FN(c_measure), GLYPHS_FN(c_glyphs), RELEASE_FN(c_release))
        self.struct = TextRasterizer(None, *self._cb)


class TrueTypeFont:
    """truetype.Parse + freetype.Context.DrawString of the library itself (ipx_font_*, csrc/ipx_font.cpp):
    what NewWatermarker holds (watermark.go:25-38).  Usable wherever a `Font` is."""

    def __init__(self, ttf_bytes):
        self.handle = C.c_void_p()
        self._bytes = bytes(ttf_bytes)
        rc = lib().ipx_font_create(self._bytes, len(self._bytes), C.byref(self.handle))
        if rc:
            raise IpxError(rc, lib().ipx_last_error().decode(errors="replace"))
        self.struct = TextRasterizer()
        lib().ipx_font_rasterizer(self.handle, C.byref(self.struct))

    @classmethod
    def from_file(cls, path):
        with open(path, "rb") as f:
            return cls(f.read())

    def index(self, rune):
        return lib().ipx_font_glyph_index(self.handle, ord(rune) if isinstance(rune, str) else int(rune))

    def glyph_advance(self, rune, size):
        v = C.c_int32()
        rc = lib().ipx_font_glyph_advance(self.handle, ord(rune) if isinstance(rune, str) else int(rune), size, C.byref(v))
        if rc:
            raise IpxError(rc, lib().ipx_last_error().decode(errors="replace"))
        return v.value

    def kern(self, r0, r1, size):
        v = C.c_int32()
        rc = lib().ipx_font_kern(self.handle, ord(r0), ord(r1), size, C.byref(v))
        if rc:
            raise IpxError(rc, lib().ipx_last_error().decode(errors="replace"))
        return v.value

    def text_width(self, text, size):
        """-> (textWidth as 26.6 fixed, int(textWidth.Ceil())) of watermark.go:108-117"""
        w, px = C.c_int32(), C.c_int()
        rc = lib().ipx_font_text_width(self.handle, text.encode(), size, C.byref(w), C.byref(px))
        if rc:
            raise IpxError(rc, lib().ipx_last_error().decode(errors="replace"))
        return w.value, px.value

    def draw_string(self, text, size, px, py, clip_w, clip_h):
        """-> ([{"mask", "dr", "mp"}] in DrawMask order, X of the returned point as 26.6 fixed)"""
        out, n, endx = C.POINTER(Glyph)(), C.c_int(), C.c_int32()
        rc = lib().ipx_font_draw_string(self.handle, text.encode() if isinstance(text, str) else text, size, px, py,
                                        clip_w, clip_h, C.byref(out), C.byref(n), C.byref(endx))
        if rc:
            raise IpxError(rc, lib().ipx_last_error().decode(errors="replace"))
        gl = []
        for i in range(n.value):
            g = out[i]
            m = np.frombuffer((C.c_uint8 * (g.mh * g.mstride)).from_address(g.mask), np.uint8).reshape(g.mh, g.mstride)
            gl.append({"mask": m[:, :g.mw].copy(), "dr": (g.dr.x0, g.dr.y0, g.dr.x1, g.dr.y1), "mp": (g.mpx, g.mpy)})
        lib().ipx_font_release_thread()
        return gl, endx.value

    def close(self):
        if self.handle:
            lib().ipx_font_destroy(self.handle)
            self.handle = C.c_void_p()


def _font_ptr(font):
    return C.byref(font.struct) if font is not None else None


class _Op:
    _fn = None

    def Process(self, ctx, img, format, params):
        """-> (frame, format) like the reference's (io.Reader, string, error) minus the encoder."""
        im, keep = _image(img)
        arr, n, keep2 = _params(params)
        out, fmt = Image(), C.create_string_buffer(8)
        rc = getattr(lib(), self._fn)(ctx.handle, C.byref(im), format.encode(), arr, n, C.byref(out), fmt)
        if rc:
            raise IpxError(rc, lib().ipx_last_error().decode(errors="replace"))
        return _take(out), fmt.value.decode()


class Resizer(_Op):           # operations/resize.go:20-24
    _fn = "ipx_resizer_process"


class Thumbnailer(_Op):       # operations/thumbnail.go:19-23
    _fn = "ipx_thumbnailer_process"


class Watermarker:            # operations/watermark.go:25-38
    def __init__(self, font=None):
        self.font = font      # None mirrors a Watermarker whose font failed to parse (:32-34)

    def Process(self, ctx, img, format, params):
        im, keep = _image(img)
        arr, n, keep2 = _params(params)
        out, fmt = Image(), C.create_string_buffer(8)
        rc = lib().ipx_watermarker_process(ctx.handle, C.byref(im), format.encode(), arr, n, _font_ptr(self.font),
                                           C.byref(out), fmt)
        if rc:
            raise IpxError(rc, lib().ipx_last_error().decode(errors="replace"))
        return _take(out), fmt.value.decode()


class ImageProcessor:         # image_processor.go:21-37
    def __init__(self, ctx, font=None):
        self.ctx, self.font = ctx, font

    def Process(self, task, decoded, decoded_format="jpeg"):
        """task: {"ID", "ImageID", "Operations": [{"Type", "Parameters"}], "Format"} (the JSON of
        domain.ProcessingTask, task.go:3-15).  -> (result dict, error text or None), result =
        {"ID", "ImageID", "Status", "ProcessedPaths", "Error", "Outputs": {op: (frame, content_type)}}"""
        im, keep = _image(decoded)
        ops = task.get("Operations") or []
        oarr = (Operation * max(1, len(ops)))()
        keep2 = []
        for i, op in enumerate(ops):
            arr, n, k = _params(op.get("Parameters") or {})
            tb = str(op.get("Type", "")).encode()
            keep2 += [arr, k, tb]
            oarr[i] = Operation(tb, C.cast(arr, C.POINTER(Param)), n)
        t = Task(str(task.get("ID", "")).encode(), str(task.get("ImageID", "")).encode(),
                 C.cast(oarr, C.POINTER(Operation)), len(ops), str(task.get("Format", "") or "").encode())
        outs = (Processed * max(1, len(ops)))()
        nout = C.c_int()
        rc = lib().ipx_processor_process(self.ctx.handle, C.byref(t), C.byref(im), decoded_format.encode(),
                                         _font_ptr(self.font), outs, C.byref(nout))
        err = lib().ipx_last_error().decode(errors="replace") if rc else None
        res = {"ID": task.get("ID", ""), "ImageID": task.get("ImageID", ""), "Status": "completed", "ProcessedPaths": {},
               "Error": "", "Outputs": {}}
        for i in range(nout.value):
            o = outs[i]
            name = o.operation.decode()
            res["ProcessedPaths"][name] = o.path.decode()
            res["Outputs"][name] = (_take(o.image), o.content_type.decode(), o.format.decode())
        if rc:   # image_processor.go:66-75
            res["Status"] = "failed"
            first = err.split(" failed: ", 1)
            res["Error"] = "Operation %s failed: %s" % (first[0][len("operation "):], first[1]) if len(first) == 2 else err
        return res, err
