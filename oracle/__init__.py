"""ctypes front end of the CPU oracle (oracle/ipx_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of bench.py -- never by anything under ``imageprocessor_amd/``.

PARITY UNPINNED: the reference holds no golden vectors for this path;
the restatement is pinned only by the hand-derived known answers under tests/golden/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libipx_oracle.so")

OP_OVER = 0
OP_SRC = 1


class Rect(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32)]


class Glyph(C.Structure):
    _fields_ = [("mask", C.c_void_p), ("mw", C.c_int32), ("mh", C.c_int32),
                ("mstride", C.c_int32), ("dr", Rect), ("mpx", C.c_int32), ("mpy", C.c_int32)]


class Pipeline(C.Structure):
    _fields_ = [("resize_w", C.c_int), ("resize_h", C.c_int), ("keep_aspect", C.c_int),
                ("thumb_size", C.c_int), ("crop_to_fit", C.c_int),
                ("glyphs", C.POINTER(Glyph)), ("n_glyphs", C.c_int), ("col", C.c_uint8 * 4)]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("ipx_oracle.c", "ipx_jpeg_oracle.c", "ipx_jpeg_dec_oracle.c", "ipx_oracle.h"))
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < src_m:
        subprocess.check_call(["make", "-s", "-C", _HERE, "libipx_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()          # (a no-op when the library is newer than its sources; a stale one would miss entry points added since)
        L = C.CDLL(_SO)
        L.ipxo_scale_bilinear_rgba8.restype = C.c_int
        L.ipxo_scale_bilinear_rgba8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, Rect,
                                                C.c_void_p, C.c_int, C.c_int, C.c_int, Rect,
                                                C.c_int]
        L.ipxo_draw_rgba8.restype = None
        L.ipxo_draw_rgba8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, Rect,
                                      C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int]
        L.ipxo_composite_glyphs_rgba8.restype = None
        L.ipxo_composite_glyphs_rgba8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                  C.POINTER(Glyph), C.c_int, C.c_void_p]
        L.ipxo_process_rgba8.restype = C.c_int
        L.ipxo_process_rgba8.argtypes = [C.POINTER(Pipeline), C.c_void_p, C.c_int, C.c_int,
                                         C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ipxo_text_height_px.restype = C.c_int
        L.ipxo_text_height_px.argtypes = [C.c_double]
        L.ipxo_parse_color.restype = C.c_int
        L.ipxo_parse_color.argtypes = [C.c_char_p, C.c_double, C.c_void_p]
        _lib = L
    return _lib


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.ndim == 3 and a.shape[2] == 4, "RGBA8 frame must be H x W x 4"
    return a


def _rect(r):
    return r if isinstance(r, Rect) else Rect(*[int(v) for v in r])


def resize_dims(ow, oh, w, h, keep_aspect):
    nw, nh = C.c_int(), C.c_int()
    lib().ipxo_resize_dims(ow, oh, w, h, int(bool(keep_aspect)), C.byref(nw), C.byref(nh))
    return nw.value, nh.value


def thumb_geometry(ow, oh, size, crop_to_fit):
    r, nw, nh = Rect(), C.c_int(), C.c_int()
    lib().ipxo_thumb_geometry(ow, oh, size, int(bool(crop_to_fit)), C.byref(r), C.byref(nw),
                              C.byref(nh))
    return (r.x0, r.y0, r.x1, r.y1), nw.value, nh.value


def watermark_anchor(position, w, h, width_px, height_px):
    px, py = C.c_int(), C.c_int()
    lib().ipxo_watermark_anchor(position.encode(), w, h, width_px, height_px, C.byref(px),
                                C.byref(py))
    return px.value, py.value


def text_height_px(font_size):
    return lib().ipxo_text_height_px(float(font_size))


def parse_color(s, opacity):
    out = (C.c_uint8 * 4)()
    err = lib().ipxo_parse_color(s.encode(), float(opacity), out)
    return tuple(out), bool(err)


def scale_bilinear(src, dw, dh, sr=None, dr=None, op=OP_OVER, dst=None):
    """BiLinear.Scale(dst, dr, src, sr, op, nil); dst defaults to a zeroed dw x dh frame."""
    src = _u8(src)
    sh, sw = src.shape[:2]
    if dst is None:
        dst = np.zeros((dh, dw, 4), np.uint8)
    dst = _u8(dst)
    assert dst.shape[:2] == (dh, dw)
    sr = _rect(sr if sr is not None else (0, 0, sw, sh))
    dr = _rect(dr if dr is not None else (0, 0, dw, dh))
    rc = lib().ipxo_scale_bilinear_rgba8(dst.ctypes.data, dw, dh, dw * 4, dr, src.ctypes.data, sw,
                                         sh, sw * 4, sr, op)
    if rc:
        raise ValueError("source rectangle leaves the source image")
    return dst


def draw(dst, r, src, sp=(0, 0), op=OP_SRC):
    """image/draw.DrawMask(dst, r, src, sp, nil, ZP, op) in place on dst."""
    src = _u8(src)
    assert dst.dtype == np.uint8 and dst.flags.c_contiguous
    dh, dw = dst.shape[:2]
    sh, sw = src.shape[:2]
    lib().ipxo_draw_rgba8(dst.ctypes.data, dw, dh, dw * 4, _rect(r), src.ctypes.data, sw, sh,
                          sw * 4, int(sp[0]), int(sp[1]), op)
    return dst


def _glyph_array(glyphs):
    keep = []
    arr = (Glyph * max(1, len(glyphs)))()
    for i, g in enumerate(glyphs):
        m = np.ascontiguousarray(g["mask"], dtype=np.uint8)
        keep.append(m)
        mp = g.get("mp", (0, 0))
        arr[i] = Glyph(m.ctypes.data, m.shape[1], m.shape[0], m.shape[1], _rect(g["dr"]),
                       int(mp[0]), int(mp[1]))
    return arr, keep


def composite_glyphs(dst, glyphs, col):
    """DrawMask(dst, dr, Uniform(col), ZP, mask, mp, Over) per glyph, in order, in place."""
    assert dst.dtype == np.uint8 and dst.flags.c_contiguous
    dh, dw = dst.shape[:2]
    arr, keep = _glyph_array(glyphs)
    c = (C.c_uint8 * 4)(*[int(v) for v in col])
    lib().ipxo_composite_glyphs_rgba8(dst.ctypes.data, dw, dh, dw * 4, arr, len(glyphs), c)
    return dst


def process(src, resize=(1024, 768, True), thumb=(200, True), glyphs=(), col=(255, 255, 255, 127),
            want=("resize", "thumbnail", "watermark")):
    """The three operators on the ORIGINAL frame (image_processor.go:64-65)."""
    src = _u8(src)
    sh, sw = src.shape[:2]
    arr, keep = _glyph_array(list(glyphs))
    p = Pipeline(resize[0], resize[1], int(bool(resize[2])), thumb[0], int(bool(thumb[1])),
                 C.cast(arr, C.POINTER(Glyph)), len(glyphs), (C.c_uint8 * 4)(*col))
    out = {}
    ptr = {"resize": None, "thumbnail": None, "watermark": None}
    if "resize" in want:
        nw, nh = resize_dims(sw, sh, *resize)
        out["resize"] = np.empty((nh, nw, 4), np.uint8)
        ptr["resize"] = out["resize"].ctypes.data
    if "thumbnail" in want:
        _, nw, nh = thumb_geometry(sw, sh, *thumb)
        out["thumbnail"] = np.empty((nh, nw, 4), np.uint8)
        ptr["thumbnail"] = out["thumbnail"].ctypes.data
    if "watermark" in want:
        out["watermark"] = np.empty((sh, sw, 4), np.uint8)
        ptr["watermark"] = out["watermark"].ctypes.data
    rc = lib().ipxo_process_rgba8(C.byref(p), src.ctypes.data, sw, sh, sw * 4, ptr["resize"],
                                  ptr["thumbnail"], ptr["watermark"])
    if rc:
        raise RuntimeError("oracle pipeline failed: %d" % rc)
    return out


# ---- source-type variants ------------------------------------------------------

class YCbCrStruct(C.Structure):
    _fields_ = [("y", C.c_void_p), ("cb", C.c_void_p), ("cr", C.c_void_p), ("ystride", C.c_int32),
                ("cstride", C.c_int32), ("w", C.c_int32), ("h", C.c_int32), ("ratio", C.c_int32)]


RATIO_444, RATIO_422, RATIO_420, RATIO_440 = 0, 1, 2, 3


def chroma_shape(w, h, ratio):
    """Plane size of Cb / Cr for an image.YCbCr with Rect.Min = (0,0) (image.NewYCbCr)."""
    cw = (w + 1) // 2 if ratio in (RATIO_422, RATIO_420) else w
    chh = (h + 1) // 2 if ratio in (RATIO_420, RATIO_440) else h
    return chh, cw


def _ycbcr(y, cb, cr, ratio):
    y = np.ascontiguousarray(y, np.uint8)
    cb = np.ascontiguousarray(cb, np.uint8)
    cr = np.ascontiguousarray(cr, np.uint8)
    h, w = y.shape
    assert cb.shape == cr.shape == chroma_shape(w, h, ratio), (cb.shape, chroma_shape(w, h, ratio))
    return YCbCrStruct(y.ctypes.data, cb.ctypes.data, cr.ctypes.data, w, cb.shape[1], w, h, ratio), (y, cb, cr)


def _decl_variants():
    L = lib()
    if getattr(L, "_variants", False):
        return L
    L.ipxo_scale_bilinear_nrgba8.restype = C.c_int
    L.ipxo_scale_bilinear_nrgba8.argtypes = L.ipxo_scale_bilinear_rgba8.argtypes
    L.ipxo_draw_nrgba8.restype = None
    L.ipxo_draw_nrgba8.argtypes = L.ipxo_draw_rgba8.argtypes
    L.ipxo_scale_bilinear_ycbcr.restype = C.c_int
    L.ipxo_scale_bilinear_ycbcr.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, Rect, C.POINTER(YCbCrStruct), Rect]
    L.ipxo_draw_ycbcr.restype = None
    L.ipxo_draw_ycbcr.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, Rect, C.POINTER(YCbCrStruct), C.c_int, C.c_int]
    L.ipxo_scale_bilinear_paletted.restype = C.c_int
    L.ipxo_scale_bilinear_paletted.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, Rect, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, Rect, C.c_int]
    L.ipxo_draw_paletted.restype = None
    L.ipxo_draw_paletted.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, Rect, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.ipxo_scale_bilinear_deep.restype = C.c_int
    L.ipxo_scale_bilinear_deep.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, Rect, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, Rect, C.c_int]
    L.ipxo_draw_deep.restype = None
    L.ipxo_draw_deep.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, Rect, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.ipxo_deep_taps.restype = None
    L.ipxo_deep_taps.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L._variants = True
    return L


def scale_bilinear_nrgba(src, dw, dh, sr=None, dr=None, op=OP_OVER, dst=None):
    src = _u8(src)
    sh, sw = src.shape[:2]
    if dst is None:
        dst = np.zeros((dh, dw, 4), np.uint8)
    dst = _u8(dst)
    rc = _decl_variants().ipxo_scale_bilinear_nrgba8(dst.ctypes.data, dw, dh, dw * 4,
                                                     _rect(dr if dr is not None else (0, 0, dw, dh)), src.ctypes.data,
                                                     sw, sh, sw * 4, _rect(sr if sr is not None else (0, 0, sw, sh)), op)
    if rc:
        raise ValueError("source rectangle leaves the source image")
    return dst


def draw_nrgba(dst, r, src, sp=(0, 0), op=OP_SRC):
    src = _u8(src)
    dh, dw = dst.shape[:2]
    sh, sw = src.shape[:2]
    _decl_variants().ipxo_draw_nrgba8(dst.ctypes.data, dw, dh, dw * 4, _rect(r), src.ctypes.data, sw, sh, sw * 4,
                                      int(sp[0]), int(sp[1]), op)
    return dst


DEEP_NRGBA64, DEEP_RGBA64, DEEP_GRAY16, DEEP_CMYK = 0, 1, 2, 3
DEEP_BPP = {DEEP_NRGBA64: 8, DEEP_RGBA64: 8, DEEP_GRAY16: 2, DEEP_CMYK: 4}


def deep_pix(values, kind):
    """Go's Pix for a deep image: `values` (h, w, 4) uint16 for NRGBA64 / RGBA64, (h, w) uint16 for Gray16 -> big-endian bytes;
    (h, w, 4) uint8 C M Y K for CMYK -> as is.  Returns (h, w * bpp) uint8."""
    v = np.asarray(values)
    if kind == DEEP_CMYK:
        return np.ascontiguousarray(v, np.uint8).reshape(v.shape[0], -1)
    return np.ascontiguousarray(v.astype(">u2")).view(np.uint8).reshape(v.shape[0], -1)


def _deep(src, kind):
    src = np.ascontiguousarray(src, np.uint8)
    sh, row = src.shape
    return src, row // DEEP_BPP[kind], sh, row


def deep_taps(src, kind):
    """At(x, y).RGBA() of every pixel of a deep frame (Pix rows as deep_pix makes them) -> (h, w, 4) uint16."""
    src, sw, sh, row = _deep(src, kind)
    out = np.zeros((sh, sw, 4), np.uint16)
    _decl_variants().ipxo_deep_taps(out.ctypes.data, src.ctypes.data, sw, sh, row, kind)
    return out


def scale_bilinear_deep(src, kind, dw, dh, sr=None, dr=None, op=OP_OVER, dst=None):
    src, sw, sh, row = _deep(src, kind)
    if dst is None:
        dst = np.zeros((dh, dw, 4), np.uint8)
    dst = _u8(dst)
    rc = _decl_variants().ipxo_scale_bilinear_deep(dst.ctypes.data, dw, dh, dw * 4, _rect(dr if dr is not None else (0, 0, dw, dh)),
                                                   src.ctypes.data, sw, sh, row, kind, _rect(sr if sr is not None else (0, 0, sw, sh)), op)
    if rc:
        raise ValueError("source rectangle leaves the source image")
    return dst


def draw_deep(dst, r, src, kind, sp=(0, 0), op=OP_SRC):
    src, sw, sh, row = _deep(src, kind)
    dh, dw = dst.shape[:2]
    _decl_variants().ipxo_draw_deep(dst.ctypes.data, dw, dh, dw * 4, _rect(r), src.ctypes.data, sw, sh, row, kind, int(sp[0]), int(sp[1]), op)
    return dst


def palette16(entries, kind="nrgba"):
    """Palette[i].RGBA() for 256 entries (missing ones: the zero colour).  kind "rgba": color.RGBA entries (the GIF decoder's, and a PNG
    palette without tRNS) -> c | c<<8; kind "nrgba": color.NRGBA entries (a PNG palette with tRNS) -> color.NRGBA.RGBA: c |= c<<8;
    c *= a; c /= 0xff; alpha a | a<<8."""
    e = np.zeros((256, 4), np.uint32)
    src = np.asarray(entries, np.uint32).reshape(-1, 4)
    e[: len(src)] = src
    out = np.zeros((256, 4), np.uint16)
    if kind == "rgba":
        out[:] = e * 0x101
    else:
        out[:, :3] = (e[:, :3] * 0x101) * e[:, 3:4] // 0xff
        out[:, 3] = e[:, 3] * 0x101
    return out


def scale_bilinear_paletted(idx, pal16, dw, dh, sr=None, dr=None, op=OP_OVER, dst=None):
    idx = np.ascontiguousarray(idx, np.uint8)
    pal16 = np.ascontiguousarray(pal16, np.uint16)
    sh, sw = idx.shape
    if dst is None:
        dst = np.zeros((dh, dw, 4), np.uint8)
    dst = _u8(dst)
    rc = _decl_variants().ipxo_scale_bilinear_paletted(dst.ctypes.data, dw, dh, dw * 4, _rect(dr if dr is not None else (0, 0, dw, dh)),
                                                       idx.ctypes.data, sw, sh, sw, pal16.ctypes.data,
                                                       _rect(sr if sr is not None else (0, 0, sw, sh)), op)
    if rc:
        raise ValueError("source rectangle leaves the source image")
    return dst


def draw_paletted(dst, r, idx, pal16, sp=(0, 0), op=OP_SRC):
    idx = np.ascontiguousarray(idx, np.uint8)
    pal16 = np.ascontiguousarray(pal16, np.uint16)
    dh, dw = dst.shape[:2]
    sh, sw = idx.shape
    _decl_variants().ipxo_draw_paletted(dst.ctypes.data, dw, dh, dw * 4, _rect(r), idx.ctypes.data, sw, sh, sw, pal16.ctypes.data,
                                        int(sp[0]), int(sp[1]), op)
    return dst


def scale_bilinear_ycbcr(y, cb, cr, ratio, dw, dh, sr=None, dr=None, dst=None):
    st, keep = _ycbcr(y, cb, cr, ratio)
    if dst is None:
        dst = np.zeros((dh, dw, 4), np.uint8)
    dst = _u8(dst)
    rc = _decl_variants().ipxo_scale_bilinear_ycbcr(dst.ctypes.data, dw, dh, dw * 4,
                                                    _rect(dr if dr is not None else (0, 0, dw, dh)), C.byref(st),
                                                    _rect(sr if sr is not None else (0, 0, st.w, st.h)))
    if rc:
        raise ValueError("source rectangle leaves the source image")
    return dst


def draw_ycbcr(dst, r, y, cb, cr, ratio, sp=(0, 0)):
    st, keep = _ycbcr(y, cb, cr, ratio)
    dh, dw = dst.shape[:2]
    _decl_variants().ipxo_draw_ycbcr(dst.ctypes.data, dw, dh, dw * 4, _rect(r), C.byref(st), int(sp[0]), int(sp[1]))
    return dst


# ---- image/jpeg encoder (oracle/ipx_jpeg_oracle.c) ----------------------------------------------------
def jpeg_quant(quality):
    """The two quantisation tables of jpeg.Encode at `quality`, zig-zag order as DQT carries them."""
    out = (C.c_uint8 * 128)()
    lib().ipxo_jpeg_quant(int(quality), out)
    return np.frombuffer(out, np.uint8).reshape(2, 64).copy()


def _take_bytes(p, n):
    data = C.string_at(p, n.value)
    lib().ipxo_free(p)
    return data


def jpeg_encode_rgba(frame, quality=85, want_coefs=False):
    """jpeg.Encode(w, *image.RGBA, &jpeg.Options{Quality}) -> bytes (and the quantised blocks in scan order)."""
    L = lib()
    L.ipxo_jpeg_encode_rgba8.restype = C.c_int
    L.ipxo_jpeg_encode_rgba8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_size_t), C.c_void_p]
    L.ipxo_free.argtypes = [C.c_void_p]
    f = _u8(frame)
    h, w = f.shape[:2]
    coefs = np.zeros((((h + 15) // 16) * ((w + 15) // 16), 6, 64), np.int16) if want_coefs else None
    p, n = C.c_void_p(), C.c_size_t()
    rc = L.ipxo_jpeg_encode_rgba8(f.ctypes.data, w, h, f.strides[0], int(quality), C.byref(p), C.byref(n),
                                  coefs.ctypes.data if want_coefs else None)
    if rc:
        raise ValueError("jpeg: image is too large to encode" if rc == -1 else "out of memory")
    data = _take_bytes(p, n)
    return (data, coefs) if want_coefs else data


def jpeg_encode_gray(plane, quality=85):
    L = lib()
    L.ipxo_jpeg_encode_gray8.restype = C.c_int
    L.ipxo_jpeg_encode_gray8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_size_t)]
    L.ipxo_free.argtypes = [C.c_void_p]
    g = np.ascontiguousarray(plane, dtype=np.uint8)
    assert g.ndim == 2
    h, w = g.shape
    p, n = C.c_void_p(), C.c_size_t()
    rc = L.ipxo_jpeg_encode_gray8(g.ctypes.data, w, h, g.strides[0], int(quality), C.byref(p), C.byref(n))
    if rc:
        raise ValueError("jpeg: image is too large to encode" if rc == -1 else "out of memory")
    return _take_bytes(p, n)


# ---- image/jpeg decoder, baseline (oracle/ipx_jpeg_dec_oracle.c) --------------------------------------------
class Decoded(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("ratio", C.c_int), ("ystride", C.c_int), ("cstride", C.c_int),
                ("yrows", C.c_int), ("crows", C.c_int), ("y", C.c_void_p), ("cb", C.c_void_p), ("cr", C.c_void_p), ("dc_wide", C.c_int)]


def jpeg_decode_full(data):
    """image.Decode with Go's whole marker loop (ipxo_jpeg_decode_full): progressive and multi-scan files, garbage between segments
    skipped, EOI required.  Same result dict as jpeg_decode (no coefficients)."""
    L = lib()
    L.ipxo_jpeg_decode_full.restype = C.c_int
    L.ipxo_jpeg_decode_full.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Decoded)]
    L.ipxo_decoded_free.argtypes = [C.POINTER(Decoded)]
    d = Decoded()
    rc = L.ipxo_jpeg_decode_full(bytes(data), len(data), C.byref(d))
    if rc:
        raise ValueError({-1: "malformed", -2: "unsupported"}.get(rc, "out of memory"))

    def plane(p, rows, stride):
        return np.frombuffer(C.string_at(p, rows * stride), np.uint8).reshape(rows, stride).copy()
    out = {"w": d.w, "h": d.h, "ratio": d.ratio, "y": plane(d.y, d.yrows, d.ystride), "cb": plane(d.cb, d.crows, d.cstride),
           "cr": plane(d.cr, d.crows, d.cstride), "dc_wide": bool(d.dc_wide)}
    L.ipxo_decoded_free(C.byref(d))
    return out


def jpeg_decode(data, want_coefs=False):
    """image.Decode of a JPEG -> dict(w, h, ratio, y, cb, cr) with the MCU-padded planes of image.NewYCbCr (y: yrows x ystride,
    cb / cr: crows x cstride).  Raises ValueError("malformed" | "unsupported").  Without want_coefs this is Go's whole decoder
    (jpeg_decode_full); with it, the single-scan baseline restatement that also returns the quantised coefficients."""
    if not want_coefs:
        return jpeg_decode_full(data)
    L = lib()
    L.ipxo_jpeg_decode.restype = C.c_int
    L.ipxo_jpeg_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Decoded), C.c_void_p, C.c_size_t]
    L.ipxo_decoded_free.argtypes = [C.POINTER(Decoded)]
    d = Decoded()
    coefs = np.zeros(max(64, len(data) * 64), np.int16) if want_coefs else None
    rc = L.ipxo_jpeg_decode(bytes(data), len(data), C.byref(d), coefs.ctypes.data if want_coefs else None,
                            coefs.size if want_coefs else 0)
    if rc:
        raise ValueError({-1: "malformed", -2: "unsupported"}.get(rc, "out of memory"))
    def plane(p, rows, stride):
        return np.frombuffer(C.string_at(p, rows * stride), np.uint8).reshape(rows, stride).copy()
    out = {"w": d.w, "h": d.h, "ratio": d.ratio, "y": plane(d.y, d.yrows, d.ystride), "cb": plane(d.cb, d.crows, d.cstride),
           "cr": plane(d.cr, d.crows, d.cstride), "dc_wide": bool(d.dc_wide)}
    L.ipxo_decoded_free(C.byref(d))
    if want_coefs:
        out["coefs"] = coefs
    return out
