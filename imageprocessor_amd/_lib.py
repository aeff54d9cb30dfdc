"""ctypes declarations for libipx.so (include/ipx.h).  Fails loudly when the HIP extension is
missing: this package has no CPU fallback for the pixel path."""
import ctypes as C
import importlib.util
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IPX_LIB") or os.path.join(HERE, "libipx.so")  # IPX_LIB: an experimental build


class Rect(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32)]


class Glyph(C.Structure):
    _fields_ = [("mask", C.c_void_p), ("mw", C.c_int32), ("mh", C.c_int32), ("mstride", C.c_int32),
                ("dr", Rect), ("mpx", C.c_int32), ("mpy", C.c_int32)]


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("lanes", C.c_int32), ("lane_bytes", C.c_size_t)]


class PlanParams(C.Structure):
    _fields_ = [("sw", C.c_int32), ("sh", C.c_int32),
                ("do_resize", C.c_int32), ("resize_w", C.c_int32), ("resize_h", C.c_int32),
                ("keep_aspect", C.c_int32),
                ("do_thumbnail", C.c_int32), ("thumb_size", C.c_int32), ("crop_to_fit", C.c_int32),
                ("do_watermark", C.c_int32), ("glyphs", C.c_void_p)]


class PlanInfo(C.Structure):
    _fields_ = [("resize_w", C.c_int32), ("resize_h", C.c_int32),
                ("thumb_w", C.c_int32), ("thumb_h", C.c_int32), ("thumb_crop", Rect),
                ("wm_w", C.c_int32), ("wm_h", C.c_int32),
                ("resize_bytes", C.c_size_t), ("thumb_bytes", C.c_size_t), ("wm_bytes", C.c_size_t),
                ("algorithmic_bytes", C.c_size_t)]


class YCbCr(C.Structure):
    _fields_ = [("y", C.c_void_p), ("cb", C.c_void_p), ("cr", C.c_void_p), ("ystride", C.c_int32),
                ("cstride", C.c_int32), ("w", C.c_int32), ("h", C.c_int32), ("ratio", C.c_int32)]


class YCbCrBatch(C.Structure):
    _fields_ = [("y", C.c_void_p), ("cb", C.c_void_p), ("cr", C.c_void_p), ("ystride", C.c_int32),
                ("cstride", C.c_int32), ("y_frame_stride", C.c_size_t), ("c_frame_stride", C.c_size_t),
                ("ratio", C.c_int32)]


class Param(C.Structure):
    _fields_ = [("key", C.c_char_p), ("type", C.c_int32), ("f64", C.c_double), ("i64", C.c_int64),
                ("str", C.c_char_p)]


class Image(C.Structure):
    _fields_ = [("pix", C.c_void_p), ("w", C.c_int32), ("h", C.c_int32), ("stride", C.c_int32)]


MEASURE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_double, C.POINTER(C.c_int))
GLYPHS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                        C.POINTER(C.POINTER(Glyph)), C.POINTER(C.c_int))
RELEASE_FN = C.CFUNCTYPE(None, C.c_void_p)


class Bytes(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", C.c_size_t)]


class TextRasterizer(C.Structure):
    _fields_ = [("user", C.c_void_p), ("measure", MEASURE_FN), ("glyphs", GLYPHS_FN), ("release", RELEASE_FN)]


class Operation(C.Structure):
    _fields_ = [("type", C.c_char_p), ("params", C.POINTER(Param)), ("nparams", C.c_int32)]


class Task(C.Structure):
    _fields_ = [("id", C.c_char_p), ("image_id", C.c_char_p), ("ops", C.POINTER(Operation)),
                ("nops", C.c_int32), ("format", C.c_char_p)]


class Processed(C.Structure):
    _fields_ = [("operation", C.c_char * 16), ("path", C.c_char * 320), ("content_type", C.c_char * 32),
                ("format", C.c_char * 8), ("image", Image)]


class PoolConfig(C.Structure):
    _fields_ = [("lanes_per_device", C.c_int32), ("lane_bytes", C.c_size_t)]


class PoolOps(C.Structure):
    _fields_ = [("sw", C.c_int32), ("sh", C.c_int32),
                ("do_resize", C.c_int32), ("resize_w", C.c_int32), ("resize_h", C.c_int32), ("keep_aspect", C.c_int32),
                ("do_thumbnail", C.c_int32), ("thumb_size", C.c_int32), ("crop_to_fit", C.c_int32),
                ("do_watermark", C.c_int32), ("glyphs", C.POINTER(Glyph)), ("n_glyphs", C.c_int32), ("col", C.c_uint8 * 4)]


class Job(C.Structure):
    _fields_ = [("kind", C.c_int32), ("ops", PoolOps), ("n", C.c_int32),
                ("src", C.c_void_p), ("sstride", C.c_int32), ("src_frame_stride", C.c_size_t),
                ("resize_out", C.c_void_p), ("resize_frame_stride", C.c_size_t),
                ("thumb_out", C.c_void_p), ("thumb_frame_stride", C.c_size_t),
                ("wm_out", C.c_void_p), ("wm_frame_stride", C.c_size_t),
                ("files", C.POINTER(Bytes)), ("quality", C.c_int32),
                ("resize_jpeg", C.POINTER(Bytes)), ("thumb_jpeg", C.POINTER(Bytes)), ("wm_jpeg", C.POINTER(Bytes)),
                ("status", C.POINTER(C.c_int32))]


class BatcherConfig(C.Structure):
    _fields_ = [("max_batch", C.c_int32), ("max_wait_us", C.c_int32), ("quality", C.c_int32)]


class BatchResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("resize", Bytes), ("thumb", Bytes), ("wm", Bytes)]


class BatcherStats(C.Structure):
    _fields_ = [(k, C.c_longlong) for k in ("files", "batches", "flushed_by_size", "flushed_by_timer", "largest_batch", "pending_files", "flushed_when_idle")]


_P = C.c_void_p
_I = C.c_int
_Z = C.c_size_t

# name -> (restype, argtypes); also the list of symbols the ABI test checks against the header
SIGNATURES = {
    "ipx_create": (_I, [C.POINTER(Config), C.POINTER(_P)]),
    "ipx_destroy": (None, [_P]),
    "ipx_last_error": (C.c_char_p, []),
    "ipx_abi_version": (_I, []),
    "ipx_device_count": (_I, []),
    "ipx_frame_supported": (_I, [_I, _I, C.c_longlong, _I]),
    "ipx_resize_dims": (_I, [_I, _I, _I, _I, _I, C.POINTER(_I), C.POINTER(_I)]),
    "ipx_thumb_geometry": (_I, [_I, _I, _I, _I, C.POINTER(Rect), C.POINTER(_I), C.POINTER(_I)]),
    "ipx_text_height_px": (_I, [C.c_double]),
    "ipx_watermark_anchor": (_I, [C.c_char_p, _I, _I, _I, _I, C.POINTER(_I), C.POINTER(_I)]),
    "ipx_parse_color": (_I, [C.c_char_p, C.c_double, _P]),
    "ipx_host_alloc": (_P, [_P, _Z]),
    "ipx_host_free": (_I, [_P, _P]),
    "ipx_dev_alloc": (_P, [_P, _Z]),
    "ipx_dev_free": (_I, [_P, _P]),
    "ipx_memcpy_h2d": (_I, [_P, _P, _P, _Z]),
    "ipx_memcpy_d2h": (_I, [_P, _P, _P, _Z]),
    "ipx_memcpy_d2d": (_I, [_P, _P, _P, _Z]),
    "ipx_stream_copy": (_I, [_P, _P, _P, _P, _Z]),
    "ipx_device_sync": (_I, [_P]),
    "ipx_stream_sync": (_I, [_P, _P]),
    "ipx_stream_create": (_P, [_P]),
    "ipx_stream_destroy": (_I, [_P, _P]),
    "ipx_scale_bilinear_rgba8": (_I, [_P, _P, _I, _I, _I, Rect, _P, _I, _I, _I, Rect, _I]),
    "ipx_draw_rgba8": (_I, [_P, _P, _I, _I, _I, Rect, _P, _I, _I, _I, _I, _I, _I]),
    "ipx_composite_glyphs_rgba8": (_I, [_P, _P, _I, _I, _I, C.POINTER(Glyph), _I, _P]),
    "ipx_scale_bilinear_nrgba8": (_I, [_P, _P, _I, _I, _I, Rect, _P, _I, _I, _I, Rect, _I]),
    "ipx_draw_nrgba8": (_I, [_P, _P, _I, _I, _I, Rect, _P, _I, _I, _I, _I, _I, _I]),
    "ipx_scale_bilinear_deep": (_I, [_P, _P, _I, _I, _I, Rect, _P, _I, _I, _I, _I, Rect, _I]),
    "ipx_draw_deep": (_I, [_P, _P, _I, _I, _I, Rect, _P, _I, _I, _I, _I, _I, _I, _I]),
    "ipx_scale_bilinear_ycbcr": (_I, [_P, _P, _I, _I, _I, Rect, C.POINTER(YCbCr), Rect]),
    "ipx_draw_ycbcr": (_I, [_P, _P, _I, _I, _I, Rect, C.POINTER(YCbCr), _I, _I]),
    "ipx_dev_scale_bilinear_rgba8": (_I, [_P, _P, _P, _I, _I, _I, Rect, _P, _I, _I, _I, Rect, _I]),
    "ipx_dev_draw_rgba8": (_I, [_P, _P, _P, _I, _I, _I, Rect, _P, _I, _I, _I, _I, _I, _I]),
    "ipx_glyphset_create": (_I, [_P, C.POINTER(Glyph), _I, _P, C.POINTER(_P)]),
    "ipx_glyphset_destroy": (None, [_P, _P]),
    "ipx_dev_composite_glyphs_rgba8": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "ipx_plan_create": (_I, [_P, C.POINTER(PlanParams), C.POINTER(_P)]),
    "ipx_plan_destroy": (None, [_P, _P]),
    "ipx_plan_query": (_I, [_P, C.POINTER(PlanInfo)]),
    "ipx_plan_run_dev": (_I, [_P, _P, _P, _I, _P, _I, _Z, _P, _Z, _P, _Z, _P, _Z]),
    "ipx_plan_run_host": (_I, [_P, _P, _I, _P, _I, _Z, _P, _Z, _P, _Z, _P, _Z]),
    "ipx_plan_run_dev_ycbcr": (_I, [_P, _P, _P, _I, C.POINTER(YCbCrBatch), _P, _Z, _P, _Z, _P, _Z]),
    "ipx_plan_run_dev_gray": (_I, [_P, _P, _P, _I, _P, _I, _Z, _P, _Z, _P, _Z, _P, _Z]),
    "ipx_plan_run_dev_paletted": (_I, [_P, _P, _P, _I, _P, _I, _Z, _P, _P, _Z, _P, _Z, _P, _Z]),
    "ipx_plan_run_host_nrgba": (_I, [_P, _P, _I, _P, _I, _Z, _P, _Z, _P, _Z, _P, _Z]),
    "ipx_plan_run_host_gray": (_I, [_P, _P, _I, _P, _I, _Z, _P, _Z, _P, _Z, _P, _Z]),
    "ipx_plan_run_host_paletted": (_I, [_P, _P, _I, _P, _I, _Z, _P, _P, _Z, _P, _Z, _P, _Z]),
    "ipx_plan_run_dev_nrgba": (_I, [_P, _P, _P, _I, _P, _I, _Z, _P, _Z, _P, _Z, _P, _Z]),
    "ipx_plan_run_dev_deep": (_I, [_P, _P, _P, _I, _I, _P, _I, _Z, _P, _Z, _P, _Z, _P, _Z]),
    "ipx_plan_run_host_deep": (_I, [_P, _P, _I, _I, _P, _I, _Z, _P, _Z, _P, _Z, _P, _Z]),
    "ipx_plan_run_host_ycbcr": (_I, [_P, _P, _I, C.POINTER(YCbCrBatch), _P, _Z, _P, _Z, _P, _Z]),
    "ipx_event_create": (_P, [_P]),
    "ipx_event_record": (_I, [_P, _P, _P]),
    "ipx_event_elapsed_ms": (_I, [_P, _P, _P, C.POINTER(C.c_float)]),
    "ipx_event_destroy": (None, [_P, _P]),
    "ipx_image_free": (None, [C.POINTER(Image)]),
    "ipx_resizer_process": (_I, [_P, C.POINTER(Image), C.c_char_p, C.POINTER(Param), _I, C.POINTER(Image), _P]),
    "ipx_thumbnailer_process": (_I, [_P, C.POINTER(Image), C.c_char_p, C.POINTER(Param), _I, C.POINTER(Image), _P]),
    "ipx_watermarker_process": (_I, [_P, C.POINTER(Image), C.c_char_p, C.POINTER(Param), _I,
                                     C.POINTER(TextRasterizer), C.POINTER(Image), _P]),
    "ipx_processor_process": (_I, [_P, C.POINTER(Task), C.POINTER(Image), C.c_char_p, C.POINTER(TextRasterizer),
                                   C.POINTER(Processed), C.POINTER(_I)]),
    "ipx_jpeg_coef_count": (_Z, [_I, _I]),
    "ipx_jpeg_quant_tables": (_I, [_I, _P]),
    "ipx_dev_jpeg_fdct_rgba8": (_I, [_P, _P, _P, _I, _I, _I, _Z, _I, _I, _P]),
    "ipx_jpeg_entropy_encode": (_I, [_P, _I, _I, _I, C.POINTER(_P), C.POINTER(_Z)]),
    "ipx_jpeg_encode_rgba8": (_I, [_P, _P, _I, _I, _I, _I, C.POINTER(_P), C.POINTER(_Z)]),
    "ipx_jpeg_encode_batch_dev": (_I, [_P, _P, _I, _I, _I, _Z, _I, _I, C.POINTER(_P), C.POINTER(_Z), C.POINTER(_Z)]),
    "ipx_buffer_free": (None, [_P]),
    "ipx_jpeg_decode_batch": (_I, [_P, _P, C.POINTER(Bytes), _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(YCbCrBatch), C.POINTER(_I),
                                   C.POINTER(_P)]),
    "ipx_jpeg_planes_free": (None, [_P, _P]),
    "ipx_plan_run_jpeg_jpeg": (_I, [_P, _P, _I, C.POINTER(Bytes), _I, C.POINTER(Bytes), C.POINTER(Bytes), C.POINTER(Bytes), C.POINTER(_I),
                                    C.POINTER(_P)]),
    "ipx_plan_run_host_jpeg": (_I, [_P, _P, _I, _P, _I, _Z, _I, C.POINTER(Bytes), C.POINTER(Bytes), C.POINTER(Bytes), C.POINTER(_P)]),
    "ipx_jpeg_result_free": (None, [_P, _P]),
    "ipx_plan_run_host_ycbcr_jpeg": (_I, [_P, _P, _I, C.POINTER(YCbCrBatch), _I, C.POINTER(Bytes), C.POINTER(Bytes), C.POINTER(Bytes),
                                          C.POINTER(_P)]),
    "ipx_font_create": (_I, [_P, _Z, C.POINTER(_P)]),
    "ipx_font_destroy": (None, [_P]),
    "ipx_font_glyph_index": (_I, [_P, C.c_uint32]),
    "ipx_font_glyph_advance": (_I, [_P, C.c_uint32, C.c_double, C.POINTER(C.c_int32)]),
    "ipx_font_kern": (_I, [_P, C.c_uint32, C.c_uint32, C.c_double, C.POINTER(C.c_int32)]),
    "ipx_font_text_width": (_I, [_P, C.c_char_p, C.c_double, C.POINTER(C.c_int32), C.POINTER(_I)]),
    "ipx_font_draw_string": (_I, [_P, C.c_char_p, C.c_double, _I, _I, _I, _I, C.POINTER(C.POINTER(Glyph)),
                                  C.POINTER(_I), C.POINTER(C.c_int32)]),
    "ipx_font_release_thread": (None, []),
    "ipx_font_rasterizer": (_I, [_P, C.POINTER(TextRasterizer)]),
    "ipx_pool_create": (_I, [C.POINTER(_I), _I, C.POINTER(PoolConfig), C.POINTER(_P)]),
    "ipx_pool_destroy": (None, [_P]),
    "ipx_pool_slots": (_I, [_P]),
    "ipx_pool_frames_done": (C.c_longlong, [_P, _I]),
    "ipx_pool_host_alloc": (_P, [_P, _I, _Z]),
    "ipx_pool_host_free": (_I, [_P, _I, _P]),
    "ipx_job_submit": (_I, [_P, C.POINTER(Job), C.POINTER(C.c_uint64)]),
    "ipx_job_poll": (_I, [_P, C.c_uint64, C.POINTER(_I)]),
    "ipx_job_wait": (_I, [_P, C.c_uint64, C.POINTER(_I)]),
    "ipx_job_release": (_I, [_P, C.c_uint64]),
    "ipx_pool_run_host": (_I, [_P, C.POINTER(Job), _I]),
    "ipx_plan_acquire": (_I, [_P, C.POINTER(PoolOps), C.POINTER(_P), C.POINTER(_I)]),
    "ipx_plan_release": (None, [_P, _P, _I]),
    "ipx_link_probe": (_I, [_P, _Z, _Z, _I, C.POINTER(C.c_double * 4)]),
    "ipx_batcher_create": (_I, [_P, C.POINTER(BatcherConfig), C.POINTER(_P)]),
    "ipx_batcher_destroy": (None, [_P]),
    "ipx_batcher_submit": (_I, [_P, C.POINTER(Bytes), C.POINTER(PoolOps), C.POINTER(C.c_uint64)]),
    "ipx_batcher_wait": (_I, [_P, C.c_uint64, C.POINTER(BatchResult)]),
    "ipx_batcher_release": (_I, [_P, C.c_uint64]),
    "ipx_batcher_get_stats": (_I, [_P, C.POINTER(BatcherStats)]),
}

_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so (soname
    libamdhip64.so.7, the same as /opt/rocm's).  If it is loaded first, libipx's NEEDED entry and a
    later `import torch` both resolve to that one copy; loading them in the other order would put
    two runtimes in the process.  IPX_SYSTEM_HIP=1 skips this (no torch in the process)."""
    if os.environ.get("IPX_SYSTEM_HIP") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load():
    """Loads libipx.so; raises if it has not been built (python -m imageprocessor_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "imageprocessor_amd: %s is missing -- build the HIP extension first "
            "(python imageprocessor_amd/build.py, or __graft_entry__.build()). "
            "There is no CPU fallback for the pixel path." % LIB_PATH)
    _preload_hip_runtime()
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L
