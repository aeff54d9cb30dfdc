/*
 * ipx.h -- C ABI of libipx, the MI355X (gfx950) pixel worker for ImageProcessor.
 *
 * This is the drop-in boundary for ONE path of the reference: the per-pixel work of
 * internal/worker -> internal/usecase/processor.  The reference has no
 * FFI today (dockerfile:12 builds with CGO_ENABLED=0); the seam is three private Go helpers
 * plus the operator method set.  Every entry point below names the reference code it replaces.
 * Plain pointers and sizes only; all functions are callable from any OS thread (goroutines
 * migrate), return an ipx_status (0 = ok, negative = error) and never abort or throw.
 * INTEGRATION.md holds the cgo binding that goes with this header.
 *
 * Pixel format everywhere: 8-bit premultiplied RGBA, as Go's image.RGBA (Pix, Stride, Rect
 * with Min = (0,0)).  Strides are in bytes.
 */
#ifndef IPX_H
#define IPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IPX_ABI_VERSION 1

typedef struct ipx_ctx ipx_ctx;           /* one per worker process and GPU            */
typedef struct ipx_glyphset ipx_glyphset; /* rasterised watermark text resident in HBM */
typedef struct ipx_plan ipx_plan;         /* fused resize+thumbnail+watermark geometry */

typedef enum {
    IPX_OK = 0,
    IPX_ERR_INVALID = -1,     /* bad argument (the Go side returns an error, image_processor.go:66-75) */
    IPX_ERR_NOMEM = -2,       /* host or device allocation failed                       */
    IPX_ERR_HIP = -3,         /* a HIP call failed; text in ipx_last_error()            */
    IPX_ERR_UNSUPPORTED = -4, /* valid in the reference but outside this library's path */
    IPX_ERR_NODEVICE = -5     /* no gfx950 device visible                               */
} ipx_status;

/* draw.Op of image/draw and x/image/draw (resize.go:123 passes xdraw.Over, watermark.go:92 draw.Src) */
enum { IPX_OP_OVER = 0, IPX_OP_SRC = 1 };

/* image.Rectangle: [x0,x1) x [y0,y1) */
typedef struct { int32_t x0, y0, x1, y1; } ipx_rect;

/* One draw.DrawMask call of freetype.Context.DrawString (watermark.go:151): an *image.Alpha
 * mask with bounds (0,0)-(mw,mh), the destination rectangle dr and the mask point mp that is
 * aligned with dr.Min.  Clipping against the frame and the mask is done by the library exactly
 * as image/draw does.  Glyphs are applied in array order; overlapping boxes are NOT merged. */
typedef struct {
    const uint8_t *mask;
    int32_t mw, mh, mstride;
    ipx_rect dr;
    int32_t mpx, mpy;
} ipx_glyph;

typedef struct {
    int32_t device;        /* HIP device ordinal; -1 = LOCAL_RANK / 0                          */
    int32_t lanes;         /* staging lanes (a stream + device scratch each); 0 = 5: a host batch pipelines its chunks over four
                              (the reference's WORKER_CONCURRENCY is 3, .env.example:38, worker.go:90; a compressed-in /
                              compressed-out batch measured best in four parts) and one stays free, so that a single-frame call is
                              served while a batch runs */
    size_t lane_bytes;     /* initial pinned+device staging per lane; grows on demand; 0 = 64 MiB */
} ipx_config;

/* ---- lifetime ----------------------------------------------------------------------------- */

/* Replaces processor.NewImageProcessor (image_processor.go:29) as the owner of per-process state. */
int ipx_create(const ipx_config *cfg, ipx_ctx **out);
void ipx_destroy(ipx_ctx *ctx);
/* Thread-local text of the last failure on the calling thread ("" if none). */
const char *ipx_last_error(void);
int ipx_abi_version(void);
/* Number of gfx950 devices visible; a negative ipx_status on failure. */
int ipx_device_count(void);

/* Whether a frame of w x h pixels with this row stride can go to the GPU path at all: the kernels address a frame through 32-bit
 * buffer descriptors, so (h-1)*stride + w*bytes_per_pixel must stay below 2 GiB and each side below 65536.  IPX_OK, or
 * IPX_ERR_UNSUPPORTED -- the answer every plan / per-operation entry gives for such a frame; the worker keeps its CPU path for it
 * (a 32 MiB PNG upload, domain/task.go:55, can decode past 23170 x 23170).  Host only. */
int ipx_frame_supported(int w, int h, long long stride, int bytes_per_pixel);

/* ---- geometry and parameter rules (host only, no GPU needed) ------------------------------- */

/* resize.go:61-75: aspect-fit in float64 with truncation, or (w,h) as given. */
int ipx_resize_dims(int ow, int oh, int w, int h, int keep_aspect, int *nw, int *nh);
/* thumbnail.go:48-65 (short side = size) and :114-127 (centre square crop). */
int ipx_thumb_geometry(int ow, int oh, int size, int crop_to_fit, ipx_rect *crop, int *nw, int *nh);
/* watermark.go:116-118: int(fixed.Int26_6(fontSize*64*1.2).Ceil()). */
int ipx_text_height_px(double font_size);
/* watermark.go:121-148: baseline point (whole pixels, as freetype.Pt) for a position string;
 * unknown strings fall to bottom-right like the reference's default arm. */
int ipx_watermark_anchor(const char *position, int w, int h, int width_px, int height_px,
                         int *px, int *py);
/* watermark.go:159-190 + :93-97: "r,g,b[,a]" -> color.RGBA bytes (NOT premultiplied, as the
 * reference builds it).  Returns IPX_OK, or 1 when the string is malformed, in which case rgba
 * holds the reference's fallback (black with alpha uint8(255*opacity)). */
int ipx_parse_color(const char *s, double opacity, uint8_t rgba[4]);

/* ---- memory the Go side may hand to asynchronous calls --------------------------------------- */

/* hipHostMalloc'd staging (cgo: wrap with unsafe.Slice; C owns it, Go must not retain it past free). */
void *ipx_host_alloc(ipx_ctx *ctx, size_t bytes);
int ipx_host_free(ipx_ctx *ctx, void *p);
void *ipx_dev_alloc(ipx_ctx *ctx, size_t bytes);
int ipx_dev_free(ipx_ctx *ctx, void *p);
/* The three copies are complete when they return (the device-to-device one runs on the context's stream and waits for it). */
int ipx_memcpy_h2d(ipx_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int ipx_memcpy_d2h(ipx_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int ipx_memcpy_d2d(ipx_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes);
/* A plain streaming copy (every workgroup copies a contiguous region of its own, 16 bytes per lane; dst, src and bytes multiples of 16),
 * asynchronous on `stream`.  Not
 * part of the path: bench.py times it on the box it runs on, because the streaming ceiling the band kernels are held against differs
 * from box to box and with where buffers land. */
int ipx_stream_copy(ipx_ctx *ctx, void *stream, void *dst_dev, const void *src_dev, size_t bytes);
/* What the host link gives pinned memory on this box: out_gbps = {up alone, down alone, up while down runs, down while up runs}, best of
 * `reps` transfers of up_bytes / down_bytes (down_bytes a multiple of 16).  Copies go in 32 MiB pieces on one stream per direction; the
 * both-at-once pair is the better of two ways down -- a copy engine, or a kernel storing into the pinned block (how the host entries
 * deliver their outputs).  Not part of the path: bench.py holds its PCIe-inclusive legs against it (`frac_of_link`). */
int ipx_link_probe(ipx_ctx *ctx, size_t up_bytes, size_t down_bytes, int reps, double out_gbps[4]);
/* Blocks until the device is idle (every stream). */
int ipx_device_sync(ipx_ctx *ctx);
/* Blocks until everything queued on `stream` (a hipStream_t, NULL = the context's stream) is done. */
int ipx_stream_sync(ipx_ctx *ctx, void *stream);

/* A stream of the caller's own on the context's device (a hipStream_t, non-blocking) for the asynchronous device-pointer entries:
 * two of them let a worker queue the next batch while the previous one runs (bench.py --mixed does exactly that).  Destroying a
 * stream waits for what is queued on it. */
void *ipx_stream_create(ipx_ctx *ctx);
int ipx_stream_destroy(ipx_ctx *ctx, void *stream);

/* ---- per-operation seam, host pointers, synchronous ------------------------------------------
 * Each call stages through a pinned lane, runs the HIP kernel and copies the result back. */

/* xdraw.BiLinear.Scale(dst, dr, src, sr, op, nil) for *image.RGBA <- *image.RGBA -- BiLinear is x/image's tent Kernel run through its
 * two-pass float64 kernel scaler (newDistrib, scaleX_RGBA, scaleY_RGBA_{Src,Over}), NOT the 2x2-tap ApproxBiLinear:
 * resizeImage (resize.go:121-125) and both Scale calls of cropAndResize (thumbnail.go:128-131; equal sizes are not simplified to a Copy).
 * dst is read as well as written when op = IPX_OP_OVER.  sr must lie inside the source
 * (IPX_ERR_UNSUPPORTED otherwise: the reference would leave its typed fast path). */
int ipx_scale_bilinear_rgba8(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride, ipx_rect dr,
                             const uint8_t *src, int sw, int sh, int sstride, ipx_rect sr, int op);
/* draw.Draw / draw.DrawMask with a nil mask, *image.RGBA <- *image.RGBA: the full-frame copy of
 * addTextWatermark (watermark.go:90-92). */
int ipx_draw_rgba8(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride, ipx_rect r,
                   const uint8_t *src, int sw, int sh, int sstride, int spx, int spy, int op);
/* freetype.Context.DrawString's compositing (watermark.go:151): draw.DrawMask(dst, dr,
 * image.Uniform{col}, ZP, mask, mp, draw.Over) per glyph, in order, in place on dst.
 * col is the color.RGBA of watermark.go:93-97 (parseColor). */
int ipx_composite_glyphs_rgba8(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride,
                               const ipx_glyph *glyphs, int n, const uint8_t col[4]);

/* ---- source-type variants of the same helpers ---------------------------------
 * image.Decode hands the reference *image.NRGBA for PNGs with alpha and *image.YCbCr for JPEGs
 * (image_processor.go:47); resizeImage / cropAndResize / draw.Draw take them as they are.  These
 * entries do the same on the GPU: per-tap conversion inside the kernel scaler exactly as x/image/draw
 * does it (scaleX_NRGBA, scaleX_YCbCr4xx), and image/draw's drawNRGBA{Src,Over} /
 * imageutil.DrawYCbCr for copies (the watermark's draw.Draw). */
int ipx_scale_bilinear_nrgba8(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride, ipx_rect dr,
                              const uint8_t *src, int sw, int sh, int sstride, ipx_rect sr, int op);
int ipx_draw_nrgba8(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride, ipx_rect r,
                    const uint8_t *src, int sw, int sh, int sstride, int spx, int spy, int op);

/* *image.YCbCr with Rect.Min = (0,0); ratio numbered as image.YCbCrSubsampleRatio */
enum { IPX_YCBCR_444 = 0, IPX_YCBCR_422 = 1, IPX_YCBCR_420 = 2, IPX_YCBCR_440 = 3,
       IPX_GRAY = 4 /* *image.Gray: only the y plane is set (one-component JPEGs) */ };
typedef struct {
    const uint8_t *y, *cb, *cr;
    int32_t ystride, cstride, w, h, ratio;
} ipx_ycbcr;
/* A YCbCr image is opaque, so the reference's Over becomes Src: there is no op argument. */
int ipx_scale_bilinear_ycbcr(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride, ipx_rect dr,
                             const ipx_ycbcr *src, ipx_rect sr);
int ipx_draw_ycbcr(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride, ipx_rect r,
                   const ipx_ycbcr *src, int spx, int spy);

/* ---- same operations on frames already resident in HBM, asynchronous on `stream` -------------
 * `stream` is a hipStream_t (NULL = the context's own stream).  Pointers are device pointers. */

int ipx_dev_scale_bilinear_rgba8(ipx_ctx *ctx, void *stream, uint8_t *dst, int dw, int dh,
                                 int dstride, ipx_rect dr, const uint8_t *src, int sw, int sh,
                                 int sstride, ipx_rect sr, int op);
int ipx_dev_draw_rgba8(ipx_ctx *ctx, void *stream, uint8_t *dst, int dw, int dh, int dstride,
                       ipx_rect r, const uint8_t *src, int sw, int sh, int sstride, int spx,
                       int spy, int op);
/* Uploads the glyph masks once (host pointers in `glyphs`); clipping happens per frame size. */
int ipx_glyphset_create(ipx_ctx *ctx, const ipx_glyph *glyphs, int n, const uint8_t col[4],
                        ipx_glyphset **out);
void ipx_glyphset_destroy(ipx_ctx *ctx, ipx_glyphset *gs);
int ipx_dev_composite_glyphs_rgba8(ipx_ctx *ctx, void *stream, uint8_t *dst, int dw, int dh,
                                   int dstride, const ipx_glyphset *gs);

/* ---- the batched worker path -------------------------------------------------------------------
 * Replaces (*ImageProcessor).Process (image_processor.go:39-102) between image.Decode (:47) and
 * the encoders (resize.go:78-91, thumbnail.go:68-81, watermark.go:66-79) for a batch of decoded
 * frames of one size: every operator is applied to the ORIGINAL frame (image_processor.go:64-65),
 * so one pass over each source frame produces all requested outputs. */

typedef struct {
    int32_t sw, sh;                 /* frame size                                          */
    int32_t do_resize;              /* operator present in task.Operations (domain/task.go) */
    int32_t resize_w, resize_h, keep_aspect; /* resize.go:26-59                             */
    int32_t do_thumbnail;
    int32_t thumb_size, crop_to_fit;         /* thumbnail.go:25-47; size 0 = 200 (task.go:56) */
    int32_t do_watermark;
    const ipx_glyphset *glyphs;     /* rasterised text; may be NULL (copy only)             */
} ipx_plan_params;

typedef struct {
    int32_t resize_w, resize_h;     /* actual output sizes after the aspect rules           */
    int32_t thumb_w, thumb_h;
    ipx_rect thumb_crop;
    int32_t wm_w, wm_h;
    size_t resize_bytes, thumb_bytes, wm_bytes; /* tightly packed bytes per frame             */
    size_t algorithmic_bytes;       /* source read once + every output written once, per frame */
} ipx_plan_info;

int ipx_plan_create(ipx_ctx *ctx, const ipx_plan_params *p, ipx_plan **out);
void ipx_plan_destroy(ipx_ctx *ctx, ipx_plan *plan);
int ipx_plan_query(const ipx_plan *plan, ipx_plan_info *info);

/* n frames resident in HBM at src + i*src_frame_stride (row stride sstride); outputs tightly
 * packed rows at out + i*out_frame_stride.  An output pointer may be NULL to skip it even when
 * the plan has the operator.  Asynchronous on `stream`. */
int ipx_plan_run_dev(ipx_ctx *ctx, void *stream, const ipx_plan *plan, int n, const uint8_t *src,
                     int sstride, size_t src_frame_stride, uint8_t *resize_out,
                     size_t resize_frame_stride, uint8_t *thumb_out, size_t thumb_frame_stride,
                     uint8_t *wm_out, size_t wm_frame_stride);

/* Host-to-host variant for the worker: frames and outputs in pinned memory from ipx_host_alloc
 * (or any host memory, slower); copies and kernels are pipelined over the context's lanes so
 * H2D, kernels and D2H of consecutive chunks overlap.  Synchronous. */
int ipx_plan_run_host(ipx_ctx *ctx, const ipx_plan *plan, int n, const uint8_t *src, int sstride,
                      size_t src_frame_stride, uint8_t *resize_out, size_t resize_frame_stride,
                      uint8_t *thumb_out, size_t thumb_frame_stride, uint8_t *wm_out,
                      size_t wm_frame_stride);

/* A batch of decoded JPEG frames (what image.Decode returns, image_processor.go:47) for the same plan:
 * planes of frame i at y + i*y_frame_stride and cb / cr + i*c_frame_stride.  Per operator the reference
 * converts differently (16-bit per tap inside resize; RGBA8 first for the crop thumbnail and the
 * watermark); the results are those of the reference's helpers on the *image.YCbCr itself.  The host
 * variant uploads 1.5 bytes per pixel for 4:2:0 instead of 4. */
typedef struct {
    const uint8_t *y, *cb, *cr;
    int32_t ystride, cstride;
    size_t y_frame_stride, c_frame_stride;
    int32_t ratio;                  /* IPX_YCBCR_* */
} ipx_ycbcr_batch;
int ipx_plan_run_dev_ycbcr(ipx_ctx *ctx, void *stream, const ipx_plan *plan, int n, const ipx_ycbcr_batch *src,
                           uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out,
                           size_t thumb_frame_stride, uint8_t *wm_out, size_t wm_frame_stride);
/* A batch of *image.NRGBA frames (PNGs with alpha): like the YCbCr batch, per operator as the reference's helpers treat the
 * type -- resize and the non-crop thumbnail weight 16-bit premultiplied taps (scaleX_NRGBA), the crop thumbnail scales its 8-bit
 * crop copy and the watermark premultiplies to RGBA8 (drawNRGBASrc).  One pass (ks_fused_kernel) for
 * 16-byte aligned frames whose width is a multiple of 4, three kernels otherwise; the same bytes either way. */
int ipx_plan_run_dev_nrgba(ipx_ctx *ctx, void *stream, const ipx_plan *plan, int n, const uint8_t *src, int sstride,
                           size_t src_frame_stride, uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out,
                           size_t thumb_frame_stride, uint8_t *wm_out, size_t wm_frame_stride);
/* A batch of *image.Gray frames (one-component JPEGs, grey PNGs): image/draw's drawGray and x/image's
 * scaleX_Gray both read a source pixel as (y, y, y, 0xff).  A YCbCr pixel with Cb = Cr = 128 converts to exactly that in both of
 * the reference's conversions, so the batch takes the planar pass with a constant chroma row (1 byte per pixel read); shapes that kernel
 * does not take are expanded to RGBA8 in HBM and take the RGBA pass.  The outputs are bit for bit the reference's either way. */
int ipx_plan_run_dev_gray(ipx_ctx *ctx, void *stream, const ipx_plan *plan, int n, const uint8_t *gray, int stride,
                          size_t frame_stride, uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out,
                          size_t thumb_frame_stride, uint8_t *wm_out, size_t wm_frame_stride);
/* A batch of *image.Paletted frames (what image.Decode returns for GIF uploads and palette PNGs, image_processor.go:47): one index
 * byte per pixel plus, per frame, 256 palette entries of 4 bytes (R, G, B, A), non-premultiplied, unused entries zero; palettes of
 * frame i at palettes + i*1024, in device memory, 4-byte aligned.  No routine of x/image or image/draw specialises on this type: the
 * generic ones (scaleX_Image, drawRGBA) read Palette[i].RGBA() per tap.  For every entry the GIF and PNG decoders produce --
 * opaque color.RGBA, the zero colour for a GIF's transparent index, color.NRGBA for a PNG's tRNS -- that is exactly the
 * premultiplication scaleX_NRGBA / drawNRGBA* apply to (R, G, B, A), so the frames are expanded to NRGBA8 in HBM and take
 * ipx_plan_run_dev_nrgba; outputs are bit for bit the generic routines' (tests/test_sources_gpu.py against an oracle of those). */
int ipx_plan_run_dev_paletted(ipx_ctx *ctx, void *stream, const ipx_plan *plan, int n, const uint8_t *index, int stride,
                              size_t frame_stride, const uint8_t *palettes, uint8_t *resize_out, size_t resize_frame_stride,
                              uint8_t *thumb_out, size_t thumb_frame_stride, uint8_t *wm_out, size_t wm_frame_stride);
/* The same three source types from host memory (frames Go's png / gif / jpeg decoders left there), chunked over the lanes like
 * ipx_plan_run_host; `palettes` is host memory here. */
int ipx_plan_run_host_nrgba(ipx_ctx *ctx, const ipx_plan *plan, int n, const uint8_t *src, int sstride, size_t src_frame_stride,
                            uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out, size_t thumb_frame_stride,
                            uint8_t *wm_out, size_t wm_frame_stride);
int ipx_plan_run_host_gray(ipx_ctx *ctx, const ipx_plan *plan, int n, const uint8_t *gray, int stride, size_t frame_stride,
                           uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out, size_t thumb_frame_stride,
                           uint8_t *wm_out, size_t wm_frame_stride);
int ipx_plan_run_host_paletted(ipx_ctx *ctx, const ipx_plan *plan, int n, const uint8_t *index, int stride, size_t frame_stride,
                               const uint8_t *palettes, uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out,
                               size_t thumb_frame_stride, uint8_t *wm_out, size_t wm_frame_stride);
/* The remaining image types image.Decode returns (image_processor.go:47) -- *image.NRGBA64, *image.RGBA64, *image.Gray16 from 16-bit
 * PNGs, *image.CMYK from four-component JPEGs ("deep" sources; SURVEY.md 8(f) N2 tail).  resizeImage (resize.go:121-125) takes them
 * through x/image's generic scaleX_Image, which reads every tap as src.At(x, y).RGBA() at full 16-bit precision; the
 * crop copy (thumbnail.go:128-130) and draw.Draw (watermark.go:92) go through image/draw's drawRGBA (drawCMYK for CMYK) and keep the
 * top byte of the same value.  `src` is Go's Pix for the type: big-endian 16-bit channels R G B A (8 bytes per pixel; NRGBA64 not
 * premultiplied, RGBA64 premultiplied), big-endian Y (2 bytes), or C M Y K bytes (4); rows `sstride` bytes apart, 2-byte aligned
 * (CMYK: 4).  One pass expands the frames to those 16-bit taps in HBM (8 bytes per pixel); the fused converted-tile kernel then reads
 * them as they are.  Bit for bit the generic routines' outputs (tests/test_deep_gpu.py against the oracle's restatement). */
enum { IPX_DEEP_NRGBA64 = 0, IPX_DEEP_RGBA64 = 1, IPX_DEEP_GRAY16 = 2, IPX_DEEP_CMYK = 3 };
/* the per-operation seam for these types (host pointers, one frame; like ipx_scale_bilinear_nrgba8 / ipx_draw_nrgba8): Scale with any
 * rectangles and either op -- Over turns into Src when (*NRGBA64).Opaque / (*RGBA64).Opaque holds, Gray16 and CMYK are opaque -- and
 * draw.Draw (drawRGBA; drawCMYK for CMYK) */
int ipx_scale_bilinear_deep(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride, ipx_rect dr, const uint8_t *src, int sw, int sh,
                            int sstride, int kind, ipx_rect sr, int op);
int ipx_draw_deep(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride, ipx_rect r, const uint8_t *src, int sw, int sh, int sstride,
                  int kind, int spx, int spy, int op);
int ipx_plan_run_dev_deep(ipx_ctx *ctx, void *stream, const ipx_plan *plan, int n, int kind, const uint8_t *src, int sstride,
                          size_t src_frame_stride, uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out,
                          size_t thumb_frame_stride, uint8_t *wm_out, size_t wm_frame_stride);
int ipx_plan_run_host_deep(ipx_ctx *ctx, const ipx_plan *plan, int n, int kind, const uint8_t *src, int sstride,
                           size_t src_frame_stride, uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out,
                           size_t thumb_frame_stride, uint8_t *wm_out, size_t wm_frame_stride);
int ipx_plan_run_host_ycbcr(ipx_ctx *ctx, const ipx_plan *plan, int n, const ipx_ycbcr_batch *src,
                            uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out,
                            size_t thumb_frame_stride, uint8_t *wm_out, size_t wm_frame_stride);

/* Average duration in milliseconds of the last `ipx_plan_run_dev` launches on `stream` is a
 * measurement concern of the caller: bracket calls with ipx_event_* below (HIP events on the
 * stream the kernels run on). */
void *ipx_event_create(ipx_ctx *ctx);
int ipx_event_record(ipx_ctx *ctx, void *event, void *stream);
int ipx_event_elapsed_ms(ipx_ctx *ctx, void *start, void *stop, float *ms); /* syncs on stop */
void ipx_event_destroy(ipx_ctx *ctx, void *event);

/* ---- operator seam: the reference's own method set on decoded frames ---------------------------
 * Resizer / Thumbnailer / Watermarker .Process(ctx, img, format, params) (resize.go:26,
 * thumbnail.go:25, watermark.go:40) and ImageProcessor.Process (image_processor.go:39) with the
 * codecs cut off: the input is the decoded frame (image_processor.go:47 stays on the Go side) and
 * each result is the *image.RGBA the reference would hand to its encoder, plus the format string
 * and object key it would use.  Parameter parsing, defaults and error texts follow the reference. */

/* one entry of Go's map[string]interface{}; the type tag reproduces the reference's type switches
 * (after the JSON round trip of a task every number is float64, domain/task.go:17-20) */
typedef enum {
    IPX_PT_FLOAT64 = 1, IPX_PT_INT = 2, IPX_PT_INT64 = 3, IPX_PT_INT32 = 4, IPX_PT_BOOL = 5, IPX_PT_STRING = 6
} ipx_param_type;
typedef struct {
    const char *key;
    int32_t type;      /* ipx_param_type */
    double f64;        /* FLOAT64 */
    int64_t i64;       /* INT / INT64 / INT32 / BOOL */
    const char *str;   /* STRING */
} ipx_param;

/* *image.RGBA with Rect.Min = (0,0).  Outputs are allocated by the library (ipx_image_free). */
typedef struct { uint8_t *pix; int32_t w, h, stride; } ipx_image;
void ipx_image_free(ipx_image *img);

/* The text rasteriser sits on the host side of the boundary (the reference uses golang/freetype
 * with the Go Regular face, watermark.go:29-38,98-118,151).  measure() returns
 * int(textWidth.Ceil()) of watermark.go:109-117; glyphs() returns, for the baseline point (px, py)
 * and a frame of w x h (c.SetClip(result.Bounds()), watermark.go:101), the DrawMask calls
 * DrawString would make (valid until release()).  A NULL rasteriser makes Watermarker.Process fail
 * with "font not loaded" like a nil font (watermark.go:87-89).  ipx_font_rasterizer() below fills
 * one in from a parsed TrueType font; a caller may also supply its own. */
typedef struct {
    void *user;
    int (*measure)(void *user, const char *text, double font_size, int *width_px);
    int (*glyphs)(void *user, const char *text, double font_size, int px, int py, int w, int h,
                  const ipx_glyph **out, int *n);
    void (*release)(void *user);
} ipx_text_rasterizer;

/* ---- glyph mask producer: truetype.Parse + freetype.Context.DrawString ------
 * A restatement of github.com/golang/freetype @ e2365dfdc4a0 (go.mod:42) without hinting, which is
 * all the reference uses: cmap formats 4 / 12, simple and compound glyphs, kern format 0, the
 * quadratic-spline cell rasteriser (even-odd), 4 horizontal sub-pixel positions.  Host code: the
 * masks are a few hundred bytes each and cached; the composite is the GPU's part. */
typedef struct ipx_font ipx_font;
/* truetype.Parse (watermark.go:31).  Failure = the reference's nil font; text in ipx_last_error(). */
int ipx_font_create(const uint8_t *ttf, size_t len, ipx_font **out);
void ipx_font_destroy(ipx_font *font);
/* (f *Font).Index */
int ipx_font_glyph_index(const ipx_font *font, uint32_t rune);
/* face.GlyphAdvance of truetype.NewFace(font, {Size, DPI: 72}) (watermark.go:105-111), 26.6 fixed */
int ipx_font_glyph_advance(const ipx_font *font, uint32_t rune, double font_size, int32_t *advance26_6);
/* (f *Font).Kern at the freetype.Context's scale (what DrawString adds between runes), 26.6 fixed */
int ipx_font_kern(const ipx_font *font, uint32_t rune0, uint32_t rune1, double font_size, int32_t *kern26_6);
/* watermark.go:108-117: textWidth = sum of advances (no kerning) and int(textWidth.Ceil()) */
int ipx_font_text_width(const ipx_font *font, const char *text, double font_size, int32_t *width26_6,
                        int *width_px);
/* c.DrawString(text, freetype.Pt(px, py)) with c.SetClip((0,0)-(clip_w,clip_h)) (watermark.go:98-104,151):
 * the DrawMask calls in rune order.  *out stays valid until the next call on the calling thread or
 * ipx_font_release_thread().  end_x26_6 (may be NULL) receives the returned point's X. */
int ipx_font_draw_string(ipx_font *font, const char *text, double font_size, int px, int py, int clip_w,
                         int clip_h, const ipx_glyph **out, int *n, int32_t *end_x26_6);
void ipx_font_release_thread(void);
/* Fills an ipx_text_rasterizer whose callbacks are the three entries above. */
int ipx_font_rasterizer(ipx_font *font, ipx_text_rasterizer *out);

/* out_format receives "jpeg" / "png" / "gif" as the reference's encoder switch would name it
 * (resize.go:78-91, thumbnail.go:68-81, watermark.go:66-79: a GIF watermark becomes JPEG). */
int ipx_resizer_process(ipx_ctx *ctx, const ipx_image *img, const char *format,
                        const ipx_param *params, int nparams, ipx_image *out, char out_format[8]);
int ipx_thumbnailer_process(ipx_ctx *ctx, const ipx_image *img, const char *format,
                            const ipx_param *params, int nparams, ipx_image *out, char out_format[8]);
int ipx_watermarker_process(ipx_ctx *ctx, const ipx_image *img, const char *format,
                            const ipx_param *params, int nparams, const ipx_text_rasterizer *font,
                            ipx_image *out, char out_format[8]);

/* domain.OperationParams / domain.ProcessingTask (domain/task.go:3-20) */
typedef struct { const char *type; const ipx_param *params; int32_t nparams; } ipx_operation;
typedef struct {
    const char *id, *image_id;
    const ipx_operation *ops; int32_t nops;
    const char *format;            /* task.Format; "" = the decoded format (image_processor.go:55-58) */
} ipx_task;
typedef struct {
    char operation[16];            /* key of ProcessingResult.ProcessedPaths                    */
    char path[320];                /* generatePath (image_processor.go:129-162)                 */
    char content_type[32];         /* getContentType (image_processor.go:164-182)               */
    char format[8];
    ipx_image image;
} ipx_processed;
/* (*ImageProcessor).Process (image_processor.go:39-102) from the decoded frame on: every operator
 * is applied to the ORIGINAL frame (:64-65), in one fused GPU pass when each type occurs at most
 * once.  `out` has room for task->nops entries; *n_out of them are filled.  On an operator failure
 * the status is negative, *n_out counts the operators that had succeeded before it, and
 * ipx_last_error() holds "operation <type> failed: ..." as the reference formats it (:66-75). */
int ipx_processor_process(ipx_ctx *ctx, const ipx_task *task, const ipx_image *decoded,
                          const char *decoded_format, const ipx_text_rasterizer *font,
                          ipx_processed *out, int *n_out);

/* ---- jpeg.Encode (encoder side) ------------------------------------------------
 * Every operator of the reference ends in jpeg.Encode(buf, img, &jpeg.Options{Quality: 85}) on its
 * *image.RGBA (resize.go:80, thumbnail.go:70, watermark.go:68,73,76).  Go's writer is restated on the GPU end to end: colour
 * conversion, 2x2 chroma box, jfdctint and the quantiser leave int16 coefficients (zig-zag order, 6 x 64 per 16x16 MCU in scan order
 * Y0 Y1 Y2 Y3 Cb Cr); the Huffman coder sizes every block while it still sits in LDS, places the bits, stuffs 0xff bytes and
 * prepends SOI / DQT / SOF0 / DHT / SOS in further kernels (ipx_jpeg_encode_batch_dev and the ipx_plan_run_*_jpeg entries), so only
 * finished streams cross the link.  ipx_jpeg_entropy_encode is the same coder on the host, for coefficients a caller has downloaded.
 * The byte stream is the one Go's encoder writes (no JFIF segment, both DQT tables, 4:2:0, Annex K Huffman tables). */
size_t ipx_jpeg_coef_count(int w, int h);                 /* int16 elements per frame                 */
int ipx_jpeg_quant_tables(int quality, uint8_t out[128]); /* the two DQT tables, zig-zag order         */
/* n frames resident in HBM -> coefficients in HBM (n * ipx_jpeg_coef_count int16).  Asynchronous. */
int ipx_dev_jpeg_fdct_rgba8(ipx_ctx *ctx, void *stream, const uint8_t *src, int w, int h, int stride,
                            size_t frame_stride, int n, int quality, int16_t *coefs);
/* Host half: coefficients (host memory) -> the complete stream.  *out is malloc'd: ipx_buffer_free. */
int ipx_jpeg_entropy_encode(const int16_t *coefs, int w, int h, int quality, uint8_t **out, size_t *len);
/* jpeg.Encode for one frame in host memory (upload, transform, download, entropy coding). */
int ipx_jpeg_encode_rgba8(ipx_ctx *ctx, const uint8_t *pix, int w, int h, int stride, int quality,
                          uint8_t **out, size_t *len);
/* n frames resident in HBM -> n streams in ONE block of host memory: transform, entropy coding (sizing,
 * placement and 0xff stuffing) and headers all happen on the GPU and only the finished streams cross the
 * link.  *blob is pinned memory owned by the library (ipx_host_free); stream i is blob[offs[i] ..
 * offs[i] + lens[i]).  cgo: wrap each with unsafe.Slice, write it out, then free the block. */
int ipx_jpeg_encode_batch_dev(ipx_ctx *ctx, const uint8_t *src, int w, int h, int stride,
                              size_t frame_stride, int n, int quality, uint8_t **blob, size_t *offs,
                              size_t *lens);
void ipx_buffer_free(void *p);

/* The worker's whole GPU leg for a batch of decoded frames in host memory (image_processor.go:64-77 without
 * image.Decode): upload, every requested operator in one pass, jpeg.Encode of each output on the GPU, and
 * only the finished streams come back.  resize_out / thumb_out / wm_out (n entries each, or NULL to skip)
 * receive pointers into pinned blocks owned by *result; release them with ipx_jpeg_result_free once the
 * streams are written out (fileRepo.SaveProcessed, image_processor.go:76).  Chunks run on the context's
 * lanes, one host thread per lane, so uploads, kernels and downloads of different chunks overlap. */
typedef struct { const uint8_t *data; size_t len; } ipx_bytes;
typedef struct ipx_jpeg_result ipx_jpeg_result;
int ipx_plan_run_host_jpeg(ipx_ctx *ctx, const ipx_plan *plan, int n, const uint8_t *src, int sstride,
                           size_t src_frame_stride, int quality, ipx_bytes *resize_out, ipx_bytes *thumb_out,
                           ipx_bytes *wm_out, ipx_jpeg_result **result);
/* The same for a batch of decoded JPEGs (*image.YCbCr planes): 1.5 bytes per pixel go up for 4:2:0 instead of 4. */
int ipx_plan_run_host_ycbcr_jpeg(ipx_ctx *ctx, const ipx_plan *plan, int n, const ipx_ycbcr_batch *src,
                                 int quality, ipx_bytes *resize_out, ipx_bytes *thumb_out, ipx_bytes *wm_out,
                                 ipx_jpeg_result **result);
void ipx_jpeg_result_free(ipx_ctx *ctx, ipx_jpeg_result *result);

/* ---- image.Decode for JPEGs (decoder side) -------------------------------------------
 * image_processor.go:47 decodes every upload; for JPEG files that is Go's image/jpeg.  A batch of
 * files of one size and one kind (three components at 4:4:4 / 4:2:2 / 4:2:0 / 4:4:0, or one component) is decoded here.  Baseline files:
 * the compressed bytes go up, Huffman decoding runs in parallel inside each scan (or per restart interval), the integer IDCT of idct.go runs
 * block-parallel.  Progressive (SOF2), multi-scan and extended-sequential files: their scans are decoded by the host threads that parse
 * the batch (the scans refine each other), the coefficients go up, IDCT onwards is the same GPU path.  Either way the *image.YCbCr planes
 * (MCU-padded strides, as image.NewYCbCr lays them out) stay in
 * HBM, ready for ipx_plan_run_dev_ycbcr (ratio IPX_GRAY: only y is set; ipx_plan_run_dev_gray).  status[i]: IPX_OK, IPX_ERR_INVALID (malformed:
 * Go's decoder fails on the file too, including a file that ends without EOI) or
 * IPX_ERR_UNSUPPORTED (CMYK / RGB, other samplings, a size or sampling
 * different from the batch's, damaged restart intervals Go would resynchronise over, coefficients beyond int16): the worker decodes those
 * with Go as before.  planes->y == NULL when no
 * image was decodable.  Free the planes with ipx_jpeg_planes_free: they are stream-ordered allocations of `stream` (NULL: the
 * context's default stream), which has to outlive them. */
typedef struct ipx_jpeg_planes ipx_jpeg_planes;
/* Compressed in, compressed out: the uploads as they arrive from the object store (image_processor.go:41-47), the
 * objects as they go back (:76).  image.Decode, every operator of the plan and jpeg.Encode run on the GPU; the
 * link carries ~0.3 MB up and ~0.5 MB down per 1080p image.  status[i] as for ipx_jpeg_decode_batch (the
 * plan's frame size is the batch's size); outputs of undecodable files are {NULL, 0}. */
int ipx_plan_run_jpeg_jpeg(ipx_ctx *ctx, const ipx_plan *plan, int n, const ipx_bytes *files, int quality,
                           ipx_bytes *resize_out, ipx_bytes *thumb_out, ipx_bytes *wm_out, int *status,
                           ipx_jpeg_result **result);
int ipx_jpeg_decode_batch(ipx_ctx *ctx, void *stream, const ipx_bytes *jpegs, int n, int *w, int *h,
                          ipx_ycbcr_batch *planes, int *status, ipx_jpeg_planes **owner);
void ipx_jpeg_planes_free(ipx_ctx *ctx, ipx_jpeg_planes *owner);

/* ---- one process, several GPUs, asynchronous jobs ----------------------------------------------------------
 * The reference worker is ONE process whose goroutines pull independent messages (worker.go:88-96, 112-149); it scales by
 * running more consumers, nothing is exchanged (kafka/consumer.go:23).  A pool is that shape behind the C ABI: one context per
 * listed device (a device may be listed more than once), `lanes_per_device` feeder threads per context with a stream each, and ONE
 * queue of chunks ordered by cost, largest first, that every feeder pulls from when it is free -- a mixed-size batch balances itself
 * (work stealing falls out of pull scheduling), a uniform batch spreads evenly.  No data-path collective, no xGMI traffic.
 * Jobs are asynchronous: ipx_job_submit returns a ticket at once, so a goroutine does not hold an OS thread for the length of a batch.
 * cgo rule: frames, outputs, files and the status array of a job must stay valid (and, for Go, be C-allocated: ipx_pool_host_alloc)
 * until ipx_job_wait has returned; the operator description and its glyph masks are copied at submit. */
typedef struct ipx_pool ipx_pool;
typedef struct {
    int32_t lanes_per_device;   /* feeder threads (one stream each) per listed device; 0 = 4 (the reference's WORKER_CONCURRENCY is 3; four chunks in flight keep the host link busy) */
    size_t lane_bytes;          /* frames per chunk are sized to about this many bytes; 0 = 256 MiB */
} ipx_pool_config;
int ipx_pool_create(const int *devices, int n_devices, const ipx_pool_config *cfg, ipx_pool **out);
void ipx_pool_destroy(ipx_pool *pool);                        /* finishes what is queued, frees what was not released */
int ipx_pool_slots(const ipx_pool *pool);
long long ipx_pool_frames_done(const ipx_pool *pool, int slot); /* frames (files) slot `slot` has processed so far */
/* hipHostMalloc'd memory, allocated and first touched on a thread bound to the CPUs next to slot `slot`'s GPU (sysfs local_cpulist):
 * staging on the GPU's NUMA node.  Free with ipx_pool_host_free. */
void *ipx_pool_host_alloc(ipx_pool *pool, int slot, size_t bytes);
int ipx_pool_host_free(ipx_pool *pool, int slot, void *p);

/* The operators of a job, device independent (ipx_plan_params names a glyph set of ONE context): the same fields, with the
 * rasterised text as host masks.  Plans and uploaded glyph sets are cached per device by content. */
typedef struct {
    int32_t sw, sh;
    int32_t do_resize, resize_w, resize_h, keep_aspect;
    int32_t do_thumbnail, thumb_size, crop_to_fit;
    int32_t do_watermark;
    const ipx_glyph *glyphs;    /* may be NULL (copy only) */
    int32_t n_glyphs;
    uint8_t col[4];             /* parseColor's color.RGBA (watermark.go:93-97) */
} ipx_pool_ops;

/* A plan (and its uploaded glyph set) from the context's cache, keyed by content: what the per-operator entries and the pool use,
 * so that describing the operators per call costs no hipMalloc / hipFree in the steady state (each of those waits for every stream
 * of the device).  *cached = 1: the context owns the plan, and it stays valid at least until the caller hands it back; 0: every plan
 * of the full cache was in use and this one is the caller's for this call.  Either way hand it back with ipx_plan_release.  The
 * cache holds IPX_PLAN_CACHE_MAX (256) plans and replaces the least recently used one nobody holds. */
int ipx_plan_acquire(ipx_ctx *ctx, const ipx_pool_ops *ops, ipx_plan **plan, int *cached);
void ipx_plan_release(ipx_ctx *ctx, ipx_plan *plan, int cached);

enum { IPX_JOB_RGBA8 = 0,       /* decoded frames in, the operators' RGBA8 outputs back (what ipx_plan_run_host does) */
       IPX_JOB_JPEG = 1,        /* uploaded JPEG files in, three JPEG streams per file back (what ipx_plan_run_jpeg_jpeg does) */
       /* decoded frames of the other packed image types, Pix as Go holds it (src / sstride / src_frame_stride describe it), RGBA8
        * outputs back: what ipx_plan_run_host_nrgba / _gray / _deep do */
       IPX_JOB_NRGBA8 = 2, IPX_JOB_GRAY8 = 3,
       IPX_JOB_NRGBA64 = 4, IPX_JOB_RGBA64 = 5, IPX_JOB_GRAY16 = 6, IPX_JOB_CMYK = 7 };
typedef struct {
    int32_t kind;
    ipx_pool_ops ops;
    int32_t n;                  /* frames / files, all of size ops.sw x ops.sh */
    /* pixel jobs (every kind but IPX_JOB_JPEG): frame i at src + i*src_frame_stride; an output pointer may be NULL to skip it */
    const uint8_t *src; int32_t sstride; size_t src_frame_stride;
    uint8_t *resize_out; size_t resize_frame_stride;
    uint8_t *thumb_out; size_t thumb_frame_stride;
    uint8_t *wm_out; size_t wm_frame_stride;
    /* IPX_JOB_JPEG: n files; n entries per output array (or NULL), pointing into pinned blocks the pool owns until ipx_job_release;
     * status[i] as ipx_plan_run_jpeg_jpeg reports it (files Go has to decode itself get IPX_ERR_UNSUPPORTED) */
    const ipx_bytes *files; int32_t quality;
    ipx_bytes *resize_jpeg, *thumb_jpeg, *wm_jpeg;
    int32_t *status;
} ipx_job;
typedef uint64_t ipx_ticket;
int ipx_job_submit(ipx_pool *pool, const ipx_job *job, ipx_ticket *ticket);     /* returns at once */
int ipx_job_poll(ipx_pool *pool, ipx_ticket ticket, int *done);                 /* *done = 1 when ipx_job_wait would not block */
int ipx_job_wait(ipx_pool *pool, ipx_ticket ticket, int *frames_done);          /* blocks; the job's status (first failing chunk's) */
int ipx_job_release(ipx_pool *pool, ipx_ticket ticket);                         /* forgets the job, frees a JPEG job's output blocks */
/* Synchronous convenience for pixel jobs: submit them all (mixed sizes welcome: the queue runs the largest first), wait, release. */
int ipx_pool_run_host(ipx_pool *pool, const ipx_job *jobs, int n_jobs);

/* ---- micro-batching of single uploads ---------------------------------------------------------------------------------------
 * The reference pulls ONE message per goroutine (internal/worker/worker.go:112-149) from a channel of concurrency * 2 (:88); what
 * turns those single files into GPU batches has to sit between the goroutines and the pool, and it sits here, below the ABI, so that
 * its policy is the library's (and tested) rather than every binding's.  ipx_batcher_submit takes ONE uploaded JPEG file with the
 * operators of its task (ops->sw x ops->sh = the frame size from the file's header, image.DecodeConfig) and returns a ticket at once;
 * files are grouped by frame size, JPEG shape (components and luma sampling of the frame header: a batch is one shape) and operator
 * content (parameters, colour, every glyph's rectangle and mask bytes); a group goes to the
 * pool as one IPX_JOB_JPEG when it holds max_batch files or when its first file has waited max_wait_us.  ipx_batcher_wait blocks until
 * the ticket's group is done and fills the file's own result: status IPX_OK and three streams (NULL for operators the task did not ask
 * for), or IPX_ERR_UNSUPPORTED / IPX_ERR_INVALID for a file the GPU path does not decode -- its neighbours are not affected, the worker
 * runs its own image.Decode path for that message.  The streams live in blocks shared by the group: ipx_batcher_release hands a file's
 * share back (call it after fileRepo.SaveProcessed, image_processor.go:76), the blocks go with the group's last file.  At-least-once
 * delivery is unchanged: a goroutine commits its message only after its own ticket came back and its objects were saved.
 * cgo rule: the file bytes must stay valid (C-allocated for Go) until ipx_batcher_wait or ipx_batcher_release has returned for the
 * ticket; the operator description and its glyph masks are copied at submit.  Every entry is thread-safe. */
typedef struct ipx_batcher ipx_batcher;
typedef struct {
    int32_t max_batch;      /* files per job; 0 = 256 (a part of ipx_plan_run_jpeg_jpeg) */
    int32_t max_wait_us;    /* how long the first file of a group may wait for company WHILE other jobs of this batcher run; 0 = 2000 */
    int32_t quality;        /* jpeg.Options.Quality of the outputs; 0 = 85 (domain.DefaultJPEGQuality, task.go:57) */
} ipx_batcher_config;
typedef uint64_t ipx_batch_ticket;
typedef struct {
    int32_t status;               /* of this file: IPX_OK, or why the worker has to decode it itself */
    ipx_bytes resize, thumb, wm;  /* valid until ipx_batcher_release(ticket) */
} ipx_batch_result;
typedef struct {
    long long files, batches, flushed_by_size, flushed_by_timer, largest_batch, pending_files;
    long long flushed_when_idle;   /* groups that left at once because nothing of this batcher was running (a job finishing releases what
                                    * gathered while it ran): a lone file does not wait max_wait_us for company that is not coming */
} ipx_batcher_stats;
int ipx_batcher_create(ipx_pool *pool, const ipx_batcher_config *cfg, ipx_batcher **out);
void ipx_batcher_destroy(ipx_batcher *b);     /* flushes what is pending, waits for it, frees what was not released */
int ipx_batcher_submit(ipx_batcher *b, const ipx_bytes *file, const ipx_pool_ops *ops, ipx_batch_ticket *ticket);
int ipx_batcher_wait(ipx_batcher *b, ipx_batch_ticket ticket, ipx_batch_result *res);
int ipx_batcher_release(ipx_batcher *b, ipx_batch_ticket ticket);
int ipx_batcher_get_stats(ipx_batcher *b, ipx_batcher_stats *out);

#ifdef __cplusplus
}
#endif
#endif /* IPX_H */
