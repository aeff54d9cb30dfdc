/*
 * ipx_oracle.h -- CPU restatement of the reference's pixel hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under imageprocessor_amd/ may include,
 * link, import or execute this; it is the checker for tests/, for
 * __graft_entry__.smoke() and for bench.py's cpu_baseline leg.
 *
 * PARITY UNPINNED: the reference (sj-shoff/ImageProcessor) holds no tests,
 * golden vectors or fixtures for this path, and its arithmetic lives in
 * un-vendored Go modules (golang.org/x/image v0.33.0, golang/freetype
 * e2365dfdc4a0, Go 1.24 stdlib image/draw) with no Go toolchain in the image.
 * Each function below cites the reference call site it stands in for and the
 * upstream routine whose published algorithm it restates.
 */
#ifndef IPX_ORACLE_H
#define IPX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { IPXO_OP_OVER = 0, IPXO_OP_SRC = 1 };

typedef struct { int32_t x0, y0, x1, y1; } ipxo_rect;

/* one DrawMask call of freetype.Context.DrawString (watermark.go:151) */
typedef struct {
    const uint8_t *mask; /* *image.Alpha pixels, origin (0,0)        */
    int32_t mw, mh;      /* mask bounds                              */
    int32_t mstride;
    ipxo_rect dr;        /* destination rectangle handed to DrawMask */
    int32_t mpx, mpy;    /* mask point aligned with dr.Min           */
} ipxo_glyph;

/* resize.go:61-75 */
void ipxo_resize_dims(int ow, int oh, int w, int h, int keep_aspect,
                      int *nw, int *nh);
/* thumbnail.go:48-65 (non-crop) and :114-127 (crop rectangle) */
void ipxo_thumb_geometry(int ow, int oh, int size, int crop_to_fit,
                         ipxo_rect *crop, int *nw, int *nh);
/* watermark.go:116-148: position string -> baseline point in whole pixels */
void ipxo_watermark_anchor(const char *position, int w, int h, int width_px,
                           int height_px, int *px, int *py);
/* watermark.go:116-118 */
int ipxo_text_height_px(double font_size);
/* watermark.go:93-97,159-190; returns 0 ok, 1 = parse error (black fallback applied) */
int ipxo_parse_color(const char *s, double opacity, uint8_t rgba[4]);

/* x/image/draw BiLinear.Scale -- the tent Kernel's two-pass float64 scaler, NOT ApproxBiLinear -- for *image.RGBA <- *image.RGBA
 * (resize.go:123, thumbnail.go:129).
 * dst bounds are (0,0)-(dw,dh); src bounds (0,0)-(sw,sh); sr must lie inside src.
 * Returns 0, or -1 when sr leaves the source (the reference would take the generic
 * Image path, which this restatement does not cover). */
int ipxo_scale_bilinear_rgba8(uint8_t *dst, int dw, int dh, int dstride, ipxo_rect dr,
                              const uint8_t *src, int sw, int sh, int sstride, ipxo_rect sr,
                              int op);

/* image/draw.DrawMask(dst, r, src *image.RGBA, sp, nil, ZP, op) (watermark.go:92 and the
 * equal-size Scale of thumbnail.go:129).  Clips like draw.clip. */
void ipxo_draw_rgba8(uint8_t *dst, int dw, int dh, int dstride, ipxo_rect r,
                     const uint8_t *src, int sw, int sh, int sstride, int spx, int spy,
                     int op);

/* image/draw.DrawMask(dst, dr, image.Uniform{col}, ZP, *image.Alpha, mp, Over), applied for
 * each glyph in order (freetype DrawString, watermark.go:151).  col is color.RGBA bytes. */
void ipxo_composite_glyphs_rgba8(uint8_t *dst, int dw, int dh, int dstride,
                                 const ipxo_glyph *glyphs, int n, const uint8_t col[4]);

/* The three operators of image_processor.go:104-117 on one RGBA8 frame, each applied to the
 * ORIGINAL frame (:64-65).  Any output pointer may be NULL to skip that operator. */
typedef struct {
    int resize_w, resize_h, keep_aspect;   /* resize.go:26-59          */
    int thumb_size, crop_to_fit;           /* thumbnail.go:25-47       */
    const ipxo_glyph *glyphs; int n_glyphs; /* rasterised text (input) */
    uint8_t col[4];                        /* watermark.go:93-97       */
} ipxo_pipeline;

int ipxo_process_rgba8(const ipxo_pipeline *p, const uint8_t *src, int sw, int sh, int sstride,
                       uint8_t *resize_out, uint8_t *thumb_out, uint8_t *wm_out);

#ifdef __cplusplus
}
#endif

/* ---- source-type variants: PNG with alpha decodes to *image.NRGBA, JPEG to
 * *image.YCbCr; the reference hands those straight to the same three helpers. ------------------ */
#ifdef __cplusplus
extern "C" {
#endif

/* x/image/draw kernelScaler with scaleX_NRGBA: each tap premultiplied (c * a16 / 0xff) before it is weighted. */
int ipxo_scale_bilinear_nrgba8(uint8_t *dst, int dw, int dh, int dstride, ipxo_rect dr,
                               const uint8_t *src, int sw, int sh, int sstride, ipxo_rect sr, int op);
void ipxo_draw_nrgba8(uint8_t *dst, int dw, int dh, int dstride, ipxo_rect r,
                      const uint8_t *src, int sw, int sh, int sstride, int spx, int spy, int op);

/* *image.Paletted through the generic routines (scaleX_Image + scaleY_RGBA_{Src,Over}, image/draw drawRGBA); pal16[i] = Palette[i].RGBA() */
int ipxo_scale_bilinear_paletted(uint8_t *dst, int dw, int dh, int dstride, ipxo_rect dr, const uint8_t *src, int sw, int sh,
                                 int sstride, const uint16_t pal16[256][4], ipxo_rect sr, int op);
void ipxo_draw_paletted(uint8_t *dst, int dw, int dh, int dstride, ipxo_rect r, const uint8_t *src, int sw, int sh,
                        int sstride, const uint16_t pal16[256][4], int spx, int spy, int op);

/* *image.NRGBA64 / *image.RGBA64 / *image.Gray16 (16-bit PNGs) and *image.CMYK through the generic routines (CMYK: drawCMYK); src is
 * Go's Pix (big-endian 16-bit channels; C M Y K bytes) */
enum { IPXO_DEEP_NRGBA64 = 0, IPXO_DEEP_RGBA64 = 1, IPXO_DEEP_GRAY16 = 2, IPXO_DEEP_CMYK = 3 };
int ipxo_deep_bpp(int kind);
void ipxo_deep_taps(uint16_t *out, const uint8_t *src, int sw, int sh, int sstride, int kind);   /* At(x, y).RGBA() per pixel */
int ipxo_scale_bilinear_deep(uint8_t *dst, int dw, int dh, int dstride, ipxo_rect dr, const uint8_t *src, int sw, int sh, int sstride,
                             int kind, ipxo_rect sr, int op);
void ipxo_draw_deep(uint8_t *dst, int dw, int dh, int dstride, ipxo_rect r, const uint8_t *src, int sw, int sh, int sstride, int kind,
                    int spx, int spy, int op);

/* *image.YCbCr with Rect.Min = (0,0); ratio = image.YCbCrSubsampleRatio (444=0, 422=1, 420=2, 440=3) */
typedef struct {
    const uint8_t *y, *cb, *cr;
    int32_t ystride, cstride, w, h, ratio;
} ipxo_ycbcr;
/* x/image/draw kernelScaler with scaleX_YCbCr{444,422,420,440} (YCbCr is opaque, so Over becomes Src): each tap
 * converted with color.YCbCr.RGBA's 16-bit integer formula, alpha 0xffff.  Equal sizes: DrawYCbCr. */
int ipxo_scale_bilinear_ycbcr(uint8_t *dst, int dw, int dh, int dstride, ipxo_rect dr,
                              const ipxo_ycbcr *src, ipxo_rect sr);
/* image/internal/imageutil.DrawYCbCr (draw.Draw from a YCbCr source): 8-bit conversion, A = 255 */
void ipxo_draw_ycbcr(uint8_t *dst, int dw, int dh, int dstride, ipxo_rect r, const ipxo_ycbcr *src,
                     int spx, int spy);

#ifdef __cplusplus
}
#endif
#endif
