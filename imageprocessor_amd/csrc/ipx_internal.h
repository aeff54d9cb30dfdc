// ipx_internal.h -- shared declarations of libipx (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>

#include "../../include/ipx.h"

namespace ipx {

// ---- error plumbing: every ABI entry returns a status, text goes to a thread-local buffer ----
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void clear_error();
// Nothing may unwind through the C ABI (a cgo or ctypes caller would be terminated): every extern "C" entry that can allocate is a
// function-try-block closed by IPX_CATCH_STATUS, which maps std::bad_alloc / std::system_error to IPX_ERR_NOMEM and anything else to
// IPX_ERR_INVALID, with the text in ipx_last_error().  Call status_of_exception() only inside a catch block.
int status_of_exception() noexcept;
#define IPX_CATCH_STATUS catch (...) { return ipx::status_of_exception(); }

#define IPX_HIP(call)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            ipx::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,  \
                           __LINE__);                                                        \
            return IPX_ERR_HIP;                                                              \
        }                                                                                    \
    } while (0)

// ---- frame geometry the kernels can address --------------------------------------------------------------------------
// Frames are addressed through buffer descriptors (32-bit num_records) and int byte offsets; an idle lane's offset is kOOB =
// 0x7fffffff, which has to lie beyond every frame.  So a frame's span (h-1)*stride + w*bpp must stay below 2 GiB, and a side
// below 65536 (grid dimensions; also jpeg.Encode's own limit).  A 32 MiB PNG upload (domain/task.go:55) can decode past that:
// such frames get IPX_ERR_UNSUPPORTED and stay on the reference's CPU path -- never wrapped offsets.
constexpr long long kMaxFrameSpan = 0x7fff0000LL;
constexpr int kMaxFrameSide = 65535;
inline bool frame_span_ok(long long w, long long h, long long stride, int bpp)
{
    if (w > kMaxFrameSide || h > kMaxFrameSide) return false;
    return w <= 0 || h <= 0 || (h - 1) * stride + w * bpp <= kMaxFrameSpan;
}

// ---- image.Rectangle arithmetic (image/geom.go semantics) ------------------------------------
struct Rect {
    int x0, y0, x1, y1;
    int dx() const { return x1 - x0; }
    int dy() const { return y1 - y0; }
    bool empty() const { return x0 >= x1 || y0 >= y1; }
    Rect shifted(int ox, int oy) const { return Rect{x0 + ox, y0 + oy, x1 + ox, y1 + oy}; }
    Rect intersect(const Rect &s) const;
};
inline Rect to_rect(const ipx_rect &r) { return Rect{r.x0, r.y0, r.x1, r.y1}; }

// image/draw.clip for DrawMask: narrows r to dst, to the source placed by sp and, when mask_w >= 0,
// to the mask placed by mp; advances sp / mp by the amount r.Min moved.  has_src false = Uniform.
bool draw_clip(Rect &r, int dw, int dh, bool has_src, int sw, int sh, int &spx, int &spy,
               bool has_mask, int mw, int mh, int &mpx, int &mpy);

// ---- kernel launchers (ipx_kernels.hip) --------------------------------------------------------
// how the pixels of a source image in HBM are laid out (the tap kinds built on them: ipx_ks.h)
enum { IPX_SRC_RGBA = 0, IPX_SRC_NRGBA = 1, IPX_SRC_YCBCR = 2,
       IPX_SRC_TAP64 = 3 /* At(x, y).RGBA() as four little-endian uint16 per pixel: the deep source types after deep_expand_kernel */ };
hipError_t launch_opaque_scan(const uint8_t *src, int sw, int sh, int sstride, int *flag,
                              hipStream_t s);
hipError_t launch_opaque_scan_tap64(const uint8_t *src, int sw, int sh, int sstride, int *flag, hipStream_t s);
hipError_t launch_draw(uint8_t *dst, int dstride, const uint8_t *src, int sstride, int w, int h,
                       int op, hipStream_t s);
hipError_t launch_draw_nrgba(uint8_t *dst, int dstride, const uint8_t *src, int sstride, int w, int h, int op,
                             hipStream_t s, int nframes = 1, size_t dst_fs = 0, size_t src_fs = 0);
// deep source types (ipx.h IPX_DEEP_*): Go's Pix -> frames of 16-bit taps (dst rows w * 8 bytes apart), and taps -> RGBA8 (drawRGBA / drawCMYK)
hipError_t launch_deep_expand(uint8_t *dst, size_t dst_fs, const uint8_t *src, int sstride, size_t src_fs, int kind, int w, int h, int nframes, hipStream_t s);
hipError_t launch_draw_tap64(uint8_t *dst, int dstride, const uint8_t *src, int sstride, int w, int h, int op, hipStream_t s,
                             int nframes = 1, size_t dst_fs = 0, size_t src_fs = 0);
hipError_t launch_draw_ycbcr(uint8_t *dst, int dstride, const uint8_t *y, int ystride, const uint8_t *cb,
                             const uint8_t *cr, int cstride, int ratio, int spx, int spy, int w, int h, hipStream_t s,
                             int nframes = 1, size_t dst_fs = 0, size_t y_fs = 0, size_t c_fs = 0);

struct DevGlyph {           // one clipped DrawMask call, masks resident in HBM
    const uint8_t *mask;    // points at mask(mpx, mpy) after clipping
    int mstride;
    int x0, y0, x1, y1;     // clipped destination rectangle
};
constexpr int kMaxGlyphs = 256;
hipError_t launch_stream_copy(void *dst, const void *src, size_t bytes, hipStream_t s);   // the box's streaming ceiling (bench.py)
hipError_t launch_composite(uint8_t *dst, int dstride, size_t frame_stride, int nframes,
                            const DevGlyph *glyphs_dev, int n, Rect bbox, uint32_t sr, uint32_t sg,
                            uint32_t sb, uint32_t sa, hipStream_t s);

hipError_t launch_gray_expand(uint8_t *dst, size_t dst_fs, const uint8_t *src, int sstride, size_t src_fs, int w, int h, int n, hipStream_t s);
hipError_t launch_palette_expand(uint8_t *dst, size_t dst_fs, const uint8_t *src, int sstride, size_t src_fs, const uint8_t *palettes, int w,
                                 int h, int n, hipStream_t s);

// Per kernel instantiation and process: the dynamic-LDS limit is raised once (it only has to be at least what a launch asks for) and the
// occupancy is cached per LDS size.  These are properties of the loaded function, not of the calling thread; done per thread, every
// short-lived worker thread repeated hipFuncSetAttribute / hipOccupancy... in the middle of other threads' launches.
struct KernelLaunchCache {
    std::mutex mu;
    size_t lds_max = 0;
    std::map<size_t, int> resident;
    bool said = false;
    hipError_t prepare(const void *fn, int threads, size_t lds, int *res)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (lds > lds_max) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            lds_max = lds;
        }
        if (res) {
            auto it = resident.find(lds);
            if (it == resident.end()) {
                int n = 0;
                hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, threads, lds);
                if (e != hipSuccess) return e;
                it = resident.emplace(lds, n > 1 ? n : 1).first;
            }
            *res = it->second;
        }
        return hipSuccess;
    }
    bool first_report() { std::lock_guard<std::mutex> lk(mu); const bool f = !said; said = true; return f; }
};


// ---- jpeg.Encode: transform on the GPU (ipx_jpeg.hip), tables / headers / entropy coder on the host ----
struct JpegTables {
    uint8_t quant[2][64];     // zig-zag order, as DQT carries them (e.quant of writer.go)
    uint16_t div8[2][64];     // 8 * quant, natural order: the divisor of writeBlock's div()
    uint32_t recip[2][64];    // ceil(2^32 / div8): floor(n / div8) == mulhi(n, recip) for n < 2^20
};
void jpeg_tables(int quality, JpegTables *t);
struct JpegArgs {
    const uint8_t *src; size_t frame_stride; int stride, w, h;
    int aligned16;            // base, row stride and frame stride are multiples of 16
    int16_t *coefs; int mcus_per_frame;
    // optional (all three or none): the AC part of every block's Huffman-coded length, and the quantised DC terms as a dense array --
    // computed while the block sits in LDS, so that no separate pass has to read the coefficients back to size the scan
    const uint32_t *huff;     // jpeg_huff_packed: four tables of 256 (len << 16 | code)
    uint32_t *aclen;          // [frame][block] bits of the AC symbols (ZRL, EOB included)
    int16_t *dcq;             // [frame][block]
    uint32_t recip[2][64];
    uint16_t div8[2][64];
};
hipError_t launch_jpeg_fdct(const JpegArgs &a, int n, hipStream_t s);


// ---- image.Decode for baseline JPEGs (ipx_jpeg_dec_host.cpp parses, ipx_jpeg_dec.hip decodes) ------------
struct JpegDecInfo {
    int w, h, h0, v0, ratio, ri, ncomp;
    int host_scans;            // 1: not one baseline scan the GPU's Huffman kernels take (progressive, several scans, table ids above 1):
                               // jpeg_host_decode walks the scans on the host, the coefficients go up, the GPU does the rest
    int progressive;           // SOF2 (reconstructProgressiveImage's rule for blocks outside the image applies)
    uint8_t td[3], ta[3];      // kernel table slots of the three components: 0,1 = DC tables, 2,3 = AC tables
    size_t scan_off, scan_len; // entropy-coded data within the file
};
struct JpegDecTables {         // per image, as the kernels read them
    uint16_t lut[4][256];      // first level: length << 8 | symbol for codes of at most 8 bits, else 0
    int32_t maxcode[4][18];    // by code length; -1 = no code of that length
    int32_t valoff[4][18];     // symbol index = valoff[len] + code
    uint8_t vals[4][256];
    uint16_t qnat[3][64];      // quantiser per component, natural order
    uint32_t bound[4][8];      // for lengths 9..16: first 16-bit left-aligned value NOT covered by codes of at most that length
};
// one independently decodable piece of a scan: the whole scan, or one restart interval of it (DC predictions and the bit
// reader start afresh after every RSTn, so intervals decode in parallel)
struct JpegDecImage {
    unsigned long long scan_off; uint32_t scan_len, img, first_mcu, n_mcu;
    uint8_t td[3], ta[3], valid, pad;   // pad: bytes to skip at the start (pieces are read in aligned 16-byte chunks)
    uint8_t strict_end, rsv[3];         // a restart interval must end exactly at its RSTn marker (Go would resynchronise: not ours to guess)
    unsigned long long uoff;            // where the piece's unstuffed copy lives (launch_jpeg_pieces), 16-byte aligned, regions do not overlap
    uint32_t tab_img, rsv2;             // the image whose Huffman tables this piece decodes with (the first of the batch with identical tables)
};
struct JpegDecArgs {
    const uint8_t *blob; const JpegDecImage *img; const JpegDecTables *tab;
    int16_t *coefs; int *status;
    int16_t *dcs;                    // [n][nblk] the DC coefficient of every block (coefs[...][0] stays zero)
    int n, mxx, myy, h0, v0, nblk;   // n = images
    int w, h;                        // the batch's frame size (progressive images reconstruct only blocks that hold image pixels)
    int bpm, ybl;                    // blocks per MCU and how many of them are luma (Gray: 1, 1)
    int nitems;                      // pieces to decode (>= images)
    int shared_tables, first_valid;   // every valid image carries the Huffman tables of image first_valid
};
// the per-image status word of the decoder kernels: 0, or a negative key whose low three bits are -IPX_ERR_* and whose upper bits order
// the reporting pieces so that atomicMin keeps the verdict of the piece a sequential decoder would have reached first
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int jpeg_status_key(uint32_t first_mcu, int status) { return -(int)(((0x3ffffffu - (first_mcu & 0x3ffffffu)) << 3) | (uint32_t)(-status)); }
// a DC value beyond int16: Go keeps int32 and decodes such a (damaged) file, this pipeline cannot represent it -> UNSUPPORTED, with the
// lowest priority: any real decoding error elsewhere in the file, which Go fails on as well, wins
constexpr uint32_t kJpegStatusLast = 0x3ffffffu;
inline int jpeg_status_of(int key) { return key >= 0 ? 0 : -(int)((uint32_t)(-key) & 7u); }
struct JpegPlanes { uint8_t *y, *cb, *cr; int ystride, cstride; size_t y_fs, c_fs; const uint8_t *valid; /* per image: bit 0 decodable, bit 1 progressive */ };
// Huffman decoding parallel inside a scan (ipx_jpeg_dec_par.hip): per image and per 1 KiB sub-sequence of its scan
struct JpegParImage { unsigned long long scan_off; uint32_t scan_len, img, nsub; size_t sub_off; uint8_t td[3], ta[3], pad[2]; };
struct JpegParArgs {
    const uint8_t *blob; const JpegDecTables *tab; const JpegParImage *img;
    uint8_t *ublob;                        // the scans without stuffing, same offsets as blob, zero filled
    int nimg, max_nsub, bpm, ybl, nblk;
    int stage_rows;                        // 1: the workgroup's 64 sub-sequences are staged in LDS (rows of sub + 4 bytes: small batches); 0: read through L1 / L2; 2: diagnostic, see CoefSink::drop
    int sub;                               // bytes of scan per sub-sequence: 128, 256, 512 or 1024 (jpeg_par_sub_bytes(): the largest)
    uint32_t *stuffed;                     // [nimg][max_nsub] stuffed zeros before each sub-sequence's chunk (after the scan)
    uint32_t *scan_end, *ulen;             // per image: first marker in the stuffed scan; bytes of the unstuffed scan
    unsigned long long *entry, *exit_a, *exit_b;
    unsigned long long *ck_state; uint32_t *ck_ends;   // per sub-sequence, jpeg_par_checkpoints() each: state and block ends at the checkpoints
    uint32_t *ends, *total_ends;           // block ends per sub-sequence (after the scan: first block index); per image total
    uint32_t *changed;
    int16_t *coefs; int *status;
    int16_t *dcs;                          // [image][nblk]: DC differences from the write pass, DC values after par_dc_kernel
};
int jpeg_par_sub_bytes();
int jpeg_par_checkpoints();
hipError_t launch_par_count(const JpegParArgs &a, hipStream_t s);
hipError_t launch_par_unstuff(const JpegParArgs &a, hipStream_t s);
hipError_t launch_par_sync(const JpegParArgs &a, int round, hipStream_t s);
hipError_t launch_par_write(const JpegParArgs &a, hipStream_t s);
hipError_t launch_par_dc(const JpegParArgs &a, hipStream_t s);
int jpeg_parse(const uint8_t *d, size_t len, JpegDecInfo *info, JpegDecTables *tab);
}  // namespace ipx
#include <vector>
namespace ipx {
int jpeg_host_decode(const uint8_t *d, size_t len, JpegDecInfo *info, int16_t *coefs, int16_t *dcs, size_t nblk, uint16_t qnat[3][64],
                     bool *progressive);
hipError_t launch_jpeg_huff(const JpegDecArgs &a, hipStream_t s);
// the same pieces through the word-wise reader of the parallel decoder: unstuff each piece into ublob (+ its length into ulen), then one
// decode pass -- the state at the start of a piece is known, so nothing is speculative.  The 64 pieces of a workgroup share one set
// of tables (img[64 * k].tab_img): the host groups the pieces by table set and pads each group to a multiple of 64.
hipError_t launch_jpeg_pieces(const JpegDecArgs &a, uint8_t *ublob, uint32_t *ulen, hipStream_t s);
hipError_t launch_jpeg_idct(const JpegDecArgs &a, const JpegPlanes &pl, hipStream_t s);

}  // namespace ipx

#include <vector>
namespace ipx {
void jpeg_write_stream(const int16_t *coefs, int w, int h, const JpegTables &t, std::vector<uint8_t> *out);
void jpeg_write_header(int w, int h, const JpegTables &t, std::vector<uint8_t> *out);
void jpeg_huff_packed(uint32_t out[1024]);
// the entropy coder on the GPU (ipx_jpeg_entropy.hip)
hipError_t launch_jpeg_len(const int16_t *coefs, int nblk, int n, const uint32_t *tables, uint32_t *len, hipStream_t s);
// len[b] += bits of block b's DC symbol (difference to the previous block of its component), from the dense DC array of the transform kernel
hipError_t launch_jpeg_dclen(const int16_t *dcq, int nblk, int n, const uint32_t *tables, uint32_t *len, hipStream_t s);
hipError_t launch_jpeg_bits(const int16_t *coefs, int nblk, int n, const uint32_t *tables, const uint32_t *off, const uint32_t *total_bits,
                            const unsigned long long *ubase, uint8_t *ustream, hipStream_t s);
hipError_t launch_scan(uint32_t *v, int per_frame, int n, uint32_t *total, hipStream_t s);
int jpeg_chunk_bytes();
hipError_t launch_jpeg_ffcount(const uint8_t *ustream, const unsigned long long *ubase, const uint32_t *ubytes, int max_chunks, int n,
                               uint32_t *ffcount, hipStream_t s);
hipError_t launch_jpeg_stuff(const uint8_t *ustream, const unsigned long long *ubase, const uint32_t *ubytes, int max_chunks, int n,
                             const uint32_t *ffoff, const uint8_t *header, int hdr_len, const unsigned long long *obase, uint8_t *ostream,
                             hipStream_t s);
}  // namespace ipx
