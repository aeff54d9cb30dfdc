// ipx_host.cpp -- host-side rules of the path: geometry, clipping, parameter parsing.
//
// These mirror the Go code that surrounds the pixel loops in the reference
// (internal/usecase/processor/operations/{resize,thumbnail,watermark}.go) and the clipping that
// image/draw and x/image/draw perform before their inner loops.  Compiled without FMA contraction
// (x86-64 baseline has no FMA; -ffp-contract=off is passed as well) so the float64 results are
// those of the reference's GOAMD64=v1 build.
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <system_error>
#include <vector>

#include "ipx_internal.h"

namespace ipx {

static thread_local char g_err[512];

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
void clear_error() { g_err[0] = 0; }

int status_of_exception() noexcept
{
    try { throw; }
    catch (const std::bad_alloc &) { set_error("out of memory"); return IPX_ERR_NOMEM; }
    catch (const std::system_error &e) { set_error("system resource unavailable: %s", e.what()); return IPX_ERR_NOMEM; }
    catch (const std::exception &e) { set_error("internal error: %s", e.what()); return IPX_ERR_INVALID; }
    catch (...) { set_error("internal error"); return IPX_ERR_INVALID; }
}

Rect Rect::intersect(const Rect &s) const
{
    Rect r = *this;
    if (r.x0 < s.x0) r.x0 = s.x0;
    if (r.y0 < s.y0) r.y0 = s.y0;
    if (r.x1 > s.x1) r.x1 = s.x1;
    if (r.y1 > s.y1) r.y1 = s.y1;
    if (r.empty()) return Rect{0, 0, 0, 0};
    return r;
}

bool draw_clip(Rect &r, int dw, int dh, bool has_src, int sw, int sh, int &spx, int &spy,
               bool has_mask, int mw, int mh, int &mpx, int &mpy)
{
    const int ox = r.x0, oy = r.y0;
    r = r.intersect(Rect{0, 0, dw, dh});
    if (has_src) r = r.intersect(Rect{0, 0, sw, sh}.shifted(ox - spx, oy - spy));
    if (has_mask) r = r.intersect(Rect{0, 0, mw, mh}.shifted(ox - mpx, oy - mpy));
    if (r.empty()) return false;
    const int dx = r.x0 - ox, dy = r.y0 - oy;
    spx += dx; spy += dy;
    mpx += dx; mpy += dy;
    return true;
}

}  // namespace ipx

using namespace ipx;

extern "C" {

const char *ipx_last_error(void) { return g_err; }
int ipx_abi_version(void) { return IPX_ABI_VERSION; }

int ipx_frame_supported(int w, int h, long long stride, int bytes_per_pixel)
{
    clear_error();
    if (w < 0 || h < 0 || bytes_per_pixel <= 0 || stride < (long long)w * bytes_per_pixel) {
        set_error("ipx_frame_supported: bad argument");
        return IPX_ERR_INVALID;
    }
    if (frame_span_ok(w, h, stride, bytes_per_pixel)) return IPX_OK;
    set_error("a %dx%d frame (stride %lld) is beyond the 2 GiB / 65535-pixel span the kernels address", w, h, stride);
    return IPX_ERR_UNSUPPORTED;
}

int ipx_resize_dims(int ow, int oh, int w, int h, int keep_aspect, int *nw, int *nh)
{
    clear_error();
    if (!nw || !nh || ow <= 0 || oh <= 0) { set_error("ipx_resize_dims: bad argument"); return IPX_ERR_INVALID; }
    // resize.go:51-53: "width and height must be positive numbers"
    if (w <= 0 || h <= 0) { set_error("width and height must be positive numbers"); return IPX_ERR_INVALID; }
    if (keep_aspect) {  // resize.go:63-72
        const double wr = (double)w / (double)ow;
        const double hr = (double)h / (double)oh;
        const double ratio = std::fmin(wr, hr);
        *nw = (int)((double)ow * ratio);
        *nh = (int)((double)oh * ratio);
    } else {            // resize.go:73-75
        *nw = w;
        *nh = h;
    }
    return IPX_OK;
}

int ipx_thumb_geometry(int ow, int oh, int size, int crop_to_fit, ipx_rect *crop, int *nw, int *nh)
{
    clear_error();
    if (!crop || !nw || !nh || ow <= 0 || oh <= 0) { set_error("ipx_thumb_geometry: bad argument"); return IPX_ERR_INVALID; }
    if (size == 0) size = 200;  // domain/task.go:56 DefaultThumbnailSize (thumbnail.go:35-37)
    if (size < 0) { set_error("size must be a positive number"); return IPX_ERR_INVALID; }  // thumbnail.go:38-40
    if (crop_to_fit) {  // thumbnail.go:114-127
        const int side = ow > oh ? oh : ow;
        const int cx = ow > oh ? (ow - oh) / 2 : 0;
        const int cy = ow > oh ? 0 : (oh - ow) / 2;
        *crop = ipx_rect{cx, cy, cx + side, cy + side};
        *nw = size;
        *nh = size;
    } else {            // thumbnail.go:52-65
        *crop = ipx_rect{0, 0, ow, oh};
        if (ow > oh) {
            *nh = size;
            *nw = (int)((double)ow * (double)size / (double)oh);
        } else {
            *nw = size;
            *nh = (int)((double)oh * (double)size / (double)ow);
        }
    }
    return IPX_OK;
}

int ipx_text_height_px(double font_size)
{
    // watermark.go:116,118: fixed.Int26_6(fontSize * 64 * 1.2).Ceil()
    const int32_t h26_6 = (int32_t)(font_size * 64 * 1.2);
    return (h26_6 + 63) >> 6;
}

int ipx_watermark_anchor(const char *position, int w, int h, int width_px, int height_px, int *px,
                         int *py)
{
    clear_error();
    if (!position || !px || !py) { set_error("ipx_watermark_anchor: bad argument"); return IPX_ERR_INVALID; }
    enum { L, R, C } hx = R;      // watermark.go:145-147: anything else is bottom-right
    enum { T, B, M } vy = B;
    const std::string p(position);
    if (p == "top-left") { hx = L; vy = T; }
    else if (p == "top-right") { hx = R; vy = T; }
    else if (p == "top-center") { hx = C; vy = T; }
    else if (p == "bottom-left") { hx = L; vy = B; }
    else if (p == "bottom-center") { hx = C; vy = B; }
    else if (p == "center") { hx = C; vy = M; }
    const int margin = 20;        // watermark.go:121
    *px = hx == L ? margin : hx == R ? w - width_px - margin : (w - width_px) / 2;
    *py = vy == T ? margin + height_px : vy == B ? h - margin : (h + height_px) / 2;
    return IPX_OK;
}

// strconv.Atoi on a field: sign, then decimal digits only
static bool atoi_field(const std::string &f, long long *v)
{
    size_t i = 0;
    bool neg = false;
    if (f.empty()) return false;
    if (f[0] == '+' || f[0] == '-') { neg = f[0] == '-'; i = 1; }
    if (i == f.size()) return false;
    long long acc = 0;
    for (; i < f.size(); i++) {
        if (f[i] < '0' || f[i] > '9') return false;
        if (acc > 900000000000000000LL) return false;  // range error
        acc = acc * 10 + (f[i] - '0');
    }
    *v = neg ? -acc : acc;
    return true;
}

int ipx_parse_color(const char *s, double opacity, uint8_t rgba[4])
{
    clear_error();
    if (!s || !rgba) { set_error("ipx_parse_color: bad argument"); return IPX_ERR_INVALID; }
    const uint8_t alpha = (uint8_t)(int32_t)(255 * opacity);  // uint8(255 * opacity)
    std::vector<std::string> parts(1);
    for (const char *c = s; *c; c++) {
        if (*c == ' ') continue;            // strings.ReplaceAll(colorStr, " ", "")
        if (*c == ',') parts.emplace_back();
        else parts.back().push_back(*c);
    }
    long long v[4] = {0, 0, 0, 0};
    bool ok = parts.size() == 3 || parts.size() == 4;
    for (int i = 0; ok && i < 3; i++) ok = atoi_field(parts[i], &v[i]);
    if (!ok) {  // watermark.go:93-97: black, opacity alpha
        rgba[0] = rgba[1] = rgba[2] = 0;
        rgba[3] = alpha;
        return 1;
    }
    for (int i = 0; i < 3; i++) rgba[i] = (uint8_t)(v[i] < 0 ? 0 : v[i] > 255 ? 255 : v[i]);
    if (parts.size() == 4 && atoi_field(parts[3], &v[3])) rgba[3] = (uint8_t)(v[3] < 0 ? 0 : v[3] > 255 ? 255 : v[3]);
    else rgba[3] = alpha;
    return IPX_OK;
}

}  // extern "C"
