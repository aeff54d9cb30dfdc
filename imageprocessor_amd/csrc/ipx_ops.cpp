// ipx_ops.cpp -- the reference's operator method set, on decoded frames, ABOVE the C ABI.
//
// Mirrors internal/usecase/processor/operations/{resize,thumbnail,watermark}.go (parameter type
// switches, defaults, error texts, output-format rules) and image_processor.go (operator loop on
// the ORIGINAL frame, wrapped error texts, generatePath, getContentType).  Everything here calls
// only the extern "C" entry points of include/ipx.h: it is the C++ stand-in for the Go adapters of
// INTEGRATION.md (there is no Go toolchain in the build image) and uses no library internals.
#include <cctype>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ipx.h"

namespace ipx {
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int status_of_exception() noexcept;   // inside a catch block: the ipx_status an exception maps to (nothing unwinds through the ABI)
}
#define IPX_CATCH_STATUS catch (...) { return ipx::status_of_exception(); }

namespace {

struct OpError {
    int status;
    std::string text;
};

std::string lower(const char *s)
{
    std::string r(s ? s : "");
    for (auto &c : r) c = (char)tolower((unsigned char)c);
    return r;
}

const ipx_param *find(const ipx_param *p, int n, const char *key)
{
    for (int i = 0; i < n; i++)
        if (p[i].key && !strcmp(p[i].key, key)) return &p[i];
    return nullptr;
}

// the `v.(float64) / .(int) / .(int64) / .(int32)` ladders of resize.go:27-50, thumbnail.go:26-37
bool number(const ipx_param *p, long long *out)
{
    if (!p) return false;
    switch (p->type) {
    case IPX_PT_FLOAT64: *out = (long long)p->f64; return true;  // int(w): truncation
    case IPX_PT_INT: case IPX_PT_INT64: case IPX_PT_INT32: *out = p->i64; return true;
    default: return false;
    }
}
bool boolean(const ipx_param *p) { return p && p->type == IPX_PT_BOOL && p->i64 != 0; }  // v, _ := x.(bool)

// encoder switch: resize.go:78-91, thumbnail.go:68-81
const char *static_format(const std::string &f)
{
    if (f == "jpg" || f == "jpeg") return "jpeg";
    if (f == "png") return "png";
    if (f == "gif") return "gif";
    return "jpeg";
}

struct ResizeReq { int w = 0, h = 0; bool keep = false; };
struct ThumbReq { int size = 0; bool crop = false; };
struct WmReq {
    std::string text, position, color;
    double opacity = 0.5, font_size = 36;
};

bool parse_resize(const ipx_param *p, int n, ResizeReq *r, std::string *err)
{
    long long w, h;
    if (!number(find(p, n, "width"), &w)) { *err = "width parameter is required and must be a number"; return false; }
    if (!number(find(p, n, "height"), &h)) { *err = "height parameter is required and must be a number"; return false; }
    if (w <= 0 || h <= 0) { *err = "width and height must be positive numbers"; return false; }
    if (w > 0x3fffffff || h > 0x3fffffff) { *err = "width and height are too large"; return false; }   // beyond 65535 the plan reports IPX_ERR_UNSUPPORTED
    r->w = (int)w; r->h = (int)h; r->keep = boolean(find(p, n, "keep_aspect"));
    return true;
}

bool parse_thumb(const ipx_param *p, int n, ThumbReq *r, std::string *err)
{
    long long s;
    if (!number(find(p, n, "size"), &s)) s = 200;  // domain.DefaultThumbnailSize (task.go:56)
    if (s <= 0) { *err = "size must be a positive number"; return false; }
    if (s > 0x3fffffff) { *err = "size is too large"; return false; }
    r->size = (int)s; r->crop = boolean(find(p, n, "crop_to_fit"));
    return true;
}

void parse_wm(const ipx_param *p, int n, WmReq *r)  // watermark.go:41-60: no parameter can fail
{
    const ipx_param *q;
    q = find(p, n, "text");
    r->text = q && q->type == IPX_PT_STRING && q->str && *q->str ? q->str : "\xC2\xA9 ImageProcessor";
    q = find(p, n, "opacity");
    r->opacity = q && q->type == IPX_PT_FLOAT64 && q->f64 > 0 ? q->f64 : 0.5;
    q = find(p, n, "position");
    r->position = q && q->type == IPX_PT_STRING && q->str ? q->str : "bottom-right";
    q = find(p, n, "font_size");
    r->font_size = q && q->type == IPX_PT_FLOAT64 && q->f64 > 0 ? q->f64 : 36;
    q = find(p, n, "font_color");
    r->color = q && q->type == IPX_PT_STRING && q->str ? q->str : "255,255,255";
}

bool alloc_image(ipx_image *im, int w, int h)
{
    if (w > 0x1fffffff) return false;               // the stride is an int32 in ipx_image (plans cap sides at 65535 anyway)
    im->w = w; im->h = h; im->stride = w * 4;
    im->pix = (uint8_t *)malloc((size_t)(w > 0 && h > 0 ? (size_t)w * h * 4 : 1));
    return im->pix != nullptr;
}

// One pass over the frame for whichever of the three operators is requested.
// Returns 0 or a negative ipx_status with *err set.
int run_ops(ipx_ctx *ctx, const ipx_image *img, const ResizeReq *rz, const ThumbReq *th, const WmReq *wm,
            const ipx_text_rasterizer *font, ipx_image *o_rz, ipx_image *o_th, ipx_image *o_wm, std::string *err)
{
    if (!img || !img->pix || img->w <= 0 || img->h <= 0 || img->stride < img->w * 4) {
        *err = "invalid source frame";
        return IPX_ERR_INVALID;
    }
    bool rasterised = false;
    int rc = IPX_OK;
    ipx_pool_ops po;
    memset(&po, 0, sizeof po);
    po.sw = img->w; po.sh = img->h;
    if (wm) {  // addTextWatermark, watermark.go:86-157
        if (!font || !font->measure || !font->glyphs) { *err = "font not loaded"; return IPX_ERR_INVALID; }
        (void)ipx_parse_color(wm->color.c_str(), wm->opacity, po.col);  // a parse error falls back to black (:93-97)
        int width_px = 0;
        if (font->measure(font->user, wm->text.c_str(), wm->font_size, &width_px)) {
            *err = "failed to draw watermark text: rasteriser failed";
            return IPX_ERR_INVALID;
        }
        const int height_px = ipx_text_height_px(wm->font_size);
        int px, py;
        ipx_watermark_anchor(wm->position.c_str(), img->w, img->h, width_px, height_px, &px, &py);
        const ipx_glyph *gl = nullptr;
        int ng = 0;
        if (font->glyphs(font->user, wm->text.c_str(), wm->font_size, px, py, img->w, img->h, &gl, &ng)) {
            *err = "failed to draw watermark text: rasteriser failed";
            return IPX_ERR_INVALID;
        }
        rasterised = true;
        po.do_watermark = 1; po.glyphs = gl; po.n_glyphs = ng;
    }
    if (rz) { po.do_resize = 1; po.resize_w = rz->w; po.resize_h = rz->h; po.keep_aspect = rz->keep; }
    if (th) { po.do_thumbnail = 1; po.thumb_size = th->size; po.crop_to_fit = th->crop; }
    // the plan (tap tables, uploaded masks) comes from the context's cache, keyed by content: the same text on the same frame size --
    // the worker's usual case -- costs no device allocation after the first call
    ipx_plan *plan = nullptr;
    int cached = 0;
    rc = ipx_plan_acquire(ctx, &po, &plan, &cached);
    ipx_plan_info info;
    if (!rc) rc = ipx_plan_query(plan, &info);
    if (!rc) {
        bool ok = true;
        if (rz) ok = ok && alloc_image(o_rz, info.resize_w, info.resize_h);
        if (th) ok = ok && alloc_image(o_th, info.thumb_w, info.thumb_h);
        if (wm) ok = ok && alloc_image(o_wm, info.wm_w, info.wm_h);
        if (!ok) { rc = IPX_ERR_NOMEM; ipx::set_error("out of memory"); }
    }
    if (!rc)
        rc = ipx_plan_run_host(ctx, plan, 1, img->pix, img->stride, 0, rz ? o_rz->pix : nullptr, 0,
                               th ? o_th->pix : nullptr, 0, wm ? o_wm->pix : nullptr, 0);
    if (rc) *err = ipx_last_error();
    if (plan) ipx_plan_release(ctx, plan, cached);
    if (rasterised && font->release) font->release(font->user);
    return rc;
}

void put(char *dst, size_t cap, const std::string &s) { snprintf(dst, cap, "%s", s.c_str()); }

std::string content_type(const std::string &path)  // image_processor.go:164-182
{
    const size_t dot = path.rfind('.');
    const std::string ext = dot == std::string::npos ? "" : lower(path.c_str() + dot);
    if (ext == ".jpg" || ext == ".jpeg") return "image/jpeg";
    if (ext == ".png") return "image/png";
    if (ext == ".gif") return "image/gif";
    if (ext == ".webp") return "image/webp";
    if (ext == ".bmp") return "image/bmp";
    if (ext == ".tiff" || ext == ".tif") return "image/tiff";
    return "image/jpeg";
}

// generatePath reads width/height/size with the SHORT ladder (float64, int) only: :133-155
long long path_number(const ipx_param *p)
{
    if (!p) return 0;
    if (p->type == IPX_PT_FLOAT64) return (long long)p->f64;
    if (p->type == IPX_PT_INT) return p->i64;
    return 0;
}

}  // namespace

extern "C" {

void ipx_image_free(ipx_image *img)
{
    if (img && img->pix) { free(img->pix); img->pix = nullptr; }
}

int ipx_resizer_process(ipx_ctx *ctx, const ipx_image *img, const char *format, const ipx_param *params,
                        int nparams, ipx_image *out, char out_format[8]) try
{
    if (!ctx || !out || !out_format) { ipx::set_error("ipx_resizer_process: bad argument"); return IPX_ERR_INVALID; }
    memset(out, 0, sizeof *out);
    ResizeReq r;
    std::string err;
    if (!parse_resize(params, nparams, &r, &err)) { ipx::set_error("%s", err.c_str()); return IPX_ERR_INVALID; }
    int rc = run_ops(ctx, img, &r, nullptr, nullptr, nullptr, out, nullptr, nullptr, &err);
    if (rc) { ipx_image_free(out); ipx::set_error("%s", err.c_str()); return rc; }
    const std::string f = lower(format);
    snprintf(out_format, 8, "%s", f == "gif" ? "gif" : static_format(f));  // resize.go:55-58,78-91
    return IPX_OK;
}
IPX_CATCH_STATUS

int ipx_thumbnailer_process(ipx_ctx *ctx, const ipx_image *img, const char *format, const ipx_param *params,
                            int nparams, ipx_image *out, char out_format[8]) try
{
    if (!ctx || !out || !out_format) { ipx::set_error("ipx_thumbnailer_process: bad argument"); return IPX_ERR_INVALID; }
    memset(out, 0, sizeof *out);
    ThumbReq t;
    std::string err;
    if (!parse_thumb(params, nparams, &t, &err)) { ipx::set_error("%s", err.c_str()); return IPX_ERR_INVALID; }
    int rc = run_ops(ctx, img, nullptr, &t, nullptr, nullptr, nullptr, out, nullptr, &err);
    if (rc) { ipx_image_free(out); ipx::set_error("%s", err.c_str()); return rc; }
    const std::string f = lower(format);
    snprintf(out_format, 8, "%s", f == "gif" ? "gif" : static_format(f));  // thumbnail.go:42-46,68-81
    return IPX_OK;
}
IPX_CATCH_STATUS

int ipx_watermarker_process(ipx_ctx *ctx, const ipx_image *img, const char *format, const ipx_param *params,
                            int nparams, const ipx_text_rasterizer *font, ipx_image *out, char out_format[8]) try
{
    if (!ctx || !out || !out_format) { ipx::set_error("ipx_watermarker_process: bad argument"); return IPX_ERR_INVALID; }
    memset(out, 0, sizeof *out);
    WmReq w;
    parse_wm(params, nparams, &w);
    std::string err;
    int rc = run_ops(ctx, img, nullptr, nullptr, &w, font, nullptr, nullptr, out, &err);
    if (rc) {  // watermark.go:61-64
        ipx_image_free(out);
        ipx::set_error("failed to add watermark: %s", err.c_str());
        return rc;
    }
    const std::string f = lower(format);
    snprintf(out_format, 8, "%s", f == "png" ? "png" : "jpeg");  // watermark.go:66-79: gif and others -> jpeg
    return IPX_OK;
}
IPX_CATCH_STATUS

int ipx_processor_process(ipx_ctx *ctx, const ipx_task *task, const ipx_image *decoded, const char *decoded_format,
                          const ipx_text_rasterizer *font, ipx_processed *out, int *n_out) try
{
    if (!ctx || !task || !out || !n_out || (task->nops && !task->ops)) {
        ipx::set_error("ipx_processor_process: bad argument");
        return IPX_ERR_INVALID;
    }
    *n_out = 0;
    const std::string target = task->format && *task->format ? task->format : (decoded_format ? decoded_format : "");
    const std::string lf = lower(target.c_str());
    const std::string id = task->image_id ? task->image_id : "";

    // Parse operators in order up to the first that fails; the ones before it still run and are
    // returned, as the reference has already stored their results when it hits the error (:64-92).
    struct Parsed { int kind; ResizeReq r; ThumbReq t; WmReq w; };
    std::vector<Parsed> ops;
    int fail_status = IPX_OK;
    std::string fail_text;
    int seen[3] = {0, 0, 0};
    bool fusable = true;
    for (int i = 0; i < task->nops; i++) {
        const ipx_operation &op = task->ops[i];
        const std::string type = op.type ? op.type : "";
        Parsed p;
        std::string err;
        bool ok = true;
        if (type == "resize") { p.kind = 0; ok = parse_resize(op.params, op.nparams, &p.r, &err); }
        else if (type == "thumbnail") { p.kind = 1; ok = parse_thumb(op.params, op.nparams, &p.t, &err); }
        else if (type == "watermark") {
            p.kind = 2; parse_wm(op.params, op.nparams, &p.w);
            if (!font || !font->measure || !font->glyphs) { ok = false; err = "failed to add watermark: font not loaded"; }
        } else {  // image_processor.go:115-117
            fail_status = IPX_ERR_UNSUPPORTED;
            fail_text = "operation " + type + " failed: unsupported operation type: " + type;
            break;
        }
        if (!ok) {  // :118-120 wrapped by :66-75
            fail_status = IPX_ERR_INVALID;
            fail_text = "operation " + type + " failed: failed to process operation " + type + ": " + err;
            break;
        }
        if (seen[p.kind]++) fusable = false;
        ops.push_back(p);
    }

    std::vector<ipx_image> imgs(ops.size());
    for (auto &im : imgs) memset(&im, 0, sizeof im);
    int rc = IPX_OK;
    std::string err;
    size_t done = 0;
    if (fusable && !ops.empty()) {
        const ResizeReq *rz = nullptr; const ThumbReq *th = nullptr; const WmReq *wm = nullptr;
        ipx_image *o[3] = {nullptr, nullptr, nullptr};
        for (size_t i = 0; i < ops.size(); i++) {
            if (ops[i].kind == 0) rz = &ops[i].r;
            if (ops[i].kind == 1) th = &ops[i].t;
            if (ops[i].kind == 2) wm = &ops[i].w;
            o[ops[i].kind] = &imgs[i];
        }
        rc = run_ops(ctx, decoded, rz, th, wm, font, o[0], o[1], o[2], &err);
        if (!rc) done = ops.size();
        else {
            // the fused pass failed as a whole (e.g. the rasteriser): go operator by operator, so that the operators before the
            // failing one still deliver their results and the error names the one that failed, as the sequential loop of
            // image_processor.go:64-92 would
            for (auto &im : imgs) { ipx_image_free(&im); memset(&im, 0, sizeof im); }
            fusable = false;
            rc = IPX_OK;
            err.clear();
        }
    }
    if (!fusable) {
        for (size_t i = 0; i < ops.size() && !rc; i++) {
            const Parsed &p = ops[i];
            rc = run_ops(ctx, decoded, p.kind == 0 ? &p.r : nullptr, p.kind == 1 ? &p.t : nullptr,
                         p.kind == 2 ? &p.w : nullptr, font, &imgs[i], &imgs[i], &imgs[i], &err);
            if (!rc) done = i + 1;
        }
    }
    if (rc) {
        static const char *names[3] = {"resize", "thumbnail", "watermark"};
        const char *nm = done < ops.size() ? names[ops[done].kind] : "";
        fail_status = rc;
        fail_text = std::string("operation ") + nm + " failed: failed to process operation " + nm + ": " +
                    (done < ops.size() && ops[done].kind == 2 ? "failed to add watermark: " : "") + err;
        for (size_t i = done; i < imgs.size(); i++) ipx_image_free(&imgs[i]);
    }

    for (size_t i = 0; i < done; i++) {
        const Parsed &p = ops[i];
        const ipx_operation &op = task->ops[i];
        ipx_processed &r = out[i];
        memset(&r, 0, sizeof r);
        std::string fmt, path;
        char buf[512];
        if (p.kind == 0) {
            fmt = lf == "gif" ? "gif" : static_format(lf);
            snprintf(buf, sizeof buf, "processed/resize/%s/%lldx%lld.%s", id.c_str(), path_number(find(op.params, op.nparams, "width")),
                     path_number(find(op.params, op.nparams, "height")), fmt.c_str());
            put(r.operation, sizeof r.operation, "resize");
        } else if (p.kind == 1) {
            fmt = lf == "gif" ? "gif" : static_format(lf);
            long long s = path_number(find(op.params, op.nparams, "size"));
            if (s == 0) s = 200;
            snprintf(buf, sizeof buf, "processed/thumbnails/%s/%lld.%s", id.c_str(), s, fmt.c_str());
            put(r.operation, sizeof r.operation, "thumbnail");
        } else {
            fmt = lf == "png" ? "png" : "jpeg";
            snprintf(buf, sizeof buf, "processed/watermarked/%s/watermarked.%s", id.c_str(), fmt.c_str());
            put(r.operation, sizeof r.operation, "watermark");
        }
        path = buf;
        put(r.path, sizeof r.path, path);
        put(r.content_type, sizeof r.content_type, content_type(path));
        put(r.format, sizeof r.format, fmt);
        r.image = imgs[i];
    }
    *n_out = (int)done;
    if (fail_status) { ipx::set_error("%s", fail_text.c_str()); return fail_status; }
    return IPX_OK;
}
IPX_CATCH_STATUS

}  // extern "C"
