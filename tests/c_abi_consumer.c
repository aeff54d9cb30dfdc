/* c_abi_consumer.c -- libipx through its C ABI from plain C99, no C++ and no Python in between: the call sequence the cgo binding
 * (the files under go/ipx) makes, replayed on the GPU and compared with the CPU oracle byte for byte.
 *
 *   1. Context, pinned staging, glyph set, plan, ipx_plan_run_host              (go/ipx/ipx.go, plan.go: New, Pinned, NewPlan, RunHost)
 *   2. ipx_scale_bilinear_rgba8 / ipx_composite_glyphs_rgba8 on one frame       (ScaleBilinear, CompositeGlyphs: the narrow seam)
 *   3. ipx_plan_run_host_jpeg: every stream vs the oracle's jpeg.Encode         (RunHostJPEG)
 *   4. ipx_plan_run_jpeg_jpeg on the streams of step 3                          (RunJPEGJPEG)
 *   5. ipx_pool_create {0, 0}, a pixel job and a JPEG job, wait, release        (go/ipx/pool.go: NewPool, SubmitPixels, SubmitJPEG)
 *   6. the error contract: a status and a text, never an abort                  (IsUnsupported)
 *
 * Built and run by tests/test_c_abi_consumer.py: gcc -std=c99 -pedantic -Wall -Werror, linked with libipx.so and the oracle.
 * The oracle is the checker here, nothing more (oracle/ipx_oracle.h). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ipx.h"
#include "ipx_oracle.h"

/* the oracle's codec entries (oracle/ipx_jpeg_oracle.c) */
int ipxo_jpeg_encode_rgba8(const uint8_t *pix, int w, int h, int stride, int quality, uint8_t **out, size_t *out_len, int16_t *coefs);
void ipxo_free(void *p);

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "c_abi_consumer: " __VA_ARGS__); fprintf(stderr, " [%s] (%s:%d)\n", ipx_last_error(), __FILE__, __LINE__); return 1; } } while (0)

enum { W = 640, H = 360, N = 5, NG = 6 };

static uint32_t lcg_state = 20261004u;
static uint8_t lcg(void) { lcg_state = lcg_state * 1664525u + 1013904223u; return (uint8_t)(lcg_state >> 24); }

int main(void)
{
    ipx_ctx *ctx = NULL;
    ipx_config cfg;
    int i, k, rc;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = 0;
    CHECK(ipx_abi_version() == IPX_ABI_VERSION, "ABI version");
    CHECK(ipx_create(&cfg, &ctx) == IPX_OK, "ipx_create");

    /* ---- 1. frames in pinned memory, glyphs, plan, one batched pass ---- */
    {
        const size_t fb = (size_t)W * H * 4;
        uint8_t *src = (uint8_t *)ipx_host_alloc(ctx, fb * N);
        uint8_t masks[NG][24 * 20];
        ipx_glyph gl[NG];
        ipxo_glyph ogl[NG];
        const uint8_t col[4] = {255, 255, 255, 127};      /* parseColor("255,255,255", 0.5): not premultiplied */
        ipx_glyphset *gs = NULL;
        ipx_plan *plan = NULL;
        ipx_plan_params pp;
        ipx_plan_info info;
        uint8_t *res, *th, *wm, *ores, *oth, *owm;
        ipxo_pipeline op;
        CHECK(src != NULL, "ipx_host_alloc");
        for (i = 0; i < (int)(fb * N); i++) src[i] = (i & 3) == 3 ? 255 : lcg();
        for (k = 0; k < NG; k++) {
            for (i = 0; i < 24 * 20; i++) { const uint8_t v = lcg(); masks[k][i] = v < 90 ? 0 : (v > 190 ? 255 : v); }
            gl[k].mask = masks[k]; gl[k].mw = 24; gl[k].mh = 20; gl[k].mstride = 24;
            gl[k].dr.x0 = W - 200 + k * 22; gl[k].dr.y0 = H - 45 + (k & 3); gl[k].dr.x1 = gl[k].dr.x0 + 24; gl[k].dr.y1 = gl[k].dr.y0 + 20;
            gl[k].mpx = 0; gl[k].mpy = 0;
            ogl[k].mask = masks[k]; ogl[k].mw = 24; ogl[k].mh = 20; ogl[k].mstride = 24;
            ogl[k].dr.x0 = gl[k].dr.x0; ogl[k].dr.y0 = gl[k].dr.y0; ogl[k].dr.x1 = gl[k].dr.x1; ogl[k].dr.y1 = gl[k].dr.y1;
            ogl[k].mpx = 0; ogl[k].mpy = 0;
        }
        CHECK(ipx_glyphset_create(ctx, gl, NG, col, &gs) == IPX_OK, "ipx_glyphset_create");
        memset(&pp, 0, sizeof pp);
        pp.sw = W; pp.sh = H;
        pp.do_resize = 1; pp.resize_w = 1024; pp.resize_h = 768; pp.keep_aspect = 1;
        pp.do_thumbnail = 1; pp.thumb_size = 200; pp.crop_to_fit = 1;
        pp.do_watermark = 1; pp.glyphs = gs;
        CHECK(ipx_plan_create(ctx, &pp, &plan) == IPX_OK, "ipx_plan_create");
        CHECK(ipx_plan_query(plan, &info) == IPX_OK, "ipx_plan_query");
        CHECK(info.resize_w == 1024 && info.resize_h == 576 && info.thumb_w == 200 && info.thumb_h == 200, "aspect rules");
        res = (uint8_t *)ipx_host_alloc(ctx, info.resize_bytes * N);
        th = (uint8_t *)ipx_host_alloc(ctx, info.thumb_bytes * N);
        wm = (uint8_t *)ipx_host_alloc(ctx, info.wm_bytes * N);
        CHECK(res && th && wm, "output staging");
        CHECK(ipx_plan_run_host(ctx, plan, N, src, W * 4, fb, res, info.resize_bytes, th, info.thumb_bytes, wm, info.wm_bytes) == IPX_OK,
              "ipx_plan_run_host");
        ores = (uint8_t *)malloc(info.resize_bytes); oth = (uint8_t *)malloc(info.thumb_bytes); owm = (uint8_t *)malloc(info.wm_bytes);
        memset(&op, 0, sizeof op);
        op.resize_w = 1024; op.resize_h = 768; op.keep_aspect = 1; op.thumb_size = 200; op.crop_to_fit = 1;
        op.glyphs = ogl; op.n_glyphs = NG; memcpy(op.col, col, 4);
        for (i = 0; i < N; i++) {
            CHECK(ipxo_process_rgba8(&op, src + fb * i, W, H, W * 4, ores, oth, owm) == 0, "oracle");
            CHECK(!memcmp(res + info.resize_bytes * i, ores, info.resize_bytes), "resize of frame %d differs from the oracle", i);
            CHECK(!memcmp(th + info.thumb_bytes * i, oth, info.thumb_bytes), "thumbnail of frame %d differs from the oracle", i);
            CHECK(!memcmp(wm + info.wm_bytes * i, owm, info.wm_bytes), "watermark of frame %d differs from the oracle", i);
        }

        /* ---- 2. the narrow seam on frame 0 ---- */
        {
            ipx_rect dr, sr;
            ipxo_rect odr, osr;
            uint8_t *d = (uint8_t *)calloc(300 * 170, 4), *od = (uint8_t *)calloc(300 * 170, 4);
            uint8_t *f0 = (uint8_t *)malloc(fb);
            dr.x0 = 0; dr.y0 = 0; dr.x1 = 300; dr.y1 = 170; sr.x0 = 10; sr.y0 = 5; sr.x1 = 630; sr.y1 = 355;
            odr.x0 = 0; odr.y0 = 0; odr.x1 = 300; odr.y1 = 170; osr.x0 = 10; osr.y0 = 5; osr.x1 = 630; osr.y1 = 355;
            CHECK(ipx_scale_bilinear_rgba8(ctx, d, 300, 170, 1200, dr, src, W, H, W * 4, sr, IPX_OP_OVER) == IPX_OK, "ipx_scale_bilinear_rgba8");
            CHECK(ipxo_scale_bilinear_rgba8(od, 300, 170, 1200, odr, src, W, H, W * 4, osr, IPXO_OP_OVER) == 0, "oracle scale");
            CHECK(!memcmp(d, od, 300 * 170 * 4), "scale differs from the oracle");
            memcpy(f0, src, fb);
            CHECK(ipx_composite_glyphs_rgba8(ctx, f0, W, H, W * 4, gl, NG, col) == IPX_OK, "ipx_composite_glyphs_rgba8");
            CHECK(!memcmp(f0, wm, fb), "composite on frame 0 differs from the fused pass");
            free(d); free(od); free(f0);
        }

        /* ---- 2b. another decoded type through the plan: the RGBA8 bytes read as *image.Gray16 Pix (go/ipx/sources.go RunHostDeep) ---- */
        {
            const int gw = W * 2, gb = gw * 2;                     /* a 2W x H frame of big-endian uint16 */
            ipx_plan_params gp;
            ipx_plan *gplan = NULL;
            ipx_plan_info gi;
            ipxo_rect gdr, gsr;
            uint8_t *gres, *ogres;
            memset(&gp, 0, sizeof gp);
            gp.sw = gw; gp.sh = H; gp.do_resize = 1; gp.resize_w = 400; gp.resize_h = 90; gp.keep_aspect = 0;
            CHECK(ipx_plan_create(ctx, &gp, &gplan) == IPX_OK && ipx_plan_query(gplan, &gi) == IPX_OK, "plan for Gray16 frames");
            gres = (uint8_t *)malloc(gi.resize_bytes * N); ogres = (uint8_t *)calloc(gi.resize_bytes, 1);
            CHECK(gres && ogres, "malloc");
            CHECK(ipx_plan_run_host_deep(ctx, gplan, N, IPX_DEEP_GRAY16, src, gb, fb, gres, gi.resize_bytes, NULL, 0, NULL, 0) == IPX_OK, "ipx_plan_run_host_deep");
            gdr.x0 = 0; gdr.y0 = 0; gdr.x1 = 400; gdr.y1 = 90; gsr.x0 = 0; gsr.y0 = 0; gsr.x1 = gw; gsr.y1 = H;
            CHECK(ipxo_scale_bilinear_deep(ogres, 400, 90, 1600, gdr, src + fb, gw, H, gb, IPXO_DEEP_GRAY16, gsr, IPXO_OP_OVER) == 0, "oracle Gray16 scale");
            CHECK(!memcmp(gres + gi.resize_bytes, ogres, gi.resize_bytes), "Gray16 resize of frame 1 differs from the oracle");
            CHECK(ipx_plan_run_host_deep(ctx, gplan, N, 9, src, gb, fb, gres, gi.resize_bytes, NULL, 0, NULL, 0) == IPX_ERR_INVALID, "an unknown deep type is an argument error");
            free(gres); free(ogres);
            ipx_plan_destroy(ctx, gplan);
        }

        /* ---- 3. operators + jpeg.Encode on the GPU ---- */
        {
            ipx_bytes jr[N], jt[N], jw[N], files[N], r2[N], t2[N], w2[N];
            ipx_jpeg_result *result = NULL, *result2 = NULL;
            int status[N];
            ipx_plan *plan2 = NULL;
            CHECK(ipx_plan_run_host_jpeg(ctx, plan, N, src, W * 4, fb, 85, jr, jt, jw, &result) == IPX_OK, "ipx_plan_run_host_jpeg");
            for (i = 0; i < N; i++) {
                uint8_t *o = NULL;
                size_t olen = 0;
                CHECK(ipxo_jpeg_encode_rgba8(res + info.resize_bytes * i, 1024, 576, 4096, 85, &o, &olen, NULL) == 0, "oracle jpeg");
                CHECK(olen == jr[i].len && !memcmp(o, jr[i].data, olen), "resize stream %d differs from the oracle's jpeg.Encode", i);
                ipxo_free(o);
                CHECK(jt[i].len > 100 && jw[i].len > 100 && jw[i].data[0] == 0xff && jw[i].data[1] == 0xd8, "stream %d", i);
            }
            /* ---- 4. compressed in, compressed out: the resize streams (1024 x 576) are the uploads now ---- */
            memset(&pp, 0, sizeof pp);
            pp.sw = 1024; pp.sh = 576; pp.do_resize = 1; pp.resize_w = 320; pp.resize_h = 180; pp.do_thumbnail = 1; pp.thumb_size = 64; pp.crop_to_fit = 1;
            pp.do_watermark = 1;
            CHECK(ipx_plan_create(ctx, &pp, &plan2) == IPX_OK, "plan for the second stage");
            for (i = 0; i < N; i++) files[i] = jr[i];
            files[2].len = 150;                                     /* a truncated upload */
            CHECK(ipx_plan_run_jpeg_jpeg(ctx, plan2, N, files, 85, r2, t2, w2, status, &result2) == IPX_OK, "ipx_plan_run_jpeg_jpeg");
            for (i = 0; i < N; i++) {
                if (i == 2) { CHECK(status[i] != IPX_OK && r2[i].data == NULL, "the truncated file must be reported, not guessed at"); continue; }
                CHECK(status[i] == IPX_OK && r2[i].len > 100 && t2[i].len > 100 && w2[i].len > 100, "file %d", i);
            }
            ipx_jpeg_result_free(ctx, result2);

            /* ---- 5. the pool: two slots on device 0, a pixel job and a JPEG job in flight together ---- */
            {
                const int devices[2] = {0, 0};
                ipx_pool *pool = NULL;
                ipx_job job, jjob;
                ipx_ticket t1 = 0, t2k = 0;
                uint8_t *psrc, *pres, *pth, *pwm;
                ipx_bytes pr[N], pt[N];
                int32_t pstatus[N];
                int done = 0;
                CHECK(ipx_pool_create(devices, 2, NULL, &pool) == IPX_OK && ipx_pool_slots(pool) == 2, "ipx_pool_create");
                psrc = (uint8_t *)ipx_pool_host_alloc(pool, 1, fb * N);
                pres = (uint8_t *)ipx_pool_host_alloc(pool, 0, info.resize_bytes * N);
                pth = (uint8_t *)ipx_pool_host_alloc(pool, 0, info.thumb_bytes * N);
                pwm = (uint8_t *)ipx_pool_host_alloc(pool, 1, info.wm_bytes * N);
                CHECK(psrc && pres && pth && pwm, "ipx_pool_host_alloc");
                memcpy(psrc, src, fb * N);
                memset(&job, 0, sizeof job);
                job.kind = IPX_JOB_RGBA8; job.n = N;
                job.ops.sw = W; job.ops.sh = H; job.ops.do_resize = 1; job.ops.resize_w = 1024; job.ops.resize_h = 768; job.ops.keep_aspect = 1;
                job.ops.do_thumbnail = 1; job.ops.thumb_size = 200; job.ops.crop_to_fit = 1; job.ops.do_watermark = 1;
                job.ops.glyphs = gl; job.ops.n_glyphs = NG; memcpy(job.ops.col, col, 4);
                job.src = psrc; job.sstride = W * 4; job.src_frame_stride = fb;
                job.resize_out = pres; job.resize_frame_stride = info.resize_bytes; job.thumb_out = pth; job.thumb_frame_stride = info.thumb_bytes;
                job.wm_out = pwm; job.wm_frame_stride = info.wm_bytes;
                CHECK(ipx_job_submit(pool, &job, &t1) == IPX_OK, "ipx_job_submit (pixels)");
                memset(&jjob, 0, sizeof jjob);
                jjob.kind = IPX_JOB_JPEG; jjob.n = N; jjob.ops.sw = 1024; jjob.ops.sh = 576; jjob.ops.do_resize = 1; jjob.ops.resize_w = 320; jjob.ops.resize_h = 180;
                jjob.ops.do_thumbnail = 1; jjob.ops.thumb_size = 64; jjob.ops.crop_to_fit = 1;
                jjob.files = jr; jjob.quality = 85; jjob.resize_jpeg = pr; jjob.thumb_jpeg = pt; jjob.status = pstatus;
                CHECK(ipx_job_submit(pool, &jjob, &t2k) == IPX_OK, "ipx_job_submit (JPEG)");
                CHECK(ipx_job_poll(pool, t1, &done) == IPX_OK, "ipx_job_poll");
                CHECK(ipx_job_wait(pool, t1, &done) == IPX_OK && done == N, "ipx_job_wait (pixels)");
                CHECK(!memcmp(pres, res, info.resize_bytes * N) && !memcmp(pth, th, info.thumb_bytes * N) && !memcmp(pwm, wm, info.wm_bytes * N),
                      "the pool's outputs differ from ipx_plan_run_host's");
                CHECK(ipx_job_wait(pool, t2k, NULL) == IPX_OK, "ipx_job_wait (JPEG)");
                for (i = 0; i < N; i++) CHECK(pstatus[i] == IPX_OK && pr[i].len > 100 && pt[i].len > 50, "JPEG job, file %d", i);
                CHECK(ipx_job_release(pool, t1) == IPX_OK && ipx_job_release(pool, t2k) == IPX_OK, "ipx_job_release");
                CHECK(ipx_job_wait(pool, t1, NULL) == IPX_ERR_INVALID, "a released ticket is unknown");
                ipx_pool_host_free(pool, 1, psrc); ipx_pool_host_free(pool, 0, pres); ipx_pool_host_free(pool, 0, pth); ipx_pool_host_free(pool, 1, pwm);
                ipx_pool_destroy(pool);
            }
            ipx_jpeg_result_free(ctx, result);
            ipx_plan_destroy(ctx, plan2);
        }

        /* ---- 6. errors are statuses with a text ---- */
        memset(&pp, 0, sizeof pp);
        pp.sw = 32768; pp.sh = 16384; pp.do_watermark = 1;
        {
            ipx_plan *bad = NULL;
            rc = ipx_plan_create(ctx, &pp, &bad);
            CHECK(rc == IPX_ERR_UNSUPPORTED && bad == NULL && strstr(ipx_last_error(), "span") != NULL, "a 2 GiB frame must be refused (got %d)", rc);
            CHECK(ipx_frame_supported(32768, 16384, 131072, 4) == IPX_ERR_UNSUPPORTED && ipx_frame_supported(1920, 1080, 7680, 4) == IPX_OK, "ipx_frame_supported");
            CHECK(ipx_plan_run_host(ctx, NULL, 1, src, W * 4, fb, NULL, 0, NULL, 0, NULL, 0) == IPX_ERR_INVALID, "a null plan is an argument error");
        }
        free(ores); free(oth); free(owm);
        ipx_host_free(ctx, res); ipx_host_free(ctx, th); ipx_host_free(ctx, wm); ipx_host_free(ctx, src);
        ipx_plan_destroy(ctx, plan);
        ipx_glyphset_destroy(ctx, gs);
    }
    ipx_destroy(ctx);
    printf("c_abi_consumer ok: plans, the narrow seam, the codec legs, the pool and the error contract through the C ABI from C99\n");
    return 0;
}
