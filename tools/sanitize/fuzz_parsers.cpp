// CPU-side sanitizer run of the parsers that read bytes from outside: the JPEG marker parser (ipx_jpeg_dec_host.cpp, fed with
// every upload), the host scan decoder of progressive / multi-scan files (ipx_jpeg_dec_prog.cpp) and the TrueType loader / rasteriser (ipx_font.cpp).  Built with -fsanitize=address,undefined by tools/sanitize/run.sh;
// inputs: seed files given on the command line (*.jpg, *.ttf), mutated with a fixed-seed generator.  Any report aborts the run.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../imageprocessor_amd/csrc/ipx_internal.h"

namespace ipx {
void set_error(const char *, ...) {}
}

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint32_t rnd()
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 32);
}

static std::vector<uint8_t> slurp(const char *path)
{
    std::vector<uint8_t> v;
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f);
    return v;
}

static void mutate(std::vector<uint8_t> &v, size_t lo, size_t hi)
{
    if (hi <= lo) return;
    switch (rnd() % 5) {
    case 0: for (int k = 1 + rnd() % 4; k > 0; k--) v[lo + rnd() % (hi - lo)] ^= (uint8_t)(1u << (rnd() % 8)); break;
    case 1: for (int k = 1 + rnd() % 4; k > 0; k--) v[lo + rnd() % (hi - lo)] = (uint8_t)rnd(); break;
    case 2: v.resize(lo + rnd() % (hi - lo)); break;
    case 3: { const size_t a = lo + rnd() % (hi - lo), n = std::min<size_t>(v.size() - a, 1 + rnd() % 64); v.erase(v.begin() + a, v.begin() + a + n); break; }
    default: { const size_t a = lo + rnd() % (hi - lo); for (int k = 0; k < 2; k++) if (a + k < v.size()) v[a + k] = k ? 0xff : (uint8_t)(0xff - rnd() % 3); break; }   // large lengths
    }
}

int main(int argc, char **argv)
{
    const int cases = argc > 1 ? atoi(argv[1]) : 2000;
    long jpeg_ok = 0, jpeg_bad = 0, font_ok = 0, font_bad = 0, host_ok = 0, host_bad = 0;
    for (int a = 2; a < argc; a++) {
        const std::string path = argv[a];
        const std::vector<uint8_t> seed = slurp(argv[a]);
        const bool is_font = path.size() > 4 && path.substr(path.size() - 4) == ".ttf";
        for (int t = 0; t < cases; t++) {
            std::vector<uint8_t> v = seed;
            if (is_font) {
                // table directory and the tables the loader reads are spread over the file: mutate anywhere, mostly near the front
                if (t) mutate(v, 0, (rnd() & 1) ? std::min<size_t>(v.size(), 4096) : v.size());
                // exact-size heap copy so that any read past the end is an ASan report
                uint8_t *heap = (uint8_t *)malloc(v.size() ? v.size() : 1);
                memcpy(heap, v.data(), v.size());
                ipx_font *font = nullptr;
                if (ipx_font_create(heap, v.size(), &font) == IPX_OK && font) {
                    font_ok++;
                    int32_t w26 = 0; int wpx = 0;
                    (void)ipx_font_text_width(font, "Sample Watermark \xc3\xa9\xe2\x82\xac", 24.0 + (t % 40), &w26, &wpx);
                    const ipx_glyph *gl = nullptr; int n = 0;
                    if (ipx_font_draw_string(font, "Wj\xc3\xa9.", 12.0 + (t % 90), 10, 60, 640, 360, &gl, &n, nullptr) == IPX_OK) ipx_font_release_thread();
                    (void)ipx_font_kern(font, 'A', 'V', 32.0, &w26);
                    ipx_font_destroy(font);
                } else font_bad++;
                free(heap);
            } else {
                size_t sos = 0;
                for (size_t i = 0; i + 1 < v.size(); i++) if (v[i] == 0xff && v[i + 1] == 0xda) { sos = i; break; }
                // header region for the marker parser; every other case anywhere in the file, so that the host scan decoder
                // (progressive / multi-scan files) sees broken entropy-coded data, lost scans and misplaced markers as well
                if (t) mutate(v, 2, (t & 1) ? v.size() : std::min(v.size(), sos + 16));
                uint8_t *heap = (uint8_t *)malloc(v.size() ? v.size() : 1);
                memcpy(heap, v.data(), v.size());
                ipx::JpegDecInfo info;
                static ipx::JpegDecTables tab;
                const int rc = ipx::jpeg_parse(heap, v.size(), &info, &tab);
                if (rc == IPX_OK) {
                    jpeg_ok++;
                    if (info.w <= 0 || info.h <= 0 || (!info.host_scans && info.scan_off + info.scan_len > v.size())) { fprintf(stderr, "inconsistent parse result\n"); abort(); }
                    if (info.host_scans) {
                        // exact-size slices, as the runtime hands them out (one past the end is an ASan report)
                        const size_t mxx = (info.w + 8 * info.h0 - 1) / (8 * info.h0), myy = (info.h + 8 * info.v0 - 1) / (8 * info.v0);
                        const size_t nblk = mxx * myy * (info.ncomp == 1 ? 1 : info.h0 * info.v0 + 2);
                        if (nblk > (size_t)1 << 22) { host_bad++; free(heap); continue; }      // (a mutated size: the runtime's batch geometry check comes first)
                        int16_t *coefs = (int16_t *)malloc(nblk * 64 * sizeof(int16_t)), *dcs = (int16_t *)malloc(nblk * sizeof(int16_t));
                        uint16_t qnat[3][64];
                        bool prog = false;
                        ipx::JpegDecInfo hi = info;
                        const int hr = ipx::jpeg_host_decode(heap, v.size(), &hi, coefs, dcs, nblk, qnat, &prog);
                        if (hr == IPX_OK) {
                            host_ok++;
                            if (hi.w != info.w || hi.h != info.h) { fprintf(stderr, "host decoder disagrees with the parser\n"); abort(); }
                        } else host_bad++;
                        free(coefs); free(dcs);
                    }
                } else jpeg_bad++;
                free(heap);
            }
        }
    }
    printf("jpeg headers: %ld parsed, %ld refused; host scan decodes: %ld done, %ld refused; fonts: %ld loaded, %ld refused; no sanitizer report\n",
           jpeg_ok, jpeg_bad, host_ok, host_bad, font_ok, font_bad);
    return 0;
}
