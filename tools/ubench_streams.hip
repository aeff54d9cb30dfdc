// ubench_streams.hip -- does HBM write (and copy) throughput depend on HOW MANY separate sequential streams the chip writes at a time?
// Grid-stride = the whole grid writes one moving window; per-block regions = G separate streams, each advancing 4 KB per iteration
// (what the persistent band kernels do: every workgroup writes the rows of "its" frames).  hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void k_write_window(uint4 *__restrict__ d, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    const uint4 v = make_uint4(1, 2, 3, 4);
    for (; i < n; i += step) d[i] = v;
}
__global__ __launch_bounds__(256) void k_write_streams(uint4 *__restrict__ d, size_t n)
{
    const size_t per = n / gridDim.x;
    uint4 *p = d + (size_t)blockIdx.x * per;
    const uint4 v = make_uint4(1, 2, 3, 4);
    for (size_t i = threadIdx.x; i < per; i += 256) p[i] = v;
}
__global__ __launch_bounds__(256) void k_copy_window(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    for (; i + 3 * step < n; i += 4 * step) {
        uint4 a = s[i], b = s[i + step], c = s[i + 2 * step], e = s[i + 3 * step];
        d[i] = a; d[i + step] = b; d[i + 2 * step] = c; d[i + 3 * step] = e;
    }
    for (; i < n; i += step) d[i] = s[i];
}
__global__ __launch_bounds__(256) void k_copy_streams(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    const size_t per = n / gridDim.x;
    const uint4 *ps = s + (size_t)blockIdx.x * per;
    uint4 *pd = d + (size_t)blockIdx.x * per;
    size_t i = threadIdx.x;
    for (; i + 768 < per; i += 1024) {
        uint4 a = ps[i], b = ps[i + 256], c = ps[i + 512], e = ps[i + 768];
        pd[i] = a; pd[i + 256] = b; pd[i + 512] = c; pd[i + 768] = e;
    }
    for (; i < per; i += 256) pd[i] = ps[i];
}
template <typename F> float best_ms(F f)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < 4; r++) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; }
    return best;
}
int main()
{
    const size_t bytes = (size_t)8 << 30, n = bytes / 16;
    uint4 *s, *d; CK(hipMalloc(&s, bytes)); CK(hipMalloc(&d, bytes)); CK(hipMemset(s, 1, bytes)); CK(hipMemset(d, 2, bytes));
    for (int G : {512, 1024, 2048, 8192}) {
        const float ww = best_ms([&] { k_write_window<<<G, 256>>>(d, n); }), ws = best_ms([&] { k_write_streams<<<G, 256>>>(d, n); });
        const float cw = best_ms([&] { k_copy_window<<<G, 256>>>(s, d, n); }), cs = best_ms([&] { k_copy_streams<<<G, 256>>>(s, d, n); });
        printf("G=%5d  write: one window %.0f GB/s, %d streams %.0f GB/s   copy: one window %.0f GB/s, %d streams %.0f GB/s\n", G, bytes / ww / 1e6, G,
               bytes / ws / 1e6, 2.0 * bytes / cw / 1e6, G, 2.0 * bytes / cs / 1e6);
    }
    return 0;
}
