// ubench_place.hip -- does a plain streaming copy depend on where its buffers land?  Several (source, destination) pairs of 8 GiB, each a new
// hipMalloc (the earlier ones kept), the same grid-stride copy on each.  hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void k_copy(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    for (; i + 3 * step < n; i += 4 * step) {
        uint4 a = s[i], b = s[i + step], c = s[i + 2 * step], e = s[i + 3 * step];
        d[i] = a; d[i + step] = b; d[i + 2 * step] = c; d[i + 3 * step] = e;
    }
    for (; i < n; i += step) d[i] = s[i];
}
int main()
{
    const size_t bytes = (size_t)8 << 30, n = bytes / 16;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int set = 0; set < 8; set++) {
        uint4 *s, *d, *pad;
        CK(hipMalloc(&s, bytes)); CK(hipMalloc(&d, bytes)); CK(hipMalloc(&pad, (size_t)(3 + 2 * set) << 20));
        CK(hipMemset(s, 1, bytes)); CK(hipMemset(d, 2, bytes));
        k_copy<<<8192, 256>>>(s, d, n); CK(hipDeviceSynchronize());
        float best = 1e9, worst = 0;
        for (int r = 0; r < 6; r++) {
            CK(hipEventRecord(e0)); k_copy<<<8192, 256>>>(s, d, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; worst = ms > worst ? ms : worst;
        }
        printf("set %d  src %p dst %p  copy 8 GiB: best %.3f ms worst %.3f ms -> %.0f GB/s\n", set, (void *)s, (void *)d, best, worst, 2.0 * bytes / best / 1e6);
        if (set >= 5) { CK(hipFree(s)); CK(hipFree(d)); }     // (HBM holds 288 GB; the last sets reuse freed ranges)
    }
    return 0;
}
