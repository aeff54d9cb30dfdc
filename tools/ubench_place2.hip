// ubench_place2.hip -- the same copy as ubench_place.hip, source and destination carved out of ONE allocation at a chosen distance:
// is the run-to-run spread a matter of the RELATIVE placement of the two streams (then an arena with the right offsets fixes it), or of
// where the pages themselves land?  hipcc -O3 --offload-arch=gfx950
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
    const size_t pads[] = {0, 4096, 65536, 1 << 20, 3 << 20, 17 << 20, 0, 1 << 20};
    for (int rep = 0; rep < 3; rep++) {
        char *arena; CK(hipMalloc(&arena, 2 * bytes + ((size_t)64 << 20)));
        CK(hipMemset(arena, 1, 2 * bytes + ((size_t)64 << 20)));
        printf("arena %d at %p:", rep, (void *)arena);
        for (size_t pad : pads) {
            uint4 *s = (uint4 *)arena, *d = (uint4 *)(arena + bytes + pad);
            k_copy<<<8192, 256>>>(s, d, n); CK(hipDeviceSynchronize());
            float best = 1e9;
            for (int r = 0; r < 4; r++) {
                CK(hipEventRecord(e0)); k_copy<<<8192, 256>>>(s, d, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
            }
            printf("  pad %zuK %.0f", pad >> 10, 2.0 * bytes / best / 1e6);
        }
        printf("  GB/s\n");
        if (rep == 1) CK(hipFree(arena));     // the third arena may reuse the second one's range
    }
    return 0;
}
