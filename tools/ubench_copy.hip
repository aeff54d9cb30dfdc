// ubench_copy.hip -- streaming ceilings on this MI355X: what a plain copy / read / write reaches,
// and what the band kernel's tile-shaped copy reaches.  hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_copy_stride(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    for (; i + 3 * step < n; i += 4 * step) {
        uint4 a = s[i], b = s[i + step], c = s[i + 2 * step], e = s[i + 3 * step];
        d[i] = a; d[i + step] = b; d[i + 2 * step] = c; d[i + 3 * step] = e;
    }
    for (; i < n; i += step) d[i] = s[i];
}

// one block per contiguous tile of TILE bytes; each thread loads U chunks first, then stores
template <int U, bool LDS>
__global__ __launch_bounds__(256) void k_copy_tile(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    extern __shared__ uint4 sm[];
    const size_t base = (size_t)blockIdx.x * (256 * U);
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { size_t i = base + u * 256 + threadIdx.x; if (i < n) v[u] = s[i]; }
#pragma unroll
    for (int u = 0; u < U; u++) {
        size_t i = base + u * 256 + threadIdx.x;
        if (LDS) sm[u * 256 + threadIdx.x] = v[u];
        if (i < n) d[i] = v[u];
    }
}

// tile of U*SUB chunks per thread handled as SUB rounds of (U loads, U stores); XCD-contiguous order
template <int U, int SUB, bool LDS, bool SWZ>
__global__ __launch_bounds__(256) void k_copy_rounds(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    extern __shared__ uint4 sm[];
    unsigned bid = blockIdx.x;
    if (SWZ) { const unsigned per = gridDim.x >> 3; if (bid < (per << 3)) bid = (bid & 7) * per + (bid >> 3); }
    const size_t base = (size_t)bid * (256 * U * SUB);
#pragma unroll
    for (int r = 0; r < SUB; r++) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { size_t i = base + (r * U + u) * 256 + threadIdx.x; if (i < n) v[u] = s[i]; }
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t i = base + (r * U + u) * 256 + threadIdx.x;
            if (LDS) sm[(r * U + u) * 256 + threadIdx.x] = v[u];
            if (i < n) d[i] = v[u];
        }
    }
}

__global__ __launch_bounds__(256) void k_read(const uint4 *__restrict__ s, uint32_t *out, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    uint32_t acc = 0;
    for (; i + 3 * step < n; i += 4 * step) {
        uint4 a = s[i], b = s[i + step], c = s[i + 2 * step], e = s[i + 3 * step];
        acc += a.x ^ b.y ^ c.z ^ e.w;
    }
    if (acc == 0x12345678) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_write(uint4 *__restrict__ d, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    const uint4 v = make_uint4(1, 2, 3, 4);
    for (; i < n; i += step) d[i] = v;
}

template <typename F>
void timeit(const char *name, double bytes, F f)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    float best = 1e9, sum = 0;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; sum += ms;
    }
    printf("%-34s best %.3f ms  avg %.3f ms  -> %.0f GB/s (best)\n", name, best, sum / 5, bytes / best / 1e6);
}

int main()
{
    const size_t bytes = (size_t)8 << 30, n = bytes / 16;
    uint4 *s, *d; uint32_t *o;
    CK(hipMalloc(&s, bytes)); CK(hipMalloc(&d, bytes)); CK(hipMalloc(&o, 4));
    CK(hipMemset(s, 1, bytes)); CK(hipMemset(d, 2, bytes));
    for (int blocks : {2048, 8192, 65536})
        timeit(blocks == 2048 ? "copy grid-stride 2048 blk" : blocks == 8192 ? "copy grid-stride 8192 blk" : "copy grid-stride 65536 blk",
               2.0 * bytes, [&] { k_copy_stride<<<blocks, 256>>>(s, d, n); });
    timeit("copy tile U=4 (16 KB/blk)", 2.0 * bytes, [&] { k_copy_tile<4, false><<<(unsigned)((n + 1023) / 1024), 256>>>(s, d, n); });
    timeit("copy tile U=8 (32 KB/blk)", 2.0 * bytes, [&] { k_copy_tile<8, false><<<(unsigned)((n + 2047) / 2048), 256>>>(s, d, n); });
    timeit("copy tile U=16 (64 KB/blk)", 2.0 * bytes, [&] { k_copy_tile<16, false><<<(unsigned)((n + 4095) / 4096), 256>>>(s, d, n); });
    timeit("copy tile U=8 via LDS 32 KB", 2.0 * bytes, [&] { k_copy_tile<8, true><<<(unsigned)((n + 2047) / 2048), 256, 32768>>>(s, d, n); });
    CK(hipFuncSetAttribute((const void *)k_copy_tile<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    timeit("copy tile U=16 via LDS 64 KB", 2.0 * bytes, [&] { k_copy_tile<16, true><<<(unsigned)((n + 4095) / 4096), 256, 65536>>>(s, d, n); });
#define ROUNDS(U, SUB, LDS, SWZ) do { \
        if (LDS) CK(hipFuncSetAttribute((const void *)k_copy_rounds<U, SUB, LDS, SWZ>, hipFuncAttributeMaxDynamicSharedMemorySize, U * SUB * 4096)); \
        char nm[96]; snprintf(nm, sizeof nm, "rounds U=%d SUB=%d (%d KB/blk) lds=%d swz=%d", U, SUB, U * SUB * 4, (int)LDS, (int)SWZ); \
        timeit(nm, 2.0 * bytes, [&] { k_copy_rounds<U, SUB, LDS, SWZ><<<(unsigned)((n + 256 * U * SUB - 1) / (256 * U * SUB)), 256, LDS ? U * SUB * 4096 : 0>>>(s, d, n); }); } while (0)
    ROUNDS(2, 1, false, false); ROUNDS(4, 1, false, false); ROUNDS(6, 1, false, false); ROUNDS(8, 1, false, false);
    ROUNDS(2, 1, true, false); ROUNDS(4, 1, true, false); ROUNDS(6, 1, true, false); ROUNDS(8, 1, true, false);
    ROUNDS(4, 2, false, false); ROUNDS(4, 4, false, false); ROUNDS(4, 2, true, false); ROUNDS(4, 4, true, false); ROUNDS(2, 8, true, false);
    ROUNDS(4, 1, false, true); ROUNDS(4, 1, true, true); ROUNDS(4, 4, true, true); ROUNDS(8, 1, true, true);
    timeit("read only grid-stride 8192 blk", 1.0 * bytes, [&] { k_read<<<8192, 256>>>(s, o, n); });
    timeit("write only grid-stride 8192 blk", 1.0 * bytes, [&] { k_write<<<8192, 256>>>(d, n); });
    timeit("hipMemcpyAsync D2D", 2.0 * bytes, [&] { CK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0)); });
    return 0;
}
