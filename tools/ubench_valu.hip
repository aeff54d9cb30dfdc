// ubench_valu.hip -- per-instruction issue cost on gfx950 for the ops the bilinear tap loop uses.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o gpurun_out/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int OP>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a, uint32_t b)
{
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    uint32_t u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { REP8(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));) }
        if (OP == 1) { REP8(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));) }
        if (OP == 2) { REP8(asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));) }
        if (OP == 3) { REP8(asm volatile("v_cvt_f64_u32 %0, %8\n v_cvt_f64_u32 %1, %9\n v_cvt_f64_u32 %2, %10\n v_cvt_f64_u32 %3, %11\n v_cvt_f64_u32 %4, %12\n v_cvt_f64_u32 %5, %13\n v_cvt_f64_u32 %6, %14\n v_cvt_f64_u32 %7, %15" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(u0), "v"(u1), "v"(u2), "v"(u3), "v"(u4), "v"(u5), "v"(u6), "v"(u7));) }
        if (OP == 4) { REP8(asm volatile("v_cvt_u32_f64 %0, %8\n v_cvt_u32_f64 %1, %9\n v_cvt_u32_f64 %2, %10\n v_cvt_u32_f64 %3, %11\n v_cvt_u32_f64 %4, %12\n v_cvt_u32_f64 %5, %13\n v_cvt_u32_f64 %6, %14\n v_cvt_u32_f64 %7, %15" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));) }
        if (OP == 5) { REP8(asm volatile("v_perm_b32 %0, %0, %1, %8\n v_perm_b32 %1, %1, %2, %8\n v_perm_b32 %2, %2, %3, %8\n v_perm_b32 %3, %3, %4, %8\n v_perm_b32 %4, %4, %5, %8\n v_perm_b32 %5, %5, %6, %8\n v_perm_b32 %6, %6, %7, %8\n v_perm_b32 %7, %7, %0, %8" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(b));) }
        if (OP == 6) { REP8(asm volatile("v_mad_u32_u24 %0, %0, %8, %1\n v_mad_u32_u24 %1, %1, %8, %2\n v_mad_u32_u24 %2, %2, %8, %3\n v_mad_u32_u24 %3, %3, %8, %4\n v_mad_u32_u24 %4, %4, %8, %5\n v_mad_u32_u24 %5, %5, %8, %6\n v_mad_u32_u24 %6, %6, %8, %7\n v_mad_u32_u24 %7, %7, %8, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(b));) }
        if (OP == 7) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(b));) }
        if (OP == 8) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(u0) : "vcc");) }
        if (OP == 9) { REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(b));) }
        if (OP == 10) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));) }
        if (OP == 11) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));) }
        if (OP == 12) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(b));) }
        if (OP == 13) { REP8(asm volatile("v_cvt_f32_ubyte1 %0, %1\n v_cvt_f32_ubyte1 %1, %2\n v_cvt_f32_ubyte1 %2, %3\n v_cvt_f32_ubyte1 %3, %4\n v_cvt_f32_ubyte1 %4, %5\n v_cvt_f32_ubyte1 %5, %6\n v_cvt_f32_ubyte1 %6, %7\n v_cvt_f32_ubyte1 %7, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));) }
        if (OP == 14) { REP8(asm volatile("v_cvt_u32_f32 %0, %1\n v_cvt_u32_f32 %1, %2\n v_cvt_u32_f32 %2, %3\n v_cvt_u32_f32 %3, %4\n v_cvt_u32_f32 %4, %5\n v_cvt_u32_f32 %5, %6\n v_cvt_u32_f32 %6, %7\n v_cvt_u32_f32 %7, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));) }
        if (OP == 15) { REP8(asm volatile("v_lshl_add_u32 %0, %0, 8, %1\n v_lshl_add_u32 %1, %1, 8, %2\n v_lshl_add_u32 %2, %2, 8, %3\n v_lshl_add_u32 %3, %3, 8, %4\n v_lshl_add_u32 %4, %4, 8, %5\n v_lshl_add_u32 %5, %5, 8, %6\n v_lshl_add_u32 %6, %6, 8, %7\n v_lshl_add_u32 %7, %7, 8, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));) }
        if (OP == 16) { REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(b));) }
        if (OP == 17) { REP8(asm volatile("v_dot4_u32_u8 %0, %1, %8, %0\n v_dot4_u32_u8 %1, %2, %8, %1\n v_dot4_u32_u8 %2, %3, %8, %2\n v_dot4_u32_u8 %3, %4, %8, %3\n v_dot4_u32_u8 %4, %5, %8, %4\n v_dot4_u32_u8 %5, %6, %8, %5\n v_dot4_u32_u8 %6, %7, %8, %6\n v_dot4_u32_u8 %7, %0, %8, %7" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(b));) }
        if (OP == 18) { REP8(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(b));) }
        if (OP == 19) { REP8(asm volatile("v_mad_u32_u16 %0, %0, %8, %1\n v_mad_u32_u16 %1, %1, %8, %2\n v_mad_u32_u16 %2, %2, %8, %3\n v_mad_u32_u16 %3, %3, %8, %4\n v_mad_u32_u16 %4, %4, %8, %5\n v_mad_u32_u16 %5, %5, %8, %6\n v_mad_u32_u16 %6, %6, %8, %7\n v_mad_u32_u16 %7, %7, %8, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(b));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (double)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7);
}

template <int OP>
void run(const char *name, double *out, int waves_per_simd)
{
    const int iters = 8000;
    const int blocks = 256 * waves_per_simd;  // 256-thread block = 4 waves = one per SIMD of a CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 10, 1.0000001, 0x0c0c0100u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, iters, 1.0000001, 0x0c0c0100u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // each SIMD ran waves_per_simd waves x iters x 64 wave-instructions
    double ns_per_inst = ms * 1e6 / ((double)waves_per_simd * iters * 64);
    printf("%-16s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instruction per SIMD (= %.1f cycles at 2.4 GHz)\n", name,
           waves_per_simd, ms, ns_per_inst, ns_per_inst * 2.4);
}

int main()
{
    double *out; hipMalloc(&out, sizeof(double) * 256 * 256 * 8);
    for (int w : {2, 8}) {
        run<0>("v_mul_f64", out, w);
        run<1>("v_add_f64", out, w);
        run<2>("v_fma_f64", out, w);
        run<3>("v_cvt_f64_u32", out, w);
        run<4>("v_cvt_u32_f64", out, w);
        run<5>("v_perm_b32", out, w);
        run<6>("v_mad_u32_u24", out, w);
        run<7>("v_mul_lo_u32", out, w);
        run<8>("v_mad_u64_u32", out, w);
        run<9>("v_add_u32", out, w);
        run<10>("v_pk_fma_f32", out, w);
        run<11>("v_pk_mul_f32", out, w);
        run<12>("v_fma_f32", out, w);
        run<13>("v_cvt_f32_ubyte1", out, w);
        run<14>("v_cvt_u32_f32", out, w);
        run<15>("v_lshl_add_u32", out, w);
        run<16>("v_mul_f32", out, w);
        run<17>("v_dot4_u32_u8", out, w);
        run<18>("v_and_b32", out, w);
        run<19>("v_mad_u32_u16", out, w);
    }
    return 0;
}
