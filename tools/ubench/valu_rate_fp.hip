// Micro-benchmark: issue rate of the fp32 / conversion VALU instructions an fp32-domain loop filter would be made of
// (all values are small integers, exact in fp32), per SIMD at 1, 2 and 4 waves per SIMD.  Same harness as valu_rate.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#define OP2(str) asm volatile(str " %0, %1" : "=v"(a[i]) : "v"(a[i]))
#define OP3(str) asm volatile(str " %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(c))
#define OP4(str) asm volatile(str " %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(c), "v"(d))
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed * (i + 1) + threadIdx.x;
    uint32_t c = seed | 1, d = seed + 77;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (OP == 0) OP3("v_sub_f32");
                if (OP == 1) OP4("v_fma_f32");
                if (OP == 2) OP2("v_floor_f32");
                if (OP == 3) OP4("v_med3_f32");
                if (OP == 4) OP4("v_max3_f32");
                if (OP == 5) asm volatile("v_max_f32 %0, |%1|, |%2|" : "=v"(a[i]) : "v"(a[i]), "v"(c));
                if (OP == 6) asm volatile("v_cmp_le_f32 vcc, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(c) : "vcc");
                if (OP == 7) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(c));
                if (OP == 8) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %2" : "=v"(a[i]) : "v"(c), "v"(a[i]));
                if (OP == 9) OP2("v_cvt_f32_ubyte0");
                if (OP == 10) OP2("v_cvt_f32_ubyte2");
                if (OP == 11) OP2("v_rndne_f32");
                if (OP == 12) OP3("v_mul_f32");
                if (OP == 13) asm volatile("v_add_f32 %0, |%1|, -%2" : "=v"(a[i]) : "v"(a[i]), "v"(c));
                if (OP == 14) OP2("v_cvt_f32_i32");
                if (OP == 15) OP2("v_cvt_i32_f32");
                if (OP == 16) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(*(uint64_t*)&a[i & ~1]) : "v"(*(uint64_t*)&a[i & ~1]), "v"(*(uint64_t*)&a[(i & ~1) ^ 2]));
                if (OP == 17) asm volatile("v_pk_fma_f32 %0, %1, %2, %1" : "=v"(*(uint64_t*)&a[i & ~1]) : "v"(*(uint64_t*)&a[i & ~1]), "v"(*(uint64_t*)&a[(i & ~1) ^ 2]));
                if (OP == 18) OP3("v_min_f32");
                if (OP == 19) asm volatile("v_cmp_gt_f32 vcc, |%1|, %2\n v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(c) : "vcc");
                if (OP == 20) OP3("v_cvt_pk_i16_i32");
                if (OP == 21) OP2("v_sat_pk_u8_i16");
                if (OP == 22) OP3("v_min_i16");
                if (OP == 23) OP4("v_mad_u16");
                if (OP == 24) asm volatile("v_cmp_gt_i16 vcc, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(c) : "vcc");
                if (OP == 25) OP3("v_ashrrev_i32");
                if (OP == 26) OP4("v_med3_i32");
                if (OP == 27) OP4("v_sad_u32");
                if (OP == 28) OP3("v_add_u32");
                if (OP == 29) OP2("v_trunc_f32");
                if (OP == 30) OP2("v_fract_f32");
                if (OP == 31) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
                if (OP == 32) OP3("v_ldexp_f32");
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s ^= a[i];
    if (s == 0x12345u) out[0] = s;
}
template <int OP> void run(const char* name, uint32_t* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    printf("%-18s", name);
    for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu *= 2) {
        float ms;
        hipLaunchKernelGGL(k<OP>, dim3(256 * wg_per_cu), dim3(256), 0, 0, d, 10, 3u);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256 * wg_per_cu), dim3(256), 0, 0, d, iters, 3u);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("  %dw/SIMD %.2f ns", wg_per_cu, ms * 1e6 / (double(iters) * 64 * wg_per_cu));
    }
    printf("   (per wave-instruction per SIMD)\n");
}
int main() {
    uint32_t* d; hipMalloc(&d, 256);
    run<28>("v_add_u32", d); run<0>("v_sub_f32", d); run<13>("v_add_f32 |a|,-b", d); run<12>("v_mul_f32", d); run<1>("v_fma_f32", d); run<31>("v_fmac_f32", d);
    run<2>("v_floor_f32", d); run<29>("v_trunc_f32", d); run<30>("v_fract_f32", d); run<11>("v_rndne_f32", d); run<32>("v_ldexp_f32", d);
    run<3>("v_med3_f32", d); run<4>("v_max3_f32", d); run<5>("v_max_f32 |a|,|b|", d); run<18>("v_min_f32", d);
    run<6>("cmp_le_f32+cndmask", d); run<19>("cmp_gt |a|+cndmsk", d); run<7>("v_cndmask_b32", d);
    run<8>("v_cvt_pk_u8_f32", d); run<9>("v_cvt_f32_ubyte0", d); run<10>("v_cvt_f32_ubyte2", d); run<14>("v_cvt_f32_i32", d); run<15>("v_cvt_i32_f32", d);
    run<16>("v_pk_add_f32", d); run<17>("v_pk_fma_f32", d);
    run<20>("v_cvt_pk_i16_i32", d); run<21>("v_sat_pk_u8_i16", d); run<22>("v_min_i16", d); run<23>("v_mad_u16", d); run<24>("cmp_gt_i16+cndmask", d);
    run<25>("v_ashrrev_i32", d); run<26>("v_med3_i32", d); run<27>("v_sad_u32", d);
    return 0;
}
