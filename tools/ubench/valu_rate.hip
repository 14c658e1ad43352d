// Micro-benchmark: issue rate of the integer VALU instructions the recon / filter kernels are made of, per SIMD, at 1, 2
// and 4 waves per SIMD (wave64 on gfx950).  16 independent chains per lane, inline asm so nothing is folded away.
#include <hip/hip_runtime.h>
#include <cstdio>
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
                if (OP == 0) OP3("v_add_u32");
                if (OP == 1) OP4("v_med3_i32");
                if (OP == 2) OP3("v_mul_i32_i24");
                if (OP == 3) OP3("v_mul_lo_u32");
                if (OP == 4) OP4("v_perm_b32");
                if (OP == 5) OP4("v_bfe_u32");
                if (OP == 6) asm volatile("v_add_u32_dpp %0, %1, %2 row_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(a[i]), "v"(c));
                if (OP == 7) OP3("v_pk_add_i16");
                if (OP == 8) OP3("v_pk_max_i16");
                if (OP == 9) OP4("v_lshl_or_b32");
                if (OP == 10) OP3("v_and_b32");
                if (OP == 11) OP3("v_ashrrev_i32");
                if (OP == 12) OP4("v_mad_i32_i24");
                if (OP == 13) OP3("v_max_i32");
                if (OP == 14) OP4("v_add3_u32");
                if (OP == 15) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(c));
                if (OP == 16) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0" : "+v"(a[i]) : "v"(c));
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
    printf("%-16s", name);
    for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu *= 2) {
        float ms;
        hipLaunchKernelGGL(k<OP>, dim3(256 * wg_per_cu), dim3(256), 0, 0, d, 10, 3u);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256 * wg_per_cu), dim3(256), 0, 0, d, iters, 3u);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_wave = double(iters) * 64;
        // a 256-thread workgroup puts one wave on each SIMD: wg_per_cu = waves per SIMD
        printf("  %dw/SIMD %.2f ns", wg_per_cu, ms * 1e6 / (instr_per_wave * wg_per_cu));
    }
    printf("   (per wave-instruction per SIMD)\n");
}
int main() {
    uint32_t* d; hipMalloc(&d, 256);
    run<0>("v_add_u32", d); run<1>("v_med3_i32", d); run<2>("v_mul_i32_i24", d); run<3>("v_mul_lo_u32", d);
    run<4>("v_perm_b32", d); run<5>("v_bfe_u32", d); run<6>("v_add_u32_dpp", d); run<7>("v_pk_add_i16", d);
    run<8>("v_pk_max_i16", d); run<9>("v_lshl_or_b32", d); run<10>("v_and_b32", d); run<11>("v_ashrrev_i32", d);
    run<12>("v_mad_i32_i24", d); run<13>("v_max_i32", d); run<14>("v_add3_u32", d); run<15>("v_cndmask_b32", d); run<16>("v_mov_b32_sdwa", d);
    return 0;
}
