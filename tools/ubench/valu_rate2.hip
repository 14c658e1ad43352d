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
                if (OP == 0) OP3("v_sub_u32");
                if (OP == 1) OP3("v_or_b32");
                if (OP == 2) OP3("v_xor_b32");
                if (OP == 3) OP3("v_lshlrev_b32");
                if (OP == 4) OP3("v_lshrrev_b32");
                if (OP == 5) OP3("v_min_i32");
                if (OP == 6) OP3("v_max_u32");
                if (OP == 7) OP3("v_mul_u32_u24");
                if (OP == 8) OP3("v_pk_sub_i16");
                if (OP == 9) OP3("v_pk_min_i16");
                if (OP == 10) OP3("v_pk_ashrrev_i16");
                if (OP == 11) OP3("v_pk_lshlrev_b16");
                if (OP == 12) OP3("v_pk_mul_lo_u16");
                if (OP == 13) OP3("v_add_u16");
                if (OP == 14) OP3("v_sub_u16");
                if (OP == 15) OP3("v_max_i16");
                if (OP == 16) OP3("v_lshlrev_b16");
                if (OP == 17) OP3("v_ashrrev_i16");
                if (OP == 18) OP3("v_mul_lo_u16");
                if (OP == 19) OP3("v_add_f32");
                if (OP == 20) OP3("v_mul_f32");
                if (OP == 21) OP3("v_max_f32");
                if (OP == 22) OP3("v_subrev_u32");
                if (OP == 23) OP4("v_and_or_b32");
                if (OP == 24) OP4("v_bfi_b32");
                if (OP == 25) OP4("v_sad_u8");
                if (OP == 26) OP4("v_alignbit_b32");
                if (OP == 27) OP4("v_mad_u32_u24");
                if (OP == 28) OP4("v_lshl_add_u32");
                if (OP == 29) OP4("v_add_lshl_u32");
                if (OP == 30) OP4("v_xad_u32");
                if (OP == 31) OP4("v_or3_b32");
                if (OP == 32) OP4("v_min3_i32");
                if (OP == 33) OP4("v_max3_i32");
                if (OP == 34) OP4("v_pk_mad_i16");
                if (OP == 35) OP4("v_bfe_i32");
                if (OP == 36) OP4("v_med3_i16");
                if (OP == 37) OP4("v_mad_i16");
                if (OP == 38) OP4("v_fma_f32");
                if (OP == 39) OP4("v_sad_u16");
                if (OP == 40) OP4("v_msad_u8");
                if (OP == 41) OP4("v_alignbyte_b32");
                if (OP == 42) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(c));
                if (OP == 43) asm volatile("v_cmp_lt_i32 vcc, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(c) : "vcc");
                if (OP == 44) asm volatile("v_cmp_lt_i32 vcc, %0, %1" : : "v"(a[i]), "v"(c) : "vcc");
                if (OP == 45) asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0" : "=v"(a[i]) : "v"(a[i]), "v"(c));
                if (OP == 46) asm volatile("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0" : "=v"(a[i]) : "v"(a[i]), "v"(c));
                if (OP == 47) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(a[i]));
                if (OP == 48) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[i]) : "v"(a[i]));
                if (OP == 49) asm volatile("v_add_u32 %0, 7, %1" : "=v"(a[i]) : "v"(a[i]));
                if (OP == 50) asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(a[i]) : "v"(a[i]));
                if (OP == 51) asm volatile("v_and_b32 %0, 0xff00ff, %1" : "=v"(a[i]) : "v"(a[i]));
                if (OP == 52) asm volatile("v_med3_i32 %0, %1, 0, %2" : "=v"(a[i]) : "v"(a[i]), "v"(c));
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
    run<0>("v_sub_u32", d);
    run<1>("v_or_b32", d);
    run<2>("v_xor_b32", d);
    run<3>("v_lshlrev_b32", d);
    run<4>("v_lshrrev_b32", d);
    run<5>("v_min_i32", d);
    run<6>("v_max_u32", d);
    run<7>("v_mul_u32_u24", d);
    run<8>("v_pk_sub_i16", d);
    run<9>("v_pk_min_i16", d);
    run<10>("v_pk_ashrrev_i16", d);
    run<11>("v_pk_lshlrev_b16", d);
    run<12>("v_pk_mul_lo_u16", d);
    run<13>("v_add_u16", d);
    run<14>("v_sub_u16", d);
    run<15>("v_max_i16", d);
    run<16>("v_lshlrev_b16", d);
    run<17>("v_ashrrev_i16", d);
    run<18>("v_mul_lo_u16", d);
    run<19>("v_add_f32", d);
    run<20>("v_mul_f32", d);
    run<21>("v_max_f32", d);
    run<22>("v_subrev_u32", d);
    run<23>("v_and_or_b32", d);
    run<24>("v_bfi_b32", d);
    run<25>("v_sad_u8", d);
    run<26>("v_alignbit_b32", d);
    run<27>("v_mad_u32_u24", d);
    run<28>("v_lshl_add_u32", d);
    run<29>("v_add_lshl_u32", d);
    run<30>("v_xad_u32", d);
    run<31>("v_or3_b32", d);
    run<32>("v_min3_i32", d);
    run<33>("v_max3_i32", d);
    run<34>("v_pk_mad_i16", d);
    run<35>("v_bfe_i32", d);
    run<36>("v_med3_i16", d);
    run<37>("v_mad_i16", d);
    run<38>("v_fma_f32", d);
    run<39>("v_sad_u16", d);
    run<40>("v_msad_u8", d);
    run<41>("v_alignbyte_b32", d);
    run<42>("v_mov_b32", d);
    run<43>("v_cmp+cndmask", d);
    run<44>("v_cmp_lt_i32", d);
    run<45>("v_add_u32_sdwa", d);
    run<46>("v_sub_u32_sdwa_w", d);
    run<47>("v_mov_b32_dpp", d);
    run<48>("v_cvt_f32_ubyte1", d);
    run<49>("v_add_u32 const", d);
    run<50>("v_lshlrev const", d);
    run<51>("v_and_b32 lit", d);
    run<52>("v_med3 const", d);
    return 0;
}
