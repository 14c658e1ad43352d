// Shared device-side helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../../include/webp_amd.h"

namespace wpk {

constexpr int kWave = 64;             // CDNA wavefront
constexpr int kBandRows = 4;          // macroblock rows handled by one wavefront (16 lanes each)
constexpr int kWavesPerWG = 8;        // wavefronts per workgroup in the wavefront-scheduled kernels
constexpr uint32_t kSeqShift = 12;    // progress word = (band sequence << 12) | macroblocks finished
constexpr uint32_t kBandDone = 0xFFF;

// Macroblock-tiled intermediate layout (workspace between recon, loop filter and the finishing pass).
// One tile = 384 contiguous bytes = three 128-byte lines: 16 luma rows of 16 bytes, 8 Cb rows of 8, 8 Cr rows of 8.
//   R tiles (recon output):        tile (x, y) of a frame at index mb_base + y*mbw + x, holds macroblock (x, y).
//   S tiles (loop-filter output):  grid (mbw+1) x (mbh+1) at index stile_base + ty*(mbw+1) + tx; tile (tx, ty) holds
//                                  luma rows 16ty-8..16ty+7 / cols 16tx-8..16tx+7 and chroma rows 8ty-4..8ty+3 / cols
//                                  8tx-4..8tx+3: the region that is FINAL once macroblock (tx, ty) has been filtered.
// Every wavefront-scheduled kernel reads and writes whole tiles (full cache lines); only the streaming finishing
// pass touches the row-major planes / ARGB, with long contiguous rows.
constexpr int kTileBytes = 384;
constexpr int kTileCb = 256, kTileCr = 320;
constexpr int kFiltShift = 8;         // luma shift of the S grid (chroma: 4)

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }   // -> v_med3_i32

// Cross-wavefront hand-over inside a workgroup (all its wavefronts run on one CU and share its L1):
// the producer finishes its global stores, executes a workgroup-scope release FENCE (all address
// spaces: s_waitcnt vmcnt(0) lgkmcnt(0)) and then stores the LDS progress word; the consumer polls the
// word and executes a workgroup-scope acquire fence before touching the data.  Fences are explicit
// because an atomic on an LDS address alone only orders LDS accesses.
__device__ __forceinline__ void wait_progress(const uint32_t* word, uint32_t need) {
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void publish_progress(uint32_t* word, uint32_t value) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Forces the wait for previously issued loads to happen HERE (loads and stores share one in-order
// counter on gfx9-class hardware: waiting for a load that was issued before a batch of stores, but
// only after those stores have been issued, would also wait for the stores' acknowledgements).
__device__ __forceinline__ void settle(uint32_t& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void settle(uint4& v) { settle(v.x); settle(v.y); settle(v.z); settle(v.w); }
__device__ __forceinline__ void settle(uint2& v) { settle(v.x); settle(v.y); }

// Non-temporal (streaming) global accesses for data that is used exactly once.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
#ifdef WP_NO_NT
__device__ __forceinline__ uint4 ld_nt(const uint4* p) { return *p; }
__device__ __forceinline__ uint2 ld_nt(const uint2* p) { return *p; }
__device__ __forceinline__ void st_nt(uint4* p, uint4 v) { *p = v; }
__device__ __forceinline__ void st_nt(uint2* p, uint2 v) { *p = v; }
#else
__device__ __forceinline__ uint4 ld_nt(const uint4* p) { const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p)); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint2 ld_nt(const uint2* p) { const u32x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(p)); return make_uint2(v.x, v.y); }
__device__ __forceinline__ void st_nt(uint4* p, uint4 v) { const u32x4_t w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<u32x4_t*>(p)); }
__device__ __forceinline__ void st_nt(uint2* p, uint2 v) { const u32x2_t w = {v.x, v.y}; __builtin_nontemporal_store(w, reinterpret_cast<u32x2_t*>(p)); }
#endif


// Packed 16-bit helpers shared by the lossy and the lossless kernels.
typedef short short2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t as_u32(short2_t v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ short2_t as_s2(uint32_t v) { return __builtin_bit_cast(short2_t, v); }
// clamp two int16 pairs (bytes 0,2 and bytes 1,3 of a pixel group) to 0..255 and interleave them into four bytes
__device__ __forceinline__ uint32_t clip_pack(short2_t p02, short2_t p13) {
    uint32_t l8, h8;                                   // v_sat_pk_u8_i16: two int16 -> two bytes (low half), clamped to 0..255
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(l8) : "v"(as_u32(p02)));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(h8) : "v"(as_u32(p13)));
    return __builtin_amdgcn_perm(h8, l8, 0x05010400u);  // bytes: l8[0], h8[0], l8[1], h8[1]
}

// The same for pairs in pixel order -- (pixels 0, 1) and (pixels 2, 3): the second saturating pack writes the upper half of
// the result directly (SDWA destination select), so no byte shuffle is needed afterwards.
__device__ __forceinline__ uint32_t clip_pack_seq(short2_t p01, short2_t p23) {
    uint32_t r;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(as_u32(p01)));
    asm("v_sat_pk_u8_i16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(r) : "v"(as_u32(p23)));
    return r;
}

// Optional per-phase cycle profile of the wavefront kernels (tuning builds only: -DWP_PROFILE).  Wave 0 of workgroup 0
// accumulates s_memtime deltas per phase and adds them to wp_prof[] at the end of each band; tools read the symbol.
#ifdef WP_PROFILE
static __device__ unsigned long long wp_prof[32];      // one copy per translation unit
#define WP_PROF_DECL unsigned long long prof_t = __builtin_amdgcn_s_memtime(), prof_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; const bool prof_on = blockIdx.x == 0 && (threadIdx.x >> 6) == 0
#define WP_PROF_MARK(i) do { if (prof_on) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); prof_acc[i] += now_ - prof_t; prof_t = now_; } } while (0)
#define WP_PROF_FLUSH(base) do { if (prof_on && (threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 10; ++i_) atomicAdd(&wp_prof[(base) + i_], prof_acc[i_]); } } while (0)
#else
#define WP_PROF_DECL
#define WP_PROF_MARK(i)
#define WP_PROF_FLUSH(base)
#endif

}  // namespace wpk
