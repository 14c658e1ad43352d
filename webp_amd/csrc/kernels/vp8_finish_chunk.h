// Finishing pass of the lossy path, one chunk: 128 output columns x one tile row (16 luma rows) of one frame, from the
// 8 (or, for shifted tiles, 9) horizontally adjacent source tiles -- parked in LDS by the workgroup -- to the row-major
// Y/U/V planes (what ImageData.Yuv420 carries, Vp8Decoder.ensureImg geometry) and tightly packed A,R,G,B (Tools.toArgb,
// Tools.hx:6-48).  Chunks are 128 columns (3.4 KB of LDS) so that two finishing workgroups fit into the 8 KB of LDS two
// workgroups of the one-kernel reconstruction + loop filter leave on a CU: the pass is memory-bound, that kernel
// issue-bound, and callers that pipeline batches run them side by side (webp_amd/device.py LossyPipeline).
#pragma once
#include "wp_device.h"
#include "wp_convert.h"

// Everything this pass touches is used exactly once: non-temporal accesses keep it out of the way of the tiles and
// coefficient blocks that the wavefront-scheduled code re-reads through L2.
// timing-only builds: the pass without its plane stores / without its ARGB stores (how much of a pipelined step is HBM time)
#ifdef WP_KO_PLANES
#define WP_PLANE_STORE(p, v) asm volatile("" :: "v"(v))
#else
#define WP_PLANE_STORE(p, v) WP_NT_STORE(p, v)
#endif
#ifdef WP_KO_ARGB
#define WP_ARGB_STORE(p, v) asm volatile("" :: "v"(v))
#else
#define WP_ARGB_STORE(p, v) WP_NT_STORE(p, v)
#endif
#ifdef WP_FINISH_NO_NT
#define WP_NT_LOAD(p) (*(p))
#define WP_NT_STORE(p, v) (*(p) = (v))
#else
#define WP_NT_LOAD(p) __builtin_nontemporal_load(p)
#define WP_NT_STORE(p, v) __builtin_nontemporal_store((v), (p))
#endif

namespace wpk {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kFinishCols = 128;                       // output luma columns per chunk
constexpr int kFinishTiles = kFinishCols / 16 + 1;      // source tiles that can overlap them
constexpr int kFinishLdsBytes = kFinishTiles * kTileBytes;

// What a chunk reads: up to kFinishTiles source tiles, contiguous in memory.
struct FinishSrc { const uint8_t* src; int nt; };
__device__ __forceinline__ FinishSrc finish_src(const wp_vp8_frame& fr, int ty, int c0, const uint8_t* __restrict__ Rp, const uint8_t* __restrict__ Sp, bool shifted) {
    const int ntx = fr.mbw + (shifted ? 1 : 0);
    const int tx0 = c0 >> 4;                                      // (c0 + shift) >> 4: first source tile
    FinishSrc r;
    r.nt = min(shifted ? kFinishTiles : kFinishTiles - 1, ntx - tx0);
    r.src = (shifted ? Sp + size_t(fr.stile_base) * kTileBytes : Rp + size_t(fr.mb_base) * kTileBytes) + (size_t(ty) * ntx + tx0) * kTileBytes;
    return r;
}
// thread t's 16 bytes of the chunk's source tiles (zeros beyond them)
__device__ __forceinline__ u32x4 finish_fetch(const FinishSrc& fs, int t) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (t < fs.nt * (kTileBytes / 16)) v = WP_NT_LOAD(reinterpret_cast<const u32x4*>(fs.src) + t);
    return v;
}

// The chunk's tiles are in LDS (tiles4: kFinishLdsBytes): rows out.  t: thread index in the workgroup of 256 (8 row pairs x
// 32 quads); nt: source tiles present.
__device__ __forceinline__ void finish_emit(const wp_vp8_frame& fr, int ty, int c0, int t, int nt, const uint4* tiles4, bool shifted,
                                            uint8_t* Yp, uint8_t* Up, uint8_t* Vp, const uint8_t* __restrict__ Ap, uint8_t* out) {
    const int mbw = fr.mbw, mbh = fr.mbh, width = fr.width, height = fr.height;
    const int shift = shifted ? kFiltShift : 0;
    const uint32_t* tiles32 = reinterpret_cast<const uint32_t*>(tiles4);
    const uint16_t* tiles16 = reinterpret_cast<const uint16_t*>(tiles4);

    // Output addresses: a wave-uniform plane / slab base (scalar registers) plus a 32-bit byte offset per lane (a frame's
    // planes and its ARGB image stay below 4 GB: at most 16383 x 16383 pixels).
    const uint32_t ystride = 16u * uint32_t(mbw), cstride = 8u * uint32_t(mbw);
    uint8_t* const Y = Yp ? Yp + fr.y_off : nullptr;
    const uint8_t* const A = (Ap != nullptr && fr.has_alpha) ? Ap + fr.alpha_off : nullptr;
    uint8_t* const dst = out ? out + fr.argb_off : nullptr;

    // ---- luma rows + ARGB: thread = 4 pixels x 2 rows (one chroma sample pair: its rounded terms are computed once),
    //      half a wavefront = 128 pixels of two rows (512 contiguous bytes of ARGB per row) ----
    const int q = t & 31, px = c0 + 4 * q;
    const int k = (4 * q + shift) >> 4, c4 = (4 * q + shift) & 15;         // source tile / column inside it
    const int rp = t >> 5;                                                 // row pair inside the tile row (shift is even)
    const int py0 = 16 * ty - shift + 2 * rp;
    if (k < nt && px < int(ystride) && py0 >= 0 && py0 < 16 * mbh) {
        const int npx = min(4, width - px);
        const uint32_t lbase = uint32_t(k) * uint32_t(kTileBytes / 4) + uint32_t(c4 >> 2);        // dword index of the tile row's first luma dword for this lane
        {
            const uint32_t ya = tiles32[lbase + 8 * rp];
            const uint32_t yb = tiles32[lbase + 8 * rp + 4];
            if (Y) {
                const uint32_t yo = uint32_t(py0) * ystride + uint32_t(px);
                WP_NT_STORE(reinterpret_cast<uint32_t*>(Y + size_t(yo)), ya);
                WP_NT_STORE(reinterpret_cast<uint32_t*>(Y + size_t(yo + ystride)), yb);
            }
            if (dst && py0 < height && px < width) {
                const uint32_t ci = 2u * lbase + 4u * uint32_t(rp) - uint32_t(c4 >> 2);   // 16-bit index of the two chroma samples: k * 192 + 4 * rp + (c4 >> 2)
                const uint32_t u2 = tiles16[ci + kTileCb / 2], v2 = tiles16[ci + kTileCr / 2];
                const ChromaTerms t0 = chroma_terms(int(u2 & 0xff), int(v2 & 0xff));
                const ChromaTerms t1 = chroma_terms(int(u2 >> 8), int(v2 >> 8));
                const bool any_tie = __builtin_amdgcn_ballot_w64(t0.tie || t1.tie) != 0;
                const uint32_t pix0 = uint32_t(py0) * uint32_t(width) + uint32_t(px);
                // both rows' alpha dwords are aligned when the plane and its rows are (px is a multiple of 4)
                const bool a_dword = A != nullptr && npx == 4 && ((reinterpret_cast<uintptr_t>(A) | uint32_t(width)) & 3) == 0;
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const int py = py0 + rr;
                    if (py >= height) break;
                    const uint32_t pix = pix0 + (rr ? uint32_t(width) : 0u);
                    uint32_t a4 = 0xffffffffu;
                    if (A) {
                        if (a_dword) {
                            a4 = *reinterpret_cast<const uint32_t*>(A + size_t(pix));     // the usual case: one load for the four pixels
                        } else {
                            a4 = 0;
                            for (int j = 0; j < npx; ++j) a4 |= uint32_t(A[size_t(pix + uint32_t(j))]) << (8 * j);
                        }
                    }
                    const uint4 o = convert_row4(rr ? yb : ya, t0, t1, a4, any_tie);
                    uint8_t* p = dst + size_t(pix * 4u);
                    if (npx == 4 && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
                        const u32x4 ov4 = {o.x, o.y, o.z, o.w};
                        WP_NT_STORE(reinterpret_cast<u32x4*>(p), ov4);
                    } else {
                        const uint32_t ov[4] = {o.x, o.y, o.z, o.w};
                        for (int j = 0; j < npx; ++j) reinterpret_cast<uint32_t*>(p)[j] = ov[j];
                    }
                }
            }
        }
    }
    // ---- chroma rows: 2 planes x 8 rows x (8 tiles x 2 dwords) ----
    if (Up && Vp) {
        {
            const int plane = t >> 7, row = (t >> 4) & 7, dw = t & 15;
            const int cy = 8 * ty - (shift >> 1) + row, cx = (c0 >> 1) + 4 * dw;
            const int kk = (4 * dw + (shift >> 1)) >> 3, half = ((4 * dw + (shift >> 1)) >> 2) & 1;
            if (kk < nt && cy >= 0 && cy < 8 * mbh && cx < int(cstride)) {
                uint8_t* P = plane ? Vp + fr.v_off : Up + fr.u_off;
                WP_NT_STORE(reinterpret_cast<uint32_t*>(P + size_t(uint32_t(cy) * cstride + uint32_t(cx))), tiles32[kk * (kTileBytes / 4) + (kTileCb / 4) + 16 * plane + 2 * row + half]);
            }
        }
    }
}

// The same for a chunk that lies wholly inside the image with every output requested and no alpha plane (finish_is_interior:
// 14 of 15 chunks of 66 of 69 tile rows of a 1080p frame): no per-lane predicate, no partial stores, row and chunk bases in
// scalar registers and 24-bit multiplies for the lane's share of the addresses.  The pass overlaps the issue-bound
// reconstruction + loop filter kernel, so its own instruction count is step time: ~100 instead of ~160 VALU per thread.
__device__ __forceinline__ bool finish_is_interior(const wp_vp8_frame& fr, int ty, int c0, bool shifted, const uint8_t* Yp, const uint8_t* Up,
                                                   const uint8_t* Vp, const uint8_t* Ap, const uint8_t* out) {
    const int row0 = 16 * ty - (shifted ? kFiltShift : 0);
    const bool alpha_ok = !(Ap != nullptr && fr.has_alpha) || ((reinterpret_cast<uintptr_t>(Ap) + fr.alpha_off) & 3) == 0;      // (rows are aligned with it: width % 4 == 0 below)
    return shifted && Yp != nullptr && Up != nullptr && Vp != nullptr && out != nullptr && alpha_ok && row0 >= 0 &&
           row0 + 16 <= fr.height && c0 + kFinishCols <= fr.width && (fr.width & 3) == 0 &&
           ((reinterpret_cast<uintptr_t>(out) + fr.argb_off) & 15) == 0;
}
// Everything about an interior chunk that does not depend on the thread, as scalars the workgroup computes once and advances
// from chunk to chunk (the descriptor lives in memory: read through `fr` inside the chunk loop it was re-loaded, and waited
// for, four times per chunk).
struct FinishBases {
    uint8_t *y, *argb, *u, *v;          // first byte of the chunk's first row in each output
    const uint8_t* alpha;               // ... and in the alpha plane (null: opaque)
    uint32_t ystride, cstride, width;
    __device__ __forceinline__ void next_chunk() { y += kFinishCols; argb += 4 * kFinishCols; u += kFinishCols / 2; v += kFinishCols / 2; if (alpha) alpha += kFinishCols; }
};
__device__ __forceinline__ FinishBases finish_bases(const wp_vp8_frame& fr, int ty, int c0, uint8_t* Yp, uint8_t* Up, uint8_t* Vp, const uint8_t* Ap, uint8_t* out) {
    FinishBases b;
    b.ystride = 16u * uint32_t(fr.mbw); b.cstride = 8u * uint32_t(fr.mbw); b.width = uint32_t(fr.width);
    const uint32_t row0 = uint32_t(16 * ty - kFiltShift);                          // first luma row of the chunk (interior: >= 0)
    b.y = Yp + fr.y_off + size_t(row0 * b.ystride + uint32_t(c0));
    b.argb = out + fr.argb_off + 4u * size_t(row0 * b.width + uint32_t(c0));
    const size_t coff = size_t((row0 >> 1) * b.cstride + (uint32_t(c0) >> 1));
    b.u = Up + fr.u_off + coff; b.v = Vp + fr.v_off + coff;
    b.alpha = (Ap != nullptr && fr.has_alpha) ? Ap + fr.alpha_off + size_t(row0 * b.width + uint32_t(c0)) : nullptr;
    return b;
}
// A wavefront that holds the chroma pair whose green term needs the Y-dependent tie correction (wp_convert.h) patches the
// pixels concerned afterwards (a wavefront-uniform branch).  Shifted (S) tiles only.
__device__ __forceinline__ void finish_emit_interior(const FinishBases& fb, int t, const uint4* tiles4) {
    const uint32_t* tiles32 = reinterpret_cast<const uint32_t*>(tiles4);
    const uint16_t* tiles16 = reinterpret_cast<const uint16_t*>(tiles4);
    constexpr uint32_t shift = uint32_t(kFiltShift);
    {
        const uint32_t q = uint32_t(t) & 31u, rp = uint32_t(t) >> 5;
        const uint32_t cs = 4u * q + shift;                                         // column inside the chunk's tiles
        const uint32_t k = cs >> 4, sub = (cs >> 2) & 3u;
        const uint32_t lbase = k * uint32_t(kTileBytes / 4) + sub + 8u * rp;
        const uint32_t ci = k * uint32_t(kTileBytes / 2) + 4u * rp + sub;
        const uint32_t u2 = tiles16[ci + kTileCb / 2], v2 = tiles16[ci + kTileCr / 2];
#if WP_FINISH_PRIO_LO != WP_FINISH_PRIO && WP_FINISH_LO_AT == 0
        __builtin_amdgcn_s_setprio(WP_FINISH_PRIO_LO);
#endif
        const ChromaTerms t0 = chroma_terms(int(u2 & 0xff), int(v2 & 0xff));
        const ChromaTerms t1 = chroma_terms(int(u2 >> 8), int(v2 >> 8));
        const bool any_tie = __builtin_amdgcn_ballot_w64(t0.tie || t1.tie) != 0;
        const ChromaPairs cp0 = chroma_pairs(t0), cp1 = chroma_pairs(t1);
        const uint32_t ya = tiles32[lbase], yb = tiles32[lbase + 4];
        const uint32_t yo = __umul24(2u * rp, fb.ystride) + 4u * q;
        WP_PLANE_STORE(reinterpret_cast<uint32_t*>(fb.y + yo), ya);
        WP_PLANE_STORE(reinterpret_cast<uint32_t*>(fb.y + fb.ystride + yo), yb);
        const uint32_t po = 4u * (__umul24(2u * rp, fb.width) + 4u * q);
        __builtin_amdgcn_sched_barrier(0);      // one row at a time: interleaved, the two conversions need 38 registers (budget: 32)
        uint4 o0 = convert_row4_opaque(ya, cp0, cp1);
        if (any_tie) tie_patch_row4(o0, ya, t0.tie, t1.tie);
        if (fb.alpha) alpha_patch_row4(o0, *reinterpret_cast<const uint32_t*>(fb.alpha + (po >> 2)));
        const u32x4 w0 = {o0.x, o0.y, o0.z, o0.w};
        WP_ARGB_STORE(reinterpret_cast<u32x4*>(fb.argb + po), w0);
#if WP_FINISH_PRIO_LO != WP_FINISH_PRIO && WP_FINISH_LO_AT == 1
        __builtin_amdgcn_s_setprio(WP_FINISH_PRIO_LO);       // the second half of a chunk at the lower priority (see vp8_finish.hip)
#endif
        __builtin_amdgcn_sched_barrier(0);
        uint4 o1 = convert_row4_opaque(yb, cp0, cp1);
        if (any_tie) tie_patch_row4(o1, yb, t0.tie, t1.tie);
        if (fb.alpha) alpha_patch_row4(o1, *reinterpret_cast<const uint32_t*>(fb.alpha + fb.width + (po >> 2)));
        const u32x4 w1 = {o1.x, o1.y, o1.z, o1.w};
        WP_ARGB_STORE(reinterpret_cast<u32x4*>(fb.argb + 4u * fb.width + po), w1);
#if WP_FINISH_PRIO_LO != WP_FINISH_PRIO && WP_FINISH_LO_AT == 2
        __builtin_amdgcn_s_setprio(WP_FINISH_PRIO_LO);
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        const uint32_t plane = uint32_t(__builtin_amdgcn_readfirstlane(t)) >> 7;      // wavefronts 0, 1: Cb; 2, 3: Cr
        const uint32_t row = (uint32_t(t) >> 4) & 7u, dw = uint32_t(t) & 15u;
        const uint32_t cs = 4u * dw + (shift >> 1);
        uint8_t* const P = plane ? fb.v : fb.u;
        WP_PLANE_STORE(reinterpret_cast<uint32_t*>(P + (__umul24(row, fb.cstride) + 4u * dw)),
                    tiles32[(cs >> 3) * uint32_t(kTileBytes / 4) + uint32_t(kTileCb / 4) + 16u * plane + 2u * row + ((cs >> 2) & 1u)]);
    }
}

}  // namespace wpk
