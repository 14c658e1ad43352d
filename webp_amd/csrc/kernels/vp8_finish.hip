// Finishing pass of the lossy path: macroblock tiles (wp_device.h) -> row-major Y/U/V planes (what
// ImageData.Yuv420 carries, Vp8Decoder.ensureImg geometry) and tightly packed A,R,G,B (Tools.toArgb,
// Tools.hx:6-48), plus the inverse helper that tiles row-major planes for the stand-alone loop filter.
//
// Pure streaming.  A workgroup produces 256 output columns x 16 rows: it parks the 16 (or, for shifted tiles,
// 17) horizontally adjacent source tiles (6.4 KB, contiguous in memory) in LDS and writes rows: a wavefront emits
// 1 KB of contiguous ARGB (256 pixels of one row), 256 B of luma or 128 B of chroma per instruction, all starting
// on 128-byte boundaries of the row (no cache line is shared between workgroups).  Frames are read from the S tiles (shift 8 luma /
// 4 chroma: what the loop filter and the one-kernel path leave, also for frames whose loop filter is off) or, on request,
// from the R tiles (no shift: the unfiltered planes wp_vp8_recon_dev can return).
#include "wp_device.h"
#include "wp_convert.h"

// Everything this pass touches is used exactly once: non-temporal accesses keep it from displacing the tiles that the
// wavefront kernels of the next launch re-read through L2.
#ifdef WP_FINISH_NO_NT
#define WP_NT_LOAD(p) (*(p))
#define WP_NT_STORE(p, v) (*(p) = (v))
#else
#define WP_NT_LOAD(p) __builtin_nontemporal_load(p)
#define WP_NT_STORE(p, v) __builtin_nontemporal_store((v), (p))
#endif

namespace wpk {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kFinishCols = 256;                       // output luma columns per workgroup
constexpr int kFinishTiles = kFinishCols / 16 + 1;      // source tiles that can overlap them

__global__ __launch_bounds__(256)
void vp8_finish_kernel(const wp_vp8_frame* __restrict__ frames, const uint8_t* __restrict__ Rp, const uint8_t* __restrict__ Sp,
                       int force_unshifted, uint8_t* Yp, uint8_t* Up, uint8_t* Vp, const uint8_t* __restrict__ Ap, uint8_t* out) {
    __shared__ uint4 tiles4[kFinishTiles * kTileBytes / 16];
    const wp_vp8_frame& fr = frames[blockIdx.z];
    const int mbw = fr.mbw, mbh = fr.mbh, width = fr.width, height = fr.height;
    const bool shifted = !force_unshifted;
    const int shift = shifted ? kFiltShift : 0;
    const int ntx = mbw + (shifted ? 1 : 0), nty = mbh + (shifted ? 1 : 0);
    const int ty = blockIdx.y, c0 = blockIdx.x * kFinishCols;     // output columns c0 .. c0+255
    if (ty >= nty || c0 >= 16 * mbw) return;
    const int tx0 = c0 >> 4;                                      // (c0 + shift) >> 4: first source tile
    const int nt = min(shifted ? kFinishTiles : kFinishTiles - 1, ntx - tx0);
    const uint8_t* src = (shifted ? Sp + size_t(fr.stile_base) * kTileBytes : Rp + size_t(fr.mb_base) * kTileBytes)
                       + (size_t(ty) * ntx + tx0) * kTileBytes;
    const int t = threadIdx.x;
    for (int i = t; i < nt * (kTileBytes / 16); i += 256) { const u32x4 v = WP_NT_LOAD(reinterpret_cast<const u32x4*>(src) + i); tiles4[i] = make_uint4(v.x, v.y, v.z, v.w); }
    __syncthreads();
    const uint32_t* tiles32 = reinterpret_cast<const uint32_t*>(tiles4);
    const uint16_t* tiles16 = reinterpret_cast<const uint16_t*>(tiles4);

    const int ystride = 16 * mbw, cstride = 8 * mbw;
    uint8_t* const Y = Yp ? Yp + fr.y_off : nullptr;
    const uint8_t* A = (Ap != nullptr && fr.has_alpha) ? Ap + fr.alpha_off : nullptr;
    uint8_t* const dst = out ? out + fr.argb_off : nullptr;

    // ---- luma rows + ARGB: thread = 4 pixels x 2 rows (one chroma sample pair: its rounded terms are computed once),
    //      wavefront = 256 pixels of two rows ----
    const int q = t & 63, px = c0 + 4 * q;
    const int k = (4 * q + shift) >> 4, c4 = (4 * q + shift) & 15;         // source tile / column inside it
    if (k < nt && px < ystride) {
        const int npx = min(4, width - px);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int rp = pass * 4 + (t >> 6);                            // row pair inside the tile row (shift is even)
            const int py0 = 16 * ty - shift + 2 * rp;
            if (py0 < 0 || py0 >= 16 * mbh) continue;
            const uint32_t ya = tiles32[k * (kTileBytes / 4) + 8 * rp + (c4 >> 2)];
            const uint32_t yb = tiles32[k * (kTileBytes / 4) + 8 * rp + 4 + (c4 >> 2)];
            if (Y) {
                uint8_t* yp = Y + size_t(py0) * ystride + px;
                WP_NT_STORE(reinterpret_cast<uint32_t*>(yp), ya);
                WP_NT_STORE(reinterpret_cast<uint32_t*>(yp + ystride), yb);
            }
            if (dst && py0 < height && px < width) {
                const int ci = k * (kTileBytes / 2) + 4 * rp + (c4 >> 2);  // 16-bit index of the two chroma samples
                const uint32_t u2 = tiles16[ci + kTileCb / 2], v2 = tiles16[ci + kTileCr / 2];
                const ChromaTerms t0 = chroma_terms(int(u2 & 0xff), int(v2 & 0xff));
                const ChromaTerms t1 = chroma_terms(int(u2 >> 8), int(v2 >> 8));
                const bool any_tie = __builtin_amdgcn_ballot_w64(t0.tie || t1.tie) != 0;
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const int py = py0 + rr;
                    if (py >= height) break;
                    const size_t pix = size_t(py) * width + px;
                    uint32_t a4 = 0xffffffffu;
                    if (A) {
                        a4 = 0;
                        for (int j = 0; j < npx; ++j) a4 |= uint32_t(A[pix + j]) << (8 * j);
                    }
                    const uint4 o = convert_row4(rr ? yb : ya, t0, t1, a4, any_tie);
                    uint8_t* p = dst + pix * 4;
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
    // ---- chroma rows: 2 planes x 8 rows x (16 tiles x 2 dwords) ----
    if (Up && Vp) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = t + 256 * it;
            const int plane = idx >> 8, row = (idx >> 5) & 7, dw = idx & 31;
            const int cy = 8 * ty - (shift >> 1) + row, cx = (c0 >> 1) + 4 * dw;
            const int kk = (4 * dw + (shift >> 1)) >> 3, half = ((4 * dw + (shift >> 1)) >> 2) & 1;
            if (kk < nt && cy >= 0 && cy < 8 * mbh && cx < cstride) {
                uint8_t* P = plane ? Vp + fr.v_off : Up + fr.u_off;
                WP_NT_STORE(reinterpret_cast<uint32_t*>(P + size_t(cy) * cstride + cx), tiles32[kk * (kTileBytes / 4) + (kTileCb / 4) + 16 * plane + 2 * row + half]);
            }
        }
    }
}

// Row-major planes -> R tiles (only used by the stand-alone loop-filter entry point).
__global__ __launch_bounds__(256)
void vp8_tile_kernel(const wp_vp8_frame* __restrict__ frames, const uint8_t* __restrict__ Yp, const uint8_t* __restrict__ Up,
                     const uint8_t* __restrict__ Vp, uint8_t* Rp) {
    const wp_vp8_frame& fr = frames[blockIdx.y];
    const int mbw = fr.mbw, n_mb = fr.mbw * fr.mbh;
    const int t = threadIdx.x, j = t & 31;
    for (int m = blockIdx.x * 8 + (t >> 5); m < n_mb; m += gridDim.x * 8) {
        const int x = m % mbw, y = m / mbw;
        uint8_t* tile = Rp + (size_t(fr.mb_base) + m) * kTileBytes;
        if (j < 16) {
            *reinterpret_cast<uint4*>(tile + 16 * j) = *reinterpret_cast<const uint4*>(Yp + fr.y_off + size_t(16 * y + j) * (16 * mbw) + 16 * x);
        } else {
            const int p = (j - 16) >> 3, r = (j - 16) & 7;
            const uint8_t* P = p ? Vp + fr.v_off : Up + fr.u_off;
            *reinterpret_cast<uint2*>(tile + kTileCb + 64 * p + 8 * r) = *reinterpret_cast<const uint2*>(P + size_t(8 * y + r) * (8 * mbw) + 8 * x);
        }
    }
}

}  // namespace wpk

extern "C" int wpk_launch_vp8_finish(int n, int chunks, int tile_rows, const wp_vp8_frame* d_frames, const uint8_t* d_rtiles,
                                     const uint8_t* d_stiles, int force_unshifted, uint8_t* d_y, uint8_t* d_u, uint8_t* d_v,
                                     const uint8_t* d_alpha, uint8_t* d_argb, hipStream_t stream) {
    hipLaunchKernelGGL(wpk::vp8_finish_kernel, dim3(chunks, tile_rows, n), dim3(256), 0, stream,
                       d_frames, d_rtiles, d_stiles, force_unshifted, d_y, d_u, d_v, d_alpha, d_argb);
    return int(hipGetLastError());
}

extern "C" int wpk_launch_vp8_tile(int n, int grid_x, const wp_vp8_frame* d_frames, const uint8_t* d_y, const uint8_t* d_u,
                                   const uint8_t* d_v, uint8_t* d_rtiles, hipStream_t stream) {
    hipLaunchKernelGGL(wpk::vp8_tile_kernel, dim3(grid_x, n), dim3(256), 0, stream, d_frames, d_y, d_u, d_v, d_rtiles);
    return int(hipGetLastError());
}
