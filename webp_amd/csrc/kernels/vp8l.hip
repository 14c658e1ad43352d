// VP8L inverse transforms and alpha-plane post-processing for a batch of images (gfx950).
//
// Replaces vp8l/Transform.hx (inversePredictor :34-188, inverseCrossColor :190-222,
// inverseSubtractGreen :225-234, inverseColorIndexing :236-281), the driver loop
// Vp8LDecoder.hx:395-401, the RGBA->ARGB swizzle WebPDecoder.hx:109-123, and the pixel half of
// readAlpha / unfilterAlpha (WebPDecoder.hx:344-349, 357-411).
//
// Pixels are RGBA bytes, i.e. the little-endian dword R | G<<8 | B<<16 | A<<24; all
// channel arithmetic wraps mod 256 as haxe.io.Bytes.set does.
//
// Chain driver (host side, wpk_launch_vp8l): transforms are undone last-read-first.  Pass p handles,
// for every frame, its transform number n_transforms-1-p.  The per-pixel transforms are pure
// streaming kernels (one dword per lane, coalesced).  The predictor is a row wavefront: pixel (x,y)
// needs L, TL, T, TR, so row y may trail row y-1 by a few pixels; one wavefront owns 64 rows
// (lane = row), rows are skewed by one 8-pixel chunk and exchange their finished chunks with DPP-free
// lane shuffles; 64-row bands are pipelined across the wavefronts of a workgroup through an LDS
// progress word, the boundary row going through memory (workgroup-scope release/acquire).
#include "wp_device.h"
#include <algorithm>
#include <mutex>

namespace wpk {

__device__ __forceinline__ int n_tiles(int dim, int bits) { return (dim + (1 << bits) - 1) >> bits; }   // Transform.hx:23-25

// Where a frame's pixels live before inverse pass `p`: the decodePix slab until a colour-indexing
// transform with pixel packing (bits > 0) has expanded them into the scratch slab.
struct Loc { bool in_tmp; int width; };
__device__ __forceinline__ Loc locate(const wp_vp8l_frame& fr, int idx /* transform about to be undone */) {
    Loc l{false, fr.coded_width};
    for (int j = fr.n_transforms - 1; j > idx; --j)
        if (fr.t[j].type == 3) { l.width = fr.t[j].old_width; if (fr.t[j].bits > 0) l.in_tmp = true; }
    return l;
}

// per-byte wrapping add of two packed pixels
__device__ __forceinline__ uint32_t add_px(uint32_t a, uint32_t b) {
    return ((a & 0x7f7f7f7fu) + (b & 0x7f7f7f7fu)) ^ ((a ^ b) & 0x80808080u);
}
// per-byte floor((a + b) / 2)  (Transform.hx:283-285 avg2)
__device__ __forceinline__ uint32_t avg_px(uint32_t a, uint32_t b) { return __builtin_amdgcn_lerp(a, b, 0u); }   // v_lerp_u8: (a + b + 0) >> 1 per byte

typedef uint32_t U32x4 __attribute__((ext_vector_type(4)));                   // one 16-byte access, 16-byte aligned (LDS)
typedef uint32_t U32x4u __attribute__((ext_vector_type(4), aligned(4)));       // ... dword-aligned (pixel rows)
struct __attribute__((aligned(4))) Px4 { uint32_t v[4]; };      // four pixels, dword-aligned (rows are not 16-byte aligned in general)

// ---------------------------------------------------------------- pass fusion
// Each pass over an image costs 8 bytes of HBM traffic per pixel, so the passes libwebp's streams need are folded into
// the kernels next to them; every kernel decides per frame, from the transform list alone, with the same predicates:
//   * a cross-colour transform undone directly before a predictor is applied by the predictor kernel as it loads;
//   * the frame's final pixels (R,G,B,A or rotated to A,R,G,B) are written into the result slab by the predictor kernel
//     when the predictor is the last transform undone, or the last but a subtract-green (applied on the way out), and by
//     the colour-indexing kernel when that is the last; the finishing pass then skips the frame.
// Anything else (and green-only output, except from a last colour-indexing pass) takes the general one-pass-per-transform route.
enum OutputWriter { kOutFinish = 0, kOutPredictor = 1, kOutPredictorSG = 2, kOutIndexing = 3 };
__device__ __forceinline__ int output_writer(const wp_vp8l_frame& fr, int swizzle) {
    if (fr.n_transforms == 0) return kOutFinish;
    if (swizzle == WP_SWIZZLE_GREEN) return fr.t[0].type == 3 ? kOutIndexing : kOutFinish;      // a palette ALPH image: green bytes straight from the indexing pass
    if (fr.t[0].type == 0) return kOutPredictor;
    if (fr.t[0].type == 2 && fr.n_transforms >= 2 && fr.t[1].type == 0) return kOutPredictorSG;
    if (fr.t[0].type == 3) return kOutIndexing;
    return kOutFinish;
}
// is the cross-colour transform at index idx undone by the predictor kernel (the transform undone right after it)?
__device__ __forceinline__ bool cross_color_fused(const wp_vp8l_frame& fr, int idx) { return idx >= 1 && fr.t[idx - 1].type == 0; }
__device__ __forceinline__ uint32_t swizzled(uint32_t v, int swizzle) { return swizzle == WP_SWIZZLE_ARGB ? ((v << 8) | (v >> 24)) : v; }

// ---------------------------------------------------------------- streaming transforms
// Frames are split into those whose chain holds a predictor and those whose does not (wpk_launch_vp8l runs the two groups'
// chains on two streams): a kernel's `pass` argument carries the group it is to touch in bits 8..9 (0 all, 1 with, 2 without).
__device__ __forceinline__ bool has_predictor(const wp_vp8l_frame& fr) {
    bool p = false;
    for (int k = 0; k < fr.n_transforms && k < WP_VP8L_MAX_TRANSFORMS; ++k) p = p || fr.t[k].type == 0;
    return p;
}
__device__ __forceinline__ bool not_my_group(const wp_vp8l_frame& fr, int only) { return only != 0 && has_predictor(fr) != (only == 1); }

__global__ __launch_bounds__(256)
void vp8l_subtract_green_kernel(const wp_vp8l_frame* __restrict__ frames, int pass_and_group, uint8_t* pix, uint8_t* tmp, int swizzle) {
    const wp_vp8l_frame& fr = frames[blockIdx.y];
    const int pass = pass_and_group & 0xff;
    if (not_my_group(fr, pass_and_group >> 8)) return;
    const int idx = fr.n_transforms - 1 - pass;
    if (idx < 0 || fr.t[idx].type != 2) return;
    if (idx == 0 && output_writer(fr, swizzle) == kOutPredictorSG) return;       // applied by the predictor kernel
    const Loc loc = locate(fr, idx);
    uint32_t* p = reinterpret_cast<uint32_t*>(loc.in_tmp ? tmp + fr.tmp_off : pix + fr.pix_off);
    const long long total = (long long)loc.width * fr.height;          // the reference walks the whole buffer (:227)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const uint32_t v = p[i];
        const uint32_t g = (v >> 8) & 0xff;
        p[i] = add_px(v, g | (g << 16));                               // R += G, B += G
    }
}

__device__ __forceinline__ int s8(uint32_t v) { return int(int8_t(v & 0xff)); }
// one pixel of inverseCrossColor (Transform.hx:202-218); t = the tile's three multipliers
__device__ __forceinline__ uint32_t cross_color_px(uint32_t v, uint32_t t) {
    const int r2b = s8(t), g2b = s8(t >> 8), g2r = s8(t >> 16);       // Transform.hx:202-204
    const int g = s8(v >> 8);
    uint32_t red = (v & 0xff) + (uint32_t(g2r * g) >> 5);              // unsigned >>> 5, low 8 bits kept (Q10)
    uint32_t blue = ((v >> 16) & 0xff) + (uint32_t(g2b * g) >> 5);
    blue = (blue & 0xff) + (uint32_t(r2b * s8(red)) >> 5);
    return (v & 0xff00ff00u) | (red & 0xff) | ((blue & 0xff) << 16);
}

__global__ __launch_bounds__(256)
void vp8l_cross_color_kernel(const wp_vp8l_frame* __restrict__ frames, int pass_and_group, uint8_t* pix, uint8_t* tmp, const uint8_t* __restrict__ aux) {
    const wp_vp8l_frame& fr = frames[blockIdx.y];
    const int pass = pass_and_group & 0xff;
    if (not_my_group(fr, pass_and_group >> 8)) return;
    const int idx = fr.n_transforms - 1 - pass;
    if (idx < 0 || fr.t[idx].type != 1 || cross_color_fused(fr, idx)) return;
    const Loc loc = locate(fr, idx);
    uint32_t* p = reinterpret_cast<uint32_t*>(loc.in_tmp ? tmp + fr.tmp_off : pix + fr.pix_off);
    const int w = fr.t[idx].old_width, bits = fr.t[idx].bits;
    const int tpr = n_tiles(w, bits);
    const uint32_t* tiles = reinterpret_cast<const uint32_t*>(aux + fr.t[idx].data_off);
    // w, height <= 16384: pixel indices fit in 31 bits; (x, y) advance incrementally (one division per lane, not per pixel)
    const int total = w * fr.height, stride = int(gridDim.x * blockDim.x);
    const int qs = stride / w, rs = stride - qs * w;
    int i = int(blockIdx.x * blockDim.x + threadIdx.x);
    int y = i / w, x = i - y * w;
    for (; i < total; i += stride) {
        p[i] = cross_color_px(p[i], tiles[(y >> bits) * tpr + (x >> bits)]);
        x += rs; y += qs;
        if (x >= w) { x -= w; ++y; }
    }
}

__global__ __launch_bounds__(256)
void vp8l_color_indexing_kernel(const wp_vp8l_frame* __restrict__ frames, int pass_and_group, uint8_t* pix, uint8_t* tmp, const uint8_t* __restrict__ aux,
                                uint8_t* out, int swizzle) {
    const wp_vp8l_frame& fr = frames[blockIdx.y];
    const int pass = pass_and_group & 0xff;
    if (not_my_group(fr, pass_and_group >> 8)) return;
    const int idx = fr.n_transforms - 1 - pass;
    if (idx < 0 || fr.t[idx].type != 3) return;
    const Loc loc = locate(fr, idx);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(loc.in_tmp ? tmp + fr.tmp_off : pix + fr.pix_off);
    const int w = fr.t[idx].old_width, bits = fr.t[idx].bits;
    // The palette (1 << (8 >> bits) entries: 256, 16, 4 or 2) goes to LDS: a gather from memory per pixel is a request to
    // the L1 per pixel, and the L1's request rate is what this pass and whatever runs beside it are bound by.
    __shared__ uint32_t palette[256];
    {
        const uint32_t* gpal = reinterpret_cast<const uint32_t*>(aux + fr.t[idx].data_off);
        const int n_pal = 1 << (8 >> bits);
        for (int k = threadIdx.x; k < n_pal; k += blockDim.x) palette[k] = gpal[k];
        __syncthreads();
    }
    const bool final_out = idx == 0 && output_writer(fr, swizzle) == kOutIndexing;      // the last transform undone: straight into the result slab
    const int sw = final_out ? swizzle : WP_SWIZZLE_NONE;
    const bool green = sw == WP_SWIZZLE_GREEN;                              // one byte per pixel out
    uint8_t* const gdst = out + fr.out_off;
    if (bits == 0) {                                                        // in place, whole buffer (:237-248)
        uint32_t* p = const_cast<uint32_t*>(src);
        uint32_t* q = final_out ? reinterpret_cast<uint32_t*>(out + fr.out_off) : p;
        const long long total = (long long)loc.width * fr.height;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
            if (green) gdst[i] = uint8_t(palette[(p[i] >> 8) & 0xff] >> 8);
            else q[i] = swizzled(palette[(p[i] >> 8) & 0xff], sw);
        return;
    }
    uint32_t* dst = reinterpret_cast<uint32_t*>(final_out ? out + fr.out_off : tmp + fr.tmp_off);     // new 4*oldWidth*h buffer (:261)
    const int pw = loc.width;                                                // packed row length = nTiles(oldWidth, bits)
    const int bpp = 8 >> bits, xmask = (1 << bits) - 1;
    const uint32_t vmask = (1u << bpp) - 1u;
    // Four pixels of one row per lane (16-byte stores; the packed source dwords they share are loaded once by the cache).
    // w, height <= 16384: indices fit in 31 bits; (quad, row) advance incrementally (one division per lane, not per pixel).
    const int qpr = (w + 3) >> 2, total = qpr * fr.height, stride = int(gridDim.x * blockDim.x);
    const int qs = stride / qpr, rs = stride - qs * qpr;
    int i = int(blockIdx.x * blockDim.x + threadIdx.x);
    int y = i / qpr, xq = i - y * qpr;
    for (; i < total; i += stride) {
        const int x0 = 4 * xq;
        const uint32_t* srow = src + y * pw;
        // the four pixels share one packed source dword (bits 2, 3) or two (bits 1): each is loaded once
        const int d0 = x0 >> bits;
        const uint32_t s0 = (srow[d0] >> 8) & 0xff;
        const uint32_t s1 = bits == 1 ? (srow[min(d0 + 1, pw - 1)] >> 8) & 0xff : s0;
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x0 + j;                    // (pixels past the row's end are computed from what was loaded and not stored)
            const uint32_t g = ((x >> bits) == d0) ? s0 : s1;
            o[j] = swizzled(palette[(g >> ((x & xmask) * bpp)) & vmask], green ? WP_SWIZZLE_NONE : sw);
        }
        if (green) {
            uint8_t* gd = gdst + y * w + x0;
            const uint32_t g4 = ((o[0] >> 8) & 0xff) | ((o[1] >> 8) & 0xff) << 8 | ((o[2] >> 8) & 0xff) << 16 | ((o[3] >> 8) & 0xff) << 24;
            if (x0 + 4 <= w && ((fr.out_off + size_t(y) * w + x0) & 3) == 0) *reinterpret_cast<uint32_t*>(gd) = g4;
            else for (int j = 0; j < 4 && x0 + j < w; ++j) gd[j] = uint8_t(g4 >> (8 * j));
            xq += rs; y += qs;
            if (xq >= qpr) { xq -= qpr; ++y; }
            continue;
        }
        uint32_t* d = dst + y * w + x0;
        if (x0 + 4 <= w) { Px4 v; v.v[0] = o[0]; v.v[1] = o[1]; v.v[2] = o[2]; v.v[3] = o[3]; *reinterpret_cast<Px4*>(d) = v; }
        else { d[0] = o[0]; if (x0 + 1 < w) d[1] = o[1]; if (x0 + 2 < w) d[2] = o[2]; }
        xq += rs; y += qs;
        if (xq >= qpr) { xq -= qpr; ++y; }
    }
}

// final copy into the result slab: as is, with the RGBA -> ARGB byte rotation of WebPDecoder.hx:109-123, or green only
__global__ __launch_bounds__(256)
void vp8l_finish_kernel(const wp_vp8l_frame* __restrict__ frames, const uint8_t* pix, const uint8_t* tmp, uint8_t* out, int swizzle, int only) {
    const wp_vp8l_frame& fr = frames[blockIdx.y];
    if (not_my_group(fr, only)) return;
    if (output_writer(fr, swizzle) != kOutFinish) return;                    // a transform kernel wrote this frame's result already
    const Loc loc = locate(fr, -1);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(loc.in_tmp ? tmp + fr.tmp_off : pix + fr.pix_off);
    const long long total = (long long)fr.width * fr.height;
    const long long first = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    if (swizzle == WP_SWIZZLE_GREEN) {
        // the alpha plane is the green channel of a VP8L-coded ALPH image (WebPDecoder.hx:330-336): one byte per pixel,
        // four pixels per lane so that loads are 16 bytes and stores whole words
        uint8_t* d8 = out + fr.out_off;
        for (long long q = first; q < ((total + 3) >> 2); q += stride) {
            const long long i = q * 4;
            if (i + 4 <= total) {
                const uint4 v = *reinterpret_cast<const uint4*>(src + i);
                *reinterpret_cast<uint32_t*>(d8 + i) = __builtin_amdgcn_perm(v.y, v.x, 0x0c0c0501) | __builtin_amdgcn_perm(v.w, v.z, 0x05010c0c);
            } else {
                for (long long k = i; k < total; ++k) d8[k] = uint8_t(src[k] >> 8);
            }
        }
        return;
    }
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + fr.out_off);
    for (long long i = first; i < total; i += stride) {
        const uint32_t v = src[i];
        dst[i] = swizzled(v, swizzle);
    }
}

// ---------------------------------------------------------------- predictor (row wavefront)
__device__ __forceinline__ int clamp255i(int v) { return clampi(v, 0, 255); }

// One predicted pixel (Transform.hx:78-182).  Modes 14 and 15 have no case: nothing is added.
__device__ __forceinline__ uint32_t predict_px(int mode, uint32_t L, uint32_t T, uint32_t TL, uint32_t TR) {
    switch (mode) {
    case 0: return 0xff000000u;
    case 1: return L;
    case 2: return T;
    case 3: return TR;
    case 4: return TL;
    case 5: return avg_px(avg_px(L, TR), T);
    case 6: return avg_px(L, TL);
    case 7: return avg_px(L, T);
    case 8: return avg_px(TL, T);
    case 9: return avg_px(T, TR);
    case 10: return avg_px(avg_px(L, TL), avg_px(T, TR));
    case 11: {   // select: l = sum|TL - T|, t = sum|TL - L|; L if l < t else T
        const uint32_t l = __builtin_amdgcn_sad_u8(TL, T, 0u), t = __builtin_amdgcn_sad_u8(TL, L, 0u);
        return l < t ? L : T; }
    // the two clamped modes work on the channel pairs (0,2) and (1,3) in packed 16-bit arithmetic
    case 12: {   // clampAddSubtractFull(L, T, TL) per channel
        // (T - TL first: it does not depend on L, the pixel-to-pixel chain)
        const short2_t de = as_s2(T & 0x00ff00ffu) - as_s2(TL & 0x00ff00ffu);
        const short2_t dO = as_s2((T >> 8) & 0x00ff00ffu) - as_s2((TL >> 8) & 0x00ff00ffu);
        return clip_pack(as_s2(L & 0x00ff00ffu) + de, as_s2((L >> 8) & 0x00ff00ffu) + dO); }
    case 13: {   // clampAddSubtractHalf(avg2(L, T), TL): a + trunc((a - b) / 2)  (Q10: C-style division, rounds towards zero)
        const uint32_t av = avg_px(L, T);
        const short2_t ae = as_s2(av & 0x00ff00ffu), ao = as_s2((av >> 8) & 0x00ff00ffu);
        const short2_t de = ae - as_s2(TL & 0x00ff00ffu), dO = ao - as_s2((TL >> 8) & 0x00ff00ffu);
        return clip_pack(ae + ((de - (de >> 15)) >> 1), ao + ((dO - (dO >> 15)) >> 1)); }
    default: return 0u;
    }
}

// value of the lane above (lane r-1) as one DPP move (wave_shr:1; gfx9 has it) instead of a trip through the LDS
// crossbar (ds_bpermute): these hand-overs sit on the per-step dependency chain.  Lane 0 gets 0 and never uses it.
__device__ __forceinline__ uint32_t from_lane_above(uint32_t v) {
    return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x138, 0xf, 0xf, true));
}
// ... with lane 0 keeping `old` (the band's first row takes the row above the band from memory instead)
__device__ __forceinline__ uint32_t from_lane_above_or(uint32_t old, uint32_t v) {
    return uint32_t(__builtin_amdgcn_update_dpp(int(old), int(v), 0x138, 0xf, 0xf, false));
}
// value of lane r + J of the same 16-lane row (row_shl:J); J = 0: the lane's own
template <int J> __device__ __forceinline__ uint32_t from_lane_plus(uint32_t v) {
    if constexpr (J == 0) return v;
    else return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x100 + J, 0xf, 0xf, true));
}
template <int N, int... J>
__device__ __forceinline__ void rows_above(uint32_t (&T)[N], const uint32_t (&P)[N], uint32_t an, std::integer_sequence<int, J...>) {
    ((T[J] = from_lane_above_or(from_lane_plus<J>(an), P[J])), ...);
}
constexpr uint32_t kPredSeqShift = 15;      // progress word = (band sequence << 15) | pixels finished in the band's last row
constexpr int kPredRows = 64;
// Wavefronts per workgroup (= bands of one frame that can be in flight).  A band trails the one above by 64 steps and
// lasts n_chunks + 63, so four or five are active at a time.  Four is best (1024 x 1080p: 10.0 ms; 5, 6, 8: 12.5 ms; a
// single frame 3.6 vs 4.0 ms): each wavefront of the workgroup then has a SIMD to itself, and the pass is bound by the
// latency of the per-step dependency chain; registers held by waiting wavefronts also cost resident frames.
#ifndef WP_PRED_WAVES
#define WP_PRED_WAVES 4
#endif
constexpr int kPredWaves = WP_PRED_WAVES;
// Pixels a lane handles per step (two 4-pixel groups: the TR hand-over of the one-chunk skew sits between them).  Every
// lane walks its own row, so a step touches 64 different rows (tools/ubench/row_access.hip: row-strided 16-byte
// accesses reach 1.1 / 1.7 TB/s for loads / stores, 32-byte ones 3.3 / 1.9 TB/s).  Measured with the earlier two-chunk
// skew on 1024 x 1080p, whole chain: 4 px 19.4 ms, 8 px 15.9 ms, 16 px 30.6 ms, 32 px 57.5 ms.
#ifndef WP_PRED_CHUNK
#define WP_PRED_CHUNK 8
#endif
constexpr int kPC = WP_PRED_CHUNK;
// A band tells the one below how far its last row has got every kPredPublishPx pixels.  Publishing is a release fence, and
// on gfx9 that is s_waitcnt vmcnt(0): the wavefront drains ALL its memory operations, the loads it has just issued for the
// next step included -- a full trip to memory on the band's critical path every time.
#ifndef WP_PRED_PUBLISH_PX
#define WP_PRED_PUBLISH_PX 16
#endif
constexpr int kPredPublishPx = WP_PRED_PUBLISH_PX;

__global__ __launch_bounds__(kPredWaves * kWave)
void vp8l_predictor_kernel(int n_frames, const wp_vp8l_frame* __restrict__ frames, int pass, uint8_t* pix, uint8_t* tmp,
                           const uint8_t* __restrict__ aux, uint8_t* out, int swizzle, uint32_t* bands_done) {
    __shared__ uint32_t progress[kPredWaves];
#ifndef WP_PRED_NO_STAGE
    // Final pixels leave through LDS: a lane produces 32 bytes of ITS row per step, so direct stores put 64 quarter cache
    // lines of 64 different rows into every instruction (row-strided 32-byte stores reach 1.9 TB/s, tools/ubench/row_access.hip,
    // and the pass ran at exactly that).  Each row collects four chunks (128 bytes) here; the rows that complete a run in a
    // step -- a quarter of them, rows are skewed by one chunk -- are written as whole 128-byte runs, eight lanes per row.
    // Row pitch 36 dwords: the 16-byte accesses of eight consecutive lanes fall on different banks.
    __shared__ __attribute__((aligned(16))) uint32_t stage[kPredWaves][kPredRows][36];
#endif
    const int wave = threadIdx.x >> 6;
    const int r = threadIdx.x & 63;                 // row within the band
    if (threadIdx.x < kPredWaves) progress[threadIdx.x] = 0;
    __syncthreads();

    uint32_t seq = 1;
    for (int f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const wp_vp8l_frame& fr = frames[f];
        const int idx = fr.n_transforms - 1 - pass;
        if (idx < 0 || fr.t[idx].type != 0) continue;
        const Loc loc = locate(fr, idx);
        uint32_t* base = reinterpret_cast<uint32_t*>(loc.in_tmp ? tmp + fr.tmp_off : pix + fr.pix_off);
        const int w = fr.t[idx].old_width, h = fr.height, bits = fr.t[idx].bits;
        if (w == 0 || h == 0) continue;
        const int tpr = n_tiles(w, bits);
        const uint8_t* tiles = aux + fr.t[idx].data_off;
        // fused neighbours (see "pass fusion" above): cross-colour on the way in, subtract-green + rotation on the way out
        const bool cc = idx + 1 < fr.n_transforms && fr.t[idx + 1].type == 1;
        const int cc_bits = cc ? fr.t[idx + 1].bits : 0;
        const int cc_tpr = n_tiles(w, cc_bits);
        const uint32_t* cc_tiles = reinterpret_cast<const uint32_t*>(aux + (cc ? fr.t[idx + 1].data_off : 0));
        const int ow = output_writer(fr, swizzle);
        const bool to_out = (ow == kOutPredictor && idx == 0) || (ow == kOutPredictorSG && idx == 1);
        const bool sg_out = to_out && ow == kOutPredictorSG;
        uint32_t* const obase = reinterpret_cast<uint32_t*>(out + fr.out_off);
        const int n_bands = (h + kPredRows - 1) / kPredRows;
        const int n_chunks = (w + kPC - 1) / kPC;

        for (int band = 0; band < n_bands; ++band) {
            const uint32_t q = seq + uint32_t(band);
            if (int(q % kPredWaves) != wave) continue;
            const int y = band * kPredRows + r;
            const bool row_ok = y < h;
            const bool has_above = band > 0, has_below = band + 1 < n_bands;
            const int prev_wave = (wave + kPredWaves - 1) % kPredWaves;
            const int rows_here = min(kPredRows, h - band * kPredRows);
            // Row r trails row r-1 by ONE chunk: of the row above, chunk k (T, TL, and TR of all pixels but the chunk's last)
            // is what lane r-1 produced in the previous step; the TR of the last pixel is the first pixel lane r-1
            // produces in this very step -- every lane runs the same instruction, so it is shuffled down between the
            // chunk's two 4-pixel groups.  The pass is bound by a frame's latency (three frames take as long as three
            // hundred): (bands - 1) x 64 rows x skew + chunks per row steps, so the skew is what counts.
            const int n_steps = n_chunks + (rows_here - 1);
            uint32_t* row = base + (long long)(row_ok ? y : 0) * w;
            const uint8_t* tile_row = tiles + 4 * ((row_ok ? y : 0) >> bits) * tpr;
            const uint32_t* cc_row = cc_tiles + ((row_ok ? y : 0) >> cc_bits) * cc_tpr;
#ifdef WP_PRED_NO_STAGE
            uint32_t* orow = obase + (long long)(row_ok ? y : 0) * w;
#endif
            // the next band reads this band's last row back (unrotated, before subtract-green): that row is also stored in place
            const bool keep_raw = !to_out || (has_below && r == kPredRows - 1);

            uint32_t P1[kPC], T4[kPC];              // my previous chunk's outputs (what the row below fetches); row above, this chunk
            // band's first row: chunk c+1 of the row above the band and the pixel after it, read a step ahead -- one pixel per
            // lane (lanes 0..kPC), handed to lane 0 with v_readlane: one load instruction per step instead of nine
            uint32_t an_v = 0;
            const uint32_t* const band_above = base + (long long)(has_above ? band * kPredRows - 1 : 0) * w;
#pragma unroll
            for (int j = 0; j < kPC; ++j) { P1[j] = 0; T4[j] = 0; }
            uint32_t TLfirst = 0, Lpx = 0, first_px = 0;
            uint32_t avail = 0;                     // pixels of the row above known to be finished (band boundary)
            uint32_t pending_publish = 0;
            // A step's own inputs (its 8 pixels, the tile modes of its two 4-pixel groups, the cross-colour multipliers)
            // depend on nothing the wavefront computes, so they are fetched one step ahead: with a few bands of a frame in
            // flight there are too few wavefronts per SIMD to hide a load issued where it is used.  Pixels come as PAIRS of
            // chunks, 64 bytes per lane every other step: every lane reads its own row, a CU's 16 wavefronts keep 1024
            // cache lines in use and the 32 KB L1 holds 256, so whatever part of a 64-byte request is not taken at once
            // is requested from L2 again (r02 PMC: 2.2 requests per 64 bytes read, the L1 stalled on pending requests
            // three quarters of the time).
            uint32_t nx_in[kPC], nx_in2[kPC], nx_mode[kPC / 4], nx_cc[kPC / 4];       // (two arrays: one indexed by k & 1 would go to scratch)
#pragma unroll
            for (int j = 0; j < kPC; ++j) { nx_in[j] = 0; nx_in2[j] = 0; }
#pragma unroll
            for (int g4 = 0; g4 < kPC / 4; ++g4) { nx_mode[g4] = 0; nx_cc[g4] = 0; }
            auto fetch = [&](int kk) {
                if (!(row_ok && kk >= 0 && kk < n_chunks)) return;
#ifdef WP_KO_FETCH                                  // timing experiment only: one load per row
                if (kk > 0) return;
#endif
                const int xx = kPC * kk;
                if ((kk & 1) == 0) {                // a pair starts: both of its chunks
                    if (xx + 2 * kPC <= w) {
#pragma unroll
                        for (int g4 = 0; g4 < kPC / 4; ++g4) {
                            const Px4 v = *reinterpret_cast<const Px4*>(row + xx + 4 * g4);
                            const Px4 v2 = *reinterpret_cast<const Px4*>(row + xx + kPC + 4 * g4);
#pragma unroll
                            for (int j = 0; j < 4; ++j) { nx_in[4 * g4 + j] = v.v[j]; nx_in2[4 * g4 + j] = v2.v[j]; }
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < kPC; ++j) {
                            nx_in[j] = (xx + j < w) ? row[xx + j] : 0u;
                            nx_in2[j] = (xx + kPC + j < w) ? row[xx + kPC + j] : 0u;
                        }
                    }
                }
#pragma unroll
                for (int g4 = 0; g4 < kPC / 4; ++g4) {
                    // a 4-pixel group never straddles a tile (bits >= 2); tiles beyond the row's end are never read
                    nx_mode[g4] = tile_row[4 * (min(xx + 4 * g4, w - 1) >> bits) + 1];
                    if (cc) nx_cc[g4] = cc_row[min(xx + 4 * g4, w - 1) >> cc_bits];
                }
            };
            fetch(-r);
            auto wait_above = [&](uint32_t need) {
                if (avail < need) {
                    const uint32_t target = ((q - 1) << kPredSeqShift) | need;
                    uint32_t v;
                    while ((v = __hip_atomic_load(&progress[prev_wave], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
                        __builtin_amdgcn_s_sleep(1);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    avail = (v >> kPredSeqShift) == (q - 1) ? (v & ((1u << kPredSeqShift) - 1u)) : uint32_t(w);
                }
            };
            if (has_above) {
                wait_above(uint32_t(min(kPC + 1, w)));
                if (r <= kPC && r < w) an_v = band_above[r];
            }

#ifndef WP_PRED_NO_STAGE
            // output flush, per lane and band: the two rows (mod 4) this lane flushes a piece of, their byte offsets in the
            // frame's result (a frame is at most 16384 x 16384 pixels: 32 bits) and the lane's 4-pixel piece of a 32-pixel run
            const int fl_x = 4 * (r & 7);
            const uint32_t w4 = 4u * uint32_t(w);
            int fl_row[2];
            uint32_t fl_off[2];
#pragma unroll
            for (int round = 0; round < 2; ++round) {
                fl_row[round] = 4 * ((r >> 3) + 8 * round);
                fl_off[round] = uint32_t(band * kPredRows + fl_row[round]) * w4 + 4u * uint32_t(fl_x);
            }
#endif
            for (int c = 0;; ++c) {
                const int k = c - r;
                const bool act = row_ok && k >= 0 && k < n_chunks;
                // this step's inputs were fetched during the previous step (a lane's first chunk one step before it starts).
                // Taking them is the step's one wait on memory, and it comes first: everything the wavefront has in flight
                // was issued near the top of the previous step, a whole step ago.
                uint32_t in[kPC], cc_of[kPC / 4];
                int mode_of[kPC / 4];
                // (real moves, hidden from the register coalescer: left to itself it loads the next step's inputs into fresh
                // registers and moves them over at the END of this step, behind an s_waitcnt vmcnt(0) that undoes the prefetch)
                const unsigned long long odd = __builtin_amdgcn_ballot_w64((k & 1) != 0);
#pragma unroll
                for (int j = 0; j < kPC; ++j)                                   // the pair's first or second chunk
                    asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(in[j]) : "v"(nx_in[j]), "v"(nx_in2[j]), "s"(odd));
#pragma unroll
                for (int g4 = 0; g4 < kPC / 4; ++g4) {
                    uint32_t m_;
                    asm volatile("v_mov_b32 %0, %1" : "=v"(m_) : "v"(nx_mode[g4]));
                    asm volatile("v_mov_b32 %0, %1" : "=v"(cc_of[g4]) : "v"(nx_cc[g4]));
                    mode_of[g4] = int(m_ & 15);
                }
                // ---- what step c-1 left behind goes to memory NOW, before this step's loads are requested: the stores then
                // have the whole step to land.  Issued at the end of step c-1 instead they stood between the wavefront and
                // its next inputs (vmcnt counts in order) and under the release fence of the progress counter: with four
                // wavefronts per SIMD the store latency of every step was exposed (r02 PMC: the wavefronts spent half their
                // time in s_waitcnt with the VALUs 42 % busy).
                if (c > 0) {
                    const int cp = c - 1, kp = cp - r;
                    if (has_below && pending_publish) {
                        // the row announced was stored a step ago: the release fence finds nothing left to wait for
                        publish_progress(&progress[wave], pending_publish);
                        pending_publish = 0;
                    }
                    if (keep_raw && row_ok && kp >= 0 && kp < n_chunks) {      // P1: the chunk this lane produced in step c-1
                        const int xp = kPC * kp;
                        if (xp + kPC <= w) {
#pragma unroll
                            for (int g4 = 0; g4 < kPC / 4; ++g4) {
                                Px4 v;
#pragma unroll
                                for (int j = 0; j < 4; ++j) v.v[j] = P1[4 * g4 + j];
                                *reinterpret_cast<Px4*>(row + xp + 4 * g4) = v;
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < kPC; ++j) if (xp + j < w) row[xp + j] = P1[j];
                        }
                    }
#if !defined(WP_PRED_NO_STAGE) && !defined(WP_KO_FLUSH)
                    if (to_out) {
                        // rows whose chunk index was 3 mod 4 in step c-1 have a complete 128-byte run in LDS: rows r0, r0 + 4, ...
                        // (16 of them: two rounds of 8 rows x 8 lanes); a row that ended on another chunk index flushes what it
                        // has.  What depends on the lane alone was computed once per band (fl_row, fl_off, fl_x).
                        __builtin_amdgcn_wave_barrier();
                        const int r0 = (cp + 1) & 3;
                        const uint32_t r0_off = uint32_t(r0) * w4;
#pragma unroll
                        for (int round = 0; round < 2; ++round) {
                            const int rr = fl_row[round] + r0;
                            const int kk = cp - rr;                                 // (kk & 3) == 3 by construction
                            const int x = (kk >> 2) * 32 + fl_x;
                            if (kk >= 0 && kk < n_chunks && rr < rows_here && x < w) {
                                // (the four dwords are pinned as one register tuple: the row's ragged end below does not use
                                // the fourth, and the compiler would otherwise split the read and the store into 12 + 4 bytes)
                                U32x4 v = *reinterpret_cast<const U32x4*>(&stage[wave][rr][fl_x]);
                                asm volatile("" : "+v"(v));
                                uint8_t* dst = reinterpret_cast<uint8_t*>(obase) + (fl_off[round] + r0_off + uint32_t(kk >> 2) * 128u);
                                if (x + 4 <= w) *reinterpret_cast<U32x4u*>(dst) = v;
                                else { uint32_t* d = reinterpret_cast<uint32_t*>(dst); d[0] = v.x; if (x + 1 < w) d[1] = v.y; if (x + 2 < w) d[2] = v.z; }
                            }
                        }
                        if (((n_chunks - 1) & 3) != 3 && r < 8) {                   // the row that has just produced its last chunk
                            const int rr = cp - (n_chunks - 1);
                            const int x = ((n_chunks - 1) >> 2) * 32 + fl_x;
                            if (rr >= 0 && rr < rows_here && x < w) {
                                const uint4 v = *reinterpret_cast<const uint4*>(&stage[wave][rr][fl_x]);
                                uint32_t* d = obase + (long long)(band * kPredRows + rr) * w + x;
                                d[0] = v.x; if (x + 1 < w) d[1] = v.y; if (x + 2 < w) d[2] = v.z; if (x + 3 < w) d[3] = v.w;
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
#endif
                    if (has_below) {
                        // the band's last row: what it has reached is announced at the top of the next step
                        const int k_last = cp - (kPredRows - 1);
                        if (k_last >= 0 && (((k_last + 1) * kPC) % kPredPublishPx == 0 || k_last == n_chunks - 1))
                            pending_publish = (q << kPredSeqShift) | uint32_t(min(kPC * (k_last + 1), w));
                    }
                }
                if (c == n_steps) break;
                // ---- chunk k of the row above: lane r-1's previous outputs, or memory for the band's first row
                // (an_v: pixel kPC * c + lane of the row above the band in lanes 0..kPC, loaded a step ago: lane 0 takes pixel j
                // from lane j with one DPP move, every other lane its T from the lane above with another)
                uint32_t an_now = 0;
                if (has_above) asm volatile("v_mov_b32 %0, %1" : "=v"(an_now) : "v"(an_v));
                rows_above(T4, P1, an_now, std::make_integer_sequence<int, kPC>());
                const uint32_t tr_mem = from_lane_plus<kPC>(an_now);
                if (has_above) {
                    if (c + 1 < n_chunks) {
                        // lane 0 is about to read pixels up to kPC*(c+2) of the row above the band
                        wait_above(uint32_t(min(kPC * (c + 2) + 1, w)));
                        const int xa = kPC * (c + 1) + r;
                        if (r <= kPC && xa < w) an_v = band_above[xa];
                    }
                }
                // the next step's inputs are requested now and land while this chunk is being computed
                fetch(k + 1);
                const int x0 = kPC * k;
                uint32_t res[kPC];
#pragma unroll
                for (int j = 0; j < kPC; ++j) res[j] = 0;
                // the row's first output does not depend on L (black or T is added): known before the groups below fold it
                // into the TR of the row's last pixel, which can sit in the same group (w <= 4)
                if (act && x0 == 0) first_px = add_px(cc ? cross_color_px(in[0], cc_of[0]) : in[0], y == 0 ? 0xff000000u : T4[0]);
                uint32_t tr_last = 0;
                static_assert(kPC == 8, "two 4-pixel groups per chunk");
                // The lanes of a wavefront hold different modes, and a 14-way switch per pixel costs more in exec-mask
                // bookkeeping (47 SALU per pixel) than in arithmetic.  So: per 4-pixel group (one tile, one mode per lane) the
                // mode becomes a handful of lane masks; everything that does not involve L is computed for the group up
                // front; the modes that average L with its neighbours (1, 5, 6, 7, 10) share one chain avg(avg(L, A), B)
                // with per-lane A, B; and only select / clamp-add-subtract (11-13) keep a branch, taken when some lane of
                // the wavefront has that mode.
#pragma unroll
                for (int g4 = 0; g4 < kPC / 4; ++g4) {
                    if (g4 == 1) {
                        // the row above has just produced the pixel that is TR of this chunk's last pixel (every lane takes part)
                        tr_last = from_lane_above_or(tr_mem, res[0]);
                    }
                    if (!act) continue;
                    if (cc) {                // inverseCrossColor of the pixels loaded (a 4-pixel group never straddles a tile: bits >= 2)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) in[4 * g4 + jj] = cross_color_px(in[4 * g4 + jj], cc_of[g4]);
                    }
                    const int xg = x0 + 4 * g4;
                    const int m = mode_of[g4];
                    // Fast path: every active lane of the wavefront has the same mode for this 4-pixel group (libwebp picks one
                    // or two modes for most of an image: the bench streams are select, mode 11, throughout) and no lane is in
                    // the first row, the first column or the ragged end of its row.  The general code below costs ~54 VALU + 47
                    // SALU instructions per pixel whatever the modes are; a single known mode without those special cases is a
                    // dozen.  Modes that read TR (3, 5, 9, 10) stay on the general path (TR of a row's last pixel is special).
                    {
                        const int m0 = __builtin_amdgcn_readfirstlane(m);
                        const bool inside = y > 0 && xg > 0 && xg + 4 <= w;
                        constexpr uint32_t kNoTR = 0xF9D7u;       // modes 0,1,2,4,6,7,8,11,12,13,14,15
                        // (the modes that read TR -- 3, 5, 9, 10; libwebp likes 10 for smooth and for fractal-like images -- also
                        // keep away from the row's last pixel, whose TR is the row's own first pixel)
                        const bool plain = m == m0 && inside && (((kNoTR >> m0) & 1u) || xg + 4 < w);
                        auto run4 = [&](auto pred) __attribute__((always_inline)) {
                            uint32_t L = Lpx;
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) {
                                const int j = 4 * g4 + jj;
                                L = add_px(in[j], pred(L, T4[j], j ? T4[j - 1] : TLfirst, j < kPC - 1 ? T4[j + 1] : tr_last));
                                res[j] = L;
                            }
                            Lpx = L;
                        };
                        if (__builtin_amdgcn_ballot_w64(!plain) == 0) {
                            // (the switch sits outside the pixel loop: one branch sequence per group, not per pixel)
                            switch (m0) {
                            case 0: run4([](uint32_t, uint32_t, uint32_t, uint32_t) { return 0xff000000u; }); break;
                            case 1: run4([](uint32_t L, uint32_t, uint32_t, uint32_t) { return L; }); break;
                            case 2: run4([](uint32_t, uint32_t T, uint32_t, uint32_t) { return T; }); break;
                            case 3: run4([](uint32_t, uint32_t, uint32_t, uint32_t TR) { return TR; }); break;
                            case 4: run4([](uint32_t, uint32_t, uint32_t TL, uint32_t) { return TL; }); break;
                            case 5: run4([](uint32_t L, uint32_t T, uint32_t, uint32_t TR) { return avg_px(avg_px(L, TR), T); }); break;
                            case 6: run4([](uint32_t L, uint32_t, uint32_t TL, uint32_t) { return avg_px(L, TL); }); break;
                            case 7: run4([](uint32_t L, uint32_t T, uint32_t, uint32_t) { return avg_px(L, T); }); break;
                            case 8: run4([](uint32_t, uint32_t T, uint32_t TL, uint32_t) { return avg_px(TL, T); }); break;
                            case 9: run4([](uint32_t, uint32_t T, uint32_t, uint32_t TR) { return avg_px(T, TR); }); break;
                            case 10: run4([](uint32_t L, uint32_t T, uint32_t TL, uint32_t TR) { return avg_px(avg_px(L, TL), avg_px(T, TR)); }); break;
                            case 11: run4([](uint32_t L, uint32_t T, uint32_t TL, uint32_t) {
                                    return __builtin_amdgcn_sad_u8(TL, T, 0u) < __builtin_amdgcn_sad_u8(TL, L, 0u) ? L : T; }); break;
                            case 12: run4([](uint32_t L, uint32_t T, uint32_t TL, uint32_t) { return predict_px(12, L, T, TL, 0u); }); break;
                            case 13: run4([](uint32_t L, uint32_t T, uint32_t TL, uint32_t) { return predict_px(13, L, T, TL, 0u); }); break;
                            default: run4([](uint32_t, uint32_t, uint32_t, uint32_t) { return 0u; }); break;      // 14, 15: nothing is added
                            }
                            continue;
                        }
                        // libwebp's photo-like streams: select and clamp-add-subtract, tile by tile, and nothing else.  Both are
                        // computed and one is taken per lane: 21 instructions per pixel where the general code takes 30.
                        if (__builtin_amdgcn_ballot_w64(!(inside && (m == 11 || m == 12))) == 0) {      // (not all the same: that was the case above)
                            const bool is12 = m == 12;
                            run4([&](uint32_t L, uint32_t T, uint32_t TL, uint32_t) {
                                const uint32_t sel = __builtin_amdgcn_sad_u8(TL, T, 0u) < __builtin_amdgcn_sad_u8(TL, L, 0u) ? L : T;
                                return is12 ? predict_px(12, L, T, TL, 0u) : sel; });
                            continue;
                        }
                        // ... and its smooth ones: the averaging modes 5 .. 10 (10 for most tiles, a few of 7 and 8).  The general
                        // code's terms for these six alone, no frame-edge cases: 28 instructions per pixel where it takes 45.
                        if (__builtin_amdgcn_ballot_w64(!(inside && xg + 4 < w && m >= 5 && m <= 10)) == 0) {
                            const bool m5 = m == 5, chain = m != 8 && m != 9, two = m5 || m == 10;
                            uint32_t A[4], B[4], N[4];
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) {
                                const int j = 4 * g4 + jj;
                                const uint32_t T = T4[j], TL = j ? T4[j - 1] : TLfirst, TR = j < kPC - 1 ? T4[j + 1] : tr_last;
                                const uint32_t aTTR = avg_px(T, TR);
                                A[jj] = m5 ? TR : (m == 7 ? T : TL);
                                B[jj] = m5 ? T : aTTR;
                                N[jj] = m == 8 ? avg_px(TL, T) : aTTR;
                            }
                            uint32_t L = Lpx;
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) {
                                const int j = 4 * g4 + jj;
                                const uint32_t X = avg_px(L, A[jj]);
                                const uint32_t c = two ? avg_px(X, B[jj]) : X;
                                L = add_px(in[j], chain ? c : N[jj]);
                                res[j] = L;
                            }
                            Lpx = L;
                            continue;
                        }
                    }
                    const bool m5 = m == 5, m10 = m == 10, is_l = m == 1;
                    const bool use_b = m5 || m10;
                    const bool chain = is_l || use_b || m == 6 || m == 7;
                    const bool any11 = __builtin_amdgcn_ballot_w64(m == 11) != 0, any12 = __builtin_amdgcn_ballot_w64(m == 12) != 0,
                               any13 = __builtin_amdgcn_ballot_w64(m == 13) != 0;
                    // modes 0..10 (constants, single neighbours and the averages): skipped as a whole when no lane has one --
                    // libwebp's photo-like streams mix select (11) and clamp-add-subtract (12) tile by tile and little else
                    const bool any_low = __builtin_amdgcn_ballot_w64(m <= 10) != 0;
                    uint32_t A[4] = {0, 0, 0, 0}, B[4] = {0, 0, 0, 0}, N[4] = {0, 0, 0, 0}, TLv[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) TLv[jj] = (4 * g4 + jj) ? T4[4 * g4 + jj - 1] : TLfirst;
                    if (any_low) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const int j = 4 * g4 + jj, x = xg + jj;
                            const uint32_t T = T4[j];
                            const uint32_t TL = TLv[jj];
                            uint32_t TR = j < kPC - 1 ? T4[j + 1] : tr_last;
                            if (x == w - 1) TR = first_px;                        // linear addressing: top+4 is the row's own first pixel
                            const uint32_t aTTR = avg_px(T, TR);
                            A[jj] = m5 ? TR : (m == 7 ? T : TL);
                            B[jj] = m5 ? T : aTTR;
                            uint32_t n = 0u;                                      // modes 14, 15: nothing is added
                            n = m == 0 ? 0xff000000u : n;
                            n = m == 2 ? T : n;
                            n = m == 3 ? TR : n;
                            n = m == 4 ? TL : n;
                            n = m == 8 ? avg_px(TL, T) : n;
                            n = m == 9 ? aTTR : n;
                            N[jj] = n;
                        }
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int j = 4 * g4 + jj, x = xg + jj;
                        const uint32_t L = Lpx, T = T4[j];
                        uint32_t pr = 0u;
                        if (any_low) {
                            const uint32_t X = avg_px(L, A[jj]);
                            pr = use_b ? avg_px(X, B[jj]) : X;
                            pr = is_l ? L : pr;
                            pr = chain ? pr : N[jj];
                        }
                        if (any11) {     // select: l = sum|TL - T|, t = sum|TL - L|; L if l < t else T
                            const uint32_t l = __builtin_amdgcn_sad_u8(TLv[jj], T, 0u), t = __builtin_amdgcn_sad_u8(TLv[jj], L, 0u);
                            pr = m == 11 ? (l < t ? L : T) : pr;
                        }
                        if (any12) pr = m == 12 ? predict_px(12, L, T, TLv[jj], 0u) : pr;
                        if (any13) pr = m == 13 ? predict_px(13, L, T, TLv[jj], 0u) : pr;
                        uint32_t pred = x == 0 ? T4[0] : pr;                  // first column: T (:61-64)
                        pred = y == 0 ? (x == 0 ? 0xff000000u : L) : pred;    // first row: black / L (:40-53)
                        res[j] = add_px(in[j], pred);
                        Lpx = x < w ? res[j] : Lpx;
                    }
                }
                if (act) {
                    if (to_out) {            // final pixels: R += G, B += G if a subtract-green follows, then the byte order asked for
                        uint32_t fin[kPC];
#pragma unroll
                        for (int j = 0; j < kPC; ++j) fin[j] = res[j];
                        if (sg_out) {            // (wavefront-uniform branches, not a select per pixel)
#pragma unroll
                            for (int j = 0; j < kPC; ++j) { const uint32_t g = (res[j] >> 8) & 0xff; fin[j] = add_px(res[j], g | (g << 16)); }
                        }
                        if (swizzle == WP_SWIZZLE_ARGB) {
#pragma unroll
                            for (int j = 0; j < kPC; ++j) fin[j] = (fin[j] << 8) | (fin[j] >> 24);
                        }
#ifndef WP_PRED_NO_STAGE
                        uint32_t* st = &stage[wave][r][(k & 3) * kPC];
                        *reinterpret_cast<uint4*>(st) = make_uint4(fin[0], fin[1], fin[2], fin[3]);
                        *reinterpret_cast<uint4*>(st + 4) = make_uint4(fin[4], fin[5], fin[6], fin[7]);
#else
                        if (x0 + kPC <= w) {
#pragma unroll
                            for (int g4 = 0; g4 < kPC / 4; ++g4) {
                                Px4 v;
#pragma unroll
                                for (int j = 0; j < 4; ++j) v.v[j] = fin[4 * g4 + j];
                                *reinterpret_cast<Px4*>(orow + x0 + 4 * g4) = v;
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < kPC; ++j) if (x0 + j < w) orow[x0 + j] = fin[j];
                        }
#endif
                    }
                    TLfirst = T4[kPC - 1];
#pragma unroll
                    for (int j = 0; j < kPC; ++j) P1[j] = res[j];
                }
            }
            if (has_below && pending_publish) publish_progress(&progress[wave], pending_publish);      // the band's last rows are out
            // how far the pass has got, for the launcher's side stream (hold_until_kernel): one count per finished band
            if (bands_done && r == 0) __hip_atomic_fetch_add(bands_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        seq += uint32_t(n_bands);
    }
}

// ---------------------------------------------------------------- alpha plane (WebPDecoder.hx:311-411)
__device__ __forceinline__ uint32_t alpha_src(const uint8_t* src, bool rgba, long long i) { return rgba ? src[4 * i + 1] : src[i]; }

// filter 0: copy / green extraction.  filters 1-3, stage 1: first row is a running sum along x and
// the first column of the other rows a running sum along y (:364-370, :380-384, :389-396).
__global__ __launch_bounds__(256)
void alpha_copy_kernel(const wp_alpha_frame* __restrict__ frames, const uint8_t* __restrict__ src, uint8_t* alpha) {
    const wp_alpha_frame& fr = frames[blockIdx.y];
    const uint8_t* s = src + fr.src_off;
    uint8_t* d = alpha + fr.dst_off;
    const long long total = (long long)fr.width * fr.height;
    const bool rgba = fr.src_is_vp8l_rgba != 0;
    if (!rgba && s == d) return;                       // unfiltering in place: the plane is where it belongs already
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        d[i] = uint8_t(alpha_src(s, rgba, i));
}

// vertical filter, stage 1: running sum along row 0 (one wavefront per frame, 64 bytes per round).
__global__ __launch_bounds__(64)
void alpha_row0_kernel(const wp_alpha_frame* __restrict__ frames, uint8_t* alpha) {
    const wp_alpha_frame& fr = frames[blockIdx.x];
    if (fr.filter != 2) return;
    uint8_t* a = alpha + fr.dst_off;
    const int lane = threadIdx.x;
    uint32_t carry = 0;
    for (int xb = 0; xb < fr.width; xb += 64) {
        const int x = xb + lane;
        uint32_t v = x < fr.width ? a[x] : 0u;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(v, d); if (lane >= d) v += t; }
        v = (v + carry) & 0xff;
        if (x < fr.width) a[x] = uint8_t(v);
        carry = __shfl(v, 63);
    }
}
// vertical filter, stage 2: running sum down every column (:383-385); thread = column, coalesced rows.
__global__ __launch_bounds__(256)
void alpha_columns_kernel(const wp_alpha_frame* __restrict__ frames, uint8_t* alpha) {
    const wp_alpha_frame& fr = frames[blockIdx.y];
    if (fr.filter != 2) return;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= fr.width) return;
    uint8_t* a = alpha + fr.dst_off + x;
    const int w = fr.width, h = fr.height;
    uint32_t acc = a[0];
    int y = 1;
    for (; y + 8 <= h; y += 8) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = a[(long long)(y + k) * w];
#pragma unroll
        for (int k = 0; k < 8; ++k) { acc = (acc + v[k]) & 0xff; a[(long long)(y + k) * w] = uint8_t(acc); }
    }
    for (; y < h; ++y) { acc = (acc + a[(long long)y * w]) & 0xff; a[(long long)y * w] = uint8_t(acc); }
}

// Horizontal filter, fast path (rows that start dword-aligned): unfiltered[y][x] = sum_{i<=y} raw[i][0] + sum_{1<=j<=x} raw[y][j]
// (mod 256; :364-376), i.e. a running sum down column 0 and then an independent running sum along every row.
__device__ __forceinline__ bool alpha_fast_horizontal(const wp_alpha_frame& fr) {
    return fr.filter == 1 && (fr.width & 3) == 0 && (fr.dst_off & 3) == 0;
}
// stage 1: column 0 (one wavefront per frame, 64 rows per round)
__global__ __launch_bounds__(64)
void alpha_col0_kernel(const wp_alpha_frame* __restrict__ frames, uint8_t* alpha) {
    const wp_alpha_frame& fr = frames[blockIdx.x];
    if (!alpha_fast_horizontal(fr)) return;
    uint8_t* a = alpha + fr.dst_off;
    const int lane = threadIdx.x, w = fr.width;
    uint32_t carry = 0;
    for (int yb = 0; yb < fr.height; yb += 64) {
        const int y = yb + lane;
        uint32_t v = y < fr.height ? a[(long long)y * w] : 0u;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(v, d); if (lane >= d) v += t; }
        v = (v + carry) & 0xff;
        if (y < fr.height) a[(long long)y * w] = uint8_t(v);
        carry = __shfl(v, 63);
    }
}
// stage 2: one wavefront per row, 256 pixels (a dword per lane) per round
__global__ __launch_bounds__(256)
void alpha_hrows_kernel(const wp_alpha_frame* __restrict__ frames, uint8_t* alpha) {
    const wp_alpha_frame& fr = frames[blockIdx.y];
    if (!alpha_fast_horizontal(fr)) return;
    const int lane = threadIdx.x & 63, w = fr.width;
    for (int y = blockIdx.x * 4 + (threadIdx.x >> 6); y < fr.height; y += gridDim.x * 4) {
        uint32_t* row = reinterpret_cast<uint32_t*>(alpha + fr.dst_off + (long long)y * w);
        uint32_t carry = 0;
        for (int xb = 0; xb < w; xb += 256) {
            const int x = xb + 4 * lane;
            const uint32_t v = x < w ? row[x >> 2] : 0u;
            const uint32_t p0 = v & 0xff, p1 = p0 + ((v >> 8) & 0xff), p2 = p1 + ((v >> 16) & 0xff), p3 = p2 + (v >> 24);
            uint32_t sc = p3;                                                  // inclusive scan of the lanes' totals
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(sc, d); if (lane >= d) sc += t; }
            const uint32_t e = sc - p3 + carry;                                // everything to the left of this lane's dword
            if (x < w) row[x >> 2] = ((p0 + e) & 0xff) | ((p1 + e) & 0xff) << 8 | ((p2 + e) & 0xff) << 16 | ((p3 + e) & 0xff) << 24;
            carry = (__shfl(sc, 63) + carry) & 0xff;
        }
    }
}

// horizontal (1, rows not dword-aligned) and gradient (3) filters: one workgroup of four wavefronts per frame; a wavefront
// owns a band of 64 rows, lane = row, rows skewed by one pixel (row r handles x = s - r): the row above has then just
// produced its pixel x.  Bands are pipelined over the wavefronts like the VP8L predictor's: the band's first row reads
// the last row of the band above from memory, in 64-byte blocks fetched one block ahead, as far as that band has
// announced it (LDS progress word, workgroup-scope release / acquire).
constexpr int kAlphaWaves = 4;
__device__ __forceinline__ void publish64(unsigned long long* word, unsigned long long value) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__global__ __launch_bounds__(kAlphaWaves * 64)
void alpha_unfilter_kernel(const wp_alpha_frame* __restrict__ frames, uint8_t* alpha) {
    // ((band + 1) << 32) | pixels of the band's last row that are in memory.  64 bits: a raw ALPH plane takes its size from
    // the 24-bit VP8X canvas (WebPDecoder.hx:133-148), so neither the pixel count nor the band number fits a 15 / 17-bit split
    __shared__ unsigned long long progress[kAlphaWaves];
    const wp_alpha_frame& fr = frames[blockIdx.x];
    const int w = fr.width, h = fr.height, filter = fr.filter;
    if ((filter != 1 && filter != 3) || alpha_fast_horizontal(fr)) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < kAlphaWaves) progress[threadIdx.x] = 0;
    __syncthreads();
    uint8_t* a = alpha + fr.dst_off;
    const int n_bands = (h + 63) / 64;
    for (int band = wave; band < n_bands; band += kAlphaWaves) {
        const int y0 = band * 64;
        const int y = y0 + lane;
        const bool row_ok = y < h;
        const int rows_here = min(64, h - y0);
        const bool has_above = band > 0, has_below = band + 1 < n_bands;
        const int prev_wave = (wave + kAlphaWaves - 1) % kAlphaWaves;
        uint8_t* row = a + (long long)(row_ok ? y : 0) * w;
        const uint8_t* above = a + (long long)(y0 > 0 ? y0 - 1 : 0) * w;      // last row of the band above
        uint32_t avail = 0;
        auto wait_above = [&](uint32_t need) {
            if (avail < need) {
                const unsigned long long target = ((unsigned long long)band << 32) | need;        // the band above announces itself as (band - 1) + 1
                unsigned long long v;
                while ((v = __hip_atomic_load(&progress[prev_wave], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
                    __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                avail = (v >> 32) == (unsigned long long)band ? uint32_t(v) : uint32_t(w);
            }
        };
        uint32_t cur = 0, up_prev = 0, abuf = 0, anext = 0;
        if (has_above) {
            wait_above(uint32_t(min(w, 128)));
            abuf = lane < w ? above[lane] : 0u;
            anext = 64 + lane < w ? above[64 + lane] : 0u;
        }
        const int n_steps = w + rows_here - 1;
        for (int s = 0; s < n_steps; ++s) {
            const int x = s - lane;
            uint32_t up = __shfl_up(cur, 1);                                   // row above at column x (lane-1 produced it last step)
            const uint32_t up0 = __shfl(abuf, s & 63);                         // the same for the band's first row
            if (lane == 0) up = up0;
            if ((s & 63) == 63) {
                abuf = anext;
                const int nb = s + 1 + 64;                                     // first column of the block after the next
                anext = 0u;
                if (has_above && nb < w) {
                    wait_above(uint32_t(min(w, nb + 64)));
                    anext = nb + lane < w ? above[nb + lane] : 0u;
                }
            }
            if (row_ok && x >= 0 && x < w) {
                const uint32_t v = row[x];
                uint32_t pred;
                if (x == 0) pred = y == 0 ? 0u : up;                            // first column: vertical (:370, :396)
                else if (y == 0 || filter == 1) pred = cur;                     // first row / horizontal: L
                else pred = uint32_t(clampi(int(cur) + int(up) - int(up_prev), 0, 255));   // gradient (:399-405)
                cur = (v + pred) & 0xff;
                row[x] = uint8_t(cur);
                up_prev = up;
            }
            if (has_below && (s & 63) == 63) {
                const int done = s - 63 + 1;                                   // pixels the band's last row (lane 63) has stored
                if (done > 0) publish64(&progress[wave], ((unsigned long long)(band + 1) << 32) | uint32_t(min(done, w)));
            }
        }
        if (has_below) publish64(&progress[wave], ((unsigned long long)(band + 1) << 32) | uint32_t(w));
    }
}

}  // namespace wpk

namespace {
int grid_for(long long items, int n_frames) {
    long long g = (items + 255) / 256;
    const long long cap = std::max(8LL, 65536LL / std::max(1, n_frames));     // frames a kernel has nothing to do for leave at once: plenty of blocks
    return int(std::max(1LL, std::min(g, cap)));
}
}  // namespace

// The predictor is a latency chain that leaves the VALUs and the memory system half idle (see above); every other kernel of
// the chain streams.  Frames are independent, so the launcher splits them into those whose chain holds a predictor and those
// whose does not and -- given a side stream for the device (created by wp_init) -- runs the second group's chains beside the
// first's instead of after them: one fork before, one join after; every kernel skips the frames of the other group.
// 1024 x 1080p, two thirds of the frames with a predictor, one third palette images: 5.39 -> 4.63 ms (see the delay below).
// Without a side stream (wp_init never ran for the device), or with one group only, the passes run one after the other as
// before.  The enqueue sequence is short and guarded, so that the fork / join events of concurrent callers do not interleave.
// The predictor pass is a chain of dependent steps per frame, and towards its end most of a frame's wavefronts have run out
// of bands: the other group's streaming kernels cost least there.  So the side stream is held back until a share of the
// predictor pass's bands is done -- the pass counts them (a relaxed agent-scope add per band) and a one-wavefront kernel on
// the side stream polls the count.  Round 2 held the stream for 0.6 x an estimate of the pass's duration (steps of its
// longest frame x 2.6 us per step x rounds of 1024 frames): right on this GPU at this clock, 1024 x 1080p with 683 predictor
// frames: held back 40 us 4.91 ms, 0.4 of the estimate 4.83, 0.5 4.73, 0.6 4.63, 0.75 4.73, 0.85 4.91; no side stream 5.39;
// streaming kernels placed first 6.1.  The share of BANDS that corresponds is measured in DESIGN.md 4.4.
#ifndef WP_VP8L_SIDE_SHARE
#define WP_VP8L_SIDE_SHARE 0.65             // the side stream starts when this share of the predictor pass's bands is done
#endif
namespace wpk {
// Holds a stream back for about `us` microseconds (one wavefront): the persistent kernel of the other stream has to be placed
// on the CUs before streaming kernels fill them, or its workgroups trickle in as those drain (lossy pipeline, pipeline.cpp).
__global__ void delay_kernel(int us) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(32);      // 100 MHz counter
}
// Holds a stream back until the predictor pass on the other stream has finished `target` of its bands (the counter its
// wavefronts bump) -- PROGRESS, not a time estimate: round 2 held the side stream for 0.6 x (steps of the longest frame x
// 2.6 us), a constant of this GPU at this clock.  Every wait ends: the kernel gives up after `max_us` whatever the counter
// says (a faulted or never-launched predictor pass must not leave a spinning wavefront behind).
__global__ void hold_until_kernel(const uint32_t* counter, uint32_t target, int min_us, int max_us) {
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        const unsigned long long dt = wall_clock64() - t0;
        if (dt >= (unsigned long long)max_us * 100ull) break;
        if (dt >= (unsigned long long)min_us * 100ull && __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
        __builtin_amdgcn_s_sleep(64);
    }
}
}
namespace {
constexpr int kMaxDev = 64;
hipStream_t g_side[kMaxDev];
hipEvent_t g_fork[kMaxDev], g_join[kMaxDev];
constexpr int kCounters = 64;               // band counters handed out round-robin: a call's counter is not re-used before 63 later calls on the device
uint32_t* g_counters[kMaxDev];
unsigned g_next_counter[kMaxDev];
std::mutex g_side_mutex;
}
// the same hold-off for the lossy pipeline (pipeline.cpp): the finishing launch waits `us` behind the persistent kernel
extern "C" int wpk_launch_delay(int us, hipStream_t stream) {
    hipLaunchKernelGGL(wpk::delay_kernel, dim3(1), dim3(64), 0, stream, us);
    return int(hipGetLastError());
}
// ... and its progress-driven form: until `target` workgroups of the persistent kernel have announced themselves
extern "C" int wpk_launch_hold_until(const uint32_t* d_counter, uint32_t target, int min_us, int max_us, hipStream_t stream) {
    hipLaunchKernelGGL(wpk::hold_until_kernel, dim3(1), dim3(64), 0, stream, d_counter, target, min_us, max_us);
    return int(hipGetLastError());
}
// wp_shutdown: the side streams and their events go; wp_init creates them again on demand
extern "C" void wpk_vp8l_side_shutdown(void) {
    std::lock_guard<std::mutex> lock(g_side_mutex);
    for (int d = 0; d < kMaxDev; ++d) {
        if (!g_side[d]) continue;
        (void)hipStreamSynchronize(g_side[d]);
        (void)hipStreamDestroy(g_side[d]);
        (void)hipEventDestroy(g_fork[d]);
        (void)hipEventDestroy(g_join[d]);
        if (g_counters[d]) (void)hipFree(g_counters[d]);
        g_side[d] = nullptr; g_fork[d] = nullptr; g_join[d] = nullptr; g_counters[d] = nullptr;
    }
}
extern "C" int wpk_vp8l_side_init(int dev) {
    if (dev < 0 || dev >= kMaxDev) return 0;
    std::lock_guard<std::mutex> lock(g_side_mutex);
    if (g_side[dev]) return 0;
    hipStream_t s; hipEvent_t a, b;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return int(hipGetLastError());
    if (hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) {
        (void)hipStreamDestroy(s);
        return int(hipGetLastError());
    }
    uint32_t* c = nullptr;
    if (hipMalloc(&c, kCounters * sizeof(uint32_t)) != hipSuccess || hipMemset(c, 0, kCounters * sizeof(uint32_t)) != hipSuccess) {
        (void)hipStreamDestroy(s); (void)hipEventDestroy(a); (void)hipEventDestroy(b);
        return int(hipGetLastError());
    }
    g_fork[dev] = a; g_join[dev] = b; g_side[dev] = s; g_counters[dev] = c;
    return 0;
}

extern "C" int wpk_launch_vp8l(int n, const wp_vp8l_frame* fh, const wp_vp8l_frame* fd, uint8_t* d_pix, const uint8_t* d_aux,
                               uint8_t* d_tmp, uint8_t* d_out, int swizzle, hipStream_t stream) {
    using namespace wpk;
    long long max_px = 1;
    for (int i = 0; i < n; ++i) max_px = std::max(max_px, (long long)fh[i].width * fh[i].height);
    int dev = -1;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_side_mutex);
    const bool have_side = dev >= 0 && dev < kMaxDev && g_side[dev] != nullptr;
    auto with_predictor = [&](int i) {
        for (int k = 0; k < fh[i].n_transforms; ++k) if (fh[i].t[k].type == 0) return true;
        return false;
    };
    for (int first = 0; first < n; first += 65535) {
        const int cnt = std::min(65535, n - first);
        const wp_vp8l_frame* fdc = fd + first;
        const int gx = grid_for(max_px, cnt);
        bool any_p = false, any_s = false;
        long long px = 0;
        for (int i = first; i < first + cnt; ++i) {
            if (with_predictor(i)) any_p = true; else any_s = true;
            px += (long long)fh[i].width * fh[i].height;
        }
        // two streams pay for themselves on large batches only
#ifdef WP_VP8L_NO_FORK                     // tuning builds: one stream whatever the batch
        const bool fork = false;
#else
        const bool fork = have_side && any_p && any_s && px >= (64LL << 20);
#endif
        uint32_t* counter = nullptr;        // bands the first predictor launch has finished (fork only)
        long long counted_bands = 0;
        // the chains of group `only` (0: every frame) on stream st
        auto chains = [&](int only, hipStream_t st) {
            bool counted = false;
            for (int pass = 0; pass < WP_VP8L_MAX_TRANSFORMS; ++pass) {
                bool have[4] = {false, false, false, false};
                long long bands = 0;
                for (int i = first; i < first + cnt; ++i) {
                    if (only && with_predictor(i) != (only == 1)) continue;
                    const int idx = fh[i].n_transforms - 1 - pass;
                    if (idx >= 0) have[fh[i].t[idx].type] = true;
                    if (idx >= 0 && fh[i].t[idx].type == 0) bands += (fh[i].height + kPredRows - 1) / kPredRows;
                }
                const int arg = pass | (only << 8);
                if (have[0]) {
                    uint32_t* c = (only == 1 && counter && !counted) ? counter : nullptr;      // the pass the side stream keys on
                    if (c) { counted = true; counted_bands = bands; }
                    hipLaunchKernelGGL(vp8l_predictor_kernel, dim3(std::min(cnt, 1024)), dim3(kPredWaves * kWave), 0, st,
                                       cnt, fdc, pass, d_pix, d_tmp, d_aux, d_out, swizzle, c);
                }
                if (have[1]) hipLaunchKernelGGL(vp8l_cross_color_kernel, dim3(gx, cnt), dim3(256), 0, st, fdc, arg, d_pix, d_tmp, d_aux);
                if (have[2]) hipLaunchKernelGGL(vp8l_subtract_green_kernel, dim3(gx, cnt), dim3(256), 0, st, fdc, arg, d_pix, d_tmp, swizzle);
                if (have[3]) hipLaunchKernelGGL(vp8l_color_indexing_kernel, dim3(gx, cnt), dim3(256), 0, st, fdc, arg, d_pix, d_tmp, d_aux, d_out, swizzle);
            }
            hipLaunchKernelGGL(vp8l_finish_kernel, dim3(gx, cnt), dim3(256), 0, st, fdc, d_pix, d_tmp, d_out, swizzle, only);
        };
        if (fork) {
            hipStream_t side = g_side[dev];
            counter = g_counters[dev] + (g_next_counter[dev]++ % kCounters);
            // zeroed on the main stream BEFORE the fork event: the side stream's hold kernel then never sees an old count
            if (hipMemsetAsync(counter, 0, sizeof(uint32_t), stream) != hipSuccess) return int(hipGetLastError());
            if (hipEventRecord(g_fork[dev], stream) != hipSuccess || hipStreamWaitEvent(side, g_fork[dev], 0) != hipSuccess) return int(hipGetLastError());
            chains(1, stream);
            // at least 40 us (the persistent predictor kernel is placed first), then until WP_VP8L_SIDE_SHARE of the pass's
            // bands are done, at most 20 ms
            const uint32_t target = uint32_t(std::max(1.0, WP_VP8L_SIDE_SHARE * double(counted_bands)));
            hipLaunchKernelGGL(hold_until_kernel, dim3(1), dim3(64), 0, side, counter, target, 40, 20000);
            chains(2, side);
            if (hipEventRecord(g_join[dev], side) != hipSuccess || hipStreamWaitEvent(stream, g_join[dev], 0) != hipSuccess) return int(hipGetLastError());
        } else {
            chains(0, stream);
        }
    }
    return int(hipGetLastError());
}

extern "C" int wpk_launch_alpha(int n, const wp_alpha_frame* fh, const wp_alpha_frame* fd, const uint8_t* d_src, uint8_t* d_alpha,
                                hipStream_t stream) {
    using namespace wpk;
    long long max_px = 1;
    for (int i = 0; i < n; ++i) { max_px = std::max(max_px, (long long)fh[i].width * fh[i].height); }
    for (int first = 0; first < n; first += 65535) {
        const int cnt = std::min(65535, n - first);
        hipLaunchKernelGGL(alpha_copy_kernel, dim3(grid_for(max_px, cnt), cnt), dim3(256), 0, stream, fd + first, d_src, d_alpha);
    }
    bool f2 = false, f13 = false;
    int max_w = 1;
    for (int i = 0; i < n; ++i) { f2 = f2 || fh[i].filter == 2; f13 = f13 || fh[i].filter == 1 || fh[i].filter == 3; max_w = std::max(max_w, fh[i].width); }
    if (f2) {
        hipLaunchKernelGGL(alpha_row0_kernel, dim3(n), dim3(64), 0, stream, fd, d_alpha);
        for (int first = 0; first < n; first += 65535)
            hipLaunchKernelGGL(alpha_columns_kernel, dim3((max_w + 255) / 256, std::min(65535, n - first)), dim3(256), 0, stream, fd + first, d_alpha);
    }
    bool f1 = false, slow = false;
    int max_h = 1;
    for (int i = 0; i < n; ++i) {
        const bool fast = fh[i].filter == 1 && (fh[i].width & 3) == 0 && (fh[i].dst_off & 3) == 0;      // alpha_fast_horizontal
        f1 = f1 || fast;
        slow = slow || ((fh[i].filter == 1 || fh[i].filter == 3) && !fast);
        max_h = std::max(max_h, fh[i].height);
    }
    if (f1) {
        hipLaunchKernelGGL(alpha_col0_kernel, dim3(n), dim3(64), 0, stream, fd, d_alpha);
        for (int first = 0; first < n; first += 65535) {
            const int cnt = std::min(65535, n - first);
            const int gx = std::max(1, std::min((max_h + 3) / 4, std::max(8, 16384 / cnt)));
            hipLaunchKernelGGL(alpha_hrows_kernel, dim3(gx, cnt), dim3(256), 0, stream, fd + first, d_alpha);
        }
    }
    if (slow) hipLaunchKernelGGL(alpha_unfilter_kernel, dim3(n), dim3(kAlphaWaves * 64), 0, stream, fd, d_alpha);
    return int(hipGetLastError());
}
