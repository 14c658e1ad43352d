// The pixel loop of decodePix on the device (Vp8LDecoder.hx:254-333): literal stores (:283-290), LZ77 copies (:308-316)
// and the lazily filled colour cache (:320-333), driven by the token stream the host's Huffman decode leaves
// (include/webp_amd.h, wp_vp8l_tokens_frame).  The loop is serial in the tokens but wide inside one: a literal run or a
// copy of n pixels is n independent lanes (a copy whose distance is shorter than its length repeats its source with
// period `dist`, so every lane still reads a pixel that existed before the token), and filling the cache with the pixels
// since the last lookup is one hash per lane with "the latest pixel wins" settled by an LDS atomic max on the pixel index.
//
// One wavefront per image.  The last 16384 pixels it wrote stay in an LDS ring, which serves every copy whose distance
// is below 12224 pixels (the distance codes 1..120 reach 7 rows up: all of them for images up to 1746 pixels wide) and
// the cache fill; longer distances re-read the output slab from L2 after a release fence.  Token words are fetched 64 at
// a time into one register per lane and walked with v_readlane, so a token costs no memory round trip unless it is a long
// literal run.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <atomic>
#include "../../../include/webp_amd.h"

namespace wpk {

constexpr int kMaxCacheBits = 11;                           // Vp8LDecoder.hx:228
// The ring is a template parameter: 16384 pixels (64 KiB) serve every distance code for images up to 1746 pixels wide but
// leave room for two images per CU -- two of its four SIMDs then idle, the other two hold one wavefront each, and the loop
// is a chain of ~480-cycle tokens.  8192 / 4096 pixels (32 / 16 KiB) admit four / eight images per CU; copies that reach
// further back than the ring take the slab path (a release fence and loads from L2), which photo-like streams rarely need.
// The launcher picks the ring from the batch size (enough images to fill the CUs at that residency) -- LDS per wavefront =
// ring + the two colour-cache arrays, sized by the largest cache of the batch (dynamic LDS).  gfx950 only (160 KiB per CU).
template <uint32_t kRingPx> struct RingCfg {
    static constexpr uint32_t kMask = kRingPx - 1;
    static constexpr uint32_t kPiece = kRingPx >= 16384 ? 4096 : kRingPx / 8;      // longest stretch copied before the ring position is re-examined (4096 = the longest copy the bitstream codes, Vp8LDecoder.hx:337-345)
    static constexpr uint32_t kReach = kRingPx - kPiece - 64;                      // copies up to this distance never see a ring slot reused under them
};
static_assert(16384u * 4u + 2u * (4u << kMaxCacheBits) <= 160u * 1024u / 2u, "vp8l_resolve_kernel: two wavefronts with the largest ring and cache must fit gfx950's 160 KiB per CU");

__device__ __forceinline__ uint32_t lane_word(uint32_t v, uint32_t l) { return uint32_t(__builtin_amdgcn_readlane(int(v), int(l))); }
__device__ __forceinline__ uint32_t far_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// key of a pixel: the reference hashes A<<24 | R<<16 | G<<8 | B (Vp8LDecoder.hx:323-326); pixels are kept as R,G,B,A bytes
__device__ __forceinline__ uint32_t cache_hash(uint32_t rgba) {
    const uint32_t argb = (rgba & 0xff00ff00u) | ((rgba & 0xffu) << 16) | ((rgba >> 16) & 0xffu);
    return argb * 0x1e35a7bdu;
}

template <uint32_t kRingPx>
__global__ __launch_bounds__(64)
void vp8l_resolve_kernel(const wp_vp8l_tokens_frame* __restrict__ frames, const uint32_t* __restrict__ tokens, uint8_t* pix_slab, int cache_entries) {
    constexpr uint32_t kRingMask = RingCfg<kRingPx>::kMask, kCopyPiece = RingCfg<kRingPx>::kPiece, kRingReach = RingCfg<kRingPx>::kReach;
    __shared__ uint32_t ring[kRingPx];
    extern __shared__ uint32_t cache_lds[];                 // 2 x cache_entries words: values, then 1 + index of the pixel that owns the entry
    uint32_t* const cval = cache_lds;
    uint32_t* const cown = cache_lds + cache_entries;
    const wp_vp8l_tokens_frame fr = frames[blockIdx.x];
    const uint32_t lane = threadIdx.x;
    const uint32_t* tok = tokens + (fr.tok_off >> 2);
    uint32_t* out = reinterpret_cast<uint32_t*>(pix_slab + fr.pix_off);
    const uint32_t total = uint32_t(fr.coded_width) * uint32_t(fr.height);
    const uint32_t n_words = uint32_t(fr.n_words);          // the launcher refuses streams of 2^32 words and more
    const uint32_t cc_size = fr.cc_bits > 0 ? 1u << fr.cc_bits : 0u;
    const int cc_shift = 32 - fr.cc_bits;
    for (uint32_t i = lane; i < cc_size; i += 64) { cval[i] = 0; cown[i] = 0; }

    uint32_t pos = 0, p = 0, cached = 0;
    // Pixels [0, fenced) are known to have reached L2: the slab path (copies and cache fills that reach further back than the
    // ring) reads this wavefront's own earlier stores, and only a release fence -- a wait for every store still in flight --
    // makes them visible to its loads.  What the slab path reads is at least a ring's reach old, so one fence serves every
    // far access of the next few thousand pixels (a photo-like 300 x 300 stream: 568 far copies, ~30 fences).
    uint32_t fenced = 0;
    // (settle: the wait for a token block's load sits where the block is loaded.  Left to the compiler it sits in front of
    // every token's v_readlane, as s_waitcnt vmcnt(0) -- which also waits for every pixel store of the token before.)
    uint32_t blk_base = 0, blk = lane < n_words ? tok[lane] : 0u;
    asm volatile("; settle %0" : "+v"(blk));
    while (pos < n_words && p < total) {
        if (pos - blk_base >= 64u) {
            blk_base = pos;
            blk = (lane < n_words - pos) ? tok[pos + lane] : 0u;
            asm volatile("; settle %0" : "+v"(blk));
        }
        const uint32_t t = lane_word(blk, pos - blk_base);
        const uint32_t kind = t >> 30, arg = t & 0x3fffffffu;
        if (kind == WP_TOK_LITERALS) {
            const uint32_t first = pos + 1;
            const uint32_t n = min(min(arg, total - p), n_words - first);
            if (first - blk_base + n <= 64u) {               // the payload is in the block already
                const uint32_t w = uint32_t(__builtin_amdgcn_ds_bpermute(int(4u * (first - blk_base + lane)), int(blk)));
                if (lane < n) { ring[(p + lane) & kRingMask] = w; out[p + lane] = w; }
            } else {
                // a long run (an image that does not compress is one literal per pixel): four loads in flight per round --
                // with one, a round is a trip to memory
                uint32_t c = 0;
                for (; c + 256 <= n; c += 256) {
                    uint32_t w[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) w[k] = tok[first + c + 64 * k + lane];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const uint32_t i = c + 64 * k + lane; ring[(p + i) & kRingMask] = w[k]; out[p + i] = w[k]; }
                }
                for (; c < n; c += 64) {
                    const uint32_t i = c + lane;
                    if (i < n) { const uint32_t w = tok[first + i]; ring[(p + i) & kRingMask] = w; out[p + i] = w; }
                }
            }
            p += n;
            if (n != arg) break;                             // malformed: the run leaves the image or the stream
            pos = first + n;
        } else if (kind == WP_TOK_COPY) {
            if (n_words - pos < 2u) break;
            const uint32_t at = pos + 1 - blk_base;
            const uint32_t dist = at < 64u ? lane_word(blk, at) : uint32_t(__builtin_amdgcn_readfirstlane(int(tok[pos + 1])));
            if (dist == 0 || dist > p || arg > total - p) break;        // "Invalid LZ77 parameters" (the host has refused these)
            uint32_t len = arg;
            while (len) {
                const uint32_t l = min(len, kCopyPiece);
                const uint32_t src0 = p - dist;
                const bool periodic = dist < l;
                if (dist <= kRingReach) {
                    // Source and destination never overlap inside a piece (dist >= l, or -- periodic -- every source lies in
                    // the `dist` pixels before p).  Pixel i of a periodic copy reads source i mod dist: lane mod dist by a
                    // float reciprocal with one correction step (lane < 64, dist < 4096: exact), advanced by 64 mod dist per
                    // round, no integer division anywhere.
                    uint32_t r = lane;
                    uint32_t step = 64u;
                    if (periodic) {
                        const float inv = __builtin_amdgcn_rcpf(float(dist));
                        r = lane - uint32_t(float(lane) * inv) * dist;
                        if (int(r) < 0) r += dist;
                        if (r >= dist) r -= dist;
                        step = 64u - uint32_t(64.0f * inv) * dist;      // uniform
                        if (int(step) < 0) step += dist;
                        if (step >= dist) step -= dist;
                    }
                    if (l <= 64u) {                          // most copies: one round
                        if (lane < l) {
                            const uint32_t w = ring[(src0 + r) & kRingMask];
                            ring[(p + lane) & kRingMask] = w;
                            out[p + lane] = w;
                        }
                    } else {                                 // four reads in flight before their four writes
                        for (uint32_t c = 0; c < l; c += 256) {
                            uint32_t w[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                w[k] = (c + 64 * k + lane < l) ? ring[(src0 + r) & kRingMask] : 0u;
                                r += step;
                                if (periodic && r >= dist) r -= dist;
                            }
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const uint32_t i = c + 64 * k + lane;
                                if (i < l) { ring[(p + i) & kRingMask] = w[k]; out[p + i] = w[k]; }
                            }
                        }
                    }
                } else {                                     // dist > kRingReach > l: plain forward copy out of the slab
                    if (src0 + l > fenced) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); fenced = p; }
                    for (uint32_t c = 0; c < l; c += 64) {
                        const uint32_t i = c + lane;
                        if (i < l) {
                            const uint32_t w = far_load(out + src0 + i);
                            ring[(p + i) & kRingMask] = w;
                            out[p + i] = w;
                        }
                    }
                }
                p += l;
                len -= l;
            }
            pos += 2;
        } else if (kind == WP_TOK_CACHE) {
            if (arg >= cc_size) break;                        // "Invalid color cache index" / "null color cache"
            // the pixels written since the last lookup enter the cache now, in order (Vp8LDecoder.hx:320-327)
            // pixels that have left the ring come back from the slab: ONE release fence makes this wavefront's own stores
            // visible to its loads (nothing is stored between here and the end of the fill)
            if (p - cached > kRingPx && p - kRingPx > fenced) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); fenced = p; }
            for (uint32_t b = cached; b < p; b += 64) {
                const uint32_t idx = b + lane;
                const bool active = idx < p;
                uint32_t w = 0;
                if (p - b <= kRingPx) {
                    if (active) w = ring[idx & kRingMask];
                } else {
                    uint32_t wf = 0;
                    if (active) wf = far_load(out + idx);
                    asm volatile("; settle %0" : "+v"(wf));       // (see above: keeps s_waitcnt vmcnt(0) out of the common path)
                    w = wf;
                }
                const uint32_t key = cache_hash(w) >> cc_shift;
                // (workgroup-scope relaxed atomics: LDS operations of one wavefront execute in order; the default
                // atomics would drain the wavefront's pixel stores first)
                if (active) __hip_atomic_fetch_max(&cown[key], idx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (active && __hip_atomic_load(&cown[key], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == idx + 1)
                    __hip_atomic_store(&cval[key], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            cached = p;
            const uint32_t w = __hip_atomic_load(&cval[arg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (lane == 0) { ring[p & kRingMask] = w; out[p] = w; }
            p += 1;
            pos += 1;
        } else {
            break;
        }
    }
}

}  // namespace wpk

namespace { std::atomic<int> g_force_ring{0}; }
extern "C" void wpk_resolve_force_ring(int px) { g_force_ring.store(px, std::memory_order_relaxed); }
// max_cc_bits: the largest colour-cache size (bits) among the frames; cus: compute units of the device
extern "C" int wpk_launch_vp8l_resolve(int n, int max_cc_bits, int cus, const wp_vp8l_tokens_frame* d_frames, const uint32_t* d_tokens, uint8_t* d_pix, hipStream_t stream) {
    const int entries = max_cc_bits > 0 ? 1 << max_cc_bits : 0;
    const size_t dyn = size_t(2) * size_t(entries) * 4;
    const int force = g_force_ring.load(std::memory_order_relaxed);      // wp_debug_set_resolve_ring: test hook
    // enough images to give every SIMD a wavefront (or two) at the smaller rings?  otherwise the big ring costs nothing
    int ring = n >= 6 * cus ? 4096 : n >= 3 * cus ? 8192 : 16384;
    if (force == 4096 || force == 8192 || force == 16384) ring = force;
    if (ring == 4096) hipLaunchKernelGGL(wpk::vp8l_resolve_kernel<4096>, dim3(n), dim3(64), dyn, stream, d_frames, d_tokens, d_pix, entries);
    else if (ring == 8192) hipLaunchKernelGGL(wpk::vp8l_resolve_kernel<8192>, dim3(n), dim3(64), dyn, stream, d_frames, d_tokens, d_pix, entries);
    else hipLaunchKernelGGL(wpk::vp8l_resolve_kernel<16384>, dim3(n), dim3(64), dyn, stream, d_frames, d_tokens, d_pix, entries);
    return int(hipGetLastError());
}
