// C ABI of the library (include/webp_amd.h): argument checking, launch geometry, host staging and the
// batch assembler that turns parsed files into the packed slabs the kernels consume.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <string>
#include <vector>
#include "host/host.h"

// kernel launchers (kernels/*.hip)
extern "C" {
int wpk_launch_vp8_recon(int n, int cus, const wp_vp8_frame* d_frames, const void* d_mb, const int16_t* d_coeff,
                         uint8_t* d_rtiles, hipStream_t stream);
int wpk_launch_vp8_filter(int n, int cus, const wp_vp8_frame* d_frames, const uint32_t* d_filt,
                          const uint8_t* d_rtiles, uint8_t* d_stiles, hipStream_t stream);
int wpk_mb_wgs_per_cu(void);
int wpk_vp8l_side_init(int dev);
void wpk_vp8l_side_shutdown(void);
void wpk_resolve_force_ring(int px);
int wpk_launch_vp8_recon_filter(int n, int cus, const wp_vp8_frame* d_frames, int use_turns, const void* d_mb, const int16_t* d_coeff,
                                const uint32_t* d_filt, uint8_t* d_rtiles, uint8_t* d_stiles, hipStream_t stream,
                                uint32_t* d_started, uint32_t* workgroups);
int wpk_vp8_recon_filter_counted(int n, const wp_vp8_frame* fh, const wp_vp8_frame* fd, const uint32_t* d_mb, const uint32_t* d_filt,
                                 const int16_t* d_coeff, void* d_work, size_t work_bytes, void* stream, uint32_t* d_started,
                                 uint32_t* workgroups);
int wpk_launch_vp8_finish(int n, int chunks, int tile_rows, int alone, const wp_vp8_frame* d_frames, const uint8_t* d_rtiles,
                          const uint8_t* d_stiles, int force_unshifted, uint8_t* d_y, uint8_t* d_u, uint8_t* d_v,
                          const uint8_t* d_alpha, uint8_t* d_argb, hipStream_t stream);
int wpk_launch_vp8_tile(int n, int grid_x, const wp_vp8_frame* d_frames, const uint8_t* d_y, const uint8_t* d_u,
                        const uint8_t* d_v, uint8_t* d_rtiles, hipStream_t stream);
int wpk_launch_yuv420_to_argb(int n, int grid_x, const wp_vp8_frame* d_frames, const uint8_t* d_y, const uint8_t* d_u,
                              const uint8_t* d_v, const uint8_t* d_alpha, uint8_t* d_argb, hipStream_t stream);
int wpk_launch_vp8l(int n, const wp_vp8l_frame* frames_host, const wp_vp8l_frame* d_frames, uint8_t* d_pix, const uint8_t* d_aux,
                    uint8_t* d_tmp, uint8_t* d_out, int swizzle, hipStream_t stream);
int wpk_launch_anim_compose(int n, const wp_anim_frame* d_frames, int cw, int ch, const uint8_t* d_argb, uint8_t* d_canvases, hipStream_t stream);
int wpk_launch_vp8l_resolve(int n, int max_cc_bits, int cus, const wp_vp8l_tokens_frame* d_frames, const uint32_t* d_tokens, uint8_t* d_pix, hipStream_t stream);
int wpk_launch_alpha(int n, const wp_alpha_frame* frames_host, const wp_alpha_frame* d_frames, const uint8_t* d_src, uint8_t* d_alpha,
                     hipStream_t stream);
}

// Workspace of wp_vp8_decode_to_argb_static_dev, the entry point that allocates nothing: device memory that exists from
// the moment the code object is loaded.  In host code this name is only a handle for hipGetSymbolAddress -- every use of
// the memory goes through the per-device address resolved by static_scratch() below.
static __device__ __align__(256) unsigned char scratch[WP_STATIC_WORK_BYTES];

namespace {

thread_local std::string g_error;
std::atomic<int> g_cus{0};      // compute units of the device (all devices of a node are the same part)
std::atomic<int> g_launch_cap{0};   // wp_debug_set_frames_per_launch: test hook, 0 = off
constexpr int kDefaultCus = 256;    // MI355X; used by the host-side layout functions until wp_init has read the real count

int fail(int code, const std::string& msg) { g_error = msg; return code; }
int hip_fail(hipError_t e, const char* what) { return fail(WP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return hip_fail(e_, #expr); } while (0)

int ensure_init() {
    if (g_cus.load(std::memory_order_relaxed) > 0) return WP_OK;
    return wp_init(-1);
}

// Address of `scratch` on the CURRENT device (every GPU holds its own copy of the array at its own address): resolved once
// per device -- by wp_init, or by the first static call on a device wp_init was never called for -- and kept.
constexpr int kMaxDevices = 64;
std::atomic<void*> g_scratch_ptr[kMaxDevices];
int static_scratch(void** out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(WP_ERR_UNSUPPORTED, "device index beyond the static workspace table");
    void* p = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!p) {
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(WP_ERR_HIP, "hipGetSymbolAddress returned a null address for the static workspace");
        g_scratch_ptr[dev].store(p, std::memory_order_release);
    }
    *out = p;
    return WP_OK;
}

// 0: usable, 1: bad geometry, 2: misaligned offsets
int vp8_frame_defect(const wp_vp8_frame& a) {
    if (a.mbw <= 0 || a.mbh <= 0 || a.mbw > 1024 || a.mbh > 1024 || a.width <= 0 || a.height <= 0 ||
        a.width > 16 * a.mbw || a.height > 16 * a.mbh || a.filter_type < 0 || a.filter_type > 2) return 1;
    if ((a.y_off & 15) || (a.u_off & 7) || (a.v_off & 7) || (a.argb_off & 3)) return 2;
    return 0;
}
int check_vp8_frames(int n, const wp_vp8_frame* f) {
    if (n < 0 || (n > 0 && !f)) return fail(WP_ERR_INVALID, "bad frame array");
    for (int i = 0; i < n; ++i) {
        const int d = vp8_frame_defect(f[i]);
        if (d == 1) return fail(WP_ERR_INVALID, "bad wp_vp8_frame at index " + std::to_string(i));
        if (d == 2) return fail(WP_ERR_INVALID, "misaligned plane offset in wp_vp8_frame " + std::to_string(i));
    }
    return WP_OK;
}

constexpr size_t kTileBytes = 384;      // one macroblock tile of the workspace (kernels/wp_device.h)
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Workspace layout: [R tiles: one per MB record slot][S tiles: (mbw+1) x (mbh+1) per frame, in frame order]
struct Vp8Work { size_t n_mb = 0, n_st = 0, r_bytes = 0, total = 0; };
Vp8Work vp8_work(int n, const wp_vp8_frame* f) {
    Vp8Work w;
    for (int i = 0; i < n; ++i) {
        w.n_mb = std::max(w.n_mb, size_t(f[i].mb_base) + size_t(f[i].mbw) * f[i].mbh);
        w.n_st += size_t(f[i].mbw + 1) * size_t(f[i].mbh + 1);
    }
    w.r_bytes = align_up(w.n_mb * kTileBytes, 256);
    w.total = w.r_bytes + w.n_st * kTileBytes;
    return w;
}

// The kernels index the workspace with mb_base / stile_base from the descriptors: check them against the size the
// caller really allocated before anything is launched.
int check_vp8_work(int n, const wp_vp8_frame* f, const void* d_work, size_t work_bytes, Vp8Work& w) {
    w = vp8_work(n, f);
    if (!d_work) return fail(WP_ERR_INVALID, "null workspace pointer");
    if (reinterpret_cast<uintptr_t>(d_work) & 255) return fail(WP_ERR_INVALID, "workspace must be 256-byte aligned");
    if (w.total > work_bytes) return fail(WP_ERR_INVALID, "workspace too small (see wp_vp8_plan)");
    size_t st = 0;
    for (int i = 0; i < n; ++i) {
        if (size_t(f[i].stile_base) != st) return fail(WP_ERR_INVALID, "stile_base of frame " + std::to_string(i) + " was not set by wp_vp8_plan");
        st += size_t(f[i].mbw + 1) * size_t(f[i].mbh + 1);
    }
    return WP_OK;
}

// The wavefront kernels number the bands a workgroup processes in a 20-bit field of their LDS progress words
// (kernels/wp_device.h: seq << 12 | macroblocks); a launch therefore takes at most as many frames as keep that count
// below 2^19 for the smallest grid that can occur (one workgroup per CU) -- far beyond any batch that fits in HBM, but
// checked rather than assumed.
int frames_per_launch(int n, const wp_vp8_frame* f, int phases = 1) {      // phases: how often a workgroup numbers the bands of a frame
    long long max_bands = 1;
    for (int i = 0; i < n; ++i) max_bands = std::max(max_bands, (long long)(f[i].mbh + 3) / 4);
    const long long per_wg = std::max(1LL, (1LL << 19) / (max_bands * phases));
    const int cus_now = g_cus.load(std::memory_order_relaxed);
    long long per = std::min<long long>(n, per_wg * (cus_now > 0 ? cus_now : kDefaultCus));
    const int cap = g_launch_cap.load(std::memory_order_relaxed);
    if (cap > 0) per = std::min<long long>(per, cap);
    return int(std::max(1LL, per));
}

// Which frame each slot of a launch works on (wp_vp8_frame.turn).  The one-kernel path's persistent workgroups take slots
// w, w + grid, w + 2 grid, ...: with frames of unequal cost in caller order the launch ends with the workgroup that drew two
// heavy ones (the bench batch: a q90 frame costs 1.5 x a q75 frame, two frames per workgroup).  Frames are therefore dealt
// heaviest first, in rounds of `grid` that alternate direction (longest-processing-time order, "snake"): the workgroup that
// got the heaviest frame of a round gets the lightest of the next.
void assign_turns(int n, wp_vp8_frame* f) {
    const int per = frames_per_launch(n, f);
    // the kernel's persistent workgroups.  Before wp_init has seen the device (a parser / worker process that only lays out
    // batches never touches the GPU) the part's own figure stands in: 256 CUs on MI355X.  A schedule laid out for another
    // grid is still a permutation -- it only balances less well.
    const int cus = g_cus.load(std::memory_order_relaxed);
    const int grid = std::max(1, wpk_mb_wgs_per_cu() * (cus > 0 ? cus : kDefaultCus));
    std::vector<uint32_t> order;
    for (int first = 0; first < n; first += per) {
        const int cnt = std::min(per, n - first);
        order.resize(size_t(cnt));
        for (int i = 0; i < cnt; ++i) order[size_t(i)] = uint32_t(i);
        auto cost = [&](uint32_t i) { const wp_vp8_frame& a = f[first + int(i)]; return a.cost ? uint64_t(a.cost) : uint64_t(a.mbw) * uint64_t(a.mbh); };
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost(a) > cost(b); });
        for (int s = 0; s < cnt; ++s) {
            const int round = s / grid, j = s % grid, in_round = std::min(grid, cnt - round * grid);
            const int k = (round & 1) ? in_round - 1 - j : j;
            f[first + s].turn = k >= 0 ? order[size_t(round * grid + k)] : order[size_t(s)];
        }
    }
}
// `turn` is trusted by the kernel as an index, and every frame must be somebody's turn: true if the descriptors of
// launch [first, first + cnt) hold a permutation of 0..cnt-1 (what assign_turns leaves for this split into launches).
// Otherwise -- descriptors that never saw wp_vp8_plan, or a different split (the test hook) -- the launch runs in slot order.
bool turns_valid(int cnt, const wp_vp8_frame* f) {
    static const bool off = std::getenv("WP_NO_TURNS") != nullptr;      // tuning switch: slot order
    if (off) return false;
    // a bitmap on the stack for launches of up to 65536 frames (every batch the static entry point accepts is far smaller:
    // it must not allocate, on the host either); the heap only beyond that
    constexpr int kStackFrames = 1 << 16;
    uint64_t stack_bits[kStackFrames / 64];
    std::vector<uint64_t> heap_bits;
    uint64_t* bits = stack_bits;
    const size_t words = (size_t(cnt) + 63) / 64;
    if (cnt > kStackFrames) {
        try { heap_bits.assign(words, 0); } catch (const std::bad_alloc&) { return false; }
        bits = heap_bits.data();
    } else {
        std::memset(stack_bits, 0, words * sizeof(uint64_t));
    }
    for (int i = 0; i < cnt; ++i) {
        const uint32_t t = f[i].turn;
        if (t >= uint32_t(cnt) || ((bits[t >> 6] >> (t & 63)) & 1)) return false;
        bits[t >> 6] |= uint64_t(1) << (t & 63);
    }
    return true;
}

struct FinishGeom { int chunks = 1, rows = 1; };
FinishGeom finish_geom(int n, const wp_vp8_frame* f) {
    FinishGeom g;
    for (int i = 0; i < n; ++i) {
        g.chunks = std::max(g.chunks, (f[i].mbw + 7) / 8);        // 128 output columns per workgroup (kernels/vp8_finish_chunk.h)
        g.rows = std::max(g.rows, f[i].mbh + 1);
    }
    return g;
}

// set while an entry point runs the finishing pass right behind its own kernels on one stream (nothing shares the GPU with it)
thread_local bool g_finish_alone = false;
std::atomic<int> g_finish_overlap{1};      // wp_hint_finish_overlap

int launch_finish(int n, const wp_vp8_frame* fh, const wp_vp8_frame* fd, const uint8_t* d_work, const Vp8Work& w, int force_unshifted,
                  uint8_t* d_y, uint8_t* d_u, uint8_t* d_v, const uint8_t* d_alpha, uint8_t* d_argb, hipStream_t stream) {
    // the z dimension of the grid is the frame index (<= 65535 per launch)
    for (int first = 0; first < n; first += 65535) {
        const int cnt = std::min(65535, n - first);
        const FinishGeom g = finish_geom(cnt, fh + first);
        const int e = wpk_launch_vp8_finish(cnt, g.chunks, g.rows, (g_finish_alone || !g_finish_overlap.load(std::memory_order_relaxed)) ? 1 : 0, fd + first, d_work, d_work + w.r_bytes, force_unshifted,
                                            d_y, d_u, d_v, d_alpha, d_argb, stream);
        if (e) return hip_fail(hipError_t(e), "vp8_finish launch");
    }
    return WP_OK;
}

}  // namespace

namespace wp { void set_error(const std::string& msg) { g_error = msg; } }

extern "C" {

int wp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int wp_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(WP_ERR_HIP, "no HIP device available (the HIP path has no CPU fallback)");
    if (device >= 0) {
        if (device >= n) return fail(WP_ERR_INVALID, "device index out of range");
        HIP_TRY(hipSetDevice(device));
    }
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    g_cus.store(prop.multiProcessorCount, std::memory_order_relaxed);
    const int side_rc = wpk_vp8l_side_init(dev);  // the VP8L chain's side stream (streaming passes beside the predictor)
    if (side_rc) return hip_fail(hipError_t(side_rc), "side stream");
    void* sp = nullptr;
    return static_scratch(&sp);             // set-up is where the static workspace's address on this device is looked up
}

// Releases what wp_init created: the VP8L chain's side stream and its two events on every device the library was
// initialised for.  The library re-initialises itself on the next compute call (or wp_init).  Pipelines are the caller's to
// destroy (wp_vp8_pipeline_destroy); device memory the library allocates lives only for the duration of a host-pointer call.
int wp_shutdown(void) {
    wpk_vp8l_side_shutdown();
    g_cus.store(0, std::memory_order_relaxed);      // ensure_init() runs wp_init again
    return WP_OK;
}

int wp_debug_set_frames_per_launch(int frames) {
    if (frames < 0) return fail(WP_ERR_INVALID, "negative frame count");
    g_launch_cap.store(frames, std::memory_order_relaxed);
    return WP_OK;
}

int wp_debug_set_resolve_ring(int pixels) {
    if (pixels != 0 && pixels != 4096 && pixels != 8192 && pixels != 16384) return fail(WP_ERR_INVALID, "ring sizes: 4096, 8192, 16384 (0 = choose by batch size)");
    wpk_resolve_force_ring(pixels);
    return WP_OK;
}

int wp_hint_finish_overlap(int on) {
    g_finish_overlap.store(on ? 1 : 0, std::memory_order_relaxed);
    return WP_OK;
}

int wp_last_error(char* buf, int cap) {
    if (!buf || cap <= 0) return int(g_error.size());
    std::snprintf(buf, size_t(cap), "%s", g_error.c_str());
    return int(g_error.size());
}

int wp_vp8_check_frames(int n, const wp_vp8_frame* frames, int* status) {
    if (n < 0 || (n > 0 && (!frames || !status))) return fail(WP_ERR_INVALID, "bad argument");
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        status[i] = vp8_frame_defect(frames[i]) ? WP_ERR_INVALID : WP_OK;
        bad += status[i] != WP_OK;
    }
    return bad;
}

int wp_vp8_plan(int n, wp_vp8_frame* frames, uint64_t* work_bytes) {
    int rc = check_vp8_frames(n, frames); if (rc) return rc;
    size_t st = 0;
    for (int i = 0; i < n; ++i) {
        if (st > 0xffffffffull) return fail(WP_ERR_INVALID, "batch too large for 32-bit tile indices");
        frames[i].stile_base = uint32_t(st);
        st += size_t(frames[i].mbw + 1) * size_t(frames[i].mbh + 1);
    }
    if (work_bytes) *work_bytes = uint64_t(vp8_work(n, frames).total);
    assign_turns(n, frames);                // host arithmetic only: planning a batch does not touch the GPU
    return WP_OK;
}

int wp_vp8_recon_dev(int n, const wp_vp8_frame* fh, const wp_vp8_frame* fd, const uint32_t* d_mb, const int16_t* d_coeff,
                     void* d_work, size_t work_bytes, uint8_t* d_y, uint8_t* d_u, uint8_t* d_v, void* stream) {
    int rc = ensure_init(); if (rc) return rc;
    rc = check_vp8_frames(n, fh); if (rc) return rc;
    if (n == 0) return WP_OK;
    if (!fd || !d_mb) return fail(WP_ERR_INVALID, "null device pointer");
    const bool planes = d_y || d_u || d_v;
    if (planes && !(d_y && d_u && d_v)) return fail(WP_ERR_INVALID, "give all three planes or none");
    Vp8Work w;
    rc = check_vp8_work(n, fh, d_work, work_bytes, w); if (rc) return rc;
    uint8_t* work = static_cast<uint8_t*>(d_work);
    const int per = frames_per_launch(n, fh);
    for (int first = 0; first < n; first += per) {
        const int e = wpk_launch_vp8_recon(std::min(per, n - first), g_cus.load(std::memory_order_relaxed), fd + first, d_mb, d_coeff, work, hipStream_t(stream));
        if (e) return hip_fail(hipError_t(e), "vp8_recon launch");
    }
    if (planes) return launch_finish(n, fh, fd, work, w, 1, d_y, d_u, d_v, nullptr, nullptr, hipStream_t(stream));
    return WP_OK;
}

int wp_vp8_loopfilter_dev(int n, const wp_vp8_frame* fh, const wp_vp8_frame* fd, const uint32_t* d_filt,
                          void* d_work, size_t work_bytes, uint8_t* d_y, uint8_t* d_u, uint8_t* d_v, void* stream) {
    int rc = ensure_init(); if (rc) return rc;
    rc = check_vp8_frames(n, fh); if (rc) return rc;
    if (n == 0) return WP_OK;
    if (!fd || !d_filt) return fail(WP_ERR_INVALID, "null device pointer");
    const bool planes = d_y || d_u || d_v;
    if (planes && !(d_y && d_u && d_v)) return fail(WP_ERR_INVALID, "give all three planes or none");
    Vp8Work w;
    rc = check_vp8_work(n, fh, d_work, work_bytes, w); if (rc) return rc;
    uint8_t* work = static_cast<uint8_t*>(d_work);
    if (planes) {
        for (int first = 0; first < n; first += 65535) {
            const int cnt = std::min(65535, n - first);
            const int e = wpk_launch_vp8_tile(cnt, 64, fd + first, d_y, d_u, d_v, work, hipStream_t(stream));
            if (e) return hip_fail(hipError_t(e), "vp8_tile launch");
        }
    }
    const int per = frames_per_launch(n, fh);
    for (int first = 0; first < n; first += per) {
        const int e = wpk_launch_vp8_filter(std::min(per, n - first), g_cus.load(std::memory_order_relaxed), fd + first, d_filt, work, work + w.r_bytes, hipStream_t(stream));
        if (e) return hip_fail(hipError_t(e), "vp8_filter launch");
    }
    if (planes) return launch_finish(n, fh, fd, work, w, 0, d_y, d_u, d_v, nullptr, nullptr, hipStream_t(stream));
    return WP_OK;
}

int wp_vp8_recon_filter_dev(int n, const wp_vp8_frame* fh, const wp_vp8_frame* fd, const uint32_t* d_mb, const uint32_t* d_filt,
                            const int16_t* d_coeff, void* d_work, size_t work_bytes, void* stream) {
    return wpk_vp8_recon_filter_counted(n, fh, fd, d_mb, d_filt, d_coeff, d_work, work_bytes, stream, nullptr, nullptr);
}

// The same with a device counter that every workgroup of the launch(es) bumps once as it starts, and the number of workgroups
// launched (added to *workgroups): wp_vp8_pipeline_* holds its finishing stream back until the two are equal.  Internal.
int wpk_vp8_recon_filter_counted(int n, const wp_vp8_frame* fh, const wp_vp8_frame* fd, const uint32_t* d_mb, const uint32_t* d_filt,
                                 const int16_t* d_coeff, void* d_work, size_t work_bytes, void* stream, uint32_t* d_started,
                                 uint32_t* workgroups) {
    int rc = ensure_init(); if (rc) return rc;
    rc = check_vp8_frames(n, fh); if (rc) return rc;
    if (n == 0) return WP_OK;
    if (!fd || !d_mb || !d_filt) return fail(WP_ERR_INVALID, "null device pointer");
    Vp8Work w;
    rc = check_vp8_work(n, fh, d_work, work_bytes, w); if (rc) return rc;
    uint8_t* work = static_cast<uint8_t*>(d_work);
    const int per = frames_per_launch(n, fh);
    for (int first = 0; first < n; first += per) {
        const int cnt = std::min(per, n - first);
        const int e = wpk_launch_vp8_recon_filter(cnt, g_cus.load(std::memory_order_relaxed), fd + first, turns_valid(cnt, fh + first) ? 1 : 0, d_mb, d_coeff,
                                                  d_filt, work, work + w.r_bytes, hipStream_t(stream), d_started, workgroups);
        if (e) return hip_fail(hipError_t(e), "vp8_mb launch");
    }
    return WP_OK;
}

int wp_vp8_finish_range_dev(int n, const wp_vp8_frame* fh, const wp_vp8_frame* fd, int first, int count, const void* d_work, size_t work_bytes,
                            uint8_t* d_y, uint8_t* d_u, uint8_t* d_v, const uint8_t* d_alpha, uint8_t* d_argb, void* stream) {
    int rc = ensure_init(); if (rc) return rc;
    rc = check_vp8_frames(n, fh); if (rc) return rc;
    if (first < 0 || count < 0 || first > n || count > n - first) return fail(WP_ERR_INVALID, "frame range outside the batch");
    if (n == 0 || count == 0) return WP_OK;
    if (!fd) return fail(WP_ERR_INVALID, "null device pointer");
    const bool planes = d_y || d_u || d_v;
    if (planes && !(d_y && d_u && d_v)) return fail(WP_ERR_INVALID, "give all three planes or none");
    if (!planes && !d_argb) return WP_OK;
    if (d_argb)
        for (int i = first; i < first + count; ++i)
            if (fh[i].has_alpha && !d_alpha) return fail(WP_ERR_INVALID, "frame has alpha but no alpha slab was given");
    Vp8Work w;                                      // the workspace layout is a property of the whole batch
    rc = check_vp8_work(n, fh, d_work, work_bytes, w); if (rc) return rc;
    return launch_finish(count, fh + first, fd + first, static_cast<const uint8_t*>(d_work), w, 0, d_y, d_u, d_v, d_alpha, d_argb, hipStream_t(stream));
}

int wp_vp8_finish_dev(int n, const wp_vp8_frame* fh, const wp_vp8_frame* fd, const void* d_work, size_t work_bytes,
                      uint8_t* d_y, uint8_t* d_u, uint8_t* d_v, const uint8_t* d_alpha, uint8_t* d_argb, void* stream) {
    return wp_vp8_finish_range_dev(n, fh, fd, 0, n, d_work, work_bytes, d_y, d_u, d_v, d_alpha, d_argb, stream);
}

int wp_yuv420_to_argb_dev(int n, const wp_vp8_frame* fh, const wp_vp8_frame* fd, const uint8_t* d_y, const uint8_t* d_u,
                          const uint8_t* d_v, const uint8_t* d_alpha, uint8_t* d_argb, void* stream) {
    int rc = ensure_init(); if (rc) return rc;
    rc = check_vp8_frames(n, fh); if (rc) return rc;
    if (n == 0) return WP_OK;
    if (!fd || !d_y || !d_u || !d_v || !d_argb) return fail(WP_ERR_INVALID, "null device pointer");
    long long max_quads = 1;
    for (int i = 0; i < n; ++i) {
        if (fh[i].has_alpha && !d_alpha) return fail(WP_ERR_INVALID, "frame has alpha but no alpha slab was given");
        max_quads = std::max(max_quads, (long long)((fh[i].width + 3) / 4) * fh[i].height);
    }
    // the y dimension of the grid is the frame index (<= 65535 per launch)
    for (int first = 0; first < n; first += 65535) {
        const int cnt = std::min(65535, n - first);
        long long gx = (max_quads + 255) / 256;
        const int slots = g_cus.load(std::memory_order_relaxed) * 32;
        const long long cap = std::max(1LL, (long long)slots / std::max(1, std::min(cnt, slots)));
        gx = std::min(gx, std::max(cap, 1LL) * 4);
        const int e = wpk_launch_yuv420_to_argb(cnt, int(gx), fd + first, d_y, d_u, d_v, d_alpha, d_argb, hipStream_t(stream));
        if (e) return hip_fail(hipError_t(e), "yuv420_to_argb launch");
    }
    return WP_OK;
}

int wp_vp8_decode_to_argb_dev(int n, const wp_vp8_frame* fh, const wp_vp8_frame* fd, const uint32_t* d_mb, const uint32_t* d_filt,
                              const int16_t* d_coeff, void* d_work, size_t work_bytes, uint8_t* d_y, uint8_t* d_u, uint8_t* d_v,
                              const uint8_t* d_alpha, uint8_t* d_argb, void* stream) {
    int rc = wp_vp8_recon_filter_dev(n, fh, fd, d_mb, d_filt, d_coeff, d_work, work_bytes, stream);      // one kernel for both
    if (rc) return rc;
    g_finish_alone = true;
    rc = wp_vp8_finish_dev(n, fh, fd, d_work, work_bytes, d_y, d_u, d_v, d_alpha, d_argb, stream);
    g_finish_alone = false;
    return rc;
}

int wp_vp8_decode_to_argb_static_dev(int n, const wp_vp8_frame* fh, const wp_vp8_frame* fd, const uint32_t* d_mb, const uint32_t* d_filt,
                                     const int16_t* d_coeff, uint8_t* d_y, uint8_t* d_u, uint8_t* d_v,
                                     const uint8_t* d_alpha, uint8_t* d_argb, void* stream) {
    int rc = ensure_init(); if (rc) return rc;
    rc = check_vp8_frames(n, fh); if (rc) return rc;
    const Vp8Work w = vp8_work(n, fh);
    if (w.total > size_t(WP_STATIC_WORK_BYTES))
        return fail(WP_ERR_UNSUPPORTED, "batch needs a workspace of " + std::to_string(w.total) + " bytes, the static one holds " +
                    std::to_string(size_t(WP_STATIC_WORK_BYTES)) + " (WP_STATIC_WORK_BYTES): use wp_vp8_decode_to_argb_dev with a caller workspace");
    void* work = nullptr;
    rc = static_scratch(&work); if (rc) return rc;
    // Nothing in the workspace has to start at zero: every tile byte the finishing pass uses is written by this call's
    // reconstruction / loop filter first (what an earlier, larger call left behind is never read).
    return wp_vp8_decode_to_argb_dev(n, fh, fd, d_mb, d_filt, d_coeff, work, size_t(WP_STATIC_WORK_BYTES), d_y, d_u, d_v, d_alpha, d_argb, stream);
}

int wp_vp8l_inverse_dev(int n, const wp_vp8l_frame* fh, const wp_vp8l_frame* fd, uint8_t* d_pix, const uint8_t* d_aux,
                        uint8_t* d_tmp, uint8_t* d_out, int swizzle, void* stream) {
    int rc = ensure_init(); if (rc) return rc;
    if (n < 0 || (n > 0 && (!fh || !fd || !d_pix || !d_out))) return fail(WP_ERR_INVALID, "bad argument");
    if (swizzle < WP_SWIZZLE_NONE || swizzle > WP_SWIZZLE_GREEN) return fail(WP_ERR_INVALID, "bad swizzle");
    for (int i = 0; i < n; ++i) {
        const wp_vp8l_frame& a = fh[i];
        // 14-bit dimensions + 1 (Vp8LDecoder.hx header): the kernels count pixels in 31 bits and a row's progress in 15
        if (a.width <= 0 || a.height <= 0 || a.width > 16384 || a.height > 16384 || a.coded_width <= 0 || a.coded_width > 16384 ||
            a.n_transforms < 0 || a.n_transforms > WP_VP8L_MAX_TRANSFORMS ||
            (a.pix_off & 15) || (a.out_off & 15) || (a.tmp_off & 15))
            return fail(WP_ERR_INVALID, "bad wp_vp8l_frame at index " + std::to_string(i));
        // The kernels derive every buffer extent from the transform list: check that the list is consistent with the frame
        // (widths chain from `width` down to `coded_width` as in Vp8LDecoder.hx:58-121) and that each transform's data is
        // as long as the kernel that undoes it will read.
        long long w = a.width;
        for (int k = 0; k < a.n_transforms; ++k) {
            const wp_vp8l_transform& t = a.t[k];
            if (t.type < 0 || t.type > 3 || t.bits < 0 || t.bits > 9 || (t.data_off & 3) || t.old_width != w)
                return fail(WP_ERR_INVALID, "bad transform in wp_vp8l_frame " + std::to_string(i));
            unsigned long long need = 0;
            if (t.type == 0 || t.type == 1) {
                if (t.bits < 2) return fail(WP_ERR_INVALID, "bad transform in wp_vp8l_frame " + std::to_string(i));
                const long long tw = (w + (1LL << t.bits) - 1) >> t.bits, th = ((long long)a.height + (1LL << t.bits) - 1) >> t.bits;
                need = 4ull * (unsigned long long)tw * (unsigned long long)th;
            } else if (t.type == 3) {
                if (t.bits > 3) return fail(WP_ERR_INVALID, "bad transform in wp_vp8l_frame " + std::to_string(i));
                need = 4ull << (8 >> t.bits);
                w = (w + (1LL << t.bits) - 1) >> t.bits;
            }
            if (t.data_len < need) return fail(WP_ERR_INVALID, "transform data too short in wp_vp8l_frame " + std::to_string(i));
            if (need && !d_aux) return fail(WP_ERR_INVALID, "wp_vp8l_frame " + std::to_string(i) + " has transform data but no aux slab was given");
            if (t.type == 3 && t.bits > 0 && !d_tmp) return fail(WP_ERR_INVALID, "wp_vp8l_frame " + std::to_string(i) + " expands packed pixels but no tmp slab was given");
        }
        if (w != a.coded_width) return fail(WP_ERR_INVALID, "coded_width of wp_vp8l_frame " + std::to_string(i) + " does not follow from its transforms");
    }
    if (n == 0) return WP_OK;
    const int e = wpk_launch_vp8l(n, fh, fd, d_pix, d_aux, d_tmp, d_out, swizzle, hipStream_t(stream));
    if (e) return hip_fail(hipError_t(e), "vp8l launch");
    return WP_OK;
}

static int bad_anim_args(int n, const wp_anim_frame* f, int cw, int ch) {
    if (n < 0 || (n > 0 && !f) || cw <= 0 || ch <= 0 || cw > (1 << 24) || ch > (1 << 24)) return 1;      // 24-bit canvas size (VP8X)
    for (int i = 0; i < n; ++i)
        if (f[i].x < 0 || f[i].y < 0 || f[i].x >= (1 << 24) || f[i].y >= (1 << 24) || f[i].width <= 0 || f[i].height <= 0 ||
            f[i].width > (1 << 24) || f[i].height > (1 << 24) || (f[i].argb_off & 3)) return 1;
    return 0;
}

int wp_anim_compose_dev(int n, const wp_anim_frame* fh, const wp_anim_frame* fd, int canvas_w, int canvas_h, const uint8_t* d_argb,
                        uint8_t* d_canvases, void* stream) {
    int rc = ensure_init(); if (rc) return rc;
    if (bad_anim_args(n, fh, canvas_w, canvas_h) || (n > 0 && (!fd || !d_argb || !d_canvases))) return fail(WP_ERR_INVALID, "bad argument");
    if (n == 0) return WP_OK;
    const int e = wpk_launch_anim_compose(n, fd, canvas_w, canvas_h, d_argb, d_canvases, hipStream_t(stream));
    if (e) return hip_fail(hipError_t(e), "anim compose launch");
    return WP_OK;
}

static int bad_tokens_frame(const wp_vp8l_tokens_frame& a) {
    return a.coded_width <= 0 || a.height <= 0 || a.coded_width > 16384 || a.height > 16384 || a.cc_bits < 0 || a.cc_bits > 11 ||
           (a.tok_off & 3) || (a.pix_off & 15) || a.n_words >= (uint64_t(1) << 32) - 256;
}

int wp_vp8l_resolve_dev(int n, const wp_vp8l_tokens_frame* fh, const wp_vp8l_tokens_frame* fd, const uint32_t* d_tokens, uint8_t* d_pix, void* stream) {
    int rc = ensure_init(); if (rc) return rc;
    if (n < 0 || (n > 0 && (!fh || !fd || !d_tokens || !d_pix))) return fail(WP_ERR_INVALID, "bad argument");
    for (int i = 0; i < n; ++i)
        if (bad_tokens_frame(fh[i])) return fail(WP_ERR_INVALID, "bad wp_vp8l_tokens_frame at index " + std::to_string(i));
    if (n == 0) return WP_OK;
    int max_cc = 0;
    for (int i = 0; i < n; ++i) max_cc = std::max(max_cc, fh[i].cc_bits);
    const int e = wpk_launch_vp8l_resolve(n, max_cc, g_cus.load(std::memory_order_relaxed), fd, d_tokens, d_pix, hipStream_t(stream));
    if (e) return hip_fail(hipError_t(e), "vp8l resolve launch");
    return WP_OK;
}

// The walk of the kernel without the pixels: positions and indices only (what decodePix checks, Vp8LDecoder.hx:283-289, 308-316).
int wp_vp8l_check_tokens(const wp_vp8l_tokens_frame* frame, const uint32_t* tok) {
    if (!frame || (!tok && frame->n_words)) return fail(WP_ERR_INVALID, "bad argument");
    if (bad_tokens_frame(*frame)) return fail(WP_ERR_INVALID, "bad wp_vp8l_tokens_frame");
    const uint64_t total = uint64_t(frame->coded_width) * uint64_t(frame->height), n = frame->n_words;
    const uint32_t cc_size = frame->cc_bits ? 1u << frame->cc_bits : 0u;
    uint64_t pos = 0, p = 0;
    while (pos < n) {
        const uint32_t kind = tok[pos] >> 30, arg = tok[pos] & 0x3fffffffu;
        if (kind == WP_TOK_LITERALS) {
            if (arg > n - pos - 1) return fail(WP_ERR_FORMAT, "token stream: literal run past the end of the stream");
            if (arg > total - p) return fail(WP_ERR_FORMAT, "token stream: more pixels than the image holds");
            pos += 1 + uint64_t(arg); p += arg;
        } else if (kind == WP_TOK_COPY) {
            if (n - pos < 2) return fail(WP_ERR_FORMAT, "token stream: copy without a distance");
            const uint32_t dist = tok[pos + 1];
            if (dist == 0 || dist > p || arg > total - p) return fail(WP_ERR_FORMAT, "Invalid LZ77 parameters");
            pos += 2; p += arg;
        } else if (kind == WP_TOK_CACHE) {
            if (cc_size == 0) return fail(WP_ERR_NULL, "null color cache");
            if (arg >= cc_size) return fail(WP_ERR_FORMAT, "Invalid color cache index");
            if (p >= total) return fail(WP_ERR_FORMAT, "token stream: more pixels than the image holds");
            pos += 1; p += 1;
        } else {
            return fail(WP_ERR_FORMAT, "token stream: unknown token kind");
        }
    }
    if (p != total) return fail(WP_ERR_FORMAT, "token stream: fewer pixels than the image holds");
    return WP_OK;
}

int wp_alpha_unfilter_dev(int n, const wp_alpha_frame* fh, const wp_alpha_frame* fd, const uint8_t* d_src, uint8_t* d_alpha, void* stream) {
    int rc = ensure_init(); if (rc) return rc;
    if (n < 0 || (n > 0 && (!fh || !fd || !d_src || !d_alpha))) return fail(WP_ERR_INVALID, "bad argument");
    for (int i = 0; i < n; ++i)
        if (fh[i].width <= 0 || fh[i].height <= 0 || fh[i].filter < 0 || fh[i].filter > 3)
            return fail(WP_ERR_INVALID, "bad wp_alpha_frame at index " + std::to_string(i));
    if (n == 0) return WP_OK;
    const int e = wpk_launch_alpha(n, fh, fd, d_src, d_alpha, hipStream_t(stream));
    if (e) return hip_fail(hipError_t(e), "alpha launch");
    return WP_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Host-pointer variants and the batch assembler
// ------------------------------------------------------------------------------------------------
namespace {

struct DevBuf {
    void* p = nullptr;
    size_t n = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        n = bytes;
        if (bytes == 0) bytes = 16;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) { p = nullptr; return hip_fail(e, "hipMalloc"); }
        return WP_OK;
    }
    int upload(const void* src, size_t bytes) {
        int rc = alloc(bytes); if (rc) return rc;
        if (bytes) HIP_TRY(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
        return WP_OK;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

struct Vp8Extents { size_t y = 0, u = 0, v = 0, argb = 0, alpha = 0; };
Vp8Extents vp8_extents(int n, const wp_vp8_frame* f) {
    Vp8Extents e;
    for (int i = 0; i < n; ++i) {
        const size_t ysz = size_t(256) * f[i].mbw * f[i].mbh, csz = size_t(64) * f[i].mbw * f[i].mbh;
        e.y = std::max(e.y, size_t(f[i].y_off) + ysz);
        e.u = std::max(e.u, size_t(f[i].u_off) + csz);
        e.v = std::max(e.v, size_t(f[i].v_off) + csz);
        e.argb = std::max(e.argb, size_t(f[i].argb_off) + size_t(4) * f[i].width * f[i].height);
        if (f[i].has_alpha) e.alpha = std::max(e.alpha, size_t(f[i].alpha_off) + size_t(f[i].width) * f[i].height);
    }
    return e;
}

}  // namespace

namespace {
// bounds of the record / coefficient arrays against the descriptors and the block indices inside the MB records
int check_vp8_records(int n, const wp_vp8_frame* frames, const uint32_t* mb, size_t n_mb, const int16_t* coeff, size_t n_coeff_blocks) {
    for (int i = 0; i < n; ++i) {
        if (size_t(frames[i].mb_base) + size_t(frames[i].mbw) * frames[i].mbh > n_mb) return fail(WP_ERR_INVALID, "mb_base out of range");
        if (!mb) continue;
        if (frames[i].coeff_base > n_coeff_blocks) return fail(WP_ERR_INVALID, "coeff_base out of range");
        size_t avail = n_coeff_blocks - size_t(frames[i].coeff_base);
        if (frames[i].flags & WP_FRAME_COEFF_BOUNDED) {
            if (size_t(frames[i].coeff_blocks) > avail) return fail(WP_ERR_INVALID, "coeff_blocks of frame " + std::to_string(i) + " reaches beyond the coefficient array");
            avail = size_t(frames[i].coeff_blocks);
        }
        if (avail && !coeff) return fail(WP_ERR_INVALID, "null coefficient array");
        const uint32_t* rec = mb + size_t(frames[i].mb_base) * WP_MB_WORDS;
        for (size_t m = 0, e = size_t(frames[i].mbw) * frames[i].mbh; m < e; ++m, rec += WP_MB_WORDS) {
            const size_t cnt = size_t(__builtin_popcount(rec[2] & 0x00ffffffu)) + ((rec[2] >> WP_MB_WHT_BIT) & 1u);
            if (cnt && size_t(rec[3]) + cnt > avail)
                return fail(WP_ERR_INVALID, "coefficient block index out of range in frame " + std::to_string(i) + ", macroblock " + std::to_string(m));
        }
    }
    return WP_OK;
}
// The host-pointer entry points know the coefficient array's length: every frame that did not bring a bound of its own gets
// the rest of the array as its bound, so the device never reads beyond the uploaded slab.
void bound_frames(std::vector<wp_vp8_frame>& f, size_t n_coeff_blocks) {
    for (wp_vp8_frame& a : f) {
        if (a.flags & WP_FRAME_COEFF_BOUNDED) continue;
        a.coeff_blocks = uint32_t(std::min<uint64_t>(0xffffffffull, uint64_t(n_coeff_blocks) - a.coeff_base));
        a.flags |= WP_FRAME_COEFF_BOUNDED;
    }
}
}  // namespace

extern "C" {

int wp_vp8_decode_to_argb(int n, const wp_vp8_frame* frames, const uint32_t* mb, size_t n_mb, const uint32_t* filt,
                          const int16_t* coeff, size_t n_coeff_blocks, uint8_t* out_y, uint8_t* out_u, uint8_t* out_v,
                          const uint8_t* alpha, uint8_t* out_argb) {
    int rc = ensure_init(); if (rc) return rc;
    rc = check_vp8_frames(n, frames); if (rc) return rc;
    if (n == 0) return WP_OK;
    if (!mb || !filt) return fail(WP_ERR_INVALID, "null record array");
    // the kernels follow the block indices in the MB records: check them here, where the records are host memory
    // (the _dev entry points take device buffers on trust)
    if ((rc = check_vp8_records(n, frames, mb, n_mb, coeff, n_coeff_blocks))) return rc;
    const Vp8Extents ex = vp8_extents(n, frames);
    std::vector<wp_vp8_frame> planned;                          // the caller's descriptors stay untouched
    try { planned.assign(frames, frames + n); } catch (const std::bad_alloc&) { return fail(WP_ERR_NOMEM, "out of memory"); }
    uint64_t work_bytes = 0;
    if ((rc = wp_vp8_plan(n, planned.data(), &work_bytes))) return rc;
    bound_frames(planned, n_coeff_blocks);
    frames = planned.data();
    const bool want_planes = out_y || out_u || out_v;
    DevBuf dfr, dmb, dfl, dco, dy, du, dv, da, dargb, dwork;
    if ((rc = dfr.upload(frames, sizeof(wp_vp8_frame) * size_t(n)))) return rc;
    if ((rc = dwork.alloc(size_t(work_bytes)))) return rc;
    if ((rc = dmb.upload(mb, n_mb * 16))) return rc;
    if ((rc = dfl.upload(filt, n_mb * 4))) return rc;
    if ((rc = dco.upload(coeff, n_coeff_blocks * 32))) return rc;
    if (want_planes && ((rc = dy.alloc(ex.y)) || (rc = du.alloc(ex.u)) || (rc = dv.alloc(ex.v)))) return rc;
    if (ex.alpha) { if (!alpha) return fail(WP_ERR_INVALID, "alpha slab missing"); if ((rc = da.upload(alpha, ex.alpha))) return rc; }
    if (out_argb && (rc = dargb.alloc(ex.argb))) return rc;
    rc = wp_vp8_decode_to_argb_dev(n, frames, dfr.as<wp_vp8_frame>(), dmb.as<uint32_t>(), dfl.as<uint32_t>(), dco.as<int16_t>(),
                                   dwork.p, size_t(work_bytes),
                                   want_planes ? dy.as<uint8_t>() : nullptr, want_planes ? du.as<uint8_t>() : nullptr,
                                   want_planes ? dv.as<uint8_t>() : nullptr, ex.alpha ? da.as<uint8_t>() : nullptr,
                                   out_argb ? dargb.as<uint8_t>() : nullptr, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    if (out_y) HIP_TRY(hipMemcpy(out_y, dy.p, ex.y, hipMemcpyDeviceToHost));
    if (out_u) HIP_TRY(hipMemcpy(out_u, du.p, ex.u, hipMemcpyDeviceToHost));
    if (out_v) HIP_TRY(hipMemcpy(out_v, dv.p, ex.v, hipMemcpyDeviceToHost));
    if (out_argb) HIP_TRY(hipMemcpy(out_argb, dargb.p, ex.argb, hipMemcpyDeviceToHost));
    return WP_OK;
}

int wp_vp8_recon(int n, const wp_vp8_frame* frames, const uint32_t* mb, size_t n_mb, const int16_t* coeff, size_t n_coeff_blocks,
                 uint8_t* out_y, uint8_t* out_u, uint8_t* out_v) {
    int rc = ensure_init(); if (rc) return rc;
    rc = check_vp8_frames(n, frames); if (rc) return rc;
    if (n == 0) return WP_OK;
    if (!mb || !out_y || !out_u || !out_v) return fail(WP_ERR_INVALID, "null pointer");
    if ((rc = check_vp8_records(n, frames, mb, n_mb, coeff, n_coeff_blocks))) return rc;
    const Vp8Extents ex = vp8_extents(n, frames);
    std::vector<wp_vp8_frame> planned;
    try { planned.assign(frames, frames + n); } catch (const std::bad_alloc&) { return fail(WP_ERR_NOMEM, "out of memory"); }
    uint64_t work_bytes = 0;
    if ((rc = wp_vp8_plan(n, planned.data(), &work_bytes))) return rc;
    bound_frames(planned, n_coeff_blocks);
    DevBuf dfr, dmb, dco, dy, du, dv, dwork;
    if ((rc = dfr.upload(planned.data(), sizeof(wp_vp8_frame) * size_t(n))) || (rc = dwork.alloc(size_t(work_bytes))) ||
        (rc = dmb.upload(mb, n_mb * 16)) || (rc = dco.upload(coeff, n_coeff_blocks * 32)) ||
        (rc = dy.alloc(ex.y)) || (rc = du.alloc(ex.u)) || (rc = dv.alloc(ex.v))) return rc;
    rc = wp_vp8_recon_dev(n, planned.data(), dfr.as<wp_vp8_frame>(), dmb.as<uint32_t>(), dco.as<int16_t>(), dwork.p, size_t(work_bytes),
                          dy.as<uint8_t>(), du.as<uint8_t>(), dv.as<uint8_t>(), nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out_y, dy.p, ex.y, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_u, du.p, ex.u, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_v, dv.p, ex.v, hipMemcpyDeviceToHost));
    return WP_OK;
}

int wp_vp8_loopfilter(int n, const wp_vp8_frame* frames, const uint32_t* filt, size_t n_mb, uint8_t* y, uint8_t* u, uint8_t* v) {
    int rc = ensure_init(); if (rc) return rc;
    rc = check_vp8_frames(n, frames); if (rc) return rc;
    if (n == 0) return WP_OK;
    if (!filt || !y || !u || !v) return fail(WP_ERR_INVALID, "null pointer");
    if ((rc = check_vp8_records(n, frames, nullptr, n_mb, nullptr, 0))) return rc;
    const Vp8Extents ex = vp8_extents(n, frames);
    std::vector<wp_vp8_frame> planned;
    try { planned.assign(frames, frames + n); } catch (const std::bad_alloc&) { return fail(WP_ERR_NOMEM, "out of memory"); }
    uint64_t work_bytes = 0;
    if ((rc = wp_vp8_plan(n, planned.data(), &work_bytes))) return rc;
    DevBuf dfr, dfl, dy, du, dv, dwork;
    if ((rc = dfr.upload(planned.data(), sizeof(wp_vp8_frame) * size_t(n))) || (rc = dwork.alloc(size_t(work_bytes))) ||
        (rc = dfl.upload(filt, n_mb * 4)) || (rc = dy.upload(y, ex.y)) || (rc = du.upload(u, ex.u)) || (rc = dv.upload(v, ex.v))) return rc;
    rc = wp_vp8_loopfilter_dev(n, planned.data(), dfr.as<wp_vp8_frame>(), dfl.as<uint32_t>(), dwork.p, size_t(work_bytes),
                               dy.as<uint8_t>(), du.as<uint8_t>(), dv.as<uint8_t>(), nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(y, dy.p, ex.y, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(u, du.p, ex.u, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(v, dv.p, ex.v, hipMemcpyDeviceToHost));
    return WP_OK;
}

int wp_yuv420_to_argb(int n, const wp_vp8_frame* frames, const uint8_t* y, const uint8_t* u, const uint8_t* v,
                      const uint8_t* alpha, uint8_t* out_argb) {
    int rc = ensure_init(); if (rc) return rc;
    rc = check_vp8_frames(n, frames); if (rc) return rc;
    if (n == 0) return WP_OK;
    if (!y || !u || !v || !out_argb) return fail(WP_ERR_INVALID, "null pointer");
    const Vp8Extents ex = vp8_extents(n, frames);
    DevBuf dfr, dy, du, dv, da, dargb;
    if ((rc = dfr.upload(frames, sizeof(wp_vp8_frame) * size_t(n)))) return rc;
    if ((rc = dy.upload(y, ex.y)) || (rc = du.upload(u, ex.u)) || (rc = dv.upload(v, ex.v))) return rc;
    if (ex.alpha) { if (!alpha) return fail(WP_ERR_INVALID, "alpha slab missing"); if ((rc = da.upload(alpha, ex.alpha))) return rc; }
    if ((rc = dargb.alloc(ex.argb))) return rc;
    rc = wp_yuv420_to_argb_dev(n, frames, dfr.as<wp_vp8_frame>(), dy.as<uint8_t>(), du.as<uint8_t>(), dv.as<uint8_t>(),
                               ex.alpha ? da.as<uint8_t>() : nullptr, dargb.as<uint8_t>(), nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out_argb, dargb.p, ex.argb, hipMemcpyDeviceToHost));
    return WP_OK;
}

int wp_vp8l_inverse(int n, const wp_vp8l_frame* frames, const uint8_t* pix, size_t pix_len, const uint8_t* aux, size_t aux_len,
                    uint8_t* out, size_t out_len, int swizzle) {
    int rc = ensure_init(); if (rc) return rc;
    if (n < 0 || (n > 0 && (!frames || !pix || !out))) return fail(WP_ERR_INVALID, "bad argument");
    if (n == 0) return WP_OK;
    size_t tmp_len = 0;
    for (int i = 0; i < n; ++i) {
        const wp_vp8l_frame& a = frames[i];
        if (a.width <= 0 || a.height <= 0 || a.coded_width <= 0 || a.width > 16384 || a.height > 16384 || a.coded_width > 16384)
            return fail(WP_ERR_INVALID, "bad wp_vp8l_frame");
        // offsets come from the caller: extent checks must not wrap (off <= len && extent <= len - off)
        auto fits = [](uint64_t off, uint64_t extent, uint64_t len) { return off <= len && extent <= len - off; };
        if (!fits(a.pix_off, uint64_t(4) * a.coded_width * a.height, pix_len)) return fail(WP_ERR_INVALID, "pix slab too small");
        if (!fits(a.out_off, uint64_t(swizzle == WP_SWIZZLE_GREEN ? 1 : 4) * a.width * a.height, out_len))
            return fail(WP_ERR_INVALID, "out slab too small");
        for (int k = 0; k < a.n_transforms && k < WP_VP8L_MAX_TRANSFORMS; ++k)
            if (!fits(a.t[k].data_off, a.t[k].data_len, aux_len)) return fail(WP_ERR_INVALID, "aux slab too small");
        if (a.tmp_off > (uint64_t(1) << 46)) return fail(WP_ERR_INVALID, "tmp_off out of range");
        tmp_len = std::max(tmp_len, size_t(a.tmp_off) + size_t(4) * a.width * a.height);
    }
    DevBuf dfr, dpix, daux, dtmp, dout;
    if ((rc = dfr.upload(frames, sizeof(wp_vp8l_frame) * size_t(n)))) return rc;
    if ((rc = dpix.upload(pix, pix_len)) || (rc = daux.upload(aux, aux_len))) return rc;
    if ((rc = dtmp.alloc(tmp_len)) || (rc = dout.alloc(out_len))) return rc;
    rc = wp_vp8l_inverse_dev(n, frames, dfr.as<wp_vp8l_frame>(), dpix.as<uint8_t>(), daux.as<uint8_t>(), dtmp.as<uint8_t>(),
                             dout.as<uint8_t>(), swizzle, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout.p, out_len, hipMemcpyDeviceToHost));
    return WP_OK;
}

int wp_anim_compose(int n, const wp_anim_frame* frames, int canvas_w, int canvas_h, const uint8_t* argb, size_t argb_len,
                    uint8_t* canvases, size_t canvases_len) {
    int rc = ensure_init(); if (rc) return rc;
    if (bad_anim_args(n, frames, canvas_w, canvas_h) || (n > 0 && (!argb || !canvases))) return fail(WP_ERR_INVALID, "bad argument");
    if (n == 0) return WP_OK;
    auto fits = [](uint64_t off, uint64_t extent, uint64_t len) { return off <= len && extent <= len - off; };
    for (int i = 0; i < n; ++i)
        if (!fits(frames[i].argb_off, uint64_t(4) * uint64_t(frames[i].width) * uint64_t(frames[i].height), argb_len))
            return fail(WP_ERR_INVALID, "frame slab too small");
    const uint64_t need = uint64_t(4) * uint64_t(canvas_w) * uint64_t(canvas_h) * uint64_t(n);
    if (need > canvases_len) return fail(WP_ERR_INVALID, "canvas slab too small");
    DevBuf dfr, dsrc, dout;
    if ((rc = dfr.upload(frames, sizeof(wp_anim_frame) * size_t(n))) || (rc = dsrc.upload(argb, argb_len)) || (rc = dout.alloc(size_t(need)))) return rc;
    rc = wp_anim_compose_dev(n, frames, dfr.as<wp_anim_frame>(), canvas_w, canvas_h, dsrc.as<uint8_t>(), dout.as<uint8_t>(), nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(canvases, dout.p, size_t(need), hipMemcpyDeviceToHost));
    return WP_OK;
}

int wp_vp8l_resolve(int n, const wp_vp8l_tokens_frame* frames, const uint32_t* tokens, size_t n_words, uint8_t* pix, size_t pix_len) {
    int rc = ensure_init(); if (rc) return rc;
    if (n < 0 || (n > 0 && (!frames || !tokens || !pix))) return fail(WP_ERR_INVALID, "bad argument");
    if (n == 0) return WP_OK;
    auto fits = [](uint64_t off, uint64_t extent, uint64_t len) { return off <= len && extent <= len - off; };
    for (int i = 0; i < n; ++i) {
        const wp_vp8l_tokens_frame& a = frames[i];
        if (bad_tokens_frame(a)) return fail(WP_ERR_INVALID, "bad wp_vp8l_tokens_frame at index " + std::to_string(i));
        if (!fits(a.tok_off / 4, a.n_words, n_words)) return fail(WP_ERR_INVALID, "token slab too small");
        if (!fits(a.pix_off, uint64_t(4) * a.coded_width * a.height, pix_len)) return fail(WP_ERR_INVALID, "pix slab too small");
        if ((rc = wp_vp8l_check_tokens(&a, tokens + a.tok_off / 4))) return rc;
    }
    DevBuf dfr, dtok, dpix;
    if ((rc = dfr.upload(frames, sizeof(wp_vp8l_tokens_frame) * size_t(n))) || (rc = dtok.upload(tokens, n_words * 4)) ||
        (rc = dpix.upload(pix, pix_len))) return rc;      // the slab may hold other frames' pixels (host-resolved ones): they go up and come back unchanged
    rc = wp_vp8l_resolve_dev(n, frames, dfr.as<wp_vp8l_tokens_frame>(), dtok.as<uint32_t>(), dpix.as<uint8_t>(), nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(pix, dpix.p, pix_len, hipMemcpyDeviceToHost));
    return WP_OK;
}

int wp_alpha_unfilter(int n, const wp_alpha_frame* frames, const uint8_t* src, size_t src_len, uint8_t* alpha, size_t alpha_len) {
    int rc = ensure_init(); if (rc) return rc;
    if (n < 0 || (n > 0 && (!frames || !src || !alpha))) return fail(WP_ERR_INVALID, "bad argument");
    if (n == 0) return WP_OK;
    for (int i = 0; i < n; ++i) {
        const wp_alpha_frame& a = frames[i];
        if (a.width <= 0 || a.height <= 0) return fail(WP_ERR_INVALID, "bad wp_alpha_frame");
        const uint64_t px = uint64_t(a.width) * uint64_t(a.height);
        auto fits = [](uint64_t off, uint64_t extent, uint64_t len) { return off <= len && extent <= len - off; };
        if (!fits(a.src_off, px * (a.src_is_vp8l_rgba ? 4 : 1), src_len)) return fail(WP_ERR_INVALID, "source slab too small");
        if (!fits(a.dst_off, px, alpha_len)) return fail(WP_ERR_INVALID, "alpha slab too small");
    }
    DevBuf dfr, dsrc, dal;
    if ((rc = dfr.upload(frames, sizeof(wp_alpha_frame) * size_t(n)))) return rc;
    if ((rc = dsrc.upload(src, src_len)) || (rc = dal.alloc(alpha_len))) return rc;
    rc = wp_alpha_unfilter_dev(n, frames, dfr.as<wp_alpha_frame>(), dsrc.as<uint8_t>(), dal.as<uint8_t>(), nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(alpha, dal.p, alpha_len, hipMemcpyDeviceToHost));
    return WP_OK;
}

// ---- host parse objects ----
int wp_parse(const uint8_t* data, size_t len, wp_parsed** out) { return wp_parse_ex(data, len, 0, out); }
int wp_parse_ex(const uint8_t* data, size_t len, int flags, wp_parsed** out) {
    if (!out) return fail(WP_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (!data && len) return fail(WP_ERR_INVALID, "null data");
    wp_parsed* p = new (std::nothrow) wp_parsed();
    if (!p) return fail(WP_ERR_NOMEM, "out of memory");
    std::string err;
    int rc;
    try {                                   // no exception crosses the C ABI
        rc = wp::parse_webp(data, len, *p, err, flags);
    } catch (const std::bad_alloc&) {
        rc = WP_ERR_NOMEM; err = "out of memory";
    } catch (const std::exception& e) {
        rc = WP_ERR_NOMEM; err = std::string("host parser: ") + e.what();
    }
    if (rc) { delete p; return fail(rc, err); }
    *out = p;
    return WP_OK;
}
void wp_parsed_free(wp_parsed* p) { delete p; }

int wp_parse_batch(int n, const uint8_t* const* data, const size_t* len, int n_threads, wp_parsed** out, int* status) {
    return wp_parse_batch_ex(n, data, len, n_threads, 0, out, status);
}
int wp_parse_batch_ex(int n, const uint8_t* const* data, const size_t* len, int n_threads, int flags, wp_parsed** out, int* status) {
    if (n < 0 || (n > 0 && (!data || !len || !out))) return fail(WP_ERR_INVALID, "bad argument");
    if (n == 0) return WP_OK;
    if (n_threads <= 0) n_threads = int(std::max(1u, std::thread::hardware_concurrency()));
    n_threads = std::min(n_threads, n);
    for (int i = 0; i < n; ++i) out[i] = nullptr;
    std::vector<int> st;
    std::vector<std::string> msg;
    try {
        st.assign(static_cast<size_t>(n), WP_OK);
        msg.resize(static_cast<size_t>(n));
    } catch (const std::bad_alloc&) {
        return fail(WP_ERR_NOMEM, "out of memory");
    }
    std::atomic<int> next{0};
    auto worker = [&]() {
        for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
            wp_parsed* p = new (std::nothrow) wp_parsed();
            if (!p) { st[size_t(i)] = WP_ERR_NOMEM; msg[size_t(i)] = "out of memory"; continue; }
            int rc;
            try {
                rc = (!data[i] && len[i]) ? WP_ERR_INVALID : wp::parse_webp(data[i], len[i], *p, msg[size_t(i)], flags);
            } catch (const std::exception&) {     // bad_alloc and anything else the containers throw
                rc = WP_ERR_NOMEM; msg[size_t(i)].clear();
            }
            if (rc) { delete p; st[size_t(i)] = rc; } else out[i] = p;
        }
    };
    std::vector<std::thread> pool;
    try {
        for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker);
    } catch (const std::exception&) {
        // the system would not start another thread: the ones that run (and this one) share the work
    }
    worker();
    for (auto& t : pool) t.join();
    int first = WP_OK;
    for (int i = n - 1; i >= 0; --i) {
        if (status) status[i] = st[size_t(i)];
        if (st[size_t(i)]) {
            first = st[size_t(i)];
            try { g_error = "file " + std::to_string(i) + ": " + (msg[size_t(i)].empty() ? std::string("out of memory") : msg[size_t(i)]); } catch (const std::exception&) {}
        }
    }
    return first;
}
// ---- background parse job ----
struct wp_parse_job {
    int n = 0;
    const uint8_t* const* data = nullptr;
    const size_t* len = nullptr;
    wp_parsed** out = nullptr;
    int* status = nullptr;
    std::atomic<int> next{0};
    std::atomic<int> n_active{0};       // n while the job runs, 0 once it is being closed
    std::mutex m;
    std::condition_variable cv;
    std::vector<char> done;             // guarded by m
    int done_prefix = 0;                // files [0, done_prefix) are done; guarded by m
    int allow = 0;                      // files [0, allow) may be started (bounds the parsed-but-unconsumed memory); guarded by m
    int window = 0;
    std::vector<std::string> msg;
    std::vector<std::thread> pool;

    void work() {
        for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return i < allow; });
            }
            if (i >= n_active.load()) break;        // the job is being closed
            int rc = WP_ERR_NOMEM;
            wp_parsed* p = new (std::nothrow) wp_parsed();
            if (p) {
                try {
                    rc = (!data[i] && len[i]) ? WP_ERR_INVALID : wp::parse_webp(data[i], len[i], *p, msg[size_t(i)]);
                } catch (const std::exception&) {
                    rc = WP_ERR_NOMEM; msg[size_t(i)].clear();
                }
                if (rc) { delete p; p = nullptr; }
            }
            out[i] = p;
            status[i] = rc;
            {
                std::lock_guard<std::mutex> lk(m);
                done[size_t(i)] = 1;
                while (done_prefix < n && done[size_t(done_prefix)]) ++done_prefix;
            }
            cv.notify_all();
        }
    }
};

int wp_parse_job_start(int n, const uint8_t* const* data, const size_t* len, int n_threads, int window, wp_parsed** out,
                       int* status, wp_parse_job** job) {
    if (!job) return fail(WP_ERR_INVALID, "null job pointer");
    *job = nullptr;
    if (n < 0 || (n > 0 && (!data || !len || !out || !status))) return fail(WP_ERR_INVALID, "bad argument");
    wp_parse_job* j = new (std::nothrow) wp_parse_job();
    if (!j) return fail(WP_ERR_NOMEM, "out of memory");
    try {
        j->n = n; j->data = data; j->len = len; j->out = out; j->status = status;
        j->n_active.store(n);
        j->window = window > 0 ? window : n;
        j->allow = std::min(n, j->window);
        j->done.assign(size_t(n), 0);
        j->msg.resize(size_t(n));
        for (int i = 0; i < n; ++i) { out[i] = nullptr; status[i] = WP_OK; }
        if (n_threads <= 0) n_threads = int(std::max(1u, std::thread::hardware_concurrency()));
        n_threads = std::max(1, std::min(n_threads, std::max(n, 1)));
        for (int t = 0; t < n_threads && n > 0; ++t) {
            try { j->pool.emplace_back([j] { j->work(); }); } catch (const std::exception&) { break; }
        }
        if (n > 0 && j->pool.empty()) { delete j; return fail(WP_ERR_NOMEM, "could not start a parse thread"); }
    } catch (const std::bad_alloc&) {
        delete j;
        return fail(WP_ERR_NOMEM, "out of memory");
    }
    *job = j;
    return WP_OK;
}

int wp_parse_job_wait(wp_parse_job* j, int upto) {
    if (!j || upto < 0 || upto > j->n) return fail(WP_ERR_INVALID, "bad argument");
    {
        std::unique_lock<std::mutex> lk(j->m);
        if (upto + j->window > j->allow) { j->allow = std::min(j->n, upto + j->window); j->cv.notify_all(); }   // the window slides with the consumer
        j->cv.wait(lk, [&] { return j->done_prefix >= upto; });
    }
    for (int i = 0; i < upto; ++i)
        if (j->status[i]) return fail(j->status[i], "file " + std::to_string(i) + ": " + (j->msg[size_t(i)].empty() ? std::string("out of memory") : j->msg[size_t(i)]));
    return WP_OK;
}

int wp_parse_job_finish(wp_parse_job* j) {
    if (!j) return WP_OK;
    j->n_active.store(0);                // files nobody has started stay unparsed (out[i] == NULL)
    j->next.store(j->n);
    { std::lock_guard<std::mutex> lk(j->m); j->allow = j->n; }
    j->cv.notify_all();
    for (auto& t : j->pool) t.join();
    delete j;
    return WP_OK;
}

namespace {
const wp::ParsedFrame* lossy_frame(const wp_parsed* const* files, const int* frame_index, int i) {
    const wp_parsed* p = files[i];
    const int k = frame_index ? frame_index[i] : 0;
    if (!p || k < 0 || size_t(k) >= p->frames.size() || p->frames[size_t(k)].type != 0) return nullptr;
    return &p->frames[size_t(k)];
}
}  // namespace

int wp_vp8_batch_layout(int n, const wp_parsed* const* files, const int* frame_index, const int* with_alpha,
                        wp_vp8_frame* frames, wp_vp8_batch_sizes* sizes) {
    if (n < 0 || (n > 0 && (!files || !frames)) || !sizes) return fail(WP_ERR_INVALID, "bad argument");
    wp_vp8_batch_sizes z = {};
    uint64_t st = 0;
    for (int i = 0; i < n; ++i) {
        const wp::ParsedFrame* f = lossy_frame(files, frame_index, i);
        if (!f) return fail(WP_ERR_INVALID, "item " + std::to_string(i) + " is not a lossy frame of a parsed file");
        const uint64_t nmb = uint64_t(f->vp8.mbw) * uint64_t(f->vp8.mbh);
        if (z.n_mb + nmb > 0xffffffffull || st > 0xffffffffull) return fail(WP_ERR_INVALID, "batch too large for 32-bit record indices");
        wp_vp8_frame& d = frames[i];
        d = wp_vp8_frame{};
        d.mbw = f->vp8.mbw; d.mbh = f->vp8.mbh;
        // Visible size: the frame's own (ANMF header for animation frames, as test/Visual.hx uses it).  An ANMF header that
        // claims more than the VP8 stream's planes hold cannot be honoured: the reference builds the planes from the VP8
        // header (Vp8Decoder.ensureImg) and so does this frame -- alone, instead of failing the whole batch.
        d.width = (f->width > 0 && f->width <= 16 * d.mbw) ? f->width : f->vp8.width;
        d.height = (f->height > 0 && f->height <= 16 * d.mbh) ? f->height : f->vp8.height;
        d.filter_type = f->vp8.filter_type;
        d.has_alpha = (with_alpha && with_alpha[i]) ? 1 : 0;
        d.mb_base = uint32_t(z.n_mb); d.stile_base = uint32_t(st); d.coeff_base = z.n_coeff_blocks;
        if (f->vp8.coeff.size() / 16 > 0xffffffffull) return fail(WP_ERR_INVALID, "frame has more coefficient blocks than a 32-bit count holds");
        d.coeff_blocks = uint32_t(f->vp8.coeff.size() / 16); d.flags = WP_FRAME_COEFF_BOUNDED;
        d.y_off = z.y_bytes; d.u_off = z.u_bytes; d.v_off = z.v_bytes; d.argb_off = z.argb_bytes; d.alpha_off = z.alpha_bytes;
        z.n_mb += nmb;
        st += uint64_t(d.mbw + 1) * uint64_t(d.mbh + 1);
        z.n_coeff_blocks += f->vp8.coeff.size() / 16;
        // relative cost on the device: every MB is reconstructed and filtered, stored blocks add transforms, an i4 MB ten
        // dependent prediction mini-steps (measured: a q90 1080p frame, 68 % i4, costs 1.5 x a q75 frame)
        d.cost = uint32_t(std::min<uint64_t>(0xffffffffull, 24 * nmb + f->vp8.coeff.size() / 16 + 16 * uint64_t(f->vp8.n_i4)));
        z.y_bytes += align_up(size_t(256 * nmb), 256);
        z.u_bytes += align_up(size_t(64 * nmb), 256);
        z.v_bytes += align_up(size_t(64 * nmb), 256);
        z.argb_bytes += align_up(size_t(4) * size_t(d.width) * size_t(d.height), 256);
        if (d.has_alpha) z.alpha_bytes += align_up(size_t(d.width) * size_t(d.height), 256);
    }
    const int rc = check_vp8_frames(n, frames); if (rc) return rc;
    assign_turns(n, frames);
    z.work_bytes = uint64_t(vp8_work(n, frames).total);
    *sizes = z;
    return WP_OK;
}

int wp_vp8_batch_gather(int n, const wp_parsed* const* files, const int* frame_index, const wp_vp8_frame* frames,
                        uint32_t* mb, uint32_t* filt, size_t n_mb_capacity, int16_t* coeff, size_t n_coeff_blocks_capacity,
                        int n_threads) {
    if (n < 0 || (n > 0 && (!files || !frames || !mb || !filt))) return fail(WP_ERR_INVALID, "bad argument");
    for (int i = 0; i < n; ++i) {
        const wp::ParsedFrame* f = lossy_frame(files, frame_index, i);
        const size_t nmb = f ? size_t(f->vp8.mbw) * size_t(f->vp8.mbh) : 0;
        if (!f || f->vp8.mbw != frames[i].mbw || f->vp8.mbh != frames[i].mbh || f->vp8.mb.size() != nmb * WP_MB_WORDS || f->vp8.filt.size() != nmb)
            return fail(WP_ERR_INVALID, "item " + std::to_string(i) + " does not match its descriptor (wp_vp8_batch_layout)");
        if (!f->vp8.coeff.empty() && !coeff) return fail(WP_ERR_INVALID, "null coefficient slab");
        if (size_t(frames[i].mb_base) + nmb > n_mb_capacity || size_t(frames[i].coeff_base) + f->vp8.coeff.size() / 16 > n_coeff_blocks_capacity)
            return fail(WP_ERR_INVALID, "slab too small for item " + std::to_string(i));
    }
    if (n_threads <= 0) n_threads = int(std::max(1u, std::thread::hardware_concurrency()));
    n_threads = std::max(1, std::min(n_threads, n));
    std::atomic<int> next{0};
    auto worker = [&]() {
        for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
            const wp::ParsedFrame* f = lossy_frame(files, frame_index, i);
            const wp_vp8_frame& d = frames[i];
            const size_t nmb = size_t(d.mbw) * size_t(d.mbh);
            uint32_t* rec = mb + size_t(d.mb_base) * WP_MB_WORDS;
            std::memcpy(rec, f->vp8.mb.data(), nmb * WP_MB_WORDS * sizeof(uint32_t));
            std::memcpy(filt + d.mb_base, f->vp8.filt.data(), nmb * sizeof(uint32_t));
            if (!f->vp8.coeff.empty()) std::memcpy(coeff + size_t(d.coeff_base) * 16, f->vp8.coeff.data(), f->vp8.coeff.size() * sizeof(int16_t));
        }
    };
    std::vector<std::thread> pool;
    try {
        for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker);
    } catch (const std::exception&) {
        // fewer threads than asked for: the ones that run share the work
    }
    worker();
    for (auto& t : pool) t.join();
    return WP_OK;
}

int wp_parsed_kind(const wp_parsed* p) { return p ? p->kind : -1; }
int wp_parsed_width(const wp_parsed* p) { return p ? p->width : 0; }
int wp_parsed_height(const wp_parsed* p) { return p ? p->height : 0; }
int wp_parsed_nframes(const wp_parsed* p) { return p ? int(p->frames.size()) : 0; }

// info: [0] type, [1] width, [2] height, [3] mbw, [4] mbh, [5] filter_type, [6] has_alpha, [7] x, [8] y, [9] duration,
//       [10] blend, [11] dispose, [12] n_transforms, [13] coded_width, [14] alpha filter, [15] alpha is vp8l-coded
int wp_parsed_frame_info(const wp_parsed* p, int i, int32_t* info) {
    if (!p || i < 0 || size_t(i) >= p->frames.size() || !info) return fail(WP_ERR_INVALID, "bad frame index");
    const wp::ParsedFrame& f = p->frames[size_t(i)];
    info[0] = f.type; info[1] = f.width; info[2] = f.height;
    info[3] = f.vp8.mbw; info[4] = f.vp8.mbh; info[5] = f.vp8.filter_type; info[6] = f.alpha.present ? 1 : 0;
    info[7] = f.x; info[8] = f.y; info[9] = f.duration; info[10] = f.blend; info[11] = f.dispose;
    info[12] = int(f.vp8l.transforms.size()); info[13] = f.vp8l.coded_width; info[14] = f.alpha.filter; info[15] = f.alpha.vp8l ? 1 : 0;
    return WP_OK;
}

// which: 0 MB records, 1 filter records, 2 coefficient blocks, 3 VP8L pix, 4 raw alpha bytes, 5 alpha VP8L pix,
//        6 VP8L token stream (WP_PARSE_VP8L_TOKENS), 10+k: data of VP8L transform k, 20+k: data of the alpha image's transform k
const void* wp_parsed_frame_buf(const wp_parsed* p, int i, int which, size_t* len) {
    if (len) *len = 0;
    if (!p || i < 0 || size_t(i) >= p->frames.size()) return nullptr;
    const wp::ParsedFrame& f = p->frames[size_t(i)];
    auto ret = [&](const void* ptr, size_t n) { if (len) *len = n; return ptr; };
    switch (which) {
    case 0: return ret(f.vp8.mb.data(), f.vp8.mb.size() * 4);
    case 1: return ret(f.vp8.filt.data(), f.vp8.filt.size() * 4);
    case 2: return ret(f.vp8.coeff.data(), f.vp8.coeff.size() * 2);
    case 3: return ret(f.vp8l.pix.data(), f.vp8l.pix.size());
    case 4: return ret(f.alpha.raw.data(), f.alpha.raw.size());
    case 5: return ret(f.alpha.coded.pix.data(), f.alpha.coded.pix.size());
    case 6: return ret(f.vp8l.tokens.data(), f.vp8l.tokens.size() * 4);
    default:
        if (which >= 10 && which < 14 && size_t(which - 10) < f.vp8l.transforms.size())
            return ret(f.vp8l.transforms[size_t(which - 10)].pix.data(), f.vp8l.transforms[size_t(which - 10)].pix.size());
        if (which >= 20 && which < 24 && size_t(which - 20) < f.alpha.coded.transforms.size())
            return ret(f.alpha.coded.transforms[size_t(which - 20)].pix.data(), f.alpha.coded.transforms[size_t(which - 20)].pix.size());
        return nullptr;
    }
}

// meta: [0] vp8l width, [1] height, [2] n_transforms, [3] colour-cache bits of a tokenized image, [4+3k..6+3k] (type, bits, old_width) of transform k;
//       [16..18] and [20+3k..22+3k]: the same for the VP8L-coded alpha image of the frame
int wp_parsed_frame_vp8l_meta(const wp_parsed* p, int i, int32_t* meta) {
    if (!p || i < 0 || size_t(i) >= p->frames.size() || !meta) return fail(WP_ERR_INVALID, "bad frame index");
    const wp::ParsedFrame& f = p->frames[size_t(i)];
    std::memset(meta, 0, 32 * sizeof(int32_t));
    const wp::Vp8lPacked* src[2] = {&f.vp8l, &f.alpha.coded};
    for (int s = 0; s < 2; ++s) {
        int32_t* m = meta + 16 * s;
        m[0] = src[s]->width; m[1] = src[s]->height; m[2] = int32_t(src[s]->transforms.size()); m[3] = src[s]->cc_bits;
        for (size_t k = 0; k < src[s]->transforms.size() && k < 4; ++k) {
            m[4 + 3 * k] = src[s]->transforms[k].type; m[5 + 3 * k] = src[s]->transforms[k].bits; m[6 + 3 * k] = src[s]->transforms[k].old_width;
        }
    }
    if (!f.alpha.vp8l) { meta[16] = f.alpha.width; meta[17] = f.alpha.height; }
    return WP_OK;
}

// hdr: FrameHeader of a lossy frame (Image.hx:10-19): keyFrame, versionNumber, showFrame, firstPartitionLen, width, height, xScale, yScale
int wp_parsed_frame_vp8_header(const wp_parsed* p, int i, int32_t* hdr) {
    if (!p || i < 0 || size_t(i) >= p->frames.size() || !hdr) return fail(WP_ERR_INVALID, "bad frame index");
    const wp::Vp8Packed& v = p->frames[size_t(i)].vp8;
    hdr[0] = v.keyFrame; hdr[1] = v.versionNumber; hdr[2] = v.showFrame; hdr[3] = v.firstPartitionLen;
    hdr[4] = v.width; hdr[5] = v.height; hdr[6] = v.xScale; hdr[7] = v.yScale;
    return WP_OK;
}

}  // extern "C"
