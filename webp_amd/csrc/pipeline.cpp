// wp_vp8_pipeline_*: the software pipeline over successive lossy batches as part of the C ABI (include/webp_amd.h), so that
// a caller in any language gets the pipelined step without re-implementing its stream plumbing:
//
//     main stream      [ recon + loop filter, batch k ]      [ recon + loop filter, batch k+1 ]   ...
//     finish stream                                         [ hold-off | finishing pass, batch k ]   ...
//
// The finishing pass is bound by memory bandwidth, the one-kernel reconstruction + loop filter by instruction issue; side by
// side they fill each other's gaps.  Batches rotate through the pipeline's own tile workspaces, so the kernel
// of batch k+1 never touches the tiles the finishing pass of batch k still reads.  The finishing launch is held back until
// every workgroup of the persistent kernel beside it has been placed: both kernels become runnable at the same moment, and if
// the streaming one is dispatched first the persistent kernel's workgroups are placed piecemeal as it drains and the step
// takes a third longer.  Held back by PROGRESS: the persistent kernel's workgroups count themselves in a device word as they
// start, and a one-wavefront kernel on the finish stream polls it (with a time-out, so that a launch that failed cannot leave
// a spinning wavefront behind); rounds 1 and 2 waited a fixed 40 us, a constant of this GPU.  wp_vp8_pipeline_set_hold_us adds
// a minimum time on top (default 0).  No counterpart in the reference (one image per call, no device).
//
// Everything here is host code over the library's public *_dev entry points.
#include <hip/hip_runtime.h>
#include <deque>
#include <new>
#include <string>
#include <vector>
#include "../../include/webp_amd.h"

namespace wp { void set_error(const std::string& msg); }
extern "C" int wpk_launch_delay(int us, hipStream_t stream);
extern "C" int wpk_launch_hold_until(const uint32_t* d_counter, uint32_t target, int min_us, int max_us, hipStream_t stream);
extern "C" int wpk_vp8_recon_filter_counted(int n, const wp_vp8_frame* fh, const wp_vp8_frame* fd, const uint32_t* d_mb, const uint32_t* d_filt,
                                            const int16_t* d_coeff, void* d_work, size_t work_bytes, void* stream, uint32_t* d_started,
                                            uint32_t* workgroups);

struct wp_vp8_pipeline {
    int device = 0;
    hipStream_t s_main = nullptr, s_fin = nullptr;
    size_t ws_bytes = 0;
    std::vector<void*> ws;
    int hold_us = 0;       // least time the finishing pass is held back behind the launch of the next batch's persistent kernel, on top of the wait for its workgroups
    uint32_t* d_started = nullptr;                 // kCounters device words, one per step in flight (step k uses k % kCounters)
    int max_in_flight = 3;
    bool timing = false;
    uint64_t k = 0;
    struct Step {
        std::vector<wp_vp8_frame> fh;
        const wp_vp8_frame* fd = nullptr;
        void* work = nullptr;
        uint8_t *y = nullptr, *u = nullptr, *v = nullptr, *argb = nullptr;
        const uint8_t* alpha = nullptr;
        hipEvent_t alpha_ready = nullptr;             // caller's event: the alpha slab is complete (the finishing pass waits for it)
        hipEvent_t mb0 = nullptr, mb1 = nullptr;      // around the reconstruction + loop filter kernel (main stream)
        uint32_t* started = nullptr;                  // this step's counter of placed workgroups ...
        uint32_t workgroups = 0;                      // ... and what it reaches
        hipEvent_t fin0 = nullptr, fin1 = nullptr;    // around the finishing pass (finish stream); fin1 = the step's outputs are complete
    };
    Step pending;                  // its finishing pass has not been enqueued yet
    bool has_pending = false;
    std::deque<Step> queued;       // finishing pass enqueued, not yet known to be complete; oldest first
    double ms_mb = 0, ms_fin = 0;  // accumulated over completed steps (timing pipelines only)
    long steps_done = 0;
};

namespace {

int fail(int code, const std::string& msg) { wp::set_error(msg); return code; }
int hip_fail(hipError_t e, const char* what) { return fail(WP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return hip_fail(e_, #expr); } while (0)

int new_event(const wp_vp8_pipeline* p, hipEvent_t* e) {
    HIP_TRY(hipEventCreateWithFlags(e, p->timing ? hipEventDefault : hipEventDisableTiming));
    return WP_OK;
}

void free_step_events(wp_vp8_pipeline::Step& s) {
    for (hipEvent_t* e : {&s.mb0, &s.mb1, &s.fin0, &s.fin1})
        if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
}

// the oldest queued step is complete: account for it and release its events
void retire_front(wp_vp8_pipeline* p) {
    wp_vp8_pipeline::Step& s = p->queued.front();
    if (p->timing) {
        float a = 0, b = 0;
        if (hipEventElapsedTime(&a, s.mb0, s.mb1) == hipSuccess && hipEventElapsedTime(&b, s.fin0, s.fin1) == hipSuccess) {
            p->ms_mb += a; p->ms_fin += b; p->steps_done += 1;
        }
    }
    free_step_events(s);
    p->queued.pop_front();
}

constexpr int kCounters = 8;                // > max_in_flight + 2: a counter is long out of use when its turn comes again
constexpr int kHoldTimeoutUs = 2000;        // the poll gives up after this long whatever the counter says

// enqueue the finishing pass of the pending step on the finish stream; `beside`: the step whose persistent kernel has just
// been launched and runs beside it (null: nothing does)
int enqueue_finish(wp_vp8_pipeline* p, const wp_vp8_pipeline::Step* beside) {
    wp_vp8_pipeline::Step& s = p->pending;
    HIP_TRY(hipStreamWaitEvent(p->s_fin, s.mb1, 0));
    if (s.alpha_ready) HIP_TRY(hipStreamWaitEvent(p->s_fin, s.alpha_ready, 0));
    if (beside) {
        // mb0 of that step was recorded after its counter was cleared and directly before its launch: polling starts then
        HIP_TRY(hipStreamWaitEvent(p->s_fin, beside->mb0, 0));
        const int e = wpk_launch_hold_until(beside->started, beside->workgroups, p->hold_us, kHoldTimeoutUs + p->hold_us, p->s_fin);
        if (e) return hip_fail(hipError_t(e), "hold-off launch");
    }
    int rc = new_event(p, &s.fin0); if (rc) return rc;
    rc = new_event(p, &s.fin1); if (rc) return rc;
    HIP_TRY(hipEventRecord(s.fin0, p->s_fin));
    rc = wp_vp8_finish_dev(int(s.fh.size()), s.fh.data(), s.fd, s.work, p->ws_bytes, s.y, s.u, s.v, s.alpha, s.argb, p->s_fin);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(s.fin1, p->s_fin));
    p->queued.push_back(std::move(s));
    p->pending = wp_vp8_pipeline::Step();
    p->has_pending = false;
    return WP_OK;
}

struct DeviceGuard {           // the pipeline's calls run with its device current, whatever the caller's thread had selected
    int prev = -1;
    bool switched = false;
    int enter(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) { HIP_TRY(hipSetDevice(dev)); switched = true; }
        return WP_OK;
    }
    ~DeviceGuard() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

extern "C" {

int wp_vp8_pipeline_create(int n_workspaces, size_t work_bytes, int flags, wp_vp8_pipeline** out) {
    if (!out) return fail(WP_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (n_workspaces < 2 || n_workspaces > 8) return fail(WP_ERR_INVALID, "a pipeline rotates through 2..8 workspaces");
    int rc = wp_init(-1); if (rc) return rc;          // the calling thread's current device
    wp_vp8_pipeline* p = new (std::nothrow) wp_vp8_pipeline();
    if (!p) return fail(WP_ERR_NOMEM, "out of memory");
    p->timing = (flags & WP_PIPELINE_TIMING) != 0;
    p->ws_bytes = work_bytes;
    auto bail = [&](int code) { (void)wp_vp8_pipeline_destroy(p); return code; };
    if (hipGetDevice(&p->device) != hipSuccess) return bail(fail(WP_ERR_HIP, "hipGetDevice"));
    // default priority, on purpose: a high-priority finishing stream once stalled for good behind the stream it waited for
    // 
    if (hipStreamCreateWithFlags(&p->s_main, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&p->s_fin, hipStreamNonBlocking) != hipSuccess) return bail(fail(WP_ERR_HIP, "hipStreamCreate"));
    try { p->ws.assign(size_t(n_workspaces), nullptr); } catch (const std::bad_alloc&) { return bail(fail(WP_ERR_NOMEM, "out of memory")); }
    for (void*& w : p->ws) {
        const hipError_t e = hipMalloc(&w, work_bytes ? work_bytes : 256);
        if (e != hipSuccess) { w = nullptr; return bail(hip_fail(e, "hipMalloc (pipeline workspace)")); }
    }
    if (hipMalloc(reinterpret_cast<void**>(&p->d_started), 256 * kCounters) != hipSuccess) { p->d_started = nullptr; return bail(fail(WP_ERR_HIP, "hipMalloc (pipeline counters)")); }
    *out = p;
    return WP_OK;
}

int wp_vp8_pipeline_submit(wp_vp8_pipeline* p, int n, const wp_vp8_frame* fh, const wp_vp8_frame* fd, const uint32_t* d_mb,
                           const uint32_t* d_filt, const int16_t* d_coeff, uint8_t* d_y, uint8_t* d_u, uint8_t* d_v,
                           const uint8_t* d_alpha, uint8_t* d_argb, void* inputs_ready, void* alpha_ready) {
    if (!p) return fail(WP_ERR_INVALID, "null pipeline");
    if (n <= 0 || !fh || !fd) return fail(WP_ERR_INVALID, "a pipeline step needs at least one frame");
    DeviceGuard g; int rc = g.enter(p->device); if (rc) return rc;
    // at most max_in_flight steps queued ahead of the GPU (the host blocks here, not in the launches)
    while (int(p->queued.size()) >= p->max_in_flight) {
        HIP_TRY(hipEventSynchronize(p->queued.front().fin1));
        retire_front(p);
    }
    while (!p->queued.empty() && hipEventQuery(p->queued.front().fin1) == hipSuccess) retire_front(p);      // complete already
    wp_vp8_pipeline::Step s;
    try { s.fh.assign(fh, fh + n); } catch (const std::bad_alloc&) { return fail(WP_ERR_NOMEM, "out of memory"); }
    s.fd = fd; s.y = d_y; s.u = d_u; s.v = d_v; s.alpha = d_alpha; s.argb = d_argb;
    s.alpha_ready = hipEvent_t(alpha_ready);
    const size_t n_ws = p->ws.size();
    s.work = p->ws[size_t(p->k % n_ws)];
    // The finishing pass that last read this workspace belongs to the step submitted n_ws steps ago.  Steps complete in
    // order on the finish stream, so it is enough to wait for the most recent finishing pass at or before that step.
    {
        const uint64_t want = p->k >= n_ws ? p->k - n_ws : ~uint64_t(0);      // index of the step whose finishing pass must be over
        if (want != ~uint64_t(0)) {
            // queued holds steps [k - has_pending - queued.size(), k - has_pending) in order
            const uint64_t first = p->k - (p->has_pending ? 1 : 0) - p->queued.size();
            if (want >= first) {
                const size_t idx = size_t(want - first);
                if (idx < p->queued.size()) HIP_TRY(hipStreamWaitEvent(p->s_main, p->queued[idx].fin1, 0));
                else return fail(WP_ERR_INVALID, "internal: workspace hazard on a step whose finishing pass is not enqueued");
            }                                                                  // else: already retired, i.e. complete
        }
    }
    if (inputs_ready) HIP_TRY(hipStreamWaitEvent(p->s_main, hipEvent_t(inputs_ready), 0));
    rc = new_event(p, &s.mb0); if (rc) return rc;
    rc = new_event(p, &s.mb1); if (rc) { free_step_events(s); return rc; }
    s.started = p->d_started + 64 * (p->k % kCounters);                         // (256 bytes apart: a line of its own)
    hipError_t he = hipMemsetAsync(s.started, 0, sizeof(uint32_t), p->s_main);
    if (he == hipSuccess) he = hipEventRecord(s.mb0, p->s_main);
    if (he != hipSuccess) { free_step_events(s); return hip_fail(he, "pipeline step set-up"); }
    rc = wpk_vp8_recon_filter_counted(n, s.fh.data(), fd, d_mb, d_filt, d_coeff, s.work, p->ws_bytes, p->s_main, s.started, &s.workgroups);
    if (rc) { free_step_events(s); return rc; }
    he = hipEventRecord(s.mb1, p->s_main);
    if (he != hipSuccess) { free_step_events(s); return hip_fail(he, "hipEventRecord"); }
    p->k += 1;
    // the finishing pass of the step before goes out now, beside the kernel just launched
    if (p->has_pending) { rc = enqueue_finish(p, &s); if (rc) { free_step_events(s); return rc; } }
    p->pending = std::move(s);
    p->has_pending = true;
    return WP_OK;
}

int wp_vp8_pipeline_flush(wp_vp8_pipeline* p, void* stream) {
    if (!p) return fail(WP_ERR_INVALID, "null pipeline");
    DeviceGuard g; int rc = g.enter(p->device); if (rc) return rc;
    if (p->has_pending) { rc = enqueue_finish(p, nullptr); if (rc) return rc; }     // nothing left to overlap it with: no hold-off
    if (!p->queued.empty()) HIP_TRY(hipStreamWaitEvent(hipStream_t(stream), p->queued.back().fin1, 0));
    return WP_OK;
}

int wp_vp8_pipeline_drain(wp_vp8_pipeline* p) {
    if (!p) return fail(WP_ERR_INVALID, "null pipeline");
    DeviceGuard g; int rc = g.enter(p->device); if (rc) return rc;
    if (p->has_pending) { rc = enqueue_finish(p, nullptr); if (rc) return rc; }
    HIP_TRY(hipStreamSynchronize(p->s_main));
    HIP_TRY(hipStreamSynchronize(p->s_fin));
    while (!p->queued.empty()) retire_front(p);
    return WP_OK;
}

int wp_vp8_pipeline_stats(wp_vp8_pipeline* p, double* ms_recon_filter, double* ms_finish, long* steps, int reset) {
    if (!p) return fail(WP_ERR_INVALID, "null pipeline");
    if (!p->timing) return fail(WP_ERR_INVALID, "pipeline was created without WP_PIPELINE_TIMING");
    if (ms_recon_filter) *ms_recon_filter = p->ms_mb;
    if (ms_finish) *ms_finish = p->ms_fin;
    if (steps) *steps = p->steps_done;
    if (reset) { p->ms_mb = p->ms_fin = 0; p->steps_done = 0; }
    return WP_OK;
}

int wp_vp8_pipeline_set_hold_us(wp_vp8_pipeline* p, int us) {
    if (!p || us < 0 || us > 10000) return fail(WP_ERR_INVALID, "bad argument");
    p->hold_us = us;
    return WP_OK;
}

int wp_vp8_pipeline_destroy(wp_vp8_pipeline* p) {
    if (!p) return WP_OK;
    DeviceGuard g; (void)g.enter(p->device);
    if (p->s_main) (void)hipStreamSynchronize(p->s_main);
    if (p->s_fin) (void)hipStreamSynchronize(p->s_fin);
    if (p->has_pending) free_step_events(p->pending);
    for (auto& s : p->queued) free_step_events(s);
    for (void* w : p->ws) if (w) (void)hipFree(w);
    if (p->d_started) (void)hipFree(p->d_started);
    if (p->s_main) (void)hipStreamDestroy(p->s_main);
    if (p->s_fin) (void)hipStreamDestroy(p->s_fin);
    delete p;
    return WP_OK;
}

}  // extern "C"
