// Micro-benchmark: how a wavefront that owns 64 rows of a 1920-byte-pitch plane should read / write them on MI355X.
// Per visit the wavefront moves SEG bytes of each of its 64 rows.  Mappings:
//   L (lane = row):   each lane issues SEG/16 back-to-back 16-byte accesses to its own row.
//   C (cooperative):  SEG/16 adjacent lanes cover one row segment, 64/(SEG/16) rows per instruction.
// Total bytes are identical for every variant; only the request shape changes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int SEG, bool COOP, bool LOAD>
__global__ __launch_bounds__(512) void k(uint8_t* base, int pitch, int rows_per_img, int nimg, uint32_t* sink) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int bands = rows_per_img / 64;
    const int img = wave / bands, band = wave % bands;
    if (img >= nimg) return;
    uint8_t* band0 = base + (size_t(img) * rows_per_img + band * 64) * pitch;
    constexpr int PER = SEG / 16;              // 16-byte pieces per row segment
    uint32_t acc = 0;
    for (int x = 0; x + SEG <= pitch; x += SEG) {
#pragma unroll
        for (int k2 = 0; k2 < PER; ++k2) {
            uint8_t* p = COOP ? band0 + size_t((lane / PER) + k2 * (64 / PER)) * pitch + x + 16 * (lane % PER)
                              : band0 + size_t(lane) * pitch + x + 16 * k2;
            if (LOAD) { const uint4 v = *reinterpret_cast<const uint4*>(p); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
            else *reinterpret_cast<uint4*>(p) = make_uint4(x, lane, img, band);
        }
        for (int k2 = 0; k2 < PER; ++k2) __builtin_amdgcn_s_sleep(20);   // same total idle time per byte
    }
    if (LOAD && acc == 0x12345678u) sink[0] = acc;
}
template <int SEG, bool COOP, bool LOAD>
static void run(const char* name, uint8_t* d, uint32_t* sink, int pitch, int rows, int nimg) {
    const int waves = nimg * (rows / 64), blocks = (waves * 64 + 511) / 512;
    const size_t bytes = size_t(nimg) * rows * pitch;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        float ms;
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SEG, COOP, LOAD>), dim3(blocks), dim3(512), 0, 0, d, pitch, rows, nimg, sink);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-22s %8.3f ms  %8.1f GB/s\n", name, best, bytes / best / 1e6);
}
int main() {
    const int pitch = 1920, rows = 1088, nimg = 1024;
    const size_t bytes = size_t(nimg) * rows * pitch;
    uint8_t* d; uint32_t* sink;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&sink, 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 1, bytes);
    run<16, false, false>("store 16 lane=row", d, sink, pitch, rows, nimg);
    run<32, false, false>("store 32 lane=row", d, sink, pitch, rows, nimg);
    run<32, true, false>("store 32 coop", d, sink, pitch, rows, nimg);
    run<64, false, false>("store 64 lane=row", d, sink, pitch, rows, nimg);
    run<64, true, false>("store 64 coop", d, sink, pitch, rows, nimg);
    run<128, false, false>("store 128 lane=row", d, sink, pitch, rows, nimg);
    run<128, true, false>("store 128 coop", d, sink, pitch, rows, nimg);
    run<16, false, true>("load 16 lane=row", d, sink, pitch, rows, nimg);
    run<32, false, true>("load 32 lane=row", d, sink, pitch, rows, nimg);
    run<32, true, true>("load 32 coop", d, sink, pitch, rows, nimg);
    run<64, false, true>("load 64 lane=row", d, sink, pitch, rows, nimg);
    run<64, true, true>("load 64 coop", d, sink, pitch, rows, nimg);
    run<128, false, true>("load 128 lane=row", d, sink, pitch, rows, nimg);
    run<128, true, true>("load 128 coop", d, sink, pitch, rows, nimg);
    return 0;
}
