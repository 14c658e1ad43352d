/* A plain C program that pipelines batches through wp_vp8_pipeline_* (include/webp_amd.h) with device memory from the HIP
 * runtime's C API -- what a serving loop in C, or a binding for a language without HIP streams of its own, would write:
 *   for every batch: upload the packed buffers, submit;  drain;  download.
 * Three one-frame batches (the same file three times, separate device buffers) go through a two-workspace pipeline, so
 * that the workspaces are re-used and the finishing pass of each batch runs beside the kernel of the next.  Writes Y, U,
 * V and A,R,G,B of every batch to a file for the test to compare with the oracle.
 *   c_pipeline <in.webp> <out.bin>      exit status 0 = ok; on failure prints wp_last_error. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "webp_amd.h"

#define NB 3

static int die(const char* what) {
    char msg[512];
    wp_last_error(msg, (int)sizeof msg);
    fprintf(stderr, "%s: %s\n", what, msg);
    return 1;
}
#define HIPCHK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: c_pipeline in.webp out.bin\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END);
    long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t* data = (uint8_t*)malloc((size_t)len);
    if (!data || fread(data, 1, (size_t)len, f) != (size_t)len) { fprintf(stderr, "read failed\n"); return 2; }
    fclose(f);

    wp_parsed* p = NULL;
    if (wp_parse(data, (size_t)len, &p) != WP_OK) return die("wp_parse");
    const wp_parsed* files[1];
    wp_vp8_frame frame;
    wp_vp8_batch_sizes sz;
    files[0] = p;
    if (wp_vp8_batch_layout(1, files, NULL, NULL, &frame, &sz) != WP_OK) return die("wp_vp8_batch_layout");
    size_t n_blocks = (size_t)sz.n_coeff_blocks ? (size_t)sz.n_coeff_blocks : 1;
    uint32_t* mb = (uint32_t*)malloc((size_t)sz.n_mb * 16);
    uint32_t* filt = (uint32_t*)malloc((size_t)sz.n_mb * 4);
    int16_t* coeff = (int16_t*)calloc(n_blocks, 32);
    if (wp_vp8_batch_gather(1, files, NULL, &frame, mb, filt, (size_t)sz.n_mb, coeff, n_blocks, 1) != WP_OK) return die("wp_vp8_batch_gather");

    if (wp_init(0) != WP_OK) return die("wp_init");
    wp_vp8_pipeline* pipe = NULL;
    if (wp_vp8_pipeline_create(2, (size_t)sz.work_bytes, 0, &pipe) != WP_OK) return die("wp_vp8_pipeline_create");

    void *d_frame[NB], *d_mb[NB], *d_filt[NB], *d_coeff[NB], *d_y[NB], *d_u[NB], *d_v[NB], *d_argb[NB];
    int b;
    for (b = 0; b < NB; ++b) {
        HIPCHK(hipMalloc(&d_frame[b], sizeof frame));
        HIPCHK(hipMalloc(&d_mb[b], (size_t)sz.n_mb * 16));
        HIPCHK(hipMalloc(&d_filt[b], (size_t)sz.n_mb * 4));
        HIPCHK(hipMalloc(&d_coeff[b], n_blocks * 32));
        HIPCHK(hipMalloc(&d_y[b], (size_t)sz.y_bytes));
        HIPCHK(hipMalloc(&d_u[b], (size_t)sz.u_bytes));
        HIPCHK(hipMalloc(&d_v[b], (size_t)sz.v_bytes));
        HIPCHK(hipMalloc(&d_argb[b], (size_t)sz.argb_bytes));
        HIPCHK(hipMemcpy(d_frame[b], &frame, sizeof frame, hipMemcpyHostToDevice));      /* blocking copies: inputs are ready at submit */
        HIPCHK(hipMemcpy(d_mb[b], mb, (size_t)sz.n_mb * 16, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_filt[b], filt, (size_t)sz.n_mb * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_coeff[b], coeff, n_blocks * 32, hipMemcpyHostToDevice));
        HIPCHK(hipMemset(d_argb[b], 0, (size_t)sz.argb_bytes));
        if (wp_vp8_pipeline_submit(pipe, 1, &frame, (const wp_vp8_frame*)d_frame[b], (const uint32_t*)d_mb[b], (const uint32_t*)d_filt[b],
                                   (const int16_t*)d_coeff[b], (uint8_t*)d_y[b], (uint8_t*)d_u[b], (uint8_t*)d_v[b], NULL,
                                   (uint8_t*)d_argb[b], NULL, NULL) != WP_OK) return die("wp_vp8_pipeline_submit");
    }
    if (wp_vp8_pipeline_drain(pipe) != WP_OK) return die("wp_vp8_pipeline_drain");

    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 2; }
    int32_t hdr[4];
    hdr[0] = frame.width; hdr[1] = frame.height; hdr[2] = frame.mbw; hdr[3] = frame.mbh;
    fwrite(hdr, sizeof hdr, 1, o);
    size_t ny = (size_t)256 * frame.mbw * frame.mbh, nc = (size_t)64 * frame.mbw * frame.mbh, na = (size_t)4 * frame.width * frame.height;
    uint8_t* host = (uint8_t*)malloc(ny + 2 * nc + na);
    for (b = 0; b < NB; ++b) {
        HIPCHK(hipMemcpy(host, d_y[b], ny, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(host + ny, d_u[b], nc, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(host + ny + nc, d_v[b], nc, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(host + ny + 2 * nc, d_argb[b], na, hipMemcpyDeviceToHost));
        fwrite(host, 1, ny + 2 * nc + na, o);
    }
    fclose(o);
    if (wp_vp8_pipeline_destroy(pipe) != WP_OK) return die("wp_vp8_pipeline_destroy");
    wp_parsed_free(p);
    wp_shutdown();
    printf("c_pipeline ok: %d batches of %dx%d\n", NB, frame.width, frame.height);
    return 0;
}
