/* A plain C99 program that drives the library through include/webp_amd.h alone, the way a foreign-function binding
 * would: parse one lossy .webp with the library's host side, lay the frame out as a one-item batch, decode it with the
 * host-pointer entry point and write Y, U, V and A,R,G,B to a file for the test to compare with the oracle.
 *   c_caller <in.webp> <out.bin>       exit status 0 = ok; on failure prints wp_last_error. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "webp_amd.h"

static int die(const char* what) {
    char msg[512];
    wp_last_error(msg, (int)sizeof msg);
    fprintf(stderr, "%s: %s\n", what, msg);
    return 1;
}

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: c_caller in.webp out.bin\n"); return 2; }
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
    if (wp_parsed_kind(p) != 0 || wp_parsed_nframes(p) != 1) { fprintf(stderr, "not a lossy still\n"); return 2; }

    const wp_parsed* files[1];
    wp_vp8_frame frame;
    wp_vp8_batch_sizes sz;
    files[0] = p;
    if (wp_vp8_batch_layout(1, files, NULL, NULL, &frame, &sz) != WP_OK) return die("wp_vp8_batch_layout");
    uint32_t* mb = (uint32_t*)malloc((size_t)sz.n_mb * 16);
    uint32_t* filt = (uint32_t*)malloc((size_t)sz.n_mb * 4);
    size_t n_blocks = (size_t)sz.n_coeff_blocks ? (size_t)sz.n_coeff_blocks : 1;
    int16_t* coeff = (int16_t*)calloc(n_blocks, 32);
    if (wp_vp8_batch_gather(1, files, NULL, &frame, mb, filt, (size_t)sz.n_mb, coeff, n_blocks, 1) != WP_OK) return die("wp_vp8_batch_gather");

    uint8_t* y = (uint8_t*)malloc((size_t)sz.y_bytes);
    uint8_t* u = (uint8_t*)malloc((size_t)sz.u_bytes);
    uint8_t* v = (uint8_t*)malloc((size_t)sz.v_bytes);
    uint8_t* argb = (uint8_t*)malloc((size_t)sz.argb_bytes);
    if (wp_init(0) != WP_OK) return die("wp_init");
    if (wp_vp8_decode_to_argb(1, &frame, mb, (size_t)sz.n_mb, filt, coeff, n_blocks, y, u, v, NULL, argb) != WP_OK) return die("wp_vp8_decode_to_argb");

    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 2; }
    int32_t hdr[4];
    hdr[0] = frame.width; hdr[1] = frame.height; hdr[2] = frame.mbw; hdr[3] = frame.mbh;
    fwrite(hdr, sizeof hdr, 1, o);
    fwrite(y, 1, (size_t)256 * frame.mbw * frame.mbh, o);
    fwrite(u, 1, (size_t)64 * frame.mbw * frame.mbh, o);
    fwrite(v, 1, (size_t)64 * frame.mbw * frame.mbh, o);
    fwrite(argb, 1, (size_t)4 * frame.width * frame.height, o);
    fclose(o);
    wp_parsed_free(p);
    wp_shutdown();
    printf("c_caller ok: %dx%d\n", frame.width, frame.height);
    return 0;
}
