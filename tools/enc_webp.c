/* Build-container-only helper: encodes a raw RGB / RGBA file with libwebp's advanced API so that
 * bench_data/ can cover encoder settings PIL does not expose (simple loop filter, sharpness,
 * segments, partitions).  Not part of the product; the GPU box never runs it.
 *   enc_webp in.raw W H channels out.webp quality method filter_type filter_strength sharpness lossless alpha_q partitions segments */
#include <stdio.h>
#include <stdlib.h>
#include <webp/encode.h>
int main(int argc, char** argv) {
    if (argc < 15) { fprintf(stderr, "usage\n"); return 2; }
    int W = atoi(argv[2]), H = atoi(argv[3]), ch = atoi(argv[4]);
    size_t n = (size_t)W * H * ch;
    unsigned char* buf = malloc(n);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(buf, 1, n, f) != n) { fprintf(stderr, "read\n"); return 1; }
    fclose(f);
    WebPConfig cfg;
    if (!WebPConfigInit(&cfg)) return 1;
    cfg.quality = (float)atof(argv[6]); cfg.method = atoi(argv[7]);
    cfg.filter_type = atoi(argv[8]); cfg.filter_strength = atoi(argv[9]); cfg.filter_sharpness = atoi(argv[10]);
    cfg.lossless = atoi(argv[11]); cfg.alpha_quality = atoi(argv[12]); cfg.partitions = atoi(argv[13]); cfg.segments = atoi(argv[14]);
    if (cfg.filter_strength < 0) { cfg.autofilter = 1; cfg.filter_strength = 60; }
    if (cfg.lossless) cfg.exact = 1;
    if (!WebPValidateConfig(&cfg)) { fprintf(stderr, "config\n"); return 1; }
    WebPPicture pic;
    if (!WebPPictureInit(&pic)) return 1;
    pic.width = W; pic.height = H; pic.use_argb = cfg.lossless;
    int ok = ch == 4 ? WebPPictureImportRGBA(&pic, buf, W * 4) : WebPPictureImportRGB(&pic, buf, W * 3);
    if (!ok) return 1;
    WebPMemoryWriter wr; WebPMemoryWriterInit(&wr);
    pic.writer = WebPMemoryWrite; pic.custom_ptr = &wr;
    if (!WebPEncode(&cfg, &pic)) { fprintf(stderr, "encode err %d\n", pic.error_code); return 1; }
    f = fopen(argv[5], "wb"); fwrite(wr.mem, 1, wr.size, f); fclose(f);
    printf("%s %zu bytes\n", argv[5], wr.size);
    return 0;
}
