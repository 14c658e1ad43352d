/*
 * webp_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded CPU restatement of the fourst4r/webp (Haxe) decoder,
 * written to follow the Haxe text literally, including its deviations from
 * libwebp / the VP8 spec.  It exists so that
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg have something
 * to check the HIP path against.  Nothing under webp_amd/ may include, link or
 * call this file.
 *
 * Parity pinning: validated against the reference's own golden vectors
 * (the PNGs under test/expected, 130 files, via tests/golden/) -- see tests/test_oracle_golden.py.
 *
 * Each function cites the reference file:line it restates (paths relative to
 * /root/reference/src/webp/).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "vp8_tables.h"

#define WO_OK 0
#define WO_ERR_FORMAT (-1)   /* a Haxe `throw "<string>"` */
#define WO_ERR_EOF (-2)      /* haxe.io.Eof */
#define WO_ERR_NULL (-3)     /* decodeFrame returned null (Q11) */
#define WO_ERR_ALLOC (-4)

static const char* g_err = "";
static int fail(int code, const char* msg) { g_err = msg; return code; }
const char* wo_last_error(void) { return g_err; }

/* ===================================================================== */
/* vp8/Partition.hx : boolean decoder                                     */
/* ===================================================================== */
typedef struct {
    const uint8_t* buf; int len;
    int r; int rangeM1; int32_t bits; int nBits; int unexpectedEOF;
} Partition;

/* Partition.hx:45-52 */
static void part_init(Partition* p, const uint8_t* buf, int len) {
    p->buf = buf; p->len = len; p->r = 0; p->rangeM1 = 254; p->bits = 0; p->nBits = 0; p->unexpectedEOF = 0;
}
/* Partition.hx:55-85 */
static int readBit(Partition* p, int prob) {
    if (p->nBits < 8) {
        if (p->r >= p->len) { p->unexpectedEOF = 1; return 0; }
        int x = p->buf[p->r];
        p->bits |= x << (8 - p->nBits);
        p->r++;
        p->nBits += 8;
    }
    int split = ((p->rangeM1 * prob) >> 8) + 1;
    int bit = p->bits >= (split << 8);
    if (bit) { p->rangeM1 -= split; p->bits -= split << 8; }
    else { p->rangeM1 = split - 1; }
    if (p->rangeM1 < 127) {
        int shift = wo_BoolShift[p->rangeM1];
        p->rangeM1 = wo_BoolRange[p->rangeM1];
        p->bits = (int32_t)((uint32_t)p->bits << shift);
        p->nBits -= shift;
    }
    return bit;
}
/* Partition.hx:88-96 */
static int readUint(Partition* p, int prob, int n) {
    int u = 0;
    for (int i = 0; i < n; i++) if (readBit(p, prob)) u |= 1 << (n - i - 1);
    return u;
}
/* Partition.hx:99-104 */
static int readInt(Partition* p, int prob, int n) {
    int u = readUint(p, prob, n);
    int b = readBit(p, prob);
    return b ? -u : u;
}
/* Partition.hx:107-112 */
static int readOptionalInt(Partition* p, int prob, int n) {
    if (!readBit(p, prob)) return 0;
    return readInt(p, prob, n);
}
#define uniformProb 128

/* ===================================================================== */
/* vp8/Vp8Decoder.hx                                                      */
/* ===================================================================== */
enum { predDC = 0, predTM, predVE, predHE, predRD, predVR, predLD, predVL, predHD, predHU,
       predDCTop, predDCLeft, predDCTopLeft };
enum { planeY1WithY2 = 0, planeY2 = 1, planeUV = 2, planeY1SansY2 = 3 };
#define bCoeffBase 256
#define rCoeffBase 320
#define whtCoeffBase 384
#define ybrYX 8
#define ybrYY 1
#define ybrBX 8
#define ybrBY 18
#define ybrRX 24
#define ybrRY 18

static const uint8_t kBands[17] = {0, 1, 2, 3, 6, 4, 5, 6, 6, 6, 6, 6, 6, 6, 6, 7, 0};      /* Reconstruct.hx:29-31 */
static const uint8_t kZigzag[16] = {0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15};  /* Reconstruct.hx:40-42 */
static const uint8_t kCat3456[4][12] = {                                                      /* Reconstruct.hx:33-38 */
    {173, 148, 140, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {176, 155, 140, 135, 0, 0, 0, 0, 0, 0, 0, 0},
    {180, 157, 141, 134, 130, 0, 0, 0, 0, 0, 0, 0},
    {254, 254, 243, 230, 196, 177, 153, 140, 133, 130, 129, 0}};

typedef struct { int level, ilevel, hlevel, inner; } FilterParam;     /* Filter.hx:5-10 */
typedef struct { uint8_t pred[4]; int nzMask; int nzY16; } MB;          /* Vp8Decoder.hx:42-51 */
typedef struct { int y1[2], y2[2], uv[2]; } Quant;                      /* Quant.hx:3-8 */

typedef struct {
    const uint8_t* data; int len; int pos;          /* BytesInput r */
    int mbw, mbh;
    /* frameHeader */
    int keyFrame, versionNumber, showFrame, firstPartitionLen, width, height, xScale, yScale;
    /* segmentHeader: Int8Vector fields are *unsigned* bytes on read-back (Q2) */
    int useSegment, updateMap, relativeDelta;
    uint8_t seg_quantizer[4], seg_filterStrength[4], seg_prob[3];
    /* filterHeader */
    int f_simple, f_level, f_sharpness, f_useLFDelta;
    uint8_t f_refLFDelta[4], f_modeLFDelta[4], f_perSegmentLevel[4];
    Partition fp; Partition op[8]; int nOP;
    Quant quant[4];
    uint8_t tokenProb[4][8][3][11];
    int useSkipProb, skipProb;
    FilterParam filterParams[4][2];
    FilterParam** perMB;            /* pointers: FilterParam is a reference type in Haxe (Q1) */
    int segment;
    MB leftMB; MB* upMB;
    int nzDCMask, nzACMask;
    int usePredY16, predY16, predC8;
    int predY4[4][4];
    int32_t coeff[400];
    int32_t ybr[26][32];
    uint8_t *Y, *Cb, *Cr; int YStride, CStride;
    uint8_t *Y0, *Cb0, *Cr0;        /* optional: copy of planes before the loop filter (test hook) */
} Vp8;

/* Vp8Decoder.hx:126-184 */
static int vp8_decodeFrameHeader(Vp8* d) {
    if (d->len - d->pos < 3) return fail(WO_ERR_EOF, "Eof");
    const uint8_t* b = d->data + d->pos; d->pos += 3;
    d->keyFrame = (b[0] & 1) == 0;
    d->versionNumber = (b[0] >> 1) & 7;
    d->showFrame = ((b[0] >> 4) & 1) == 1;
    d->firstPartitionLen = (b[0] >> 5) | (b[1] << 3) | (b[2] << 11);
    d->width = d->height = d->xScale = d->yScale = 0;
    if (!d->keyFrame) return WO_OK;
    if (d->len - d->pos < 7) return fail(WO_ERR_EOF, "Eof");
    b = d->data + d->pos; d->pos += 7;
    if (b[0] != 0x9d || b[1] != 0x01 || b[2] != 0x2a) return fail(WO_ERR_FORMAT, "vp8: invalid format");
    d->width = ((b[4] & 0x3f) << 8) | b[3];
    d->height = ((b[6] & 0x3f) << 8) | b[5];
    d->xScale = b[4] >> 6;
    d->yScale = b[6] >> 6;
    d->mbw = (d->width + 0x0f) >> 4;
    d->mbh = (d->height + 0x0f) >> 4;
    d->seg_prob[0] = d->seg_prob[1] = d->seg_prob[2] = 0xff;
    memset(d->seg_filterStrength, 0, 4); memset(d->seg_quantizer, 0, 4);
    d->useSegment = d->updateMap = d->relativeDelta = 0;
    memcpy(d->tokenProb, wo_TokenProbDefault, sizeof d->tokenProb);
    d->segment = 0;
    return WO_OK;
}

/* Vp8Decoder.hx:203-227 */
static void vp8_parseSegmentHeader(Vp8* d) {
    Partition* fp = &d->fp;
    d->useSegment = readBit(fp, uniformProb);
    if (!d->useSegment) { d->updateMap = 0; return; }
    d->updateMap = readBit(fp, uniformProb);
    if (readBit(fp, uniformProb)) {
        d->relativeDelta = !readBit(fp, uniformProb);
        for (int i = 0; i < 4; i++) d->seg_quantizer[i] = (uint8_t)readOptionalInt(fp, uniformProb, 7);
        for (int i = 0; i < 4; i++) d->seg_filterStrength[i] = (uint8_t)readOptionalInt(fp, uniformProb, 6);
    }
    if (!d->updateMap) return;
    for (int i = 0; i < 3; i++)
        d->seg_prob[i] = readBit(fp, uniformProb) ? (uint8_t)readUint(fp, uniformProb, 8) : 0xff;
}

/* Vp8Decoder.hx:488-539 */
static void vp8_computeFilterParams(Vp8* d) {
    for (int i = 0; i < 4; i++) {
        int baseLevel = d->f_level;
        if (d->useSegment) {
            baseLevel = d->seg_filterStrength[i];           /* unsigned read-back (Q2) */
            if (d->relativeDelta) baseLevel += d->f_level;
        }
        for (int j = 0; j < 2; j++) {
            FilterParam* p = &d->filterParams[i][j];
            p->inner = j != 0;
            int level = baseLevel;
            if (d->f_useLFDelta) {
                level += d->f_refLFDelta[0];
                if (j != 0) level += d->f_modeLFDelta[0];
            }
            if (level <= 0) { p->level = 0; continue; }
            if (level > 63) level = 63;
            int ilevel = level;
            if (d->f_sharpness > 0) {
                ilevel >>= (d->f_sharpness > 4 ? 2 : 1);
                if (ilevel > 9 - d->f_sharpness) ilevel = 9 - d->f_sharpness;
            }
            if (ilevel < 1) ilevel = 1;
            p->ilevel = ilevel;
            p->level = 2 * level + ilevel;
            if (d->keyFrame) p->hlevel = level < 15 ? 0 : (level < 40 ? 1 : 2);
            else p->hlevel = level < 15 ? 0 : (level < 20 ? 1 : (level < 40 ? 2 : 3));
        }
    }
}

/* Vp8Decoder.hx:229-258 */
static void vp8_parseFilterHeader(Vp8* d) {
    Partition* fp = &d->fp;
    d->f_simple = readBit(fp, uniformProb);
    d->f_level = readUint(fp, uniformProb, 6);
    d->f_sharpness = readUint(fp, uniformProb, 3);
    d->f_useLFDelta = readBit(fp, uniformProb);
    if (d->f_useLFDelta && readBit(fp, uniformProb)) {
        for (int i = 0; i < 4; i++) d->f_refLFDelta[i] = (uint8_t)readOptionalInt(fp, uniformProb, 6);
        for (int i = 0; i < 4; i++) d->f_modeLFDelta[i] = (uint8_t)readOptionalInt(fp, uniformProb, 6);
    }
    if (d->f_level == 0) return;
    if (d->useSegment) {
        for (int i = 0; i < 4; i++) {
            int strength = d->seg_filterStrength[i];
            if (d->relativeDelta) strength += d->f_level;
            d->f_perSegmentLevel[i] = (uint8_t)strength;
        }
    } else {
        d->f_perSegmentLevel[0] = (uint8_t)d->f_level;
    }
    vp8_computeFilterParams(d);
}

/* Vp8Decoder.hx:260-287 (Q3: every partition is sliced from offset 0) */
static int vp8_parseOtherPartitions(Vp8* d) {
    int partLens[8] = {0};
    d->nOP = 1 << readUint(&d->fp, uniformProb, 2);
    int n = 3 * (d->nOP - 1);
    partLens[d->nOP - 1] = (d->len - d->pos) - n;
    if (partLens[d->nOP - 1] < 0) return 0;
    if (n > 0) {
        const uint8_t* buf = d->data + d->pos; d->pos += n;
        for (int i = 0; i < d->nOP - 1; i++) {
            int pl = buf[3 * i] | (buf[3 * i + 1] << 8) | (buf[3 * i + 2] << 16);
            if (pl > partLens[d->nOP - 1]) return 0;
            partLens[i] = pl;
            partLens[d->nOP - 1] -= pl;
        }
    }
    if ((1 << 24) <= partLens[d->nOP - 1]) return 0;
    const uint8_t* buf = d->data + d->pos; d->pos = d->len;
    for (int i = 0; i < d->nOP; i++) part_init(&d->op[i], buf, partLens[i]);
    return 1;
}

static int clipi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* Vp8Decoder.hx:382-415 */
static void vp8_parseQuant(Vp8* d) {
    Partition* fp = &d->fp;
    int baseQ0 = readUint(fp, uniformProb, 7);
    int dqy1DC = readOptionalInt(fp, uniformProb, 4);
    int dqy2DC = readOptionalInt(fp, uniformProb, 4);
    int dqy2AC = readOptionalInt(fp, uniformProb, 4);
    int dquvDC = readOptionalInt(fp, uniformProb, 4);
    int dquvAC = readOptionalInt(fp, uniformProb, 4);
    for (int i = 0; i < 4; i++) {
        int q = baseQ0;
        if (d->useSegment) {
            if (d->relativeDelta) q += d->seg_quantizer[i];     /* unsigned (Q2) */
            else q = d->seg_quantizer[i];
        }
        Quant* Q = &d->quant[i];
        Q->y1[0] = wo_DequantDC[clipi(q + dqy1DC, 0, 127)];
        Q->y1[1] = wo_DequantAC[clipi(q, 0, 127)];
        Q->y2[0] = wo_DequantDC[clipi(q + dqy2DC, 0, 127)] * 2;
        Q->y2[1] = (int)(wo_DequantAC[clipi(q + dqy2AC, 0, 127)] * 155 / 100);
        if (Q->y2[1] < 8) Q->y2[1] = 8;
        Q->uv[0] = wo_DequantDC[clipi(q + dquvDC, 0, 117)];
        Q->uv[1] = wo_DequantAC[clipi(q + dquvAC, 0, 127)];
    }
}

/* Vp8Decoder.hx:326-333 */
static void vp8_parseTokenProb(Vp8* d) {
    const uint8_t* upd = wo_TokenProbUpdate;
    uint8_t* tp = &d->tokenProb[0][0][0][0];
    for (int i = 0; i < 4 * 8 * 3 * 11; i++)
        if (readBit(&d->fp, upd[i])) tp[i] = (uint8_t)readUint(&d->fp, uniformProb, 8);
}

/* Vp8Decoder.hx:289-324 */
static int vp8_parseOtherHeaders(Vp8* d) {
    if (d->len - d->pos < d->firstPartitionLen) return fail(WO_ERR_EOF, "Eof");
    part_init(&d->fp, d->data + d->pos, d->firstPartitionLen); d->pos += d->firstPartitionLen;
    if (d->keyFrame) { readBit(&d->fp, uniformProb); readBit(&d->fp, uniformProb); }
    vp8_parseSegmentHeader(d);
    vp8_parseFilterHeader(d);
    if (!vp8_parseOtherPartitions(d)) return 0;
    vp8_parseQuant(d);
    if (!d->keyFrame) return 0;
    readBit(&d->fp, uniformProb);
    vp8_parseTokenProb(d);
    d->useSkipProb = readBit(&d->fp, uniformProb);
    if (d->useSkipProb) d->skipProb = readUint(&d->fp, uniformProb, 8);
    if (d->fp.unexpectedEOF) return 0;
    return 1;
}

/* Vp8Decoder.hx:541-577 */
static void vp8_prepareYBR(Vp8* d, int mbx, int mby) {
    int32_t (*ybr)[32] = d->ybr;
    if (mbx == 0) {
        for (int y = 0; y < 17; y++) ybr[y][7] = 0x81;
        for (int y = 17; y < 26; y++) { ybr[y][7] = 0x81; ybr[y][23] = 0x81; }
    } else {
        for (int y = 0; y < 17; y++) ybr[y][7] = ybr[y][7 + 16];
        for (int y = 17; y < 26; y++) { ybr[y][7] = ybr[y][15]; ybr[y][23] = ybr[y][31]; }
    }
    if (mby == 0) {
        for (int x = 7; x < 28; x++) ybr[0][x] = 0x7f;
        for (int x = 7; x < 16; x++) ybr[17][x] = 0x7f;
        for (int x = 23; x < 32; x++) ybr[17][x] = 0x7f;
    } else {
        for (int i = 0; i < 16; i++) ybr[0][8 + i] = d->Y[(16 * mby - 1) * d->YStride + 16 * mbx + i];
        for (int i = 0; i < 8; i++) ybr[17][8 + i] = d->Cb[(8 * mby - 1) * d->CStride + 8 * mbx + i];
        for (int i = 0; i < 8; i++) ybr[17][24 + i] = d->Cr[(8 * mby - 1) * d->CStride + 8 * mbx + i];
        if (mbx == d->mbw - 1) {
            for (int i = 16; i < 20; i++) ybr[0][8 + i] = d->Y[(16 * mby - 1) * d->YStride + 16 * mbx + 15];
        } else {
            for (int i = 16; i < 20; i++) ybr[0][8 + i] = d->Y[(16 * mby - 1) * d->YStride + 16 * mbx + i];
        }
    }
    for (int y = 4; y < 16; y += 4)
        for (int x = 24; x < 28; x++) ybr[y][x] = ybr[0][x];
}

/* Vp8Decoder.hx:579-637 (the dead pstop/eq computation at :584-590 is omitted: no side effects) */
static int vp8_parseResiduals4(Vp8* d, Partition* r, int plane, int context, const int* quant, int skipFirstCoeff, int coeffBase) {
    uint8_t (*prob)[3][11] = d->tokenProb[plane];
    int n = skipFirstCoeff ? 1 : 0;
    const uint8_t* p = prob[kBands[n]][context];
    if (!readBit(r, p[0])) return 0;
    while (n != 16) {
        n++;
        if (!readBit(r, p[1])) { p = prob[kBands[n]][0]; continue; }
        int v = 0;
        if (!readBit(r, p[2])) {
            v = 1;
            p = prob[kBands[n]][1];
        } else {
            if (!readBit(r, p[3])) {
                v = !readBit(r, p[4]) ? 2 : 3 + readUint(r, p[5], 1);
            } else if (!readBit(r, p[6])) {
                if (!readBit(r, p[7])) v = 5 + readUint(r, 159, 1);
                else { v = 7 + 2 * readUint(r, 165, 1); v += readUint(r, 145, 1); }
            } else {
                int b1 = readUint(r, p[8], 1);
                int b0 = readUint(r, p[9 + b1], 1);
                int cat = 2 * b1 + b0;
                const uint8_t* tab = kCat3456[cat];
                for (int i = 0; i < 12; i++) {
                    if (tab[i] == 0) break;
                    v *= 2;
                    v += readUint(r, tab[i], 1);
                }
                v += 3 + (8 << cat);
            }
            p = prob[kBands[n]][2];
        }
        int z = kZigzag[n - 1];
        int32_t c = v * quant[z > 0 ? 1 : 0];
        c = readBit(r, uniformProb) ? -c : c;
        d->coeff[coeffBase + z] = (int32_t)((uint32_t)c << 16) >> 16;
        if (n == 16 || !readBit(r, p[0])) return 1;
    }
    return 1;
}

/* Vp8Decoder.hx:960-989 (Q5: outputs are not truncated to int16) */
static void vp8_inverseWHT16(Vp8* d) {
    int32_t m[16];
    int32_t* c = d->coeff;
    for (int i = 0; i < 4; i++) {
        int32_t a0 = c[384 + 0 + i] + c[384 + 12 + i];
        int32_t a1 = c[384 + 4 + i] + c[384 + 8 + i];
        int32_t a2 = c[384 + 4 + i] - c[384 + 8 + i];
        int32_t a3 = c[384 + 0 + i] - c[384 + 12 + i];
        m[0 + i] = a0 + a1; m[8 + i] = a0 - a1; m[4 + i] = a3 + a2; m[12 + i] = a3 - a2;
    }
    int out = 0;
    for (int i = 0; i < 4; i++) {
        int32_t dc = m[0 + i * 4] + 3;
        int32_t a0 = dc + m[3 + i * 4];
        int32_t a1 = m[1 + i * 4] + m[2 + i * 4];
        int32_t a2 = m[1 + i * 4] - m[2 + i * 4];
        int32_t a3 = dc - m[3 + i * 4];
        c[out + 0] = (a0 + a1) >> 3;
        c[out + 16] = (a3 + a2) >> 3;
        c[out + 32] = (a0 - a1) >> 3;
        c[out + 48] = (a3 - a2) >> 3;
        out += 64;
    }
}

static const uint8_t kUnpack[16][4] = {   /* Reconstruct.hx:22-27 */
    {0,0,0,0},{1,0,0,0},{0,1,0,0},{1,1,0,0},{0,0,1,0},{1,0,1,0},{0,1,1,0},{1,1,1,0},
    {0,0,0,1},{1,0,0,1},{0,1,0,1},{1,1,0,1},{0,0,1,1},{1,0,1,1},{0,1,1,1},{1,1,1,1}};
static int pack4(const int* x, int shift) { return (x[0] << 0 | x[1] << 1 | x[2] << 2 | x[3] << 3) << shift; }

/* Vp8Decoder.hx:639-714 */
static int vp8_parseResiduals(Vp8* d, int mbx, int mby) {
    Partition* partition = &d->op[mby & (d->nOP - 1)];
    int plane = planeY1SansY2;
    Quant* quant = &d->quant[d->segment];
    MB* up = &d->upMB[mbx];
    if (d->usePredY16) {
        int nz = vp8_parseResiduals4(d, partition, planeY2, d->leftMB.nzY16 + up->nzY16, quant->y2, 0, whtCoeffBase);
        d->leftMB.nzY16 = nz;
        up->nzY16 = nz;
        vp8_inverseWHT16(d);
        plane = planeY1WithY2;
    }
    int nzDC[4] = {0,0,0,0}, nzAC[4] = {0,0,0,0};
    int nzDCMask = 0, nzACMask = 0, coeffBase = 0;
    int lnz[4], unz[4];
    for (int i = 0; i < 4; i++) { lnz[i] = kUnpack[d->leftMB.nzMask & 0x0f][i]; unz[i] = kUnpack[up->nzMask & 0x0f][i]; }
    for (int y = 0; y < 4; y++) {
        int nz = lnz[y];
        for (int x = 0; x < 4; x++) {
            nz = vp8_parseResiduals4(d, partition, plane, nz + unz[x], quant->y1, d->usePredY16, coeffBase);
            unz[x] = nz;
            nzAC[x] = nz;
            nzDC[x] = d->coeff[coeffBase] != 0 ? 1 : 0;
            coeffBase += 16;
        }
        lnz[y] = nz;
        nzDCMask |= pack4(nzDC, y * 4);
        nzACMask |= pack4(nzAC, y * 4);
    }
    int lnzMask = pack4(lnz, 0);
    int unzMask = pack4(unz, 0);
    for (int i = 0; i < 4; i++) { lnz[i] = kUnpack[d->leftMB.nzMask >> 4][i]; unz[i] = kUnpack[up->nzMask >> 4][i]; }
    for (int c = 0; c < 4; c += 2) {
        for (int y = 0; y < 2; y++) {
            int nz = lnz[y + c];
            for (int x = 0; x < 2; x++) {
                nz = vp8_parseResiduals4(d, partition, planeUV, nz + unz[x + c], quant->uv, 0, coeffBase);
                unz[x + c] = nz;
                nzAC[y * 2 + x] = nz;
                nzDC[y * 2 + x] = d->coeff[coeffBase] != 0 ? 1 : 0;
                coeffBase += 16;
            }
            lnz[y + c] = nz;
        }
        nzDCMask |= pack4(nzDC, 16 + c * 2);
        nzACMask |= pack4(nzAC, 16 + c * 2);
    }
    lnzMask |= pack4(lnz, 4);
    unzMask |= pack4(unz, 4);
    d->leftMB.nzMask = lnzMask & 0xFF;
    up->nzMask = unzMask & 0xFF;
    d->nzDCMask = nzDCMask;
    d->nzACMask = nzACMask;
    return nzDCMask == 0 && nzACMask == 0;
}

/* ---------------- vp8/PredFunc.hx ---------------- */
/* PredFunc.hx:57-68 */
static int checkTopLeftPred(int mbx, int mby, int p) {
    if (p != predDC) return p;
    if (mbx == 0) { if (mby == 0) return predDCTopLeft; return predDCLeft; }
    if (mby == 0) return predDCTop;
    return predDC;
}
#define YB(yy, xx) (z->ybr[(yy)][(xx)])
static void fill(Vp8* z, int y, int x, int n, int v) { for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) YB(y + j, x + i) = v; }
/* PredFunc.hx:118-133 */
static void predFunc4DC(Vp8* z, int y, int x) {
    int sum = 4;
    for (int i = 0; i < 4; i++) sum += YB(y - 1, x + i);
    for (int j = 0; j < 4; j++) sum += YB(y + j, x - 1);
    fill(z, y, x, 4, sum / 8);
}
/* PredFunc.hx:135-144, 436-445, 513-522 */
static void predFuncTM(Vp8* z, int y, int x, int n) {
    int delta0 = -YB(y - 1, x - 1);
    for (int j = 0; j < n; j++) {
        int delta1 = delta0 + YB(y + j, x - 1);
        for (int i = 0; i < n; i++) YB(y + j, x + i) = clipi(delta1 + YB(y - 1, x + i), 0, 255);
    }
}
/* PredFunc.hx:146-165 */
static void predFunc4VE(Vp8* z, int y, int x) {
    int a = YB(y-1,x-1), b = YB(y-1,x), c = YB(y-1,x+1), d = YB(y-1,x+2), e = YB(y-1,x+3), f = YB(y-1,x+4);
    int abc = (a + 2*b + c + 2) / 4, bcd = (b + 2*c + d + 2) / 4, cde = (c + 2*d + e + 2) / 4, def = (d + 2*e + f + 2) / 4;
    for (int j = 0; j < 4; j++) { YB(y+j,x) = abc; YB(y+j,x+1) = bcd; YB(y+j,x+2) = cde; YB(y+j,x+3) = def; }
}
/* PredFunc.hx:167-185 */
static void predFunc4HE(Vp8* z, int y, int x) {
    int s = YB(y+3,x-1), r = YB(y+2,x-1), q = YB(y+1,x-1), p = YB(y,x-1), a = YB(y-1,x-1);
    int ssr = (s + 2*s + r + 2) / 4, srq = (s + 2*r + q + 2) / 4, rqp = (r + 2*q + p + 2) / 4, apq = (a + 2*p + q + 2) / 4;
    for (int i = 0; i < 4; i++) { YB(y,x+i) = apq; YB(y+1,x+i) = rqp; YB(y+2,x+i) = srq; YB(y+3,x+i) = ssr; }
}
/* PredFunc.hx:187-225 */
static void predFunc4RD(Vp8* z, int y, int x) {
    int s = YB(y+3,x-1), r = YB(y+2,x-1), q = YB(y+1,x-1), p = YB(y,x-1), a = YB(y-1,x-1);
    int b = YB(y-1,x), c = YB(y-1,x+1), d = YB(y-1,x+2), e = YB(y-1,x+3);
    int srq = (s + 2*r + q + 2) / 4, rqp = (r + 2*q + p + 2) / 4, qpa = (q + 2*p + a + 2) / 4, pab = (p + 2*a + b + 2) / 4;
    int abc = (a + 2*b + c + 2) / 4, bcd = (b + 2*c + d + 2) / 4, cde = (c + 2*d + e + 2) / 4;
    YB(y,x) = pab; YB(y,x+1) = abc; YB(y,x+2) = bcd; YB(y,x+3) = cde;
    YB(y+1,x) = qpa; YB(y+1,x+1) = pab; YB(y+1,x+2) = abc; YB(y+1,x+3) = bcd;
    YB(y+2,x) = rqp; YB(y+2,x+1) = qpa; YB(y+2,x+2) = pab; YB(y+2,x+3) = abc;
    YB(y+3,x) = srq; YB(y+3,x+1) = rqp; YB(y+3,x+2) = qpa; YB(y+3,x+3) = pab;
}
/* PredFunc.hx:227-268 */
static void predFunc4VR(Vp8* z, int y, int x) {
    int r = YB(y+2,x-1), q = YB(y+1,x-1), p = YB(y,x-1), a = YB(y-1,x-1);
    int b = YB(y-1,x), c = YB(y-1,x+1), d = YB(y-1,x+2), e = YB(y-1,x+3);
    int ab = (a + b + 1) / 2, bc = (b + c + 1) / 2, cd = (c + d + 1) / 2, de = (d + e + 1) / 2;
    int rqp = (r + 2*q + p + 2) / 4, qpa = (q + 2*p + a + 2) / 4, pab = (p + 2*a + b + 2) / 4;
    int abc = (a + 2*b + c + 2) / 4, bcd = (b + 2*c + d + 2) / 4, cde = (c + 2*d + e + 2) / 4;
    YB(y,x) = ab; YB(y,x+1) = bc; YB(y,x+2) = cd; YB(y,x+3) = de;
    YB(y+1,x) = pab; YB(y+1,x+1) = abc; YB(y+1,x+2) = bcd; YB(y+1,x+3) = cde;
    YB(y+2,x) = qpa; YB(y+2,x+1) = ab; YB(y+2,x+2) = bc; YB(y+2,x+3) = cd;
    YB(y+3,x) = rqp; YB(y+3,x+1) = pab; YB(y+3,x+2) = abc; YB(y+3,x+3) = bcd;
}
/* PredFunc.hx:270-307 */
static void predFunc4LD(Vp8* z, int y, int x) {
    int a = YB(y-1,x), b = YB(y-1,x+1), c = YB(y-1,x+2), d = YB(y-1,x+3), e = YB(y-1,x+4), f = YB(y-1,x+5), g = YB(y-1,x+6), h = YB(y-1,x+7);
    int abc = (a + 2*b + c + 2) / 4, bcd = (b + 2*c + d + 2) / 4, cde = (c + 2*d + e + 2) / 4, def = (d + 2*e + f + 2) / 4;
    int efg = (e + 2*f + g + 2) / 4, fgh = (f + 2*g + h + 2) / 4, ghh = (g + 2*h + h + 2) / 4;
    YB(y,x) = abc; YB(y,x+1) = bcd; YB(y,x+2) = cde; YB(y,x+3) = def;
    YB(y+1,x) = bcd; YB(y+1,x+1) = cde; YB(y+1,x+2) = def; YB(y+1,x+3) = efg;
    YB(y+2,x) = cde; YB(y+2,x+1) = def; YB(y+2,x+2) = efg; YB(y+2,x+3) = fgh;
    YB(y+3,x) = def; YB(y+3,x+1) = efg; YB(y+3,x+2) = fgh; YB(y+3,x+3) = ghh;
}
/* PredFunc.hx:309-347 */
static void predFunc4VL(Vp8* z, int y, int x) {
    int a = YB(y-1,x), b = YB(y-1,x+1), c = YB(y-1,x+2), d = YB(y-1,x+3), e = YB(y-1,x+4), f = YB(y-1,x+5), g = YB(y-1,x+6), h = YB(y-1,x+7);
    int ab = (a + b + 1) / 2, bc = (b + c + 1) / 2, cd = (c + d + 1) / 2, de = (d + e + 1) / 2;
    int abc = (a + 2*b + c + 2) / 4, bcd = (b + 2*c + d + 2) / 4, cde = (c + 2*d + e + 2) / 4, def = (d + 2*e + f + 2) / 4;
    int efg = (e + 2*f + g + 2) / 4, fgh = (f + 2*g + h + 2) / 4;
    YB(y,x) = ab; YB(y,x+1) = bc; YB(y,x+2) = cd; YB(y,x+3) = de;
    YB(y+1,x) = abc; YB(y+1,x+1) = bcd; YB(y+1,x+2) = cde; YB(y+1,x+3) = def;
    YB(y+2,x) = bc; YB(y+2,x+1) = cd; YB(y+2,x+2) = de; YB(y+2,x+3) = efg;
    YB(y+3,x) = bcd; YB(y+3,x+1) = cde; YB(y+3,x+2) = def; YB(y+3,x+3) = fgh;
}
/* PredFunc.hx:349-387 */
static void predFunc4HD(Vp8* z, int y, int x) {
    int s = YB(y+3,x-1), r = YB(y+2,x-1), q = YB(y+1,x-1), p = YB(y,x-1), a = YB(y-1,x-1);
    int b = YB(y-1,x), c = YB(y-1,x+1), d = YB(y-1,x+2);
    int sr = (s + r + 1) / 2, rq = (r + q + 1) / 2, qp = (q + p + 1) / 2, pa = (p + a + 1) / 2;
    int srq = (s + 2*r + q + 2) / 4, rqp = (r + 2*q + p + 2) / 4, qpa = (q + 2*p + a + 2) / 4, pab = (p + 2*a + b + 2) / 4;
    int abc = (a + 2*b + c + 2) / 4, bcd = (b + 2*c + d + 2) / 4;
    YB(y,x) = pa; YB(y,x+1) = pab; YB(y,x+2) = abc; YB(y,x+3) = bcd;
    YB(y+1,x) = qp; YB(y+1,x+1) = qpa; YB(y+1,x+2) = pa; YB(y+1,x+3) = pab;
    YB(y+2,x) = rq; YB(y+2,x+1) = rqp; YB(y+2,x+2) = qp; YB(y+2,x+3) = qpa;
    YB(y+3,x) = sr; YB(y+3,x+1) = srq; YB(y+3,x+2) = rq; YB(y+3,x+3) = rqp;
}
/* PredFunc.hx:389-418 */
static void predFunc4HU(Vp8* z, int y, int x) {
    int s = YB(y+3,x-1), r = YB(y+2,x-1), q = YB(y+1,x-1), p = YB(y,x-1);
    int pq = (p + q + 1) / 2, qr = (q + r + 1) / 2, rs = (r + s + 1) / 2;
    int pqr = (p + 2*q + r + 2) / 4, qrs = (q + 2*r + s + 2) / 4, rss = (r + 2*s + s + 2) / 4, sss = s;
    YB(y,x) = pq; YB(y,x+1) = pqr; YB(y,x+2) = qr; YB(y,x+3) = qrs;
    YB(y+1,x) = qr; YB(y+1,x+1) = qrs; YB(y+1,x+2) = rs; YB(y+1,x+3) = rss;
    YB(y+2,x) = rs; YB(y+2,x+1) = rss; YB(y+2,x+2) = sss; YB(y+2,x+3) = sss;
    YB(y+3,x) = sss; YB(y+3,x+1) = sss; YB(y+3,x+2) = sss; YB(y+3,x+3) = sss;
}
static void predFunc4(Vp8* z, int mode, int y, int x) {      /* PredFunc.hx:70-84 */
    switch (mode) {
    case predDC: predFunc4DC(z, y, x); break;
    case predTM: predFuncTM(z, y, x, 4); break;
    case predVE: predFunc4VE(z, y, x); break;
    case predHE: predFunc4HE(z, y, x); break;
    case predRD: predFunc4RD(z, y, x); break;
    case predVR: predFunc4VR(z, y, x); break;
    case predLD: predFunc4LD(z, y, x); break;
    case predVL: predFunc4VL(z, y, x); break;
    case predHD: predFunc4HD(z, y, x); break;
    case predHU: predFunc4HU(z, y, x); break;
    }
}
/* PredFunc.hx:420-572: the 8x8 and 16x16 predictors share their shape; n is 8 or 16.
 * Note the naming quirk kept from the text: "DCTop" (mby==0) sums the LEFT column,
 * "DCLeft" (mbx==0) sums the TOP row. */
static void predFuncN(Vp8* z, int mode, int y, int x, int n) {
    switch (mode) {
    case predDC: {
        int sum = n;
        for (int i = 0; i < n; i++) sum += YB(y - 1, x + i);
        for (int j = 0; j < n; j++) sum += YB(y + j, x - 1);
        fill(z, y, x, n, sum / (2 * n));
        break; }
    case predTM: predFuncTM(z, y, x, n); break;
    case predVE: for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) YB(y + j, x + i) = YB(y - 1, x + i); break;
    case predHE: for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) YB(y + j, x + i) = YB(y + j, x - 1); break;
    case predDCTop: {
        int sum = n / 2;
        for (int j = 0; j < n; j++) sum += YB(y + j, x - 1);
        fill(z, y, x, n, sum / n);
        break; }
    case predDCLeft: {
        int sum = n / 2;
        for (int i = 0; i < n; i++) sum += YB(y - 1, x + i);
        fill(z, y, x, n, sum / n);
        break; }
    case predDCTopLeft: fill(z, y, x, n, 0x80); break;
    }
}

/* Vp8Decoder.hx:991-996 */
static int clip8(int i) { return i < 0 ? 0 : (i > 255 ? 255 : i); }
/* wrapping 32-bit multiply then arithmetic >>16 (Q6) */
static int32_t mul16(int32_t a, int32_t c) { return (int32_t)((uint32_t)a * (uint32_t)c) >> 16; }
/* Vp8Decoder.hx:905-935 */
static void vp8_inverseDCT4(Vp8* z, int y, int x, int coeffBase) {
    const int32_t c1 = 85627, c2 = 35468;
    int32_t m[4][4];
    int32_t* co = z->coeff;
    for (int i = 0; i < 4; i++) {
        int32_t a = co[coeffBase + 0] + co[coeffBase + 8];
        int32_t b = co[coeffBase + 0] - co[coeffBase + 8];
        int32_t c = mul16(co[coeffBase + 4], c2) - mul16(co[coeffBase + 12], c1);
        int32_t dd = mul16(co[coeffBase + 4], c1) + mul16(co[coeffBase + 12], c2);
        m[i][0] = a + dd; m[i][1] = b + c; m[i][2] = b - c; m[i][3] = a - dd;
        coeffBase++;
    }
    for (int j = 0; j < 4; j++) {
        int32_t dc = m[0][j] + 4;
        int32_t a = dc + m[2][j];
        int32_t b = dc - m[2][j];
        int32_t c = mul16(m[1][j], c2) - mul16(m[3][j], c1);
        int32_t dd = mul16(m[1][j], c1) + mul16(m[3][j], c2);
        YB(y + j, x + 0) = clip8(YB(y + j, x + 0) + ((a + dd) >> 3));
        YB(y + j, x + 1) = clip8(YB(y + j, x + 1) + ((b + c) >> 3));
        YB(y + j, x + 2) = clip8(YB(y + j, x + 2) + ((b - c) >> 3));
        YB(y + j, x + 3) = clip8(YB(y + j, x + 3) + ((a - dd) >> 3));
    }
}
/* Vp8Decoder.hx:937-944 */
static void vp8_inverseDCT4DCOnly(Vp8* z, int y, int x, int coeffBase) {
    int32_t dc = (z->coeff[coeffBase + 0] + 4) >> 3;
    for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) YB(y + j, x + i) = clip8(YB(y + j, x + i) + dc);
}
/* Vp8Decoder.hx:946-958 */
static void vp8_inverseDCT8(Vp8* z, int y, int x, int cb) {
    vp8_inverseDCT4(z, y, x, cb); vp8_inverseDCT4(z, y, x + 4, cb + 16);
    vp8_inverseDCT4(z, y + 4, x, cb + 32); vp8_inverseDCT4(z, y + 4, x + 4, cb + 48);
}
static void vp8_inverseDCT8DCOnly(Vp8* z, int y, int x, int cb) {
    vp8_inverseDCT4DCOnly(z, y, x, cb); vp8_inverseDCT4DCOnly(z, y, x + 4, cb + 16);
    vp8_inverseDCT4DCOnly(z, y + 4, x, cb + 32); vp8_inverseDCT4DCOnly(z, y + 4, x + 4, cb + 48);
}

/* Vp8Decoder.hx:716-770 */
static void vp8_reconstructMacroblock(Vp8* d, int mbx, int mby) {
    if (d->usePredY16) {
        int p = checkTopLeftPred(mbx, mby, d->predY16);
        predFuncN(d, p, 1, 8, 16);
        for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) {
            int n = 4 * j + i, y = 4 * j + 1, x = 4 * i + 8, mask = 1 << n;
            if ((d->nzACMask & mask) != 0) vp8_inverseDCT4(d, y, x, 16 * n);
            else if ((d->nzDCMask & mask) != 0) vp8_inverseDCT4DCOnly(d, y, x, 16 * n);
        }
    } else {
        for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) {
            int n = 4 * j + i, y = 4 * j + 1, x = 4 * i + 8, mask = 1 << n;
            predFunc4(d, d->predY4[j][i], y, x);
            if ((d->nzACMask & mask) != 0) vp8_inverseDCT4(d, y, x, 16 * n);
            else if ((d->nzDCMask & mask) != 0) vp8_inverseDCT4DCOnly(d, y, x, 16 * n);
        }
    }
    int p = checkTopLeftPred(mbx, mby, d->predC8);
    predFuncN(d, p, ybrBY, ybrBX, 8);
    if ((d->nzACMask & 0x0f0000) != 0) vp8_inverseDCT8(d, ybrBY, ybrBX, bCoeffBase);
    else if ((d->nzDCMask & 0x0f0000) != 0) vp8_inverseDCT8DCOnly(d, ybrBY, ybrBX, bCoeffBase);
    predFuncN(d, p, ybrRY, ybrRX, 8);
    if ((d->nzACMask & 0xf00000) != 0) vp8_inverseDCT8(d, ybrRY, ybrRX, rCoeffBase);
    else if ((d->nzDCMask & 0xf00000) != 0) vp8_inverseDCT8DCOnly(d, ybrRY, ybrRX, rCoeffBase);
}

/* Vp8Decoder.hx:838-856 */
static void vp8_parsePredModeY16(Vp8* d, int mbx) {
    Partition* fp = &d->fp;
    int p;
    if (!readBit(fp, 156)) p = !readBit(fp, 163) ? predDC : predVE;
    else if (!readBit(fp, 128)) p = predHE;
    else p = predTM;
    for (int i = 0; i < 4; i++) { d->upMB[mbx].pred[i] = (uint8_t)p; d->leftMB.pred[i] = (uint8_t)p; }
    d->predY16 = p;
}
/* Vp8Decoder.hx:858-868 */
static void vp8_parsePredModeC8(Vp8* d) {
    Partition* fp = &d->fp;
    if (!readBit(fp, 142)) d->predC8 = predDC;
    else if (!readBit(fp, 114)) d->predC8 = predVE;
    else if (!readBit(fp, 183)) d->predC8 = predHE;
    else d->predC8 = predTM;
}
/* Vp8Decoder.hx:870-903 */
static void vp8_parsePredModeY4(Vp8* d, int mbx) {
    Partition* fp = &d->fp;
    for (int j = 0; j < 4; j++) {
        int p = d->leftMB.pred[j];
        for (int i = 0; i < 4; i++) {
            const uint8_t* prob = &wo_PredProb[(d->upMB[mbx].pred[i] * 10 + p) * 9];
            if (!readBit(fp, prob[0])) p = predDC;
            else if (!readBit(fp, prob[1])) p = predTM;
            else if (!readBit(fp, prob[2])) p = predVE;
            else if (!readBit(fp, prob[3])) {
                if (!readBit(fp, prob[4])) p = predHE;
                else if (!readBit(fp, prob[5])) p = predRD;
                else p = predVR;
            } else if (!readBit(fp, prob[6])) p = predLD;
            else if (!readBit(fp, prob[7])) p = predVL;
            else if (!readBit(fp, prob[8])) p = predHD;
            else p = predHU;
            d->predY4[j][i] = p;
            d->upMB[mbx].pred[i] = (uint8_t)p;
        }
        d->leftMB.pred[j] = (uint8_t)p;
    }
}

/* Vp8Decoder.hx:772-824 */
static int vp8_reconstruct(Vp8* d, int mbx, int mby) {
    Partition* fp = &d->fp;
    if (d->updateMap) {
        d->segment = !readBit(fp, d->seg_prob[0]) ? readUint(fp, d->seg_prob[1], 1) : readUint(fp, d->seg_prob[2], 1) + 2;
    }
    int skip = d->useSkipProb ? readBit(fp, d->skipProb) : 0;
    memset(d->coeff, 0, sizeof d->coeff);
    vp8_prepareYBR(d, mbx, mby);
    d->usePredY16 = readBit(fp, 145);
    if (d->usePredY16) vp8_parsePredModeY16(d, mbx); else vp8_parsePredModeY4(d, mbx);
    vp8_parsePredModeC8(d);
    if (!skip) {
        skip = vp8_parseResiduals(d, mbx, mby);
    } else {
        if (d->usePredY16) { d->leftMB.nzY16 = 0; d->upMB[mbx].nzY16 = 0; }
        d->leftMB.nzMask = 0; d->upMB[mbx].nzMask = 0;
        d->nzDCMask = 0; d->nzACMask = 0;
    }
    vp8_reconstructMacroblock(d, mbx, mby);
    for (int y = 0; y < 16; y++) {
        int i = (mby * d->YStride + mbx) * 16 + y * d->YStride;
        for (int k = 0; k < 16; k++) d->Y[i + k] = (uint8_t)d->ybr[ybrYY + y][ybrYX + k];
    }
    for (int y = 0; y < 8; y++) {
        int i = (mby * d->CStride + mbx) * 8 + y * d->CStride;
        for (int k = 0; k < 8; k++) { d->Cb[i + k] = (uint8_t)d->ybr[ybrBY + y][ybrBX + k]; d->Cr[i + k] = (uint8_t)d->ybr[ybrRY + y][ybrRX + k]; }
    }
    return skip;
}

/* ---------------- vp8/Filter.hx ---------------- */
static int iabs(int x) { return x < 0 ? -x : x; }
static int clamp127(int x) { return x < -127 ? -127 : (x > 127 ? 127 : x); }     /* Filter.hx:16-18 (Q4) */
static int clamp15(int x) { return x < -15 ? -15 : (x > 15 ? 15 : x); }          /* Filter.hx:20-22 (Q4) */
static int clamp255(int x) { return x < 0 ? 0 : (x > 255 ? 255 : x); }
/* Filter.hx:29-50 */
static void filter2(uint8_t* pix, int level, int index, int iStep, int jStep) {
    for (int n = 0; n < 16; n++) {
        int p1 = pix[index - 2 * jStep], p0 = pix[index - 1 * jStep], q0 = pix[index], q1 = pix[index + jStep];
        if ((iabs(p0 - q0) << 1) + (iabs(p1 - q1) >> 1) > level) { index += iStep; continue; }
        int a = 3 * (q0 - p0) + clamp127(p1 - q1);
        int a1 = clamp15((a + 4) >> 3);
        int a2 = clamp15((a + 3) >> 3);
        pix[index - jStep] = (uint8_t)clamp255(p0 + a2);
        pix[index] = (uint8_t)clamp255(q0 - a1);
        index += iStep;
    }
}
/* Filter.hx:53-111 */
static void filter246(uint8_t* pix, int n, int level, int ilevel, int hlevel, int index, int iStep, int jStep, int fourNotSix) {
    while (n > 0) {
        int p3 = pix[index - 4 * jStep], p2 = pix[index - 3 * jStep], p1 = pix[index - 2 * jStep], p0 = pix[index - jStep];
        int q0 = pix[index], q1 = pix[index + jStep], q2 = pix[index + 2 * jStep], q3 = pix[index + 3 * jStep];
        if ((iabs(p0 - q0) << 1) + (iabs(p1 - q1) >> 1) > level ||
            iabs(p3 - p2) > ilevel || iabs(p2 - p1) > ilevel || iabs(p1 - p0) > ilevel ||
            iabs(q1 - q0) > ilevel || iabs(q2 - q1) > ilevel || iabs(q3 - q2) > ilevel) {
            index += iStep; n--; continue;
        }
        if (iabs(p1 - p0) > hlevel || iabs(q1 - q0) > hlevel) {
            int a = 3 * (q0 - p0) + clamp127(p1 - q1);
            int a1 = clamp15((a + 4) >> 3);
            int a2 = clamp15((a + 3) >> 3);
            pix[index - jStep] = (uint8_t)clamp255(p0 + a2);
            pix[index] = (uint8_t)clamp255(q0 - a1);
        } else if (fourNotSix) {
            int a = 3 * (q0 - p0);
            int a1 = clamp15((a + 4) >> 3);
            int a2 = clamp15((a + 3) >> 3);
            int a3 = (a1 + 1) >> 1;
            pix[index - 2 * jStep] = (uint8_t)clamp255(p1 + a3);
            pix[index - jStep] = (uint8_t)clamp255(p0 + a2);
            pix[index] = (uint8_t)clamp255(q0 - a1);
            pix[index + jStep] = (uint8_t)clamp255(q1 - a3);
        } else {
            int a = clamp127(3 * (q0 - p0) + clamp127(p1 - q1));
            int a1 = (27 * a + 63) >> 7;
            int a2 = (18 * a + 63) >> 7;
            int a3 = (9 * a + 63) >> 7;
            pix[index - 3 * jStep] = (uint8_t)clamp255(p2 + a3);
            pix[index - 2 * jStep] = (uint8_t)clamp255(p1 + a2);
            pix[index - jStep] = (uint8_t)clamp255(p0 + a1);
            pix[index] = (uint8_t)clamp255(q0 - a1);
            pix[index + jStep] = (uint8_t)clamp255(q1 - a2);
            pix[index + 2 * jStep] = (uint8_t)clamp255(q2 - a3);
        }
        index += iStep; n--;
    }
}

/* Vp8Decoder.hx:417-443 */
static void vp8_simpleFilter(Vp8* d) {
    for (int mby = 0; mby < d->mbh; mby++) for (int mbx = 0; mbx < d->mbw; mbx++) {
        FilterParam* f = d->perMB[d->mbw * mby + mbx];
        if (f->level == 0) continue;
        int l = f->level, YS = d->YStride;
        int yIndex = (mby * YS + mbx) * 16;
        if (mbx > 0) filter2(d->Y, l + 4, yIndex, YS, 1);
        if (f->inner) { filter2(d->Y, l, yIndex + 4, YS, 1); filter2(d->Y, l, yIndex + 8, YS, 1); filter2(d->Y, l, yIndex + 12, YS, 1); }
        if (mby > 0) filter2(d->Y, l + 4, yIndex, 1, YS);
        if (f->inner) { filter2(d->Y, l, yIndex + YS * 4, 1, YS); filter2(d->Y, l, yIndex + YS * 8, 1, YS); filter2(d->Y, l, yIndex + YS * 12, 1, YS); }
    }
}
/* Vp8Decoder.hx:445-486 */
static void vp8_normalFilter(Vp8* d) {
    for (int mby = 0; mby < d->mbh; mby++) for (int mbx = 0; mbx < d->mbw; mbx++) {
        FilterParam* f = d->perMB[d->mbw * mby + mbx];
        if (f->level == 0) continue;
        int l = f->level, il = f->ilevel, hl = f->hlevel, YS = d->YStride, CS = d->CStride;
        int yIndex = (mby * YS + mbx) * 16;
        int cIndex = (mby * CS + mbx) * 8;
        if (mbx > 0) {
            filter246(d->Y, 16, l + 4, il, hl, yIndex, YS, 1, 0);
            filter246(d->Cb, 8, l + 4, il, hl, cIndex, CS, 1, 0);
            filter246(d->Cr, 8, l + 4, il, hl, cIndex, CS, 1, 0);
        }
        if (f->inner) {
            filter246(d->Y, 16, l, il, hl, yIndex + 4, YS, 1, 1);
            filter246(d->Y, 16, l, il, hl, yIndex + 8, YS, 1, 1);
            filter246(d->Y, 16, l, il, hl, yIndex + 12, YS, 1, 1);
            filter246(d->Cb, 8, l, il, hl, cIndex + 4, CS, 1, 1);
            filter246(d->Cr, 8, l, il, hl, cIndex + 4, CS, 1, 1);
        }
        if (mby > 0) {
            filter246(d->Y, 16, l + 4, il, hl, yIndex, 1, YS, 0);
            filter246(d->Cb, 8, l + 4, il, hl, cIndex, 1, CS, 0);
            filter246(d->Cr, 8, l + 4, il, hl, cIndex, 1, CS, 0);
        }
        if (f->inner) {
            filter246(d->Y, 16, l, il, hl, yIndex + YS * 4, 1, YS, 1);
            filter246(d->Y, 16, l, il, hl, yIndex + YS * 8, 1, YS, 1);
            filter246(d->Y, 16, l, il, hl, yIndex + YS * 12, 1, YS, 1);
            filter246(d->Cb, 8, l, il, hl, cIndex + CS * 4, 1, CS, 1);
            filter246(d->Cr, 8, l, il, hl, cIndex + CS * 4, 1, CS, 1);
        }
    }
}

/* Vp8Decoder.hx:335-380 (+ ensureImg :186-201) */
static int vp8_decodeFrame(Vp8* d, int keep_unfiltered) {
    int width = d->mbw * 16, height = d->mbh * 16;
    d->YStride = width; d->CStride = width >> 1;
    size_t ysz = (size_t)d->YStride * height, csz = (size_t)d->CStride * (height >> 1);
    d->Y = (uint8_t*)calloc(ysz ? ysz : 1, 1); d->Cb = (uint8_t*)calloc(csz ? csz : 1, 1); d->Cr = (uint8_t*)calloc(csz ? csz : 1, 1);
    int nmb = d->mbw * d->mbh;
    d->perMB = (FilterParam**)calloc(nmb ? nmb : 1, sizeof(FilterParam*));
    d->upMB = (MB*)calloc(d->mbw ? d->mbw : 1, sizeof(MB));
    if (!d->Y || !d->Cb || !d->Cr || !d->perMB || !d->upMB) return fail(WO_ERR_ALLOC, "alloc");
    int ok = vp8_parseOtherHeaders(d);
    if (ok < 0) return ok;
    if (!ok) return fail(WO_ERR_NULL, "vp8: decodeFrame returned null");
    for (int mby = 0; mby < d->mbh; mby++) {
        memset(&d->leftMB, 0, sizeof d->leftMB);
        for (int mbx = 0; mbx < d->mbw; mbx++) {
            int skip = vp8_reconstruct(d, mbx, mby);
            /* Q1: fs aliases the shared per-(segment, i16/i4) record */
            FilterParam* fs = &d->filterParams[d->segment][d->usePredY16 ? 0 : 1];
            fs->inner = fs->inner || !skip;
            d->perMB[d->mbw * mby + mbx] = fs;
        }
    }
    if (d->fp.unexpectedEOF) return fail(WO_ERR_NULL, "vp8: decodeFrame returned null");
    for (int i = 0; i < d->nOP; i++) if (d->op[i].unexpectedEOF) return fail(WO_ERR_NULL, "vp8: decodeFrame returned null");
    if (keep_unfiltered) {
        d->Y0 = (uint8_t*)malloc(ysz ? ysz : 1); d->Cb0 = (uint8_t*)malloc(csz ? csz : 1); d->Cr0 = (uint8_t*)malloc(csz ? csz : 1);
        memcpy(d->Y0, d->Y, ysz); memcpy(d->Cb0, d->Cb, csz); memcpy(d->Cr0, d->Cr, csz);
    }
    if (d->f_level != 0) { if (d->f_simple) vp8_simpleFilter(d); else vp8_normalFilter(d); }
    return WO_OK;
}

/* Test hook: the pixel half of Vp8Decoder.decodeFrame driven from the PACKED buffers of include/webp_amd.h (MB records,
 * sparse coefficient blocks, filter records) instead of the bitstream, so that reconstruction and loop filter can be
 * checked on inputs no encoder produces (full-range coefficients, every mode at every position, arbitrary filter
 * records).  Follows reconstruct() :785-821 per MB in raster order, then simpleFilter / normalFilter.  Absent blocks are
 * zero; every block goes through the full inverseDCT4 (bit-identical to the DC-only / skipped paths).
 * y0/u0/v0 (optional) receive the unfiltered planes. */
int wo_vp8_from_packed(int mbw, int mbh, const uint32_t* mb, const int16_t* coeff, const uint32_t* filt, int filter_type,
                       uint8_t* y, uint8_t* u, uint8_t* v, uint8_t* y0, uint8_t* u0, uint8_t* v0) {
    Vp8* d = (Vp8*)calloc(1, sizeof(Vp8));
    if (!d) return fail(WO_ERR_ALLOC, "alloc");
    d->mbw = mbw; d->mbh = mbh;
    d->YStride = 16 * mbw; d->CStride = 8 * mbw;
    d->Y = y; d->Cb = u; d->Cr = v;
    const int nmb = mbw * mbh;
    FilterParam* fps = (FilterParam*)calloc(nmb ? nmb : 1, sizeof(FilterParam));
    d->perMB = (FilterParam**)calloc(nmb ? nmb : 1, sizeof(FilterParam*));
    if (!fps || !d->perMB) { free(fps); free(d->perMB); free(d); return fail(WO_ERR_ALLOC, "alloc"); }
    for (int mby = 0; mby < mbh; mby++) for (int mbx = 0; mbx < mbw; mbx++) {
        const int idx = mby * mbw + mbx;
        const uint32_t* w = mb + 4 * idx;
        const int has_wht = (w[2] >> 24) & 1;
        memset(d->coeff, 0, sizeof d->coeff);
        vp8_prepareYBR(d, mbx, mby);
        d->usePredY16 = (w[2] >> 25) & 1;
        d->predC8 = (w[2] >> 26) & 3;
        if (d->usePredY16) d->predY16 = w[0] & 15;
        else for (int n = 0; n < 16; n++) d->predY4[n >> 2][n & 3] = (w[n >> 3] >> (4 * (n & 7))) & 15;
        const int16_t* blk = coeff + (size_t)w[3] * 16;
        if (has_wht) { for (int k = 0; k < 16; k++) d->coeff[whtCoeffBase + k] = blk[k]; blk += 16; }
        for (int n = 0; n < 24; n++)
            if ((w[2] >> n) & 1) { for (int k = 0; k < 16; k++) d->coeff[16 * n + k] = blk[k]; blk += 16; }
        if (has_wht) vp8_inverseWHT16(d);
        d->nzACMask = 0xffffff; d->nzDCMask = 0;
        vp8_reconstructMacroblock(d, mbx, mby);
        for (int yy = 0; yy < 16; yy++)
            for (int k = 0; k < 16; k++) d->Y[(size_t)(16 * mby + yy) * d->YStride + 16 * mbx + k] = (uint8_t)d->ybr[ybrYY + yy][ybrYX + k];
        for (int yy = 0; yy < 8; yy++)
            for (int k = 0; k < 8; k++) {
                d->Cb[(size_t)(8 * mby + yy) * d->CStride + 8 * mbx + k] = (uint8_t)d->ybr[ybrBY + yy][ybrBX + k];
                d->Cr[(size_t)(8 * mby + yy) * d->CStride + 8 * mbx + k] = (uint8_t)d->ybr[ybrRY + yy][ybrRX + k];
            }
        fps[idx].level = filt[idx] & 0xff; fps[idx].ilevel = (filt[idx] >> 8) & 0xff;
        fps[idx].hlevel = (filt[idx] >> 16) & 0xff; fps[idx].inner = (filt[idx] >> 24) & 1;
        d->perMB[idx] = &fps[idx];
    }
    if (y0) memcpy(y0, y, (size_t)256 * nmb);
    if (u0) memcpy(u0, u, (size_t)64 * nmb);
    if (v0) memcpy(v0, v, (size_t)64 * nmb);
    if (filter_type == 1) vp8_simpleFilter(d); else if (filter_type == 2) vp8_normalFilter(d);
    free(fps); free(d->perMB); free(d);
    return WO_OK;
}

static void vp8_init(Vp8* d, const uint8_t* data, int len) {   /* Vp8Decoder.hx:88-124 */
    memset(d, 0, sizeof *d);
    d->data = data; d->len = len; d->pos = 0;
}
static void vp8_free_tmp(Vp8* d) { free(d->perMB); free(d->upMB); d->perMB = NULL; d->upMB = NULL; }

/* ===================================================================== */
/* vp8l/HTree.hx, vp8l/Vp8LDecoder.hx, vp8l/Transform.hx                  */
/* ===================================================================== */
/* A byte source that can be the concatenation of two buffers (types/ConcatInput.hx). */
typedef struct { const uint8_t* a; int alen; const uint8_t* b; int blen; int pos; } Src;
static int src_readByte(Src* s) {       /* returns -1 on Eof */
    if (s->pos < s->alen) return s->a[s->pos++];
    if (s->pos - s->alen < s->blen) { int v = s->b[s->pos - s->alen]; s->pos++; return v; }
    return -1;
}

typedef struct { Src* r; int32_t bits; int nBits; int err; } L;    /* Vp8LDecoder.hx:13-16 ; err = sticky exception */

/* Vp8LDecoder.hx:66-76 */
static int l_read(L* d, int n) {
    if (d->err) return 0;
    while (d->nBits < n) {
        int c = src_readByte(d->r);
        if (c < 0) { d->err = WO_ERR_EOF; return 0; }
        d->bits |= (c & 0xFF) << d->nBits;
        d->nBits += 8;
    }
    int u = d->bits & ((1 << n) - 1);
    d->bits >>= n;
    d->nBits -= n;
    return u;
}

typedef struct { int symbol; int children; } HNode;
typedef struct { HNode* nodes; int nnodes, cap; int lut[128]; } HTree;   /* HTree.hx:36-43 */
#define leafNode (-1)
#define lutSize 7
#define lutMask 127

static uint8_t reverseBits8(int v) {      /* HTree.hx:13-30, computed instead of tabulated */
    int r = 0;
    for (int i = 0; i < 8; i++) if (v & (1 << i)) r |= 1 << (7 - i);
    return (uint8_t)r;
}
static int ht_push(HTree* h) {
    if (h->nnodes == h->cap) {
        int nc = h->cap ? h->cap * 2 : 16;
        HNode* nn = (HNode*)realloc(h->nodes, nc * sizeof(HNode));
        if (!nn) return -1;
        h->nodes = nn; h->cap = nc;
    }
    h->nodes[h->nnodes].symbol = 0; h->nodes[h->nnodes].children = 0;
    return h->nnodes++;
}
static int ht_new(HTree* h) { memset(h, 0, sizeof *h); return ht_push(h) < 0 ? WO_ERR_ALLOC : WO_OK; }
static void ht_reset_nodes(HTree* h) { h->nnodes = 0; ht_push(h); }
static void ht_free(HTree* h) { free(h->nodes); h->nodes = NULL; }

/* HTree.hx:45-89 */
static int ht_insert(HTree* h, int symbol, int code, int codeLength) {
    if (symbol > 0xFFFF || codeLength > 0xFE) return fail(WO_ERR_FORMAT, "Invalid Huffman tree");
    int baseCode = 0;
    if (codeLength > lutSize) {
        baseCode = reverseBits8((code >> (codeLength - lutSize)) & 0xFF) >> (8 - lutSize);
    } else {
        baseCode = reverseBits8(code & 0xFF) >> (8 - codeLength);
        for (int i = 0; i < (1 << (lutSize - codeLength)); i++)
            h->lut[baseCode | (i << codeLength)] = (symbol << 8) | (codeLength + 1);
    }
    int n = 0, jump = lutSize;
    while (codeLength > 0) {
        codeLength--;
        if (n >= h->nnodes) return fail(WO_ERR_FORMAT, "Invalid Huffman tree");
        if (h->nodes[n].children == leafNode) return fail(WO_ERR_FORMAT, "Invalid Huffman tree");
        if (h->nodes[n].children == 0) {
            h->nodes[n].children = h->nnodes;
            if (ht_push(h) < 0 || ht_push(h) < 0) return fail(WO_ERR_ALLOC, "alloc");
        }
        n = h->nodes[n].children + ((code >> codeLength) & 1);
        jump--;
        if (jump == 0 && h->lut[baseCode] == 0) h->lut[baseCode] = n << 8;
    }
    if (h->nodes[n].children == 0) h->nodes[n].children = leafNode;
    else if (h->nodes[n].children != leafNode) return fail(WO_ERR_FORMAT, "Invalid Huffman tree");
    h->nodes[n].symbol = symbol;
    return WO_OK;
}

/* HTree.hx:171-202 */
static int codeLengthsToCodes(const int* codeLengths, int ncl, int* codes) {
    int maxCodeLength = 0;
    for (int i = 0; i < ncl; i++) if (maxCodeLength < codeLengths[i]) maxCodeLength = codeLengths[i];
    if (ncl == 0 || maxCodeLength > 15) return fail(WO_ERR_FORMAT, "Invalid Huffman tree");
    int histogram[16] = {0}, nextCodes[16] = {0};
    for (int i = 0; i < ncl; i++) histogram[codeLengths[i]]++;
    int currCode = 0;
    for (int cl = 1; cl < 16; cl++) { currCode = (currCode + histogram[cl - 1]) << 1; nextCodes[cl] = currCode; }
    for (int s = 0; s < ncl; s++) { codes[s] = 0; if (codeLengths[s] > 0) { codes[s] = nextCodes[codeLengths[s]]; nextCodes[codeLengths[s]]++; } }
    return WO_OK;
}

/* HTree.hx:91-115 */
static int ht_build(HTree* h, const int* codeLengths, int ncl) {
    int nSymbols = 0, lastSymbol = 0;
    for (int s = 0; s < ncl; s++) if (codeLengths[s] != 0) { nSymbols++; lastSymbol = s; }
    if (nSymbols == 0) return fail(WO_ERR_FORMAT, "Invalid Huffman tree");
    ht_reset_nodes(h);
    if (nSymbols == 1) return ht_insert(h, lastSymbol, 0, 0);
    int* codes = (int*)malloc(sizeof(int) * ncl);
    if (!codes) return fail(WO_ERR_ALLOC, "alloc");
    int rc = codeLengthsToCodes(codeLengths, ncl, codes);
    for (int s = 0; rc == WO_OK && s < ncl; s++)
        if (codeLengths[s] > 0) rc = ht_insert(h, s, codes[s], codeLengths[s]);
    free(codes);
    return rc;
}
/* HTree.hx:118-128.  (`_nodes.resize(2*nSymbols-1)` leaves unreachable null slots in the
 * Haxe array; here the node indices are shifted but the tree shape is identical.) */
static int ht_buildSimple(HTree* h, int nSymbols, const int* symbols, int alphabetSize) {
    ht_reset_nodes(h);
    for (int i = 0; i < nSymbols; i++) {
        if (symbols[i] >= alphabetSize) return fail(WO_ERR_FORMAT, "Invalid huffman tree");
        int rc = ht_insert(h, symbols[i], i, nSymbols - 1);
        if (rc) return rc;
    }
    return WO_OK;
}
/* HTree.hx:130-169 */
static int ht_slow(HTree* h, L* d, int n) {
    while (h->nodes[n].children != leafNode) {
        if (d->nBits == 0) {
            int c = src_readByte(d->r);
            if (c < 0) { d->err = WO_ERR_EOF; return 0; }
            d->bits = c; d->nBits = 8;
        }
        n = h->nodes[n].children + (d->bits & 1);
        d->bits >>= 1;
        d->nBits--;
    }
    return h->nodes[n].symbol;
}
static int ht_next(HTree* h, L* d) {
    if (d->err) return 0;
    if (d->nBits < lutSize) {
        int c = src_readByte(d->r);
        if (c < 0) return ht_slow(h, d, 0);
        d->bits |= c << d->nBits;
        d->nBits += 8;
    }
    int n = h->lut[d->bits & lutMask];
    if ((n & 0xFF) != 0) {
        int b = (n & 0xFF) - 1;
        d->bits >>= b;
        d->nBits -= b;
        return n >> 8;
    }
    n >>= 8;
    d->bits >>= lutSize;
    d->nBits -= lutSize;
    return ht_slow(h, d, n);
}

/* Transform.hx:13-19 */
typedef struct { int oldWidth; int transformType; int bits; uint8_t* pix; int pixlen; } Transform;
static int nTiles(int dim, int bits) { return (dim + (1 << bits) - 1) >> bits; }   /* Transform.hx:23-25 */

static const uint8_t kCodeLengthCodeOrder[19] = {17, 18, 0, 1, 2, 3, 4, 5, 16, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};
static const int kRepeatBits[3] = {2, 3, 7}, kRepeatOffsets[3] = {3, 3, 11};
static const int kAlphabetSizes[5] = {256 + 24, 256, 256, 256, 40};

/* Vp8LDecoder.hx:50-59 */
static int distanceMap(int w, int code) {
    if (code > 120) return code - 120;
    int distCode = wo_DistanceMap[code - 1];
    int yOffset = distCode >> 4;
    int xOffset = 8 - (distCode & 0xF);
    int dd = yOffset * w + xOffset;
    return dd >= 1 ? dd : 1;
}
/* Vp8LDecoder.hx:337-345 */
static uint32_t lz77Param(L* d, uint32_t symbol) {
    if (symbol < 4) return symbol + 1;
    uint32_t extraBits = (symbol - 2) >> 1;
    uint32_t offset = (2 + (symbol & 1)) << extraBits;
    uint32_t n = (uint32_t)l_read(d, (int)extraBits);
    return offset + n + 1;
}

/* Vp8LDecoder.hx:124-158 */
static int decodeCodeLengths(L* d, int* dst, int ndst, const int* clcl) {
    HTree h; int rc = ht_new(&h); if (rc) return rc;
    rc = ht_build(&h, clcl, 19);
    if (rc) { ht_free(&h); return rc; }
    int maxSymbol = ndst;
    int useLength = l_read(d, 1);
    if (useLength != 0) {
        int n = l_read(d, 3);
        n = 2 + 2 * n;
        int ms = l_read(d, n);
        maxSymbol = ms + 2;
        if (maxSymbol > ndst) { ht_free(&h); return fail(WO_ERR_FORMAT, "Invalid code lengths"); }
    }
    int prevCodeLength = 8, symbol = 0;
    while (symbol < ndst && !d->err) {
        if (maxSymbol == 0) break;
        maxSymbol--;
        int codeLength = ht_next(&h, d);
        if (d->err) break;
        if (codeLength < 16) {
            dst[symbol++] = codeLength;
            if (codeLength != 0) prevCodeLength = codeLength;
            continue;
        }
        int repeat = l_read(d, kRepeatBits[codeLength - 16]);
        repeat += kRepeatOffsets[codeLength - 16];
        if (symbol + repeat > ndst) { ht_free(&h); return fail(WO_ERR_FORMAT, "Invalid code lengths"); }
        int cl = (codeLength == 16) ? prevCodeLength : 0;
        for (int i = 0; i < repeat; i++) dst[symbol++] = cl;
    }
    ht_free(&h);
    return d->err ? fail(d->err, "Eof") : WO_OK;
}

/* Vp8LDecoder.hx:160-183 */
static int decodeHuffmanTree(L* d, HTree* h, int alphabetSize) {
    int useSimple = l_read(d, 1);
    if (useSimple != 0) {
        int nSymbols = l_read(d, 1) + 1;
        int firstSymbolLengthCode = l_read(d, 1) * 7 + 1;
        int symbols[2] = {0, 0};
        symbols[0] = l_read(d, firstSymbolLengthCode);
        if (nSymbols == 2) symbols[1] = l_read(d, 8);
        if (d->err) return fail(d->err, "Eof");
        return ht_buildSimple(h, nSymbols, symbols, alphabetSize);
    }
    int nCodes = l_read(d, 4) + 4;
    if (nCodes > 19) return fail(WO_ERR_FORMAT, "Invalid Huffman tree");
    int clcl[19] = {0};
    for (int i = 0; i < nCodes; i++) clcl[kCodeLengthCodeOrder[i]] = l_read(d, 3);
    if (d->err) return fail(d->err, "Eof");
    int* codeLengths = (int*)calloc(alphabetSize, sizeof(int));
    if (!codeLengths) return fail(WO_ERR_ALLOC, "alloc");
    int rc = decodeCodeLengths(d, codeLengths, alphabetSize, clcl);
    if (rc == WO_OK) rc = ht_build(h, codeLengths, alphabetSize);
    free(codeLengths);
    return rc;
}

static int decodePix(L* d, int w, int h, int minCap, int topLevel, uint8_t** out, int* outlen);

typedef struct { HTree t[5]; } HGroup;

/* Vp8LDecoder.hx:222-335 (with decodeHuffmanGroups :185-220 inlined at the top) */
static int decodePix(L* d, int w, int h, int minCap, int topLevel, uint8_t** out, int* outlen) {
    (void)minCap;
    int rc = WO_OK;
    int ccBits = 0, ccShift = 0; uint32_t* ccEntries = NULL; int ccLen = 0;
    HGroup* hGroups = NULL; int nGroups = 0; uint8_t* hPix = NULL; int hPixLen = 0; int hBits = 0;
    uint8_t* pix = NULL;
    if (l_read(d, 1) != 0) {
        ccBits = l_read(d, 4);
        if (d->err) return fail(d->err, "Eof");
        if (ccBits < 1 || ccBits > 11) return fail(WO_ERR_FORMAT, "Invalid color cache parameters");
        ccShift = 32 - ccBits;
        ccLen = 1 << ccBits;
        ccEntries = (uint32_t*)calloc(ccLen, sizeof(uint32_t));
    }
    /* decodeHuffmanGroups */
    int maxHGroupIndex = 0;
    if (topLevel) {
        int useMeta = l_read(d, 1);
        if (useMeta != 0) {
            hBits = l_read(d, 3) + 2;
            rc = decodePix(d, nTiles(w, hBits), nTiles(h, hBits), 0, 0, &hPix, &hPixLen);
            if (rc) goto done;
            for (int p = 0; p < hPixLen; p += 4) {
                int i = (hPix[p] << 8) | hPix[p + 1];
                if (maxHGroupIndex < i) maxHGroupIndex = i;
            }
        }
    }
    nGroups = maxHGroupIndex + 1;
    hGroups = (HGroup*)calloc(nGroups, sizeof(HGroup));
    if (!hGroups) { rc = fail(WO_ERR_ALLOC, "alloc"); nGroups = 0; goto done; }
    for (int i = 0; i < nGroups; i++) for (int j = 0; j < 5; j++) ht_new(&hGroups[i].t[j]);
    for (int i = 0; i < nGroups; i++) {
        for (int j = 0; j < 5; j++) {
            int alphabetSize = kAlphabetSizes[j];
            if (j == 0 && ccBits > 0) alphabetSize += 1 << ccBits;
            rc = decodeHuffmanTree(d, &hGroups[i].t[j], alphabetSize);
            if (rc) goto done;
        }
    }
    if (d->err) { rc = fail(d->err, "Eof"); goto done; }
    {
        int hMask = hBits != 0 ? (1 << hBits) - 1 : 0;
        int tilesPerRow = hBits != 0 ? nTiles(w, hBits) : 0;
        long long plen64 = 4LL * w * h;
        int plen = (int)plen64;
        pix = (uint8_t*)calloc(plen ? plen : 1, 1);
        if (!pix) { rc = fail(WO_ERR_ALLOC, "alloc"); goto done; }
        int p = 0, cachedP = 0, x = 0, y = 0;
        HGroup* hg = &hGroups[0];
        int lookupHG = hMask != 0;
        while (p < plen) {
            if (lookupHG) {
                int i = 4 * (tilesPerRow * (y >> hBits) + (x >> hBits));
                hg = &hGroups[(hPix[i] << 8) | hPix[i + 1]];
            }
            int green = ht_next(&hg->t[0], d);
            if (d->err) { rc = fail(d->err, "Eof"); goto done; }
            if (green < 256) {
                int red = ht_next(&hg->t[1], d);
                int blue = ht_next(&hg->t[2], d);
                int alpha = ht_next(&hg->t[3], d);
                if (d->err) { rc = fail(d->err, "Eof"); goto done; }
                pix[p++] = (uint8_t)red; pix[p++] = (uint8_t)green; pix[p++] = (uint8_t)blue; pix[p++] = (uint8_t)alpha;
                x++;
                if (x == w) { x = 0; y++; }
                lookupHG = hMask != 0 && (x & hMask) == 0;
            } else if (green < 256 + 24) {
                int length = (int)lz77Param(d, (uint32_t)(green - 256));
                int distSym = ht_next(&hg->t[4], d);
                int distCode = (int)lz77Param(d, (uint32_t)distSym);
                if (d->err) { rc = fail(d->err, "Eof"); goto done; }
                int dist = distanceMap(w, distCode);
                int pEnd = p + 4 * length;
                int q = p - 4 * dist;
                int qEnd = pEnd - 4 * dist;
                if (p < 0 || q < 0 || pEnd > plen || qEnd > plen) { rc = fail(WO_ERR_FORMAT, "Invalid LZ77 parameters"); goto done; }
                /* both branches of :290-299 are a forward byte copy */
                while (p < pEnd) pix[p++] = pix[q++];
                x += length;
                while (x >= w) { x -= w; y++; }
                lookupHG = hMask != 0;
            } else {
                const uint32_t colorCacheMultiplier = 0x1e35a7bd;
                if (!ccEntries) { rc = fail(WO_ERR_NULL, "null color cache"); goto done; }
                while (cachedP < p) {
                    uint32_t argb = ((uint32_t)pix[cachedP] << 16) | ((uint32_t)pix[cachedP + 1] << 8) | pix[cachedP + 2] | ((uint32_t)pix[cachedP + 3] << 24);
                    ccEntries[(argb * colorCacheMultiplier) >> ccShift] = argb;
                    cachedP += 4;
                }
                green -= 256 + 24;
                if (green >= ccLen) { rc = fail(WO_ERR_FORMAT, "Invalid color cache index"); goto done; }
                uint32_t argb = ccEntries[green];
                pix[p++] = (argb >> 16) & 0xFF; pix[p++] = (argb >> 8) & 0xFF; pix[p++] = argb & 0xFF; pix[p++] = (argb >> 24) & 0xFF;
                x++;
                if (x == w) { x = 0; y++; }
                lookupHG = hMask != 0 && (x & hMask) == 0;
            }
        }
        *out = pix; *outlen = plen; pix = NULL;
    }
done:
    for (int i = 0; i < nGroups; i++) for (int j = 0; j < 5; j++) ht_free(&hGroups[i].t[j]);
    free(hGroups); free(hPix); free(ccEntries); free(pix);
    return rc;
}

/* The pixel loop of decodePix (Vp8LDecoder.hx:254-333) driven by a token stream (include/webp_amd.h, wp_vp8l_tokens_frame)
 * instead of by Huffman reads: the checker for wp_vp8l_resolve[_dev].  Literal store :283-290, LZ77 copy :308-316 (a forward
 * byte copy), colour-cache fill up to the current pixel and lookup :320-333.  Returns 0, or the error decodePix would raise. */
int wo_resolve_tokens(int w, int h, int ccBits, const uint32_t* tok, long long nWords, uint8_t* pix) {
    long long plen = 4LL * w * h, p = 0, cachedP = 0, pos = 0;
    int ccLen = ccBits ? 1 << ccBits : 0, ccShift = 32 - ccBits, rc = 0;
    uint32_t* ccEntries = ccLen ? (uint32_t*)calloc(ccLen, sizeof(uint32_t)) : NULL;
    while (pos < nWords) {
        uint32_t kind = tok[pos] >> 30, arg = tok[pos] & 0x3fffffffu;
        if (kind == 0) {
            if (arg > nWords - pos - 1 || 4LL * arg > plen - p) { rc = fail(WO_ERR_FORMAT, "token stream: literal run out of range"); break; }
            for (uint32_t i = 0; i < arg; i++) {
                uint32_t v = tok[pos + 1 + i];
                pix[p++] = (uint8_t)v; pix[p++] = (uint8_t)(v >> 8); pix[p++] = (uint8_t)(v >> 16); pix[p++] = (uint8_t)(v >> 24);
            }
            pos += 1 + (long long)arg;
        } else if (kind == 1) {
            if (nWords - pos < 2) { rc = fail(WO_ERR_FORMAT, "token stream: copy without a distance"); break; }
            long long dist = tok[pos + 1], pEnd = p + 4LL * arg, q = p - 4 * dist;
            if (dist == 0 || q < 0 || pEnd > plen) { rc = fail(WO_ERR_FORMAT, "Invalid LZ77 parameters"); break; }
            while (p < pEnd) pix[p++] = pix[q++];
            pos += 2;
        } else if (kind == 2) {
            if (!ccEntries) { rc = fail(WO_ERR_NULL, "null color cache"); break; }
            if (p >= plen) { rc = fail(WO_ERR_FORMAT, "token stream: too many pixels"); break; }
            while (cachedP < p) {
                uint32_t argb = ((uint32_t)pix[cachedP] << 16) | ((uint32_t)pix[cachedP + 1] << 8) | pix[cachedP + 2] | ((uint32_t)pix[cachedP + 3] << 24);
                ccEntries[(argb * 0x1e35a7bdu) >> ccShift] = argb;
                cachedP += 4;
            }
            if ((int)arg >= ccLen) { rc = fail(WO_ERR_FORMAT, "Invalid color cache index"); break; }
            uint32_t argb = ccEntries[arg];
            pix[p++] = (argb >> 16) & 0xFF; pix[p++] = (argb >> 8) & 0xFF; pix[p++] = argb & 0xFF; pix[p++] = (argb >> 24) & 0xFF;
            pos += 1;
        } else { rc = fail(WO_ERR_FORMAT, "token stream: unknown token kind"); break; }
    }
    if (!rc && p != plen) rc = fail(WO_ERR_FORMAT, "token stream: fewer pixels than the image holds");
    free(ccEntries);
    return rc;
}

/* Vp8LDecoder.hx:78-122 */
static int decodeTransform(L* d, int* w, int h, Transform* t) {
    t->oldWidth = *w;
    t->transformType = l_read(d, 2);
    t->bits = 0; t->pix = NULL; t->pixlen = 0;
    if (d->err) return fail(d->err, "Eof");
    switch (t->transformType) {
    case 0: case 1: {
        t->bits = l_read(d, 3) + 2;
        return decodePix(d, nTiles(*w, t->bits), nTiles(h, t->bits), 0, 0, &t->pix, &t->pixlen);
    }
    case 2: return WO_OK;
    default: {
        int nColors = l_read(d, 8) + 1;
        t->bits = nColors <= 2 ? 3 : (nColors <= 4 ? 2 : (nColors <= 16 ? 1 : 0));
        int finalNumColors = 1 << (8 >> t->bits);
        *w = nTiles(*w, t->bits);
        uint8_t* pix = NULL; int plen = 0;
        int rc = decodePix(d, nColors, 1, 4 * finalNumColors, 0, &pix, &plen);
        if (rc) return rc;
        uint8_t* ex = (uint8_t*)calloc(4 * finalNumColors, 1);
        ex[0] = pix[0]; ex[1] = pix[1]; ex[2] = pix[2]; ex[3] = pix[3];
        for (int p = 4; p < 4 * nColors; p += 4) {
            ex[p] = (uint8_t)(pix[p] + ex[p - 4]); ex[p + 1] = (uint8_t)(pix[p + 1] + ex[p - 3]);
            ex[p + 2] = (uint8_t)(pix[p + 2] + ex[p - 2]); ex[p + 3] = (uint8_t)(pix[p + 3] + ex[p - 1]);
        }
        free(pix);
        t->pix = ex; t->pixlen = 4 * finalNumColors;
        return WO_OK; }
    }
}

/* ---- Transform.hx inverse transforms (exported individually for kernel unit tests) ---- */
static int avg2(int a, int b) { return ((a + b) / 2) & 0xff; }                                    /* Transform.hx:283-285 */
static int clampAddSubtractFull(int a, int b, int c) { int x = a + b - c; return x < 0 ? 0 : (x > 255 ? 255 : x); }
static int clampAddSubtractHalf(int a, int b) { int x = a + (a - b) / 2; return x < 0 ? 0 : (x > 255 ? 255 : x); }  /* truncating / (Q10) */

/* Transform.hx:34-188.  pix is RGBA (Q9); all byte stores wrap mod 256. */
void wo_inverse_predictor(int oldWidth, int bits, const uint8_t* tpix, uint8_t* pix, int h) {
    if (oldWidth == 0 || h == 0) return;
    pix[3] = (uint8_t)(pix[3] + 0xFF);
    int p = 4, mask = (1 << bits) - 1;
    for (int x = 1; x < oldWidth; x++) {
        for (int k = 0; k < 4; k++) pix[p + k] = (uint8_t)(pix[p + k] + pix[p - 4 + k]);
        p += 4;
    }
    int top = 0, tilesPerRow = nTiles(oldWidth, bits);
    for (int y = 1; y < h; y++) {
        for (int k = 0; k < 4; k++) pix[p + k] = (uint8_t)(pix[p + k] + pix[top + k]);
        p += 4; top += 4;
        int q = 4 * (y >> bits) * tilesPerRow;
        int predictorMode = tpix[q + 1] & 0x0F;
        q += 4;
        for (int x = 1; x < oldWidth; x++) {
            if ((x & mask) == 0) { predictorMode = tpix[q + 1] & 0x0F; q += 4; }
            switch (predictorMode) {
            case 0: pix[p + 3] = (uint8_t)(pix[p + 3] + 0xFF); break;
            case 1: for (int k = 0; k < 4; k++) pix[p + k] = (uint8_t)(pix[p + k] + pix[p - 4 + k]); break;
            case 2: for (int k = 0; k < 4; k++) pix[p + k] = (uint8_t)(pix[p + k] + pix[top + k]); break;
            case 3: for (int k = 0; k < 4; k++) pix[p + k] = (uint8_t)(pix[p + k] + pix[top + 4 + k]); break;
            case 4: for (int k = 0; k < 4; k++) pix[p + k] = (uint8_t)(pix[p + k] + pix[top - 4 + k]); break;
            case 5: for (int k = 0; k < 4; k++) pix[p + k] = (uint8_t)(pix[p + k] + avg2(avg2(pix[p - 4 + k], pix[top + 4 + k]), pix[top + k])); break;
            case 6: for (int k = 0; k < 4; k++) pix[p + k] = (uint8_t)(pix[p + k] + avg2(pix[p - 4 + k], pix[top - 4 + k])); break;
            case 7: for (int k = 0; k < 4; k++) pix[p + k] = (uint8_t)(pix[p + k] + avg2(pix[p - 4 + k], pix[top + k])); break;
            case 8: for (int k = 0; k < 4; k++) pix[p + k] = (uint8_t)(pix[p + k] + avg2(pix[top - 4 + k], pix[top + k])); break;
            case 9: for (int k = 0; k < 4; k++) pix[p + k] = (uint8_t)(pix[p + k] + avg2(pix[top + k], pix[top + 4 + k])); break;
            case 10: for (int k = 0; k < 4; k++) pix[p + k] = (uint8_t)(pix[p + k] + avg2(avg2(pix[p - 4 + k], pix[top - 4 + k]), avg2(pix[top + k], pix[top + 4 + k]))); break;
            case 11: {
                int l = 0, t = 0;
                for (int k = 0; k < 4; k++) { l += iabs(pix[top - 4 + k] - pix[top + k]); t += iabs(pix[top - 4 + k] - pix[p - 4 + k]); }
                if (l < t) { for (int k = 0; k < 4; k++) pix[p + k] = (uint8_t)(pix[p + k] + pix[p - 4 + k]); }
                else { for (int k = 0; k < 4; k++) pix[p + k] = (uint8_t)(pix[p + k] + pix[top + k]); }
                break; }
            case 12: for (int k = 0; k < 4; k++) pix[p + k] = (uint8_t)(pix[p + k] + clampAddSubtractFull(pix[p - 4 + k], pix[top + k], pix[top - 4 + k])); break;
            case 13: for (int k = 0; k < 4; k++) pix[p + k] = (uint8_t)(pix[p + k] + clampAddSubtractHalf(avg2(pix[p - 4 + k], pix[top + k]), pix[top - 4 + k])); break;
            default: break;   /* modes 14, 15: no case in the text (Q10) */
            }
            p += 4; top += 4;
        }
    }
}
static int uint8ToInt8(int x) { return (x & 0xFF) >= 0x80 ? (x & 0xFF) - 0x100 : (x & 0xFF); }
/* Transform.hx:190-222 */
void wo_inverse_cross_color(int oldWidth, int bits, const uint8_t* tpix, uint8_t* pix, int h) {
    int greenToRed = 0, greenToBlue = 0, redToBlue = 0;
    int p = 0, mask = (1 << bits) - 1, tilesPerRow = nTiles(oldWidth, bits);
    for (int y = 0; y < h; y++) {
        int q = 4 * ((y >> bits) * tilesPerRow);
        for (int x = 0; x < oldWidth; x++) {
            if ((x & mask) == 0) {
                redToBlue = uint8ToInt8(tpix[q + 0]); greenToBlue = uint8ToInt8(tpix[q + 1]); greenToRed = uint8ToInt8(tpix[q + 2]);
                q += 4;
            }
            int32_t red = pix[p + 0], green = pix[p + 1], blue = pix[p + 2];
            red += (int32_t)((uint32_t)(greenToRed * uint8ToInt8(green)) >> 5);
            blue += (int32_t)((uint32_t)(greenToBlue * uint8ToInt8(green)) >> 5);
            blue = (blue & 0xff) + (int32_t)((uint32_t)(redToBlue * uint8ToInt8(red)) >> 5);
            pix[p + 0] = (uint8_t)red; pix[p + 2] = (uint8_t)blue;
            p += 4;
        }
    }
}
/* Transform.hx:225-234 */
void wo_inverse_subtract_green(uint8_t* pix, int len) {
    for (int p = 0; p < len; p += 4) { int g = pix[p + 1]; pix[p] = (uint8_t)(pix[p] + g); pix[p + 2] = (uint8_t)(pix[p + 2] + g); }
}
/* Transform.hx:236-281.  bits==0 works in place (dst may equal pix); otherwise dst is 4*oldWidth*h. */
void wo_inverse_color_indexing(int oldWidth, int bits, const uint8_t* tpix, const uint8_t* pix, int pixlen, uint8_t* dst, int h) {
    if (bits == 0) {
        for (int p = 0; p < pixlen; p += 4) { int i = 4 * pix[p + 1]; dst[p] = tpix[i]; dst[p + 1] = tpix[i + 1]; dst[p + 2] = tpix[i + 2]; dst[p + 3] = tpix[i + 3]; }
        return;
    }
    int vMask = 0, xMask = 0, bitsPerPixel = 8 >> bits;
    switch (bits) { case 1: vMask = 0x0f; xMask = 0x01; break; case 2: vMask = 0x03; xMask = 0x03; break; case 3: vMask = 0x01; xMask = 0x07; break; }
    int dd = 0, p = 0; uint32_t v = 0;
    for (int y = 0; y < h; y++) for (int x = 0; x < oldWidth; x++) {
        if ((x & xMask) == 0) { v = pix[p + 1]; p += 4; }
        int i = 4 * (v & vMask);
        dst[dd] = tpix[i]; dst[dd + 1] = tpix[i + 1]; dst[dd + 2] = tpix[i + 2]; dst[dd + 3] = tpix[i + 3];
        dd += 4;
        v >>= bitsPerPixel;
    }
}

/* Vp8LDecoder.hx:347-409.  Returns RGBA pixels (Q9); caller swizzles. */
static int vp8l_decode(Src* src, uint8_t** outpix, int* ow, int* oh) {
    L d; memset(&d, 0, sizeof d); d.r = src;
    int magic = l_read(&d, 8);
    if (d.err) return fail(d.err, "Eof");
    if (magic != 0x2f) return fail(WO_ERR_FORMAT, "vp8l: invalid header");
    int w = l_read(&d, 14) + 1;
    int h = l_read(&d, 14) + 1;
    l_read(&d, 1);
    if (l_read(&d, 3) != 0) return d.err ? fail(d.err, "Eof") : fail(WO_ERR_FORMAT, "vp8l: invalid version");
    if (d.err) return fail(d.err, "Eof");
    Transform tr[4]; int nTransforms = 0; int seen[4] = {0, 0, 0, 0};
    int originalW = w, rc = WO_OK;
    uint8_t* pix = NULL; int plen = 0;
    for (;;) {
        int more = l_read(&d, 1);
        if (d.err) { rc = fail(d.err, "Eof"); goto done; }
        if (more == 0) break;
        Transform t;
        rc = decodeTransform(&d, &w, h, &t);
        if (rc) { free(t.pix); goto done; }
        if (seen[t.transformType]) { free(t.pix); rc = fail(WO_ERR_FORMAT, "vp8l: repeated transform"); goto done; }
        seen[t.transformType] = 1;
        tr[nTransforms++] = t;
    }
    rc = decodePix(&d, w, h, 0, 1, &pix, &plen);
    if (rc) goto done;
    for (int i = nTransforms - 1; i >= 0; i--) {
        Transform* t = &tr[i];
        switch (t->transformType) {
        case 0: wo_inverse_predictor(t->oldWidth, t->bits, t->pix, pix, h); break;
        case 1: wo_inverse_cross_color(t->oldWidth, t->bits, t->pix, pix, h); break;
        case 2: wo_inverse_subtract_green(pix, plen); break;
        case 3:
            if (t->bits == 0) wo_inverse_color_indexing(t->oldWidth, 0, t->pix, pix, plen, pix, h);
            else {
                int nl = 4 * t->oldWidth * h;
                uint8_t* dst = (uint8_t*)calloc(nl ? nl : 1, 1);
                wo_inverse_color_indexing(t->oldWidth, t->bits, t->pix, pix, plen, dst, h);
                free(pix); pix = dst; plen = nl;
            }
            break;
        }
    }
    *outpix = pix; pix = NULL; *ow = originalW; *oh = h;
done:
    for (int i = 0; i < nTransforms; i++) free(tr[i].pix);
    free(pix);
    return rc;
}

/* ===================================================================== */
/* Tools.hx, WebPDecoder.hx                                               */
/* ===================================================================== */
/* Tools.hx:16-39 (Q7).  Compile this file with -ffp-contract=off. */
void wo_yuv420_to_argb(const uint8_t* y, int ystride, const uint8_t* cb, const uint8_t* cr, int cstride,
                       const uint8_t* a, int width, int height, uint8_t* argb) {
    for (int row = 0; row < height; row++) for (int col = 0; col < width; col++) {
        int yIndex = row * ystride + col;
        int cbCrIndex = (row >> 1) * cstride + (col >> 1);
        int Y = y[yIndex] & 0xFF;
        int Cb = (cb[cbCrIndex] & 0xFF) - 128;
        int Cr = (cr[cbCrIndex] & 0xFF) - 128;
        double R = Y + (1.402 * Cr);
        double G = Y - (0.344136 * Cb) - (0.714136 * Cr);
        double B = Y + (1.772 * Cb);
        R = floor(fmax(0, fmin(255, R)) + 0.5);      /* Haxe Math.round = floor(x + 0.5) */
        G = floor(fmax(0, fmin(255, G)) + 0.5);
        B = floor(fmax(0, fmin(255, B)) + 0.5);
        int i = (row * width + col) * 4;
        argb[i] = a ? a[row * width + col] : 0xFF;
        argb[i + 1] = (uint8_t)(int)R; argb[i + 2] = (uint8_t)(int)G; argb[i + 3] = (uint8_t)(int)B;
    }
}

/* WebPDecoder.hx:357-411 */
void wo_unfilter_alpha(uint8_t* alpha, int len, int alphaStride, int filter) {
    if (len == 0 || alphaStride == 0) return;
    switch (filter) {
    case 1:
        for (int i = 1; i < alphaStride; i++) alpha[i] = (uint8_t)(alpha[i] + alpha[i - 1]);
        for (int i = alphaStride; i < len; i += alphaStride) {
            alpha[i] = (uint8_t)(alpha[i] + alpha[i - alphaStride]);
            for (int j = 1; j < alphaStride; j++) alpha[i + j] = (uint8_t)(alpha[i + j] + alpha[i + j - 1]);
        }
        break;
    case 2:
        for (int i = 1; i < alphaStride; i++) alpha[i] = (uint8_t)(alpha[i] + alpha[i - 1]);
        for (int i = alphaStride; i < len; i++) alpha[i] = (uint8_t)(alpha[i] + alpha[i - alphaStride]);
        break;
    case 3:
        for (int i = 1; i < alphaStride; i++) alpha[i] = (uint8_t)(alpha[i] + alpha[i - 1]);
        for (int i = alphaStride; i < len; i += alphaStride) {
            alpha[i] = (uint8_t)(alpha[i] + alpha[i - alphaStride]);
            for (int j = 1; j < alphaStride; j++) {
                int c = alpha[i + j - alphaStride - 1], b = alpha[i + j - alphaStride], a = alpha[i + j - 1];
                int x = a + b - c;
                x = x < 0 ? 0 : (x > 255 ? 255 : x);
                alpha[i + j] = (uint8_t)(alpha[i + j] + x);
            }
        }
        break;
    }
}

/* ---- result objects (mirror of Image.hx) ---- */
enum { WO_YUV420 = 0, WO_ARGB = 1, WO_ANIM = 2 };
typedef struct wo_image {
    int kind;
    int width, height;                 /* header.width/height (canvas for Anim; AnimFrameHeader for frames) */
    int keyFrame, versionNumber, showFrame, firstPartitionLen, xScale, yScale;
    int has_header;
    /* Yuv420 */
    uint8_t *y, *u, *v, *a; int ystride, uvstride, mbw, mbh; int alen;
    uint8_t *y0, *u0, *v0;             /* unfiltered planes (test hook) */
    /* Argb */
    uint8_t* argb; int stride; int argblen;
    /* Anim */
    struct wo_image* frames; int nframes;
    int fx, fy, duration, blend, dispose;
} wo_image;

static void img_free_fields(wo_image* im) {
    free(im->y); free(im->u); free(im->v); free(im->a); free(im->argb); free(im->y0); free(im->u0); free(im->v0);
    for (int i = 0; i < im->nframes; i++) img_free_fields(&im->frames[i]);
    free(im->frames);
    memset(im, 0, sizeof *im);
}
void wo_free(wo_image* im) { if (im) { img_free_fields(im); free(im); } }

static int g_keep_unfiltered = 0;
void wo_set_keep_unfiltered(int v) { g_keep_unfiltered = v; }

/* WebPDecoder.hx:90-103 */
static int readVp8(const uint8_t* data, int len, wo_image* im) {
    Vp8* d = (Vp8*)malloc(sizeof(Vp8));
    if (!d) return fail(WO_ERR_ALLOC, "alloc");
    vp8_init(d, data, len);
    int rc = vp8_decodeFrameHeader(d);
    if (rc == WO_OK) rc = vp8_decodeFrame(d, g_keep_unfiltered);
    vp8_free_tmp(d);
    if (rc != WO_OK) { free(d->Y); free(d->Cb); free(d->Cr); free(d->Y0); free(d->Cb0); free(d->Cr0); free(d); return rc; }
    free(im->y); free(im->u); free(im->v); free(im->y0); free(im->u0); free(im->v0);
    im->y = d->Y; im->u = d->Cb; im->v = d->Cr; im->ystride = d->YStride; im->uvstride = d->CStride;
    im->y0 = d->Y0; im->u0 = d->Cb0; im->v0 = d->Cr0;
    im->mbw = d->mbw; im->mbh = d->mbh;
    im->width = d->width; im->height = d->height;
    im->keyFrame = d->keyFrame; im->versionNumber = d->versionNumber; im->showFrame = d->showFrame;
    im->firstPartitionLen = d->firstPartitionLen; im->xScale = d->xScale; im->yScale = d->yScale;
    im->has_header = 1;
    free(d);
    return WO_OK;
}

/* WebPDecoder.hx:105-131 (decode + in-place RGBA->ARGB swizzle) */
static int readVp8l(const uint8_t* data, int len, uint8_t** argb, int* w, int* h) {
    Src s = {data, len, NULL, 0, 0};
    uint8_t* pix = NULL;
    int rc = vp8l_decode(&s, &pix, w, h);
    if (rc) return rc;
    int n = 4 * (*w) * (*h);
    for (int i = 0; i < n; i += 4) {
        uint8_t r = pix[i], g = pix[i + 1], b = pix[i + 2], a = pix[i + 3];
        pix[i] = a; pix[i + 1] = r; pix[i + 2] = g; pix[i + 3] = b;
    }
    *argb = pix;
    return WO_OK;
}

/* WebPDecoder.hx:311-355 */
static int readAlpha(const uint8_t* data, int len, int widthMinusOne, int heightMinusOne, int compression, uint8_t** out, int* outlen) {
    if (compression == 0) {
        int w = widthMinusOne + 1, h = heightMinusOne + 1;
        if (len < w * h) return fail(WO_ERR_FORMAT, "Failed to read alpha values: Eof");
        uint8_t* alpha = (uint8_t*)malloc((w * h) != 0 ? w * h : 1);
        memcpy(alpha, data, (size_t)w * h);
        *out = alpha; *outlen = w * h;
        return WO_OK;
    } else if (compression == 1) {
        if (widthMinusOne > 0x3fff || heightMinusOne > 0x3fff) return fail(WO_ERR_FORMAT, "Invalid format: dimensions too large");
        uint8_t header[5];
        header[0] = 0x2f;
        header[1] = widthMinusOne & 0xff;
        header[2] = (uint8_t)((widthMinusOne >> 8) | ((heightMinusOne & 0x3f) << 6));
        header[3] = (heightMinusOne >> 2) & 0xff;
        header[4] = (heightMinusOne >> 10) & 0xff;
        Src s = {header, 5, data, len, 0};
        uint8_t* pix = NULL; int w, h;
        int rc = vp8l_decode(&s, &pix, &w, &h);
        if (rc) return rc;
        int n = w * h;
        uint8_t* alpha = (uint8_t*)malloc(n ? n : 1);
        for (int i = 0; i < n; i++) alpha[i] = pix[i * 4 + 1];
        free(pix);
        *out = alpha; *outlen = n;
        return WO_OK;
    }
    return fail(WO_ERR_FORMAT, "Invalid format: unsupported compression type");
}

static uint32_t rd32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static int rd24(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16); }

/* WebPDecoder.hx:258-309 */
static int readFrameData(const uint8_t* data, int len, int pos, int fwm1, int fhm1, wo_image* fr) {
    uint8_t* alpha = NULL; int alen = 0;
    int have_lossy = 0, have_lossless = 0;
    uint8_t* argb = NULL; int lw = 0, lh = 0;
    int rc = WO_OK;
    while (pos < len) {
        if (len - pos < 8) break;
        const uint8_t* id = data + pos;
        int32_t dataLen = (int32_t)rd32(data + pos + 4);
        pos += 8;
        if (dataLen < 0 || dataLen > (len - pos)) { rc = fail(WO_ERR_FORMAT, "Invalid ANMF subchunk length"); goto bad; }
        const uint8_t* cd = data + pos;
        pos += dataLen;
        if ((dataLen & 1) == 1 && pos < len) pos++;
        if (!memcmp(id, "ALPH", 4)) {
            free(alpha); alpha = NULL; alen = 0;
            if (dataLen >= 1) {
                int flags = cd[0];
                uint8_t* a = NULL; int al = 0;
                if (readAlpha(cd + 1, dataLen - 1, fwm1, fhm1, flags & 0x03, &a, &al) == WO_OK) {
                    wo_unfilter_alpha(a, al, fwm1 + 1, (flags >> 2) & 0x03);
                    alpha = a; alen = al;
                }   /* failures are swallowed -> opaque (:283-290) */
            }
        } else if (!memcmp(id, "VP8 ", 4)) {
            rc = readVp8(cd, dataLen, fr);
            if (rc) goto bad;
            have_lossy = 1;
        } else if (!memcmp(id, "VP8L", 4)) {
            free(argb); argb = NULL;
            rc = readVp8l(cd, dataLen, &argb, &lw, &lh);
            if (rc) goto bad;
            have_lossless = 1;
        }
    }
    if (have_lossless) {
        fr->kind = WO_ARGB; fr->argb = argb; fr->stride = 4 * lw; fr->argblen = 4 * lw * lh;
        free(alpha);
        return WO_OK;
    }
    if (have_lossy) {
        fr->kind = WO_YUV420; fr->a = alpha; fr->alen = alen;
        return WO_OK;
    }
    rc = fail(WO_ERR_FORMAT, "Invalid format: no VP8/VP8L chunk found in ANMF");
bad:
    free(alpha); free(argb);
    return rc;
}

/* WebPDecoder.hx:179-256 */
int wo_decode(const uint8_t* data, int len, wo_image** out) {
    *out = NULL;
    wo_image* im = (wo_image*)calloc(1, sizeof(wo_image));
    if (!im) return fail(WO_ERR_ALLOC, "alloc");
    int rc = WO_OK;
    int pos = 0;
    /* readRiffHeader :37-45 */
    if (len < 4) { rc = fail(WO_ERR_EOF, "Eof"); goto bad; }
    if (memcmp(data, "RIFF", 4)) { rc = fail(WO_ERR_FORMAT, "Missing RIFF chunk header"); goto bad; }
    if (len < 12) { rc = fail(WO_ERR_EOF, "Eof"); goto bad; }
    int32_t chunkLen = (int32_t)rd32(data + 4);
    if (chunkLen < 4) { rc = fail(WO_ERR_FORMAT, "Short chunk data"); goto bad; }
    pos = 12;
    int totalLen = chunkLen - 4;
    int widthMinusOne = 0, heightMinusOne = 0;
    uint8_t* alpha = NULL; int alen = 0;
    int have_lossy = 0, have_lossless = 0;
    uint8_t* argb = NULL; int lw = 0, lh = 0;
    wo_image lossy; memset(&lossy, 0, sizeof lossy);
    wo_image* frames = NULL; int nframes = 0;
    /* The reference first collects all chunks (decodeChunks :246-256: a container error anywhere in the file is raised
     * before any payload is looked at), then processes them in order. */
    {
        int p2 = pos, t2 = totalLen;
        while (t2 > 0) {
            if (len - p2 < 8) { rc = fail(WO_ERR_EOF, "Eof"); goto bad2; }
            int32_t dl = (int32_t)rd32(data + p2 + 4);
            p2 += 8;
            if (dl < 0 || dl > len - p2) { rc = fail(WO_ERR_EOF, "Eof"); goto bad2; }
            p2 += dl;
            if ((dl & 1) == 1) {
                if (p2 >= len) { rc = fail(WO_ERR_EOF, "Eof"); goto bad2; }
                p2++; t2 -= 1;
            }
            t2 -= dl + 8;
        }
    }
    while (totalLen > 0) {
        if (len - pos < 8) { rc = fail(WO_ERR_EOF, "Eof"); goto bad2; }
        const uint8_t* id = data + pos;
        int32_t dataLen = (int32_t)rd32(data + pos + 4);
        pos += 8;
        if (dataLen < 0 || dataLen > len - pos) { rc = fail(WO_ERR_EOF, "Eof"); goto bad2; }
        const uint8_t* cd = data + pos;
        pos += dataLen;
        if ((dataLen & 1) == 1) {
            if (pos >= len) { rc = fail(WO_ERR_EOF, "Eof"); goto bad2; }
            pos++; totalLen -= 1;
        }
        totalLen -= dataLen + 8;
        if (!memcmp(id, "VP8X", 4)) {                    /* readVp8x :133-141 */
            if (dataLen < 10) { rc = fail(WO_ERR_EOF, "Eof"); goto bad2; }
            widthMinusOne = rd24(cd + 4); heightMinusOne = rd24(cd + 7);
        } else if (!memcmp(id, "ALPH", 4)) {             /* readAlph :78-88 */
            if (dataLen < 1) { rc = fail(WO_ERR_EOF, "Eof"); goto bad2; }
            int flags = cd[0];
            uint8_t* a = NULL; int al = 0;
            rc = readAlpha(cd + 1, dataLen - 1, widthMinusOne, heightMinusOne, flags & 0x03, &a, &al);
            if (rc) goto bad2;
            wo_unfilter_alpha(a, al, widthMinusOne + 1, (flags >> 2) & 0x03);
            free(alpha); alpha = a; alen = al;
        } else if (!memcmp(id, "VP8 ", 4)) {
            rc = readVp8(cd, dataLen, &lossy);
            if (rc) goto bad2;
            have_lossy = 1;
        } else if (!memcmp(id, "VP8L", 4)) {
            free(argb); argb = NULL;
            rc = readVp8l(cd, dataLen, &argb, &lw, &lh);
            if (rc) goto bad2;
            have_lossless = 1;
        } else if (!memcmp(id, "ANIM", 4)) {
            if (dataLen < 6) { rc = fail(WO_ERR_EOF, "Eof"); goto bad2; }
        } else if (!memcmp(id, "ANMF", 4)) {             /* readAnmf :150-177 */
            if (dataLen < 16) { rc = fail(WO_ERR_EOF, "Eof"); goto bad2; }
            wo_image* nf = (wo_image*)realloc(frames, sizeof(wo_image) * (nframes + 1));
            if (!nf) { rc = fail(WO_ERR_ALLOC, "alloc"); goto bad2; }
            frames = nf;
            wo_image* fr = &frames[nframes];
            memset(fr, 0, sizeof *fr);
            int fwm1 = rd24(cd + 6), fhm1 = rd24(cd + 9);
            int flags = cd[15];
            rc = readFrameData(cd, dataLen, 16, fwm1, fhm1, fr);
            if (rc) { img_free_fields(fr); goto bad2; }
            /* the frame's own width/height are the AnimFrameHeader's */
            fr->width = fwm1 + 1; fr->height = fhm1 + 1;
            fr->fx = rd24(cd); fr->fy = rd24(cd + 3); fr->duration = rd24(cd + 12);
            fr->dispose = (flags & 0x01) != 0; fr->blend = (flags & 0x02) == 0;
            nframes++;
        }
    }
    if (nframes > 0) {
        im->kind = WO_ANIM; im->frames = frames; im->nframes = nframes; frames = NULL;
        im->width = widthMinusOne + 1; im->height = heightMinusOne + 1; im->keyFrame = 1; im->showFrame = 1; im->has_header = 1;
        free(alpha); free(argb); img_free_fields(&lossy);
        *out = im; return WO_OK;
    }
    if (have_lossless) {
        im->kind = WO_ARGB; im->argb = argb; im->stride = 4 * lw; im->argblen = 4 * lw * lh;
        im->width = lw; im->height = lh; im->keyFrame = 1; im->showFrame = 1; im->has_header = 1;
        free(alpha); img_free_fields(&lossy);
        *out = im; return WO_OK;
    }
    if (have_lossy) {
        *im = lossy; im->kind = WO_YUV420; im->a = alpha; im->alen = alen;
        *out = im; return WO_OK;
    }
    rc = fail(WO_ERR_FORMAT, "Invalid format: no VP8/VP8L chunk found");
bad2:
    free(alpha); free(argb); img_free_fields(&lossy);
    for (int i = 0; i < nframes; i++) img_free_fields(&frames[i]);
    free(frames);
bad:
    free(im);
    return rc;
}

/* Tools.toArgb for one Yuv420/Argb image or frame (Tools.hx:6-48 plus test/Visual.hx:45-89 for frames).
 * out must hold 4*width*height bytes. */
int wo_to_argb(const wo_image* im, uint8_t* out) {
    if (im->kind == WO_ARGB) {
        for (int row = 0; row < im->height; row++) memcpy(out + (size_t)row * im->width * 4, im->argb + (size_t)row * im->stride, (size_t)im->width * 4);
        return WO_OK;
    }
    if (im->kind == WO_YUV420) {
        wo_yuv420_to_argb(im->y, im->ystride, im->u, im->v, im->uvstride, im->a, im->width, im->height, out);
        return WO_OK;
    }
    return fail(WO_ERR_FORMAT, "Nested Anim not supported");
}

/* ---- accessors for ctypes ---- */
int wo_kind(const wo_image* im) { return im->kind; }
int wo_width(const wo_image* im) { return im->width; }
int wo_height(const wo_image* im) { return im->height; }
int wo_nframes(const wo_image* im) { return im->nframes; }
const wo_image* wo_frame(const wo_image* im, int i) { return &im->frames[i]; }
const uint8_t* wo_plane(const wo_image* im, int which) {
    switch (which) { case 0: return im->y; case 1: return im->u; case 2: return im->v; case 3: return im->a;
                     case 4: return im->argb; case 5: return im->y0; case 6: return im->u0; case 7: return im->v0; }
    return NULL;
}
int wo_info(const wo_image* im, int which) {
    switch (which) { case 0: return im->ystride; case 1: return im->uvstride; case 2: return im->mbw; case 3: return im->mbh;
                     case 4: return im->stride; case 5: return im->alen; case 6: return im->argblen;
                     case 7: return im->fx; case 8: return im->fy; case 9: return im->duration; case 10: return im->blend; case 11: return im->dispose; }
    return 0;
}

/* ---- Canvas compositing.  The reference parses AnimFrameHeader (Image.hx:21-34, WebPDecoder.hx:150-177)
 * and does no compositing, so this follows the WebP container specification ("Assembling the Canvas from Frames") in the
 * integer arithmetic of libwebp's WebPAnimDecoder (src/demux/anim_decode.c: IsKeyFrame, ZeroFillFrameRect,
 * BlendPixelNonPremult, FindBlendRangeAtRow), restated frame by frame on a whole canvas the way that decoder works.
 * Pinned by tests/golden/anim/ (canvases produced by that decoder through PIL).  Pixels: A,R,G,B bytes.
 * frames: n x {x, y, width, height, blend, dispose} (x, y in units of 2 pixels); pix[k]: frame k's pixels, tight. */
static uint32_t wo_blend_non_premult(const uint8_t* s, const uint8_t* d) {        /* returns A | R<<8 | G<<16 | B<<24 */
    uint32_t src_a = s[0], dst_a = d[0];
    uint32_t dst_factor_a = (dst_a * (256 - src_a)) >> 8;
    uint32_t blend_a = src_a + dst_factor_a;
    uint32_t scale = (1u << 24) / blend_a;
    uint32_t out = blend_a;
    for (int c = 1; c < 4; c++) out |= (((uint32_t)s[c] * src_a + (uint32_t)d[c] * dst_factor_a) * scale >> 24) << (8 * c);
    return out;
}
int wo_anim_compose(int n, const int32_t* frames, int cw, int ch, const uint8_t* const* pix, uint8_t* canvases) {
    size_t csz = (size_t)4 * cw * ch;
    uint8_t* prev = (uint8_t*)calloc(csz ? csz : 1, 1);         /* previous canvas with the previous frame's disposal applied */
    if (!prev) return fail(WO_ERR_ALLOC, "alloc");
    int prev_key = 0, prev_full = 0, prev_dispose = 0, px0 = 0, py0 = 0, pw = 0, ph = 0;
    for (int k = 0; k < n; k++) {
        const int32_t* f = frames + 6 * k;
        int x0 = 2 * f[0], y0 = 2 * f[1], w = f[2], h = f[3], blend = f[4], dispose = f[5];
        int full = x0 == 0 && y0 == 0 && w == cw && h == ch;
        int key = k == 0 || (!blend && full) || (prev_dispose && (prev_full || prev_key));
        uint8_t* cur = canvases + (size_t)k * csz;
        if (key) memset(prev, 0, csz);
        else if (prev_dispose)
            for (int y = py0; y < py0 + ph && y < ch; y++)
                for (int x = px0; x < px0 + pw && x < cw; x++) memset(prev + 4 * ((size_t)y * cw + x), 0, 4);
        memcpy(cur, prev, csz);
        for (int y = 0; y < h; y++) {
            int cy = y0 + y;
            if (cy >= ch) break;
            for (int x = 0; x < w; x++) {
                int cx = x0 + x;
                if (cx >= cw) break;
                const uint8_t* s = pix[k] + 4 * ((size_t)y * w + x);
                uint8_t* d = cur + 4 * ((size_t)cy * cw + cx);
                int in_prev_rect = prev_dispose && cx >= px0 && cx < px0 + pw && cy >= py0 && cy < py0 + ph;
                if (blend && !key && !in_prev_rect && s[0] != 0xff) {
                    if (s[0] != 0) { uint32_t v = wo_blend_non_premult(s, d); d[0] = v & 0xff; d[1] = (v >> 8) & 0xff; d[2] = (v >> 16) & 0xff; d[3] = v >> 24; }
                    /* alpha 0: the canvas pixel stays */
                } else {
                    memcpy(d, s, 4);
                }
            }
        }
        memcpy(prev, cur, csz);
        prev_key = key; prev_full = full; prev_dispose = dispose != 0; px0 = x0; py0 = y0; pw = w; ph = h;
    }
    free(prev);
    return 0;
}
