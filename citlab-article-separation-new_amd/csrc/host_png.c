/* libasep_host.so: PNG row un-filtering for the scan decode in front of the GPU path (include/asep_host.h).
 * Pillow spends more than half of a 3000 x 4500 scan's decode time in its byte-serial Paeth loop; here the common filters run
 * per pixel with the previous pixel in registers, Up and None are memcpy / vector adds. */
#include <stdlib.h>
#include <string.h>

#include "../../include/asep_host.h"

/* one Paeth step: left a, above b, upper left c, filtered byte x -> reconstructed byte (the predictor in
 * differences: p - a = b - c, p - b = a - c, p - c = their sum; ties a, then b, then c) */
static inline int paeth_step(int a, int b, int c, int x) {
    int pa = b - c, pb = a - c;
    int pc = pa + pb;
    int s;
    s = pa >> 31; pa = (pa ^ s) - s;                                    /* |.| and the two selections as masks: scan content makes */
    s = pb >> 31; pb = (pb ^ s) - s;                                    /* these comparisons unpredictable for a branch */
    s = pc >> 31; pc = (pc ^ s) - s;
    const int m1 = -(pb < pa);
    pa = (pb & m1) | (pa & ~m1);
    int pred = (b & m1) | (a & ~m1);
    const int m2 = -(pc < pa);
    pred = (c & m2) | (pred & ~m2);
    return (x + pred) & 0xff;
}

/* Four consecutive Paeth rows of one-byte pixels as a wavefront: row k runs k pixels behind row k - 1, so the four dependency
 * chains (each byte needs its left neighbour) are independent inside one iteration and overlap in the pipeline -- a single row
 * alone is a chain of ~10 dependent operations per byte.  up = the row above the first of the four. */
static void paeth4_rows(const uint8_t* const src[4], uint8_t* const dst[4], const uint8_t* up, long stride) {
    int a[4] = {0, 0, 0, 0}, c[4] = {0, 0, 0, 0};
    const uint8_t* above[4] = {up, dst[0], dst[1], dst[2]};
    for (long i = 0; i < stride + 3; ++i) {
        if (i >= 3 && i < stride) {                                      /* all four rows inside: no bounds tests */
            const int b0 = above[0][i], b1 = above[1][i - 1], b2 = above[2][i - 2], b3 = above[3][i - 3];
            const int v0 = paeth_step(a[0], b0, c[0], src[0][i]);
            const int v1 = paeth_step(a[1], b1, c[1], src[1][i - 1]);
            const int v2 = paeth_step(a[2], b2, c[2], src[2][i - 2]);
            const int v3 = paeth_step(a[3], b3, c[3], src[3][i - 3]);
            dst[0][i] = (uint8_t)v0; dst[1][i - 1] = (uint8_t)v1; dst[2][i - 2] = (uint8_t)v2; dst[3][i - 3] = (uint8_t)v3;
            a[0] = v0; a[1] = v1; a[2] = v2; a[3] = v3;
            c[0] = b0; c[1] = b1; c[2] = b2; c[3] = b3;
        } else {
            for (int k = 0; k < 4; ++k) {
                const long j = i - k;
                if (j < 0 || j >= stride) continue;
                const int b = above[k][j];
                const int v = paeth_step(a[k], b, c[k], src[k][j]);
                dst[k][j] = (uint8_t)v;
                a[k] = v; c[k] = b;
            }
        }
    }
}

long asep_png_unfilter(const uint8_t* filtered, long rows, long stride, int bpp, uint8_t* out) {
    if (!filtered || !out || rows < 0 || stride < 0 || bpp < 1 || bpp > 8) return -1;
    uint8_t* zero = (uint8_t*)calloc((size_t)stride + 1, 1);            /* the row above the first one */
    if (!zero) return -1;
    const uint8_t* up = zero;
    for (long y = 0; y < rows; ++y) {
        const uint8_t* src = filtered + (size_t)y * (stride + 1);
        uint8_t* dst = out + (size_t)y * stride;
        const int ft = src[0];
        if (ft == 4 && bpp == 1 && y + 3 < rows && src[stride + 1] == 4 && src[2 * (stride + 1)] == 4 && src[3 * (stride + 1)] == 4) {
            const uint8_t* s4[4];
            uint8_t* d4[4];
            for (int k = 0; k < 4; ++k) { s4[k] = src + (size_t)k * (stride + 1) + 1; d4[k] = dst + (size_t)k * stride; }
            paeth4_rows(s4, d4, up, stride);
            y += 3;
            up = d4[3];
            continue;
        }
        ++src;
        switch (ft) {
        case 0:
            memcpy(dst, src, (size_t)stride);
            break;
        case 1:                                                          /* Sub: + the pixel to the left */
            for (long i = 0; i < stride && i < bpp; ++i) dst[i] = src[i];
            for (long i = bpp; i < stride; ++i) dst[i] = (uint8_t)(src[i] + dst[i - bpp]);
            break;
        case 2:                                                          /* Up */
            for (long i = 0; i < stride; ++i) dst[i] = (uint8_t)(src[i] + up[i]);
            break;
        case 3:                                                          /* Average of left and up (integer floor) */
            for (long i = 0; i < stride && i < bpp; ++i) dst[i] = (uint8_t)(src[i] + (up[i] >> 1));
            for (long i = bpp; i < stride; ++i) dst[i] = (uint8_t)(src[i] + ((dst[i - bpp] + up[i]) >> 1));
            break;
        case 4:                                                          /* Paeth predictor of left, up, upper left */
            for (long i = 0; i < stride && i < bpp; ++i) dst[i] = (uint8_t)(src[i] + up[i]);
            if (bpp == 1) {
                int a = stride > 0 ? dst[0] : 0, c = stride > 0 ? up[0] : 0;
                for (long i = 1; i < stride; ++i) {
                    const int b = up[i];
                    a = paeth_step(a, b, c, src[i]);
                    dst[i] = (uint8_t)a;
                    c = b;
                }
            } else {
                for (long i = bpp; i < stride; ++i) dst[i] = (uint8_t)paeth_step(dst[i - bpp], up[i], up[i - bpp], src[i]);
            }
            break;
        default:
            free(zero);
            return -(y + 1);
        }
        up = dst;
    }
    free(zero);
    return 0;
}

void asep_rgb_to_bgr(const uint8_t* in, size_t n, uint8_t* out) {
    for (size_t i = 0; i < n; ++i) {
        const uint8_t r = in[3 * i], g = in[3 * i + 1], b = in[3 * i + 2];
        out[3 * i] = b; out[3 * i + 1] = g; out[3 * i + 2] = r;
    }
}
