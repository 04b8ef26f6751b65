// Classical (non-neural) image stages either side of the ARU-Net, as HBM-bound byte / bit kernels for gfx950.
//
//   a1   scale_image + BGR2GRAY/255           net_post_processing_helper.py:14-33
//   a9   apply_cc_analysis                    region_net_post_processor_base.py:230-251
//        post_process (openings, subtract)    separator_net_post_processor.py:26-97
//   a12  StrokeWidthDistanceTransform         swt_dist_trafo.py:18-29
//
// Binary images are kept as bit planes (one uint64 per 64 pixels of a row, LSB = smallest x): a 3000x4500 mask
// is 1.7 MB, so rectangular morphology is word-parallel AND/OR of shifted words out of L2 instead of byte traffic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace asep {

// --------------------------------------------------------------------------------------------------------------
// bit planes
// --------------------------------------------------------------------------------------------------------------
// grid (WW, ceil(H/4)), block 256: one wave packs 64 pixels of one row.
__global__ void __launch_bounds__(256)
post_pack_kernel(const uint8_t* __restrict__ in, int H, int W, int stride, int ch, uint64_t* __restrict__ bits,
                 int WW) {
    const int lane = threadIdx.x & 63;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= H) return;
    const int x = blockIdx.x * 64 + lane;
    const bool fg = x < W && in[((size_t)y * W + x) * stride + ch] != 0;
    const uint64_t m = __ballot(fg);
    if (lane == 0) bits[(size_t)y * WW + blockIdx.x] = m;
}

__global__ void __launch_bounds__(256)
post_unpack_kernel(const uint64_t* __restrict__ bits, int H, int W, int WW, uint8_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // one thread = 4 pixels
    const int W4 = (W + 3) >> 2;
    if (i >= (size_t)H * W4) return;
    const int y = (int)(i / W4), x = (int)(i % W4) * 4;
    const uint64_t w = bits[(size_t)y * WW + (x >> 6)] >> (x & 63);
    uint8_t* o = out + (size_t)y * W + x;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (x + k < W) o[k] = ((w >> k) & 1) ? 255 : 0;
}

// bit i of the result = source bit (64*wx + i + o); `fill` is the value of bits outside [0, W).
__device__ __forceinline__ uint64_t post_row_word(const uint64_t* __restrict__ row, int WW, int W, int q,
                                                  uint64_t fill) {
    if (q < 0 || q >= WW) return fill;
    uint64_t w = row[q];
    const int tail = W - q * 64;                                  // valid bits in this word
    if (tail < 64) {
        const uint64_t valid = (tail <= 0) ? 0ull : ((1ull << tail) - 1ull);
        w = (w & valid) | (fill & ~valid);
    }
    return w;
}

__device__ __forceinline__ uint64_t post_shifted(const uint64_t* __restrict__ row, int WW, int W, int wx, int o,
                                                 uint64_t fill) {
    const int p = wx * 64 + o;                                    // first source bit
    const int q = p >> 6;                                         // floor division (arithmetic shift)
    const int r = p & 63;
    const uint64_t lo = post_row_word(row, WW, W, q, fill);
    if (r == 0) return lo;
    const uint64_t hi = post_row_word(row, WW, W, q + 1, fill);
    return (lo >> r) | (hi << (64 - r));
}

// cv window along x: offsets [-a, b], a = k/2, b = k-1-a.  ERODE: AND with +inf border; DILATE: OR with -inf border.
template <bool ERODE>
__global__ void __launch_bounds__(256)
post_morph_h_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int H, int W, int WW, int a,
                    int b) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)H * WW) return;
    const int y = (int)(i / WW), wx = (int)(i % WW);
    const uint64_t* row = in + (size_t)y * WW;
    const uint64_t fill = ERODE ? ~0ull : 0ull;
    uint64_t acc = fill;
    for (int o = -a; o <= b; ++o) {
        const uint64_t s = post_shifted(row, WW, W, wx, o, fill);
        acc = ERODE ? (acc & s) : (acc | s);
    }
    const int tail = W - wx * 64;
    if (tail < 64) acc &= (1ull << tail) - 1ull;
    out[i] = acc;
}

template <bool ERODE>
__global__ void __launch_bounds__(256)
post_morph_v_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int H, int W, int WW, int a,
                    int b) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)H * WW) return;
    const int y = (int)(i / WW), wx = (int)(i % WW);
    const int y0 = max(y - a, 0), y1 = min(y + b, H - 1);
    uint64_t acc = ERODE ? ~0ull : 0ull;
    for (int yy = y0; yy <= y1; ++yy) {
        const uint64_t s = in[(size_t)yy * WW + wx];
        acc = ERODE ? (acc & s) : (acc | s);
    }
    const int tail = W - wx * 64;
    if (tail < 64) acc &= (1ull << tail) - 1ull;
    out[i] = acc;
}

// cv2.subtract on 0/255 masks == a AND NOT b.
__global__ void __launch_bounds__(256)
post_andnot_kernel(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, uint64_t* __restrict__ out,
                   size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] & ~b[i];
}

// --------------------------------------------------------------------------------------------------------------
// 8-connected components: lock-free union-find on pixel indices (label = smallest index of the component).
//   init     every pixel starts on the first pixel of its horizontal run inside a 64-pixel row segment (one ballot)
//   union    only run contacts are linked: the segment's first pixel with its left neighbour, and per run of the
//            row above the left-most pixel that touches it (8-neighbourhood) -- O(#runs) unions, not O(#pixels)
//   find     path halving with atomicMin stores (labels only ever decrease, so concurrent writers cannot lose links)
//   area     per 64x16 tile the runs are accumulated in an LDS hash (root -> pixels), one global atomic per
//            (tile, component) instead of one per run
// --------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
cc_init_kernel(const uint8_t* __restrict__ in, int stride, int ch, int H, int W, int32_t* __restrict__ L) {
    const int lane = threadIdx.x & 63;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= H) return;
    const int x = blockIdx.x * 64 + lane;
    const bool fg = x < W && in[((size_t)y * W + x) * stride + ch] != 0;
    const uint64_t m = __ballot(fg);
    if (x >= W) return;
    int32_t lab = -1;
    if (fg) {
        const uint64_t below = ~m & ((1ull << lane) - 1ull);     // background lanes to my left
        const int start = below ? 64 - __clzll((long long)below) : 0;
        lab = (int32_t)((size_t)y * W + blockIdx.x * 64 + start);
    }
    L[(size_t)y * W + x] = lab;
}

__device__ __forceinline__ int32_t cc_find(int32_t* L, int32_t i) {
    for (;;) {
        const int32_t p = __atomic_load_n(&L[i], __ATOMIC_RELAXED);
        if (p == i) return i;
        const int32_t gp = __atomic_load_n(&L[p], __ATOMIC_RELAXED);
        if (gp == p) return p;
        atomicMin(&L[i], gp);                                     // halve the path (monotone, race-free)
        i = gp;
    }
}

__device__ __forceinline__ void cc_union(int32_t* L, int32_t a, int32_t b) {
    for (;;) {
        a = cc_find(L, a);
        b = cc_find(L, b);
        if (a == b) return;
        if (a > b) {
            const int32_t t = a;
            a = b;
            b = t;
        }
        const int32_t old = atomicMin(&L[b], a);                // b was a root when we looked
        if (old == b) return;
        b = old;                                                  // somebody re-parented b meanwhile: link that too
    }
}

__global__ void __launch_bounds__(256)
cc_union_kernel(int32_t* L, int H, int W) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)H * W) return;
    if (__atomic_load_n(&L[i], __ATOMIC_RELAXED) < 0) return;
    const int y = (int)(i / W), x = (int)(i % W);
    const int32_t me = (int32_t)i;
    const bool left = x > 0 && __atomic_load_n(&L[i - 1], __ATOMIC_RELAXED) >= 0;
    if (left && (x & 63) == 0) cc_union(L, me, me - 1);          // runs are pre-joined inside a 64-pixel segment
    if (y > 0) {
        const int32_t up = me - W;
        const bool n = __atomic_load_n(&L[up], __ATOMIC_RELAXED) >= 0;
        const bool ne = x + 1 < W && __atomic_load_n(&L[up + 1], __ATOMIC_RELAXED) >= 0;
        if (left) {
            // my left neighbour already sees the upper pixels x-1 and x; only a run starting at x+1 is new
            if (!n && ne) cc_union(L, me, up + 1);
        } else if (n) {
            cc_union(L, me, up);                                  // nw / ne belong to the same upper run
        } else {
            if (x > 0 && __atomic_load_n(&L[up - 1], __ATOMIC_RELAXED) >= 0) cc_union(L, me, up - 1);
            if (ne) cc_union(L, me, up + 1);
        }
    }
}

// flatten every pixel onto its root and zero the per-root area counters
__global__ void __launch_bounds__(256)
cc_flatten_kernel(int32_t* L, int32_t* __restrict__ area, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (__atomic_load_n(&L[i], __ATOMIC_RELAXED) < 0) return;
    const int32_t r = cc_find(L, (int32_t)i);
    L[i] = r;
    if (r == (int32_t)i) area[i] = 0;
}

// grid (WW, ceil(H/16)), block 256: wave w handles rows 4w..4w+3 of a 64 x 16 tile
#define CC_HASH 128
__global__ void __launch_bounds__(256)
cc_area_kernel(const int32_t* __restrict__ L, int32_t* __restrict__ area, int H, int W) {
    __shared__ int32_t keys[CC_HASH];
    __shared__ int32_t vals[CC_HASH];
    if (threadIdx.x < CC_HASH) {
        keys[threadIdx.x] = -1;
        vals[threadIdx.x] = 0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane;
    for (int k = 0; k < 4; ++k) {
        const int y = blockIdx.y * 16 + wave * 4 + k;
        const int32_t r = (y < H && x < W) ? L[(size_t)y * W + x] : -1;
        const int32_t prev = __shfl_up(r, 1);
        const bool brk = lane == 0 || prev != r;
        const uint64_t brks = __ballot(brk);
        if (brk && r >= 0) {
            const uint64_t above = lane == 63 ? 0ull : (brks >> (lane + 1));
            const int len = above ? __ffsll((long long)above) : 64 - lane;
            unsigned slot = ((unsigned)r * 2654435761u) >> 25;   // 7 bits
            bool done = false;
            for (int probe = 0; probe < CC_HASH && !done; ++probe) {
                const int32_t old = atomicCAS(&keys[slot], -1, r);
                if (old == -1 || old == r) {
                    atomicAdd(&vals[slot], len);
                    done = true;
                }
                slot = (slot + 1) & (CC_HASH - 1);
            }
            if (!done) atomicAdd(&area[r], len);                  // table full (very noisy tile)
        }
    }
    __syncthreads();
    if (threadIdx.x < CC_HASH && keys[threadIdx.x] >= 0) atomicAdd(&area[keys[threadIdx.x]], vals[threadIdx.x]);
}

// grid (WW, ceil(H/4)): keep pixels of components with area >= min_size; emit the bit plane (and optional u8).
__global__ void __launch_bounds__(256)
cc_filter_kernel(const int32_t* __restrict__ L, const int32_t* __restrict__ area, int H, int W, int min_size,
                 uint64_t* __restrict__ bits, int WW, uint8_t* __restrict__ out_u8) {
    const int lane = threadIdx.x & 63;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= H) return;
    const int x = blockIdx.x * 64 + lane;
    bool keep = false;
    if (x < W) {
        const int32_t r = L[(size_t)y * W + x];
        keep = r >= 0 && area[r] >= min_size;
        if (out_u8) out_u8[(size_t)y * W + x] = keep ? 255 : 0;
    }
    const uint64_t m = __ballot(keep);
    if (lane == 0) bits[(size_t)y * WW + blockIdx.x] = m;
}

// --------------------------------------------------------------------------------------------------------------
// a1: resize + gray
// --------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint8_t sat_u8_rn(float v) {
    const int r = __float2int_rn(v);                              // round half to even (cvRound)
    return (uint8_t)min(max(r, 0), 255);
}

// INTER_AREA, integer scale s (ResizeAreaFast): full blocks sum*float(1/(s*s)) (2x2: (sum+2)>>2), ragged border
// blocks sum/count.
__global__ void __launch_bounds__(256)
prep_area_int_kernel(const uint8_t* __restrict__ src, int H, int W, int C, int s, uint8_t* __restrict__ dst, int dh,
                     int dw) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)dh * dw * C) return;
    const int c = (int)(i % C);
    const int dx = (int)((i / C) % dw), dy = (int)(i / ((size_t)C * dw));
    const int y0 = dy * s, y1 = min(y0 + s, H), x0 = dx * s, x1 = min(x0 + s, W);
    int sum = 0;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) sum += src[((size_t)y * W + x) * C + c];
    uint8_t v;
    if (y1 - y0 == s && x1 - x0 == s) {
        if (s == 2) v = (uint8_t)((sum + 2) >> 2);
        else v = sat_u8_rn(__fmul_rn((float)sum, 1.0f / (float)(s * s)));
    } else {
        const int cnt = max((y1 - y0) * (x1 - x0), 1);
        v = sat_u8_rn(__fdiv_rn((float)sum, (float)cnt));
    }
    dst[i] = v;
}

// INTER_AREA, fractional scale: per-axis tables (start offsets per destination index, source index, weight).
__global__ void __launch_bounds__(256)
prep_area_tab_kernel(const uint8_t* __restrict__ src, int H, int W, int C, uint8_t* __restrict__ dst, int dh, int dw,
                     const int32_t* __restrict__ xofs, const int32_t* __restrict__ xsrc,
                     const float* __restrict__ xw, const int32_t* __restrict__ yofs,
                     const int32_t* __restrict__ ysrc, const float* __restrict__ yw) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)dh * dw * C) return;
    const int c = (int)(i % C);
    const int dx = (int)((i / C) % dw), dy = (int)(i / ((size_t)C * dw));
    const int xb = xofs[dx], xe = xofs[dx + 1];
    float sum = 0.f;
    bool first = true;
    for (int j = yofs[dy]; j < yofs[dy + 1]; ++j) {
        const uint8_t* row = src + (size_t)ysrc[j] * W * C + c;
        float buf = 0.f;
        for (int k = xb; k < xe; ++k) buf = __fadd_rn(buf, __fmul_rn((float)row[(size_t)xsrc[k] * C], xw[k]));
        const float t = __fmul_rn(yw[j], buf);
        sum = first ? t : __fadd_rn(sum, t);
        first = false;
    }
    dst[i] = sat_u8_rn(sum);
}

// INTER_CUBIC (a = -0.75) with OpenCV's 11-bit fixed-point taps; tables hold the floor index and 4 taps.
__global__ void __launch_bounds__(256)
prep_cubic_kernel(const uint8_t* __restrict__ src, int H, int W, int C, uint8_t* __restrict__ dst, int dh, int dw,
                  const int32_t* __restrict__ xi, const int16_t* __restrict__ xw, const int32_t* __restrict__ yi,
                  const int16_t* __restrict__ yw) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)dh * dw * C) return;
    const int c = (int)(i % C);
    const int dx = (int)((i / C) % dw), dy = (int)(i / ((size_t)C * dw));
    const int sx = xi[dx], sy = yi[dy];
    long long acc = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int yy = min(max(sy - 1 + j, 0), H - 1);
        const uint8_t* row = src + (size_t)yy * W * C + c;
        int h = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int xx = min(max(sx - 1 + k, 0), W - 1);
            h += (int)row[(size_t)xx * C] * (int)xw[dx * 4 + k];
        }
        acc += (long long)h * (long long)yw[dy * 4 + j];
    }
    const long long v = (acc + (1ll << 21)) >> 22;
    dst[i] = (uint8_t)min(max(v, 0ll), 255ll);
}

// BGR2GRAY (OpenCV 4.x 15-bit fixed point) and /255.0 in double like the reference, then the feed's float32 cast.
__global__ void __launch_bounds__(256)
prep_gray_kernel(const uint8_t* __restrict__ img, size_t n, int C, float* __restrict__ gray,
                 uint8_t* __restrict__ gray_u8) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int g;
    if (C == 3) {
        const uint8_t* p = img + i * 3;
        g = ((int)p[0] * 3735 + (int)p[1] * 19235 + (int)p[2] * 9798 + 16384) >> 15;
    } else {
        g = img[i];
    }
    if (gray) gray[i] = (float)((double)g / 255.0);
    if (gray_u8) gray_u8[i] = (uint8_t)g;
}

// --------------------------------------------------------------------------------------------------------------
// a12: 255-gray -> Gaussian 5x5 -> Otsu -> exact Euclidean distance transform -> uint8
// --------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

// inverted input, taps [1,4,6,4,1]^2, (sum + 128) >> 8; also accumulates the 256-bin histogram of the result.
// grid (ceil(W/64), ceil(H/16)), block 256: 64 x 16 output tile, (64+4) x (16+4) inverted input tile in LDS,
// horizontal pass into LDS, vertical pass from LDS.
__global__ void __launch_bounds__(256)
swt_blur_hist_kernel(const uint8_t* __restrict__ gray, int H, int W, uint8_t* __restrict__ blur,
                     unsigned int* __restrict__ hist) {
    __shared__ unsigned int lh[256];
    __shared__ uint8_t tin[20][72];
    __shared__ uint16_t th[20][64];
    const int tid = threadIdx.x;
    lh[tid] = 0;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 16;
    for (int i = tid; i < 20 * 68; i += 256) {
        const int r = i / 68, c = i % 68;
        tin[r][c] = (uint8_t)(255 - gray[(size_t)reflect101(y0 + r - 2, H) * W + reflect101(x0 + c - 2, W)]);
    }
    __syncthreads();
    for (int i = tid; i < 20 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        th[r][c] = (uint16_t)(tin[r][c] + 4 * tin[r][c + 1] + 6 * tin[r][c + 2] + 4 * tin[r][c + 3] + tin[r][c + 4]);
    }
    __syncthreads();
    for (int i = tid; i < 16 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        const int y = y0 + r, x = x0 + c;
        if (y < H && x < W) {
            const int acc = th[r][c] + 4 * th[r + 1][c] + 6 * th[r + 2][c] + 4 * th[r + 3][c] + th[r + 4][c];
            const int v = (acc + 128) >> 8;
            blur[(size_t)y * W + x] = (uint8_t)v;
            atomicAdd(&lh[v], 1u);
        }
    }
    __syncthreads();
    if (lh[tid]) atomicAdd(&hist[tid], lh[tid]);
}

// getThreshVal_Otsu_8u in double, one thread; no fused multiply-adds so the host restatement matches bit for bit.
__global__ void swt_otsu_kernel(const unsigned int* __restrict__ hist, int* __restrict__ thr_out) {
    if (threadIdx.x || blockIdx.x) return;
    double n = 0;
    for (int i = 0; i < 256; ++i) n += (double)hist[i];
    const double scale = 1.0 / n;
    double mu = 0;
    for (int i = 0; i < 256; ++i) mu = __dadd_rn(mu, __dmul_rn((double)i, (double)hist[i]));
    mu = __dmul_rn(mu, scale);
    double mu1 = 0, q1 = 0, max_sigma = 0;
    int max_val = 0;
    const double eps = 1.1920928955078125e-07;
    for (int i = 0; i < 256; ++i) {
        const double p_i = __dmul_rn((double)hist[i], scale);
        mu1 = __dmul_rn(mu1, q1);
        q1 = __dadd_rn(q1, p_i);
        const double q2 = __dsub_rn(1.0, q1);
        if (fmin(q1, q2) < eps || fmax(q1, q2) > 1.0 - eps) continue;
        mu1 = __ddiv_rn(__dadd_rn(mu1, __dmul_rn((double)i, p_i)), q1);
        const double mu2 = __ddiv_rn(__dsub_rn(mu, __dmul_rn(q1, mu1)), q2);
        const double d = __dsub_rn(mu1, mu2);
        const double sigma = __dmul_rn(__dmul_rn(__dmul_rn(q1, q2), d), d);
        if (sigma > max_sigma) {
            max_sigma = sigma;
            max_val = i;
        }
    }
    *thr_out = max_val;
}

// vertical pass in two kernels so that a page-high column is not one serial walk:
//   swt_edt_seg_kernel   per (column, 32-row segment): first / last zero row inside the segment (-1 = none)
//   swt_edt_cols_kernel  per (column, segment): nearest zero above / below from the segment table, then
//                        g[y][x] = min(y - zero_above, zero_below - y), capped at SWT_INF (= no zero in the column)
#define SWT_INF 0x3fff
#define SWT_SEG 32
__global__ void __launch_bounds__(256)
swt_edt_seg_kernel(const uint8_t* __restrict__ blur, const int* __restrict__ thr, int H, int W,
                   int32_t* __restrict__ seg_first, int32_t* __restrict__ seg_last) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int sg = blockIdx.y;
    if (x >= W) return;
    const int t = *thr;
    const int y0 = sg * SWT_SEG, y1 = min(y0 + SWT_SEG, H);
    int first = -1, last = -1;
    for (int y = y0; y < y1; ++y) {
        if ((int)blur[(size_t)y * W + x] <= t) {
            if (first < 0) first = y;
            last = y;
        }
    }
    seg_first[(size_t)sg * W + x] = first;
    seg_last[(size_t)sg * W + x] = last;
}

__global__ void __launch_bounds__(256)
swt_edt_cols_kernel(const uint8_t* __restrict__ blur, const int* __restrict__ thr, int H, int W, int nseg,
                    const int32_t* __restrict__ seg_first, const int32_t* __restrict__ seg_last,
                    uint16_t* __restrict__ g) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int sg = blockIdx.y;
    if (x >= W) return;
    const int t = *thr;
    const int y0 = sg * SWT_SEG, y1 = min(y0 + SWT_SEG, H);
    int above = -(1 << 20), below = 1 << 20;                      // nearest zero rows outside the segment
    for (int s = sg - 1; s >= 0; --s) {
        const int v = seg_last[(size_t)s * W + x];
        if (v >= 0) { above = v; break; }
    }
    for (int s = sg + 1; s < nseg; ++s) {
        const int v = seg_first[(size_t)s * W + x];
        if (v >= 0) { below = v; break; }
    }
    int dn[SWT_SEG];
    int z = above;
#pragma unroll
    for (int r = 0; r < SWT_SEG; ++r) {
        const int y = y0 + r;
        if (y < y1) {
            if ((int)blur[(size_t)y * W + x] <= t) z = y;
            dn[r] = y - z;
        } else {
            dn[r] = 0;
        }
    }
    z = below;
#pragma unroll
    for (int r = SWT_SEG - 1; r >= 0; --r) {
        const int y = y0 + r;
        if (y < y1) {
            if (dn[r] == 0) z = y;
            const int d = min(min(dn[r], z - y), SWT_INF);
            g[(size_t)y * W + x] = (uint16_t)d;
        }
    }
}

// horizontal pass: d2 = min_x' (x-x')^2 + g(x')^2, searched outwards until (x-x')^2 >= best (exact);
// result trunc(sqrt(d2)) mod 256 like ``dist.astype(np.uint8)``.
__global__ void __launch_bounds__(256)
swt_edt_rows_kernel(const uint16_t* __restrict__ g, int H, int W, uint8_t* __restrict__ out,
                    int32_t* __restrict__ d2_out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)H * W) return;
    const int y = (int)(i / W), x = (int)(i % W);
    const uint16_t* row = g + (size_t)y * W;
    const int g0 = row[x];
    long long best = g0 >= SWT_INF ? (1ll << 40) : (long long)g0 * g0;
    if (best) {
        for (int dx = 1; (long long)dx * dx < best && (x - dx >= 0 || x + dx < W); ++dx) {
            const long long dd = (long long)dx * dx;
            if (x - dx >= 0) {
                const int gv = row[x - dx];
                if (gv < SWT_INF) best = min(best, dd + (long long)gv * gv);
            }
            if (x + dx < W) {
                const int gv = row[x + dx];
                if (gv < SWT_INF) best = min(best, dd + (long long)gv * gv);
            }
        }
    }
    const int32_t b32 = best > 0x7fffffffll ? 0x7fffffff : (int32_t)best;
    if (d2_out) d2_out[i] = b32;
    const float d = __fsqrt_rn((float)b32);
    out[i] = (uint8_t)(((int)d) & 255);
}


// --------------------------------------------------------------------------------------------------------------
// a10 (device half): boundary of the pixels equal to `value` as maximal straight segments.  A unit edge is the side
// of a foreground pixel that faces background (or the image border), directed so that the foreground lies to its
// right: heading 0 (+x) top side, 1 (+y) right side, 2 (-x) bottom side, 3 (-y) left side.  Collinear unit edges of
// one heading form a segment; only its first and last unit edge are emitted:
//   starts[] = ((vy * (W + 1) + vx) * 4 + heading) of the segment's start vertex
//   ends[]   = the same encoding of its end vertex
// in arbitrary order (the host sorts both per heading; segments of one heading on one grid line are disjoint, so
// the k-th start pairs with the k-th end).  counter[0] / counter[1] count starts / ends even beyond `capacity`.
// --------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void post_wave_append(int n, unsigned long long* counter, unsigned long long& first) {
    const int lane = threadIdx.x & 63;
    int incl = n;
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    const int total = __shfl(incl, 63);
    unsigned long long base = 0;
    if (lane == 63 && total > 0) base = atomicAdd(counter, (unsigned long long)total);
    base = __shfl(base, 63);
    first = base + (unsigned long long)(incl - n);
}

__global__ void __launch_bounds__(256)
post_boundary_segments_kernel(const uint8_t* __restrict__ mask, int H, int W, int value, int32_t* __restrict__ starts,
                              int32_t* __restrict__ ends, unsigned long long capacity,
                              unsigned long long* __restrict__ counter) {
    // a thread owns 16 consecutive pixels (one 16-byte load when the mask is aligned); a separator mask is almost empty, and a
    // wave whose 1024 pixels hold no foreground leaves at once -- only the others walk their pixels, all lanes together, because the
    // appends are wave-wide scans
    const size_t n = (size_t)H * W;
    const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    uint8_t px[16];
    bool any = false;
    if (i0 + 16 <= n && ((size_t)mask & 15) == 0) {
        const uint4 q = *reinterpret_cast<const uint4*>(mask + i0);
        const unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            px[k] = (uint8_t)(w4[k >> 2] >> (8 * (k & 3)));
            any |= px[k] == value;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            px[k] = (i0 + k < n && mask[i0 + k] == value) ? (uint8_t)value : (uint8_t)~value;
            any |= i0 + k < n && px[k] == value;
        }
    }
    if (__ballot(any) == 0) return;
    const int VW = W + 1;
    for (int k = 0; k < 16; ++k) {
        const size_t i = i0 + k;
        int sk[4], ek[4];
        int ns = 0, ne = 0;
        if (i < n && px[k] == value) {
            const int y = (int)(i / W), x = (int)(i % W);
            auto fg = [&](int yy, int xx) -> bool {
                return yy >= 0 && yy < H && xx >= 0 && xx < W && mask[(size_t)yy * W + xx] == value;
            };
            const bool nn = fg(y - 1, x), s = fg(y + 1, x), w = fg(y, x - 1), e = fg(y, x + 1);
            const bool nw = fg(y - 1, x - 1), nev = fg(y - 1, x + 1), sw = fg(y + 1, x - 1), se = fg(y + 1, x + 1);
            if (!nn) {                                                // top side, +x: (x, y) -> (x+1, y)
                if (!(w && !nw)) sk[ns++] = (y * VW + x) * 4 + 0;
                if (!(e && !nev)) ek[ne++] = (y * VW + x + 1) * 4 + 0;
            }
            if (!e) {                                                 // right side, +y: (x+1, y) -> (x+1, y+1)
                if (!(nn && !nev)) sk[ns++] = (y * VW + x + 1) * 4 + 1;
                if (!(s && !se)) ek[ne++] = ((y + 1) * VW + x + 1) * 4 + 1;
            }
            if (!s) {                                                 // bottom side, -x: (x+1, y+1) -> (x, y+1)
                if (!(e && !se)) sk[ns++] = ((y + 1) * VW + x + 1) * 4 + 2;
                if (!(w && !sw)) ek[ne++] = ((y + 1) * VW + x) * 4 + 2;
            }
            if (!w) {                                                 // left side, -y: (x, y+1) -> (x, y)
                if (!(s && !sw)) sk[ns++] = ((y + 1) * VW + x) * 4 + 3;
                if (!(nn && !nw)) ek[ne++] = (y * VW + x) * 4 + 3;
            }
        }
        if (__ballot(ns | ne) == 0) continue;                         // (wave-uniform: nobody has anything at this position)
        unsigned long long o;
        post_wave_append(ns, counter, o);
        for (int q = 0; q < ns; ++q, ++o)
            if (o < capacity) starts[o] = sk[q];
        post_wave_append(ne, counter + 1, o);
        for (int q = 0; q < ne; ++q, ++o)
            if (o < capacity) ends[o] = ek[q];
    }
}


// --------------------------------------------------------------------------------------------------------------
// a11: the heading pipeline's two other per-page reductions, so that neither the decoded scan nor the net output has to
// visit the host a second time:
//   post_gray_u8_kernel    cv2.imread(path, IMREAD_GRAYSCALE) of a decoded BGR image (swt_dist_trafo.py:19): OpenCV's
//                          fixed-point weights (B 3735 + G 19235 + R 9798 + 2^14) >> 15, four pixels per thread
//   post_box_sums_kernel   heading_net_post_processor.py:247-270: the sum of one channel of the uint8 net output over a
//                          text line's bounding box, as an exact integer (one workgroup per box, a wave per row)
// --------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
post_gray_u8_kernel(const uint8_t* __restrict__ bgr, size_t n, uint8_t* __restrict__ out) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;            // four pixels = 12 bytes in, 4 bytes out
    auto gray = [](unsigned b, unsigned g, unsigned r) -> unsigned { return (b * 3735u + g * 19235u + r * 9798u + 16384u) >> 15; };
    if (q * 4 + 4 <= n && (((size_t)bgr | (size_t)out) & 3) == 0) {
        const uint32_t* src = (const uint32_t*)(bgr + q * 12);
        const uint32_t a = src[0], b = src[1], c = src[2];
        const unsigned p0 = gray(a & 255, (a >> 8) & 255, (a >> 16) & 255);
        const unsigned p1 = gray(a >> 24, b & 255, (b >> 8) & 255);
        const unsigned p2 = gray((b >> 16) & 255, b >> 24, c & 255);
        const unsigned p3 = gray((c >> 8) & 255, (c >> 16) & 255, c >> 24);
        *(uint32_t*)(out + q * 4) = p0 | (p1 << 8) | (p2 << 16) | (p3 << 24);
    } else {
        for (size_t i = q * 4; i < n && i < q * 4 + 4; ++i) out[i] = (uint8_t)gray(bgr[3 * i], bgr[3 * i + 1], bgr[3 * i + 2]);
    }
}

struct PostBox { int x0, y0, x1, y1; };           // rows [y0, y1), columns [x0, x1), already clipped to the image

__global__ void __launch_bounds__(256)
post_box_sums_kernel(const uint8_t* __restrict__ img, int W, int pix_stride, int channel, const PostBox* __restrict__ boxes,
                     unsigned long long* __restrict__ sums) {
    __shared__ unsigned long long part[4];
    const PostBox b = boxes[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long s = 0;
    for (int y = b.y0 + wave; y < b.y1; y += 4) {
        const uint8_t* row = img + ((size_t)y * W) * pix_stride + channel;
        unsigned r = 0;                                                  // a row of at most 2^24 pixels fits 32 bits
        for (int x = b.x0 + lane; x < b.x1; x += 64) r += row[(size_t)x * pix_stride];
        s += r;
    }
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d);
    if (lane == 0) part[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}


// --------------------------------------------------------------------------------------------------------------
// a11 / f4: per text line statistics on the stroke-width image (heading_net_post_processor.py:218-245,
// feature_generation.py:106-159): inside the line's crop, 8-connected components of the non-zero pixels ->
// bounding boxes -> reject (w < 3 or h < 3 or w > 500 or h > 500, then w/h > 8 or h/w > 8) -> per accepted
// component the maximum of the crop over its bounding box -> median of those maxima and the largest height.
// One workgroup per line; labels live in a global scratch slice of the line, component boxes in LDS.
// --------------------------------------------------------------------------------------------------------------
#define SWTL_MAXC 1024
#define SWTL_LDS_PIXELS 16384                    // 64 KB of labels
struct SwtLineBox { int x0, y0, x1, y1; };      // crop = rows [y0, y1), columns [x0, x1), already clipped

__global__ void __launch_bounds__(256)
swt_line_features_kernel(const uint8_t* __restrict__ swt, int W, const SwtLineBox* __restrict__ boxes,
                         const unsigned long long* __restrict__ scratch_ofs, int32_t* scratch,
                         float* __restrict__ out_sw, int32_t* __restrict__ out_h, int32_t* __restrict__ out_flag) {
    __shared__ int ncomp;
    __shared__ int maxh;
    __shared__ int bx0[SWTL_MAXC], bx1[SWTL_MAXC], by0[SWTL_MAXC], by1[SWTL_MAXC];
    __shared__ unsigned int hist[256];
    const int line = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const SwtLineBox b = boxes[line];
    const int cw = b.x1 - b.x0, chh = b.y1 - b.y0;
    const int n = cw > 0 && chh > 0 ? cw * chh : 0;
    if (tid == 0) { ncomp = 0; maxh = 0; }
    hist[tid] = 0;
    if (n == 0) {
        if (tid == 0) { out_sw[line] = 0.f; out_h[line] = 0; out_flag[line] = 0; }
        return;
    }
    // the labels of a crop of up to SWTL_LDS_PIXELS live in LDS (the union-find's atomics and the three sweeps over the labels
    // are then on-chip: 1.6 -> 1.3 ms per page of 700 lines beside the next page's net, 0.36 ms alone); larger crops keep the
    // global scratch slice
    extern __shared__ __attribute__((aligned(16))) int32_t swtl_labels[];
    int32_t* L = n <= SWTL_LDS_PIXELS ? swtl_labels : scratch + scratch_ofs[line];
    int32_t* aux = scratch + scratch_ofs[line] + n;
    const uint8_t* src = swt + (size_t)b.y0 * W + b.x0;
    for (int p = tid; p < n; p += 256) {
        const int y = p / cw, x = p - y * cw;
        L[p] = src[(size_t)y * W + x] != 0 ? p : -1;
    }
    __syncthreads();
    for (int p = tid; p < n; p += 256) {
        if (__atomic_load_n(&L[p], __ATOMIC_RELAXED) < 0) continue;
        const int y = p / cw, x = p - y * cw;
        const bool left = x > 0 && __atomic_load_n(&L[p - 1], __ATOMIC_RELAXED) >= 0;
        if (left) cc_union(L, p, p - 1);
        if (y > 0) {
            const int up = p - cw;
            const bool nn = __atomic_load_n(&L[up], __ATOMIC_RELAXED) >= 0;
            const bool ne = x + 1 < cw && __atomic_load_n(&L[up + 1], __ATOMIC_RELAXED) >= 0;
            if (left) {
                if (!nn && ne) cc_union(L, p, up + 1);
            } else if (nn) {
                cc_union(L, p, up);
            } else {
                if (x > 0 && __atomic_load_n(&L[up - 1], __ATOMIC_RELAXED) >= 0) cc_union(L, p, up - 1);
                if (ne) cc_union(L, p, up + 1);
            }
        }
    }
    __syncthreads();
    for (int p = tid; p < n; p += 256) {
        if (__atomic_load_n(&L[p], __ATOMIC_RELAXED) < 0) continue;
        const int r = cc_find(L, p);
        L[p] = r;
        if (r == p) aux[p] = atomicAdd(&ncomp, 1);
    }
    __syncthreads();
    const int nc = ncomp;
    if (nc > SWTL_MAXC) {                       // pathological crop: the host falls back for this line
        if (tid == 0) { out_sw[line] = 0.f; out_h[line] = 0; out_flag[line] = 1; }
        return;
    }
    for (int c = tid; c < nc; c += 256) { bx0[c] = 1 << 30; by0[c] = 1 << 30; bx1[c] = -1; by1[c] = -1; }
    __syncthreads();
    for (int p = tid; p < n; p += 256) {
        const int r = L[p];
        if (r < 0) continue;
        const int c = aux[r];
        const int y = p / cw, x = p - y * cw;
        atomicMin(&bx0[c], x); atomicMax(&bx1[c], x);
        atomicMin(&by0[c], y); atomicMax(&by1[c], y);
    }
    __syncthreads();
    for (int c = wave; c < nc; c += 4) {
        const int w = bx1[c] - bx0[c] + 1, h = by1[c] - by0[c] + 1;
        if (w < 3 || h < 3 || h > 500 || w > 500) continue;           // swt_dist_trafo.py:53-56
        if (w > 8 * h || h > 8 * w) continue;                           // :57-60 (w/h > 8 <=> w > 8h for positive ints)
        int m = 0;
        for (int i = lane; i < w * h; i += 64) {
            const int yy = i / w, xx = i - yy * w;
            m = max(m, (int)src[(size_t)(by0[c] + yy) * W + bx0[c] + xx]);
        }
        for (int d = 32; d > 0; d >>= 1) m = max(m, __shfl_xor(m, d));
        if (lane == 0) {
            atomicAdd(&hist[m], 1u);
            atomicMax(&maxh, h);
        }
    }
    __syncthreads();
    if (tid == 0) {
        unsigned int total = 0;
        for (int v = 0; v < 256; ++v) total += hist[v];
        float med = 0.f;
        if (total > 0) {
            const unsigned int k1 = (total - 1) / 2, k2 = total / 2;   // np.median: mean of the two middle values
            unsigned int cum = 0;
            int v1 = -1, v2 = -1;
            for (int v = 0; v < 256; ++v) {
                cum += hist[v];
                if (v1 < 0 && cum > k1) v1 = v;
                if (v2 < 0 && cum > k2) { v2 = v; break; }
            }
            med = 0.5f * (float)(v1 + v2);
        }
        out_sw[line] = med;
        out_h[line] = maxh;
        out_flag[line] = 0;
    }
}

}  // namespace asep
