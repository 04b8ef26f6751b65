// Host side of the classical image stages (include/asep_hip.h, "classical image stages" block).
// Compiled with -ffp-contract=off: the float / double sequences below restate OpenCV's scalar code paths and must
// not be fused into FMAs (the CPU restatement used by the parity tests evaluates them step by step).
#include <cmath>
#include <vector>

#include <mutex>

#include "asep_common.h"
#include "post_kernels.h"

using namespace asep;

namespace {

struct AreaTab {
    std::vector<int32_t> ofs, src;
    std::vector<float> w;
};

// computeResizeAreaTab (OpenCV resize.cpp), double arithmetic, float weights.
AreaTab build_area_tab(int ssize, int dsize, double scale) {
    AreaTab t;
    t.ofs.push_back(0);
    for (int dx = 0; dx < dsize; ++dx) {
        const double fsx1 = dx * scale;
        const double fsx2 = fsx1 + scale;
        const double cell = std::min(scale, ssize - fsx1);
        int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) {
            t.src.push_back(sx1 - 1);
            t.w.push_back((float)((sx1 - fsx1) / cell));
        }
        for (int sx = sx1; sx < sx2; ++sx) {
            t.src.push_back(sx);
            t.w.push_back((float)(1.0 / cell));
        }
        if (fsx2 - sx2 > 1e-3) {
            t.src.push_back(sx2);
            t.w.push_back((float)(std::min(std::min(fsx2 - sx2, 1.0), cell) / cell));
        }
        t.ofs.push_back((int32_t)t.src.size());
    }
    return t;
}

struct CubicTab {
    std::vector<int32_t> idx;
    std::vector<int16_t> w;
};

CubicTab build_cubic_tab(int dsize, double scale) {
    CubicTab t;
    t.idx.resize(dsize);
    t.w.resize((size_t)dsize * 4);
    for (int dx = 0; dx < dsize; ++dx) {
        float fx = (float)((dx + 0.5) * scale - 0.5);
        const int sx = (int)std::floor(fx);
        fx -= sx;
        const float A = -0.75f;
        float c[4];
        c[0] = ((A * (fx + 1) - 5 * A) * (fx + 1) + 8 * A) * (fx + 1) - 4 * A;
        c[1] = ((A + 2) * fx - (A + 3)) * fx * fx + 1;
        c[2] = ((A + 2) * (1 - fx) - (A + 3)) * (1 - fx) * (1 - fx) + 1;
        c[3] = 1.f - c[0] - c[1] - c[2];
        t.idx[dx] = sx;
        for (int k = 0; k < 4; ++k) {
            long v = lrintf(c[k] * 2048.0f);
            v = std::min(std::max(v, -32768l), 32767l);
            t.w[(size_t)dx * 4 + k] = (int16_t)v;
        }
    }
    return t;
}

template <typename T>
T* upload(const std::vector<T>& v) {
    T* d = nullptr;
    ASEP_HIP_CHECK_THROW(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
    if (!v.empty()) ASEP_HIP_CHECK_THROW(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

inline unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

struct asep_post {
    hipStream_t s = nullptr;
    BufferPool pool;
    // cached resize tables
    int tab_H = 0, tab_W = 0;
    double tab_sc = 0;
    int tab_kind = 0;                    // 1 area (fractional), 2 cubic
    void* d_tab[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    void free_tabs() {
        for (auto& p : d_tab) {
            if (p) (void)hipFree(p);
            p = nullptr;
        }
        tab_kind = 0;
    }
    ~asep_post() {
        free_tabs();
        pool.release();
        if (s) (void)hipStreamDestroy(s);
    }
};

namespace {

int cv_round(double v) { return (int)lrint(v); }

void scaled_size(int H, int W, double sc, int& h, int& w) {
    if (sc == 1.0) {
        h = H;
        w = W;
        return;
    }
    h = cv_round(H * sc);
    w = cv_round(W * sc);
}

// ---- 1-D passes on bit planes ---------------------------------------------------------------------------------
void morph_1d(hipStream_t st, bool erode, bool horizontal, const uint64_t* in, uint64_t* out, int H, int W, int WW,
              int k) {
    const int a = k / 2, b = k - 1 - a;
    const size_t n = (size_t)H * WW;
    if (horizontal) {
        if (erode) post_morph_h_kernel<true><<<blocks_for(n), 256, 0, st>>>(in, out, H, W, WW, a, b);
        else post_morph_h_kernel<false><<<blocks_for(n), 256, 0, st>>>(in, out, H, W, WW, a, b);
    } else {
        if (erode) post_morph_v_kernel<true><<<blocks_for(n), 256, 0, st>>>(in, out, H, W, WW, a, b);
        else post_morph_v_kernel<false><<<blocks_for(n), 256, 0, st>>>(in, out, H, W, WW, a, b);
    }
}

// erode or dilate with a kw x kh rectangle: cur -> (result pointer); tmp buffers ping-pong.
uint64_t* morph_rect(hipStream_t st, bool erode, uint64_t* cur, uint64_t* t0, uint64_t* t1, int H, int W, int WW,
                     int kw, int kh) {
    if (kw > 1) {
        uint64_t* dst = (cur == t0) ? t1 : t0;
        morph_1d(st, erode, true, cur, dst, H, W, WW, kw);
        cur = dst;
    }
    if (kh > 1) {
        uint64_t* dst = (cur == t0) ? t1 : t0;
        morph_1d(st, erode, false, cur, dst, H, W, WW, kh);
        cur = dst;
    }
    return cur;
}

// op: 0 erode, 1 dilate, 2 open, 3 close.  `src` is never written; the result lands in `dst`.
void morph_op(hipStream_t st, int op, const uint64_t* src, uint64_t* dst, uint64_t* t0, uint64_t* t1, int H, int W,
              int WW, int kw, int kh) {
    const size_t bytes = (size_t)H * WW * sizeof(uint64_t);
    ASEP_HIP_CHECK_THROW(hipMemcpyAsync(t0, src, bytes, hipMemcpyDeviceToDevice, st));
    uint64_t* cur = t0;
    if (op == 0 || op == 2) cur = morph_rect(st, true, cur, t0, t1, H, W, WW, kw, kh);
    if (op == 1 || op == 3) cur = morph_rect(st, false, cur, t0, t1, H, W, WW, kw, kh);
    if (op == 2) cur = morph_rect(st, false, cur, t0, t1, H, W, WW, kw, kh);
    if (op == 3) cur = morph_rect(st, true, cur, t0, t1, H, W, WW, kw, kh);
    ASEP_HIP_CHECK_THROW(hipMemcpyAsync(dst, cur, bytes, hipMemcpyDeviceToDevice, st));
}

// CC area filter: d_mask (stride, channel) -> bit plane `bits` (and optional u8).
void cc_filter_dev(asep_post* p, hipStream_t st, const uint8_t* d_mask, int H, int W, int stride, int ch,
                   int min_size, uint64_t* bits, int WW, uint8_t* d_out_u8) {
    const size_t n = (size_t)H * W;
    int32_t* L = (int32_t*)p->pool.get(n * sizeof(int32_t));
    int32_t* area = (int32_t*)p->pool.get(n * sizeof(int32_t));
    cc_init_kernel<<<dim3(WW, cdiv(H, 4)), 256, 0, st>>>(d_mask, stride, ch, H, W, L);
    cc_union_kernel<<<blocks_for(n), 256, 0, st>>>(L, H, W);
    cc_flatten_kernel<<<blocks_for(n), 256, 0, st>>>(L, area, n);
    cc_area_kernel<<<dim3(WW, cdiv(H, 16)), 256, 0, st>>>(L, area, H, W);
    cc_filter_kernel<<<dim3(WW, cdiv(H, 4)), 256, 0, st>>>(L, area, H, W, min_size, bits, WW, d_out_u8);
}

int check_image(const char* fn, int H, int W) {
    if (H < 1 || W < 1 || (size_t)H * W > 0x7fffffffull) {
        set_error("%s: unsupported image size %dx%d", fn, W, H);
        return ASEP_ERR_ARG;
    }
    return ASEP_OK;
}

int separator_dev(asep_post* p, hipStream_t st, const uint8_t* d_mask, int H, int W, int stride, int ch, int min_size,
                  int k_h, int k_v, int k_clean, uint8_t* d_out_h, uint8_t* d_out_v) {
    if (k_h < 1 || k_v < 1 || k_clean < 1) {
        // cv2.getStructuringElement asserts ksize > 0 (separator_net_post_processor.py:70-86 on tiny images)
        set_error("separator post-processing: structuring element sizes must be >= 1 (got %d, %d, %d)", k_h, k_v,
                  k_clean);
        return ASEP_ERR_ARG;
    }
    const int WW = cdiv(W, 64);
    const size_t words = (size_t)H * WW;
    uint64_t* cc = (uint64_t*)p->pool.get(words * 8);
    uint64_t* hz = (uint64_t*)p->pool.get(words * 8);
    uint64_t* vt = (uint64_t*)p->pool.get(words * 8);
    uint64_t* t0 = (uint64_t*)p->pool.get(words * 8);
    uint64_t* t1 = (uint64_t*)p->pool.get(words * 8);
    cc_filter_dev(p, st, d_mask, H, W, stride, ch, min_size, cc, WW, nullptr);
    morph_op(st, 2, cc, hz, t0, t1, H, W, WW, k_h, 1);
    morph_op(st, 2, cc, vt, t0, t1, H, W, WW, 1, k_v);
    post_andnot_kernel<<<blocks_for(words), 256, 0, st>>>(hz, vt, cc, words);
    morph_op(st, 2, cc, hz, t0, t1, H, W, WW, k_clean, 1);
    const size_t n4 = (size_t)H * ((W + 3) / 4);
    post_unpack_kernel<<<blocks_for(n4), 256, 0, st>>>(hz, H, W, WW, d_out_h);
    post_unpack_kernel<<<blocks_for(n4), 256, 0, st>>>(vt, H, W, WW, d_out_v);
    ASEP_HIP_CHECK(hipGetLastError());
    return ASEP_OK;
}

int prep_dev(asep_post* p, hipStream_t st, const uint8_t* d_img, int H, int W, int C, double sc, uint8_t* d_out_image,
             float* d_out_gray) {
    if (C != 1 && C != 3) {
        set_error("asep_prep_scale_gray: C must be 1 or 3 (got %d)", C);
        return ASEP_ERR_ARG;
    }
    if (!(sc > 0)) {
        set_error("asep_prep_scale_gray: scaling factor must be positive");
        return ASEP_ERR_ARG;
    }
    int h, w;
    scaled_size(H, W, sc, h, w);
    if (h < 1 || w < 1) {
        set_error("asep_prep_scale_gray: scaled size %dx%d is empty", w, h);
        return ASEP_ERR_ARG;
    }
    const size_t nout = (size_t)h * w * C;
    const uint8_t* scaled = d_img;
    if (sc != 1.0) {
        uint8_t* dst = d_out_image ? d_out_image : (uint8_t*)p->pool.get(nout);
        const double scale = 1.0 / sc;
        if (sc < 1.0) {
            const int iscale = cv_round(scale);
            if (std::fabs(scale - iscale) < 2.220446049250313e-16) {
                prep_area_int_kernel<<<blocks_for(nout), 256, 0, st>>>(d_img, H, W, C, iscale, dst, h, w);
            } else {
                if (!(p->tab_kind == 1 && p->tab_H == H && p->tab_W == W && p->tab_sc == sc)) {
                    p->free_tabs();
                    AreaTab tx = build_area_tab(W, w, scale), ty = build_area_tab(H, h, scale);
                    p->d_tab[0] = upload(tx.ofs);
                    p->d_tab[1] = upload(tx.src);
                    p->d_tab[2] = upload(tx.w);
                    p->d_tab[3] = upload(ty.ofs);
                    p->d_tab[4] = upload(ty.src);
                    p->d_tab[5] = upload(ty.w);
                    p->tab_kind = 1;
                    p->tab_H = H;
                    p->tab_W = W;
                    p->tab_sc = sc;
                }
                prep_area_tab_kernel<<<blocks_for(nout), 256, 0, st>>>(
                    d_img, H, W, C, dst, h, w, (const int32_t*)p->d_tab[0], (const int32_t*)p->d_tab[1],
                    (const float*)p->d_tab[2], (const int32_t*)p->d_tab[3], (const int32_t*)p->d_tab[4],
                    (const float*)p->d_tab[5]);
            }
        } else {
            if (!(p->tab_kind == 2 && p->tab_H == H && p->tab_W == W && p->tab_sc == sc)) {
                p->free_tabs();
                CubicTab tx = build_cubic_tab(w, scale), ty = build_cubic_tab(h, scale);
                p->d_tab[0] = upload(tx.idx);
                p->d_tab[1] = upload(tx.w);
                p->d_tab[2] = upload(ty.idx);
                p->d_tab[3] = upload(ty.w);
                p->tab_kind = 2;
                p->tab_H = H;
                p->tab_W = W;
                p->tab_sc = sc;
            }
            prep_cubic_kernel<<<blocks_for(nout), 256, 0, st>>>(d_img, H, W, C, dst, h, w,
                                                                  (const int32_t*)p->d_tab[0],
                                                                  (const int16_t*)p->d_tab[1],
                                                                  (const int32_t*)p->d_tab[2],
                                                                  (const int16_t*)p->d_tab[3]);
        }
        scaled = dst;
    } else if (d_out_image) {
        ASEP_HIP_CHECK(hipMemcpyAsync(d_out_image, d_img, nout, hipMemcpyDeviceToDevice, st));
    }
    if (d_out_gray)
        prep_gray_kernel<<<blocks_for((size_t)h * w), 256, 0, st>>>(scaled, (size_t)h * w, C, d_out_gray, nullptr);
    ASEP_HIP_CHECK(hipGetLastError());
    return ASEP_OK;
}

int swt_dev(asep_post* p, hipStream_t st, const uint8_t* d_gray, int H, int W, uint8_t* d_out, int32_t* d_d2,
            int** d_thr_out) {
    const size_t n = (size_t)H * W;
    uint8_t* blur = (uint8_t*)p->pool.get(n);
    unsigned int* hist = (unsigned int*)p->pool.get(257 * sizeof(unsigned int));
    int* thr = (int*)(hist + 256);
    const int nseg = cdiv(H, SWT_SEG);
    int32_t* seg_first = (int32_t*)p->pool.get((size_t)nseg * W * sizeof(int32_t));
    int32_t* seg_last = (int32_t*)p->pool.get((size_t)nseg * W * sizeof(int32_t));
    uint16_t* g = (uint16_t*)p->pool.get(n * sizeof(uint16_t));
    ASEP_HIP_CHECK(hipMemsetAsync(hist, 0, 257 * sizeof(unsigned int), st));
    swt_blur_hist_kernel<<<dim3(cdiv(W, 64), cdiv(H, 16)), 256, 0, st>>>(d_gray, H, W, blur, hist);
    swt_otsu_kernel<<<1, 64, 0, st>>>(hist, thr);
    swt_edt_seg_kernel<<<dim3(cdiv(W, 256), nseg), 256, 0, st>>>(blur, thr, H, W, seg_first, seg_last);
    swt_edt_cols_kernel<<<dim3(cdiv(W, 256), nseg), 256, 0, st>>>(blur, thr, H, W, nseg, seg_first, seg_last, g);
    swt_edt_rows_kernel<<<blocks_for(n), 256, 0, st>>>(g, H, W, d_out, d_d2);
    ASEP_HIP_CHECK(hipGetLastError());
    if (d_thr_out) *d_thr_out = thr;
    return ASEP_OK;
}

// RAII device staging for the host-pointer entry points
struct Staged {
    void* p = nullptr;
    ~Staged() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t n) {
        ASEP_HIP_CHECK(hipMalloc(&p, std::max<size_t>(n, 1)));
        return ASEP_OK;
    }
};

}  // namespace

#define POST_GUARD_BEGIN ASEP_GUARD_BEGIN
#define POST_GUARD_END ASEP_GUARD_END

extern "C" {

asep_post* asep_post_create(void) {
    asep_post* p = new asep_post();
    if (hipStreamCreateWithFlags(&p->s, hipStreamNonBlocking) != hipSuccess) {
        set_error("asep_post_create: hipStreamCreate failed (no usable GPU?)");
        delete p;
        return nullptr;
    }
    return p;
}

void asep_post_free(asep_post* p) { delete p; }

int asep_prep_scaled_size(int H, int W, double sc, int32_t* out_h, int32_t* out_w) {
    if (!out_h || !out_w || H < 1 || W < 1 || !(sc > 0)) {
        set_error("asep_prep_scaled_size: bad arguments");
        return ASEP_ERR_ARG;
    }
    int h, w;
    scaled_size(H, W, sc, h, w);
    *out_h = h;
    *out_w = w;
    return ASEP_OK;
}

int asep_prep_scale_gray_dev(asep_post* p, const uint8_t* d_img, int H, int W, int C, double sc,
                             uint8_t* d_out_image, float* d_out_gray, void* stream) {
    if (!p || !d_img) {
        set_error("asep_prep_scale_gray_dev: null argument");
        return ASEP_ERR_ARG;
    }
    if (int rc = check_image("asep_prep_scale_gray_dev", H, W)) return rc;
    POST_GUARD_BEGIN
    p->pool.begin();
    return prep_dev(p, (hipStream_t)stream, d_img, H, W, C, sc, d_out_image, d_out_gray);
    POST_GUARD_END
}

int asep_prep_scale_gray(asep_post* p, const uint8_t* img, int H, int W, int C, double sc, uint8_t* out_image,
                         float* out_gray) {
    if (!p || !img) {
        set_error("asep_prep_scale_gray: null argument");
        return ASEP_ERR_ARG;
    }
    if (int rc = check_image("asep_prep_scale_gray", H, W)) return rc;
    if (!(sc > 0)) {
        set_error("asep_prep_scale_gray: scaling factor must be positive");
        return ASEP_ERR_ARG;
    }
    POST_GUARD_BEGIN
    int h, w;
    scaled_size(H, W, sc, h, w);
    if (h < 1 || w < 1) {
        set_error("asep_prep_scale_gray: scaled size %dx%d is empty", w, h);
        return ASEP_ERR_ARG;
    }
    Staged din, dimg, dgray;
    const size_t nin = (size_t)H * W * C, nout = (size_t)h * w * C;
    if (int rc = din.alloc(nin)) return rc;
    if (int rc = dimg.alloc(nout)) return rc;
    if (int rc = dgray.alloc((size_t)h * w * sizeof(float))) return rc;
    ASEP_HIP_CHECK(hipMemcpyAsync(din.p, img, nin, hipMemcpyHostToDevice, p->s));
    p->pool.begin();
    if (int rc = prep_dev(p, p->s, (const uint8_t*)din.p, H, W, C, sc, (uint8_t*)dimg.p, (float*)dgray.p)) return rc;
    if (out_image) ASEP_HIP_CHECK(hipMemcpyAsync(out_image, dimg.p, nout, hipMemcpyDeviceToHost, p->s));
    if (out_gray)
        ASEP_HIP_CHECK(hipMemcpyAsync(out_gray, dgray.p, (size_t)h * w * sizeof(float), hipMemcpyDeviceToHost, p->s));
    ASEP_HIP_CHECK(hipStreamSynchronize(p->s));
    return ASEP_OK;
    POST_GUARD_END
}

int asep_post_cc_filter(asep_post* p, const uint8_t* mask, int H, int W, int pix_stride, int channel, int min_size,
                        uint8_t* out) {
    if (!p || !mask || !out || pix_stride < 1 || channel < 0 || channel >= pix_stride) {
        set_error("asep_post_cc_filter: bad arguments");
        return ASEP_ERR_ARG;
    }
    if (int rc = check_image("asep_post_cc_filter", H, W)) return rc;
    POST_GUARD_BEGIN
    const size_t n = (size_t)H * W;
    Staged din, dout;
    if (int rc = din.alloc(n * pix_stride)) return rc;
    if (int rc = dout.alloc(n)) return rc;
    ASEP_HIP_CHECK(hipMemcpyAsync(din.p, mask, n * pix_stride, hipMemcpyHostToDevice, p->s));
    p->pool.begin();
    const int WW = cdiv(W, 64);
    uint64_t* bits = (uint64_t*)p->pool.get((size_t)H * WW * 8);
    cc_filter_dev(p, p->s, (const uint8_t*)din.p, H, W, pix_stride, channel, min_size, bits, WW, (uint8_t*)dout.p);
    ASEP_HIP_CHECK(hipGetLastError());
    ASEP_HIP_CHECK(hipMemcpyAsync(out, dout.p, n, hipMemcpyDeviceToHost, p->s));
    ASEP_HIP_CHECK(hipStreamSynchronize(p->s));
    return ASEP_OK;
    POST_GUARD_END
}

int asep_post_morph_rect(asep_post* p, int op, const uint8_t* mask, int H, int W, int kw, int kh, uint8_t* out) {
    if (!p || !mask || !out || op < 0 || op > 3) {
        set_error("asep_post_morph_rect: bad arguments");
        return ASEP_ERR_ARG;
    }
    if (kw < 1 || kh < 1) {
        set_error("asep_post_morph_rect: structuring element %dx%d must be at least 1x1", kw, kh);
        return ASEP_ERR_ARG;
    }
    if (int rc = check_image("asep_post_morph_rect", H, W)) return rc;
    POST_GUARD_BEGIN
    const size_t n = (size_t)H * W;
    Staged din, dout;
    if (int rc = din.alloc(n)) return rc;
    if (int rc = dout.alloc(n)) return rc;
    ASEP_HIP_CHECK(hipMemcpyAsync(din.p, mask, n, hipMemcpyHostToDevice, p->s));
    p->pool.begin();
    const int WW = cdiv(W, 64);
    const size_t words = (size_t)H * WW;
    uint64_t* src = (uint64_t*)p->pool.get(words * 8);
    uint64_t* dst = (uint64_t*)p->pool.get(words * 8);
    uint64_t* t0 = (uint64_t*)p->pool.get(words * 8);
    uint64_t* t1 = (uint64_t*)p->pool.get(words * 8);
    post_pack_kernel<<<dim3(WW, cdiv(H, 4)), 256, 0, p->s>>>((const uint8_t*)din.p, H, W, 1, 0, src, WW);
    morph_op(p->s, op, src, dst, t0, t1, H, W, WW, kw, kh);
    post_unpack_kernel<<<blocks_for((size_t)H * ((W + 3) / 4)), 256, 0, p->s>>>(dst, H, W, WW, (uint8_t*)dout.p);
    ASEP_HIP_CHECK(hipGetLastError());
    ASEP_HIP_CHECK(hipMemcpyAsync(out, dout.p, n, hipMemcpyDeviceToHost, p->s));
    ASEP_HIP_CHECK(hipStreamSynchronize(p->s));
    return ASEP_OK;
    POST_GUARD_END
}

int asep_post_separator_dev(asep_post* p, const uint8_t* d_mask, int H, int W, int pix_stride, int channel,
                            int min_size, int k_h, int k_v, int k_clean, uint8_t* d_out_horizontal,
                            uint8_t* d_out_vertical, void* stream) {
    if (!p || !d_mask || !d_out_horizontal || !d_out_vertical || pix_stride < 1 || channel < 0 ||
        channel >= pix_stride) {
        set_error("asep_post_separator_dev: bad arguments");
        return ASEP_ERR_ARG;
    }
    if (int rc = check_image("asep_post_separator_dev", H, W)) return rc;
    POST_GUARD_BEGIN
    p->pool.begin();
    return separator_dev(p, (hipStream_t)stream, d_mask, H, W, pix_stride, channel, min_size, k_h, k_v, k_clean,
                         d_out_horizontal, d_out_vertical);
    POST_GUARD_END
}

int asep_post_separator(asep_post* p, const uint8_t* mask, int H, int W, int pix_stride, int channel, int min_size,
                        int k_h, int k_v, int k_clean, uint8_t* out_horizontal, uint8_t* out_vertical) {
    if (!p || !mask || !out_horizontal || !out_vertical || pix_stride < 1 || channel < 0 || channel >= pix_stride) {
        set_error("asep_post_separator: bad arguments");
        return ASEP_ERR_ARG;
    }
    if (int rc = check_image("asep_post_separator", H, W)) return rc;
    POST_GUARD_BEGIN
    const size_t n = (size_t)H * W;
    Staged din, dh, dv;
    if (int rc = din.alloc(n * pix_stride)) return rc;
    if (int rc = dh.alloc(n)) return rc;
    if (int rc = dv.alloc(n)) return rc;
    ASEP_HIP_CHECK(hipMemcpyAsync(din.p, mask, n * pix_stride, hipMemcpyHostToDevice, p->s));
    p->pool.begin();
    if (int rc = separator_dev(p, p->s, (const uint8_t*)din.p, H, W, pix_stride, channel, min_size, k_h, k_v, k_clean,
                               (uint8_t*)dh.p, (uint8_t*)dv.p))
        return rc;
    ASEP_HIP_CHECK(hipMemcpyAsync(out_horizontal, dh.p, n, hipMemcpyDeviceToHost, p->s));
    ASEP_HIP_CHECK(hipMemcpyAsync(out_vertical, dv.p, n, hipMemcpyDeviceToHost, p->s));
    ASEP_HIP_CHECK(hipStreamSynchronize(p->s));
    return ASEP_OK;
    POST_GUARD_END
}

long asep_post_boundary_segments_dev(asep_post* p, const uint8_t* d_mask, int H, int W, int value, int32_t* d_starts,
                                     int32_t* d_ends, long capacity, void* stream) {
    if (!p || !d_mask || capacity < 0 || (capacity > 0 && (!d_starts || !d_ends))) {
        set_error("asep_post_boundary_segments_dev: bad arguments");
        return ASEP_ERR_ARG;
    }
    if (int rc = check_image("asep_post_boundary_segments_dev", H, W)) return rc;
    if ((size_t)(H + 1) * (W + 1) * 4 > 0x7fffffffull) {
        set_error("asep_post_boundary_segments_dev: image too large for 32-bit vertex keys");
        return ASEP_ERR_UNSUPPORTED;
    }
    POST_GUARD_BEGIN
    hipStream_t st = (hipStream_t)stream;
    p->pool.begin();
    unsigned long long* counter = (unsigned long long*)p->pool.get(2 * sizeof(unsigned long long));
    ASEP_HIP_CHECK(hipMemsetAsync(counter, 0, 2 * sizeof(unsigned long long), st));
    post_boundary_segments_kernel<<<blocks_for(((size_t)H * W + 15) / 16), 256, 0, st>>>(d_mask, H, W, value, d_starts, d_ends,
                                                                             (unsigned long long)capacity, counter);
    ASEP_HIP_CHECK(hipGetLastError());
    unsigned long long total[2] = {0, 0};
    ASEP_HIP_CHECK(hipMemcpyAsync(total, counter, sizeof(total), hipMemcpyDeviceToHost, st));
    ASEP_HIP_CHECK(hipStreamSynchronize(st));
    if (total[0] != total[1]) {
        set_error("boundary segments: %llu starts but %llu ends", total[0], total[1]);
        return ASEP_ERR_HIP;
    }
    return (long)total[0];
    POST_GUARD_END
}

int asep_post_boundary_segments_enqueue_dev(asep_post* p, const uint8_t* d_mask, int H, int W, int value,
                                            int32_t* d_starts, int32_t* d_ends, long capacity,
                                            unsigned long long* d_totals, void* stream) {
    if (!p || !d_mask || !d_totals || capacity < 0 || (capacity > 0 && (!d_starts || !d_ends))) {
        set_error("asep_post_boundary_segments_enqueue_dev: bad arguments");
        return ASEP_ERR_ARG;
    }
    if (int rc = check_image("asep_post_boundary_segments_enqueue_dev", H, W)) return rc;
    if ((size_t)(H + 1) * (W + 1) * 4 > 0x7fffffffull) {
        set_error("asep_post_boundary_segments_enqueue_dev: image too large for 32-bit vertex keys");
        return ASEP_ERR_UNSUPPORTED;
    }
    POST_GUARD_BEGIN
    hipStream_t st = (hipStream_t)stream;
    ASEP_HIP_CHECK(hipMemsetAsync(d_totals, 0, 2 * sizeof(unsigned long long), st));
    post_boundary_segments_kernel<<<blocks_for(((size_t)H * W + 15) / 16), 256, 0, st>>>(d_mask, H, W, value, d_starts, d_ends,
                                                                             (unsigned long long)capacity, d_totals);
    ASEP_HIP_CHECK(hipGetLastError());
    return ASEP_OK;
    POST_GUARD_END
}

long asep_post_boundary_segments(asep_post* p, const uint8_t* mask, int H, int W, int value, int32_t* out_starts,
                                 int32_t* out_ends, long capacity) {
    if (!p || !mask || capacity < 0 || (capacity > 0 && (!out_starts || !out_ends))) {
        set_error("asep_post_boundary_segments: bad arguments");
        return ASEP_ERR_ARG;
    }
    if (int rc = check_image("asep_post_boundary_segments", H, W)) return rc;
    POST_GUARD_BEGIN
    const size_t n = (size_t)H * W;
    Staged din, ds, de;
    if (int rc = din.alloc(n)) return rc;
    if (int rc = ds.alloc((size_t)capacity * sizeof(int32_t))) return rc;
    if (int rc = de.alloc((size_t)capacity * sizeof(int32_t))) return rc;
    ASEP_HIP_CHECK(hipMemcpyAsync(din.p, mask, n, hipMemcpyHostToDevice, p->s));
    const long total = asep_post_boundary_segments_dev(p, (const uint8_t*)din.p, H, W, value, (int32_t*)ds.p,
                                                       (int32_t*)de.p, capacity, p->s);
    if (total < 0) return total;
    const long ncopy = total < capacity ? total : capacity;
    if (ncopy > 0) {
        ASEP_HIP_CHECK(hipMemcpy(out_starts, ds.p, (size_t)ncopy * sizeof(int32_t), hipMemcpyDeviceToHost));
        ASEP_HIP_CHECK(hipMemcpy(out_ends, de.p, (size_t)ncopy * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    return total;
    POST_GUARD_END
}

int asep_swt_distance_transform_dev(asep_post* p, const uint8_t* d_gray, int H, int W, uint8_t* d_out, void* stream) {
    if (!p || !d_gray || !d_out) {
        set_error("asep_swt_distance_transform_dev: null argument");
        return ASEP_ERR_ARG;
    }
    if (int rc = check_image("asep_swt_distance_transform_dev", H, W)) return rc;
    POST_GUARD_BEGIN
    p->pool.begin();
    return swt_dev(p, (hipStream_t)stream, d_gray, H, W, d_out, nullptr, nullptr);
    POST_GUARD_END
}

int asep_swt_line_features_dev(asep_post* p, const uint8_t* d_swt, int H, int W, int n_lines, const int32_t* boxes,
                                float* out_stroke_width, int32_t* out_height, int32_t* out_flag, void* stream) {
    if (!p || !d_swt || n_lines < 0 || (n_lines > 0 && (!boxes || !out_stroke_width || !out_height || !out_flag))) {
        set_error("asep_swt_line_features_dev: bad arguments");
        return ASEP_ERR_ARG;
    }
    if (int rc = check_image("asep_swt_line_features_dev", H, W)) return rc;
    if (n_lines == 0) return ASEP_OK;
    POST_GUARD_BEGIN
    hipStream_t st = (hipStream_t)stream;
    std::vector<SwtLineBox> hb(n_lines);
    std::vector<unsigned long long> ofs(n_lines);
    unsigned long long total = 0;
    long long max_crop = 0;
    for (int i = 0; i < n_lines; ++i) {
        // numpy slicing semantics of swt[y0:y1, x0:x1]: bounds are clipped to the image
        SwtLineBox b{boxes[4 * i + 0], boxes[4 * i + 1], boxes[4 * i + 2], boxes[4 * i + 3]};
        b.x0 = std::min(std::max(b.x0, 0), W); b.x1 = std::min(std::max(b.x1, 0), W);
        b.y0 = std::min(std::max(b.y0, 0), H); b.y1 = std::min(std::max(b.y1, 0), H);
        hb[i] = b;
        ofs[i] = total;
        const long long cw = b.x1 - b.x0, chh = b.y1 - b.y0;
        if (cw > 0 && chh > 0) {
            if (cw * chh > 0x3fffffffll) { set_error("asep_swt_line_features_dev: line %d crop too large", i); return ASEP_ERR_UNSUPPORTED; }
            total += 2ull * (unsigned long long)(cw * chh);
            max_crop = std::max(max_crop, cw * chh);
        }
    }
    p->pool.begin();
    SwtLineBox* d_boxes = (SwtLineBox*)p->pool.get((size_t)n_lines * sizeof(SwtLineBox));
    unsigned long long* d_ofs = (unsigned long long*)p->pool.get((size_t)n_lines * sizeof(unsigned long long));
    int32_t* d_scratch = (int32_t*)p->pool.get(std::max<size_t>((size_t)total, 1) * sizeof(int32_t));
    float* d_sw = (float*)p->pool.get((size_t)n_lines * sizeof(float));
    int32_t* d_h = (int32_t*)p->pool.get((size_t)n_lines * sizeof(int32_t));
    int32_t* d_f = (int32_t*)p->pool.get((size_t)n_lines * sizeof(int32_t));
    ASEP_HIP_CHECK(hipMemcpyAsync(d_boxes, hb.data(), (size_t)n_lines * sizeof(SwtLineBox), hipMemcpyHostToDevice, st));
    ASEP_HIP_CHECK(hipMemcpyAsync(d_ofs, ofs.data(), (size_t)n_lines * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
    ASEP_HIP_CHECK(hipStreamSynchronize(st));          // hb / ofs are stack-owned: finish the copies before they die
    // label tile in LDS: sized for the largest crop of THIS call that fits (a page of small lines keeps more workgroups per CU
    // than one that reserves 64 KB each); crops beyond SWTL_LDS_PIXELS label in the global scratch slice.  The attribute is a
    // property of the function ON A DEVICE: one flag per device id, not one per process.
    const size_t lds_bytes = (size_t)std::min<long long>(std::max<long long>(max_crop, 1), SWTL_LDS_PIXELS) * sizeof(int32_t);
    {
        static std::mutex mu;
        static std::map<int, bool> ok_on;
        int dev = 0;
        ASEP_HIP_CHECK(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(mu);
        auto it = ok_on.find(dev);
        if (it == ok_on.end())
            it = ok_on.emplace(dev, hipFuncSetAttribute((const void*)swt_line_features_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        SWTL_LDS_PIXELS * (int)sizeof(int32_t)) == hipSuccess).first;
        if (!it->second) { set_error("asep_swt_line_features_dev: cannot reserve the label tile in LDS on device %d", dev); return ASEP_ERR_HIP; }
    }
    swt_line_features_kernel<<<n_lines, 256, lds_bytes, st>>>(d_swt, W, d_boxes, d_ofs, d_scratch, d_sw, d_h, d_f);
    ASEP_HIP_CHECK(hipGetLastError());
    ASEP_HIP_CHECK(hipMemcpyAsync(out_stroke_width, d_sw, (size_t)n_lines * sizeof(float), hipMemcpyDeviceToHost, st));
    ASEP_HIP_CHECK(hipMemcpyAsync(out_height, d_h, (size_t)n_lines * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    ASEP_HIP_CHECK(hipMemcpyAsync(out_flag, d_f, (size_t)n_lines * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    ASEP_HIP_CHECK(hipStreamSynchronize(st));
    return ASEP_OK;
    POST_GUARD_END
}

int asep_prep_gray_u8_dev(asep_post* p, const uint8_t* d_bgr, int H, int W, uint8_t* d_out, void* stream) {
    if (!p || !d_bgr || !d_out) {
        set_error("asep_prep_gray_u8_dev: null argument");
        return ASEP_ERR_ARG;
    }
    if (int rc = check_image("asep_prep_gray_u8_dev", H, W)) return rc;
    POST_GUARD_BEGIN
    const size_t n = (size_t)H * W;
    post_gray_u8_kernel<<<blocks_for((n + 3) / 4), 256, 0, (hipStream_t)stream>>>(d_bgr, n, d_out);
    ASEP_HIP_CHECK(hipGetLastError());
    return ASEP_OK;
    POST_GUARD_END
}

int asep_post_box_sums_dev(asep_post* p, const uint8_t* d_img, int H, int W, int pix_stride, int channel, int n_boxes,
                           const int32_t* boxes, int64_t* out_sums, void* stream) {
    if (!p || !d_img || n_boxes < 0 || (n_boxes > 0 && (!boxes || !out_sums)) || pix_stride < 1 || channel < 0 ||
        channel >= pix_stride) {
        set_error("asep_post_box_sums_dev: bad arguments");
        return ASEP_ERR_ARG;
    }
    if (int rc = check_image("asep_post_box_sums_dev", H, W)) return rc;
    if (n_boxes == 0) return ASEP_OK;
    POST_GUARD_BEGIN
    hipStream_t st = (hipStream_t)stream;
    std::vector<PostBox> hb(n_boxes);
    for (int i = 0; i < n_boxes; ++i) {
        // numpy slicing semantics of img[y0:y1, x0:x1] for non-negative bounds: clipped to the image
        PostBox b{boxes[4 * i + 0], boxes[4 * i + 1], boxes[4 * i + 2], boxes[4 * i + 3]};
        b.x0 = std::min(std::max(b.x0, 0), W); b.x1 = std::min(std::max(b.x1, 0), W);
        b.y0 = std::min(std::max(b.y0, 0), H); b.y1 = std::min(std::max(b.y1, 0), H);
        hb[i] = b;
    }
    p->pool.begin();
    PostBox* d_boxes = (PostBox*)p->pool.get((size_t)n_boxes * sizeof(PostBox));
    unsigned long long* d_sums = (unsigned long long*)p->pool.get((size_t)n_boxes * sizeof(unsigned long long));
    ASEP_HIP_CHECK(hipMemcpyAsync(d_boxes, hb.data(), (size_t)n_boxes * sizeof(PostBox), hipMemcpyHostToDevice, st));
    post_box_sums_kernel<<<n_boxes, 256, 0, st>>>(d_img, W, pix_stride, channel, d_boxes, d_sums);
    ASEP_HIP_CHECK(hipGetLastError());
    ASEP_HIP_CHECK(hipMemcpyAsync(out_sums, d_sums, (size_t)n_boxes * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    ASEP_HIP_CHECK(hipStreamSynchronize(st));          // hb is stack-owned and the sums are the caller's next input
    return ASEP_OK;
    POST_GUARD_END
}

int asep_swt_line_features(asep_post* p, const uint8_t* swt, int H, int W, int n_lines, const int32_t* boxes,
                            float* out_stroke_width, int32_t* out_height, int32_t* out_flag) {
    if (!p || !swt) {
        set_error("asep_swt_line_features: null argument");
        return ASEP_ERR_ARG;
    }
    if (int rc = check_image("asep_swt_line_features", H, W)) return rc;
    POST_GUARD_BEGIN
    Staged din;
    if (int rc = din.alloc((size_t)H * W)) return rc;
    ASEP_HIP_CHECK(hipMemcpyAsync(din.p, swt, (size_t)H * W, hipMemcpyHostToDevice, p->s));
    return asep_swt_line_features_dev(p, (const uint8_t*)din.p, H, W, n_lines, boxes, out_stroke_width, out_height,
                                      out_flag, p->s);
    POST_GUARD_END
}

int asep_swt_distance_transform(asep_post* p, const uint8_t* gray, int H, int W, uint8_t* out, int32_t* out_otsu,
                                int32_t* out_d2) {
    if (!p || !gray || !out) {
        set_error("asep_swt_distance_transform: null argument");
        return ASEP_ERR_ARG;
    }
    if (int rc = check_image("asep_swt_distance_transform", H, W)) return rc;
    POST_GUARD_BEGIN
    const size_t n = (size_t)H * W;
    Staged din, dout, dd2;
    if (int rc = din.alloc(n)) return rc;
    if (int rc = dout.alloc(n)) return rc;
    if (out_d2)
        if (int rc = dd2.alloc(n * sizeof(int32_t))) return rc;
    ASEP_HIP_CHECK(hipMemcpyAsync(din.p, gray, n, hipMemcpyHostToDevice, p->s));
    p->pool.begin();
    int* d_thr = nullptr;
    if (int rc = swt_dev(p, p->s, (const uint8_t*)din.p, H, W, (uint8_t*)dout.p, (int32_t*)dd2.p, &d_thr)) return rc;
    ASEP_HIP_CHECK(hipMemcpyAsync(out, dout.p, n, hipMemcpyDeviceToHost, p->s));
    if (out_d2) ASEP_HIP_CHECK(hipMemcpyAsync(out_d2, dd2.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, p->s));
    if (out_otsu) ASEP_HIP_CHECK(hipMemcpyAsync(out_otsu, d_thr, sizeof(int32_t), hipMemcpyDeviceToHost, p->s));
    ASEP_HIP_CHECK(hipStreamSynchronize(p->s));
    return ASEP_OK;
    POST_GUARD_END
}

}  // extern "C"
