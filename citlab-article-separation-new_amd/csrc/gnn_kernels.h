// GNN relation-predictor device kernels for gfx950.
//
// Reference semantics (file:line in /root/reference):
//   gnn/model/graph_util/misc.py:7-151       edge-list correction      -> edge_table / edge_count / edge_scan / edge_emit
//   gnn/model/graph/message_fn_chunk.py:148-418  edge MLP + 1/indeg + segmented sum -> gnn_message_kernel
//   gnn/model/graph/update_fn_lstm.py:31-101     LSTM node update        -> gnn_lstm_kernel
//   gnn/model/graph/graph_relation.py:229-287    pair classifier + softmax -> gnn_pair_pre / gnn_pair_cls
//
// The message-passing aggregation is a segmented reduction over edges grouped by TARGET node
// (column 1 of the edge list): one workgroup per target, node state of the target kept in registers,
// the edge-MLP weights staged once per workgroup in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace asep {

constexpr int GNN_H = 32;            // hidden / interaction width handled by the lane mapping
constexpr int EMPTY_SLOT = 0x7f7f7f7f;

// table[from*N+to] = first index of that directed edge in the (symmetrised) list  (misc.py:47-88)
__global__ __launch_bounds__(256) void edge_table_kernel(const int32_t* __restrict__ edges, int E, int N,
                                                         int undirected, int* __restrict__ table) {
    const int total = undirected ? 2 * E : E;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        int f, t;
        if (idx < E) { f = edges[2 * idx]; t = edges[2 * idx + 1]; }
        else { f = edges[2 * (idx - E) + 1]; t = edges[2 * (idx - E)]; }
        if (f == t || f < 0 || t < 0 || f >= N || t >= N) continue;     // self loops are removed (misc.py:68-72)
        atomicMin(&table[(size_t)f * N + t], idx);
    }
}

// per-row (from) and per-column (to) counts of occupied table cells; one wave per row / column
__global__ __launch_bounds__(64) void edge_count_kernel(const int* __restrict__ table, int N,
                                                        int* __restrict__ rowcnt, int* __restrict__ colcnt) {
    const int r = blockIdx.x, lane = threadIdx.x;
    int rc = 0, cc = 0;
    for (int k = lane; k < N; k += 64) {
        rc += table[(size_t)r * N + k] != EMPTY_SLOT;
        cc += table[(size_t)k * N + r] != EMPTY_SLOT;
    }
    for (int o = 32; o > 0; o >>= 1) { rc += __shfl_down(rc, o); cc += __shfl_down(cc, o); }
    if (lane == 0) { rowcnt[r] = rc; colcnt[r] = cc; }
}

// exclusive scans of rowcnt / colcnt (N is a few hundred: one workgroup)
__global__ __launch_bounds__(1024) void edge_scan_kernel(const int* __restrict__ rowcnt, const int* __restrict__ colcnt,
                                                         int N, int* __restrict__ rowptr, int* __restrict__ colptr) {
    __shared__ int sa[1024], sb[1024];
    const int tid = threadIdx.x;
    const int per = (N + 1023) / 1024;
    int a = 0, b = 0;
    for (int k = tid * per; k < min(N, (tid + 1) * per); ++k) { a += rowcnt[k]; b += colcnt[k]; }
    sa[tid] = a; sb[tid] = b;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        int va = tid >= o ? sa[tid - o] : 0, vb = tid >= o ? sb[tid - o] : 0;
        __syncthreads();
        sa[tid] += va; sb[tid] += vb;
        __syncthreads();
    }
    int pa = sa[tid] - a, pb = sb[tid] - b;
    for (int k = tid * per; k < min(N, (tid + 1) * per); ++k) {
        rowptr[k] = pa; colptr[k] = pb;
        pa += rowcnt[k]; pb += colcnt[k];
    }
    if (tid == 1023) { rowptr[N] = sa[1023]; colptr[N] = sb[1023]; }
}

// emit (a) the corrected edge list sorted by from*N+to with first-occurrence indices (misc.py:91-104)
//      (b) the same edges grouped by target (CSR over column 1) for the aggregation
__global__ __launch_bounds__(64) void edge_emit_kernel(const int* __restrict__ table, int N,
                                                       const int* __restrict__ rowptr, const int* __restrict__ colptr,
                                                       int32_t* __restrict__ sorted_edges, int* __restrict__ sorted_first,
                                                       int* __restrict__ tsrc, int* __restrict__ tfirst) {
    const int r = blockIdx.x, lane = threadIdx.x;
    int wr = rowptr[r], wc = colptr[r];
    for (int k0 = 0; k0 < N; k0 += 64) {
        const int k = k0 + lane;
        const int vr = k < N ? table[(size_t)r * N + k] : EMPTY_SLOT;
        const int vc = k < N ? table[(size_t)k * N + r] : EMPTY_SLOT;
        const unsigned long long mr = __ballot(vr != EMPTY_SLOT), mc = __ballot(vc != EMPTY_SLOT);
        const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
        if (vr != EMPTY_SLOT) {
            const int p = wr + __popcll(mr & below);
            sorted_edges[2 * p] = r; sorted_edges[2 * p + 1] = k; sorted_first[p] = vr;
        }
        if (vc != EMPTY_SLOT) {
            const int p = wc + __popcll(mc & below);
            tsrc[p] = k; tfirst[p] = vc;
        }
        wr += __popcll(mr); wc += __popcll(mc);
    }
}

__global__ __launch_bounds__(256) void edge_feat_gather_kernel(const float* __restrict__ ef, int E, int Ed,
                                                               const int* __restrict__ first, int n,
                                                               float* __restrict__ out) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n * Ed; i += gridDim.x * 256) {
        const int e = i / Ed, d = i - e * Ed;
        const int src = first[e] % E;            // tiled x2 for the reversed half (misc.py:56)
        out[i] = ef[(size_t)src * Ed + d];
    }
}

// ------------------------------------------------------------------------------------------------
// message function: one workgroup per target node j, eight half-waves process eight in-edges at a
// time; lane o of a half-wave owns hidden unit / output unit o.
// ------------------------------------------------------------------------------------------------
struct MsgArgs {
    const float* u;      // [N,U] node input features
    const float* h;      // [N,32] hidden state
    const float* ef;     // [E,Ed] original edge features (indexed by first-occurrence % E)
    const int* tptr;     // [N+1] CSR over targets
    const int* tsrc;     // [E'] source node of each in-edge
    const int* tfirst;   // [E'] first-occurrence index -> edge feature row (% E)
    const float* W1; const float* b1;   // [K,32], [32]
    const float* W2; const float* b2;   // [32,32], [32]
    float* x;            // [N,32] aggregated messages
    int N, U, Ed, E, K;
};

__global__ __launch_bounds__(256) void gnn_message_kernel(const MsgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* W1s = sm;                       // K*32
    float* W2s = W1s + a.K * 32;           // 32*32
    float* zs = W2s + 32 * 32;             // 8*K
    float* hs = zs + 8 * a.K;              // 8*32
    const int tid = threadIdx.x, half = tid >> 5, o = tid & 31;
    const int j = blockIdx.x;
    for (int i = tid; i < a.K * 32; i += 256) W1s[i] = a.W1[i];
    for (int i = tid; i < 32 * 32; i += 256) W2s[i] = a.W2[i];
    const int beg = a.tptr[j], end = a.tptr[j + 1];
    const int U = a.U, Ed = a.Ed, K = a.K;
    const float bias1 = a.b1[o], bias2 = a.b2[o];
    float xacc = 0.f;
    float* z = zs + half * K;
    const float* uj = a.u + (size_t)j * U;
    const float* hj = a.h + (size_t)j * 32;
    for (int base = beg; base < end; base += 8) {
        const int e = base + half;
        const bool valid = e < end;
        __syncthreads();                   // previous round's z / hs fully consumed (also covers weight staging)
        if (valid) {
            const int i = a.tsrc[e];
            const float* ui = a.u + (size_t)i * U;
            const float* hi = a.h + (size_t)i * 32;
            const float* efr = a.ef + (size_t)(a.tfirst[e] % a.E) * Ed;
            // z = [u_i, u_j, u_j-u_i, (u_j-u_i)^2, ef, h_i, h_j, h_j-h_i, (h_j-h_i)^2]  (message_fn_chunk.py:313-350)
            for (int k = o; k < U; k += 32) {
                const float vi = ui[k], vj = uj[k], d = vj - vi;
                z[k] = vi; z[U + k] = vj; z[2 * U + k] = d; z[3 * U + k] = d * d;
            }
            for (int k = o; k < Ed; k += 32) z[4 * U + k] = efr[k];
            {
                const float vi = hi[o], vj = hj[o], d = vj - vi;
                float* zh = z + 4 * U + Ed;
                zh[o] = vi; zh[32 + o] = vj; zh[64 + o] = d; zh[96 + o] = d * d;
            }
        }
        __syncthreads();
        float acc = bias1;
        if (valid) {
            for (int k = 0; k < K; ++k) acc = fmaf(z[k], W1s[k * 32 + o], acc);
            hs[half * 32 + o] = fmaxf(acc, 0.f);
        }
        __syncthreads();
        if (valid) {
            float acc2 = bias2;
#pragma unroll
            for (int k = 0; k < 32; ++k) acc2 = fmaf(hs[half * 32 + k], W2s[k * 32 + o], acc2);
            xacc += tanhf(acc2);
        }
    }
    __syncthreads();
    hs[half * 32 + o] = xacc;
    __syncthreads();
    if (tid < 32) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) s += hs[q * 32 + tid];
        const int deg = end - beg;
        a.x[(size_t)j * 32 + tid] = deg > 0 ? s / (float)deg : 0.f;   // a_ij = 1/indeg(j) (message_fn_chunk.py:369-386)
    }
}

// ------------------------------------------------------------------------------------------------
// Fused message passing step (message_fn_chunk.py:148-418 + update_fn_lstm.py:31-85), MFMA version.
// One workgroup per target node j; each of its 4 waves walks the in-edges of j in tiles of 16 edges:
//   layer 1  D1[unit][edge] = W1^T z        v_mfma_f32_16x16x4_f32, A = W1 fragments held in registers,
//                                            B = z built in registers from gathered node rows (no LDS);
//   layer 2  D2 = W2^T relu(D1 + b1)        the D layout of layer 1 IS the B layout of layer 2 (rows 4kk+r);
//   tanh, masked by edge validity, accumulated in registers; one cross-lane + cross-wave reduction per target
//   gives x_j = (1/indeg) sum_i m_ij; the same workgroup then applies the LSTM update of node j.
// The K axis (rows of W1) is permuted into "quads" of 4 consecutive features so that every lane builds its
// 4 slots of a 16-slot chunk from one 16-byte piece of a node row: quad descriptor = (type, element offset).
// ------------------------------------------------------------------------------------------------
typedef float gf32x4 __attribute__((ext_vector_type(4)));
enum { GQ_ZERO = 0, GQ_UI, GQ_UJ, GQ_DU, GQ_DU2, GQ_EF, GQ_HI, GQ_HJ, GQ_DH, GQ_DH2 };
constexpr int GNN_MAXCH = 16;          // K chunks of 16 slots (U <= 8, Ed <= 4 -> 11 chunks)

#ifdef ASEP_ABLATION   // (make ABLATION=1: the stage-by-stage form of the batched relation nets, measured and not adopted -- DESIGN_LESSONS 36)
// Several pages' launches of one graph kernel as ONE launch (asep_gnn_forward_visual_batch_dev): blockIdx.y = page, p[page] = the arguments the
// per-page launch would get, nx[page] = its grid size along x (blocks beyond it leave at once).  By value in the kernel arguments (< 4 KB).
constexpr int GNN_BATCH = 16;
template <class A>
struct GnnBatch {
    A p[GNN_BATCH];
    int nx[GNN_BATCH];
};
#endif

struct StepArgs {
    const float* u; const float* h_in; const float* c_in; const float* ef;
    const int* tptr; const int* tsrc; const int* tfirst;
    const gf32x4* A1;      // [nch][2 m-tiles][64 lanes]
    const gf32x4* A2;      // [2 chunks][2 m-tiles][64 lanes]
    const float* b1; const float* b2;
    const float* Wg[4]; const float* bg[4];
    float* h_out; float* c_out;
    int N, U, Ed, E, nch;
    const unsigned char* qdesc;   // [nch*4][2] = (type, offset) of every quad, device memory
};

__device__ __forceinline__ float gsig(float v) { return 1.f / (1.f + expf(-v)); }

__device__ __forceinline__ void gnn_step_kernel_body(const StepArgs& a, const int bx, const int gx) {
    __shared__ float xs[4][32];
    __shared__ float vs[32 + 32 + 64];
    __shared__ float gs[4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int tgt = bx;
    const int beg = a.tptr[tgt], end = a.tptr[tgt + 1];
    const int U = a.U, Ed = a.Ed;

    gf32x4 xacc[2] = {gf32x4{0.f, 0.f, 0.f, 0.f}, gf32x4{0.f, 0.f, 0.f, 0.f}};
    const int ntiles = (end - beg + 15) >> 4;
    if (wave < ntiles) {
        gf32x4 A1[GNN_MAXCH][2];
#pragma unroll
        for (int c = 0; c < GNN_MAXCH; ++c)
            if (c < a.nch) { A1[c][0] = a.A1[(c * 2 + 0) * 64 + lane]; A1[c][1] = a.A1[(c * 2 + 1) * 64 + lane]; }
        gf32x4 A2[2][2];
#pragma unroll
        for (int c = 0; c < 2; ++c) { A2[c][0] = a.A2[(c * 2 + 0) * 64 + lane]; A2[c][1] = a.A2[(c * 2 + 1) * 64 + lane]; }
        const gf32x4 b1v[2] = {*reinterpret_cast<const gf32x4*>(a.b1 + kk * 4), *reinterpret_cast<const gf32x4*>(a.b1 + 16 + kk * 4)};
        const gf32x4 b2v[2] = {*reinterpret_cast<const gf32x4*>(a.b2 + kk * 4), *reinterpret_cast<const gf32x4*>(a.b2 + 16 + kk * 4)};
        // target-side rows (identical for every edge of this block)
        const float* hjp = a.h_in + (size_t)tgt * 32;
        const gf32x4 hj[2] = {*reinterpret_cast<const gf32x4*>(hjp + kk * 4), *reinterpret_cast<const gf32x4*>(hjp + 16 + kk * 4)};
        float uj[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) uj[q] = q < U ? a.u[(size_t)tgt * U + q] : 0.f;

        for (int t = wave; t < ntiles; t += 4) {
            const int e = beg + t * 16 + j;
            const bool valid = e < end;
            const int ec = valid ? e : beg;                        // clamp: padded columns read a real edge, masked later
            const int src = a.tsrc[ec];
            const int fi = a.tfirst[ec] % a.E;
            const float* hip = a.h_in + (size_t)src * 32;
            const gf32x4 hi[2] = {*reinterpret_cast<const gf32x4*>(hip + kk * 4), *reinterpret_cast<const gf32x4*>(hip + 16 + kk * 4)};
            float ui[8], efv[4];
#pragma unroll
            for (int q = 0; q < 8; ++q) ui[q] = q < U ? a.u[(size_t)src * U + q] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) efv[q] = q < Ed ? a.ef[(size_t)fi * Ed + q] : 0.f;

            gf32x4 acc1[2] = {gf32x4{0.f, 0.f, 0.f, 0.f}, gf32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int c = 0; c < GNN_MAXCH; ++c) {
                if (c < a.nch) {
                    const int ty = a.qdesc[(c * 4 + kk) * 2], off = a.qdesc[(c * 4 + kk) * 2 + 1];
                    gf32x4 z;
                    if (ty >= GQ_HI) {
                        const gf32x4 x0 = off ? hi[1] : hi[0], x1 = off ? hj[1] : hj[0];   // off = 16*c' (+4kk implied)
                        const gf32x4 dd = x1 - x0;
                        z = ty == GQ_HI ? x0 : (ty == GQ_HJ ? x1 : (ty == GQ_DH ? dd : dd * dd));
                    } else if (ty == GQ_EF) {
                        z = gf32x4{efv[0], efv[1], efv[2], efv[3]};
                    } else if (ty == GQ_ZERO) {
                        z = gf32x4{0.f, 0.f, 0.f, 0.f};
                    } else {
                        const gf32x4 x0 = off ? gf32x4{ui[4], ui[5], ui[6], ui[7]} : gf32x4{ui[0], ui[1], ui[2], ui[3]};
                        const gf32x4 x1 = off ? gf32x4{uj[4], uj[5], uj[6], uj[7]} : gf32x4{uj[0], uj[1], uj[2], uj[3]};
                        const gf32x4 dd = x1 - x0;
                        z = ty == GQ_UI ? x0 : (ty == GQ_UJ ? x1 : (ty == GQ_DU ? dd : dd * dd));
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[c][0][r], z[r], acc1[0], 0, 0, 0);
                        acc1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[c][1][r], z[r], acc1[1], 0, 0, 0);
                    }
                }
            }
            // hidden = relu(D1 + b1): rows 4kk+r of m-tile m == B slots 4kk+r of chunk m of the second layer
            gf32x4 hid[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                hid[m] = acc1[m] + b1v[m];
                hid[m].x = fmaxf(hid[m].x, 0.f); hid[m].y = fmaxf(hid[m].y, 0.f); hid[m].z = fmaxf(hid[m].z, 0.f); hid[m].w = fmaxf(hid[m].w, 0.f);
            }
            gf32x4 acc2[2] = {gf32x4{0.f, 0.f, 0.f, 0.f}, gf32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[c][0][r], hid[c][r], acc2[0], 0, 0, 0);
                    acc2[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[c][1][r], hid[c][r], acc2[1], 0, 0, 0);
                }
            if (valid) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const gf32x4 v = acc2[m] + b2v[m];
                    xacc[m].x += tanhf(v.x); xacc[m].y += tanhf(v.y); xacc[m].z += tanhf(v.z); xacc[m].w += tanhf(v.w);
                }
            }
        }
    }
    // ---- sum over the 16 edge columns (lanes j), then over the 4 waves ----
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = xacc[m][r];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            if (j == 0) xs[wave][16 * m + 4 * kk + r] = v;
        }
    __syncthreads();
    if (tid < 32) {
        const int deg = end - beg;
        const float s = xs[0][tid] + xs[1][tid] + xs[2][tid] + xs[3][tid];
        vs[tid] = deg > 0 ? s / (float)deg : 0.f;                    // a_ij = 1/indeg(j)
        vs[32 + tid] = a.h_in[(size_t)tgt * 32 + tid];
    }
    if (tid >= 64 && tid < 64 + U) vs[tid] = a.u[(size_t)tgt * U + (tid - 64)];
    __syncthreads();
    // ---- LSTM update of node tgt: v = [x, h, u] ; gate order ingate, outgate, forgetgate, cellinput ----
    if (tid < 128) {
        const int q = tid >> 5, o = tid & 31;
        const float* Wq = a.Wg[q];
        float g = a.bg[q][o];
        const int V = 64 + U;
        for (int k = 0; k < V; ++k) g = fmaf(vs[k], Wq[k * 32 + o], g);
        gs[q][o] = g;
    }
    __syncthreads();
    if (tid < 32) {
        const float ig = gsig(gs[0][tid]), og = gsig(gs[1][tid]), fg = gsig(gs[2][tid]), cg = tanhf(gs[3][tid]);
        const float c = fg * a.c_in[(size_t)tgt * 32 + tid] + ig * cg;
        a.c_out[(size_t)tgt * 32 + tid] = c;
        a.h_out[(size_t)tgt * 32 + tid] = og * tanhf(c);
    }
}
__global__ __launch_bounds__(256) void gnn_step_kernel(const StepArgs a) { gnn_step_kernel_body(a, (int)blockIdx.x, (int)gridDim.x); }
#ifdef ASEP_ABLATION
__global__ __launch_bounds__(256) void gnn_step_kernel_batch(const GnnBatch<StepArgs> b) {
    if ((int)blockIdx.x >= b.nx[blockIdx.y]) return;
    gnn_step_kernel_body(b.p[blockIdx.y], (int)blockIdx.x, b.nx[blockIdx.y]);
}
#endif


// ------------------------------------------------------------------------------------------------
// The same fused step for WIDE node features (visual branch: 7 geometric + 3 x 16 visual = 55 features, K = 350;
// graph_relation.py:84-139).  W1's A fragments no longer fit the register file (23 chunks x 2 m-tiles x 4 VGPRs), so
// they are staged once per workgroup in LDS ([nch][2][64 lanes] x 16 B = 46 KB at U = 55) and read one chunk ahead of
// their MFMAs; a quad of z is built from ONE 16-byte load of the source node's row (rows are padded to a multiple of
// four floats, u_pad) and one LDS read of the target's row instead of from register copies of whole rows.
// Quad descriptor (4 bytes): kind (GQ_*), offset / 4 inside the row, unused, unused.
// ------------------------------------------------------------------------------------------------
struct StepBigArgs {
    const float* u;        // [N, Upad] zero-padded node features
    const float* h_in; const float* c_in; const float* ef;
    const int* tptr; const int* tsrc; const int* tfirst;
    const gf32x4* A1; const gf32x4* A2;
    const float* b1; const float* b2;
    const float* Wg[4]; const float* bg[4];
    float* h_out; float* c_out;
    int N, U, Upad, Ed, E, nch;
    const unsigned char* qdesc;    // [nch*4][4]
};

__device__ __forceinline__ void gnn_step_big_kernel_body(const StepBigArgs& a, const int bx, const int gx) {
    extern __shared__ __attribute__((aligned(16))) float sm_big[];
    gf32x4* A1s = reinterpret_cast<gf32x4*>(sm_big);                   // [nch][2][64]
    float* trow = sm_big + (size_t)a.nch * 2 * 64 * 4;                  // target row: [Upad] u | [32] h
    __shared__ float xs[4][32];
    __shared__ float gs[4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int tgt = bx;
    const int beg = a.tptr[tgt], end = a.tptr[tgt + 1];
    const int Upad = a.Upad, Ed = a.Ed, nch = a.nch;
    const int ntiles = (end - beg + 15) >> 4;
    if (ntiles > 0)
        for (int i = tid; i < nch * 128; i += 256) A1s[i] = a.A1[i];
    for (int i = tid; i < Upad; i += 256) trow[i] = a.u[(size_t)tgt * Upad + i];
    if (tid < 32) trow[Upad + tid] = a.h_in[(size_t)tgt * 32 + tid];
    __syncthreads();

    gf32x4 xacc[2] = {gf32x4{0.f, 0.f, 0.f, 0.f}, gf32x4{0.f, 0.f, 0.f, 0.f}};
    if (wave < ntiles) {
        gf32x4 A2[2][2];
#pragma unroll
        for (int c = 0; c < 2; ++c) { A2[c][0] = a.A2[(c * 2 + 0) * 64 + lane]; A2[c][1] = a.A2[(c * 2 + 1) * 64 + lane]; }
        const gf32x4 b1v[2] = {*reinterpret_cast<const gf32x4*>(a.b1 + kk * 4), *reinterpret_cast<const gf32x4*>(a.b1 + 16 + kk * 4)};
        const gf32x4 b2v[2] = {*reinterpret_cast<const gf32x4*>(a.b2 + kk * 4), *reinterpret_cast<const gf32x4*>(a.b2 + 16 + kk * 4)};
        const uchar4* qd = reinterpret_cast<const uchar4*>(a.qdesc);
        for (int t = wave; t < ntiles; t += 4) {
            const int e = beg + t * 16 + j;
            const bool valid = e < end;
            const int ec = valid ? e : beg;
            const int src = a.tsrc[ec];
            const int fi = a.tfirst[ec] % a.E;
            const float* urow = a.u + (size_t)src * Upad;
            const float* hrow = a.h_in + (size_t)src * 32;
            gf32x4 efv = gf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q < Ed) efv[q] = a.ef[(size_t)fi * Ed + q];
            // quad fetch: source-side piece from global (L1 / L2), target-side piece from the LDS row
            auto fetch = [&](int c, gf32x4& x0, gf32x4& x1, int& kind) {
                const uchar4 d = qd[c * 4 + kk];
                kind = d.x;
                const int off = 4 * (int)d.y;
                const bool is_h = kind >= GQ_HI;
                const float* p0 = is_h ? hrow + off : urow + off;
                x0 = (kind == GQ_ZERO || kind == GQ_EF) ? efv : *reinterpret_cast<const gf32x4*>(p0);
                x1 = *reinterpret_cast<const gf32x4*>(trow + (is_h ? Upad + off : ((kind == GQ_ZERO || kind == GQ_EF) ? 0 : off)));
            };
            gf32x4 acc1[2] = {gf32x4{0.f, 0.f, 0.f, 0.f}, gf32x4{0.f, 0.f, 0.f, 0.f}};
            gf32x4 x0, x1, a0, a1;
            int kind;
            fetch(0, x0, x1, kind);
            a0 = A1s[lane]; a1 = A1s[64 + lane];
#pragma unroll 1
            for (int c = 0; c < nch; ++c) {
                gf32x4 nx0 = x0, nx1 = x1, na0 = a0, na1 = a1;
                int nkind = kind;
                if (c + 1 < nch) {                                    // next chunk's operands fly under this chunk's MFMAs
                    fetch(c + 1, nx0, nx1, nkind);
                    na0 = A1s[(c + 1) * 128 + lane]; na1 = A1s[(c + 1) * 128 + 64 + lane];
                }
                const gf32x4 dd = x1 - x0;
                const int role = kind == GQ_EF ? 0 : (kind >= GQ_HI ? kind - GQ_HI : kind - GQ_UI);   // 0 i, 1 j, 2 d, 3 d^2
                gf32x4 z = role == 0 ? x0 : (role == 1 ? x1 : (role == 2 ? dd : dd * dd));
                if (kind == GQ_ZERO) z = gf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[r], z[r], acc1[0], 0, 0, 0);
                    acc1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[r], z[r], acc1[1], 0, 0, 0);
                }
                x0 = nx0; x1 = nx1; a0 = na0; a1 = na1; kind = nkind;
            }
            gf32x4 hid[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                hid[m] = acc1[m] + b1v[m];
                hid[m].x = fmaxf(hid[m].x, 0.f); hid[m].y = fmaxf(hid[m].y, 0.f); hid[m].z = fmaxf(hid[m].z, 0.f); hid[m].w = fmaxf(hid[m].w, 0.f);
            }
            gf32x4 acc2[2] = {gf32x4{0.f, 0.f, 0.f, 0.f}, gf32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[c][0][r], hid[c][r], acc2[0], 0, 0, 0);
                    acc2[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[c][1][r], hid[c][r], acc2[1], 0, 0, 0);
                }
            if (valid) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const gf32x4 v = acc2[m] + b2v[m];
                    xacc[m].x += tanhf(v.x); xacc[m].y += tanhf(v.y); xacc[m].z += tanhf(v.z); xacc[m].w += tanhf(v.w);
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = xacc[m][r];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            if (j == 0) xs[wave][16 * m + 4 * kk + r] = v;
        }
    __syncthreads();
    // v = [x (32) | h (32) | u (U)] for the LSTM gates: x into the A1 area (free now), h / u are in trow
    float* xv = sm_big;
    if (tid < 32) {
        const int deg = end - beg;
        const float s = xs[0][tid] + xs[1][tid] + xs[2][tid] + xs[3][tid];
        xv[tid] = deg > 0 ? s / (float)deg : 0.f;
    }
    __syncthreads();
    if (tid < 128) {
        const int q = tid >> 5, o = tid & 31;
        const float* Wq = a.Wg[q];
        float g = a.bg[q][o];
        for (int k = 0; k < 32; ++k) g = fmaf(xv[k], Wq[k * 32 + o], g);
        for (int k = 0; k < 32; ++k) g = fmaf(trow[Upad + k], Wq[(32 + k) * 32 + o], g);
        for (int k = 0; k < a.U; ++k) g = fmaf(trow[k], Wq[(64 + k) * 32 + o], g);
        gs[q][o] = g;
    }
    __syncthreads();
    if (tid < 32) {
        const float ig = gsig(gs[0][tid]), og = gsig(gs[1][tid]), fg = gsig(gs[2][tid]), cg = tanhf(gs[3][tid]);
        const float c = fg * a.c_in[(size_t)tgt * 32 + tid] + ig * cg;
        a.c_out[(size_t)tgt * 32 + tid] = c;
        a.h_out[(size_t)tgt * 32 + tid] = og * tanhf(c);
    }
}
__global__ __launch_bounds__(256) void gnn_step_big_kernel(const StepBigArgs a) { gnn_step_big_kernel_body(a, (int)blockIdx.x, (int)gridDim.x); }
#ifdef ASEP_ABLATION
__global__ __launch_bounds__(256) void gnn_step_big_kernel_batch(const GnnBatch<StepBigArgs> b) {
    if ((int)blockIdx.x >= b.nx[blockIdx.y]) return;
    gnn_step_big_kernel_body(b.p[blockIdx.y], (int)blockIdx.x, b.nx[blockIdx.y]);
}
#endif


// ------------------------------------------------------------------------------------------------
// The FACTORED step (round 6; the default for the reference's widths, any U).  The first layer of the edge MLP is linear in
//   z = [u_f, u_t, u_t - u_f, (u_t - u_f)^2, e, h_f, h_t, h_t - h_f, (h_t - h_f)^2]        (message_fn_chunk.py:266-350; f = from, t = to),
// and with W1's row blocks Wa .. Wi in that order
//   z W1 = u_f (Wa - Wc) + h_f (Wf - Wh)            per SOURCE node:  Pf[f]   (its u half once per page, its h half per step)
//        + u_t (Wb + Wc) + h_t (Wg + Wh) + b1       per TARGET node:  Pt[t]
//        + (u_t - u_f)^2 Wd + e We                  per edge, the same in all T steps:  C[e]   (once per page, CSR-by-target order)
//        + (h_t - h_f)^2 Wi                         per edge and step: K = 32 instead of 4 U + Ed + 128 (350 for the visual nets).
// A step's edge work is then 16 + 16 MFMAs per tile of 16 edges (were 176 + 16 at U = 55) with both layers' fragments in registers (no
// LDS staging of W1: 45 KB per workgroup before), and three 128-byte row gathers per edge (Pf[f], h[f], C[e]; were two rows of 87 floats).
// The pair classifier has used the same factorisation since round 2 (gnn_pair_pre_kernel).  Same one-workgroup-per-target segmented
// sum, no atomics.  Sums are associated differently from the unfactored kernels (which stay in the tree: ASEP_GNN_FACTOR=0): both are held
// to 1e-5 against the fp64 oracle, and the factored filters are formed in double on the host.
// ------------------------------------------------------------------------------------------------
struct FactPreArgs {
    const float* u;        // [N, U]
    const float* ef;       // [E, Ed]
    const int* tptr; const int* tsrc; const int* tfirst;
    const float* Wuu;      // [U][64]: columns 0..31 Wa - Wc, 32..63 Wb + Wc
    const float* Wde;      // [U + Ed][32]: Wd, then We
    const float* b1;
    float* Pu;             // [N][64]: u (Wa - Wc) | u (Wb + Wc) + b1
    float* C;              // [E'][32]
    int N, U, Ed, E;
};

// one workgroup per target node: its Pu row, then C for its in-edges (32 edges x 8 threads x 4 outputs at a time)
__global__ __launch_bounds__(256) void gnn_fact_pre_kernel(const FactPreArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm_pre[];
    float* Ws = sm_pre;                                   // [(U + Ed)][32]
    float* ut = sm_pre + (size_t)(a.U + a.Ed) * 32;       // [U] the target's features
    const int tid = threadIdx.x, tgt = blockIdx.x;
    const int U = a.U, Ed = a.Ed;
    const int beg = a.tptr[tgt], end = a.tptr[tgt + 1];
    for (int i = tid; i < (U + Ed) * 32; i += 256) Ws[i] = a.Wde[i];
    for (int i = tid; i < U; i += 256) ut[i] = a.u[(size_t)tgt * U + i];
    __syncthreads();
    if (tid < 64) {
        float acc = tid >= 32 ? a.b1[tid - 32] : 0.f;
        for (int k = 0; k < U; ++k) acc = fmaf(ut[k], a.Wuu[k * 64 + tid], acc);
        a.Pu[(size_t)tgt * 64 + tid] = acc;
    }
    const int le = tid >> 3, oq = (tid & 7) * 4;
    for (int base = beg; base < end; base += 32) {
        const int e = base + le;
        if (e >= end) continue;
        const float* uf = a.u + (size_t)a.tsrc[e] * U;
        const float* efr = a.ef + (size_t)(a.tfirst[e] % a.E) * Ed;
        gf32x4 acc = gf32x4{0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 8 <= U; k += 8) {                                  // the source row eight values at a time: all eight loads in flight together
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = uf[k + q];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float d = ut[k + q] - v[q];
                acc += (d * d) * *reinterpret_cast<const gf32x4*>(Ws + (k + q) * 32 + oq);
            }
        }
        for (; k < U; ++k) {
            const float d = ut[k] - uf[k];
            acc += (d * d) * *reinterpret_cast<const gf32x4*>(Ws + k * 32 + oq);
        }
        for (int k = 0; k < Ed; ++k) acc += efr[k] * *reinterpret_cast<const gf32x4*>(Ws + (U + k) * 32 + oq);
        *reinterpret_cast<gf32x4*>(a.C + (size_t)e * 32 + oq) = acc;
    }
}

struct StepFactArgs {
    const float* u;        // [N, U] (LSTM input)
    const float* h_in; const float* c_in;
    const int* tptr; const int* tsrc;
    const float* P_in;     // [N][64] this step's Pf | Pt (step 0: Pu itself, h = 0)
    const float* Pu;       // [N][64]
    const float* C;        // [E'][32]
    const gf32x4* A1;      // Wi fragments [2 chunks][2 m-tiles][64 lanes]
    const gf32x4* A2;      // W2 fragments [2][2][64]
    const float* b2;
    const float* Whh;      // [32][64]: columns 0..31 Wf - Wh, 32..63 Wg + Wh
    const float* Wg[4]; const float* bg[4];
    float* h_out; float* c_out;
    float* P_out;          // next step's [N][64], or null behind the last step
    int N, U;
    int first;             // the first step: h and c are zero (not read: no memset in front of the steps)
};

__global__ __launch_bounds__(256) void gnn_step_fact_kernel(const StepFactArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm_fact[];
    float* vs = sm_fact;                                  // [32 x | 32 h | U u] of the target: the LSTM's input row
    __shared__ float xs[4][32];
    __shared__ float gs[4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int tgt = blockIdx.x;
    const int beg = a.tptr[tgt], end = a.tptr[tgt + 1];
    const int U = a.U;
    const int ntiles = (end - beg + 15) >> 4;
    const bool first = a.first != 0;
    if (tid < 32) vs[32 + tid] = first ? 0.f : a.h_in[(size_t)tgt * 32 + tid];
    for (int i = tid; i < U; i += 256) vs[64 + i] = a.u[(size_t)tgt * U + i];

    gf32x4 xacc[2] = {gf32x4{0.f, 0.f, 0.f, 0.f}, gf32x4{0.f, 0.f, 0.f, 0.f}};
    if (wave < ntiles) {
        gf32x4 A1[2][2], A2[2][2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            A1[c][0] = a.A1[(c * 2 + 0) * 64 + lane]; A1[c][1] = a.A1[(c * 2 + 1) * 64 + lane];
            A2[c][0] = a.A2[(c * 2 + 0) * 64 + lane]; A2[c][1] = a.A2[(c * 2 + 1) * 64 + lane];
        }
        const gf32x4 b2v[2] = {*reinterpret_cast<const gf32x4*>(a.b2 + kk * 4), *reinterpret_cast<const gf32x4*>(a.b2 + 16 + kk * 4)};
        const float* htp = a.h_in + (size_t)tgt * 32;
        const gf32x4 zero4 = gf32x4{0.f, 0.f, 0.f, 0.f};
        const gf32x4 ht[2] = {first ? zero4 : *reinterpret_cast<const gf32x4*>(htp + kk * 4), first ? zero4 : *reinterpret_cast<const gf32x4*>(htp + 16 + kk * 4)};
        const float* ptp = a.P_in + (size_t)tgt * 64 + 32;
        const gf32x4 pt[2] = {*reinterpret_cast<const gf32x4*>(ptp + kk * 4), *reinterpret_cast<const gf32x4*>(ptp + 16 + kk * 4)};
        // a tile's three row gathers; the next tile's are requested before this tile's MFMAs
        auto gather = [&](int t, gf32x4 (&hs)[2], gf32x4 (&acc)[2], bool& valid) {
            const int e = beg + t * 16 + j;
            valid = e < end;
            const int ec = valid ? e : beg;                           // padded columns read a real edge, masked later
            const int src = a.tsrc[ec];
            const float* hsp = a.h_in + (size_t)src * 32;
            const float* pfp = a.P_in + (size_t)src * 64;
            const float* cep = a.C + (size_t)ec * 32;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                hs[m] = first ? zero4 : *reinterpret_cast<const gf32x4*>(hsp + 16 * m + kk * 4);
                acc[m] = *reinterpret_cast<const gf32x4*>(pfp + 16 * m + kk * 4) + *reinterpret_cast<const gf32x4*>(cep + 16 * m + kk * 4);
            }
        };
        gf32x4 hs[2], acc1[2];
        bool valid;
        gather(wave, hs, acc1, valid);
        for (int t = wave; t < ntiles; t += 4) {
            gf32x4 nhs[2] = {hs[0], hs[1]}, nacc[2] = {acc1[0], acc1[1]};
            bool nvalid = false;
            if (t + 4 < ntiles) gather(t + 4, nhs, nacc, nvalid);
#pragma unroll
            for (int m = 0; m < 2; ++m) acc1[m] += pt[m];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const gf32x4 dd = ht[c] - hs[c];
                const gf32x4 z = dd * dd;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[c][0][r], z[r], acc1[0], 0, 0, 0);
                    acc1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[c][1][r], z[r], acc1[1], 0, 0, 0);
                }
            }
            gf32x4 hid[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                hid[m] = acc1[m];
                hid[m].x = fmaxf(hid[m].x, 0.f); hid[m].y = fmaxf(hid[m].y, 0.f); hid[m].z = fmaxf(hid[m].z, 0.f); hid[m].w = fmaxf(hid[m].w, 0.f);
            }
            gf32x4 acc2[2] = {b2v[0], b2v[1]};
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[c][0][r], hid[c][r], acc2[0], 0, 0, 0);
                    acc2[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[c][1][r], hid[c][r], acc2[1], 0, 0, 0);
                }
            if (valid) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    xacc[m].x += tanhf(acc2[m].x); xacc[m].y += tanhf(acc2[m].y); xacc[m].z += tanhf(acc2[m].z); xacc[m].w += tanhf(acc2[m].w);
                }
            }
            hs[0] = nhs[0]; hs[1] = nhs[1]; acc1[0] = nacc[0]; acc1[1] = nacc[1]; valid = nvalid;
        }
    }
    // ---- sum over the 16 edge columns (lanes j), then over the 4 waves ----
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = xacc[m][r];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            if (j == 0) xs[wave][16 * m + 4 * kk + r] = v;
        }
    __syncthreads();
    if (tid < 32) {
        const int deg = end - beg;
        const float s = xs[0][tid] + xs[1][tid] + xs[2][tid] + xs[3][tid];
        vs[tid] = deg > 0 ? s / (float)deg : 0.f;                    // a_ij = 1 / indeg(j)
    }
    __syncthreads();
    // ---- LSTM update of node tgt: v = [x, h, u]; gate order ingate, outgate, forgetgate, cellinput (update_fn_lstm.py:41-76) ----
    if (tid < 128) {
        const int q = tid >> 5, o = tid & 31;
        const float* Wq = a.Wg[q];
        float g = a.bg[q][o];
        const int V = 64 + U;
        for (int k = 0; k < V; ++k) g = fmaf(vs[k], Wq[k * 32 + o], g);
        gs[q][o] = g;
    }
    __syncthreads();
    if (tid < 32) {
        const float ig = gsig(gs[0][tid]), og = gsig(gs[1][tid]), fg = gsig(gs[2][tid]), cg = tanhf(gs[3][tid]);
        const float c = (first ? 0.f : fg * a.c_in[(size_t)tgt * 32 + tid]) + ig * cg;
        const float hn = og * tanhf(c);
        a.c_out[(size_t)tgt * 32 + tid] = c;
        a.h_out[(size_t)tgt * 32 + tid] = hn;
        xs[0][tid] = hn;
    }
    if (!a.P_out) return;
    __syncthreads();
    // ---- the node's rows of the NEXT step: Pf = u (Wa - Wc) + h' (Wf - Wh),  Pt = u (Wb + Wc) + b1 + h' (Wg + Wh) ----
    if (tid < 64) {
        float acc = a.Pu[(size_t)tgt * 64 + tid];
#pragma unroll 8
        for (int k = 0; k < 32; ++k) acc = fmaf(xs[0][k], a.Whh[k * 64 + tid], acc);
        a.P_out[(size_t)tgt * 64 + tid] = acc;
    }
}

// zero-padded copy of the node features: [N, U] -> [N, Upad]
__global__ __launch_bounds__(256) void gnn_pad_rows_kernel(const float* __restrict__ src, int N, int U, float* __restrict__ dst, int Upad) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * Upad) return;
    const int n = i / Upad, k = i - n * Upad;
    dst[i] = k < U ? src[(size_t)n * U + k] : 0.f;
}

// ------------------------------------------------------------------------------------------------
// Generic-width fallbacks (hidden_node_feature_dim / interaction_feature_dim / hidden layer of the edge MLP other
// than 32; message_fn_chunk.py:13-40): plain FMA loops, one workgroup per target node / node.  Correct for any
// width; the MFMA kernels above are the path for the reference's default widths.
// ------------------------------------------------------------------------------------------------
// An MLP of the reference (layers.py:468-490 mlp): hidden layers fully_connected_layer_h<i> (ReLU) + fully_connected_logit_layer_out.
// num_hidden_units_interaction_fct / num_hidden_units_attention_fct (message_fn_chunk.py:24,40) and the classifier's
// num_hidden_units (graph_relation.py:196) are LISTS: up to GNN_MLP_MAX hidden layers are served here.
constexpr int GNN_MLP_MAX = 4;
struct MlpW {
    const float* W[GNN_MLP_MAX + 1];    // dense layer l: [dims[l], dims[l + 1]], row = input unit
    const float* b[GNN_MLP_MAX + 1];
    int dims[GNN_MLP_MAX + 2];          // dims[0] = input width, dims[nl] = output width
    int nl;                             // dense layers = hidden layers + 1
};
// the hidden layers on the LDS vector `in` (all 256 threads of the block): relu(W x + b) layer by layer through two scratch
// vectors; -> the last hidden activations (width dims[nl - 1]); ends behind a barrier
__device__ __forceinline__ const float* mlp_hidden(const MlpW& m, const float* in, float* bufA, float* bufB, int tid) {
    const float* cur = in;
    for (int l = 0; l + 1 < m.nl; ++l) {
        float* out = (l & 1) ? bufB : bufA;
        const int din = m.dims[l], dout = m.dims[l + 1];
        for (int o = tid; o < dout; o += 256) {
            float s = m.b[l][o];
            for (int k = 0; k < din; ++k) s = fmaf(cur[k], m.W[l][(size_t)k * dout + o], s);
            out[o] = fmaxf(s, 0.f);
        }
        __syncthreads();
        cur = out;
    }
    return cur;
}
// z of one interaction (message_fn_chunk.py:313-350: u_from, u_to, u_diff, u_sq | edge | h_from, h_to, h_diff, h_sq) -> LDS
__device__ __forceinline__ void build_z(float* z, const float* ui, const float* uj, const float* hi, const float* hj, const float* efr,
                                        int U, int Ed, int H, int tid) {
    for (int k = tid; k < U; k += 256) {
        const float vi = ui[k], vj = uj[k], d = vj - vi;
        z[k] = vi; z[U + k] = vj; z[2 * U + k] = d; z[3 * U + k] = d * d;
    }
    for (int k = tid; k < Ed; k += 256) z[4 * U + k] = efr[k];
    for (int k = tid; k < H; k += 256) {
        const float vi = hi[k], vj = hj[k], d = vj - vi;
        float* zh = z + 4 * U + Ed;
        zh[k] = vi; zh[H + k] = vj; zh[2 * H + k] = d; zh[3 * H + k] = d * d;
    }
}

struct MsgGenArgs {
    const float* u; const float* h; const float* ef;
    const int* tptr; const int* tsrc; const int* tfirst;
    MlpW mlp;                           // [K] -> hidden ... -> [I], tanh output
    float* x;                           // [N, I]
    int N, U, Ed, E, H, I;
    int maxh;                           // widest hidden layer (size of each scratch vector)
    int agg_max;                        // message_fn_chunk.py:16,57-62 aggregation_type: 0 = 'sum' (tf.sparse.reduce_sum), 1 = 'max'
};

// Balanced neighbour weighting (no attention): every interaction is scaled by 1 / indegree(target) and the scaled features are
// aggregated over the target's in-edges -- 'sum': their sum; 'max' (tf.sparse.reduce_max over the stored entries only: a target
// without in-edges gets 0, a target with in-edges may get a negative value): max_e (m_e / deg) = (max_e m_e) / deg, deg > 0.
__global__ __launch_bounds__(256) void gnn_message_generic_kernel(const MsgGenArgs a) {
    extern __shared__ float smg[];
    const int K = 4 * a.U + a.Ed + 4 * a.H;
    float* z = smg;                 // K
    float* bufA = z + K;            // maxh
    float* bufB = bufA + a.maxh;    // maxh
    float* acc = bufB + a.maxh;     // I
    const int tid = threadIdx.x, tgt = blockIdx.x;
    const int beg = a.tptr[tgt], end = a.tptr[tgt + 1];
    for (int o = tid; o < a.I; o += 256) acc[o] = a.agg_max ? -INFINITY : 0.f;
    const float* uj = a.u + (size_t)tgt * a.U;
    const float* hj = a.h + (size_t)tgt * a.H;
    const int nl = a.mlp.nl, dlast = a.mlp.dims[nl - 1];
    for (int e = beg; e < end; ++e) {
        const int i = a.tsrc[e];
        __syncthreads();
        build_z(z, a.u + (size_t)i * a.U, uj, a.h + (size_t)i * a.H, hj, a.ef + (size_t)(a.tfirst[e] % a.E) * a.Ed, a.U, a.Ed, a.H, tid);
        __syncthreads();
        const float* hid = mlp_hidden(a.mlp, z, bufA, bufB, tid);
        for (int o = tid; o < a.I; o += 256) {
            float s = a.mlp.b[nl - 1][o];
            for (int k = 0; k < dlast; ++k) s = fmaf(hid[k], a.mlp.W[nl - 1][(size_t)k * a.I + o], s);
            const float m = tanhf(s);
            acc[o] = a.agg_max ? fmaxf(acc[o], m) : acc[o] + m;
        }
    }
    __syncthreads();
    const int deg = end - beg;
    for (int o = tid; o < a.I; o += 256) a.x[(size_t)tgt * a.I + o] = deg > 0 ? acc[o] / (float)deg : 0.f;
}

// ------------------------------------------------------------------------------------------------
// Attention-weighted aggregation (message_fn_chunk.py:35-41,167-245,420-454; use_attention = True): per head k an interaction MLP
// (tanh output, x_dim = interaction_dim / heads for 'concat', interaction_dim for 'average') and an attention MLP (one output, no
// output activation) over the same z; the unnormalised values are put into a sparse [from, to] tensor, TRANSPOSED, soft-maxed
// over its rows (= over the in-edges of a target) and read back with `.values` -- i.e. in (to, from) order -- and multiplied
// element by element with the interaction features, which are in (from, to) order: interaction e gets the e-th soft-max value of
// the (to, from)-sorted list.  That pairing is restated literally here (for an undirected graph it is the attention of the
// REVERSE edge): csr position p <-> (to, from) order, edge index e <-> (from, to) order, eidx[p] = the edge at csr position p.
// Three launches per transition step, one workgroup per target node, plain FMA loops (these nets are the exception).
// ------------------------------------------------------------------------------------------------
// eidx[p] of csr entry p = (tsrc[p] -> t): binary search of t in row tsrc[p] of the (from, to)-sorted list
__global__ __launch_bounds__(64) void edge_rank_kernel(const int* __restrict__ colptr, const int* __restrict__ tsrc, const int* __restrict__ rowptr,
                                                      const int32_t* __restrict__ sorted_edges, int N, int* __restrict__ eidx) {
    const int t = blockIdx.x;
    for (int p = colptr[t] + threadIdx.x; p < colptr[t + 1]; p += 64) {
        const int s = tsrc[p];
        int lo = rowptr[s], hi = rowptr[s + 1] - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sorted_edges[2 * mid + 1] < t) lo = mid + 1; else hi = mid;
        }
        eidx[p] = lo;
    }
}

// Interaction chunks (message_fn_chunk.py:76-110): the reference runs the message function once per chunk of S = max(1, 100000 / N)
// target nodes on the interactions that end in the chunk.  Inside a chunk the value pairing above is between the chunk's
// (to, from)-sorted list -- csr positions tptr[c S] ... -- and its (from, to)-sorted list -- the chunk's edges in edge-index order.
// widx[e] = tptr[c S] + (number of edges e' < e that end in chunk c): the csr position whose soft-max value edge e is multiplied
// with (one chunk: widx[e] = e).  One workgroup per chunk scans the edge list.
__global__ __launch_bounds__(256) void edge_chunk_rank_kernel(const int32_t* __restrict__ sorted_edges, const int* __restrict__ rowptr,
                                                            const int* __restrict__ colptr, int N, int S, int* __restrict__ widx) {
    __shared__ int wsum[4];
    __shared__ int carry;
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Etot = rowptr[N], lo = c * S, hi = min(N, lo + S);
    if (tid == 0) carry = colptr[lo];
    __syncthreads();
    for (int e0 = 0; e0 < Etot; e0 += 256) {
        const int e = e0 + tid;
        const int t = e < Etot ? sorted_edges[2 * e + 1] : -1;
        const bool in = t >= lo && t < hi;
        const unsigned long long m = __ballot(in);
        const int before = __popcll(m & (lane == 0 ? 0ull : (~0ull >> (64 - lane))));
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int base = carry;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        if (in) widx[e] = base + before;
        __syncthreads();
        if (tid == 0) carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
}

constexpr int GNN_MAX_HEADS = 8;
struct AttHeadW { MlpW inter, att; };     // interaction MLP ([K] -> ... -> [xd], tanh) and attention MLP ([K] -> ... -> [1], linear)
struct MsgAttArgs {
    const float* u; const float* h; const float* ef;
    const int* tptr; const int* tsrc; const int* tfirst; const int* eidx;
    const AttHeadW* hd;  // [heads] in device memory (eight heads of two five-layer MLPs do not fit the kernel-argument segment)
    float* M;            // [E', heads * xd] interaction features by edge index e
    float* A;            // [heads][E'] unnormalised attention values by edge index e
    int N, U, Ed, E, H, xd, heads, Etot;
    int maxh;            // widest hidden layer of any of the MLPs
};
// per target t, per in-edge p: z, then for every head m = tanh(MLP_int(z)) -> M[eidx[p]], a = MLP_att(z) -> A[head][eidx[p]]
__global__ __launch_bounds__(256) void gnn_msg_att_kernel(const MsgAttArgs a) {
    extern __shared__ float smg[];
    const int K = 4 * a.U + a.Ed + 4 * a.H;
    float* z = smg;                 // K
    float* bufA = z + K;            // maxh
    float* bufB = bufA + a.maxh;    // maxh
    const int tid = threadIdx.x, tgt = blockIdx.x;
    const int beg = a.tptr[tgt], end = a.tptr[tgt + 1];
    const float* uj = a.u + (size_t)tgt * a.U;
    const float* hj = a.h + (size_t)tgt * a.H;
    for (int p = beg; p < end; ++p) {
        const int i = a.tsrc[p], e = a.eidx[p];
        __syncthreads();
        build_z(z, a.u + (size_t)i * a.U, uj, a.h + (size_t)i * a.H, hj, a.ef + (size_t)(a.tfirst[p] % a.E) * a.Ed, a.U, a.Ed, a.H, tid);
        __syncthreads();
        for (int hdi = 0; hdi < a.heads; ++hdi) {
            const MlpW& wi = a.hd[hdi].inter;
            const MlpW& wa = a.hd[hdi].att;
            const float* hid = mlp_hidden(wi, z, bufA, bufB, tid);
            const int li = wi.nl - 1, di = wi.dims[li];
            for (int o = tid; o < a.xd; o += 256) {
                float s = wi.b[li][o];
                for (int k = 0; k < di; ++k) s = fmaf(hid[k], wi.W[li][(size_t)k * a.xd + o], s);
                a.M[(size_t)e * a.heads * a.xd + hdi * a.xd + o] = tanhf(s);
            }
            __syncthreads();
            hid = mlp_hidden(wa, z, bufA, bufB, tid);
            const int la = wa.nl - 1, da = wa.dims[la];
            if (tid == 0) {
                float s = wa.b[la][0];
                for (int k = 0; k < da; ++k) s = fmaf(hid[k], wa.W[la][k], s);
                a.A[(size_t)hdi * a.Etot + e] = s;
            }
            __syncthreads();
        }
    }
}
// tf.sparse.softmax over the rows of the transposed tensor: per target t and head, over its csr entries p -> S[head][p]
__global__ __launch_bounds__(64) void gnn_att_softmax_kernel(const int* __restrict__ tptr, const int* __restrict__ eidx, const float* __restrict__ A,
                                                            int heads, int Etot, float* __restrict__ S) {
    const int t = blockIdx.x, lane = threadIdx.x;
    const int beg = tptr[t], end = tptr[t + 1];
    for (int hdi = 0; hdi < heads; ++hdi) {
        const float* Ah = A + (size_t)hdi * Etot;
        float mx = -INFINITY;
        for (int p = beg + lane; p < end; p += 64) mx = fmaxf(mx, Ah[eidx[p]]);
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float den = 0.f;
        for (int p = beg + lane; p < end; p += 64) den += expf(Ah[eidx[p]] - mx);
        for (int o = 32; o > 0; o >>= 1) den += __shfl_xor(den, o);
        for (int p = beg + lane; p < end; p += 64) S[(size_t)hdi * Etot + p] = expf(Ah[eidx[p]] - mx) / den;
    }
}
// x[t] = merge over heads of sum over in-edges e of S[head][widx[e]] * M[e][head]  -- the pairing above (widx[e] = e in the one-chunk case);
// merge 'concat' (heads * xd = I columns) or 'average' (xd = I, mean over heads)
__global__ __launch_bounds__(64) void gnn_att_aggregate_kernel(const int* __restrict__ tptr, const int* __restrict__ eidx, const int* __restrict__ widx,
                                                              const float* __restrict__ S, const float* __restrict__ M, int heads, int xd, int Etot,
                                                              int average, int agg_max, float* __restrict__ x) {
    const int t = blockIdx.x;
    const int beg = tptr[t], end = tptr[t + 1];
    const int I = average ? xd : heads * xd;
    // agg_max (aggregation_type 'max'): the maximum of the attenuated features over the in-edges instead of their sum; 0 without in-edges
    auto head_col = [&](int hdi, int col) {
        if (agg_max) {
            float m = -INFINITY;
            for (int p = beg; p < end; ++p) { const int e = eidx[p]; m = fmaxf(m, S[(size_t)hdi * Etot + widx[e]] * M[(size_t)e * heads * xd + col]); }
            return end > beg ? m : 0.f;
        }
        float s = 0.f;
        for (int p = beg; p < end; ++p) { const int e = eidx[p]; s = fmaf(S[(size_t)hdi * Etot + widx[e]], M[(size_t)e * heads * xd + col], s); }
        return s;
    };
    for (int o = threadIdx.x; o < I; o += 64) {
        float acc = 0.f;
        if (average) {
            for (int hdi = 0; hdi < heads; ++hdi) acc += head_col(hdi, hdi * xd + o);
            acc /= (float)heads;
        } else {
            acc = head_col(o / xd, o);
        }
        x[(size_t)t * I + o] = acc;
    }
}

struct LstmGenArgs {
    const float* x; const float* h_in; const float* c_in; const float* u;
    const float* Wg[4]; const float* bg[4];     // [I + (use_h ? H : 0) + (use_u ? U : 0), H], [H]
    float* h_out; float* c_out;
    int N, U, H, I;
    int use_h, use_u;    // update_fn_lstm.py:13-16,43-50: incorporate_hidden_features_in_update / incorporate_node_input_features_in_update
};

__global__ __launch_bounds__(256) void gnn_lstm_generic_kernel(const LstmGenArgs a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.N * a.H) return;
    const int node = idx / a.H, o = idx - node * a.H;
    float g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] = a.bg[q][o];
    const float* xr = a.x + (size_t)node * a.I;
    const float* hr = a.h_in + (size_t)node * a.H;
    const float* ur = a.u + (size_t)node * a.U;
    for (int k = 0; k < a.I; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = fmaf(xr[k], a.Wg[q][(size_t)k * a.H + o], g[q]);
    int row = a.I;
    if (a.use_h) {
        for (int k = 0; k < a.H; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] = fmaf(hr[k], a.Wg[q][(size_t)(row + k) * a.H + o], g[q]);
        row += a.H;
    }
    if (a.use_u)
        for (int k = 0; k < a.U; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] = fmaf(ur[k], a.Wg[q][(size_t)(row + k) * a.H + o], g[q]);
    const float ig = gsig(g[0]), og = gsig(g[1]), fg = gsig(g[2]), cg = tanhf(g[3]);
    const float c = fg * a.c_in[idx] + ig * cg;
    a.c_out[idx] = c;
    a.h_out[idx] = og * tanhf(c);
}

// ------------------------------------------------------------------------------------------------
// LSTM update (update_fn_lstm.py:55-76): v = [x, h, u]; four dense gates; c = f*c + i*g; h = o*tanh(c)
// one thread per (node, unit); gate order in Wg/bg: ingate, outgate, forgetgate, cellinput
// ------------------------------------------------------------------------------------------------
struct LstmArgs {
    const float* x; const float* h_in; const float* c_in; const float* u;
    const float* Wg[4]; const float* bg[4];     // [V,32], [32]
    float* h_out; float* c_out;
    int N, U;
};

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

__global__ __launch_bounds__(256) void gnn_lstm_kernel(const LstmArgs a) {
    const int tid = threadIdx.x, o = tid & 31;
    const int node = blockIdx.x * 8 + (tid >> 5);
    if (node >= a.N) return;
    float g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] = a.bg[q][o];
    const float* xr = a.x + (size_t)node * 32;
    const float* hr = a.h_in + (size_t)node * 32;
    const float* ur = a.u + (size_t)node * a.U;
    for (int k = 0; k < 32; ++k) {
        const float v = xr[k];
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = fmaf(v, a.Wg[q][k * 32 + o], g[q]);
    }
    for (int k = 0; k < 32; ++k) {
        const float v = hr[k];
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = fmaf(v, a.Wg[q][(32 + k) * 32 + o], g[q]);
    }
    for (int k = 0; k < a.U; ++k) {
        const float v = ur[k];
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = fmaf(v, a.Wg[q][(64 + k) * 32 + o], g[q]);
    }
    const float ig = sigmoidf_(g[0]), og = sigmoidf_(g[1]), fg = sigmoidf_(g[2]), cg = tanhf(g[3]);
    const float c = fg * a.c_in[(size_t)node * 32 + o] + ig * cg;
    a.c_out[(size_t)node * 32 + o] = c;
    a.h_out[(size_t)node * 32 + o] = og * tanhf(c);
}

// ------------------------------------------------------------------------------------------------
// pair classifier (graph_relation.py:253-266): logits = MLP([h_a || h_b]).  The first layer is linear in
// the concatenation, so it is evaluated once per node:  P[a] = h_a . W1[0:32],  Q[b] = h_b . W1[32:64]
// (stored transposed [H1][N] for coalesced per-pair reads); per pair: relu(P+Q+b1) -> H2 -> classes.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gnn_pair_pre_kernel(const float* __restrict__ h, int N, int H, const float* __restrict__ W1,
                                                           int H1, float* __restrict__ Pt, float* __restrict__ Qt) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N * H1; i += gridDim.x * 256) {
        const int n = i % N, k = i / N;
        float p = 0.f, q = 0.f;
        for (int d = 0; d < H; ++d) {
            const float hv = h[(size_t)n * H + d];
            p = fmaf(hv, W1[d * H1 + k], p);
            q = fmaf(hv, W1[(H + d) * H1 + k], q);
        }
        Pt[(size_t)k * N + n] = p;
        Qt[(size_t)k * N + n] = q;
    }
}

struct PairPreArgs {
    const float* h; int N, H; const float* W1; int H1; float* Pt; float* Qt;
};
#ifdef ASEP_ABLATION
// gnn_pair_pre_kernel for several pages (GnnBatch): the same grid-stride loop per page, nx[page] blocks
__global__ __launch_bounds__(256) void gnn_pair_pre_kernel_batch(const GnnBatch<PairPreArgs> b) {
    if ((int)blockIdx.x >= b.nx[blockIdx.y]) return;
    const PairPreArgs& a = b.p[blockIdx.y];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.N * a.H1; i += b.nx[blockIdx.y] * 256) {
        const int n = i % a.N, k = i / a.N;
        float p = 0.f, q = 0.f;
        for (int d = 0; d < a.H; ++d) {
            const float hv = a.h[(size_t)n * a.H + d];
            p = fmaf(hv, a.W1[d * a.H1 + k], p);
            q = fmaf(hv, a.W1[(a.H + d) * a.H1 + k], q);
        }
        a.Pt[(size_t)k * a.N + n] = p;
        a.Qt[(size_t)k * a.N + n] = q;
    }
}
#endif

struct PairArgs {
    const float* Pt; const float* Qt;   // [H1][N]
    const float* b1;                    // [H1]
    const float* W2; const float* b2;   // [H1,H2], [H2]
    const float* W3; const float* b3;   // [H2,NC], [NC]
    const int32_t* rel;                 // [R,2] or nullptr = all ordered pairs row major
    float* out;                         // [R,NC]
    int N, R;
};

template <int H1, int H2, int NC>
__device__ __forceinline__ void gnn_pair_cls_kernel_body(const PairArgs& a, const int bx, const int gx) {
    __shared__ __attribute__((aligned(16))) float W2s[H1 * H2];
    __shared__ float misc[H1 + H2 + H2 * NC + NC];
    float* b1s = misc; float* b2s = b1s + H1; float* W3s = b2s + H2; float* b3s = W3s + H2 * NC;
    const int tid = threadIdx.x;
    for (int i = tid; i < H1 * H2; i += 256) W2s[i] = a.W2[i];
    for (int i = tid; i < H1; i += 256) b1s[i] = a.b1[i];
    for (int i = tid; i < H2; i += 256) b2s[i] = a.b2[i];
    for (int i = tid; i < H2 * NC; i += 256) W3s[i] = a.W3[i];
    for (int i = tid; i < NC; i += 256) b3s[i] = a.b3[i];
    __syncthreads();
    const int r = bx * 256 + tid;
    if (r >= a.R) return;
    int na, nb;
    if (a.rel) { na = a.rel[2 * r]; nb = a.rel[2 * r + 1]; }
    else { na = r / a.N; nb = r - na * a.N; }
    if ((unsigned)na >= (unsigned)a.N || (unsigned)nb >= (unsigned)a.N) {      // a pair that names no node: visible, not UB
#pragma unroll
        for (int c = 0; c < NC; ++c) a.out[(size_t)r * NC + c] = __builtin_nanf("");
        return;
    }
    float acc[H2];
#pragma unroll
    for (int k = 0; k < H2; ++k) acc[k] = b2s[k];
    for (int d = 0; d < H1; ++d) {
        const float v = fmaxf(a.Pt[(size_t)d * a.N + na] + a.Qt[(size_t)d * a.N + nb] + b1s[d], 0.f);
#pragma unroll
        for (int k = 0; k < H2; ++k) acc[k] = fmaf(v, W2s[d * H2 + k], acc[k]);
    }
    float lg[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) lg[c] = b3s[c];
#pragma unroll
    for (int k = 0; k < H2; ++k) {
        const float v = fmaxf(acc[k], 0.f);
#pragma unroll
        for (int c = 0; c < NC; ++c) lg[c] = fmaf(v, W3s[k * NC + c], lg[c]);
    }
    float mx = lg[0];
#pragma unroll
    for (int c = 1; c < NC; ++c) mx = fmaxf(mx, lg[c]);
    float den = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) { lg[c] = expf(lg[c] - mx); den += lg[c]; }
#pragma unroll
    for (int c = 0; c < NC; ++c) a.out[(size_t)r * NC + c] = lg[c] / den;
}
template <int H1, int H2, int NC>
__global__ __launch_bounds__(256) void gnn_pair_cls_kernel(const PairArgs a) { gnn_pair_cls_kernel_body<H1, H2, NC>(a, (int)blockIdx.x, (int)gridDim.x); }
#ifdef ASEP_ABLATION
template <int H1, int H2, int NC>
__global__ __launch_bounds__(256) void gnn_pair_cls_kernel_batch(const GnnBatch<PairArgs> b) {
    if ((int)blockIdx.x >= b.nx[blockIdx.y]) return;
    gnn_pair_cls_kernel_body<H1, H2, NC>(b.p[blockIdx.y], (int)blockIdx.x, b.nx[blockIdx.y]);
}
#endif


// any classifier widths (trainer_rel.py:17 num_hidden_units is a free parameter): one thread per pair, the second
// layer recomputes the first layer's activations per output unit (L1-resident P / Q columns) -- a fallback
struct PairGenArgs {
    const float* Pt; const float* Qt;   // [h1][N]: the first hidden layer's two halves, evaluated per node (gnn_pair_pre_kernel)
    const float* b1;                    // [h1]
    MlpW rest;                          // the layers behind the first hidden one: [h1] -> ... -> [NC] (nl >= 1)
    const int32_t* rel;                 // [R,2] or nullptr = all ordered pairs row major
    float* out;                         // [R,NC]
    int N, R, maxw;                     // maxw = widest layer (row pitch of the two scratch tiles)
};
constexpr int PAIRG_P = 32;             // pairs per block; 8 threads per pair

// any number of hidden layers of any width (graph_relation.py:196 num_hidden_units is a free list): 32 pairs per block, the
// activations of a layer in an LDS tile [pair][unit], 8 threads of a pair share the units of the next layer -- a fallback
__global__ __launch_bounds__(256) void gnn_pair_cls_generic_kernel(const PairGenArgs a) {
    extern __shared__ float spg[];
    float* cur = spg;                                   // [PAIRG_P][maxw]
    float* nxt = spg + PAIRG_P * a.maxw;
    const int tid = threadIdx.x, pl = tid >> 3, sub = tid & 7;
    const int r = blockIdx.x * PAIRG_P + pl;
    const int NC = a.rest.dims[a.rest.nl];
    int na = -1, nb = -1;
    if (r < a.R) {
        if (a.rel) { na = a.rel[2 * r]; nb = a.rel[2 * r + 1]; }
        else { na = r / a.N; nb = r - na * a.N; }
    }
    const bool valid = r < a.R && (unsigned)na < (unsigned)a.N && (unsigned)nb < (unsigned)a.N;
    const int h1 = a.rest.dims[0];
    for (int d = sub; d < h1; d += 8)
        cur[pl * a.maxw + d] = valid ? fmaxf(a.Pt[(size_t)d * a.N + na] + a.Qt[(size_t)d * a.N + nb] + a.b1[d], 0.f) : 0.f;
    __syncthreads();
    for (int l = 0; l < a.rest.nl; ++l) {
        const int din = a.rest.dims[l], dout = a.rest.dims[l + 1];
        const bool last = l == a.rest.nl - 1;
        for (int o = sub; o < dout; o += 8) {
            float s = a.rest.b[l][o];
            for (int k = 0; k < din; ++k) s = fmaf(cur[pl * a.maxw + k], a.rest.W[l][(size_t)k * dout + o], s);
            nxt[pl * a.maxw + o] = last ? s : fmaxf(s, 0.f);
        }
        __syncthreads();
        float* t = cur; cur = nxt; nxt = t;
    }
    if (sub == 0 && r < a.R) {
        if (!valid) {                                   // a pair that names no node: visible, not UB
            for (int c = 0; c < NC; ++c) a.out[(size_t)r * NC + c] = __builtin_nanf("");
            return;
        }
        const float* lg = cur + pl * a.maxw;
        float mx = lg[0];
        for (int c = 1; c < NC; ++c) mx = fmaxf(mx, lg[c]);
        float den = 0.f;
        for (int c = 0; c < NC; ++c) den += expf(lg[c] - mx);
        for (int c = 0; c < NC; ++c) a.out[(size_t)r * NC + c] = expf(lg[c] - mx) / den;
    }
}


// ---------------------------------------------------------------------------------------------------------------
// visual node features (graph_relation.py:84-139, misc.py:282-381): per node the paraxial bounding rectangle of
// its region (relative coordinates) -> floor-scaled ROI on a backbone feature map -> per-channel max over the ROI
// -> ff(ReLU) compression -> columns [col0, col0 + d) of the node feature matrix.
// ---------------------------------------------------------------------------------------------------------------
struct RoiArgs {
    const float* fm;          // [fh, fw, C] NHWC; fp32, or bf16 (2 bytes per value) for gnn_roi_compress_kernel<true>
    int fh, fw, C;
    const float* regions;     // [N, 2, P]: row 0 = x, row 1 = y, relative to the image size
    int P;
    const int32_t* npts;      // [N]
    const float* Wc;          // [C, d]
    const float* bc;          // [d]
    int d;
    float* u_out;             // [N, ustride]
    int ustride, col0;
    float* vmax_out;          // optional [N, C] (tests)
};

template <bool BF>
__device__ __forceinline__ void gnn_roi_compress_kernel_body(const RoiArgs& a, const int bx, const int gx) {
    // value i of the map (a bf16 value widens exactly; the maximum of a region is therefore the same value the fp32 form of a
    // bf16-rounded map would give)
    auto fmv = [&](size_t i) -> float {
        if constexpr (BF) return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(a.fm)[i] << 16);
        else return a.fm[i];
    };
    __shared__ float red[256];
    __shared__ float vmax[256];
    const int n = bx, tid = threadIdx.x;
    const int np = min(a.npts[n], a.P);
    float xmin = 0.f, xmax = 0.f, ymin = 0.f, ymax = 0.f;     // misc.py:503-508: no points -> zeros
    if (np > 0) {
        const float* rx = a.regions + (size_t)n * 2 * a.P;
        const float* ry = rx + a.P;
        xmin = xmax = rx[0];
        ymin = ymax = ry[0];
        for (int i = 1; i < np; ++i) {
            xmin = fminf(xmin, rx[i]); xmax = fmaxf(xmax, rx[i]);
            ymin = fminf(ymin, ry[i]); ymax = fmaxf(ymax, ry[i]);
        }
    }
    // misc.py:322-337: floor(rel * size) clamped into the map; at least one cell
    const int x0 = max(min((int)floorf(xmin * (float)a.fw), a.fw - 1), 0);
    const int x1 = max(min((int)floorf(xmax * (float)a.fw), a.fw - 1), 0);
    const int y0 = max(min((int)floorf(ymin * (float)a.fh), a.fh - 1), 0);
    const int y1 = max(min((int)floorf(ymax * (float)a.fh), a.fh - 1), 0);
    const int nx = max(x1 - x0 + 1, 1), ny = max(y1 - y0 + 1, 1);
    const int C = a.C;
    float m = -INFINITY;
    constexpr int VEC = BF ? 8 : 4;                            // values per 16-byte load
    const int groups = C / VEC;                                // channel groups of VEC: a lane's group is lane % groups
    if (C % VEC == 0 && groups <= 8 && (groups & (groups - 1)) == 0 && ((size_t)a.fm & 15) == 0) {
        // 16-byte loads: a thread keeps VEC consecutive channels (the same ones for every vector it visits, because its vector
        // index advances by 256 and 256 * VEC is a multiple of C); maxima are exact, so the result is that of the scalar form
        float mv[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) mv[k] = -INFINITY;
        const int rowvec = nx * C / VEC;
        for (int y = 0; y < ny; ++y) {
            const uint4* row = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.fm) +
                                                              ((size_t)(y0 + y) * a.fw + x0) * C * (BF ? 2 : 4));
            for (int i = tid; i < rowvec; i += 256) {
                const uint4 q = row[i];
                if constexpr (BF) {
                    const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        mv[2 * k] = fmaxf(mv[2 * k], __uint_as_float(w[k] << 16));
                        mv[2 * k + 1] = fmaxf(mv[2 * k + 1], __uint_as_float(w[k] & 0xffff0000u));
                    }
                } else {
                    mv[0] = fmaxf(mv[0], __uint_as_float(q.x)); mv[1] = fmaxf(mv[1], __uint_as_float(q.y));
                    mv[2] = fmaxf(mv[2], __uint_as_float(q.z)); mv[3] = fmaxf(mv[3], __uint_as_float(q.w));
                }
            }
        }
        // lanes of one group first (xor distances that are multiples of `groups`), then the four waves through 1 KB of LDS:
        // the kernel keeps its LDS footprint small enough to share a CU with the page net's 155 KB blocks
        __shared__ float wred[4][8][VEC];
        for (int off = groups; off < 64; off <<= 1)
#pragma unroll
            for (int k = 0; k < VEC; ++k) mv[k] = fmaxf(mv[k], __shfl_xor(mv[k], off));
        if ((tid & 63) < groups)
#pragma unroll
            for (int k = 0; k < VEC; ++k) wred[tid >> 6][tid & 63][k] = mv[k];
        __syncthreads();
        if (tid < C) {
            const int g = tid / VEC, slot = tid % VEC;
            vmax[tid] = fmaxf(fmaxf(wred[0][g][slot], wred[1][g][slot]), fmaxf(wred[2][g][slot], wred[3][g][slot]));
        }
    } else if (256 % C == 0) {
        // a thread keeps one channel: rows of nx*C contiguous floats are read coalesced
        const int rowlen = nx * C;
        for (int y = 0; y < ny; ++y) {
            const size_t row = ((size_t)(y0 + y) * a.fw + x0) * C;
            for (int i = tid; i < rowlen; i += 256) m = fmaxf(m, fmv(row + i));
        }
        red[tid] = m;
        __syncthreads();
        if (tid < C) {
            float v = red[tid];
            for (int k = tid + C; k < 256; k += C) v = fmaxf(v, red[k]);
            vmax[tid] = v;
        }
    } else {
        for (int c = tid; c < C; c += 256) {
            float v = -INFINITY;
            for (int y = 0; y < ny; ++y)
                for (int x = 0; x < nx; ++x) v = fmaxf(v, fmv(((size_t)(y0 + y) * a.fw + x0 + x) * C + c));
            vmax[c] = v;
        }
    }
    __syncthreads();
    if (a.vmax_out && tid < C) a.vmax_out[(size_t)n * C + tid] = vmax[tid];
    for (int j = tid; j < a.d; j += 256) {
        float acc = a.bc[j];
        for (int c = 0; c < C; ++c) acc = fmaf(vmax[c], a.Wc[(size_t)c * a.d + j], acc);
        a.u_out[(size_t)n * a.ustride + a.col0 + j] = fmaxf(acc, 0.f);
    }
}
template <bool BF>
__global__ __launch_bounds__(256) void gnn_roi_compress_kernel(const RoiArgs a) { gnn_roi_compress_kernel_body<BF>(a, (int)blockIdx.x, (int)gridDim.x); }
#ifdef ASEP_ABLATION
template <bool BF>
__global__ __launch_bounds__(256) void gnn_roi_compress_kernel_batch(const GnnBatch<RoiArgs> b) {
    if ((int)blockIdx.x >= b.nx[blockIdx.y]) return;
    gnn_roi_compress_kernel_body<BF>(b.p[blockIdx.y], (int)blockIdx.x, b.nx[blockIdx.y]);
}
#endif


// graph_gnn.py:102-109 compress_node_feature_dim: y[n][d] = tanh(b[d] + sum_k x[n][k] W[k][d])  (layers.ff_layer with tanh)
__global__ void __launch_bounds__(256)
gnn_compress_kernel(const float* __restrict__ x, int N, int K, const float* __restrict__ W, const float* __restrict__ b, int D,
                    float* __restrict__ y) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * D) return;
    const int n = i / D, d = i - n * D;
    float s = b[d];
    for (int k = 0; k < K; ++k) s = fmaf(x[(size_t)n * K + k], W[(size_t)k * D + d], s);
    y[i] = tanhf(s);
}

// graph_gnn.py:158-166 output_type: 1 = h[n][d] += sum_k x[n][k] W[k][d] (ff_layer without bias / activation), written to y [N, D];
// 2 = y[n] = [h[n] | x[n]] (width D + K)
__global__ void __launch_bounds__(256)
gnn_output_type_kernel(const float* __restrict__ h, const float* __restrict__ x, int N, int D, int K, const float* __restrict__ W,
                       int mode, float* __restrict__ y) {
    const int wid = mode == 1 ? D : D + K;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * wid) return;
    const int n = i / wid, d = i - n * wid;
    if (mode == 1) {
        float s = 0.f;
        for (int k = 0; k < K; ++k) s = fmaf(x[(size_t)n * K + k], W[(size_t)k * D + d], s);
        y[i] = h[(size_t)n * D + d] + s;
    } else {
        y[i] = d < D ? h[(size_t)n * D + d] : x[(size_t)n * K + d - D];
    }
}

// copies the geometric node features into the first `ug` columns of the concatenated node feature matrix
__global__ void __launch_bounds__(256)
gnn_copy_cols_kernel(const float* __restrict__ src, int N, int ug, float* __restrict__ dst, int ustride) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * ug) return;
    dst[(size_t)(i / ug) * ustride + i % ug] = src[i];
}

}  // namespace asep
