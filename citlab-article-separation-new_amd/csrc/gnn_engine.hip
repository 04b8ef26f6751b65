// GNN relation-predictor engine (host side).  Entry points: include/asep_hip.h.
// Schedule of one page (batch size 1, graph_gnn.py:46-167 + graph_relation.py:194-203):
//   [visual branch: backbone -> ROI max -> compression -> concat]
//   edge correction -> T x (message, LSTM update) -> per-node first classifier layer -> per-pair classifier.
// Three step kernels, chosen at load time:
//   STEP_SMALL  gnn_step_kernel      widths 32/32/32, U <= 8, Ed <= 4: W1 fragments in registers (the 7-feature nets)
//   STEP_BIG    gnn_step_big_kernel  widths 32/32/32, U <= 120, Ed <= 4: W1 fragments in LDS (visual nets, U = 55)
//   STEP_GENERIC  gnn_message_generic_kernel + gnn_lstm_generic_kernel: any widths (plain FMA loops)
// Round 6: for the widths of STEP_SMALL / STEP_BIG the default is the FACTORED step (gnn_fact_pre_kernel once per page +
// gnn_step_fact_kernel per step: per-node terms of the edge MLP's first layer per node, K = 32 per edge and step); ASEP_GNN_FACTOR=0
// keeps the unfactored kernels (the tests compare the two).
#include <algorithm>
#include <memory>

#include "asep_common.h"
#include "gnn_kernels.h"

using namespace asep;

enum { STEP_GENERIC = 0, STEP_SMALL = 1, STEP_BIG = 2, STEP_FACT = 3 };

struct asep_gnn {
    asep_gnn_cfg cfg{};
    int U = 0, Ed = 0, H = 0, I = 0, Hm = 0;   // node / edge feature widths, hidden, interaction, MLP hidden
    int K = 0, V = 0;                 // message-MLP input width, LSTM input width
    float *b1 = nullptr, *b2 = nullptr;          // biases of the default one-hidden-layer edge MLP (fused step kernels)
    MlpW msg{};                       // the interaction MLP (no attention): [K] -> hidden ... -> [I]
    int msg_maxh = 0;                 // widest hidden layer of the interaction / attention MLPs
    float* Wg[4] = {nullptr, nullptr, nullptr, nullptr};
    float* bg[4] = {nullptr, nullptr, nullptr, nullptr};
    float *C1 = nullptr, *cb1 = nullptr, *C2 = nullptr, *cb2 = nullptr, *C3 = nullptr, *cb3 = nullptr;
    MlpW cls_rest{};                  // classifier layers behind the first hidden one: [cls_hidden1] -> ... -> [num_classes]
    int cls_maxw = 0;
    bool cls_fast = false;            // 64, 32 -> 2: gnn_pair_cls_kernel<64,32,2>
    int Iout = 0;                     // width of x (message_fn_chunk.py:229-241): heads * x_dim for 'concat', interaction_dim otherwise
    int Ed_fed = 0;                   // width of the edge features as FED (Ed minus the visual edge dims)
    // fused MFMA step kernels: permuted W1 / W2 fragments + quad descriptors
    float *A1 = nullptr, *A2 = nullptr;
    unsigned char* qdesc = nullptr;
    int nch = 0, Upad = 0;
    int mode = STEP_GENERIC;
    size_t big_lds = 0;
    bool use_step = true;             // ASEP_GNN_STEP=0 selects the separate (generic) message / LSTM kernels
    bool use_fact = true;             // ASEP_GNN_FACTOR=0: the unfactored fused step kernels (STEP_SMALL / STEP_BIG)
    float *Wuu = nullptr, *Wde = nullptr, *Whh = nullptr, *A1f = nullptr;   // factored first layer (gnn_kernels.h, "The FACTORED step")
    size_t fact_pre_lds = 0, fact_lds = 0;
    std::vector<void*> owned;
    BufferPool pool;                  // buffers of one forward (requested in a fixed order)
    BufferPool vis_pool;              // buffers of the visual stage in front of it
    BufferPool host_stage;            // device staging of the host-pointer entry points (grow-only)
    // state of the last forward
    int N = 0;
    float* d_h = nullptr;
    int* d_rowptr = nullptr;
    hipStream_t stream = nullptr;
    // visual branch (graph_relation.py:84-139): backbone + per-map compression layers
    // graph_gnn.py:102-109 compress_node_feature_dim: fed features [N, Uin] -> tanh(Wc x + bc) [N, U]; Uin == U when off
    int Uin = 0;
    float* Wc = nullptr;
    float* bc = nullptr;
    std::map<std::string, HostTensor> vis_blob;      // visual_node_feature_compression_fm_<i>/dense/{weights,bias}
    asep_aru* backbone = nullptr;                    // not owned
    std::vector<std::string> vis_names;
    std::vector<float*> vis_W, vis_b;                // owned through vis_owned
    std::vector<void*> vis_owned;
    std::vector<int> vis_C, vis_d;
    int vis_total = 0;
    float* Wout = nullptr;                           // output_type 1: GraphLSTM1/dense/weights [Uin, H]
    AttHeadW att_w[GNN_MAX_HEADS] = {};              // attention_heads > 0: per-head interaction + attention MLPs
    AttHeadW* d_att_w = nullptr;                     // the same in device memory (kernel argument of gnn_msg_att_kernel)
    int att_xd = 0;                                  // interaction width per head
    // visual EDGE features (graph_relation.py:141-172): compression layers per feature map, concatenated behind the fed edge features
    std::vector<float*> vise_W, vise_b;
    std::vector<int> vise_d;
    int vise_total = 0;
    float* d_ef_cat = nullptr;                       // [E, Ed] per page of the last visual forward (own buffer)
    size_t ef_cat_cap = 0;
    float* d_u_cat = nullptr;                        // concatenated node features of the last visual forward (own buffer)
    size_t u_cat_cap = 0;
    // Page lanes of the batch entry point (ASEP_GNN_LANES, default 1 = everything on the caller's stream): the graph part of a page
    // is a chain of ~16 launches of at most N workgroups each (N = 200: less than one workgroup per CU), so the pages of a batch CAN
    // be dealt over several streams, each with its own arena, that fork behind the grouped backbone forward and join the caller's
    // stream at the end.  Measured in round 4 (profiles/README.md, r4b): slower -- 412.2 -> 403.1 pages/s (bf16), 121.3 -> 119.9
    // (fp32) with 4 lanes, the same with 8: the graph chains already run beside the page net's kernels on their own stream, and
    // four of them at once take more issue slots from those kernels than the shorter chain gives back.  Kept as a switch.
    struct PageLane {
        hipStream_t s = nullptr;
        hipEvent_t done = nullptr;
        BufferPool pool;
        ~PageLane() {
            if (done) (void)hipEventDestroy(done);
            if (s) (void)hipStreamDestroy(s);
        }
    };
    std::vector<std::unique_ptr<PageLane>> page_lanes;
    hipEvent_t ev_fork = nullptr;
    int n_page_lanes = 1;
    bool batch_graph = false;         // ASEP_GNN_BATCH=1: the pages of asep_gnn_forward_visual_batch_dev stage by stage (forward_batch_impl: ROI kernels, steps,
                                      // classifier as ONE launch over the pages) instead of one after the other.  Measured (profiles/r4_ab, r4k): 240 -> 90 launches
                                      // per 16-page call, bit-identical results, step rate unchanged (fp32 120.5 -> 120.8, bf16 419.3 -> 417.9 pages/s): the graph
                                      // chains already hide behind the page net on their own stream -- off by default
    void free_visual() {
        for (void* p : vis_owned)
            if (p) (void)hipFree(p);
        vis_owned.clear(); vis_W.clear(); vis_b.clear(); vis_names.clear(); vis_C.clear(); vis_d.clear();
        vise_W.clear(); vise_b.clear(); vise_d.clear();
        vis_total = 0; vise_total = 0;
        backbone = nullptr;
    }
    ~asep_gnn() {
        free_visual();
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (d_u_cat) (void)hipFree(d_u_cat);
        if (d_ef_cat) (void)hipFree(d_ef_cat);
        for (void* p : owned)
            if (p) (void)hipFree(p);
    }
};

namespace {

const char* MSG = "GraphLSTM1/message_fn_default/head_0/calculation_interaction_features/concat_u_and_h/interaction_features";
const char* UPD = "GraphLSTM1/update_function_LSTM";
const char* CLS = "Classification/logits";

int upload_named(std::vector<void*>& owner, const std::map<std::string, HostTensor>& blob, const std::string& name,
                 std::vector<int> dims, float** d) {
    auto it = blob.find(name);
    if (it == blob.end()) { set_error("weights: missing tensor %s", name.c_str()); return ASEP_ERR_WEIGHTS; }
    if (it->second.dims != dims) {
        std::string want, got;
        for (int v : dims) want += std::to_string(v) + ",";
        for (int v : it->second.dims) got += std::to_string(v) + ",";
        set_error("weights: %s has shape [%s] expected [%s]", name.c_str(), got.c_str(), want.c_str());
        return ASEP_ERR_WEIGHTS;
    }
    const auto& h = it->second.data;
    ASEP_HIP_CHECK(hipMalloc((void**)d, std::max<size_t>(h.size(), 4) * sizeof(float)));
    owner.push_back(*d);
    ASEP_HIP_CHECK(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return ASEP_OK;
}

template <typename T>
int upload_vec(asep_gnn* g, const std::vector<T>& v, T** d) {
    ASEP_HIP_CHECK(hipMalloc((void**)d, std::max<size_t>(v.size(), 4) * sizeof(T)));
    g->owned.push_back(*d);
    ASEP_HIP_CHECK(hipMemcpy(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return ASEP_OK;
}

// An MLP of layers.py:468-490 under `scope`: fully_connected_layer_h<i> for the hidden widths, fully_connected_logit_layer_out
int upload_mlp(asep_gnn* g, const std::map<std::string, HostTensor>& blob, const std::string& scope, int d_in,
               const std::vector<int>& hidden, int d_out, MlpW* m, int first_layer = 0) {
    // first_layer = 1: the first hidden layer is held elsewhere (classifier: evaluated per node); dims[0] = its width
    if ((int)hidden.size() > GNN_MLP_MAX) { set_error("%s: %zu hidden layers, at most %d are served", scope.c_str(), hidden.size(), GNN_MLP_MAX); return ASEP_ERR_UNSUPPORTED; }
    *m = MlpW{};
    int cur = d_in, l = 0;
    for (size_t i = 0; i < hidden.size(); ++i) {
        if ((int)i >= first_layer) {
            const std::string ly = scope + "/fully_connected_layer_h" + std::to_string(i + 1);
            float *W = nullptr, *b = nullptr;
            int rc = upload_named(g->owned, blob, ly + "/weights", {cur, hidden[i]}, &W);
            if (!rc) rc = upload_named(g->owned, blob, ly + "/bias", {hidden[i]}, &b);
            if (rc) return rc;
            m->W[l] = W; m->b[l] = b; m->dims[l] = cur; ++l;
        }
        cur = hidden[i];
    }
    float *W = nullptr, *b = nullptr;
    int rc = upload_named(g->owned, blob, scope + "/fully_connected_logit_layer_out/weights", {cur, d_out}, &W);
    if (!rc) rc = upload_named(g->owned, blob, scope + "/fully_connected_logit_layer_out/bias", {d_out}, &b);
    if (rc) return rc;
    m->W[l] = W; m->b[l] = b; m->dims[l] = cur; ++l;
    m->dims[l] = d_out;
    m->nl = l;
    return ASEP_OK;
}

std::vector<int> hidden_list(std::initializer_list<int> v) {      // the non-zero prefix of a cfg's hidden-width fields
    std::vector<int> out;
    for (int x : v) {
        if (x <= 0) break;
        out.push_back(x);
    }
    return out;
}

// K axis of the edge MLP permuted into quads of four consecutive features (gnn_kernels.h): fragments + descriptors
struct Quad { int type, off; };

int pack_step_fragments(asep_gnn* g, const std::map<std::string, HostTensor>& blob, bool big) {
    const int U = g->U, Ed = g->Ed;
    const std::string m = MSG;
    std::vector<Quad> quads;
    const int uq = (U + 3) / 4;
    for (int t : {GQ_UI, GQ_UJ, GQ_DU, GQ_DU2})
        for (int q = 0; q < uq; ++q) quads.push_back({t, 4 * q});
    if (Ed > 0) quads.push_back({GQ_EF, 0});
    while (quads.size() % 4) quads.push_back({GQ_ZERO, 0});
    // H-type quads come as whole chunks: lane group k4 owns h[16c' + 4 k4 ..]
    for (int t : {GQ_HI, GQ_HJ, GQ_DH, GQ_DH2})
        for (int c = 0; c < 2; ++c)
            for (int k4 = 0; k4 < 4; ++k4) quads.push_back({t, big ? 16 * c + 4 * k4 : 16 * c});
    g->nch = (int)quads.size() / 4;
    const HostTensor& W1h = blob.find(m + "/fully_connected_layer_h1/weights")->second;            // [K,32]
    const HostTensor& W2h = blob.find(m + "/fully_connected_logit_layer_out/weights")->second;     // [32,32]
    auto w1row = [&](const Quad& q, int k4, int r) -> int {              // original row of W1 or -1 (zero)
        const int hoff = big ? q.off + r : q.off + 4 * k4 + r;
        switch (q.type) {
            case GQ_UI: return q.off + r < U ? 0 * U + q.off + r : -1;
            case GQ_UJ: return q.off + r < U ? 1 * U + q.off + r : -1;
            case GQ_DU: return q.off + r < U ? 2 * U + q.off + r : -1;
            case GQ_DU2: return q.off + r < U ? 3 * U + q.off + r : -1;
            case GQ_EF: return r < Ed ? 4 * U + r : -1;
            case GQ_HI: return 4 * U + Ed + 0 * 32 + hoff;
            case GQ_HJ: return 4 * U + Ed + 1 * 32 + hoff;
            case GQ_DH: return 4 * U + Ed + 2 * 32 + hoff;
            case GQ_DH2: return 4 * U + Ed + 3 * 32 + hoff;
            default: return -1;
        }
    };
    std::vector<float> a1((size_t)g->nch * 2 * 64 * 4), a2((size_t)2 * 2 * 64 * 4);
    const int dstride = big ? 4 : 2;
    std::vector<unsigned char> qd((size_t)g->nch * 4 * dstride, 0);
    for (int c = 0; c < g->nch; ++c)
        for (int k4 = 0; k4 < 4; ++k4) {
            const Quad& q = quads[c * 4 + k4];
            qd[(c * 4 + k4) * dstride] = (unsigned char)q.type;
            qd[(c * 4 + k4) * dstride + 1] = (unsigned char)(big ? q.off / 4 : q.off);
            for (int mt = 0; mt < 2; ++mt)
                for (int i = 0; i < 16; ++i)
                    for (int r = 0; r < 4; ++r) {
                        const int row = w1row(q, k4, r), lane = k4 * 16 + i;
                        a1[(((size_t)c * 2 + mt) * 64 + lane) * 4 + r] = row < 0 ? 0.f : W1h.data[(size_t)row * GNN_H + 16 * mt + i];
                    }
        }
    for (int c = 0; c < 2; ++c)
        for (int mt = 0; mt < 2; ++mt)
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r)
                    a2[(((size_t)c * 2 + mt) * 64 + lane) * 4 + r] = W2h.data[(size_t)(16 * c + 4 * (lane >> 4) + r) * GNN_H + 16 * mt + (lane & 15)];
    int rc = upload_vec(g, a1, &g->A1);
    if (!rc) rc = upload_vec(g, a2, &g->A2);
    if (!rc) rc = upload_vec(g, qd, &g->qdesc);
    return rc;
}

// The factored first layer of the edge MLP (gnn_kernels.h): W1's row blocks [Wa Wb Wc Wd | We | Wf Wg Wh Wi] combined in double
int pack_fact(asep_gnn* g, const std::map<std::string, HostTensor>& blob) {
    const int U = g->U, Ed = g->Ed;
    const std::string m = MSG;
    const HostTensor& W1 = blob.find(m + "/fully_connected_layer_h1/weights")->second;            // [K,32]
    auto w = [&](int row, int o) -> double { return (double)W1.data[(size_t)row * GNN_H + o]; };
    const int ha = 4 * U + Ed;                                   // first row of Wf
    std::vector<float> wuu((size_t)std::max(U, 1) * 64), wde((size_t)std::max(U + Ed, 1) * 32), whh((size_t)32 * 64), a1((size_t)2 * 2 * 64 * 4);
    for (int k = 0; k < U; ++k)
        for (int o = 0; o < 32; ++o) {
            wuu[(size_t)k * 64 + o] = (float)(w(k, o) - w(2 * U + k, o));                  // Wa - Wc
            wuu[(size_t)k * 64 + 32 + o] = (float)(w(U + k, o) + w(2 * U + k, o));         // Wb + Wc
            wde[(size_t)k * 32 + o] = (float)w(3 * U + k, o);                              // Wd
        }
    for (int k = 0; k < Ed; ++k)
        for (int o = 0; o < 32; ++o) wde[(size_t)(U + k) * 32 + o] = (float)w(4 * U + k, o);   // We
    for (int k = 0; k < 32; ++k)
        for (int o = 0; o < 32; ++o) {
            whh[(size_t)k * 64 + o] = (float)(w(ha + k, o) - w(ha + 64 + k, o));           // Wf - Wh
            whh[(size_t)k * 64 + 32 + o] = (float)(w(ha + 32 + k, o) + w(ha + 64 + k, o)); // Wg + Wh
        }
    for (int c = 0; c < 2; ++c)                                   // Wi as A fragments: slot 16 c + 4 kk + r, unit 16 mt + i
        for (int mt = 0; mt < 2; ++mt)
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r)
                    a1[(((size_t)c * 2 + mt) * 64 + lane) * 4 + r] = (float)w(ha + 96 + 16 * c + 4 * (lane >> 4) + r, 16 * mt + (lane & 15));
    int rc = upload_vec(g, wuu, &g->Wuu);
    if (!rc) rc = upload_vec(g, wde, &g->Wde);
    if (!rc) rc = upload_vec(g, whh, &g->Whh);
    if (!rc) rc = upload_vec(g, a1, &g->A1f);
    return rc;
}

struct EdgeBufs {
    int* table; int* rowcnt; int* colcnt; int* rowptr; int* colptr;
    int32_t* sorted; int* sfirst; int* tsrc; int* tfirst;
};

// misc.py:7-151 on the device; fills the CSR-by-target used by the message kernel
int correct_edges_dev(asep_gnn* g, BufferPool& pool, int N, int E, const int32_t* d_edges, EdgeBufs& eb, hipStream_t s) {
    const int und = g->cfg.undirected_graph ? 1 : 0;
    const size_t maxE = (size_t)(und ? 2 : 1) * std::max(E, 1);
    eb.table = (int*)pool.get((size_t)N * N * sizeof(int));
    eb.rowcnt = (int*)pool.get((size_t)N * sizeof(int));
    eb.colcnt = (int*)pool.get((size_t)N * sizeof(int));
    eb.rowptr = (int*)pool.get((size_t)(N + 1) * sizeof(int));
    eb.colptr = (int*)pool.get((size_t)(N + 1) * sizeof(int));
    eb.sorted = (int32_t*)pool.get(maxE * 2 * sizeof(int32_t));
    eb.sfirst = (int*)pool.get(maxE * sizeof(int));
    eb.tsrc = (int*)pool.get(maxE * sizeof(int));
    eb.tfirst = (int*)pool.get(maxE * sizeof(int));
    ASEP_HIP_CHECK(hipMemsetAsync(eb.table, 0x7f, (size_t)N * N * sizeof(int), s));
    if (E > 0) {
        const int total = und ? 2 * E : E;
        hipLaunchKernelGGL(edge_table_kernel, dim3(std::min(cdiv(total, 256), 2048)), dim3(256), 0, s, d_edges, E, N, und,
                           eb.table);
    }
    hipLaunchKernelGGL(edge_count_kernel, dim3(N), dim3(64), 0, s, eb.table, N, eb.rowcnt, eb.colcnt);
    hipLaunchKernelGGL(edge_scan_kernel, dim3(1), dim3(1024), 0, s, eb.rowcnt, eb.colcnt, N, eb.rowptr, eb.colptr);
    hipLaunchKernelGGL(edge_emit_kernel, dim3(N), dim3(64), 0, s, eb.table, N, eb.rowptr, eb.colptr, eb.sorted,
                       eb.sfirst, eb.tsrc, eb.tfirst);
    ASEP_HIP_CHECK(hipGetLastError());
    g->d_rowptr = eb.rowptr;
    return ASEP_OK;
}

// the step kernel of this model under the switches: the factored form wherever a fused step kernel serves the widths
inline int step_mode_of(const asep_gnn* g) {
    if (!g->use_step || g->mode == STEP_GENERIC) return STEP_GENERIC;
    return g->use_fact && g->A1f ? STEP_FACT : g->mode;
}

int forward_impl(asep_gnn* g, BufferPool& pool, int N, int E, const int32_t* d_edges, const float* d_u, const float* d_ef, int R,
                 const int32_t* d_rel, float* d_out, hipStream_t s) {
    const asep_gnn_cfg& c = g->cfg;
    if (N < 1 || E < 0 || R < 0) { set_error("asep_gnn_forward: bad sizes N=%d E=%d R=%d", N, E, R); return ASEP_ERR_ARG; }
    if ((size_t)N * N > (size_t)1 << 30) { set_error("asep_gnn_forward: N=%d too large for the dense edge table", N); return ASEP_ERR_UNSUPPORTED; }
    g->stream = s;
    g->N = N;
    pool.begin();
    EdgeBufs eb{};
    int rc = correct_edges_dev(g, pool, N, E, d_edges, eb, s);
    if (rc) return rc;
    const int H = g->H, I = g->I;
    const size_t nh = (size_t)N * H;
    float* h[2] = {(float*)pool.get(nh * 4), (float*)pool.get(nh * 4)};
    float* cs[2] = {(float*)pool.get(nh * 4), (float*)pool.get(nh * 4)};
    float* x = (float*)pool.get((size_t)N * std::max(I, g->Iout) * 4);
    float* Pt = (float*)pool.get((size_t)N * c.cls_hidden1 * 4);
    float* Qt = (float*)pool.get((size_t)N * c.cls_hidden1 * 4);
    const float* d_fed = d_u;                                // node features as fed (output_type add / concat read these)
    if (g->Wc) {                                             // compress_node_feature_dim: the fed features -> tanh(Wc x + bc)
        float* uc = (float*)pool.get((size_t)N * g->U * 4);
        hipLaunchKernelGGL(gnn_compress_kernel, dim3(cdiv(N * g->U, 256)), dim3(256), 0, s, d_u, N, g->Uin, g->Wc, g->bc, g->U, uc);
        d_u = uc;
    }
    // attention-weighted aggregation: edge index of every csr entry + per-edge interaction features / attention values
    int* att_eidx = nullptr;
    int* att_widx = nullptr;
    float *att_M = nullptr, *att_A = nullptr, *att_S = nullptr;
    const size_t att_maxE = (size_t)(c.undirected_graph ? 2 : 1) * std::max(E, 1);
    if (c.attention_heads > 0) {
        att_eidx = (int*)pool.get(att_maxE * sizeof(int));
        att_M = (float*)pool.get(att_maxE * c.attention_heads * g->att_xd * sizeof(float));
        att_A = (float*)pool.get(att_maxE * c.attention_heads * sizeof(float));
        att_S = (float*)pool.get(att_maxE * c.attention_heads * sizeof(float));
        att_widx = (int*)pool.get(att_maxE * sizeof(int));
        hipLaunchKernelGGL(edge_rank_kernel, dim3(N), dim3(64), 0, s, eb.colptr, eb.tsrc, eb.rowptr, eb.sorted, N, att_eidx);
        const int chunk_nodes = std::max(1, 100000 / N);      // message_fn_chunk.py:77-78
        hipLaunchKernelGGL(edge_chunk_rank_kernel, dim3(cdiv(N, chunk_nodes)), dim3(256), 0, s, eb.sorted, eb.rowptr, eb.colptr, N, chunk_nodes,
                           att_widx);
    }
    float* upad = nullptr;
    const int mode = step_mode_of(g);
    if (mode == STEP_BIG) {
        upad = (float*)pool.get((size_t)N * g->Upad * 4);
        hipLaunchKernelGGL(gnn_pad_rows_kernel, dim3(cdiv(N * g->Upad, 256)), dim3(256), 0, s, d_u, N, g->U, upad, g->Upad);
    }
    float *fPu = nullptr, *fP[2] = {nullptr, nullptr}, *fC = nullptr;
    if (mode == STEP_FACT && c.num_transition_steps > 0) {     // the step-independent parts of the edge MLP's first layer, once per page
        fPu = (float*)pool.get((size_t)N * 64 * 4);
        fP[0] = (float*)pool.get((size_t)N * 64 * 4);
        fP[1] = (float*)pool.get((size_t)N * 64 * 4);
        fC = (float*)pool.get(att_maxE * 32 * 4);
        FactPreArgs fa{};
        fa.u = d_u; fa.ef = d_ef; fa.tptr = eb.colptr; fa.tsrc = eb.tsrc; fa.tfirst = eb.tfirst;
        fa.Wuu = g->Wuu; fa.Wde = g->Wde; fa.b1 = g->b1; fa.Pu = fPu; fa.C = fC;
        fa.N = N; fa.U = g->U; fa.Ed = g->Ed; fa.E = std::max(E, 1);
        hipLaunchKernelGGL(gnn_fact_pre_kernel, dim3(N), dim3(256), g->fact_pre_lds, s, fa);
    }
    if (mode != STEP_FACT || c.num_transition_steps == 0) {  // (the factored step's first launch does not read h / c: no memsets)
        ASEP_HIP_CHECK(hipMemsetAsync(h[0], 0, nh * 4, s));
        ASEP_HIP_CHECK(hipMemsetAsync(cs[0], 0, nh * 4, s));
    }
    int cur = 0;
    for (int t = 0; t < c.num_transition_steps; ++t) {
        if (mode == STEP_FACT) {
            StepFactArgs sa{};
            sa.u = d_u; sa.h_in = h[cur]; sa.c_in = cs[cur]; sa.tptr = eb.colptr; sa.tsrc = eb.tsrc;
            sa.P_in = t == 0 ? fPu : fP[t & 1];                // h = 0 in front of the first step: the rows are their u halves
            sa.Pu = fPu; sa.C = fC;
            sa.A1 = (const gf32x4*)g->A1f; sa.A2 = (const gf32x4*)g->A2; sa.b2 = g->b2; sa.Whh = g->Whh;
            for (int q = 0; q < 4; ++q) { sa.Wg[q] = g->Wg[q]; sa.bg[q] = g->bg[q]; }
            sa.h_out = h[cur ^ 1]; sa.c_out = cs[cur ^ 1];
            sa.P_out = t + 1 < c.num_transition_steps ? fP[(t + 1) & 1] : nullptr;
            sa.N = N; sa.U = g->U; sa.first = t == 0;
            hipLaunchKernelGGL(gnn_step_fact_kernel, dim3(N), dim3(256), g->fact_lds, s, sa);
        } else if (mode == STEP_SMALL) {
            StepArgs sa{};
            sa.u = d_u; sa.h_in = h[cur]; sa.c_in = cs[cur]; sa.ef = d_ef;
            sa.tptr = eb.colptr; sa.tsrc = eb.tsrc; sa.tfirst = eb.tfirst;
            sa.A1 = (const gf32x4*)g->A1; sa.A2 = (const gf32x4*)g->A2; sa.b1 = g->b1; sa.b2 = g->b2;
            for (int q = 0; q < 4; ++q) { sa.Wg[q] = g->Wg[q]; sa.bg[q] = g->bg[q]; }
            sa.h_out = h[cur ^ 1]; sa.c_out = cs[cur ^ 1];
            sa.N = N; sa.U = g->U; sa.Ed = g->Ed; sa.E = std::max(E, 1); sa.nch = g->nch;
            sa.qdesc = g->qdesc;
            hipLaunchKernelGGL(gnn_step_kernel, dim3(N), dim3(256), 0, s, sa);
        } else if (mode == STEP_BIG) {
            StepBigArgs sa{};
            sa.u = upad; sa.h_in = h[cur]; sa.c_in = cs[cur]; sa.ef = d_ef;
            sa.tptr = eb.colptr; sa.tsrc = eb.tsrc; sa.tfirst = eb.tfirst;
            sa.A1 = (const gf32x4*)g->A1; sa.A2 = (const gf32x4*)g->A2; sa.b1 = g->b1; sa.b2 = g->b2;
            for (int q = 0; q < 4; ++q) { sa.Wg[q] = g->Wg[q]; sa.bg[q] = g->bg[q]; }
            sa.h_out = h[cur ^ 1]; sa.c_out = cs[cur ^ 1];
            sa.N = N; sa.U = g->U; sa.Upad = g->Upad; sa.Ed = g->Ed; sa.E = std::max(E, 1); sa.nch = g->nch;
            sa.qdesc = g->qdesc;
            hipLaunchKernelGGL(gnn_step_big_kernel, dim3(N), dim3(256), g->big_lds, s, sa);
        } else if (c.attention_heads > 0) {
            MsgAttArgs aa{};
            aa.u = d_u; aa.h = h[cur]; aa.ef = d_ef; aa.tptr = eb.colptr; aa.tsrc = eb.tsrc; aa.tfirst = eb.tfirst; aa.eidx = att_eidx;
            aa.hd = g->d_att_w;
            aa.M = att_M; aa.A = att_A;
            aa.N = N; aa.U = g->U; aa.Ed = g->Ed; aa.E = std::max(E, 1); aa.H = H;
            aa.xd = g->att_xd; aa.heads = c.attention_heads; aa.Etot = (int)att_maxE; aa.maxh = g->msg_maxh;
            const size_t lds = ((size_t)g->K + 2 * (size_t)g->msg_maxh) * sizeof(float);
            hipLaunchKernelGGL(gnn_msg_att_kernel, dim3(N), dim3(256), lds, s, aa);
            hipLaunchKernelGGL(gnn_att_softmax_kernel, dim3(N), dim3(64), 0, s, eb.colptr, att_eidx, att_A, c.attention_heads, (int)att_maxE, att_S);
            hipLaunchKernelGGL(gnn_att_aggregate_kernel, dim3(N), dim3(64), 0, s, eb.colptr, att_eidx, att_widx, att_S, att_M, c.attention_heads,
                               g->att_xd, (int)att_maxE, c.attention_merge, c.aggregation_type, x);
        } else {
            MsgGenArgs ma{};
            ma.u = d_u; ma.h = h[cur]; ma.ef = d_ef; ma.tptr = eb.colptr; ma.tsrc = eb.tsrc; ma.tfirst = eb.tfirst;
            ma.mlp = g->msg; ma.x = x;
            ma.N = N; ma.U = g->U; ma.Ed = g->Ed; ma.E = std::max(E, 1); ma.H = H; ma.I = I; ma.maxh = g->msg_maxh;
            ma.agg_max = c.aggregation_type;
            const size_t lds = ((size_t)g->K + 2 * (size_t)g->msg_maxh + I) * sizeof(float);
            hipLaunchKernelGGL(gnn_message_generic_kernel, dim3(N), dim3(256), lds, s, ma);
        }
        if (mode == STEP_GENERIC) {                          // (the fused step kernels contain the LSTM update)
            LstmGenArgs la{};
            la.x = x; la.h_in = h[cur]; la.c_in = cs[cur]; la.u = d_u;
            for (int q = 0; q < 4; ++q) { la.Wg[q] = g->Wg[q]; la.bg[q] = g->bg[q]; }
            la.h_out = h[cur ^ 1]; la.c_out = cs[cur ^ 1]; la.N = N; la.U = g->U; la.H = H; la.I = g->Iout;
            la.use_h = c.lstm_use_hidden; la.use_u = c.lstm_use_input;
            hipLaunchKernelGGL(gnn_lstm_generic_kernel, dim3(cdiv(N * H, 256)), dim3(256), 0, s, la);
        }
        cur ^= 1;
    }
    g->d_h = h[cur];
    const float* hcls = h[cur];                              // what the pair classifier reads, and its width
    int Hc = H;
    if (c.output_type != 0) {                                // graph_gnn.py:158-166
        Hc = c.output_type == 2 ? H + g->Uin : H;
        float* y = (float*)pool.get((size_t)N * Hc * 4);
        hipLaunchKernelGGL(gnn_output_type_kernel, dim3(cdiv(N * Hc, 256)), dim3(256), 0, s, h[cur], d_fed, N, H, g->Uin, g->Wout,
                           c.output_type, y);
        hcls = y;
    }
    if (R > 0) {
        hipLaunchKernelGGL(gnn_pair_pre_kernel, dim3(std::min(cdiv(N * c.cls_hidden1, 256), 1024)), dim3(256), 0, s,
                           hcls, N, Hc, g->C1, c.cls_hidden1, Pt, Qt);
        if (g->cls_fast) {
            PairArgs pa{};
            pa.Pt = Pt; pa.Qt = Qt; pa.b1 = g->cb1; pa.W2 = g->C2; pa.b2 = g->cb2; pa.W3 = g->C3; pa.b3 = g->cb3;
            pa.rel = d_rel; pa.out = d_out; pa.N = N; pa.R = R;
            hipLaunchKernelGGL((gnn_pair_cls_kernel<64, 32, 2>), dim3(cdiv(R, 256)), dim3(256), 0, s, pa);
        } else {
            PairGenArgs pg{};
            pg.Pt = Pt; pg.Qt = Qt; pg.b1 = g->cb1; pg.rest = g->cls_rest; pg.rel = d_rel; pg.out = d_out; pg.N = N; pg.R = R;
            pg.maxw = g->cls_maxw;
            hipLaunchKernelGGL(gnn_pair_cls_generic_kernel, dim3(cdiv(R, PAIRG_P)), dim3(256), 2 * (size_t)PAIRG_P * g->cls_maxw * sizeof(float), s, pg);
        }
    }
    ASEP_HIP_CHECK(hipGetLastError());
    return ASEP_OK;
}

#ifdef ASEP_ABLATION
// ---- several pages' graphs, stage by stage (asep_gnn_forward_visual_batch_dev) --------------------------------------------------
// The graph of one page is ~15 launches of at most 200 workgroups whose duration is the latency of ONE workgroup's work (a transition
// step: 60-100 us for 200 nodes); sixteen pages one after the other are 48 such step launches per call.  Here every stage that is a
// fused kernel runs ONCE for all pages (blockIdx.y = page, GnnBatch): the ROI kernels per feature map, the transition steps, the pair
// classifier.  Same kernel bodies on the same values: bit-identical to the page-by-page form (tests/test_gnn_visual_gpu.py).
struct GraphCtx {
    int N = 0, E = 0, R = 0;
    const float* d_u = nullptr;       // node features the steps read (compressed if the model compresses)
    const float* d_fed = nullptr;     // as fed
    const float* d_ef = nullptr;
    const int32_t* d_rel = nullptr;
    const int32_t* d_edges_in = nullptr;
    float* d_out = nullptr;
    EdgeBufs eb{};
    float* h[2] = {nullptr, nullptr};
    float* cs[2] = {nullptr, nullptr};
    float *Pt = nullptr, *Qt = nullptr, *upad = nullptr;
};

bool graph_batch_eligible(const asep_gnn* g) {
    const int mode = g->use_step ? g->mode : STEP_GENERIC;
    return (mode == STEP_BIG || mode == STEP_SMALL) && g->cfg.attention_heads == 0 && g->cfg.output_type == 0 && !g->Wc;
}

// n <= GNN_BATCH pages: edges / buffers per page, then the steps and the classifier as batched launches.  The pool is NOT reset between pages.
int forward_batch_impl(asep_gnn* g, BufferPool& pool, int n, GraphCtx* cx, hipStream_t s) {
    const asep_gnn_cfg& c = g->cfg;
    const int H = g->H;
    const int mode = g->use_step ? g->mode : STEP_GENERIC;
    int maxN = 0, maxR = 0;
    for (int b = 0; b < n; ++b) {
        GraphCtx& q = cx[b];
        if (q.N < 1 || q.E < 0 || q.R < 0) { set_error("asep_gnn_forward: bad sizes N=%d E=%d R=%d", q.N, q.E, q.R); return ASEP_ERR_ARG; }
        if ((size_t)q.N * q.N > (size_t)1 << 30) { set_error("asep_gnn_forward: N=%d too large for the dense edge table", q.N); return ASEP_ERR_UNSUPPORTED; }
        int rc = correct_edges_dev(g, pool, q.N, q.E, q.d_edges_in, q.eb, s);
        if (rc) return rc;
        const size_t nh = (size_t)q.N * H;
        q.h[0] = (float*)pool.get(nh * 4); q.h[1] = (float*)pool.get(nh * 4);
        q.cs[0] = (float*)pool.get(nh * 4); q.cs[1] = (float*)pool.get(nh * 4);
        q.Pt = (float*)pool.get((size_t)q.N * c.cls_hidden1 * 4);
        q.Qt = (float*)pool.get((size_t)q.N * c.cls_hidden1 * 4);
        if (mode == STEP_BIG) {
            q.upad = (float*)pool.get((size_t)q.N * g->Upad * 4);
            hipLaunchKernelGGL(gnn_pad_rows_kernel, dim3(cdiv(q.N * g->Upad, 256)), dim3(256), 0, s, q.d_u, q.N, g->U, q.upad, g->Upad);
        }
        ASEP_HIP_CHECK(hipMemsetAsync(q.h[0], 0, nh * 4, s));
        ASEP_HIP_CHECK(hipMemsetAsync(q.cs[0], 0, nh * 4, s));
        maxN = std::max(maxN, q.N);
        maxR = std::max(maxR, q.R);
    }
    int cur = 0;
    for (int t = 0; t < c.num_transition_steps; ++t) {
        if (mode == STEP_BIG) {
            GnnBatch<StepBigArgs> ba{};
            for (int b = 0; b < n; ++b) {
                GraphCtx& q = cx[b];
                StepBigArgs& sa = ba.p[b];
                sa.u = q.upad; sa.h_in = q.h[cur]; sa.c_in = q.cs[cur]; sa.ef = q.d_ef;
                sa.tptr = q.eb.colptr; sa.tsrc = q.eb.tsrc; sa.tfirst = q.eb.tfirst;
                sa.A1 = (const gf32x4*)g->A1; sa.A2 = (const gf32x4*)g->A2; sa.b1 = g->b1; sa.b2 = g->b2;
                for (int k = 0; k < 4; ++k) { sa.Wg[k] = g->Wg[k]; sa.bg[k] = g->bg[k]; }
                sa.h_out = q.h[cur ^ 1]; sa.c_out = q.cs[cur ^ 1];
                sa.N = q.N; sa.U = g->U; sa.Upad = g->Upad; sa.Ed = g->Ed; sa.E = std::max(q.E, 1); sa.nch = g->nch;
                sa.qdesc = g->qdesc;
                ba.nx[b] = q.N;
            }
            hipLaunchKernelGGL(gnn_step_big_kernel_batch, dim3(maxN, n), dim3(256), g->big_lds, s, ba);
        } else {
            GnnBatch<StepArgs> ba{};
            for (int b = 0; b < n; ++b) {
                GraphCtx& q = cx[b];
                StepArgs& sa = ba.p[b];
                sa.u = q.d_u; sa.h_in = q.h[cur]; sa.c_in = q.cs[cur]; sa.ef = q.d_ef;
                sa.tptr = q.eb.colptr; sa.tsrc = q.eb.tsrc; sa.tfirst = q.eb.tfirst;
                sa.A1 = (const gf32x4*)g->A1; sa.A2 = (const gf32x4*)g->A2; sa.b1 = g->b1; sa.b2 = g->b2;
                for (int k = 0; k < 4; ++k) { sa.Wg[k] = g->Wg[k]; sa.bg[k] = g->bg[k]; }
                sa.h_out = q.h[cur ^ 1]; sa.c_out = q.cs[cur ^ 1];
                sa.N = q.N; sa.U = g->U; sa.Ed = g->Ed; sa.E = std::max(q.E, 1); sa.nch = g->nch;
                sa.qdesc = g->qdesc;
                ba.nx[b] = q.N;
            }
            hipLaunchKernelGGL(gnn_step_kernel_batch, dim3(maxN, n), dim3(256), 0, s, ba);
        }
        cur ^= 1;
    }
    g->d_h = cx[n - 1].h[cur];
    g->N = cx[n - 1].N;
    if (maxR > 0) {
        GnnBatch<PairPreArgs> bp{};
        int maxpre = 0;
        for (int b = 0; b < n; ++b) {
            GraphCtx& q = cx[b];
            bp.p[b] = PairPreArgs{q.h[cur], q.N, H, g->C1, c.cls_hidden1, q.Pt, q.Qt};
            bp.nx[b] = q.R > 0 ? std::min(cdiv(q.N * c.cls_hidden1, 256), 1024) : 0;
            maxpre = std::max(maxpre, bp.nx[b]);
        }
        hipLaunchKernelGGL(gnn_pair_pre_kernel_batch, dim3(maxpre, n), dim3(256), 0, s, bp);
        if (g->cls_fast) {
            GnnBatch<PairArgs> ba{};
            for (int b = 0; b < n; ++b) {
                GraphCtx& q = cx[b];
                PairArgs& pa = ba.p[b];
                pa.Pt = q.Pt; pa.Qt = q.Qt; pa.b1 = g->cb1; pa.W2 = g->C2; pa.b2 = g->cb2; pa.W3 = g->C3; pa.b3 = g->cb3;
                pa.rel = q.d_rel; pa.out = q.d_out; pa.N = q.N; pa.R = q.R;
                ba.nx[b] = cdiv(q.R, 256);
            }
            hipLaunchKernelGGL((gnn_pair_cls_kernel_batch<64, 32, 2>), dim3(cdiv(maxR, 256), n), dim3(256), 0, s, ba);
        } else {
            for (int b = 0; b < n; ++b) {
                GraphCtx& q = cx[b];
                if (q.R < 1) continue;
                PairGenArgs pg{};
                pg.Pt = q.Pt; pg.Qt = q.Qt; pg.b1 = g->cb1; pg.rest = g->cls_rest; pg.rel = q.d_rel; pg.out = q.d_out; pg.N = q.N; pg.R = q.R;
                pg.maxw = g->cls_maxw;
                hipLaunchKernelGGL(gnn_pair_cls_generic_kernel, dim3(cdiv(q.R, PAIRG_P)), dim3(256), 2 * (size_t)PAIRG_P * g->cls_maxw * sizeof(float), s, pg);
            }
        }
    }
    ASEP_HIP_CHECK(hipGetLastError());
    return ASEP_OK;
}
#endif

// graph_relation.py:84-139 in front of the graph: backbone, ROI max + compression per feature map, concatenation.
// ROI max + compression of one page's nodes from the backbone end points "<prefix><name>" of the forward that is queued
// on s; d_ug [N, U - visual dims] -> d_u [N, U]
int visual_rois_dev(asep_gnn* g, int N, const float* d_ug, const std::string& prefix, const float* d_reg, int P,
                    const int32_t* d_np, float* d_u, hipStream_t s) {
    const int U = g->Uin, ug = U - g->vis_total;            // width of the FED (concatenated) features
    if (ug > 0) hipLaunchKernelGGL(gnn_copy_cols_kernel, dim3(cdiv(N * ug, 256)), dim3(256), 0, s, d_ug, N, ug, d_u, U);
    int col = ug;
    for (size_t i = 0; i < g->vis_names.size(); ++i) {
        const float* fm = nullptr;
        int dims[3];
        int bf = 0;                                         // a bf16 backbone (compute_dtype 1) hands over bf16 maps
        int rc = aru_endpoint_dev(g->backbone, (prefix + g->vis_names[i]).c_str(), &fm, dims, &bf);
        if (rc) return rc;
        if (dims[2] != g->vis_C[i]) { set_error("end point %s has %d channels, expected %d", g->vis_names[i].c_str(), dims[2], g->vis_C[i]); return ASEP_ERR_ARG; }
        RoiArgs a{};
        a.fm = fm; a.fh = dims[0]; a.fw = dims[1]; a.C = dims[2];
        a.regions = d_reg; a.P = P; a.npts = d_np; a.Wc = g->vis_W[i]; a.bc = g->vis_b[i]; a.d = g->vis_d[i];
        a.u_out = d_u; a.ustride = U; a.col0 = col; a.vmax_out = nullptr;
        if (bf) hipLaunchKernelGGL(gnn_roi_compress_kernel<true>, dim3(N), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(gnn_roi_compress_kernel<false>, dim3(N), dim3(256), 0, s, a);
        col += g->vis_d[i];
    }
    ASEP_HIP_CHECK(hipGetLastError());
    return ASEP_OK;
}

// graph_relation.py:141-172 + misc.py:384-470: the same ROI max + compression for the E interactions' regions; the compressed
// features go behind the fed edge features: d_ef_fed [E, Ed_fed] -> d_ef [E, Ed]
int visual_edge_rois_dev(asep_gnn* g, int E, const float* d_ef_fed, const std::string& prefix, const float* d_reg, int P,
                         const int32_t* d_np, float* d_ef, hipStream_t s) {
    if (E < 1) return ASEP_OK;
    const int Ed = g->Ed, ef = g->Ed_fed;
    if (ef > 0) hipLaunchKernelGGL(gnn_copy_cols_kernel, dim3(cdiv(E * ef, 256)), dim3(256), 0, s, d_ef_fed, E, ef, d_ef, Ed);
    int col = ef;
    for (size_t i = 0; i < g->vis_names.size(); ++i) {
        const float* fm = nullptr;
        int dims[3];
        int bf = 0;
        int rc = aru_endpoint_dev(g->backbone, (prefix + g->vis_names[i]).c_str(), &fm, dims, &bf);
        if (rc) return rc;
        RoiArgs a{};
        a.fm = fm; a.fh = dims[0]; a.fw = dims[1]; a.C = dims[2];
        a.regions = d_reg; a.P = P; a.npts = d_np; a.Wc = g->vise_W[i]; a.bc = g->vise_b[i]; a.d = g->vise_d[i];
        a.u_out = d_ef; a.ustride = Ed; a.col0 = col; a.vmax_out = nullptr;
        if (bf) hipLaunchKernelGGL(gnn_roi_compress_kernel<true>, dim3(E), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(gnn_roi_compress_kernel<false>, dim3(E), dim3(256), 0, s, a);
        col += g->vise_d[i];
    }
    ASEP_HIP_CHECK(hipGetLastError());
    return ASEP_OK;
}

int reserve_ef_cat(asep_gnn* g, size_t n) {
    if (g->ef_cat_cap < n) {
        if (g->d_ef_cat) (void)hipFree(g->d_ef_cat);
        g->d_ef_cat = nullptr; g->ef_cat_cap = 0;
        ASEP_HIP_CHECK(hipMalloc((void**)&g->d_ef_cat, n * sizeof(float)));
        g->ef_cat_cap = n;
    }
    return ASEP_OK;
}

int reserve_u_cat(asep_gnn* g, size_t nu) {
    if (g->u_cat_cap < nu) {
        if (g->d_u_cat) (void)hipFree(g->d_u_cat);
        g->d_u_cat = nullptr; g->u_cat_cap = 0;
        ASEP_HIP_CHECK(hipMalloc((void**)&g->d_u_cat, nu * sizeof(float)));
        g->u_cat_cap = nu;
    }
    return ASEP_OK;
}

// d_ug [N, U - visual dims]; everything on stream s; returns the concatenated features [N, U] in *d_u_out.
int visual_features_dev(asep_gnn* g, int N, const float* d_ug, const float* d_img, int h, int w, const float* d_reg, int P,
                        const int32_t* d_np, float** d_u_out, hipStream_t s) {
    g->vis_pool.begin();
    const int ncls = std::max(1, aru_num_classes(g->backbone));
    float* d_bo = (float*)g->vis_pool.get((size_t)h * w * ncls * sizeof(float));   // backbone logits (not used by the graph)
    int rc = reserve_u_cat(g, (size_t)N * g->Uin);
    if (rc) return rc;
    rc = asep_aru_forward_dev(g->backbone, d_img, h, w, d_bo, nullptr, nullptr, 0.f, s);
    if (rc) return rc;
    rc = visual_rois_dev(g, N, d_ug, std::string(), d_reg, P, d_np, g->d_u_cat, s);
    if (rc) return rc;
    *d_u_out = g->d_u_cat;
    return ASEP_OK;
}

// host-side checks of the index arrays (the reference's tf.gather would raise on them)
int check_indices(const char* what, const int32_t* idx, size_t n_pairs, int N) {
    for (size_t i = 0; i < 2 * n_pairs; ++i)
        if (idx[i] < 0 || idx[i] >= N) {
            set_error("%s[%zu] = (%d, %d) names a node outside 0..%d", what, i / 2, idx[i & ~(size_t)1], idx[i | 1], N - 1);
            return ASEP_ERR_ARG;
        }
    return ASEP_OK;
}

}  // namespace

extern "C" {

asep_gnn* asep_gnn_load(const void* weight_blob, size_t nbytes, const asep_gnn_cfg* cfg) {
    ASEP_GUARD_BEGIN
    if (!cfg || !weight_blob) { set_error("asep_gnn_load: null argument"); return nullptr; }
    if (cfg->struct_size != (int32_t)sizeof(asep_gnn_cfg)) {
        set_error("asep_gnn_load: cfg.struct_size is %d, this library's asep_gnn_cfg has %zu bytes (ABI version %d): the binding was "
                  "written against another include/asep_hip.h", cfg->struct_size, sizeof(asep_gnn_cfg), ASEP_ABI_VERSION);
        return nullptr;
    }
    if (cfg->node_feature_dim < 1 || cfg->edge_feature_dim < 0 || cfg->num_transition_steps < 0 || cfg->hidden_dim < 1 ||
        cfg->interaction_dim < 1 || cfg->interaction_hidden < 1 || cfg->cls_hidden1 < 1 || cfg->cls_hidden2 < 0 ||
        cfg->num_classes < 1 || cfg->num_classes > 16 || cfg->visual_edge_dims < 0 || cfg->visual_edge_dims > cfg->edge_feature_dim) {
        set_error("asep_gnn_load: bad cfg");
        return nullptr;
    }
    if (cfg->aggregation_type != 0 && cfg->aggregation_type != 1) {
        set_error("asep_gnn_load: aggregation_type %d unknown (0 = 'sum', 1 = 'max'; message_fn_chunk.py:57-62)", cfg->aggregation_type);
        return nullptr;
    }
    std::map<std::string, HostTensor> blob;
    if (!parse_blob(weight_blob, nbytes, blob)) return nullptr;
    std::unique_ptr<asep_gnn> g(new asep_gnn());
    g->cfg = *cfg;
    g->cfg.lstm_use_hidden = cfg->lstm_use_hidden != 0;
    g->cfg.lstm_use_input = cfg->lstm_use_input != 0;
    const int U = g->U = cfg->node_feature_dim, Ed = g->Ed = cfg->edge_feature_dim;
    const int H = g->H = cfg->hidden_dim, I = g->I = cfg->interaction_dim, Hm = g->Hm = cfg->interaction_hidden;
    g->Ed_fed = Ed - cfg->visual_edge_dims;
    g->K = 4 * U + Ed + 4 * H;
    g->Uin = cfg->compress_input_dim > 0 ? cfg->compress_input_dim : U;
    const std::vector<int> ih = hidden_list({cfg->interaction_hidden, cfg->interaction_hidden2, cfg->interaction_hidden3, cfg->interaction_hidden4});
    const std::vector<int> ah = hidden_list({cfg->attention_hidden, cfg->attention_hidden2, cfg->attention_hidden3, cfg->attention_hidden4});
    const std::vector<int> ch = hidden_list({cfg->cls_hidden1, cfg->cls_hidden2, cfg->cls_hidden3, cfg->cls_hidden4});
    const std::string m = MSG, u = UPD, c = CLS;
    const int heads = cfg->attention_heads;
    int rc = ASEP_OK;
    if (heads < 0 || heads > GNN_MAX_HEADS || (heads > 0 && (cfg->attention_hidden < 1 || (cfg->attention_merge != 0 && cfg->attention_merge != 1)))) {
        set_error("asep_gnn_load: bad attention configuration (heads %d of at most %d, hidden %d, merge %d)", heads, GNN_MAX_HEADS,
                  cfg->attention_hidden, cfg->attention_merge);
        return nullptr;
    }
    g->msg_maxh = *std::max_element(ih.begin(), ih.end());
    if (heads > 0) g->msg_maxh = std::max(g->msg_maxh, *std::max_element(ah.begin(), ah.end()));
    // LDS of the generic message kernels: z + two scratch vectors (+ the accumulators); refused here, not at the first forward
    if (((size_t)g->K + 2 * (size_t)g->msg_maxh + I) * sizeof(float) > 60 * 1024) {
        set_error("asep_gnn_load: edge-MLP input width %d with hidden layers up to %d needs more LDS than a workgroup may use", g->K, g->msg_maxh);
        return nullptr;
    }
    g->Iout = I;
    if (heads > 0) {
        // message_fn_chunk.py:69-72: x_dim = interaction_dim // heads for 'concat' (the LSTM then reads heads * x_dim columns)
        if (cfg->attention_merge == 0 && I % heads != 0) {
            set_error("asep_gnn_load: interaction_dim %d is not a multiple of %d attention heads (merge 'concat')", I, heads);
            return nullptr;
        }
        g->att_xd = cfg->attention_merge == 0 ? I / heads : I;
        g->Iout = cfg->attention_merge == 0 ? heads * g->att_xd : I;
        for (int k = 0; k < heads && !rc; ++k) {
            const std::string hk = "GraphLSTM1/message_fn_default/head_" + std::to_string(k) + "/";
            const std::string mi = hk + "calculation_interaction_features/concat_u_and_h/interaction_features";
            const std::string ma = hk + "calculation_unnormalized_attention_values/calculation_interaction_features/concat_u_and_h/interaction_features";
            rc = upload_mlp(g.get(), blob, mi, g->K, ih, g->att_xd, &g->att_w[k].inter);
            if (!rc) rc = upload_mlp(g.get(), blob, ma, g->K, ah, 1, &g->att_w[k].att);
        }
        if (!rc) {
            std::vector<AttHeadW> hw(g->att_w, g->att_w + heads);
            rc = upload_vec(g.get(), hw, &g->d_att_w);
        }
    } else {
        rc = upload_mlp(g.get(), blob, m, g->K, ih, I, &g->msg);
        if (!rc && ih.size() == 1) { g->b1 = const_cast<float*>(g->msg.b[0]); g->b2 = const_cast<float*>(g->msg.b[1]); }
    }
    g->V = g->Iout + (g->cfg.lstm_use_hidden ? H : 0) + (g->cfg.lstm_use_input ? U : 0);      // update_fn_lstm.py:41-50
    const char* gates[4] = {"ingate", "outgate", "forgetgate", "cellinput"};
    for (int q = 0; q < 4 && !rc; ++q) {
        rc = upload_named(g->owned, blob, u + "/" + gates[q] + "_activation/dense/weights", {g->V, H}, &g->Wg[q]);
        if (!rc) rc = upload_named(g->owned, blob, u + "/" + gates[q] + "_activation/dense/bias", {H}, &g->bg[q]);
    }
    if (cfg->output_type < 0 || cfg->output_type > 2) { set_error("asep_gnn_load: output_type %d unknown (0 hidden, 1 add, 2 concat)", cfg->output_type); return nullptr; }
    const int Hc = cfg->output_type == 2 ? H + g->Uin : H;   // width of the node vectors the classifier pairs up
    if (!rc) rc = upload_named(g->owned, blob, c + "/fully_connected_layer_h1/weights", {2 * Hc, cfg->cls_hidden1}, &g->C1);
    if (!rc && cfg->output_type == 1) rc = upload_named(g->owned, blob, "GraphLSTM1/dense/weights", {g->Uin, H}, &g->Wout);
    if (!rc) rc = upload_named(g->owned, blob, c + "/fully_connected_layer_h1/bias", {cfg->cls_hidden1}, &g->cb1);
    if (!rc) rc = upload_mlp(g.get(), blob, c, 2 * Hc, ch, cfg->num_classes, &g->cls_rest, /*first_layer=*/1);
    if (!rc) {
        g->cls_maxw = std::max(cfg->num_classes, *std::max_element(ch.begin(), ch.end()));
        g->cls_fast = ch.size() == 2 && ch[0] == 64 && ch[1] == 32 && cfg->num_classes == 2;
        if (g->cls_fast) {                                   // the specialised kernel's operand names
            g->C2 = const_cast<float*>(g->cls_rest.W[0]); g->cb2 = const_cast<float*>(g->cls_rest.b[0]);
            g->C3 = const_cast<float*>(g->cls_rest.W[1]); g->cb3 = const_cast<float*>(g->cls_rest.b[1]);
        }
        if (2 * (size_t)PAIRG_P * g->cls_maxw * sizeof(float) > 60 * 1024) {
            set_error("asep_gnn_load: classifier layers up to %d wide need more LDS than a workgroup may use", g->cls_maxw);
            return nullptr;
        }
    }
    if (!rc && cfg->compress_input_dim > 0) {
        rc = upload_named(g->owned, blob, "GraphLSTM1/compress_input/ff_compress_input/weights", {g->Uin, U}, &g->Wc);
        if (!rc) rc = upload_named(g->owned, blob, "GraphLSTM1/compress_input/ff_compress_input/bias", {U}, &g->bc);
    }
    if (rc) return nullptr;
    for (auto& kv : blob)
        if (kv.first.rfind("visual_node_feature_compression_fm_", 0) == 0 || kv.first.rfind("visual_edge_feature_compression_fm_", 0) == 0)
            g->vis_blob[kv.first] = kv.second;
    warn_ignored_switches();
    if (const char* ev = getenv("ASEP_GNN_STEP")) g->use_step = atoi(ev) != 0;
    if (const char* ev = getenv("ASEP_GNN_FACTOR")) g->use_fact = atoi(ev) != 0;
#ifdef ASEP_ABLATION   // measured and not adopted (DESIGN_LESSONS 28, 36); `make ABLATION=1` builds them for scripts/r4_ab.sh
    if (const char* ev = getenv("ASEP_GNN_BATCH")) g->batch_graph = atoi(ev) != 0; if (const char* e2 = getenv("ASEP_GNN_LANES")) g->n_page_lanes = std::max(1, std::min(16, atoi(e2)));
#endif
    // the fused MFMA step kernels serve the reference's defaults: widths 32 / [32] / 32, degree-normalised SUM, both LSTM inputs
    const bool default_widths = H == GNN_H && I == GNN_H && Hm == GNN_H && ih.size() == 1 && heads == 0 && cfg->aggregation_type == 0 &&
                                g->cfg.lstm_use_hidden && g->cfg.lstm_use_input;
    g->mode = STEP_GENERIC;
    if (default_widths && Ed <= 4) {
        if (U <= 8) {
            if (pack_step_fragments(g.get(), blob, false)) return nullptr;
            if (g->nch <= GNN_MAXCH) g->mode = STEP_SMALL;
        }
        if (g->mode == STEP_GENERIC && U <= 120) {
            g->Upad = (U + 3) & ~3;
            if (pack_step_fragments(g.get(), blob, true)) return nullptr;
            g->big_lds = ((size_t)g->nch * 512 + g->Upad + 32) * sizeof(float);
            if (g->big_lds <= 150 * 1024 &&
                hipFuncSetAttribute((const void*)gnn_step_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->big_lds) == hipSuccess)
                g->mode = STEP_BIG;
        }
    }
    if (g->mode != STEP_GENERIC) {                          // the factored step for the same widths (any U whose filters fit the pre kernel's LDS)
        g->fact_pre_lds = ((size_t)(U + Ed) * 32 + U) * sizeof(float);
        g->fact_lds = (size_t)(64 + U) * sizeof(float);
        if (g->fact_pre_lds <= 60 * 1024) {
            if (pack_fact(g.get(), blob)) return nullptr;
        }
    }
    const size_t lds = ((size_t)g->K + 2 * (size_t)g->msg_maxh + I) * sizeof(float);
    if (hipFuncSetAttribute((const void*)gnn_message_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)gnn_msg_att_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)gnn_pair_cls_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(2 * (size_t)PAIRG_P * g->cls_maxw * sizeof(float))) != hipSuccess) {
        set_error("asep_gnn_load: cannot reserve %zu bytes of LDS", lds);
        return nullptr;
    }
    return g.release();
    ASEP_GUARD_END_PTR
}

void asep_gnn_free(asep_gnn* g) { delete g; }

int asep_gnn_forward_dev(asep_gnn* g, int N, int E, const int32_t* d_edges, const float* d_node_feat,
                         const float* d_edge_feat, int R, const int32_t* d_relations, float* d_probs_out, void* stream) {
    ASEP_GUARD_BEGIN
    if (!g || !d_node_feat || (E > 0 && !d_edges) || (R > 0 && !d_probs_out)) { set_error("asep_gnn_forward_dev: null argument"); return ASEP_ERR_ARG; }
    if (g->cfg.edge_feature_dim > 0 && E > 0 && !d_edge_feat) { set_error("asep_gnn_forward_dev: edge features required"); return ASEP_ERR_ARG; }
    if (g->cfg.visual_edge_dims > 0) { set_error("asep_gnn_forward_dev: this graph assigns visual features to edges: use asep_gnn_forward_visual*"); return ASEP_ERR_ARG; }
    return forward_impl(g, g->pool, N, E, d_edges, d_node_feat, d_edge_feat, R, d_relations, d_probs_out, (hipStream_t)stream);
    ASEP_GUARD_END
}

int asep_gnn_forward(asep_gnn* g, int N, int E, const int32_t* edges, const float* node_feat, const float* edge_feat,
                     int R, const int32_t* relations, float* probs_out) {
    ASEP_GUARD_BEGIN
    if (!g || !node_feat || (E > 0 && !edges) || (R > 0 && !probs_out)) { set_error("asep_gnn_forward: null argument"); return ASEP_ERR_ARG; }
    if (g->cfg.visual_edge_dims > 0) { set_error("asep_gnn_forward: this graph assigns visual features to edges: use asep_gnn_forward_visual*"); return ASEP_ERR_ARG; }
    if (N < 1 || E < 0 || R < 0) { set_error("asep_gnn_forward: bad sizes N=%d E=%d R=%d", N, E, R); return ASEP_ERR_ARG; }
    int rc;
    if (E > 0 && (rc = check_indices("interacting_nodes", edges, (size_t)E, N))) return rc;
    if (relations && R > 0 && (rc = check_indices("relations", relations, (size_t)R, N))) return rc;
    // grow-only device staging in the handle: seven hipMalloc / hipFree pairs per page would cost more than the GNN
    const size_t ne = (size_t)E * 2, nu = (size_t)N * g->Uin, nf = (size_t)E * g->Ed;      // (features as fed: width Uin)
    const size_t nr = relations ? (size_t)R * 2 : 0, no = (size_t)R * g->cfg.num_classes;
    g->host_stage.begin();
    int32_t* d_e = (int32_t*)g->host_stage.get(std::max<size_t>(ne, 1) * sizeof(int32_t));
    float* d_u = (float*)g->host_stage.get(std::max<size_t>(nu, 1) * sizeof(float));
    float* d_f = (float*)g->host_stage.get(std::max<size_t>(nf, 1) * sizeof(float));
    int32_t* d_r = (int32_t*)g->host_stage.get(std::max<size_t>(nr, 1) * sizeof(int32_t));
    float* d_o = (float*)g->host_stage.get(std::max<size_t>(no, 1) * sizeof(float));
    if (ne) ASEP_HIP_CHECK(hipMemcpyAsync(d_e, edges, ne * sizeof(int32_t), hipMemcpyHostToDevice, nullptr));
    ASEP_HIP_CHECK(hipMemcpyAsync(d_u, node_feat, nu * sizeof(float), hipMemcpyHostToDevice, nullptr));
    if (nf && edge_feat) ASEP_HIP_CHECK(hipMemcpyAsync(d_f, edge_feat, nf * sizeof(float), hipMemcpyHostToDevice, nullptr));
    if (nr) ASEP_HIP_CHECK(hipMemcpyAsync(d_r, relations, nr * sizeof(int32_t), hipMemcpyHostToDevice, nullptr));
    rc = asep_gnn_forward_dev(g, N, E, ne ? d_e : nullptr, d_u, (nf && edge_feat) ? d_f : nullptr, R,
                              nr ? d_r : nullptr, d_o, nullptr);
    if (rc) return rc;
    if (R > 0) ASEP_HIP_CHECK(hipMemcpyAsync(probs_out, d_o, no * sizeof(float), hipMemcpyDeviceToHost, nullptr));
    ASEP_HIP_CHECK(hipStreamSynchronize(nullptr));
    return ASEP_OK;
    ASEP_GUARD_END
}

int asep_gnn_correct_edges(asep_gnn* g, int N, int E, const int32_t* edges, const float* edge_feat, int32_t* out_edges,
                           float* out_feat) {
    ASEP_GUARD_BEGIN
    if (!g || N < 1 || E < 0 || (E > 0 && !edges) || !out_edges) { set_error("asep_gnn_correct_edges: bad argument"); return ASEP_ERR_ARG; }
    if ((size_t)N * N > (size_t)1 << 30) { set_error("asep_gnn_correct_edges: N too large"); return ASEP_ERR_UNSUPPORTED; }
    const int Ed = g->Ed;
    const bool feats = out_feat && Ed > 0 && edge_feat;
    g->host_stage.begin();
    int32_t* d_e = (int32_t*)g->host_stage.get(std::max<size_t>((size_t)E * 2, 1) * sizeof(int32_t));
    float* d_f = (float*)g->host_stage.get(std::max<size_t>(feats ? (size_t)E * Ed : 0, 1) * sizeof(float));
    float* d_of = (float*)g->host_stage.get(std::max<size_t>(feats ? (size_t)E * 2 * Ed : 0, 1) * sizeof(float));
    if (E > 0) ASEP_HIP_CHECK(hipMemcpy(d_e, edges, (size_t)E * 2 * sizeof(int32_t), hipMemcpyHostToDevice));
    if (feats && E > 0) ASEP_HIP_CHECK(hipMemcpy(d_f, edge_feat, (size_t)E * Ed * sizeof(float), hipMemcpyHostToDevice));
    g->pool.begin();
    EdgeBufs eb{};
    int rc = correct_edges_dev(g, g->pool, N, E, d_e, eb, nullptr);
    if (rc) return rc;
    int ecorr = 0;
    ASEP_HIP_CHECK(hipStreamSynchronize(nullptr));
    ASEP_HIP_CHECK(hipMemcpy(&ecorr, eb.rowptr + N, sizeof(int), hipMemcpyDeviceToHost));
    if (ecorr > 0) {
        ASEP_HIP_CHECK(hipMemcpy(out_edges, eb.sorted, (size_t)ecorr * 2 * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (feats) {
            hipLaunchKernelGGL(edge_feat_gather_kernel, dim3(std::min(cdiv(ecorr * Ed, 256), 1024)), dim3(256), 0, nullptr,
                               d_f, E, Ed, eb.sfirst, ecorr, d_of);
            ASEP_HIP_CHECK(hipMemcpy(out_feat, d_of, (size_t)ecorr * Ed * sizeof(float), hipMemcpyDeviceToHost));
        }
    }
    return ecorr;
    ASEP_GUARD_END
}

int asep_gnn_get_hidden(asep_gnn* g, float* out, size_t max_floats) {
    ASEP_GUARD_BEGIN
    if (!g || !out || !g->d_h) { set_error("asep_gnn_get_hidden: no forward has run"); return ASEP_ERR_ARG; }
    const size_t n = (size_t)g->N * g->H;
    if (max_floats < n) { set_error("asep_gnn_get_hidden: buffer too small"); return ASEP_ERR_ARG; }
    ASEP_HIP_CHECK(hipStreamSynchronize(g->stream));
    ASEP_HIP_CHECK(hipMemcpy(out, g->d_h, n * sizeof(float), hipMemcpyDeviceToHost));
    return ASEP_OK;
    ASEP_GUARD_END
}

int asep_gnn_step_mode(const asep_gnn* g) { return g ? step_mode_of(g) : ASEP_ERR_ARG; }

int asep_gnn_attach_backbone(asep_gnn* g, asep_aru* backbone, int n_maps, const char* const* endpoint_names) {
    ASEP_GUARD_BEGIN
    if (!g || !backbone || n_maps < 1 || !endpoint_names) { set_error("asep_gnn_attach_backbone: bad argument"); return ASEP_ERR_ARG; }
    g->free_visual();                                   // a second attach replaces (and frees) the first one's uploads
    for (int i = 0; i < n_maps; ++i) {
        if (!endpoint_names[i]) { set_error("asep_gnn_attach_backbone: null end-point name"); g->free_visual(); return ASEP_ERR_ARG; }
        const int C = aru_endpoint_channels(backbone, endpoint_names[i]);
        if (C < 1 || C > 256) {
            set_error("asep_gnn_attach_backbone: '%s' is not a feature map of this backbone (only unet conv/deconv "
                      "end points with layer_depth -1 are supported)", endpoint_names[i]);
            g->free_visual();
            return ASEP_ERR_UNSUPPORTED;
        }
        const std::string scope = "visual_node_feature_compression_fm_" + std::to_string(i) + "/dense/";
        auto w = g->vis_blob.find(scope + "weights"), b = g->vis_blob.find(scope + "bias");
        if (w == g->vis_blob.end() || b == g->vis_blob.end()) { set_error("weights: missing tensor %sweights|bias", scope.c_str()); g->free_visual(); return ASEP_ERR_WEIGHTS; }
        if (w->second.dims.size() != 2 || w->second.dims[0] != C || b->second.dims.size() != 1 || b->second.dims[0] != w->second.dims[1]) {
            set_error("weights: %sweights must be [%d, d] with a matching bias", scope.c_str(), C);
            g->free_visual();
            return ASEP_ERR_WEIGHTS;
        }
        float *dW = nullptr, *db = nullptr;
        int rc = upload_named(g->vis_owned, g->vis_blob, scope + "weights", w->second.dims, &dW);
        if (!rc) rc = upload_named(g->vis_owned, g->vis_blob, scope + "bias", b->second.dims, &db);
        if (rc) { g->free_visual(); return rc; }
        g->vis_names.push_back(endpoint_names[i]);
        g->vis_W.push_back(dW); g->vis_b.push_back(db);
        g->vis_C.push_back(C); g->vis_d.push_back(w->second.dims[1]);
        g->vis_total += w->second.dims[1];
    }
    if (g->cfg.visual_edge_dims > 0) {                     // assign_visual_features_to_edges: one compression layer per feature map
        for (int i = 0; i < n_maps; ++i) {
            const std::string scope = "visual_edge_feature_compression_fm_" + std::to_string(i) + "/dense/";
            auto w = g->vis_blob.find(scope + "weights"), b = g->vis_blob.find(scope + "bias");
            if (w == g->vis_blob.end() || b == g->vis_blob.end()) { set_error("weights: missing tensor %sweights|bias", scope.c_str()); g->free_visual(); return ASEP_ERR_WEIGHTS; }
            if (w->second.dims.size() != 2 || w->second.dims[0] != g->vis_C[i] || b->second.dims.size() != 1 || b->second.dims[0] != w->second.dims[1]) {
                set_error("weights: %sweights must be [%d, d] with a matching bias", scope.c_str(), g->vis_C[i]);
                g->free_visual();
                return ASEP_ERR_WEIGHTS;
            }
            float *dW = nullptr, *db = nullptr;
            int rc = upload_named(g->vis_owned, g->vis_blob, scope + "weights", w->second.dims, &dW);
            if (!rc) rc = upload_named(g->vis_owned, g->vis_blob, scope + "bias", b->second.dims, &db);
            if (rc) { g->free_visual(); return rc; }
            g->vise_W.push_back(dW); g->vise_b.push_back(db); g->vise_d.push_back(w->second.dims[1]);
            g->vise_total += w->second.dims[1];
        }
        if (g->vise_total != g->cfg.visual_edge_dims) {
            set_error("asep_gnn_attach_backbone: the visual edge compression layers are %d wide in all, cfg.visual_edge_dims says %d", g->vise_total,
                      g->cfg.visual_edge_dims);
            g->free_visual();
            return ASEP_ERR_ARG;
        }
    }
    if (g->vis_total >= g->Uin + 1) {
        set_error("asep_gnn_attach_backbone: %d visual dims exceed the fed node feature width %d", g->vis_total, g->Uin);
        g->free_visual();
        return ASEP_ERR_ARG;
    }
    g->backbone = backbone;
    return ASEP_OK;
    ASEP_GUARD_END
}

int asep_gnn_forward_visual_dev(asep_gnn* g, int N, int E, const int32_t* d_edges, const float* d_node_feat,
                                const float* d_edge_feat, const float* d_image, int h, int w, const float* d_regions, int P,
                                const int32_t* d_num_points, const float* d_edge_regions, const int32_t* d_edge_num_points,
                                int R, const int32_t* d_relations, float* d_probs_out, void* stream) {
    ASEP_GUARD_BEGIN
    if (!g || !g->backbone) { set_error("asep_gnn_forward_visual_dev: no backbone attached"); return ASEP_ERR_ARG; }
    const int ug = g->Uin - g->vis_total;
    if (N < 1 || h < 1 || w < 1 || P < 1 || !d_image || !d_regions || !d_num_points || (ug > 0 && !d_node_feat) ||
        (E > 0 && !d_edges) || (R > 0 && !d_probs_out) || (g->Ed_fed > 0 && E > 0 && !d_edge_feat) ||
        (g->vise_total > 0 && E > 0 && (!d_edge_regions || !d_edge_num_points))) {
        set_error("asep_gnn_forward_visual_dev: bad argument");
        return ASEP_ERR_ARG;
    }
    float* d_u = nullptr;
    int rc = visual_features_dev(g, N, d_node_feat, d_image, h, w, d_regions, P, d_num_points, &d_u, (hipStream_t)stream);
    if (rc) return rc;
    if (g->vise_total > 0 && E > 0) {
        rc = reserve_ef_cat(g, (size_t)E * g->Ed);
        if (!rc) rc = visual_edge_rois_dev(g, E, d_edge_feat, std::string(), d_edge_regions, P, d_edge_num_points, g->d_ef_cat, (hipStream_t)stream);
        if (rc) return rc;
        d_edge_feat = g->d_ef_cat;
    }
    return forward_impl(g, g->pool, N, E, d_edges, d_u, d_edge_feat, R, d_relations, d_probs_out, (hipStream_t)stream);
    ASEP_GUARD_END
}

int asep_gnn_forward_visual_batch_dev(asep_gnn* g, int n_pages, const asep_gnn_page* pages, int h, int w, int P, void* stream) {
    ASEP_GUARD_BEGIN
    if (!g || !g->backbone) { set_error("asep_gnn_forward_visual_batch_dev: no backbone attached"); return ASEP_ERR_ARG; }
    if (n_pages < 1 || !pages || h < 1 || w < 1 || P < 1) { set_error("asep_gnn_forward_visual_batch_dev: bad argument"); return ASEP_ERR_ARG; }
    const int ug = g->Uin - g->vis_total;
    size_t nu = 0, nef = 0;
    for (int b = 0; b < n_pages; ++b) {
        const asep_gnn_page& q = pages[b];
        if (q.N < 1 || !q.d_image || !q.d_regions || !q.d_num_points || (ug > 0 && !q.d_node_feat) || (q.E > 0 && !q.d_edges) ||
            (q.R > 0 && !q.d_probs_out) || (g->Ed_fed > 0 && q.E > 0 && !q.d_edge_feat) ||
            (g->vise_total > 0 && q.E > 0 && (!q.d_edge_regions || !q.d_edge_num_points))) {
            set_error("asep_gnn_forward_visual_batch_dev: bad argument in page %d", b);
            return ASEP_ERR_ARG;
        }
        nu += (size_t)q.N * g->Uin;
        if (g->vise_total > 0) nef += (size_t)q.E * g->Ed;
    }
    hipStream_t s = (hipStream_t)stream;
    g->vis_pool.begin();
    const int ncls = std::max(1, aru_num_classes(g->backbone));
    std::vector<const float*> imgs(n_pages);
    std::vector<float*> outs(n_pages);
    for (int b = 0; b < n_pages; ++b) {
        imgs[b] = pages[b].d_image;
        outs[b] = (float*)g->vis_pool.get((size_t)h * w * ncls * sizeof(float));      // backbone logits (not used by the graph)
    }
    int rc = reserve_u_cat(g, nu);
    if (!rc && nef) rc = reserve_ef_cat(g, nef);
    if (rc) return rc;
    // ONE grouped backbone forward for all pages (every layer is one launch over the page list), then per page the ROI
    // kernels and the graph, queued back to back on the same stream
    rc = asep_aru_forward_batch_dev(g->backbone, n_pages, imgs.data(), h, w, outs.data(), nullptr, nullptr, 0.f, s);
    if (rc) return rc;
    // page lanes: ROI kernels + graph of page b on lane b % L, forked behind the backbone, joined into s at the end
    const int L = std::max(1, std::min(g->n_page_lanes, n_pages));
    if (L > 1) {
        while ((int)g->page_lanes.size() < L) {
            std::unique_ptr<asep_gnn::PageLane> pl(new asep_gnn::PageLane());
            ASEP_HIP_CHECK(hipStreamCreateWithFlags(&pl->s, hipStreamNonBlocking));
            ASEP_HIP_CHECK(hipEventCreateWithFlags(&pl->done, hipEventDisableTiming));
            g->page_lanes.push_back(std::move(pl));
        }
        if (!g->ev_fork) ASEP_HIP_CHECK(hipEventCreateWithFlags(&g->ev_fork, hipEventDisableTiming));
        ASEP_HIP_CHECK(hipEventRecord(g->ev_fork, s));
        for (int l = 0; l < L; ++l) ASEP_HIP_CHECK(hipStreamWaitEvent(g->page_lanes[l]->s, g->ev_fork, 0));
    }
    float* d_u = g->d_u_cat;
    float* d_efc = g->d_ef_cat;
#ifdef ASEP_ABLATION
    if (L == 1 && g->batch_graph && n_pages > 1 && g->vise_total == 0 && graph_batch_eligible(g)) {
        // stage by stage over all pages (forward_batch_impl): ROI kernels per feature map, steps, classifier as one launch each
        const int U = g->Uin, ugc = U - g->vis_total;
        g->stream = s;
        g->pool.begin();
        for (int b0 = 0; b0 < n_pages; b0 += GNN_BATCH) {
            const int nb = std::min(GNN_BATCH, n_pages - b0);
            GraphCtx cx[GNN_BATCH];
            int maxN = 0;
            float* du = d_u;
            for (int b = 0; b < nb; ++b) {
                const asep_gnn_page& q = pages[b0 + b];
                if (ugc > 0) hipLaunchKernelGGL(gnn_copy_cols_kernel, dim3(cdiv(q.N * ugc, 256)), dim3(256), 0, s, q.d_node_feat, q.N, ugc, du, U);
                cx[b].N = q.N; cx[b].E = q.E; cx[b].R = q.R;
                cx[b].d_u = du; cx[b].d_fed = du; cx[b].d_ef = q.d_edge_feat;
                cx[b].d_edges_in = q.d_edges; cx[b].d_rel = q.d_relations; cx[b].d_out = q.d_probs_out;
                du += (size_t)q.N * U;
                maxN = std::max(maxN, q.N);
            }
            int col = ugc;
            for (size_t i = 0; i < g->vis_names.size(); ++i) {
                GnnBatch<RoiArgs> ba{};
                int any_bf = -1;
                for (int b = 0; b < nb; ++b) {
                    const asep_gnn_page& q = pages[b0 + b];
                    const std::string prefix = (b0 + b) ? "p" + std::to_string(b0 + b) + "/" : std::string();
                    const float* fm = nullptr;
                    int dims[3];
                    int bf = 0;
                    rc = aru_endpoint_dev(g->backbone, (prefix + g->vis_names[i]).c_str(), &fm, dims, &bf);
                    if (rc) return rc;
                    if (dims[2] != g->vis_C[i]) { set_error("end point %s has %d channels, expected %d", g->vis_names[i].c_str(), dims[2], g->vis_C[i]); return ASEP_ERR_ARG; }
                    if (any_bf >= 0 && any_bf != bf) { set_error("end points of one call differ in precision"); return ASEP_ERR_ARG; }
                    any_bf = bf;
                    RoiArgs& a = ba.p[b];
                    a.fm = fm; a.fh = dims[0]; a.fw = dims[1]; a.C = dims[2];
                    a.regions = q.d_regions; a.P = P; a.npts = q.d_num_points; a.Wc = g->vis_W[i]; a.bc = g->vis_b[i]; a.d = g->vis_d[i];
                    a.u_out = const_cast<float*>(cx[b].d_u); a.ustride = U; a.col0 = col; a.vmax_out = nullptr;
                    ba.nx[b] = q.N;
                }
                if (any_bf) hipLaunchKernelGGL(gnn_roi_compress_kernel_batch<true>, dim3(maxN, nb), dim3(256), 0, s, ba);
                else hipLaunchKernelGGL(gnn_roi_compress_kernel_batch<false>, dim3(maxN, nb), dim3(256), 0, s, ba);
                col += g->vis_d[i];
            }
            rc = forward_batch_impl(g, g->pool, nb, cx, s);
            if (rc) return rc;
            d_u = du;
        }
        return ASEP_OK;
    }
#endif
    for (int b = 0; b < n_pages; ++b) {
        const asep_gnn_page& q = pages[b];
        hipStream_t ls = L > 1 ? g->page_lanes[b % L]->s : s;
        BufferPool& lp = L > 1 ? g->page_lanes[b % L]->pool : g->pool;
        const std::string prefix = b ? "p" + std::to_string(b) + "/" : std::string();
        rc = visual_rois_dev(g, q.N, q.d_node_feat, prefix, q.d_regions, P, q.d_num_points, d_u, ls);
        if (rc) return rc;
        const float* d_ef = q.d_edge_feat;
        if (g->vise_total > 0 && q.E > 0) {
            rc = visual_edge_rois_dev(g, q.E, q.d_edge_feat, prefix, q.d_edge_regions, P, q.d_edge_num_points, d_efc, ls);
            if (rc) return rc;
            d_ef = d_efc;
            d_efc += (size_t)q.E * g->Ed;
        }
        rc = forward_impl(g, lp, q.N, q.E, q.d_edges, d_u, d_ef, q.R, q.d_relations, q.d_probs_out, ls);
        if (rc) return rc;
        d_u += (size_t)q.N * g->Uin;
    }
    if (L > 1) {
        for (int l = 0; l < L; ++l) {
            ASEP_HIP_CHECK(hipEventRecord(g->page_lanes[l]->done, g->page_lanes[l]->s));
            ASEP_HIP_CHECK(hipStreamWaitEvent(s, g->page_lanes[l]->done, 0));
        }
        g->stream = s;                                       // everything the lanes queued is ordered in front of what follows on s
    }
    return ASEP_OK;
    ASEP_GUARD_END
}

int asep_gnn_forward_visual(asep_gnn* g, int N, int E, const int32_t* edges, const float* node_feat, const float* edge_feat,
                            const float* image, int h, int w, const float* regions, int P, const int32_t* num_points,
                            const float* edge_regions, const int32_t* edge_num_points, int R, const int32_t* relations, float* probs_out) {
    ASEP_GUARD_BEGIN
    if (!g || !g->backbone) { set_error("asep_gnn_forward_visual: no backbone attached"); return ASEP_ERR_ARG; }
    const int U = g->Uin, ug = U - g->vis_total;
    if (N < 1 || E < 0 || R < 0 || h < 1 || w < 1 || P < 1 || !image || !regions || !num_points || (ug > 0 && !node_feat) ||
        (E > 0 && !edges) || (R > 0 && !probs_out)) { set_error("asep_gnn_forward_visual: bad argument"); return ASEP_ERR_ARG; }
    int rc;
    if (E > 0 && (rc = check_indices("interacting_nodes", edges, (size_t)E, N))) return rc;
    if (relations && R > 0 && (rc = check_indices("relations", relations, (size_t)R, N))) return rc;
    const size_t ne = (size_t)E * 2, nug = (size_t)N * ug, nf = edge_feat ? (size_t)E * g->Ed_fed : 0;
    const bool vedges = g->vise_total > 0 && E > 0;
    if (vedges && (!edge_regions || !edge_num_points)) { set_error("asep_gnn_forward_visual: this graph assigns visual features to edges: edge_regions / edge_num_points required"); return ASEP_ERR_ARG; }
    const size_t nr = relations ? (size_t)R * 2 : 0, no = (size_t)R * g->cfg.num_classes, ni = (size_t)h * w;
    const size_t nreg = (size_t)N * 2 * P;
    auto up = [&](const void* src, size_t bytes) -> void* {          // grow-only staging + async copy on the null stream
        void* d = g->host_stage.get(std::max<size_t>(bytes, 4));
        if (bytes && src) ASEP_HIP_CHECK_THROW(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, nullptr));
        return d;
    };
    g->host_stage.begin();
    int32_t* d_e = (int32_t*)up(edges, ne * 4);
    float* d_ug = (float*)up(node_feat, nug * 4);
    float* d_f = (float*)up(edge_feat, nf * 4);
    int32_t* d_r = (int32_t*)up(relations, nr * 4);
    float* d_img = (float*)up(image, ni * 4);
    float* d_reg = (float*)up(regions, nreg * 4);
    int32_t* d_np = (int32_t*)up(num_points, (size_t)N * 4);
    float* d_ereg = (float*)up(vedges ? edge_regions : nullptr, vedges ? (size_t)E * 2 * P * 4 : 0);
    int32_t* d_enp = (int32_t*)up(vedges ? edge_num_points : nullptr, vedges ? (size_t)E * 4 : 0);
    float* d_o = (float*)up(nullptr, no * 4);
    rc = asep_gnn_forward_visual_dev(g, N, E, ne ? d_e : nullptr, nug ? d_ug : nullptr, nf ? d_f : nullptr, d_img, h, w, d_reg,
                                     P, d_np, vedges ? d_ereg : nullptr, vedges ? d_enp : nullptr, R, nr ? d_r : nullptr, d_o, nullptr);
    if (rc) return rc;
    if (R > 0) ASEP_HIP_CHECK(hipMemcpyAsync(probs_out, d_o, no * sizeof(float), hipMemcpyDeviceToHost, nullptr));
    ASEP_HIP_CHECK(hipStreamSynchronize(nullptr));
    return ASEP_OK;
    ASEP_GUARD_END
}

int asep_gnn_get_node_features(asep_gnn* g, float* out, size_t max_floats) {
    ASEP_GUARD_BEGIN
    if (!g || !out || !g->d_u_cat) { set_error("asep_gnn_get_node_features: no visual forward has run"); return ASEP_ERR_ARG; }
    const size_t n = (size_t)g->N * g->Uin;
    if (max_floats < n) { set_error("asep_gnn_get_node_features: buffer too small"); return ASEP_ERR_ARG; }
    ASEP_HIP_CHECK(hipStreamSynchronize(g->stream));
    ASEP_HIP_CHECK(hipMemcpy(out, g->d_u_cat, n * sizeof(float), hipMemcpyDeviceToHost));
    return ASEP_OK;
    ASEP_GUARD_END
}

double asep_gnn_flops(const asep_gnn* g, int N, int E_corrected, int R) {
    if (!g) return 0.0;
    const asep_gnn_cfg& c = g->cfg;
    auto mlp_mac = [](const MlpW& m) {
        double s = 0;
        for (int l = 0; l < m.nl; ++l) s += (double)m.dims[l] * m.dims[l + 1];
        return s;
    };
    double per_edge = 0;
    if (c.attention_heads > 0)
        for (int k = 0; k < c.attention_heads; ++k) per_edge += mlp_mac(g->att_w[k].inter) + mlp_mac(g->att_w[k].att);
    else per_edge = mlp_mac(g->msg);
    double mac = (double)c.num_transition_steps * ((double)E_corrected * per_edge + (double)N * 4.0 * g->V * g->H);
    mac += (double)R * ((double)2 * g->H * c.cls_hidden1 + mlp_mac(g->cls_rest));
    return 2.0 * mac;
}

}  // extern "C"
