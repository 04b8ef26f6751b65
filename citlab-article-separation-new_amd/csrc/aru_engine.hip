// ARU-Net inference engine (host side): weight packing for the MFMA fragment order, buffer
// management and the layer schedule of ARU_v1.py:62-294.  Entry points: include/asep_hip.h.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <memory>

#include "aru_kernels.h"
#include "res8_kernels.h"
#include "res8v_kernels.h"
#include "bf16_kernels.h"
#include "res8w_kernels.h"
#include "convr_kernels.h"
#include "split_kernels.h"
#include "asep_common.h"

using namespace asep;

namespace {

struct Tensor {
    float* p = nullptr;    // bf: the buffer holds bf16 (2 bytes per element; native bf16 path), accessed through bp()
    int H = 0, W = 0, C = 0;
    bool bf = false;
    size_t count() const { return (size_t)H * W * C; }
    bf16_t* bp() const { return reinterpret_cast<bf16_t*>(p); }
};

// A convolution whose weights are packed as the A operand of v_mfma_f32_16x16x4_f32.
struct PackedConv {
    int kh = 0, kw = 0, cin = 0, cout = 0;
    bool c8 = false;       // Cin == 8: two taps per 16-slot chunk
    bool c12 = false;      // Cin == 12, 4x4 taps (attention conv2): dense rows of 48 floats = 3 chunks, no channel padding
    bool deconv = false;
    int groups = 0, mtiles = 0, nchunks = 0;
    float* d_w = nullptr;
    float* d_b = nullptr;
    float* d_wino = nullptr;   // Winograd F(2x2,3x3) transformed weights U = G g G^T, packed [g][pos][mtile][lane][4]
    float* d_wv = nullptr;     // scalar-operand filters of the vector-ALU kernels: deconv 16 -> 8 [tap][ci][co] (deconv8v_kernel),
                               // 4x4 conv 32 -> 1 [tap][ci] (conv_c1out_kernel)
    // native bf16 path (bf16_kernels.h): A fragments of v_mfma_f32_16x16x32_bf16, 8 bf16 per lane
    int bmode = -1;            // convb / deconvb MODE (0: Cin 8, 1: Cin 16, 2: Cin % 32 == 0); -1: not packed
    int bchunks = 0;
    bf16_t* d_wb = nullptr;    // conv: [chunk][mtile][lane][8]; deconv: MODE 2 [G][tap][mtile][lane][8], MODE 1 [frag 0..5][mtile][lane][8]
    bf16_t* d_wb8 = nullptr;   // deconv 16 -> 8 (level 0): the three class-pair fragments of deconvb8_kernel [3][lane][8]
    // fp32 with split products (split_kernels.h): the filter as three bf16 parts, [chunk][part h, m, l][mtile][lane][8]
    int smode = -1;            // 1: Cin 12 / 16 (chunk = two taps), 2: Cin % 32 == 0 (chunk = tap x 32 channels); -1: not packed
    bf16_t* d_ws = nullptr;
    bf16_t* d_ws16 = nullptr;  // 3x3, Cin % 16 == 0, Cin >= 32: stages of 16 channels, chunk = two taps (convs16_kernel): [stage][chunk 5][part][mtile][lane][8]
};

struct DirectConv {        // Cin == 1 first layers
    int k = 0, cout = 0;
    float* d_w = nullptr;  // [k*k][cout]
    float* d_b = nullptr;
};

int upload(const std::vector<float>& h, float** d) {
    ASEP_HIP_CHECK(hipMalloc((void**)d, std::max<size_t>(h.size(), 4) * sizeof(float)));
    ASEP_HIP_CHECK(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return ASEP_OK;
}

}  // namespace

struct asep_aru {
    asep_aru_cfg cfg{};
    std::map<std::string, PackedConv> convs;   // keyed by variable scope, e.g. "aru_net/featMapG/unet_down_1/convR_0"
    struct ResB { int C = 0; bf16_t* d_w = nullptr; float* d_b = nullptr; };
    std::map<std::string, ResB> resb;          // bf16 path: fused residual-block tails (8- / 16-channel levels), keyed by block scope
    // bf16 path, whole level-0 blocks in one kernel each (res8b_kernel): pixel-pair A fragments
    bf16_t* d_r8b_down_w = nullptr;  // [3 convs][3 ky][64][8]
    float* d_r8b_down_b = nullptr;   // [3][8]
    bf16_t* d_r8b_up_w1 = nullptr;   // conv1 of unet_up_0 [3 ky][2 halves][64][8]
    bf16_t* d_r8f_up_w1 = nullptr;   // the same for res8f_kernel's planar input tile [3 ky][2 sources][64][8]: k = 8 (window pixel kk) + channel of the source
    bf16_t* d_r8f_down_w1 = nullptr; // conv1 of unet_down_0 as ONE pair fragment [64][8] (k = window row / column, res8f_kernel)
    float* d_r8b_down_w1r = nullptr; // the same filter [9][8] as fp32 values rounded to bfloat16 (border tiles, res8b_tile)
    bool use_res32 = true;           // ASEP_BF_RES32=0: the 32-channel residual tails layer by layer (convb_kernel)
    int walk_mode = 1;               // ASEP_BF_WALK: 1 both level-0 blocks on the walkers (default), 2 the UP block only, 0 neither
    bool use_deconvs = true;         // ASEP_SPLIT_DECONV=0: the deconvolutions of the f32s engine on the fp32 MFMA (deconv_mfma_kernel) instead of split products
    bool use_convr = true;           // ASEP_BF_CONVR=0: the 64 -> 64 layers on convb_kernel instead of the register-resident form (convr_kernels.h)
    unsigned char* d_zero_trash = nullptr;   // 16 zero bytes (padding source of convr_kernel) + 4 KB behind them that nobody reads (its dump)
    int fused_act = 0;               // bf16 engine, elu / leaky RESIDUAL graphs (round 6): the level-0 blocks and the 16-channel tails on the general fused forms
                                     // res8b_kernel<UP, ACT> / resb_tail_kernel<16, ACT> (activation on the fp32 sums before the rounding); 0: ReLU graph, or layer by layer
    bool use_walk = true;            // ASEP_BF_WALK=0: the level-0 blocks as 16 x 32 tiles (res8f_kernel) instead of the column-strip walkers (res8w_kernels.h)
    bf16_t* d_r8b_up_w = nullptr;    // [3][3][64][8]
    float* d_r8b_up_b = nullptr;     // [3][8]
    float* d_r8b_up_b1 = nullptr;    // [8]
    DirectConv det_first, att_first;
    // fused level-0 residual blocks (feat_root == 8, res_depth == 3): pixel-pair MFMA fragments
    float* d_r8_down_wr = nullptr;   // [3][6][64][4]
    float* d_r8_down_br = nullptr;   // [3][8]
    float* d_r8_up_w1 = nullptr;     // unet_up_0/conv1 as two 8-channel pixel-pair passes [2][6][64][4]
    float* d_r8_up_wr = nullptr;     // [3][6][64][4]
    float* d_r8_up_br = nullptr;     // [3][8]
    float* d_r8_up_b1 = nullptr;     // [8]
    // the same filters in scalar layout for the fp32 vector-ALU kernels (res8v_kernels.h)
    float* d_r8v_down_wr = nullptr;  // [3][R8V_FILTER]
    float* d_r8v_up_w1 = nullptr;    // [2][R8V_FILTER]
    float* d_r8v_up_wr = nullptr;    // [3][R8V_FILTER]
    bool r8_valu = true;             // fp32 only; ASEP_R8_VALU=0 runs the fp32 MFMA variants instead
    bool use_fused8 = true;          // ASEP_FUSED8=0 falls back to the layer-by-layer kernels
    bool fused8_wanted = true;       // what ASEP_FUSED8 said (use_fused8 is also switched off for the graph variants)
    bool fused8_var = false;         // elu / leaky RESIDUAL graphs: the level-0 blocks on res8v_*_kernel<activation> (round 4)
    float* d_att_head = nullptr;     // A fragment of attPart/conv1 for att_head_kernel (12 output channels, 4x4 taps)
    bf16_t* d_att_headb = nullptr;   // the same as ONE bf16 fragment [64][8] (att_headb_kernel, bf16 path)
    float* d_logit_w = nullptr;
    float* d_logit_b = nullptr;
    float* d_logit_wd = nullptr;     // two classes: [16][feat_root] class-1 minus class-0 filter + the bias difference (combine_kernel behind a soft-max)
    float* d_stats = nullptr;      // mvn {mean, 1/std}
    double* d_sums = nullptr;
    // A lane = one in-order chain of launches (stream + its buffer pool + a side stream for the attention branch).
    // A batch of pages is split over the lanes so that two independent chains fill each other's launch tails.
    struct Lane {
        hipStream_t s = nullptr;         // lane 0 uses the caller's stream
        bool own_stream = false;
        hipStream_t side = nullptr;
        hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_begin = nullptr, ev_done = nullptr;
        hipStream_t bside = nullptr;     // the border tiles of the strip walkers run beside the walkers (run_res8w); not the attention branch's
        hipEvent_t ev_bfork = nullptr, ev_bjoin = nullptr;   // stream: its chain is busy while the level-0 down block runs
        BufferPool pool;
        ~Lane() {
            if (ev_fork) (void)hipEventDestroy(ev_fork);
            if (ev_bfork) (void)hipEventDestroy(ev_bfork);
            if (ev_bjoin) (void)hipEventDestroy(ev_bjoin);
            if (ev_join) (void)hipEventDestroy(ev_join);
            if (ev_begin) (void)hipEventDestroy(ev_begin);
            if (ev_done) (void)hipEventDestroy(ev_done);
            if (side) (void)hipStreamDestroy(side);
            if (bside) (void)hipStreamDestroy(bside);
            if (own_stream && s) (void)hipStreamDestroy(s);
        }
    };
    std::vector<std::unique_ptr<Lane>> lanes;
    Lane* cur = nullptr;
    int last_nl = 0;                     // lanes of the previous call: a call with another count releases every lane's arena first (below)
    bool lanes_forced = false;           // ASEP_LANES given: split any batch of >= 2 pages
    int num_lanes = 2;                   // page lanes of a batch call (ASEP_LANES overrides).  Two lanes since round 6: two independent chains fill each other's launch
                                         // tails (r4j: 1 / 2 / 3 / 4 lanes = 119.3 / 121.2 / 120.3 / 115.8 pages/s fp32; round 5: 1 / 2 / 3 = 467 / 477 / 478 bf16, 137.6 / 139.8
                                         // f32s; the round-5 driver run: f32s 135.8 -> 138.4, bf16 469.6 -> 487.5, outputs bit-identical).  A lane takes at least four pages, so
                                         // calls of fewer than eight pages stay on one lane.  Two launches of one kernel then share the chip and each takes about twice as long,
                                         // so a per-launch figure must not be taken from that schedule: while a handle records launch times (asep_aru_profile, any mode) its
                                         // calls run on ONE lane -- the bench line's roofline block and rocprofv3's averages of the event-timed steps describe a kernel, not the
                                         // sharing (DESIGN_LESSONS 47, 49)
    std::map<std::string, Tensor> endpoints;
    hipStream_t stream = nullptr;
    std::vector<void*> owned;

    // optional per-launch timing with HIP events on the launch stream (bench.py roofline leg)
    struct ProfRec { int kid; double flops, bytes, xflops; hipEvent_t a, b; };
    int num_cus = 256;
    BufferPool host_stage;         // device staging of the host-pointer entry point (grow-only)
    hipStream_t host_stream = nullptr;   // transfers + forward of the host-pointer entry point (created on first use)
    bool use_xcd_sched = true;     // ASEP_XCD_SCHED=0: identity tile order (persistent kernels: no tile table; one-shot kernels: no XCD bands)
    std::map<std::string, const int32_t*> sched_cache;
    bool bf16 = false;             // cfg.compute_dtype == 1: native bf16 data path (bf16_kernels.h): bf16 activations in HBM / LDS,
                                   // v_mfma_f32_16x16x32_bf16 with fp32 accumulation; fp32 image in, fp32 probabilities out
    bool split = false;            // cfg.compute_dtype == 2: fp32 tensors and accumulation, every product of the convolutions with >= 12 input channels
                                   // as six bf16 x bf16 partial products (split_kernels.h).  Level 0 stays on the vector-ALU blocks (DESIGN_LESSONS 32)
    bool use_c12 = true;           // ASEP_C12=0: 12-channel inputs padded to a 16-channel group (read when the weights are packed)
    bool fuse_pool = true;         // ASEP_FUSE_POOL=0: separate maxpool2_kernel after every conv
    bool fuse_act = true;          // ASEP_FUSE_ACT=0: elu / leaky of the graph variants as a separate act_kernel pass behind every conv
    bool profiling = false;
    bool prof_detail = false;      // per-layer names (scope + spatial size) instead of per-kernel names
    bool prof_in_situ = false;     // keep the attention side stream while recording (times include what shares the chip)
    std::vector<std::string> prof_names;
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_next = 0;

    ~asep_aru() {
        for (void* p : owned)
            if (p) (void)hipFree(p);
        for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
        if (host_stream) (void)hipStreamDestroy(host_stream);
    }
    hipEvent_t next_event() {
        if (ev_next == ev_pool.size()) {
            hipEvent_t e;
            ASEP_HIP_CHECK_THROW(hipEventCreate(&e));
            ev_pool.push_back(e);
        }
        return ev_pool[ev_next++];
    }
    int prof_kid(const std::string& name) {
        for (size_t i = 0; i < prof_names.size(); ++i)
            if (prof_names[i] == name) return (int)i;
        prof_names.push_back(name);
        return (int)prof_names.size() - 1;
    }
    int feat(int l) const { return cfg.feat_root << l; }
};

namespace {

// Brackets one kernel launch with two events when profiling is on (no-op otherwise).
// Names are the kernels' rocprofv3 names without "void ", "asep::", blanks and the argument list (every template
// argument spelled out), so that scripts/roofline_from_profiles.py can join the two sources without a name table.
struct ProfScope {
    asep_aru* m;
    hipEvent_t a = nullptr, b = nullptr;
    bool on = false;
    double flops;
    double xflops = -1;        // EXECUTED FLOPs where they differ from the algorithmic credit (< 0: the same)
    double bytes = 0;          // ALGORITHMIC HBM bytes of the launch: every input tensor read once, every output written once, the filter
                               // once (SURVEY.md section 8d per-unit figure x the units of the launch); set by the launcher
    std::string name, detail;
    ProfScope(asep_aru* m_, const std::string& name_, double flops_, const std::string& detail_ = std::string())
        : m(m_), flops(flops_), name(name_), detail(detail_) {
        if (!m->profiling) return;
        on = true;
        a = m->next_event();
        b = m->next_event();
        ASEP_HIP_CHECK_THROW(hipEventRecord(a, m->stream));
    }
    void set_name(const std::string& n) { name = n; }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(b, m->stream);
        m->prof_recs.push_back({m->prof_kid(m->prof_detail && !detail.empty() ? name + " " + detail : name), flops, bytes, xflops < 0 ? flops : xflops, a, b});
    }
};
std::string targs(std::initializer_list<std::string> l) {
    std::string s = "<";
    for (const std::string& x : l) s += (s.size() > 1 ? "," : "") + x;
    return s + ">";
}
inline std::string tb(bool b) { return b ? "true" : "false"; }
inline std::string ti(int i) { return std::to_string(i); }

// ---- weight packing -----------------------------------------------------------------------------
int pack_conv_bf(asep_aru* m, PackedConv& pc, const HostTensor& w);   // bf16 fragments (native bf16 path), defined further down
int pack_conv_split(asep_aru* m, PackedConv& pc, const HostTensor& w);   // three-part bf16 fragments (split_kernels.h)
int pack_deconv_split(asep_aru* m, PackedConv& pc, const HostTensor& w); // the same for deconvs_kernel

// conv   W[kh][kw][cin][cout]  (layers.py:219);  deconv W[kh][kw][cout][cin] (layers.py:352, ARU_v1.py:257)
int pack_conv(asep_aru* m, const std::map<std::string, HostTensor>& blob, const std::string& scope,
              const char* bias_name, bool deconv) {
    auto wi = blob.find(scope + "/weights");
    auto bi = blob.find(scope + "/" + bias_name);
    if (wi == blob.end() || bi == blob.end()) {
        set_error("weights: missing tensor %s/{weights,%s}", scope.c_str(), bias_name);
        return ASEP_ERR_WEIGHTS;
    }
    const HostTensor& w = wi->second;
    if (w.dims.size() != 4) {
        set_error("weights: %s/weights must have rank 4", scope.c_str());
        return ASEP_ERR_WEIGHTS;
    }
    PackedConv pc;
    pc.kh = w.dims[0];
    pc.kw = w.dims[1];
    pc.deconv = deconv;
    pc.cin = deconv ? w.dims[3] : w.dims[2];
    pc.cout = deconv ? w.dims[2] : w.dims[3];
    if ((int)bi->second.count() != pc.cout) {
        set_error("weights: %s bias has %zu elements, expected %d", scope.c_str(), bi->second.count(), pc.cout);
        return ASEP_ERR_WEIGHTS;
    }
    if (pc.cin % 4 != 0) {
        set_error("weights: %s has Cin=%d; the MFMA path needs Cin %% 4 == 0", scope.c_str(), pc.cin);
        return ASEP_ERR_UNSUPPORTED;
    }
    const int taps = pc.kh * pc.kw;
    pc.c8 = (!deconv && pc.cin == 8);
    pc.c12 = (!deconv && pc.cin == 12 && pc.kh == 4 && pc.kw == 4 && pc.cout <= 16 && m->use_c12);
    pc.mtiles = cdiv(pc.cout, 16);
    pc.groups = (pc.c8 || pc.c12) ? 1 : cdiv(pc.cin, 16);
    pc.nchunks = pc.c8 ? (taps + 1) / 2 : (pc.c12 ? pc.kh * (pc.kw * 12 / 16) : pc.groups * taps);
    auto W = [&](int tap, int ci, int co) -> float {
        if (ci >= pc.cin || co >= pc.cout || tap >= taps) return 0.f;
        return deconv ? w.data[((size_t)tap * pc.cout + co) * pc.cin + ci]
                      : w.data[((size_t)tap * pc.cin + ci) * pc.cout + co];
    };
    std::vector<float> pk((size_t)pc.nchunks * pc.mtiles * 64 * 4);
    for (int ch = 0; ch < pc.nchunks; ++ch)
        for (int mt = 0; mt < pc.mtiles; ++mt)
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r) {
                    const int kk = lane >> 4, co = mt * 16 + (lane & 15);
                    int tap, ci;
                    if (pc.c8) {
                        tap = 2 * ch + (kk >> 1);
                        ci = 4 * (kk & 1) + r;
                    } else if (pc.c12) {
                        const int cpr = pc.kw * 12 / 16, ky = ch / cpr, flat = (ch % cpr) * 16 + 4 * kk + r;   // float of the row's run
                        tap = ky * pc.kw + flat / 12;
                        ci = flat % 12;
                    } else {
                        const int g = ch / taps;
                        tap = ch % taps;
                        ci = 16 * g + 4 * kk + r;
                    }
                    pk[(((size_t)ch * pc.mtiles + mt) * 64 + lane) * 4 + r] = W(tap, ci, co);
                }
    int rc = upload(pk, &pc.d_w);
    if (rc) return rc;
    rc = upload(bi->second.data, &pc.d_b);
    if (rc) return rc;
    m->owned.push_back(pc.d_w);
    m->owned.push_back(pc.d_b);
    if (!deconv && pc.kh == 3 && pc.kw == 3 && pc.cin % 16 == 0 && pc.cout % 16 == 0) {
        // U[a][b] = sum_ij G[a][i] g[i][j] G[b][j], G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] (double accumulation)
        static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
        std::vector<float> wk((size_t)pc.groups * 16 * pc.mtiles * 64 * 4);
        for (int g = 0; g < pc.groups; ++g)
            for (int pos = 0; pos < 16; ++pos)
                for (int mt = 0; mt < pc.mtiles; ++mt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int r = 0; r < 4; ++r) {
                            const int ci = 16 * g + 4 * (lane >> 4) + r, co = mt * 16 + (lane & 15);
                            const int ua = pos >> 2, ub = pos & 3;
                            double u = 0;
                            for (int i = 0; i < 3; ++i)
                                for (int j2 = 0; j2 < 3; ++j2) u += G[ua][i] * (double)W(i * 3 + j2, ci, co) * G[ub][j2];
                            wk[((((size_t)g * 16 + pos) * pc.mtiles + mt) * 64 + lane) * 4 + r] = (float)u;
                        }
        rc = upload(wk, &pc.d_wino);
        if (rc) return rc;
        m->owned.push_back(pc.d_wino);
    }
    if (deconv && pc.kh == 3 && pc.kw == 3 && pc.cin == 16 && pc.cout == 8) {
        std::vector<float> wv;
        for (int tap = 0; tap < 9; ++tap)
            for (int ci = 0; ci < 16; ++ci)
                for (int co = 0; co < 8; ++co) wv.push_back(W(tap, ci, co));
        rc = upload(wv, &pc.d_wv);
        if (rc) return rc;
        m->owned.push_back(pc.d_wv);
    }
    if (!deconv && pc.kh == 4 && pc.kw == 4 && pc.cin == 32 && pc.cout == 1) {
        std::vector<float> wv;
        for (int tap = 0; tap < 16; ++tap)
            for (int ci = 0; ci < 32; ++ci) wv.push_back(W(tap, ci, 0));
        rc = upload(wv, &pc.d_wv);
        if (rc) return rc;
        m->owned.push_back(pc.d_wv);
    }
    if (m->bf16) {
        rc = pack_conv_bf(m, pc, w);
        if (rc) return rc;
    }
    if (m->split && !deconv) {
        rc = pack_conv_split(m, pc, w);
        if (rc) return rc;
    }
    if (m->split && deconv) {
        rc = pack_deconv_split(m, pc, w);
        if (rc) return rc;
    }
    m->convs[scope] = pc;
    return ASEP_OK;
}

int pack_direct(asep_aru* m, const std::map<std::string, HostTensor>& blob, const std::string& scope,
                DirectConv* dc) {
    auto wi = blob.find(scope + "/weights");
    auto bi = blob.find(scope + "/biases");
    if (wi == blob.end() || bi == blob.end()) {
        set_error("weights: missing tensor %s/{weights,biases}", scope.c_str());
        return ASEP_ERR_WEIGHTS;
    }
    const HostTensor& w = wi->second;
    if (w.dims.size() != 4 || w.dims[2] != 1 || w.dims[0] != w.dims[1]) {
        set_error("weights: %s must be [k,k,1,cout]", scope.c_str());
        return ASEP_ERR_UNSUPPORTED;
    }
    dc->k = w.dims[0];
    dc->cout = w.dims[3];
    int rc = upload(w.data, &dc->d_w);
    if (rc) return rc;
    rc = upload(bi->second.data, &dc->d_b);
    if (rc) return rc;
    m->owned.push_back(dc->d_w);
    m->owned.push_back(dc->d_b);
    return ASEP_OK;
}

struct TileDims { int tx, ty, begin; };

// work unit -> tile table of the PERSISTENT kernels (res8v_*, res32_tail_kernel: resident blocks walk the units with a grid stride) for the
// problems' tile grids `probs` (tile numbers begin + ty * tx + x, `total` tiles in all): the tiles of every problem in 4 x 8 super-tile
// order, cut into eight chunks, unit k = the (k / 8)-th tile of chunk k mod 8 (block b runs on XCD b mod 8).  nullptr on failure: identity.
const int32_t* xcd_schedule(asep_aru* m, const std::vector<TileDims>& probs, int total) {
    std::string key = "u";
    for (const TileDims& q : probs) key += ":" + std::to_string(q.tx) + "x" + std::to_string(q.ty);
    auto it = m->sched_cache.find(key);
    if (it != m->sched_cache.end()) return it->second;
    std::vector<int32_t> order;
    order.reserve(total);
    for (const TileDims& q : probs)
        for (int gc = 0; gc * 8 < q.tx; ++gc)
            for (int gr = 0; gr * 4 < q.ty; ++gr)
                for (int r = 0; r < 4; ++r)
                    for (int c = 0; c < 8; ++c) {
                        const int ty = gr * 4 + r, tx = gc * 8 + c;
                        if (ty < q.ty && tx < q.tx) order.push_back(q.begin + ty * q.tx + tx);
                    }
    if ((int)order.size() != total) return nullptr;
    std::vector<int32_t> sched(total);
    int off[9];
    off[0] = 0;
    for (int x = 0; x < 8; ++x) off[x + 1] = off[x] + (total - x + 7) / 8;
    for (int k = 0; k < total; ++k) sched[k] = order[off[k % 8] + k / 8];
    int32_t* d = nullptr;
    if (hipMalloc((void**)&d, (size_t)total * sizeof(int32_t)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, sched.data(), (size_t)total * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    m->owned.push_back(d);
    m->sched_cache[key] = d;
    return d;
}

// the persistent fp32 level-0 kernels: nblocks resident blocks, unit k = block + i * nblocks
const int32_t* tile_schedule(asep_aru* m, const Res8Args& a, int nblocks, int unit_h) {
    if (!m->use_xcd_sched || nblocks % 8 != 0 || a.total_tiles < 2 * nblocks) return nullptr;
    std::vector<TileDims> probs;
    for (int i = 0; i < a.nprob; ++i) probs.push_back({a.p[i].tiles_x, (a.p[i].H + unit_h - 1) / unit_h, a.p[i].tile_begin});
    return xcd_schedule(m, probs, a.total_tiles);
}

// one-shot kernels (one block per tile): XCD bands (XcdMap, aru_kernels.h) from a few waves of blocks per XCD on.  *n_units = blocks to launch
// along x (the grid is padded to eight equal chunks; surplus blocks leave at once).
XcdMap oneshot_map(asep_aru* m, int total, int* n_units) {
    XcdMap xm{0, total};
    if (n_units) *n_units = total;
    if (!m->use_xcd_sched || total < 8 * 64) return xm;
    xm.chunk = (total + 7) / 8;
    if (n_units) *n_units = 8 * xm.chunk;
    return xm;
}

// ---- kernel launchers: every launch covers one layer of a LIST of problems (pages x scales) -----------------
typedef std::vector<Tensor> TL;

Tensor new_tensor(asep_aru* m, int H, int W, int C) {
    Tensor t;
    t.H = H; t.W = W; t.C = C;
    t.p = (float*)m->cur->pool.get(t.count() * sizeof(float));
    return t;
}

inline double tbytes(const Tensor& t) { return (double)t.count() * (t.bf ? 2.0 : 4.0); }

std::string dims_of(const TL& l) {
    std::string d;
    for (size_t i = 0; i < l.size() && i < 3; ++i) d += (i ? "+" : "") + std::to_string(l[i].H) + "x" + std::to_string(l[i].W);
    if (l.size() > 3) d += "+..(" + std::to_string(l.size()) + ")";
    return d;
}

// launches conv_mfma_kernel<...> and gives the profiler record that instantiation's exact name
#define ASEP_CONV_LAUNCH(KH_, KW_, MT_, C8_, TH_, DB_, BF_, C12_, MB_)                                                   \
    do {                                                                                                                 \
        ps.set_name("conv_mfma_kernel" + targs({ti(KH_), ti(KW_), ti(MT_), tb(C8_), ti(TH_), tb(DB_), tb(BF_), tb(C12_), ti(MB_)})); \
        hipLaunchKernelGGL((conv_mfma_kernel<KH_, KW_, MT_, C8_, TH_, DB_, BF_, C12_, MB_>), grid, dim3(256), 0, s, a);  \
    } while (0)

template <int KH, int KW>
void launch_conv_k(asep_aru* m, const PackedConv& pc, const ConvArgs& a, int total_tiles, double flops, double bytes,
                   const std::string& scope, const TL& in0, bool big_tile) {
    const int mt = pc.c8 ? 1 : (pc.mtiles % 4 == 0 ? 4 : (pc.mtiles % 2 == 0 ? 2 : 1));
    dim3 grid(total_tiles, pc.mtiles / mt);                  // (total_tiles = the schedule's units: padded to 8 when grid.y > 1)
    ProfScope ps(m, "conv_mfma_kernel", flops, scope + " " + dims_of(in0) + " " + std::to_string(pc.cin) + "->" + std::to_string(pc.cout));
    ps.bytes = bytes;
    hipStream_t s = m->stream;
    const bool res_op = a.p[0].res != nullptr;
    const bool has_res = res_op || KH != 3;      // (the four-blocks-per-CU variant exists for 3x3 only: 4x4 needs 140 VGPRs)
    if constexpr (KW == 4) {
        if (pc.c12) {                                        // one m-tile, one channel group: 16 x 32 tiles, single LDS buffer
            ASEP_CONV_LAUNCH(KH, KW, 1, false, 16, false, false, true, 2);
            return;
        }
    }
    if constexpr (KH == 3) {
        if (pc.c8 && big_tile && !res_op) {                  // 8 -> 16 (level-1 conv1): 16 x 32-pixel blocks, four per CU
            ASEP_CONV_LAUNCH(3, 3, 1, true, 16, false, false, false, 4);
            return;
        }
    }
    if (pc.c8) ASEP_CONV_LAUNCH(KH, KW, 1, true, CONV_TH, true, false, false, 2);
    else if (mt == 1 && big_tile && !has_res) ASEP_CONV_LAUNCH(3, 3, 1, false, 16, false, false, false, 4);
    else if (mt == 1 && big_tile) ASEP_CONV_LAUNCH(KH, KW, 1, false, 16, false, false, false, 2);
    else if (mt == 4) ASEP_CONV_LAUNCH(KH, KW, 4, false, CONV_TH, true, false, false, 2);
    else if (mt == 2 && !res_op) ASEP_CONV_LAUNCH(KH, KW, 2, false, CONV_TH, true, false, false, 3);   // no residual prefetch: three blocks per CU
    else if (mt == 2) ASEP_CONV_LAUNCH(KH, KW, 2, false, CONV_TH, true, false, false, 2);
    else ASEP_CONV_LAUNCH(KH, KW, 1, false, CONV_TH, true, false, false, 2);
}

enum PoolKind { POOL_MAX, POOL_AVG_C1, POOL_CHANSUM };
TL run_pool(asep_aru* m, const TL& in, PoolKind kind);

// launches convs_kernel<...> under that instantiation's name
#define ASEP_CONVS_LAUNCH(KH_, KW_, C16_, MT_, TH_, MB_)                                                              \
    do {                                                                                                              \
        ps.set_name("convs_kernel" + targs({ti(KH_), ti(KW_), tb(C16_), ti(MT_), ti(TH_), ti(MB_)}));                 \
        hipLaunchKernelGGL((convs_kernel<KH_, KW_, C16_, MT_, TH_, MB_>), grid, dim3(256), 0, m->stream, a);          \
    } while (0)

// a conv layer on the split-product kernel (split_kernels.h): same operands and results as run_conv's fp32 kernels
TL run_conv_split(asep_aru* m, const PackedConv& pc, const std::string& scope, const TL& in0, const TL* in1, bool relu_in, bool relu_out,
                  const TL* res, TL* pooled, bool keep_full, int act) {
    const bool fuse_pool = pooled && m->fuse_pool;
    TL out;
    if (keep_full || !fuse_pool)
        for (const Tensor& t : in0) out.push_back(new_tensor(m, t.H, t.W, pc.cout));
    if (fuse_pool) {
        pooled->clear();
        for (const Tensor& t : in0) pooled->push_back(new_tensor(m, (t.H + 1) / 2, (t.W + 1) / 2, pc.cout));
    }
    const bool c16 = pc.smode == 1;
    // (measured, 4 pages per launch: 32 -> 16 1230 -> 930 us on convs16_kernel, but 64 -> 64 378 -> 405 and 32 -> 32 431 -> 534: with more than one
    //  m-tile the barrier per chunk and the 16-channel stages cost more than the shared fragments save: one-m-tile layers only.
    //  The same kernel with the fragments fetched per wave (ALDS = false; 16-channel stages + halo prefetch only) spills and was slower still:
    //  521 / 527 us for those two layers -- not instantiated)
    const bool alds = pc.d_ws16 && pc.mtiles == 1;
    const int mt = (pc.mtiles % 4 == 0 && !c16) ? 4 : (pc.mtiles % 2 == 0 ? 2 : 1);
    const int th = (c16 && pc.kh == 3 && mt == 1) ? 16 : 8;
    for (size_t b0 = 0; b0 < in0.size(); b0 += MAXP) {
        const size_t b1 = std::min(in0.size(), b0 + MAXP);
        ConvArgs a{};
        int tiles = 0;
        double flops = 0, bytes = (double)pc.kh * pc.kw * pc.cin * pc.cout * 4.0;
        for (size_t i = b0; i < b1; ++i) {
            ConvProb& p = a.p[i - b0];
            p.in0 = in0[i].p; p.in1 = in1 ? (*in1)[i].p : nullptr; p.res = res ? (*res)[i].p : nullptr;
            p.out = out.empty() ? nullptr : out[i].p;
            p.pool = fuse_pool ? (*pooled)[i].p : nullptr;
            p.H = p.Ho = in0[i].H; p.W = p.Wo = in0[i].W;
            p.tiles_x = (in0[i].W + 31) / 32;
            p.tile_begin = tiles;
            tiles += p.tiles_x * ((in0[i].H + th - 1) / th);
            flops += 2.0 * in0[i].H * in0[i].W * pc.kh * pc.kw * (double)pc.cin * pc.cout;
            bytes += tbytes(in0[i]) + (in1 ? tbytes((*in1)[i]) : 0.0) + (res ? tbytes((*res)[i]) : 0.0) + (out.empty() ? 0.0 : tbytes(out[i])) +
                     (fuse_pool ? tbytes((*pooled)[i]) : 0.0);
        }
        a.nprob = (int)(b1 - b0);
        a.total_tiles = tiles;
        a.c0 = in0[0].C; a.c1 = in1 ? (*in1)[0].C : 0;
        a.wpk = (const f32x4*)(alds ? pc.d_ws16 : pc.d_ws); a.bias = pc.d_b;
        a.cout = pc.cout; a.mtiles = pc.mtiles; a.groups = alds ? pc.cin / 16 : pc.cin / 32;
        a.relu_in = relu_in; a.relu_out = relu_out; a.act = act;
        a.skip_full = fuse_pool && !keep_full;
        int units = tiles;
        a.xm = oneshot_map(m, tiles, &units);
        dim3 grid(units, pc.mtiles / mt);
        TL sub(in0.begin() + b0, in0.begin() + b1);
        ProfScope ps(m, "convs_kernel", flops, scope + " " + dims_of(sub) + " " + std::to_string(pc.cin) + "->" + std::to_string(pc.cout));
        ps.bytes = bytes;
        if (alds) {                                          // (one m-tile: mt == 1)
            ps.set_name("convs16_kernel<1,8,3,true>");
            hipLaunchKernelGGL((convs16_kernel<1, 8, 3, true>), grid, dim3(256), 0, m->stream, a);
        } else if (pc.kh == 3) {
            if (c16 && th == 16) ASEP_CONVS_LAUNCH(3, 3, true, 1, 16, 2);
            else if (c16 && mt == 2) ASEP_CONVS_LAUNCH(3, 3, true, 2, 8, 3);
            else if (c16) ASEP_CONVS_LAUNCH(3, 3, true, 1, 8, 3);
            else if (mt == 4) ASEP_CONVS_LAUNCH(3, 3, false, 4, 8, 2);
            else if (mt == 2) ASEP_CONVS_LAUNCH(3, 3, false, 2, 8, 2);
            else ASEP_CONVS_LAUNCH(3, 3, false, 1, 8, 2);
        } else {
            if (c16 && mt == 2) ASEP_CONVS_LAUNCH(4, 4, true, 2, 8, 2);
            else if (c16) ASEP_CONVS_LAUNCH(4, 4, true, 1, 8, 2);
            else { set_error("internal: conv %s (4x4, %d input channels) was packed for the split-product kernel", scope.c_str(), pc.cin); throw ArgError(); }
        }
    }
    if (pooled && !fuse_pool) *pooled = run_pool(m, out, POOL_MAX);
    return out;
}

// stride-1 SAME conv on the (optionally concatenated) inputs of every problem

// pooled != nullptr: also produce maxpool2 of the output.  The direct kernels and the register-resident Winograd kernel take
// the 2x2 max in their epilogue (ConvProb::pool); the others are followed by maxpool2_kernel.  keep_full = false: the caller
// reads only the pooled tensor (attention CNN), so the unpooled one is not stored (and the returned list is empty) when the
// pool is fused.
TL run_conv(asep_aru* m, const std::string& scope, const TL& in0, const TL* in1, bool relu_in, bool relu_out,
            const TL* res, TL* pooled = nullptr, bool keep_full = true, int act = 0) {
    auto it = m->convs.find(scope);
    if (it == m->convs.end()) { set_error("internal: conv %s not packed", scope.c_str()); throw ArgError(); }
    const PackedConv& pc = it->second;
    const int cin = in0[0].C + (in1 ? (*in1)[0].C : 0);
    if (cin != pc.cin) {
        set_error("internal: conv %s expects Cin=%d, got %d", scope.c_str(), pc.cin, cin);
        throw ArgError();
    }
    if (!((pc.kh == 3 && pc.kw == 3) || (pc.kh == 4 && pc.kw == 4))) {
        set_error("conv %s: unsupported kernel size %dx%d", scope.c_str(), pc.kh, pc.kw);
        throw ArgError();
    }
    if (pc.d_wv && pc.cout == 1 && m->r8_valu && !in1 && !res && !pooled) {
        // single output channel (attention conv4): one pixel per thread on the vector ALU
        TL out1;
        for (const Tensor& t : in0) out1.push_back(new_tensor(m, t.H, t.W, 1));
        for (size_t b0 = 0; b0 < in0.size(); b0 += MAXP) {
            const size_t b1 = std::min(in0.size(), b0 + MAXP);
            ConvArgs a{};
            int tiles = 0;
            double flops = 0, bytes = 0;
            for (size_t i = b0; i < b1; ++i) {
                ConvProb& p = a.p[i - b0];
                p.in0 = in0[i].p; p.out = out1[i].p;
                p.H = p.Ho = in0[i].H; p.W = p.Wo = in0[i].W;
                p.tiles_x = cdiv(in0[i].W, C1O_T);
                p.tile_begin = tiles;
                tiles += p.tiles_x * cdiv(in0[i].H, C1O_T);
                flops += 2.0 * in0[i].H * in0[i].W * 16.0 * pc.cin;
                bytes += tbytes(in0[i]) + tbytes(out1[i]);
            }
            a.nprob = (int)(b1 - b0);
            a.total_tiles = tiles;
            a.c0 = pc.cin; a.cout = 1;
            a.wpk = (const f32x4*)pc.d_wv; a.bias = pc.d_b;
            a.relu_in = relu_in; a.relu_out = relu_out; a.act = act;
            int units = tiles;
            a.xm = oneshot_map(m, tiles, &units);
            ProfScope ps(m, "conv_c1out_kernel", flops, scope);
            ps.bytes = bytes;
            hipLaunchKernelGGL(conv_c1out_kernel, dim3(units), dim3(256), 0, m->stream, a);
        }
        return out1;
    }
    if (m->split && pc.d_ws) return run_conv_split(m, pc, scope, in0, in1, relu_in, relu_out, res, pooled, keep_full, act);
    const bool wino = pc.d_wino && pc.mtiles > 1;            // (Winograd pays from 32 output channels: DESIGN_LESSONS 4, 16)
    const int wino_mt = pc.mtiles % 4 == 0 ? 4 : (pc.mtiles % 2 == 0 ? 2 : 1);
    const bool fuse_pool = pooled && m->fuse_pool && pc.cout % 4 == 0 && (!wino || wino_mt <= 2);
    TL out;
    if (keep_full || !fuse_pool)
        for (const Tensor& t : in0) out.push_back(new_tensor(m, t.H, t.W, pc.cout));
    if (fuse_pool) {
        pooled->clear();
        for (const Tensor& t : in0) pooled->push_back(new_tensor(m, cdiv(t.H, 2), cdiv(t.W, 2), pc.cout));
    }
    // single channel group, one 16-channel output tile: 16 x 32 pixel blocks, single LDS buffer (more MFMA work per
    // block against the fixed load latency of these short blocks)
    // (two channel groups only for the residual-free 3x3 variant: four blocks per CU hide the refill of its single LDS buffer)
    const bool big_tile = !wino && (!pc.c8 || (pc.kh == 3 && !res)) && (pc.groups == 1 || (pc.groups == 2 && !res && pc.kh == 3)) && pc.mtiles == 1;
    const int th = big_tile ? 16 : CONV_TH;
    for (size_t b0 = 0; b0 < in0.size(); b0 += MAXP) {
        const size_t b1 = std::min(in0.size(), b0 + MAXP);
        ConvArgs a{};
        int tiles = 0;
        double flops = 0, bytes = (double)pc.kh * pc.kw * pc.cin * pc.cout * 4.0;
        for (size_t i = b0; i < b1; ++i) {
            ConvProb& p = a.p[i - b0];
            p.in0 = in0[i].p; p.in1 = in1 ? (*in1)[i].p : nullptr; p.res = res ? (*res)[i].p : nullptr;
            p.out = out.empty() ? nullptr : out[i].p;
            p.pool = fuse_pool ? (*pooled)[i].p : nullptr;
            p.H = p.Ho = in0[i].H; p.W = p.Wo = in0[i].W;
            p.tiles_x = cdiv(in0[i].W, CONV_TW);
            p.tile_begin = tiles;
            tiles += p.tiles_x * cdiv(in0[i].H, th);
            flops += 2.0 * in0[i].H * in0[i].W * pc.kh * pc.kw * (double)pc.cin * pc.cout;
            bytes += tbytes(in0[i]) + (in1 ? tbytes((*in1)[i]) : 0.0) + (res ? tbytes((*res)[i]) : 0.0) + (out.empty() ? 0.0 : tbytes(out[i])) +
                     (fuse_pool ? tbytes((*pooled)[i]) : 0.0);
        }
        a.nprob = (int)(b1 - b0);
        a.total_tiles = tiles;
        a.c0 = in0[0].C; a.c1 = in1 ? (*in1)[0].C : 0;
        a.wpk = (const f32x4*)pc.d_w; a.bias = pc.d_b;
        a.cout = pc.cout; a.mtiles = pc.mtiles; a.groups = pc.groups;
        a.relu_in = relu_in; a.relu_out = relu_out; a.act = act;
        a.skip_full = fuse_pool && !keep_full;
        TL sub(in0.begin() + b0, in0.begin() + b1);
        if (wino) {
            a.wpk = (const f32x4*)pc.d_wino;
            int wt = 0;
            for (size_t i = b0; i < b1; ++i) {              // Winograd blocks are 4 x 32 output pixels
                ConvProb& p = a.p[i - b0];
                p.tiles_x = cdiv(in0[i].W, WINO_TW);
                p.tile_begin = wt;
                wt += p.tiles_x * cdiv(in0[i].H, WINO_TH);
            }
            const int mt = wino_mt;
            if (mt == 1) {                                   // register-resident variant for one m-tile: 8 x 32 pixel blocks
                wt = 0;
                for (size_t i = b0; i < b1; ++i) {
                    ConvProb& p = a.p[i - b0];
                    p.tile_begin = wt;
                    wt += p.tiles_x * cdiv(in0[i].H, 2 * WINO_TH);
                }
            }
            a.total_tiles = wt;
            const int ny = pc.mtiles / mt;
            int wunits = wt;
            a.xm = oneshot_map(m, wt, &wunits);
            dim3 grid(wunits, ny);
            std::string pname;
            if (mt == 1) pname = "conv_winor_kernel<false,1,true>";
            else if (mt == 2) pname = !res ? "conv_winor_kernel<false,2,false>" : "conv_winor_kernel<false,2,true>";
            else pname = "conv_wino_kernel" + targs({ti(mt), tb(false)});
            ProfScope ps(m, pname, flops, scope + " " + dims_of(sub) + " " + std::to_string(pc.cin) + "->" + std::to_string(pc.cout));
            ps.bytes = bytes;
            if (mt == 1) {
                hipLaunchKernelGGL((conv_winor_kernel<false, 1, true>), grid, dim3(256), 0, m->stream, a);
            } else if (mt == 2) {
                // register-resident variant: a wave per (tile row, m-tile); grid.y counts pairs of m-tiles
                if (!res) hipLaunchKernelGGL((conv_winor_kernel<false, 2, false>), grid, dim3(256), 0, m->stream, a);
                else hipLaunchKernelGGL((conv_winor_kernel<false, 2, true>), grid, dim3(256), 0, m->stream, a);
            } else hipLaunchKernelGGL((conv_wino_kernel<4>), grid, dim3(256), 0, m->stream, a);
        } else {
            int units = tiles;
            a.xm = oneshot_map(m, tiles, &units);
            if (pc.kh == 3) launch_conv_k<3, 3>(m, pc, a, units, flops, bytes, scope, sub, big_tile);
            else launch_conv_k<4, 4>(m, pc, a, units, flops, bytes, scope, sub, big_tile);
        }
    }
    if (pooled && !fuse_pool) *pooled = run_pool(m, out, POOL_MAX);
    return out;
}

// conv2d_transpose 3x3 stride 2 SAME to the spatial sizes of `like` (ARU_v1.py:255-259)
TL run_deconv(asep_aru* m, const std::string& scope, const TL& in, const TL& like, bool relu_out, int act = 0) {
    auto it = m->convs.find(scope);
    if (it == m->convs.end()) { set_error("internal: deconv %s not packed", scope.c_str()); throw ArgError(); }
    const PackedConv& pc = it->second;
    if (pc.kh != 3 || pc.kw != 3 || in[0].C != pc.cin) {
        set_error("deconv %s: unsupported shape (k=%d, Cin %d vs %d)", scope.c_str(), pc.kh, pc.cin, in[0].C);
        throw ArgError();
    }
    TL out;
    for (size_t i = 0; i < in.size(); ++i) {
        if (cdiv(like[i].H, 2) != in[i].H || cdiv(like[i].W, 2) != in[i].W) {
            set_error("deconv %s: output %dx%d incompatible with input %dx%d", scope.c_str(), like[i].H, like[i].W, in[i].H, in[i].W);
            throw ArgError();
        }
        out.push_back(new_tensor(m, like[i].H, like[i].W, pc.cout));
    }
    const int mt = pc.mtiles % 2 == 0 ? 2 : 1;
    const bool valu = pc.d_wv && m->r8_valu;                 // level 0: one input position per thread on the vector ALU
#ifndef DS_ROWS
#define DS_ROWS 8                  // input rows per block of deconvs_kernel (16: 280 / 310 / 477 us against 260 / 312 / 440)
#endif
    const bool splitd = m->split && m->use_deconvs && pc.d_ws && pc.smode == 2 && !valu;   // >= 32 input channels: split products (deconvs_kernel)
    for (size_t b0 = 0; b0 < in.size(); b0 += MAXP) {
        const size_t b1 = std::min(in.size(), b0 + MAXP);
        ConvArgs a{};
        int tiles = 0;
        double flops = 0, bytes = 9.0 * pc.cin * pc.cout * 4.0;
        for (size_t i = b0; i < b1; ++i) {
            ConvProb& p = a.p[i - b0];
            p.in0 = in[i].p; p.in1 = nullptr; p.res = nullptr; p.out = out[i].p;
            p.H = in[i].H; p.W = in[i].W; p.Ho = out[i].H; p.Wo = out[i].W;
            p.pbh = std::max((in[i].H - 1) * 2 + 3 - out[i].H, 0) / 2;
            p.pbw = std::max((in[i].W - 1) * 2 + 3 - out[i].W, 0) / 2;
            p.tiles_x = cdiv(in[i].W, valu ? DCV_T : DC_TW);
            p.tile_begin = tiles;
            tiles += p.tiles_x * cdiv(in[i].H, valu ? DCV_T : (splitd ? DS_ROWS : DC_TH));
            flops += 2.0 * in[i].H * in[i].W * 9.0 * pc.cin * pc.cout;
            bytes += tbytes(in[i]) + tbytes(out[i]);
        }
        a.nprob = (int)(b1 - b0);
        a.c0 = in[0].C; a.c1 = 0;
        a.wpk = (const f32x4*)pc.d_w; a.bias = pc.d_b;
        a.cout = pc.cout; a.mtiles = pc.mtiles; a.groups = pc.groups;
        a.relu_in = 0; a.relu_out = relu_out; a.act = act;
        int units = tiles;
        a.xm = oneshot_map(m, tiles, &units);
        dim3 grid(units, pc.mtiles / mt);
        const std::string dname = valu ? std::string("deconv8v_kernel") : (splitd ? "deconvs_kernel" + targs({ti(mt), ti(DS_ROWS)}) : "deconv_mfma_kernel" + targs({ti(mt), tb(false)}));
        TL sub(in.begin() + b0, in.begin() + b1);
        ProfScope ps(m, dname, flops, scope + " " + dims_of(sub) + " " + std::to_string(pc.cin) + "->" + std::to_string(pc.cout));
        ps.bytes = bytes;
        if (valu) {
            a.wpk = (const f32x4*)pc.d_wv;
            hipLaunchKernelGGL(deconv8v_kernel, dim3(units), dim3(256), 0, m->stream, a);
        } else if (splitd) {
            a.wpk = (const f32x4*)pc.d_ws; a.groups = pc.cin / 32;
            if (mt == 2) hipLaunchKernelGGL((deconvs_kernel<2, DS_ROWS>), grid, dim3(256), 0, m->stream, a);
            else hipLaunchKernelGGL((deconvs_kernel<1, DS_ROWS>), grid, dim3(256), 0, m->stream, a);
        } else if (mt == 2) hipLaunchKernelGGL((deconv_mfma_kernel<2>), grid, dim3(256), 0, m->stream, a);
        else hipLaunchKernelGGL((deconv_mfma_kernel<1>), grid, dim3(256), 0, m->stream, a);
    }
    return out;
}

// first layer (Cin == 1); stats[i] = per-problem {mean, 1/std} pointer or nullptr
TL run_direct(asep_aru* m, const DirectConv& dc, const TL& imgs, bool relu, const std::vector<const float*>& stats, int act = 0) {
    TL out;
    for (const Tensor& t : imgs) out.push_back(new_tensor(m, t.H, t.W, dc.cout));
    for (size_t b0 = 0; b0 < imgs.size(); b0 += MAXP) {
        const size_t b1 = std::min(imgs.size(), b0 + MAXP);
        C1Args a{};
        int tiles = 0;
        double flops = 0, bytes = 0;
        for (size_t i = b0; i < b1; ++i) {
            C1Prob& p = a.p[i - b0];
            p.img = imgs[i].p; p.out = out[i].p; p.stats = stats.empty() ? nullptr : stats[i];
            p.H = imgs[i].H; p.W = imgs[i].W;
            p.tiles_x = cdiv(imgs[i].W, 64);
            p.tile_begin = tiles;
            tiles += p.tiles_x * cdiv(imgs[i].H, 4);
            flops += 2.0 * imgs[i].H * imgs[i].W * dc.k * dc.k * dc.cout;
            bytes += tbytes(imgs[i]) + tbytes(out[i]);
        }
        a.nprob = (int)(b1 - b0);
        a.w = dc.d_w; a.bias = dc.d_b; a.relu = relu ? 1 : 0; a.act = act;
        ProfScope ps(m, "conv_c1_kernel<" + std::to_string(dc.k) + "," + std::to_string(dc.cout) + ">", flops);
        ps.bytes = bytes;
        dim3 grid(tiles);
        if (dc.k == 3 && dc.cout == 8) hipLaunchKernelGGL((conv_c1_kernel<3, 8>), grid, dim3(256), 0, m->stream, a);
        else if (dc.k == 3 && dc.cout == 16) hipLaunchKernelGGL((conv_c1_kernel<3, 16>), grid, dim3(256), 0, m->stream, a);
        else if (dc.k == 4 && dc.cout == 12) hipLaunchKernelGGL((conv_c1_kernel<4, 12>), grid, dim3(256), 0, m->stream, a);
        else { set_error("first-layer conv k=%d cout=%d not instantiated", dc.k, dc.cout); throw ArgError(); }
    }
    return out;
}

TL run_pool(asep_aru* m, const TL& in, PoolKind kind) {
    TL out;
    for (const Tensor& t : in) {
        if (kind == POOL_CHANSUM) out.push_back(new_tensor(m, t.H, t.W, 1));
        else out.push_back(new_tensor(m, cdiv(t.H, 2), cdiv(t.W, 2), t.C));
    }
    for (size_t b0 = 0; b0 < in.size(); b0 += MAXP) {
        const size_t b1 = std::min(in.size(), b0 + MAXP);
        PoolArgs a{};
        int blocks = 0;
        double bytes = 0;
        for (size_t i = b0; i < b1; ++i) {
            bytes += tbytes(in[i]) + tbytes(out[i]);
            PoolProb& p = a.p[i - b0];
            p.in = in[i].p; p.out = out[i].p; p.H = in[i].H; p.W = in[i].W; p.Ho = out[i].H; p.Wo = out[i].W;
            p.blk_begin = blocks;
            const size_t items = kind == POOL_MAX ? out[i].count() / 4 : (size_t)out[i].H * out[i].W;
            blocks += (int)((items + POOL_ITEMS - 1) / POOL_ITEMS);
        }
        a.nprob = (int)(b1 - b0);
        a.C = in[0].C;
        ProfScope ps(m, kind == POOL_MAX ? "maxpool2_kernel" : (kind == POOL_AVG_C1 ? "avgpool2_c1_kernel" : "chansum_kernel"), 0.0);
        ps.bytes = bytes;
        if (kind == POOL_MAX) hipLaunchKernelGGL(maxpool2_kernel, dim3(blocks), dim3(256), 0, m->stream, a);
        else if (kind == POOL_AVG_C1) hipLaunchKernelGGL(avgpool2_c1_kernel, dim3(blocks), dim3(256), 0, m->stream, a);
        else hipLaunchKernelGGL(chansum_kernel, dim3(blocks), dim3(256), 0, m->stream, a);
    }
    return out;
}

int grid_1d(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 256 * 8); }

// pixel-pair A fragments of a 3x3 conv with 8 input channels starting at input channel ci0 of W[3][3][cin][8]:
// rows = (pixel parity e, cout), chunk = (ky, h), slot s: kx' = 2h + (s>>3), ci = s&7, kx = kx' - e
void pack_pair8(const HostTensor& w, int cin, int ci0, std::vector<float>& dst) {
    for (int ky = 0; ky < 3; ++ky)
        for (int h = 0; h < 2; ++h)
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r) {
                    const int row = lane & 15, kk = lane >> 4, s = 4 * kk + r;
                    const int e = row >> 3, co = row & 7;
                    const int kx = 2 * h + (s >> 3) - e, ci = s & 7;
                    float v = 0.f;
                    if (kx >= 0 && kx <= 2) v = w.data[(((size_t)ky * 3 + kx) * cin + ci0 + ci) * 8 + co];
                    dst.push_back(v);
                }
}

// scalar layout of a 3x3 conv with 8 input channels starting at input channel ci0 of W[3][3][cin][8] (res8v_kernels.h).
// Direct: [g = (ky*2 + hf)*3 + kx][c][co], input channel ci0 + hf*4 + c.  Winograd F(2,3) along x: [(ky*2 + hf)*4 + j][c][co],
// U_j = sum_kx G[j][kx] g[ky][kx], G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] (double accumulation)
void pack_scalar8(const HostTensor& w, int cin, int ci0, std::vector<float>& dst) {
    auto W = [&](int ky, int kx, int ci, int co) { return (double)w.data[(((size_t)ky * 3 + kx) * cin + ci0 + ci) * 8 + co]; };
    if (R8V_WINO) {
        static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
        for (int rh = 0; rh < 6; ++rh)
            for (int j = 0; j < 4; ++j)
                for (int c = 0; c < 4; ++c)
                    for (int co = 0; co < 8; ++co) {
                        double u = 0;
                        for (int kx = 0; kx < 3; ++kx) u += G[j][kx] * W(rh >> 1, kx, (rh & 1) * 4 + c, co);
                        dst.push_back((float)u);
                    }
        return;
    }
    for (int g = 0; g < 18; ++g) {
        const int ky = g / 6, hf = (g / 3) % 2, kx = g % 3;
        for (int c = 0; c < 4; ++c)
            for (int co = 0; co < 8; ++co) dst.push_back((float)W(ky, kx, hf * 4 + c, co));
    }
}

int pack_res8(asep_aru* m, const std::map<std::string, HostTensor>& blob) {
    const std::string s = "aru_net/featMapG/unet_down_0";
    std::vector<float> wr, br, vwr;
    for (int r = 0; r < 3; ++r) {
        auto wi = blob.find(s + "/convR_" + std::to_string(r) + "/weights");
        auto bi = blob.find(s + "/convR_" + std::to_string(r) + "/biases");
        if (wi == blob.end() || bi == blob.end()) { set_error("weights: missing %s/convR_%d", s.c_str(), r); return ASEP_ERR_WEIGHTS; }
        pack_pair8(wi->second, 8, 0, wr);
        pack_scalar8(wi->second, 8, 0, vwr);
        br.insert(br.end(), bi->second.data.begin(), bi->second.data.end());
    }
    int rc = upload(wr, &m->d_r8_down_wr);
    if (!rc) rc = upload(br, &m->d_r8_down_br);
    if (!rc) rc = upload(vwr, &m->d_r8v_down_wr);
    if (rc) return rc;
    m->owned.push_back(m->d_r8_down_wr);
    m->owned.push_back(m->d_r8_down_br);
    m->owned.push_back(m->d_r8v_down_wr);
    if (m->cfg.scale_space_num > 1) {
        const std::string u = "aru_net/featMapG/unet_up_0";
        auto w1 = blob.find(u + "/conv1/weights");
        auto b1 = blob.find(u + "/conv1/biases");
        if (w1 == blob.end() || b1 == blob.end()) { set_error("weights: missing %s/conv1", u.c_str()); return ASEP_ERR_WEIGHTS; }
        std::vector<float> pw1, pwr, pbr, vw1, vwr2;
        pack_pair8(w1->second, 16, 0, pw1);      // skip channels 0..7
        pack_pair8(w1->second, 16, 8, pw1);      // deconv channels 8..15
        pack_scalar8(w1->second, 16, 0, vw1);
        pack_scalar8(w1->second, 16, 8, vw1);
        for (int r = 0; r < 3; ++r) {
            auto wi = blob.find(u + "/convR_" + std::to_string(r) + "/weights");
            auto bi = blob.find(u + "/convR_" + std::to_string(r) + "/biases");
            if (wi == blob.end() || bi == blob.end()) { set_error("weights: missing %s/convR_%d", u.c_str(), r); return ASEP_ERR_WEIGHTS; }
            pack_pair8(wi->second, 8, 0, pwr);
            pack_scalar8(wi->second, 8, 0, vwr2);
            pbr.insert(pbr.end(), bi->second.data.begin(), bi->second.data.end());
        }
        rc = upload(pw1, &m->d_r8_up_w1);
        if (!rc) rc = upload(pwr, &m->d_r8_up_wr);
        if (!rc) rc = upload(pbr, &m->d_r8_up_br);
        if (!rc) rc = upload(b1->second.data, &m->d_r8_up_b1);
        if (!rc) rc = upload(vw1, &m->d_r8v_up_w1);
        if (!rc) rc = upload(vwr2, &m->d_r8v_up_wr);
        if (rc) return rc;
        m->owned.push_back(m->d_r8v_up_w1); m->owned.push_back(m->d_r8v_up_wr);
        m->owned.push_back(m->d_r8_up_w1); m->owned.push_back(m->d_r8_up_wr);
        m->owned.push_back(m->d_r8_up_br); m->owned.push_back(m->d_r8_up_b1);
        if (hipFuncSetAttribute((const void*)res8_up_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)R8_UP_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)res8v_up_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)R8_UP_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)res8v_up_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)R8_UP_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)res8v_up_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)R8_UP_LDS) != hipSuccess) {
            set_error("cannot reserve %zu bytes of LDS for the fused up block", R8_UP_LDS);
            return ASEP_ERR_HIP;
        }
    }
    if (hipFuncSetAttribute((const void*)res8_down_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)R8_DOWN_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)res8v_down_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)R8_DOWN_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)res8v_down_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)R8_DOWN_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)res8v_down_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)R8_DOWN_LDS) != hipSuccess) {
        set_error("cannot reserve %zu bytes of LDS for the fused residual block", R8_DOWN_LDS);
        return ASEP_ERR_HIP;
    }
    return ASEP_OK;
}

// fused level-0 down block: images -> d0 (and maxpool2(d0) if want_pool)
// XCD-aware order of the persistent fused kernels' tiles.  Workgroups are dealt round-robin to the 8 XCDs (block b
// runs on XCD b % 8, each with its own 4 MB L2).  The tiles of every problem are first put into "super-tile" order
// (groups of 4 x 8 tiles, groups walked down a column of groups first), the concatenated list is cut into 8 equal
// chunks, and the k-th unit of work (k = block + i * grid) takes the (k / 8)-th tile of chunk k % 8: the 32 blocks of
// an XCD work on spatially adjacent tiles at the same time, and on the rows just below right after, so the 8-row /
// 14-column halo overlap of neighbouring tiles is served by that XCD's L2 instead of being fetched again.
// the vector-ALU level-0 kernels address their tensors with 32-bit element offsets (< 2^28 pixels per tensor)
bool r8v_fits(const TL& l) {
    for (const Tensor& t : l)
        if ((size_t)t.H * t.W >= ((size_t)1 << 28)) return false;
    return true;
}

void run_res8_down(asep_aru* m, const TL& imgs, const std::vector<const float*>& stats, bool want_pool, TL* d_out, TL* pool_out) {
    for (const Tensor& t : imgs) {
        d_out->push_back(new_tensor(m, t.H, t.W, 8));
        if (want_pool) pool_out->push_back(new_tensor(m, cdiv(t.H, 2), cdiv(t.W, 2), 8));
    }
    for (size_t b0 = 0; b0 < imgs.size(); b0 += MAXP) {
        const size_t b1 = std::min(imgs.size(), b0 + MAXP);
        Res8Args a{};
        int tiles = 0;
        double flops = 0, bytes = 0;
        for (size_t i = b0; i < b1; ++i) {
            bytes += tbytes(imgs[i]) + tbytes((*d_out)[i]) + (want_pool ? tbytes((*pool_out)[i]) : 0.0);
            Res8Prob& p = a.p[i - b0];
            p.img = imgs[i].p; p.in1 = nullptr; p.stats = stats.empty() ? nullptr : stats[i];
            p.out = (*d_out)[i].p; p.pool = want_pool ? (*pool_out)[i].p : nullptr;
            p.H = imgs[i].H; p.W = imgs[i].W;
            p.tiles_x = cdiv(imgs[i].W, R8_OW);
            p.tile_begin = tiles;
            tiles += p.tiles_x * cdiv(imgs[i].H, R8_OH * R8_NP);
            flops += 2.0 * imgs[i].H * imgs[i].W * (9.0 * 8 + 3 * 9.0 * 64);
        }
        a.nprob = (int)(b1 - b0);
        a.total_tiles = tiles;
        a.w1 = m->det_first.d_w; a.b1 = m->det_first.d_b;
        bool valu = m->r8_valu;                  // vector-ALU kernels (32-bit element offsets: < 2^29 pixels per tensor)
        for (size_t i = b0; i < b1; ++i) valu = valu && (size_t)imgs[i].H * imgs[i].W < ((size_t)1 << 28);
        a.wr = (const f32x4*)(valu ? m->d_r8v_down_wr : m->d_r8_down_wr); a.br = m->d_r8_down_br;
        TL sub(imgs.begin() + b0, imgs.begin() + b1);
        const std::string pname = valu ? std::string("res8v_down_kernel<0>") : std::string("res8_down_kernel<false>");   // (rocprofv3's names: template arguments spelled out)
        ProfScope ps(m, pname, flops, "unet_down_0 (conv1+3xconvR+add+pool) " + dims_of(sub));
        ps.bytes = bytes;
        a.sched = tile_schedule(m, a, std::min(tiles, m->num_cus), R8_OH * R8_NP);
        const dim3 gd(std::min(tiles, m->num_cus));
        const int actv = m->cfg.activation;                  // graph variants (fused8_var): the same block with elu / leaky (vector-ALU form only)
        if (actv && !valu) { set_error("level-0 block of an elu / leaky graph: image too large for the vector-ALU kernel"); throw ArgError(); }
        if (valu && actv == 1) { ps.set_name("res8v_down_kernel<1>"); hipLaunchKernelGGL(res8v_down_kernel<1>, gd, dim3(R8_THREADS), R8_DOWN_LDS, m->stream, a); }
        else if (valu && actv == 2) { ps.set_name("res8v_down_kernel<2>"); hipLaunchKernelGGL(res8v_down_kernel<2>, gd, dim3(R8_THREADS), R8_DOWN_LDS, m->stream, a); }
        else if (valu) hipLaunchKernelGGL(res8v_down_kernel<0>, gd, dim3(R8_THREADS), R8_DOWN_LDS, m->stream, a);
        else hipLaunchKernelGGL(res8_down_kernel<false>, gd, dim3(R8_THREADS), R8_DOWN_LDS, m->stream, a);
    }
}

// fused level-0 up block: [skip, deconv] -> block output
TL run_res8_up(asep_aru* m, const TL& skip, const TL& v) {
    TL out;
    for (const Tensor& t : skip) out.push_back(new_tensor(m, t.H, t.W, 8));
    for (size_t b0 = 0; b0 < skip.size(); b0 += MAXP) {
        const size_t b1 = std::min(skip.size(), b0 + MAXP);
        Res8Args a{};
        int tiles = 0;
        double flops = 0, bytes = 0;
        for (size_t i = b0; i < b1; ++i) {
            bytes += tbytes(skip[i]) + tbytes(v[i]) + tbytes(out[i]);
            Res8Prob& p = a.p[i - b0];
            p.img = skip[i].p; p.in1 = v[i].p; p.stats = nullptr; p.out = out[i].p; p.pool = nullptr;
            p.H = skip[i].H; p.W = skip[i].W;
            p.tiles_x = cdiv(skip[i].W, R8_OW);
            p.tile_begin = tiles;
            tiles += p.tiles_x * cdiv(skip[i].H, R8_OH * R8_NP);
            flops += 2.0 * skip[i].H * skip[i].W * (9.0 * 16 * 8 + 3 * 9.0 * 64);
        }
        a.nprob = (int)(b1 - b0);
        a.total_tiles = tiles;
        bool valu = m->r8_valu;                  // vector-ALU kernels (32-bit element offsets: < 2^29 pixels per tensor)
        for (size_t i = b0; i < b1; ++i) valu = valu && (size_t)skip[i].H * skip[i].W < ((size_t)1 << 28);
        a.w1 = valu ? m->d_r8v_up_w1 : m->d_r8_up_w1; a.b1 = m->d_r8_up_b1;
        a.wr = (const f32x4*)(valu ? m->d_r8v_up_wr : m->d_r8_up_wr); a.br = m->d_r8_up_br;
        TL sub(skip.begin() + b0, skip.begin() + b1);
        const std::string pname = valu ? std::string("res8v_up_kernel<0>") : std::string("res8_up_kernel<false>");
        ProfScope ps(m, pname, flops, "unet_up_0 (conv1[16->8]+3xconvR+add) " + dims_of(sub));
        ps.bytes = bytes;
        a.sched = tile_schedule(m, a, std::min(tiles, m->num_cus), R8_OH * R8_NP);
        const dim3 grid(std::min(tiles, m->num_cus));
        const int actv = m->cfg.activation;
        if (actv && !valu) { set_error("level-0 block of an elu / leaky graph: image too large for the vector-ALU kernel"); throw ArgError(); }
        if (valu && actv == 1) { ps.set_name("res8v_up_kernel<1>"); hipLaunchKernelGGL(res8v_up_kernel<1>, grid, dim3(R8_THREADS), R8_UP_LDS, m->stream, a); }
        else if (valu && actv == 2) { ps.set_name("res8v_up_kernel<2>"); hipLaunchKernelGGL(res8v_up_kernel<2>, grid, dim3(R8_THREADS), R8_UP_LDS, m->stream, a); }
        else if (valu) hipLaunchKernelGGL(res8v_up_kernel<0>, grid, dim3(R8_THREADS), R8_UP_LDS, m->stream, a);
        else hipLaunchKernelGGL(res8_up_kernel<false>, grid, dim3(R8_THREADS), R8_UP_LDS, m->stream, a);
    }
    return out;
}

// ================================================================================================
// Native bf16 data path (cfg.compute_dtype == 1): packing and launchers of bf16_kernels.h
// ================================================================================================
bf16_t f2bf(float f) {                                 // round-to-nearest-even like v_cvt_pk_bf16_f32 (weights have no NaN)
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

int upload_bf(const std::vector<bf16_t>& h, bf16_t** d) {
    ASEP_HIP_CHECK(hipMalloc((void**)d, std::max<size_t>(h.size(), 8) * sizeof(bf16_t)));
    ASEP_HIP_CHECK(hipMemcpy(*d, h.data(), h.size() * sizeof(bf16_t), hipMemcpyHostToDevice));
    return ASEP_OK;
}

// A fragments of a conv for convb_kernel / resb_tail_kernel.  W(tap, ci, co) = weight or 0 outside the filter.
// mode 0: chunk = ky, k = 8 kx + ci (kx = 3: zero);  mode 1: chunk c, k = 16 (tap - 2c) + ci;  mode 2: chunk = G taps + tap, k = ci - 32 G
template <class WF>
void pack_frags_conv(int mode, int kh, int kw, int cin, int mtiles, WF W, std::vector<bf16_t>& dst, int* nchunks) {
    const int taps = kh * kw;
    const int chunks = mode == 0 ? kh : (mode == 1 ? (taps + 1) / 2 : (cin / 32) * taps);
    *nchunks = chunks;
    for (int ch = 0; ch < chunks; ++ch)
        for (int mt = 0; mt < mtiles; ++mt)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int kk = lane >> 4, co = mt * 16 + (lane & 15);
                    int tap, ci;
                    if (mode == 0) { tap = kk < kw ? ch * kw + kk : -1; ci = j; }
                    else if (mode == 1) { tap = 2 * ch + (kk >> 1); ci = (kk & 1) * 8 + j; if (tap >= taps) tap = -1; }
                    else { const int G = ch / taps; tap = ch % taps; ci = 32 * G + kk * 8 + j; }
                    dst.push_back(f2bf(tap < 0 ? 0.f : W(tap, ci, co)));
                }
}

int conv_bmode(int cin) { return cin == 8 ? 0 : (cin == 16 ? 1 : (cin % 32 == 0 ? 2 : -1)); }

// bf16 fragments of one packed conv / deconv (called from pack_conv when the model is bf16)
int pack_conv_bf(asep_aru* m, PackedConv& pc, const HostTensor& w) {
    const int taps = pc.kh * pc.kw;
    auto W = [&](int tap, int ci, int co) -> float {
        if (ci >= pc.cin || co >= pc.cout || tap >= taps) return 0.f;
        return pc.deconv ? w.data[((size_t)tap * pc.cout + co) * pc.cin + ci] : w.data[((size_t)tap * pc.cin + ci) * pc.cout + co];
    };
    std::vector<bf16_t> pk;
    if (!pc.deconv) {
        // the 12-channel output of the attention head is stored as a 16-channel plane (4 zero channels): Cin 12 -> mode 1
        const int cin_eff = pc.cin == 12 ? 16 : pc.cin;
        pc.bmode = conv_bmode(cin_eff);
        if (pc.bmode < 0 || (pc.bmode == 0 && pc.kh != 3)) return ASEP_OK;      // not served by convb (refused at run time if it is needed)
        pack_frags_conv(pc.bmode, pc.kh, pc.kw, cin_eff, pc.mtiles, W, pk, &pc.bchunks);
    } else {
        pc.bmode = pc.cin == 16 ? 1 : (pc.cin % 32 == 0 ? 2 : -1);
        if (pc.bmode < 0 || pc.kh != 3 || pc.kw != 3) { pc.bmode = -1; return ASEP_OK; }
        if (pc.bmode == 2) {
            const int G = pc.cin / 32;
            pc.bchunks = G * 9;
            for (int g = 0; g < G; ++g)
                for (int tap = 0; tap < 9; ++tap)
                    for (int mt = 0; mt < pc.mtiles; ++mt)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j)
                                pk.push_back(f2bf(W(tap, 32 * g + (lane >> 4) * 8 + j, mt * 16 + (lane & 15))));
        } else {
            // fragment f: dy = f >= 4; class (py, px) = dy ? (0, f & 1) : (f >> 1, f & 1); k = 16 dx + ci
            pc.bchunks = 6;
            for (int f = 0; f < 6; ++f)
                for (int mt = 0; mt < pc.mtiles; ++mt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int kk = lane >> 4, dx = kk >> 1, ci = (kk & 1) * 8 + j, co = mt * 16 + (lane & 15);
                            const int dy = f >= 4, py = dy ? 0 : (f >> 1), px = f & 1;
                            const int ky = py ? 1 : (dy ? 2 : 0);
                            const int kx = px ? (dx ? -1 : 1) : (dx ? 2 : 0);
                            pk.push_back(f2bf(kx < 0 ? 0.f : W(ky * 3 + kx, ci, co)));
                        }
        }
    }
    int rc = upload_bf(pk, &pc.d_wb);
    if (rc) return rc;
    m->owned.push_back(pc.d_wb);
    if (pc.deconv && pc.bmode == 1 && pc.cout == 8) {
        // deconvb8_kernel: fragment q = (dy = 0, py = 0), (dy = 0, py = 1), (dy = 1, py = 0); row m = 8 px + co; k = 16 dx + ci
        std::vector<bf16_t> p8;
        for (int q = 0; q < 3; ++q)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int kk = lane >> 4, dx = kk >> 1, ci = (kk & 1) * 8 + j, mrow = lane & 15, px = mrow >> 3, co = mrow & 7;
                    const int dy = q == 2, py = q == 1;
                    const int ky = py ? 1 : (dy ? 2 : 0);
                    const int kx = px ? (dx ? -1 : 1) : (dx ? 2 : 0);
                    p8.push_back(f2bf(kx < 0 ? 0.f : W(ky * 3 + kx, ci, co)));
                }
        rc = upload_bf(p8, &pc.d_wb8);
        if (rc) return rc;
        m->owned.push_back(pc.d_wb8);
    }
    return ASEP_OK;
}

// fp32 filter -> its three bfloat16 parts (round to nearest at every cut: w = h + m + l exactly), in convs_kernel's fragment order
int pack_conv_split(asep_aru* m, PackedConv& pc, const HostTensor& w) {
    const int taps = pc.kh * pc.kw;
    if (!((pc.kh == 3 && pc.kw == 3) || (pc.kh == 4 && pc.kw == 4)) || pc.cout % 16 != 0) return ASEP_OK;
    pc.smode = (pc.cin == 16 || pc.cin == 12) ? 1 : (pc.cin % 32 == 0 ? 2 : -1);
    if (pc.smode == 2 && pc.kh != 3) pc.smode = -1;          // 4x4 filters are instantiated for the 12- / 16-channel form only: such a layer keeps the fp32 MFMA kernel
    if (pc.smode < 0) return ASEP_OK;
    auto W = [&](int tap, int ci, int co) -> float {
        if (ci >= pc.cin || co >= pc.cout || tap >= taps) return 0.f;
        return w.data[((size_t)tap * pc.cin + ci) * pc.cout + co];
    };
    auto bfval = [](bf16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; };
    const int chunks = pc.smode == 1 ? (taps + 1) / 2 : (pc.cin / 32) * taps;
    std::vector<bf16_t> pk((size_t)chunks * 3 * pc.mtiles * 64 * 8);
    for (int ch = 0; ch < chunks; ++ch)
        for (int mt = 0; mt < pc.mtiles; ++mt)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int kk = lane >> 4, co = mt * 16 + (lane & 15);
                    int tap, ci;
                    if (pc.smode == 1) { tap = 2 * ch + (kk >> 1); ci = (kk & 1) * 8 + j; if (tap >= taps) tap = -1; }
                    else { const int G = ch / taps; tap = ch % taps; ci = 32 * G + kk * 8 + j; }
                    const float v = tap < 0 ? 0.f : W(tap, ci, co);
                    const bf16_t h = f2bf(v);
                    const float r = v - bfval(h);
                    const bf16_t mm = f2bf(r);
                    const bf16_t l = f2bf(r - bfval(mm));
                    const bf16_t part[3] = {h, mm, l};
                    for (int s = 0; s < 3; ++s) pk[((((size_t)ch * 3 + s) * pc.mtiles + mt) * 64 + lane) * 8 + j] = part[s];
                }
    int rc = upload_bf(pk, &pc.d_ws);
    if (rc) return rc;
    m->owned.push_back(pc.d_ws);
    if (pc.kh == 3 && pc.kw == 3 && pc.cin % 16 == 0 && pc.cin >= 32) {
        const int stages = pc.cin / 16;
        std::vector<bf16_t> pk16((size_t)stages * 5 * 3 * pc.mtiles * 64 * 8);
        for (int g = 0; g < stages; ++g)
            for (int t = 0; t < 5; ++t)
                for (int mt = 0; mt < pc.mtiles; ++mt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int kk = lane >> 4, co = mt * 16 + (lane & 15);
                            const int tap = 2 * t + (kk >> 1), ci = 16 * g + (kk & 1) * 8 + j;
                            const float v = tap < taps ? W(tap, ci, co) : 0.f;
                            const bf16_t h = f2bf(v);
                            const float r = v - bfval(h);
                            const bf16_t mm = f2bf(r);
                            const bf16_t part[3] = {h, mm, f2bf(r - bfval(mm))};
                            for (int s = 0; s < 3; ++s)
                                pk16[(((((size_t)g * 5 + t) * 3 + s) * pc.mtiles + mt) * 64 + lane) * 8 + j] = part[s];
                        }
        rc = upload_bf(pk16, &pc.d_ws16);
        if (rc) return rc;
        m->owned.push_back(pc.d_ws16);
    }
    return ASEP_OK;
}

// a 3x3 deconvolution filter with Cin % 32 == 0 as three bfloat16 parts for deconvs_kernel: [stage of 32 channels][tap][part h, m, l][m-tile][lane][8]
int pack_deconv_split(asep_aru* m, PackedConv& pc, const HostTensor& w) {
    if (pc.kh != 3 || pc.kw != 3 || pc.cin % 32 != 0 || pc.cout % 16 != 0) return ASEP_OK;      // (level 0, 16 -> 8: deconv8v_kernel)
    auto W = [&](int tap, int ci, int co) -> float { return w.data[((size_t)tap * pc.cout + co) * pc.cin + ci]; };
    auto bfval = [](bf16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; };
    const int G = pc.cin / 32;
    std::vector<bf16_t> pk((size_t)G * 9 * 3 * pc.mtiles * 64 * 8);
    for (int g = 0; g < G; ++g)
        for (int tap = 0; tap < 9; ++tap)
            for (int mt = 0; mt < pc.mtiles; ++mt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const float v = W(tap, 32 * g + (lane >> 4) * 8 + j, mt * 16 + (lane & 15));
                        const bf16_t h = f2bf(v);
                        const float r = v - bfval(h);
                        const bf16_t mm = f2bf(r);
                        const bf16_t part[3] = {h, mm, f2bf(r - bfval(mm))};
                        for (int s = 0; s < 3; ++s) pk[(((((size_t)g * 9 + tap) * 3 + s) * pc.mtiles + mt) * 64 + lane) * 8 + j] = part[s];
                    }
    int rc = upload_bf(pk, &pc.d_ws);
    if (rc) return rc;
    m->owned.push_back(pc.d_ws);
    pc.smode = 2;
    return ASEP_OK;
}

// the three convR filters of a residual block for resb_tail_kernel<C>: [3][CPC][64][8] + biases [3][C]
int pack_resb(asep_aru* m, const std::map<std::string, HostTensor>& blob, const std::string& scope, int C) {
    std::vector<bf16_t> pk;
    std::vector<float> br;
    for (int r = 0; r < 3; ++r) {
        auto wi = blob.find(scope + "/convR_" + std::to_string(r) + "/weights");
        auto bi = blob.find(scope + "/convR_" + std::to_string(r) + "/biases");
        if (wi == blob.end() || bi == blob.end()) { set_error("weights: missing %s/convR_%d", scope.c_str(), r); return ASEP_ERR_WEIGHTS; }
        const HostTensor& w = wi->second;
        if (w.dims.size() != 4 || w.dims[0] != 3 || w.dims[1] != 3 || w.dims[2] != C || w.dims[3] != C) return ASEP_OK;   // not this shape: layer-by-layer
        auto W = [&](int tap, int ci, int co) -> float { return (ci < C && co < C) ? w.data[((size_t)tap * C + ci) * C + co] : 0.f; };
        int nch = 0;
        if (C == 32) pack_frags_conv(2, 3, 3, 32, 2, W, pk, &nch);          // [9 taps][2 m-tiles][64 lanes] (res32_tail_kernel)
        else pack_frags_conv(C == 8 ? 0 : 1, 3, 3, C, 1, W, pk, &nch);
        br.insert(br.end(), bi->second.data.begin(), bi->second.data.end());
    }
    asep_aru::ResB rb;
    rb.C = C;
    int rc = upload_bf(pk, &rb.d_w);
    if (!rc) rc = upload(br, &rb.d_b);
    if (rc) return rc;
    m->owned.push_back(rb.d_w);
    m->owned.push_back(rb.d_b);
    m->resb[scope] = rb;
    return ASEP_OK;
}

// pixel-pair A fragments of a 3x3 conv with 8 output channels for res8b_kernel: row m = (pixel parity e, cout), one fragment
// per filter row (and per half of the 4-pixel window when there are 16 input channels): k = 8 kk + j.
//   cin 8:  window pixel p = kk, ci = j;   cin 16: window pixel p = 2 half + (kk >> 1), ci = 8 (kk & 1) + j;   kx = p - e
void pack_pair_frags(const HostTensor& w, int cin, std::vector<bf16_t>& dst) {
    const int halves = cin == 16 ? 2 : 1;
    for (int ky = 0; ky < 3; ++ky)
        for (int hf = 0; hf < halves; ++hf)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int m = lane & 15, kk = lane >> 4, e = m >> 3, co = m & 7;
                    const int p = cin == 16 ? 2 * hf + (kk >> 1) : kk;
                    const int ci = cin == 16 ? (kk & 1) * 8 + j : j;
                    const int kx = p - e;
                    dst.push_back(f2bf((kx >= 0 && kx <= 2) ? w.data[(((size_t)ky * 3 + kx) * cin + ci) * 8 + co] : 0.f));
                }
}

int pack_res8b(asep_aru* m, const std::map<std::string, HostTensor>& blob) {
    auto tail = [&](const std::string& scope, bf16_t** d_w, float** d_b) -> int {
        std::vector<bf16_t> pk;
        std::vector<float> br;
        for (int r = 0; r < 3; ++r) {
            auto wi = blob.find(scope + "/convR_" + std::to_string(r) + "/weights");
            auto bi = blob.find(scope + "/convR_" + std::to_string(r) + "/biases");
            if (wi == blob.end() || bi == blob.end()) { set_error("weights: missing %s/convR_%d", scope.c_str(), r); return ASEP_ERR_WEIGHTS; }
            const HostTensor& w = wi->second;
            if (w.dims.size() != 4 || w.dims[0] != 3 || w.dims[1] != 3 || w.dims[2] != 8 || w.dims[3] != 8) return 1;   // other shape: generic kernels
            pack_pair_frags(w, 8, pk);
            br.insert(br.end(), bi->second.data.begin(), bi->second.data.end());
        }
        int rc = upload_bf(pk, d_w);
        if (!rc) rc = upload(br, d_b);
        if (rc) return rc;
        m->owned.push_back(*d_w);
        m->owned.push_back(*d_b);
        return ASEP_OK;
    };
    int rc = tail("aru_net/featMapG/unet_down_0", &m->d_r8b_down_w, &m->d_r8b_down_b);
    if (rc == 1) { m->d_r8b_down_w = nullptr; return ASEP_OK; }
    if (rc) return rc;
    {
        // conv1 (1 -> 8) as one pair fragment: row m = (parity e, cout); k = 8 kk + jj: window row 2 kk + (jj >> 2) (kk < 2), column jj & 3
        auto w1 = blob.find("aru_net/featMapG/unet_down_0/conv1/weights");
        if (w1 != blob.end() && w1->second.dims.size() == 4 && w1->second.dims[0] == 3 && w1->second.dims[1] == 3 && w1->second.dims[2] == 1 &&
            w1->second.dims[3] == 8) {
            std::vector<bf16_t> pk;
            for (int lane = 0; lane < 64; ++lane)
                for (int jj = 0; jj < 8; ++jj) {
                    const int mrow = lane & 15, kk = lane >> 4, e = mrow >> 3, co = mrow & 7;
                    const int ky = 2 * kk + (jj >> 2), kx = (jj & 3) - e;
                    pk.push_back(f2bf((kk < 2 && ky <= 2 && kx >= 0 && kx <= 2) ? w1->second.data[(size_t)(ky * 3 + kx) * 8 + co] : 0.f));
                }
            rc = upload_bf(pk, &m->d_r8f_down_w1);
            if (rc) return rc;
            m->owned.push_back(m->d_r8f_down_w1);
            std::vector<float> wr(w1->second.data.size());
            for (size_t i = 0; i < wr.size(); ++i) {
                const uint32_t u = (uint32_t)f2bf(w1->second.data[i]) << 16;
                memcpy(&wr[i], &u, 4);
            }
            rc = upload(wr, &m->d_r8b_down_w1r);
            if (rc) return rc;
            m->owned.push_back(m->d_r8b_down_w1r);
        }
    }
    if (m->cfg.scale_space_num > 1) {
        const std::string u = "aru_net/featMapG/unet_up_0";
        auto w1 = blob.find(u + "/conv1/weights");
        auto b1 = blob.find(u + "/conv1/biases");
        if (w1 == blob.end() || b1 == blob.end()) { set_error("weights: missing %s/conv1", u.c_str()); return ASEP_ERR_WEIGHTS; }
        const HostTensor& w = w1->second;
        if (w.dims.size() != 4 || w.dims[0] != 3 || w.dims[1] != 3 || w.dims[2] != 16 || w.dims[3] != 8) return ASEP_OK;
        rc = tail(u, &m->d_r8b_up_w, &m->d_r8b_up_b);
        if (rc == 1) { m->d_r8b_up_w = nullptr; return ASEP_OK; }
        if (rc) return rc;
        std::vector<bf16_t> pk;
        pack_pair_frags(w, 16, pk);
        rc = upload_bf(pk, &m->d_r8b_up_w1);
        if (!rc) rc = upload(b1->second.data, &m->d_r8b_up_b1);
        if (rc) return rc;
        m->owned.push_back(m->d_r8b_up_w1);
        m->owned.push_back(m->d_r8b_up_b1);
        // res8f_kernel (interior tiles) keeps skip and deconv as two 16-byte planes: fragment (ky, source), k = 8 kk + j <-> window pixel kk, channel 8 source + j
        std::vector<bf16_t> pf;
        for (int ky = 0; ky < 3; ++ky)
            for (int src = 0; src < 2; ++src)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int mrow = lane & 15, kk = lane >> 4, e = mrow >> 3, co = mrow & 7, kx = kk - e;
                        pf.push_back(f2bf((kx >= 0 && kx <= 2) ? w.data[(((size_t)ky * 3 + kx) * 16 + src * 8 + j) * 8 + co] : 0.f));
                    }
        rc = upload_bf(pf, &m->d_r8f_up_w1);
        if (rc) return rc;
        m->owned.push_back(m->d_r8f_up_w1);
    }
    return ASEP_OK;
}

Tensor new_tensor_bf(asep_aru* m, int H, int W, int C);
void run_res8b_tiles(asep_aru* m, bool up, const TL& a0, const TL* a1, const std::vector<const float*>& stats, bool want_pool, const TL& outs, const TL* pool_out);

// whole level-0 blocks of the bf16 path (res8b_kernel)
// A level-0 block (up: [skip, deconv] in; down: the fp32 image in, pool out) of the pages the strip walker serves (res8w_kernels.h): one launch of
// walker items (one wave each) + one launch of the border tiles around the walkers' regions.  `outs` / `pools` are the pages' output tensors.
void run_res8w(asep_aru* m, bool up, const TL& in0, const TL* dec, const std::vector<const float*>& stats, const TL& outs, const TL* pools) {
    for (size_t b0 = 0; b0 < in0.size(); b0 += MAXP) {
        const size_t b1 = std::min(in0.size(), b0 + MAXP);
        Res8WArgs wa{};
        Res8WBArgs ba{};
        double flops = 0, bytes = 0, wshare = 0;
        long strip_rows = 0;
        for (size_t i = b0; i < b1; ++i) {
            Res8WProb& p = wa.p[i - b0];
            if (up) { p.skip = in0[i].bp(); p.dec = (*dec)[i].bp(); }
            else { p.img = in0[i].p; p.stats = stats.empty() ? nullptr : stats[i]; p.pool = pools ? (*pools)[i].bp() : nullptr; }
            p.out = outs[i].bp();
            p.H = in0[i].H; p.W = in0[i].W;
            p.n_strips = (p.W - 4 - R8W_X0) / R8W_TW;
            p.y_end = R8W_Y0 + 2 * ((p.H - 4 - R8W_Y0) / 2);
            strip_rows += (long)p.n_strips * (p.y_end - R8W_Y0);
            bytes += tbytes(in0[i]) + (up ? tbytes((*dec)[i]) : 0.0) + tbytes(outs[i]) + (pools ? tbytes((*pools)[i]) : 0.0);
            flops += 2.0 * p.H * p.W * (9.0 * (up ? 16 : 1) * 8 + 3 * 9.0 * 64);
            wshare += (double)p.n_strips * R8W_TW * (p.y_end - R8W_Y0);
        }
        // rows of an item: ~6 items per resident wave of the chip (eight waves per CU), so that the hardware's block dispatch balances the tail;
        // an item's head and tail cost about fifteen iterations of the general form (profiles/r6_walk: 64 / 128 / 256 / 400 / 800 rows per item =
        // 952 / 798 / 747 / 752 / 739 us per 4-page launch)
        const long slots = 8L * m->num_cus;
        int band = (int)std::min<long>(256, std::max<long>(32, strip_rows / (6 * slots)));
        band = (band + 1) & ~1;
#ifdef ASEP_ABLATION
        if (const char* e = getenv("ASEP_BF_WALK_BAND")) band = std::max(2, atoi(e) & ~1);      // (measurement knob of ablation builds)
#endif
        int items = 0, btiles = 0;
        for (size_t i = b0; i < b1; ++i) {
            Res8WProb& p = wa.p[i - b0];
            p.band = band;
            p.tile_begin = items;
            items += p.n_strips * cdiv(p.y_end - R8W_Y0, band);
            Res8BProb& q = ba.b.p[i - b0];
            q.skip = p.skip; q.dec = p.dec; q.img = p.img; q.stats = p.stats; q.out = p.out; q.pool = p.pool; q.H = p.H; q.W = p.W;
            q.tile_begin = btiles;
            ba.nbx[i - b0] = cdiv(p.W, 32); ba.nby[i - b0] = cdiv(p.y_end - R8W_Y0, 16);
            ba.y_end[i - b0] = p.y_end; ba.xr[i - b0] = R8W_X0 + R8W_TW * p.n_strips;
            btiles += 2 * ba.nbx[i - b0] + 2 * ba.nby[i - b0];
        }
        wa.nprob = ba.b.nprob = (int)(b1 - b0);
        if (up) {
            wa.b1 = m->d_r8b_up_b1; wa.w1pf = (const u32x4*)m->d_r8f_up_w1; wa.wpk = (const u32x4*)m->d_r8b_up_w; wa.bias = m->d_r8b_up_b;
            ba.b.w1pk = (const u32x4*)m->d_r8b_up_w1; ba.b.b1 = m->d_r8b_up_b1; ba.b.wpk = (const u32x4*)m->d_r8b_up_w; ba.b.bias = m->d_r8b_up_b;
        } else {
            wa.b1 = m->det_first.d_b; wa.w1pf = (const u32x4*)m->d_r8f_down_w1; wa.wpk = (const u32x4*)m->d_r8b_down_w; wa.bias = m->d_r8b_down_b;
            ba.b.w1 = m->d_r8b_down_w1r ? m->d_r8b_down_w1r : m->det_first.d_w; ba.b.b1 = m->det_first.d_b; ba.b.wpk = (const u32x4*)m->d_r8b_down_w; ba.b.bias = m->d_r8b_down_b;
        }
        TL sub(in0.begin() + b0, in0.begin() + b1);
        const std::string what = (up ? "unet_up_0 (conv1[16->8]+3xconvR+add) " : "unet_down_0 (conv1+3xconvR+add+pool) ") + dims_of(sub);
        double area = 0;
        for (size_t i = b0; i < b1; ++i) area += (double)in0[i].H * in0[i].W;
        const double wf = wshare / area;                     // the walker's share of the pages' pixels
        int units = items;
        wa.xm = oneshot_map(m, items, &units);
        // the border tiles (4 % of the pixels in 256-thread blocks) run BESIDE the walkers on a stream of their own: they fill the walkers' tail.
        // While launch times are recorded everything stays on one stream.
        asep_aru::Lane& L = *m->cur;
        const bool beside = !m->profiling && L.bside;
        if (beside) {
            ASEP_HIP_CHECK_THROW(hipEventRecord(L.ev_bfork, m->stream));
            ASEP_HIP_CHECK_THROW(hipStreamWaitEvent(L.bside, L.ev_bfork, 0));
        }
        {
            ProfScope ps(m, up ? "res8w_kernel<true>" : "res8w_kernel<false>", flops * wf, what);
            ps.bytes = bytes * wf;
            if (up) hipLaunchKernelGGL(res8w_kernel<true>, dim3(units), dim3(64), 0, m->stream, wa);
            else hipLaunchKernelGGL(res8w_kernel<false>, dim3(units), dim3(64), 0, m->stream, wa);
        }
        {
            ProfScope ps(m, up ? "res8wb_kernel<true>" : "res8wb_kernel<false>", flops * (1.0 - wf), what);
            ps.bytes = bytes * (1.0 - wf);
            if (up) hipLaunchKernelGGL(res8wb_kernel<true>, dim3(btiles), dim3(256), 0, beside ? L.bside : m->stream, ba);
            else hipLaunchKernelGGL(res8wb_kernel<false>, dim3(btiles), dim3(256), 0, beside ? L.bside : m->stream, ba);
        }
        if (beside) {
            ASEP_HIP_CHECK_THROW(hipEventRecord(L.ev_bjoin, L.bside));
            ASEP_HIP_CHECK_THROW(hipStreamWaitEvent(m->stream, L.ev_bjoin, 0));
        }
    }
}

void run_res8b(asep_aru* m, bool up, const TL& a0, const TL* a1, const std::vector<const float*>& stats, bool want_pool, TL* d_out, TL* pool_out) {
    for (const Tensor& t : a0) {
        d_out->push_back(new_tensor_bf(m, t.H, t.W, 8));
        if (want_pool) pool_out->push_back(new_tensor_bf(m, cdiv(t.H, 2), cdiv(t.W, 2), 8));
    }
    if (!m->fused_act && m->use_walk && (up || m->walk_mode == 1) && (up ? m->d_r8f_up_w1 != nullptr : m->d_r8f_down_w1 != nullptr)) {
        // pages with room for at least four strips and two tile rows of walker region go to the strip walker, the others stay on the tile kernels
        TL ws, wd, wo, wp, rs, rd, ro, rp;
        std::vector<const float*> wst, rst;
        for (size_t i = 0; i < a0.size(); ++i) {
            const Tensor& t = a0[i];
            const bool fits = (t.W - 4 - R8W_X0) / R8W_TW >= 4 && t.H - 4 - R8W_Y0 >= 32 && (size_t)t.H * t.W < ((size_t)1 << 28);
            (fits ? ws : rs).push_back(t);
            if (up) (fits ? wd : rd).push_back((*a1)[i]);
            (fits ? wo : ro).push_back((*d_out)[i]);
            if (want_pool) (fits ? wp : rp).push_back((*pool_out)[i]);
            if (!stats.empty()) (fits ? wst : rst).push_back(stats[i]);
        }
        if (!ws.empty()) {
            run_res8w(m, up, ws, up ? &wd : nullptr, wst, wo, want_pool ? &wp : nullptr);
            if (!rs.empty()) run_res8b_tiles(m, up, rs, up ? &rd : nullptr, rst, want_pool, ro, want_pool ? &rp : nullptr);
            return;
        }
    }
    run_res8b_tiles(m, up, a0, a1, stats, want_pool, *d_out, pool_out);
}

// the tile kernels (res8f_kernel for interior tiles + res8b_tile for border tiles in one launch) on the given output tensors
void run_res8b_tiles(asep_aru* m, bool up, const TL& a0, const TL* a1, const std::vector<const float*>& stats, bool want_pool, const TL& outs, const TL* pool_out) {
    const TL* d_out = &outs;
    for (size_t b0 = 0; b0 < a0.size(); b0 += MAXP) {
        const size_t b1 = std::min(a0.size(), b0 + MAXP);
        Res8BArgs a{};
        int tiles = 0;
        double flops = 0, bytes = 0;
        for (size_t i = b0; i < b1; ++i) {
            bytes += tbytes(a0[i]) + (up ? tbytes((*a1)[i]) : 0.0) + tbytes((*d_out)[i]) + (want_pool ? tbytes((*pool_out)[i]) : 0.0);
            Res8BProb& p = a.p[i - b0];
            if (up) { p.skip = a0[i].bp(); p.dec = (*a1)[i].bp(); }
            else { p.img = a0[i].p; p.stats = stats.empty() ? nullptr : stats[i]; }
            p.out = (*d_out)[i].bp(); p.pool = want_pool ? (*pool_out)[i].bp() : nullptr;
            p.H = a0[i].H; p.W = a0[i].W;
            p.tiles_x = cdiv(a0[i].W, 32);
            p.tile_begin = tiles;
            tiles += p.tiles_x * cdiv(a0[i].H, 16);
            flops += 2.0 * a0[i].H * a0[i].W * (9.0 * (up ? 16 : 1) * 8 + 3 * 9.0 * 64);
        }
        a.nprob = (int)(b1 - b0);
        if (up) { a.w1pk = (const u32x4*)m->d_r8b_up_w1; a.b1 = m->d_r8b_up_b1; a.wpk = (const u32x4*)m->d_r8b_up_w; a.bias = m->d_r8b_up_b; }
        else { a.w1 = m->d_r8b_down_w1r ? m->d_r8b_down_w1r : m->det_first.d_w; a.b1 = m->det_first.d_b; a.wpk = (const u32x4*)m->d_r8b_down_w; a.bias = m->d_r8b_down_b; }
        TL sub(a0.begin() + b0, a0.begin() + b1);
        int units = tiles;
        a.xm = oneshot_map(m, tiles, &units);
        const std::string what = (up ? "unet_up_0 (conv1[16->8]+3xconvR+add) " : "unet_down_0 (conv1+3xconvR+add+pool) ") + dims_of(sub);
        bool small = true;                                   // res8f_kernel addresses its tensors with 32-bit byte offsets (16 bytes per pixel)
        for (size_t i = b0; i < b1; ++i) small = small && (size_t)a0[i].H * a0[i].W < ((size_t)1 << 28);
        if (m->fused_act) {                                  // elu / leaky: the general form for every tile
            ProfScope ps(m, std::string("res8b_kernel") + targs({tb(up), ti(m->fused_act)}), flops, what);
            ps.bytes = bytes;
            if (up && m->fused_act == 1) hipLaunchKernelGGL((res8b_kernel<true, 1>), dim3(units), dim3(256), 0, m->stream, a);
            else if (up) hipLaunchKernelGGL((res8b_kernel<true, 2>), dim3(units), dim3(256), 0, m->stream, a);
            else if (m->fused_act == 1) hipLaunchKernelGGL((res8b_kernel<false, 1>), dim3(units), dim3(256), 0, m->stream, a);
            else hipLaunchKernelGGL((res8b_kernel<false, 2>), dim3(units), dim3(256), 0, m->stream, a);
        } else if (small && (up || m->d_r8f_down_w1)) {
            // lean form for interior tiles (their 24 x 40 input window inside the image), general form for border tiles, one launch
            Res8BArgs f = a;
            if (!up) f.w1pk = (const u32x4*)m->d_r8f_down_w1;
            else f.w1pf = (const u32x4*)m->d_r8f_up_w1;
            ProfScope ps(m, up ? "res8f_kernel<true>" : "res8f_kernel<false>", flops, what);
            ps.bytes = bytes;
            if (up) hipLaunchKernelGGL(res8f_kernel<true>, dim3(units), dim3(256), 0, m->stream, f);
            else hipLaunchKernelGGL(res8f_kernel<false>, dim3(units), dim3(256), 0, m->stream, f);
        } else {
            ProfScope ps(m, up ? "res8b_kernel<true,0>" : "res8b_kernel<false,0>", flops, what);
            ps.bytes = bytes;
            if (up) hipLaunchKernelGGL(res8b_kernel<true>, dim3(units), dim3(256), 0, m->stream, a);
            else hipLaunchKernelGGL(res8b_kernel<false>, dim3(units), dim3(256), 0, m->stream, a);
        }
    }
}

Tensor new_tensor_bf(asep_aru* m, int H, int W, int C) {
    Tensor t;
    t.H = H; t.W = W; t.C = C; t.bf = true;
    t.p = (float*)m->cur->pool.get(t.count() * sizeof(bf16_t));
    return t;
}

#define ASEP_CONVB_LAUNCH(KH_, KW_, MODE_, MT_, WM_, TH_, MB_)                                                         \
    do {                                                                                                               \
        ps.set_name("convb_kernel" + targs({ti(KH_), ti(KW_), ti(MODE_), ti(MT_), ti(WM_), ti(TH_), ti(MB_), tb(false), ti(4)})); \
        hipLaunchKernelGGL((convb_kernel<KH_, KW_, MODE_, MT_, WM_, TH_, MB_, false, 4>), grid, dim3(256), 0, m->stream, a); \
    } while (0)
#define ASEP_CONVB_LAUNCH8(KH_, KW_, MODE_, MT_, WM_, TH_, MB_, RES_)                                                  \
    do {                                                                                                               \
        ps.set_name("convb_kernel" + targs({ti(KH_), ti(KW_), ti(MODE_), ti(MT_), ti(WM_), ti(TH_), ti(MB_), tb(RES_), ti(8)})); \
        hipLaunchKernelGGL((convb_kernel<KH_, KW_, MODE_, MT_, WM_, TH_, MB_, RES_, 8>), grid, dim3(512), 0, m->stream, a); \
    } while (0)
#define ASEP_CONVB_LAUNCH_RES(KH_, KW_, MODE_, MT_, WM_, TH_, MB_)                                                     \
    do {                                                                                                               \
        ps.set_name("convb_kernel" + targs({ti(KH_), ti(KW_), ti(MODE_), ti(MT_), ti(WM_), ti(TH_), ti(MB_), tb(true), ti(4)})); \
        hipLaunchKernelGGL((convb_kernel<KH_, KW_, MODE_, MT_, WM_, TH_, MB_, true, 4>), grid, dim3(256), 0, m->stream, a); \
    } while (0)

// maxpool2 of a ReLU layer's bf16 output (behind convr_kernel's RES form)
TL run_maxpool2b(asep_aru* m, const TL& in) {
    TL out;
    for (const Tensor& t : in) out.push_back(new_tensor_bf(m, cdiv(t.H, 2), cdiv(t.W, 2), t.C));
    for (size_t b0 = 0; b0 < in.size(); b0 += MAXP) {
        const size_t b1 = std::min(in.size(), b0 + MAXP);
        MaxPoolBArgs a{};
        int blocks = 0;
        double bytes = 0;
        for (size_t i = b0; i < b1; ++i) {
            bytes += tbytes(in[i]) + tbytes(out[i]);
            MaxPoolBProb& p = a.p[i - b0];
            p.in = in[i].bp(); p.out = out[i].bp(); p.H = in[i].H; p.W = in[i].W;
            p.blk_begin = blocks;
            blocks += (int)((out[i].count() / 8 + 255) / 256);
        }
        a.nprob = (int)(b1 - b0); a.C = in[0].C;
        ProfScope ps(m, "maxpool2b_kernel", 0.0);
        ps.bytes = bytes;
        hipLaunchKernelGGL(maxpool2b_kernel, dim3(blocks), dim3(256), 0, m->stream, a);
    }
    return out;
}

// the 64 -> 64 3x3 layers with the filter in registers (convr_kernels.h): one wave per SIMD, a wave = the pipeline of a 32-column strip
TL run_convr(asep_aru* m, const std::string& scope, const PackedConv& pc, const TL& in0, bool relu_in, bool relu_out, const TL* res) {
    TL out;
    for (const Tensor& t : in0) out.push_back(new_tensor_bf(m, t.H, t.W, 64));
    for (size_t b0 = 0; b0 < in0.size(); b0 += MAXP) {
        const size_t b1 = std::min(in0.size(), b0 + MAXP);
        ConvRArgs a{};
        int total = 0;
        const int cin = in0[0].C;
        double flops = 0, bytes = 9.0 * cin * 64 * 2.0;
        for (size_t i = b0; i < b1; ++i) {
            bytes += tbytes(in0[i]) + tbytes(out[i]) + (res ? tbytes((*res)[i]) : 0.0);
            ConvRProb& p = a.p[i - b0];
            p.in = in0[i].bp(); p.res = res ? (*res)[i].bp() : nullptr; p.out = out[i].bp();
            p.H = in0[i].H; p.W = in0[i].W; p.strips = cdiv(in0[i].W, 32); p.begin = total;
            total += p.strips * p.H;
            flops += 2.0 * in0[i].H * in0[i].W * 9.0 * cin * 64.0;
        }
        a.nprob = (int)(b1 - b0); a.total = total;
        a.wpk = (const u32x4*)pc.d_wb; a.bias = pc.d_b;
        a.zero = m->d_zero_trash;
        // one wave per SIMD; a wave's range = total / waves rows (never fewer than 8: a range starts with three rows of latency)
        const int blocks = std::max(1, std::min(m->num_cus, total / 32));
        TL sub(in0.begin() + b0, in0.begin() + b1);
        ProfScope ps(m, "convr_kernel", flops, scope + " " + dims_of(sub) + " " + std::to_string(cin) + "->64");
        ps.bytes = bytes;
        ps.set_name("convr_kernel" + targs({tb(relu_in), tb(relu_out), tb(res != nullptr), ti(cin)}));
        if (cin == 32) hipLaunchKernelGGL((convr_kernel<false, false, false, 32>), dim3(blocks), dim3(256), 0, m->stream, a);
        else if (res) hipLaunchKernelGGL((convr_kernel<false, true, true>), dim3(blocks), dim3(256), 0, m->stream, a);
        else if (relu_in && relu_out) hipLaunchKernelGGL((convr_kernel<true, true>), dim3(blocks), dim3(256), 0, m->stream, a);
        else if (relu_in) hipLaunchKernelGGL((convr_kernel<true, false>), dim3(blocks), dim3(256), 0, m->stream, a);
        else if (relu_out) hipLaunchKernelGGL((convr_kernel<false, true>), dim3(blocks), dim3(256), 0, m->stream, a);
        else hipLaunchKernelGGL((convr_kernel<false, false>), dim3(blocks), dim3(256), 0, m->stream, a);
    }
    return out;
}

// stride-1 SAME conv of the bf16 path.  pooled != nullptr: the epilogue also writes maxpool2 of the output (always fused here);
// keep_full = false: only the pooled tensor is stored; pool_f32: the pooled tensor is fp32 (input of conv_c1out_kernel)
// relu_out = true on an elu / leaky graph (cfg.activation != 0): the layer's activation is that function (ConvBArgs::act), applied to the
// fp32 sums before the rounding; relu_out = false: identity (block-opening conv1)
TL run_convb(asep_aru* m, const std::string& scope, const TL& in0, const TL* in1, bool relu_in, bool relu_out, const TL* res,
             TL* pooled = nullptr, bool keep_full = true, bool pool_f32 = false) {
    const int act = relu_out ? m->cfg.activation : 0;
    if (act) relu_out = false;
    auto it = m->convs.find(scope);
    if (it == m->convs.end()) { set_error("internal: conv %s not packed", scope.c_str()); throw ArgError(); }
    const PackedConv& pc = it->second;
    const int cin = in0[0].C + (in1 ? (*in1)[0].C : 0);
    const int cin_w = pc.cin == 12 ? 16 : pc.cin;           // attention conv2 reads the head's zero-padded 16-channel plane
    if (cin != cin_w || !in0[0].bf) { set_error("internal: conv %s expects %d bf16 channels, got %d", scope.c_str(), cin_w, cin); throw ArgError(); }
    if (pc.bmode < 0 || !pc.d_wb || !((pc.kh == 3 && pc.kw == 3) || (pc.kh == 4 && pc.kw == 4)) || in0[0].C % 8 != 0 || pc.cout % 8 != 0) {
        set_error("conv %s (%dx%d, %d -> %d channels) is not served by the bf16 kernels", scope.c_str(), pc.kh, pc.kw, pc.cin, pc.cout);
        throw ArgError();
    }
    // output-channel tiles per block: 1 (cout 8 / 16), 2 (cout 32: one wave row, 16 x 32 pixels), 4 (cout >= 64: two wave
    // rows of two m-tiles, 8 x 32 pixels)
    if (m->use_convr && pc.kh == 3 && pc.kw == 3 && pc.bmode == 2 && !in1 && pc.cout == 64 && pc.mtiles == 4 && !act &&
        ((in0[0].C == 64 && (!res || (!relu_in && relu_out)) && (!pooled || (relu_out && keep_full && !pool_f32))) ||
         (in0[0].C == 32 && !res && !pooled && !relu_in && !relu_out))) {
        TL out = run_convr(m, scope, pc, in0, relu_in, relu_out, res);
        // (the block-closing layer of unet_down_3 also pools: convr_kernel has no fused pool; a ReLU output's 2 x 2 maxima are taken from the stored
        // tensor -- the same values convb_kernel's epilogue compares -- by a streaming kernel: 100 + 41 us against 160)
        if (pooled) *pooled = run_maxpool2b(m, out);
        return out;
    }
    const int mtb = pc.mtiles >= 4 ? 4 : pc.mtiles;
    if (pc.mtiles % mtb != 0 || mtb == 3) { set_error("conv %s: %d output tiles not instantiated", scope.c_str(), pc.mtiles); throw ArgError(); }
    const int th = (mtb == 4 || (mtb == 2 && pc.bmode == 2)) ? 8 : 16;
    TL out;
    if (keep_full || !pooled)
        for (const Tensor& t : in0) out.push_back(new_tensor_bf(m, t.H, t.W, pc.cout));
    if (pooled) {
        pooled->clear();
        for (const Tensor& t : in0) {
            Tensor q = pool_f32 ? new_tensor(m, cdiv(t.H, 2), cdiv(t.W, 2), pc.cout) : new_tensor_bf(m, cdiv(t.H, 2), cdiv(t.W, 2), pc.cout);
            pooled->push_back(q);
        }
    }
    for (size_t b0 = 0; b0 < in0.size(); b0 += MAXP) {
        const size_t b1 = std::min(in0.size(), b0 + MAXP);
        ConvBArgs a{};
        int tiles = 0;
        double flops = 0, bytes = (double)pc.kh * pc.kw * pc.cin * pc.cout * 2.0;
        for (size_t i = b0; i < b1; ++i) {
            bytes += tbytes(in0[i]) + (in1 ? tbytes((*in1)[i]) : 0.0) + (res ? tbytes((*res)[i]) : 0.0) + (out.empty() ? 0.0 : tbytes(out[i])) +
                     (pooled ? tbytes((*pooled)[i]) : 0.0);
            ConvBProb& p = a.p[i - b0];
            p.in0 = in0[i].bp(); p.in1 = in1 ? (*in1)[i].bp() : nullptr; p.res = res ? (*res)[i].bp() : nullptr;
            p.out = out.empty() ? nullptr : out[i].bp();
            p.pool = pooled ? (void*)(*pooled)[i].p : nullptr;
            p.H = in0[i].H; p.W = in0[i].W;
            p.tiles_x = cdiv(in0[i].W, 32);
            p.tile_begin = tiles;
            tiles += p.tiles_x * cdiv(in0[i].H, th);
            flops += 2.0 * in0[i].H * in0[i].W * pc.kh * pc.kw * (double)pc.cin * pc.cout;
        }
        a.nprob = (int)(b1 - b0);
        a.wpk = (const u32x4*)pc.d_wb; a.bias = pc.d_b;
        a.c0 = in0[0].C; a.c1 = in1 ? (*in1)[0].C : 0;
        a.cout = pc.cout; a.mtiles = pc.mtiles; a.groups = cin / 32;
        a.relu_in = relu_in; a.relu_out = relu_out; a.act = act; a.skip_full = pooled && !keep_full; a.pool_f32 = pool_f32;
        int units = tiles;
        a.xm = oneshot_map(m, tiles, &units);
        dim3 grid(units, pc.mtiles / mtb);
        TL sub(in0.begin() + b0, in0.begin() + b1);
        ProfScope ps(m, "convb_kernel", flops, scope + " " + dims_of(sub) + " " + std::to_string(pc.cin) + "->" + std::to_string(pc.cout));
        ps.bytes = bytes;
        bool res32 = res != nullptr;                          // the RESP kernels address the residual operand with 32-bit byte offsets
        for (size_t i = b0; res && i < b1; ++i) res32 = res32 && tbytes((*res)[i]) < 4294967296.0;
        const int key = pc.kh * 100 + pc.bmode * 10 + mtb + (th == 8 && mtb == 2 ? 1000 : 0) + (res32 && pc.bmode == 2 && pc.kh == 3 && mtb >= 2 ? 2000 : 0);
        switch (key) {
            case 3322: ASEP_CONVB_LAUNCH_RES(3, 3, 2, 2, 1, 8, 3); break;       // residual operand in the initial value (32- and >= 64-channel convR_2)
            case 2324: ASEP_CONVB_LAUNCH8(3, 3, 2, 2, 2, 8, 2, true); break;    // (the residual is part of the accumulators' initial value: the eight-wave form's registers)
            case 1322: ASEP_CONVB_LAUNCH(3, 3, 2, 2, 1, 8, 4); break;
            case 301: ASEP_CONVB_LAUNCH(3, 3, 0, 1, 1, 16, 3); break;
            case 302: ASEP_CONVB_LAUNCH(3, 3, 0, 2, 1, 16, 3); break;
            case 304: ASEP_CONVB_LAUNCH(3, 3, 0, 2, 2, 8, 3); break;
            case 311: ASEP_CONVB_LAUNCH(3, 3, 1, 1, 1, 16, 3); break;
            case 312: ASEP_CONVB_LAUNCH(3, 3, 1, 2, 1, 16, 3); break;
            case 314: ASEP_CONVB_LAUNCH(3, 3, 1, 2, 2, 8, 3); break;
            case 321: ASEP_CONVB_LAUNCH(3, 3, 2, 1, 1, 16, 3); break;
            case 324: ASEP_CONVB_LAUNCH8(3, 3, 2, 2, 2, 8, 2, false); break;   // >= 64 channels: eight waves over the same LDS tile
            case 411: ASEP_CONVB_LAUNCH(4, 4, 1, 1, 1, 16, 3); break;
            case 412: ASEP_CONVB_LAUNCH(4, 4, 1, 2, 1, 16, 3); break;
            case 414: ASEP_CONVB_LAUNCH(4, 4, 1, 2, 2, 8, 3); break;
            case 421: ASEP_CONVB_LAUNCH(4, 4, 2, 1, 1, 16, 2); break;
            case 422: ASEP_CONVB_LAUNCH(4, 4, 2, 2, 1, 16, 2); break;
            case 424: ASEP_CONVB_LAUNCH(4, 4, 2, 2, 2, 8, 1); break;       // (16 taps x 4 m-tiles of fragments: one block per CU is what its registers allow)
            default: set_error("conv %s: bf16 kernel variant %d not instantiated", scope.c_str(), key); throw ArgError();
        }
    }
    return out;
}

// fused tail of a residual block (3 x convR + t + ReLU [+ pool]) for the 8- / 16-channel levels
bool has_resb(asep_aru* m, const std::string& scope) { return m->resb.count(scope) != 0; }

TL run_resb_tail(asep_aru* m, const std::string& scope, const TL& t, TL* pooled) {
    const asep_aru::ResB& rb = m->resb.at(scope);
    if (t[0].C != rb.C || !t[0].bf) { set_error("internal: residual tail %s expects %d bf16 channels", scope.c_str(), rb.C); throw ArgError(); }
    TL out;
    for (const Tensor& x : t) out.push_back(new_tensor_bf(m, x.H, x.W, rb.C));
    if (pooled) {
        pooled->clear();
        for (const Tensor& x : t) pooled->push_back(new_tensor_bf(m, cdiv(x.H, 2), cdiv(x.W, 2), rb.C));
    }
    for (size_t b0 = 0; b0 < t.size(); b0 += MAXP) {
        const size_t b1 = std::min(t.size(), b0 + MAXP);
        ResBArgs a{};
        int tiles = 0;
        double flops = 0, bytes = 3.0 * 9.0 * rb.C * rb.C * 2.0;
        for (size_t i = b0; i < b1; ++i) {
            bytes += tbytes(t[i]) + tbytes(out[i]) + (pooled ? tbytes((*pooled)[i]) : 0.0);
            ResBProb& p = a.p[i - b0];
            p.t = t[i].bp(); p.out = out[i].bp(); p.pool = pooled ? (*pooled)[i].bp() : nullptr;
            p.H = t[i].H; p.W = t[i].W;
            p.tiles_x = cdiv(t[i].W, RB_TW);
            p.tile_begin = tiles;
            tiles += p.tiles_x * cdiv(t[i].H, RB_TH);
            flops += 2.0 * t[i].H * t[i].W * 3 * 9.0 * rb.C * rb.C;
        }
        a.nprob = (int)(b1 - b0);
        a.wpk = (const u32x4*)rb.d_w; a.bias = rb.d_b;
        int units = tiles;
        if (rb.C == 32) {                                    // persistent kernel: table (its units are walked with a grid stride)
            std::vector<TileDims> probs;
            for (int i = 0; i < a.nprob; ++i) probs.push_back({a.p[i].tiles_x, cdiv(a.p[i].H, RB_TH), a.p[i].tile_begin});
            a.sched = (m->use_xcd_sched && tiles >= 8 * 64) ? xcd_schedule(m, probs, tiles) : nullptr;
        } else {
            a.xm = oneshot_map(m, tiles, &units);
        }
        TL sub(t.begin() + b0, t.begin() + b1);
        bool small16 = true;                                 // res16f_kernel addresses its tensors with 32-bit byte offsets (32 bytes per pixel)
        for (const Tensor& x : sub) small16 = small16 && (size_t)x.H * x.W < ((size_t)1 << 27);
        const std::string what = scope + " (3xconvR+add" + (pooled ? "+pool) " : ") ") + dims_of(sub);
        if (rb.C == 32) {
            static bool attr[3] = {false, false, false};
            const int act = m->fused_act;
            const void* fn = act == 1 ? (const void*)res32_tail_kernel<1> : act == 2 ? (const void*)res32_tail_kernel<2> : (const void*)res32_tail_kernel<0>;
            if (!attr[act]) {
                ASEP_HIP_CHECK_THROW(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, Res32Layout::BYTES));
                attr[act] = true;
            }
            ProfScope ps(m, "res32_tail_kernel" + targs({ti(act)}), flops, what);
            ps.bytes = bytes;
            a.ntiles = tiles;
            const dim3 g32(std::min(tiles, m->num_cus));
            if (act == 1) hipLaunchKernelGGL(res32_tail_kernel<1>, g32, dim3(512), Res32Layout::BYTES, m->stream, a);
            else if (act == 2) hipLaunchKernelGGL(res32_tail_kernel<2>, g32, dim3(512), Res32Layout::BYTES, m->stream, a);
            else hipLaunchKernelGGL(res32_tail_kernel<0>, g32, dim3(512), Res32Layout::BYTES, m->stream, a);
        } else if (m->fused_act) {                     // elu / leaky: the general form
            ProfScope ps(m, "resb_tail_kernel" + targs({ti(rb.C), ti(m->fused_act)}), flops, what);
            ps.bytes = bytes;
            if (rb.C == 8 && m->fused_act == 1) hipLaunchKernelGGL((resb_tail_kernel<8, 1>), dim3(units), dim3(256), 0, m->stream, a);
            else if (rb.C == 8) hipLaunchKernelGGL((resb_tail_kernel<8, 2>), dim3(units), dim3(256), 0, m->stream, a);
            else if (m->fused_act == 1) hipLaunchKernelGGL((resb_tail_kernel<16, 1>), dim3(units), dim3(256), 0, m->stream, a);
            else hipLaunchKernelGGL((resb_tail_kernel<16, 2>), dim3(units), dim3(256), 0, m->stream, a);
        } else if (rb.C == 16 && small16) {            // lean form for interior tiles, general form for border tiles, one launch
            ProfScope ps(m, "res16f_kernel", flops, what);
            ps.bytes = bytes;
            hipLaunchKernelGGL(res16f_kernel, dim3(units), dim3(256), 0, m->stream, a);
        } else {
            ProfScope ps(m, "resb_tail_kernel" + targs({ti(rb.C), ti(0)}), flops, what);
            ps.bytes = bytes;
            if (rb.C == 8) hipLaunchKernelGGL(resb_tail_kernel<8>, dim3(units), dim3(256), 0, m->stream, a);
            else hipLaunchKernelGGL(resb_tail_kernel<16>, dim3(units), dim3(256), 0, m->stream, a);
        }
    }
    return out;
}

TL run_deconvb(asep_aru* m, const std::string& scope, const TL& in, const TL& like, bool relu_out) {
    const int act = relu_out ? m->cfg.activation : 0;        // (elu / leaky graphs: the deconvolution's activation, layers.py:342-367)
    if (act) relu_out = false;
    auto it = m->convs.find(scope);
    if (it == m->convs.end()) { set_error("internal: deconv %s not packed", scope.c_str()); throw ArgError(); }
    const PackedConv& pc = it->second;
    if (pc.bmode < 1 || !pc.d_wb || in[0].C != pc.cin || !in[0].bf || pc.cout % 8 != 0) {
        set_error("deconv %s (%d -> %d channels) is not served by the bf16 kernels", scope.c_str(), pc.cin, pc.cout);
        throw ArgError();
    }
    TL out;
    for (size_t i = 0; i < in.size(); ++i) {
        if (cdiv(like[i].H, 2) != in[i].H || cdiv(like[i].W, 2) != in[i].W) {
            set_error("deconv %s: output %dx%d incompatible with input %dx%d", scope.c_str(), like[i].H, like[i].W, in[i].H, in[i].W);
            throw ArgError();
        }
        out.push_back(new_tensor_bf(m, like[i].H, like[i].W, pc.cout));
    }
    const int mt = pc.mtiles % 2 == 0 ? 2 : 1;
    const bool d8 = pc.d_wb8 && relu_out && !act;           // level 0 (16 -> 8): no LDS, whole pixels straight to HBM
    const int dth = d8 ? D8_RW : 8;                          // input rows per block (16 rows for one m-tile measured slower: 148 -> 187 us at level 0)
    const int dtw = d8 ? D8_TW : DCB_TW;
    for (size_t b0 = 0; b0 < in.size(); b0 += MAXP) {
        const size_t b1 = std::min(in.size(), b0 + MAXP);
        DeconvBArgs a{};
        int tiles = 0;
        double flops = 0, bytes = 9.0 * pc.cin * pc.cout * 2.0;
        for (size_t i = b0; i < b1; ++i) {
            bytes += tbytes(in[i]) + tbytes(out[i]);
            DeconvBProb& p = a.p[i - b0];
            p.in = in[i].bp(); p.out = out[i].bp();
            p.Hi = in[i].H; p.Wi = in[i].W; p.Ho = out[i].H; p.Wo = out[i].W;
            p.pbh = std::max((in[i].H - 1) * 2 + 3 - out[i].H, 0) / 2;
            p.pbw = std::max((in[i].W - 1) * 2 + 3 - out[i].W, 0) / 2;
            p.tiles_x = cdiv(in[i].W, dtw);
            p.tile_begin = tiles;
            tiles += p.tiles_x * cdiv(in[i].H, dth);
            flops += 2.0 * in[i].H * in[i].W * 9.0 * pc.cin * pc.cout;
        }
        a.nprob = (int)(b1 - b0);
        a.wpk = (const u32x4*)(d8 ? pc.d_wb8 : pc.d_wb); a.bias = pc.d_b;
        a.cin = pc.cin; a.cout = pc.cout; a.mtiles = pc.mtiles; a.groups = pc.cin / 32; a.relu_out = relu_out; a.act = act;
        int units = tiles;
        a.xm = oneshot_map(m, tiles, &units);
        dim3 grid(units, pc.mtiles / mt);
        TL sub(in.begin() + b0, in.begin() + b1);
        ProfScope ps(m, d8 ? std::string("deconvb8_kernel") : "deconvb_kernel" + targs({ti(pc.bmode), ti(mt), ti(dth)}), flops,
                     scope + " " + dims_of(sub) + " " + std::to_string(pc.cin) + "->" + std::to_string(pc.cout));
        ps.bytes = bytes;
        if (d8) hipLaunchKernelGGL(deconvb8_kernel, dim3(units), dim3(256), 0, m->stream, a);
        else if (pc.bmode == 1 && mt == 1) hipLaunchKernelGGL((deconvb_kernel<1, 1, 8>), grid, dim3(256), 0, m->stream, a);
        else if (pc.bmode == 1) hipLaunchKernelGGL((deconvb_kernel<1, 2, 8>), grid, dim3(256), 0, m->stream, a);
        else if (mt == 1) hipLaunchKernelGGL((deconvb_kernel<2, 1, 8>), grid, dim3(256), 0, m->stream, a);
        else hipLaunchKernelGGL((deconvb_kernel<2, 2, 8>), grid, dim3(256), 0, m->stream, a);
    }
    return out;
}

// first layer of the feature CNN on the bf16 path: fp32 image -> bf16 [H,W,8] (pre-ReLU t of unet_down_0)
TL run_direct_bf(asep_aru* m, const DirectConv& dc, const TL& imgs, const std::vector<const float*>& stats, bool activated = false) {
    if (dc.k != 3 || dc.cout != 8) { set_error("bf16 path: first-layer conv k=%d cout=%d not instantiated", dc.k, dc.cout); throw ArgError(); }
    TL out;
    for (const Tensor& t : imgs) out.push_back(new_tensor_bf(m, t.H, t.W, dc.cout));
    for (size_t b0 = 0; b0 < imgs.size(); b0 += MAXP) {
        const size_t b1 = std::min(imgs.size(), b0 + MAXP);
        C1Args a{};
        int tiles = 0;
        double flops = 0, bytes = 0;
        for (size_t i = b0; i < b1; ++i) {
            bytes += tbytes(imgs[i]) + tbytes(out[i]);
            C1Prob& p = a.p[i - b0];
            p.img = imgs[i].p; p.out = out[i].p; p.stats = stats.empty() ? nullptr : stats[i];
            p.H = imgs[i].H; p.W = imgs[i].W;
            p.tiles_x = cdiv(imgs[i].W, 64);
            p.tile_begin = tiles;
            tiles += p.tiles_x * cdiv(imgs[i].H, 4);
            flops += 2.0 * imgs[i].H * imgs[i].W * dc.k * dc.k * dc.cout;
        }
        a.nprob = (int)(b1 - b0);
        a.w = dc.d_w; a.bias = dc.d_b; a.relu = (activated && m->cfg.activation == 0) ? 1 : 0; a.act = activated ? m->cfg.activation : 0;
        ProfScope ps(m, "conv_c1_kernel<3,8,true>", flops);
        ps.bytes = bytes;
        hipLaunchKernelGGL((conv_c1_kernel<3, 8, true>), dim3(tiles), dim3(256), 0, m->stream, a);
    }
    return out;
}

TL run_chansum_bf(asep_aru* m, const TL& in) {
    TL out;
    for (const Tensor& t : in) out.push_back(new_tensor(m, t.H, t.W, 1));
    for (size_t b0 = 0; b0 < in.size(); b0 += MAXP) {
        const size_t b1 = std::min(in.size(), b0 + MAXP);
        PoolBArgs a{};
        int blocks = 0;
        double bytes = 0;
        for (size_t i = b0; i < b1; ++i) {
            bytes += tbytes(in[i]) + tbytes(out[i]);
            PoolBProb& p = a.p[i - b0];
            p.in = in[i].bp(); p.out = out[i].p; p.H = in[i].H; p.W = in[i].W;
            p.blk_begin = blocks;
            blocks += (int)(((size_t)in[i].H * in[i].W + POOL_ITEMS - 1) / POOL_ITEMS);
        }
        a.nprob = (int)(b1 - b0);
        a.C = in[0].C;
        ProfScope ps(m, "chansumb_kernel", 0.0);
        ps.bytes = bytes;
        hipLaunchKernelGGL(chansumb_kernel, dim3(blocks), dim3(256), 0, m->stream, a);
    }
    return out;
}

// ---- network schedule (ARU_v1.py), evaluated for all problems in lock step ---------------------------------
// residual block: conv1 (identity) -> t ; relu ; (res_depth-1) x conv+relu ; conv (identity) ; +t ; relu
// ---- graph variants (asep_aru_cfg.activation != 0 and / or plain_u; fp32 path only): the layer kernels store pre-activation values,
//      act_kernel follows (on the pooled tensor too: the activations are increasing, the fused 2x2 max commutes with them) ----
void apply_act(asep_aru* m, TL& l) {
    if (l.empty()) return;
    for (size_t b0 = 0; b0 < l.size(); b0 += MAXP) {
        const size_t b1 = std::min(l.size(), b0 + MAXP);
        PoolArgs a{};
        int blocks = 0;
        double bytes = 0;
        for (size_t i = b0; i < b1; ++i) {
            bytes += 2.0 * tbytes(l[i]);
            PoolProb& p = a.p[i - b0];
            p.in = l[i].p; p.out = l[i].p; p.H = p.Ho = l[i].H; p.W = p.Wo = l[i].W;
            p.blk_begin = blocks;
            blocks += (int)((l[i].count() + (size_t)POOL_ITEMS * 4 - 1) / ((size_t)POOL_ITEMS * 4));
        }
        a.nprob = (int)(b1 - b0);
        a.C = l[0].C;
        ProfScope ps(m, "act_kernel", 0.0);
        ps.bytes = bytes;
        hipLaunchKernelGGL(act_kernel, dim3(blocks), dim3(256), 0, m->stream, a, m->cfg.activation);
    }
}
// conv / deconv / first conv followed by the graph's activation (act = false: identity layers).  Round 4: elu / leaky are applied in the
// epilogue of the producing kernel (ConvArgs::act: the same arithmetic as act_kernel, so the results are bit-identical to the separate pass
// of round 3, which ASEP_FUSE_ACT=0 still selects): a variant's layer is one launch and one write instead of launch + read + write.
// The fused 2x2 max pool sees the ACTIVATED values (both functions are increasing: the same as activating the pooled maximum).
TL conv_act(asep_aru* m, const std::string& scope, const TL& in0, const TL* in1, bool relu_in, bool act, const TL* res,
            TL* pooled = nullptr, bool keep_full = true) {
    if (m->cfg.activation == 0 || !act) return run_conv(m, scope, in0, in1, relu_in, act, res, pooled, keep_full);
    if (m->fuse_act) return run_conv(m, scope, in0, in1, relu_in, false, res, pooled, keep_full, m->cfg.activation);
    TL out = run_conv(m, scope, in0, in1, relu_in, false, res, pooled, keep_full);
    apply_act(m, out);
    if (pooled) apply_act(m, *pooled);
    return out;
}
TL deconv_act(asep_aru* m, const std::string& scope, const TL& in, const TL& like) {
    if (m->cfg.activation == 0) return run_deconv(m, scope, in, like, true);
    if (m->fuse_act) return run_deconv(m, scope, in, like, false, m->cfg.activation);
    TL out = run_deconv(m, scope, in, like, false);
    apply_act(m, out);
    return out;
}
TL direct_act(asep_aru* m, const DirectConv& dc, const TL& imgs, bool act, const std::vector<const float*>& stats) {
    if (m->cfg.activation == 0 || !act) return run_direct(m, dc, imgs, act, stats);
    if (m->fuse_act) return run_direct(m, dc, imgs, false, stats, m->cfg.activation);
    TL out = run_direct(m, dc, imgs, false, stats);
    apply_act(m, out);
    return out;
}

TL res_block_tail(asep_aru* m, const std::string& scope, const TL& t, TL* pooled = nullptr) {
    if (m->bf16 && has_resb(m, scope)) return run_resb_tail(m, scope, t, pooled);      // one kernel for the whole tail
    TL r = t;
    const int rd = m->cfg.res_depth;
    for (int i = 0; i < rd; ++i) {
        const bool last = (i == rd - 1);
        const std::string sc = scope + "/convR_" + std::to_string(i);
        // (relu_in of the first conv: the ReLU between conv1 and convR_0 -- a ReLU in every activation variant, ARU_v1.py:214)
        r = m->bf16 ? run_convb(m, sc, r, nullptr, /*relu_in=*/i == 0, /*relu_out=*/true, last ? &t : nullptr, last ? pooled : nullptr)
                    : conv_act(m, sc, r, nullptr, /*relu_in=*/i == 0, /*act=*/true, last ? &t : nullptr, last ? pooled : nullptr);
    }
    return r;
}

// names[i] = end-point prefix of problem i ("scale_<s>" for page 0, "p<b>/scale_<s>" otherwise)
TL det_cnn(asep_aru* m, const TL& imgs, const std::vector<std::string>& names, const std::vector<const float*>& stats) {
    const int n = m->cfg.scale_space_num;
    std::vector<TL> skips;
    TL u = imgs;
    auto publish = [&](const TL& l, const std::string& suffix) {
        for (size_t i = 0; i < l.size(); ++i) m->endpoints[names[i] + suffix] = l[i];
    };
    for (int l = 0; l < n; ++l) {
        const std::string scope = "aru_net/featMapG/unet_down_" + std::to_string(l);
        if (m->bf16 && l == 0 && m->d_r8b_down_w && m->det_first.k == 3 && m->det_first.cout == 8) {
            TL d, pooled;                                    // the whole block in one kernel: image -> d0 (+ pool)
            run_res8b(m, false, imgs, nullptr, stats, n > 1, &d, &pooled);
            skips.push_back(d);
            publish(d, "_unet_down_0_conv");
            u = n > 1 ? pooled : d;
            continue;
        }
        if (m->bf16 && m->cfg.plain_u) {                     // graph 'U' (ARU_v1.py:228-233): conv1 + conv2, both activated, layer by layer
            TL c1 = (l == 0) ? run_direct_bf(m, m->det_first, imgs, stats, true) : run_convb(m, scope + "/conv1", u, nullptr, false, true, nullptr);
            TL pooled;
            TL d = run_convb(m, scope + "/conv2", c1, nullptr, false, true, nullptr, l < n - 1 ? &pooled : nullptr);
            skips.push_back(d);
            publish(d, "_unet_down_" + std::to_string(l) + "_conv");
            u = (l < n - 1) ? pooled : d;
            continue;
        }
        if (m->bf16) {
            // native bf16 path: conv1 -> t (bf16), then the block tail (one kernel at 8 / 16 channels, three convs above)
            TL t = (l == 0) ? run_direct_bf(m, m->det_first, imgs, stats) : run_convb(m, scope + "/conv1", u, nullptr, false, false, nullptr);
            TL pooled;
            TL d = res_block_tail(m, scope, t, l < n - 1 ? &pooled : nullptr);
            skips.push_back(d);
            publish(d, "_unet_down_" + std::to_string(l) + "_conv");
            u = (l < n - 1) ? pooled : d;
            continue;
        }
        if (l == 0 && (m->use_fused8 || (m->fused8_var && r8v_fits(imgs))) && m->d_r8_down_wr) {
            TL d, pooled;
            run_res8_down(m, imgs, stats, n > 1, &d, &pooled);
            skips.push_back(d);
            publish(d, "_unet_down_0_conv");
            u = n > 1 ? pooled : d;
            continue;
        }
        TL pooled, d;
        if (m->cfg.plain_u) {                                // graph 'U': conv1 + conv2, both activated (ARU_v1.py:228-233)
            TL c1 = (l == 0) ? direct_act(m, m->det_first, imgs, true, stats) : conv_act(m, scope + "/conv1", u, nullptr, false, true, nullptr);
            d = conv_act(m, scope + "/conv2", c1, nullptr, false, true, nullptr, l < n - 1 ? &pooled : nullptr);
        } else {
            TL t = (l == 0) ? run_direct(m, m->det_first, imgs, false, stats)
                            : run_conv(m, scope + "/conv1", u, nullptr, false, false, nullptr);
            d = res_block_tail(m, scope, t, l < n - 1 ? &pooled : nullptr);    // the block's last conv also emits maxpool2(d)
        }
        skips.push_back(d);
        publish(d, "_unet_down_" + std::to_string(l) + "_conv");
        u = (l < n - 1) ? pooled : d;
    }
    for (int l = n - 2; l >= 0; --l) {
        const std::string scope = "aru_net/featMapG/unet_up_" + std::to_string(l);
        const TL& skip = skips[l];
        TL v = m->bf16 ? run_deconvb(m, scope + "/deconv", u, skip, true) : deconv_act(m, scope + "/deconv", u, skip);
        publish(v, "_unet_up_" + std::to_string(l) + "_deconv");
        if (m->bf16 && l == 0 && m->d_r8b_up_w1) {
            TL d, none;                                      // conv1 over [skip, deconv] + the tail in one kernel
            run_res8b(m, true, skip, &v, {}, false, &d, &none);
            u = d;
        } else if (m->bf16 && m->cfg.plain_u) {              // ARU_v1.py:283-288
            TL c1 = run_convb(m, scope + "/conv1", skip, &v, false, true, nullptr);
            u = run_convb(m, scope + "/conv2", c1, nullptr, false, true, nullptr);
        } else if (m->bf16) {
            TL t = run_convb(m, scope + "/conv1", skip, &v, false, false, nullptr);   // concat [skip, deconv]
            u = res_block_tail(m, scope, t);
        } else if (l == 0 && (m->use_fused8 || (m->fused8_var && r8v_fits(skip))) && m->d_r8_up_w1) {
            u = run_res8_up(m, skip, v);
        } else if (m->cfg.plain_u) {                         // ARU_v1.py:283-288
            TL c1 = conv_act(m, scope + "/conv1", skip, &v, false, true, nullptr);
            u = conv_act(m, scope + "/conv2", c1, nullptr, false, true, nullptr);
        } else {
            TL t = run_conv(m, scope + "/conv1", skip, &v, false, false, nullptr);   // concat [skip, deconv]
            u = res_block_tail(m, scope, t);
        }
        publish(u, "_unet_up_" + std::to_string(l) + "_conv");
    }
    return u;
}

TL att_cnn(asep_aru* m, const TL& imgs, const std::vector<const float*>& stats) {
    const std::string p = "aru_net/attMapG/attPart/conv";
    TL y;
    if (m->bf16 && !m->d_att_head) { set_error("bf16 path: attention head 4x4 / 12 channels expected"); throw ArgError(); }
    // the fused head (conv1 + activation + pool, one pooled pixel per thread) also serves the elu / leaky variants (round 4): the pool
    // is taken on the pre-activation values, the activation on the maximum
    const bool head_variant = m->d_att_head && !m->use_fused8 && m->fused8_wanted && m->fuse_act && m->cfg.activation != 0 && m->r8_valu && !m->bf16;
    if (m->d_att_head && (m->use_fused8 || head_variant || m->bf16)) {
        // conv1 + ReLU + pool1 fused (the full-resolution 12-channel tensor is never materialised)
        // (bf16 path: the head writes a 16-channel bf16 plane, channels 12..15 zero)
        for (const Tensor& t : imgs) y.push_back(m->bf16 ? new_tensor_bf(m, cdiv(t.H, 2), cdiv(t.W, 2), 16) : new_tensor(m, cdiv(t.H, 2), cdiv(t.W, 2), 12));
        bool headb = m->bf16 && m->d_att_headb && m->cfg.activation == 0;
        // (att_headb_kernel addresses the pooled plane with 32-bit element offsets: images of 2^28 pixels and more take the vector-ALU head below,
        //  which refuses them with an error instead of wrapping)
        for (const Tensor& t : imgs) headb = headb && (size_t)t.H * t.W < ((size_t)1 << 28);
        for (size_t b0 = 0; headb && b0 < imgs.size(); b0 += MAXP) {
            // bf16 path, ReLU graph: the head on the bf16 MFMA (image and filter as bfloat16 like the feature CNN's first layer)
            const size_t b1 = std::min(imgs.size(), b0 + MAXP);
            AttHeadBArgs a{};
            int tiles = 0;
            double flops = 0, bytes = 0;
            for (size_t i = b0; i < b1; ++i) {
                bytes += tbytes(imgs[i]) + tbytes(y[i]);
                C1Prob& q = a.p[i - b0];
                q.img = imgs[i].p; q.out = y[i].p; q.stats = stats.empty() ? nullptr : stats[i];
                q.H = imgs[i].H; q.W = imgs[i].W;
                q.tiles_x = cdiv(imgs[i].W, ATTB_TW);
                q.tile_begin = tiles;
                tiles += q.tiles_x * cdiv(imgs[i].H, ATTB_TH);
                flops += 2.0 * imgs[i].H * imgs[i].W * 16.0 * 12;
            }
            a.nprob = (int)(b1 - b0);
            a.wpk = (const u32x4*)m->d_att_headb; a.bias = m->att_first.d_b;
            int units = tiles;
            a.xm = oneshot_map(m, tiles, &units);
            ProfScope ps(m, "att_headb_kernel", flops);
            ps.bytes = bytes;
            hipLaunchKernelGGL(att_headb_kernel, dim3(units), dim3(256), 0, m->stream, a);
        }
        for (size_t b0 = 0; !headb && b0 < imgs.size(); b0 += MAXP) {
            const size_t b1 = std::min(imgs.size(), b0 + MAXP);
            AttHeadArgs a{};
            int tiles = 0;
            double flops = 0, bytes = 0;
            for (size_t i = b0; i < b1; ++i) {
                bytes += tbytes(imgs[i]) + tbytes(y[i]);
                C1Prob& q = a.p[i - b0];
                q.img = imgs[i].p; q.out = y[i].p; q.stats = stats.empty() ? nullptr : stats[i];
                q.H = imgs[i].H; q.W = imgs[i].W;
                q.tiles_x = cdiv(imgs[i].W, ATT_TW);
                q.tile_begin = tiles;
                tiles += q.tiles_x * cdiv(imgs[i].H, ATT_TH);
                flops += 2.0 * imgs[i].H * imgs[i].W * 16.0 * 12;
            }
            a.nprob = (int)(b1 - b0);
            a.wpk = (const f32x4*)m->d_att_head; a.bias = m->att_first.d_b; a.w = m->att_first.d_w; a.act = m->cfg.activation;
            bool valu = m->r8_valu || m->bf16;               // vector-ALU form (32-bit output offsets, see run_res8_down)
            for (size_t i = b0; i < b1; ++i) valu = valu && (size_t)imgs[i].H * imgs[i].W < ((size_t)1 << 28);
            if (m->bf16 && !valu) { set_error("bf16 path: image too large for the attention head kernel"); throw ArgError(); }
            if (head_variant && !valu) { set_error("attention head of an elu / leaky graph: image too large for the vector-ALU kernel"); throw ArgError(); }
            ProfScope ps(m, m->bf16 ? "att_headv_kernel<true>" : (valu ? "att_headv_kernel<false>" : "att_head_kernel"), flops);
            ps.bytes = bytes;
            if (m->bf16) hipLaunchKernelGGL(att_headv_kernel<true>, dim3(tiles), dim3(256), 0, m->stream, a);
            else if (valu) hipLaunchKernelGGL(att_headv_kernel<false>, dim3(tiles), dim3(256), 0, m->stream, a);
            else hipLaunchKernelGGL(att_head_kernel, dim3(tiles), dim3(256), 0, m->stream, a);
        }
    } else {
        y = direct_act(m, m->att_first, imgs, true, stats);
        y = run_pool(m, y, POOL_MAX);
    }
    TL pooled;
    if (m->bf16) {
        run_convb(m, p + "2", y, nullptr, false, true, nullptr, &pooled, /*keep_full=*/false);
        y = pooled;
        // the last pooled tensor (1/8 resolution, 32 channels) is written as fp32: conv4 is the fp32 vector-ALU kernel
        run_convb(m, p + "3", y, nullptr, false, true, nullptr, &pooled, /*keep_full=*/false, /*pool_f32=*/true);
        y = pooled;
        return conv_act(m, p + "4", y, nullptr, false, true, nullptr);       // (fp32 vector-ALU kernel; the graph's activation)
    }
    conv_act(m, p + "2", y, nullptr, false, true, nullptr, &pooled, /*keep_full=*/false);   // conv + ReLU + pool in one kernel
    y = pooled;
    conv_act(m, p + "3", y, nullptr, false, true, nullptr, &pooled, /*keep_full=*/false);
    y = pooled;
    y = conv_act(m, p + "4", y, nullptr, false, true, nullptr);
    return y;
}

// B pages of identical size through the net; problems = pages x scales
int forward_impl(asep_aru* m, asep_aru::Lane& L, int page0, int B, const float* const* d_imgs, const int32_t* Hs, const int32_t* Ws,
                 float* const* d_outs, uint8_t* const* d_u8s, uint8_t* const* d_masks, float threshold) {
    const asep_aru_cfg& cfg = m->cfg;
    hipStream_t stream = L.s;
    m->stream = stream;
    m->cur = &L;
    L.pool.begin();
    try {
        const int nsc = cfg.use_attention ? cfg.num_scales_att : 1;
        if (nsc > MAX_SCALES) { set_error("num_scales_att %d > %d", nsc, MAX_SCALES); return ASEP_ERR_UNSUPPORTED; }
        // problem order: page-major, scale-minor
        TL level0;
        std::vector<const float*> stats0;
        for (int b = 0; b < B; ++b) {
            Tensor img;
            img.p = const_cast<float*>(d_imgs[b]);
            img.H = Hs[b]; img.W = Ws[b]; img.C = 1;            // (pages of a call may differ in size: every launch carries per-problem dims)
            level0.push_back(img);
            const float* st = nullptr;
            if (cfg.mvn) {
                const int nparts = grid_1d((img.count() + 3) / 4);        // 16-byte loads: a thread takes four values per step
                double* sums = (double*)L.pool.get(2 * (size_t)nparts * sizeof(double));
                float* stt = (float*)L.pool.get(2 * sizeof(float));
                hipLaunchKernelGGL(moments_kernel, dim3(nparts), dim3(256), 0, stream, img.p, img.count(), sums);
                hipLaunchKernelGGL(moments_finish_kernel, dim3(1), dim3(256), 0, stream, sums, nparts, img.count(), stt);
                st = stt;
            }
            stats0.push_back(st);
        }
        // image pyramid.  With mvn the pyramid is built from the standardised image; avg-pooling commutes with
        // the affine map, so scales >= 1 standardise on load with the page's statistics.
        std::vector<TL> pyr{level0};
        for (int s = 1; s < nsc; ++s) pyr.push_back(run_pool(m, pyr.back(), POOL_AVG_C1));
        TL all;
        std::vector<std::string> names;
        std::vector<const float*> stats;
        for (int b = 0; b < B; ++b)
            for (int s = 0; s < nsc; ++s) {
                all.push_back(pyr[s][b]);
                names.push_back((page0 + b ? "p" + std::to_string(page0 + b) + "/" : std::string()) + "scale_" + std::to_string(s));
                stats.push_back(stats0[b]);
            }
        if (!cfg.mvn) stats.clear();

        TL att;
        bool forked = false;
        if (cfg.use_attention) {
            // the attention CNN is a chain of small launches that cannot fill the chip: run it on a side stream next
            // to the feature branch (fork after the pyramid, join before the combine).  Per-launch profiling keeps
            // everything on one stream so that kernel times are not inflated by the overlap.
            if ((!m->profiling || m->prof_in_situ) && L.side) {
                ASEP_HIP_CHECK(hipEventRecord(L.ev_fork, stream));
                ASEP_HIP_CHECK(hipStreamWaitEvent(L.side, L.ev_fork, 0));
                m->stream = L.side;
                forked = true;
            }
            att = att_cnn(m, all, stats);
            if (forked) {
                ASEP_HIP_CHECK(hipEventRecord(L.ev_join, L.side));
                m->stream = stream;
            }
            for (size_t i = 0; i < att.size(); ++i) {
                const int pg = page0 + (int)i / nsc;
                m->endpoints[(pg ? "p" + std::to_string(pg) + "/" : std::string()) + "att_" + std::to_string(i % nsc)] = att[i];
            }
        }
        TL feat = det_cnn(m, all, names, stats);
        if (forked) ASEP_HIP_CHECK(hipStreamWaitEvent(stream, L.ev_join, 0));
        TL fsum;
        if (nsc > 1) {
            TL coarse;
            for (int b = 0; b < B; ++b)
                for (int s = 1; s < nsc; ++s) coarse.push_back(feat[b * nsc + s]);
            fsum = m->bf16 ? run_chansum_bf(m, coarse) : run_pool(m, coarse, POOL_CHANSUM);
        }
        for (int b = 0; b < B; ++b) {
            const int H = Hs[b], W = Ws[b];
            CombineArgs ca{};
            ca.nsc = nsc; ca.H = H; ca.W = W;
            ca.f0 = feat[b * nsc].p;
            int up = 8;
            for (int s = 0; s < nsc && cfg.use_attention; ++s) {
                const Tensor& a = att[b * nsc + s];
                if (cdiv(H, up) != a.H || cdiv(W, up) != a.W) { set_error("internal: attention map shape"); return ASEP_ERR_ARG; }
                ca.att[s] = a.p; ca.ah[s] = a.H; ca.aw[s] = a.W; ca.aup[s] = up; ca.ash[s] = 3 + s;
                ca.aph[s] = (a.H * up - H) / 2; ca.apw[s] = (a.W * up - W) / 2;
                up *= 2;
            }
            up = 1;
            for (int s = 1; s < nsc; ++s) {
                const Tensor& f = feat[b * nsc + s];
                up *= 2;
                ca.fsum[s] = fsum[b * (nsc - 1) + (s - 1)].p; ca.fh[s] = f.H; ca.fw[s] = f.W; ca.fup[s] = up; ca.fsh[s] = s;
                ca.fph[s] = (f.H * up - H) / 2; ca.fpw[s] = (f.W * up - W) / 2;
            }
            ca.wl = m->d_logit_w; ca.bl = m->d_logit_b;
            ca.wd = (cfg.apply_softmax && cfg.n_classes == 2) ? m->d_logit_wd : nullptr;
            ca.out = d_outs[b]; ca.out_u8 = d_u8s ? d_u8s[b] : nullptr; ca.out_mask = d_masks ? d_masks[b] : nullptr;
            ca.thr255 = (double)threshold * 255.0;
            ca.softmax = cfg.apply_softmax;
            ca.tiles_x = cdiv(W, COMBINE_TW);
            const int ctiles = ca.tiles_x * cdiv(H, 16);
            int cunits = ctiles;
            ca.xm = oneshot_map(m, ctiles, &cunits);
            dim3 grid(cunits);
            // number of scales as a template constant (1 = no attention, 3 = the default ARU-Net) with 32-bit offsets, for tensors
            // below 4 GB; anything else takes the run-time form
            const bool small = (size_t)H * W * std::max(cfg.feat_root, cfg.n_classes) * sizeof(float) < ((size_t)1 << 32);
            const int nsct = small && (nsc == 1 || nsc == 3) ? nsc : 0;
            ProfScope ps(m, "combine_kernel" + targs({ti(cfg.feat_root), ti(cfg.n_classes), tb(m->bf16), ti(nsct)}), 2.0 * H * W * 16.0 * cfg.feat_root * cfg.n_classes);
            // scale-0 feature map + per further scale its channel sum + the attention maps in; probabilities (+ uint8 / threshold images) out
            ps.bytes = tbytes(feat[b * nsc]) + (double)H * W * cfg.n_classes * (4.0 + (ca.out_u8 ? 1.0 : 0.0) + (ca.out_mask ? 1.0 : 0.0));
            if (ca.wd) ps.xflops = 2.0 * H * W * 16.0 * cfg.feat_root;   // two classes behind a soft-max: the logits conv runs on the difference filter (half the products)
            for (int s2 = 1; s2 < nsc; ++s2) ps.bytes += tbytes(fsum[b * (nsc - 1) + (s2 - 1)]);
            for (int s2 = 0; s2 < nsc && cfg.use_attention; ++s2) ps.bytes += tbytes(att[b * nsc + s2]);
#define ASEP_COMB_N(FR, NC, BF)                                                                    \
        if (nsct == 3) hipLaunchKernelGGL((combine_kernel<FR, NC, BF, 3>), grid, dim3(256), 0, stream, ca);      \
        else if (nsct == 1) hipLaunchKernelGGL((combine_kernel<FR, NC, BF, 1>), grid, dim3(256), 0, stream, ca); \
        else hipLaunchKernelGGL((combine_kernel<FR, NC, BF, 0>), grid, dim3(256), 0, stream, ca);
#define ASEP_COMB(FR, NC)                                                                          \
    if (cfg.feat_root == FR && cfg.n_classes == NC && !m->bf16) {                                  \
        ASEP_COMB_N(FR, NC, false)                                                                 \
    } else
#define ASEP_COMBB(NC)                                                                             \
    if (cfg.feat_root == 8 && cfg.n_classes == NC && m->bf16) {                                    \
        ASEP_COMB_N(8, NC, true)                                                                   \
    } else
            ASEP_COMBB(1) ASEP_COMBB(2) ASEP_COMBB(3) ASEP_COMBB(4)
            ASEP_COMB(8, 1) ASEP_COMB(8, 2) ASEP_COMB(8, 3) ASEP_COMB(8, 4) ASEP_COMB(16, 2)
            { set_error("combine: feat_root=%d n_classes=%d not instantiated", cfg.feat_root, cfg.n_classes); return ASEP_ERR_UNSUPPORTED; }
#undef ASEP_COMB
#undef ASEP_COMBB
#undef ASEP_COMB_N
        }
        ASEP_HIP_CHECK(hipGetLastError());
    } catch (const HipError&) {
        return ASEP_ERR_HIP;
    } catch (const ArgError&) {
        return ASEP_ERR_ARG;
    }
    return ASEP_OK;
}

// 2*MAC of every conv / deconv of one forward (SURVEY.md section 8d formula)
double flops_impl(const asep_aru_cfg& cfg, int H, int W) {
    auto det = [&](int h, int w) {
        double mac = 0;
        int last = cfg.channels;
        std::vector<std::pair<int, int>> dims;
        int hh = h, ww = w;
        for (int l = 0; l < cfg.scale_space_num; ++l) {
            const int f = cfg.feat_root << l;
            dims.push_back({hh, ww});
            mac += (double)hh * ww * (9.0 * last * f + cfg.res_depth * 9.0 * f * f);
            last = f;
            if (l < cfg.scale_space_num - 1) { hh = cdiv(hh, 2); ww = cdiv(ww, 2); }
        }
        for (int l = cfg.scale_space_num - 2; l >= 0; --l) {
            const int f = cfg.feat_root << l;
            const auto& di = dims[l + 1];
            const auto& d = dims[l];
            mac += (double)di.first * di.second * 9.0 * last * f;                       // deconv, input resolution
            mac += (double)d.first * d.second * (9.0 * 2 * f * f + cfg.res_depth * 9.0 * f * f);
            last = f;
        }
        return mac;
    };
    auto att = [&](int h, int w) {
        double mac = 0;
        int chans[5] = {cfg.channels, 12, 16, 32, 1};
        int hh = h, ww = w;
        for (int i = 0; i < 4; ++i) {
            mac += (double)hh * ww * 16.0 * chans[i] * chans[i + 1];
            if (i < 3) { hh = cdiv(hh, 2); ww = cdiv(ww, 2); }
        }
        return mac;
    };
    double mac = 0;
    const int nsc = cfg.use_attention ? cfg.num_scales_att : 1;
    int h = H, w = W;
    for (int s = 0; s < nsc; ++s) {
        mac += det(h, w);
        if (cfg.use_attention) mac += att(h, w);
        h = cdiv(h, 2); w = cdiv(w, 2);
    }
    mac += (double)H * W * 16.0 * cfg.feat_root * cfg.n_classes;
    return 2.0 * mac;
}

}  // namespace

namespace asep {

int aru_endpoint_dev(asep_aru* m, const char* name, const float** d_ptr, int dims[3], int* is_bf16) {
    if (!m || !name || !d_ptr) { set_error("aru_endpoint_dev: bad argument"); return ASEP_ERR_ARG; }
    auto it = m->endpoints.find(name);
    if (it == m->endpoints.end()) { set_error("backbone has no end point '%s' (run a forward first)", name); return ASEP_ERR_ARG; }
    if (it->second.bf && !is_bf16) { set_error("end point '%s' is bf16 (compute_dtype 1) and the caller reads fp32 maps", name); return ASEP_ERR_UNSUPPORTED; }
    if (is_bf16) *is_bf16 = it->second.bf ? 1 : 0;
    *d_ptr = it->second.p;                                  // bf16 maps: the same address, 2 bytes per value
    if (dims) { dims[0] = it->second.H; dims[1] = it->second.W; dims[2] = it->second.C; }
    return ASEP_OK;
}

// "scale_<s>_unet_{down,up}_<l>_{conv,deconv}" -> feat_root * 2^l (ARU_v1.py:208-292)
int aru_endpoint_channels(const asep_aru* m, const char* name) {
    if (!m || !name) return -1;
    int s = 0, l = 0;
    char kind[16] = {0}, what[16] = {0};
    if (sscanf(name, "scale_%d_unet_%15[a-z]_%d_%15[a-z]", &s, kind, &l, what) != 4) return -1;
    if (l < 0 || l >= m->cfg.scale_space_num) return -1;
    return m->cfg.feat_root << l;
}

int aru_num_classes(const asep_aru* m) { return m ? m->cfg.n_classes : -1; }

}  // namespace asep

extern "C" {

asep_aru* asep_aru_load(const void* weight_blob, size_t nbytes, const asep_aru_cfg* cfg) {
    ASEP_GUARD_BEGIN
    if (!cfg || !weight_blob) { set_error("asep_aru_load: null argument"); return nullptr; }
    if (cfg->struct_size != (int32_t)sizeof(asep_aru_cfg)) {
        set_error("asep_aru_load: cfg.struct_size is %d, this library's asep_aru_cfg has %zu bytes (ABI version %d): the binding was "
                  "written against another include/asep_hip.h", cfg->struct_size, sizeof(asep_aru_cfg), ASEP_ABI_VERSION);
        return nullptr;
    }
    if (cfg->channels != 1) { set_error("asep_aru_load: only 1-channel input is supported (ARU_v1.py:115)"); return nullptr; }
    if (cfg->compute_dtype < 0 || cfg->compute_dtype > 2) { set_error("asep_aru_load: compute_dtype %d unknown (0 = fp32, 1 = bf16 MFMA, 2 = fp32 with split bf16 products)", cfg->compute_dtype); return nullptr; }
    if (cfg->scale_space_num < 1 || cfg->res_depth < 1) { set_error("asep_aru_load: bad cfg"); return nullptr; }
    if (cfg->activation < 0 || cfg->activation > 2) { set_error("asep_aru_load: activation %d unknown (0 = relu, 1 = elu, 2 = leaky)", cfg->activation); return nullptr; }
    const bool variant = cfg->activation != 0 || cfg->plain_u != 0;
    if (cfg->plain_u && cfg->use_attention) { set_error("asep_aru_load: graph 'U' has no attention branch (ARU_v1.py:92-97)"); return nullptr; }
    std::map<std::string, HostTensor> blob;
    if (!parse_blob(weight_blob, nbytes, blob)) return nullptr;
    warn_ignored_switches();
    std::unique_ptr<asep_aru> m(new asep_aru());
    m->cfg = *cfg;
    m->bf16 = cfg->compute_dtype == 1;
    m->split = cfg->compute_dtype == 2;
    // Engine switches (environment, read at load): each selects an INDEPENDENT implementation of the same arithmetic that the test suite
    // compares with the default one (fused against unfused, vector-ALU against MFMA form), or a schedule knob of the bench; the switches of
    // experiments that lost left the tree in round 5 (table: DESIGN.md section 4.5).
    if (const char* e = getenv("ASEP_FUSE_POOL")) m->fuse_pool = atoi(e) != 0;
    if (const char* e = getenv("ASEP_FUSE_ACT")) m->fuse_act = atoi(e) != 0;
    if (const char* e = getenv("ASEP_C12")) m->use_c12 = atoi(e) != 0;
    if (const char* e = getenv("ASEP_FUSED8")) m->use_fused8 = atoi(e) != 0;
    m->fused8_wanted = m->use_fused8;
    if (variant) m->use_fused8 = false;                      // the fused level-0 blocks / attention head are ReLU residual kernels
    if (const char* e = getenv("ASEP_R8_VALU")) m->r8_valu = atoi(e) != 0;
    m->fused8_var = variant && !cfg->plain_u && cfg->activation != 0 && m->fused8_wanted && m->r8_valu && m->fuse_act && !m->bf16;
    if (const char* e = getenv("ASEP_XCD_SCHED")) m->use_xcd_sched = atoi(e) != 0;
    if (const char* e = getenv("ASEP_BF_RES32")) m->use_res32 = atoi(e) != 0;
    if (const char* e = getenv("ASEP_BF_CONVR")) m->use_convr = atoi(e) != 0;
    if (const char* e = getenv("ASEP_SPLIT_DECONV")) m->use_deconvs = atoi(e) != 0;
    if (const char* e = getenv("ASEP_BF_WALK")) { m->walk_mode = atoi(e); m->use_walk = m->walk_mode != 0; }
    if (const char* e = getenv("ASEP_LANES")) { m->num_lanes = std::max(1, std::min(4, atoi(e))); m->lanes_forced = true; }
    for (int l = 0; l < m->num_lanes; ++l) {
        std::unique_ptr<asep_aru::Lane> L(new asep_aru::Lane());
        bool ok = hipStreamCreateWithFlags(&L->side, hipStreamNonBlocking) == hipSuccess &&
                  hipStreamCreateWithFlags(&L->bside, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreateWithFlags(&L->ev_fork, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&L->ev_bfork, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&L->ev_bjoin, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&L->ev_join, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&L->ev_begin, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&L->ev_done, hipEventDisableTiming) == hipSuccess;
        if (ok && l > 0) { ok = hipStreamCreateWithFlags(&L->s, hipStreamNonBlocking) == hipSuccess; L->own_stream = ok; }
        if (!ok) { set_error("asep_aru_load: cannot create streams/events for lane %d", l); return nullptr; }
        m->lanes.push_back(std::move(L));
    }
    m->cur = m->lanes[0].get();
    int rc = ASEP_OK;
    const int n = cfg->scale_space_num;
    if (cfg->use_attention) {
        rc = pack_direct(m.get(), blob, "aru_net/attMapG/attPart/conv1", &m->att_first);
        if (!rc && m->att_first.k == 4 && m->att_first.cout == 12) {
            const HostTensor& w = blob.find("aru_net/attMapG/attPart/conv1/weights")->second;   // [4][4][1][12]
            std::vector<float> pk(64 * 4, 0.f);
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r) {
                    const int co = lane & 15, ky = lane >> 4, kx = r;
                    if (co < 12) pk[lane * 4 + r] = w.data[(size_t)(ky * 4 + kx) * 12 + co];
                }
            rc = upload(pk, &m->d_att_head);
            if (!rc) m->owned.push_back(m->d_att_head);
            if (!rc && m->bf16) {                            // att_headb_kernel: k = 8 kk + 4 r + c <-> tap (2 kk + r, c) for kk < 2, zero rows / slots elsewhere
                std::vector<bf16_t> pb(64 * 8, 0);
                for (int lane = 0; lane < 64; ++lane) {
                    const int co = lane & 15, kk = lane >> 4;
                    for (int i = 0; i < 8 && kk < 2 && co < 12; ++i) pb[lane * 8 + i] = f2bf(w.data[(size_t)((2 * kk + (i >> 2)) * 4 + (i & 3)) * 12 + co]);
                }
                rc = upload_bf(pb, &m->d_att_headb);
                if (!rc) m->owned.push_back(m->d_att_headb);
            }
        }
        for (int i = 2; i <= 4 && !rc; ++i)
            rc = pack_conv(m.get(), blob, "aru_net/attMapG/attPart/conv" + std::to_string(i), "biases", false);
    }
    if (!rc) rc = pack_direct(m.get(), blob, "aru_net/featMapG/unet_down_0/conv1", &m->det_first);
    for (int l = 0; l < n && !rc; ++l) {
        const std::string s = "aru_net/featMapG/unet_down_" + std::to_string(l);
        if (l > 0) rc = pack_conv(m.get(), blob, s + "/conv1", "biases", false);
        if (cfg->plain_u) { if (!rc) rc = pack_conv(m.get(), blob, s + "/conv2", "biases", false); continue; }
        for (int r = 0; r < cfg->res_depth && !rc; ++r)
            rc = pack_conv(m.get(), blob, s + "/convR_" + std::to_string(r), "biases", false);
    }
    for (int l = n - 2; l >= 0 && !rc; --l) {
        const std::string s = "aru_net/featMapG/unet_up_" + std::to_string(l);
        rc = pack_conv(m.get(), blob, s + "/deconv", "bias", true);
        if (!rc) rc = pack_conv(m.get(), blob, s + "/conv1", "biases", false);
        if (cfg->plain_u) { if (!rc) rc = pack_conv(m.get(), blob, s + "/conv2", "biases", false); continue; }
        for (int r = 0; r < cfg->res_depth && !rc; ++r)
            rc = pack_conv(m.get(), blob, s + "/convR_" + std::to_string(r), "biases", false);
    }
    if (!rc && (!variant || m->fused8_var) && !m->bf16 && cfg->feat_root == 8 && cfg->res_depth == 3 && m->det_first.k == 3) rc = pack_res8(m.get(), blob);
    // (bf16 path, elu / leaky / 'U' graphs -- round 5: layer by layer on convb_kernel / deconvb_kernel with the activation in their general
    //  epilogues; the fused blocks below bake the ReLU into packed-bf16 maxima and serve the ReLU residual graphs)
    // (round 6: the elu / leaky RESIDUAL graphs take the GENERAL fused forms of the 8- and 16-channel levels -- res8b_tile / resb_tail_tile apply the
    //  activation to fp32 values and take float maxima in their pools, a template parameter serves them; the 32-channel tail the same way; the lean forms and
    //  the walkers stay the ReLU graphs')
    m->fused_act = (m->bf16 && variant && !cfg->plain_u && cfg->activation != 0 && m->fused8_wanted && m->fuse_act) ? cfg->activation : 0;
    const bool fusedb = m->bf16 && (!variant || m->fused_act);
    if (!rc && fusedb && cfg->res_depth == 3 && cfg->feat_root == 8) rc = pack_res8b(m.get(), blob);
    if (!rc && fusedb && cfg->res_depth == 3)
        for (int l = 0; l < n && !rc; ++l) {
            const int f = cfg->feat_root << l;
            if (f != 8 && f != 16 && !(f == 32 && m->use_res32)) continue;
            rc = pack_resb(m.get(), blob, "aru_net/featMapG/unet_down_" + std::to_string(l), f);
            if (!rc && l < n - 1) rc = pack_resb(m.get(), blob, "aru_net/featMapG/unet_up_" + std::to_string(l), f);
        }
    if (rc) return nullptr;
    {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            m->num_cus = prop.multiProcessorCount;
    }
    auto lw = blob.find("aru_net/logit/class/weights");
    auto lb = blob.find("aru_net/logit/class/biases");
    if (lw == blob.end() || lb == blob.end()) { set_error("weights: missing aru_net/logit/class"); return nullptr; }
    const auto& d = lw->second.dims;
    if (d.size() != 4 || d[0] != 4 || d[1] != 4 || d[2] != cfg->feat_root || d[3] != cfg->n_classes) {
        set_error("weights: aru_net/logit/class/weights must be [4,4,%d,%d]", cfg->feat_root, cfg->n_classes);
        return nullptr;
    }
    if (upload(lw->second.data, &m->d_logit_w) || upload(lb->second.data, &m->d_logit_b)) return nullptr;
    m->owned.push_back(m->d_logit_w);
    m->owned.push_back(m->d_logit_b);
    if (cfg->n_classes == 2) {
        const std::vector<float>& w = lw->second.data;       // [4][4][feat_root][2]
        std::vector<float> wd((size_t)16 * cfg->feat_root + 1);
        for (size_t i = 0; i < (size_t)16 * cfg->feat_root; ++i) wd[i] = w[2 * i + 1] - w[2 * i];
        wd.back() = lb->second.data[1] - lb->second.data[0];
        if (upload(wd, &m->d_logit_wd)) return nullptr;
        m->owned.push_back(m->d_logit_wd);
    }
    if (hipMalloc((void**)&m->d_stats, 2 * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&m->d_sums, 2 * sizeof(double)) != hipSuccess) {
        set_error("asep_aru_load: hipMalloc failed");
        return nullptr;
    }
    m->owned.push_back(m->d_stats);
    m->owned.push_back(m->d_sums);
    if (hipMalloc((void**)&m->d_zero_trash, 8192) != hipSuccess || hipMemset(m->d_zero_trash, 0, 8192) != hipSuccess) {
        set_error("asep_aru_load: hipMalloc failed");
        return nullptr;
    }
    m->owned.push_back(m->d_zero_trash);
    return m.release();
    ASEP_GUARD_END_PTR
}

void asep_aru_free(asep_aru* m) { delete m; }

// splits the pages over the lanes (lane 0 = the caller's stream), forks / joins with events
static int forward_lanes(asep_aru* m, int n_pages, const float* const* d_imgs, const int32_t* Hs, const int32_t* Ws, float* const* d_outs,
                         uint8_t* const* d_u8, uint8_t* const* d_mask, float threshold, hipStream_t stream) {
    m->endpoints.clear();
    // (per-launch profiling runs on one lane in every mode: the isolated mode brackets one kernel at a time, the in-situ mode keeps the side
    //  stream and whatever the caller runs beside the net, but not a second launch of the SAME kernel beside the bracketed one)
    // a lane takes at least four pages (= one full 12-problem launch per layer with three scales): fewer pages per launch cost the deep levels
    // more than a second lane returns; ASEP_LANES (lanes_forced) keeps the plain split for the measurement scripts and tests
    const int by_pages = m->lanes_forced ? n_pages : n_pages / 4;
    const int nl = (m->profiling || n_pages < 2) ? 1 : std::max(1, std::min<int>((int)m->lanes.size(), by_pages));
    asep_aru::Lane& L0 = *m->lanes[0];
    L0.s = stream;
    // A lane's arena keeps the buffers of the largest call it has served.  When the number of lanes of a call changes (16 pages on two lanes, then
    // the same 16 pages on one lane while launch times are recorded) the arenas would add up to 1.5 x the pages in flight -- 144 GB instead of 96 for
    // 16 fp32 pages, and a second process of the same size beside it no longer fits into 288 GB (round 6: the children of the default bench line
    // ran out of memory).  A change of the lane count therefore releases every arena first (hipFree waits for the device: nothing is in use).
    if (nl != m->last_nl && n_pages > 1) {
        if (m->last_nl != 0)
            for (auto& Lp : m->lanes) Lp->pool.release();
        m->last_nl = nl;
    }
    if (nl == 1) return forward_impl(m, L0, 0, n_pages, d_imgs, Hs, Ws, d_outs, d_u8, d_mask, threshold);
    ASEP_HIP_CHECK(hipEventRecord(L0.ev_begin, stream));
    int page0 = 0;
    for (int l = 0; l < nl; ++l) {
        asep_aru::Lane& L = *m->lanes[l];
        const int cnt = n_pages / nl + (l < n_pages % nl ? 1 : 0);
        if (l > 0) ASEP_HIP_CHECK(hipStreamWaitEvent(L.s, L0.ev_begin, 0));
        int rc = forward_impl(m, L, page0, cnt, d_imgs + page0, Hs + page0, Ws + page0, d_outs + page0, d_u8 ? d_u8 + page0 : nullptr,
                              d_mask ? d_mask + page0 : nullptr, threshold);
        if (rc) return rc;
        if (l > 0) {
            ASEP_HIP_CHECK(hipEventRecord(L.ev_done, L.s));
            ASEP_HIP_CHECK(hipStreamWaitEvent(stream, L.ev_done, 0));
        }
        page0 += cnt;
    }
    m->stream = stream;
    return ASEP_OK;
}

int asep_aru_forward_dev(asep_aru* m, const float* d_img, int H, int W, float* d_out, uint8_t* d_out_u8,
                         uint8_t* d_out_mask, float threshold, void* stream) {
    ASEP_GUARD_BEGIN
    if (!m || !d_img || !d_out || H < 1 || W < 1) { set_error("asep_aru_forward_dev: bad argument"); return ASEP_ERR_ARG; }
    const int32_t h1 = H, w1 = W;
    return forward_lanes(m, 1, &d_img, &h1, &w1, &d_out, d_out_u8 ? &d_out_u8 : nullptr, d_out_mask ? &d_out_mask : nullptr,
                         threshold, (hipStream_t)stream);
    ASEP_GUARD_END
}

int asep_aru_forward_batch_dev(asep_aru* m, int n_pages, const float* const* d_imgs, int H, int W, float* const* d_outs,
                               uint8_t* const* d_out_u8, uint8_t* const* d_out_mask, float threshold, void* stream) {
    ASEP_GUARD_BEGIN
    if (!m || !d_imgs || !d_outs || n_pages < 1 || H < 1 || W < 1) { set_error("asep_aru_forward_batch_dev: bad argument"); return ASEP_ERR_ARG; }
    for (int b = 0; b < n_pages; ++b)
        if (!d_imgs[b] || !d_outs[b] || (d_out_u8 && !d_out_u8[b]) || (d_out_mask && !d_out_mask[b])) {
            set_error("asep_aru_forward_batch_dev: null page pointer at %d", b);
            return ASEP_ERR_ARG;
        }
    const std::vector<int32_t> hs(n_pages, H), ws(n_pages, W);
    return forward_lanes(m, n_pages, d_imgs, hs.data(), ws.data(), d_outs, d_out_u8, d_out_mask, threshold, (hipStream_t)stream);
    ASEP_GUARD_END
}

// ABI 6: the same for pages of DIFFERENT sizes (the reference runs page by page on whatever size --fixed_height / --scaling_factor leave:
// ARU_v1.py:64, run_net_post_processing.py:61-82; real scans differ in width).  Every grouped launch already carries per-problem dims (the
// three scales of a page), so pages of any sizes share the launches of a layer.
int asep_aru_forward_batch_dev2(asep_aru* m, int n_pages, const float* const* d_imgs, const int32_t* H, const int32_t* W, float* const* d_outs,
                                uint8_t* const* d_out_u8, uint8_t* const* d_out_mask, float threshold, void* stream) {
    ASEP_GUARD_BEGIN
    if (!m || !d_imgs || !d_outs || !H || !W || n_pages < 1) { set_error("asep_aru_forward_batch_dev2: bad argument"); return ASEP_ERR_ARG; }
    for (int b = 0; b < n_pages; ++b)
        if (!d_imgs[b] || !d_outs[b] || (d_out_u8 && !d_out_u8[b]) || (d_out_mask && !d_out_mask[b]) || H[b] < 1 || W[b] < 1) {
            set_error("asep_aru_forward_batch_dev2: null page pointer or empty page at %d", b);
            return ASEP_ERR_ARG;
        }
    return forward_lanes(m, n_pages, d_imgs, H, W, d_outs, d_out_u8, d_out_mask, threshold, (hipStream_t)stream);
    ASEP_GUARD_END
}

int asep_aru_forward(asep_aru* m, const float* img_hw, int H, int W, float* out_hwc, uint8_t* out_u8,
                     uint8_t* out_mask, float threshold) {
    ASEP_GUARD_BEGIN
    if (!m || !img_hw || !out_hwc || H < 1 || W < 1) { set_error("asep_aru_forward: bad argument"); return ASEP_ERR_ARG; }
    const size_t npix = (size_t)H * W, nout = npix * m->cfg.n_classes;
    // staging buffers live in the handle and only grow (a page-sized hipMalloc / hipFree pair per call costs
    // milliseconds, comparable to the net itself on small inputs)
    float *d_img = nullptr, *d_out = nullptr;
    uint8_t *d_u8 = nullptr, *d_mask = nullptr;
    try {
        m->host_stage.begin();
        d_img = (float*)m->host_stage.get(npix * sizeof(float));
        d_out = (float*)m->host_stage.get(nout * sizeof(float));
        d_u8 = (uint8_t*)m->host_stage.get(nout);
        d_mask = (uint8_t*)m->host_stage.get(nout);
    } catch (const HipError&) {
        return ASEP_ERR_HIP;
    }
    // All transfers and the forward are queued on one stream, one synchronisation at the end.  Page-locked caller buffers
    // (asep_host_alloc / asep_host_register) are device-visible: the net's last kernel then writes its outputs straight
    // into them over the link (no staging buffer, no separate download: 108 MB of probabilities leave while the kernel
    // runs); the input is still copied once (it is read by several kernels).  Pageable buffers take staged copies.
    if (!m->host_stream) ASEP_HIP_CHECK(hipStreamCreateWithFlags(&m->host_stream, hipStreamNonBlocking));
    hipStream_t hs = m->host_stream;
    auto device_view = [](void* host) -> void* {
        if (!host) return nullptr;
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, host) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return at.type == hipMemoryTypeHost ? at.devicePointer : nullptr;
    };
    float* v_out = (float*)device_view(out_hwc);
    uint8_t* v_u8 = (uint8_t*)device_view(out_u8);
    uint8_t* v_mask = (uint8_t*)device_view(out_mask);
    ASEP_HIP_CHECK(hipMemcpyAsync(d_img, img_hw, npix * sizeof(float), hipMemcpyHostToDevice, hs));
    const int rc = asep_aru_forward_dev(m, d_img, H, W, v_out ? v_out : d_out, out_u8 ? (v_u8 ? v_u8 : d_u8) : nullptr,
                                        out_mask ? (v_mask ? v_mask : d_mask) : nullptr, threshold, hs);
    if (rc) return rc;
    if (out_u8 && !v_u8) ASEP_HIP_CHECK(hipMemcpyAsync(out_u8, d_u8, nout, hipMemcpyDeviceToHost, hs));
    if (out_mask && !v_mask) ASEP_HIP_CHECK(hipMemcpyAsync(out_mask, d_mask, nout, hipMemcpyDeviceToHost, hs));
    if (!v_out) ASEP_HIP_CHECK(hipMemcpyAsync(out_hwc, d_out, nout * sizeof(float), hipMemcpyDeviceToHost, hs));
    ASEP_HIP_CHECK(hipStreamSynchronize(hs));
    return ASEP_OK;
    ASEP_GUARD_END
}

long asep_aru_get_endpoint(asep_aru* m, const char* name, float* out, size_t max_floats, int32_t dims[3]) {
    ASEP_GUARD_BEGIN
    if (!m || !name) { set_error("asep_aru_get_endpoint: bad argument"); return ASEP_ERR_ARG; }
    auto it = m->endpoints.find(name);
    if (it == m->endpoints.end()) { set_error("asep_aru_get_endpoint: unknown end point '%s'", name); return ASEP_ERR_ARG; }
    const Tensor& t = it->second;
    if (dims) { dims[0] = t.H; dims[1] = t.W; dims[2] = t.C; }
    if (!out) return (long)t.count();
    if (max_floats < t.count()) { set_error("asep_aru_get_endpoint: buffer too small"); return ASEP_ERR_ARG; }
    ASEP_HIP_CHECK(hipStreamSynchronize(m->stream));
    if (t.bf) {                                             // bf16 tensor: widened on the host
        std::vector<bf16_t> tmp(t.count());
        ASEP_HIP_CHECK(hipMemcpy(tmp.data(), t.p, t.count() * sizeof(bf16_t), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < tmp.size(); ++i) {
            const uint32_t u = (uint32_t)tmp[i] << 16;
            memcpy(out + i, &u, 4);
        }
        return (long)t.count();
    }
    ASEP_HIP_CHECK(hipMemcpy(out, t.p, t.count() * sizeof(float), hipMemcpyDeviceToHost));
    return (long)t.count();
    ASEP_GUARD_END
}

// ABI 6: give the handle's device arenas back (they are rebuilt by the next forward call): a process that holds a model and starts another
// GPU process of its size beside it calls this first
int asep_aru_trim(asep_aru* m) {
    ASEP_GUARD_BEGIN
    if (!m) { set_error("asep_aru_trim: null handle"); return ASEP_ERR_ARG; }
    ASEP_HIP_CHECK(hipDeviceSynchronize());
    m->endpoints.clear();
    for (auto& Lp : m->lanes) Lp->pool.release();
    m->host_stage.release();
    m->last_nl = 0;
    return ASEP_OK;
    ASEP_GUARD_END
}

int asep_aru_profile(asep_aru* m, int enable) {
    ASEP_GUARD_BEGIN
    if (!m) { set_error("asep_aru_profile: null handle"); return ASEP_ERR_ARG; }
    m->profiling = enable != 0;
    m->prof_detail = enable == 2;
    m->prof_in_situ = enable == 3;
    if (enable) { m->prof_recs.clear(); m->ev_next = 0; m->prof_names.clear(); }
    return ASEP_OK;
    ASEP_GUARD_END
}

long asep_aru_profile_report(asep_aru* m, char* buf, size_t buflen) {
    ASEP_GUARD_BEGIN
    if (!m || !buf || buflen < 2) { set_error("asep_aru_profile_report: bad argument"); return ASEP_ERR_ARG; }
    ASEP_HIP_CHECK(hipStreamSynchronize(m->stream));
    const size_t nk = m->prof_names.size();
    std::vector<double> ms(nk, 0.0), fl(nk, 0.0), by(nk, 0.0), xf(nk, 0.0);
    std::vector<long> calls(nk, 0);
    for (const auto& r : m->prof_recs) {
        float t = 0.f;
        ASEP_HIP_CHECK(hipEventElapsedTime(&t, r.a, r.b));
        ms[r.kid] += t; fl[r.kid] += r.flops; by[r.kid] += r.bytes; xf[r.kid] += r.xflops; calls[r.kid] += 1;
    }
    std::string js = "[";
    for (size_t i = 0; i < nk; ++i) {
        if (!calls[i]) continue;
        char line[512];
        snprintf(line, sizeof(line), "%s{\"kernel\":\"%s\",\"calls\":%ld,\"total_ms\":%.6f,\"flops\":%.6e,\"bytes\":%.6e,\"executed_flops\":%.6e}",
                 js.size() > 1 ? "," : "", m->prof_names[i].c_str(), calls[i], ms[i], fl[i], by[i], xf[i]);
        js += line;
    }
    js += "]";
    if (js.size() + 1 > buflen) { set_error("asep_aru_profile_report: buffer too small (%zu needed)", js.size() + 1); return ASEP_ERR_ARG; }
    memcpy(buf, js.c_str(), js.size() + 1);
    return (long)js.size();
    ASEP_GUARD_END
}

double asep_aru_flops(const asep_aru* m, int H, int W) {
    if (!m) return 0.0;
    return flops_impl(m->cfg, H, W);
}

}  // extern "C"

#if defined(R8F_TRACE)
// debug builds only (see R8F_MARK in bf16_kernels.h); not part of include/asep_hip.h
extern "C" int asep_debug_r8f_trace(int up, unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(asep::g_r8f_trace), sizeof(unsigned long long) * n, (size_t)(up ? 1 : 0) * 4096 * 12 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}
#endif

#if defined(CVR_TRACE)
extern "C" int asep_debug_cvr_trace(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(asep::g_cvr_trace), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost);
}
#endif
#if defined(CVB_TRACE)
// debug builds only (see CVB_MARK in bf16_kernels.h); not part of include/asep_hip.h
extern "C" int asep_debug_cvb_trace(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(asep::g_cvb_trace), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost);
}
#endif
