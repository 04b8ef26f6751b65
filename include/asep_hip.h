/* asep_hip.h -- C ABI of libasep_hip.so: the MI355X (gfx950) replacement for the two
 * TensorFlow session calls on the article-separation hot path.
 *
 * The reference has no FFI of its own; its seam is two Python call sites that hand numpy arrays
 * to tf.Session.run.  Each entry point below names the reference call it replaces:
 *
 *   ARU-Net   article_separation/image_segmentation/net_post_processing/net_post_processing_helper.py
 *             :36-53  load_graph(path_to_pb)                    -> asep_aru_load
 *             :56-72  get_net_output(image, pb_graph, gpu)      -> asep_aru_forward[_dev]
 *             :75-78  apply_threshold + separator_net_post_processor.py:147 (uint8 truncation)
 *                                                               -> fused u8 outputs of asep_aru_forward
 *   GNN       article_separation/gnn/io.py:12-25 load_graph     -> asep_gnn_load
 *             article_separation/gnn/run_gnn_clustering.py:259-269 sess.run(
 *                 'output_belong_to_same_instance:0', feed_dict) -> asep_gnn_forward[_dev]
 *             article_separation/gnn/model/graph_util/misc.py:7-151
 *                 check_and_correct_interacting_nodes           -> asep_gnn_correct_edges
 *
 * Conventions: plain pointers and sizes; caller-owned buffers; 0 = success, negative = error
 * (text via asep_last_error()); no exceptions cross the ABI; one handle per (process, device);
 * a handle is re-entrant but must not be used concurrently from two threads.
 * Pointers named d_* are device (HBM) pointers; all others are host pointers.
 * `stream` is a hipStream_t passed as void* (NULL = the default stream).
 */
#ifndef ASEP_HIP_H
#define ASEP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASEP_OK 0
#define ASEP_ERR_ARG -1
#define ASEP_ERR_HIP -2
#define ASEP_ERR_WEIGHTS -3
#define ASEP_ERR_UNSUPPORTED -4

/* ---- runtime ------------------------------------------------------------------------------- */
int asep_device_count(void);
int asep_init(int device_id);                 /* hipSetDevice + arch check (gfx950) */
const char* asep_last_error(void);
const char* asep_version(void);
/* Version of THIS header's struct layouts and signatures (ASEP_ABI_VERSION of the build).  A binding checks it once after
 * dlopen; independently of it every configuration struct starts with its own size in bytes (`struct_size`), and the load
 * functions refuse a struct whose size is not the one the library was built with -- a caller written against an older or newer
 * header gets ASEP_ERR_ARG + a message instead of fields read from whatever follows its struct on the stack. */
#define ASEP_ABI_VERSION 6
int asep_abi_version(void);
/* ABI 6: the environment switches this build of the library reads when a model is loaded (one name per line).
 * Anything else in the environment is not a switch: a caller that records the switches a measurement ran under (bench.py) filters
 * against this list, and the library names a set-but-ignored switch of an earlier round once on stderr. */
const char* asep_engine_switches(void);

/* Page-locked host memory.  The host-pointer entry points (asep_aru_forward, asep_gnn_forward ...) copy with
 * asynchronous transfers on one stream: from / to page-locked buffers these are DMAs at link speed, from / to ordinary
 * (pageable) memory the runtime stages them.  asep_host_alloc returns page-locked memory; asep_host_register page-locks
 * a range the caller already owns (e.g. a shared-memory slot that decoded pages arrive in). */
void* asep_host_alloc(size_t nbytes);
void asep_host_free(void* p);
int asep_host_register(void* p, size_t nbytes);
int asep_host_unregister(void* p);

/* ---- ARU-Net (ARU_v1.py:35-43 hyper-parameters) ---------------------------------------------- */
typedef struct asep_aru_cfg {
    int32_t struct_size;       /* = sizeof(asep_aru_cfg) of the header the caller was compiled / written against */
    int32_t channels;          /* image channels, 1 */
    int32_t n_classes;
    int32_t feat_root;         /* 8 */
    int32_t scale_space_num;   /* 5 */
    int32_t res_depth;         /* 3 */
    int32_t num_scales_att;    /* 3; 0/1 with use_attention=0 */
    int32_t use_attention;     /* graph contains 'ARU' */
    int32_t mvn;               /* per-image standardisation of the input */
    int32_t apply_softmax;     /* export-time class softmax */
    int32_t compute_dtype;     /* 0 = fp32 (f32 MFMA), 1 = bf16 tensors + bf16 MFMA with fp32 accumulation, 2 = fp32 tensors and accumulation with
                                  SPLIT products: x = xh + xm + xl, w = wh + wm + wl (bfloat16 parts, exact), x w as the six largest of the nine
                                  bf16 products on the bf16 MFMA (dropped terms <= 2^-23 |x w|): fp32 results, the fp32 parity gates */
    int32_t activation;        /* ARU_v1.py:70-75 activation_name: 0 = relu, 1 = elu, 2 = leaky (leak 0.1, layers.py:10-30).  It is the
                                  activation of the convR / conv2 layers, of the block ends, the deconvolutions and the attention CNN; the
                                  ReLU between conv1 and convR_0 of a residual block is a ReLU in every variant (ARU_v1.py:214,268) */
    int32_t plain_u;           /* 1 = graph 'U' (ARU_v1.py:228-233,283-288): every block is conv1 + conv2 with the activation, no
                                  residual add, tensors <block>/conv2/{weights,biases} instead of convR_<i>; 0 = residual blocks (RU / ARU) */
} asep_aru_cfg;

typedef struct asep_aru asep_aru;

/* weight_blob: "ASEPW001" container (see citlab-article-separation-new_amd/weights.py) holding
 * the tensors of ARU_v1.py under the reference's variable-scope names. */
asep_aru* asep_aru_load(const void* weight_blob, size_t nbytes, const asep_aru_cfg* cfg);
void asep_aru_free(asep_aru* m);

/* Host-buffer call == get_net_output(): img_hw is H*W*channels floats (row major, values as fed to
 * 'inImg:0'); out_hwc receives H*W*n_classes floats ('output:0'[0]).  The optional u8 outputs are
 * the two consumers that directly follow the net in the reference:
 *   out_u8   = uint8(prob*255)               (truncation, separator_net_post_processor.py:147)
 *   out_mask = (out_u8 > threshold*255)*255  (net_post_processing_helper.py:75-78)
 * Pass NULL to skip either. */
int asep_aru_forward(asep_aru* m, const float* img_hw, int H, int W,
                     float* out_hwc, uint8_t* out_u8, uint8_t* out_mask, float threshold);

/* Device-resident variant (input and outputs already in HBM); asynchronous on `stream`. */
int asep_aru_forward_dev(asep_aru* m, const float* d_img, int H, int W,
                         float* d_out, uint8_t* d_out_u8, uint8_t* d_out_mask, float threshold,
                         void* stream);

/* A batch of n_pages equally sized device-resident pages in one call (BASELINE configs[1]: "batch of
 * 3000x4500 px full pages").  Every layer is launched once for all pages and scale-space levels, which keeps
 * the coarse levels from under-filling the 256 CUs.  d_imgs / d_outs / d_out_u8 / d_out_mask are host arrays
 * of n_pages device pointers (the u8 arrays themselves may be NULL). */
int asep_aru_forward_batch_dev(asep_aru* m, int n_pages, const float* const* d_imgs, int H, int W,
                               float* const* d_outs, uint8_t* const* d_out_u8, uint8_t* const* d_out_mask,
                               float threshold, void* stream);
/* ABI 6: asep_aru_forward_batch_dev for pages of DIFFERENT sizes: H[b] x W[b] is page b's size, d_imgs[b] / d_outs[b] / d_out_u8[b] /
 * d_out_mask[b] its buffers ([H[b], W[b]] fp32 in; [H[b], W[b], n_classes] out).  Replaces the reference's page-by-page loop over scans of
 * arbitrary size (run_net_post_processing.py:61-82 -> get_net_output per scan, ARU_v1.py:64 `inImg` [1, None, None, 1]); results are those
 * of the single-page calls bit for bit.  The pages of a call share every layer's launches. */
int asep_aru_forward_batch_dev2(asep_aru* m, int n_pages, const float* const* d_imgs, const int32_t* H, const int32_t* W,
                                float* const* d_outs, uint8_t* const* d_out_u8, uint8_t* const* d_out_mask, float threshold,
                                void* stream);

/* Named intermediate tensors of the last forward (tests / GNN visual branch): copies the NHWC fp32
 * tensor `name` (e.g. "scale_0_unet_up_0_conv", ARU_v1.py:11-29 end-point names) to host.
 * Returns the number of floats written, or a negative error; dims receives {H, W, C}. */
long asep_aru_get_endpoint(asep_aru* m, const char* name, float* out, size_t max_floats, int32_t dims[3]);

/* ABI 6: release the handle's device arenas (intermediate tensors of the largest call served so far: ~6 GB per 3000 x 4500 fp32 page in flight);
 * the next forward call rebuilds them.  For a process that keeps a model loaded while another GPU process of its size runs beside it. */
int asep_aru_trim(asep_aru* m);

/* Per-launch timing with HIP events recorded on the launch stream (measurement aid for bench.py).
 * asep_aru_profile(m,1) clears the records and starts recording; (m,0) stops.  Mode 1 serialises the net on the
 * launch stream (isolated kernel times), mode 2 additionally names every layer, mode 3 keeps the attention side
 * stream running ("in situ": what rocprofv3 sees in the real schedule).  The report is a JSON array
 * [{"kernel","calls","total_ms","flops"}] aggregated per kernel, "kernel" being the rocprofv3 name without
 * "void ", "asep::", blanks and the argument list; returns its length. */
int asep_aru_profile(asep_aru* m, int enable);
long asep_aru_profile_report(asep_aru* m, char* buf, size_t buflen);

/* Algorithmic FLOPs (2*MAC of conv/deconv layers) of one forward at H x W. */
double asep_aru_flops(const asep_aru* m, int H, int W);

/* ---- GNN relation predictor ------------------------------------------------------------------ */
typedef struct asep_gnn_cfg {
    int32_t struct_size;           /* = sizeof(asep_gnn_cfg) of the header the caller was compiled / written against */
    int32_t node_feature_dim;      /* u (after masking), e.g. 7 */
    int32_t edge_feature_dim;      /* e, e.g. 2; counts the fed (geometric) + compressed visual edge dims (visual_edge_dims) */
    int32_t num_transition_steps;  /* 3 */
    int32_t hidden_dim;            /* 32 (any width; the fused MFMA kernels serve 32/32/32) */
    int32_t interaction_dim;       /* 32 */
    int32_t interaction_hidden;    /* first hidden layer of the interaction MLP (32); further layers: interaction_hidden2..4 below */
    int32_t cls_hidden1;           /* classifier num_hidden_units (graph_relation.py:196), first entry: 64 (64,32 -> 2 has a specialised kernel) */
    int32_t cls_hidden2;           /* 32; 0 = the classifier has ONE hidden layer */
    int32_t num_classes;           /* 2 (<= 16) */
    int32_t undirected_graph;      /* 1 */
    int32_t compress_input_dim;    /* graph_gnn.py:20,102-109 compress_node_feature_dim: 0 = off; > 0: node features are FED with this
                                      width and go through tanh(W x + b) (GraphLSTM1/compress_input/ff_compress_input/{weights,bias})
                                      to node_feature_dim before the message passing */
    int32_t output_type;           /* graph_gnn.py:23,158-166: 0 = 'hidden' (the classifier reads the final hidden states), 1 =
                                      'add_final_hidden_and_input' (h += x W, GraphLSTM1/dense/weights [fed width, hidden], no bias), 2 =
                                      'concat_final_hidden_and_input' (the classifier reads [h | x]: its first layer has 2 (hidden + fed
                                      width) rows); x = the node features AS FED (before compress_input) */
    int32_t attention_heads;       /* message_fn_chunk.py:35-41 use_attention: 0 = off (degree-normalised sum, the reference's default);
                                      k >= 1 = learned attention with k heads (head_<i>/calculation_interaction_features/... and
                                      head_<i>/calculation_unnormalized_attention_values/... MLPs); the reference's chunking of the
                                      interactions by target-node ranges (message_fn_chunk.py:76-110) is reproduced */
    int32_t attention_merge;       /* multihead_attention_merge_type: 0 = 'concat' (x_dim = interaction_dim / heads), 1 = 'average' */
    int32_t attention_hidden;      /* num_hidden_units_attention_fct, first entry (16); further layers: attention_hidden2..4 below */
    /* ---- ABI 5 ---- */
    int32_t aggregation_type;      /* message_fn_chunk.py:16,57-62: 0 = 'sum' (tf.sparse.reduce_sum, the default), 1 = 'max'
                                      (tf.sparse.reduce_max over the stored entries: a node without in-edges gets 0) */
    int32_t interaction_hidden2;   /* num_hidden_units_interaction_fct is a list (message_fn_chunk.py:24): entries 2..4, 0 = absent */
    int32_t interaction_hidden3;
    int32_t interaction_hidden4;
    int32_t attention_hidden2;     /* num_hidden_units_attention_fct entries 2..4 (message_fn_chunk.py:40), 0 = absent */
    int32_t attention_hidden3;
    int32_t attention_hidden4;
    int32_t cls_hidden3;           /* classifier hidden layers 3, 4; 0 = absent */
    int32_t cls_hidden4;
    int32_t lstm_use_hidden;       /* update_fn_lstm.py:13-16,43-50 incorporate_hidden_features_in_update (default 1): h is an input of the gates */
    int32_t lstm_use_input;        /* incorporate_node_input_features_in_update (default 1): the node features are an input of the gates */
    int32_t visual_edge_dims;      /* graph_relation.py:141-172 assign_visual_features_to_edges: total width of the compressed visual EDGE
                                      features (ROI max over the backbone end points of every interaction's region ->
                                      visual_edge_feature_compression_fm_<i>/dense) appended to the fed edge features; 0 = off */
} asep_gnn_cfg;

typedef struct asep_gnn asep_gnn;

asep_gnn* asep_gnn_load(const void* weight_blob, size_t nbytes, const asep_gnn_cfg* cfg);
void asep_gnn_free(asep_gnn* g);

/* misc.py:7-151 on the device: symmetrise (if undirected), de-duplicate, drop self loops, sort by
 * from*N+to, first-occurrence edge features.  out_edges must hold 2*E*2 ints, out_feat 2*E*e
 * floats (may be NULL).  Returns E' (>= 0) or a negative error. */
int asep_gnn_correct_edges(asep_gnn* g, int N, int E, const int32_t* edges, const float* edge_feat,
                           int32_t* out_edges, float* out_feat);

/* One page: == sess.run('output_belong_to_same_instance:0', feed_dict) at batch size 1.
 *   edges [E,2] (from,to), node_feat [N,u], edge_feat [E,e], relations [R,2] -> probs_out [R,num_classes].
 * relations == NULL means "all N*N ordered pairs, row major" (input_dataset.py:444-457). */
int asep_gnn_forward(asep_gnn* g, int N, int E, const int32_t* edges, const float* node_feat,
                     const float* edge_feat, int R, const int32_t* relations, float* probs_out);

int asep_gnn_forward_dev(asep_gnn* g, int N, int E, const int32_t* d_edges, const float* d_node_feat,
                         const float* d_edge_feat, int R, const int32_t* d_relations, float* d_probs_out,
                         void* stream);

/* Final hidden node states h [N, hidden_dim] of the last forward (tests). */
int asep_gnn_get_hidden(asep_gnn* g, float* out, size_t max_floats);

double asep_gnn_flops(const asep_gnn* g, int N, int E_corrected, int R);

/* Visual node features (graph_relation.py:84-139 with image_input; misc.py:272-381): `backbone` is an ARU-Net
 * handle loaded from the `aru_net/...` tensors of the same frozen graph (graph 'RU'/'ARU', mvn per the GNN's flag,
 * apply_softmax 0); endpoint_names are the `feature_map_generation_params from_layer` entries (layer_depth -1),
 * e.g. "scale_0_unet_up_2_conv".  The compression layers visual_node_feature_compression_fm_<i>/dense/{weights,bias} come from
 * the GNN's own weight blob.  cfg.node_feature_dim counts geometric + compressed visual dims (e.g. 7 + 3*16).
 * The backbone may be loaded with compute_dtype 1 (bf16, BASELINE configs[4]): the ROI kernel then reads its bf16 end points;
 * compression layers, graph and classifier stay fp32. */
int asep_gnn_attach_backbone(asep_gnn* g, asep_aru* backbone, int n_maps, const char* const* endpoint_names);

/* run_gnn_clustering.py:259-269 with the image feeds: node_feat [N, node_feature_dim - visual dims],
 * image float32 [h,w] (0..255 as fed, input_dataset.py:279-280), regions [N,2,P] relative coordinates (row 0 = x,
 * row 1 = y), num_points [N].  Backbone -> ROI max -> compression -> concat -> GNN -> probabilities [R, classes].
 * edge_regions [E,2,P] / edge_num_points [E] ('visual_regions_edges', 'num_points_visual_regions_edges'): the regions of the
 * interactions of a graph exported with assign_visual_features_to_edges (cfg.visual_edge_dims > 0; NULL otherwise): their
 * compressed ROI maxima are appended to edge_feat [E, edge_feature_dim - visual_edge_dims] BEFORE the edge correction, which
 * keeps the first occurrence's features like every other edge feature. */
int asep_gnn_forward_visual(asep_gnn* g, int N, int E, const int32_t* edges, const float* node_feat,
                            const float* edge_feat, const float* image, int h, int w, const float* regions, int P,
                            const int32_t* num_points, const float* edge_regions, const int32_t* edge_num_points,
                            int R, const int32_t* relations, float* probs_out);

/* The same with every array already in HBM (image [h,w] float32 included), launched on `stream` without any host
 * synchronisation: backbone, ROI kernels and the graph are queued back to back.  Index arrays are not validated here:
 * an edge that names a node outside 0..N-1 is ignored, a relation that does yields NaN probabilities. */
int asep_gnn_forward_visual_dev(asep_gnn* g, int N, int E, const int32_t* d_edges, const float* d_node_feat,
                                const float* d_edge_feat, const float* d_image, int h, int w, const float* d_regions,
                                int P, const int32_t* d_num_points, const float* d_edge_regions,
                                const int32_t* d_edge_num_points, int R, const int32_t* d_relations,
                                float* d_probs_out, void* stream);

/* A batch of pages through the visual net (bench.py's step; a GPU owner that holds several decoded pages): the backbones
 * of all pages run as ONE grouped forward (asep_aru_forward_batch_dev: every layer is one launch over the page list, so the
 * small 683 x 1024 images fill the chip together), then ROI kernels + graph per page, all queued on `stream`, no host
 * synchronisation.  Every image is [h,w] float32 and every region array [N,2,P]; the other sizes are per page.
 * asep_gnn_get_node_features afterwards returns page 0's features. */
typedef struct asep_gnn_page {
    int32_t N, E, R;
    const int32_t* d_edges;        /* [E,2] */
    const float* d_node_feat;      /* [N, node_feature_dim - visual dims] */
    const float* d_edge_feat;      /* [E, edge_feature_dim] */
    const float* d_image;          /* [h,w] */
    const float* d_regions;        /* [N,2,P] */
    const int32_t* d_num_points;   /* [N] */
    const float* d_edge_regions;   /* [E,2,P] or NULL (cfg.visual_edge_dims == 0) */
    const int32_t* d_edge_num_points; /* [E] or NULL */
    const int32_t* d_relations;    /* [R,2] or NULL = all N*N ordered pairs (then R must be N*N) */
    float* d_probs_out;            /* [R, num_classes] */
} asep_gnn_page;
int asep_gnn_forward_visual_batch_dev(asep_gnn* g, int n_pages, const asep_gnn_page* pages, int h, int w, int P,
                                      void* stream);

/* Which message-passing kernel the handle uses: 0 = generic FMA kernels (any widths), 1 = fused MFMA step with the
 * edge-MLP filter in registers (widths 32, node_feature_dim <= 8), 2 = fused MFMA step with the filter in LDS (widths 32,
 * node_feature_dim <= 120: the visual nets), 3 (ABI 6, the default wherever 1 or 2 would serve) = the FACTORED fused step: the per-node
 * terms of the edge MLP's first layer once per node, its step-independent per-edge term once per page, K = 32 per edge and step
 * (ASEP_GNN_FACTOR=0 keeps 1 / 2). */
int asep_gnn_step_mode(const asep_gnn* g);

/* Concatenated node features [N, node_feature_dim] of the last asep_gnn_forward_visual[_dev] (tests). */
int asep_gnn_get_node_features(asep_gnn* g, float* out, size_t max_floats);

/* ---- classical image stages around the ARU-Net ---------------------
 * The reference runs these on the host with OpenCV; here they are byte / bit kernels next to the nets so that a
 * page never leaves HBM between decode and polygon extraction.  Binary images: 0 = background, non-zero =
 * foreground on input; 0 / 255 on output.  One workspace handle per (process, device); buffers grow on demand. */
typedef struct asep_post asep_post;
asep_post* asep_post_create(void);
void asep_post_free(asep_post* p);

/* net_post_processing_helper.py:14-33 (scale_image + cvtColor(BGR2GRAY)/255.0) without the file decode.
 * img: uint8 [H,W,C] with C = 3 (BGR) or 1 (gray).  sc < 1: cv2 INTER_AREA, sc > 1: INTER_CUBIC, sc == 1: copy.
 * Output size (round-half-even of H*sc, W*sc) is returned by asep_prep_scaled_size.
 * out_image (optional): uint8 [h,w,C]; out_gray: float32 [h,w] = gray/255 (the net input). */
int asep_prep_scaled_size(int H, int W, double sc, int32_t* out_h, int32_t* out_w);
int asep_prep_scale_gray(asep_post* p, const uint8_t* img, int H, int W, int C, double sc,
                         uint8_t* out_image, float* out_gray);
int asep_prep_scale_gray_dev(asep_post* p, const uint8_t* d_img, int H, int W, int C, double sc,
                             uint8_t* d_out_image, float* d_out_gray, void* stream);

/* region_net_post_processor_base.py:230-251 apply_cc_analysis: drop 8-connected components whose pixel count is
 * below min_size (the caller evaluates int(size * threshold) in double like the reference).
 * mask: uint8 [H,W,pix_stride], channel `channel` is used. */
int asep_post_cc_filter(asep_post* p, const uint8_t* mask, int H, int W, int pix_stride, int channel,
                        int min_size, uint8_t* out);

/* cv2.erode / dilate / morphologyEx(MORPH_OPEN | MORPH_CLOSE) with a kw x kh rectangle, default anchor
 * (kw/2, kh/2), default constant border; op: 0 erode, 1 dilate, 2 open, 3 close. */
int asep_post_morph_rect(asep_post* p, int op, const uint8_t* mask, int H, int W, int kw, int kh, uint8_t* out);

/* separator_net_post_processor.py:26-97 post_process: CC filter, horizontal opening (k_h x 1), vertical opening
 * (1 x k_v), horizontal minus vertical, clean-up opening (k_clean x 1).  mask: uint8 [H,W,pix_stride] (the
 * thresholded net output, channel 0 is the separator class).  Outputs uint8 [H,W] each. */
int asep_post_separator(asep_post* p, const uint8_t* mask, int H, int W, int pix_stride, int channel,
                        int min_size, int k_h, int k_v, int k_clean, uint8_t* out_horizontal,
                        uint8_t* out_vertical);
int asep_post_separator_dev(asep_post* p, const uint8_t* d_mask, int H, int W, int pix_stride, int channel,
                            int min_size, int k_h, int k_v, int k_clean, uint8_t* d_out_horizontal,
                            uint8_t* d_out_vertical, void* stream);

/* Device half of region_net_post_processor_base.py:186-197 (rasterio.features.shapes): the boundary of the pixels
 * equal to `value` as maximal straight segments, each directed with the foreground on its right (heading 0 = +x top
 * sides, 1 = +y right sides, 2 = -x bottom sides, 3 = -y left sides).  out_starts / out_ends receive the keys
 * ((vy*(W+1)+vx)*4 + heading) of the start / end vertices in arbitrary order; segments of one heading on one grid
 * line are disjoint, so after sorting both arrays per heading the k-th start pairs with the k-th end.  Returns the
 * number of segments (may exceed `capacity`: call again with larger buffers) or a negative error.  The host chains
 * the segments into exterior / interior rings (polygonize.py). */
long asep_post_boundary_segments(asep_post* p, const uint8_t* mask, int H, int W, int value, int32_t* out_starts,
                                 int32_t* out_ends, long capacity);
long asep_post_boundary_segments_dev(asep_post* p, const uint8_t* d_mask, int H, int W, int value, int32_t* d_starts,
                                     int32_t* d_ends, long capacity, void* stream);
/* The same without the host round trip, for a caller that keeps more than one page in flight: the kernel is
 * enqueued on `stream` and the two counts (starts, ends; equal on success, either may exceed `capacity`) are left in
 * d_totals[0..1] (device memory, caller-owned).  Nothing is synchronised: the caller copies d_totals and the first
 * min(total, capacity) keys back in stream order and falls back to asep_post_boundary_segments_dev with larger
 * buffers when total > capacity. */
int asep_post_boundary_segments_enqueue_dev(asep_post* p, const uint8_t* d_mask, int H, int W, int value,
                                            int32_t* d_starts, int32_t* d_ends, long capacity,
                                            unsigned long long* d_totals, void* stream);

/* cv2.imread(path, IMREAD_GRAYSCALE) of an image that is already decoded and on the device (swt_dist_trafo.py:19):
 * d_bgr uint8 [H,W,3] in B,G,R order -> d_out uint8 [H,W] with OpenCV's fixed-point weights
 * (B*3735 + G*19235 + R*9798 + 16384) >> 15.  Enqueued on `stream`, nothing is synchronised. */
int asep_prep_gray_u8_dev(asep_post* p, const uint8_t* d_bgr, int H, int W, uint8_t* d_out, void* stream);

/* heading_net_post_processor.py:247-270 (get_net_prob_for_text_line) without the net output leaving the device:
 * out_sums[i] = sum of d_img[y0:y1, x0:x1, channel] over box i = {x0, y0, x1, y1} (clipped to the image like a numpy
 * slice with non-negative bounds), d_img uint8 [H,W,pix_stride].  Exact integers; the caller divides by 255 and by
 * the nominal box size.  boxes and out_sums are host pointers; returns after the sums have arrived. */
int asep_post_box_sums_dev(asep_post* p, const uint8_t* d_img, int H, int W, int pix_stride, int channel, int n_boxes,
                           const int32_t* boxes, int64_t* out_sums, void* stream);

/* swt_dist_trafo.py:18-29 distance_transform without the file decode: gray uint8 [H,W] -> 255-gray ->
 * GaussianBlur 5x5 -> Otsu -> exact Euclidean distance to the nearest zero pixel -> astype(uint8).
 * out_otsu (optional) receives the Otsu threshold; out_d2 (optional) the exact squared distances (int32). */
int asep_swt_distance_transform(asep_post* p, const uint8_t* gray, int H, int W, uint8_t* out,
                                int32_t* out_otsu, int32_t* out_d2);
int asep_swt_distance_transform_dev(asep_post* p, const uint8_t* d_gray, int H, int W, uint8_t* d_out,
                                    void* stream);

/* heading_net_post_processor.py:218-245 / feature_generation.py:106-159 for all text lines of a page at once.
 * swt: the uint8 distance-transform image [H,W]; boxes: n_lines x {x0, y0, x1, y1} = the crop swt[y0:y1, x0:x1]
 * the reference takes per line (numpy slicing: bounds are clipped to the image).  Per line: 8-connected components
 * of the non-zero crop pixels (connected_components_cv), size / aspect cleaning (clean_connected_components),
 * the crop's maximum over each surviving component's bounding box; out_stroke_width = median of those maxima
 * (0.0 if none), out_height = largest surviving component height.  out_flag[i] = 1 marks a line with more than
 * 1024 components whose results were not computed (the caller evaluates that line on the host).
 * boxes and the three outputs are host pointers in both variants. */
int asep_swt_line_features(asep_post* p, const uint8_t* swt, int H, int W, int n_lines, const int32_t* boxes,
                           float* out_stroke_width, int32_t* out_height, int32_t* out_flag);
int asep_swt_line_features_dev(asep_post* p, const uint8_t* d_swt, int H, int W, int n_lines, const int32_t* boxes,
                               float* out_stroke_width, int32_t* out_height, int32_t* out_flag, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ASEP_HIP_H */
