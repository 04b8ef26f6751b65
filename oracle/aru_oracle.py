"""CPU ORACLE (test infrastructure, NOT product code) -- ARU-Net forward pass.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product path (``citlab-article-separation-new_amd``) never does.

PARITY UNPINNED: the reference model path needs TensorFlow 1.12-1.14 and the frozen ``.pb``
files, neither of which exists here and the reference has no tests or
golden vectors for it.  This file is a restatement of the reference *graph definition*

    article_separation/backbones/ARU_v1.py:62-294            (_create_aru_net, _attCNN, _detCNN)
    article_separation/gnn/model/graph_util/layers.py:191-247 (conv2d), :342-367 (deconv2d),
        :526-544 (avg/max pool), :672-711 (per_image_standardization), :716-720 (upsample_simple)

with the TensorFlow op semantics (SAME padding, conv2d_transpose cropping, avg-pool divisor)
restated from the TF 1.x documentation.  Two independent
implementations are kept and cross-checked in ``tests/test_oracle_aru.py``:

  * ``forward_numpy``  -- explicit shifted-slice sums in numpy (float32 or float64), no library conv;
  * ``forward_torch``  -- torch-CPU ``F.conv2d`` / ``F.conv_transpose2d`` with explicit padding/cropping
                          (fast enough for full pages; also the ``cpu_baseline`` of bench.py).
"""
import numpy as np


# ----------------------------------------------------------------------------------------------
# numpy ops (NHWC without batch: [H, W, C])
# ----------------------------------------------------------------------------------------------
def same_pad(k: int):
    """TF SAME, stride 1: pad_total = k-1, pad_before = pad_total // 2 (3x3 -> (1,1); 4x4 -> (1,2))."""
    total = k - 1
    return total // 2, total - total // 2


def conv2d_same(x, w, b=None):
    """layers.py:191-247 -> tf.nn.conv2d(stride 1, SAME) + bias.  x [H,W,Ci], w [kh,kw,Ci,Co]."""
    H, W, _ = x.shape
    kh, kw, ci, co = w.shape
    pt, pb = same_pad(kh)
    pl, pr = same_pad(kw)
    xp = np.pad(x, ((pt, pb), (pl, pr), (0, 0)))
    out = np.zeros((H, W, co), dtype=x.dtype)
    for ky in range(kh):
        for kx in range(kw):
            out += xp[ky:ky + H, kx:kx + W, :] @ w[ky, kx].astype(x.dtype)
    if b is not None:
        out += b.astype(x.dtype)
    return out


def conv2d_transpose_same(x, w, out_hw, stride):
    """tf.nn.conv2d_transpose(x, w[kh,kw,Co,Ci], out_shape, stride, SAME) (layers.py:362).

    y[i] = sum_{o,k : o*s + k - pad_before = i} x[o] * w[k]  with
    pad_total = max((h_in-1)*s + k - H_out, 0), pad_before = pad_total // 2 (no kernel flip)."""
    h, wd, ci = x.shape
    kh, kw, co, ci2 = w.shape
    assert ci == ci2
    Ho, Wo = out_hw
    assert -(-Ho // stride) == h and -(-Wo // stride) == wd, "TF would reject this out_shape"
    pbh = max((h - 1) * stride + kh - Ho, 0) // 2
    pbw = max((wd - 1) * stride + kw - Wo, 0) // 2
    full_h = (h - 1) * stride + kh
    full_w = (wd - 1) * stride + kw
    full = np.zeros((max(full_h, Ho + pbh), max(full_w, Wo + pbw), co), dtype=x.dtype)
    for ky in range(kh):
        for kx in range(kw):
            contrib = x @ w[ky, kx].astype(x.dtype).T          # [h, wd, co]
            full[ky:ky + (h - 1) * stride + 1:stride, kx:kx + (wd - 1) * stride + 1:stride, :] += contrib
    return full[pbh:pbh + Ho, pbw:pbw + Wo, :]


def deconv2d(x, w, b, out_hw, stride=2):
    """layers.py:342-367: conv2d_transpose + bias (+ activation applied by the caller)."""
    return conv2d_transpose_same(x, w, out_hw, stride) + b.astype(x.dtype)


def _pool_windows(x, k=2):
    H, W, C = x.shape
    Ho, Wo = -(-H // k), -(-W // k)
    return Ho, Wo


def max_pool2(x):
    """tf.nn.max_pool2d(2x2, s2, SAME): out = ceil(in/2), padding at the end only, ignored by max."""
    H, W, C = x.shape
    Ho, Wo = _pool_windows(x)
    xp = np.full((Ho * 2, Wo * 2, C), -np.inf, dtype=x.dtype)
    xp[:H, :W] = x
    return xp.reshape(Ho, 2, Wo, 2, C).max(axis=(1, 3))


def avg_pool2(x):
    """tf.nn.avg_pool2d(2x2, s2, SAME): divisor = number of *valid* elements."""
    H, W, C = x.shape
    Ho, Wo = _pool_windows(x)
    xp = np.zeros((Ho * 2, Wo * 2, C), dtype=x.dtype)
    xp[:H, :W] = x
    cnt = np.zeros((Ho * 2, Wo * 2, 1), dtype=x.dtype)
    cnt[:H, :W] = 1
    s = xp.reshape(Ho, 2, Wo, 2, C).sum(axis=(1, 3))
    n = cnt.reshape(Ho, 2, Wo, 2, 1).sum(axis=(1, 3))
    return s / n


def upsample_simple(x, out_hw, up):
    """layers.py:716-720: conv2d_transpose with an all-ones [up,up,C,C] filter, stride up, SAME.

    => nearest-neighbour upsample AND channel sum (every output channel = sum of input channels),
    cropped with offset (h*up - H_out)//2.  Returns [H_out, W_out, C]."""
    h, w, C = x.shape
    Ho, Wo = out_hw
    assert -(-Ho // up) == h and -(-Wo // up) == w
    ph = (h * up - Ho) // 2
    pw = (w * up - Wo) // 2
    s = x.sum(axis=2, keepdims=True)
    iy = (np.arange(Ho) + ph) // up
    ix = (np.arange(Wo) + pw) // up
    y = s[iy][:, ix]
    return np.repeat(y, C, axis=2)


def relu(x):
    return np.maximum(x, 0)


def activation_fn(name):
    """ARU_v1.py:70-75: the graph's activation -- layers.relu, layers.elu (tf.nn.elu: x > 0 ? x : exp(x) - 1) or layers.leaky_relu
    (layers.py:10-30: max(0, x) + 0.1 * min(0, x))."""
    if name == "relu":
        return relu
    if name == "elu":
        return lambda x: np.where(x > 0, x, np.exp(np.minimum(x, 0)) - 1).astype(x.dtype)
    if name == "leaky":
        return lambda x: (np.maximum(x, 0) + x.dtype.type(0.1) * np.minimum(x, 0)).astype(x.dtype)
    raise ValueError(f"activation_name {name!r} (ARU_v1.py:43: relu, elu, leaky)")


def softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def per_image_standardization(x):
    """layers.py:672-711: (x-mean)/max(sqrt(relu(E[x^2]-mean^2)), 1e-4)."""
    mean = x.mean(dtype=x.dtype)
    var = max((x * x).mean(dtype=x.dtype) - mean * mean, 0)
    return (x - mean) / max(np.sqrt(var), 1e-4)


# ----------------------------------------------------------------------------------------------
# network (numpy)
# ----------------------------------------------------------------------------------------------
def _res_block(x, w, prefix, res_depth, act=relu):
    """ARU_v1.py:212-227 / :266-281.  conv1 (identity) -> save -> relu (ALWAYS relu, :214) -> (res_depth-1) x conv+act
    -> conv (identity) -> add -> act."""
    t = conv2d_same(x, w[prefix + "/conv1/weights"], w[prefix + "/conv1/biases"])
    r = relu(t)
    for a in range(res_depth):
        r = conv2d_same(r, w[prefix + f"/convR_{a}/weights"], w[prefix + f"/convR_{a}/biases"])
        if a < res_depth - 1:
            r = act(r)
    return act(r + t)


def _plain_block(x, w, prefix, act):
    """ARU_v1.py:228-233 / :283-288 (graph 'U'): conv1 + act -> conv2 + act."""
    c1 = act(conv2d_same(x, w[prefix + "/conv1/weights"], w[prefix + "/conv1/biases"]))
    return act(conv2d_same(c1, w[prefix + "/conv2/weights"], w[prefix + "/conv2/biases"]))


def det_cnn(x, w, cfg, end_points=None, sc=0):
    """ARU_v1.py:186-294."""
    n = cfg.scale_space_num
    act = activation_fn(getattr(cfg, "activation_name", "relu"))
    residual = getattr(cfg, "use_residual", True)
    block = (lambda y, p: _res_block(y, w, p, cfg.res_depth, act)) if residual else (lambda y, p: _plain_block(y, w, p, act))
    skips = []
    u = x
    for l in range(n):
        d = block(u, f"aru_net/featMapG/unet_down_{l}")
        skips.append(d)
        if end_points is not None:
            end_points[f"scale_{sc}_unet_down_{l}_conv"] = d
        u = max_pool2(d) if l < n - 1 else d
    for l in range(n - 2, -1, -1):
        p = f"aru_net/featMapG/unet_up_{l}"
        skip = skips[l]
        v = act(deconv2d(u, w[p + "/deconv/weights"], w[p + "/deconv/bias"], skip.shape[:2], cfg.pool_size))
        if end_points is not None:
            end_points[f"scale_{sc}_unet_up_{l}_deconv"] = v
        c = np.concatenate([skip, v], axis=2)                      # skip first (ARU_v1.py:264)
        u = block(c, p)
        if end_points is not None:
            end_points[f"scale_{sc}_unet_up_{l}_conv"] = u
    return u


def att_cnn(x, w, act=relu):
    """ARU_v1.py:165-184: 4x4 conv 12 -> pool -> 16 -> pool -> 32 -> pool -> 4x4 conv 1, all with the graph's activation."""
    p = "aru_net/attMapG/attPart/conv"
    y = act(conv2d_same(x, w[p + "1/weights"], w[p + "1/biases"]))
    y = max_pool2(y)
    y = act(conv2d_same(y, w[p + "2/weights"], w[p + "2/biases"]))
    y = max_pool2(y)
    y = act(conv2d_same(y, w[p + "3/weights"], w[p + "3/biases"]))
    y = max_pool2(y)
    y = act(conv2d_same(y, w[p + "4/weights"], w[p + "4/biases"]))
    return y


def forward_numpy(image, w, cfg, dtype=np.float32, return_intermediates=False):
    """image [H,W] or [H,W,C] -> probabilities (or logits if not cfg.apply_softmax) [H,W,n_classes].

    ARU_v1.py:62-163."""
    x = np.asarray(image, dtype=dtype)
    if x.ndim == 2:
        x = x[:, :, None]
    w = {k: v.astype(dtype) for k, v in w.items()}
    H, W, _ = x.shape
    inter = {}
    if cfg.mvn:
        x = per_image_standardization(x)
    scales = [x]
    if cfg.use_attention:
        for _ in range(1, cfg.num_scales_att):
            scales.append(avg_pool2(scales[-1]))                   # ARU_v1.py:106-109
        att = []
        up = 8
        for s in range(cfg.num_scales_att):                        # ARU_v1.py:113-118
            a = att_cnn(scales[s], w, activation_fn(getattr(cfg, "activation_name", "relu")))
            inter[f"att_{s}"] = a
            att.append(upsample_simple(a, (H, W), up)[:, :, :1])   # out shape = input shape (1 ch)
            up *= 2
    feats = [det_cnn(x, w, cfg, inter, 0)]
    if cfg.use_attention:
        up = 1
        for s in range(1, cfg.num_scales_att):                     # ARU_v1.py:129-138
            f = det_cnn(scales[s], w, cfg, inter, s)
            up *= 2
            feats.append(upsample_simple(f, (H, W), up))
        a = softmax(np.concatenate(att, axis=2), axis=2)           # over the scale axis
        m = sum(feats[s] * a[:, :, s:s + 1] for s in range(cfg.num_scales_att))
    else:
        m = feats[0]
    inter["sum_att_feat_map"] = m
    logits = conv2d_same(m, w["aru_net/logit/class/weights"], w["aru_net/logit/class/biases"])
    inter["logits"] = logits
    out = softmax(logits, axis=2) if cfg.apply_softmax else logits
    if return_intermediates:
        return out, inter
    return out


# ----------------------------------------------------------------------------------------------
# torch-CPU implementation (NCHW internally)
# ----------------------------------------------------------------------------------------------
def _t_conv(x, w, b, F, torch):
    kh, kw = w.shape[0], w.shape[1]
    pt, pb = same_pad(kh)
    pl, pr = same_pad(kw)
    wt = w.permute(3, 2, 0, 1).contiguous()                       # [Co,Ci,kh,kw]
    return F.conv2d(F.pad(x, (pl, pr, pt, pb)), wt, b)


def _t_deconv(x, w, b, out_hw, stride, F, torch):
    kh, kw = w.shape[0], w.shape[1]
    h, wd = x.shape[2], x.shape[3]
    Ho, Wo = out_hw
    pbh = max((h - 1) * stride + kh - Ho, 0) // 2
    pbw = max((wd - 1) * stride + kw - Wo, 0) // 2
    # torch conv_transpose2d weight: [Cin, Cout, kh, kw]; y[o*s + k] += x[o] * w[ci,co,k]  (no flip)
    wt = w.permute(3, 2, 0, 1).contiguous()
    full = F.conv_transpose2d(x, wt, None, stride=stride)
    need_h, need_w = pbh + Ho, pbw + Wo
    if full.shape[2] < need_h or full.shape[3] < need_w:
        full = F.pad(full, (0, max(0, need_w - full.shape[3]), 0, max(0, need_h - full.shape[2])))
    return full[:, :, pbh:pbh + Ho, pbw:pbw + Wo] + b.view(1, -1, 1, 1)


def _t_upsample(x, out_hw, up, torch):
    h, w = x.shape[2], x.shape[3]
    Ho, Wo = out_hw
    ph = (h * up - Ho) // 2
    pw = (w * up - Wo) // 2
    s = x.sum(dim=1, keepdim=True)
    iy = (torch.arange(Ho) + ph) // up
    ix = (torch.arange(Wo) + pw) // up
    return s[:, :, iy][:, :, :, ix]                                # [1,1,Ho,Wo]; caller broadcasts


def forward_torch(image, w, cfg, num_threads=None, dtype=None, return_intermediates=False, bn=None, storage="f32", teacher=None):
    """``bn``: optional {layer scope: (gamma, beta, moving_mean, moving_variance, epsilon)} -- inference-mode batch
    normalisation applied UNFOLDED between bias and activation (layers.py:241-242 ``batchNorm`` switch); the importer
    folds it into the weights, this is the independent evaluation it is compared with.

    ``storage="bf16"``: the SAME graph with the roundings of the engine's bf16 data path (BASELINE configs[4] "bf16 convs";
    csrc/bf16_kernels.h): the filters of the feature CNN and of the attention CNN's conv2 / conv3 are rounded to
    bfloat16 (round to nearest even), every tensor those layers -- and the deconvolutions and the attention head -- write is rounded to
    bfloat16 after bias and residual add (ReLU / max pool commute with the rounding), the feature CNN's first layer reads the
    (standardised) image rounded to bfloat16 -- and so does, in ReLU graphs, the attention head with its filter rounded to bfloat16
    (att_headb_kernel, round 5; elu / leaky graphs keep the fp32 head); sums stay fp32, and so do biases, conv4's filter, the
    channel sums, the blend over the scales, the logits layer and the soft-maxes.  This is NOT the reference's arithmetic (that is
    ``storage="f32"``): it is the checker that tells a wrong bf16 kernel from bf16 rounding -- against it the engine's bf16 path has
    to agree far more closely than the 2e-2 it is allowed against the fp32 graph (tests/test_full_frame_gpu.py).

    ``teacher``: {end-point name: [H,W,C] array} -- "teacher forcing": every end point is still computed (and returned in the
    intermediates) from its inputs, but then REPLACED by the teacher's tensor before anything downstream reads it.  With the engine's
    own end points as the teacher, each returned end point is what the oracle makes of the ENGINE's upstream tensors, i.e. the
    comparison isolates one block (conv1 + residual tail, or one deconvolution) instead of letting forty layers of flipped roundings
    pile up -- two bf16 evaluations of the whole net that differ only in summation order drift apart almost as far as bf16 drifts
    from fp32 (measured: rms 2.0e-3 of max|ref| against 3.4e-3), block by block they agree to a bfloat16 step on a few elements."""
    import torch
    import torch.nn.functional as F
    if num_threads:
        torch.set_num_threads(num_threads)
    dtype = dtype or torch.float32
    with torch.no_grad():
        x = torch.as_tensor(np.asarray(image)).to(dtype)
        if x.ndim == 2:
            x = x[:, :, None]
        x = x.permute(2, 0, 1)[None].contiguous()
        tw = {k: torch.as_tensor(v).to(dtype) for k, v in w.items()}
        H, W = x.shape[2], x.shape[3]
        inter = {}
        if storage not in ("f32", "bf16"):
            raise ValueError(f"storage {storage!r}")
        emu = storage == "bf16"

        def force(name, t):
            """record the oracle's value of an end point, hand the teacher's on to the layers downstream"""
            inter[name] = t
            if teacher is not None and name in teacher:
                tt = torch.as_tensor(np.asarray(teacher[name])).to(dtype)
                tt = tt.permute(2, 0, 1)[None].contiguous()
                if tt.shape != t.shape:
                    raise ValueError(f"teacher tensor {name}: shape {tuple(tt.shape)}, the graph has {tuple(t.shape)}")
                return tt
            return t
        q = (lambda t: t.to(torch.bfloat16).to(dtype)) if emu else (lambda t: t)      # torch rounds to nearest even
        bf_head = emu and getattr(cfg, "activation_name", "relu") == "relu"       # the attention head on the bf16 MFMA (ReLU graphs)
        if emu:
            for k in list(tw):
                bf_filter = (k.startswith("aru_net/featMapG/") or k.startswith("aru_net/attMapG/attPart/conv2")
                             or k.startswith("aru_net/attMapG/attPart/conv3")
                             or (bf_head and k.startswith("aru_net/attMapG/attPart/conv1/")))
                if bf_filter and k.endswith("/weights"):
                    tw[k] = q(tw[k])

        def batch_norm(y, p):
            if not bn or p not in bn:
                return y
            gamma, beta, mean, var = (torch.as_tensor(np.asarray(v)).to(dtype).view(1, -1, 1, 1) for v in bn[p][:4])
            return (y - mean) / torch.sqrt(var + float(bn[p][4])) * gamma + beta

        def conv(x, p, bias_name="biases"):
            return batch_norm(_t_conv(x, tw[p + "/weights"], tw[p + "/" + bias_name], F, torch), p)

        act_name = getattr(cfg, "activation_name", "relu")
        if act_name == "relu":
            act = F.relu
        elif act_name == "elu":
            act = lambda y: torch.where(y > 0, y, torch.exp(torch.clamp(y, max=0)) - 1)      # tf.nn.elu
        elif act_name == "leaky":
            act = lambda y: torch.clamp(y, min=0) + 0.1 * torch.clamp(y, max=0)              # layers.py:10-30
        else:
            raise ValueError(f"activation_name {act_name!r}")

        # a stored tensor of the bf16 data path = round(activation(fp32 sums)): the engine applies the activation to the fp32 accumulators and
        # rounds once (for ReLU the order does not matter -- rounding is monotone and keeps the sign --, for elu / leaky it does: round 5)
        aq = lambda y: q(act(y))

        def block(x, p):
            if not getattr(cfg, "use_residual", True):       # graph 'U' (ARU_v1.py:228-233)
                return aq(conv(aq(conv(x, p + "/conv1")), p + "/conv2"))
            t = q(conv(x, p + "/conv1"))
            r = F.relu(t)                                    # always a ReLU (ARU_v1.py:214)
            for a in range(cfg.res_depth):
                r = conv(r, p + f"/convR_{a}")
                if a < cfg.res_depth - 1:
                    r = aq(r)
            return aq(r + t)

        def det(x, sc):
            n = cfg.scale_space_num
            skips = []
            u = q(x)                                         # bf16 path: conv1 of the first block reads the image as bfloat16
            for l in range(n):
                d = force(f"scale_{sc}_unet_down_{l}_conv", block(u, f"aru_net/featMapG/unet_down_{l}"))
                skips.append(d)
                u = F.max_pool2d(d, 2, 2, ceil_mode=True) if l < n - 1 else d
            for l in range(n - 2, -1, -1):
                p = f"aru_net/featMapG/unet_up_{l}"
                skip = skips[l]
                v = aq(batch_norm(_t_deconv(u, tw[p + "/deconv/weights"], tw[p + "/deconv/bias"],
                                            skip.shape[2:], cfg.pool_size, F, torch), p + "/deconv"))
                v = force(f"scale_{sc}_unet_up_{l}_deconv", v)
                u = force(f"scale_{sc}_unet_up_{l}_conv", block(torch.cat([skip, v], dim=1), p))
            return u

        def att(x):
            p = "aru_net/attMapG/attPart/conv"
            y = aq(conv(q(x) if bf_head else x, p + "1"))    # (bf16 path: the pooled output stored as bfloat16; ReLU graphs: image and filter as bfloat16)
            y = F.max_pool2d(y, 2, 2, ceil_mode=True)
            y = aq(conv(y, p + "2"))
            y = F.max_pool2d(y, 2, 2, ceil_mode=True)
            y = aq(conv(y, p + "3"))
            y = F.max_pool2d(y, 2, 2, ceil_mode=True)
            return act(conv(y, p + "4"))

        if cfg.mvn:
            mean = x.mean()
            var = torch.clamp((x * x).mean() - mean * mean, min=0)
            x = (x - mean) / torch.clamp(var.sqrt(), min=1e-4)
        scales = [x]
        if cfg.use_attention:
            for _ in range(1, cfg.num_scales_att):
                scales.append(F.avg_pool2d(scales[-1], 2, 2, ceil_mode=True, count_include_pad=False))
            atts = []
            up = 8
            for s in range(cfg.num_scales_att):
                a = force(f"att_{s}", att(scales[s]))
                atts.append(_t_upsample(a, (H, W), up, torch))
                up *= 2
        feats = [det(x, 0)]
        if cfg.use_attention:
            up = 1
            for s in range(1, cfg.num_scales_att):
                f = det(scales[s], s)
                up *= 2
                feats.append(_t_upsample(f, (H, W), up, torch))   # channel-sum, broadcast over C
            a = torch.softmax(torch.cat(atts, dim=1), dim=1)
            m = sum(feats[s] * a[:, s:s + 1] for s in range(cfg.num_scales_att))
        else:
            m = feats[0]
        inter["sum_att_feat_map"] = m
        logits = conv(m, "aru_net/logit/class")
        inter["logits"] = logits
        out = torch.softmax(logits, dim=1) if cfg.apply_softmax else logits
        res = out[0].permute(1, 2, 0).contiguous().numpy()
        if return_intermediates:
            return res, {k: v[0].permute(1, 2, 0).contiguous().numpy() for k, v in inter.items()}
        return res


# ----------------------------------------------------------------------------------------------
# consumers of the probability map that sit directly behind the net (fused into the HIP epilogue)
# ----------------------------------------------------------------------------------------------
def to_uint8(prob):
    """separator_net_post_processor.py:147: np.array(net_output * 255, dtype=np.uint8) (truncation)."""
    return np.array(prob * 255, dtype=np.uint8)


def apply_threshold(net_output, threshold):
    """net_post_processing_helper.py:75-78."""
    if net_output.dtype == np.uint8:
        threshold = threshold * 255
    return np.array((net_output > threshold) * 255, dtype=np.uint8)
