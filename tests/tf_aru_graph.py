"""TEST INFRASTRUCTURE: an ARU_v1 inference graph laid out the way TensorFlow 1.x freezes it.

Builds the NodeDef list that ``ARU_v1_CNN._create_aru_net`` (``/root/reference/article_separation/backbones/
ARU_v1.py:62-294``) + ``layers.py`` produce after ``convert_variables_to_constants``: Const + ``/read`` Identity per
variable, Conv2D / BiasAdd / Relu per layer, MaxPool / AvgPool, Conv2DBackpropInput with a Shape -> StridedSlice -> Pack
output-shape sub-graph for the transposed convolutions, ``upsample_simple`` as Conv2DBackpropInput with a splat
constant filter, ConcatV2 / Softmax / Split / Mul / AddN for the attention fusion, an ``output`` Softmax.

Options exercise what a net exported by ANOTHER project may look like: arbitrary scope names (``rename``), inference
batch normalisation after the bias as FusedBatchNorm nodes or as the folded Mul / Add pair (``bn``), AddV2 instead of
Add, bias as Add instead of BiasAdd, no class softmax, no ``/read`` identities.
The graph is serialised by google.protobuf (tests/tf_graphdef_proto.py), not by the product's own encoder.
"""
import numpy as np

import tf_graphdef_proto as tp

F32 = tp.DType(tp.DT_FLOAT)


class AruGraphBuilder:
    def __init__(self, ns, weights, cfg, rename=None, bn=None, bn_style="fused", add_op="Add", bias_op="BiasAdd",
                 read_identities=True, tensor_encoding="content", output_softmax=True, activation="Relu"):
        self.ns, self.w, self.cfg = ns, weights, cfg
        self.rename = rename or (lambda s: s)
        self.bn, self.bn_style = bn or {}, bn_style
        self.add_op, self.bias_op, self.read_identities = add_op, bias_op, read_identities
        self.enc, self.output_softmax, self.activation = tensor_encoding, output_softmax, activation
        self.nodes, self.uid = [], 0
        self.leak = 0.1                                     # layers.py:10

    # ---- node helpers ------------------------------------------------------------------------------------------
    def add(self, name, op, inputs=(), **attrs):
        self.nodes.append(tp.node(self.ns, name, op, inputs, tensor_encoding=self.enc, **attrs))
        return name

    def fresh(self, base):
        self.uid += 1
        return f"{base}_{self.uid}"

    def const(self, name, arr, encoding=None):
        arr = np.asarray(arr)
        self.nodes.append(tp.node(self.ns, name, "Const", (), tensor_encoding=encoding or self.enc,
                                  dtype=tp.DType(tp._NP2DT[arr.dtype]), value=arr))
        return name

    def variable(self, scope, leaf):
        """Const named like the variable (+ Identity '/read'); created once, reused by every scale"""
        key = f"{scope}/{leaf}"
        name = self.rename(key)
        if not any(n.name == name for n in self.nodes):
            self.const(name, self.w[key])
            if self.read_identities:
                self.add(name + "/read", "Identity", [name], T=F32)
        return name + "/read" if self.read_identities else name

    # ---- layers (layers.py:191-247, 342-367, 526-544, 716-720) --------------------------------------------------
    def conv(self, x, scope, tag, act=True):
        s = self.rename(scope)
        y = self.add(f"{s}/{tag}conv", "Conv2D", [x, self.variable(scope, "weights")], T=F32, strides=[1, 1, 1, 1],
                     padding="SAME", data_format="NHWC", dilations=[1, 1, 1, 1], use_cudnn_on_gpu=True)
        return self._tail(y, scope, s, tag, "biases", act)

    def deconv(self, x, like, scope, tag, cout):
        s = self.rename(scope)
        shp = self.add(f"{s}/{tag}Shape", "Shape", [like], T=F32, out_type=tp.DType(tp.DT_INT32))
        parts = []
        for i in range(3):
            b, e, st = (self.const(self.fresh(f"{s}/{tag}ss"), np.array([v], np.int32)) for v in (i, i + 1, 1))
            parts.append(self.add(f"{s}/{tag}strided_slice_{i}", "StridedSlice", [shp, b, e, st], T=tp.DType(tp.DT_INT32),
                                  Index=tp.DType(tp.DT_INT32), shrink_axis_mask=1, begin_mask=0, end_mask=0))
        parts.append(self.const(self.fresh(f"{s}/{tag}nf"), np.array(cout, np.int32)))
        out_shape = self.add(f"{s}/{tag}stack", "Pack", parts, N=4, T=tp.DType(tp.DT_INT32), axis=0)
        y = self.add(f"{s}/{tag}conv", "Conv2DBackpropInput", [out_shape, self.variable(scope, "weights"), x], T=F32,
                     strides=[1, 2, 2, 1], padding="SAME", data_format="NHWC", dilations=[1, 1, 1, 1])
        return self._tail(y, scope, s, tag, "bias", True)

    def _tail(self, y, scope, s, tag, bias_leaf, act):
        b = self.variable(scope, bias_leaf)
        if self.bias_op == "BiasAdd":
            y = self.add(f"{s}/{tag}preActivation", "BiasAdd", [y, b], T=F32, data_format="NHWC")
        else:
            y = self.add(f"{s}/{tag}preActivation", self.bias_op, [y, b], T=F32)
        if scope in self.bn:
            gamma, beta, mean, var, eps = self.bn[scope]
            first = not any(n.name == f"{s}/bn/gamma" for n in self.nodes)
            if self.bn_style == "fused":
                if first:
                    for leaf, v in (("gamma", gamma), ("beta", beta), ("moving_mean", mean), ("moving_variance", var)):
                        self.const(f"{s}/bn/{leaf}", v.astype(np.float32))
                y = self.add(f"{s}/{tag}bn/FusedBatchNorm", "FusedBatchNormV3",
                             [y] + [f"{s}/bn/{leaf}" for leaf in ("gamma", "beta", "moving_mean", "moving_variance")],
                             T=F32, U=F32, epsilon=float(eps), is_training=False, data_format="NHWC")
            else:                                          # folded by a graph transform: y * scale + shift
                scale = (gamma / np.sqrt(var + np.float32(eps))).astype(np.float32)
                shift = (beta - mean * scale).astype(np.float32)
                if first:
                    self.const(f"{s}/bn/gamma", scale)
                    self.const(f"{s}/bn/shift", shift)
                y = self.add(f"{s}/{tag}bn/mul", "Mul", [y, f"{s}/bn/gamma"], T=F32)
                y = self.add(f"{s}/{tag}bn/add", self.add_op, [y, f"{s}/bn/shift"], T=F32)
        if act:
            y = self.act(y, f"{s}/{tag}activation")
        return y

    def act(self, y, name):
        """the graph's activation (ARU_v1.py:70-75) as TF writes it: Relu / Elu ops, or layers.leaky_relu's composite
        maximum(0, x) + leak * minimum(0, x) (layers.py:10-30).  `activation` != "Relu" (constructor) overrides the op type."""
        kind = getattr(self.cfg, "activation_name", "relu")
        if self.activation != "Relu" or kind == "relu":
            return self.add(name, self.activation, [y], T=F32)
        if kind == "elu":
            return self.add(name, "Elu", [y], T=F32)
        zero = self.const(self.fresh(name + "/zero"), np.array(0.0, np.float32))
        leak = self.const(self.fresh(name + "/leak"), np.array(self.leak, np.float32))
        hi = self.add(name + "/Maximum", "Maximum", [zero, y], T=F32)
        lo = self.add(name + "/Minimum", "Minimum", [zero, y], T=F32)
        lo = self.add(name + "/Mul", "Mul", [leak, lo], T=F32)
        return self.add(name + "/add", self.add_op, [hi, lo], T=F32)

    def pool(self, x, name, op):
        return self.add(name, op, [x], T=F32, ksize=[1, 2, 2, 1], strides=[1, 2, 2, 1], padding="SAME", data_format="NHWC")

    def upsample(self, x, like_or_shape, name, up, c, shape_is_tensor):
        filt = self.const(name + "/Const", np.full((up, up, c, c), 1.0, np.float32), encoding="splat")
        shape = like_or_shape if shape_is_tensor else self.add(name + "/Shape", "Shape", [like_or_shape], T=F32,
                                                              out_type=tp.DType(tp.DT_INT32))
        return self.add(name + "/conv2d_transpose", "Conv2DBackpropInput", [shape, filt, x], T=F32, strides=[1, up, up, 1],
                        padding="SAME", data_format="NHWC")

    # ---- ARU_v1 -------------------------------------------------------------------------------------------------
    def att_cnn(self, x, sc):
        tag = f"s{sc}_" if sc else ""
        base = "aru_net/attMapG/attPart"
        for i in range(1, 5):
            x = self.conv(x, f"{base}/conv{i}", tag, act=True)
            if i < 4:
                x = self.pool(x, self.rename(f"{base}/{tag}pool{i}"), "MaxPool")
        return x

    def res_block(self, x, scope, tag):
        if not getattr(self.cfg, "use_residual", True):     # graph 'U' (ARU_v1.py:228-233): conv1 + conv2, both activated
            return self.conv(self.conv(x, f"{scope}/conv1", tag, act=True), f"{scope}/conv2", tag, act=True)
        R = self.cfg.res_depth
        orig = self.conv(x, f"{scope}/conv1", tag, act=False)
        # ARU_v1.py:214: layers.relu, whatever the graph's activation is
        x = self.add(self.rename(f"{scope}/{tag}activation"), self.activation if self.activation != "Relu" else "Relu", [orig], T=F32)
        for r in range(R):
            x = self.conv(x, f"{scope}/convR_{r}", tag, act=r < R - 1)
        x = self.add(self.rename(f"{scope}/{tag}add"), self.add_op, [x, orig], T=F32)
        return self.act(x, self.rename(f"{scope}/{tag}activation_out"))

    def det_cnn(self, x, sc):
        tag = f"s{sc}_" if sc else ""
        n, base = self.cfg.scale_space_num, "aru_net/featMapG"
        skips = []
        for l in range(n):
            x = self.res_block(x, f"{base}/unet_down_{l}", tag)
            skips.append(x)
            if l < n - 1:
                x = self.pool(x, self.rename(f"{base}/unet_down_{l}/{tag}pool"), "MaxPool")
        for l in range(n - 2, -1, -1):
            scope = f"{base}/unet_up_{l}"
            d = self.deconv(x, skips[l], f"{scope}/deconv", tag, self.cfg.feat(l))
            axis = self.const(self.fresh(self.rename(f"{scope}/{tag}concat/axis")), np.array(3, np.int32))
            cat = self.add(self.rename(f"{scope}/{tag}concat"), "ConcatV2", [skips[l], d, axis], N=2, T=F32,
                           Tidx=tp.DType(tp.DT_INT32))
            x = self.res_block(cat, scope, tag)
        return x

    def build(self):
        cfg, ren = self.cfg, self.rename
        img = self.add("inImg", "Placeholder", dtype=F32, shape=tp.Shape(-1, -1, -1, cfg.channels))
        scales = [img]
        n_sc = cfg.num_scales_att if cfg.use_attention else 1
        for sc in range(1, n_sc):
            scales.append(self.pool(scales[-1], ren(f"aru_net/attMapG/AvgPool_{sc}"), "AvgPool"))
        att = []
        if cfg.use_attention:
            up = 8
            for sc in range(n_sc):
                a = self.att_cnn(scales[sc], sc)
                att.append(self.upsample(a, img, ren(f"aru_net/attMapG/up_{sc}"), up, 1, shape_is_tensor=False))
                up *= 2
        det = [self.det_cnn(img, 0)]
        if cfg.use_attention:
            ishape = self.add(ren("aru_net/misc/Shape"), "Shape", [img], T=F32, out_type=tp.DType(tp.DT_INT32))
            parts = []
            for i in range(3):
                b, e, st = (self.const(self.fresh(ren("aru_net/misc/ss")), np.array([v], np.int32)) for v in (i, i + 1, 1))
                parts.append(self.add(ren(f"aru_net/misc/strided_slice_{i}"), "StridedSlice", [ishape, b, e, st],
                                      T=tp.DType(tp.DT_INT32), Index=tp.DType(tp.DT_INT32), shrink_axis_mask=1))
            parts.append(self.const(ren("aru_net/misc/stack/3"), np.array(cfg.feat_root, np.int32)))
            o_shape = self.add(ren("aru_net/misc/stack"), "Pack", parts, N=4, T=tp.DType(tp.DT_INT32), axis=0)
            up = 1
            for sc in range(1, n_sc):
                up *= 2
                d = self.det_cnn(scales[sc], sc)
                det.append(self.upsample(d, o_shape, ren(f"aru_net/featMapG/up_{sc}"), up, cfg.feat_root, shape_is_tensor=True))
            axis = self.const(ren("aru_net/logit/concat/axis"), np.array(3, np.int32))
            cat = self.add(ren("aru_net/logit/concat"), "ConcatV2", att + [axis], N=n_sc, T=F32, Tidx=tp.DType(tp.DT_INT32))
            sm = self.add(ren("aru_net/logit/Softmax"), "Softmax", [cat], T=F32)
            sdim = self.const(ren("aru_net/logit/split/split_dim"), np.array(3, np.int32))
            sp = self.add(ren("aru_net/logit/split"), "Split", [sdim, sm], T=F32, num_split=n_sc)
            prods = [self.add(ren(f"aru_net/logit/Mul_{sc}"), "Mul", [det[sc], sp if sc == 0 else f"{sp}:{sc}"], T=F32)
                     for sc in range(n_sc)]
            fmap = self.add(ren("aru_net/logit/AddN"), "AddN", prods, N=n_sc, T=F32)
        else:
            fmap = det[0]
        logits = self.conv(fmap, "aru_net/logit/class", "", act=False)
        logits = self.add(ren("aru_net/logit/logits"), "Identity", [logits], T=F32)
        if self.output_softmax:
            self.add("output", "Softmax", [logits], T=F32)
        else:
            self.add("output", "Identity", [logits], T=F32)
        return tp.graphdef(self.ns, self.nodes).SerializeToString()


def build_aru_pb(weights, cfg, packed_repeated=True, **kw):
    return AruGraphBuilder(tp.build_messages(packed_repeated), weights, cfg, **kw).build()
