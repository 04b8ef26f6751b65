"""Model hyper-parameters of the two hot-path nets.

Defaults follow the reference's defaults:
  * ARU-Net: ``article_separation/backbones/ARU_v1.py:35-43`` (graph 'ARU' switches the
    attention branch on, ``ARU_v1.py:94-98``).
  * GNN: ``gnn/model/graph/graph_gnn.py:19-25``, ``message_fn_chunk.py:13-40``,
    ``update_fn_lstm.py:12-19``, ``gnn/trainer/trainer_rel.py:15-17``.
"""
from dataclasses import dataclass, field, asdict
from typing import List


@dataclass
class AruConfig:
    graph: str = "ARU"            # 'U' (plain conv1 + conv2 blocks), 'RU' (residual blocks) or 'ARU' (+ attention), ARU_v1.py:92-97
    channels: int = 1             # image channels (the attention branch needs 1)
    n_classes: int = 2
    feat_root: int = 8
    scale_space_num: int = 5
    res_depth: int = 3
    num_scales_att: int = 3
    filter_size: int = 3
    pool_size: int = 2
    activation_name: str = "relu"  # ARU_v1.py:43,70-75: 'relu', 'elu' or 'leaky' (leak 0.1, layers.py:10-30)
    mvn: bool = False
    apply_softmax: bool = True    # export-time class softmax -> 'output:0'
    # The engine's arithmetic -- an engine option, not a property of the weights:
    #   'f32s' (default since round 5): fp32 tensors, fp32 accumulation, fp32 results; the products of the convolutions with >= 12 input
    #           channels as six bf16 x bf16 partial products of the exact three-way bfloat16 split of both factors (dropped terms
    #           <= 2^-23 |x w|); level 0, first / last layers and deconvolutions on the fp32 FMA / MFMA.  Held to the fp32 parity gates.
    #   'f32':  every product on the fp32 pipes (v_mfma_f32_16x16x4_f32, Winograd F(2x2,3x3) from 32 channels, v_pk_fma_f32 at level 0)
    #   'bf16': BASELINE configs[4] "bf16 convs": bf16 tensors and MFMA operands, fp32 accumulation (its own, wider gates)
    compute_dtype: str = "f32s"

    @property
    def use_attention(self) -> bool:
        return "ARU" in self.graph

    @property
    def use_residual(self) -> bool:
        return "RU" in self.graph                 # ARU_v1.py:94-95 ('U' alone: conv1 + conv2 blocks)

    @property
    def activation_code(self) -> int:
        """asep_aru_cfg.activation"""
        try:
            return {"relu": 0, "elu": 1, "leaky": 2}[self.activation_name]
        except KeyError:
            raise ValueError(f"activation_name must be 'relu', 'elu' or 'leaky' (ARU_v1.py:43), got {self.activation_name!r}")

    def feat(self, level: int) -> int:
        return self.feat_root * (self.pool_size ** level)

    def to_dict(self):
        return asdict(self)


@dataclass
class GnnConfig:
    node_feature_dim: int = 7         # after masking (vn7e2: 7 of 15, nets/README.md:24-27)
    edge_feature_dim: int = 2
    num_transition_steps: int = 3
    hidden_dim: int = 32              # hidden_node_feature_dim
    interaction_dim: int = 32         # interaction_feature_dim
    interaction_hidden: List[int] = field(default_factory=lambda: [32])
    classifier_hidden: List[int] = field(default_factory=lambda: [64, 32])
    num_classes: int = 2
    undirected_graph: bool = True
    # graph_gnn.py:20,102-109: > 0: the (concatenated) node features go through ff_layer(tanh) to this width before the GNN
    compress_node_feature_dim: int = 0
    # graph_gnn.py:23,158-166: 'hidden', 'add_final_hidden_and_input' (h += x W) or 'concat_final_hidden_and_input' ([h | x]); x = the
    # node features as fed
    output_type: str = "hidden"
    # message_fn_chunk.py:35-41: learned attention over the in-edges instead of the degree-normalised sum (the default, False)
    use_attention: bool = False
    num_attention_heads: int = 1
    multihead_attention_merge_type: str = "concat"        # 'concat' (x_dim = interaction_dim // heads) or 'average'
    attention_hidden: List[int] = field(default_factory=lambda: [16])   # num_hidden_units_attention_fct
    aggregation_type: str = "sum"     # message_fn_chunk.py:16,57-62: 'sum' (tf.sparse.reduce_sum) or 'max' (tf.sparse.reduce_max)
    # update_fn_lstm.py:13-16,43-50: which tensors the four gates read besides x
    incorporate_hidden_features_in_update: bool = True
    incorporate_node_input_features_in_update: bool = True
    # graph_relation.py:141-172 assign_visual_features_to_edges: ROI features of every interaction's region, compressed like the
    # nodes' (same layer_compressed_dim per feature map), appended to the fed edge features
    visual_edges: bool = False
    # visual branch (GraphRelation image_input); 0 maps -> disabled
    visual_dims: List[int] = field(default_factory=list)   # layer_compressed_dim per feature map
    # feature_map_generation_params from_layer (layer_depth -1): backbone end points, e.g. scale_0_unet_up_2_conv
    visual_layers: List[str] = field(default_factory=list)
    mvn: bool = False                 # GraphRelation flag: per-image standardisation of the fed image
    backbone: dict = field(default_factory=dict)            # AruConfig fields of the ARU_v1 backbone (graph 'RU')

    def backbone_cfg(self) -> "AruConfig":
        kw = dict(graph="RU", apply_softmax=False)
        kw.update(self.backbone)
        kw["mvn"] = bool(kw.get("mvn", False) or self.mvn)
        return AruConfig(**kw)

    def visual_channels(self) -> List[int]:
        """Channels of the selected end points: feat_root * 2^level (ARU_v1.py:208-292)."""
        import re
        bc = self.backbone_cfg()
        out = []
        for name in self.visual_layers:
            m = re.fullmatch(r"scale_\d+_unet_(down|up)_(\d+)_(conv|deconv)", name)
            if not m or int(m.group(2)) >= bc.scale_space_num:
                raise ValueError(f"'{name}' is not a feature-map end point of the ARU_v1 backbone")
            out.append(bc.feat(int(m.group(2))))
        return out

    @property
    def heads(self) -> int:
        """message_fn_chunk.py:167-169: one head without attention"""
        return self.num_attention_heads if self.use_attention else 1

    @property
    def head_interaction_dim(self) -> int:
        """message_fn_chunk.py:68-72: x_dim of one head"""
        if self.use_attention and self.multihead_attention_merge_type == "concat":
            return self.interaction_dim // self.num_attention_heads
        return self.interaction_dim

    @property
    def output_type_code(self) -> int:
        """asep_gnn_cfg.output_type"""
        try:
            return {"hidden": 0, "add_final_hidden_and_input": 1, "concat_final_hidden_and_input": 2}[self.output_type]
        except KeyError:
            raise ValueError(f"output_type {self.output_type!r} (graph_gnn.py:23: hidden, add_final_hidden_and_input, "
                             f"concat_final_hidden_and_input)")

    @property
    def classifier_node_dim(self) -> int:
        """width of the node vectors the pair classifier concatenates"""
        return self.hidden_dim + (self.u_in_dim if self.output_type_code == 2 else 0)

    @property
    def u_in_dim(self) -> int:
        """width of the node features as fed (geometric + compressed visual dims)"""
        return self.node_feature_dim + sum(self.visual_dims)

    @property
    def u_dim(self) -> int:
        """width of the node features the message / update functions see"""
        return self.compress_node_feature_dim if self.compress_node_feature_dim > 0 else self.u_in_dim

    @property
    def visual_edge_dim(self) -> int:
        return sum(self.visual_dims) if self.visual_edges else 0

    @property
    def edge_in_dim(self) -> int:
        """width of the edge features the message function sees: fed (geometric) + compressed visual edge dims"""
        return self.edge_feature_dim + self.visual_edge_dim

    @property
    def aggregation_code(self) -> int:
        """asep_gnn_cfg.aggregation_type"""
        try:
            return {"sum": 0, "max": 1}[self.aggregation_type]
        except KeyError:
            raise ValueError(f"aggregation_type {self.aggregation_type!r} (message_fn_chunk.py:57-62: 'sum' or 'max')")

    @property
    def message_in_dim(self) -> int:
        return 4 * self.u_dim + self.edge_in_dim + 4 * self.hidden_dim

    @property
    def update_in_dim(self) -> int:
        return (self.message_out_dim + (self.hidden_dim if self.incorporate_hidden_features_in_update else 0)
                + (self.u_dim if self.incorporate_node_input_features_in_update else 0))

    @property
    def message_out_dim(self) -> int:
        """width of x (message_fn_chunk.py:229-241): heads * x_dim for 'concat', x_dim for 'average' / no attention"""
        if self.use_attention and self.multihead_attention_merge_type == "concat":
            return self.heads * self.head_interaction_dim
        return self.interaction_dim

    def to_dict(self):
        return asdict(self)
