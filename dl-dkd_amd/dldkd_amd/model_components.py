"""Parameter containers for the encoder towers.

Module / parameter names reproduce the reference's state dict (method/model_components.py:269-312,
339-353,379-450) so its checkpoints load unchanged; the forward passes call the HIP kernels in
libdldkd_hip.so through `ops` instead of ATen.
"""
import torch
import torch.nn as nn

from . import functional as F_
from . import ops


class TrainablePositionalEncoding(nn.Module):
    """Learned position rows + LayerNorm (+ dropout in training).  Keys: position_embeddings.weight,
    LayerNorm.{weight,bias} (model_components.py:269-284)."""

    def __init__(self, max_position_embeddings, hidden_size, dropout=0.1):
        super().__init__()
        self.position_embeddings = nn.Embedding(max_position_embeddings, hidden_size)
        self.LayerNorm = nn.LayerNorm(hidden_size)
        self.dropout = nn.Dropout(dropout)

    def forward(self, input_feat):
        L = input_feat.shape[1]
        if L > self.position_embeddings.num_embeddings:
            raise IndexError(f"sequence length {L} exceeds {self.position_embeddings.num_embeddings} positions")
        pos = self.position_embeddings.weight[:L]
        return F_.layernorm(input_feat, self.LayerNorm.weight, self.LayerNorm.bias, add=pos, add_mod=L,
                            p_drop=self.dropout.p, training=self.training)            # LayerNorm + Dropout: one kernel


class LinearLayer(nn.Module):
    """LayerNorm -> Dropout -> Linear -> ReLU.  Keys: LayerNorm.*, net.1.* (model_components.py:294-312)."""

    def __init__(self, in_hsz, out_hsz, layer_norm=True, dropout=0.1, relu=True):
        super().__init__()
        self.relu = relu
        self.layer_norm = layer_norm
        if layer_norm:
            self.LayerNorm = nn.LayerNorm(in_hsz)
        self.net = nn.Sequential(nn.Dropout(dropout), nn.Linear(in_hsz, out_hsz))

    def forward(self, x, row_mask=None, grad_premasked=False):
        """row_mask (optional, training): the (n, L) mask of the padded batch x (n, L, K) - rows of the padding may then be skipped
        (their outputs are don't-care values nothing downstream of the towers reads).  grad_premasked: the consumer's backward pass
        applies this layer's ReLU mask itself (functional._TowerTrain)."""
        lin = self.net[1]
        if self.layer_norm and F_.in_proj_train_ok(x, lin.weight):
            # training on raw features, throughput mode: one autograd node whose backward pass skips the input gradient
            return F_.in_proj_train(x, self.LayerNorm.weight, self.LayerNorm.bias, lin.weight, lin.bias, self.net[0].p,
                                    self.training, relu=self.relu, row_mask=row_mask, grad_premasked=grad_premasked)
        if self.layer_norm:
            x = F_.layernorm(x, self.LayerNorm.weight, self.LayerNorm.bias, p_drop=self.net[0].p, training=self.training)
        else:
            x = F_.dropout(x, self.net[0].p, self.training)
        lin = self.net[1]
        return F_.linear(x, lin.weight, lin.bias, relu=self.relu)


class BertSelfAttention(nn.Module):
    """Keys: query/key/value.{weight,bias} (model_components.py:379-391)."""

    def __init__(self, config):
        super().__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)" % (
                config.hidden_size, config.num_attention_heads))
        if config.hidden_size != ops.HIDDEN or config.num_attention_heads != 4:
            raise ValueError("the gfx950 attention kernel is specialised for hidden 384 = 4 heads x 96")
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = config.hidden_size // config.num_attention_heads
        self.all_head_size = config.hidden_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)

    def fused_qkv(self):
        w = torch.cat([self.query.weight, self.key.weight, self.value.weight], 0).contiguous()
        b = torch.cat([self.query.bias, self.key.bias, self.value.bias], 0).contiguous()
        return w, b

    def forward(self, query_states, key_states, value_states, attention_mask=None):
        if not (query_states is key_states and key_states is value_states):
            raise NotImplementedError("only self-attention is on the DL-DKD path (BertAttention.forward, :351)")
        if ops.rows_kernel_ok(query_states):                      # inference, throughput mode: full-row bf16 kernel
            if getattr(self, "_packed_qkv", None) is None:
                self._packed_qkv = ops.PackedLinear([self.query, self.key, self.value])
            # q|k|v handed to the attention kernel as bf16 (it rounds them to bf16 itself otherwise: same numbers, half
            # the bytes of the largest activation of the layer)
            qkv = ops.linear_rows(query_states.float(), self._packed_qkv, out_bf16=True)
            mask = None
            if attention_mask is not None:
                mask = attention_mask.reshape(attention_mask.shape[0], -1).float().contiguous()
            return ops.attention(qkv.view(*query_states.shape[:-1], 3 * ops.HIDDEN), mask)
        elif ops.rows_x3_ok(query_states):                        # inference, parity mode: fp32-grade full-row kernel
            if getattr(self, "_packed_qkv_x3", None) is None:
                self._packed_qkv_x3 = ops.PackedLinearX3([self.query, self.key, self.value])
            qkv = ops.linear_rows_x3(query_states.float(), self._packed_qkv_x3)
        else:
            w, b = self.fused_qkv()
            qkv = F_.linear(query_states, w, b)                   # one GEMM for the three projections
        mask = None
        if attention_mask is not None:                            # (N, 1, L) as encode_input passes it (model.py:242)
            mask = attention_mask.reshape(attention_mask.shape[0], -1).contiguous()
        return F_.attention(qkv, mask, self.dropout.p, self.training)


class BertSelfOutput(nn.Module):
    """Keys: dense.*, LayerNorm.* (model_components.py:439-450)."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        if ops.rows_kernel_ok(hidden_states):
            if getattr(self, "_packed_dense", None) is None:
                self._packed_dense = ops.PackedLinear([self.dense])
            h = ops.linear_rows(hidden_states.float(), self._packed_dense)
        elif ops.rows_x3_ok(hidden_states):
            if getattr(self, "_packed_dense_x3", None) is None:
                self._packed_dense_x3 = ops.PackedLinearX3([self.dense])
            h = ops.linear_rows_x3(hidden_states.float(), self._packed_dense_x3)
        else:
            h = F_.linear(hidden_states, self.dense.weight, self.dense.bias)
        h = F_.dropout(h, self.dropout.p, self.training)
        return F_.layernorm(h, self.LayerNorm.weight, self.LayerNorm.bias, add=input_tensor, add_mod=0)


class BertAttention(nn.Module):
    """self-attention sub-layer with post-LN, no FFN (model_components.py:339-353)."""

    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)

    def forward(self, input_tensor, attention_mask=None):
        ctx = self.self(input_tensor, input_tensor, input_tensor, attention_mask)
        return self.output(ctx, input_tensor)
