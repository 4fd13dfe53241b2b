"""DLDKD: host-side mirror of the reference's model class for the scoring + distillation hot path.

Same constructor `(config, opt)`, same 74 state-dict keys, same public methods and return shapes as
reference method/model.py:13-387, so checkpoints and calling code carry over; every tensor op below the
method boundary is a HIP kernel from libdldkd_hip.so (no ATen math on the path, no CPU fallback).
"""
import copy
import types

import torch
import torch.nn as nn

from . import native, ops, scoring
from .model_components import BertAttention, LinearLayer, TrainablePositionalEncoding


def _cfg_get(config, name, default=None):
    if isinstance(config, dict):
        return config.get(name, default)
    return getattr(config, name, default)


def _enc_cfg(hidden, drop, heads):
    return types.SimpleNamespace(hidden_size=hidden, intermediate_size=hidden, hidden_dropout_prob=drop,
                                 num_attention_heads=heads, attention_probs_dropout_prob=drop)


class DLDKD(nn.Module):
    def __init__(self, config, opt):
        super().__init__()
        self.config = config
        self.double_branch = opt.double_branch
        c = lambda k, d=None: _cfg_get(config, k, d)   # noqa: E731

        def towers(prefix, hidden):
            # module attribute names = the reference's (model.py:20-61) -> identical state-dict keys
            setattr(self, prefix + "query_pos_embed", TrainablePositionalEncoding(c("max_desc_l"), hidden, c("input_drop")))
            setattr(self, prefix + "query_input_proj", LinearLayer(c("query_input_size"), hidden, layer_norm=True,
                                                                   dropout=c("input_drop"), relu=True))
            setattr(self, prefix + "query_encoder", BertAttention(_enc_cfg(hidden, c("drop"), c("n_heads"))))
            setattr(self, prefix + "modular_vector_mapping", nn.Linear(hidden, 1, bias=False))
            setattr(self, prefix + "visual_pos_embed", TrainablePositionalEncoding(c("max_ctx_l"), hidden, c("input_drop")))
            setattr(self, prefix + "visual_input_proj", LinearLayer(c("visual_input_size"), hidden, layer_norm=True,
                                                                    dropout=c("input_drop"), relu=True))
            setattr(self, prefix + "visual_encoder", BertAttention(_enc_cfg(hidden, c("drop"), c("n_heads"))))
            setattr(self, prefix + "out_mapping_linear", nn.Linear(hidden, hidden))

        towers("", c("inheritance_hidden"))
        if self.double_branch:
            towers("exp_", c("exploration_hidden"))

        self.weight = 1
        self.kl_intra_weight = opt.kl_intra_weight
        self.inher_nce_weight = opt.inher_nce_weight
        self.explore_nce_weight = opt.explore_nce_weight
        self.collection = opt.collection
        self.alpha = opt.alpha
        self.belta = opt.belta
        # label_style is read by forward() but never put into model_config by the reference's train.py
        # (SURVEY quirk table): accept it from either place.
        self.label_style = c("label_style", getattr(opt, "label_style", "soft"))
        self.reset_parameters()

    # ------------------------------------------------------------------ init / config
    def reset_parameters(self):
        """N(0, initializer_range) for Linear/Embedding weights, LayerNorm = (1, 0), biases 0 (model.py:80-93)."""
        std = _cfg_get(self.config, "initializer_range", 0.02)
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Embedding)):
                m.weight.data.normal_(mean=0.0, std=std)
            elif isinstance(m, nn.LayerNorm):
                m.bias.data.zero_()
                m.weight.data.fill_(1.0)
            if isinstance(m, nn.Linear) and m.bias is not None:
                m.bias.data.zero_()

    def set_hard_negative(self, use_hard_negative, hard_pool_size):
        if isinstance(self.config, dict):
            self.config["use_hard_negative"] = use_hard_negative
            self.config["hard_pool_size"] = hard_pool_size
        else:
            self.config.use_hard_negative = use_hard_negative
            self.config.hard_pool_size = hard_pool_size

    # ------------------------------------------------------------------ encoders (model.py:199-258)
    @staticmethod
    def encode_input(feat, mask, input_proj_layer, encoder_layer, pos_embed_layer):
        feat = feat.float().contiguous()
        h = input_proj_layer(feat)
        h = pos_embed_layer(h)
        if mask is not None:
            mask = mask.float().unsqueeze(1)
        return encoder_layer(h, mask)

    def encode_context(self, frame_video_feat, video_mask=None):
        out = []
        for pre in ("", "exp_") if self.double_branch else ("",):
            h = self.encode_input(frame_video_feat, video_mask, getattr(self, pre + "visual_input_proj"),
                                  getattr(self, pre + "visual_encoder"), getattr(self, pre + "visual_pos_embed"))
            lin = getattr(self, pre + "out_mapping_linear")
            out.append(ops.linear(h, lin.weight, lin.bias))
        return (out[0], out[1]) if self.double_branch else (out[0], None)

    def get_modularized_queries(self, encoded_query, query_mask, inheritance=False):
        w = (self.modular_vector_mapping if inheritance else self.exp_modular_vector_mapping).weight
        return ops.modpool(encoded_query.contiguous(), query_mask.float().contiguous(), w.reshape(-1).contiguous())

    def encode_query(self, query_feat, query_mask):
        if query_feat.dim() == 2:            # the reference's collate .squeeze() drops a batch of one
            query_feat, query_mask = query_feat.unsqueeze(0), query_mask.reshape(1, -1)
        out = []
        for pre in ("", "exp_") if self.double_branch else ("",):
            h = self.encode_input(query_feat, query_mask, getattr(self, pre + "query_input_proj"),
                                  getattr(self, pre + "query_encoder"), getattr(self, pre + "query_pos_embed"))
            out.append(self.get_modularized_queries(h, query_mask, inheritance=(pre == "")))
        return (out[0], out[1]) if self.double_branch else (out[0], None)

    # ------------------------------------------------------------------ scoring (model.py:307-350)
    @staticmethod
    def pooled_scores(queries, galleries, mask=None, normalize=True, w=(0.7, 0.3), want_branches=True):
        """All-pairs key-clip max-pooled scores on the bf16 MFMA scorer (K1).

        queries: list of (Nq, 384); galleries: list of (Nv, L, 384) or a scoring.PackedGallery.
        Returns (fused, s0, s1)."""
        pq = scoring.pack_queries(list(queries), normalize=normalize)
        pg = galleries if isinstance(galleries, scoring.PackedGallery) else scoring.pack_gallery(list(galleries), mask, normalize)
        return scoring.simpool_eval(pq, pg, w=w, want_fused=True, want_branches=want_branches)

    def get_pred_from_raw_query(self, query_feat, query_mask, ctx_info):
        """encode_query -> get_sim_scores(inher) -> get_sim_scores(explore) (eval.py:200-208).
        Returns (inher_scores, explore_scores), each (Nq, Nv)."""
        q_inh, q_exp = self.encode_query(query_feat, query_mask)
        pg = ctx_info.get("_packed")
        if pg is None:
            gs = [ctx_info["inher_frame_feat"]] + ([ctx_info["explore_frame_feat"]] if self.double_branch else [])
            pg = scoring.pack_gallery(gs, ctx_info["video_mask"])
        qs = [q_inh] + ([q_exp] if self.double_branch else [])
        _, s0, s1 = self.pooled_scores(qs, pg, want_branches=True)
        return s0, s1
