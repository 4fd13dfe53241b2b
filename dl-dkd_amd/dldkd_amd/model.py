"""DLDKD: host-side mirror of the reference's model class for the scoring + distillation hot path.

Same constructor `(config, opt)`, same 74 state-dict keys, same public methods and return shapes as
reference method/model.py:13-387, so checkpoints and calling code carry over; every tensor op below the
method boundary is a HIP kernel from libdldkd_hip.so (no ATen math on the path, no CPU fallback).
"""
import types

import numpy as np
import torch
import torch.nn as nn

from . import functional as F_
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

        self.tower_streams = False         # training: the four towers on four streams (_encode_towers)
        self._side_streams = None
        self._tower_runner = None
        self._pre_ln_hook = None
        self._branch_runner = None
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
        # inference only: run the raw-feature input projections on the bf16 K4 kernel (LayerNorm folded, both
        # branches in one pass over the features) instead of the fp32 parity path
        self.fast_input_proj = False
        self.fused_parity_input_proj = True
        self._folded = {}
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
        if mask is not None and input_proj_layer.training and feat.dim() == 3 and mask.dim() == 2 and mask.shape == feat.shape[:2]:
            h = input_proj_layer(feat, row_mask=mask)        # training: the input projection skips the rows of the padding
            rg = F_.take_group_flags()
            if rg is not None and DLDKD.TOWER_SKIPS_PADDING:
                # ... and so does everything behind it: the flags stay current (ops.row_groups) until the tower's caller clears
                # them (_video_tower / encode_context)
                ops.set_row_groups(*rg)
        else:
            h = input_proj_layer(feat)
        h = pos_embed_layer(h)
        if mask is not None:
            mask = mask.float().unsqueeze(1)
        return encoder_layer(h, mask)

    def _fast_proj(self, kind, feat, groups=None):
        """K4: both branches' LayerNorm+Linear+ReLU of the raw features in one pass (inference only): bf16 when
        fast_input_proj is set, otherwise the parity-grade three-plane kernel."""
        if not self.fast_input_proj:
            if kind + "_x3" not in self._folded:
                self._folded[kind + "_x3"] = ops.FoldedInProjX3([getattr(self, pre + kind + "_input_proj") for pre in ("", "exp_")])
            return ops.in_proj_x3(feat.float().contiguous(), self._folded[kind + "_x3"])
        if kind not in self._folded:
            layers = [getattr(self, pre + kind + "_input_proj") for pre in (("", "exp_") if self.double_branch else ("",))]
            self._folded[kind] = ops.FoldedInProj(layers)
        return ops.in_proj_h16(feat.float().contiguous(), self._folded[kind], groups=groups)

    def _use_fast(self, feat):
        if self.training or torch.is_grad_enabled():
            return False
        if self.fast_input_proj:
            return feat.shape[-1] % 32 == 0
        # parity mode, inference: both branches' LayerNorm + Linear + ReLU in one fp32-grade pass (three bf16 planes per
        # operand: the numerics of the gemm_f32x3 path, 40 % less time than LayerNorm + two GEMMs)
        return (self.fused_parity_input_proj and self.double_branch and ops.gemm_precision() in ("fp32", "fp32x3")
                and ops.in_proj_x3_ok(feat.shape[-1]))

    @staticmethod
    def _encode_after_proj(h, mask, encoder_layer, pos_embed_layer):
        h = pos_embed_layer(h)
        if mask is not None:
            mask = mask.float().unsqueeze(1)
        return encoder_layer(h, mask)

    def _tower_packs(self, kind):
        """ops.TowerPack per branch for the fused per-sequence tower kernel ("visual": with the out mapping)."""
        key = kind + "_tower"
        if key not in self._folded:
            pres = ("", "exp_") if self.double_branch else ("",)
            if kind == "visual":
                self._folded[key] = [ops.TowerPack(getattr(self, pre + "visual_pos_embed"), getattr(self, pre + "visual_encoder"),
                                                   out_linear=getattr(self, pre + "out_mapping_linear")) for pre in pres]
            else:
                self._folded[key] = [ops.TowerPack(getattr(self, pre + "query_pos_embed"), getattr(self, pre + "query_encoder"),
                                                   mod_linear=getattr(self, pre + "modular_vector_mapping")) for pre in pres]
        return self._folded[key]

    def _upload_tables(self, tables_np, device):
        """Host-planned int32 tables of the fused gallery encode (slot table of the tower kernel, row-group table of the input
        projection) -> device in ONE upload through a ring of pinned slots (staging.PinnedRing): a pageable copy parks the host
        until the stream drains, a fresh pin_memory() per call is a multi-millisecond driver allocation whenever the host runs
        ahead of the GPU, and the copy engine costs two cross-engine hand-offs per upload (the slot is read by a kernel)."""
        from .staging import PinnedRing
        sizes = [t.nbytes for t in tables_np]
        nbytes = sum(sizes)
        ring = getattr(self, "_item_ring", None)
        if ring is None or ring.bufs[0].numel() < nbytes:
            ring = self._item_ring = PinnedRing(max(nbytes, 64 * 1024), device, slots=32)     # 2 MB pinned: an epoch's batches
        slot = ring.next()
        off = 0
        for t, nb in zip(tables_np, sizes):
            slot[off:off + nb].view(torch.int32).copy_(torch.from_numpy(t.reshape(-1)))
            off += nb
        dev = torch.empty(nbytes, dtype=torch.uint8, device=device)
        ring.upload(dev, by_kernel=True)
        out, off = [], 0
        for nb in sizes:
            out.append(dev[off:off + nb].view(torch.int32))
            off += nb
        return out

    def encode_context_into(self, packer, frame_video_feat, video_mask, lens_host=None):
        """Throughput-mode gallery encode straight into the scorer's resident bf16 gallery (scoring.GalleryPacker): input
        projection (K4) + ONE fused tower kernel per (32-row slot, branch) that L2-normalises and writes the packed rows - the
        fp32 (n, L, 384) tower outputs are never formed.  lens_host (numpy, optional): the lengths on the host; with them short
        videos share workgroups (ops.plan_tower_items).  Returns False when the fused path does not apply."""
        if not (self.fast_input_proj and self._use_fast(frame_video_feat)):
            return False
        items = groups = None
        n, L = frame_video_feat.shape[0], frame_video_feat.shape[1]
        if lens_host is not None:
            tables = [ops.plan_tower_items(lens_host)]
            if (L % 32 == 0 and self.double_branch and ops.INPROJ_KERNEL == "rows128"
                    and native.lib().dldkd_in_proj_h16_rows128_ok(frame_video_feat.shape[-1])):
                # the input projection visits only the 32-row groups that hold valid clips (the tower never reads the others)
                tables.append(ops.plan_row_groups(lens_host, L))
            tables = self._upload_tables(tables, frame_video_feat.device)
            items = tables[0].view(-1, 4)
            groups = tables[1] if len(tables) > 1 else None
        h0 = self._fast_proj("visual", frame_video_feat, groups=groups)
        if not ops.tower_seq_ok(h0[0]):
            return False
        lens = self._lens(video_mask, n, L, frame_video_feat.device)
        v0 = packer.reserve(n, L)
        ops.tower_seq(h0, self._tower_packs("visual"), lens, seq_rows=L, items=items, out_mode=1, gallery=packer.blobs, v0=v0,
                      Lp=packer.Lp, lens_out=packer.lens)
        return True

    def resident_encode_ok(self):
        """The gallery encode from resident bf16 rows (ops.ResidentRows -> K4b -> fused tower kernel): throughput mode, inference,
        two branches, a feature width the K4b kernel takes."""
        return bool(self.fast_input_proj and self.double_branch and not self.training and not torch.is_grad_enabled()
                    and ops.TOWER_SEQ and ops.gemm_precision() == "bf16" and ops.INPROJ_KERNEL == "rows128"
                    and ops.in_proj_rows_ok(self.visual_input_proj.net[1].weight.shape[1]))

    def encode_resident_into(self, packer, res):
        """Gallery encode of a resident feature table (eval.ResidentGallery) straight into the scorer's packed bf16 gallery: per
        chunk ONE input-projection launch over its rows (K4b) and ONE fused tower launch over its videos (K5, rows addressed by
        row0: the projection's output stays ragged)."""
        if "visual" not in self._folded:
            self._folded["visual"] = ops.FoldedInProj([self.visual_input_proj, self.exp_visual_input_proj])
        packs = self._tower_packs("visual")
        for (va, n, r0, r1, lens_d, row0_d, items) in res.chunks:
            y = ops.in_proj_resident(res.table, r0, r1, self._folded["visual"], out_h16=ops.RESIDENT_H0_H16)
            v0 = packer.reserve(n, int(res.lens_host[va:va + n].max(initial=0)))
            if items.shape[0]:
                # (a packer whose buffers are zero beyond every video's last 16-row tile: those rows are not written again)
                ops.tower_seq(y, packs, lens_d, seq_rows=0, row0=row0_d, items=items, out_mode=1, gallery=packer.blobs, v0=v0,
                              Lp=packer.Lp, lens_out=packer.lens,
                              skip_zero_rows=bool(getattr(packer, "zero_padded", False)) and ops.RESIDENT_H0_H16 and ops.SKIP_ZERO_ROWS)

    def encode_context(self, frame_video_feat, video_mask=None):
        out = []
        fast = self._fast_proj("visual", frame_video_feat) if self._use_fast(frame_video_feat) else None
        if fast is not None and self.fast_input_proj and ops.tower_seq_ok(fast[0]):
            # throughput mode: everything behind the input projection is one kernel (clips past a video's length included: they
            # are queries like any other, as in the reference)
            n, L = frame_video_feat.shape[0], frame_video_feat.shape[1]
            out = ops.tower_seq(fast, self._tower_packs("visual"), self._lens(video_mask, n, L, frame_video_feat.device), seq_rows=L)
            return (out[0], out[1]) if self.double_branch else (out[0], None)
        for bi, pre in enumerate(("", "exp_") if self.double_branch else ("",)):
            if fast is None and self._tower_fused_ok(frame_video_feat, video_mask):
                out.append(self._tower_fused(pre, "visual", frame_video_feat, video_mask))
                continue
            if fast is not None:
                h = self._encode_after_proj(fast[bi], video_mask, getattr(self, pre + "visual_encoder"),
                                            getattr(self, pre + "visual_pos_embed"))
            else:
                h = self.encode_input(frame_video_feat, video_mask, getattr(self, pre + "visual_input_proj"),
                                      getattr(self, pre + "visual_encoder"), getattr(self, pre + "visual_pos_embed"))
            lin = getattr(self, pre + "out_mapping_linear")
            if ops.rows_kernel_ok(h):
                if pre + "out_map" not in self._folded:
                    self._folded[pre + "out_map"] = ops.PackedLinear([lin])
                out.append(ops.linear_rows(h, self._folded[pre + "out_map"]))
            elif ops.rows_x3_ok(h):
                if pre + "out_map_x3" not in self._folded:
                    self._folded[pre + "out_map_x3"] = ops.PackedLinearX3([lin])
                out.append(ops.linear_rows_x3(h, self._folded[pre + "out_map_x3"]))
            else:
                out.append(F_.linear(h, lin.weight, lin.bias))
            ops.set_row_groups(None, 0)          # a training tower's row groups (encode_input) end with the tower
        return (out[0], out[1]) if self.double_branch else (out[0], None)

    def get_modularized_queries(self, encoded_query, query_mask, inheritance=False):
        w = (self.modular_vector_mapping if inheritance else self.exp_modular_vector_mapping).weight
        return F_.modpool(encoded_query, query_mask, w.reshape(-1))

    def encode_query(self, query_feat, query_mask):
        if query_feat.dim() == 2:            # the reference's collate .squeeze() drops a batch of one
            query_feat, query_mask = query_feat.unsqueeze(0), query_mask.reshape(1, -1)
        out = []
        fast = self._fast_proj("query", query_feat) if self._use_fast(query_feat) else None
        if fast is not None and self.fast_input_proj and ops.tower_seq_ok(fast[0]) and query_feat.shape[1] <= 32:
            # throughput mode: input projection, then ONE kernel for position LayerNorm, attention block and modular pooling, four
            # queries per workgroup
            n, lq = query_feat.shape[0], query_feat.shape[1]
            out = ops.tower_seq(fast, self._tower_packs("query"), self._lens(query_mask, n, lq, query_feat.device), seq_rows=lq,
                                out_mode=2)
            return (out[0], out[1]) if self.double_branch else (out[0], None)
        for bi, pre in enumerate(("", "exp_") if self.double_branch else ("",)):
            if fast is None and self._tower_fused_ok(query_feat, query_mask):
                out.append(self._query_tower(pre, query_feat, query_mask))
                continue
            if fast is not None:
                h = self._encode_after_proj(fast[bi], query_mask, getattr(self, pre + "query_encoder"),
                                            getattr(self, pre + "query_pos_embed"))
            else:
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

    # ------------------------------------------------------------------ fp32 scoring with clip-level output
    _LENS_CACHE = [None] * 6              # (mask storage pointer, mask._version, shape, stream) of the last masks -> their lengths

    @staticmethod
    def _lens(mask, nv, L, device):
        """Valid clips / words per sequence, int32.  GPU masks: one kernel, and the towers of one step ask for the same two masks
        (video, query) again and again - the last few results are kept (keyed on the mask's storage and version)."""
        if mask is None:
            return torch.full((nv,), L, dtype=torch.int32, device=device)
        if not mask.is_cuda or mask.dtype != torch.float32 or not mask.is_contiguous() or mask.dim() != 2:
            return (mask > 0).sum(1).to(torch.int32)
        # (per stream: the towers of a step run on their own streams / graphs and do not wait for each other's small kernels)
        key = (mask.data_ptr(), mask._version, tuple(mask.shape), torch.cuda.is_current_stream_capturing(),
               torch.cuda.current_stream(mask.device).cuda_stream)
        for ent in DLDKD._LENS_CACHE:
            if ent is not None and ent[0] == key and ent[2]() is mask:
                return ent[1]
        import weakref
        lens = torch.empty(mask.shape[0], dtype=torch.int32, device=mask.device)
        native.check(native.lib().dldkd_mask_lens_f32(native.ptr(mask), mask.shape[0], mask.shape[1], native.ptr(lens), native.stream()),
                     "mask_lens")
        DLDKD._LENS_CACHE = [(key, lens, weakref.ref(mask))] + DLDKD._LENS_CACHE[:5]
        return lens

    @staticmethod
    def _clip_level(q, ctx, mask, normalize):
        """fp32 (parity-grade) pooled + clip-level scores: returns (pooled (Nq,Nv), clip (Nq,Nv,L) masked)."""
        q = q.float()
        if q.dim() == 1:
            q = q.unsqueeze(0)
        ctx = ctx.float()
        lens = DLDKD._lens(mask, ctx.shape[0], ctx.shape[1], ctx.device)
        if normalize:
            q, ctx = F_.normalize(q), F_.normalize(ctx)
        pooled, clip, _ = F_.clip_pool(F_.clip_scores(q, ctx), lens)
        return pooled, clip

    @staticmethod
    def get_sim_scores(modularied_query, context_feat, mask=None):
        """Cosine scores: (pooled (Nq, Nv), clip-level (Nq, L, Nv)) as model.py:307-329; masked clips are
        exactly -1e10.  fp32-grade GEMM path (differentiable); the clip-level tensor is a permuted view of the
        (Nq, Nv, L) buffer the kernels use."""
        pooled, clip = DLDKD._clip_level(modularied_query, context_feat, mask, True)
        return pooled, clip.permute(0, 2, 1)

    @staticmethod
    def get_unnormalized_sim_scores(modularied_query, context_feat, mask=None):
        """Raw dot-product twin (model.py:331-350): pooled (Nq, Nv)."""
        return DLDKD._clip_level(modularied_query, context_feat, mask, False)[0]

    # ------------------------------------------------------------------ losses (model.py:166-197, 353-387)
    def compute_kl_loss(self, predict, target, cnn_mask, temp, mode="batch_score", query_labels=None):
        if mode != "frame_score":
            raise NotImplementedError("only mode='frame_score' is on the DL-DKD++ path (model.py:154-155)")
        # predict / target arrive as (Nq, L, Nv) like the reference's; the kernels use (Nq, Nv, L)
        Sp = predict.permute(0, 2, 1).contiguous()
        St = target.permute(0, 2, 1).contiguous()
        labels = torch.as_tensor(np.asarray(query_labels), dtype=torch.int32, device=Sp.device)
        return F_.kl_frame(Sp, St, labels, self._lens(cnn_mask, Sp.shape[1], Sp.shape[2], Sp.device), temp)

    def _draw_triplet(self, labels_np, nv):
        """The reference's CPU torch.randint calls, same order and arguments (model.py:366-368,377-380)."""
        hard = bool(_cfg_get(self.config, "use_hard_negative"))
        r_v2t = None
        if not hard:
            r = [int(torch.randint(0, int((labels_np != i).sum()), size=(1,))) for i in range(nv)]
            r_v2t = torch.tensor(r, dtype=torch.int32)
        hi = min(1 + _cfg_get(self.config, "hard_pool_size"), nv) if hard else nv
        r_t2v = torch.randint(1, hi, size=(len(labels_np),)).to(torch.int32)
        return hard, r_t2v, r_v2t

    def get_clip_triplet_loss(self, query_context_scores, labels, _staged=None):
        """model.py:353-387.  `_staged` (internal, train.GraphedTrainStep): (labels, r_t2v, r_v2t) already on the device -
        the CPU torch.randint draws were made by the caller, in the reference's order, before the captured step replays."""
        hard = bool(_cfg_get(self.config, "use_hard_negative"))
        if _staged is not None:
            lab, r_t2v, r_v2t = _staged
            return F_.triplet(query_context_scores, lab, r_t2v, None if hard else r_v2t, hard, _cfg_get(self.config, "margin"))
        labels_np = np.asarray(labels)
        dev = query_context_scores.device
        hard, r_t2v, r_v2t = self._draw_triplet(labels_np, query_context_scores.shape[1])
        lab = torch.as_tensor(labels_np, dtype=torch.int32, device=dev)
        return F_.triplet(query_context_scores, lab, r_t2v.to(dev), None if r_v2t is None else r_v2t.to(dev), hard,
                          _cfg_get(self.config, "margin"))

    # ------------------------------------------------------------------ training forward (model.py:100-163)
    def forward(self, batch):
        loss, parts = self.forward_tensors(batch)
        out = {"loss_overall": float(loss.detach())}          # the reference returns a Python float here (model.py:160)
        out.update(parts)
        return loss, out

    # ------------------------------------------------------------------ the four towers side by side (training)
    TOWER_SKIPS_PADDING = True

    def _tower_fused_ok(self, feat, mask):
        """Training, throughput mode: the tower behind the input projection as the fused row kernels (functional._TowerTrain)."""
        return (self.training and mask is not None and feat.dim() == 3 and mask.dim() == 2 and tuple(mask.shape) == tuple(feat.shape[:2])
                and F_.tower_train_ok(feat.is_cuda, feat.shape[1]))

    def _tower_fused(self, pre, kind, feat, mask):
        proj, pos, enc = (getattr(self, pre + kind + s) for s in ("_input_proj", "_pos_embed", "_encoder"))
        feat = feat.float().contiguous()
        n, L = feat.shape[0], feat.shape[1]
        if L > pos.position_embeddings.num_embeddings:
            raise IndexError(f"sequence length {L} exceeds {pos.position_embeddings.num_embeddings} positions")
        fused_proj = F_.in_proj_train_ok(feat, proj.net[1].weight) and proj.relu
        out_lin = getattr(self, pre + "out_mapping_linear") if kind == "visual" else None
        lens = None
        if F_.TOWER_PREPACK:
            # every weight operand of the tower - the projection's bf16 weight, the fragment packs - and the batch's sequence lengths
            # from ONE launch
            # ("mixed" precision: the projection's forward GEMM takes the fp32 weight - no bf16 cast of it)
            lens = F_.tower_prepack(proj.net[1].weight if (fused_proj and not F_.tower_train_mixed()) else None, enc.self.query.weight, enc.self.key.weight,
                                    enc.self.value.weight, enc.output.dense.weight, None if out_lin is None else out_lin.weight, mask=mask,
                                    mixed=(proj.net[1].weight if fused_proj else None, enc.self.query.bias, enc.self.key.bias, enc.self.value.bias))
        if lens is None:
            lens = self._lens(mask, n, L, feat.device)
        y0 = proj(feat, row_mask=mask, grad_premasked=True) if fused_proj else proj(feat)
        rg = F_.take_group_flags()
        flags = rg[0] if (rg is not None and self.TOWER_SKIPS_PADDING and rg[1] == n * L) else None
        # (the WHOLE position table goes in - the kernels read its first L rows: a [:L] view here costs a zeros + copy pair per tower
        # in autograd's slice backward)
        return F_.tower_train(y0, pos.position_embeddings.weight, pos.LayerNorm.weight, pos.LayerNorm.bias,
                              (enc.self.query, enc.self.key, enc.self.value), enc.output.dense, enc.output.LayerNorm.weight,
                              enc.output.LayerNorm.bias, out_lin, mask, lens, flags,
                              pos.dropout.p, enc.self.dropout.p, enc.output.dropout.p, self.training, relu_mask=fused_proj)

    def _video_tower(self, pre, feat, mask):
        if self._tower_fused_ok(feat, mask):
            return self._tower_fused(pre, "visual", feat, mask)
        try:
            h = self.encode_input(feat, mask, getattr(self, pre + "visual_input_proj"), getattr(self, pre + "visual_encoder"),
                                  getattr(self, pre + "visual_pos_embed"))
            lin = getattr(self, pre + "out_mapping_linear")
            return F_.linear(h, lin.weight, lin.bias)
        finally:
            ops.set_row_groups(None, 0)

    def _query_tower(self, pre, feat, mask):
        if self._tower_fused_ok(feat, mask):
            return self.get_modularized_queries(self._tower_fused(pre, "query", feat, mask), mask, inheritance=(pre == ""))
        try:
            h = self.encode_input(feat, mask, getattr(self, pre + "query_input_proj"), getattr(self, pre + "query_encoder"),
                                  getattr(self, pre + "query_pos_embed"))
            return self.get_modularized_queries(h, mask, inheritance=(pre == ""))
        finally:
            ops.set_row_groups(None, 0)

    TOWER_FORK = "pair"
    QUERY_TOWERS_FIRST = True

    def _encode_towers(self, video, vmask, text, tmask):
        """(g_inh, g_exp, q_inh, q_exp) of the training forward.  With tower_streams set (train.GraphedTrainStep sets it) the
        four towers are enqueued on four streams forked from the current one and joined behind the last: they share nothing
        but the raw features, and most of their kernels fill less than the chip (a 16,384 x 384 x 384 GEMM is 384 workgroups
        for 512 slots, the attention kernels 512-640, the LayerNorm kernels fewer), so the hardware - and a captured hipGraph,
        whose fork / join edges these become - runs them side by side.  Autograd runs every node's backward on the stream of
        its forward, so the backward pass of the towers overlaps the same way."""
        if not (self.tower_streams and self.training and video.is_cuda and torch.is_grad_enabled()):
            try:
                if self.training and self.QUERY_TOWERS_FIRST:   # same order (= same dropout draws) as the four-stream form below
                    q_inh, q_exp = self.encode_query(text, tmask)
                    g_inh, g_exp = self.encode_context(video, vmask)
                else:
                    g_inh, g_exp = self.encode_context(video, vmask)
                    q_inh, q_exp = self.encode_query(text, tmask)
            finally:
                ops.set_row_groups(None, 0)                     # (encode_input leaves a training tower's row groups current)
            return g_inh, g_exp, q_inh, q_exp
        if text.dim() == 2:                  # the reference's collate .squeeze() drops a batch of one
            text, tmask = text.unsqueeze(0), tmask.reshape(1, -1)
        dev = video.device
        cur = torch.cuda.current_stream(dev)
        if self._side_streams is None or self._side_streams[0].device != dev:
            self._side_streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
        pres = ("", "exp_") if self.double_branch else ("",)
        jobs = [(self._video_tower, pre, video, vmask) for pre in pres] + [(self._query_tower, pre, text, tmask) for pre in pres]
        nvt = len(pres)
        if self.QUERY_TOWERS_FIRST:
            # autograd runs the towers' backward chains in reverse creation order, and the graph executor starts the parallel
            # branches of a replayed step in capture order, two or three at a time: created LAST, the two long video towers are
            # the first backward chains to start instead of the last
            jobs = jobs[nvt:] + jobs[:nvt]
        if self._tower_runner is not None:
            # train.GraphedTrainStep, one GPU: every tower is captured into its OWN graph on its own stream (the runner closes the
            # graph that is open, captures the thunks one by one and opens the graph of the losses); the replay launches the four
            # forward graphs - and later the four backward graphs - on four real streams
            outs = self._tower_runner([(lambda fn=fn, pre=pre, x=x, m=m: fn(pre, x, m)) for fn, pre, x, m in jobs],
                                      [int(x.numel()) for _, _, x, _ in jobs])
            if self.QUERY_TOWERS_FIRST:
                outs = outs[len(jobs) - nvt:] + outs[:len(jobs) - nvt]
            return (outs[0], outs[1], outs[2], outs[3]) if self.double_branch else (outs[0], None, outs[1], None)
        outs = []
        # Where the side streams fork matters: a stream that waits for `cur` AFTER tower 0 was enqueued there waits for tower 0
        # (the first 0.6 ms of the C3 step ran one tower alone).  TOWER_FORK = "pair": the second video tower forks before
        # anything is enqueued, so the two long towers run side by side from the start; the query towers fork behind tower 0.
        # "all" forks the three side streams up front (C3 bf16 step 4.45 instead of 4.7 ms) but hipGraphLaunch of ROCm 7.0.2
        # segfaults on the second or third graph captured that way in one process (tests/test_train_loop_gpu.py: three batch
        # signatures; a chain of waits or a first kernel per branch does not help) - not used.  "late": every side stream forks
        # behind tower 0.
        early = {"pair": 1, "all": len(jobs) - 1, "late": 0}[self.TOWER_FORK]
        for i in range(1, early + 1):
            self._side_streams[i - 1].wait_stream(cur)
        for i, (fn, pre, x, m) in enumerate(jobs):
            st = cur if i == 0 else self._side_streams[i - 1]
            if i > early:
                st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(fn(pre, x, m))
        for i in range(1, len(jobs)):
            cur.wait_stream(self._side_streams[i - 1])
            outs[i].record_stream(cur)       # allocated on a side stream, consumed by the losses on this one
        if self.QUERY_TOWERS_FIRST:
            outs = outs[len(jobs) - nvt:] + outs[:len(jobs) - nvt]
        if self.double_branch:
            return outs[0], outs[1], outs[2], outs[3]
        return outs[0], None, outs[1], None

    # ------------------------------------------------------------------ data-parallel gradient buckets
    TOWERS = (("g_inh", ("visual_", "out_mapping_linear.")), ("g_exp", ("exp_visual_", "exp_out_mapping_linear.")),
              ("q_inh", ("query_", "modular_vector_mapping.")), ("q_exp", ("exp_query_", "exp_modular_vector_mapping.")))

    def grad_buckets(self):
        """One gradient bucket per tower, in the order train.backward_in_phases runs the towers' backward passes: the two video
        towers first (8 MB of gradients each at TVR sizes - their all-reduce hides under the towers that follow), the query
        towers (3.6 MB each) last, so the one collective nothing can hide is the smallest.  Every parameter of the model sits
        in exactly one tower: the losses behind the tower outputs have none."""
        named = list(self.named_parameters())
        out = [[p for n, p in named if n.startswith(prefixes)] for _, prefixes in self.TOWERS]
        return [b for b in out if b]

    def forward_phased(self, batch, staged=None):
        """forward_tensors plus the phases of its backward pass: [(tower output, that tower's parameters)] in grad_buckets()
        order.  The loss depends on a tower's parameters only through its output, so d loss / d outputs first and then one
        tower at a time yields the same gradients as loss.backward() (train.backward_in_phases)."""
        taps = {}
        loss, parts = self.forward_tensors(batch, staged=staged, taps=taps)
        buckets = self.grad_buckets()
        names = [n for n, _ in self.TOWERS if n in taps]
        if len(names) != len(buckets):
            raise RuntimeError("forward_phased: tower outputs and gradient buckets do not match")
        return loss, parts, [(taps[n], b) for n, b in zip(names, buckets)]

    def forward_tensors(self, batch, staged=None, taps=None):
        """The training forward without its one host synchronisation: returns (loss, {inher_trip, ..., kl_intra}) as
        tensors.  `staged` (train.GraphedTrainStep): an object with .labels_dev (int32) and .draws = [(r_t2v, r_v2t), ...]
        on the device; nothing in here then touches host memory, so the whole step can be captured into a hipGraph.
        taps (dict, optional): receives the four tower outputs (forward_phased)."""
        labels = batch["text_labels"]
        mask = batch["student_videos_mask"].float()
        dev = mask.device
        F_.drop_pre_ln()
        dual = False
        if self.training and self.double_branch and torch.is_grad_enabled() and mask.is_cuda:
            # both video towers normalise the same raw features (model.py:229-243): their dropout slots are drawn here, up front
            # (whatever form the step takes, the masks are the same), and in throughput mode ONE pass writes both branches'
            # LayerNorm-dropout rows - on the first video tower's stream when the stepper captures (the hook), so that the main
            # stream's chain (teacher scores, a query tower) is not held up by it
            vid = batch["student_videos"]
            layers = [self.visual_input_proj, self.exp_visual_input_proj]
            p_in = float(layers[0].net[0].p)
            if p_in == float(layers[1].net[0].p) and vid.dtype == torch.float32 and vid.is_contiguous():
                F_.predraw_in_proj_slots(vid, layers, p_in)
                dual = (mask.is_contiguous() and self._tower_fused_ok(vid, mask) and F_.in_proj_ln_dual_ok(vid, layers, p_in, True))
        hook = self._pre_ln_hook
        if hook is not None:
            with hook(dual):
                if dual:
                    F_.in_proj_ln_dual(vid, mask, layers, p_in)
        elif dual:
            F_.in_proj_ln_dual(vid, mask, layers, p_in)
        if self.training and torch.is_grad_enabled() and mask.is_cuda:
            F_.begin_zero_arena(dev)          # one fill for the step's small zero-initialised gradient buffers
        else:
            F_.end_zero_arena()
        # A query axis padded to a bucket (train.GraphedTrainStep, batches of variable caption counts: Charades / ActivityNet): the
        # text tensors hold nq_rows >= len(labels) queries, the rows behind the real ones are padding (one valid zero word each).
        # Towers and pooled scores run over all rows; the losses run over the first len(labels) (functional.branch_losses nq_valid).
        nq_rows = int(batch["student_text"].shape[0])
        nq_valid = None
        if nq_rows != len(labels):
            if nq_rows < len(labels) or not (mask.is_cuda and F_.BRANCH_LOSS_FUSED and F_.simpool_train_ok()):
                raise ValueError(f"forward_tensors: {nq_rows} text rows for {len(labels)} labels (a padded query axis needs the fused loss path)")
            nq_valid = len(labels)
        if staged is not None:
            lab = staged.labels_dev
        else:
            lab_np = np.zeros(nq_rows, dtype=np.int32)
            lab_np[:len(labels)] = np.asarray(labels)
            lab = torch.as_tensor(lab_np, dtype=torch.int32, device=dev)
        nv, L = mask.shape
        lens = self._lens(mask, nv, L, dev)

        t_text = batch["teacher_text"].float().reshape(nq_rows, -1)          # .squeeze() of model.py:114
        t_vid = batch["teacher_videos"].float()

        fused = F_.simpool_train_ok()

        def both(q, g, want_clip):
            if fused:
                # one MFMA GEMM with a pooling epilogue: cosine and raw maxima + the positive column of the clip-level
                # cosines (all compute_kl_loss reads, model.py:184); the (Nq, Nv, L) tensors are never formed
                return F_.simpool_train(q, g, lens, lab, want_clip)
            pooled_cos, clip_cos, _ = F_.clip_pool(F_.clip_scores(F_.normalize(q), F_.normalize(g)), lens)      # "fp32_exact"
            pooled_raw, _, _ = F_.clip_pool(F_.clip_scores(q, g), lens)
            return pooled_cos, pooled_raw, (clip_cos if want_clip else None)

        def trip(scores, call):
            st = None if staged is None else (lab,) + tuple(staged.draws[call])
            return self.get_clip_triplet_loss(scores, labels, _staged=st)

        hard_neg = bool(_cfg_get(self.config, "use_hard_negative"))
        fused_losses = fused and F_.BRANCH_LOSS_FUSED and mask.is_cuda

        def draws(call):
            """(r_t2v, r_v2t) of get_clip_triplet_loss call `call` on the device: staged by the graph stepper, or drawn here with the
            reference's CPU torch.randint calls (model.py:366-380), in its order."""
            if staged is not None:
                r_t2v, r_v2t = staged.draws[call]
                return r_t2v, (None if hard_neg else r_v2t)
            _, r_t2v, r_v2t = self._draw_triplet(np.asarray(labels), nv)
            return r_t2v.to(dev), (None if r_v2t is None else r_v2t.to(dev))

        # the teacher's scores need nothing of the towers: enqueued in front of them they run beside the tower graphs (the
        # stepper's main stream is idle there) instead of at the head of the serial loss section
        with torch.no_grad():
            _, t_raw, t_clip = both(t_text, t_vid, True)
        g_inh, g_exp, q_inh, q_exp = self._encode_towers(batch["student_videos"], mask, batch["student_text"],
                                                         batch["student_text_mask"])
        if taps is not None:
            taps.update({k: v for k, v in (("g_inh", g_inh), ("g_exp", g_exp), ("q_inh", q_inh), ("q_exp", q_exp)) if v is not None})
        soft = self.label_style == "soft"

        def inh_part(q, g):
            i_cos, i_raw, i_clip = both(q, g, True)
            if fused_losses:
                r_t2v, r_v2t = draws(0)
                return F_.branch_losses(i_cos, i_raw, t_raw, i_clip, t_clip, lab, lens, r_t2v, r_v2t, hard_neg, _cfg_get(self.config, "margin"),
                                        soft, self.alpha, self.belta, self.inher_nce_weight, self.kl_intra_weight * self.weight, False,
                                        kd_factor=self.kl_intra_weight, nq_valid=nq_valid)
            inher_trip = trip(i_cos, 0)
            if soft:
                inher_nce = self.inher_nce_weight * F_.nce_soft(lab, i_raw, t_raw, self.alpha, self.belta)
            else:
                inher_nce = self.inher_nce_weight * F_.nce_hard(lab, i_raw)
            kl_intra = self.kl_intra_weight * self.weight * F_.kl_frame(i_clip, t_clip, lab, lens, 0.2)
            return inher_trip, inher_nce, kl_intra

        def exp_part(q, g):
            e_cos, e_raw, _ = both(q, g, False)
            if fused_losses:
                r_t2v, r_v2t = draws(1)
                return F_.branch_losses(e_cos, e_raw, None, None, None, lab, lens, r_t2v, r_v2t, hard_neg, _cfg_get(self.config, "margin"),
                                        soft, self.alpha, self.belta, self.explore_nce_weight, 0.0, True, kd_factor=0.0, nq_valid=nq_valid)[:2]
            explore_trip = trip(e_cos, 1)
            if soft:
                explore_nce = self.explore_nce_weight * F_.nce_soft(lab, e_raw, e_raw, self.alpha, self.belta)
            else:
                explore_nce = self.explore_nce_weight * F_.nce_hard(lab, e_raw)
            return explore_trip, explore_nce

        explore_trip, explore_nce = 0, 0
        if self.double_branch and self._branch_runner is not None:
            # train.GraphedTrainStep, one GPU: the two branches' losses - and their backward passes down to the tower outputs -
            # are independent of each other: the runner captures each into its own graph on its own stream
            (inher_trip, inher_nce, kl_intra), (explore_trip, explore_nce) = self._branch_runner(
                [(inh_part, (q_inh, g_inh)), (exp_part, (q_exp, g_exp))])
            kl = kl_intra
            loss = F_.sum_scalars(inher_trip, inher_nce, kl, explore_trip, explore_nce)     # one launch, the reference's order
        else:
            inher_trip, inher_nce, kl_intra = inh_part(q_inh, g_inh)
            if self.double_branch:
                explore_trip, explore_nce = exp_part(q_exp, g_exp)
            kl = kl_intra
            loss = (F_.sum_scalars(inher_trip, inher_nce, kl, explore_trip, explore_nce) if self.double_branch and inher_trip.is_cuda
                    else inher_trip + inher_nce + kl + explore_trip + explore_nce)
        F_.drop_pre_ln()
        return loss, {"inher_trip": inher_trip, "inher_nce": inher_nce, "explore_trip": explore_trip,
                      "explore_nce": explore_nce, "kl": kl, "kl_intra": kl_intra}
