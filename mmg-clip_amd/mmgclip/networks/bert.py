"""BERT text tower on the HIP kernels (forward + backward), state-dict compatible with Hugging Face `BertModel`.

Reference: `BertEncoder` wraps `AutoModel.from_pretrained(<name>)` and returns `last_hidden_state`
(mmgclip/networks/encoder.py:131-156); architecture values from notebooks/bert_experimental.ipynb:609-624
(vocab 28996, hidden 768, 12 layers x 12 heads, FFN 3072, 512 positions, LayerNorm eps 1e-12, erf-GELU, post-LN).
The reference freezes every BERT parameter (encoder.py:141-142); `freeze=False` enables the north star's fine-tuning.
Dropout (p = 0.1 in the HF config, live because the reference calls model.train(), ClassifierExperiment.py:97) is applied in
training mode at HF's four positions with counter-based masks (csrc/dropout.h): the backward regenerates them from the step's
seed, and the oracle restates them (oracle/dropout_oracle.py), so parity holds mask for mask.  eval() / `dropout=False` / p = 0
give the deterministic tower.

Device layout: hidden states bf16 [B*S, 768]; Q, K, V come from ONE fused GEMM ([2304, 768] weight = the three HF
matrices stacked, contiguous in the parameter arena so its gradient needs no scatter).
"""
import os

import torch
import torch.nn as nn

from .. import _hip
from .. import kernels as K
from .. import linalg as L
from ..params import ParamArena, backward_finished, last_backward, note_forward, stream_anchor


class BertConfigLite:
    """The subset of HF BertConfig the tower needs (defaults = Bio_ClinicalBERT, notebooks/bert_experimental.ipynb:609-624)."""

    def __init__(self, vocab_size=28996, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12,
                 initializer_range=0.02, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, **_):
        self.vocab_size, self.hidden_size = vocab_size, hidden_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.intermediate_size, self.max_position_embeddings = intermediate_size, max_position_embeddings
        self.type_vocab_size, self.layer_norm_eps, self.initializer_range = type_vocab_size, layer_norm_eps, initializer_range
        self.hidden_dropout_prob, self.attention_probs_dropout_prob = float(hidden_dropout_prob), float(attention_probs_dropout_prob)
        assert 0.0 <= self.hidden_dropout_prob < 1.0 and 0.0 <= self.attention_probs_dropout_prob < 1.0
        assert hidden_size == 64 * num_attention_heads, "the attention kernel is specialised for head_dim 64"


def _hf_layout(cfg):
    """Module tree whose state-dict keys equal Hugging Face BertModel's (incl. the unused pooler)."""
    H, F = cfg.hidden_size, cfg.intermediate_size
    m = nn.Module()
    m.embeddings = nn.Module()
    m.embeddings.word_embeddings = nn.Embedding(cfg.vocab_size, H, padding_idx=0)
    m.embeddings.position_embeddings = nn.Embedding(cfg.max_position_embeddings, H)
    m.embeddings.token_type_embeddings = nn.Embedding(cfg.type_vocab_size, H)
    m.embeddings.LayerNorm = nn.LayerNorm(H, eps=cfg.layer_norm_eps)
    m.encoder = nn.Module()
    layers = []
    for _ in range(cfg.num_hidden_layers):
        lyr = nn.Module()
        lyr.attention = nn.Module()
        lyr.attention.self = nn.Module()
        lyr.attention.self.query, lyr.attention.self.key, lyr.attention.self.value = nn.Linear(H, H), nn.Linear(H, H), nn.Linear(H, H)
        lyr.attention.output = nn.Module()
        lyr.attention.output.dense = nn.Linear(H, H)
        lyr.attention.output.LayerNorm = nn.LayerNorm(H, eps=cfg.layer_norm_eps)
        lyr.intermediate = nn.Module()
        lyr.intermediate.dense = nn.Linear(H, F)
        lyr.output = nn.Module()
        lyr.output.dense = nn.Linear(F, H)
        lyr.output.LayerNorm = nn.LayerNorm(H, eps=cfg.layer_norm_eps)
        layers.append(lyr)
    m.encoder.layer = nn.ModuleList(layers)
    m.pooler = nn.Module()
    m.pooler.dense = nn.Linear(H, H)
    for mod in m.modules():                               # HF _init_weights
        if isinstance(mod, nn.Linear):
            nn.init.normal_(mod.weight, std=cfg.initializer_range)
            nn.init.zeros_(mod.bias)
        elif isinstance(mod, nn.Embedding):
            nn.init.normal_(mod.weight, std=cfg.initializer_range)
            if mod.padding_idx is not None:
                with torch.no_grad():
                    mod.weight[mod.padding_idx].zero_()
    return m


class BertTower(nn.Module):
    def __init__(self, config=None, micro_batch=4096, dropout=True):
        super().__init__()
        self.config = config or BertConfigLite()
        self.dropout = bool(dropout)          # training-mode dropout on (the reference's behaviour); eval() turns it off as in HF
        self.next_dropout_seed = None         # tests: force the seed of the next training-mode forward
        self.model = _hf_layout(self.config)
        self.model.config = self.config
        self.model_output_dimension = self.config.hidden_size
        self.micro_batch = micro_batch        # sequences per pass
        self.packed = os.environ.get("MMG_BERT_PACKED", "1") != "0"     # run the layers on the valid tokens only
        self._arena = None
        self._wc = None
        self._wc_version = None
        self._anchor = None
        self.post_backward_hook = None      # called with the arena once this tower's gradients are complete

    # ---- parameter plumbing: arena order puts q,k,v (weights, then biases) of a layer next to each other -----
    def _ordered_named_parameters(self):
        named = dict(self.model.named_parameters())
        order = [n for n in named if n.startswith("embeddings.")]
        for i in range(self.config.num_hidden_layers):
            p = f"encoder.layer.{i}."
            order += [p + f"attention.self.{x}.weight" for x in ("query", "key", "value")]
            order += [p + f"attention.self.{x}.bias" for x in ("query", "key", "value")]
            order += [n for n in named if n.startswith(p) and ".attention.self." not in n]
        order += [n for n in named if n.startswith("pooler.")]
        assert len(order) == len(named)
        return [(n, named[n]) for n in order]

    def _materialize(self, device):
        if self._arena is not None and self._arena.device == device and self._arena.is_bound():
            return
        self._arena = ParamArena(self._ordered_named_parameters(), device)
        self._wc_version = None
        self._anchor = torch.zeros(1, device=device, requires_grad=True)

    @property
    def arena(self):
        return self._arena

    def _refresh_working_copies(self):
        A = self._arena
        v = A.version()
        if self._wc_version == v:
            return
        H = self.config.hidden_size
        wc = {}
        e = self.model.embeddings
        wc["word"], wc["pos"], wc["type"] = (K.cast_bf16(e.word_embeddings.weight.data), K.cast_bf16(e.position_embeddings.weight.data),
                                             K.cast_bf16(e.token_type_embeddings.weight.data))
        for i, lyr in enumerate(self.model.encoder.layer):
            p = f"encoder.layer.{i}."
            wqkv = A.span(p + "attention.self.query.weight", p + "attention.self.value.weight")[:3 * H * H].view(3 * H, H)
            wc[f"{i}.wqkv"] = K.cast_bf16(wqkv)
            wc[f"{i}.wqkvt"] = K.transpose_cast_bf16(wqkv)
            wc[f"{i}.bqkv"] = A.span(p + "attention.self.query.bias", p + "attention.self.value.bias")
            for tag, lin in (("wo", lyr.attention.output.dense), ("wi", lyr.intermediate.dense), ("wf", lyr.output.dense)):
                wc[f"{i}.{tag}"] = K.cast_bf16(lin.weight.data)
                wc[f"{i}.{tag}t"] = K.transpose_cast_bf16(lin.weight.data)
        self._wc, self._wc_version = wc, v

    def _check_qkv_contiguous(self):
        H = self.config.hidden_size
        A = self._arena
        assert (H * H) % 64 == 0 and H % 64 == 0, "fused QKV views need 64-element aligned slices"
        for i in range(self.config.num_hidden_layers):
            p = f"encoder.layer.{i}.attention.self."
            assert A.offsets[p + "key.weight"] - A.offsets[p + "query.weight"] == H * H
            assert A.offsets[p + "key.bias"] - A.offsets[p + "query.bias"] == H

    # ---- packed ("unpadded") layout -------------------------------------------------------------------------------
    @staticmethod
    def sequence_lengths(mask):
        """Python list of lengths when `mask` [B,S] is a right-padded prompt batch (ones then zeros, no empty row), else None.
        Reads the mask on the host: free for a CPU tensor, one small device read otherwise."""
        m = mask.detach().to("cpu", torch.int64)
        lens = m.sum(1)
        S = m.shape[1]
        if not bool((lens >= 1).all()) or not bool((m == (torch.arange(S)[None, :] < lens[:, None])).all()):
            return None
        return lens.tolist()

    def _lengths_of(self, mask):
        """Lengths of the whole batch, remembered ON the tensor object (so a batch that is fed again costs no device read,
        and a new or modified tensor can never pick up stale lengths)."""
        tag = getattr(mask, "_mmg_seq_lens", None)
        if tag is not None and tag[0] == mask._version:
            return tag[1]
        lens = self.sequence_lengths(mask)
        try:
            mask._mmg_seq_lens = (mask._version, lens)
        except Exception:           # noqa: BLE001  (tensor subclasses without attribute support)
            pass
        return lens

    def _packing(self, lens, b0, b1, S, device):
        """(row indices of the valid tokens of sequences b0..b1 inside that slice [T] int64, cu_seqlens int32) or None.
        The reference pads to max_length (dataset.py:347) and reads only the [SEP] row (mmgclip_model.py:110-111): the layers
        then run on the valid tokens only."""
        if lens is None or S > 256:
            return None
        sl = lens[b0:b1]
        if sum(sl) >= len(sl) * S:
            return None
        rows = torch.cat([b * S + torch.arange(int(n)) for b, n in enumerate(sl)])
        cu = torch.zeros(len(sl) + 1, dtype=torch.int32)
        cu[1:] = torch.tensor(sl, dtype=torch.int32).cumsum(0)
        return rows.to(device), cu.to(device)

    # ---- dropout -------------------------------------------------------------------------------------------------
    # sites of one step's masks (oracle/dropout_oracle.py uses the same numbering)
    SITE_EMBEDDINGS = 0

    @staticmethod
    def _site(layer, which):
        """which: 1 attention probabilities, 2 attention output dense, 3 FFN output dense."""
        return 4 * layer + which

    def reseed_dropout(self, seed=None):
        """Restart the tower's private dropout-seed stream from (seed, rank); seed=None: the last `seeding()` call's seed, or
        torch.initial_seed() when `seeding` never ran.  Called by itself whenever `utils.global_utils.seeding` has run since the
        last draw, so the reference's `seeding(config.base.seed)` reproduces a run."""
        from ..utils.global_utils import seed_epoch
        epoch, base = seed_epoch()
        if seed is None:
            seed = base if base is not None else torch.initial_seed()
        rank = int(os.environ.get("RANK", "0"))
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank = torch.distributed.get_rank()
        # splitmix-style fold of (seed, rank): every data-parallel rank draws its own masks for its own samples
        mixed = (int(seed) * 0x9E3779B97F4A7C15 + (rank + 1) * 0xBF58476D1CE4E5B9 + 0x94D049BB133111EB) & 0x7FFFFFFFFFFFFFFF
        self._drop_gen = torch.Generator().manual_seed(mixed)
        self._drop_epoch = epoch

    def _draw_dropout(self):
        """(p_hidden, p_attention, seed) for this forward, or None (eval mode, dropout=False, both p = 0).  The seed comes from a
        generator the tower owns, derived from (base seed, rank): torch's global CPU generator is never consumed (the reference's
        nn.Dropout does not touch it either: its draws belong to the DataLoader's sampler), `seeding(s)` reproduces a run, and
        data-parallel ranks draw different masks."""
        cfg = self.config
        if not (self.training and self.dropout) or (cfg.hidden_dropout_prob == 0.0 and cfg.attention_probs_dropout_prob == 0.0):
            return None
        if os.environ.get("MMG_BERT_DROPOUT", "1") == "0":        # operational off-switch (deterministic runs, the parity tests)
            return None
        if self.next_dropout_seed is not None:
            seed, self.next_dropout_seed = int(self.next_dropout_seed), None
        else:
            from ..utils.global_utils import seed_epoch
            if getattr(self, "_drop_gen", None) is None or self._drop_epoch != seed_epoch()[0]:
                self.reseed_dropout()
            seed = int(torch.randint(0, 2 ** 62, (1,), generator=self._drop_gen).item())
        return cfg.hidden_dropout_prob, cfg.attention_probs_dropout_prob, seed

    _warned_long_attention_dropout = False

    def _attention_dropout_ok(self, S):
        if S <= 512:                         # (= max_position_embeddings of the reference's BERT: every length it can run)
            return True
        if not BertTower._warned_long_attention_dropout:
            import warnings
            warnings.warn(f"attention-probability dropout is implemented for S <= 512 (got S = {S}): skipped; the hidden-state "
                          "dropouts stay on")
            BertTower._warned_long_attention_dropout = True
        return False

    # ---- one micro-batch ---------------------------------------------------------------------------------------
    def _forward_mb(self, ids, tt, mask, save, pack=None, drop=None, b0=0):
        cfg, wc = self.config, self._wc
        B, S = ids.shape
        heads, eps = cfg.num_attention_heads, cfg.layer_norm_eps
        e = self.model.embeddings
        emb = K.bert_embed_fwd(ids, tt, wc["word"], wc["pos"], wc["type"], S)
        rows, cu = pack if pack is not None else (None, None)
        if rows is not None:
            emb = emb.index_select(0, rows)                  # [T, H]: the valid tokens, sequence after sequence
        tok = None                                           # token id (b * S + s in the whole batch) of every row, for the masks
        p_h = p_a = 0.0
        if drop is not None:
            p_h, p_a, seed = drop
            if not self._attention_dropout_ok(S):
                p_a = 0.0
            if rows is not None or b0 > 0:
                tok = (rows if rows is not None else torch.arange(B * S, device=ids.device)) + b0 * S
        # The residual stream and the pre-LayerNorm sums x + sublayer(x) are kept in fp32 (`xf`, `a`, `f`); every GEMM still reads a
        # bf16 copy of the stream (`x`).  Measured at BASELINE config C1 (tests/test_c1_gpu.py): with a bf16 stream the text tower
        # owned the end-to-end loss error (1e-3 relative, the north star's bar); BERT is 5 % of the step, so the fp32 rows are free.
        x, xf, mean, rstd = K.layernorm_fwd_f32(emb.float(), e.LayerNorm.weight.data, e.LayerNorm.bias.data, eps, want_stats=save,
                                                want_f32=True)
        if p_h > 0:                                          # HF BertEmbeddings: dropout after the LayerNorm
            x = K.dropout_f32_(xf, p_h, seed, self.SITE_EMBEDDINGS, rows=tok, want_bf16=True)
        saved = {"emb": (emb, mean, rstd), "layers": [], "shape": (B, S), "tok": (ids, tt, mask), "pack": pack,
                 "drop": (p_h, p_a, seed, tok, b0) if drop is not None else None} if save else None
        for i, lyr in enumerate(self.model.encoder.layer):
            qkv = L.gemm_nt(x, wc[f"{i}.wqkv"], bias=wc[f"{i}.bqkv"][:3 * cfg.hidden_size])
            if p_a > 0:                                      # HF BertSelfAttention: dropout on the probabilities
                ctx, lse = K.attention_dropout_fwd(qkv, mask, B, S, heads, p_a, seed, self._site(i, 1), want_lse=save, cu=cu,
                                                   first_sequence=b0)
            else:
                ctx, lse = K.attention_fwd(qkv, mask, B, S, heads, want_lse=save, cu=cu)
            a = L.gemm_nt(ctx, wc[f"{i}.wo"], bias=lyr.attention.output.dense.bias.data, out_dtype=torch.float32)
            if p_h > 0:                                      # HF BertSelfOutput: dropout(dense(ctx)) + residual
                K.dropout_f32_(a, p_h, seed, self._site(i, 2), rows=tok)
            x1, x1f, m1, r1 = K.layernorm_fwd_f32(a, lyr.attention.output.LayerNorm.weight.data, lyr.attention.output.LayerNorm.bias.data,
                                                  eps, want_stats=save, want_f32=True, res=xf)      # a <- a + xf, then LayerNorm
            hpre = torch.empty(x.shape[0], cfg.intermediate_size, device=x.device, dtype=torch.bfloat16) if save else None
            g = L.gemm_nt(x1, wc[f"{i}.wi"], bias=lyr.intermediate.dense.bias.data, epi=L.EPI_GELU, aux_out=hpre)
            f = L.gemm_nt(g, wc[f"{i}.wf"], bias=lyr.output.dense.bias.data, out_dtype=torch.float32)
            if p_h > 0:                                      # HF BertOutput: dropout(dense(g)) + residual
                K.dropout_f32_(f, p_h, seed, self._site(i, 3), rows=tok)
            x2, x2f, m2, r2 = K.layernorm_fwd_f32(f, lyr.output.LayerNorm.weight.data, lyr.output.LayerNorm.bias.data, eps,
                                                  want_stats=save, want_f32=True, res=x1f)          # f <- f + x1f, then LayerNorm
            if save:
                saved["layers"].append((x, qkv, ctx, lse, a, m1, r1, x1, hpre, f, m2, r2))
            x, xf = x2, x2f
        x = xf                                               # the tower's output is the fp32 stream (EOS pooling gathers fp32 rows)
        if rows is not None:                                 # back to the padded [B*S, H] layout (padding rows: zeros)
            full = torch.zeros(B * S, x.shape[1], device=x.device, dtype=x.dtype)
            full.index_copy_(0, rows, x)
            x = full
        return x, saved

    def _backward_mb(self, dx, saved, final=False):
        """final: last micro-batch of the step's last backward of this tower - a layer's gradients are complete once it has
        passed, and are announced to the gradient all-reduce (ParamArena.mark_ready)."""
        cfg, wc, A = self.config, self._wc, self._arena
        B, S = saved["shape"]
        ids, tt, mask = saved["tok"]
        rows, cu = saved["pack"] if saved["pack"] is not None else (None, None)
        if rows is not None:
            dx = dx.index_select(0, rows)
        heads, H = cfg.num_attention_heads, cfg.hidden_size
        p_h, p_a, seed, tok, b0 = saved["drop"] if saved.get("drop") is not None else (0.0, 0.0, 0, None, 0)
        # a dropped sub-layer output y~ = mask y / (1 - p) sits next to the residual: the LayerNorm backward's gradient goes to the
        # residual path as it is and to the sub-layer's GEMMs through the same mask
        masked = (lambda t, site: K.dropout_bf16(t, p_h, seed, site, rows=tok)) if p_h > 0 else (lambda t, site: t)
        for i in range(cfg.num_hidden_layers - 1, -1, -1):
            lyr = self.model.encoder.layer[i]
            p = f"encoder.layer.{i}."
            x, qkv, ctx, lse, a, m1, r1, x1, hpre, f, m2, r2 = saved["layers"][i]
            df = K.layernorm_bwd_f32(dx, f, m2, r2, lyr.output.LayerNorm.weight.data, A.g(p + "output.LayerNorm.weight"),
                                     A.g(p + "output.LayerNorm.bias"))
            dfm = masked(df, self._site(i, 3))
            g = torch.empty_like(hpre)                     # GELU(hpre), rebuilt by the same epilogue that applies GELU'
            dh = L.gemm_nt(dfm, wc[f"{i}.wft"], epi=L.EPI_DGELU, aux_in=hpre, aux_out=g)
            L.gemm_tn_acc(dfm, g, A.g(p + "output.dense.weight"), colsum=A.g(p + "output.dense.bias"))
            del g, dfm
            L.gemm_tn_acc(dh, x1, A.g(p + "intermediate.dense.weight"), colsum=A.g(p + "intermediate.dense.bias"))
            dx1 = L.gemm_nt(dh, wc[f"{i}.wit"], residual=df)          # + residual path of the FFN block
            del dh, df
            da = K.layernorm_bwd_f32(dx1, a, m1, r1, lyr.attention.output.LayerNorm.weight.data,
                                     A.g(p + "attention.output.LayerNorm.weight"), A.g(p + "attention.output.LayerNorm.bias"))
            del dx1
            dam = masked(da, self._site(i, 2))
            L.gemm_tn_acc(dam, ctx, A.g(p + "attention.output.dense.weight"), colsum=A.g(p + "attention.output.dense.bias"))
            dctx = L.gemm_nt(dam, wc[f"{i}.wot"])
            del dam
            if p_a > 0:
                dqkv = K.attention_dropout_bwd(qkv, mask, ctx, lse, dctx, B, S, heads, p_a, seed, self._site(i, 1), cu=cu,
                                               first_sequence=b0)
            else:
                dqkv = K.attention_bwd(qkv, mask, ctx, lse, dctx, B, S, heads, cu=cu)
            del dctx
            gw = A.gspan(p + "attention.self.query.weight", p + "attention.self.value.weight")[:3 * H * H].view(3 * H, H)
            gb = A.gspan(p + "attention.self.query.bias", p + "attention.self.value.bias")[:3 * H]
            L.gemm_tn_acc(dqkv, x, gw, colsum=gb)
            dx = L.gemm_nt(dqkv, wc[f"{i}.wqkvt"], residual=da)       # + residual path of the attention block
            del dqkv, da
            saved["layers"][i] = None
            if final:
                A.mark_ready(p)
        emb, mean, rstd = saved["emb"]
        e = self.model.embeddings
        dx = masked(dx, self.SITE_EMBEDDINGS)
        demb = K.layernorm_bwd(dx, emb, mean, rstd, e.LayerNorm.weight.data, A.g("embeddings.LayerNorm.weight"),
                               A.g("embeddings.LayerNorm.bias"))
        if rows is not None:
            full = torch.zeros(B * S, demb.shape[1], device=demb.device, dtype=demb.dtype)
            full.index_copy_(0, rows, demb)
            demb = full
        K.bert_embed_bwd(demb, ids, tt, A.g("embeddings.word_embeddings.weight"), A.g("embeddings.position_embeddings.weight"),
                         A.g("embeddings.token_type_embeddings.weight"), B, S)

    # ---- public -----------------------------------------------------------------------------------------------------
    def forward(self, input_ids, attention_mask=None, token_type_ids=None, packed=None, **_):
        """-> last_hidden_state as fp32 [B*S, H] (the tower's fp32 residual stream; .view(B, S, H) for the HF-shaped tensor).
        packed=True (default: self.packed) computes the rows of valid tokens only and leaves zeros in the padding rows;
        packed=False reproduces HF's values there too."""
        _hip.require_gpu(input_ids)
        self._materialize(input_ids.device)
        self._check_qkv_contiguous()
        ids = input_ids.to(torch.int64).contiguous()
        mask = attention_mask.to(torch.int64).contiguous() if attention_mask is not None else None
        tt = token_type_ids.to(torch.int64).contiguous() if token_type_ids is not None else None
        if ids.shape[1] > self.config.max_position_embeddings:
            raise ValueError(f"sequence length {ids.shape[1]} exceeds max_position_embeddings")
        needs_grad = torch.is_grad_enabled() and self._arena.any_trainable()
        use_packed = self.packed if packed is None else bool(packed)
        lens = self._lengths_of(attention_mask) if (use_packed and attention_mask is not None and ids.shape[1] <= 256) else None
        note_forward(self, needs_grad)
        return _BertFn.apply(self, ids, tt, mask, stream_anchor(self, self._anchor.device) if needs_grad else None, lens)


class _BertFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tower, ids, tt, mask, anchor, lens=None):
        tower._refresh_working_copies()
        save = anchor is not None
        outs, saved = [], []
        mb = tower.micro_batch
        drop = tower._draw_dropout()
        for i in range(0, ids.shape[0], mb):
            sl = slice(i, i + mb)
            pack = tower._packing(lens, i, min(i + mb, ids.shape[0]), ids.shape[1], ids.device)
            h, sv = tower._forward_mb(ids[sl], tt[sl] if tt is not None else None, mask[sl] if mask is not None else None, save,
                                      pack, drop, i)
            outs.append(h)
            saved.append(sv)
        ctx.tower, ctx.saved_mb = tower, saved if save else None
        out = torch.cat(outs, 0) if len(outs) > 1 else outs[0]
        return out

    @staticmethod
    def backward(ctx, dh):
        tower = ctx.tower
        tower._arena.prepare_grads()
        if dh.is_cuda:                       # (two-stream mode: the gradient was produced on another stream than this backward's)
            dh.record_stream(torch.cuda.current_stream())
        dh = dh.to(torch.bfloat16).contiguous()
        row = 0
        for k, sv in enumerate(ctx.saved_mb):
            B, S = sv["shape"]
            tower._backward_mb(dh[row:row + B * S], sv, final=(k == len(ctx.saved_mb) - 1) and last_backward(tower))
            row += B * S
        ctx.saved_mb = None
        backward_finished(tower)
        # The anchor gets a (zero) gradient so that autograd accumulates on THIS stream: the engine then makes the caller's stream wait
        # for it at the end of backward() - in two-stream mode (mmgclip_model.py) the parameter gradients written above are complete for
        # whoever reads them next, with or without an explicit MMGCLIP.join_streams().
        return None, None, None, None, torch.zeros(1, device=dh.device), None


class EosPool(torch.autograd.Function):
    """text_features = hidden[arange(n), attention_mask.sum(-1) - 1]  (mmgclip_model.py:110-111); fp32 [B, H]."""

    @staticmethod
    def forward(ctx, hidden, mask, B, S):
        out, idx = K.eos_pool_fwd(hidden, mask.to(torch.int64).contiguous(), B, S)
        ctx.save_for_backward(idx)
        ctx.B, ctx.S = B, S
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        if dout.is_cuda:
            dout.record_stream(torch.cuda.current_stream())
        return K.eos_pool_bwd(dout.float().contiguous(), idx, ctx.B, ctx.S), None, None, None


def hf_config_dict(cfg):
    return dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                max_position_embeddings=cfg.max_position_embeddings, type_vocab_size=cfg.type_vocab_size,
                layer_norm_eps=cfg.layer_norm_eps, hidden_act="gelu", hidden_dropout_prob=cfg.hidden_dropout_prob,
                attention_probs_dropout_prob=cfg.attention_probs_dropout_prob, initializer_range=cfg.initializer_range, pad_token_id=0,
                position_embedding_type="absolute", model_type="bert")
