"""Training-mode dropout of the text tower (csrc/dropout.h, dropout.hip, the DROP attention kernels, networks/bert.py) against
the numpy restatement of the mask (oracle/dropout_oracle.py) and the oracle BERT that applies it at HF's four positions.

The reference runs HF BertModel under model.train() (ClassifierExperiment.py:97 -> encoder.py:156): hidden_dropout_prob =
attention_probs_dropout_prob = 0.1.  Mask generation is bit-exact (integer hash); the arithmetic around it has the bf16
tolerances of the dropout-free tests (tests/test_towers_gpu.py, tests/test_kernels_gpu.py)."""
import numpy as np
import pytest
import torch

from oracle import dropout_oracle as D
from oracle import encoders_oracle as E

pytestmark = [pytest.mark.gpu, pytest.mark.bert_dropout]


def _rel(a, b):
    a, b = a.detach().float().cpu().double().flatten(), b.detach().float().cpu().double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30)), float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))


def test_elementwise_dropout_is_bit_exact_against_the_numpy_mask(dev):
    from mmgclip import kernels as K
    g = torch.Generator().manual_seed(0)
    M, C, p, seed, site = 1000, 768, 0.1, 0x1234_5678_9ABC_DEF0, 7
    x = torch.randn(M, C, generator=g)
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    # rows = None: token = row
    xd = x.to(dev).clone()
    xb = K.dropout_f32_(xd, p, seed, site, want_bf16=True)
    keep = D.hidden_mask(M, C, p, seed, site)
    want = np.where(keep, x.numpy() * scale, np.float32(0)).astype(np.float32)
    assert np.array_equal(xd.cpu().numpy(), want)
    assert torch.equal(xb.cpu(), torch.from_numpy(want).to(torch.bfloat16))
    assert abs(keep.mean() - 0.9) < 3e-3
    # packed layout: row m holds token rows[m]
    rows = torch.randperm(5000, generator=g)[:M].sort().values
    xd = x.to(dev).clone()
    K.dropout_f32_(xd, p, seed, site, rows=rows.to(dev))
    keep_r = D.hidden_mask(5000, C, p, seed, site)[rows.numpy()]
    assert np.array_equal(xd.cpu().numpy(), np.where(keep_r, x.numpy() * scale, np.float32(0)).astype(np.float32))
    # bf16 gradient path, another site / seed: a different mask
    gb = torch.randn(M, C, generator=g).to(torch.bfloat16)
    out = K.dropout_bf16(gb.to(dev), p, seed + 1, site + 1, rows=rows.to(dev))
    keep2 = D.hidden_mask(5000, C, p, seed + 1, site + 1)[rows.numpy()]
    want2 = torch.from_numpy(np.where(keep2, gb.float().numpy() * scale, np.float32(0)).astype(np.float32)).to(torch.bfloat16)
    assert torch.equal(out.cpu(), want2)
    assert (keep2 != keep_r).mean() > 0.1
    # p = 0 is the identity
    xd = x.to(dev).clone()
    K.dropout_f32_(xd, 0.0, seed, site)
    assert torch.equal(xd.cpu(), x)
    with pytest.raises(RuntimeError, match="probability"):
        K.dropout_f32_(xd, 1.0, seed, site)


def _attention_reference(qkv, lens, S, heads, keep, p):
    """fp32 torch attention of the padded layout with the given keep mask [B, heads, S, S]; returns ctx [B*S, heads*64]."""
    B = len(lens)
    Hd = heads * 64
    q, k, v = [t.reshape(B, S, heads, 64).permute(0, 2, 1, 3) for t in qkv.float().split(Hd, dim=1)]
    s = q @ k.transpose(-1, -2) * 0.125
    key_ok = (torch.arange(S)[None, :] < torch.tensor(lens)[:, None])
    s = s + (~key_ok)[:, None, None, :].float() * -1e30
    prob = s.softmax(-1) * keep.float() * (1.0 / (1.0 - float(np.float32(p))))
    return (prob @ v).permute(0, 2, 1, 3).reshape(B * S, Hd)


@pytest.mark.parametrize("S,lens", [(77, [77, 40, 9]), (256, [256, 130, 31]), (40, [40, 33, 17]), (384, [384, 200, 33]), (512, [512, 257])])
def test_attention_dropout_forward_backward_match_torch_with_the_same_mask(dev, S, lens):
    from mmgclip import kernels as K
    heads, p, seed, site, B = 12, 0.1, 99, 5, len(lens)
    Hd = heads * 64
    g = torch.Generator().manual_seed(S)
    qkv = (torch.randn(B * S, 3 * Hd, generator=g) * 0.8).to(torch.bfloat16)
    dctx = torch.randn(B * S, Hd, generator=g).to(torch.bfloat16)
    mask = (torch.arange(S)[None, :] < torch.tensor(lens)[:, None]).long()
    valid = mask.reshape(-1, 1).float()
    keep = torch.from_numpy(D.attention_mask(B, heads, S, p, seed, site))
    ref_in = qkv.float().requires_grad_(True)
    ref = _attention_reference(ref_in, lens, S, heads, keep, p)
    (ref * dctx.float() * valid).sum().backward()
    dq_ref = ref_in.grad * valid
    # padded layout
    ctx, lse = K.attention_dropout_fwd(qkv.to(dev), mask.to(dev), B, S, heads, p, seed, site)
    r, c = _rel(ctx.float() * valid.to(dev), ref * valid)
    assert r < 2e-2 and c > 0.9995, (r, c)
    plain, lse0 = K.attention_fwd(qkv.to(dev), mask.to(dev), B, S, heads)
    assert torch.equal(lse, lse0)                                 # the log-sum-exp is that of the undropped softmax
    assert _rel(plain.float() * valid.to(dev), ref * valid)[0] > 0.1          # and the mask does act
    dqkv = K.attention_dropout_bwd(qkv.to(dev), mask.to(dev), ctx, lse, (dctx * valid.to(torch.bfloat16)).to(dev), B, S, heads, p, seed, site)
    r, c = _rel(dqkv.float() * valid.to(dev), dq_ref)
    assert r < 3e-2 and c > 0.999, (r, c)
    if S > 256:                      # (the tiled backward above; the packed layout is used up to S = 256 only)
        return
    # packed layout: the same sequences stored back to back draw the same masks (index = position inside the sequence)
    rows = torch.cat([b * S + torch.arange(n) for b, n in enumerate(lens)])
    cu = torch.zeros(B + 1, dtype=torch.int32)
    cu[1:] = torch.tensor(lens, dtype=torch.int32).cumsum(0)
    qp = qkv[rows].contiguous().to(dev)
    ctx_p, lse_p = K.attention_dropout_fwd(qp, None, B, S, heads, p, seed, site, cu=cu.to(dev))
    assert torch.equal(ctx_p.cpu(), ctx.cpu()[rows])
    dq_p = K.attention_dropout_bwd(qp, None, ctx_p, lse_p, dctx[rows].contiguous().to(dev), B, S, heads, p, seed, site, cu=cu.to(dev))
    r, c = _rel(dq_p, dq_ref[rows])
    assert r < 3e-2 and c > 0.999, (r, c)
    # a micro-batch that starts at sequence 1 of the batch reproduces that sequence's rows
    one = qkv[S:2 * S].contiguous().to(dev)
    ctx_1, _ = K.attention_dropout_fwd(one, mask[1:2].contiguous().to(dev), 1, S, heads, p, seed, site, first_sequence=1)
    assert torch.equal((ctx_1.float() * valid[S:2 * S].to(dev)).cpu(), (ctx.float() * valid.to(dev)).cpu()[S:2 * S])


def test_attention_dropout_masks_do_not_repeat_across_large_batches(dev):
    """round 2's index ((bh * 512 + q) * 512 + k wrapped at bh = 16384, so sequence b and b + 1366 (12 heads) shared their
    masks.  The sequence-head number now keys the hash: a micro-batch starting at sequence 1366 / 5000 draws the oracle's mask for THOSE
    sequences, and that mask differs from sequence 0's."""
    from mmgclip import kernels as K
    heads, p, seed, site, S, B = 12, 0.1, 77, 9, 32, 2
    Hd = heads * 64
    g = torch.Generator().manual_seed(3)
    qkv = (torch.randn(B * S, 3 * Hd, generator=g) * 0.8).to(torch.bfloat16)
    mask = torch.ones(B, S, dtype=torch.long)
    keep0 = D.attention_mask(B, heads, S, p, seed, site)
    outs = {}
    for first in (0, 1366, 5000):
        keep = D.attention_mask(B, heads, S, p, seed, site, first_sequence=first)
        if first:
            assert (keep != keep0).mean() > 0.1
        ref = _attention_reference(qkv.float(), [S] * B, S, heads, torch.from_numpy(keep), p)
        ctx, _ = K.attention_dropout_fwd(qkv.to(dev), mask.to(dev), B, S, heads, p, seed, site, first_sequence=first)
        r, c = _rel(ctx, ref)
        assert r < 2e-2 and c > 0.9995, (first, r, c)
        outs[first] = ctx.float().cpu()
    assert _rel(outs[1366], outs[0])[0] > 0.1 and _rel(outs[5000], outs[1366])[0] > 0.1


def _tower(dev, layers=3, seed=4):
    from mmgclip.networks.bert import BertConfigLite
    from mmgclip.networks.encoder import BertEncoder
    from tests.test_towers_gpu import _randomize
    torch.manual_seed(0)
    cfg = BertConfigLite(vocab_size=3000, num_hidden_layers=layers)
    enc = BertEncoder(pretrained=None, random_init=True, freeze=False, config=cfg)
    _randomize(enc, seed)
    sd = {k[len("model."):]: v.clone() for k, v in enc.state_dict().items()}
    return enc.to(dev), sd


@pytest.mark.parametrize("packed,S", [(False, 77), (True, 77), (True, 300)])
def test_bert_tower_training_mode_matches_the_oracle_mask_for_mask(dev, packed, S):
    """S = 300: beyond the whole-sequence backward (and beyond packing): the tiled dQ / dK,dV kernels regenerate the same masks."""
    from mmgclip.dataset.synthetic import synthetic_tokens
    enc, sd = _tower(dev, layers=3 if S == 77 else 2)
    tok = synthetic_tokens(4, S, 3000, torch.Generator().manual_seed(5))
    wgt = torch.randn(4 * S, 768, generator=torch.Generator().manual_seed(6))
    valid = tok["attention_mask"].reshape(-1, 1).float()
    seed = 0x5EED_0000_0001
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = E.bert_forward(osd, tok["input_ids"], tok["attention_mask"], tok["token_type_ids"], dropout=(0.1, 0.1, seed))
    (ref.reshape(-1, 768) * wgt * valid).sum().backward()
    with torch.no_grad():
        plain = E.bert_forward(sd, tok["input_ids"], tok["attention_mask"], tok["token_type_ids"])
    enc.train()
    enc.next_dropout_seed = seed
    hid = enc.hidden_states({k: v.to(dev) for k, v in tok.items()}, packed=packed)
    r, c = _rel(hid.float() * valid.to(dev), ref.reshape(-1, 768) * valid)
    assert r < 3e-2 and c > 0.999, (r, c)
    assert _rel(hid.float() * valid.to(dev), plain.reshape(-1, 768) * valid)[0] > 0.2        # far from the eval-mode output
    (hid.float() * (wgt * valid).to(dev)).sum().backward()
    bad = {}
    for name, p in enc.model.named_parameters():
        if name.startswith("pooler.") or name.endswith("attention.self.key.bias"):
            continue
        r, c = _rel(p.grad, osd[name].grad)
        if not (c > 0.99 and r < 0.12):
            bad[name] = (r, c)
    assert not bad, f"{len(bad)} gradients off: {list(bad.items())[:8]}"


def test_bert_tower_dropout_switches(dev, monkeypatch):
    """train() draws a fresh seed per forward from the tower's own generator (torch's global CPU stream is left alone: ADVICE r2);
    eval(), dropout=False and MMG_BERT_DROPOUT=0 give the deterministic tower; `seeding(s)` reproduces a training-mode run;
    another rank draws other masks; micro-batching does not change the masks."""
    from mmgclip.dataset.synthetic import synthetic_tokens
    from mmgclip.utils.global_utils import seeding
    enc, sd = _tower(dev, layers=2)
    tok = {k: v.to(dev) for k, v in synthetic_tokens(6, 40, 3000, torch.Generator().manual_seed(1)).items()}
    run = lambda: enc.hidden_states(tok).detach().clone()              # noqa: E731
    enc.eval()
    e0, e1 = run(), run()
    assert torch.equal(e0, e1)
    enc.train()
    seeding(11)
    state = torch.get_rng_state()
    t0, t1 = run(), run()
    assert not torch.equal(t0, t1) and not torch.equal(t0, e0)
    assert torch.equal(torch.get_rng_state(), state)                   # the global CPU generator was not consumed
    seeding(11)
    assert torch.equal(run(), t0)
    monkeypatch.setenv("RANK", "1")                                    # the same base seed on another data-parallel rank
    seeding(11)
    assert not torch.equal(run(), t0)
    monkeypatch.setenv("RANK", "0")
    enc.next_dropout_seed = 5
    whole = run()
    enc.micro_batch = 4
    enc.next_dropout_seed = 5
    assert _rel(run(), whole)[0] < 1e-3                                # (a different mask would be an O(1) difference)
    enc.micro_batch = 4096
    enc.dropout = False
    assert torch.equal(run(), e0)
    enc.dropout = True
    monkeypatch.setenv("MMG_BERT_DROPOUT", "0")
    assert torch.equal(run(), e0)


def test_whole_model_step_with_dropout_is_the_same_on_one_or_two_streams(dev, monkeypatch):
    """Product defaults together: HF training-mode dropout live in the text tower AND the text tower on its side stream.  With the
    same torch seed the masks are the same, so the one-stream run must give the same losses and gradients; and a second run with the
    same seed reproduces the first bit for bit in the loss."""
    from tests.test_model_gpu import _cfg, _small_bert
    from mmgclip.dataset.synthetic import synthetic_batch
    from mmgclip.loss.loss_controller import create_loss
    from mmgclip.networks.mmgclip_model import MMGCLIP
    _small_bert(monkeypatch)
    res = {}
    for tag, stream in (("two", "1"), ("one", "0"), ("two_again", "1")):
        monkeypatch.setenv("MMG_TEXT_STREAM", stream)
        torch.manual_seed(0)
        cfg = _cfg("networks=clip_convnexttiny_bert_pixels", "tokenizer=bert_clinical_seqlen=77", "networks/dropout=dropout0",
                   "networks.image_encoder.micro_batch=4", "networks.image_encoder.image_size=64", "loss.config.loss_name=MMGCLIPLoss")
        model = MMGCLIP(cfg).train()
        assert model.text_encoder.dropout and model.text_encoder.training
        crit = create_loss("MMGCLIPLoss")()
        from mmgclip.utils.global_utils import seeding
        seeding(123)                                               # restarts the tower's dropout-seed stream
        losses = []
        for step in range(2):
            batch = synthetic_batch(8, S=77, image_size=64, vocab_size=3000, seed=9 + step, with_impression=True)
            model.zero_grad(set_to_none=True)
            loss, _ = crit(**model(batch, materialize_logits=False))
            loss.backward()
            losses.append(loss.item())
        torch.cuda.current_stream().synchronize()
        res[tag] = (losses, {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None})
    assert res["two"][0] == res["one"][0] == res["two_again"][0], (res["two"][0], res["one"][0], res["two_again"][0])
    for n, g0 in res["one"][1].items():
        g1 = res["two"][1][n]
        assert float((g0 - g1).abs().max()) <= 1e-5 * float(g0.abs().max()) + 1e-9, n
    # and dropout really is on: the deterministic tower gives another loss
    monkeypatch.setenv("MMG_BERT_DROPOUT", "0")
    torch.manual_seed(0)
    model = MMGCLIP(_cfg("networks=clip_convnexttiny_bert_pixels", "tokenizer=bert_clinical_seqlen=77", "networks/dropout=dropout0",
                         "networks.image_encoder.micro_batch=4", "networks.image_encoder.image_size=64",
                         "loss.config.loss_name=MMGCLIPLoss")).train()
    batch = synthetic_batch(8, S=77, image_size=64, vocab_size=3000, seed=9, with_impression=True)
    loss, _ = create_loss("MMGCLIPLoss")()(**model(batch, materialize_logits=False))
    assert abs(loss.item() - res["two"][0][0]) > 1e-4
