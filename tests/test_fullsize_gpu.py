"""Size-independent properties at BASELINE.json's full sizes (config C2: 1024 x 1024 x 1 images, 77-token prompts), where the
fp32 CPU oracle is too slow to be the checker: micro-batch independence, linearity of the linear kernels on the full stage-1
geometry, and agreement of the fused / unfused and packed / padded code paths on the real shapes."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().float().flatten().double(), b.detach().float().flatten().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_convnext_1024_micro_batch_independence_and_fused_vs_gemm_pair(dev, monkeypatch):
    """Features and parameter gradients of 1024^2 images do not depend on how the batch is cut into micro-batches, and the
    fused CNBlock kernels agree with the LayerNorm + GEMM + GEMM path on the full-resolution geometry (256^2 ... 32^2 maps)."""
    from mmgclip.networks.encoder import ConvNextTinyEncoder
    torch.manual_seed(0)
    img = torch.rand(3, 1, 1024, 1024, generator=torch.Generator().manual_seed(1)).to(dev)
    wgt = torch.randn(3, 768, generator=torch.Generator().manual_seed(2)).to(dev)
    ref_tower = ConvNextTinyEncoder(micro_batch=3)
    state = {k: v.clone() for k, v in ref_tower.state_dict().items()}
    results = {}
    for name, mb, fused in (("mb3", 3, "1"), ("mb1", 1, "1"), ("unfused", 2, "0")):
        monkeypatch.setenv("MMG_FUSED_MLP", fused)
        tower = ConvNextTinyEncoder(micro_batch=mb)
        tower.load_state_dict(state)
        tower = tower.to(dev)
        feat = tower(img)
        (feat * wgt).sum().backward()
        results[name] = (feat.detach().clone(), {n: p.grad.detach().clone() for n, p in tower.model.named_parameters()})
        del tower
        torch.cuda.empty_cache()
    f0, g0 = results["mb3"]
    assert torch.isfinite(f0).all() and f0.shape == (3, 768)
    # same kernels, different batching: equal up to the order of fp32 atomics (average pool, weight gradients)
    assert _rel(results["mb1"][0], f0) < 1e-5
    for n, g in results["mb1"][1].items():
        assert _rel(g, g0[n]) < 2e-3, n
    # different kernels (fused vs LayerNorm + two GEMMs): bf16-level agreement
    assert _rel(results["unfused"][0], f0) < 1e-2
    worst = max((_rel(g, g0[n]), n) for n, g in results["unfused"][1].items())
    assert worst[0] < 6e-2, worst


def test_convnext_256_images_in_one_micro_batch_equals_four_of_64(dev):
    """The benchmarked configuration: 256 images of 1024^2 as ONE micro-batch.  Its stage-1 hidden tensors hold
    256 * 65536 * 384 = 6.4e9 elements (> 2^32), so this is also the addressing test of every kernel on the path: features
    and parameter gradients must equal those of four 64-image micro-batches."""
    from mmgclip.networks.encoder import ConvNextTinyEncoder
    torch.manual_seed(0)
    img = torch.rand(256, 1, 1024, 1024, generator=torch.Generator().manual_seed(1)).to(dev)
    wgt = torch.randn(256, 768, generator=torch.Generator().manual_seed(2)).to(dev)
    state = None
    results = {}
    for mb in (256, 64):
        tower = ConvNextTinyEncoder(micro_batch=mb)
        if state is None:
            state = {k: v.clone() for k, v in tower.state_dict().items()}
        tower.load_state_dict(state)
        tower = tower.to(dev)
        feat = tower(img)
        (feat * wgt).sum().backward()
        results[mb] = (feat.detach().clone(), {n: p.grad.detach().clone() for n, p in tower.model.named_parameters()})
        del tower, feat
        torch.cuda.empty_cache()
    assert torch.isfinite(results[256][0]).all()
    assert _rel(results[256][0], results[64][0]) < 1e-5
    for n, g in results[256][1].items():
        assert _rel(g, results[64][1][n]) < 2e-3, n


def test_dwconv_and_gemm_linearity_on_stage1_geometry(dev):
    """conv(a x1 + x2) = a conv(x1) + conv(x2) (no bias) and the same for the NT GEMM, on one image of the 256 x 256 x 96 map."""
    from mmgclip import kernels as K, linalg as L
    n, H, C = 1, 256, 96
    g = torch.Generator().manual_seed(3)
    x1 = torch.randn(n * H * H, C, generator=g).to(dev)
    x2 = torch.randn(n * H * H, C, generator=g).to(dev)
    w49 = (0.1 * torch.randn(49, C, generator=g)).to(dev)
    a = 0.5                                                 # exact in bf16: the combination is formed before rounding
    mix = (a * x1 + x2).to(torch.bfloat16)
    y_mix = K.dwconv7(mix, w49, None, n, H, H, C).float()
    y_sum = a * K.dwconv7(x1.to(torch.bfloat16), w49, None, n, H, H, C).float() + K.dwconv7(x2.to(torch.bfloat16), w49, None, n, H, H, C).float()
    assert _rel(y_mix, y_sum) < 1e-2
    wq = (torch.randn(384, C, generator=g) / C ** 0.5).to(dev).to(torch.bfloat16)
    z_mix = L.gemm_nt(mix, wq, out_dtype=torch.float32)
    z_sum = a * L.gemm_nt(x1.to(torch.bfloat16), wq, out_dtype=torch.float32) + L.gemm_nt(x2.to(torch.bfloat16), wq, out_dtype=torch.float32)
    assert _rel(z_mix, z_sum) < 1e-2


def test_bert_base_77_tokens_packed_equals_padded_at_batch_256(dev):
    """BERT-base on 256 prompts of up to 77 tokens: the unpadded path returns the padded path's [SEP] rows (what the model reads)."""
    from mmgclip.networks.encoder import BertEncoder
    from mmgclip.networks.bert import EosPool
    from mmgclip.dataset.synthetic import synthetic_tokens
    torch.manual_seed(0)
    enc = BertEncoder(pretrained=None, random_init=True, freeze=True).to(dev)
    tok = {k: v.to(dev) for k, v in synthetic_tokens(256, 77, 28996, torch.Generator().manual_seed(5)).items()}
    with torch.no_grad():
        pooled = [EosPool.apply(enc.hidden_states(tok, packed=p), tok["attention_mask"], 256, 77) for p in (False, True)]
    assert pooled[0].shape == (256, 768) and torch.isfinite(pooled[0]).all()
    assert _rel(pooled[1], pooled[0]) < 5e-3
