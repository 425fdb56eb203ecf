"""Phase split of the fused CNBlock forward from in-kernel s_memtime probes (debug builds: -DMLP_PROBE=1|2, see cnblock_mlp.hip).
Run on the GPU box with MMGCLIP_HIP_LIB=tools/libmmg_ab_probe{1,2}.so."""
import sys, os, ctypes, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mmg-clip_amd"))
import torch
from mmgclip import kernels as K, _hip

lib = _hip.load()
probe = lib.mmg_debug_mlp_probe
probe.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda")
for C, M in ((96, 1 << 22), (192, 1 << 20), (384, 1 << 18)):
    g = torch.Generator().manual_seed(0)
    xd = torch.randn(M // 64, C, generator=g).to(torch.bfloat16).repeat(64, 1).to(dev)
    res = torch.randn_like(xd)
    lnw, lnb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    w1, b1 = (torch.randn(4 * C, C, generator=g) / C ** 0.5).to(dev), torch.zeros(4 * C, device=dev)
    w2, b2 = (torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5).to(dev), torch.zeros(C, device=dev)
    gamma = torch.ones(C, device=dev)
    packed = K.cnblock_pack(w1, w2)
    run = lambda: K.cnblock_mlp_fwd(xd, lnw, lnb, 1e-6, packed, b1, b2, gamma, res, want_hpre=False, want_stats=False)
    for _ in range(3): run()
    torch.cuda.synchronize()
    probe(None, 1)
    t = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    out = (ctypes.c_ulonglong * 8)()
    probe(out, 1)
    tot, wait, g1, gelu, g2, waves = [int(v) for v in out[:6]]
    print(f"C={C} M={M}: {dt*1e3:.3f} ms/launch; per wave: total {tot/waves:.0f} ticks, chunk wait+barrier {100*wait/tot:.1f} %, "
          f"GEMM1 {100*g1/tot:.1f} %, GELU {100*gelu/tot:.1f} %, GEMM2 {100*g2/tot:.1f} %  (waves {waves // 10} per launch)", flush=True)
