import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mmg-clip_amd"))
import torch
from mmgclip import kernels as K
dev = torch.device("cuda")
n, H, C = 16, 256, 96
x = torch.randn(n * H * H, C, device=dev).bfloat16(); w = torch.randn(49, C, device=dev) * 0.1; b = torch.randn(C, device=dev)
out = torch.empty_like(x)
for _ in range(3):
    K.dwconv7(x, w, b, n, H, H, C, out=out)
torch.cuda.synchronize()
dy = torch.randn_like(x); dw = torch.zeros(49, C, device=dev); db = torch.zeros(C, device=dev)
for _ in range(3):
    K.dwconv7_wgrad(x, dy, dw, db, n, H, H, C)
torch.cuda.synchronize()
