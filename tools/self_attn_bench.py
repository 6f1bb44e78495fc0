import torch, sys
sys.path.insert(0, '.')
from nmrf_amd import kernels as K
T = 47 * 156 * 4
qkv = torch.randn(T, 384, device='cuda')
for _ in range(3): K.self_attn(qkv, 4, 4)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): K.self_attn(qkv, 4, 4)
e1.record(); torch.cuda.synchronize()
print("self_attn KITTI B=1: %.2f us / call" % (e0.elapsed_time(e1) * 1e3 / 50))
