import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench; bench._imports()
from transcar_amd import ops
dev = torch.device('cuda:0')
rng = np.random.RandomState(7)
B, H, D, Q = 1, 8, 32, 900
C = H * D
q = rng.standard_normal((B, Q, C)).astype(np.float32); k = rng.standard_normal((B, Q, C)).astype(np.float32); v = rng.standard_normal((B, Q, C)).astype(np.float32)
f = rng.uniform(-3, 3, Q).astype(np.float32); f[:16] = -60.0
for h in range(H):
    q[:, :, h * D] = 1.0; k[:, :, h * D] = f[None, :]
qs = torch.from_numpy(q) * (1.4426950408889634 / np.sqrt(D))
vt = torch.zeros((B, C, 912)); vt[:, :, :Q] = torch.from_numpy(v).permute(0, 2, 1)
a = ops.sdpa(qs.to(dev), torch.from_numpy(k).to(dev), vt.to(dev), matrix_path='f16x2').cpu()
torch.cuda.synchronize()
print('nan rows', torch.isnan(a)[0].any(-1).nonzero().flatten()[:8].tolist())
