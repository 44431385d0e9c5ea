import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench; bench._imports()
from transcar_amd import ops
dev = torch.device('cuda:0')
rng = np.random.RandomState(7)
B, H, D, Q = 1, 8, 32, 900
C = H * D
q = rng.standard_normal((B, Q, C)).astype(np.float32); k = rng.standard_normal((B, Q, C)).astype(np.float32); v = rng.standard_normal((B, Q, C)).astype(np.float32)
def run(bad_keys, off, heads=range(H), ones_v=False, tag=''):
    f = rng.uniform(-3, 3, Q).astype(np.float32); f[bad_keys] = off
    qq, kk = q.copy(), k.copy()
    for h in heads:
        qq[:, :, h * D] = 1.0; kk[:, :, h * D] = f[None, :]
    qs = torch.from_numpy(qq) * (1.4426950408889634 / np.sqrt(D))
    vv = np.ones_like(v) if ones_v else v
    vt = torch.zeros((B, C, 912)); vt[:, :, :Q] = torch.from_numpy(vv).permute(0, 2, 1)
    a = ops.sdpa(qs.to(dev), torch.from_numpy(kk).to(dev), vt.to(dev), matrix_path='f16x2').cpu()
    nan = torch.isnan(a)[0]            # [Q, C]
    per_head = nan.view(Q, H, D).any(-1).sum(0).tolist()
    rows = nan.any(-1).nonzero().flatten()
    print(tag, 'nan per head (queries)', per_head, 'first nan queries', rows[:6].tolist(), 'inf', int(torch.isinf(a).sum()))
run(np.arange(16), -60.0, tag='keys 0-15 all heads      ')
run(np.arange(16), -60.0, heads=[0], tag='keys 0-15 head 0 only    ')
run(np.arange(16), -60.0, ones_v=True, tag='keys 0-15, V = 1         ')
run(np.arange(32, 48), -60.0, tag='keys 32-47 (wave 1 first)')
run(np.arange(256, 272), -60.0, tag='keys 256-271 (wave 0 2nd)')
run(np.arange(0, 4), -60.0, tag='keys 0-3                 ')
run(np.array([5]), -60.0, tag='key 5                    ')
run(np.array([5]), -30.0, tag='key 5 -30                ')
run(np.array([5]), -45.0, tag='key 5 -45                ')
