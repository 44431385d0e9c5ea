import sys, torch
sys.path.insert(0, '.')
import bench; bench._imports()
import ctypes as C
from transcar_amd import ops, _lib as L
dev = torch.device('cuda:0')
B, Q, Cd, H = 9, 900, 256, 8
qk = torch.randn(B, Q, 2 * Cd, device=dev) * 0.5
vt = torch.zeros(B, Cd, 912, device=dev); vt[:, :, :Q] = torch.randn(B, Cd, Q, device=dev)
out = torch.empty(B, Q, Cd, device=dev)
ws = torch.empty(L.lib().tc_sdpa_f16x2_workspace_bytes(B, Q, H), dtype=torch.uint8, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def f32(): L.lib().tc_sdpa_fwd(qk.data_ptr(), qk.data_ptr() + 4 * Cd, 2 * Cd, vt.data_ptr(), 912, out.data_ptr(), Cd, B, Q, H, st)
def h(): L.lib().tc_sdpa_fwd_f16x2(qk.data_ptr(), vt.data_ptr(), 912, out.data_ptr(), Cd, B, Q, H, ws.data_ptr(), ws.numel(), st)
for name, fn in (('f32', f32), ('f16x2 (planes + core)', h)):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): fn()
    e1.record(); torch.cuda.synchronize()
    print(sys.argv[1] if len(sys.argv) > 1 else '', name, '%.1f us' % (e0.elapsed_time(e1) / 100 * 1e3))
