"""self-attention core: the fp32 MFMA form against the staged two-plane f16 form (events over 100 launches)."""
import sys, torch
sys.path.insert(0, '.')
import bench; bench._imports()
import ctypes as C
from transcar_amd import ops, _lib as L
dev = torch.device('cuda:0')
Q, Cd, H = 900, 256, 8
for B in (1, 2, 4, 9):
    qk = torch.randn(B, Q, 2 * Cd, device=dev) * 0.5
    vt = torch.zeros(B, Cd, 912, device=dev); vt[:, :, :Q] = torch.randn(B, Cd, Q, device=dev)
    o1 = torch.empty(B, Q, Cd, device=dev); o2 = torch.empty(B, Q, Cd, device=dev)
    ws = torch.empty(max(1, L.lib().tc_sdpa_f16x2_workspace_bytes(B, Q, H)), dtype=torch.uint8, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def f32(): L.lib().tc_sdpa_fwd(qk.data_ptr(), qk.data_ptr() + 4 * Cd, 2 * Cd, vt.data_ptr(), 912, o1.data_ptr(), Cd, B, Q, H, st)
    def h(): L.lib().tc_sdpa_fwd_f16x2(qk.data_ptr(), vt.data_ptr(), 912, o2.data_ptr(), Cd, B, Q, H, ws.data_ptr(), ws.numel(), st)
    res = []
    for name, fn in (('f32', f32), ('f16x2 staged', h)):
        for _ in range(5): fn()
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): fn()
        e1.record(); torch.cuda.synchronize()
        res.append('%s %.1f us' % (name, e0.elapsed_time(e1) / 100 * 1e3))
    # float64 reference of the same operands
    q = qk[..., :Cd].double().view(B, Q, H, 32).transpose(1, 2); k = qk[..., Cd:].double().view(B, Q, H, 32).transpose(1, 2)
    v = vt[:, :, :Q].double().view(B, H, 32, Q).transpose(2, 3)
    ref = (torch.softmax((q @ k.transpose(2, 3)) * 0.6931471805599453, -1) @ v).transpose(1, 2).reshape(B, Q, Cd)
    print('B', B, ' | '.join(res), '| max err f32 %.2e f16x2 %.2e | f32 vs f16x2 %.2e' % (
        float((o1.double() - ref).abs().max()), float((o2.double() - ref).abs().max()), float((o1 - o2).abs().max())))
