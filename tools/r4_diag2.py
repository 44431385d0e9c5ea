"""round 4 diagnosis 2: the radar part (encoders / fusion chain) on the two matrix paths, B = 8, 16-row tiles"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench._imports()
from transcar_amd import ops
from transcar_amd.detr3d_head import head_options
dev = torch.device('cuda:0')
head, _ = bench.build_head(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
inp = bench.make_inputs(head, dev, 'tiny', B, seed=71, host_feats=False)
o = head.forward_nhwc(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'], inp['pad_mult'], aux=True,
                      options=head_options(tile_rows=16, matrix_path='f32'))
torch.cuda.synchronize()
hs5 = o['aux']['inter_states'][-1].contiguous().clone()
ref5 = o['aux']['inter_references'][-1].contiguous().clone()
lbox = o['aux']['last_box'].contiguous().clone()
T = inp['tokens'].shape[1]
import ctypes as C
from transcar_amd import _lib as L
ws = torch.empty(L.lib().tc_head_workspace_bytes(C.byref(head._packed_view), B, T), dtype=torch.uint8, device=dev)

def run(mp, reuse=0, compact=None):
    opt = head_options(tile_rows=16, matrix_path=mp, radar_compact=compact)
    opt.reuse_radar_kv = reuse
    c, b, h = ops.radar_fusion(head, hs5, ref5, lbox, inp['tokens'], inp['pad_mult'], 0, 3, options=opt, ws=ws)
    torch.cuda.synchronize()
    return c.clone(), b.clone(), h.clone()

def cmp(name, a, b):
    print('%-46s cls max diff %.3e  box %.3e  hits equal %s  rows with cls diff > 1e-3: %d' % (
        name, float((a[0] - b[0]).abs().max()), float((a[1] - b[1]).abs().max()), bool(torch.equal(a[2], b[2])),
        int(((a[0] - b[0]).abs().amax(-1) > 1e-3).sum())))

for compact in (None, False):
    print('--- radar_compact =', compact)
    r0 = run('f32', compact=compact)
    cmp('f32 again', run('f32', compact=compact), r0)
    h1 = run('f16x2', compact=compact)
    cmp('f16x2 enc + chain vs f32', h1, r0)
    cmp('f16x2 again vs itself', run('f16x2', compact=compact), h1)
    cmp('f16x2 third vs first', run('f16x2', compact=compact), h1)
    run('f32', compact=compact)                        # K | V of the f32 encoders in ws
    cmp('f32 K|V, f16x2 chain vs f32', run('f16x2', reuse=1, compact=compact), r0)
    cmp('f32 K|V, f16x2 chain again', run('f16x2', reuse=1, compact=compact), r0)
    run('f16x2', compact=compact)
    cmp('f16x2 K|V, f32 chain vs f32', run('f32', reuse=1, compact=compact), r0)
    run('f16x2', compact=compact)
    cmp('f16x2 K|V (2nd), f32 chain vs f32', run('f32', reuse=1, compact=compact), r0)
