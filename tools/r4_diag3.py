"""round 4 diagnosis 3: WHICH rows of the f16x2 radar chain go wrong (B = 8, two workgroups per CU)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench._imports()
from transcar_amd import ops
from transcar_amd.detr3d_head import head_options
import ctypes as C
from transcar_amd import _lib as L
dev = torch.device('cuda:0')
head, _ = bench.build_head(dev)
B = 8
inp = bench.make_inputs(head, dev, 'tiny', B, seed=71, host_feats=False)
o = head.forward_nhwc(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'], inp['pad_mult'], aux=True,
                      options=head_options(tile_rows=16, matrix_path='f32'))
torch.cuda.synchronize()
hs5 = o['aux']['inter_states'][-1].contiguous().clone()
ref5 = o['aux']['inter_references'][-1].contiguous().clone()
lbox = o['aux']['last_box'].contiguous().clone()
T = inp['tokens'].shape[1]
ws = torch.empty(L.lib().tc_head_workspace_bytes(C.byref(head._packed_view), B, T), dtype=torch.uint8, device=dev)

def run(mp, reuse=0, compact=None, layers=(0, 3)):
    opt = head_options(tile_rows=16, matrix_path=mp, radar_compact=compact)
    opt.reuse_radar_kv = reuse
    c, b, h = ops.radar_fusion(head, hs5, ref5, lbox, inp['tokens'], inp['pad_mult'], layers[0], layers[1], options=opt, ws=ws)
    torch.cuda.synchronize()
    return c.clone(), b.clone(), h.clone()

for compact in (False, None):
    r0 = run('f32', compact=compact)
    for rep in range(3):
        r1 = run('f16x2', reuse=1, compact=compact)
        d = (r1[0] - r0[0]).abs().amax(-1)            # [3, B, Q]
        db = (r1[1] - r0[1]).abs().amax(-1)
        bad = (d > 1e-3) | (db > 1e-3)
        idx = bad.nonzero()
        print('compact', compact, 'rep', rep, 'bad (layer, b, q):', idx.shape[0])
        first = {}
        for l, b, q in idx.tolist():
            first.setdefault((b, q), l)
        for (b, q), l in sorted(first.items())[:24]:
            print('   b %d q %3d (flat row %4d, tile %3d pos %2d) first bad layer %d  hits %s  cls diff %s box diff %s' % (
                b, q, b * 900 + q, (b * 900 + q) // 16, (b * 900 + q) % 16, l, r0[2][:, b, q].tolist(),
                ['%.1e' % x for x in d[:, b, q].tolist()], ['%.1e' % x for x in db[:, b, q].tolist()]))
print('single layers, f16x2 chain on f32 K|V, compact None:')
r0 = run('f32')
for rep in range(2):
    r1 = run('f16x2', reuse=1, layers=(0, 1))
    d = (r1[0][0] - r0[0][0]).abs().amax(-1); db = (r1[1][0] - r0[1][0]).abs().amax(-1)
    print('  layer 1 alone: bad rows', int(((d > 1e-3) | (db > 1e-3)).sum()), ' of which with hits', int((((d > 1e-3) | (db > 1e-3)) & (r0[2][0] > 0)).sum()),
          ' cls-only bad', int(((d > 1e-3) & (db <= 1e-3)).sum()), ' box-only bad', int(((d <= 1e-3) & (db > 1e-3)).sum()))
