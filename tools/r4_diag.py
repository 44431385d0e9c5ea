"""round 4 diagnosis: where does a frame of an 8-frame launch differ from the frame alone on the f16x2 path?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench._imports()
from transcar_amd.detr3d_head import head_options
dev = torch.device('cuda:0')
head, _ = bench.build_head(dev)
shapes = sys.argv[1] if len(sys.argv) > 1 else 'res101'
B = 8
inp = bench.make_inputs(head, dev, shapes, B, seed=71, host_feats=False)
for mp in ('f16x2', 'f32'):
    opt = head_options(tile_rows=16, matrix_path=mp)
    runs = []
    for rep in range(2):
        o = head.forward_nhwc(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'], inp['pad_mult'], aux=True, options=opt)
        torch.cuda.synchronize()
        runs.append({k: o[k].clone() for k in ('all_cls_scores', 'all_bbox_preds')} | {'hs': o['aux']['inter_states'].clone(), 'refs': o['aux']['inter_references'].clone(), 'hits': o['aux']['radar_hit_counts'].clone()})
    print(mp, 'run-to-run identical:', all(torch.equal(runs[0][k], runs[1][k]) for k in runs[0]))
    full = runs[0]
    for b in (0, 3, 7):
        one = head.forward_nhwc([f[6 * b:6 * b + 6] for f in inp['nhwc']], inp['l2i'][b:b + 1], inp['hw'],
                                inp['tokens'][b:b + 1], inp['pad_mult'], aux=True, options=opt)
        torch.cuda.synchronize()
        hs1, hsf = one['aux']['inter_states'][:, 0], full['hs'][:, b]
        for l in range(6):
            d = (hs1[l] - hsf[l]).abs()
            rows = (d.amax(-1) > 0).nonzero().flatten()
            print(mp, 'frame', b, 'decoder layer', l, 'max diff %.3e' % float(d.max()), 'rows differing', int(rows.numel()), rows[:8].tolist(), rows[-4:].tolist())
        d = (one['all_cls_scores'][:, 0] - full['all_cls_scores'][:, b]).abs()
        print(mp, 'frame', b, 'cls max diff per level', [float(x) for x in d.amax((1, 2))], 'hits equal', bool(torch.equal(one['aux']['radar_hit_counts'][:, 0], full['hits'][:, b])))
