import hashlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from transcar_amd import detr3d_head as D
dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
head, _ = bench.build_head(dev)
for rows in (32, 16, 8, 4):
    for B in (9, 2):
        inp = bench.make_inputs(head, dev, 'res101', B, seed=3, host_feats=False)
        o = head.forward_nhwc(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'], inp['pad_mult'], options=D.head_options(tile_rows=rows))
        h = hashlib.sha256()
        for k in ('all_cls_scores', 'all_bbox_preds'):
            h.update(o[k].contiguous().cpu().numpy().tobytes())
        print('rows', rows, 'B', B, h.hexdigest()[:16], flush=True)
