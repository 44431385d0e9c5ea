#!/usr/bin/env python3
"""Host-side profile of the training iteration (cProfile), and the iteration's wall time with the pieces
bracketed by device syncs:   gpurun -- 'python tools/r3_trainhost.py'"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                              # noqa: E402
import transcar_amd as T                                   # noqa: E402
from transcar_amd import configs, synth                    # noqa: E402
from transcar_amd.trainer import FusionTrainer             # noqa: E402

dev = torch.device('cuda:0')
head, sd = bench.build_head(dev)
inp = bench.make_inputs(head, dev, 'res101', 1, seed=1)
cfg = configs.head_cfg()
cfg['train_cfg'] = configs.train_cfg_pts
thead = T.build_head(cfg)
thead.load_state_dict(head.state_dict(), strict=True)
thead = thead.to(dev)
boxes, labels = synth.make_gt(seed=7, n=24)
gt = torch.from_numpy(boxes).clone()
gt[:, 2] += gt[:, 5] * 0.5
gts, lbs = [gt.to(dev)], [torch.from_numpy(labels).to(dev)]
torch.set_grad_enabled(True)
tr = FusionTrainer(thead)


def step():
    return tr.step_fused_nhwc(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'], inp['pad_mult'], gts, lbs)


for _ in range(30):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
torch.cuda.synchronize()
print('wall per iteration: %.3f ms' % ((time.perf_counter() - t0) / 200 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(38)
print(s.getvalue()[:7000])
