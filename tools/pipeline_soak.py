#!/usr/bin/env python3
"""Soak test of transcar_amd.pipeline.FramePipeline: thousands of overlapping launches on static
inputs must keep giving bit-identical outputs per lane (a lane sharing a buffer with another would not;
neither would a weight fragment overwritten before its MFMAs have read it -- the 16-row tiles refill
their one weight buffer in place).
    python tools/pipeline_soak.py [launches] [frames per launch: 1 (4-row tiles) | 2 (8) | 4 (16) | 9 (16, 507 workgroups: the bench default)]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                              # noqa: E402
from transcar_amd.pipeline import FramePipeline            # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    fpl = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dev = torch.device('cuda:0')
    torch.set_grad_enabled(False)
    head, _ = bench.build_head(dev)
    lanes = [bench.make_inputs(head, dev, 'res101', fpl, seed=1 + i, host_feats=fpl <= 2) for i in range(3)]
    pipe = FramePipeline(head, lanes)
    for _ in range(3):
        pipe.launch()
    pipe.synchronize()
    ref = [[pipe.outputs[i][0]['all_cls_scores'].clone(), pipe.outputs[i][0]['all_bbox_preds'].clone(),
            pipe.outputs[i][1][0].clone(), pipe.outputs[i][1][1].clone()] for i in range(3)]
    bad = 0
    for it in range(n):
        lane, (outs, dec) = pipe.launch()
        if it % 97 == 0:                     # read a lane back while the others keep running
            pipe.wait(lane)
            got = [outs['all_cls_scores'], outs['all_bbox_preds'], dec[0], dec[1]]
            bad += sum(0 if torch.equal(a, b) else 1 for a, b in zip(got, ref[lane]))
    pipe.synchronize()
    for i in range(3):
        outs, dec = pipe.outputs[i]
        got = [outs['all_cls_scores'], outs['all_bbox_preds'], dec[0], dec[1]]
        bad += sum(0 if torch.equal(a, b) else 1 for a, b in zip(got, ref[i]))
    print('launches %d of %d frame(s), mismatching tensors %d' % (n, fpl, bad))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
