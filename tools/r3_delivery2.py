#!/usr/bin/env python3
"""Input delivery into the lanes: the producer's copies on the default stream (bench.py with_input_delivery) vs on a
stream per lane.  python tools/r3_delivery2.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                     # noqa: E402

bench._imports()
from transcar_amd.pipeline import FramePipeline, resident_frames_per_launch      # noqa: E402


def main():
    dev = torch.device('cuda:0')
    torch.set_grad_enabled(False)
    head, _ = bench.build_head(dev)
    P = resident_frames_per_launch(head.num_query, dev)
    lanes = [bench.make_inputs(head, dev, 'res101', P, seed=11 + 7 * i, host_feats=False) for i in range(3)]
    pipe = FramePipeline(head, lanes)
    host = [dict(tokens=pipe.inputs[i]['tokens'].cpu().pin_memory(), l2i=pipe.inputs[i]['l2i'].cpu().pin_memory())
            for i in range(pipe.lanes)]
    prod = [torch.cuda.Stream() for _ in range(pipe.lanes)]
    state = {'i': 0}

    def step(mode):
        i = state['i']
        state['i'] = (i + 1) % pipe.lanes
        if mode == 'none':
            pipe.launch(i)
        elif mode == 'default':
            pipe.write_inputs(i, l2i=host[i]['l2i'], tokens=host[i]['tokens'])
            pipe.launch(i)
        else:
            with torch.cuda.stream(prod[i]):
                pipe.write_inputs(i, l2i=host[i]['l2i'], tokens=host[i]['tokens'])
                pipe.launch(i)
    for mode in ('none', 'default', 'per-lane', 'none', 'default', 'per-lane'):
        for _ in range(12):
            step(mode)
        torch.cuda.synchronize()
        n = 120
        t0 = time.perf_counter()
        for _ in range(n):
            step(mode)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('%-9s %.0f frames/s' % (mode, n * P / dt))


if __name__ == '__main__':
    main()
