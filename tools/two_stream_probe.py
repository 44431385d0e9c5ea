#!/usr/bin/env python3
"""Frames in flight: N hipGraphs of one frame each (own head workspace, own inputs) replayed
round-robin on N streams vs one graph on one stream.  Usage: two_stream_probe.py [nstreams] [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                              # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    dev = torch.device('cuda:0')
    torch.set_grad_enabled(False)
    lanes = []
    for i in range(n):
        head, _ = bench.build_head(dev)
        inp = bench.make_inputs(head, dev, 'res101', 1, seed=1 + i)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                bench.one_step(head, inp)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out = bench.one_step(head, inp)
        lanes.append((head, inp, s, g, out))
    torch.cuda.synchronize()
    for use in (1, n):
        for w in range(20):
            with torch.cuda.stream(lanes[w % use][2]):
                lanes[w % use][3].replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            with torch.cuda.stream(lanes[i % use][2]):
                lanes[i % use][3].replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('%d stream(s): %.4f ms/frame, %.1f frames/s' % (use, dt / steps * 1e3, steps / dt))


if __name__ == '__main__':
    main()
