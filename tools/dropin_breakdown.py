#!/usr/bin/env python3
"""Where the plugin entry's time goes when it is handed nine frames per call (VERDICT r4 item 5c:
`dropin_forward.batch9` against a one-lane FramePipeline):

    python tools/dropin_breakdown.py            (on the MI355X box; ~20 s)

Prints, for `outs = head(feats, metas); head.get_bboxes(outs, metas); synchronize()` with [9,6,256,H,W]
channels_last maps and raw radar sweeps in img_metas:
  * wall time per call (median / p99 of 60 calls) and the device time of the same call (HIP events),
  * the host time in front of the first kernel / inside forward / inside get_bboxes / waiting in the final sync,
  * cProfile's top functions by cumulative time over 30 calls,
  * the one-lane FramePipeline's time for the same nine frames (graph replay) for comparison."""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                              # noqa: E402
from transcar_amd import configs, synth                    # noqa: E402


def main():
    dev = torch.device('cuda:0')
    torch.set_grad_enabled(False)
    head, _ = bench.build_head(dev)
    shapes = configs.LEVEL_SHAPES['res101']
    nb = 9
    metas = [synth.make_img_metas(1, synth.make_lidar2img(), radar=synth.make_radar_frame(seed=2 + i))[0] for i in range(nb)]
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    feats = [torch.randn((nb * 6, 256, h, w), device=dev, generator=g).to(memory_format=torch.channels_last)
             for (h, w) in shapes]
    feats = [f.view(nb, 6, *f.shape[1:]) for f in feats]

    def call(stamps=None):
        t0 = time.perf_counter()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        outs = head(feats, metas)
        t1 = time.perf_counter()
        boxes = head.get_bboxes(outs, metas)
        t2 = time.perf_counter()
        e1.record()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        if stamps is not None:
            stamps.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0, e0.elapsed_time(e1) * 1e-3))
        return boxes
    for _ in range(8):
        call()
    st = []
    for _ in range(60):
        call(st)
    a = np.array(st) * 1e3
    med = np.median(a, 0)
    print('nine frames per call, 60 calls; ms per CALL (per frame = / 9)')
    print('  wall              median %.3f  p99 %.3f  max %.3f   (per frame %.4f)' % (med[3], np.percentile(a[:, 3], 99), a[:, 3].max(), med[3] / nb))
    print('  device (events)   median %.3f' % med[4])
    print('  host: forward() returns after        %.3f' % med[0])
    print('  host: get_bboxes() returns after     %.3f more (it reads the kept-row counts: a device sync)' % med[1])
    print('  host: final synchronize              %.3f' % med[2])
    slow = np.argsort(-a[:, 3])[:5]
    print('  the five slowest calls (index: wall, forward, get_bboxes):', ', '.join('%d: %.3f, %.3f, %.3f' % (i, a[i, 3], a[i, 0], a[i, 1]) for i in slow))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(30):
        call()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(22)
    print('\ncProfile, 30 calls, by cumulative time:')
    print('\n'.join(ln for ln in s.getvalue().split('\n') if ln.strip())[:6000])
    # the same nine frames through a one-lane pipeline (graph replay)
    from transcar_amd.pipeline import FramePipeline
    inp = bench.make_inputs(head, dev, 'res101', nb, seed=5, host_feats=False)
    pipe = FramePipeline(head, [inp])
    for _ in range(5):
        pipe.launch()
    pipe.synchronize()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        pipe.launch()
        pipe.synchronize()
        ts.append(time.perf_counter() - t0)
    print('\none-lane FramePipeline, the same nine frames per launch (graph replay + sync): median %.3f ms per launch = %.4f per frame'
          % (np.median(ts) * 1e3, np.median(ts) * 1e3 / nb))


if __name__ == '__main__':
    main()
