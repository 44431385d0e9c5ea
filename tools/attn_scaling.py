#!/usr/bin/env python3
"""Attention core (tc_sdpa_fwd) time against problem size: us per launch and fraction of the f32 MFMA peak
for B x 8 heads x Q x Q; separates the fixed cost of a launch from the per-tile rate.
    python tools/attn_scaling.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                              # noqa: E402
from transcar_amd import _lib as L                         # noqa: E402


def main():
    dev = torch.device('cuda:0')
    lib = L.lib()
    H, Cd = 8, 256
    pmc = len(sys.argv) > 1 and sys.argv[1] == 'pmc'     # under rocprofv3 --pmc: a few eager launches of one size
    sizes = ((2, 3600),) if pmc else None
    for B, Q in sizes or ((1, 900), (2, 900), (4, 900), (8, 900), (1, 1800), (1, 3600), (2, 3600), (1, 450), (16, 900)):
        qpad = ((Q + 15) // 16) * 16
        qk = torch.randn((B * Q, 2 * Cd), device=dev)
        vt = torch.randn((B, Cd, qpad), device=dev)
        ao = torch.empty((B * Q, Cd), device=dev)

        def run():
            L.check(lib.tc_sdpa_fwd(qk.data_ptr(), qk.data_ptr() + Cd * 4, 2 * Cd, vt.data_ptr(), qpad,
                                    ao.data_ptr(), Cd, B, Q, H, bench.cur_stream()), 'sdpa')
        if pmc:
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            continue
        ms = bench.time_events(run, iters=30)
        flop = 4.0 * Q * Q * 32 * H * B
        print('B %2d Q %4d: %7.1f us  %5.1f TFLOP/s = %4.1f %% of the f32 MFMA peak'
              % (B, Q, ms * 1e3, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100))


if __name__ == '__main__':
    main()
