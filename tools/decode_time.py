#!/usr/bin/env python3
"""box_decode_kernel alone, HIP-event timed: python tools/decode_time.py  (B = 1 and 9, 900 x 10 scores, top 300)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transcar_amd import configs, ops                      # noqa: E402


def main():
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(3)
    pcr = configs.pts_bbox_head['bbox_coder']['post_center_range']
    for B in (1, 9):
        cls = torch.from_numpy(rng.standard_normal((B, 900, 10)).astype(np.float32) - 2.0).to(dev)
        box = torch.from_numpy(rng.standard_normal((B, 900, 10)).astype(np.float32) * 0.3).to(dev)
        for _ in range(5):
            ops.box_decode_topk(cls, box, pcr, 300)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 200
        e0.record()
        for _ in range(n):
            ops.box_decode_topk(cls, box, pcr, 300)
        e1.record()
        torch.cuda.synchronize()
        print('B=%d: %.2f us per call (back to back, launch included)' % (B, e0.elapsed_time(e1) / n * 1e3))
        from transcar_amd import _lib as L
        dll = L.lib()
        if hasattr(dll, 'tc_debug_decode_stamps'):        # STAMPS=1 build: s_memtime per phase, workgroup 0 thread 0
            import ctypes as C
            buf = np.zeros(16, dtype=np.int64)
            dll.tc_debug_decode_stamps.argtypes = [C.c_void_p]
            assert dll.tc_debug_decode_stamps(buf.ctypes.data) == 0
            names = ['entry', 'keys+zero', 'bucket hist', 'bucket chosen', 'filed', 'byte pass 1', 'byte pass 2', 'byte pass 3', 'byte pass 4',
                     'byte pass 5', 'byte pass 6', 'selected', 'compacted', 'ranked', 'written']
            print('  stamps (ticks since entry):', ', '.join('%s %d' % (nm, buf[i] - buf[0]) for i, nm in enumerate(names) if buf[i] >= buf[0] and buf[i] > 0))


if __name__ == '__main__':
    main()
