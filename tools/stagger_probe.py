#!/usr/bin/env python3
"""Round 6: what are two INDEPENDENT tiles per CU worth when they are kept half a layer out of phase?

The 16-row f16x2 decoder chain runs two workgroups per CU (232 VGPRs, four LDS units).  Launched together they run the
same program in lockstep: both in an item loop (matrix cores busy, VALU idle), then both in an epilogue / LayerNorm /
camera sampling (matrix cores idle).  In the STAMPS build (make -C transcar_amd/csrc STAMPS=1) the SECOND arrival on a CU
(atomic ticket per CU from HW_ID / XCC_ID) can be started late: TRANSCAR_CHAIN_DBG bits 8..15 x 4 096 cycles.  With more
workgroups than slots (18 / 27 / 36 frames per launch: 2 / 3 / 4 rounds of 512) a later workgroup inherits the slot and
the phase of the one that ended, so the offset persists and its cost (one delay per launch) is amortised.

Measures, per frames-per-launch B and delay d: microseconds per launch of decoder layer 3 (hipGraph of N launches, best
of 4), per frame, and the gain over d = 0.  Then, for one CU, the step stamps of the pair (block 100 and its partner):
where each of the two is while the other is in its item loops.

    make -C transcar_amd/csrc STAMPS=1
    TRANSCAR_ALLOW_STAMPS=1 TRANSCAR_HIP_LIB=build/hip_stamps/libtranscar_hip_stamps.so python tools/stagger_probe.py
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                              # noqa: E402
from transcar_amd import _lib as L, ops                    # noqa: E402

N = 10
DEC = ['load attn_o', 'load x', 'out_proj', 'norm0', 'attn_w(24)', 'posenc l0', 'SAMPLE', 'pe.3', 'output_proj', 'norm1',
       'ffn0', 'ffn1', 'norm2', 'reg.0', 'next QK', 'next V', 'reg.2', 'reg.4', 'refupd']
LINEAR = {'out_proj', 'attn_w(24)', 'pe.3', 'output_proj', 'ffn0', 'ffn1', 'reg.0', 'next QK', 'next V', 'reg.2'}


def main():
    dev = torch.device('cuda:0')
    torch.set_grad_enabled(False)
    head, _ = bench.build_head(dev)
    lib = L.lib()
    for name in ('tc_debug_set_chain_dbg', 'tc_debug_stagger_reset', 'tc_debug_wg_cu', 'tc_debug_chain_stamps2'):
        assert hasattr(lib, name), 'needs the STAMPS build (see the module docstring)'
    lib.tc_debug_stagger_reset.argtypes = [C.c_void_p]
    lib.tc_debug_wg_cu.argtypes = [C.c_void_p]
    lib.tc_debug_chain_stamps2.argtypes = [C.c_int, C.c_void_p]
    lib.tc_debug_chain_stamps.argtypes = [C.c_void_p]
    lib.tc_debug_wg_spans.argtypes = [C.c_void_p]
    qe = head.query_embedding.weight
    Q, Cd, code = head.num_query, head.embed_dims, head.code_size
    qpad = ((Q + 15) // 16) * 16
    pc = L.f6(head.pc_range)

    def cs():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def make(B, seed):
        inp = bench.make_inputs(head, dev, 'res101', B, seed=seed, host_feats=False)
        o = head.forward_nhwc(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'], inp['pad_mult'], aux=True)
        M = B * Q
        st = dict(B=B, inp=inp, fv=ops.feats_view(inp['nhwc']), ref=o['aux']['inter_references'][2].contiguous(),
                  hs=o['aux']['inter_states'][2].contiguous(), attn_o=torch.randn((M, Cd), device=dev),
                  hs_out=torch.empty((M, Cd), device=dev), ref_out=torch.empty((M, 3), device=dev),
                  qk=torch.empty((M, 2 * Cd), device=dev), vt=torch.zeros((B, Cd, qpad), device=dev))
        del o
        return st

    def chain(st, rows=16):
        inp = st['inp']
        pv = head._packed_view
        lib.tc_debug_stagger_reset(cs())
        L.check(lib.tc_decoder_layer_tail_fwd(
            C.byref(pv.layers[3]), C.byref(pv.layers[4].self_attn.in_proj), C.byref(st['fv']), st['B'], Q, 6,
            code, st['attn_o'].data_ptr(), st['hs'].data_ptr(), qe.data_ptr(), inp['l2i'].data_ptr(),
            st['ref'].data_ptr(), pc, float(inp['hw'][0]), float(inp['hw'][1]), st['hs_out'].data_ptr(),
            st['ref_out'].data_ptr(), st['qk'].data_ptr(), st['vt'].data_ptr(), qpad, rows, cs()), 'tail')

    def timed(st, rows=16):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(2):
                chain(st, rows)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
                for _ in range(N):
                    chain(st, rows)
        best = None
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(s):
                g.replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / N
            best = dt if best is None else min(best, dt)
        return best * 1e6

    print('decoder chain of layer 3, us per launch (per frame) -- 16-row f16x2 tiles, two workgroups per CU, second arrival on a CU delayed')
    results = {}
    for B in (9, 18, 27, 36):
        st = make(B, 1)
        lib.tc_debug_set_chain_dbg(0)
        base32 = timed(st, 32)
        row = []
        for d in (0, 4, 8, 12, 16, 20, 24, 28, 32):
            lib.tc_debug_set_chain_dbg(d << 8)
            row.append((d, timed(st, 16)))
        results[B] = row
        t0 = row[0][1]
        print('B = %2d frames (%4d workgroups): 32-row production tiles %7.1f us (%5.2f per frame) | 16-row, delay x 4096 cycles: ' % (B, -(-B * Q // 16), base32, base32 / B)
              + '  '.join('%d: %.1f (%.2f, %+.1f %%)' % (d, t, t / B, 100.0 * (t0 / t - 1.0)) for d, t in row))
        del st
    # the pair on block 100's CU, at the best delay of the 18-frame launch
    B = 18
    best_d = min(results[B], key=lambda x: x[1])[0] or 16
    st = make(B, 1)
    for d in sorted({0, best_d, 16, 32}):
        lib.tc_debug_set_chain_dbg((d << 8) | (1 << 16))        # (bit 16: draw the per-CU tickets also at delay 0)
        lib.tc_debug_chain_stamps2(-1, None)
        chain(st)
        torch.cuda.synchronize()
        cu = np.zeros((2048, 2), dtype=np.int32)
        lib.tc_debug_wg_cu(cu.ctypes.data)
        nb = -(-B * Q // 16)
        partner = [b for b in range(min(nb, 1024)) if b != 100 and cu[b, 0] == cu[100, 0] and cu[b, 1] in (0, 1)]
        print('\ndelay %d x 4096 cycles: block 100 on CU key %#x (ticket %d); first-round partner blocks on that CU: %s' % (d, cu[100, 0], cu[100, 1], partner))
        if not partner:
            continue
        lib.tc_debug_chain_stamps2(int(partner[0]), None)
        chain(st)
        torch.cuda.synchronize()
        a = np.zeros((8, 64), dtype=np.int64)
        b2 = np.zeros((8, 64), dtype=np.int64)
        lib.tc_debug_chain_stamps(a.ctypes.data)
        lib.tc_debug_chain_stamps2(0, b2.ctypes.data)
        t0 = min(a[0, 0], b2[0, 0])
        print('  step (wave 0): block 100 [start, end) cycles since the pair\'s first stamp | partner block %d [start, end) | %s' % (partner[0], 'L = linear step'))
        for i, name in enumerate(DEC):
            print('  %-12s %s  %8d %8d   |  %8d %8d' % (name, 'L' if name in LINEAR else ' ', a[0, 2 * i] - t0, a[0, 1 + 2 * i] - t0,
                                                     b2[0, 2 * i] - t0, b2[0, 1 + 2 * i] - t0))
        def lin_intervals(x):
            return [(x[0, 2 * i] - t0, x[0, 1 + 2 * i] - t0) for i, n in enumerate(DEC) if n in LINEAR]
        ia, ib = lin_intervals(a), lin_intervals(b2)
        both = sum(max(0, min(e1, e2) - max(s1, s2)) for s1, e1 in ia for s2, e2 in ib)
        print('  cycles in linear steps: block 100 %d, partner %d; BOTH in a linear step at the same time: %d; layer spans %d / %d'
              % (sum(e - s for s, e in ia), sum(e - s for s, e in ib), both, a[0, 2 * len(DEC)] - a[0, 0], b2[0, 2 * len(DEC)] - b2[0, 0]))


if __name__ == '__main__':
    main()
