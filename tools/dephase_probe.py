#!/usr/bin/env python3
"""What would workgroups in DIFFERENT phases be worth for the decoder row chain?  (at 32-row tiles the
kernel is a sequence of phases that each load one resource -- matrix pipe, issue slots, memory system -- with every
workgroup of the chip in the same phase.)

The 32-row decoder chain of layer 3 is replayed (hipGraph, N launches per stream)
  (a) on ONE stream with 9 frames per launch: 254 workgroups, the whole chip in lockstep (the production geometry);
  (b) on TWO streams with 4 frames per launch each: 2 x 113 workgroups on disjoint CUs, the second stream started half
      a kernel late -- the two halves of the chip are then in different phases for the whole run;
  (c) as (b) with both streams started together (the halves in lockstep again: the control);
  (d) on THREE streams with 3 frames each (3 x 85 workgroups), staggered by a third.
Prints frames per millisecond of each: (b) / (c) is what de-phasing itself buys, (b) / (a) what a launch geometry of two
half-chip kernels would (it also pays a second weight stream per XCD).

    python tools/dephase_probe.py            (on the MI355X box; ~15 s)"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                              # noqa: E402
from transcar_amd import _lib as L, ops                    # noqa: E402

N = 40


def main():
    dev = torch.device('cuda:0')
    torch.set_grad_enabled(False)
    head, _ = bench.build_head(dev)
    lib = L.lib()
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
        return st

    def chain(st):
        inp = st['inp']
        pv = head._packed_view                      # (exists after the first forward)
        L.check(lib.tc_decoder_layer_tail_fwd(
            C.byref(pv.layers[3]), C.byref(pv.layers[4].self_attn.in_proj), C.byref(st['fv']), st['B'], Q, 6,
            code, st['attn_o'].data_ptr(), st['hs'].data_ptr(), qe.data_ptr(), inp['l2i'].data_ptr(),
            st['ref'].data_ptr(), pc, float(inp['hw'][0]), float(inp['hw'][1]), st['hs_out'].data_ptr(),
            st['ref_out'].data_ptr(), st['qk'].data_ptr(), st['vt'].data_ptr(), qpad, 32, cs()), 'tail')   # 32-row tiles, automatic matrix path

    def graph_of(st, stream):
        with torch.cuda.stream(stream):
            for _ in range(3):
                chain(st)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            with torch.cuda.graph(g, stream=stream, capture_error_mode='thread_local'):
                for _ in range(N):
                    chain(st)
        return g

    def run(states, delays_us, label):
        streams = [torch.cuda.Stream() for _ in states]
        graphs = [graph_of(st, s) for st, s in zip(states, streams)]
        frames = sum(st['B'] for st in states)
        best = None
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for g, s, d in zip(graphs, streams, delays_us):
                if d:
                    t1 = time.perf_counter()
                    while (time.perf_counter() - t1) * 1e6 < d:
                        pass
                with torch.cuda.stream(s):
                    g.replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print('%-58s %6.1f us per round of launches, %6.2f frames per ms' % (label, best / N * 1e6, frames * N / best * 1e-3))
        return frames * N / best

    a = run([make(9, 1)], [0], '(a) one stream, 9 frames per launch (254 workgroups)')
    two = [make(4, 2), make(4, 3)]
    c = run(two, [0, 0], '(c) two streams x 4 frames, started together')
    b = run(two, [0, 45], '(b) two streams x 4 frames, the second 45 us late')
    three = [make(3, 4), make(3, 5), make(3, 6)]
    d = run(three, [0, 30, 30], '(d) three streams x 3 frames, 30 us apart')
    print('de-phasing itself (b) / (c): %.3f;  two half-chip kernels against the production launch (b) / (a): %.3f;  '
          'three thirds (d) / (a): %.3f' % (b / c, b / a, d / a))


if __name__ == '__main__':
    main()
