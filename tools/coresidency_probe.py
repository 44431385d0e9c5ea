#!/usr/bin/env python3
"""How well do the kernels of the path share the GPU when frames overlap?  S streams each replay a
hipGraph of N launches of ONE kernel (decoder row chain / attention core / radar chain), alone and
mixed; the aggregate time against the one-stream time is the stretch co-residency costs.
    python tools/coresidency_probe.py [frames per launch: 1 | 2 | 4]
Prints one line per experiment: streams, kernels, us per launch per stream, aggregate launches/ms."""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                              # noqa: E402
from transcar_amd import _lib as L, ops                    # noqa: E402
from transcar_amd.detr3d_head import head_options          # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    dev = torch.device('cuda:0')
    torch.set_grad_enabled(False)
    head, _ = bench.build_head(dev)
    inp = bench.make_inputs(head, dev, 'res101', B, seed=1)
    Q, Cd = head.num_query, head.embed_dims
    M, H, code = B * Q, 8, head.code_size
    qpad = ((Q + 15) // 16) * 16
    o = head.forward_nhwc(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'], inp['pad_mult'], aux=True)
    ref = o['aux']['inter_references'][2].contiguous()
    hs2 = o['aux']['inter_states'][2].contiguous()
    hs5 = o['aux']['inter_states'][-1].contiguous()
    ref5 = o['aux']['inter_references'][-1].contiguous()
    lbox = o['aux']['last_box'].contiguous()
    fv = ops.feats_view(inp['nhwc'])
    pc = L.f6(head.pc_range)
    lib = L.lib()
    pv = head._packed_view
    qe = head.query_embedding.weight
    T_tok = int(inp['tokens'].shape[1])
    NS = 3

    def cs():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    bufs = []
    for s in range(NS):
        b = dict(attn_o=torch.randn((M, Cd), device=dev), hs_out=torch.empty((M, Cd), device=dev),
                 ref_out=torch.empty((M, 3), device=dev), qk=torch.empty((M, 2 * Cd), device=dev),
                 vt=torch.zeros((B, Cd, qpad), device=dev), qk_in=torch.randn((M, 2 * Cd), device=dev),
                 vt_in=torch.randn((B, Cd, qpad), device=dev), ao=torch.empty((M, Cd), device=dev),
                 rws=torch.empty(lib.tc_head_workspace_bytes(C.byref(pv), B, T_tok), dtype=torch.uint8, device=dev),
                 rcls=torch.empty((3, B, Q, head.cls_out_channels), device=dev),
                 rbox=torch.empty((3, B, Q, code), device=dev), ropt=head_options())
        bufs.append(b)

    def chain(b):
        L.check(lib.tc_decoder_layer_tail_fwd(
            C.byref(pv.layers[3]), C.byref(pv.layers[4].self_attn.in_proj), C.byref(fv), B, Q, 6,
            code, b['attn_o'].data_ptr(), hs2.data_ptr(), qe.data_ptr(), inp['l2i'].data_ptr(),
            ref.data_ptr(), pc, float(inp['hw'][0]), float(inp['hw'][1]), b['hs_out'].data_ptr(),
            b['ref_out'].data_ptr(), b['qk'].data_ptr(), b['vt'].data_ptr(), qpad, 0, cs()), 'tail')

    def attn(b):
        L.check(lib.tc_sdpa_fwd(b['qk_in'].data_ptr(), b['qk_in'].data_ptr() + Cd * 4, 2 * Cd,
                                b['vt_in'].data_ptr(), qpad, b['ao'].data_ptr(), Cd, B, Q, H, cs()), 'sdpa')

    def radar(b):
        L.check(lib.tc_radar_fusion_fwd(
            C.byref(pv), hs5.data_ptr(), ref5.data_ptr(), lbox.data_ptr(), inp['tokens'].data_ptr(), B, T_tok,
            int(inp['pad_mult']), 0, 3, b['rcls'].data_ptr(), b['rbox'].data_ptr(), None, C.byref(b['ropt']),
            b['rws'].data_ptr(), b['rws'].numel(), cs()), 'radar')

    for b in bufs:
        radar(b)
        b['ropt'].reuse_radar_kv = 1
    torch.cuda.synchronize()
    fns = dict(chain=chain, attn=attn, radar=radar)
    N = 30
    streams = [torch.cuda.Stream() for _ in range(NS)]

    def capture(kind, s, n=N):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(streams[s]):
            fns[kind](bufs[s])
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=streams[s], capture_error_mode='thread_local'):
            for _ in range(n):
                fns[kind](bufs[s])
        return g

    graphs = {(k, s): capture(k, s) for k in fns for s in range(NS)}

    def run(kinds, reps=6, gs=None):
        gs = gs or [graphs[(k, s)] for s, k in enumerate(kinds)]

        def once():
            for s, g in enumerate(gs):
                with torch.cuda.stream(streams[s]):
                    g.replay()
        once()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            once()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6      # us per replay of all the streams' graphs

    solo = {k: run([k]) / N for k in fns}
    for k in fns:
        print('%d frame(s)/launch  1 stream  %-6s %7.1f us/launch' % (B, k, solo[k]))
    # every stream gets ~3 ms of its own kernel, so that the streams stay busy together
    cnt = {k: max(4, int(round(3000.0 / solo[k]))) for k in fns}
    bal = {(k, s): capture(k, s, cnt[k]) for k in fns for s in range(NS)}
    for kinds in (['chain'] * 2, ['chain'] * 3, ['attn'] * 2, ['attn'] * 3, ['radar'] * 2, ['radar'] * 3,
                  ['chain', 'attn'], ['chain', 'radar'], ['chain', 'chain', 'attn'], ['chain', 'attn', 'radar']):
        t = run(kinds, gs=[bal[(k, s)] for s, k in enumerate(kinds)])
        serial = sum(solo[k] * cnt[k] for k in kinds)
        print('%d frame(s)/launch  %d streams %-20s %8.1f us together, %8.1f one after the other: overlap gain %.2fx'
              % (B, len(kinds), '+'.join(kinds), t, serial, serial / t))


if __name__ == '__main__':
    main()
