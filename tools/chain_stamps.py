#!/usr/bin/env python3
"""Per-step cycle counts of the fused row chains (debug build with s_memtime
stamps):   make -C transcar_amd/csrc STAMPS=1
           TRANSCAR_ALLOW_STAMPS=1 TRANSCAR_HIP_LIB=build/hip_stamps/libtranscar_hip_stamps.so python tools/chain_stamps.py [decoder|radar]
STAMPS_COLD=1 (round 6, decoder only): the stamped launch is decoder layer 3 IN A SEQUENCE (a forward that stops behind it,
other data streamed in between) instead of the layer repeated on top of a forward: profiles/r6_chain_stamps_cold.txt.
Prints, for workgroup 100 and each of its 4 waves, the cycles (100 MHz s_memtime
ticks x 24 = 2.4 GHz core cycles) each step took and the wait at its barrier."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                              # noqa: E402
from transcar_amd import _lib as L                         # noqa: E402

DEC = ['load attn_o', 'load x', 'out_proj', 'norm0 (+x+pos)', 'attn_w(24)', 'posenc l0', 'SAMPLE', 'pe.3',
       'output_proj', 'norm1', 'ffn0', 'ffn1', 'norm2 (+x+pos)', 'reg.0', 'next QK', 'next V', 'reg.2', 'reg.4 (dot)', 'refupd']
RAD = ['radar gate', 'q proj', 'RADAR ATTN', 'out_proj', 'norm2', 'linear1', 'linear2', 'norm3', 'cls.0', 'reg.0', 'cls LN1',
       'reg.2', 'cls.3', 'cls LN4', 'reg.4 (dot)', 'cls.6 (dot)', 'boxadd']


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'decoder'
    dev = torch.device('cuda:0')
    torch.set_grad_enabled(False)
    head, _ = bench.build_head(dev)
    batch = int(os.environ.get('STAMPS_BATCH', '1'))        # 4: the 16-row tiles
    inp = bench.make_inputs(head, dev, os.environ.get('STAMPS_SHAPES', 'res101'), batch, seed=1)   # 'tiny': every tap an L2 hit
    lib = L.lib()
    lib.tc_debug_chain_stamps.restype = C.c_int
    lib.tc_debug_chain_stamps.argtypes = [C.c_void_p]
    sub = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    lib.tc_debug_chain_sub.restype = C.c_int
    lib.tc_debug_chain_sub.argtypes = [C.c_int, C.c_void_p]
    lib.tc_debug_chain_sub(sub, None)
    block2 = int(os.environ.get('STAMPS_BLOCK2', '-1'))      # a second stamped workgroup (radar, 9 frames: block 0 is a tile of hit rows)
    lib.tc_debug_chain_stamps2.argtypes = [C.c_int, C.c_void_p]
    lib.tc_debug_chain_stamps2(block2, None)
    rows_env = int(os.environ.get('STAMPS_ROWS', '0'))       # 0: automatic; 32: the 8-wave tiles
    from transcar_amd.detr3d_head import head_options
    NWV = 8 if rows_env == 32 else 4
    bench.roofline.tile_rows = rows_env
    for _ in range(3):
        head.forward_nhwc(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'], inp['pad_mult'], options=head_options(tile_rows=rows_env or None))
    if which == 'decoder' and os.environ.get('STAMPS_COLD') == '1':
        # IN SEQUENCE (round 6): a forward that stops after decoder layer 3 (tc_head_options.phase = 1) -- the stamped
        # launch is that layer as a frame runs it, its weights last touched a sequence ago; in between, other data of
        # the size of a whole sequence's taps is streamed through the caches
        junk = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)
        for _ in range(2):
            junk.add_(1.0)
            head.forward_nhwc(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'], inp['pad_mult'],
                              options=head_options(tile_rows=rows_env or None, phase=1))
        names = DEC
    elif which == 'decoder':
        # the forward's LAST chain launch is the radar chain; run one decoder tail on top
        bench.roofline_chain_once(head, inp, dev)
        names = DEC
    else:
        names = RAD * 3
    torch.cuda.synchronize()
    buf = np.zeros((8, 64), dtype=np.int64)
    assert lib.tc_debug_chain_stamps(buf.ctypes.data) == 0
    t = buf - buf[:, :1]                                   # s_memtime ticks (= core cycles here)
    if sub >= 0:
        sb = np.zeros((8, 64), dtype=np.int64)
        assert lib.tc_debug_chain_sub(0, sb.ctypes.data) == 0
        print('sub-step stamps of table step %d (%s): 0 dispatch, 1 spec built, 2 first load issued, 3.. after item i' % (sub, names[sub]))
        st = sb[0, 40:47]
        print('kernel start (cycles since entry): early loads issued %d, resolve done %d, early stores %d, barrier %d, prefetch records %d, ready %d'
              % tuple(int(x - st[0]) for x in st[1:]))
        for w in range(NWV):
            row = sb[w]
            n = int((row[:20] > 0).sum())
            print('  wave%d:' % w, ' '.join('%d:%d' % (j, row[j] - row[0]) for j in range(20) if row[j] > 0))
            if NWV == 8:
                continue
            print('     item0 [start, A read, setup, mfma+prefetch]:', ' '.join('%6d' % (row[j] - row[0]) for j in range(20, 24)),
                  ' item1:', ' '.join('%6d' % (row[j] - row[0]) for j in range(25, 29)))
    if which == 'radar':
        # the radar chain is the forward's last chain launch: its workgroups' spans, by tile position inside a sample
        # (beyond one frame per launch the rows of a sample are ordered hits first: the gated part -- q projection,
        # attention over the hit tokens, out_proj -- runs in the first tiles only)
        wg = np.zeros((1024, 2), dtype=np.int64)
        lib.tc_debug_wg_spans.restype = C.c_int
        lib.tc_debug_wg_spans.argtypes = [C.c_void_p]
        assert lib.tc_debug_wg_spans(wg.ctypes.data) == 0
        rows = batch * head.num_query
        R = rows_env or (4 if rows <= 1024 else 8 if rows <= 2048 else 16)
        nb = min(-(-rows // R), 1024)
        w = wg[:nb]
        t0 = w[:, 0].min()
        start, span = w[:, 0] - t0, w[:, 1] - w[:, 0]
        print('radar chain: %d workgroups of %d rows; span min/median/max %d/%d/%d'
              % (nb, R, span.min(), np.median(span), span.max()))
        print('  span deciles:', ' '.join('%d' % x for x in np.percentile(span, [0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100])))
        hist, edges = np.histogram(span, bins=12)
        print('  span histogram (cycles: workgroups):', ', '.join('%d-%d: %d' % (edges[i], edges[i + 1], hist[i]) for i in range(len(hist))))
        tiles_per_sample = -(-head.num_query // R)
        pos = (np.arange(nb) * R % head.num_query) // R if rows % head.num_query == 0 else np.arange(nb) % tiles_per_sample
        first = np.array([np.median(span[(np.arange(nb) * R // head.num_query == b) & ((np.arange(nb) * R % head.num_query) < 4 * R)])
                          for b in range(batch)])
        rest = np.array([np.median(span[(np.arange(nb) * R // head.num_query == b) & ((np.arange(nb) * R % head.num_query) >= 8 * R)])
                         for b in range(batch)])
        print('  median span of a sample\'s first 4 tiles (hit rows) %d, of its tiles from the 8th on (no hits) %d'
              % (np.median(first), np.median(rest)))
    if which == 'decoder':
        wg = np.zeros((1024, 2), dtype=np.int64)
        lib.tc_debug_wg_spans.restype = C.c_int
        lib.tc_debug_wg_spans.argtypes = [C.c_void_p]
        assert lib.tc_debug_wg_spans(wg.ctypes.data) == 0
        nb = min(int((wg[:, 1] > 0).sum()), 1024)      # stale entries beyond: earlier dual launches
        w = wg[:nb]
        t0 = w[:, 0].min()
        start, end, span = w[:, 0] - t0, w[:, 1] - t0, w[:, 1] - w[:, 0]
        print('workgroups: %d; entry skew min/median/max %d/%d/%d; span min/median/max %d/%d/%d; last exit %d'
              % (nb, start.min(), np.median(start), start.max(), span.min(), np.median(span), span.max(), end.max()))
        print('  span deciles:', ' '.join('%d' % x for x in np.percentile(span, [0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100])))
        print('  entry deciles:', ' '.join('%d' % x for x in np.percentile(start, [0, 10, 50, 90, 100])))
        print('  block 100: span %d; step stamps cover %d..%d after its entry' % (span[100], buf[0, 0] - w[100, 0], buf[0, :].max() - w[100, 0]))
        order = np.argsort(-end)[:8]
        print('  last to finish (block: entry, span):', ', '.join('%d: %d, %d' % (b, start[b], span[b]) for b in order))
        cb = np.zeros((4, 8), dtype=np.int64)
        lib.tc_debug_cam_stamps.restype = C.c_int
        lib.tc_debug_cam_stamps.argtypes = [C.c_void_p]
        assert lib.tc_debug_cam_stamps(cb.ctypes.data) == 0
        print('camera sampling (cycles relative to the entry of cam_sample_row): step entry, projected + ballot, '
              'taps issued (last visible camera), accumulated, row done, rows done, pair counter added')
        for w in range(4):
            print('  wave%d:' % w, ' '.join('%6d' % (cb[w, j] - cb[w, 5]) for j in (5, 0, 1, 2, 3, 4, 6, 7)), ' (relative to the step entry; row stamps: the wave\'s LAST row)')
    print('%-14s %s' % ('step', '   '.join('wave%d work / wait' % w for w in range(NWV))))
    n = min(len(names), 31)
    for i in range(n):
        cols = []
        for w in range(NWV):
            work = t[w, 1 + 2 * i] - t[w, 2 * i]
            wait = t[w, 2 + 2 * i] - t[w, 1 + 2 * i]
            cols.append('%6d /%5d' % (work, wait))
        print('%-14s %s' % (names[i], '   '.join(cols)))
    print('total cycles (wave 0): %d = %.1f us at 2.4 GHz' % (t[0, 2 * n], t[0, 2 * n] / 2400.0))
    if block2 >= 0:
        b2 = np.zeros((8, 64), dtype=np.int64)
        assert lib.tc_debug_chain_stamps2(0, b2.ctypes.data) == 0
        t2 = b2 - b2[:, :1]
        print('\nworkgroup %d:' % block2)
        for i in range(n):
            print('%-14s %s' % (names[i], '   '.join('%6d /%5d' % (t2[w, 1 + 2 * i] - t2[w, 2 * i], t2[w, 2 + 2 * i] - t2[w, 1 + 2 * i]) for w in range(NWV))))
        print('total cycles (wave 0): %d' % t2[0, 2 * n])


if __name__ == '__main__':
    main()
