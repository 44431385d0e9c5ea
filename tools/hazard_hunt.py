"""Round 5: where do the flaky rows of the IN-PLACE f16 item loop (round 4's "variant A") come from?

tools/refill_hazard_probe.hip shows the refill itself is exact outside chain.hip (512 workgroups, partners out of
phase, every repetition verified), so the cause sits in the radar program.  This script runs the radar part
(tc_radar_fusion_fwd, 8 frames = 450 workgroups of 16 rows, two per CU) on a DIAGNOSTIC build.  The in-place loop
and its INPLACE make target left the product source in round 6: `git apply tools/experiments/r5_chain_experiments.patch`
brings them back (with the other rejected round-5 experiments), then:

  make -C transcar_amd/csrc INPLACE=1            -> build/hip_inplace/libtranscar_hip_inplace.so       (stage `final`)
  make -C transcar_amd/csrc INPLACE=1 DUMP=1     -> build/hip_inplace_dump/...inplace_dump.so          (stage `dump`)
  make -C transcar_amd/csrc DUMP=1               -> build/hip_dump/libtranscar_hip_dump.so             (production loop + dump)

  TRANSCAR_ALLOW_STAMPS=1 TRANSCAR_HIP_LIB=<lib> python tools/hazard_hunt.py final|dump [runs]

`final`: rows whose class logits / boxes differ from the f32 path's by more than 1e-3, per run and layer.
`dump` : the LDS destination of every step of the radar program is copied out ([step][row][256]); run-to-run
         differences name the FIRST step that produced a different value for a row, and which columns.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

bench._imports()
from transcar_amd import _lib as L  # noqa: E402
from transcar_amd import ops  # noqa: E402
from transcar_amd.detr3d_head import head_options  # noqa: E402

STEP_NAMES = ['gate', 'q_proj', 'ATTN', 'out_proj', 'norm2', 'linear1', 'linear2', 'norm3', 'cls.0', 'reg.0', 'cls.n1',
              'reg.2', 'cls.3', 'cls.n4', 'reg.4', 'cls.6', 'boxadd']


def main():
    stage = sys.argv[1] if len(sys.argv) > 1 else 'final'
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    B = int(os.environ.get('HUNT_B', '8'))
    dev = torch.device('cuda:0')
    head, _ = bench.build_head(dev)
    inp = bench.make_inputs(head, dev, 'tiny', B, seed=71, host_feats=False)
    o = head.forward_nhwc(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'], inp['pad_mult'], aux=True,
                          options=head_options(tile_rows=16, matrix_path='f32'))
    torch.cuda.synchronize()
    hs5 = o['aux']['inter_states'][-1].contiguous().clone()
    ref5 = o['aux']['inter_references'][-1].contiguous().clone()
    lbox = o['aux']['last_box'].contiguous().clone()
    T = inp['tokens'].shape[1]
    ws = torch.empty(L.lib().tc_head_workspace_bytes(C.byref(head._packed_view), B, T), dtype=torch.uint8, device=dev)
    M = B * 900

    def run(mp, reuse=0, compact=None):
        opt = head_options(tile_rows=16, matrix_path=mp, radar_compact=compact)
        opt.reuse_radar_kv = reuse
        c, b, h = ops.radar_fusion(head, hs5, ref5, lbox, inp['tokens'], inp['pad_mult'], 0, 3, options=opt, ws=ws)
        torch.cuda.synchronize()
        return c.clone(), b.clone(), h.clone()

    print('library:', L.LIB_PATH, ' diag build:', L.lib().tc_debug_diag_build() if hasattr(L.lib(), 'tc_debug_diag_build') else 0)
    compact = {'none': None, '0': False}[os.environ.get('HUNT_COMPACT', 'none')]
    r0 = run('f32', compact=compact)
    print('hit rows per layer (f32 path):', [(int((r0[2][l] > 0).sum())) for l in range(3)])
    if stage == 'final':
        for rep in range(runs):
            r1 = run('f16x2', reuse=1, compact=compact)
            d = (r1[0] - r0[0]).abs().amax(-1)
            db = (r1[1] - r0[1]).abs().amax(-1)
            bad = (d > 1e-3) | (db > 1e-3) | ~torch.isfinite(d) | ~torch.isfinite(db)
            per_layer = [int(bad[l].sum()) for l in range(3)]
            with_hits = int((bad & (r0[2] > 0)).sum())
            print('run %2d: bad rows per layer %s (of them with a hit in that layer: %d); max cls diff %.2e' % (
                rep, per_layer, with_hits, float(d[torch.isfinite(d)].max())))
            if compact is False:
                # own row order: workgroup = flat row // 16; blocks b and b + 256 share a CU
                blk = (torch.arange(M, device=dev) // 16).view(B, 900)
                first_bad = bad.any(0)
                lo = int((first_bad & (blk < 256)).sum()); hi = int((first_bad & (blk >= 256)).sum())
                alone = int((first_bad & (blk >= M // 16 - 256) & (blk < 256)).sum())
                print('        bad rows in blocks < 256: %d (of them in blocks without a partner: %d), in blocks >= 256: %d' % (lo, alone, hi))
        return
    # ---- dump stage
    dll = L.lib()
    dll.tc_debug_set_chain_dump.restype = C.c_int
    dll.tc_debug_set_chain_dump.argtypes = [C.c_void_p, C.c_longlong]
    nsteps = 3 * len(STEP_NAMES)
    dump = torch.zeros((nsteps, M, 256), dtype=torch.float32, device=dev)
    assert dll.tc_debug_set_chain_dump(dump.data_ptr(), dump.numel()) == 0
    snaps = []
    outs = []
    for rep in range(runs):
        dump.zero_()
        r1 = run('f16x2', reuse=1, compact=compact)
        snaps.append(dump.clone())
        outs.append(r1)
        d = (r1[0] - r0[0]).abs().amax(-1)
        db = (r1[1] - r0[1]).abs().amax(-1)
        bad = (d > 1e-3) | (db > 1e-3) | ~torch.isfinite(d) | ~torch.isfinite(db)
        print('run %2d: bad rows per layer vs f32 %s' % (rep, [int(bad[l].sum()) for l in range(3)]))
    # consensus per element = the median over the runs (errors are sparse); a run's deviations from it
    stack = torch.stack(snaps)                       # [runs, steps, M, 256]
    med = stack.median(dim=0).values
    for rep in range(runs):
        diff = (stack[rep] != med) & ~(torch.isnan(stack[rep]) & torch.isnan(med))
        rows_bad = diff.any(-1)                      # [steps, M]
        if not bool(rows_bad.any()):
            print('run %2d: every step output equals the consensus' % rep)
            continue
        first = {}
        idx = rows_bad.nonzero().tolist()
        for s, m in idx:
            if m not in first or s < first[m]:
                first[m] = s
        print('run %2d: %d rows deviate somewhere; first deviating step per row:' % (rep, len(first)))
        by_step = {}
        for m, s in first.items():
            by_step.setdefault(s, []).append(m)
        for s in sorted(by_step):
            rows = sorted(by_step[s])
            print('   step %2d (layer %d %-8s): %3d rows' % (s, s // len(STEP_NAMES), STEP_NAMES[s % len(STEP_NAMES)], len(rows)))
            for m in rows[:6]:
                cols = diff[s, m].nonzero().flatten().tolist()
                a = stack[rep, s, m, cols[0]].item()
                b = med[s, m, cols[0]].item()
                tiles = sorted(set(c // 16 for c in cols))
                print('      row %5d (sample %d query %3d, hits %s): %3d columns differ, 16-column groups %s; e.g. col %d: %.6g vs %.6g' % (
                    m, m // 900, m % 900, r0[2][:, m // 900, m % 900].tolist(), len(cols), tiles, cols[0], a, b))
                if STEP_NAMES[s % len(STEP_NAMES)] != 'q_proj' or r0[2][s // len(STEP_NAMES), m // 900, m % 900] > 0:
                    print('         columns:', cols[:48])
                    print('         got    :', ['%.4g' % stack[rep, s, m, c].item() for c in cols[:16]])
                    print('         want   :', ['%.4g' % med[s, m, c].item() for c in cols[:16]])


if __name__ == '__main__':
    main()
