#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (written by tools/profile_round.sh on the MI355X box)
into the committed summaries under profiles/:

    python tools/collect_profile.py r1f r1

  profiles/<name>_bench.json         bench.py lines (B=1, and B=2/4 side runs)
  profiles/<name>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary
  profiles/<name>_frame_trace.txt    one replayed frame, kernel by kernel
  profiles/<name>_pmc.json           FETCH_SIZE / WRITE_SIZE per launch
                                     (separate --pmc passes) -> HBM-side bytes
"""
import csv
import glob
import re
import json
import os
import sys
from collections import OrderedDict, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(pattern):
    # the NEWEST match: gpurun merges a call's files into gpurun_out/ next to those of earlier calls (pid-named)
    g = sorted(glob.glob(pattern), key=os.path.getmtime)
    if not g:
        raise SystemExit('missing ' + pattern)
    return g[-1]


def newest(pattern):
    """matches, newest first"""
    return sorted(glob.glob(pattern), key=os.path.getmtime, reverse=True)


def short(name):
    n = name.replace('(anonymous namespace)::', '').split('(')[0]
    n = n.replace('void ', '').replace('tc::', '')
    return n.strip()


def label_chain(rows, num_layers=6):
    """chain_kernel launches are told apart by their neighbours in dispatch
    order: the small grid is the radar encoder (side stream); a full-grid
    launch right after self_attn is a decoder layer, one right before the first
    self_attn of a frame is decoder layer 0, the one after the last decoder layer
    of a frame is the radar chain; back-to-back launches without attention in
    between are bench.py's roofline replay of the decoder layer."""
    for r in rows:
        r['K'] = short(r['Kernel_Name'])
        if r['K'].startswith('chain_dual_kernel'):
            r['K'] = 'chain_dual_kernel(decoder layer + radar encoder half)'
    main_rows = []
    for r in rows:
        if r['K'].startswith('chain_kernel') and int(r['Grid_Size']) < 256 * 128:
            r['K'] = 'chain_kernel(radar encoders)'
        elif r['K'].startswith('chain_') or r['K'].startswith('self_attn') or 'box_decode' in r['K']:
            main_rows.append(r)
    n_dec = 0
    for i, r in enumerate(main_rows):
        if r['K'].startswith('chain_dual_kernel'):      # decoder layers 0 and 1 carry the radar encoders
            prev = main_rows[i - 1]['K'] if i else ''
            n_dec = n_dec + 1 if prev.startswith('self_attn') else 1
            continue
        if not r['K'].startswith('chain_kernel'):
            if 'box_decode' in r['K']:
                n_dec = 0
            continue
        prev = main_rows[i - 1]['K'] if i else ''
        nxt = main_rows[i + 1]['K'] if i + 1 < len(main_rows) else ''
        if prev.startswith('self_attn'):
            r['K'] = 'chain_kernel(decoder layer)'
            n_dec += 1
        elif nxt.startswith('self_attn'):
            # first launch of a frame: decoder layer 0 (its self-attention is folded into the
            # packed weights; builds before that had a prologue chain here)
            r['K'] = 'chain_kernel(decoder layer)'
            n_dec = 1
        elif n_dec == num_layers and 'box_decode' in nxt:
            r['K'] = 'chain_kernel(radar)'
        else:
            r['K'] = 'chain_kernel(decoder layer)'      # roofline replay


def prog_label(kernel_name):
    """Label from the template arguments (chain_kernel<rows, program>): works in any launch order."""
    import re
    # round 4: a trailing template argument 1 = the two-plane f16 matrix path of the 16-row tiles (chain.hip MM)
    # round 6: a fifth / fifth template argument `true` = the PRE instantiation (the sampling step reads pre-gathered values)
    m = re.search(r'chain_dual_kernel<(\d+), (\d+), (\d+)(?:, (\d+))?(?:, (true|false))?>', kernel_name)
    if m:
        return 'chain_dual_kernel(decoder layer + radar encoder half %s, %s-row tiles%s%s)' % (
            'A' if m.group(3) == '4' else 'B', m.group(1), ', f16x2' if m.group(4) == '1' else '', ', pre-gathered' if m.group(5) == 'true' else '')
    m = re.search(r'chain_kernel<(\d+), (\d+)(?:, (\w+))?(?:, (\d+))?(?:, (true|false))?>', kernel_name)
    if m:
        return 'chain_kernel(%s, %s-row tiles%s%s)' % (
            {'0': 'prologue', '1': 'decoder layer', '2': 'radar encoders', '3': 'radar fusion', '6': 'radar fusion (train)',
             '7': 'radar encoders (train)', '8': 'radar backward'}.get(m.group(2), 'program ' + m.group(2)),
            m.group(1), ', f16x2' if m.group(4) == '1' else '', ', pre-gathered' if m.group(5) == 'true' else '')
    if 'self_attn_kernel' in kernel_name or 'self_attn_x_kernel' in kernel_name:     # (x: the staged f16x2 form, round 4)
        return 'self_attn_kernel'
    return short(kernel_name)


def main():
    tag, name = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, 'gpurun_out', tag)
    dst = os.path.join(ROOT, 'profiles')
    os.makedirs(dst, exist_ok=True)

    lines = []
    for extra, target in (('bench_driver.json', '_driver_cmd.json'), ('bench_vovnet.json', '_vovnet_bench.json'),
                          ('bench_f32.json', '_f32_path_bench.json')):
        p = os.path.join(src, extra)
        if os.path.exists(p):
            el = [ln.strip() for ln in open(p) if ln.strip().startswith('{')]
            if el:
                open(os.path.join(dst, name + target), 'w').write('\n'.join(el) + '\n')
    fs = newest(os.path.join(src, 'prof_f32', '*', '*kernel_stats.csv'))
    if fs:
        open(os.path.join(dst, name + '_f32_path_kernel_stats.csv'), 'w').write(open(fs[0]).read())
    for f in ('bench.json', 'bench_pair1.json', 'bench_pair2.json', 'bench_pair4.json', 'bench_steps20.json'):
        p = os.path.join(src, f)
        if os.path.exists(p):
            for ln in open(p):
                ln = ln.strip()
                if ln.startswith('{'):
                    lines.append(ln)
    open(os.path.join(dst, name + '_bench.json'), 'w').write('\n'.join(lines) + '\n')

    tb = os.path.join(src, 'bench_train.json')
    if os.path.exists(tb):
        tl = [ln.strip() for ln in open(tb) if ln.strip().startswith('{')]
        open(os.path.join(dst, name + '_train_bench.json'), 'w').write('\n'.join(tl) + '\n')
    ts = newest(os.path.join(src, 'prof_train', '*', '*kernel_stats.csv'))
    if ts:
        open(os.path.join(dst, name + '_train_kernel_stats.csv'), 'w').write(open(ts[0]).read())

    stats = one(os.path.join(src, 'prof', '*', '*kernel_stats.csv'))
    open(os.path.join(dst, name + '_kernel_stats.csv'), 'w').write(open(stats).read())

    # one frame of the trace: the last full graph replay
    trace = one(os.path.join(src, 'prof', '*', '*kernel_trace.csv'))
    rows = list(csv.DictReader(open(trace)))
    for r in rows:       # the trace csv splits the grid per axis, the counter csv does not
        if 'Grid_Size' not in r:
            r['Grid_Size'] = str(int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']))
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    for r in rows:
        r['K'] = prog_label(r['Kernel_Name'])
    dec = [i for i, r in enumerate(rows) if 'box_decode' in r['Kernel_Name']]
    out = ['# one replay of `bench.py --lanes 1` (hipGraph; 9 frames per launch sequence, the default) from rocprofv3 --kernel-trace on MI355X; us',
           '# %-44s %10s %10s %10s' % ('kernel', 'grid', 'start', 'dur')]
    if len(dec) >= 2:
        a, b = dec[-2] + 1, dec[-1] + 1
        t0 = int(rows[a]['Start_Timestamp'])
        for r in rows[a:b]:
            s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            out.append('%-46s %10s %10.1f %10.1f' % (r['K'][:46], r['Grid_Size'],
                                                     (s - t0) / 1e3, (e - s) / 1e3))
        out.append('# frame span %.1f us' % ((int(rows[b - 1]['End_Timestamp']) - t0) / 1e3))
    # average duration per (kernel, grid)
    agg = defaultdict(list)
    for r in rows:
        agg[(r['K'], r['Grid_Size'])].append(
            (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    out.append('')
    out.append('# average duration per (kernel, grid size) over the whole run')
    for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        out.append('%-46s %10s  n=%-6d avg %8.1f us  total %10.1f us' % (k[:46], g, len(v), sum(v) / len(v), sum(v)))
    open(os.path.join(dst, name + '_frame_trace.txt'), 'w').write('\n'.join(out) + '\n')

    # the default command: several frames in flight (transcar_amd/pipeline.py)
    lt = newest(os.path.join(src, 'prof_lanes', '*', '*kernel_trace.csv'))
    ls = newest(os.path.join(src, 'prof_lanes', '*', '*kernel_stats.csv'))
    if lt and ls:
        open(os.path.join(dst, name + '_lanes_kernel_stats.csv'), 'w').write(open(ls[0]).read())
        lrows = list(csv.DictReader(open(lt[0])))
        for r in lrows:
            if 'Grid_Size' not in r:
                r['Grid_Size'] = str(int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']))
        for r in lrows:
            r['K'] = prog_label(r['Kernel_Name'])
        lrows.sort(key=lambda r: int(r['Start_Timestamp']))
        dec = [i for i, r in enumerate(lrows) if 'box_decode' in r['Kernel_Name']]
        lo = ['# `bench.py` (default: 3 lanes in flight, one hipGraph + HIP stream each, 9 frames per launch sequence) from rocprofv3',
              '# --kernel-trace on MI355X: the kernels of ~3 consecutive launch sequences in start order; us',
              '# %-44s %8s %10s %10s' % ('kernel', 'stream', 'start', 'dur')]
        if len(dec) >= 40:
            a, b = dec[30] + 1, dec[33] + 1          # inside the timed region (the tail of the run is single-lane)
            t0 = int(lrows[a]['Start_Timestamp'])
            for r in lrows[a:b]:
                st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
                lo.append('%-46s %8s %10.1f %10.1f' % (r['K'][:46], r.get('Stream_Id', r.get('Queue_Id', '?')), (st - t0) / 1e3, (en - st) / 1e3))
            span = (max(int(r['End_Timestamp']) for r in lrows[a:b]) - t0) / 1e3
            lo.append('# %d box decodes (= launch sequences of frames_per_launch frames) completed in this window of %.1f us' % (3, span))
        agg2 = defaultdict(list)
        for r in lrows:
            agg2[(r['K'], r['Grid_Size'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        lo.append('')
        lo.append('# average duration per (kernel, grid size) with frames in flight (kernels of different frames share the CUs)')
        for (k, g), v in sorted(agg2.items(), key=lambda kv: -sum(kv[1]))[:8]:
            lo.append('%-46s %10s  n=%-6d avg %8.1f us' % (k[:46], g, len(v), sum(v) / len(v)))
        open(os.path.join(dst, name + '_lanes_trace.txt'), 'w').write('\n'.join(lo) + '\n')

    # PMC passes (one counter per run, kernel-trace only; `bench.py --no-graph --no-roofline --batch B`:
    # B = 2 is the launch shape of the default bench command, 2 frames per launch), per launch and
    # kernel; kernels are told apart by their template arguments (prog_label)
    res = OrderedDict()
    res['_comment'] = ('rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of '
                       '`bench.py --no-graph --no-roofline --batch B --steps 5`, MI355X; second key = B, the frames '
                       'per launch.  KB per launch as reported; traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 '
                       '(gfx950: FETCH_SIZE tallies 64 B per 128-B request, MI355X_MICROARCH.md, HBM section).')
    out_m = OrderedDict()
    out_m['_comment'] = ('rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES of the same command; second key = '
                         'frames per launch.  The counter sums matrix-pipe busy cycles over the 1024 SIMDs (8 per '
                         'v_mfma_f32_4x4x1, 32 per 16x16x4 f32); utilisation = busy / (1024 * duration * 2.4 GHz).')
    for B in ('1', '2', '4', '8', '9', '10'):
        per = defaultdict(dict)
        for ctr in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_VALU_MFMA_BUSY_CYCLES'):
            fs = newest(os.path.join(src, 'pmc_%s_b%s' % (ctr, B), '*', '*counter_collection.csv'))
            if not fs:
                continue
            acc = defaultdict(lambda: [0.0, 0.0, 0])
            for r in csv.DictReader(open(fs[0])):
                if r['Counter_Name'] != ctr:
                    continue
                a = acc[re.sub(r', \d+-row tiles(, f16x2)?', '', prog_label(r['Kernel_Name']))]
                a[0] += float(r['Counter_Value'])
                a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
                a[2] += 1
            for k, (v, dur, n) in acc.items():
                per[k][ctr] = (v / n, dur / n, n)
        for k, v in sorted(per.items()):
            if not (k.startswith('chain') or k.startswith('self_attn') or 'box_decode' in k or 'nchw' in k):
                continue
            if 'FETCH_SIZE' in v or 'WRITE_SIZE' in v:
                fk, wk = v.get('FETCH_SIZE', (0.0,))[0], v.get('WRITE_SIZE', (0.0,))[0]
                res.setdefault(k, OrderedDict())[B] = {
                    'fetch_kb': round(fk, 1), 'write_kb': round(wk, 1), 'traffic_bytes': int((2 * fk + wk) * 1024),
                    'launches': v.get('FETCH_SIZE', v.get('WRITE_SIZE'))[2]}
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in v:
                busy, dur, n = v['SQ_VALU_MFMA_BUSY_CYCLES']
                out_m.setdefault(k, OrderedDict())[B] = {
                    'mfma_busy_cycles': round(busy), 'duration_us': round(dur, 1),
                    'mfma_utilisation': round(busy / (1024 * dur * 1e-6 * 2.4e9), 4), 'launches': n}
    json.dump(res, open(os.path.join(dst, name + '_pmc.json'), 'w'), indent=1)
    json.dump(out_m, open(os.path.join(dst, name + '_mfma_busy.json'), 'w'), indent=1)
    # one launch sequence of one frame at a time (--pair 1 --lanes 1): the latency anatomy
    p1 = newest(os.path.join(src, 'prof_pair1', '*', '*kernel_stats.csv'))
    if p1:
        open(os.path.join(dst, name + '_pair1_kernel_stats.csv'), 'w').write(open(p1[0]).read())
    # round 6: the opt-in camera pre-gather (bench line + the kernel stats of one launch sequence at a time), the plugin
    # entry's breakdown, the pair-exchange probe
    pg = newest(os.path.join(src, 'prof_pregather', '*', '*kernel_stats.csv'))
    if pg:
        open(os.path.join(dst, name + '_pregather_kernel_stats.csv'), 'w').write(open(pg[0]).read())
    for f, target in (('bench_pregather.json', '_pregather_bench.json'), ('dropin_breakdown.txt', '_dropin_breakdown.txt'),
                      ('pair_exchange_probe.txt', '_pair_exchange_probe.txt'), ('bench_train_det.json', '_train_det_bench.json')):
        pth = os.path.join(src, f)
        if os.path.exists(pth):
            txt = open(pth).read()
            if f.endswith('.json'):
                txt = '\n'.join(ln.strip() for ln in txt.splitlines() if ln.strip().startswith('{')) + '\n'
            else:
                txt = '\n'.join(ln for ln in txt.splitlines() if 'amdgpu.ids' not in ln) + '\n'
            open(os.path.join(dst, name + target), 'w').write(txt)
    print(open(os.path.join(dst, name + '_frame_trace.txt')).read())
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
