#!/usr/bin/env python3
"""Benchmark of the TransCAR fusion-decoder hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over one synthetic frame:
Detr3DHead.forward (6 decoder layers + radar encoders + 3 gated radar fusion
layers) + the NMS-free box decode, with the FPN feature maps (BASELINE.json
configs[1]: 6 cameras, ResNet-101 FPN shapes, 900 queries, 255 radar points)
already resident in HBM in channels-last layout.  One process per GPU, frames
are independent (data parallel, no collective on the data path): "weak"
scaling.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import transcar_amd as T                                   # noqa: E402
from transcar_amd import _lib as L                         # noqa: E402
from transcar_amd import configs, dist as D, ops, synth    # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
F32_MFMA_PEAK_TFLOPS = 157.3   # dense f32 matrix peak (v_mfma_f32_*_f32)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--shapes', default='res101', choices=['res101', 'vovnet', 'tiny'])
    ap.add_argument('--batch', type=int, default=1, help='frames per step and GPU')
    ap.add_argument('--no-graph', action='store_true', help='eager launches (no hipGraph)')
    ap.add_argument('--lanes', type=int, default=3,
                    help='frames in flight per GPU: one hipGraph + HIP stream each '
                         '(transcar_amd/pipeline.py); 1 = strictly one frame at a time')
    ap.add_argument('--tile-rows', type=int, default=0, choices=[0, 4, 8, 16],
                    help='row-tile height of the fused chains in the frame pipeline (0 = automatic)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--unfused', action='store_true',
                    help='operator-by-operator launches instead of the fused row chains')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--no-batched', action='store_true',
                    help='skip the 4-frames-per-step side measurement')
    ap.add_argument('--train-autograd', action='store_true',
                    help='with --train: the per-operator autograd path instead of the two-call '
                         'fused forward/backward of the trainable stack')
    ap.add_argument('--train', action='store_true',
                    help='time one DDP training iteration of the fusion head (configs[2]) '
                         'instead of inference')
    return ap.parse_args()


def build_head(dev):
    sd = synth.make_state_dict(seed=3)
    head = T.build_head(configs.head_cfg())
    head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return head.to(dev).eval(), sd


def make_inputs(head, dev, shapes, batch, seed):
    """Synthetic frame(s) of BASELINE.md section 3, resident on the device."""
    feats = synth.make_feats(shapes, seed=seed, batch=batch)      # iid N(0,1)
    nhwc = [ops.to_nhwc(torch.from_numpy(f).to(dev)) for f in feats]
    l2i_np = synth.make_lidar2img()
    l2i = torch.from_numpy(np.stack([l2i_np] * batch).astype(np.float32)).to(dev)
    hw = configs.IMG_SHAPE[:2]
    # pass 1 (uniform radar) to learn where the decoder puts its boxes, then
    # 80 % of the 255 radar returns are placed near predicted centres so the
    # gated attention has realistic work
    from transcar_amd import radar as R
    f0 = [R.build_radar_features(synth.make_radar_frame(seed=2 + b)) for b in range(batch)]
    tok0, pm0 = R.pack_tokens(f0)
    o = head.forward_nhwc(nhwc, l2i, hw, torch.from_numpy(tok0).to(dev), pm0, aux=True)
    r = o['aux']['inter_references'][-1].cpu().numpy().astype(np.float64)
    pcr = configs.point_cloud_range
    fl = []
    for b in range(batch):
        c = np.round(np.stack([r[b, :, 0] * (pcr[3] - pcr[0]) + pcr[0],
                               r[b, :, 1] * (pcr[4] - pcr[1]) + pcr[1]], 1), 2)
        fl.append(R.build_radar_features(synth.make_radar_frame(seed=2 + b, centres=c)))
    tok, pm = R.pack_tokens(fl)
    return dict(feats_np=feats, nhwc=nhwc, l2i=l2i, l2i_np=l2i_np, hw=hw,
                tokens=torch.from_numpy(tok).to(dev), pad_mult=pm, radar_feats=fl)


def one_step(head, inp):
    outs = head.forward_nhwc(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'], inp['pad_mult'])
    dec = ops.box_decode_topk(outs['all_cls_scores'][-1], outs['all_bbox_preds'][-1],
                              head.bbox_coder.post_center_range, head.bbox_coder.max_num)
    return outs, dec


def time_events(fn, iters=50, warm=3):
    """Average device time of fn() in ms: `iters` launches are captured into one
    hipGraph (no host launch gaps) and the replay is bracketed by HIP events on
    the stream the kernels run on."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        for _ in range(iters):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def cur_stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def roofline(head, inp, dev):
    """Live timing of the kernels of the path.  The dominant one (largest share
    of a frame: the fused decoder row chain, 6 launches per frame) is reported
    against its roofline; the others ride along under "others"."""
    B = inp['l2i'].shape[0]
    Q, Cd, F = head.num_query, head.embed_dims, 512
    M = B * Q
    H, code, NL = 8, head.code_size, 24
    qpad = ((Q + 15) // 16) * 16
    o = head.forward_nhwc(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'], inp['pad_mult'], aux=True)
    ref = o['aux']['inter_references'][2].contiguous()
    hs2 = o['aux']['inter_states'][2].contiguous()
    fv = ops.feats_view(inp['nhwc'])
    pc = L.f6(head.pc_range)
    lib = L.lib()
    # -- fused decoder row chain (f32 MFMA): everything of a layer after the attention core
    pv = head._packed_view
    attn_o = torch.randn((M, Cd), device=dev)
    hs_out = torch.empty((M, Cd), device=dev)
    ref_out = torch.empty((M, 3), device=dev)
    qk = torch.empty((M, 2 * Cd), device=dev)
    vt = torch.zeros((B, Cd, qpad), device=dev)
    qe = head.query_embedding.weight

    def run_chain():
        L.check(lib.tc_decoder_layer_tail_fwd(
            C.byref(pv.layers[3]), C.byref(pv.layers[4].self_attn.in_proj), C.byref(fv), B, Q, 6,
            code, attn_o.data_ptr(), hs2.data_ptr(), qe.data_ptr(), inp['l2i'].data_ptr(),
            ref.data_ptr(), pc, float(inp['hw'][0]), float(inp['hw'][1]), hs_out.data_ptr(),
            ref_out.data_ptr(), qk.data_ptr(), vt.data_ptr(), qpad, cur_stream()), 'decoder_layer_tail')
    if getattr(roofline, 'chain_only', False):        # tools/chain_stamps.py: one launch, no timing
        run_chain()
        return None
    chain_ms = time_events(run_chain)
    chain_flop = 2.0 * M * (5 * Cd * Cd + Cd * NL + 2 * Cd * F + 3 * Cd * Cd + Cd * code)
    # -- camera sampling stand-alone (HBM/L2 gather): visibility-aware algorithmic bytes
    logits = torch.randn((B, Q, NL), device=dev)
    out = torch.empty((B, Q, Cd), device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)

    def run_cam(counter=None):
        L.check(lib.tc_cam_sample_fuse_fwd(
            C.byref(fv), B, Q, Cd, 6, inp['l2i'].data_ptr(), ref.data_ptr(), logits.data_ptr(),
            pc, float(inp['hw'][0]), float(inp['hw'][1]), out.data_ptr(), None,
            counter, cur_stream()), 'cam_sample')
    run_cam(C.c_void_p(cnt.data_ptr()))
    torch.cuda.synchronize()
    pairs = int(cnt.item())
    cam_ms = time_events(run_cam)
    cam_bytes = pairs * 16 * Cd * 4 + M * Cd * 4
    # -- self-attention core (f32 MFMA): 4*Q*Q*32 flop per (batch, head)
    qk_in = torch.randn((M, 2 * Cd), device=dev)
    vt_in = torch.randn((B, Cd, qpad), device=dev)
    ao = torch.empty((M, Cd), device=dev)

    def run_attn():
        L.check(lib.tc_sdpa_fwd(qk_in.data_ptr(), qk_in.data_ptr() + Cd * 4, 2 * Cd,
                                vt_in.data_ptr(), qpad, ao.data_ptr(), Cd, B, Q, H, cur_stream()), 'sdpa')
    attn_ms = time_events(run_attn)
    attn_flop = 4.0 * Q * Q * 32 * H * B
    kern = {
        'chain_kernel(decoder layer)': dict(
            bound='mfma', achieved=chain_flop / chain_ms / 1e9, peak=F32_MFMA_PEAK_TFLOPS,
            unit='TFLOP/s', ms=chain_ms, per_frame=6, alg_flop=chain_flop),
        'self_attn_kernel': dict(
            bound='mfma', achieved=attn_flop / attn_ms / 1e9, peak=F32_MFMA_PEAK_TFLOPS,
            unit='TFLOP/s', ms=attn_ms, per_frame=6, alg_flop=attn_flop),
        'cam_sample_kernel': dict(
            bound='hbm', achieved=cam_bytes / cam_ms / 1e6, peak=HBM_PEAK_GBS, unit='GB/s',
            ms=cam_ms, per_frame=0, alg_bytes=cam_bytes, visible_pairs=pairs,
            note='stand-alone operator; inside the fused forward it is a step of the chain'),
    }
    for kk in kern.values():
        kk['frac'] = kk['achieved'] / kk['peak']
    dom = max(kern, key=lambda n: kern[n]['ms'] * kern[n]['per_frame'])
    r = dict(kern[dom])
    # HBM-side traffic per launch comes from the committed PMC passes (rocprofv3 cannot
    # run inside this process); null when no profile of this kernel is committed
    traffic, src = None, None
    try:
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r1_pmc.json')))
        if dom in pmc and B == 1:
            traffic, src = pmc[dom]['traffic_bytes'], 'profiles/r1_pmc.json'
    except Exception:
        pass
    # algorithmic flop of the whole path per frame: 6 decoder chains (the last without the next
    # layer's QKV), 5 attention cores (layer 0's is a constant of the checkpoint), radar encoders
    # (T tokens) and 3 radar fusion layers
    T_tok = int(inp['tokens'].shape[1])
    nly = head.head_weights().num_layers
    path_flop = (nly * chain_flop - 2.0 * M * 3 * Cd * Cd
                 + (nly - 1) * attn_flop
                 + 2.0 * B * T_tok * (3 * Cd + Cd * Cd + 36 * 64 + 64 * 128 + 128 * Cd + 3 * Cd * 2 * Cd)
                 + 3 * 2.0 * M * (6 * Cd * Cd + 2 * Cd * F + 2 * Cd * code)) / B
    r.update(path_flop_per_frame=path_flop)
    r.update(kernel=dom, traffic=traffic, traffic_source=src,
             others={n: {kk: v[kk] for kk in ('bound', 'achieved', 'peak', 'unit', 'frac', 'ms')}
                     for n, v in kern.items() if n != dom})
    return r


def batched_side_run(head, dev, args, frames=4):
    """Not the headline: the same path with `frames` frames per step (one hipGraph
    replay per step, two steps in flight), reported beside the B = 1 value because at B = 1 a workgroup of the
    row chains is bound by its weight stream; larger row
    tiles move the same kernels toward the MFMA bound."""
    from transcar_amd.pipeline import FramePipeline
    inp = make_inputs(head, dev, args.shapes, frames, seed=11)
    nl = 1 if args.no_graph else min(2, max(1, args.lanes))
    pipe = FramePipeline(head, [inp] + [make_inputs(head, dev, args.shapes, frames, seed=13)
                                        for _ in range(nl - 1)])
    for _ in range(10):
        pipe.launch()
    torch.cuda.synchronize()
    n = max(20, args.steps // 4)
    t0 = time.perf_counter()
    for _ in range(n):
        pipe.launch()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    r = roofline(head, inp, dev)
    return {'frames_per_step': frames, 'frames_in_flight': frames * pipe.lanes,
            'value': frames * n / dt, 'unit': 'frames/s', 'ms_per_step': dt / n * 1e3,
            'roofline_frac': r['frac'], 'roofline_kernel': r['kernel'],
            'self_attn_frac': r['others']['self_attn_kernel']['frac']}


def roofline_chain_once(head, inp, dev):
    """One launch of the decoder row chain exactly as `roofline` times it."""
    roofline.chain_only = True
    try:
        roofline(head, inp, dev)
    finally:
        roofline.chain_only = False


def cpu_baseline(sd, inp, seconds):
    """The CPU oracle (oracle/transcar_oracle.py, a port of the reference's
    PyTorch path, proven equal to it on the golden fixtures) timed on this
    box's host cores on the same frame."""
    from oracle import transcar_oracle as O
    tsd = O.to_torch_sd(sd)
    feats = [torch.from_numpy(f[:1]) for f in inp['feats_np']]
    l2i = torch.from_numpy(inp['l2i_np']).float()[None]
    f36 = inp['radar_feats'][0]
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    pcr = configs.point_cloud_range
    with torch.no_grad():
        # the ops are small (900 x 256): all hardware threads of a big host
        # thrash (73 s/frame at 256 threads measured); pick the best of a
        # few thread counts on one frame each, then time with that count
        best = None
        for n in sorted({min(cores, c) for c in (8, 16, 32, 64)}):
            torch.set_num_threads(n)
            O.head_forward(tsd, feats, l2i, inp['hw'], f36, pcr)      # warm-up
            t0 = time.perf_counter()
            O.head_forward(tsd, feats, l2i, inp['hw'], f36, pcr)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, n)
            if dt > 5.0:
                break
        cores = best[1]
        torch.set_num_threads(cores)
        times = []
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end and len(times) < 50:
            t0 = time.perf_counter()
            outs = O.head_forward(tsd, feats, l2i, inp['hw'], f36, pcr)
            O.get_bboxes(outs, configs.pts_bbox_head['bbox_coder']['post_center_range'])
            times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return dict(value=1.0 / med, unit='frames/s', cores=cores, kind='port',
                ms_per_frame=med * 1e3, min_ms_per_frame=float(np.min(times)) * 1e3,
                sample='%d frames of the bench workload (B=1), torch CPU fp32, %d threads'
                       % (len(times), cores))


def train_bench(args, head, inp, dev, rank, world):
    """BASELINE.json configs[2]: batch-per-GPU 1 DDP training of the trainable
    (radar) part of the head.  A step = frozen decoder forward + radar stack
    forward (tc_radar_train_fwd) + Hungarian/focal/L1 loss (host PyTorch +
    scipy, as the reference) + HIP backward + ONE all-reduce of the flat gradient
    bucket over RCCL + device-side clip + AdamW + weight re-pack."""
    from transcar_amd.trainer import FusionTrainer
    cfg = configs.head_cfg()
    cfg['train_cfg'] = configs.train_cfg_pts
    thead = T.build_head(cfg)
    thead.load_state_dict(head.state_dict(), strict=True)
    thead = thead.to(dev)
    B = args.batch
    gts, lbs = [], []
    for b in range(B):
        boxes, labels = synth.make_gt(seed=7 + rank * 16 + b, n=24)
        gt = torch.from_numpy(boxes).clone()
        gt[:, 2] += gt[:, 5] * 0.5
        gts.append(gt.to(dev))
        lbs.append(torch.from_numpy(labels).to(dev))
    torch.set_grad_enabled(True)
    tr = FusionTrainer(thead)

    def step():
        fn = tr.step_nhwc if args.train_autograd else tr.step_fused_nhwc
        return fn(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'], inp['pad_mult'], gts, lbs)

    for _ in range(args.warmup):
        step()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = step()
    torch.cuda.synchronize()
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev if world > 1 else None)
    line = {
        'metric': 'training frames/sec: fusion head iteration (frozen DETR3D decoder fwd + radar '
                  'stack fwd/bwd + loss + grad all-reduce + AdamW), FPN features resident in HBM',
        'value': args.steps * B * world / elapsed, 'unit': 'frames/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': 'BASELINE.json configs[2]: %s FPN shapes, 900 queries, 255 radar points, '
                               '24 GT boxes, batch-per-GPU %d, DDP' % (args.shapes, B),
                   'trainable_parameters': tr.bucket.numel,
                   'grad_bucket_bytes': tr.bucket.numel * 4,
                   'parallelism': 'dp%d, one flat-bucket all-reduce per step (RCCL)' % world,
                   'final_loss': float(sum(losses.values()))},
    }
    if rank == 0:
        print(json.dumps(line))
    D.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    args = parse()
    if args.unfused:
        os.environ['TRANSCAR_UNFUSED'] = '1'
    local = int(os.environ.get('LOCAL_RANK', 0))
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
    torch.cuda.set_device(local)              # before the RCCL communicator is created
    dev = torch.device('cuda', local)
    rank, world = D.init_process_group()
    torch.set_grad_enabled(False)
    head, sd = build_head(dev)
    inp = make_inputs(head, dev, args.shapes, args.batch, seed=1 + rank)
    if args.train:
        return train_bench(args, head, inp, dev, rank, world)

    pipe = None
    if not args.no_graph:
        # a frame (13 kernel launches + decode) is captured once into a hipGraph per lane;
        # lane i works on its own synthetic frame
        from transcar_amd.pipeline import FramePipeline
        lanes = [inp] + [make_inputs(head, dev, args.shapes, args.batch, seed=101 + rank + 7 * i)
                         for i in range(1, max(1, args.lanes))]
        pipe = FramePipeline(head, lanes, tile_rows=args.tile_rows or None)

    def step():
        if pipe is not None:
            return pipe.launch()[1]
        return one_step(head, inp)

    for _ in range(args.warmup):
        step()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev if world > 1 else None)

    frames = args.steps * args.batch * world
    line = {
        'metric': 'nuScenes frames/sec (6-cam+radar, 900 queries): fusion decoder '
                  '(Detr3DHead.forward + box decode), FPN features resident in HBM',
        'value': frames / elapsed,
        'unit': 'frames/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3,
        'decoder_ms_per_frame': elapsed / args.steps * 1e3 / args.batch,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': 'BASELINE.json configs[1]: synthetic 6 cameras, ResNet-101 FPN '
                               'levels %s x 256 ch (fp32, channels-last), 900 queries, 255 radar '
                               'points, %d frame(s)/step/GPU, 1xMI355X inference'
                               % (configs.LEVEL_SHAPES[args.shapes], args.batch),
                   'shapes': args.shapes, 'frames_per_step_per_gpu': args.batch,
                   'launch': 'eager' if pipe is None else 'hipGraph replay',
                   'frames_in_flight': 1 if pipe is None else pipe.lanes,
                   'chain_tile_rows': args.tile_rows or 'auto',
                   'parallelism': 'dp%d (frames sharded, no data-path collective)' % world},
    }
    if rank == 0:
        if pipe is not None and pipe.lanes > 1:
            # one frame at a time on one lane: the latency of a frame with nothing else in flight
            pipe.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                pipe.launch(0)
            pipe.synchronize()
            dt1 = time.perf_counter() - t1
            line['single_lane'] = {'frames_in_flight': 1, 'value': args.steps * args.batch / dt1,
                                   'unit': 'frames/s', 'ms_per_frame': dt1 / args.steps * 1e3 / args.batch}
        line['roofline'] = roofline(head, inp, dev)      # rank 0's GPU; the other ranks wait at the barrier
        # every kernel of the path together, at the measured whole-job rate
        pf = line['roofline']['path_flop_per_frame']
        line['roofline']['path_achieved_tflops'] = pf * line['value'] / world / 1e12
        line['roofline']['path_frac'] = line['roofline']['path_achieved_tflops'] / line['roofline']['peak']
        if world == 1:
            if args.batch == 1 and not args.no_batched:
                line['batched'] = batched_side_run(head, dev, args, frames=4)
            if not args.no_cpu_baseline:
                line['cpu_baseline'] = cpu_baseline(sd, inp, args.cpu_seconds)
        print(json.dumps(line))
    D.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
