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
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# numpy / torch / transcar_amd are imported by `_imports()` in the worker processes only:
# the launcher parent of `--gpus N` (N > 1) must never initialise the GPU (it starts N fresh
# children and waits; a process that has touched the GPU is never re-exec'ed)
np = torch = T = L = configs = D = ops = synth = None


def _imports():
    global np, torch, T, L, configs, D, ops, synth
    import numpy as np_
    import torch as torch_
    import transcar_amd as T_
    from transcar_amd import _lib as L_
    from transcar_amd import configs as configs_, dist as D_, ops as ops_, synth as synth_
    np, torch, T, L, configs, D, ops, synth = np_, torch_, T_, L_, configs_, D_, ops_, synth_


if __name__ != '__main__':          # used as a library (tests, tools/): a worker by definition
    _imports()


HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
F32_MFMA_PEAK_TFLOPS = 157.3   # dense f32 matrix peak (v_mfma_f32_*_f32)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1,
                    help='GPUs of this node, one process each.  Without a torchrun environment '
                         '(WORLD_SIZE unset) and N > 1 this process only spawns the N ranks')
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--shapes', default='res101', choices=['res101', 'vovnet', 'tiny'])
    ap.add_argument('--batch', type=int, default=1, help='frames per step and GPU')
    ap.add_argument('--no-graph', action='store_true', help='eager launches (no hipGraph)')
    ap.add_argument('--lanes', type=int, default=3,
                    help='frames in flight per GPU: one hipGraph + HIP stream each '
                         '(transcar_amd/pipeline.py); 1 = strictly one frame at a time')
    ap.add_argument('--pair', type=int, default=0,
                    help='frames per launch: the pipeline hands the head P consecutive frames (one per '
                         'step, the per-frame API is unchanged) as ONE launch sequence -- P >= 3: 16-row tiles '
                         'on the 16x16x4 MFMA, every streamed weight fragment feeds 16 rows; 2: 8-row tiles; '
                         '1 = one frame per launch.  0 (default) = the most frames whose 16-row tiles are resident '
                         'at once, two workgroups per CU (900 queries: 9 frames = 507 workgroups); a timed window '
                         'of K steps ends with one partial launch of K %% P frames.  Other settings are measured '
                         'beside the headline (`frames_per_launch_sweep`), the like-for-like latency '
                         '`latency_ms_per_frame` always at 1')
    ap.add_argument('--tile-rows', type=int, default=0, choices=[0, 4, 8, 16, 32],
                    help='row-tile height of the fused chains in the frame pipeline (0 = automatic)')
    ap.add_argument('--matrix-path', default='auto', choices=['auto', 'f32', 'f16x2'],
                    help='tc_head_options.matrix_path of the 16-row tiles: f16x2 (= auto) two-plane f16 operands on the '
                         'matrix cores with fp32 accumulation, f32 the exact fp32 MFMA chains')
    ap.add_argument('--no-configs', action='store_true',
                    help='skip the side measurements of BASELINE.json configs[4] (VoVNet shapes) and configs[2] (a training '
                         'iteration) that ride in the default line')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--last-cls-only', action='store_true',
                    help='inference opt-in tc_head_options.last_level_cls_only: final_cls / final_cls2 are not '
                         'evaluated (get_bboxes decodes level 3 only); NOT the reference output contract')
    ap.add_argument('--no-radar-compact', action='store_true',
                    help='tc_head_options.radar_row_order = 1: the radar chain keeps the queries in their own order '
                         '(default: beyond one frame per launch, queries with a radar hit go first)')
    ap.add_argument('--pregather', action='store_true',
                    help='tc_head_options.cam_pregather = 1 (round 6, opt-in): extra workgroups of the attention-core launch in '
                         'front of a decoder chain gather its camera taps (bit-identical outputs; default: the chain gathers '
                         'itself -- with three launch sequences in flight the pre-gather costs 2.8 %%)')
    ap.add_argument('--main-only', action='store_true',
                    help='only the main timed loop: no single-lane, roofline, delivery, hand-off, batched or CPU '
                         'side measurements (kernel traces of tools/profile_round.sh)')
    ap.add_argument('--no-live-pmc', action='store_true',
                    help='roofline.traffic: do not measure it in this run (two child passes of rocprofv3 --pmc, ~20 s '
                         'each), take the figure of the latest committed profiles/r*_pmc.json')
    ap.add_argument('--no-roofline', action='store_true',
                    help='skip the per-kernel timing replays (the PMC passes of tools/profile_round.sh: '
                         'only the launches of real frames are to be counted)')
    ap.add_argument('--unfused', action='store_true',
                    help='operator-by-operator launches instead of the fused row chains')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--no-batched', action='store_true',
                    help='skip the frames-per-launch sweep (side measurement)')
    ap.add_argument('--no-handoff', action='store_true',
                    help='skip the side measurement with the NCHW -> NHWC hand-off inside the frame')
    ap.add_argument('--train-autograd', action='store_true',
                    help='with --train: the per-operator autograd path instead of the two-call '
                         'fused forward/backward of the trainable stack')
    ap.add_argument('--prefetch-depth', type=int, default=0,
                    help='with --train: frames of look-ahead whose FROZEN decoder forward runs as one batched launch '
                         'sequence (FusionTrainer(prefetch_depth=...)); 0 = the frames whose 16-row tiles are resident at '
                         'once (9 for 900 queries), 1 = round 3\'s one-frame look-ahead')
    ap.add_argument('--deterministic', action='store_true',
                    help='--train: FusionTrainer(deterministic=True) -- the backward accumulates order-free '
                         '(tc_radar_train_bwd_fused_det: bit-identical gradients run to run); the default keeps the float atomics')
    ap.add_argument('--no-prefetch', action='store_true',
                    help='with --train: do not enqueue the next iteration\'s frozen decoder forward while the host '
                         'solves the assignment')
    ap.add_argument('--train', action='store_true',
                    help='time one DDP training iteration of the fusion head (configs[2]) '
                         'instead of inference')
    ap.add_argument('--min-window-s', type=float, default=1.0,
                    help='the K-step window is repeated until this much time has been measured; '
                         'the MEDIAN window is reported (a 20-step window is 7 ms)')
    ap.add_argument('--warmup-s', type=float, default=0.5,
                    help='after the W warm-up steps, keep stepping until this much time has passed')
    ap.add_argument('--backend', default=None, choices=['nccl', 'gloo'],
                    help='torch.distributed backend (default: nccl = RCCL on the GPU)')
    ap.add_argument('--share-gpu', action='store_true',
                    help='testing on a 1-GPU box only: rank r uses GPU r %% device_count '
                         '(RCCL refuses two ranks on one GPU: use --backend gloo)')
    ap.add_argument('--no-pin', action='store_true',
                    help='multi-GPU: do not restrict a rank to the CPUs of the NUMA node of its GPU')
    ap.add_argument('--dry-run', action='store_true',
                    help='no GPU work: spawn / rendezvous (gloo) / rank census / the collectives of '
                         'the chosen mode on CPU tensors / the JSON line.  Runs in a CPU container')
    return ap.parse_args(argv)


# ---- one process per GPU from a single command (tools/dist_train.sh:7-9) -----------------
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def kfd_gpu_order(sys_root='/sys'):
    """PCI addresses of the GPUs in the order the ROCm runtime enumerates them: the KFD topology nodes with SIMDs,
    by node id (/sys/class/kfd/kfd/topology/nodes/<id>/properties: simd_count, domain, location_id = bus << 8 |
    device << 3 | function).  [] when the tree is not there or not readable."""
    import glob
    out = []
    try:
        paths = glob.glob(os.path.join(sys_root, 'class', 'kfd', 'kfd', 'topology', 'nodes', '[0-9]*', 'properties'))
        for p in sorted(paths, key=lambda q: int(os.path.basename(os.path.dirname(q)))):
            props = {}
            for line in open(p):
                f = line.split()
                if len(f) >= 2:
                    props[f[0]] = f[1]
            if int(props.get('simd_count', '0')) <= 0:
                continue                                    # a CPU node
            loc, dom = int(props['location_id']), int(props.get('domain', '0'))
            out.append('%04x:%02x:%02x.%x' % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7))
    except (OSError, ValueError, KeyError):
        return []
    return out


def _visible(order):
    """ROCR_VISIBLE_DEVICES, then HIP_/CUDA_VISIBLE_DEVICES (plain index lists only) applied to an enumeration"""
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var, '').strip()
        if var == 'CUDA_VISIBLE_DEVICES' and os.environ.get('HIP_VISIBLE_DEVICES', '').strip():
            continue
        if v and all(t.strip().isdigit() for t in v.split(',')):
            idx = [int(t) for t in v.split(',')]
            order = [order[i] for i in idx if i < len(order)]
    return order


def gpu_numa_topology(sys_root='/sys'):
    """[(pci address, numa node)] of the AMD GPUs of this host in the order HIP enumerates them: the KFD topology's
    node order when it is readable (with the *_VISIBLE_DEVICES index lists applied), else PCI order of
    /sys/class/drm/card*/device -- no GPU call, usable before torch is imported (main() checks the rank's entry
    against the device's own PCI id afterwards: cpu_affinity.pci_matches_hip)."""
    import glob
    gpus = {}
    for dev in glob.glob(os.path.join(sys_root, 'class', 'drm', 'card[0-9]*', 'device')):
        try:
            if open(os.path.join(dev, 'vendor')).read().strip().lower() != '0x1002':
                continue
            if not os.path.isdir(os.path.join(dev, 'drm')):
                continue
            # render-capable display/compute functions only (class 0x03xxxx display, 0x12xxxx processing accelerator)
            cls = open(os.path.join(dev, 'class')).read().strip().lower()
            if not (cls.startswith('0x03') or cls.startswith('0x12')):
                continue
            addr = os.path.basename(os.path.realpath(dev))
            gpus[addr] = int(open(os.path.join(dev, 'numa_node')).read().strip())
        except (OSError, ValueError):
            continue
    order = [a for a in kfd_gpu_order(sys_root) if a in gpus]
    if order:
        return [(a, gpus[a]) for a in _visible(order)]
    return sorted(gpus.items())


def parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(','):
        if not part:
            continue
        a, _, b = part.partition('-')
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def pin_to_gpu_numa_node(local_rank, sys_root='/sys'):
    """One process per GPU on a two-socket host (tools/dist_train.sh:7-9 leaves placement to the OS): restrict this
    rank to the CPUs of the NUMA node its GPU hangs off, BEFORE torch / HIP start their threads, so that launch
    threads, pinned staging buffers and the RCCL proxy thread live next to the GPU.  Returns what was done (for
    the JSON line); a host without the sysfs entries, or with numa_node -1, is left alone."""
    info = {'local_rank': int(local_rank), 'numa_node': None, 'cpus': None, 'pinned': False}
    try:
        topo = gpu_numa_topology(sys_root)
        if not topo or local_rank >= len(topo):
            info['why'] = 'no sysfs entry for GPU %d (%d AMD GPUs listed)' % (local_rank, len(topo))
            return info
        addr, node = topo[local_rank]
        info.update(pci=addr, numa_node=node)
        if node < 0:
            info['why'] = 'numa_node -1 (single-node host or not reported)'
            return info
        cpus = parse_cpulist(open(os.path.join(sys_root, 'devices', 'system', 'node', 'node%d' % node, 'cpulist')).read())
        allowed = cpus & set(os.sched_getaffinity(0))
        if not allowed:
            info['why'] = 'the node has no CPU this process may use'
            return info
        os.sched_setaffinity(0, allowed)
        info.update(cpus=len(allowed), pinned=True)
    except (OSError, ValueError, AttributeError) as e:
        info['why'] = repr(e)
    return info


def hip_pci_address(index):
    """'dddd:bb:dd.f' of HIP device `index` (torch device properties), or None"""
    try:
        p = torch.cuda.get_device_properties(index)
        return '%04x:%02x:%02x.0' % (int(p.pci_domain_id), int(p.pci_bus_id), int(p.pci_device_id))
    except (AttributeError, RuntimeError):
        return None


def check_pin_against_device(affinity, local, sys_root='/sys'):
    """After torch is up: was the sysfs guess (pin_to_gpu_numa_node ran before any GPU call) the PCI function HIP
    calls device `local`?  Recorded as cpu_affinity.pci_matches_hip; a pinned rank on the wrong node is moved."""
    if not affinity.get('pci'):
        return affinity
    mine = hip_pci_address(local)
    if mine is None:
        return affinity
    affinity['pci_matches_hip'] = mine[:10] == affinity['pci'][:10]        # domain:bus:device
    if affinity['pci_matches_hip'] or not affinity.get('pinned'):
        return affinity
    try:
        node = dict((a[:10], n) for a, n in gpu_numa_topology(sys_root)).get(mine[:10], -1)
        if node >= 0:
            cpus = parse_cpulist(open(os.path.join(sys_root, 'devices', 'system', 'node', 'node%d' % node,
                                                   'cpulist')).read())
            allowed = cpus & set(os.sched_getaffinity(0)) or cpus
            for tid in os.listdir('/proc/self/task'):          # torch's threads exist by now: move them all
                try:
                    os.sched_setaffinity(int(tid), allowed)
                except (OSError, ValueError):
                    pass
            affinity.update(pci=mine, numa_node=node, cpus=len(allowed), repinned=True)
    except (OSError, ValueError):
        pass
    return affinity


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` outside torchrun: start N fresh children (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set), one per GPU, and wait.  This parent never makes a GPU call (no
    torch import at all); rank 0 prints the JSON line on the inherited stdout."""
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   TRANSCAR_BENCH_LAUNCHER='bench.py')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.05)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:             # a rank died: the others would hang in a collective
                    q.terminate()
    return rc


def build_head(dev):
    sd = synth.make_state_dict(seed=3)
    head = T.build_head(configs.head_cfg())
    head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return head.to(dev).eval(), sd


def make_inputs(head, dev, shapes, batch, seed, host_feats=True):
    """Synthetic frame(s) of BASELINE.md section 3, resident on the device.
    host_feats=False: the iid N(0,1) maps are drawn on the device, channels-last (a lane of ten
    frames is 1.9 GB: a second per frame through numpy); `feats_np` is None then."""
    if host_feats:
        feats = synth.make_feats(shapes, seed=seed, batch=batch)      # iid N(0,1)
        nhwc = [ops.to_nhwc(torch.from_numpy(f).to(dev)) for f in feats]
    else:
        feats = None
        g = torch.Generator(device=dev)
        g.manual_seed(1000 + seed)
        nhwc = [torch.randn((batch * 6, h, w, 256), device=dev, generator=g)
                for (h, w) in configs.LEVEL_SHAPES[shapes]]
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
    # ONE side stream for every replay of the process: a process has four hardware queues and streams are dealt onto
    # them in creation order -- a fresh stream per call shifted which queue the head's ingest stream and the trainer's
    # look-ahead stream landed on, i.e. the drop-in and training side measurements moved by 5 % with the NUMBER of replays
    # the roofline happened to make before them
    s = getattr(time_events, 'stream', None)
    if s is None:
        s = time_events.stream = torch.cuda.Stream()
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


F16_MFMA_PEAK_TFLOPS = 2516.6  # dense f16 / bf16 matrix peak: 1024 flop / clk / SIMD x 1024 SIMDs x 2.4 GHz
L1_PATH_GBS_PER_CU = 64 * 2.4            # bytes per clock of a CU's vector-memory (L1) path x the clock in GHz
F16X2_PRODUCTS = 3             # v_mfma_f32_16x16x32_f16 per 32 k of one fp32-accurate product (chain.hip linear_step16h)
MATRIX_PATH_CODE = {'auto': 0, 'f32': 1, 'f16x2': 2}


def matrix_peak(f16x2):
    """(peak TFLOP/s, definition) of a chain kernel's roofline: fp32-EQUIVALENT flop per second the matrix pipe can
    deliver on that kernel's arithmetic."""
    if f16x2:
        return F16_MFMA_PEAK_TFLOPS / F16X2_PRODUCTS, ('dense f16 MFMA peak %.1f TFLOP/s / %d products per fp32-accurate '
                                                       'product (two-plane f16 operands)' % (F16_MFMA_PEAK_TFLOPS, F16X2_PRODUCTS))
    return F32_MFMA_PEAK_TFLOPS, 'dense f32 MFMA peak (v_mfma_f32_16x16x4_f32 = the fp32 vector rate)'


def roofline(head, inp, dev, matrix_path='auto', tile_rows=0):
    """Live timing of the kernels of the path.  The dominant one (largest share
    of a frame: the fused decoder row chain, 6 launches per frame) is reported
    against its roofline; the others ride along under "others".  The chain kernels of a launch with 16-row tiles
    (more than 2048 rows) compute on the f16 matrix cores unless matrix_path is 'f32': their peak is the f16 peak
    divided by the products per fp32-accurate product, the fraction of the f32 MFMA peak rides along."""
    B = inp['l2i'].shape[0]
    f16x2 = B * head.num_query > 2048 and matrix_path != 'f32'
    chain_peak, chain_peak_def = matrix_peak(f16x2)
    mp_code = MATRIX_PATH_CODE[matrix_path]
    Q, Cd, F = head.num_query, head.embed_dims, 512
    M = B * Q
    H, code, NL = 8, head.code_size, 24
    qpad = ((Q + 15) // 16) * 16
    tile_rows = tile_rows or getattr(roofline, 'tile_rows', 0)       # (tools/chain_stamps.py sets the attribute)
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

    # the inputs of decoder layer l as the forward hands them over: the states and reference points layer l - 1 wrote
    hs_of = [o['aux']['inter_states'][l].contiguous() for l in range(5)]
    ref_of = [o['aux']['inter_references'][l].contiguous() for l in range(5)]

    def run_chain(layer=3):
        L.check(lib.tc_decoder_layer_tail_fwd(
            C.byref(pv.layers[layer]), C.byref(pv.layers[layer + 1].self_attn.in_proj), C.byref(fv), B, Q, 6,
            code, attn_o.data_ptr(), hs_of[layer - 1].data_ptr(), qe.data_ptr(), inp['l2i'].data_ptr(),
            ref_of[layer - 1].data_ptr(), pc, float(inp['hw'][0]), float(inp['hw'][1]), hs_out.data_ptr(),
            ref_out.data_ptr(), qk.data_ptr(), vt.data_ptr(), qpad, (mp_code << 8) | int(tile_rows), cur_stream()), 'decoder_layer_tail')
    if getattr(roofline, 'chain_only', False):        # tools/chain_stamps.py: one launch, no timing
        run_chain()
        return None
    chain_b2b_ms = time_events(run_chain)         # back to back: weights L2-warm, the same taps every launch
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
        if f16x2:       # what tc_head_forward launches beside 16-row f16x2 chains: the staged two-plane core (self_attn.hip)
            L.check(lib.tc_sdpa_fwd_f16x2(qk_in.data_ptr(), vt_in.data_ptr(), qpad, ao.data_ptr(), Cd, B, Q, H, None, 0,
                                          cur_stream()), 'sdpa_f16x2')
        else:
            L.check(lib.tc_sdpa_fwd(qk_in.data_ptr(), qk_in.data_ptr() + Cd * 4, 2 * Cd,
                                    vt_in.data_ptr(), qpad, ao.data_ptr(), Cd, B, Q, H, cur_stream()), 'sdpa')
    attn_ms = time_events(run_attn)
    attn_flop = 4.0 * Q * Q * 32 * H * B
    # The decoder chain AS A FRAME LAUNCHES IT: behind an attention core (which has swept the L2s in between), not back
    # to back with itself -- the pair replayed, the attention core's own time taken off.  This is the duration the
    # rocprofv3 summary of the frame sequence shows (profiles/r5_kernel_stats.csv: 99.8 us against 90.6 back to back; r6: the same).
    # ... and as ANOTHER layer every launch (layers 1..4 in turn, each with its own weights, states and reference points),
    # as consecutive layers of a frame do: the 3.18 MB of packed planes and the camera taps of the moved reference points
    # then come from the Infinity Cache / HBM, not from L2s the previous replay of the same launch warmed.
    turn = [0]

    def run_pair():
        run_attn()
        run_chain(1 + turn[0] % 4)
        turn[0] += 1
    chain_ms = max(time_events(run_pair) - attn_ms, chain_b2b_ms)
    # -- fused radar chain (three fusion layers in one launch) on this frame's decoder outputs
    from transcar_amd.detr3d_head import head_options
    hs5 = o['aux']['inter_states'][-1].contiguous()
    ref5 = o['aux']['inter_references'][-1].contiguous()
    lbox = o['aux']['last_box'].contiguous()
    T_tok = int(inp['tokens'].shape[1])
    rws = torch.empty(lib.tc_head_workspace_bytes(C.byref(pv), B, T_tok), dtype=torch.uint8, device=dev)
    rcls = torch.empty((3, B, Q, head.cls_out_channels), device=dev)
    rbox = torch.empty((3, B, Q, code), device=dev)
    ropt = head_options(matrix_path=matrix_path, tile_rows=tile_rows or None)

    def run_radar():
        L.check(lib.tc_radar_fusion_fwd(
            C.byref(pv), hs5.data_ptr(), ref5.data_ptr(), lbox.data_ptr(), inp['tokens'].data_ptr(), B, T_tok,
            int(inp['pad_mult']), 0, 3, rcls.data_ptr(), rbox.data_ptr(), None, C.byref(ropt),
            rws.data_ptr(), rws.numel(), cur_stream()), 'radar_fusion')
    run_radar()                           # encoders + K/V once ...
    ropt.reuse_radar_kv = 1               # ... then the chain alone
    radar_ms = time_events(run_radar)
    radar_flop = 3 * 2.0 * M * (6 * Cd * Cd + 2 * Cd * F + 2 * Cd * code)        # the reference's flop (HEAD:538-729)
    # EXECUTED flop (the canonical count): the q projection and the out_proj of a fusion layer run only in
    # row tiles that hold a query with a radar return inside its gate (the others are x + 0 * (...), skipped); the
    # rows are ordered hits first per sample beyond one frame per launch, so a sample's n hit rows occupy
    # ceil(n / R) tiles (+ 1 where a tile straddles two samples)
    hits_l = o['aux']['radar_hit_counts'].cpu().numpy() > 0                         # [3, B, Q]
    R_tile = tile_rows or (4 if M <= 1024 else 8 if M <= 2048 else 32 if (M > 4096 and matrix_path != 'f32') else 16)
    gated_rows = 0
    for l in range(3):
        if M > 1024:                                                                # compacted (automatic rule)
            gated_rows += sum(min(Q, (-(-int(hits_l[l, b].sum()) // R_tile) + 1) * R_tile) for b in range(B))
        else:                                                                       # own order: tiles that hold a hit
            flat = hits_l[l].reshape(-1)
            pad = (-len(flat)) % R_tile
            tiles = np.concatenate([flat, np.zeros(pad, bool)]).reshape(-1, R_tile).any(1)
            gated_rows += int(tiles.sum()) * R_tile
    radar_flop_exec = radar_flop - 2.0 * (3 * M - gated_rows) * 2 * Cd * Cd
    kern = {
        'chain_kernel(decoder layer)': dict(
            bound='mfma', achieved=chain_flop / chain_ms / 1e9, peak=chain_peak, peak_definition=chain_peak_def,
            unit='TFLOP/s', ms=chain_ms, per_frame=6, alg_flop=chain_flop,
            ms_back_to_back=chain_b2b_ms,
            timing='HIP events over a 50-fold graph replay of [attention core, decoder chain of layer 1 + i % 4] minus the '
                   'attention core\'s own 50-fold replay: the chain as a frame launches it -- behind an attention core, another '
                   'layer\'s weights every launch (back to back with itself, warm L2s, it takes ms_back_to_back)',
            launches_per_frame='6 = 4 plain launches + 2 that carry the radar encoders (chain_dual_kernel: others)',
            arithmetic='f16x2 split operands, fp32 accumulate' if f16x2 else 'f32',
            # every workgroup streams the layer's packed weights (3.18 MB) through its CU's vector-memory path: what
            # binds the 4- / 8- / 16-row item loops (DESIGN.md section 9 "Round 4"); at 32 rows the stream is worth 14 % of
            # the kernel (section 5 "Round 5")
            weight_stream_gbs_per_cu=(-(-B * head.num_query // R_tile) * 795136 * 4 / 256.0) / chain_ms / 1e6,
            tile_rows=R_tile, workgroups=-(-B * head.num_query // R_tile),
            # ... against what a CU's vector-memory path delivers (64 B / clk at 2.4 GHz): the resource the kernel's item
            # loops run against (inside them ~57 of 64 B / clk, tools/split_mfma_probe.hip; averaged over the whole
            # kernel -- sampling, LayerNorms, epilogues included -- the figure below)
            weight_stream_peak_gbs_per_cu=L1_PATH_GBS_PER_CU,
            # ... and what a CU GETS when all 32 CUs of its XCD stream the same layer out of the one L2 (254 workgroups of 8
            # waves, one 16 KiB item in flight per wave: tools/wpolicy_probe.hip, profiles/r6_weight_stream_policy.txt, a
            # committed measurement, not taken in this run): the stream alone is then ~26 us of a nine-frame layer
            weight_stream_measured_ceiling_gbs_per_cu=124.0),
        'self_attn_kernel': dict(
            bound='mfma', achieved=attn_flop / attn_ms / 1e9, peak=chain_peak if f16x2 else F32_MFMA_PEAK_TFLOPS,
            unit='TFLOP/s', ms=attn_ms, per_frame=6, alg_flop=attn_flop,
            arithmetic='f16x2 split operands (self_attn_x_kernel), fp32 accumulate' if f16x2 else 'f32',
            note='issue bound, not matrix bound: per 32 keys x 16 queries a SIMD issues 12 MFMAs beside ~60 VALU '
                 'instructions (8 v_exp, the fp32 -> two-plane split of the probabilities)' if f16x2 else None),
        'chain_kernel(radar fusion)': dict(
            bound='mfma', achieved=radar_flop_exec / radar_ms / 1e9, peak=chain_peak,
            unit='TFLOP/s', ms=radar_ms, per_frame=1, alg_flop=radar_flop_exec, reference_flop=radar_flop,
            gated_rows=gated_rows, note='achieved / frac count EXECUTED flop: q-proj and out_proj only in row tiles '
                                        'that hold a query with a radar hit (the reference computes them for every query)'),
        'cam_sample_kernel': dict(
            bound='hbm', achieved=cam_bytes / cam_ms / 1e6, peak=HBM_PEAK_GBS, unit='GB/s',
            ms=cam_ms, per_frame=0, alg_bytes=cam_bytes, visible_pairs=pairs,
            note='stand-alone operator; inside the fused forward it is a step of the chain'),
    }
    for kk in kern.values():
        kk['frac'] = kk['achieved'] / kk['peak']
        if kk['bound'] == 'mfma':
            kk['frac_of_f32_mfma_peak'] = kk['achieved'] / F32_MFMA_PEAK_TFLOPS
    dom = max(kern, key=lambda n: kern[n]['ms'] * kern[n]['per_frame'])
    r = dict(kern[dom])
    # HBM-side traffic per launch comes from the committed PMC passes (rocprofv3 cannot
    # run inside this process); null when no profile of this kernel is committed
    traffic, src = None, None
    pmc = {}
    try:
        # {kernel: {"<frames per launch>": {"traffic_bytes": ...}}}, written by tools/collect_profile.py
        import glob
        newest = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc.json')),
                        key=lambda f: int(os.path.basename(f).split('_')[0][1:]))[-1]      # the latest round's passes
        pmc = json.load(open(newest))
        ent = pmc.get(dom, {}).get(str(B))
        if ent:
            traffic, src = ent['traffic_bytes'], 'profiles/' + os.path.basename(newest)
    except Exception:
        pass
    # algorithmic flop of the whole path per frame: 6 decoder chains (the last without the next
    # layer's QKV), 5 attention cores (layer 0's is a constant of the checkpoint), radar encoders
    # (T tokens) and 3 radar fusion layers
    nly = head.head_weights().num_layers
    path_flop = (nly * chain_flop - 2.0 * M * 3 * Cd * Cd
                 + (nly - 1) * attn_flop
                 + 2.0 * B * T_tok * (3 * Cd + Cd * Cd + 36 * 64 + 64 * 128 + 128 * Cd + 3 * Cd * 2 * Cd)
                 + radar_flop_exec) / B
    r.update(path_flop_per_frame=path_flop)
    r.update(kernel=dom, traffic=traffic, traffic_source=src,
             others={n: dict({kk: v[kk] for kk in ('bound', 'achieved', 'peak', 'unit', 'frac', 'ms', 'per_frame', 'arithmetic',
                                                  'frac_of_f32_mfma_peak') if kk in v},
                             traffic=(pmc.get(n, {}).get(str(B)) or {}).get('traffic_bytes'))
                     for n, v in kern.items() if n != dom})
    for n, v in kern.items():                       # (the radar chain's executed / reference flop ride along)
        if n != dom and 'reference_flop' in v:
            r['others'][n].update(alg_flop=v['alg_flop'], reference_flop=v['reference_flop'], note=v['note'])
    # the two decoder launches of a frame that carry the radar encoders (chain_dual_kernel) have no entry point of their
    # own to time here: their durations come from the committed kernel trace of the same command (profiles/)
    r['others'].update(dual_kernels_from_profile(B, chain_flop, chain_peak))
    return r


def dual_kernels_from_profile(B, chain_flop, chain_peak):
    import csv
    import glob
    import re
    out = {}
    try:
        cands = [f for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*_kernel_stats.csv'))
                 if re.match(r'^r\d+_kernel_stats\.csv$', os.path.basename(f))]
        newest = sorted(cands, key=lambda f: int(os.path.basename(f).split('_')[0][1:]))[-1]
        for row in csv.DictReader(open(newest)):
            m = re.search(r'chain_dual_kernel<(\d+), \d+, (\d+), (\d+)(?:, (?:true|false))?>', row['Name'])
            if m and int(row['Calls']) >= 10:
                ms = float(row['AverageNs']) * 1e-6
                part = {'4': 'A (encoders + K|V of fusion layer 1)', '5': 'B (K|V of fusion layers 2, 3)', '2': 'whole'}.get(m.group(2), m.group(2))
                out['chain_dual_kernel(decoder layer + radar encoders %s)' % part] = dict(
                    bound='mfma', unit='TFLOP/s', ms=ms, per_frame=1, tile_rows=int(m.group(1)), peak=chain_peak,
                    achieved=chain_flop / ms / 1e9, frac=chain_flop / ms / 1e9 / chain_peak,
                    note='duration from %s (nine frames per launch, one lane); flop = the decoder layer\'s only'
                         % ('profiles/' + os.path.basename(newest)))
    except Exception:
        pass
    return out


PMC_KERNELS = (('chain_kernel(decoder layer)', r'chain_kernel<\d+, 1, '),
               ('chain_kernel(radar fusion)', r'chain_kernel<\d+, 3, '),
               ('self_attn_kernel', r'self_attn_(x_)?kernel'))


def parse_counter_csv(path, ctr):
    """rocprofv3 `*counter_collection.csv` of one --pmc pass -> {kernel label: (mean counter value per launch,
    launches, mean duration in seconds)} for the path's kernels (the decoder chain's plain launches only: the two
    that carry the radar encoders are chain_dual_kernel)."""
    import csv
    import re
    rx = [(name, re.compile(pat)) for name, pat in PMC_KERNELS]
    acc = {}
    for row in csv.DictReader(open(path)):
        if row.get('Counter_Name') != ctr:
            continue
        for name, r in rx:
            if r.search(row['Kernel_Name']):
                a = acc.setdefault(name, [0.0, 0, 0.0])
                a[0] += float(row['Counter_Value'])
                a[1] += 1
                a[2] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-9
                break
    return {name: (v / n, n, dur / n) for name, (v, n, dur) in acc.items() if n}


def _pmc_tools():
    """(rocprofv3, python) for a --pmc child pass, or None: rocprofv3 missing, this process itself profiled, or the
    interpreter not a real ELF binary (under --pmc the profiler's preloaded library has initialised the GPU before the
    program starts, so a shim script that re-execs would be an exec from a GPU-initialised process, which this pool
    forbids -- ADVICE r3: no PATH lookup of "python3")."""
    import shutil
    if any(k.startswith('ROCPROF') or k.startswith('ROCP_') for k in os.environ):
        return None
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return None
    py = os.path.realpath(sys.executable or '')
    try:
        with open(py, 'rb') as f:
            if f.read(4) != b'\x7fELF':
                return None
    except OSError:
        return None
    return exe, py


def _pmc_pass(exe, py, tmp, ctr, bench_args, timeout_s):
    """One `rocprofv3 --kernel-trace --pmc <ctr> -- python3 bench.py <bench_args>` child (kernel trace only, the program
    itself behind `--`: the GPU box's rules) -> path of its counter csv, or None."""
    import glob
    import signal
    import subprocess
    out = os.path.join(tmp, ctr)
    cmd = [exe, '--kernel-trace', '--pmc', ctr, '--output-format', 'csv', '-d', out, '--', py,
           os.path.join(ROOT, 'bench.py')] + list(bench_args)
    env = dict(os.environ, TMPDIR='/tmp')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    # its own session: on a timeout the WHOLE group goes (rocprofv3 and the profiled bench.py behind it), or the
    # grandchild would keep running on the GPU beside the side measurements that follow
    proc = subprocess.Popen(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                            start_new_session=True)
    try:
        rc = proc.wait(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except OSError:
            pass
        proc.wait()
        return None
    files = glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True)
    return files[0] if rc == 0 and files else None


def iteration_traffic_from_csvs(fetch_csv, write_csv):
    """HBM-side bytes per TRAINING ITERATION from the two counter csv files of a `bench.py --train` child: every
    dispatch of the process counted, (2 FETCH_SIZE + WRITE_SIZE) * 1024 summed, divided by the iterations the child ran
    (= launches of adamw_kernel: exactly one per iteration) -> dict or None."""
    import csv
    tot, iters = {}, {}
    for ctr, path in (('FETCH_SIZE', fetch_csv), ('WRITE_SIZE', write_csv)):
        t, n = 0.0, 0
        for row in csv.DictReader(open(path)):
            if row.get('Counter_Name') != ctr:
                continue
            t += float(row['Counter_Value'])
            n += 'adamw_kernel' in row['Kernel_Name']
        tot[ctr], iters[ctr] = t, n
    if not iters['FETCH_SIZE'] or not iters['WRITE_SIZE']:
        return None
    fetch, write = tot['FETCH_SIZE'] / iters['FETCH_SIZE'], tot['WRITE_SIZE'] / iters['WRITE_SIZE']
    return dict(traffic_bytes=int((2 * fetch + write) * 1024), fetch_kb=round(fetch, 1), write_kb=round(write, 1),
                iterations=iters['FETCH_SIZE'])


def live_train_traffic(extra_args=(), timeout_s=120):
    """`train.roofline.traffic`: two --pmc child passes of `bench.py --train` (FETCH_SIZE, WRITE_SIZE;
    all kernels of an iteration incl. its share of the look-ahead decoder) -> iteration_traffic_from_csvs, or None."""
    import shutil
    import tempfile
    tools = _pmc_tools()
    if tools is None:
        return None
    tmp = tempfile.mkdtemp(prefix='tc_pmc_train_', dir='/tmp')
    try:
        paths = []
        for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
            pth = _pmc_pass(tools[0], tools[1], tmp, ctr, ['--train', '--steps', '9', '--warmup', '2', '--no-roofline', '--main-only',
                                                          '--min-window-s', '0.02', '--warmup-s', '0.02'] + list(extra_args),
                            timeout_s)
            if pth is None:
                return None
            paths.append(pth)
        return iteration_traffic_from_csvs(*paths)
    except (OSError, ValueError, KeyError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def live_traffic(frames_per_launch, timeout_s=75, extra_args=()):
    """HBM-side bytes per launch of the path's kernels, MEASURED in this run (the figure used to
    come from a committed profile): child passes `rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES --
    python3 bench.py --batch P --no-graph --main-only --steps 3` (separate passes, kernel trace only, the program itself
    behind `--`: the GPU box's rules), the counters averaged per launch and corrected as MI355X_MICROARCH.md's HBM
    section prescribes for gfx950: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  {} when rocprofv3 is missing, this
    process is itself being profiled, or a pass fails -- the caller keeps the committed figure then."""
    import shutil
    import subprocess
    import tempfile
    tools = _pmc_tools()
    if tools is None:
        return {}
    exe, py = tools
    got = {}
    tmp = tempfile.mkdtemp(prefix='tc_pmc_', dir='/tmp')
    try:
        for ctr in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_VALU_MFMA_BUSY_CYCLES'):
            csv_path = _pmc_pass(exe, py, tmp, ctr, ['--batch', str(frames_per_launch), '--steps', '3', '--warmup', '1',
                                                     '--main-only', '--no-graph', '--min-window-s', '0.02',
                                                     '--warmup-s', '0.02'] + list(extra_args), timeout_s)
            if csv_path is None:
                return {}
            for name, (v, n, dur) in parse_counter_csv(csv_path, ctr).items():
                got.setdefault(name, {})[ctr] = v
                got[name]['launches'] = n
                if ctr == 'SQ_VALU_MFMA_BUSY_CYCLES':
                    got[name]['busy_dur_s'] = dur
    except (OSError, ValueError, KeyError, subprocess.SubprocessError):
        return {}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {}
    for name, v in got.items():
        if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
            res[name] = dict(traffic_bytes=int((2 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024),
                             fetch_kb=round(v['FETCH_SIZE'], 1), write_kb=round(v['WRITE_SIZE'], 1), launches=v['launches'])
            if v.get('busy_dur_s'):
                # matrix-pipe busy cycles summed over the 1024 SIMDs / (1024 * duration * 2.4 GHz)
                res[name]['mfma_busy'] = v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * v['busy_dur_s'] * 2.4e9)
    return res


def _replay_rate(launch, sync, n, min_s=0.3):
    """seconds per launch() over windows of n launches, repeated until min_s has been measured
    (median window)"""
    wins, total = [], 0.0
    while total < min_s and len(wins) < 200:
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            launch()
        sync()
        wins.append(time.perf_counter() - t0)
        total += wins[-1]
    return float(np.median(wins)) / n


def single_lane(pipe, args):
    """One frame at a time on lane 0 (of a one-frame-per-launch pipeline).  `ms_per_frame_synced`: the host waits for every frame
    before it launches the next, as the reference's benchmark loop does
    (tools/analysis_tools/benchmark.py:64-91) = the latency of a frame incl. the replay launch;
    `ms_per_frame`: back-to-back replays on the one stream (no host wait in between)."""
    pipe.synchronize()

    def synced():
        pipe.launch(0)
        pipe.wait(0)
    for _ in range(10):
        synced()
    n = max(20, args.steps)
    t_sync = _replay_rate(synced, pipe.synchronize, n)
    t_b2b = _replay_rate(lambda: pipe.launch(0), pipe.synchronize, n)
    return {'frames_in_flight': 1, 'value': args.batch / t_b2b, 'unit': 'frames/s',
            'ms_per_frame': t_b2b * 1e3 / args.batch,
            'ms_per_frame_synced': t_sync * 1e3 / args.batch}


def handoff_side_run(head, dev, args, streams=None):
    """The same frame WITH the reference's hand-off inside it: the FPN returns NCHW maps
    (DET:62-66), so every lane's graph starts with the NCHW -> NHWC transposes of its four levels
    (tc_nchw_to_nhwc_levels, one launch).  A channels_last FPN skips this (zero-copy)."""
    from transcar_amd.pipeline import FramePipeline
    lanes = []
    fpl = args.batch * max(1, args.pair)
    for i in range(max(1, args.lanes)):
        inp = make_inputs(head, dev, args.shapes, fpl, seed=201 + 7 * i, host_feats=False)
        # the same maps as the FPN would hand them over: [B, N, C, H, W]
        inp['nchw'] = [f.view(fpl, 6, f.shape[1], f.shape[2], f.shape[3]).permute(0, 1, 4, 2, 3).contiguous()
                       for f in inp['nhwc']]
        lanes.append(inp)
    pipe = FramePipeline(head, lanes, tile_rows=args.tile_rows or None, streams=streams)
    for _ in range(3 * pipe.lanes):
        pipe.launch()
    n = max(20, args.steps)
    t = _replay_rate(pipe.launch, pipe.synchronize, n)
    t1 = _replay_rate(lambda: pipe.launch(0), pipe.synchronize, n)
    nbytes = 2 * sum(int(f.numel()) * 4 for f in lanes[0]['nchw'])

    def tr():
        ops.to_nhwc_levels(lanes[0]['nchw'], out=lanes[0]['nhwc'])
    tr_ms = time_events(tr)
    return {'frames_in_flight': pipe.lanes * fpl, 'frames_per_launch': fpl, 'value': fpl / t, 'unit': 'frames/s',
            'ms_per_frame': t * 1e3 / fpl, 'single_lane_ms_per_frame': t1 * 1e3 / fpl,
            'transpose': {'bound': 'hbm', 'ms': tr_ms, 'bytes': nbytes,
                          'achieved': nbytes / tr_ms / 1e6, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                          'frac': nbytes / tr_ms / 1e6 / HBM_PEAK_GBS}}


def producer_side_run(pipe, args):
    """Frames whose inputs ARRIVE: before every replay the producer (current stream) refills the
    lane's static inputs -- radar tokens and lidar2img from pinned host memory (H2D; they come from
    the data loader), and in the second figure also the four feature maps by a device copy (the
    FPN writing a new frame's maps into the lane).  Ordering is the pipeline's contract: the
    producer waits for the lane's previous replay (event), the lane waits for the producer."""
    lanes = pipe.lanes
    host = [dict(tokens=pipe.inputs[i]['tokens'].cpu().pin_memory(),
                 l2i=pipe.inputs[i]['l2i'].cpu().pin_memory()) for i in range(lanes)]
    spare = [f.clone() for f in pipe.inputs[0]['nhwc']]
    state = {'i': 0}

    def step(with_feats):
        i = state['i']
        state['i'] = (i + 1) % lanes
        pipe.write_inputs(i, l2i=host[i]['l2i'], tokens=host[i]['tokens'],
                          nhwc=spare if with_feats else None)
        pipe.launch(i)
    n = max(20, args.steps)
    fpl = pipe.frames_per_launch            # a replay = fpl frames; their inputs are written together
    for _ in range(2 * lanes):
        step(True)
    t_small = _replay_rate(lambda: step(False), torch.cuda.synchronize, n) / fpl
    t_all = _replay_rate(lambda: step(True), torch.cuda.synchronize, n) / fpl
    # restore lane 0's own frame (spare was a copy of it: nothing changed)
    return {'frames_in_flight': lanes * fpl,
            'tokens_l2i_h2d': {'value': 1.0 / t_small, 'unit': 'frames/s', 'ms_per_frame': t_small * 1e3},
            'tokens_l2i_h2d_plus_feature_copy': {
                'value': 1.0 / t_all, 'unit': 'frames/s', 'ms_per_frame': t_all * 1e3,
                'feature_bytes_per_frame': sum(int(f.numel()) * 4 for f in spare) // fpl}}


def sweep_side_run(head, dev, args, skip, streams=None):
    """Not the headline: the same pipeline at the other frames-per-launch settings (same lanes,
    the caller submits one frame per step everywhere), so that one bench line shows what pairing
    buys: 1 = one frame per launch (4-row tiles, a workgroup bound by its weight stream, DESIGN.md
    section 5), 2 = 8-row tiles, 4 = 16-row tiles on the 16x16x4 MFMA.  Decoder-chain time and
    fraction of the f32 MFMA peak of each setting ride along."""
    from transcar_amd.pipeline import FramePipeline
    out = {}
    for fpl in (1, 2, 4, 8, 9):
        if fpl == skip:
            continue
        lanes = [make_inputs(head, dev, args.shapes, fpl, seed=31 + 5 * i, host_feats=False)
                 for i in range(max(1, args.lanes))]
        from transcar_amd.detr3d_head import head_options as _ho
        pipe = FramePipeline(head, lanes, streams=streams, options=_ho(matrix_path=args.matrix_path))      # the headline pipeline's (idle) streams
        step = (lambda: pipe.launch()) if fpl == 1 else pipe.submit

        def sync():
            if fpl > 1:
                pipe.flush()
            torch.cuda.synchronize()
        t_end = time.perf_counter() + max(0.3, args.warmup_s)     # clocks ramp under load: same warm-up as the headline
        while time.perf_counter() < t_end:
            for _ in range(fpl * pipe.lanes):
                step()
            sync()
        # windows of WHOLE launches (round 2 timed max(20, steps) submits: at 8 frames per launch every window
        # ended in a flush of a partly filled lane and the sweep showed 8 below 4)
        rounds = max(2, -(-max(20, args.steps) // (fpl * pipe.lanes)))
        t = _replay_rate(step, sync, rounds * fpl * pipe.lanes, min_s=0.6)
        r = roofline(head, lanes[0], dev, args.matrix_path, tile_rows=args.tile_rows)
        allk = dict(r['others'])
        allk[r['kernel']] = r
        out[str(fpl)] = {'frames_per_launch': fpl, 'frames_in_flight': fpl * pipe.lanes,
                         'value': 1.0 / t, 'unit': 'frames/s',
                         'decoder_chain_us': allk['chain_kernel(decoder layer)']['ms'] * 1e3,
                         'decoder_chain_frac': allk['chain_kernel(decoder layer)']['frac'],
                         'self_attn_frac': allk['self_attn_kernel']['frac'],
                         'radar_chain_frac': allk['chain_kernel(radar fusion)']['frac']}
        del pipe, lanes
        torch.cuda.empty_cache()
    return out


def dropin_side_run(head, dev, args, n=40):
    """What a plugin-swap user calls, timed the reference's way (tools/analysis_tools/benchmark.py:64-91: one
    frame at a time, a device sync per frame): ``outs = head(mlvl_feats, img_metas)`` +
    ``head.get_bboxes(outs, img_metas)`` with the FPN maps handed over as [B,N,C,H,W] -- NCHW as the reference's
    FPN returns them (DET:62-66: one transposition launch per frame) and channels_last (taken zero-copy) -- and
    the radar as RAW sweeps in img_metas (the head builds the 36-feature tokens itself, HEAD:301-536)."""
    shapes = configs.LEVEL_SHAPES[args.shapes]
    frame = synth.make_radar_frame(seed=2)
    metas = synth.make_img_metas(1, synth.make_lidar2img(), radar=frame)
    out = {'method': 'head(mlvl_feats, img_metas) + get_bboxes, one frame at a time, device sync per frame '
                     '(tools/analysis_tools/benchmark.py:64-91); raw radar sweeps in img_metas',
           'radar_ingest': getattr(head, 'radar_ingest', 'host')}
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    for fmt in ('nchw', 'channels_last', 'batch9'):
        # batch9: the same entry handed nine frames at once ([9,6,256,H,W] channels_last maps, nine img_metas):
        # the library then runs 16-row tiles (8100 query rows), as a FramePipeline launch of nine frames does
        nb = 9 if fmt == 'batch9' else 1
        if nb > 1:
            metas = [synth.make_img_metas(1, synth.make_lidar2img(), radar=synth.make_radar_frame(seed=2 + i))[0]
                     for i in range(nb)]
        feats = [torch.randn((nb * 6, 256, h, w), device=dev, generator=g) for (h, w) in shapes]
        if fmt != 'nchw':
            feats = [f.to(memory_format=torch.channels_last) for f in feats]
        feats = [f.view(nb, 6, *f.shape[1:]) for f in feats]    # [B,N,C,H,W]

        def frame_once():
            outs = head(feats, metas)
            boxes = head.get_bboxes(outs, metas)
            torch.cuda.synchronize()
            return boxes
        for _ in range(5):
            frame_once()
        times = []
        for _ in range(n):
            t0 = time.perf_counter()
            frame_once()
            times.append(time.perf_counter() - t0)
        out[fmt] = {'ms_per_frame': float(np.median(times)) * 1e3 / nb, 'p99_ms': float(np.percentile(times, 99)) * 1e3,
                    'value': nb / float(np.median(times)), 'unit': 'frames/s', 'frames': n * nb,
                    'frames_per_call': nb}
        del feats
        torch.cuda.empty_cache()
    return out


def pipeline_latency_side_run(pipe, args, n_frames=None):
    """Submit -> result-ready latency of a frame in the HEADLINE configuration (frames_per_launch x lanes in
    flight), closed loop: the producer hands over frames as fast as the pipeline takes them, but re-fills a lane
    only once that lane's previous launch has completed (its outputs were consumed) -- at most
    lanes x frames_per_launch frames are in flight.  A frame's clock starts at its submit() and stops when the
    launch that carries it has finished on the device (a watcher thread blocks on one event per launch)."""
    import queue
    import threading
    P, nl = pipe.frames_per_launch, pipe.lanes
    n_frames = n_frames or max(20 * P * nl, 200)
    n_frames -= n_frames % P
    pend, lat = queue.Queue(), []

    def watcher():
        while True:
            item = pend.get()
            if item is None:
                return
            ev, stamps = item
            ev.synchronize()
            t = time.perf_counter()
            lat.extend(t - ts for ts in stamps)
    th = threading.Thread(target=watcher, daemon=True)
    th.start()
    pipe.synchronize()
    last_ev = [None] * nl
    stamps = []
    t_begin = time.perf_counter()
    for _ in range(n_frames):
        lane = pipe._fill_lane
        if pipe._filled[lane] == 0 and last_ev[lane] is not None:
            last_ev[lane].synchronize()
        stamps.append(time.perf_counter())
        lane, _, launched = pipe.submit()
        if launched:
            ev = torch.cuda.Event()
            ev.record(pipe.streams[lane])
            last_ev[lane] = ev
            pend.put((ev, stamps))
            stamps = []
    pend.put(None)
    th.join()
    pipe.synchronize()
    wall = time.perf_counter() - t_begin
    skip = P * nl                                      # the first launches start on an empty pipeline
    a = np.asarray(lat[skip:]) * 1e3
    return {'p50': float(np.percentile(a, 50)), 'p99': float(np.percentile(a, 99)), 'mean': float(a.mean()),
            'max': float(a.max()), 'unit': 'ms', 'frames': int(a.size), 'frames_per_launch': P, 'lanes': nl,
            'max_frames_in_flight': P * nl, 'closed_loop_frames_per_s': n_frames / wall,
            'definition': 'submit() of a frame -> its launch finished on the device; closed loop, a lane is re-filled '
                          'after its previous launch completed'}


def end_to_end_side_run(head, dev, args, n=12):
    """A PROXY for the detector's per-frame loop (tools/analysis_tools/benchmark.py:64-91 times the whole model):
    a stock PyTorch-ROCm conv backbone + FPN (plain torch.nn / MIOpen, channels_last fp32 -- a LOAD GENERATOR with
    the reference FPN's output contract: 4 levels x 256 channels at strides 8..64 of 6 x 928 x 1600 images, NOT the
    reference's ResNet-101-DCN and no part of this framework) feeds the head zero-copy; one frame at a time, device
    sync per frame.  Says what the head costs next to a convolutional producer and that the channels_last hand-off
    really is free at full size."""
    import torch.nn as nn

    class StockBackboneFPN(nn.Module):
        def __init__(self):
            super().__init__()
            def block(ci, co, s):
                return nn.Sequential(nn.Conv2d(ci, co, 3, s, 1), nn.ReLU(inplace=True))
            self.stem = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1))
            self.c3, self.c4, self.c5 = block(64, 128, 2), block(128, 256, 2), block(256, 512, 2)
            self.lat = nn.ModuleList([nn.Conv2d(c, 256, 1) for c in (128, 256, 512)])
            self.out = nn.ModuleList([nn.Conv2d(256, 256, 3, 1, 1) for _ in range(3)])
            self.extra = nn.Conv2d(256, 256, 3, 2, 1)            # FPN add_extra_convs='on_output' (CFG:44-51)

        def forward(self, img):
            c3 = self.c3(self.stem(img))
            c4 = self.c4(c3)
            c5 = self.c5(c4)
            p5 = self.lat[2](c5)
            p4 = self.lat[1](c4) + nn.functional.interpolate(p5, size=c4.shape[-2:], mode='nearest')
            p3 = self.lat[0](c3) + nn.functional.interpolate(p4, size=c3.shape[-2:], mode='nearest')
            outs = [self.out[0](p3), self.out[1](p4), self.out[2](p5)]
            outs.append(self.extra(outs[-1]))
            return outs

    if args.shapes != 'res101':
        return None
    net = StockBackboneFPN().to(dev).to(memory_format=torch.channels_last).eval()
    img = torch.randn((6, 3, configs.IMG_SHAPE[0], configs.IMG_SHAPE[1]), device=dev).to(memory_format=torch.channels_last)
    metas = synth.make_img_metas(1, synth.make_lidar2img(), radar=synth.make_radar_frame(seed=2))
    res = {}

    def run(with_head):
        feats = net(img)
        assert [tuple(f.shape[-2:]) for f in feats] == [tuple(x) for x in configs.LEVEL_SHAPES['res101']]
        zero_copy = all(f.is_contiguous(memory_format=torch.channels_last) and not f.is_contiguous() for f in feats)
        if with_head:
            outs = head([f.unsqueeze(0) for f in feats], metas)
            head.get_bboxes(outs, metas)
        torch.cuda.synchronize()
        return zero_copy
    for with_head in (False, True):
        for _ in range(2):
            zc = run(with_head)
        times = []
        for _ in range(n):
            t0 = time.perf_counter()
            run(with_head)
            times.append(time.perf_counter() - t0)
        res['backbone_fpn_plus_head_ms' if with_head else 'backbone_fpn_ms'] = float(np.median(times)) * 1e3
    res['head_share_ms'] = res['backbone_fpn_plus_head_ms'] - res['backbone_fpn_ms']
    res['fpn_output_taken_zero_copy'] = bool(zc)
    res['proxy'] = ('stock torch.nn conv backbone + FPN (12 conv layers, channels_last fp32, MIOpen) as a load generator '
                    'with the FPN output contract of CFG:44-51 -- not the reference ResNet-101-DCN')
    return res


def roofline_chain_once(head, inp, dev):
    """One launch of the decoder row chain exactly as `roofline` times it."""
    roofline.chain_only = True
    try:
        roofline(head, inp, dev)
    finally:
        roofline.chain_only = False


def host_cpu_info():
    """CPU model, physical / logical core counts of this host."""
    model, phys = 'unknown', set()
    try:
        pid = cid = None
        for line in open('/proc/cpuinfo'):
            k, _, v = line.partition(':')
            k, v = k.strip(), v.strip()
            if k == 'model name':
                model = v
            elif k == 'physical id':
                pid = v
            elif k == 'core id':
                cid = v
            elif not k and pid is not None:
                phys.add((pid, cid))
                pid = cid = None
        if pid is not None:
            phys.add((pid, cid))
    except OSError:
        pass
    logical = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:
        usable = logical
    return dict(cpu_model=model, physical_cores=len(phys) or logical, logical_cores=logical,
                usable_cores=usable)


def cpu_baseline(sd, inp, seconds):
    """The CPU oracle (oracle/transcar_oracle.py, a port of the reference's
    PyTorch path, proven equal to it on the golden fixtures) timed on this
    box's host cores on the same frame: 1 thread, all physical cores, and the
    best of a few thread counts (`value`, `cores`)."""
    from oracle import transcar_oracle as O
    tsd = O.to_torch_sd(sd)
    feats = [torch.from_numpy(f[:1]) for f in inp['feats_np']]
    l2i = torch.from_numpy(inp['l2i_np']).float()[None]
    f36 = inp['radar_feats'][0]
    info = host_cpu_info()
    pcr = configs.point_cloud_range
    rng = configs.pts_bbox_head['bbox_coder']['post_center_range']

    def frame():
        t0 = time.perf_counter()
        outs = O.head_forward(tsd, feats, l2i, inp['hw'], f36, pcr)
        O.get_bboxes(outs, rng)
        return time.perf_counter() - t0

    def run(n, warm, frames, budget):
        """`warm` warm-up forwards, then `frames` timed ones at n threads -> median / min.
        `budget` (seconds) bounds a thrashing thread count: the run stops early once it is spent (>= 1 timed frame)."""
        torch.set_num_threads(n)
        t_end = time.perf_counter() + budget
        for _ in range(warm):
            frame()
            if time.perf_counter() > t_end:
                break
        times = []
        while len(times) < frames and (not times or time.perf_counter() < t_end):
            times.append(frame())
        return dict(threads=n, warmups=warm, frames=len(times), ms_per_frame=float(np.median(times)) * 1e3,
                    min_ms_per_frame=float(np.min(times)) * 1e3)

    usable = info['usable_cores']
    allc = max(1, min(info['physical_cores'], usable))
    scale = max(0.25, seconds / 12.0)                    # --cpu-seconds 12 (default) = the full procedure
    with torch.no_grad():
        # which thread count is best: short probes (1 warm-up + 2 frames) -- the ops are small (900 x 256), all cores
        # of a big host thrash (73 s / frame at 256 threads in round 1), so the all-cores figure is a bounded probe too
        probes = {n: run(n, 1, 2, 3.0 * scale) for n in sorted({min(usable, c) for c in (8, 16, 32)} - {1})}
        every = run(allc, 1, 2, 4.0 * scale) if allc > 1 and allc not in probes else probes.get(allc)
        cand = dict(probes)
        if every is not None:
            cand[allc] = every
        best_n = min(cand, key=lambda n: cand[n]['ms_per_frame']) if cand else 1
        # the procedure itself: 5 warm-ups + 20 timed forwards at 1 thread and at the best count
        one = run(1, 5, 20, 16.0 * scale)
        best = run(best_n, 5, 20, 10.0 * scale) if best_n != 1 else one
        if one['ms_per_frame'] < best['ms_per_frame']:
            best = one
    out = dict(value=1e3 / best['ms_per_frame'], unit='frames/s', cores=best['threads'], kind='port',
               ms_per_frame=best['ms_per_frame'], min_ms_per_frame=best['min_ms_per_frame'],
               sample='BASELINE.md section 3: %d warm-ups + %d timed frames of the bench workload (B=1: Detr3DHead.forward + '
                      'box decode), torch CPU fp32, at 1 thread and at %d threads (the best of the probed counts %s); the '
                      'other counts are 2-frame probes' % (best['warmups'], best['frames'], best['threads'],
                                                          sorted(cand) if cand else [1]),
               one_thread=one, all_physical_cores=dict(every or one, note='bounded probe: 1 warm-up + 2 frames'),
               probes={str(n): r for n, r in probes.items()})
    out.update(info)
    out['box_to_box'] = cpu_baseline_spread(best['ms_per_frame'])
    return out


def cpu_baseline_spread(this_ms=None):
    """the CPU figure moves between boxes of the pool (and with the thread count the 2-frame probes
    pick).  The spread of `cpu_baseline.ms_per_frame` over the committed bench lines of all rounds (profiles/r*_bench.json,
    r*_driver_cmd.json: each a run on another box) and this run."""
    import glob
    vals = []
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench.json')) +
                    glob.glob(os.path.join(ROOT, 'profiles', 'r*_driver_cmd.json'))):
        try:
            with open(f) as fh:
                for ln in fh:
                    ln = ln.strip()
                    if ln.startswith('{'):
                        cb = json.loads(ln).get('cpu_baseline') or {}
                        if cb.get('ms_per_frame'):
                            vals.append((float(cb['ms_per_frame']), int(cb.get('cores', 0)), os.path.basename(f)))
                        break
        except (OSError, ValueError):
            continue
    ms = [v[0] for v in vals] + ([float(this_ms)] if this_ms else [])
    if not ms:
        return None
    return dict(runs=len(ms), min_ms_per_frame=min(ms), max_ms_per_frame=max(ms), median_ms_per_frame=float(np.median(ms)),
                threads_chosen=sorted({v[1] for v in vals}),
                note='cpu_baseline.ms_per_frame of the committed bench lines of every round (one box each) and of this run: '
                     'the baseline is a reported figure with this spread, not a constant')


def rank_census(dev, world):
    """Number of ranks the collective backend really connects: all_reduce of ones
    (RCCL on the GPU; 1 without a process group)."""
    if world == 1 or not torch.distributed.is_initialized():
        return 1
    t = torch.ones(1, dtype=torch.float32, device=dev)
    torch.distributed.all_reduce(t)
    return int(round(float(t.item())))


def timed_windows(step, sync, args, dev, world):
    """W warm-up steps (+ more until --warmup-s has passed), then windows of EXACTLY K steps, each
    bracketed by barrier + device sync on both sides and reduced with MAX over ranks, repeated
    until --min-window-s has been measured.  Every rank derives the loop counts from all-reduced
    numbers, so the ranks stay in step (a training step contains a collective)."""
    cpu_or_dev = dev if world > 1 else None
    t0 = time.perf_counter()
    for _ in range(args.warmup):
        step()
    sync()
    dt = D.max_over_ranks(time.perf_counter() - t0, cpu_or_dev)
    extra = 0
    if args.warmup > 0 and dt < args.warmup_s:
        extra = min(100000, int((args.warmup_s - dt) / max(dt / args.warmup, 1e-6)) + 1)
        for _ in range(extra):
            step()
    windows, own, total = [], [], 0.0
    while True:
        D.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        own.append(time.perf_counter() - t0)          # this rank alone, before it waits for the slowest one
        D.barrier()
        dt = D.max_over_ranks(time.perf_counter() - t0, cpu_or_dev)
        windows.append(dt)
        total += dt
        if total >= args.min_window_s or len(windows) >= 1000:
            break
    med = float(np.median(windows))
    return med, dict(windows=len(windows), warmup_steps_run=args.warmup + extra,
                     window_ms_min=min(windows) * 1e3, window_ms_median=med * 1e3,
                     window_ms_max=max(windows) * 1e3, own_window_ms_median=float(np.median(own)) * 1e3)


def per_rank_summary(own_ms, steps, frames_per_step):
    """frames/s of every rank from its OWN median window (no waiting for the slowest rank inside it):
    the spread says whether `value` (MAX over ranks) is one slow rank or all of them."""
    rates = [steps * frames_per_step / (ms * 1e-3) if ms > 0 else 0.0 for ms in own_ms]
    return {'frames_per_s': rates, 'min': min(rates), 'median': float(np.median(rates)), 'max': max(rates),
            'unit': 'frames/s', 'definition': 'steps / own median window of the rank (value uses the MAX over ranks)'}


class LookAheadFrames:
    """The synthetic loader of the training bench: P frames collated on the device, stored twice in a row so that
    every window of P consecutive frames (cyclically) is ONE contiguous view -- the look-ahead a data loader provides.
    ``next()``: the frame of this iteration (B samples); ``prefetch()``: the window of the P frames that follow it."""

    def __init__(self, head, dev, shapes, P, B, seed):
        self.P, self.B = P, B
        one = make_inputs(head, dev, shapes, P * B, seed=seed, host_feats=False)
        self.hw, self.pad_mult = one['hw'], one['pad_mult']
        self.nhwc = [torch.cat([f, f], 0).contiguous() for f in one['nhwc']]
        self.l2i = torch.cat([one['l2i'], one['l2i']], 0).contiguous()
        self.tokens = torch.cat([one['tokens'], one['tokens']], 0).contiguous()
        self.ncam = self.nhwc[0].shape[0] // (2 * P * B)
        self.cur = 0                             # iterations handed out so far
        self.fifo = []                           # positions of the frames of look-aheads under way
        del one

    def _view(self, pos, n):
        B, c = self.B, self.ncam
        return dict(nhwc=[f[c * B * pos:c * B * (pos + n)] for f in self.nhwc], l2i=self.l2i[B * pos:B * (pos + n)],
                    tokens=self.tokens[B * pos:B * (pos + n)], hw=self.hw, pad_mult=self.pad_mult)

    def frame(self, i):
        return self._view(i % self.P, 1)

    def window(self, start=0):
        return self._view(start % self.P, self.P)

    def next(self):
        """iteration i trains frame i % P; its tensors are the slice of the look-ahead window it belongs to (if any)"""
        pos = self.fifo.pop(0) if self.fifo else self.cur % self.P
        self.cur += 1
        return self._view(pos, 1)

    def prefetch(self, skip=0):
        """the P frames that follow the `skip` frames of a look-ahead still to be returned by next()"""
        s0 = (self.cur + skip) % self.P          # [s0, s0 + P) lies inside the doubled storage
        self.fifo.extend(range(s0, s0 + self.P))
        w = self._view(s0, self.P)
        return dict(feats_nhwc=w['nhwc'], lidar2img=w['l2i'], img_hw=w['hw'], tokens=w['tokens'], pad_mult=w['pad_mult'])


def _train_setup(head, dev, rank, B, prefetch_depth=1, deterministic=False):
    """A trainable copy of the head (tools/train.py's freeze list), its FusionTrainer and B synthetic GT sets."""
    from transcar_amd.trainer import FusionTrainer
    cfg = configs.head_cfg()
    cfg['train_cfg'] = configs.train_cfg_pts
    thead = T.build_head(cfg)
    thead.load_state_dict(head.state_dict(), strict=True)
    thead = thead.to(dev)
    gts, lbs = [], []
    for b in range(B):
        boxes, labels = synth.make_gt(seed=7 + rank * 16 + b, n=24)
        gt = torch.from_numpy(boxes).clone()
        gt[:, 2] += gt[:, 5] * 0.5
        gts.append(gt.to(dev))
        lbs.append(torch.from_numpy(labels).to(dev))
    was = torch.is_grad_enabled()
    torch.set_grad_enabled(True)
    try:
        tr = FusionTrainer(thead, prefetch_depth=prefetch_depth, deterministic=deterministic)
    finally:
        torch.set_grad_enabled(was)
    return thead, tr, gts, lbs


def train_bench(args, head, inp, dev, rank, world, affinity=None):
    """BASELINE.json configs[2]: batch-per-GPU 1 DDP training of the trainable
    (radar) part of the head.  A step = frozen decoder forward + radar stack
    forward (tc_radar_train_fwd) + Hungarian/focal/L1 loss (device kernels, scipy
    assignment on the host as in the reference) + HIP backward + ONE all-reduce of
    the flat gradient bucket over RCCL + device-side clip + AdamW + weight re-pack."""
    B = args.batch
    depth = 1 if (args.train_autograd or args.no_prefetch) else \
        (args.prefetch_depth or max(1, auto_frames_per_launch(head, dev) // B))
    thead, tr, gts, lbs = _train_setup(head, dev, rank, B, depth, deterministic=args.deterministic)
    torch.set_grad_enabled(True)
    last = {}

    # the next iterations' frames are known ahead (a data loader's look-ahead; here synthetic frames collated on the
    # device): their FROZEN decoder forward is enqueued, as one batched launch sequence, while the host solves an
    # iteration's assignment
    loader = LookAheadFrames(head, dev, args.shapes, depth, B, seed=1 + rank) if depth > 1 else None
    nxt = None if (args.train_autograd or args.no_prefetch) else (loader.prefetch if loader is not None else dict(
        feats_nhwc=inp['nhwc'], lidar2img=inp['l2i'], img_hw=inp['hw'], tokens=inp['tokens'], pad_mult=inp['pad_mult']))

    def step():
        f = loader.next() if loader is not None else inp
        if args.train_autograd:
            last['losses'] = tr.step_nhwc(f['nhwc'], f['l2i'], f['hw'], f['tokens'], f['pad_mult'], gts, lbs)
        else:
            last['losses'] = tr.step_fused_nhwc(f['nhwc'], f['l2i'], f['hw'], f['tokens'], f['pad_mult'],
                                                gts, lbs, prefetch=nxt)

    census = rank_census(dev, world)
    med, win = timed_windows(step, torch.cuda.synchronize, args, dev, world)
    own = D.gather_floats(win['own_window_ms_median'], dev if world > 1 else None)
    # what the gradient exchange costs the compute stream: between the end of the backward and the optimizer's first
    # kernel (events of 20 extra iterations, outside the timed windows).  One rank: no collective, the figure is the
    # events' own distance; more ranks: the all-reduce's exposed part (it starts asynchronously behind the backward and
    # runs beside the look-ahead decoder; FusionTrainer.step_fused_nhwc)
    exposed = None
    if not args.train_autograd:
        tr.exchange_events = []
        tr.exchange_chunk_events = []
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        gaps = sorted(a.elapsed_time(b) for a, b in tr.exchange_events)
        tr.exchange_events = None
        exposed = {'median': gaps[len(gaps) // 2], 'max': gaps[-1], 'iterations': len(gaps)}
        if tr.exchange_chunk_events:
            # the exchange in chunks (more than one rank): the compute stream's wait for each chunk's all-reduce, in the
            # order it waits (fusion layer 3, 2, 1, radar encoders) -- chunk k travels while chunk k + 1 is computed, so
            # only the last one should show
            per = list(zip(*[[evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1)] for evs in tr.exchange_chunk_events]))
            exposed['per_chunk_median'] = [sorted(c)[len(c) // 2] for c in per]
            exposed['chunk_bytes'] = [4 * (b - a) for a, b in tr.bucket.chunk_ranges]
        tr.exchange_chunk_events = None
    # the cost of the order-free backward beside the default ("report the cost in bench.py --train"):
    # the SAME trainer flipped to deterministic=True for three windows of the same length (one rank, fused path only)
    det_cost = None
    if world == 1 and not args.train_autograd and not args.deterministic and not args.main_only:
        tr.deterministic = True
        try:
            for _ in range(max(3, args.warmup)):
                step()
            torch.cuda.synchronize()
            wins = []
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                torch.cuda.synchronize()
                wins.append((time.perf_counter() - t0) / args.steps * 1e3)
            det_cost = {'ms_per_step': float(np.median(wins)), 'relative_to_default': float(np.median(wins)) / (med / args.steps * 1e3),
                        'what': 'FusionTrainer(deterministic=True): tc_radar_train_bwd_fused_det, bit-identical gradients run to run'}
        finally:
            tr.deterministic = False
    line = {
        'exposed_collective_ms': exposed,
        'deterministic': bool(args.deterministic),
        'deterministic_cost': det_cost,
        'per_rank': per_rank_summary(own, args.steps, B), 'cpu_affinity': affinity,
        'metric': 'training frames/sec: fusion head iteration (frozen DETR3D decoder fwd + radar '
                  'stack fwd/bwd + loss + grad all-reduce + AdamW), FPN features resident in HBM',
        'value': args.steps * B * world / med, 'unit': 'frames/s', 'n_gpus': world,
        'rccl_ranks': census,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': med / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic', 'timing': win,
        'config': {'workload': 'BASELINE.json configs[2]: %s FPN shapes, 900 queries, 255 radar points, '
                               '24 GT boxes, batch-per-GPU %d, DDP' % (args.shapes, B),
                   'decoder_lookahead_frames': depth,
                   'trainable_parameters': tr.bucket.numel,
                   'grad_bucket_bytes': tr.bucket.numel * 4,
                   'parallelism': 'dp%d, one flat-bucket all-reduce per step (%s)'
                                  % (world, backend_name(args)),
                   'launcher': launcher_name(),
                   'final_loss': float(sum(last['losses'].values()))},
    }
    # EVERY rank: the roofline's one full iteration carries the all-reduce of the loss normalisers (HEAD:889-902) --
    # run by rank 0 alone it waited for its peers for ever while they sat in the barrier below
    roof = None if args.no_roofline else train_roofline(tr, thead, loader.frame(0) if loader is not None else inp, gts, lbs,
                                                        nxt, dev, loader.window() if loader is not None else None)
    if rank == 0:
        if roof is not None:
            # HBM-side bytes per iteration, measured by two --pmc child passes of this command (world size 1 only: the
            # children run beside nothing then)
            if world == 1 and not args.no_live_pmc and not args.train_autograd:
                tt = live_train_traffic(extra_args=('--shapes', args.shapes) + (('--no-prefetch',) if args.no_prefetch else ()) +
                                        (('--deterministic',) if args.deterministic else ()))
                if tt is not None:
                    roof['traffic'], roof['traffic_source'] = tt['traffic_bytes'], 'this run'
                    roof['traffic_detail'] = dict(tt, unit='bytes per iteration = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 summed '
                                                           'over every dispatch / iterations (adamw_kernel launches)')
            line['roofline'] = roof
        print(json.dumps(line), flush=True)
    D.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


def train_roofline(tr, thead, inp, gts, lbs, nxt, dev, batch=None):
    """The training iteration's kernels timed live (HIP events around eager launches of the same C calls the timed
    loop makes, nothing else on the GPU) against the f32 MFMA peak.  Algorithmic flop of one frame (900 queries,
    T radar tokens): frozen decoder chain 1.431 GF per layer (the bench's inference figure) and the attention core;
    the trainable stack's forward = 3 fusion layers (all rows: nothing is skipped in training) + the encoders; its
    backward = the data gradients (the same products transposed, without the first layer's input) + the weight
    gradients (one product per weight over all rows)."""
    import ctypes as C
    head = thead
    lib = L.lib()
    B, T_tok = inp['l2i'].shape[0], int(inp['tokens'].shape[1])
    Q, Cd, F, code, ncls, RI = head.num_query, head.embed_dims, 512, head.code_size, head.cls_out_channels, 36
    M = B * Q

    def ev_time(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    # a full iteration once, so that tape / workspace / outputs of the stack exist
    tr.step_fused_nhwc(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'], inp['pad_mult'], gts, lbs, update=False)
    torch.cuda.synchronize()
    seed = 12345
    tr._pre = None                               # (the look-ahead of the iteration above is not consumed here)
    if batch is not None:                        # the decoder as the iterations run it: P frames per launch sequence
        Pn = int(batch['l2i'].shape[0]) // B
        dec_ms = ev_time(lambda: tr._decoder_forward(batch['nhwc'], batch['l2i'], batch['hw'], batch['tokens'],
                                                     batch['pad_mult'], seed, 1)) / Pn
    else:
        dec_ms = ev_time(lambda: tr._decoder_forward(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'], inp['pad_mult'], seed, 0))
    base = tr._decoder_forward(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'], inp['pad_mult'], seed, 0)
    aux = base['aux']
    hs_last, ref_last, last_box = aux['inter_states'][-1].contiguous(), aux['inter_references'][-1].contiguous(), aux['last_box']
    w, pv = head.head_weights(), head._packed_view
    all_cls = torch.empty((3, B, Q, ncls), device=dev)
    all_box = torch.empty((3, B, Q, code), device=dev)
    tape, bws = tr._tape, tr._bws
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def fwd():
        L.check(lib.tc_head_repack_trainable_ex(C.byref(w), C.byref(pv), 1, st), 'repack')
        L.check(lib.tc_radar_train_fwd_fused(C.byref(pv), hs_last.data_ptr(), ref_last.data_ptr(), last_box.data_ptr(),
                                             inp['tokens'].data_ptr(), B, T_tok, int(inp['pad_mult']), all_cls.data_ptr(),
                                             all_box.data_ptr(), tape.data_ptr(), tape.numel(), tr.dropout, seed, st), 'fwd')
    fwd_ms = ev_time(fwd)
    from transcar_amd.trainer import grad_table
    g = grad_table(head)
    d_cls, d_box = torch.randn_like(all_cls) * 1e-3, torch.randn_like(all_box) * 1e-3

    def bwd():
        L.check(lib.tc_radar_train_bwd_fused(C.byref(w), C.byref(g), hs_last.data_ptr(), last_box.data_ptr(),
                                             inp['tokens'].data_ptr(), B, T_tok, int(inp['pad_mult']), all_box.data_ptr(),
                                             d_cls.data_ptr(), d_box.data_ptr(), tape.data_ptr(), tape.numel(),
                                             bws.data_ptr(), bws.numel(), tr.dropout, seed, None, None, st), 'bwd')
    bwd_ms = ev_time(bwd)
    tr.bucket.zero_grad()
    layer = Cd * Cd * 6 + 2 * Cd * F + Cd * (code + ncls)           # MAC per row of one fusion layer's linears
    enc = 3 * Cd + Cd * Cd + RI * 64 + 64 * 128 + 128 * Cd + 3 * Cd * 2 * Cd
    stack_fwd = 2.0 * (3 * M * layer + B * T_tok * enc)
    dec = 6 * 2.0 * M * (5 * Cd * Cd + Cd * 24 + 2 * Cd * F + 3 * Cd * Cd + Cd * code) - 2.0 * M * 3 * Cd * Cd \
        + 6 * 4.0 * Q * Q * 32 * 8 * B
    stack_bwd = 2.0 * stack_fwd - 2.0 * M * Cd * Cd                   # no data gradient into the frozen decoder
    parts = {'frozen decoder forward (6 chains + 6 attention cores, train mode)': (dec, dec_ms),
             'stack forward (re-pack + encoder chain + fusion chain with tape)': (stack_fwd, fwd_ms),
             'stack backward (pack^T + backward chain + token side + grouped weight GEMM)': (stack_bwd, bwd_ms)}
    out = {'bound': 'mfma', 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'parts': {}}
    tot_f, tot_ms = 0.0, 0.0
    for k, (fl, ms) in parts.items():
        out['parts'][k] = {'alg_flop': fl, 'ms': ms, 'achieved': fl / ms / 1e9, 'frac': fl / ms / 1e9 / F32_MFMA_PEAK_TFLOPS}
        tot_f += fl
        tot_ms += ms
    out.update(alg_flop=tot_f, ms=tot_ms, achieved=tot_f / tot_ms / 1e9, frac=tot_f / tot_ms / 1e9 / F32_MFMA_PEAK_TFLOPS,
               kernel='the three device phases of an iteration (decoder forward, stack forward, stack backward)',
               traffic=None, decoder_lookahead_frames=tr.prefetch_depth,
               note='one frame per GPU (CFG:188): the trainable stack runs 4-row tiles, bound by the weight stream of a '
                    'workgroup (DESIGN.md section 5), not by the matrix pipe; the frozen decoder runs for '
                    '`decoder_lookahead_frames` frames per launch sequence (its ms is per frame)')
    return out


def _pipeline_rate(head, dev, args, shapes, fpl, options, streams=None, min_s=0.6, seed0=41):
    """frames/s of a FramePipeline (args.lanes lanes, fpl frames per launch, whole launches per window) and the
    pipeline's first lane inputs (for a roofline of the same geometry)."""
    from transcar_amd.pipeline import FramePipeline
    lanes = [make_inputs(head, dev, shapes, fpl, seed=seed0 + 5 * i, host_feats=False) for i in range(max(1, args.lanes))]
    pipe = FramePipeline(head, lanes, streams=streams, options=options)
    step = (lambda: pipe.launch()) if fpl == 1 else pipe.submit

    def sync():
        if fpl > 1:
            pipe.flush()
        torch.cuda.synchronize()
    t_end = time.perf_counter() + max(0.3, args.warmup_s)
    while time.perf_counter() < t_end:
        for _ in range(fpl * pipe.lanes):
            step()
        sync()
    rounds = max(2, -(-max(20, args.steps) // (fpl * pipe.lanes)))
    t = _replay_rate(step, sync, rounds * fpl * pipe.lanes, min_s=min_s)
    return 1.0 / t, pipe, lanes


def f32_path_side_run(head, dev, args, fpl, streams=None):
    """The SAME pipeline with tc_head_options.matrix_path = f32 (v_mfma_f32_16x16x4_f32 in the 16-row tiles, round 3's
    arithmetic): the plain-fp32 figure beside the headline."""
    from transcar_amd.detr3d_head import head_options
    rate, pipe, lanes = _pipeline_rate(head, dev, args, args.shapes, fpl, head_options(matrix_path='f32'), streams)
    r = roofline(head, lanes[0], dev, 'f32')
    allk = dict(r['others'])
    allk[r['kernel']] = r
    out = {'matrix_path': 'f32', 'value': rate, 'unit': 'frames/s', 'frames_per_launch': fpl,
           'decoder_chain_us': allk['chain_kernel(decoder layer)']['ms'] * 1e3,
           'decoder_chain_frac_of_f32_mfma_peak': allk['chain_kernel(decoder layer)']['frac'],
           'radar_chain_us': allk['chain_kernel(radar fusion)']['ms'] * 1e3}
    del pipe, lanes
    torch.cuda.empty_cache()
    return out


def vovnet_side_run(head, dev, args, fpl, streams=None):
    """BASELINE.json configs[4]'s FPN shapes (VoVNet, start_level = 0: 232x400 ... 29x50 = 757 MB of maps per frame;
    CFG_VOV:39-47) through the same pipeline: frames/s, the dominant kernel against its roofline, one frame at a time."""
    from transcar_amd.detr3d_head import head_options
    from transcar_amd.pipeline import FramePipeline
    rate, pipe, lanes = _pipeline_rate(head, dev, args, 'vovnet', fpl, head_options(matrix_path=args.matrix_path), streams,
                                       seed0=61)
    r = roofline(head, lanes[0], dev, args.matrix_path, tile_rows=args.tile_rows)
    one = make_inputs(head, dev, 'vovnet', 1, seed=67, host_feats=False)
    pipe1 = FramePipeline(head, [one], streams=streams)
    lat = single_lane(pipe1, args)
    out = {'workload': 'BASELINE.json configs[4] at one GPU: VoVNet FPN levels %s x 256 ch, 900 queries, 255 radar points'
                       % (configs.LEVEL_SHAPES['vovnet'],),
           'value': rate, 'unit': 'frames/s', 'frames_per_launch': fpl, 'lanes': pipe.lanes,
           'dominant_kernel': r['kernel'], 'dominant_kernel_us': r['ms'] * 1e3, 'dominant_kernel_frac': r['frac'],
           'dominant_kernel_frac_of_f32_mfma_peak': r.get('frac_of_f32_mfma_peak'),
           'latency_ms_per_frame': lat['ms_per_frame_synced']}
    del pipe, pipe1, lanes, one
    torch.cuda.empty_cache()
    return out


def train_side_run(head, inp, dev, args):
    """BASELINE.json configs[2] at one GPU inside the inference line: a short `--train` run --
    ms per iteration of the fused training path (frozen decoder forward in train mode, stack forward / backward as row
    chains, device loss, flat-bucket clip + AdamW), its launches per iteration and the roofline of its three phases."""
    depth = max(1, auto_frames_per_launch(head, dev))
    thead, tr, gts, lbs = _train_setup(head, dev, 0, 1, depth)
    loader = LookAheadFrames(head, dev, args.shapes, depth, 1, seed=3)
    nxt = loader.prefetch
    was = torch.is_grad_enabled()
    torch.set_grad_enabled(True)
    try:
        def step():
            f = loader.next()
            tr.step_fused_nhwc(f['nhwc'], f['l2i'], f['hw'], f['tokens'], f['pad_mult'], gts, lbs, prefetch=nxt)
        for _ in range(2 * depth):
            step()
        torch.cuda.synchronize()
        t = _replay_rate(step, torch.cuda.synchronize, 6 * depth, min_s=0.8)
        roof = train_roofline(tr, thead, loader.frame(0), gts, lbs, nxt, dev, loader.window())
    finally:
        torch.set_grad_enabled(was)
    out = {'workload': 'BASELINE.json configs[2] at one GPU: one frame per iteration (CFG:188), %s FPN shapes, 24 GT boxes'
                       % args.shapes,
           'ms_per_iteration': t * 1e3, 'value': 1.0 / t, 'unit': 'frames/s', 'prefetch_depth': getattr(tr, 'prefetch_depth', 1),
           'roofline': {k: roof[k] for k in ('bound', 'peak', 'unit', 'achieved', 'frac', 'ms', 'parts', 'note')}}
    del tr, thead
    torch.cuda.empty_cache()
    return out


def auto_frames_per_launch(head, dev):
    """--pair 0: the largest number of frames whose 16-row tiles are resident at once (two workgroups per CU):
    9 frames of 900 queries = 507 workgroups on the 512 slots of an MI355X.  The step count does not matter
    any more: a window of K steps is K // 9 full launches and ONE partial launch (FramePipeline.flush runs a
    graph over the filled slots only), e.g. the driver's --steps 20 = 9 + 9 + 2 frames on three lanes."""
    from transcar_amd.pipeline import resident_frames_per_launch
    return resident_frames_per_launch(head.num_query, dev)


def backend_name(args):
    b = args.backend or ('gloo' if args.dry_run else 'nccl')
    return 'RCCL' if b == 'nccl' else b


def launcher_name():
    return os.environ.get('TRANSCAR_BENCH_LAUNCHER') or \
        ('torch.distributed.run' if 'TORCHELASTIC_RUN_ID' in os.environ else 'single process')


def dry_run(args, affinity=None):
    """CPU stand-in of a rank (no GPU anywhere): rendezvous over gloo, the rank census, and the
    collectives of the chosen mode on CPU tensors of the real sizes -- inference: barrier + MAX of
    the timing only; training: one all-reduce of the 10 MB flat gradient bucket per step."""
    rank, world = D.init_process_group(backend='gloo')
    census = rank_census('cpu', world)
    bucket = None
    if args.train:
        from transcar_amd.trainer import FlatBucket
        cfg = configs.head_cfg()
        cfg['train_cfg'] = configs.train_cfg_pts
        bucket = FlatBucket(T.build_head(cfg).freeze_decoder().trainable_parameters())

    def step():
        if bucket is not None:
            bucket.grads.fill_(float(rank + 1))
            bucket.all_reduce()

    med, win = timed_windows(step, lambda: None, args, None, 1 if world == 1 else world)
    own = D.gather_floats(win['own_window_ms_median'])
    ok = True
    if bucket is not None and world > 1:
        ok = bool((bucket.grads == world * (world + 1) / 2).all())
    line = {'metric': 'dry run (CPU, gloo): launcher / rendezvous / collectives only', 'value': 0.0,
            'unit': 'frames/s', 'n_gpus': world, 'rccl_ranks': census, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': med / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'dry_run': True, 'timing': win, 'bucket_all_reduce_ok': ok, 'cpu_affinity': affinity,
            'per_rank': per_rank_summary(own, args.steps, 1), 'roofline': None,
            'cpu_baseline': None,
            'config': {'workload': 'none (dry run)', 'mode': 'train' if args.train else 'inference',
                       'launcher': launcher_name(),
                       'parallelism': 'dp%d (%s)' % (world, backend_name(args)),
                       'grad_bucket_bytes': bucket.numel * 4 if bucket is not None else 0}}
    if rank == 0:
        print(json.dumps(line), flush=True)
    D.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0 if (census == world and ok) else 1


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.main_only:
        args.no_roofline = args.no_handoff = args.no_batched = args.no_cpu_baseline = args.no_configs = True
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and args.gpus > 1:
        return spawn_ranks(args, argv)          # launcher parent: no torch, no GPU
    if env_world is not None and int(env_world) != args.gpus:
        print('bench.py: --gpus %d but WORLD_SIZE=%s: the launcher environment wins'
              % (args.gpus, env_world), file=sys.stderr)
    local = int(os.environ.get('LOCAL_RANK', 0))
    # before torch (its thread pools inherit the mask): a rank of a multi-GPU job lives on its GPU's NUMA node
    affinity = pin_to_gpu_numa_node(local) if (env_world is not None and int(env_world) > 1 and not args.no_pin) \
        else {'local_rank': local, 'pinned': False, 'why': 'single rank: all host cores (cpu_baseline uses them)'}
    _imports()
    if args.unfused:
        os.environ['TRANSCAR_UNFUSED'] = '1'
    if args.dry_run:
        return dry_run(args, affinity)
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs (--dry-run for the CPU launcher check)'
    if args.share_gpu:
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)              # before the RCCL communicator is created
    dev = torch.device('cuda', local)
    check_pin_against_device(affinity, local)
    rank, world = D.init_process_group(backend=args.backend)
    torch.set_grad_enabled(False)
    head, sd = build_head(dev)
    inp = make_inputs(head, dev, args.shapes, args.batch, seed=1 + rank)
    if args.train:
        return train_bench(args, head, inp, dev, rank, world, affinity)

    pipe, pair = None, 1
    if not args.no_graph:
        # a frame (13 kernel launches + decode) is captured once into a hipGraph per lane; lane i
        # works on its own synthetic frame(s).  --pair P: a lane holds P frame slots; the bench
        # submits ONE frame per step and the lane is replayed when its slots are filled
        from transcar_amd.pipeline import FramePipeline
        # automatic: the most frames whose 16-row tiles are resident at once (900 queries: 9 = 507 workgroups
        # on 512 slots); a window's remainder is one partial launch
        pair = args.pair if args.pair > 0 else auto_frames_per_launch(head, dev)
        args.pair = pair                      # the side runs use the same grouping
        fpl = args.batch * pair
        first = inp if pair == 1 else make_inputs(head, dev, args.shapes, fpl, seed=1 + rank, host_feats=False)
        lanes = [first] + [make_inputs(head, dev, args.shapes, fpl, seed=101 + rank + 7 * i, host_feats=False)
                           for i in range(1, max(1, args.lanes))]
        from transcar_amd.detr3d_head import head_options
        pipe = FramePipeline(head, lanes, options=head_options(tile_rows=args.tile_rows or None,
                                                              last_level_cls_only=args.last_cls_only,
                                                              radar_compact=False if args.no_radar_compact else None,
                                                              matrix_path=args.matrix_path,
                                                              cam_pregather=bool(args.pregather)))

    def step():
        if pipe is None:
            return one_step(head, inp)
        if pair == 1:
            return pipe.launch()[1]
        return pipe.submit()

    def sync():
        if pipe is not None and pair > 1:
            pipe.flush()                     # a window ends with every submitted frame launched
        torch.cuda.synchronize()

    census = rank_census(dev, world)
    med, win = timed_windows(step, sync, args, dev, world)
    own = D.gather_floats(win['own_window_ms_median'], dev if world > 1 else None)
    frames = args.steps * args.batch * world
    in_flight = 1 if pipe is None else min(pipe.lanes * pipe.frames_per_launch, args.steps * args.batch)
    line = {
        'metric': 'nuScenes frames/sec (6-cam+radar, 900 queries): fusion decoder '
                  '(Detr3DHead.forward + box decode), FPN features resident in HBM',
        'value': frames / med,
        'unit': 'frames/s',
        'n_gpus': world,
        'rccl_ranks': census,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': med / args.steps * 1e3,
        # 1 / throughput with `frames_in_flight` frames overlapping -- NOT the latency of a frame
        'throughput_inverse_ms_per_frame': med / args.steps * 1e3 / args.batch,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        # the arithmetic the path computes in: fp32 everywhere; the linear steps and the attention core of launches with 16-row tiles form each
        # fp32 product from two-plane f16 operands on the matrix cores (three MFMAs, fp32 accumulate) unless
        # --matrix-path f32 (`f32_path` carries that figure)
        'dtype': ('f32 (linear steps and attention core of the batched launches: f16x2 split operands, %d '
                  'v_mfma_f32_16x16x32_f16 products, fp32 accumulate; activations of the 32-row tiles held as two f16 planes; '
                  '4- / 8-row tiles: f32 MFMA)' % F16X2_PRODUCTS)
                 if (pipe is not None and pipe.tile_rows_of() >= 16 and args.matrix_path != 'f32') else 'f32',
        'data': 'synthetic',
        'timing': win,
        'per_rank': per_rank_summary(own, args.steps, args.batch),
        'cpu_affinity': affinity,
        'config': {'workload': '%s FPN levels %s x 256 ch (fp32, channels-last), 900 queries, 255 radar '
                               'points, %d frame(s)/step/GPU, %dxMI355X inference'
                               % ({'res101': 'BASELINE.json configs[1]: synthetic 6 cameras, ResNet-101',
                                   'vovnet': 'BASELINE.json configs[4] shapes (CFG_VOV:39-47) at %d GPU(s): synthetic 6 cameras, '
                                             'VoVNet' % world,
                                   'tiny': 'parity-test shapes (NOT a BASELINE.json configuration): 6 cameras, tiny'}[args.shapes],
                                  configs.LEVEL_SHAPES[args.shapes], args.batch, world),
                   'matrix_path': args.matrix_path,
                   'shapes': args.shapes, 'frames_per_step_per_gpu': args.batch,
                   'launch': 'eager' if pipe is None else 'hipGraph replay',
                   'frames_per_launch': 1 if pipe is None else pipe.frames_per_launch,
                   # a window of K steps never has more than K frames in flight
                   'frames_in_flight': in_flight,
                   'chain_tile_rows': args.tile_rows or 'auto',
                   'last_level_cls_only': bool(args.last_cls_only),
                   'launcher': launcher_name(),
                   'parallelism': 'dp%d (frames sharded, no data-path collective)' % world},
    }
    if rank == 0:
        if pipe is not None and not args.main_only:
            # one frame at a time, host sync per frame: the reference's own method
            # (tools/analysis_tools/benchmark.py:64-91) -- the latency of a frame
            from transcar_amd.pipeline import FramePipeline
            pipe1 = pipe if pipe.frames_per_launch == args.batch else \
                FramePipeline(head, [inp], options=head_options(radar_compact=False if args.no_radar_compact else None))
            line['single_lane'] = single_lane(pipe1, args)
            line['latency_ms_per_frame'] = line['single_lane']['ms_per_frame_synced']
        if not args.no_roofline:
            # the dominant kernel as the timed region launches it (frames_per_launch frames per launch)
            line['roofline'] = roofline(head, pipe.inputs[0] if pipe is not None else inp, dev, args.matrix_path, tile_rows=args.tile_rows)
            line['roofline']['frames_per_launch'] = 1 if pipe is None else pipe.frames_per_launch
            # every kernel of the path together, at the measured whole-job rate -- against the f32 MFMA peak (the
            # attention core and the small tiles compute on it; > 1 would only say that the f16x2 chains beat it)
            pf = line['roofline']['path_flop_per_frame']
            line['roofline']['path_achieved_tflops'] = pf * line['value'] / world / 1e12
            line['roofline']['path_frac'] = line['roofline']['path_achieved_tflops'] / F32_MFMA_PEAK_TFLOPS
            line['roofline']['path_frac_peak'] = 'f32 MFMA peak %.1f TFLOP/s' % F32_MFMA_PEAK_TFLOPS
            if world == 1 and not args.no_live_pmc and not args.no_graph:
                # HBM-side bytes of the same launches, measured now (two rocprofv3 --pmc child passes)
                rl = line['roofline']
                live = live_traffic(rl['frames_per_launch'], extra_args=('--matrix-path', args.matrix_path))
                src = 'this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE child passes, (2 F + W) * 1024'
                if rl['kernel'] in live:
                    rl['traffic_committed_profile'] = rl.get('traffic')
                    rl['traffic'], rl['traffic_source'] = live[rl['kernel']]['traffic_bytes'], src
                    if 'mfma_busy' in live[rl['kernel']]:
                        # SQ_VALU_MFMA_BUSY_CYCLES of the same launches (a third pass): matrix-pipe utilisation
                        rl['mfma_busy'] = live[rl['kernel']]['mfma_busy']
                for n_, o_ in rl['others'].items():
                    if n_ in live:
                        o_['traffic'], o_['traffic_source'] = live[n_]['traffic_bytes'], 'this run'
                        if 'mfma_busy' in live[n_]:
                            o_['mfma_busy'] = live[n_]['mfma_busy']
        if world == 1:
            if pipe is not None and not args.main_only:
                line['pipeline_latency_ms'] = pipeline_latency_side_run(pipe, args)
                line['dropin_forward'] = dropin_side_run(head, dev, args)
                line['with_input_delivery'] = producer_side_run(pipe, args)
                if not args.no_handoff:
                    line['end_to_end'] = end_to_end_side_run(head, dev, args)
            if not args.no_handoff and not args.no_graph:
                line['with_handoff'] = handoff_side_run(head, dev, args, streams=pipe.streams)
                line['with_handoff_ms'] = line['with_handoff']['ms_per_frame']
            if args.batch == 1 and not args.no_batched and pipe is not None:
                line['frames_per_launch_sweep'] = sweep_side_run(head, dev, args, pipe.frames_per_launch,
                                                                 streams=pipe.streams)
            if pipe is not None and not args.main_only and args.batch == 1:
                if args.matrix_path != 'f32' and pipe.tile_rows_of() >= 16:
                    line['f32_path'] = f32_path_side_run(head, dev, args, pipe.frames_per_launch, streams=pipe.streams)
                if args.shapes == 'res101' and not args.no_configs:
                    # configs[4] (VoVNet FPN shapes) and configs[2] (a training iteration) at one GPU, in the same line
                    line['vovnet'] = vovnet_side_run(head, dev, args, pipe.frames_per_launch, streams=pipe.streams)
                    line['train'] = train_side_run(head, inp, dev, args)
        # the reported CPU baseline (the oracle on this host's cores): in the line at every world size, so that
        # the N = 1 point of a scaling run and the plain bench line have one schema.  A rank of a multi-GPU job
        # is pinned to its GPU's NUMA node: `usable_cores` says what the baseline could use.
        if not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(sd, inp, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    D.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
