"""One data-parallel training iteration of the fusion head (BASELINE.json
configs[2]: batch-per-GPU 1, DDP, gradient all-reduce over RCCL/xGMI).

What the reference does per iteration (tools/train.py -> mmcv EpochBasedRunner
-> MMDistributedDataParallel + OptimizerHook, CFG:206-221):
forward -> loss dict -> sum -> backward -> bucketed NCCL all-reduce of the
2.6 M trainable gradients -> clip_grad_norm_(35) -> AdamW step.  Here:

  * the trainable, used parameters live in ONE flat fp32 buffer and their
    gradients in another (parameters and .grad are views): the whole gradient
    exchange is a single all-reduce of 10.5 MB, issued on the
    RCCL stream by torch.distributed (backend "nccl" = RCCL on ROCm);
  * sum-of-squares, clip coefficient, 1/world_size and the AdamW update run on
    the flat buffers in two HIP kernels (tc_sq_norm, tc_adamw_step) -- the clip
    coefficient never leaves the device, so the step has no host sync;
  * the packed weights of the fused forward chains are refreshed in place.
"""
import ctypes as C
import math

import torch
import torch.distributed as dist

from . import _lib as L


def exchange_chunk_of(name):
    """Chunk of the gradient exchange a trainable parameter of the fusion stack belongs to, in the order the backward
    finishes the weight gradients (tc_radar_train_bwd_weights): 0 / 1 / 2 = fusion layer 3 / 2 / 1 (HEAD:129-171: the
    modules with suffix `3` / `_3`, `2` / `_2`, none), 3 = the radar encoders and anything else."""
    head = name.split('.')[0]
    stems = ('final_cls', 'final_reg', 'rf_multihead_attn', 'rf_linear1', 'rf_linear2', 'rf_norm1', 'rf_norm2', 'rf_norm3')
    for stem in stems:
        if head.startswith(stem):
            tail = head[len(stem):]
            if tail in ('3', '_3'):
                return 0
            if tail in ('2', '_2'):
                return 1
            if tail == '':
                return 2
    return 3


class FlatBucket:
    """Flat parameter / gradient storage for [(name, parameter)].

    chunk_of (round 6): name -> chunk index.  The parameters of a chunk are stored contiguously, chunk 0 first
    (``chunk_ranges``: [(begin, end)] element ranges): the gradient exchange can then travel chunk by chunk while the
    backward is still producing the next one (``all_reduce_chunk_begin``).  ``names`` / ``items`` / ``offsets`` stay in
    the order of ``named_params``."""

    def __init__(self, named_params, chunk_of=None):
        self.names = [n for n, _ in named_params]
        self.items = [p for _, p in named_params]
        if not self.items:
            raise ValueError('no trainable parameters')
        chunks = [int(chunk_of(n)) if chunk_of is not None else 0 for n in self.names]
        order = sorted(range(len(self.items)), key=lambda i: (chunks[i], i))
        dev = self.items[0].device
        n = sum(p.numel() for p in self.items)
        self.params = torch.empty(n, dtype=torch.float32, device=dev)
        # one allocation: the gradients and, behind them, the squared-norm slots of the clip (tc_sq_norm: one per
        # workgroup, added up in index order by tc_adamw_step; zeroed by the same fill)
        self._store = torch.zeros(n + L.TC_SQ_NORM_PARTIALS, dtype=torch.float32, device=dev)
        self.grads = self._store[:n]
        self.sq = self._store[n:]
        self.offsets = [0] * len(self.items)
        self.chunk_ranges = []
        off = 0
        for i in order:
            p = self.items[i]
            k = p.numel()
            self.params[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.params[off:off + k].view(p.shape)
            p.grad = self.grads[off:off + k].view(p.shape)
            self.offsets[i] = off
            if not self.chunk_ranges or chunks[i] != self._last_chunk:
                self.chunk_ranges.append([off, off])
                self._last_chunk = chunks[i]
            off += k
            self.chunk_ranges[-1][1] = off
        self.chunk_ranges = [tuple(r) for r in self.chunk_ranges]
        self.numel = n

    def zero_grad(self, check_views=True):
        self._store.zero_()
        if not check_views:          # nobody re-bound a .grad since the last call (the fused iteration)
            return
        for p, off in zip(self.items, self.offsets):     # a None grad would detach the view
            if p.grad is None or p.grad.data_ptr() != self.grads.data_ptr() + 4 * off:
                p.grad = self.grads[off:off + p.numel()].view(p.shape)

    def all_reduce(self):
        """SUM over ranks, one collective for the whole bucket."""
        return self.all_reduce_end(self.all_reduce_begin())

    def all_reduce_begin(self):
        """Start the bucket's ONE all-reduce (SUM) without waiting for it: ordered behind everything already enqueued on
        the current stream (the backward's weight-gradient kernel), it runs on the backend's communication stream
        (RCCL) beside whatever other streams have in flight -- the look-ahead of the frozen decoder.  -> handle."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return dist.all_reduce(self.grads, op=dist.ReduceOp.SUM, async_op=True), dist.get_world_size()
        return None, 1

    def all_reduce_chunk_begin(self, i):
        """Start the all-reduce (SUM) of chunk i alone -- ordered behind what is enqueued on the current stream (the launch
        that produced the chunk's last gradients), beside whatever the compute stream does next (the next chunk's weight
        gradients).  -> (handle or None, world size), for ``all_reduce_end``."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            a, b = self.chunk_ranges[i]
            return dist.all_reduce(self.grads[a:b], op=dist.ReduceOp.SUM, async_op=True), dist.get_world_size()
        return None, 1

    @staticmethod
    def all_reduce_end(pending):
        """The current stream (not the host, on RCCL) waits for the exchange -- one handle or a list of chunk handles, in
        the order they were started; -> world size."""
        world = 1
        for work, w in (pending if isinstance(pending, list) else [pending]):
            if work is not None:
                work.wait()
            world = w
        return world


def cosine_lr(base_lr, cur_iter, cur_epoch, max_epochs, warmup_iters=4000, warmup_ratio=1.0 / 3,
              min_lr_ratio=1e-3):
    """CFG:216-221: mmcv CosineAnnealing (by epoch) with linear warm-up (by
    iteration) [3p-memory: mmcv.runner.hooks.lr_updater]."""
    target = base_lr * min_lr_ratio
    regular = target + 0.5 * (base_lr - target) * (1 + math.cos(math.pi * cur_epoch / max_epochs))
    if cur_iter < warmup_iters:
        k = (1 - cur_iter / warmup_iters) * (1 - warmup_ratio)
        return regular * (1 - k)
    return regular


def _g(t):
    return None if t is None or t.grad is None else C.c_void_p(t.grad.data_ptr())


def grad_table(head):
    """tc_head_weights whose pointers are the .grad buffers of the trainable stack's
    parameters (the table tc_radar_train_bwd adds into)."""
    def lin(m):
        return L.tc_linear(_g(m.weight), _g(m.bias))

    def ln(m):
        return L.tc_lnorm(_g(m.weight), _g(m.bias))

    g = L.tc_head_weights()
    g.abi_version = L.TC_ABI_VERSION
    rpe, rfe = head.radar_position_encoder, head.radar_feat_encoder
    g.radar_position_encoder = L.tc_pos_encoder(lin(rpe[0]), ln(rpe[1]), lin(rpe[3]), ln(rpe[4]))
    g.radar_feat0, g.radar_feat2, g.radar_feat4 = lin(rfe[0]), lin(rfe[2]), lin(rfe[4])
    for r, (sfx, asfx) in enumerate((('', ''), ('_2', '2'), ('_3', '3'))):
        rl = g.radar[r]
        attn = getattr(head, 'rf_multihead_attn' + asfx)
        rl.attn = L.tc_mha(L.tc_linear(_g(attn.in_proj_weight), _g(attn.in_proj_bias)),
                           lin(attn.out_proj))
        rl.norm2, rl.norm3 = ln(getattr(head, 'rf_norm2' + sfx)), ln(getattr(head, 'rf_norm3' + sfx))
        rl.linear1, rl.linear2 = lin(getattr(head, 'rf_linear1' + sfx)), lin(getattr(head, 'rf_linear2' + sfx))
        fc, fr = getattr(head, 'final_cls' + asfx), getattr(head, 'final_reg' + asfx)
        rl.final_cls = L.tc_cls_branch(lin(fc[0]), ln(fc[1]), lin(fc[3]), ln(fc[4]), lin(fc[6]))
        rl.final_reg = L.tc_reg_branch(lin(fr[0]), lin(fr[2]), lin(fr[4]))
    return g


class FusionTrainer:
    """head: transcar_amd.Detr3DHead on the GPU, built with ``train_cfg``."""

    #: coefficient of the forward counter in Detr3DHead.next_dropout_seed: consecutive training forwards draw
    #: seeds this far apart, which is what lets ONE launch give every frame of a look-ahead batch its own masks
    SEED_STRIDE = 0x85EBCA77

    def __init__(self, head, lr=1.5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01,
                 max_norm=35.0, device_loss=True, dropout=0.1, seed=0, decoder_dropout=None,
                 chain_forward=True, chain_backward=True, prefetch_depth=1, deterministic=False):
        self.head = head.freeze_decoder()
        # deterministic=True (round 5): the fused backward accumulates order-free
        # (tc_radar_train_bwd_fused_det: integer atomics on a fixed-point shadow of the gradient bucket and of dK | dV) --
        # two runs from the same state give bit-identical gradients and parameters, on any schedule.  The default keeps
        # the float atomics (what torch's own backward does in the reference): bench.py --train reports both.
        self.deterministic = bool(deterministic)
        if self.deterministic and not chain_backward:
            raise ValueError('FusionTrainer(deterministic=True) needs the fused backward (chain_backward=True): the '
                             'per-operator backward (tc_radar_train_bwd) accumulates with float atomics')
        self._shadow = None
        # Round 6: the bucket is laid out in EXCHANGE CHUNKS -- fusion layer 3, 2, 1, radar encoders,
        # the order in which the backward finishes their weight gradients -- and with more than one rank an iteration
        # launches the weight gradients chunk by chunk (tc_radar_train_bwd_weights) with one asynchronous all-reduce
        # behind each: chunk k travels while chunk k + 1 is computed (`chunked_exchange`; tools/train.py:253-260: DDP's
        # buckets).  One rank, or deterministic=True (the order-free backward flushes its shadow once, behind ONE grouped
        # launch): the single grouped launch and the single all-reduce of round 5.
        self.chunked_exchange = True
        self.bucket = FlatBucket(head.trainable_parameters(), chunk_of=exchange_chunk_of)
        head.refresh_weights()                      # parameter addresses moved into the bucket
        self.m = torch.zeros_like(self.bucket.params)
        self.v = torch.zeros_like(self.bucket.params)
        self.sq = self.bucket.sq               # zeroed with the gradients (FlatBucket.zero_grad)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.weight_decay, self.max_norm = weight_decay, max_norm
        self.iter = 0
        self.device_loss = device_loss     # step_fused_nhwc: losses + their gradients from HIP kernels
        # step_fused_nhwc: dropout of the fusion layers as the reference trains them (HEAD:129-171,
        # p = 0.1); counter-based masks from (seed, iteration).  The frozen decoder and the
        # per-operator autograd path (step / step_nhwc) run without dropout.
        self.dropout, self.seed = float(dropout), int(seed)
        head.dropout_seed = int(seed)
        # the frozen decoder's own dropouts (the reference leaves them active, tools/train.py:245-252):
        # None = as the modules are configured (0.1) when the fusion layers train with dropout, else off
        if decoder_dropout is None:
            decoder_dropout = head.decoder_dropout_p() if self.dropout > 0 else 0.0
        self.decoder_dropout = float(decoder_dropout)
        # step_fused_nhwc: the stack's forward as launches of the fused row chains with the tape stored as it
        # is produced (tc_radar_train_fwd_fused: 3 launches + 1 re-pack instead of ~66); False = the
        # operator-by-operator forward tc_radar_train_fwd (the same tape: the cross-check of the tests)
        self.chain_forward = bool(chain_forward)
        # ... and its backward as one launch of the backward row chain (query side), a handful for the token
        # side and one grouped weight-gradient launch (tc_radar_train_bwd_fused: ~16 launches instead of ~130);
        # False = tc_radar_train_bwd, one launch per operator
        self.chain_backward = bool(chain_backward)
        self._sq_clean = False
        # Look-ahead of the FROZEN decoder (tools/train.py:245-252: nothing an iteration trains feeds it).  1 = the next
        # frame's decoder forward runs on a side stream during this iteration (round 3: one frame, 4-row tiles, 0.44 ms
        # of a 1.0 ms iteration at 19 % of the matrix peak).  P > 1 (round 4): the decoder forwards of the next P
        # frames as ONE batched launch sequence (16-row tiles on the f16 matrix cores, the inference rate), every
        # frame with the dropout masks of its own seed (tc_head_options.dropout_seed_stride); the iterations then
        # consume the batch frame by frame.  The decoder's arithmetic is fixed by the depth, not by whether a frame
        # was prefetched (`decoder_tile_rows`): with P > 1 a frame that missed the look-ahead runs the same 16- / 32-row
        # kernels alone, so losses and parameters do not depend on the schedule.
        # Round 5: a look-ahead of more than 4 096 rows (nine frames: 8 100) runs the 32-row tiles, like inference.
        self.prefetch_depth = max(1, int(prefetch_depth))
        rows = self.prefetch_depth * int(head.num_query)
        self.decoder_tile_rows = (32 if rows > 4096 else 16) if self.prefetch_depth > 1 else None

    def _decoder_tile_rows_now(self):
        """`decoder_tile_rows`, resolved at the call: 32-row tiles exist on the f16x2 matrix path only (chain.hip
        tile_rows()), so with ``decoder_matrix_path = 'f32'`` -- or after the f16-range guard's fall-back -- a depth
        that asks for 32 rows runs the f32 kernels' 16 (every look-ahead raised otherwise)."""
        rows = self.decoder_tile_rows
        mp = getattr(self, 'decoder_matrix_path', None)
        if rows == 32 and (mp == 'f32' or (mp in (None, 'auto') and self.head.matrix_fallback)):
            return 16
        return rows

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def backward_and_step(self, losses, lr=None):
        """losses: dict from head.loss(); every key containing 'loss' is summed
        (mmdet BaseDetector._parse_losses)."""
        total = sum(v for k, v in losses.items() if 'loss' in k)
        self.bucket.zero_grad()
        total.backward()
        self._optimizer_step(lr)
        return total.detach()

    # ------------------------------------------------------------------
    # fast path: the trainable stack as two C calls (tc_radar_train_fwd / _bwd) instead of
    # ~160 autograd nodes; torch autograd only differentiates the loss itself
    # ------------------------------------------------------------------
    def _decoder_forward(self, feats_nhwc, lidar2img, img_hw, tokens, pad_mult, seed, lane, lookahead=False):
        """The frozen decoder (train mode) on B >= 1 frames.  lookahead: the frames are consumed one per iteration, frame
        b draws the masks of seed + b * SEED_STRIDE (the seed of the iteration that will take it).  An ordinary batch
        (B > 1 frames of ONE iteration) draws from one index space with the iteration's seed: with the stride, sample 1
        of iteration n would repeat the masks of sample 0 of iteration n + 1."""
        from .detr3d_head import head_options
        with torch.no_grad():
            return self.head.forward_nhwc(feats_nhwc, lidar2img, img_hw, tokens, pad_mult, aux='train',
                                          _allow_train=True, decoder_only=True, lane=lane,
                                          options=head_options(tile_rows=self._decoder_tile_rows_now(),
                                                               matrix_path=getattr(self, 'decoder_matrix_path', None),
                                                               decoder_dropout_p=self.decoder_dropout,
                                                               dropout_seed=seed,
                                                               dropout_seed_stride=self.SEED_STRIDE if lookahead else 0))

    def _input_key(self, feats_nhwc, lidar2img, tokens, pad_mult):
        # address, shape, the tensors' version counters AND this trainer's look-ahead generation.  The version counter
        # sees torch's in-place operators only: a loader that refills the same static tensors through this package's
        # raw-pointer kernels (ops.to_nhwc into a preallocated buffer, RadarRawStage.build, FramePipeline write hooks,
        # any ctypes tc_* call) must call ``invalidate_lookahead()`` -- it bumps the generation, and a pending
        # look-ahead of the old contents is dropped instead of handing their decoder states to the new frame
        return (int(getattr(self, '_generation', 0)),
                tuple((int(f.data_ptr()), int(f._version), tuple(f.shape)) for f in feats_nhwc),
                (int(lidar2img.data_ptr()), int(lidar2img._version)), (int(tokens.data_ptr()), int(tokens._version)),
                int(pad_mult), tuple(tokens.shape))

    def invalidate_lookahead(self):
        """The tensors handed to ``prefetch_decoder`` / ``step_fused_nhwc(prefetch=...)`` were rewritten in place by
        something torch's version counters do not see: pending look-ahead frames are stale."""
        self._generation = int(getattr(self, '_generation', 0)) + 1
        self._pre = self._pre_next = None

    def prefetch_decoder(self, feats_nhwc, lidar2img, img_hw, tokens, pad_mult, skip=0):
        """Enqueue the FROZEN decoder's forward of the NEXT iteration(s) now, on a side stream: it depends on nothing
        this iteration trains (tools/train.py:245-252 freezes it), so it runs while the host waits for the cost
        matrix and solves the Hungarian assignment (device idle otherwise: ~0.25 ms of a 1.5 ms iteration) and
        beside the backward.  The tensors hold the next P >= 1 frames (batch dimension P: a loader's look-ahead,
        collated); the following P calls of step_fused_nhwc pick their frame up when they are handed the
        corresponding slices (``feats[6 i : 6 i + 6]``, ``lidar2img[i : i + 1]``, ``tokens[i : i + 1]``), in order.
        The inputs must stay untouched until then.  The dropout seeds are those the P iterations would draw
        themselves: the forward counter is NOT advanced here but when a frame is consumed, and a look-ahead whose
        seeds no longer match the counter (another training forward came in between) is dropped.
        skip: frames of an earlier look-ahead still to be consumed before these (their iterations draw the seeds in
        between).  step_fused_nhwc(prefetch=...) calls this at the right moment."""
        head = self.head
        if not head.training:
            head.train()
        cur = torch.cuda.current_stream()
        if getattr(self, '_pre_stream', None) is None:
            self._pre_stream = torch.cuda.Stream()
        P = int(lidar2img.shape[0])
        ncam = feats_nhwc[0].shape[0] // P
        counter0 = getattr(head, '_train_forwards', 0) + int(skip)
        seed0 = head.peek_dropout_seed(1 + int(skip))          # the seed of the training forward that takes frame 0
        pend = getattr(self, '_pre', None)
        lane = 1 - (pend['lane'] if pend is not None else getattr(self, '_lane', 0))
        self._pre_stream.wait_stream(cur)                      # inputs written on the current stream are complete
        with torch.cuda.stream(self._pre_stream):
            base = self._decoder_forward(feats_nhwc, lidar2img, img_hw, tokens, pad_mult, seed0, lane, lookahead=True)
            ev = torch.cuda.Event()
            ev.record(self._pre_stream)
        for t in base['aux'].values():
            if torch.is_tensor(t):
                t.record_stream(cur)                           # allocated on the side stream, consumed on this one
        keys = [self._input_key([f[ncam * i:ncam * (i + 1)] for f in feats_nhwc], lidar2img[i:i + 1], tokens[i:i + 1],
                                pad_mult) for i in range(P)]
        new = dict(keys=keys, seed0=seed0, counter0=counter0, base=base, ev=ev, lane=lane, next=0, count=P)
        if pend is not None and skip:
            self._pre_next = new                               # behind the frames of the look-ahead in use
        else:
            self._pre, self._pre_next = new, None

    def _take_prefetched(self, feats_nhwc, lidar2img, tokens, pad_mult):
        """-> (aux of this frame, its dropout seed) from the pending look-ahead, or None."""
        pre = getattr(self, '_pre', None)
        if pre is None and getattr(self, '_pre_next', None) is not None:
            pre = self._pre = self._pre_next
            self._pre_next = None
        if pre is None:
            return None
        head, i = self.head, pre['next']
        ok = (i < pre['count'] and getattr(head, '_train_forwards', 0) == pre['counter0'] + i and
              lidar2img.shape[0] == 1 and pre['keys'][i] == self._input_key(feats_nhwc, lidar2img, tokens, pad_mult))
        if not ok:
            self._pre = self._pre_next = None                  # stale: nothing was drawn from the seed counter for it
            return None
        if i == 0:
            torch.cuda.current_stream().wait_event(pre['ev'])
            self._lane = pre['lane']
        seed = head.next_dropout_seed()
        assert seed == (pre['seed0'] + i * self.SEED_STRIDE) & 0xFFFFFFFFFFFFFFFF
        aux = pre['base']['aux']
        out = dict(inter_states=aux['inter_states'][:, i:i + 1], inter_references=aux['inter_references'][:, i:i + 1],
                   last_box=aux['last_box'][i:i + 1])
        pre['next'] = i + 1
        if pre['next'] >= pre['count']:
            self._pre, self._pre_next = getattr(self, '_pre_next', None), None
        self.lookahead_hits = getattr(self, 'lookahead_hits', 0) + 1
        return out, seed

    def lookahead_pending(self):
        """Frames of look-ahead batches not consumed yet."""
        n = 0
        for pre in (getattr(self, '_pre', None), getattr(self, '_pre_next', None)):
            if pre is not None:
                n += pre['count'] - pre['next']
        return n

    def step_fused_nhwc(self, feats_nhwc, lidar2img, img_hw, tokens, pad_mult, gt_bboxes_list,
                        gt_labels_list, lr=None, update=True, prefetch=None):
        """update=False stops after the backward: the gradients sit in the bucket.
        prefetch: dict(feats_nhwc, lidar2img, img_hw, tokens, pad_mult) of the NEXT iteration's frame -- or, with
        ``prefetch_depth`` P > 1, of the next P frames (batch dimension P): their frozen decoder forward is enqueued
        on a side stream once this iteration's cost matrix is on its way (``prefetch_decoder``), as ONE batched
        launch sequence; the following calls pick their frame up when they are handed the same tensors (the
        slices of the batch, in order).  A callable is called as ``prefetch(skip)`` (-> the dict) only when a look-ahead
        is started: the frames that follow the `skip` frames of the look-ahead still in use (P > 1: the next batch is
        started one iteration before the current one runs out)."""
        head, lib = self.head, L.lib()
        if not head.training:                      # (Module.train() walks ~380 submodules: 0.7 ms of host time per call)
            head.train()
        if self.device_loss:
            from .device_loss import check_assign_status
            check_assign_status(head)              # a non-finite cost matrix of an earlier iteration: ValueError, as scipy
        got = self._take_prefetched(feats_nhwc, lidar2img, tokens, pad_mult)
        if got is not None:
            aux, drop_seed = got
        else:
            drop_seed = head.next_dropout_seed()   # (seed, rank, forward counter): shared with forward_train_nhwc
            aux = self._decoder_forward(feats_nhwc, lidar2img, img_hw, tokens, pad_mult, drop_seed,
                                        getattr(self, '_lane', 0))['aux']
        w = head.head_weights()
        B, T = lidar2img.shape[0], tokens.shape[1]
        hs_last = aux['inter_states'][-1].contiguous()
        ref_last = aux['inter_references'][-1].contiguous()
        last_box = aux['last_box']
        key = (B, T)
        if getattr(self, '_tape_key', None) != key:
            nbytes = lib.tc_radar_train_tape_bytes(C.byref(w), B, T)
            if nbytes == 0:
                raise L.TransCARHipError(lib.tc_last_error().decode())
            self._tape = torch.empty(nbytes, dtype=torch.uint8, device=tokens.device)
            self._tape_key = key
        tape = self._tape
        Q = head.num_query
        all_cls = torch.empty((3, B, Q, head.cls_out_channels), dtype=torch.float32, device=tokens.device)
        all_box = torch.empty((3, B, Q, head.code_size), dtype=torch.float32, device=tokens.device)
        self._wT_ready = False
        if self.chain_forward:
            # the packed (4x4x1) copy of the trainable weights for the CURRENT parameters -- and, when the backward is
            # the fused one too, its transposed packed weights in the same launch (the parameters do not change
            # between this forward and its backward): one launch instead of two
            pv = head._packed_view
            if self.chain_backward:
                bws = self._backward_workspace(lib, w, key, B, T, tokens.device)
                L.check(lib.tc_radar_train_repack(C.byref(w), C.byref(pv), bws.data_ptr(), bws.numel(), B, T,
                                                  self._stream()), 'tc_radar_train_repack')
                self._wT_ready = True
            else:
                L.check(lib.tc_head_repack_trainable_ex(C.byref(w), C.byref(pv), 1, self._stream()),
                        'tc_head_repack_trainable_ex')
            head._packed_dirty = True                  # the 16x16x4 copy (inference at >= 3 frames) is stale
            L.check(lib.tc_radar_train_fwd_fused(
                C.byref(pv), hs_last.data_ptr(), ref_last.data_ptr(), last_box.data_ptr(), tokens.data_ptr(),
                B, T, int(pad_mult), all_cls.data_ptr(), all_box.data_ptr(), tape.data_ptr(), tape.numel(),
                self.dropout, drop_seed, self._stream()), 'tc_radar_train_fwd_fused')
        else:
            L.check(lib.tc_radar_train_fwd(
                C.byref(w), hs_last.data_ptr(), ref_last.data_ptr(), last_box.data_ptr(), tokens.data_ptr(),
                B, T, int(pad_mult), all_cls.data_ptr(), all_box.data_ptr(), tape.data_ptr(), tape.numel(),
                self.dropout, drop_seed, self._stream()), 'tc_radar_train_fwd')
        raw_losses = None
        if self.device_loss:
            from .device_loss import detr_loss_device
            # (a callable is asked for the frames only when a look-ahead is really started: a loader hands over the
            # window that FOLLOWS the frame this iteration trains on)
            # a batched look-ahead (P > 1) is started ONE iteration before the frames in use run out: the batch is
            # ~1.2 ms of device work beside a 1 ms iteration, and the iteration that takes its first frame would wait
            # for all of it (measured: 1.70 instead of 0.96 ms once per batch)
            pend = self.lookahead_pending()
            start = prefetch is not None and getattr(self, '_pre_next', None) is None and \
                pend <= (1 if self.prefetch_depth > 1 else 0)
            hook = (lambda: self.prefetch_decoder(skip=pend, **(prefetch(pend) if callable(prefetch) else prefetch))) \
                if start else None
            if self.chain_backward:      # the non-finite guard of the loss gradients happens inside the backward chain
                losses, d_cls, d_box, _, raw_losses = detr_loss_device(
                    head, all_cls, all_box, gt_bboxes_list, gt_labels_list, before_sync=hook, defer_guard=True)
            else:
                losses, d_cls, d_box, _ = detr_loss_device(head, all_cls, all_box, gt_bboxes_list,
                                                           gt_labels_list, before_sync=hook)
        else:                                              # the reference's PyTorch loss + autograd
            cls_leaf = all_cls.requires_grad_(True)
            box_leaf = all_box.requires_grad_(True)
            outs = {'all_cls_scores': cls_leaf, 'all_bbox_preds': box_leaf,
                    'enc_cls_scores': None, 'enc_bbox_preds': None}
            losses = head.loss(gt_bboxes_list, gt_labels_list, outs)
            total = sum(v for k, v in losses.items() if 'loss' in k)
            total.backward()                               # d loss / d outputs only
            d_cls, d_box = cls_leaf.grad.contiguous(), box_leaf.grad.contiguous()
        self.bucket.zero_grad(check_views=getattr(self, '_gtab_key', None) is None)
        # the gradient pointers are views into the flat bucket: they do not move between iterations
        gkey = self.bucket.grads.data_ptr()
        if getattr(self, '_gtab_key', None) != gkey:
            self._gtab, self._gtab_key = grad_table(head), gkey
        g = self._gtab
        if self.chain_backward:
            bws = self._backward_workspace(lib, w, key, B, T, tokens.device)
            clean = torch.empty_like(raw_losses) if raw_losses is not None else None
            common = (C.byref(w), C.byref(g), hs_last.data_ptr(), last_box.data_ptr(), tokens.data_ptr(), B, T,
                      int(pad_mult), all_box.data_ptr(), d_cls.data_ptr(), d_box.data_ptr(), tape.data_ptr(),
                      tape.numel(), bws.data_ptr(), bws.numel(), self.dropout, drop_seed,
                      raw_losses.data_ptr() if raw_losses is not None else None,
                      clean.data_ptr() if clean is not None else None)
            chunked = self._exchange_in_chunks(update)
            flags = (1 if self._wT_ready else 0) | (2 if chunked else 0)
            common = common + (flags,)
            if self.deterministic:
                need = self.bucket.numel + 3 * B * T * 2 * head.embed_dims + 8
                if self._shadow is None or self._shadow.numel() < need or self._shadow.device != tokens.device:
                    self._shadow = torch.zeros(need, dtype=torch.int64, device=tokens.device)   # the call leaves it zero
                L.check(lib.tc_radar_train_bwd_fused_det(*common, self.bucket.grads.data_ptr(), self.bucket.numel,
                                                         self._shadow.data_ptr(), self._shadow.numel(), self._stream()),
                        'tc_radar_train_bwd_fused_det')
            else:
                L.check(lib.tc_radar_train_bwd_fused_ex(*common, self._stream()), 'tc_radar_train_bwd_fused_ex')
                if chunked:
                    # the weight gradients chunk by chunk, each chunk's all-reduce started behind its launch: it travels
                    # (RCCL's stream) while the compute stream forms the next chunk
                    chunk_pending = []
                    for gi in range(len(self.bucket.chunk_ranges)):
                        L.check(lib.tc_radar_train_bwd_weights(
                            C.byref(w), C.byref(g), hs_last.data_ptr(), tokens.data_ptr(), B, T, tape.data_ptr(),
                            tape.numel(), bws.data_ptr(), bws.numel(), gi, self._stream()), 'tc_radar_train_bwd_weights')
                        if getattr(self, 'chunk_events', None) is not None:
                            ev = torch.cuda.Event(enable_timing=True)
                            ev.record()
                            self.chunk_events.append((gi, ev))
                        chunk_pending.append(self.bucket.all_reduce_chunk_begin(gi))
            if clean is not None:
                from .device_loss import loss_dict
                losses = loss_dict(clean)
        else:
            L.check(lib.tc_radar_train_bwd(
                C.byref(w), C.byref(g), hs_last.data_ptr(), last_box.data_ptr(), tokens.data_ptr(), B, T,
                int(pad_mult), all_box.data_ptr(), d_cls.data_ptr(), d_box.data_ptr(), tape.data_ptr(),
                tape.numel(), self.dropout, drop_seed, self._stream()), 'tc_radar_train_bwd')
        self.last_dropout_seed = drop_seed
        self._sq_clean = True                  # bucket.zero_grad above cleared it and nothing has added to it since
        if getattr(self, 'keep_last', False):        # tests: the operands of the backward call
            self._last = dict(w=w, g=g, hs_last=hs_last, last_box=last_box, tokens=tokens, B=B, T=T, pad_mult=int(pad_mult),
                              all_box=all_box, d_cls=d_cls, d_box=d_box, tape=tape, seed=drop_seed)
        if update:
            # the gradient exchange starts the moment the backward is enqueued; the optimizer waits on its handle
            # (tools/train.py:253-260: DDP's buckets overlap the same way).  `exchange_events` (bench.py --train):
            # how long the compute stream stood still between the backward's end and the optimizer's first kernel
            pending = chunk_pending if (self.chain_backward and chunked) else self.bucket.all_reduce_begin()
            ev = None
            if getattr(self, 'exchange_events', None) is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            self._optimizer_step(lr, pending=pending, mark=ev)
        return {k: v.detach() for k, v in losses.items()}

    def finish(self):
        """Call at the end of an epoch, before an evaluation and before a checkpoint is written (``state_dict`` below does):
        waits for the status words of every iteration still in flight and raises the ValueError scipy raises in the
        reference (ASSIGN:117-125) if one of them met a non-finite cost matrix -- `step_fused_nhwc` polls without a
        synchronisation and would report the LAST iteration only at a next one that never comes."""
        if self.device_loss:
            from .device_loss import check_assign_status
            check_assign_status(self.head, wait=True)

    def state_dict(self):
        """Checkpoint of the head's parameters (the reference's keys) + the optimizer state; `finish()` first."""
        self.finish()
        return dict(head=self.head.state_dict(), m=self.m.clone(), v=self.v.clone(), iter=self.iter)

    def _exchange_in_chunks(self, update=True):
        """Chunked weight-gradient launches + one all-reduce per chunk: only where there is an exchange to hide (more than
        one rank), on the fused float-atomic backward, and only with the four chunks the C entry knows."""
        return bool(update and self.chunked_exchange and self.chain_backward and not self.deterministic
                    and len(self.bucket.chunk_ranges) == 4
                    and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)

    def _backward_workspace(self, lib, w, key, B, T, device):
        if getattr(self, '_bws_key', None) != key:
            nb = lib.tc_radar_train_bwd_workspace_bytes(C.byref(w), B, T)
            if nb == 0:
                raise L.TransCARHipError(lib.tc_last_error().decode())
            self._bws = torch.empty(nb, dtype=torch.uint8, device=device)
            self._bws_key = key
        return self._bws

    def _optimizer_step(self, lr=None, pending=None, mark=None):
        if mark is not None and isinstance(pending, list):
            # (bench.py --train: how long the compute stream waits for EACH chunk's all-reduce, in the order it waits)
            evs, world = [mark[0]], 1
            for work, wld in pending:
                if work is not None:
                    work.wait()
                world = wld
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                evs.append(e)
            if getattr(self, 'exchange_chunk_events', None) is not None:
                self.exchange_chunk_events.append(evs)
        else:
            world = self.bucket.all_reduce_end(pending) if pending is not None else self.bucket.all_reduce()
        if mark is not None:
            mark[1].record()
            self.exchange_events.append(mark)
        b, lib = self.bucket, L.lib()
        self.iter += 1
        if not self._sq_clean:                 # (zeroed with the gradients in the fused iteration)
            self.sq.zero_()
        self._sq_clean = False
        L.check(lib.tc_sq_norm(b.grads.data_ptr(), b.numel, self.sq.data_ptr(), self._stream()),
                'tc_sq_norm')
        L.check(lib.tc_adamw_step(
            b.params.data_ptr(), b.grads.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), b.numel,
            float(self.lr if lr is None else lr), self.betas[0], self.betas[1], self.eps,
            self.weight_decay, self.iter, 1.0 / world, float(self.max_norm or 0.0),
            self.sq.data_ptr(), self._stream()), 'tc_adamw_step')
        self.head.mark_trainable_dirty()      # re-packed lazily by the next inference forward / replay

    def step(self, mlvl_feats, img_metas, gt_bboxes_list, gt_labels_list, lr=None):
        """One iteration on this rank's frame(s); returns the loss dict (detached)."""
        self.head.train()
        outs = self.head(mlvl_feats, img_metas)
        losses = self.head.loss(gt_bboxes_list, gt_labels_list, outs)
        self.backward_and_step(losses, lr)
        return {k: v.detach() for k, v in losses.items()}

    def step_nhwc(self, feats_nhwc, lidar2img, img_hw, tokens, pad_mult, gt_bboxes_list,
                  gt_labels_list, lr=None):
        """Same with everything already resident on the device (bench)."""
        self.head.train()
        outs = self.head.forward_train_nhwc(feats_nhwc, lidar2img, img_hw, tokens, pad_mult)
        losses = self.head.loss(gt_bboxes_list, gt_labels_list, outs)
        self.backward_and_step(losses, lr)
        return {k: v.detach() for k, v in losses.items()}
