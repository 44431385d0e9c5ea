"""Frames in flight: the inference loop of the fusion decoder as N hipGraphs on N
HIP streams.

The reference's test loop (tools/test.py -> mmdet ``single_gpu_test`` /
``multi_gpu_test``) runs one frame per GPU at a time; on one stream a frame of
this decoder is a dependent chain of 13 kernels, each of which fills 225 of the
256 CUs with one workgroup per CU and ends in a tail.  Consecutive frames are
independent, so a second (third) frame on its own stream runs in the CUs,
tails and launch gaps the first leaves idle; the latency of a single frame
rises accordingly.

Each lane owns a captured graph (optional NCHW -> NHWC hand-off + head forward
+ box decode), a stream, static input tensors the producer (FPN, radar
pipeline) writes into, and its own head workspace
(``Detr3DHead.forward_nhwc(lane=i)``); the weights are shared.

Frames per launch: with ``frames_per_launch = P > 1`` a lane's graph processes P
frames in ONE launch sequence (its static inputs are batch-P tensors, frame j of
the lane is slot j).  The caller still hands over one frame at a time
(``submit``); the lane is replayed when its P slots are filled.  Two frames
share every weight fragment the row chains stream (8-row tiles: each weight
register feeds two MFMAs), which at one frame per launch is what binds them
(DESIGN.md section 5): 2 frames per launch x 3 lanes give ~1.35x the frames/s
of 1 x 3 at twice the latency.

Ordering contract (the producer works on the CURRENT stream):
  * ``launch(i)`` makes lane i's stream wait for everything enqueued so far on
    the current stream, so inputs written there (FPN kernels, H2D copies) are
    complete before the replay reads them;
  * after the replay the lane records an event; ``producer_wait(i)`` /
    ``write_inputs(i, ...)`` make the current stream wait for it before a
    lane's static inputs (or outputs) are touched again.
"""
import torch

from . import _lib as _L
from . import ops
from ._lib import TransCARHipError


def resident_frames_per_launch(num_query, device=None, tile_rows=16, workgroups_per_cu=2, num_cus=None):
    """The largest number of frames whose row tiles are all resident at once: the fused chains run
    `workgroups_per_cu` workgroups of `tile_rows` queries per CU, so P frames are ceil(P * Q / tile_rows)
    workgroups on 2 x 256 slots of an MI355X -- Q = 900: 9 frames = 507 workgroups.  One frame more (563) sends
    51 workgroups into a second round on a nearly empty chip (decoder chain 236.8 us instead of 156: 38 % of
    the f32 MFMA peak instead of 52 %, profiles/r2_*, r3_*).  A stream of K frames is launched as K // P full
    launches and one partial launch (`FramePipeline.flush`)."""
    cus = num_cus or torch.cuda.get_device_properties(
        device if device is not None else torch.cuda.current_device()).multi_processor_count
    return max(1, (cus * workgroups_per_cu * tile_rows) // int(num_query))


class FramePipeline:
    """``lanes`` frames in flight.

    static_inputs: list (one per lane) of dict(nhwc=[...], l2i, hw, tokens,
    pad_mult[, nchw=[...]]) -- the tensors a lane's graph reads.  With
    ``nchw`` (list of [B,N,C,H,W] / [BN,C,H,W] maps, what the reference's FPN
    hands over, DET:62-66) the graph starts with the one-launch transposition
    of all levels into ``nhwc``.
    tile_rows (4 | 8 | 16): row-tile height of the fused chains in THIS
    pipeline's graphs (tc_head_options.chain_tile_rows; None = automatic).
    8 at one frame per lane buys throughput with >= 3 lanes and costs latency.
    A PARTIAL launch (``flush`` with fewer filled slots) of a pipeline with more than one lane keeps the tile
    height of the full launch (``tile_rows_of``): it shares the chip with the other lanes' launches, where the
    matrix-pipe time per row counts (16x16x4 MFMA tiles), not the latency of a launch that is alone (the automatic
    rule's 8- or 4-row tiles for few rows): the driver's 20-step window (9 + 9 + 2 frames) 5 372 -> 5 490 frames/s.
    """

    def __init__(self, head, static_inputs, decode=True, tile_rows=None, options=None, streams=None,
                 partial_graphs=True, radar_raw_capacity=None):
        if not static_inputs:
            raise ValueError('at least one lane')
        from .detr3d_head import head_options
        self.head, self.decode = head, decode
        self.partial_graphs = bool(partial_graphs)
        self.inputs = list(static_inputs)
        #: frames per launch = the batch size of the lanes' static inputs
        self.frames_per_launch = int(self.inputs[0]['l2i'].shape[0])
        if any(int(i['l2i'].shape[0]) != self.frames_per_launch for i in self.inputs):
            raise ValueError('all lanes must have the same number of frame slots')
        self._filled = [0] * len(self.inputs)
        self._fill_lane = 0
        # streams: reuse the HIP streams of another (idle) pipeline.  A process has a handful of hardware
        # queues (GPU_MAX_HW_QUEUES, 4 by default) and streams are mapped onto them as they are created:
        # the lanes of a THIRD or fourth pipeline of a process can end up sharing a queue, i.e. serialised
        # (bench.py's side runs measured 3 lanes at the 2-lane rate that way).
        if streams is not None and len(streams) < len(self.inputs):
            raise ValueError('streams: need one per lane')
        self.streams = list(streams[:len(self.inputs)]) if streams is not None else \
            [torch.cuda.Stream() for _ in self.inputs]
        self.done = [torch.cuda.Event() for _ in self.inputs]
        # one lane = one launch sequence at a time: the attention launches have the chip to themselves, which is where the
        # camera pre-gather pays (+1.3 %, round 6; with three lanes it costs 2.8 %)
        solo = len(self.inputs) == 1
        self.options = options if options is not None else head_options(tile_rows=tile_rows, cam_pregather=solo)
        # radar_raw_capacity = N: every lane also owns a raw-radar stage (ops.RadarRawStage: device slabs for
        # N raw points per frame slot + descriptors) and its graph STARTS with the device-side radar ingest
        # (tc_radar_build_tokens_batch, HEAD:301-536) writing the lane's static `tokens`; the producer hands
        # over raw sweeps (`write_inputs(lane, radar_frame=..., slot=j)`: host pack + three small H2D copies)
        self.radar_stage = None
        if radar_raw_capacity:
            dev = self.inputs[0]['l2i'].device
            self.radar_stage = [ops.RadarRawStage(self.frames_per_launch, int(radar_raw_capacity), dev)
                                for _ in self.inputs]
        self.graphs, self.outputs = [], []
        #: (lane, n) -> (graph, outputs): a partly filled lane (``flush`` with n < frames_per_launch valid
        #: slots) replays a graph over its first n slots only, captured the first time that n occurs
        self._partial = {}
        self.last_flush = None
        self._last = [None] * len(self.streams)        # per lane: (valid frame slots, outputs) of its last launch
        # f16-range guard (tc_head_options.range_status, round 6 / ADVICE r5): every lane's graph ENDS with a copy of the
        # head's sticky status word into a pinned host word of the lane -- `wait(lane)` reads it without another
        # synchronisation, reports it (`range_status(lane)`) and lets the head fall back; nobody calls get_bboxes here
        self._status_host = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in self.inputs]
        self._status_seen = [0] * len(self.inputs)
        self._next = 0
        self._capture()

    def _auto_tile_rows(self, rows):
        """the library's automatic rule (chain.hip tile_rows()): 4 up to 1024 rows per launch, 8 up to 2048, 16 up to
        4096, 32 beyond (16 when the matrix path is pinned to f32)"""
        mp = int(self.options.matrix_path)
        f32 = mp == _L.TC_MATRIX_F32 or (mp == _L.TC_MATRIX_AUTO and self.head.matrix_fallback)
        if rows > 4096 and not f32:
            return 32
        return 4 if rows <= 1024 else 8 if rows <= 2048 else 16

    def tile_rows_of(self, n=None):
        """Row-tile height of a launch over n (default: all) frame slots of a lane."""
        P = self.frames_per_launch
        n = P if n is None else int(n)
        if self.options.chain_tile_rows:
            return int(self.options.chain_tile_rows)
        full = self._auto_tile_rows(P * self.head.num_query)
        if n == P or len(self.inputs) < 2:
            return self._auto_tile_rows(n * self.head.num_query)
        return full

    def _options_of(self, n):
        if n is None or n == self.frames_per_launch or self.options.chain_tile_rows:
            return self.options
        import ctypes as C
        o = type(self.options)()
        C.memmove(C.byref(o), C.byref(self.options), C.sizeof(o))
        o.chain_tile_rows = self.tile_rows_of(n)
        return o

    def _step(self, i, n=None):
        inp = self.inputs[i]
        P = self.frames_per_launch
        if n is not None and n != P:
            # the first n frame slots of the lane's static tensors (dim 0 = P or P * num_cams), as views
            def head_of(t):
                return t[:(t.shape[0] // P) * n]
            inp = dict(inp, nhwc=[head_of(t) for t in inp['nhwc']], l2i=inp['l2i'][:n], tokens=inp['tokens'][:n])
            if inp.get('nchw') is not None:
                inp['nchw'] = [head_of(t) for t in self.inputs[i]['nchw']]
        if inp.get('nchw') is not None:
            ops.to_nhwc_levels(inp['nchw'], out=inp['nhwc'])
        if self.radar_stage is not None:       # raw sweeps -> this launch's tokens (first node of the graph)
            self.radar_stage[i].build(self.inputs[i]['tokens'], n=n)
        outs = self.head.forward_nhwc(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'],
                                      inp['pad_mult'], lane=i, options=self._options_of(n))
        dec = None
        if self.decode:
            dec = ops.box_decode_topk(outs['all_cls_scores'][-1], outs['all_bbox_preds'][-1],
                                      self.head.bbox_coder.post_center_range, self.head.bbox_coder.max_num)
        sb = self.head._status_buf
        if sb is not None and sb.device == inp['l2i'].device:
            self._status_host[i].copy_(sb[:1], non_blocking=True)      # (a memcpy node of the lane's graph)
        return outs, dec

    def _capture(self):
        self.graphs, self.outputs = [], []
        self._partial = {}
        with torch.no_grad():
            for i, s in enumerate(self.streams):
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    for _ in range(3):                      # allocate workspaces, warm the allocator
                        self._step(i)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                # thread_local: a RCCL watchdog thread of torch.distributed must not break the capture
                with torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
                    out = self._step(i)
                self.graphs.append(g)
                self.outputs.append(out)
        torch.cuda.synchronize()
        # the graphs hold raw pointers into the head's packed weights and lane workspaces
        self._generation = self.head.buffers_generation

    def _capture_partial(self, lane, n):
        """Graph of lane `lane` over its first n frame slots (first use of that n: host-synchronous)."""
        self.synchronize()
        s = self.streams[lane]
        with torch.no_grad():
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):
                    self._step(lane, n)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
                out = self._step(lane, n)
        torch.cuda.synchronize()
        self._partial[(lane, n)] = (g, out)
        return g, out

    def recapture(self):
        """Capture the lanes again (after the head re-allocated its packed weights or
        workspaces: ``.to()``, a checkpoint of another size)."""
        self.synchronize()
        self._capture()

    @property
    def lanes(self):
        return len(self.graphs)

    def launch(self, lane=None, n=None):
        """Enqueue one launch sequence on the next lane (round robin) and return
        (lane, (outs, decoded)): the lane's static output tensors, valid once
        ``wait(lane)`` returns and until the lane is launched again.
        n (1 <= n < frames_per_launch): only the lane's first n frame slots (a graph of its own, with
        output tensors of batch n)."""
        if n is not None and (n == self.frames_per_launch or not self.partial_graphs):
            n = None
        if n is not None and not 1 <= n < self.frames_per_launch:
            raise TransCARHipError('launch: n=%r of %d frame slots' % (n, self.frames_per_launch))
        if self.head.buffers_generation != self._generation and \
                self.head.matrix_fallback_generation == self.head.buffers_generation:
            # the f16-range guard fired (wait() below, or a get_bboxes of the plugin path): the head is on the exact-fp32
            # kernels now, these graphs still hold the f16x2 kernels -- capture them again (host-synchronous, once)
            self.recapture()
        if self.head.buffers_generation != self._generation:
            raise TransCARHipError(
                'FramePipeline: the head re-allocated device buffers (packed weights / workspaces) '
                'after these graphs were captured; call recapture()')
        if getattr(self.head, '_packed_dirty', False):
            # an optimizer step since the last replay: the re-pack (current stream) overwrites the packed
            # buffer that EVERY lane's graph reads -- it must come after all replays still in flight
            cur = torch.cuda.current_stream()
            for ev in self.done:
                cur.wait_event(ev)
            self.head.sync_packed_weights()
        i = self._next if lane is None else lane
        self._next = (i + 1) % self.lanes
        graph, outputs = (self.graphs[i], self.outputs[i]) if n is None else \
            (self._partial.get((i, n)) or self._capture_partial(i, n))
        s = self.streams[i]
        s.wait_stream(torch.cuda.current_stream())     # the producer's writes (and a re-pack) come first
        with torch.cuda.stream(s):
            graph.replay()
            self.done[i].record(s)
        self._last[i] = (self.frames_per_launch if n is None else n, outputs)
        return i, outputs

    def results(self, lane):
        """(n, (outs, decoded)) of the lane's LAST launch, full or partial: the first n frame slots are valid (a
        partial launch -- ``flush`` -- has output tensors of batch n of its own; ``outputs[lane]`` are the FULL
        graph's tensors and keep the frames of the lane's last full launch).  Valid after ``wait(lane)``."""
        if self._last[lane] is None:
            raise TransCARHipError('lane %d has not been launched' % lane)
        return self._last[lane]

    def submit(self, write=None):
        """Hand over ONE frame: it takes the next free slot (lane, slot) of the lane being filled;
        ``write(pipe, lane, slot)`` (optional) refills that slot's static inputs (e.g.
        ``pipe.write_inputs(lane, slot=slot, ...)``); when the lane's ``frames_per_launch`` slots
        are filled the lane is replayed and the next lane starts filling.  Returns
        (lane, slot, launched)."""
        lane, slot = self._fill_lane, self._filled[self._fill_lane]
        if write is not None:
            write(self, lane, slot)
        self._filled[lane] = slot + 1
        launched = False
        if self._filled[lane] == self.frames_per_launch:
            self.launch(lane)
            self._filled[lane] = 0
            self._fill_lane = (lane + 1) % self.lanes
            launched = True
        return lane, slot, launched

    def flush(self):
        """Launch a partly filled lane: only its n filled slots run (a graph over the first n frame slots,
        captured the first time that n occurs; with ``partial_graphs=False`` the full graph is replayed and
        the unfilled slots hold stale frames whose results the caller ignores).  Returns the number of valid
        slots launched (0: nothing pending); ``last_flush`` = (lane, n, (outs, decoded)) of that launch."""
        lane = self._fill_lane
        n = self._filled[lane]
        if n:
            _, out = self.launch(lane, n=n)
            self.last_flush = (lane, n, out)
            self._filled[lane] = 0
            self._fill_lane = (lane + 1) % self.lanes
        return n

    def producer_wait(self, lane):
        """The current stream waits until lane's last replay has finished: after this, work
        enqueued on the current stream may overwrite the lane's static inputs."""
        torch.cuda.current_stream().wait_event(self.done[lane])

    def write_inputs(self, lane, nhwc=None, nchw=None, l2i=None, tokens=None, pad_mult=None, slot=None,
                     radar_frame=None):
        """Refill a lane's static inputs in place from the current stream (device tensors or
        pinned host tensors: ``copy_`` is asynchronous), ordered after the lane's previous
        replay.  tokens must have the captured shape and pad_mult (radar.pack_tokens(T=...)).
        slot=None: the whole lane (tensors of the captured batch size); slot=j: ONE frame
        (batch-1 tensors) into frame slot j of a ``frames_per_launch > 1`` lane."""
        inp = self.inputs[lane]
        P = self.frames_per_launch
        if slot is not None and not 0 <= slot < P:
            raise TransCARHipError('slot %r of %d' % (slot, P))
        self.producer_wait(lane)
        if radar_frame is not None:
            # raw sweeps of ONE frame (slot j; slot None: a list with one frame per slot) for the lane's
            # device-side ingest node
            if self.radar_stage is None:
                raise TransCARHipError('FramePipeline was built without radar_raw_capacity')
            frames = [radar_frame] if slot is not None or isinstance(radar_frame, dict) else list(radar_frame)
            if slot is None and len(frames) != P:
                raise TransCARHipError('radar_frame: %d frames for %d slots' % (len(frames), P))
            for j, fr in enumerate(frames):
                self.radar_stage[lane].put(slot if slot is not None else j, fr)

        def view(dst):          # the part of a static tensor one frame slot owns (dim 0 = P or P * num_cams)
            if slot is None:
                return dst
            n = dst.shape[0] // P
            return dst[slot * n:(slot + 1) * n]
        for key, new in (('nhwc', nhwc), ('nchw', nchw)):
            if new is None:
                continue
            if inp.get(key) is None or len(new) != len(inp[key]):
                raise TransCARHipError('lane %d has no static %r inputs of %d levels' % (lane, key, len(new)))
            for dst, src in zip(inp[key], new):
                dst = view(dst.reshape(-1, *dst.shape[-3:]))
                src = src.reshape(-1, *src.shape[-3:])
                if tuple(dst.shape) != tuple(src.shape):
                    raise TransCARHipError('%s level shape %s != captured %s' % (key, tuple(src.shape),
                                                                                 tuple(dst.shape)))
                dst.copy_(src, non_blocking=True)
        if l2i is not None:
            view(inp['l2i']).copy_(l2i.reshape(-1, *inp['l2i'].shape[1:]), non_blocking=True)
        if tokens is not None:
            dst = view(inp['tokens'])
            tokens = tokens.reshape(-1, *tokens.shape[-2:])
            if tuple(tokens.shape) != tuple(dst.shape) or \
                    (pad_mult is not None and int(pad_mult) != int(inp['pad_mult'])):
                raise TransCARHipError(
                    'tokens %s / pad_mult %s differ from the captured %s / %d: pack every frame with '
                    'radar.pack_tokens(T=%d)' % (tuple(tokens.shape), pad_mult, tuple(dst.shape),
                                                 inp['pad_mult'], inp['tokens'].shape[1]))
            dst.copy_(tokens, non_blocking=True)

    def radar_overflow(self, lane):
        """After ``wait(lane)``: True if a frame of the lane's last launch kept more radar points than the
        captured token count holds (T - 1 + the pad row): its result is the truncated frame's, use a larger T."""
        if self.radar_stage is None:
            return False
        T = int(self.inputs[lane]['tokens'].shape[1])
        from . import radar as R
        # only the slots the last launch filled: after a partial launch the others hold counts of older frames
        n = self._last[lane][0] if self._last[lane] is not None else self.frames_per_launch
        return bool(T < R.NUM_RADAR_TOKENS and int(self.radar_stage[lane].count[:n].max().item()) > T - 1)

    def wait(self, lane):
        """Block until the lane's last launch has finished.  Also reads the f16-range status the launch left in the
        lane's pinned word: non-zero = a linear step of the f16x2 path produced inf / NaN somewhere in a forward since the
        word was last cleared (the affected rows of THIS launch's outputs may be non-finite: `range_status(lane)`); on
        the automatic matrix path the head falls back to the exact-fp32 kernels and the next `launch` re-captures."""
        self.streams[lane].synchronize()
        v = int(self._status_host[lane][0])
        self._status_seen[lane] = v
        if v:
            self._status_host[lane][0] = 0
            self.head._range_overflow(pinned=int(self.options.matrix_path) == _L.TC_MATRIX_F16X2)

    def range_status(self, lane):
        """The f16-range status word as `wait(lane)` last read it (0: in range)."""
        return self._status_seen[lane]

    def synchronize(self):
        for s in self.streams:
            s.synchronize()
