"""The plugin entry -- ``Detr3DHead.forward(mlvl_feats, img_metas)`` + ``get_bboxes`` -- as cached hipGraphs.

Round 6.  The entry a reference user calls (HEAD:248-261, 1003-1023) enqueued ~20
dependent launches per call; the device spent 0.17 ms of a nine-frame call (1.30 ms) in the gaps between them
(profiles/r5_dropin_breakdown.txt) -- the one-lane FramePipeline replays the same work as ONE graph in 1.136 ms, but
that is not the reference's API.  Here the entry itself replays graphs:

  * key = what the launches' arguments are made of: the feature maps' ADDRESSES, shapes and strides (the caller's
    tensors are read in place, zero-copy), batch size, token count, the forward options, the head's buffer
    generation.  A backbone in a steady inference loop hands over the same addresses frame after frame (PyTorch's
    caching allocator returns the block it was just given back); a key seen for the SECOND time is captured, from
    then on it is replayed.  Any other call -- new addresses, aux outputs, training mode, prebuilt [n,36] radar
    rows, the operator-by-operator path -- takes the eager path, unchanged.  (HIP can patch a captured kernel node's
    arguments, hipGraphExecKernelNodeSetParams, which would lift the "same addresses" condition; PyTorch does not
    expose it and the arguments here are structs resolved on the host per launch: not built.)
  * two graphs per key, the two phases of the eager forward (tc_head_options.phase): G1 = the hand-off
    (NCHW levels only) + lidar2img H2D + decoder layers 0 .. L-3; the host packs the raw radar sweeps into the
    entry's pinned mirrors while the device runs G1; G2 = three H2D copies + the device ingest + the last two decoder
    layers (with the radar encoders riding) + the fusion chain + the box decode + ONE D2H of the range status and the
    kept-row counts.
  * results are written into one static buffer of the graph and handed out as ONE clone per call (class scores,
    boxes, decoded rows): the caller owns what it gets, as with the eager path.  ``get_bboxes`` takes the decode the
    graph already did (same kernel, same arguments) when the dict it is given is the one this call returned.

Bit-identical to the eager path (tests/test_gpu_parity.py::test_plugin_entry_graphs_*)."""
import ctypes as C
import threading

import numpy as np
import torch

from . import _lib as L
from . import ops, radar


class _Entry:
    pass


class PluginGraphs:
    MAX_ENTRIES = 4          # LRU: an entry owns static outputs, a radar stage and (NCHW input) the channels-last copy

    def __init__(self, head):
        self.head = head
        self.entries = {}    # key -> _Entry
        self.seen = {}       # key -> sightings before capture
        self.order = []
        self.lock = threading.Lock()
        self.stats = dict(replays=0, captures=0, eager=0)

    # ------------------------------------------------------------------
    def eligible(self, mlvl_feats, img_metas, aux):
        h = self.head
        if aux or h.training or not getattr(h, 'plugin_graphs', True):
            return False
        o = h.forward_options
        if o is not None and (o.unfused or o.phase != 0 or o.decoder_dropout_p > 0.0):
            return False
        if getattr(h, 'radar_ingest', 'device') != 'device':
            return False
        if not all(isinstance(m.get('radar'), dict) for m in img_metas):
            return False
        return all(f.is_cuda and f.dtype == torch.float32 for f in mlvl_feats)

    def key_of(self, mlvl_feats, img_metas, n_raw):
        h = self.head
        T = ops.radar_tokens_T(max(n_raw))
        cap = max(256, ((max(n_raw) + 255) // 256) * 256)
        o = h.forward_options
        ob = bytes(o) if o is not None else b''
        fk = tuple((f.data_ptr(), tuple(f.shape), tuple(f.stride())) for f in mlvl_feats)
        hw = tuple(img_metas[0]['img_shape'][0][:2])
        ncam = len(img_metas[0]['lidar2img'])
        return (len(img_metas), str(mlvl_feats[0].device), fk, T, cap, hw, ob, h.buffers_generation, bool(h.matrix_fallback),
                torch.cuda.current_stream().cuda_stream, ncam)

    # ------------------------------------------------------------------
    def forward(self, mlvl_feats, img_metas):
        """-> outputs dict, or None: take the eager path."""
        n_raw = [sum(int(np.asarray(m['radar']['points'][c]).shape[1]) for c in radar.RADAR_CHANNELS) for m in img_metas]
        key = self.key_of(mlvl_feats, img_metas, n_raw)
        with self.lock:
            e = self.entries.get(key)
            if e is None:
                n = self.seen.get(key, 0) + 1
                if len(self.seen) > 64:
                    self.seen.clear()
                self.seen[key] = n
                if n < 2:
                    self.stats['eager'] += 1
                    return None
                e = self._capture(key, mlvl_feats, img_metas)
                self.seen.pop(key, None)
            else:
                self.order.remove(key)
            self.order.append(key)
            return self._replay(e, img_metas)

    def _options(self, e, phase):
        o = L.tc_head_options()
        src = self.head.forward_options
        if src is not None:
            C.memmove(C.byref(o), C.byref(src), C.sizeof(o))
        o.phase = phase
        # the plugin entry runs ONE launch sequence at a time: the round-6 opt-in that pays exactly there (and costs with
        # several sequences in flight) is on -- outputs bit-identical either way
        o.cam_pregather = 1
        return o

    def _capture(self, key, mlvl_feats, img_metas):
        h = self.head
        B, dev = len(img_metas), mlvl_feats[0].device
        T, cap, hw = key[3], key[4], key[5]
        while len(self.order) >= self.MAX_ENTRIES:
            self.entries.pop(self.order.pop(0), None)
        e = _Entry()
        e.key, e.B, e.T, e.hw = key, B, T, hw
        e.serial = 0
        e.done = None
        # (the caller's tensors are NOT kept alive: a replay happens only when a call hands over tensors at exactly these
        # addresses, i.e. while they exist; holding references would keep the backbone's blocks out of the caching
        # allocator's hands and change the very addresses the key relies on)
        l2i_np = np.asarray([m['lidar2img'] for m in img_metas], dtype=np.float32)           # [B, N, 4, 4] (XFMR:382-386)
        e.l2i_host = torch.zeros(l2i_np.shape, dtype=torch.float32).pin_memory()
        e.stage = ops.RadarRawStage(B, cap, dev)
        e.status_host = torch.zeros(1 + B, dtype=torch.int32).pin_memory()
        Q, ncls, code, mx = h.num_query, h.cls_out_channels, h.code_size, h.bbox_coder.max_num
        e.Q, e.ncls, e.code = Q, ncls, code
        # ONE static result buffer: labels (int64) first, then class scores | boxes (forward_nhwc's own layout), then
        # the decoded boxes and scores
        n_lab, n_out, n_box, n_sc = B * mx * 2, 3 * B * Q * (ncls + code), B * mx * 9, B * mx
        e.sizes = (n_lab, n_out, n_box, n_sc)
        side = torch.cuda.Stream(device=dev)
        h.sync_packed_weights()
        side.wait_stream(torch.cuda.current_stream())
        status = h.status_buffer(dev)
        pool = None

        def run(phase, flat):
            # one phase of the forward on the current stream, into the views of `flat`
            views = self._views(flat, e)
            if phase == 1:
                e.nhwc = ops.to_nhwc_levels(mlvl_feats)        # zero-copy for channels_last levels
                e.l2i_dev.copy_(e.l2i_host, non_blocking=True)
                h.forward_nhwc(e.nhwc, e.l2i_dev, hw, e.tokens, e.pad_mult, options=self._options(e, 1), _out=views['out'])
            else:
                e.stage.copy_to_device(B)
                e.stage.build(e.tokens)
                h.forward_nhwc(e.nhwc, e.l2i_dev, hw, e.tokens, e.pad_mult, options=self._options(e, 2), _out=views['out'])
                ops.box_decode_kept(views['cls'][-1], views['box'][-1], h.bbox_coder.post_center_range, mx,
                                    score_threshold=h.bbox_coder.score_threshold, z_shift=True,
                                    count_out=status[1:1 + B], out=(views['boxes'], views['scores'], views['labels']))
                e.status_host.copy_(status[:1 + B], non_blocking=True)
        with torch.no_grad(), torch.cuda.stream(side):
            e.l2i_dev = torch.empty(l2i_np.shape, dtype=torch.float32, device=dev)
            e.tokens = torch.empty((B, T, radar.NUM_FEATURES), dtype=torch.float32, device=dev)
            e.pad_mult = radar.NUM_RADAR_TOKENS - T + 1
            e.l2i_host.numpy()[...] = l2i_np
            e.stage.pack_host([m['radar'] for m in img_metas])
            warm = torch.empty(n_lab + n_out + n_box + n_sc, dtype=torch.float32, device=dev)
            for _ in range(2):                        # allocate workspaces, warm the allocator
                run(1, warm)
                run(2, warm)
            side.synchronize()
            e.g1, e.g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(e.g1, stream=side, capture_error_mode='thread_local'):
                e.flat = torch.empty(n_lab + n_out + n_box + n_sc, dtype=torch.float32, device=dev)
                run(1, e.flat)
            pool = e.g1.pool()
            with torch.cuda.graph(e.g2, stream=side, pool=pool, capture_error_mode='thread_local'):
                run(2, e.flat)
            del warm
        torch.cuda.current_stream().wait_stream(side)
        self.entries[key] = e
        self.stats['captures'] += 1
        return e

    @staticmethod
    def _views(flat, e):
        n_lab, n_out, n_box, n_sc = e.sizes
        B, Q, ncls, code = e.B, e.Q, e.ncls, e.code
        lab = flat[:n_lab].view(torch.int64)
        out = flat[n_lab:n_lab + n_out]
        boxes = flat[n_lab + n_out:n_lab + n_out + n_box]
        scores = flat[n_lab + n_out + n_box:]
        return dict(out=out, cls=out[:3 * B * Q * ncls].view(3, B, Q, ncls), box=out[3 * B * Q * ncls:].view(3, B, Q, code),
                    labels=lab.view(B, -1), boxes=boxes.view(B, -1, 9), scores=scores.view(B, -1))

    def _replay(self, e, img_metas):
        h = self.head
        if e.done is not None:
            e.done.synchronize()          # the entry's pinned mirrors and static results are free again
        e.l2i_host.numpy()[...] = np.asarray([m['lidar2img'] for m in img_metas], dtype=np.float32)
        h.sync_packed_weights()           # an optimizer step since the last call: re-pack in place, in front of the replay
        e.g1.replay()
        e.stage.pack_host([m['radar'] for m in img_metas])     # the host packs while the device runs G1
        e.g2.replay()
        res = e.flat.clone()              # the caller owns its results (one copy kernel for everything)
        done = torch.cuda.Event()
        done.record()
        e.done = done
        e.serial += 1
        self.stats['replays'] += 1
        v = self._views(res, e)
        outs = {'all_cls_scores': v['cls'], 'all_bbox_preds': v['box'], 'enc_cls_scores': None, 'enc_bbox_preds': None}
        outs['_decoded'] = dict(entry=e, serial=e.serial, boxes=v['boxes'], scores=v['scores'], labels=v['labels'])
        return outs

    # ------------------------------------------------------------------
    @staticmethod
    def decoded(preds_dicts):
        """get_bboxes: the decode the graph of THIS forward already did -> (list of dict(bboxes, scores, labels), status)
        or None (the dict is not the last one its entry produced: decode again)."""
        d = preds_dicts.get('_decoded') if isinstance(preds_dicts, dict) else None
        if d is None:
            return None
        e = d['entry']
        if e.serial != d['serial'] or e.done is None:
            return None
        e.done.synchronize()
        vals = e.status_host.tolist()
        status, counts = vals[0], vals[1:]
        return [{'bboxes': d['boxes'][b, :n], 'scores': d['scores'][b, :n], 'labels': d['labels'][b, :n]}
                for b, n in enumerate(counts)], status
