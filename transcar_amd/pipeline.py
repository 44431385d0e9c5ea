"""Frames in flight: the inference loop of the fusion decoder as N hipGraphs on N
HIP streams.

The reference's test loop (tools/test.py -> mmdet ``single_gpu_test`` /
``multi_gpu_test``) runs one frame per GPU at a time; on one stream a frame of
this decoder is a dependent chain of 13 kernels, each of which fills 225 of the
256 CUs with one workgroup per CU and ends in a tail.  Consecutive frames are
independent, so a second (third) frame on its own stream runs in the CUs,
tails and launch gaps the first leaves idle: measured on MI355X 1880 -> 2290
(2 lanes) -> 2390 (3 lanes) frames/s at one frame per step; the latency of a
single frame rises accordingly (0.53 -> 0.87 -> 1.26 ms).

Each lane owns a captured graph (head forward + box decode), a stream, static
input tensors the producer (FPN, radar pipeline) writes into, and its own head
workspace (``Detr3DHead.forward_nhwc(lane=i)``); the weights are shared.
"""
import torch

from . import ops


class FramePipeline:
    """``lanes`` frames in flight.

    static_inputs: list (one per lane) of dict(nhwc=[...], l2i, hw, tokens,
    pad_mult) -- the tensors a lane's graph reads; fill them in place before
    ``launch``.
    """

    def __init__(self, head, static_inputs, decode=True, tile_rows=None):
        if not static_inputs:
            raise ValueError('at least one lane')
        self.head, self.decode = head, decode
        self.inputs = list(static_inputs)
        self.streams = [torch.cuda.Stream() for _ in self.inputs]
        self.graphs, self.outputs = [], []
        self._next = 0
        # tile_rows (4 | 8 | 16): row-tile height of the fused chains in THIS pipeline's graphs
        # (tc_set_chain_tile_rows; None = the library's choice).  8 at one frame per lane buys
        # throughput with >= 3 lanes and costs latency.
        from . import _lib as L
        if tile_rows:
            L.check(L.lib().tc_set_chain_tile_rows(int(tile_rows)), 'tc_set_chain_tile_rows')
        try:
            self._capture()
        finally:
            if tile_rows:
                L.lib().tc_set_chain_tile_rows(0)

    def _step(self, i):
        inp = self.inputs[i]
        outs = self.head.forward_nhwc(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'],
                                      inp['pad_mult'], lane=i)
        if not self.decode:
            return outs, None
        dec = ops.box_decode_topk(outs['all_cls_scores'][-1], outs['all_bbox_preds'][-1],
                                  self.head.bbox_coder.post_center_range, self.head.bbox_coder.max_num)
        return outs, dec

    def _capture(self):
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

    @property
    def lanes(self):
        return len(self.graphs)

    def launch(self, lane=None):
        """Enqueue one frame on the next lane (round robin) and return
        (lane, (outs, decoded)): the lane's static output tensors, valid once
        ``wait(lane)`` returns and until the lane is launched again."""
        i = self._next if lane is None else lane
        self._next = (i + 1) % self.lanes
        with torch.cuda.stream(self.streams[i]):
            self.graphs[i].replay()
        return i, self.outputs[i]

    def wait(self, lane):
        self.streams[lane].synchronize()

    def synchronize(self):
        for s in self.streams:
            s.synchronize()
