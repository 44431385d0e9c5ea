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


class FlatBucket:
    """Flat parameter / gradient storage for [(name, parameter)]."""

    def __init__(self, named_params):
        self.names = [n for n, _ in named_params]
        self.items = [p for _, p in named_params]
        if not self.items:
            raise ValueError('no trainable parameters')
        dev = self.items[0].device
        n = sum(p.numel() for p in self.items)
        self.params = torch.empty(n, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(n, dtype=torch.float32, device=dev)
        self.offsets = []
        off = 0
        for p in self.items:
            k = p.numel()
            self.params[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.params[off:off + k].view(p.shape)
            p.grad = self.grads[off:off + k].view(p.shape)
            self.offsets.append(off)
            off += k
        self.numel = n

    def zero_grad(self):
        self.grads.zero_()
        for p, off in zip(self.items, self.offsets):     # a None grad would detach the view
            if p.grad is None or p.grad.data_ptr() != self.grads.data_ptr() + 4 * off:
                p.grad = self.grads[off:off + p.numel()].view(p.shape)

    def all_reduce(self):
        """SUM over ranks, one collective for the whole bucket."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.grads, op=dist.ReduceOp.SUM)
            return dist.get_world_size()
        return 1


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


class FusionTrainer:
    """head: transcar_amd.Detr3DHead on the GPU, built with ``train_cfg``."""

    def __init__(self, head, lr=1.5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01,
                 max_norm=35.0):
        self.head = head.freeze_decoder()
        self.bucket = FlatBucket(head.trainable_parameters())
        head.refresh_weights()                      # parameter addresses moved into the bucket
        self.m = torch.zeros_like(self.bucket.params)
        self.v = torch.zeros_like(self.bucket.params)
        self.sq = torch.zeros(1, dtype=torch.float32, device=self.bucket.params.device)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.weight_decay, self.max_norm = weight_decay, max_norm
        self.iter = 0

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def backward_and_step(self, losses, lr=None):
        """losses: dict from head.loss(); every key containing 'loss' is summed
        (mmdet BaseDetector._parse_losses)."""
        total = sum(v for k, v in losses.items() if 'loss' in k)
        self.bucket.zero_grad()
        total.backward()
        world = self.bucket.all_reduce()
        b = self.bucket
        lib = L.lib()
        self.iter += 1
        self.sq.zero_()
        L.check(lib.tc_sq_norm(b.grads.data_ptr(), b.numel, self.sq.data_ptr(), self._stream()),
                'tc_sq_norm')
        L.check(lib.tc_adamw_step(
            b.params.data_ptr(), b.grads.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), b.numel,
            float(self.lr if lr is None else lr), self.betas[0], self.betas[1], self.eps,
            self.weight_decay, self.iter, 1.0 / world, float(self.max_norm or 0.0),
            self.sq.data_ptr(), self._stream()), 'tc_adamw_step')
        self.head.repack_weights()
        return total.detach()

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
