"""Training losses and match costs of the head (HEAD:849-917, config
CFG:95-114).  In the reference these are mmdet's ``FocalLoss`` /
``L1Loss`` / ``FocalLossCost`` (un-vendored third party) and the plugin's own
``BBox3DL1Cost`` (match_cost.py:15-26): small elementwise reductions over
[900,10] tensors that run in PyTorch in the reference as well, followed by a
host-side Hungarian match.  This module keeps
the reference's formulation for ``Detr3DHead.loss`` (drop-in semantics, torch
autograd); the device-side equivalent with closed-form gradients is
``transcar_amd/device_loss.py`` over ``csrc/loss.hip`` (row f4).
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F

from .registry import MATCH_COST


def reduce_mean(t):
    """mmdet.core.reduce_mean: all-reduce SUM / world size (HEAD:891-893, 901-902)."""
    if not (dist.is_available() and dist.is_initialized()):
        return t
    t = t.clone()
    dist.all_reduce(t.div_(dist.get_world_size()), op=dist.ReduceOp.SUM)
    return t


def sigmoid_focal_loss(pred, labels, weight=None, gamma=2.0, alpha=0.25,
                       avg_factor=None, loss_weight=1.0):
    """pred [n,C] logits; labels [n] in [0,C], C = background."""
    C = pred.size(1)
    t = F.one_hot(labels, num_classes=C + 1)[:, :C].type_as(pred)
    p = pred.sigmoid()
    pt = (1 - p) * t + p * (1 - t)
    fw = (alpha * t + (1 - alpha) * (1 - t)) * pt.pow(gamma)
    loss = F.binary_cross_entropy_with_logits(pred, t, reduction='none') * fw
    if weight is not None:
        loss = loss * (weight.view(-1, 1) if weight.dim() == 1 else weight)
    loss = loss.sum() / avg_factor if avg_factor is not None else loss.mean()
    return loss_weight * loss


def l1_loss(pred, target, weight=None, avg_factor=None, loss_weight=1.0):
    if target.numel() == 0:
        return pred.sum() * 0
    loss = torch.abs(pred - target)
    if weight is not None:
        loss = loss * weight
    loss = loss.sum() / avg_factor if avg_factor is not None else loss.mean()
    return loss_weight * loss


@MATCH_COST.register_module()
class FocalLossCost:
    def __init__(self, weight=1., alpha=0.25, gamma=2, eps=1e-12):
        self.weight, self.alpha, self.gamma, self.eps = weight, alpha, gamma, eps

    def __call__(self, cls_pred, gt_labels):
        p = cls_pred.sigmoid()
        neg = -(1 - p + self.eps).log() * (1 - self.alpha) * p.pow(self.gamma)
        pos = -(p + self.eps).log() * self.alpha * (1 - p).pow(self.gamma)
        return (pos[:, gt_labels] - neg[:, gt_labels]) * self.weight


@MATCH_COST.register_module(export=True)
class BBox3DL1Cost:
    """match_cost.py:5-26."""

    def __init__(self, weight=1.):
        self.weight = weight

    def __call__(self, bbox_pred, gt_bboxes):
        return torch.cdist(bbox_pred, gt_bboxes, p=1) * self.weight


@MATCH_COST.register_module()
class IoUCost:
    """Placeholder with weight 0 ("Fake cost", CFG:113); never evaluated."""

    def __init__(self, iou_mode='giou', weight=1.):
        self.weight = weight
