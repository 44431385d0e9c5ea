"""Restatement of the un-vendored mmdet pieces the reference's TRAINING path
uses (TEST INFRASTRUCTURE; SURVEY.md Appendix B, "[3p-memory]"): FocalLoss,
L1Loss, FocalLossCost, PseudoSampler, AssignResult.  Parity for this file is
UNPINNED by the reference (mmdet source is not available, no version pinned);
the reference-owned code that calls them (HungarianAssigner3D, BBox3DL1Cost,
Detr3DHead.loss/loss_single/get_targets) runs unmodified on top of them in
tests/golden/make_golden.py.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .mmcv_bricks import MATCH_COST


def _reduce(loss, weight, reduction, avg_factor):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return loss.mean() if reduction == 'mean' else loss.sum() if reduction == 'sum' else loss
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction == 'none':
        return loss
    raise ValueError('avg_factor can not be used with reduction="sum"')


class FocalLoss(nn.Module):
    """mmdet FocalLoss(use_sigmoid=True): labels in [0, num_classes], the
    value num_classes = background."""

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25,
                 reduction='mean', loss_weight=1.0, **kw):
        super().__init__()
        assert use_sigmoid
        self.use_sigmoid = use_sigmoid
        self.gamma, self.alpha = gamma, alpha
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None,
                reduction_override=None):
        num_classes = pred.size(1)
        t = F.one_hot(target, num_classes=num_classes + 1)[:, :num_classes].type_as(pred)
        p = pred.sigmoid()
        pt = (1 - p) * t + p * (1 - t)
        fw = (self.alpha * t + (1 - self.alpha) * (1 - t)) * pt.pow(self.gamma)
        loss = F.binary_cross_entropy_with_logits(pred, t, reduction='none') * fw
        if weight is not None:
            weight = weight.view(-1, 1) if weight.dim() == 1 else weight
        return self.loss_weight * _reduce(loss, weight,
                                          reduction_override or self.reduction, avg_factor)


class L1Loss(nn.Module):
    def __init__(self, reduction='mean', loss_weight=1.0, **kw):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None,
                reduction_override=None):
        if target.numel() == 0:
            return pred.sum() * 0
        loss = torch.abs(pred - target)
        return self.loss_weight * _reduce(loss, weight,
                                          reduction_override or self.reduction, avg_factor)


class GIoULoss(nn.Module):
    def __init__(self, loss_weight=0.0, **kw):
        super().__init__()
        self.loss_weight = loss_weight


LOSS_TYPES = dict(FocalLoss=FocalLoss, L1Loss=L1Loss, GIoULoss=GIoULoss)


def build_loss(cfg):
    cfg = dict(cfg)
    return LOSS_TYPES[cfg.pop('type')](**cfg)


@MATCH_COST.register_module()
class FocalLossCost:
    def __init__(self, weight=1., alpha=0.25, gamma=2, eps=1e-12):
        self.weight, self.alpha, self.gamma, self.eps = weight, alpha, gamma, eps

    def __call__(self, cls_pred, gt_labels):
        p = cls_pred.sigmoid()
        neg = -(1 - p + self.eps).log() * (1 - self.alpha) * p.pow(self.gamma)
        pos = -(p + self.eps).log() * self.alpha * (1 - p).pow(self.gamma)
        return (pos[:, gt_labels] - neg[:, gt_labels]) * self.weight


@MATCH_COST.register_module()
class IoUCost:
    def __init__(self, iou_mode='giou', weight=1.):
        self.weight = weight


def build_match_cost(cfg):
    return MATCH_COST.build(cfg)


class AssignResult:
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds = num_gts, gt_inds
        self.max_overlaps, self.labels = max_overlaps, labels


class BaseAssigner:
    pass


class SamplingResult:
    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        if gt_bboxes.numel() == 0:
            self.pos_gt_bboxes = gt_bboxes.view(-1, gt_bboxes.shape[-1] if gt_bboxes.dim() > 1 else 4)
        else:
            self.pos_gt_bboxes = gt_bboxes[self.pos_assigned_gt_inds.long(), :]


class PseudoSampler:
    def sample(self, assign_result, bboxes, gt_bboxes, **kw):
        pos = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        neg = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        return SamplingResult(pos, neg, bboxes, gt_bboxes, assign_result)
