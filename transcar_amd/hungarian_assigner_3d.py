"""HungarianAssigner3D, mirror of
projects/mmdet3d_plugin/core/bbox/assigners/hungarian_assigner_3d.py:16-134:
one-to-one matching of the 900 predictions to the ground-truth boxes on the
weighted sum of the classification and L1 box costs; the assignment itself is
scipy's linear_sum_assignment on the host, exactly as in the reference."""
import torch
from scipy.optimize import linear_sum_assignment

from .bbox_util import normalize_bbox
from .registry import BBOX_ASSIGNERS, MATCH_COST


class AssignResult:
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds = num_gts, gt_inds
        self.max_overlaps, self.labels = max_overlaps, labels


@BBOX_ASSIGNERS.register_module(export=True)
class HungarianAssigner3D:
    def __init__(self, cls_cost=dict(type='ClassificationCost', weight=1.),
                 reg_cost=dict(type='BBoxL1Cost', weight=1.0),
                 iou_cost=dict(type='IoUCost', weight=0.0), pc_range=None):
        self.cls_cost = MATCH_COST.build(cls_cost)
        self.reg_cost = MATCH_COST.build(reg_cost)
        self.iou_cost = MATCH_COST.build(iou_cost)
        self.pc_range = pc_range

    def assign(self, bbox_pred, cls_pred, gt_bboxes, gt_labels,
               gt_bboxes_ignore=None, eps=1e-7):
        assert gt_bboxes_ignore is None, \
            'Only case when gt_bboxes_ignore is None is supported.'
        num_gts, num_bboxes = gt_bboxes.size(0), bbox_pred.size(0)
        gt_inds = bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long)
        labels = bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long)
        if num_gts == 0 or num_bboxes == 0:
            if num_gts == 0:
                gt_inds[:] = 0
            return AssignResult(num_gts, gt_inds, None, labels=labels)
        cls_cost = self.cls_cost(cls_pred, gt_labels)
        reg_cost = self.reg_cost(bbox_pred[:, :10],
                                 normalize_bbox(gt_bboxes, self.pc_range)[:, :10])
        cost = (cls_cost + reg_cost).detach().cpu()
        rows, cols = linear_sum_assignment(cost)
        rows = torch.from_numpy(rows).to(bbox_pred.device)
        cols = torch.from_numpy(cols).to(bbox_pred.device)
        gt_inds[:] = 0
        gt_inds[rows] = cols + 1
        labels[rows] = gt_labels[cols]
        return AssignResult(num_gts, gt_inds, None, labels=labels)
