"""NMSFreeCoder, mirror of
projects/mmdet3d_plugin/core/bbox/coders/nms_free_coder.py (CODER:8-111).
``decode`` runs tc_box_decode_topk on the GPU (top-k select + denormalise +
centre-range mask in one kernel)."""
import torch

from . import ops
from .registry import BBOX_CODERS


@BBOX_CODERS.register_module()
class NMSFreeCoder:
    def __init__(self, pc_range, voxel_size=None, post_center_range=None,
                 max_num=100, score_threshold=None, num_classes=10):
        self.pc_range = pc_range
        self.voxel_size = voxel_size
        self.post_center_range = post_center_range
        self.max_num = max_num
        self.score_threshold = score_threshold
        self.num_classes = num_classes

    def encode(self):
        pass

    def decode_batch(self, cls_scores, bbox_preds, z_shift):
        """[B,Q,ncls], [B,Q,code] -> list of dict(bboxes, scores, labels)."""
        if self.post_center_range is None:
            raise NotImplementedError(
                'Need to reorganize output as a batch, only support '
                'post_center_range is not None for now!')       # CODER:86-89
        boxes, scores, labels, valid = ops.box_decode_topk(
            cls_scores.contiguous(), bbox_preds.contiguous(),
            self.post_center_range, self.max_num)
        if not z_shift:          # CODER returns gravity-centre z; HEAD:1018 shifts
            boxes = boxes.clone()
            boxes[..., 2] = boxes[..., 2] + boxes[..., 5] * 0.5
        out = []
        for i in range(boxes.shape[0]):
            m = valid[i].bool()
            if self.score_threshold:
                m = m & (scores[i] > self.score_threshold)
            out.append({'bboxes': boxes[i][m], 'scores': scores[i][m],
                        'labels': labels[i][m].long()})
        return out

    def decode_single(self, cls_scores, bbox_preds):
        return self.decode_batch(cls_scores[None], bbox_preds[None], False)[0]

    def decode(self, preds_dicts, z_shift=False):
        """Last decoder level only (CODER:103-104)."""
        return self.decode_batch(preds_dicts['all_cls_scores'][-1],
                                 preds_dicts['all_bbox_preds'][-1], z_shift)
