"""NMSFreeCoder, mirror of
projects/mmdet3d_plugin/core/bbox/coders/nms_free_coder.py (CODER:8-111).
``decode`` runs tc_box_decode_topk on the GPU (top-k select + denormalise +
centre-range mask in one kernel)."""
import torch

from . import ops
from .registry import BBOX_CODERS


@BBOX_CODERS.register_module(export=True)
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
        # CODER:66-84.  One host sync for the whole batch (the kept rows' indices), then plain gathers: three
        # boolean-mask selects per sample were three syncs and nine small launches each
        m = valid.bool()
        if self.score_threshold:
            m = m & (scores > self.score_threshold)
        kept = torch.nonzero(m)                                  # [n, 2] (sample, row), row-major: rows stay sorted
        counts = torch.bincount(kept[:, 0], minlength=boxes.shape[0]).tolist() if boxes.shape[0] > 1 \
            else [int(kept.shape[0])]
        flat = kept[:, 0] * boxes.shape[1] + kept[:, 1]
        b_all = boxes.reshape(-1, boxes.shape[-1]).index_select(0, flat)
        s_all = scores.reshape(-1).index_select(0, flat)
        l_all = labels.reshape(-1).index_select(0, flat).long()
        out, o = [], 0
        for n in counts:
            out.append({'bboxes': b_all[o:o + n], 'scores': s_all[o:o + n], 'labels': l_all[o:o + n]})
            o += n
        return out

    def decode_single(self, cls_scores, bbox_preds):
        return self.decode_batch(cls_scores[None], bbox_preds[None], False)[0]

    def decode(self, preds_dicts, z_shift=False):
        """Last decoder level only (CODER:103-104)."""
        return self.decode_batch(preds_dicts['all_cls_scores'][-1],
                                 preds_dicts['all_bbox_preds'][-1], z_shift)
