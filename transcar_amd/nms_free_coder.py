"""NMSFreeCoder, mirror of
projects/mmdet3d_plugin/core/bbox/coders/nms_free_coder.py (CODER:8-111).
``decode`` runs tc_box_decode_kept on the GPU (top-k select + denormalise +
centre-range / threshold mask + compaction of the kept rows in one kernel)."""
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

    def decode_batch(self, cls_scores, bbox_preds, z_shift, status_buf=None):
        """[B,Q,ncls], [B,Q,code] -> list of dict(bboxes, scores, labels).
        status_buf (Detr3DHead.get_bboxes): int32 device tensor [1 + n], word 0 = the head's f16-range status; the
        counts go to the words behind it and both come back in the one D2H of a decode (``self.last_status``)."""
        if self.post_center_range is None:
            raise NotImplementedError(
                'Need to reorganize output as a batch, only support '
                'post_center_range is not None for now!')       # CODER:86-89
        # CODER:62-84 on the device: the kernel writes the kept rows compacted in score order and counts them; the
        # host reads the counts (the one sync of a decode: B ints) and slices -- no mask select, no gather
        B = cls_scores.shape[0]
        piggy = status_buf is not None and status_buf.numel() > B
        boxes, scores, labels, count = ops.box_decode_kept(
            cls_scores.contiguous(), bbox_preds.contiguous(), self.post_center_range, self.max_num,
            score_threshold=self.score_threshold, z_shift=z_shift, count_out=status_buf[1:1 + B] if piggy else None)
        if piggy:
            vals = status_buf[:1 + B].tolist()
            self.last_status, counts = vals[0], vals[1:]
        else:
            self.last_status, counts = None, count.tolist()
        return [{'bboxes': boxes[b, :n], 'scores': scores[b, :n], 'labels': labels[b, :n]}
                for b, n in enumerate(counts)]

    def decode_single(self, cls_scores, bbox_preds):
        return self.decode_batch(cls_scores[None], bbox_preds[None], False)[0]

    def decode(self, preds_dicts, z_shift=False, status_buf=None):
        """Last decoder level only (CODER:103-104)."""
        return self.decode_batch(preds_dicts['all_cls_scores'][-1],
                                 preds_dicts['all_bbox_preds'][-1], z_shift, status_buf=status_buf)
