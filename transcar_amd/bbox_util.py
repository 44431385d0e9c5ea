"""Box code <-> box conversions, mirror of
projects/mmdet3d_plugin/core/bbox/util.py (UTIL:4-52).  Pure tensor
reshuffling used by the training targets; the inference decode runs in the
HIP library (tc_box_decode_topk)."""
import torch


def normalize_bbox(bboxes, pc_range=None):
    """(cx,cy,cz,w,l,h,rot[,vx,vy]) -> (cx,cy,log w,log l,cz,log h,sin,cos[,vx,vy])."""
    cx, cy, cz = bboxes[..., 0:1], bboxes[..., 1:2], bboxes[..., 2:3]
    w, l, h = bboxes[..., 3:4].log(), bboxes[..., 4:5].log(), bboxes[..., 5:6].log()
    rot = bboxes[..., 6:7]
    parts = [cx, cy, w, l, cz, h, rot.sin(), rot.cos()]
    if bboxes.size(-1) > 7:
        parts += [bboxes[..., 7:8], bboxes[..., 8:9]]
    return torch.cat(parts, dim=-1)


def denormalize_bbox(nb, pc_range=None):
    rot = torch.atan2(nb[..., 6:7], nb[..., 7:8])
    parts = [nb[..., 0:1], nb[..., 1:2], nb[..., 4:5], nb[..., 2:3].exp(),
             nb[..., 3:4].exp(), nb[..., 5:6].exp(), rot]
    if nb.size(-1) > 8:
        parts += [nb[..., 8:9], nb[..., 9:10]]
    return torch.cat(parts, dim=-1)
