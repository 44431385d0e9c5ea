"""Detr3DHead.loss on the device: same numbers as
``Detr3DHead.loss`` (HEAD:742-1001) + the gradients of the summed loss with
respect to ``all_cls_scores`` / ``all_bbox_preds``, from three kernel launches
and ONE device->host copy per iteration (the Hungarian assignment itself stays
scipy's ``linear_sum_assignment`` on the host, exactly as in the reference)."""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist
from scipy.optimize import linear_sum_assignment

from . import _lib as L


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def gt_tensors(gt_bboxes_list, device):
    """mmdet3d LiDARInstance3DBoxes (gravity_center / tensor) or plain [n,9]
    gravity-centre tensors -> list of [n,9] fp32 tensors (HEAD:963-966)."""
    out = []
    for g in gt_bboxes_list:
        if hasattr(g, 'gravity_center'):
            g = torch.cat((g.gravity_center, g.tensor[:, 3:]), dim=1)
        out.append(g.to(device=device, dtype=torch.float32))
    return out


#: The Hungarian assignment on the device (tc_lsa_assign: the algorithm scipy's linear_sum_assignment implements, in
#: float64, one wavefront per problem) instead of D2H + scipy + H2D: the iteration then has no host synchronisation
#: at all.  False = the reference's route (ASSIGN:117-125), kept as the cross-check of the tests.
DEVICE_ASSIGN = True


def detr_loss_device(head, all_cls, all_box, gt_bboxes_list, gt_labels_list, before_sync=None, defer_guard=False,
                     device_assign=None):
    """-> (loss dict with the reference's keys, d_all_cls, d_all_box, assigned [Lyr,B,Q] (numpy; a device tensor with
    device_assign)).  The gradients are those of sum(losses) (mmdet ``_parse_losses``)."""
    if head.assigner is None:
        raise L.TransCARHipError('loss needs train_cfg=dict(assigner=...) at construction')
    lib = L.lib()
    dev = all_cls.device
    Lyr, B, Q, ncls = all_cls.shape
    code = all_box.shape[-1]
    all_cls = all_cls.detach().contiguous()
    all_box = all_box.detach().contiguous()
    # ground truth of the batch, padded to [B, Gmax]: built once per distinct set of GT tensors (a data loader
    # hands new tensors every iteration; the per-sample slice copies + a synchronous torch.tensor(list) cost
    # ~0.5 ms of host time per call)
    # (the cache holds the tensors, so an address cannot be re-used by another live tensor; _version catches
    # in-place edits)
    def _id(t):
        t = t.tensor if hasattr(t, 'tensor') else t
        return (int(t.data_ptr()), int(t._version), tuple(t.shape))
    key = tuple((_id(g), _id(l)) for g, l in zip(gt_bboxes_list, gt_labels_list))
    cache = getattr(head, '_gt_cache', None)
    if cache is None or cache[0] != key:
        gts = gt_tensors(gt_bboxes_list, dev)
        counts = [int(g.shape[0]) for g in gts]
        Gmax = max(max(counts), 1)
        gt9 = torch.ones((B, Gmax, 9), dtype=torch.float32, device=dev)
        lab = torch.zeros((B, Gmax), dtype=torch.int32, device=dev)
        for b, (g, l) in enumerate(zip(gts, gt_labels_list)):
            if counts[b]:
                gt9[b, :counts[b]] = g[:, :9]
                lab[b, :counts[b]] = l.to(device=dev, dtype=torch.int32)
        cnt = torch.from_numpy(np.asarray(counts, dtype=np.int32)).to(dev)
        gtn = torch.empty((B, Gmax, 10), dtype=torch.float32, device=dev)
        L.check(lib.tc_normalize_bbox(gt9.data_ptr(), B * Gmax, gtn.data_ptr(), _stream()), 'tc_normalize_bbox')
        cache = head._gt_cache = (key, counts, Gmax, lab, cnt, gtn, list(gt_bboxes_list), list(gt_labels_list))
    _, counts, Gmax, lab, cnt, gtn = cache[:6]
    a = head.assigner
    cost = torch.empty((Lyr, B, Q, Gmax), dtype=torch.float32, device=dev)
    L.check(lib.tc_match_cost(
        all_cls.data_ptr(), all_box.data_ptr(), Lyr, B, Q, ncls, code, gtn.data_ptr(), lab.data_ptr(),
        cnt.data_ptr(), Gmax, float(a.cls_cost.weight), float(a.reg_cost.weight),
        float(getattr(a.cls_cost, 'alpha', 0.25)), float(getattr(a.cls_cost, 'gamma', 2.0)),
        float(getattr(a.cls_cost, 'eps', 1e-12)), cost.data_ptr(), _stream()), 'tc_match_cost')
    if before_sync is not None:        # work the device can do while the host waits and solves the assignment
        before_sync()
    if device_assign is None:
        device_assign = DEVICE_ASSIGN
    if device_assign and Q <= 1024 and Gmax <= min(128, Q):
        return _loss_with_device_assignment(head, lib, all_cls, all_box, cost, cache, defer_guard)
    cost_h = cost.cpu().numpy()                                   # the iteration's one sync
    assigned = np.full((Lyr, B, Q), -1, dtype=np.int32)
    num_pos = np.zeros(Lyr, dtype=np.float32)
    for l in range(Lyr):
        for b in range(B):
            if counts[b]:
                rows, cols = linear_sum_assignment(cost_h[l, b, :, :counts[b]])
                assigned[l, b, rows] = cols
                num_pos[l] += len(rows)
    # normalisers, HEAD:885-902: mean over ranks of the number of positives, at least 1
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        box_avg = torch.from_numpy(num_pos).to(dev)
        dist.all_reduce(box_avg, op=dist.ReduceOp.SUM)
        box_avg = box_avg / dist.get_world_size()
        cls_avg = box_avg if head.sync_cls_avg_factor else torch.from_numpy(num_pos).to(dev)
        avg = torch.stack((cls_avg.clamp(min=1.0), box_avg.clamp(min=1.0)), dim=1).contiguous()   # [Lyr,2]
        asg = torch.from_numpy(assigned).to(dev)
        losses = torch.zeros((Lyr, 2), dtype=torch.float32, device=dev)
    else:
        # one rank: the clamp on the host, and ONE upload of [assignment | normalisers | zeroed loss accumulators]
        # (three 4-byte arrays in one staging buffer: one H2D instead of two H2D and a fill)
        one = np.maximum(num_pos, 1.0).astype(np.float32)
        na = Lyr * B * Q
        stage = np.zeros(na + 4 * Lyr, dtype=np.int32)
        stage[:na] = assigned.reshape(-1)
        stage[na:na + 2 * Lyr].view(np.float32)[:] = np.stack((one, one), axis=1).reshape(-1)
        up = torch.from_numpy(stage).to(dev)
        asg = up[:na].view(Lyr, B, Q)
        avg = up[na:na + 2 * Lyr].view(torch.float32).view(Lyr, 2)
        losses = up[na + 2 * Lyr:].view(torch.float32).view(Lyr, 2)
    d_cls = torch.empty_like(all_cls)
    d_box = torch.empty_like(all_box)
    lc, lb = head.loss_cls_cfg, head.loss_bbox_cfg
    L.check(lib.tc_detr_loss_fwd_bwd(
        all_cls.data_ptr(), all_box.data_ptr(), Lyr, B, Q, ncls, code, gtn.data_ptr(), lab.data_ptr(),
        Gmax, asg.data_ptr(), avg.data_ptr(), head.code_weights.data_ptr(),
        float(lc.get('alpha', 0.25)), float(lc.get('gamma', 2.0)), float(lc.get('loss_weight', 1.0)),
        float(lb.get('loss_weight', 1.0)), losses.data_ptr(), d_cls.data_ptr(), d_box.data_ptr(),
        _stream()), 'tc_detr_loss_fwd_bwd')
    # HEAD:915-916 zeroes a NaN loss (`loss[torch.isnan(loss)] = 0`); its gradient must not reach the flat
    # bucket / the Adam moments either: a layer whose loss is not finite contributes nothing
    if defer_guard:
        # the consumer applies the guard itself (tc_radar_train_bwd_fused(layer_losses=...): inside the backward
        # chain instead of eight elementwise launches here); the raw per-level losses travel with the gradients
        # (... and hands back the NaN-zeroed losses from the same launch: `loss_dict` of that tensor is the result)
        return None, d_cls, d_box, assigned, losses
    fin = torch.isfinite(losses)
    d_cls = torch.where(fin[:, 0].view(Lyr, 1, 1, 1), torch.nan_to_num(d_cls, nan=0.0, posinf=0.0, neginf=0.0),
                        torch.zeros_like(d_cls))
    d_box = torch.where(fin[:, 1].view(Lyr, 1, 1, 1), torch.nan_to_num(d_box, nan=0.0, posinf=0.0, neginf=0.0),
                        torch.zeros_like(d_box))
    return loss_dict(losses.masked_fill(torch.isnan(losses), 0.0)), d_cls, d_box, assigned


def _loss_with_device_assignment(head, lib, all_cls, all_box, cost, cache, defer_guard):
    """cost matrix -> tc_lsa_assign -> losses + gradients, nothing leaves the device (no host sync)."""
    dev = all_cls.device
    Lyr, B, Q, ncls = all_cls.shape
    code = all_box.shape[-1]
    _, counts, Gmax, lab, cnt, gtn = cache[:6]
    # one fill: [matched boxes per level x 2 | loss accumulators per level x 2 | status]
    z = torch.zeros(4 * Lyr + 1, dtype=torch.float32, device=dev)
    num_pos, losses, status = z[:2 * Lyr].view(Lyr, 2), z[2 * Lyr:4 * Lyr].view(Lyr, 2), z[4 * Lyr:].view(torch.int32)
    asg = torch.empty((Lyr, B, Q), dtype=torch.int32, device=dev)
    L.check(lib.tc_lsa_assign_ex(cost.data_ptr(), cnt.data_ptr(), Lyr, B, Q, Gmax, asg.data_ptr(), num_pos.data_ptr(),
                                 status.data_ptr(), losses.data_ptr(), _stream()), 'tc_lsa_assign_ex')
    # > 0: a sample had a non-finite cost.  scipy raises there (ASSIGN:117-125) and the reference stops; here the
    # output's losses are poisoned (NaN: the backward's guard then sends no gradient down) and the word travels to
    # the host without a synchronisation -- FusionTrainer raises at its next step (check_assign_status)
    head.last_assign_status = status
    post_assign_status(head, status)
    avg = num_pos
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        # normalisers, HEAD:885-902: mean over ranks of the number of positives, at least 1 -- on the device tensor
        box_avg = num_pos[:, 1].clone()
        dist.all_reduce(box_avg, op=dist.ReduceOp.SUM)
        box_avg = box_avg / dist.get_world_size()
        cls_avg = box_avg if head.sync_cls_avg_factor else num_pos[:, 0]
        avg = torch.stack((cls_avg, box_avg), dim=1).contiguous()          # (the kernel clamps at 1)
    d_cls = torch.empty_like(all_cls)
    d_box = torch.empty_like(all_box)
    lc, lb = head.loss_cls_cfg, head.loss_bbox_cfg
    L.check(lib.tc_detr_loss_fwd_bwd_counts(
        all_cls.data_ptr(), all_box.data_ptr(), Lyr, B, Q, ncls, code, gtn.data_ptr(), lab.data_ptr(),
        Gmax, asg.data_ptr(), avg.data_ptr(), head.code_weights.data_ptr(),
        float(lc.get('alpha', 0.25)), float(lc.get('gamma', 2.0)), float(lc.get('loss_weight', 1.0)),
        float(lb.get('loss_weight', 1.0)), losses.data_ptr(), d_cls.data_ptr(), d_box.data_ptr(),
        _stream()), 'tc_detr_loss_fwd_bwd_counts')
    if defer_guard:
        return None, d_cls, d_box, asg, losses
    fin = torch.isfinite(losses)
    d_cls = torch.where(fin[:, 0].view(Lyr, 1, 1, 1), torch.nan_to_num(d_cls, nan=0.0, posinf=0.0, neginf=0.0),
                        torch.zeros_like(d_cls))
    d_box = torch.where(fin[:, 1].view(Lyr, 1, 1, 1), torch.nan_to_num(d_box, nan=0.0, posinf=0.0, neginf=0.0),
                        torch.zeros_like(d_box))
    return loss_dict(losses.masked_fill(torch.isnan(losses), 0.0)), d_cls, d_box, asg


def post_assign_status(head, status):
    """Asynchronous D2H of the assignment's status word into pinned memory + an event: read by check_assign_status."""
    ring = getattr(head, '_assign_pending', None)
    if ring is None:
        ring = head._assign_pending = []
        head._assign_slots = torch.zeros(64, dtype=torch.int32).pin_memory()      # one pinned allocation, 64 words in turn
        head._assign_events = [torch.cuda.Event() for _ in range(64)]
        head._assign_next = 0
    if len(ring) >= 64:                               # nobody looked for 64 iterations: look now
        check_assign_status(head, wait=True)
    i = head._assign_next
    head._assign_next = (i + 1) % 64
    host = head._assign_slots[i:i + 1]
    host.copy_(status, non_blocking=True)
    ev = head._assign_events[i]
    ev.record()
    ring.append((host, ev))


def check_assign_status(head, wait=False):
    """Raises ValueError (as scipy.optimize.linear_sum_assignment does inside the reference's assigner, ASSIGN:117-125)
    when an iteration whose status word has arrived -- every pending one with wait=True -- met a cost matrix with
    non-finite entries.  No synchronisation unless wait."""
    ring = getattr(head, '_assign_pending', None)
    while ring:
        host, ev = ring[0]
        if wait:
            ev.synchronize()
        elif not ev.query():
            return
        ring.pop(0)
        if int(host[0]) != 0:
            ring.clear()
            raise ValueError('matrix contains invalid numeric entries (cost matrix of the Hungarian assignment; '
                             '%d sample(s): their outputs sent no gradient)' % int(host[0]))


def loss_dict(losses):
    """[Lyr, 2] (cls, bbox) per level -> the reference's keys (HEAD:990-1000: the last level unprefixed, d<i>. before)"""
    Lyr = losses.shape[0]
    out = {'loss_cls': losses[Lyr - 1, 0], 'loss_bbox': losses[Lyr - 1, 1]}
    for i in range(Lyr - 1):
        out['d%d.loss_cls' % i] = losses[i, 0]
        out['d%d.loss_bbox' % i] = losses[i, 1]
    return out
