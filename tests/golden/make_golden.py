#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the
REFERENCE's own code (imported unmodified from /root/reference through
oracle/ref_harness.py) on seeded synthetic inputs.

Run only in the authoring container:   python tests/golden/make_golden.py

Inputs and weights are regenerated from seeds by transcar_amd/synth.py and
are never stored; only small outputs / intermediates are committed
(fixtures G1-G6).  The script also asserts that the
state-dict key set of synth.py equals the reference head's own.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_harness as RH                     # noqa: E402
from transcar_amd import configs, synth                  # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)
#: smooth random fields for the end-to-end rigs (synth.make_feats docstring)
SMOOTH = (4, 6)


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print('wrote %s (%.1f KB)' % (name, os.path.getsize(path) / 1024))


def ref_head():
    head = RH.build_reference_head(configs.head_cfg())
    sd = synth.make_state_dict(seed=3)
    ref_keys = {k: tuple(v.shape) for k, v in head.state_dict().items()}
    my_keys = {k: tuple(v.shape) for k, v in sd.items()}
    assert ref_keys == my_keys, (set(ref_keys) ^ set(my_keys))
    head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()},
                         strict=True)
    head.eval()
    return head, sd


def g1_feature_sampling(ref):
    """XFMR:381-422 on tiny maps, with the edge cases of the mask logic."""
    rng = np.random.RandomState(11)
    C, Q = 8, 32
    feats = synth.make_feats('tiny', seed=12, channels=C)
    l2i = synth.make_lidar2img()
    pts = rng.uniform(0, 1, (1, Q, 3)).astype(np.float32)
    pts[0, 0] = [0.5, 0.5, 0.5]          # at the rig origin: z<=eps for all
    pts[0, 1] = [1.0, 0.5, 0.6]          # far ahead of cam 0
    pts[0, 2] = [0.0, 0.5, 0.6]          # far behind
    pts[0, 3] = [0.0, 0.0, 0.0]
    pts[0, 4] = [1.0, 1.0, 1.0]
    metas = synth.make_img_metas(1, l2i)
    _, sampled, mask = ref.XFMR.feature_sampling(
        [torch.from_numpy(f) for f in feats], torch.from_numpy(pts),
        configs.point_cloud_range, metas)
    save('g1_feature_sampling.npz', ref_points=pts,
         sampled=sampled.numpy(), mask=mask.numpy())


def g2_cross_atten(head):
    """Detr3DCrossAtten.forward (XFMR:302-378), C=256, Q=900, tiny maps."""
    rng = np.random.RandomState(21)
    feats = synth.make_feats('tiny', seed=22)
    l2i = synth.make_lidar2img()
    metas = synth.make_img_metas(1, l2i)
    Q = 900
    query = rng.standard_normal((Q, 1, 256)).astype(np.float32)
    qpos = rng.standard_normal((Q, 1, 256)).astype(np.float32)
    refp = rng.uniform(0.02, 0.98, (1, Q, 3)).astype(np.float32)
    attn = head.transformer.decoder.layers[2].attentions[1]
    out = attn(torch.from_numpy(query), None,
               [torch.from_numpy(f) for f in feats],
               query_pos=torch.from_numpy(qpos),
               reference_points=torch.from_numpy(refp), img_metas=metas)
    save('g2_cross_atten.npz', out=out.numpy()[::4])


def run_head(head, feats, l2i, frame):
    RH.RADAR_FRAME.clear()
    RH.RADAR_FRAME.update(frame)
    metas = synth.make_img_metas(1, l2i)
    cap = {}

    def hook_tokens(mod, inp):
        cap['tokens'] = inp[0].detach().clone()

    def mk_hook(i):
        def h(mod, args, kwargs):
            cap['Lq%d' % i] = args[0].shape[0]
            cap['mask%d' % i] = kwargs['attn_mask'].detach().clone()
        return h
    hs = [head.radar_feat_encoder.register_forward_pre_hook(hook_tokens)]
    for i, m in enumerate([head.rf_multihead_attn, head.rf_multihead_attn2,
                           head.rf_multihead_attn3]):
        hs.append(m.register_forward_pre_hook(mk_hook(i), with_kwargs=True))

    tcap = {}

    def hook_tr(mod, inp, out):
        tcap['hs'], tcap['init_ref'], tcap['inter_refs'] = \
            [o.detach().clone() for o in out]
    hs.append(head.transformer.register_forward_hook(hook_tr))
    outs = head([torch.from_numpy(f) for f in feats], metas)
    for h in hs:
        h.remove()
    return outs, cap, tcap


def g345_head(head, ref, shapes, tag):
    feats = synth.make_feats(shapes, seed=1, smooth=SMOOTH)
    l2i = synth.make_lidar2img()
    # pass 1: uniform radar, to learn where the decoder puts its boxes
    frame0 = synth.make_radar_frame(seed=2, n_per_radar=51)
    _, _, tcap = run_head(head, feats, l2i, frame0)
    r = tcap['inter_refs'][-1][0].numpy().astype(np.float64)
    pcr = configs.point_cloud_range
    # centres are rounded to 1 cm and STORED in the fixture: they are an
    # input of pass 2, and must not depend on anyone's decoder arithmetic
    centres = np.round(np.stack([r[:, 0] * (pcr[3] - pcr[0]) + pcr[0],
                                 r[:, 1] * (pcr[4] - pcr[1]) + pcr[1]], 1), 2)
    # pass 2: 80 % of the radar returns near predicted centres
    frame = synth.make_radar_frame(seed=2, n_per_radar=51, centres=centres)
    outs, cap, tcap = run_head(head, feats, l2i, frame)
    tokens = cap['tokens'][0].numpy()
    fill_in = int((tokens[:, 0] != 500.0).sum())
    hit_counts = []
    for i in range(3):
        m = cap['mask%d' % i].numpy()
        hit_counts.append((~m).sum(1).astype(np.int32))     # per selected row
    dec = ref.CODER.NMSFreeCoder(**{k: v for k, v in
                                    configs.pts_bbox_head['bbox_coder'].items()
                                    if k != 'type'})
    preds = dec.decode({'all_cls_scores': outs['all_cls_scores'],
                        'all_bbox_preds': outs['all_bbox_preds']})[0]
    bb = preds['bboxes'].clone()
    bb[:, 2] = bb[:, 2] - bb[:, 5] * 0.5                    # HEAD:1018
    hs = tcap['hs'].numpy()                                  # [6,Q,1,C]
    save('g5_head_%s.npz' % tag,
         all_cls_scores=outs['all_cls_scores'].numpy(),
         all_bbox_preds=outs['all_bbox_preds'].numpy(),
         inter_refs=tcap['inter_refs'].numpy(),
         init_ref=tcap['init_ref'].numpy(),
         hs_rows=hs[:, ::16, 0, :],
         hs_sum=hs.astype(np.float64).sum(axis=(1, 2, 3)),
         radar_centres=centres,
         radar_tokens=tokens[:fill_in], fill_in=fill_in,
         Lq=np.array([cap['Lq%d' % i] for i in range(3)]),
         hit_counts0=hit_counts[0], hit_counts1=hit_counts[1],
         hit_counts2=hit_counts[2],
         dec_boxes=bb.numpy(), dec_scores=preds['scores'].numpy(),
         dec_labels=preds['labels'].numpy())
    print(tag, 'fill_in', fill_in, 'Lq', [cap['Lq%d' % i] for i in range(3)])


def g4_radar_empty(head):
    """Radar ingest with one empty channel and with no radar at all."""
    feats = synth.make_feats('tiny', seed=1, smooth=SMOOTH)
    l2i = synth.make_lidar2img()
    frame = synth.make_radar_frame(seed=5, n_per_radar=[7, 0, 3, 0, 12])
    outs, cap, _ = run_head(head, feats, l2i, frame)
    tokens = cap['tokens'][0].numpy()
    fill_in = int((tokens[:, 0] != 500.0).sum())
    save('g4_radar_ragged.npz', radar_tokens=tokens[:fill_in],
         fill_in=fill_in, all_bbox_preds=outs['all_bbox_preds'].numpy(),
         all_cls_scores=outs['all_cls_scores'].numpy(),
         Lq=np.array([cap.get('Lq%d' % i, 0) for i in range(3)]))


def g7_loss(ref):
    """Detr3DHead.loss (HEAD:919-1001) incl. HungarianAssigner3D / BBox3DL1Cost of the
    reference on the G5 (tiny) head outputs and a seeded ground truth."""
    head = RH.build_reference_head(configs.head_cfg(), configs.train_cfg_pts)
    sd = synth.make_state_dict(seed=3)
    head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    g5 = np.load(os.path.join(HERE, 'g5_head_tiny.npz'))
    outs = {'all_cls_scores': torch.from_numpy(g5['all_cls_scores']),
            'all_bbox_preds': torch.from_numpy(g5['all_bbox_preds']),
            'enc_cls_scores': None, 'enc_bbox_preds': None}
    boxes, labels = synth.make_gt(seed=7, n=24)
    gt = RH.GtBoxes(torch.from_numpy(boxes))
    losses = head.loss([gt], [torch.from_numpy(labels)], outs)
    gc = torch.cat((gt.gravity_center, gt.tensor[:, 3:]), 1)
    match = [head.assigner.assign(outs['all_bbox_preds'][i, 0], outs['all_cls_scores'][i, 0],
                                  gc, torch.from_numpy(labels)).gt_inds.numpy() for i in range(3)]
    save('g7_loss.npz', gt_inds=np.stack(match),
         **{k.replace('.', '_'): float(v) for k, v in losses.items()})
    # no ground truth at all
    e = head.loss([RH.GtBoxes(torch.zeros(0, 9))], [torch.zeros(0, dtype=torch.long)], outs)
    save('g7_loss_empty.npz', **{k.replace('.', '_'): float(v) for k, v in e.items()})
    print({k: float(v) for k, v in losses.items()})


def freeze_like_train_py(head):
    """tools/train.py:245-252: the DETR3D part of the head is frozen."""
    for grp in (head.transformer, head.cls_branches, head.reg_branches, head.query_embedding):
        for p in grp.parameters():
            p.requires_grad = False


def g8_train_grads(ref, tag='tiny'):
    """One training iteration's gradients from the reference: Detr3DHead.forward (tiny
    shapes -- or, tag 'res101', the ResNet-101 FPN shapes of BASELINE.json configs[2] -- radar near the G5 centres) -> loss() -> sum of the six losses (mmdet
    `_parse_losses`) -> backward, dropout off (eval mode), frozen groups as in
    tools/train.py:245-252.  Stored per trainable parameter: [sum, sum|.|, l2] in
    float64 and the first 16 entries."""
    head = RH.build_reference_head(configs.head_cfg(), configs.train_cfg_pts)
    sd = synth.make_state_dict(seed=3)
    head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    head.eval()
    freeze_like_train_py(head)
    g5 = np.load(os.path.join(HERE, 'g5_head_%s.npz' % tag))
    feats = synth.make_feats(tag, seed=1, smooth=SMOOTH)
    l2i = synth.make_lidar2img()
    frame = synth.make_radar_frame(seed=2, n_per_radar=51, centres=g5['radar_centres'])
    boxes, labels = synth.make_gt(seed=7, n=24)
    with torch.enable_grad():
        outs, cap, _ = run_head(head, feats, l2i, frame)
        d = np.abs(outs['all_cls_scores'].detach().numpy() - g5['all_cls_scores']).max()
        assert d < 5e-4, d                      # same frame as fixture G5
        losses = head.loss([RH.GtBoxes(torch.from_numpy(boxes))], [torch.from_numpy(labels)], outs)
        total = sum(v for k, v in losses.items() if 'loss' in k)
        total.backward()
    out = {'total_loss': float(total),
           'all_cls_scores': outs['all_cls_scores'].detach().numpy(),
           'all_bbox_preds': outs['all_bbox_preds'].detach().numpy(),
           'Lq': np.array([cap['Lq%d' % i] for i in range(3)])}
    out.update({'loss__' + k.replace('.', '_'): float(v) for k, v in losses.items()})
    names = []
    for k, p in head.named_parameters():
        if not p.requires_grad:
            continue
        key = k.replace('.', '__')
        if p.grad is None:                       # attention_weights2/3, output_proj2/3
            out[key + '__none'] = np.zeros(1)
            continue
        g = p.grad.detach().double().flatten()
        out[key + '__stats'] = np.array([g.sum(), g.abs().sum(), g.norm()], np.float64)
        out[key + '__head'] = g[:16].float().numpy()
        names.append(k)
    save('g8_train_grads.npz' if tag == 'tiny' else 'g8_train_grads_%s.npz' % tag, **out)
    print('g8: total loss', float(total), len(names), 'parameters with gradients,',
          sum(p.numel() for p in head.parameters() if p.requires_grad), 'trainable scalars')


def main():
    ref = RH.load_reference()
    head, _ = ref_head()
    if len(sys.argv) > 1 and sys.argv[1] == 'vovnet':
        # round 3: BASELINE.json configs[4] (VoVNet FPN shapes 232x400 ... 29x50): the head's inference
        # fixture and one training iteration's gradients, from the reference itself
        g345_head(head, ref, 'vovnet', 'vovnet')
        g8_train_grads(ref, 'vovnet')
        return
    g1_feature_sampling(ref)
    g2_cross_atten(head)
    g345_head(head, ref, 'tiny', 'tiny')
    g345_head(head, ref, 'res101', 'res101')
    g4_radar_empty(head)
    g7_loss(ref)
    g8_train_grads(ref)
    g8_train_grads(ref, 'res101')


if __name__ == '__main__':
    main()
