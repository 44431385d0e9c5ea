"""Whole-head parity ON THE BENCH WORKLOAD by per-layer teacher forcing (-m gpu).

The end-to-end rigs of test_gpu_parity.py are conditioned (smooth feature fields, last layer of
every box-regression MLP scaled by 0.1) because the decoder's reference-point -> sampling ->
reference-point loop amplifies fp32 rounding by ~4x per layer on iid-noise maps: after six layers
no two fp32 implementations agree.  Here the loop is cut instead of tamed:
on the UN-conditioned inputs (``make_feats('res101', smooth=None)`` = the maps bench.py times,
``reg_out_scale=1.0`` = full xavier-scale refinements) every HIP layer is fed the ORACLE's state of
the layer before -- query features and reference points -- and must reproduce the oracle's output of
that one layer.  The kernels are the ones the timed path runs: tc_sdpa_fwd (attention core),
tc_decoder_layer_tail_fwd (the fused decoder row chain incl. camera sampling, FFN, box refinement
and the next layer's QKV projection) and tc_radar_fusion_fwd (radar encoders + the fused radar
chain), each fusion layer teacher-forced from the oracle's previous box / features as well.

Reference lines: XFMR:178-214 (decoder loop + refinement), XFMR:346-378 (cross attention),
HEAD:538-729 (three fusion layers).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import transcar_oracle as O
from transcar_amd import configs, radar as R, synth
from transcar_amd.detr3d_head import MATRIX_PATHS

pytestmark = pytest.mark.gpu

PCR = configs.point_cloud_range
HW = configs.IMG_SHAPE[:2]
# One layer on identical inputs; activations are LayerNorm outputs, |x| up to ~5.  On these
# un-conditioned inputs the fp32 REFERENCE formula itself deviates from its fp64 evaluation by up to
# 1.6e-4 per layer (max over 900 x 256 values; mean 4e-6 -- measured, and re-measured by the test):
# iid-noise maps turn the fp32 rounding of a projected pixel coordinate (1e-4 px at |u| ~ 1600)
# into an O(1e-4) change of the sampled feature.  So the test measures both implementations against
# the fp64 evaluation ("truth") and asserts that HIP is as close to it as the fp32 oracle is.
LAYER_MAX_TOL = 3e-4      # max |hip - fp64| (the fp32 oracle reaches 1.6e-4)
LAYER_MEAN_TOL = 1e-5     # mean |hip - fp64| (the fp32 oracle: 4e-6)
LAYER_TOL = 1e-4          # radar layers (no sampling: plain linear algebra on O(1) values)
REF_TOL = 3e-5            # refined reference points (sigmoid of a full-xavier-scale MLP output; the fp32 oracle: 1.2e-5)


def dev():
    return torch.device('cuda:0')


def gpu(x):
    return torch.as_tensor(x).float().contiguous().to(dev())


@pytest.fixture(autouse=True)
def _no_grad():
    with torch.no_grad():
        yield


@pytest.fixture(scope='module')
def rig():
    """Un-conditioned bench inputs + the oracle's trace of every layer."""
    import transcar_amd as T
    sd_np = synth.make_state_dict(seed=3, reg_out_scale=1.0)
    sd = O.to_torch_sd(sd_np)
    head = T.build_head(configs.head_cfg())
    head.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    head = head.to(dev()).eval()
    head.head_weights()
    feats_np = synth.make_feats('res101', seed=1, smooth=None)        # iid N(0,1): bench.py:make_inputs
    feats = [torch.from_numpy(f) for f in feats_np]
    l2i = torch.from_numpy(synth.make_lidar2img()).float()[None]
    hs, init_ref, inter_refs, _ = O.transformer(sd, feats, PCR, l2i, HW)   # hs [L,Q,1,C]
    nhwc = [head_ops().to_nhwc(gpu(f)) for f in feats_np]
    return dict(T=T, sd=sd, sd64={k: v.double() for k, v in sd.items()}, head=head, feats=feats,
                nhwc=nhwc, l2i=l2i, hs=hs, init_ref=init_ref, inter_refs=inter_refs)


def head_ops():
    from transcar_amd import ops
    return ops


def _qkv_from_oracle_state(sd, lid, x, pos):
    """q (pre-scaled for the 2^x softmax), k, v^T of decoder layer `lid` from the oracle's layer
    input, computed on the host in fp64 and rounded once: the attention core's operands without
    any HIP arithmetic in front of it."""
    name = 'transformer.decoder.layers.%d.attentions.0.attn' % lid
    W, b = sd[name + '.in_proj_weight'].double(), sd[name + '.in_proj_bias'].double()
    C = 256
    qk_in = (x + pos).double()
    q = F.linear(qk_in, W[:C], b[:C]) * (1.4426950408889634 / np.sqrt(32.0))
    k = F.linear(qk_in, W[C:2 * C], b[C:2 * C])
    v = F.linear(x.double(), W[2 * C:], b[2 * C:])
    Q = x.shape[1]
    qpad = ((Q + 15) // 16) * 16
    vt = torch.zeros((x.shape[0], C, qpad), dtype=torch.float64)
    vt[:, :, :Q] = v.permute(0, 2, 1)
    return q.float(), k.float(), vt.float()


@pytest.mark.parametrize('tile_rows,matrix', [(0, None), (8, None), (16, 'f16x2'), (16, 'f32'), (32, 'f16x2')])
def test_decoder_layers_teacher_forced_on_bench_inputs(rig, tile_rows, matrix):
    """Every tile height of the row chain (0 = the automatic choice, 4-row tiles at one frame; 8; 16 on BOTH
    matrix paths: the two-plane f16 operands on the matrix cores -- round 4, the default -- and the
    v_mfma_f32_16x16x4 path on the second weight copy; 32 rows with the activations as f16 planes in LDS -- round 5's
    headline tile height, here since round 6; the tolerances are the same for all):
    HIP layer l (attention core + fused row chain) on the oracle's layer-(l-1) state and
    reference points, l = 0..5, iid-noise ResNet-101 maps, full-scale refinement MLPs -- measured
    against the fp64 evaluation of the reference formula on the same inputs, next to the fp32
    oracle's own deviation from it.  Also: the refined reference points and the next layer's
    projected q / k / v^T the chain hands to the following attention core."""
    ops = head_ops()
    sd, sd64, head = rig['sd'], rig['sd64'], rig['head']
    qe = sd['query_embedding.weight']
    pos = qe[:, :256][None]                                   # [1,Q,C]
    l2i = gpu(rig['l2i'])
    pv = head._packed_view
    feats64 = [f.double() for f in rig['feats']]
    L = 6
    report = []
    for lid in range(L):
        x_prev = qe[:, 256:][None] if lid == 0 else rig['hs'][lid - 1].permute(1, 0, 2)   # [1,Q,C]
        ref_prev = rig['init_ref'] if lid == 0 else rig['inter_refs'][lid - 1]
        q, k, vt = _qkv_from_oracle_state(sd, lid, x_prev, pos)
        attn_o = ops.sdpa(gpu(q), gpu(k), gpu(vt), matrix_path='f16x2' if matrix == 'f16x2' else 'f32')
        nxt = pv.layers[lid + 1].self_attn.in_proj if lid + 1 < L else None
        hs, ref_out, qk_next, vt_next = ops.decoder_layer_tail(
            pv.layers[lid], nxt, rig['nhwc'], attn_o, gpu(x_prev), gpu(qe), l2i, gpu(ref_prev), PCR, HW,
            tile_rows=tile_rows, matrix_path=MATRIX_PATHS[matrix])
        # the same layer: fp32 oracle (from the rig's trace) and fp64 evaluation of the same formula
        p = 'transformer.decoder.layers.%d.' % lid
        truth = O.decoder_layer(sd64, p, x_prev.permute(1, 0, 2).double(), pos.permute(1, 0, 2).double(),
                                feats64, ref_prev.double(), PCR, rig['l2i'].double(), HW).permute(1, 0, 2)
        tmp64 = O.reg_branch(sd64, 'reg_branches.%d' % lid, truth)
        new_ref = torch.zeros_like(ref_prev.double())
        new_ref[..., :2] = tmp64[..., :2] + O.inverse_sigmoid(ref_prev.double()[..., :2])
        new_ref[..., 2:3] = tmp64[..., 4:5] + O.inverse_sigmoid(ref_prev.double()[..., 2:3])
        truth_ref = new_ref.sigmoid()
        oracle32 = rig['hs'][lid].permute(1, 0, 2)            # [1,Q,C]
        e_hip = (hs.cpu().double() - truth).abs()
        e_o32 = (oracle32.double() - truth).abs()
        e_ref_hip = (ref_out.cpu().double() - truth_ref).abs()
        e_ref_o32 = (rig['inter_refs'][lid].double() - truth_ref).abs()
        report.append('layer %d: max|hs - fp64| hip %.2e / fp32 oracle %.2e; mean %.2e / %.2e; '
                      'max|ref - fp64| %.1e / %.1e' % (lid, e_hip.max(), e_o32.max(), e_hip.mean(),
                                                       e_o32.mean(), e_ref_hip.max(), e_ref_o32.max()))
        # a query whose projected point sits within fp32 rounding of an image border / depth
        # threshold may flip its visibility mask (a discontinuity of the reference, XFMR:399-409):
        # allow <= 2 such rows per layer, everything else must hold the tolerance
        bad = (e_hip.amax(-1) > LAYER_MAX_TOL)[0]
        assert int(bad.sum()) <= 2, report[-1]
        ok = ~bad
        assert float(e_hip[0][ok].mean()) <= LAYER_MEAN_TOL, report[-1]
        # as close to the fp64 truth as the fp32 reference formula is (same order of magnitude)
        assert float(e_hip[0][ok].mean()) <= 2.0 * float(e_o32.mean()) + 1e-6, report[-1]
        assert float(e_hip[0][ok].max()) <= 2.0 * float(e_o32.max()) + 5e-5, report[-1]
        assert float(e_ref_hip[0][ok].max()) <= min(REF_TOL, 2.0 * float(e_ref_o32.max()) + 1e-6), report[-1]
        if nxt is not None:
            # the chain's projection of ITS hs for the next attention core, against the host
            # (fp64, rounded once) projection of the SAME hs
            q2, k2, vt2 = _qkv_from_oracle_state(sd, lid + 1, hs.cpu(), pos)
            got_q, got_k = qk_next[..., :256].cpu(), qk_next[..., 256:].cpu()
            assert float((got_q - q2).abs().max()) <= 2e-5, lid
            assert float((got_k - k2).abs().max()) <= 2e-5, lid
            Q = hs.shape[1]
            assert float((vt_next.cpu()[0, :, :Q] - vt2[0, :, :Q]).abs().max()) <= 2e-5, lid
    print('\n'.join(['teacher-forced decoder layers on the bench workload:'] + report))


@pytest.mark.parametrize('matrix', ['f32', 'f16x2'])
def test_attention_core_teacher_forced(rig, matrix):
    """tc_sdpa_fwd (fp32 MFMA) / tc_sdpa_fwd_f16x2 (two-plane f16 operands on the matrix cores, round 4) + out_proj
    residual on the oracle's layer-3 input vs torch MHA semantics: the same tolerance for both."""
    ops = head_ops()
    sd = rig['sd']
    qe = sd['query_embedding.weight']
    pos = qe[:, :256][None]
    x_prev = rig['hs'][2].permute(1, 0, 2)
    q, k, vt = _qkv_from_oracle_state(sd, 3, x_prev, pos)
    attn_o = ops.sdpa(gpu(q), gpu(k), gpu(vt), matrix_path=matrix).cpu()
    name = 'transformer.decoder.layers.3.attentions.0.attn'
    got = F.linear(attn_o, sd[name + '.out_proj.weight'], sd[name + '.out_proj.bias'])
    qk_in = (x_prev + pos).permute(1, 0, 2)
    want = O.multihead_attention(sd, name, qk_in, qk_in, x_prev.permute(1, 0, 2)).permute(1, 0, 2)
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=3e-5, rtol=0)


def _hit_aware(got, want, got_hits, want_hits, tol, what):
    """compare rows whose gate decision agrees; bound the number of disagreeing rows"""
    agree = (got_hits == want_hits)
    assert int((~agree).sum()) <= 2, '%s: gate decisions differ on %d queries' % (what, int((~agree).sum()))
    d = np.abs(got - want)[agree]
    assert d.max() <= tol, '%s: max|d| = %.3g' % (what, d.max())
    return agree


@pytest.mark.parametrize('tile_rows,matrix', [(0, None), (16, 'f16x2'), (16, 'f32'), (32, 'f16x2')])
def test_radar_layers_teacher_forced_on_bench_inputs(rig, tile_rows, matrix):
    """(4-row tiles, the 16-row tiles on both matrix paths: two-plane f16 and the f32 16x16x4, and the 32-row tiles.)  The fused radar chain, ONE fusion layer at a time: layer r is fed the oracle's query
    features and box of layer r-1 (hs[5] / the decoder's last box for r = 0) and must reproduce the
    oracle's class scores, boxes and hit counts of layer r; then all three layers in one launch from
    the oracle's hs[5] (the launch tc_head_forward makes)."""
    ops = head_ops()
    sd, head = rig['sd'], rig['head']
    # radar frame with 80 % of the returns near the boxes the oracle's decoder predicts (bench.py)
    refs = rig['inter_refs'][-1][0].double().numpy()
    centres = np.round(np.stack([refs[:, 0] * (PCR[3] - PCR[0]) + PCR[0],
                                 refs[:, 1] * (PCR[4] - PCR[1]) + PCR[1]], 1), 2)
    frame = synth.make_radar_frame(seed=2, centres=centres)
    f36 = O.build_radar_features(frame)
    # the oracle's own trace of the radar part, layer by layer (HEAD:538-729)
    want, dbg = O.head_forward(sd, rig['feats'], rig['l2i'], HW, f36, PCR, return_debug=True)
    want_cls = want['all_cls_scores'][:, 0].numpy()
    want_box = want['all_bbox_preds'][:, 0].numpy()
    want_hits = np.stack([h.numpy() for h in dbg['hit_counts']])
    assert want_hits.astype(bool).sum() > 300, 'the rig must exercise the gated attention'
    tok_np, pad_mult = R.pack_tokens([R.build_radar_features(frame)])
    tokens = gpu(tok_np)
    hs5 = gpu(dbg['hs'][-1])                                  # [1,Q,C]
    ref5 = gpu(dbg['inter_refs'][-1])
    tmp = gpu(dbg['tmp'])                                     # the decoder's last box, metres
    # -- all three layers in one launch (what tc_head_forward does), from the oracle's decoder state
    from transcar_amd.detr3d_head import head_options
    opt = head_options(tile_rows=tile_rows or None, matrix_path=matrix)
    cls, box, hits = ops.radar_fusion(head, hs5, ref5, tmp, tokens, pad_mult, 0, 3, options=opt)
    agree0 = _hit_aware(box[0, 0].cpu().numpy(), want_box[0], hits[0, 0].cpu().numpy(), want_hits[0],
                        LAYER_TOL, 'fusion layer 1 box')
    _hit_aware(cls[0, 0].cpu().numpy(), want_cls[0], hits[0, 0].cpu().numpy(), want_hits[0],
               LAYER_TOL, 'fusion layer 1 cls')
    # fusion layers 2 and 3 of the SAME launch: layer r gates on layer r-1's own box, so a
    # query is compared once its gate decisions agree in every layer so far (a flipped gate changes the query's
    # later inputs); at most 2 new disagreements per layer, the layer-wise tolerance accumulates
    agree_all = agree0.copy()
    for r in (1, 2):
        now = hits[r, 0].cpu().numpy() == want_hits[r]
        assert int((agree_all & ~now).sum()) <= 2, 'fusion layer %d: %d new gate disagreements' % (r + 1, int((agree_all & ~now).sum()))
        agree_all &= now
        for name, got_, want_ in (('box', box[r, 0], want_box[r]), ('cls', cls[r, 0], want_cls[r])):
            d = np.abs(got_.cpu().numpy() - want_)[agree_all]
            assert d.max() <= LAYER_TOL * (r + 1), 'one launch, fusion layer %d %s: max|d| = %.3g' % (r + 1, name, d.max())
    assert int(agree_all.sum()) >= want_hits.shape[1] - 6
    # -- one layer at a time, teacher-forced.  Layer r's query features are not an output of the
    # head; recompute them with the oracle's layer function from its own previous state.
    qf = dbg['hs'][-1].permute(1, 0, 2)                       # [Q,1,C]
    radar_feat = dbg['radar_feat']
    tokens_full, _ = O.radar_tokens_from_features(f36)
    prev_box = dbg['tmp']
    for r, (sa, sf, rmin, rmax) in enumerate((('', '', 1.0, 2.0), ('2', '_2', 1.0, 2.0), ('3', '_3', 0.5, 1.0))):
        if r == 0:
            ref = dbg['inter_refs'][-1]
            cxy = torch.stack([ref[..., 0] * (PCR[3] - PCR[0]) + PCR[0],
                               ref[..., 1] * (PCR[4] - PCR[1]) + PCR[1]], -1)
        else:
            cxy = prev_box[..., :2]
        mask = O.circle_mask(cxy, prev_box[..., 3], prev_box[..., 6], prev_box[..., 7],
                             tokens_full[:, :, :2], rmin, rmax)
        cls_r, box_r, hits_r = ops.radar_fusion(head, gpu(qf.permute(1, 0, 2)),
                                                ref5 if r == 0 else None, gpu(prev_box), tokens,
                                                pad_mult, r, 1, options=opt)
        agree = _hit_aware(box_r[r, 0].cpu().numpy(), want_box[r], hits_r[r, 0].cpu().numpy(),
                           want_hits[r], LAYER_TOL, 'fusion layer %d box (teacher-forced)' % (r + 1))
        _hit_aware(cls_r[r, 0].cpu().numpy(), want_cls[r], hits_r[r, 0].cpu().numpy(), want_hits[r],
                   LAYER_TOL, 'fusion layer %d cls (teacher-forced)' % (r + 1))
        assert torch.isnan(cls_r[(r + 1) % 3]).all()          # only the requested layer was written
        qf, _ = O.radar_layer(sd, sa, sf, qf, radar_feat, mask)
        prev_box = want['all_bbox_preds'][r]
    del agree0, agree


def test_last_level_cls_only_option(rig):
    """tc_head_options.last_level_cls_only (inference opt-in): boxes of all levels and the class
    scores of the decoded level are bit-identical to the default; the two skipped class slices are
    left untouched."""
    from transcar_amd.detr3d_head import head_options
    ops = head_ops()
    head = rig['head']
    frame = synth.make_radar_frame(seed=2)
    tok_np, pad_mult = R.pack_tokens([R.build_radar_features(frame)])
    tokens = gpu(tok_np)
    l2i = gpu(rig['l2i'])
    full = head.forward_nhwc(rig['nhwc'], l2i, HW, tokens, pad_mult)
    torch.cuda.synchronize()
    fast = head.forward_nhwc(rig['nhwc'], l2i, HW, tokens, pad_mult,
                             options=head_options(last_level_cls_only=True))
    assert torch.equal(full['all_bbox_preds'], fast['all_bbox_preds'])
    assert torch.equal(full['all_cls_scores'][2], fast['all_cls_scores'][2])
    a = head.get_bboxes(full, synth.make_img_metas(1))[0]
    b = head.get_bboxes(fast, synth.make_img_metas(1))[0]
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    del ops
