"""Backward kernels and one training iteration on the MI355X (pytest -m gpu).

* every backward entry point of the C ABI against a plain PyTorch fp32
  reference of the same operator (CPU autograd) on identical inputs;
* the whole iteration (frozen decoder -> radar stack -> loss -> backward)
  against fixture G8 = the REFERENCE's own gradients, and against the oracle;
* AdamW + grad clip on the flat bucket against torch.optim.AdamW.
Tolerances are relative to the gradient's magnitude: 1e-4 per operator (fp32,
different summation order, atomics), 2e-3 end to end (as the oracle-vs-reference
check in tests/test_training.py).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import transcar_oracle as O
from test_training import check_grads_against_g8, g8_inputs, g8_name, trainable
from transcar_amd import configs, synth

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def A():
    import transcar_amd
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    transcar_amd.lib()
    from transcar_amd import autograd_ops
    return autograd_ops


def rel(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))


def leaf(arr, grad=True):
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))
    return t.requires_grad_(grad)


def on_gpu(t):
    return t.detach().clone().to(dev()).requires_grad_(t.requires_grad)


@pytest.mark.parametrize('M,K,N,act', [(900, 256, 512, 1), (900, 512, 256, 0), (255, 36, 64, 1),
                                       (255, 4, 256, 0), (900, 256, 10, 0), (1800, 256, 256, 1),
                                       (7, 128, 256, 1)])
def test_linear_backward(A, M, K, N, act):
    rng = np.random.RandomState(M + K + N)
    x, w = leaf(rng.standard_normal((M, K))), leaf(rng.standard_normal((N, K)) / np.sqrt(K))
    b = leaf(rng.standard_normal(N))
    dy = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32))
    y = F.linear(x, w, b)
    if act:
        y = F.relu(y)
    y.backward(dy)
    xg, wg, bg = on_gpu(x), on_gpu(w), on_gpu(b)
    yg = A.linear(xg, wg, bg, act)
    assert rel(yg, y) < 2e-5
    yg.backward(dy.to(dev()))
    assert rel(xg.grad, x.grad) < 1e-4
    assert rel(wg.grad, w.grad) < 1e-4
    assert rel(bg.grad, b.grad) < 1e-4


def test_linear_backward_frozen_weight_and_no_input_grad(A):
    rng = np.random.RandomState(5)
    x, w, b = leaf(rng.standard_normal((64, 36)), False), leaf(rng.standard_normal((64, 36))), leaf(rng.standard_normal(64))
    xg, wg, bg = on_gpu(x), on_gpu(w), on_gpu(b)
    A.linear(xg, wg, bg, 1).sum().backward()
    F.relu(F.linear(x, w, b)).sum().backward()
    assert xg.grad is None and rel(wg.grad, w.grad) < 1e-4 and rel(bg.grad, b.grad) < 1e-4


def test_gated_linear_residual_backward(A):
    rng = np.random.RandomState(9)
    M, C = 900, 256
    x, w, b = leaf(rng.standard_normal((M, C))), leaf(rng.standard_normal((C, C)) / 16), leaf(rng.standard_normal(C))
    res = leaf(rng.standard_normal((M, C)))
    gate = torch.from_numpy((rng.uniform(size=M) < 0.3).astype(np.int32) * rng.randint(1, 5, M).astype(np.int32))
    dy = torch.from_numpy(rng.standard_normal((M, C)).astype(np.float32))
    y = res + (gate > 0).float()[:, None] * F.linear(x, w, b)
    y.backward(dy)
    g = [on_gpu(t) for t in (x, w, b, res)]
    yg = A.gated_linear_residual(g[0], g[1], g[2], g[3], gate.to(dev()))
    assert rel(yg, y) < 2e-5
    yg.backward(dy.to(dev()))
    for a, r in zip(g, (x, w, b, res)):
        assert rel(a.grad, r.grad) < 1e-4


@pytest.mark.parametrize('M,relu,addend', [(900, False, True), (900, True, False), (255, True, False),
                                           (3, False, False)])
def test_layernorm_backward(A, M, relu, addend):
    rng = np.random.RandomState(M)
    a = leaf(rng.standard_normal((M, 256)) * 2 + 0.3)
    b = leaf(rng.standard_normal((M, 256))) if addend else None
    gam, bet = leaf(rng.standard_normal(256)), leaf(rng.standard_normal(256))
    dy = torch.from_numpy(rng.standard_normal((M, 256)).astype(np.float32))
    y = F.layer_norm(a + b if addend else a, (256,), gam, bet, 1e-5)
    if relu:
        y = F.relu(y)
    y.backward(dy)
    ag, gg, bg = on_gpu(a), on_gpu(gam), on_gpu(bet)
    b2 = on_gpu(b) if addend else None
    yg = A.add_layernorm(ag, b2, gg, bg, relu)
    assert rel(yg, y) < 2e-5
    yg.backward(dy.to(dev()))
    assert rel(ag.grad, a.grad) < 1e-4
    assert rel(gg.grad, gam.grad) < 1e-4 and rel(bg.grad, bet.grad) < 1e-4
    if addend:
        assert rel(b2.grad, b.grad) < 1e-4


def _attn_reference(qproj, kv, centre_xy, box, radar_xy, rmin, rmax, pad_mult, heads=8):
    """Masked multi-head attention on projected operands, torch CPU."""
    Bn, Q, C = qproj.shape
    T = kv.shape[1]
    hd = C // heads
    outs, hit_counts = [], []
    for b in range(Bn):
        mask = O.circle_mask(centre_xy[b:b + 1], box[b:b + 1, :, 3], box[b:b + 1, :, 6],
                             box[b:b + 1, :, 7], radar_xy[b:b + 1], rmin, rmax)      # True = masked
        q = (qproj[b] / hd ** 0.5).view(Q, heads, hd).transpose(0, 1)
        k = kv[b, :, :C].view(T, heads, hd).transpose(0, 1)
        v = kv[b, :, C:].view(T, heads, hd).transpose(0, 1)
        s = q @ k.transpose(1, 2)
        mult = torch.ones(T)
        mult[-1] = pad_mult
        s = s + torch.log(mult)[None, None]
        s = s.masked_fill(mask[None], float('-inf'))
        any_hit = (~mask).any(1)
        p = torch.softmax(s, -1)
        p = torch.where(any_hit[None, :, None], p, torch.zeros_like(p))
        outs.append((p @ v).transpose(0, 1).reshape(Q, C))
        hit_counts.append(((~mask).float() * mult[None]).sum(1))
    return torch.stack(outs), torch.stack(hit_counts)


@pytest.mark.parametrize('Bn,T,pad_mult,ld_c', [(1, 255, 1, 2), (2, 130, 1, 10), (1, 64, 1245, 2)])
def test_radar_attn_core_backward(A, Bn, T, pad_mult, ld_c):
    rng = np.random.RandomState(T)
    Q, C = 900, 256
    centres = rng.uniform(-40, 40, (Bn, Q, 2)).astype(np.float32)
    box = rng.standard_normal((Bn, Q, 10)).astype(np.float32) * 0.5
    box[..., 3] = rng.uniform(0.0, 1.8, (Bn, Q))
    if ld_c == 10:
        box[..., :2] = centres
    tokens = rng.standard_normal((Bn, T, 36)).astype(np.float32)
    idx = rng.randint(0, Q, (Bn, T))
    for b in range(Bn):                        # most radar returns sit near a query centre
        tokens[b, :, :2] = centres[b, idx[b]] + rng.uniform(-1.2, 1.2, (T, 2))
    if pad_mult > 1:
        tokens[:, -1, :] = 500.0
        centres[0, 5] = [499.5, 500.2]         # one query that does hit the pad tokens
        if ld_c == 10:
            box[0, 5, :2] = centres[0, 5]
    qp = leaf(rng.standard_normal((Bn, Q, C)))
    kv = leaf(rng.standard_normal((Bn, T, 2 * C)))
    d_out = torch.from_numpy(rng.standard_normal((Bn, Q, C)).astype(np.float32))
    cen_t, box_t, tok_t = torch.from_numpy(centres), torch.from_numpy(box), torch.from_numpy(tokens)
    ref, hits_ref = _attn_reference(qp, kv, cen_t, box_t, tok_t[..., :2].contiguous(), 1.0, 2.0, pad_mult)
    ref.backward(d_out)
    qg, kg = on_gpu(qp), on_gpu(kv)
    centre_arg = box_t.to(dev()) if ld_c == 10 else cen_t.to(dev())
    out, hits = A.radar_attn_core(qg, kg, centre_arg, ld_c, box_t.to(dev()), tok_t.to(dev()),
                                  pad_mult, 1.0, 2.0)
    agree = (hits.cpu().float() == hits_ref)
    assert float(agree.float().mean()) > 0.995          # cdist rounding at the gate edge
    assert int((hits_ref > 0).sum()) > 100
    d = (out.detach().cpu() - ref.detach()).abs().amax(-1)
    assert float(d[agree].max()) < 2e-5
    if bool(agree.all()):
        out.backward(d_out.to(dev()))
        assert rel(qg.grad, qp.grad) < 1e-4
        assert rel(kg.grad, kv.grad) < 1e-4
    else:                                               # compare on the agreeing rows only
        w = agree.float()[..., None]
        ref2, _ = _attn_reference(qp, kv, cen_t, box_t, tok_t[..., :2].contiguous(), 1.0, 2.0, pad_mult)
        qp.grad = None
        kv.grad = None
        ref2.backward(d_out * w)
        out.backward((d_out * w).to(dev()))
        assert rel(qg.grad, qp.grad) < 1e-4
        assert rel(kg.grad, kv.grad) < 1e-4


def test_box_add_ref_backward(A):
    rng = np.random.RandomState(3)
    reg, prev = leaf(rng.standard_normal((1, 900, 10))), leaf(rng.standard_normal((1, 900, 10)))
    dy = torch.from_numpy(rng.standard_normal((1, 900, 10)).astype(np.float32))
    ref = reg.clone()
    ref[..., 0:2] = ref[..., 0:2] + prev[..., 0:2]
    ref[..., 4:5] = ref[..., 4:5] + prev[..., 4:5]
    ref.backward(dy)
    rg, pg = on_gpu(reg), on_gpu(prev)
    out = A.box_add_ref(rg, pg, None)
    assert rel(out, ref) < 1e-6
    out.backward(dy.to(dev()))
    assert rel(rg.grad, reg.grad) == 0.0 and rel(pg.grad, prev.grad) == 0.0
    addref = torch.from_numpy(rng.standard_normal((1, 900, 3)).astype(np.float32))
    out = A.box_add_ref(on_gpu(reg), None, addref.to(dev()))
    exp = reg.detach().clone()
    exp[..., 0:2] += addref[..., 0:2]
    exp[..., 4] += addref[..., 2]
    assert rel(out, exp) < 1e-6


# --------------------------------------------------------------------------
# one training iteration
# --------------------------------------------------------------------------
def train_head(golden_dir):
    import transcar_amd as T
    cfg = configs.head_cfg()
    cfg['train_cfg'] = configs.train_cfg_pts
    h = T.build_head(cfg)
    h.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(3).items()})
    # deterministic training forward for the comparisons with the (dropout-free) fixtures; the
    # dropout tests set their own p
    return h.to(dev()).freeze_decoder().set_dropout(0.0)


def frame_inputs(golden_dir, tag='tiny'):
    feats, l2i, frame, boxes, labels = g8_inputs(golden_dir, tag)
    metas = synth.make_img_metas(1, l2i)
    metas[0]['radar'] = frame
    gt = torch.from_numpy(boxes).clone()
    gt[:, 2] += gt[:, 5] * 0.5
    return [torch.from_numpy(f).to(dev()) for f in feats], metas, gt.to(dev()), torch.from_numpy(labels).to(dev())


def test_train_forward_equals_eval_forward(A, golden_dir):
    h = train_head(golden_dir)
    feats, metas, _, _ = frame_inputs(golden_dir)
    with torch.no_grad():
        e = h.eval()(feats, metas)
    t = h.train()(feats, metas)
    assert t['all_cls_scores'].requires_grad and t['all_bbox_preds'].requires_grad
    assert float((t["all_cls_scores"].detach() - e['all_cls_scores']).abs().max()) < 2e-4
    assert float((t['all_bbox_preds'] - e['all_bbox_preds']).abs().max()) < 2e-4


@pytest.mark.parametrize('tag', ['tiny', 'res101', 'vovnet'])
def test_training_iteration_gradients_match_reference(A, golden_dir, tag):
    """tag res101: BASELINE.json configs[2] at its full FPN shapes (every training test
    ran at tiny shapes only); vovnet: configs[4]'s VoVNet FPN shapes (only inference was
    tested there)."""
    g8 = np.load(os.path.join(golden_dir, g8_name(tag)))
    h = train_head(golden_dir).train()
    feats, metas, gt, labels = frame_inputs(golden_dir, tag)
    outs = h(feats, metas)
    assert float((outs['all_cls_scores'].detach().cpu() - torch.from_numpy(g8['all_cls_scores'])).abs().max()) < 1e-3
    losses = h.loss([gt], [labels], outs)
    for k, v in losses.items():
        ref = float(g8['loss__' + k.replace('.', '_')])
        assert abs(float(v) - ref) < 2e-3 * max(1.0, abs(ref)), (k, float(v), ref)
    total = sum(v for k, v in losses.items() if 'loss' in k)
    total.backward()
    grads = {k: p.grad for k, p in h.named_parameters() if trainable(k)}
    assert check_grads_against_g8(grads, g8, 4e-3, 'hip') == 98
    used = dict(h.trainable_parameters())
    assert len(used) == 98 and sum(p.numel() for p in used.values()) == 2646316 - 2 * 256 * 3 \
        - 2 * (24 * 256 + 24) - 2 * (256 * 256 + 256)


def test_adamw_step_matches_torch(A):
    import ctypes as C
    from transcar_amd import _lib as L
    rng = np.random.RandomState(0)
    n = 100003
    p0 = rng.standard_normal(n).astype(np.float32)
    p_ref = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.AdamW([p_ref], lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    p = torch.from_numpy(p0.copy()).to(dev())
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for step in range(1, 4):
        g = rng.standard_normal(n).astype(np.float32) * (1.0 if step != 2 else 0.01)
        p_ref.grad = torch.from_numpy(g.copy())
        torch.nn.utils.clip_grad_norm_([p_ref], 35.0)
        opt.step()
        gd = torch.from_numpy(g * 2.0).to(dev())        # SUM over 2 ranks, averaged by grad_scale
        sq = torch.zeros(L.TC_SQ_NORM_PARTIALS, device=dev())
        L.check(L.lib().tc_sq_norm(gd.data_ptr(), n, sq.data_ptr(), s), 'tc_sq_norm')
        assert abs(float(sq.sum()) - float((g.astype(np.float64) * 2) ** 2 @ np.ones(n))) < 1e-3 * float(sq.sum())
        L.check(L.lib().tc_adamw_step(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n,
                                      2e-4, 0.9, 0.999, 1e-8, 0.01, step, 0.5, 35.0, sq.data_ptr(), s),
                'tc_adamw_step')
        assert float((p.cpu() - p_ref.detach()).abs().max()) < 2e-6


def test_trainer_step_updates_flat_bucket_and_packed_weights(A, golden_dir):
    from transcar_amd.trainer import FusionTrainer
    h = train_head(golden_dir)
    feats, metas, gt, labels = frame_inputs(golden_dir)
    tr = FusionTrainer(h, lr=1e-3, weight_decay=0.01, max_norm=35.0, dropout=0.0)
    assert tr.bucket.numel == sum(p.numel() for _, p in h.trainable_parameters())
    p0 = tr.bucket.params.clone()
    with torch.no_grad():
        before = h.eval()(feats, metas)['all_cls_scores'].clone()
    l1 = tr.step(feats, metas, [gt], [labels])
    g = tr.bucket.grads.clone()
    # every parameter's .grad still aliases the flat gradient buffer
    for (n, p), off in zip(h.trainable_parameters(), tr.bucket.offsets):
        assert p.grad.data_ptr() == tr.bucket.grads.data_ptr() + 4 * off, n
        assert p.data_ptr() == tr.bucket.params.data_ptr() + 4 * off, n
    # first AdamW step: p*(1-lr*wd) - lr * g_clipped/(|g_clipped| + eps)
    norm = float(g.double().norm())
    coef = min(1.0, 35.0 / (norm + 1e-6))
    gc = g * coef
    expect = p0 * (1 - 1e-3 * 0.01) - 1e-3 * gc / (gc.abs() + 1e-8)
    assert float((tr.bucket.params - expect).abs().max()) < 2e-6
    # the fused eval forward sees the updated weights and agrees with the train forward
    with torch.no_grad():
        after = h.eval()(feats, metas)
    assert float((after['all_cls_scores'] - before).abs().max()) > 1e-3
    t = h.train()(feats, metas)
    assert float((t['all_cls_scores'].detach() - after['all_cls_scores']).abs().max()) < 5e-4
    # a few more steps on the same frame reduce the loss
    first = float(sum(l1.values()))
    for _ in range(5):
        last = tr.step(feats, metas, [gt], [labels])
    assert float(sum(last.values())) < first


@pytest.mark.parametrize('tag', ['tiny', 'res101', 'vovnet'])
def test_fused_training_path_gradients_match_reference(A, golden_dir, tag):
    """tc_radar_train_fwd / _bwd (the trainable stack as two C calls) against fixture G8
    (tiny, ResNet-101 and VoVNet FPN shapes) and against the per-operator autograd path on the same frame."""
    from transcar_amd import ops
    from transcar_amd.trainer import FusionTrainer
    g8 = np.load(os.path.join(golden_dir, g8_name(tag)))
    h = train_head(golden_dir)
    feats, metas, gt, labels = frame_inputs(golden_dir, tag)
    nhwc = [ops.to_nhwc(f) for f in feats]
    l2i = ops.lidar2img_tensor(metas, dev())
    img_hw = metas[0]['img_shape'][0][:2]
    tokens, pad_mult = h.radar_tokens(metas, dev())
    tr = FusionTrainer(h, dropout=0.0)
    losses = tr.step_fused_nhwc(nhwc, l2i, img_hw, tokens, pad_mult, [gt], [labels], update=False)
    for k, v in losses.items():
        ref = float(g8['loss__' + k.replace('.', '_')])
        assert abs(float(v) - ref) < 2e-3 * max(1.0, abs(ref)), (k, float(v), ref)
    grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in h.named_parameters() if trainable(k)}
    used = {n for n, _ in h.trainable_parameters()}
    for k in grads:                       # unused tensors are not in the bucket
        if k not in used:
            grads[k] = None
    assert check_grads_against_g8(grads, g8, 4e-3, 'fused') == 98
    # same numbers as the autograd path
    fused = tr.bucket.grads.clone()
    tr.bucket.zero_grad()
    outs = h.train()(feats, metas)
    total = sum(v for k, v in h.loss([gt], [labels], outs).items() if 'loss' in k)
    total.backward()
    assert_same_gradients_up_to_one_relu_tie(h, tr, fused, tr.bucket.grads)


def assert_params_equal_up_to_adam_noise(p_a, p_b, p_again, lr, steps, slack):
    """Parameters after `steps` AdamW iterations of two runs that differ only in the SCHEDULE (p_a, p_b), with a repeat
    of the plain run (p_again) as the yardstick.  AdamW's step is lr * m / (sqrt(v) + eps): a gradient element that is
    rounding noise of the backward's float atomics (+-1e-10) takes a step of +-lr either way, so any two runs differ by
    up to 2 lr per step in a FEW elements however equal their gradients are.  Required: all but 1e-4 of the elements
    within 3 x the repeat's own deviation (+ slack), and none beyond what sign flips can do."""
    scale = float(p_a.abs().max())
    noise = float((p_again - p_a).abs().max()) / scale
    # the yardstick itself is capped by a FIXED number (a bound that only scales with the code's own
    # noise cannot catch a change that raises the noise): two plain runs may differ by what AdamW sign flips of
    # rounding-size gradient elements can do over `steps` iterations, and by nothing more
    assert noise <= 2.5 * lr * steps / scale + 1e-6, (noise, lr, steps, scale)
    d = (p_b - p_a).abs() / scale
    bound = 3.0 * noise + slack
    frac = float((d > bound).float().mean())
    assert frac <= 1e-4, (frac, float(d.max()), noise)
    assert float(d.max()) <= 2.5 * lr * steps / scale + bound, (float(d.max()), noise, lr, steps)


def assert_same_gradients_up_to_one_relu_tie(h, tr, fused, auto, tol=1e-4):
    """(flat gradient buffers of the trainer's bucket: see assert_grad_dicts_equal_up_to_one_relu_tie)"""
    d = float((auto - fused).abs().max() / fused.abs().max())
    if d < tol:
        return
    assert d < 4e-3, d
    got, want = {}, {}
    for (n, p), off in zip(h.trainable_parameters(), tr.bucket.offsets):
        got[n], want[n] = (b[off:off + p.numel()].view_as(p) for b in (auto, fused))
    assert_grad_dicts_equal_up_to_one_relu_tie(got, want, tol, per_tensor=False)


# Why 3 and not 1.  A tie is a LayerNorm output within an ulp of zero in front of a ReLU, which the
# two paths' summation orders put on different sides.  One iteration evaluates 900 queries x 3 layers x 512 such values
# (final_cls*.{1,4}) = 1.4e6 unit-variance numbers, density 0.4 at zero; the two paths' LayerNorm outputs differ by a
# few 1e-7 (measured: MFMA chain order, split reductions), so the expected number of values that land on different
# sides is 1.4e6 x 0.4 x ~5e-7 ~ 0.3 per comparison -- Poisson: P(>= 2) ~ 4 %, P(>= 4) < 3e-4.  A cap of 1 would
# fail a correct build in ~4 % of runs by this estimate (round 5 raised it for that reason); 3 keeps a false alarm below 1e-3.  The assertion is on
# RANK, not on size: singular value number MAX_TIES of the difference must vanish, i.e. at most three ROWS may differ,
# each by its rank-one term; a real defect touches every row (rank >> 3) and still fails, and everything must agree
# to 4e-3 regardless.
MAX_TIES = 3


def assert_grad_dicts_equal_up_to_one_relu_tie(got, want, tol, per_tensor=True):
    """Two paths that sum in different orders can put a LayerNorm output within an ulp of zero on different sides of
    the ReLU behind it (final_cls*.3 -> LayerNorm -> ReLU on these fixtures).  ONE query row then contributes
    differently, which is a rank-one term in every query-side weight gradient (every operator between the radar
    attention and the losses is row-wise).  Allowed: every tensor within `tol` (relative to its own maximum when
    `per_tensor`) -- or, if some are not, at most one such row: with the rank-one term removed the query-side weight
    gradients agree to `tol`, and everything agrees to 4e-3."""
    worst, ties = 0.0, 0
    for n, w in want.items():
        scale = max(float(w.abs().max()), 1e-6)
        worst = max(worst, float((got[n] - w).abs().max()) / scale)
    if per_tensor and worst <= tol:
        return
    for n, w in want.items():
        scale = max(float(w.abs().max()), 1e-6)
        d = float((got[n] - w).abs().max())
        assert d <= 4e-3 * scale + 1e-7, (n, d, scale)
        if w.dim() != 2 or not n.startswith(('final_cls', 'final_reg', 'rf_linear', 'rf_multihead_attn')) or 'in_proj' in n:
            continue
        # (a handful of rows can sit on a tie -- 900 queries x 3 layers x 768 LayerNorm / ReLU outputs per iteration: up to
        # MAX_TIES rank-one terms, nothing of rank MAX_TIES + 1)
        sv = torch.linalg.svdvals((got[n] - w).double())
        assert float(sv[MAX_TIES]) < tol * scale, (n, [float(x) for x in sv[:MAX_TIES + 2]], scale)
        ties += float(sv[0]) >= tol * scale
    assert ties > 0, ('gradients differ by %g (relative) without a rank-one explanation' % worst)


def test_fused_training_batch_of_two_matches_autograd_path(A, golden_dir):
    """Two different frames per iteration (the reference hard-codes one in its radar part): the
    two-call stack + device loss against the per-operator autograd path + PyTorch loss."""
    from transcar_amd import ops
    from transcar_amd.trainer import FusionTrainer
    h = train_head(golden_dir)
    feats, metas, gt, labels = frame_inputs(golden_dir)
    feats2_np = synth.make_feats('tiny', seed=9, smooth=(4, 6))
    feats_b = [torch.cat([f, torch.from_numpy(g_).to(dev())], 0) for f, g_ in zip(feats, feats2_np)]
    g5 = np.load(os.path.join(golden_dir, 'g5_head_tiny.npz'))
    frame2 = synth.make_radar_frame(seed=5, n_per_radar=33, centres=g5['radar_centres'])
    metas_b = synth.make_img_metas(2, synth.make_lidar2img(), radar=[metas[0]['radar'], frame2])
    boxes2, labels2 = synth.make_gt(seed=8, n=11)
    gt2 = torch.from_numpy(boxes2).clone()
    gt2[:, 2] += gt2[:, 5] * 0.5
    gts, lbs = [gt, gt2.to(dev())], [labels, torch.from_numpy(labels2).to(dev())]
    nhwc = [ops.to_nhwc(f) for f in feats_b]
    l2i = ops.lidar2img_tensor(metas_b, dev())
    img_hw = metas_b[0]['img_shape'][0][:2]
    tokens, pad_mult = h.radar_tokens(metas_b, dev())
    tr = FusionTrainer(h, dropout=0.0)
    l_fused = tr.step_fused_nhwc(nhwc, l2i, img_hw, tokens, pad_mult, gts, lbs, update=False)
    fused = tr.bucket.grads.clone()
    tr.bucket.zero_grad()
    outs = h.train()(feats_b, metas_b)
    l_auto = h.loss(gts, lbs, outs)
    sum(v for k, v in l_auto.items() if 'loss' in k).backward()
    for k in l_auto:
        assert abs(float(l_auto[k]) - float(l_fused[k])) < 1e-4 * max(1.0, abs(float(l_auto[k]))), k
    d = (tr.bucket.grads - fused).abs().max() / fused.abs().max()
    assert float(d) < 2e-4, float(d)


@pytest.mark.parametrize('device_assign', [True, False])
@pytest.mark.parametrize('n_gt', [24, 0])
def test_device_loss_matches_reference_loss_and_autograd(A, golden_dir, n_gt, device_assign):
    """tc_match_cost / tc_detr_loss_fwd_bwd against the reference's loss values (G7), its
    Hungarian assignment, and torch autograd through Detr3DHead.loss for the gradients -- with the assignment solved
    on the device (tc_lsa_assign, round 4) and on the host by scipy (the reference's route)."""
    from transcar_amd.device_loss import detr_loss_device
    g5 = np.load(os.path.join(golden_dir, 'g5_head_tiny.npz'))
    g7 = np.load(os.path.join(golden_dir, 'g7_loss.npz' if n_gt else 'g7_loss_empty.npz'))
    h = train_head(golden_dir)
    boxes, labels = synth.make_gt(seed=7, n=24)
    gt = torch.from_numpy(boxes[:n_gt]).clone()
    gt[:, 2] += gt[:, 5] * 0.5
    gt, lab = gt.to(dev()), torch.from_numpy(labels[:n_gt]).to(dev())
    cls = torch.from_numpy(g5['all_cls_scores']).to(dev())
    box = torch.from_numpy(g5['all_bbox_preds']).to(dev())
    losses, d_cls, d_box, assigned = detr_loss_device(h, cls, box, [gt], [lab], device_assign=device_assign)
    if torch.is_tensor(assigned):
        assert device_assign and int(h.last_assign_status.item()) == 0
        assigned = assigned.cpu().numpy()
    for k, v in losses.items():
        ref = float(g7[k.replace('.', '_')])
        assert abs(float(v) - ref) <= 2e-5 * max(1.0, abs(ref)), (k, float(v), ref)
    if n_gt:
        want = g7['gt_inds'] - 1                    # the reference stores gt index + 1, 0 = unmatched
        assert np.array_equal(assigned[:, 0], want)
    cl, bl = cls.clone().requires_grad_(True), box.clone().requires_grad_(True)
    ref_losses = h.loss([gt], [lab], {'all_cls_scores': cl, 'all_bbox_preds': bl})
    sum(ref_losses.values()).backward()
    assert rel(d_cls, cl.grad) < 2e-5
    if n_gt:
        assert rel(d_box, bl.grad) < 2e-5
    else:
        assert float(d_box.abs().max()) == 0.0


@pytest.mark.parametrize('Q,G,B', [(900, 24, 2), (900, 1, 1), (900, 128, 1), (37, 37, 3), (1024, 60, 1), (64, 5, 2)])
def test_device_assignment_equals_scipy(A, Q, G, B):
    """tc_lsa_assign against scipy.optimize.linear_sum_assignment (ASSIGN:117-125) on random cost matrices, structured
    ones (every ground-truth box has a few cheap queries, as a trained head produces) and ragged batches (samples with
    fewer boxes, one without any): the same assignment, query for query; `num_pos` = the matched boxes per output;
    a NaN cost leaves that sample unassigned and raises `status` (scipy raises ValueError there)."""
    import ctypes as C
    from scipy.optimize import linear_sum_assignment
    from transcar_amd import _lib as L
    lib = L.lib()
    rng = np.random.RandomState(Q * 131 + G)
    Lyr = 3
    cost = rng.rand(Lyr, B, Q, G).astype(np.float32) * 4.0
    # structure: box g is close to queries around 7 g (+ noise): low costs there
    for g in range(G):
        cost[:, :, (7 * g) % Q, g] *= 0.05
        cost[1, :, (7 * g + 3) % Q, g] *= 0.02
    counts = np.full(B, G, dtype=np.int32)
    if B > 1:
        counts[1] = max(0, G // 3)
    if B > 2:
        counts[2] = 0
    for b in range(B):
        cost[:, b, :, counts[b]:] = 0.0                       # tc_match_cost writes 0 beyond a sample's count
    c_d, n_d = torch.from_numpy(cost).to(dev()), torch.from_numpy(counts).to(dev())
    asg = torch.full((Lyr, B, Q), -7, dtype=torch.int32, device=dev())
    z = torch.zeros(2 * Lyr + 1, dtype=torch.float32, device=dev())
    num_pos, status = z[:2 * Lyr].view(Lyr, 2), z[2 * Lyr:].view(torch.int32)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.tc_lsa_assign(c_d.data_ptr(), n_d.data_ptr(), Lyr, B, Q, G, asg.data_ptr(), num_pos.data_ptr(),
                              status.data_ptr(), st), 'tc_lsa_assign')
    got = asg.cpu().numpy()
    assert int(status.item()) == 0
    for l in range(Lyr):
        for b in range(B):
            want = np.full(Q, -1, dtype=np.int32)
            if counts[b]:
                rows, cols = linear_sum_assignment(cost[l, b, :, :counts[b]])
                want[rows] = cols
            assert np.array_equal(got[l, b], want), (l, b, int((got[l, b] != want).sum()))
    assert np.allclose(num_pos.cpu().numpy(), float(counts.sum()))
    # a non-finite cost: that sample unassigned, status raised, the others untouched
    cost[0, 0, 5, 0] = np.nan
    c_d = torch.from_numpy(cost).to(dev())
    z.zero_()
    L.check(lib.tc_lsa_assign(c_d.data_ptr(), n_d.data_ptr(), Lyr, B, Q, G, asg.data_ptr(), num_pos.data_ptr(),
                              status.data_ptr(), st), 'tc_lsa_assign')
    got2 = asg.cpu().numpy()
    assert int(status.item()) == 1 and (got2[0, 0] == -1).all()
    assert np.array_equal(got2[1:], got[1:]) and np.array_equal(got2[0, 1:], got[0, 1:])


def test_fused_step_with_device_loss_equals_torch_loss(A, golden_dir):
    from transcar_amd import ops
    from transcar_amd.trainer import FusionTrainer
    h = train_head(golden_dir)
    feats, metas, gt, labels = frame_inputs(golden_dir)
    nhwc = [ops.to_nhwc(f) for f in feats]
    l2i = ops.lidar2img_tensor(metas, dev())
    img_hw = metas[0]['img_shape'][0][:2]
    tokens, pad_mult = h.radar_tokens(metas, dev())
    tr = FusionTrainer(h, dropout=0.0)
    tr.device_loss = False
    l_torch = tr.step_fused_nhwc(nhwc, l2i, img_hw, tokens, pad_mult, [gt], [labels], update=False)
    g_torch = tr.bucket.grads.clone()
    tr.device_loss = True
    l_dev = tr.step_fused_nhwc(nhwc, l2i, img_hw, tokens, pad_mult, [gt], [labels], update=False)
    for k in l_torch:
        assert abs(float(l_torch[k]) - float(l_dev[k])) < 1e-5 * max(1.0, abs(float(l_torch[k]))), k
    assert float((tr.bucket.grads - g_torch).abs().max() / g_torch.abs().max()) < 1e-4


def test_fused_iteration_on_a_frame_without_ground_truth(A, golden_dir):
    """A sample with no annotated box (nuScenes has them): every query is background, the normalisers clamp to 1
    (HEAD:885-902), only the classification loss is non-zero.  The device loss + backward chain give the torch
    loss path's losses and gradients, and an optimizer step on them leaves finite parameters."""
    from transcar_amd import ops
    from transcar_amd.trainer import FusionTrainer
    h = train_head(golden_dir)
    feats, metas, gt, labels = frame_inputs(golden_dir)
    nhwc = [ops.to_nhwc(f) for f in feats]
    l2i = ops.lidar2img_tensor(metas, dev())
    img_hw = metas[0]['img_shape'][0][:2]
    tokens, pad_mult = h.radar_tokens(metas, dev())
    gt0, lb0 = gt[:0], labels[:0]
    tr = FusionTrainer(h, dropout=0.0, lr=1e-3)
    tr.device_loss = False
    l_torch = tr.step_fused_nhwc(nhwc, l2i, img_hw, tokens, pad_mult, [gt0], [lb0], update=False)
    g_torch = tr.bucket.grads.clone()
    tr.device_loss = True
    l_dev = tr.step_fused_nhwc(nhwc, l2i, img_hw, tokens, pad_mult, [gt0], [lb0], update=False)
    assert float(g_torch.abs().max()) > 0
    for k in l_torch:
        assert abs(float(l_torch[k]) - float(l_dev[k])) < 1e-5 * max(1.0, abs(float(l_torch[k]))), k
        if 'bbox' in k:
            assert float(l_dev[k]) == 0.0
    assert float((tr.bucket.grads - g_torch).abs().max() / g_torch.abs().max()) < 1e-4
    p0 = tr.bucket.params.clone()
    tr.step_fused_nhwc(nhwc, l2i, img_hw, tokens, pad_mult, [gt0], [lb0])
    torch.cuda.synchronize()
    assert torch.isfinite(tr.bucket.params).all() and not torch.equal(tr.bucket.params, p0)


@pytest.mark.parametrize('path', ['fused', 'autograd'])
def test_fused_training_with_dropout_matches_reference_formula(A, golden_dir, path):
    """The four dropout sites of every fusion layer (HEAD:129-171, 581-585; p = 0.1) in
    tc_radar_train_fwd / _bwd.  The masks are counter-based, so the check is: with the SAME masks
    (read back through tc_dropout_mask) the reference formula -- the oracle in train mode with given
    masks, torch autograd -- gives the same outputs, losses and gradients."""
    import ctypes as C
    from transcar_amd import _lib as L
    from transcar_amd import ops
    from transcar_amd.trainer import FusionTrainer
    p = 0.1
    h = train_head(golden_dir)
    feats, metas, gt, labels = frame_inputs(golden_dir)
    nhwc = [ops.to_nhwc(f) for f in feats]
    l2i = ops.lidar2img_tensor(metas, dev())
    img_hw = metas[0]['img_shape'][0][:2]
    tokens, pad_mult = h.radar_tokens(metas, dev())
    tr = FusionTrainer(h, dropout=p, seed=5)
    if path == 'fused':          # tc_radar_train_fwd / _bwd
        h._train_forwards = 3
        losses = tr.step_fused_nhwc(nhwc, l2i, img_hw, tokens, pad_mult, [gt], [labels], update=False)
        seed = tr.last_dropout_seed
        hip_grads = {n: q.grad.detach().cpu().clone() for n, q in h.trainable_parameters()}
        # a different seed gives different masks and a different loss; the same seed the same
        h._train_forwards = 3                                    # rewind the forward counter: same masks
        l_same = tr.step_fused_nhwc(nhwc, l2i, img_hw, tokens, pad_mult, [gt], [labels], update=False)
        assert tr.last_dropout_seed == seed                      # (the loss sums use atomics: equal up to rounding)
        assert all(abs(float(l_same[k]) - float(losses[k])) <= 1e-5 * max(1.0, abs(float(losses[k]))) for k in losses)
        # the next forward (e.g. a gradient-accumulation step: no optimizer step in between) draws new masks
        l_other = tr.step_fused_nhwc(nhwc, l2i, img_hw, tokens, pad_mult, [gt], [labels], update=False)
        assert tr.last_dropout_seed != seed
        assert any(abs(float(l_other[k]) - float(losses[k])) > 1e-4 * max(1.0, abs(float(losses[k]))) for k in losses)
    else:                        # the module API: head.train()(feats, metas) -> loss -> backward
        h.set_dropout(p, decoder=False)          # this test: the fusion layers' sites only
        h.dropout_seed = 11
        tr.bucket.zero_grad()
        outs = h.train()(feats, metas)
        seed = h.last_dropout_seed
        losses = h.loss([gt], [labels], outs)
        sum(v for k, v in losses.items() if 'loss' in k).backward()
        losses = {k: v.detach() for k, v in losses.items()}
        hip_grads = {n: q.grad.detach().cpu().clone() for n, q in h.trainable_parameters()}
        with torch.no_grad():
            again = h(feats, metas)                               # next forward: next masks
        assert h.last_dropout_seed != seed
        assert float((again['all_cls_scores'] - outs['all_cls_scores']).abs().max()) > 1e-4

    # ---- the masks
    lib = L.lib()
    Q, Cd, Fd, H, TR = h.num_query, 256, 512, 8, 1500

    def mask(site, n):
        out = torch.empty(n, dtype=torch.float32, device=dev())
        L.check(lib.tc_dropout_mask(p, seed, site, n, out.data_ptr(),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'tc_dropout_mask')
        return out.cpu()
    drop = []
    for r in range(3):
        pm = mask(4 * r + 0, Q * H * TR).view(Q, H, TR).permute(1, 0, 2).contiguous()
        drop.append(dict(probs=pm, d2=mask(4 * r + 1, Q * Cd).view(Q, Cd),
                         ffn=mask(4 * r + 2, Q * Fd).view(Q, Fd), d3=mask(4 * r + 3, Q * Cd).view(Q, Cd)))
    m = drop[0]['ffn']
    assert set(np.unique(m.numpy()).tolist()) == {0.0, np.float32(1.0 / (1.0 - p))}
    keep = float((m > 0).float().mean())
    assert abs(keep - (1 - p)) < 4e-3, keep                     # 460 800 Bernoulli draws
    assert not torch.equal(drop[0]['d2'], drop[1]['d2']) and not torch.equal(drop[0]['d2'], drop[0]['d3'])

    # ---- the reference formula with these masks (CPU, autograd)
    g_feats, g_l2i, frame, boxes, lab = g8_inputs(golden_dir)
    sd = O.to_torch_sd(synth.make_state_dict(3))
    for k, v in sd.items():
        if trainable(k):
            v.requires_grad_(True)
    outs = O.head_forward(sd, [torch.from_numpy(f) for f in g_feats], torch.from_numpy(g_l2i).float()[None],
                          configs.IMG_SHAPE[:2], O.build_radar_features(frame), configs.point_cloud_range,
                          drop=drop)
    res, _ = O.loss(outs, torch.from_numpy(boxes), torch.from_numpy(lab), sd['code_weights'])
    for k, v in losses.items():
        ref = float(res[k])
        assert abs(float(v) - ref) < 2e-3 * max(1.0, abs(ref)), (k, float(v), ref)
    sum(res.values()).backward()
    worst = 0.0
    for n, g in hip_grads.items():
        ref = sd[n].grad
        d = float((g.double() - ref.double()).norm() / max(float(ref.double().norm()), 1e-12))
        worst = max(worst, d)
        assert d < 5e-3, (n, d)
    # and they are NOT the no-dropout gradients
    g8 = np.load(os.path.join(golden_dir, 'g8_train_grads.npz'))
    k0 = 'rf_linear2.weight'
    assert abs(float(hip_grads[k0].double().norm()) - float(g8[k0.replace('.', '__') + '__stats'][2])) > \
        1e-3 * float(g8[k0.replace('.', '__') + '__stats'][2])


def test_decoder_train_mode_dropout_matches_reference_formula(A, golden_dir):
    """tools/train.py:245-252 freezes the decoder's parameters but leaves it in train mode, so
    the five dropout sites of every decoder layer (CFG:68-80: the mmcv attention wrapper's
    probability and output dropout, the FFN's two; XFMR:378: Detr3DCrossAtten.dropout; p = 0.1)
    act on configs[2].  tc_head_options.decoder_dropout_p switches them on in the HIP decoder
    (attention core + row chains, layer 0 no longer folded into the checkpoint constants).  The
    masks are counter-based, so the check is: with the SAME masks (read back through
    tc_dropout_mask) the reference formula -- the oracle's decoder with given masks -- gives the
    same decoder states and reference points."""
    import ctypes as C
    from transcar_amd import _lib as L
    from transcar_amd import ops
    from transcar_amd.detr3d_head import head_options
    p, seed = 0.1, 0x5EED1234ABCD
    h = train_head(golden_dir)
    assert h.decoder_dropout_p() == 0.0                       # train_head: set_dropout(0.0)
    h.set_decoder_dropout(p)
    assert h.decoder_dropout_p() == p
    feats, metas, _, _ = frame_inputs(golden_dir)
    nhwc = ops.to_nhwc_levels(feats)
    l2i = ops.lidar2img_tensor(metas, dev())
    img_hw = metas[0]['img_shape'][0][:2]
    tokens, pad_mult = h.radar_tokens(metas, dev())
    with torch.no_grad():
        ev = h.eval().forward_nhwc(nhwc, l2i, img_hw, tokens, pad_mult, aux=True)
        h.train()
        tr_ = h.forward_nhwc(nhwc, l2i, img_hw, tokens, pad_mult, aux=True, _allow_train=True,
                             options=head_options(decoder_dropout_p=p, dropout_seed=seed))
        tr2 = h.forward_nhwc(nhwc, l2i, img_hw, tokens, pad_mult, aux=True, _allow_train=True,
                             options=head_options(decoder_dropout_p=p, dropout_seed=seed))
        tr3 = h.forward_nhwc(nhwc, l2i, img_hw, tokens, pad_mult, aux=True, _allow_train=True,
                             options=head_options(decoder_dropout_p=p, dropout_seed=seed + 1))
        off = h.forward_nhwc(nhwc, l2i, img_hw, tokens, pad_mult, aux=True, _allow_train=True,
                             options=head_options(decoder_dropout_p=0.0, dropout_seed=seed))
    hs = tr_['aux']['inter_states']
    assert torch.equal(hs, tr2['aux']['inter_states'])               # same seed, same masks
    assert float((hs - tr3['aux']['inter_states']).abs().max()) > 1e-2   # another seed
    assert float((hs - ev['aux']['inter_states']).abs().max()) > 1e-2    # not the eval forward
    assert torch.equal(off['aux']['inter_states'], ev['aux']['inter_states'])   # p = 0: the eval kernels

    # ---- the masks, site = 16 + 8 * layer + {0 probs, 1 self-attn out, 2 cross-attn out, 3 FFN hidden, 4 FFN out}
    lib = L.lib()
    Q, Cd, Fd, H = h.num_query, 256, 512, 8

    def mask(site, n):
        out = torch.empty(n, dtype=torch.float32, device=dev())
        L.check(lib.tc_dropout_mask(p, seed, site, n, out.data_ptr(),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'tc_dropout_mask')
        return out.cpu()
    dec_drop = []
    for l in range(6):
        s0 = 16 + 8 * l
        dec_drop.append(dict(
            probs=mask(s0 + 0, H * Q * Q).view(H, Q, Q),              # [B*heads, Q, Q], B = 1
            sa=mask(s0 + 1, Q * Cd).view(Q, 1, Cd), ca=mask(s0 + 2, Q * Cd).view(Q, 1, Cd),
            ffn_h=mask(s0 + 3, Q * Fd).view(Q, 1, Fd), ffn_o=mask(s0 + 4, Q * Cd).view(Q, 1, Cd)))
    keep = float((dec_drop[2]['probs'] > 0).float().mean())
    assert abs(keep - (1 - p)) < 1e-3, keep                      # 6.5 M Bernoulli draws
    g_feats, g_l2i, frame, _, _ = g8_inputs(golden_dir)
    sd = O.to_torch_sd(synth.make_state_dict(3))
    with torch.no_grad():
        want_hs, init_ref, want_refs, _ = O.transformer(
            sd, [torch.from_numpy(f) for f in g_feats], configs.point_cloud_range,
            torch.from_numpy(g_l2i).float()[None], configs.IMG_SHAPE[:2], dec_drop=dec_drop)
    np.testing.assert_allclose(tr_['aux']['init_reference'].cpu().numpy(), init_ref.numpy(), atol=1e-6, rtol=0)
    # layer by layer; the loop gain of the refinement makes late layers looser
    got_hs = hs.cpu().numpy()[:, 0]
    got_refs = tr_['aux']['inter_references'].cpu().numpy()
    np.testing.assert_allclose(got_hs[0], want_hs[0][:, 0].numpy(), atol=2e-4, rtol=0)
    np.testing.assert_allclose(got_refs[0], want_refs[0].numpy(), atol=2e-5, rtol=0)
    np.testing.assert_allclose(got_refs, want_refs.numpy(), atol=2e-4, rtol=0)
    np.testing.assert_allclose(got_hs, want_hs[:, :, 0].numpy(), atol=2e-3, rtol=0)
    # the training iteration uses it: FusionTrainer draws decoder masks from the iteration's seed
    from transcar_amd.trainer import FusionTrainer
    t = FusionTrainer(h, dropout=p, seed=3)
    assert t.decoder_dropout == p


def test_optimizer_step_defers_the_repack_to_the_next_inference_consumer(A, golden_dir):
    """The trainer marks the packed radar weights stale instead of re-packing them every iteration;
    the next eval forward AND the next FramePipeline replay must see the updated weights."""
    from transcar_amd import ops
    from transcar_amd.pipeline import FramePipeline
    from transcar_amd.trainer import FusionTrainer
    h = train_head(golden_dir)
    feats, metas, gt, labels = frame_inputs(golden_dir)
    nhwc = [ops.to_nhwc(f) for f in feats]
    l2i = ops.lidar2img_tensor(metas, dev())
    img_hw = metas[0]['img_shape'][0][:2]
    tokens, pad_mult = h.radar_tokens(metas, dev())
    lane = dict(nhwc=nhwc, l2i=l2i, hw=img_hw, tokens=tokens, pad_mult=pad_mult)
    with torch.no_grad():
        pipe = FramePipeline(h.eval(), [lane], decode=False)
        pipe.launch()
        pipe.synchronize()
        before = pipe.outputs[0][0]['all_cls_scores'].clone()
    tr = FusionTrainer(h, lr=1e-3, dropout=0.0)       # moves the trainable parameters into its flat bucket
    with pytest.raises(Exception, match='recapture'):
        pipe.launch()
    with torch.no_grad():
        h.eval()
        pipe.recapture()
    h.train()
    tr.step(feats, metas, [gt], [labels])
    assert h._packed_dirty
    with torch.no_grad():
        h.eval()
        pipe.launch()
        pipe.synchronize()
        assert not h._packed_dirty
        after = pipe.outputs[0][0]['all_cls_scores'].clone()
        assert float((after - before).abs().max()) > 1e-3
        fresh = h.forward_nhwc(nhwc, l2i, img_hw, tokens, pad_mult)['all_cls_scores']
        torch.cuda.synchronize()
    assert torch.equal(fresh, after)


@pytest.mark.parametrize('tag', ['tiny', 'res101'])
def test_chain_forward_equals_operator_forward_with_dropout(A, golden_dir, tag):
    """tc_radar_train_fwd_fused (the stack's forward as launches of the fused row chains, the tape stored as it
    is produced) against tc_radar_train_fwd (one launch per operator) with the SAME dropout masks (p = 0.1, same
    seed): same outputs, same losses, and -- through the one backward both tapes feed -- the same gradients."""
    from transcar_amd import ops
    from transcar_amd.trainer import FusionTrainer
    h = train_head(golden_dir)
    feats, metas, gt, labels = frame_inputs(golden_dir, tag)
    nhwc = [ops.to_nhwc(f) for f in feats]
    l2i = ops.lidar2img_tensor(metas, dev())
    img_hw = metas[0]['img_shape'][0][:2]
    tokens, pad_mult = h.radar_tokens(metas, dev())
    tr = FusionTrainer(h, dropout=0.1, seed=7, decoder_dropout=0.0)
    res = {}
    for chain in (True, False):
        tr.chain_forward = chain
        h._train_forwards = 11                                   # same forward counter: same masks
        losses = tr.step_fused_nhwc(nhwc, l2i, img_hw, tokens, pad_mult, [gt], [labels], update=False)
        res[chain] = ({k: float(v) for k, v in losses.items()}, tr.bucket.grads.clone(), tr.last_dropout_seed)
    assert res[True][2] == res[False][2]
    for k, v in res[False][0].items():
        assert abs(res[True][0][k] - v) < 2e-4 * max(1.0, abs(v)), (k, res[True][0][k], v)
    d = (res[True][1] - res[False][1]).abs().max() / res[False][1].abs().max()
    assert float(d) < 5e-4, float(d)


@pytest.mark.parametrize('T_tok,n_per', [(576, 108), (1500, 298)])
def test_fused_training_with_many_radar_tokens(A, golden_dir, T_tok, n_per):
    """a real frame can carry more than 511 radar points (5 radars x 5 sweeps, padded to 1500
    tokens, HEAD:526-530); round 3's tc_radar_train_fwd_fused refused T > 512, so the default trainer raised
    mid-training.  A training iteration at T = 576 and at the reference's own T = 1500 (more than 2 Q tokens:
    also the du0 scratch the backward used to take from a query-sized tape slot): the fused chains (forward and
    backward) against the operator-level forward / backward with the same dropout masks -- same losses, same
    gradients, all finite, and the radar stack is really exercised (queries with hits)."""
    from transcar_amd import ops, radar as R
    from transcar_amd.trainer import FusionTrainer
    h = train_head(golden_dir)
    feats, metas, gt, labels = frame_inputs(golden_dir, 'tiny')
    base = metas[0]['radar']
    frame = synth.make_radar_frame(seed=11, n_per_radar=n_per)            # 5 x n_per points inside the range
    f36 = R.build_radar_features(frame)
    assert 511 < f36.shape[0] < T_tok or T_tok == 1500, f36.shape
    del base
    nhwc = [ops.to_nhwc(f) for f in feats]
    l2i = ops.lidar2img_tensor(metas, dev())
    img_hw = metas[0]['img_shape'][0][:2]
    tok_np, pad_mult = R.pack_tokens([f36], T=T_tok)
    tokens = torch.from_numpy(tok_np).to(dev())
    assert tokens.shape[1] == T_tok and pad_mult == 1500 - T_tok + 1
    tr = FusionTrainer(h, dropout=0.1, seed=5, decoder_dropout=0.0)
    res = {}
    for fused in (True, False):
        tr.chain_forward = tr.chain_backward = fused
        h._train_forwards = 3                                     # same forward counter: same masks
        losses = tr.step_fused_nhwc(nhwc, l2i, img_hw, tokens, pad_mult, [gt], [labels], update=False)
        res[fused] = ({k: float(v) for k, v in losses.items()},
                      {n: q.grad.detach().clone() for n, q in h.trainable_parameters()})
        assert all(np.isfinite(v) for v in res[fused][0].values())
        assert all(torch.isfinite(v).all() for v in res[fused][1].values())
    for k, v in res[False][0].items():
        assert abs(res[True][0][k] - v) < 2e-4 * max(1.0, abs(v)), (k, res[True][0][k], v)
    assert_grad_dicts_equal_up_to_one_relu_tie(res[True][1], res[False][1], 5e-4)
    # the attention path carried gradient (the frame has radar returns inside the gates)
    assert float(res[True][1]['rf_multihead_attn.in_proj_weight'].abs().max()) > 0


@pytest.mark.parametrize('tag,p', [('tiny', 0.0), ('tiny', 0.1), ('res101', 0.1)])
def test_chain_backward_equals_operator_backward(A, golden_dir, tag, p):
    """tc_radar_train_bwd_fused (the query side of the three fusion layers as ONE launch of the backward row
    chain, the token side as a few launches, every weight gradient in one grouped GEMM launch) against
    tc_radar_train_bwd (one launch per operator) on the SAME forward (same tape, same dropout masks): every one of
    the 98 gradient tensors agrees."""
    from transcar_amd import ops
    from transcar_amd.trainer import FusionTrainer
    h = train_head(golden_dir)
    feats, metas, gt, labels = frame_inputs(golden_dir, tag)
    nhwc = [ops.to_nhwc(f) for f in feats]
    l2i = ops.lidar2img_tensor(metas, dev())
    img_hw = metas[0]['img_shape'][0][:2]
    tokens, pad_mult = h.radar_tokens(metas, dev())
    tr = FusionTrainer(h, dropout=p, seed=3, decoder_dropout=0.0)
    res = {}
    for chain in (True, False):
        tr.chain_backward = chain
        h._train_forwards = 5
        losses = tr.step_fused_nhwc(nhwc, l2i, img_hw, tokens, pad_mult, [gt], [labels], update=False)
        res[chain] = {n: q.grad.detach().clone() for n, q in h.trainable_parameters()}
        assert all(torch.isfinite(v).all() for v in res[chain].values())
    worst = 0.0
    for n, want in res[False].items():
        got = res[True][n]
        scale = float(want.abs().max())
        d = float((got - want).abs().max())
        assert d <= 2e-4 * max(scale, 1e-6) + 1e-7, (n, d, scale)
        worst = max(worst, d / max(scale, 1e-12))
    assert len(res[False]) == 98 - 14 or len(res[False]) >= 84


def test_training_iteration_without_any_radar_return(A, golden_dir):
    """A frame whose radar sweeps are empty (every token a pad row, no query has a hit): the fusion layers pass the
    decoder state through, the attention path carries no gradient.  The backward chain, the operator-level backward
    and the autograd path agree on every gradient, all finite; the attention projections' gradients are exactly 0."""
    from transcar_amd import ops
    from transcar_amd.trainer import FusionTrainer
    h = train_head(golden_dir)
    feats, metas, gt, labels = frame_inputs(golden_dir)
    metas[0]['radar'] = synth.make_radar_frame(seed=5, n_per_radar=[0, 0, 0, 0, 0])
    nhwc = [ops.to_nhwc(f) for f in feats]
    l2i = ops.lidar2img_tensor(metas, dev())
    img_hw = metas[0]['img_shape'][0][:2]
    tokens, pad_mult = h.radar_tokens(metas, dev())
    assert torch.all(tokens[..., 0] == 500.0)
    tr = FusionTrainer(h, dropout=0.0, decoder_dropout=0.0)
    res = {}
    for name, chain in (('chain', True), ('operators', False)):
        tr.chain_backward = chain
        tr.step_fused_nhwc(nhwc, l2i, img_hw, tokens, pad_mult, [gt], [labels], update=False)
        res[name] = {n: q.grad.detach().clone() for n, q in h.trainable_parameters()}
    tr.bucket.zero_grad()
    h.train()
    outs = h.forward_train_nhwc(nhwc, l2i, img_hw, tokens, pad_mult)
    losses = h.loss([gt], [labels], outs)
    sum(v for k_, v in losses.items() if 'loss' in k_).backward()
    res['autograd'] = {n: q.grad.detach().clone() for n, q in h.trainable_parameters()}
    for n, want in res['autograd'].items():
        scale = float(want.abs().max())
        for other in ('chain', 'operators'):
            got = res[other][n]
            assert torch.isfinite(got).all(), (other, n)
            assert float((got - want).abs().max()) <= 3e-4 * max(scale, 1e-6) + 1e-7, (other, n)
        if 'rf_multihead_attn' in n:
            assert scale == 0.0, n


def test_decoder_prefetch_changes_nothing_but_the_schedule(A, golden_dir):
    """step_fused_nhwc(prefetch=next frame): the frozen decoder's forward of iteration i + 1 is enqueued on a side
    stream while the host solves iteration i's assignment.  Three optimizer steps with and without it (two frames
    alternating, dropout on, same seeds): same losses, same parameters up to the rounding of the atomics."""
    from transcar_amd import ops
    from transcar_amd.trainer import FusionTrainer
    feats, metas, gt, labels = frame_inputs(golden_dir)
    frames = []
    for i in range(2):
        f = feats if i == 0 else [torch.from_numpy(x).to(dev()) for x in synth.make_feats('tiny', seed=9, smooth=(4, 6))]
        nhwc = [ops.to_nhwc(x) for x in f]
        frames.append(dict(feats_nhwc=nhwc, lidar2img=ops.lidar2img_tensor(metas, dev()),
                           img_hw=metas[0]['img_shape'][0][:2]))
    res = {}
    for pre in (False, 'again', True):                 # 'again': the run-to-run noise of the backward's atomics
        h = train_head(golden_dir)
        tokens, pad_mult = h.radar_tokens(metas, dev())
        for fr in frames:
            fr.update(tokens=tokens, pad_mult=pad_mult)
        # (a small learning rate: a parameter difference of rounding size can flip a hard decision of the next forward --
        # a radar gate, a ReLU at zero -- and a 1e-3 step made that a 1-in-8 event; the subject here is the schedule)
        tr = FusionTrainer(h, dropout=0.1, seed=2, lr=1e-5)
        hist = []
        for it in range(3):
            cur, nxt = frames[it % 2], frames[(it + 1) % 2]
            losses = tr.step_fused_nhwc(cur['feats_nhwc'], cur['lidar2img'], cur['img_hw'], cur['tokens'], cur['pad_mult'],
                                        [gt], [labels], prefetch=nxt if pre is True else None)
            hist.append({k: float(v) for k, v in losses.items()})
        torch.cuda.synchronize()
        res[pre] = (hist, tr.bucket.params.clone())
    # AdamW normalises the step, so the last-bit differences of the atomically summed gradients reach the parameters at
    # ~1e-5 and the later losses with them: the yardstick is what two plain runs differ by at the same iteration
    for a_, b_, c_ in zip(res[False][0], res[True][0], res['again'][0]):
        for k in a_:
            assert abs(a_[k] - c_[k]) <= 2e-4 * max(1.0, abs(a_[k])), (k, a_[k], c_[k])        # (fixed cap on the yardstick)
            assert abs(a_[k] - b_[k]) <= 5e-5 * max(1.0, abs(a_[k])) + 3.0 * abs(a_[k] - c_[k]), (k, a_[k], b_[k], c_[k])
    for k, v in res[False][0][0].items():              # the first iteration has no history: exact up to the loss kernel's atomics
        assert abs(v - res[True][0][0][k]) <= 2e-6 * max(1.0, abs(v)), (k, v, res[True][0][0][k])
    assert_params_equal_up_to_adam_noise(res[False][1], res[True][1], res['again'][1], lr=1e-5, steps=3, slack=1e-5)


def test_deterministic_backward_is_bit_identical_on_any_schedule(A, golden_dir):
    """FusionTrainer(deterministic=True) (round 5; tc_radar_train_bwd_fused_det): every partial sum the
    backward adds with a float atomic -- weight-gradient row chunks, bias column sums, LayerNorm parameter gradients of
    the row chain and the token side, dK | dV of the attention backward -- goes through an integer atomic on a fixed-point
    shadow, so the sums do not depend on the order the workgroups arrive in.
    (a) one backward, twice from the same state: gradients torch.equal (the float-atomic mode is compared with the same
        code and is allowed to differ -- and must agree with the deterministic gradients to 2e-6 of their scale);
    (b) three optimizer steps with dropout, with and without the decoder look-ahead on its side stream, and a repeat:
        losses and parameters torch.equal -- the `==` the schedule test of the atomic mode cannot ask for."""
    from transcar_amd import ops
    from transcar_amd.trainer import FusionTrainer
    feats, metas, gt, labels = frame_inputs(golden_dir)
    frames = []
    for i in range(2):
        f = feats if i == 0 else [torch.from_numpy(x).to(dev()) for x in synth.make_feats('tiny', seed=9, smooth=(4, 6))]
        frames.append(dict(feats_nhwc=[ops.to_nhwc(x) for x in f], lidar2img=ops.lidar2img_tensor(metas, dev()),
                           img_hw=metas[0]['img_shape'][0][:2]))
    # (a)
    grads = {}
    for mode in ('det', 'det2', 'atomic'):
        h = train_head(golden_dir)
        tokens, pad_mult = h.radar_tokens(metas, dev())
        tr = FusionTrainer(h, dropout=0.1, seed=2, lr=1e-5, deterministic=mode != 'atomic')
        fr = frames[0]
        tr.step_fused_nhwc(fr['feats_nhwc'], fr['lidar2img'], fr['img_hw'], tokens, pad_mult, [gt], [labels], update=False)
        torch.cuda.synchronize()
        grads[mode] = tr.bucket.grads.clone()
        if mode != 'atomic':
            assert int(tr._shadow[:-8].abs().max()) == 0        # the call leaves its shadow zero (the last 8 words: scratch)
    assert float(grads['det'].abs().max()) > 0
    assert torch.equal(grads['det'], grads['det2'])
    scale = float(grads['det'].abs().max())
    assert float((grads['det'] - grads['atomic']).abs().max()) <= 2e-6 * scale
    # (b)
    res = {}
    for pre in (False, 'again', True):
        h = train_head(golden_dir)
        tokens, pad_mult = h.radar_tokens(metas, dev())
        for fr in frames:
            fr.update(tokens=tokens, pad_mult=pad_mult)
        tr = FusionTrainer(h, dropout=0.1, seed=2, lr=1e-3, deterministic=True)
        hist = []
        for it in range(3):
            cur, nxt = frames[it % 2], frames[(it + 1) % 2]
            losses = tr.step_fused_nhwc(cur['feats_nhwc'], cur['lidar2img'], cur['img_hw'], cur['tokens'], cur['pad_mult'],
                                        [gt], [labels], prefetch=nxt if pre is True else None)
            hist.append(losses)
        torch.cuda.synchronize()
        res[pre] = (hist, tr.bucket.params.clone())
    for other in ('again', True):
        assert torch.equal(res[False][1], res[other][1]), other
        for a_, b_ in zip(res[False][0], res[other][0]):
            for k in a_:
                # the loss VALUES still meet in float atomics of the loss kernel (they are reported, nothing is derived
                # from them): equal to rounding
                assert abs(float(a_[k]) - float(b_[k])) <= 2e-6 * max(1.0, abs(float(a_[k]))), (other, k)


def test_deterministic_backward_with_a_varying_token_count(A, golden_dir):
    """radar.pack_tokens rounds T to multiples of 64, so T varies from frame to frame on real data.
    The device copy of the accumulation ranges used to sit right behind the dK | dV shadows -- an offset that moves with T --
    and a shorter frame left its six pointer words inside the shadow range of a later, longer frame (times 2^-40: ~1e2 added
    to dK | dV).  Now the copy lives in the LAST words of the shadow buffer.  One trainer sees T = 128, 64, 128 (update=False:
    same parameters); the gradients of the third call must be torch.equal to a fresh trainer's on the same frame, the
    shadow is zero in front of its tail after every call, and a non-representable partial sum is not silently finite."""
    from transcar_amd import ops
    from transcar_amd.trainer import FusionTrainer
    feats, metas, gt, labels = frame_inputs(golden_dir)
    g5 = np.load(os.path.join(golden_dir, 'g5_head_tiny.npz'))
    nhwc = [ops.to_nhwc(x) for x in feats]
    l2i = ops.lidar2img_tensor(metas, dev())
    img_hw = metas[0]['img_shape'][0][:2]

    def toks(h, n_per_radar):
        m = synth.make_img_metas(1, synth.make_lidar2img(), radar=synth.make_radar_frame(seed=5, n_per_radar=n_per_radar, centres=g5['radar_centres']))
        return h.radar_tokens(m, dev())
    h = train_head(golden_dir)
    tr = FusionTrainer(h, dropout=0.1, seed=2, lr=1e-5, deterministic=True)
    seen = []
    for n in (20, 10, 20):                     # 100 / 50 / 100 points -> T = 128 / 64 / 128
        tokens, pad_mult = toks(h, n)
        seen.append(tokens.shape[1])
        tr.bucket.zero_grad()
        tr.step_fused_nhwc(nhwc, l2i, img_hw, tokens, pad_mult, [gt], [labels], update=False)
        torch.cuda.synchronize()
        assert int(tr._shadow[:-8].abs().max()) == 0
    assert seen == [128, 64, 128], seen
    third = tr.bucket.grads.clone()
    h2 = train_head(golden_dir)
    tr2 = FusionTrainer(h2, dropout=0.1, seed=2, lr=1e-5, deterministic=True)
    tokens, pad_mult = toks(h2, 20)
    h2._train_forwards = h._train_forwards - 1          # the fresh trainer draws the dropout seed of the first one's third call
    tr2.step_fused_nhwc(nhwc, l2i, img_hw, tokens, pad_mult, [gt], [labels], update=False)
    torch.cuda.synchronize()
    assert h2.last_dropout_seed == h.last_dropout_seed
    assert float(third.abs().max()) > 0
    assert torch.equal(third, tr2.bucket.grads)


@pytest.mark.parametrize('rows', [16, 32])
def test_batched_decoder_lookahead_is_the_single_frame_decoder(A, golden_dir, rows):
    """FusionTrainer(prefetch_depth=P) (round 4): the FROZEN decoder of the next P frames runs as ONE
    batched launch sequence at 16-row tiles, frame b with the dropout masks of seed + b * SEED_STRIDE
    (tc_head_options.dropout_seed_stride) -- the seeds those iterations draw themselves.
    (a) decoder states / references / last box of frame b of the batch are BIT-IDENTICAL to frame b run alone with its
        own seed at the same tile height, dropout on (all five sites per layer incl. the attention probabilities);
    (b) 2 P optimizer steps with the look-ahead and without it: every frame came from the look-ahead, same losses, same
        parameters (up to the rounding of the backward's atomics, as in the depth-1 test);
    (c) a look-ahead whose inputs were overwritten in place, or whose seeds no longer match the forward counter, is
        dropped and nothing is drawn from the counter for it."""
    from transcar_amd import ops
    from transcar_amd.trainer import FusionTrainer
    P = 3
    feats, metas, gt, labels = frame_inputs(golden_dir)
    l2i1 = ops.lidar2img_tensor(metas, dev())
    img_hw = metas[0]['img_shape'][0][:2]
    per_frame = [feats] + [[torch.from_numpy(x).to(dev()) for x in synth.make_feats('tiny', seed=9 + i, smooth=(4, 6))]
                           for i in range(1, P)]
    # the frames twice in a row: every window of P consecutive frames (cyclically) is one contiguous view
    nhwc_b = [torch.cat([ops.to_nhwc(f[l]) for f in per_frame] * 2, 0).contiguous() for l in range(len(feats))]
    l2i_b = torch.cat([l2i1] * (2 * P), 0).contiguous()

    def make(depth, deterministic=False):
        h = train_head(golden_dir)
        tok1, pad_mult = h.radar_tokens(metas, dev())
        tok_b = torch.cat([tok1] * (2 * P), 0).contiguous()
        tr = FusionTrainer(h, dropout=0.1, seed=4, lr=1e-5, prefetch_depth=depth, decoder_dropout=0.1,
                           deterministic=deterministic)
        tr.decoder_tile_rows = rows                                # the same decoder arithmetic with and without look-ahead
        ncam = nhwc_b[0].shape[0] // (2 * P)

        def view(pos, n):
            return dict(nhwc=[f[ncam * pos:ncam * (pos + n)] for f in nhwc_b], l2i=l2i_b[pos:pos + n], tokens=tok_b[pos:pos + n])

        def window(start):                                         # the P frames from `start` on, as prefetch_decoder takes them
            w = view(start % P, P)
            return dict(feats_nhwc=w['nhwc'], lidar2img=w['l2i'], img_hw=img_hw, tokens=w['tokens'], pad_mult=pad_mult)
        return h, tr, view, window, pad_mult
    # (a)
    h, tr, view, window, pad_mult = make(P)
    assert tr.decoder_dropout > 0
    seed0 = 0x1234567
    w0 = window(0)
    full = tr._decoder_forward(w0['feats_nhwc'], w0['lidar2img'], img_hw, w0['tokens'], pad_mult, seed0, 1, lookahead=True)['aux']
    torch.cuda.synchronize()
    full = {k: v.clone() for k, v in full.items() if torch.is_tensor(v)}
    frames = [view(i, 1) for i in range(P)]
    for i, f in enumerate(frames):
        one = tr._decoder_forward(f['nhwc'], f['l2i'], img_hw, f['tokens'], pad_mult,
                                  (seed0 + i * tr.SEED_STRIDE) & 0xFFFFFFFFFFFFFFFF, 0)['aux']
        torch.cuda.synchronize()
        assert torch.equal(one['inter_states'][:, 0], full['inter_states'][:, i]), i
        assert torch.equal(one['inter_references'][:, 0], full['inter_references'][:, i]), i
        assert torch.equal(one['last_box'][0], full['last_box'][i]), i
    other = tr._decoder_forward(frames[1]['nhwc'], frames[1]['l2i'], img_hw, frames[1]['tokens'], pad_mult, seed0, 0)['aux']
    assert not torch.equal(other['inter_states'][:, 0], full['inter_states'][:, 1])      # (the seed matters: dropout is on)
    # (b)
    res = {}
    # ('again': the run-to-run noise of the backward's float atomics; 'det' / 'det-look': the same two schedules with the
    # order-free backward -- VERDICT r4 item 4: `==` instead of a yardstick)
    for look in (False, 'again', True, 'det', 'det-look'):
        h, tr, view, window, pad_mult = make(P, deterministic=str(look).startswith('det'))
        hist = []
        fifo = []                                # positions (in the doubled storage) of the frames of look-aheads under way
        for it in range(2 * P + 1):
            f = view(fifo.pop(0) if fifo else it % P, 1)           # iteration `it` trains frame it % P

            def lookahead(skip=0, it=it):
                # the loader's answer: the P frames that follow the `skip` pending ones, i.e. those of iterations
                # it + 1 + skip ... it + skip + P -- one contiguous window of the doubled storage
                s0 = (it + 1 + skip) % P
                fifo.extend(range(s0, s0 + P))
                return window(s0)
            losses = tr.step_fused_nhwc(f['nhwc'], f['l2i'], img_hw, f['tokens'], pad_mult, [gt], [labels],
                                        prefetch=lookahead if look in (True, 'det-look') else None)
            hist.append({k: float(v) for k, v in losses.items()})
        torch.cuda.synchronize()
        res[look] = (hist, tr.bucket.params.clone(), getattr(tr, 'lookahead_hits', 0))
    assert res[False][2] == 0 and res[True][2] == 2 * P              # every frame but the very first
    assert res['det'][2] == 0 and res['det-look'][2] == 2 * P
    assert torch.equal(res['det'][1], res['det-look'][1])            # seven optimizer steps, look-ahead or not: the same bits
    # (the losses of later iterations carry the run-to-run noise of the earlier steps' atomically summed gradients:
    # the yardstick is what two plain runs differ by at the same iteration)
    for a_, b_, c_ in zip(res[False][0], res[True][0], res['again'][0]):
        for k in a_:
            assert abs(a_[k] - b_[k]) <= 5e-5 * max(1.0, abs(a_[k])) + 3.0 * abs(a_[k] - c_[k]), (k, a_[k], b_[k], c_[k])
    # AdamW normalises the step, so the last-bit differences of the atomically summed gradients grow over the seven
    # steps: the look-ahead may differ from the plain run by no more than two plain runs differ from each other
    assert_params_equal_up_to_adam_noise(res[False][1], res[True][1], res['again'][1], lr=tr.lr, steps=2 * P + 1, slack=1e-5)
    # (c)
    h, tr, view, window, pad_mult = make(P)
    f = view(0, 1)
    tr.step_fused_nhwc(f['nhwc'], f['l2i'], img_hw, f['tokens'], pad_mult, [gt], [labels], prefetch=lambda skip: window(1))
    assert tr.lookahead_pending() == P
    counter = h._train_forwards
    nhwc_b[0].add_(0.0)                                               # the loader refills its static tensors in place
    f = view(1, 1)
    tr.step_fused_nhwc(f['nhwc'], f['l2i'], img_hw, f['tokens'], pad_mult, [gt], [labels], prefetch=None)
    assert tr.lookahead_pending() == 0 and getattr(tr, 'lookahead_hits', 0) == 0
    assert h._train_forwards == counter + 1                           # one seed per training forward, none for the dropped batch


@pytest.mark.parametrize('rows', [16, 32])
def test_train_mode_decoder_on_the_matrix_cores_draws_the_fp32_kernels_masks(A, golden_dir, rows):
    """The frozen decoder in train mode at 16- and 32-row tiles (32: what a nine-frame look-ahead runs since round 5;
    the f32 side of the comparison stays at 16 rows, the f32 MFMA kernels' largest tile): the two-plane f16 kernels (chains AND the staged attention core
    with dropout on the probabilities, round 4) against the fp32 MFMA kernels with the same seed -- the masks are a
    function of (seed, site, element index) only, so the two agree to fp32 rounding; without the seed they do not."""
    from transcar_amd import ops
    from transcar_amd.trainer import FusionTrainer
    h = train_head(golden_dir)
    feats, metas, gt, labels = frame_inputs(golden_dir, 'tiny')
    nhwc = [ops.to_nhwc(f) for f in feats]
    l2i = ops.lidar2img_tensor(metas, dev())
    img_hw = metas[0]['img_shape'][0][:2]
    tokens, pad_mult = h.radar_tokens(metas, dev())
    tr = FusionTrainer(h, dropout=0.1, seed=4, decoder_dropout=0.1, prefetch_depth=9)
    assert tr.decoder_tile_rows == 32                          # 9 x 900 rows > 4 096
    out = {}
    for mp, seed in (('f16x2', 77), ('f32', 77), ('f16x2', 78)):
        tr.decoder_matrix_path = mp
        tr.decoder_tile_rows = rows if mp == 'f16x2' else 16
        aux = tr._decoder_forward(nhwc, l2i, img_hw, tokens, pad_mult, seed, 0)['aux']
        torch.cuda.synchronize()
        out[(mp, seed)] = {k: v.clone() for k, v in aux.items() if torch.is_tensor(v)}
    tr.decoder_matrix_path = None
    a, b, c = out[('f16x2', 77)], out[('f32', 77)], out[('f16x2', 78)]
    for k in ('inter_states', 'inter_references', 'last_box'):
        assert torch.isfinite(a[k]).all()
        assert float((a[k] - b[k]).abs().max()) < 5e-4, (k, float((a[k] - b[k]).abs().max()))
    assert float((a['inter_states'] - c['inter_states']).abs().max()) > 1e-2      # another seed: other masks


def test_a_non_finite_cost_matrix_sends_no_gradient_and_raises_as_scipy_does(A, golden_dir):
    """ADVICE r4 / VERDICT r4 weak 7: the reference's assigner hands the cost matrix to scipy, which raises ValueError on
    a non-finite entry (ASSIGN:117-125) -- training stops.  The device assignment cannot raise inside a launch: it
    poisons the losses of the affected outputs (NaN -> the backward's guard: no gradient, loss reported as 0, exactly
    HEAD:915-916's treatment of a NaN loss) and FusionTrainer raises the ValueError when the status word has arrived,
    at the latest with ``check_assign_status(head, wait=True)``."""
    from transcar_amd import ops
    from transcar_amd.device_loss import check_assign_status
    from transcar_amd.trainer import FusionTrainer
    h = train_head(golden_dir)
    feats, metas, gt, labels = frame_inputs(golden_dir)
    nhwc = [ops.to_nhwc(f) for f in feats]
    l2i = ops.lidar2img_tensor(metas, dev())
    tokens, pad_mult = h.radar_tokens(metas, dev())
    hw = metas[0]['img_shape'][0][:2]
    tr = FusionTrainer(h, dropout=0.0)
    good = tr.step_fused_nhwc(nhwc, l2i, hw, tokens, pad_mult, [gt], [labels], update=False)
    assert float(tr.bucket.grads.abs().max()) > 0 and all(np.isfinite(float(v)) for v in good.values())
    check_assign_status(h, wait=True)                       # nothing to report
    bad_gt = gt.clone()
    bad_gt[3, 0] = float('inf')                             # an L1 cost of inf: finite class logits, the loss would stay finite
    losses = tr.step_fused_nhwc(nhwc, l2i, hw, tokens, pad_mult, [bad_gt], [labels], update=False)
    torch.cuda.synchronize()
    assert all(float(v) == 0.0 for v in losses.values()), losses          # every output of the sample: NaN -> 0
    assert float(tr.bucket.grads.abs().max()) == 0.0                      # ... and no gradient reached the bucket
    with pytest.raises(ValueError, match='invalid numeric entries'):
        tr.step_fused_nhwc(nhwc, l2i, hw, tokens, pad_mult, [gt], [labels], update=False)
    check_assign_status(h, wait=True)                       # (cleared by the raise)
    # a bad cost matrix in the LAST iteration of an epoch: no next step would report it -- finish() / state_dict() do
    tr.step_fused_nhwc(nhwc, l2i, hw, tokens, pad_mult, [bad_gt], [labels], update=False)
    with pytest.raises(ValueError, match='invalid numeric entries'):
        tr.state_dict()
    assert set(tr.state_dict()) == {'head', 'm', 'v', 'iter'}            # (cleared: the checkpoint is written now)


@pytest.mark.parametrize('tag', ['tiny', 'res101'])
def test_weight_gradients_in_exchange_chunks_equal_the_grouped_launch(A, golden_dir, tag):
    """Round 6: tc_radar_train_bwd_fused_ex(flags bit 1) + tc_radar_train_bwd_weights(group 0..3) --
    fusion layer 3, 2, 1, radar encoders, one launch each, what a multi-rank iteration runs so that a chunk's all-reduce
    travels under the next chunk's launch -- add exactly the gradients of the single grouped launch (up to the float
    atomics' order); every chunk writes ONLY its own range of the bucket (the exchange of chunk g may start behind launch
    g); the chunk ranges tile the bucket in exchange order."""
    import ctypes as C
    from transcar_amd import _lib as L, ops
    from transcar_amd.trainer import FusionTrainer, exchange_chunk_of
    h = train_head(golden_dir)
    feats, metas, gt, labels = frame_inputs(golden_dir, tag)
    nhwc = [ops.to_nhwc(f) for f in feats]
    l2i = ops.lidar2img_tensor(metas, dev())
    tokens, pad_mult = h.radar_tokens(metas, dev())
    tr = FusionTrainer(h, dropout=0.1, seed=4)
    tr.keep_last = True
    tr.step_fused_nhwc(nhwc, l2i, metas[0]['img_shape'][0][:2], tokens, pad_mult, [gt], [labels], update=False)
    torch.cuda.synchronize()
    want = tr.bucket.grads.clone()
    assert float(want.abs().max()) > 0
    k, lib = tr._last, L.lib()
    rngs = tr.bucket.chunk_ranges
    assert len(rngs) == 4 and rngs[0][0] == 0 and rngs[-1][1] == tr.bucket.numel and all(a[1] == b[0] for a, b in zip(rngs, rngs[1:]))
    for n, off in zip(tr.bucket.names, tr.bucket.offsets):
        a, b = rngs[exchange_chunk_of(n)]
        assert a <= off < b, n
    tr.bucket.zero_grad()
    L.check(lib.tc_radar_train_bwd_fused_ex(
        C.byref(k['w']), C.byref(k['g']), k['hs_last'].data_ptr(), k['last_box'].data_ptr(), k['tokens'].data_ptr(),
        k['B'], k['T'], k['pad_mult'], k['all_box'].data_ptr(), k['d_cls'].data_ptr(), k['d_box'].data_ptr(),
        k['tape'].data_ptr(), k['tape'].numel(), tr._bws.data_ptr(), tr._bws.numel(), tr.dropout, k['seed'], None, None,
        2, tr._stream()), 'bwd (weights deferred)')
    torch.cuda.synchronize()
    before = tr.bucket.grads.clone()              # LayerNorm / bias-free parts the chain itself adds; no weight gradient yet
    for gi, (a, b) in enumerate(rngs):
        L.check(lib.tc_radar_train_bwd_weights(
            C.byref(k['w']), C.byref(k['g']), k['hs_last'].data_ptr(), k['tokens'].data_ptr(), k['B'], k['T'],
            k['tape'].data_ptr(), k['tape'].numel(), tr._bws.data_ptr(), tr._bws.numel(), gi, tr._stream()), 'weights %d' % gi)
        torch.cuda.synchronize()
        now = tr.bucket.grads.clone()
        changed = (now != before).nonzero().flatten()
        assert changed.numel() > 0 and int(changed.min()) >= a and int(changed.max()) < b, (gi, a, b, int(changed.min()), int(changed.max()))
        before = now
    scale = float(want.abs().max())
    assert float((before - want).abs().max()) <= 2e-6 * scale
    with pytest.raises(L.TransCARHipError):
        L.check(lib.tc_radar_train_bwd_weights(
            C.byref(k['w']), C.byref(k['g']), k['hs_last'].data_ptr(), k['tokens'].data_ptr(), k['B'], k['T'],
            k['tape'].data_ptr(), k['tape'].numel(), tr._bws.data_ptr(), tr._bws.numel(), 4, tr._stream()), 'weights 4')


def test_backward_chain_guards_non_finite_loss_gradients(A, golden_dir):
    """HEAD:915-916 zeroes a non-finite loss; its gradient must not reach the bucket either.  With the fused
    backward that guard lives INSIDE the backward chain (tc_radar_train_bwd_fused(layer_losses=...)): a level whose
    loss is NaN / inf sends nothing down and non-finite gradient elements count as 0 -- the same gradients as
    applying torch.where / nan_to_num in front of an unguarded call."""
    import ctypes as C
    from transcar_amd import _lib as L, ops
    from transcar_amd.trainer import FusionTrainer
    h = train_head(golden_dir)
    feats, metas, gt, labels = frame_inputs(golden_dir)
    nhwc = [ops.to_nhwc(f) for f in feats]
    l2i = ops.lidar2img_tensor(metas, dev())
    tokens, pad_mult = h.radar_tokens(metas, dev())
    tr = FusionTrainer(h, dropout=0.1, seed=4, decoder_dropout=0.0)
    tr.keep_last = True
    tr.step_fused_nhwc(nhwc, l2i, metas[0]['img_shape'][0][:2], tokens, pad_mult, [gt], [labels], update=False)
    k = tr._last
    lib = L.lib()
    d_cls, d_box = k['d_cls'].clone(), k['d_box'].clone()
    d_cls[2, 0, 5, 3] = float('nan')
    d_box[0, 0, 7, 1] = float('inf')
    losses = torch.tensor([[0.7, 1.1], [float('nan'), 0.9], [0.5, float('inf')]], dtype=torch.float32, device=dev())

    def run(dc, db, lv):
        tr.bucket.zero_grad()
        L.check(lib.tc_radar_train_bwd_fused(
            C.byref(k['w']), C.byref(k['g']), k['hs_last'].data_ptr(), k['last_box'].data_ptr(), k['tokens'].data_ptr(),
            k['B'], k['T'], k['pad_mult'], k['all_box'].data_ptr(), dc.data_ptr(), db.data_ptr(), k['tape'].data_ptr(),
            k['tape'].numel(), tr._bws.data_ptr(), tr._bws.numel(), tr.dropout, k['seed'],
            lv.data_ptr() if lv is not None else None, clean.data_ptr() if lv is not None else None,
            tr._stream()), 'bwd')
        torch.cuda.synchronize()
        return tr.bucket.grads.clone()
    clean = torch.full((3, 2), -7.0, dtype=torch.float32, device=dev())
    got = run(d_cls, d_box, losses)
    assert torch.equal(clean, losses.masked_fill(torch.isnan(losses), 0.0))      # HEAD:915-916: the loss dict's values
    fin = torch.isfinite(losses)
    gc = torch.where(fin[:, 0].view(3, 1, 1, 1), torch.nan_to_num(d_cls, nan=0.0, posinf=0.0, neginf=0.0), torch.zeros_like(d_cls))
    gb = torch.where(fin[:, 1].view(3, 1, 1, 1), torch.nan_to_num(d_box, nan=0.0, posinf=0.0, neginf=0.0), torch.zeros_like(d_box))
    want = run(gc.contiguous(), gb.contiguous(), None)
    assert torch.isfinite(got).all() and torch.isfinite(want).all()
    assert float(want.abs().max()) > 0
    d = (got - want).abs().max() / want.abs().max()
    assert float(d) < 1e-5, float(d)
