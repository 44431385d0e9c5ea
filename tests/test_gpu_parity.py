"""Parity of the HIP path (through the C ABI) against the CPU oracle and the
reference-generated golden fixtures.  Needs a real MI355X:  pytest -m gpu

Tolerances (fp32 everywhere; stated per test):
  * single operators on identical inputs: 2e-5 .. 1e-4 absolute on O(1) values;
  * end to end (6 decoder + 3 radar layers): 1e-3 on box codes / logits, the
    tolerance BASELINE.json's north_star states, on queries whose radar gate
    decisions agree (the gate is discontinuous).
"""
import os

import numpy as np
import pytest
import torch

from oracle import transcar_oracle as O
from parity_util import assert_rows_match
from transcar_amd import configs, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _no_grad():
    with torch.no_grad():
        yield

PCR = configs.point_cloud_range
HW = configs.IMG_SHAPE[:2]
SMOOTH = (4, 6)
E2E_TOL = 1e-3


def dev():
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def T():
    import transcar_amd
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    transcar_amd.lib()                      # fail loudly if not built
    return transcar_amd


@pytest.fixture(scope='module')
def sd_np():
    return synth.make_state_dict(seed=3)


@pytest.fixture(scope='module')
def sd(sd_np):
    return O.to_torch_sd(sd_np)


@pytest.fixture(scope='module')
def head(T, sd_np):
    h = T.build_head(configs.head_cfg())
    h.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()},
                      strict=True)
    return h.to(dev()).eval()


def g(name):
    return np.load(os.path.join(os.path.dirname(__file__), 'golden', name))


def gpu(x):
    return torch.as_tensor(x).float().contiguous().to(dev())


# --------------------------------------------------------------------------
# single operators
# --------------------------------------------------------------------------
@pytest.mark.parametrize('M,K,N,act', [(900, 256, 256, 0), (900, 256, 512, 1),
                                       (900, 512, 256, 0), (255, 36, 64, 1),
                                       (900, 256, 10, 0), (900, 256, 24, 0),
                                       (7, 256, 768, 0), (1, 128, 256, 2)])
def test_linear(T, M, K, N, act):
    rng = np.random.RandomState(M + K + N)
    x = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32))
    x2 = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(N).astype(np.float32))
    res = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32))
    ref = torch.nn.functional.linear((x + x2).double(), w.double(), b.double())
    ref = ref.relu() if act == 1 else ref.sigmoid() if act == 2 else ref
    ref = ref + res.double()
    y = T.ops.linear(gpu(x), gpu(w), gpu(b), x2=gpu(x2), res=gpu(res), act=act)
    np.testing.assert_allclose(y.cpu().numpy(), ref.float().numpy(), atol=2e-5, rtol=1e-5)


def test_add_layernorm(T):
    rng = np.random.RandomState(5)
    a = torch.from_numpy(rng.standard_normal((901, 256)).astype(np.float32)) * 3
    b = torch.from_numpy(rng.standard_normal((901, 256)).astype(np.float32))
    gm = torch.from_numpy(rng.standard_normal(256).astype(np.float32))
    bt = torch.from_numpy(rng.standard_normal(256).astype(np.float32))
    ref = torch.nn.functional.layer_norm(a + b, (256,), gm, bt, 1e-5).relu()
    y = T.ops.add_layernorm(gpu(a), gpu(b), gpu(gm), gpu(bt), relu=True)
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), atol=2e-5, rtol=0)


def test_nchw_to_nhwc_roundtrip(T):
    rng = np.random.RandomState(6)
    x = torch.from_numpy(rng.standard_normal((3, 256, 29, 50)).astype(np.float32))
    y = T.ops.to_nhwc(gpu(x))
    assert torch.equal(y.cpu(), x.permute(0, 2, 3, 1).contiguous())   # bit exact
    # a channels_last tensor is taken as is (zero copy)
    xc = gpu(x).contiguous(memory_format=torch.channels_last)
    yc = T.ops.to_nhwc(xc)
    assert yc.data_ptr() == xc.data_ptr() and torch.equal(yc.cpu(), y.cpu())


def test_nchw_to_nhwc_levels_one_launch(T):
    """All levels of a frame in one launch (tc_nchw_to_nhwc_levels): the 16-byte path (H*W % 4 == 0),
    the scalar path (29x50, 15x25), channel counts that are not multiples of 64 / 4, ragged tiles."""
    rng = np.random.RandomState(8)
    for C_, shapes in ((256, [(116, 200), (58, 100), (29, 50), (15, 25)]), (40, [(8, 12), (5, 7)]),
                       (7, [(4, 8), (3, 3)])):
        xs = [torch.from_numpy(rng.standard_normal((2, 3, C_, h, w)).astype(np.float32)) for h, w in shapes]
        ys = T.ops.to_nhwc_levels([gpu(x) for x in xs])
        for x, y in zip(xs, ys):
            assert torch.equal(y.cpu(), x.reshape(-1, *x.shape[2:]).permute(0, 2, 3, 1).contiguous())
    # into preallocated outputs (a pipeline lane's static inputs)
    outs = [torch.zeros_like(y) for y in ys]
    got = T.ops.to_nhwc_levels([gpu(x) for x in xs], out=outs)
    assert all(g_.data_ptr() == o_.data_ptr() and torch.equal(g_, y) for g_, o_, y in zip(got, outs, ys))


class _StockFPN(torch.nn.Module):
    """A plain torch.nn FPN with the reference's settings (CFG:43-50: start_level=1,
    add_extra_convs='on_output', num_outs=4, relu_before_extra_convs): laterals on C3..C5,
    3x3 output convs, one stride-2 extra conv on the last output."""

    def __init__(self, in_channels=(64, 128, 256), out_channels=256):
        super().__init__()
        nn = torch.nn
        self.lat = nn.ModuleList([nn.Conv2d(c, out_channels, 1) for c in in_channels])
        self.out = nn.ModuleList([nn.Conv2d(out_channels, out_channels, 3, padding=1) for _ in in_channels])
        self.extra = nn.Conv2d(out_channels, out_channels, 3, stride=2, padding=1)

    def forward(self, xs):
        lat = [l(x) for l, x in zip(self.lat, xs)]
        for i in range(len(lat) - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + torch.nn.functional.interpolate(lat[i], size=lat[i - 1].shape[2:], mode='nearest')
        outs = [o(x) for o, x in zip(self.out, lat)]
        outs.append(self.extra(torch.relu(outs[-1])))
        return outs


def test_channels_last_fpn_feeds_the_head_zero_copy(T, head):
    """a stock torch.nn FPN run in ``channels_last`` (MIOpen NHWC convolutions) hands
    the head its maps without any transposition -- ops.to_nhwc_levels returns views of the very
    same storage -- and the head's output equals, bit for bit, the reference hand-off (NCHW maps,
    DET:62-66) of the same values through tc_nchw_to_nhwc_levels."""
    torch.manual_seed(0)
    fpn = _StockFPN().to(dev()).eval().to(memory_format=torch.channels_last)
    N = 6
    sizes = [(32, 48), (16, 24), (8, 12)]                      # C3..C5 of a small image
    xs = [torch.randn(N, c, h, w, device=dev()).contiguous(memory_format=torch.channels_last)
          for c, (h, w) in zip((64, 128, 256), sizes)]
    feats_cl = fpn(xs)                                           # 4 levels [N,256,H,W]
    assert all(f.is_contiguous(memory_format=torch.channels_last) and not f.is_contiguous() for f in feats_cl), \
        'the FPN did not stay in channels_last'
    nhwc = T.ops.to_nhwc_levels([f[None] for f in feats_cl])
    for f, v in zip(feats_cl, nhwc):
        assert v.data_ptr() == f.data_ptr() and v.shape == (N, f.shape[2], f.shape[3], 256)   # zero copy
    frame = synth.make_radar_frame(seed=5, n_per_radar=20)
    metas = synth.make_img_metas(1, radar=frame)
    out_cl = head([f[None] for f in feats_cl], metas)
    feats_nchw = [f.contiguous()[None] for f in feats_cl]        # what the reference's FPN returns
    assert all(f.is_contiguous() for f in feats_nchw)
    out_nchw = head(feats_nchw, metas)
    for k in ('all_cls_scores', 'all_bbox_preds'):
        assert torch.equal(out_cl[k], out_nchw[k]), k


@pytest.mark.parametrize('shapes,smooth,atol', [
    ('tiny', None, 3e-5), ('res101', SMOOTH, 1e-4),
    # BASELINE.json configs[4]: VoVNet FPN levels (232x400 ... 29x50), 757 MB of maps
    ('vovnet', SMOOTH, 1e-4),
    # iid-noise maps at stride 8: the fp32 rounding of the projected pixel
    # coordinate (~1e-4 px at |u| ~ 1600) times an O(1)/px feature gradient
    ('res101', None, 1e-3)])
def test_cam_sample_vs_oracle(T, shapes, smooth, atol):
    """feature_sampling + weighting (XFMR:365-373) on identical inputs."""
    rng = np.random.RandomState(31)
    feats = synth.make_feats(shapes, seed=32, smooth=smooth)
    l2i = synth.make_lidar2img()
    Q = 900
    ref = rng.uniform(0, 1, (1, Q, 3)).astype(np.float32)
    ref[0, 0] = [0.5, 0.5, 0.5]
    logits = rng.standard_normal((1, Q, 24)).astype(np.float32)
    tf = [torch.from_numpy(f) for f in feats]
    l2i_t = torch.from_numpy(l2i).float()[None]
    sampled, mask = O.feature_sampling(tf, torch.from_numpy(ref), PCR, l2i_t, HW)
    aw = torch.from_numpy(logits).view(1, 1, Q, 6, 1, 4).sigmoid() * mask
    want = (sampled * aw).sum(-1).sum(-1).sum(-1).permute(0, 2, 1)     # [B,Q,C]
    nhwc = [T.ops.to_nhwc(gpu(f)) for f in feats]
    got, vis = T.ops.cam_sample_fuse(nhwc, gpu(l2i_t), gpu(ref), gpu(logits),
                                     PCR, HW, return_mask=True)
    want_vis = mask[0, 0, :, :, 0, 0].numpy()
    flips = (vis[0].cpu().numpy().astype(bool) != want_vis).any(1)
    assert flips.sum() <= 1, 'visibility mask differs on %d queries' % flips.sum()
    ok = ~flips
    np.testing.assert_allclose(got[0].cpu().numpy()[ok], want[0].numpy()[ok],
                               atol=atol, rtol=1e-5)


def test_cam_sample_nan_and_linearity(T):
    """XFMR:367 NaN->0, and the op is linear in the feature maps."""
    rng = np.random.RandomState(33)
    feats = synth.make_feats('tiny', seed=34)
    l2i = gpu(synth.make_lidar2img())[None]
    ref = gpu(rng.uniform(0.05, 0.95, (1, 900, 3)))
    logits = gpu(rng.standard_normal((1, 900, 24)))
    a = [T.ops.to_nhwc(gpu(f)) for f in feats]
    b = [T.ops.to_nhwc(gpu(f)) for f in synth.make_feats('tiny', seed=35)]
    fa = T.ops.cam_sample_fuse(a, l2i, ref, logits, PCR, HW)
    fb = T.ops.cam_sample_fuse(b, l2i, ref, logits, PCR, HW)
    fab = T.ops.cam_sample_fuse([x + 2 * y for x, y in zip(a, b)], l2i, ref,
                                logits, PCR, HW)
    np.testing.assert_allclose(fab.cpu().numpy(), (fa + 2 * fb).cpu().numpy(),
                               atol=1e-4, rtol=1e-5)
    nan = [x.clone() for x in a]
    nan[3][:] = float('nan')           # coarsest level entirely NaN
    fn = T.ops.cam_sample_fuse(nan, l2i, ref, logits, PCR, HW)
    assert torch.isfinite(fn).all()
    zero = [x.clone() for x in a]
    zero[3][:] = 0
    fz = T.ops.cam_sample_fuse(zero, l2i, ref, logits, PCR, HW)
    assert torch.equal(fn, fz)


def test_self_attn_vs_oracle(T, sd, head):
    rng = np.random.RandomState(41)
    x = torch.from_numpy(rng.standard_normal((900, 2, 256)).astype(np.float32))
    pos = torch.from_numpy(rng.standard_normal((900, 2, 256)).astype(np.float32))
    name = 'transformer.decoder.layers.1.attentions.0.attn'
    want = x + O.multihead_attention(sd, name, x + pos, x + pos, x)
    mod = head.transformer.decoder.layers[1].attentions[0]
    got = mod(gpu(x), query_pos=gpu(pos))
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=3e-5, rtol=1e-5)


def test_cross_atten_golden(T, head):
    """Detr3DCrossAtten.forward against the reference's own output (G2)."""
    gold = g('g2_cross_atten.npz')
    rng = np.random.RandomState(21)
    feats = [gpu(f) for f in synth.make_feats('tiny', seed=22)]
    metas = synth.make_img_metas(1)
    query = gpu(rng.standard_normal((900, 1, 256)))
    qpos = gpu(rng.standard_normal((900, 1, 256)))
    refp = gpu(rng.uniform(0.02, 0.98, (1, 900, 3)))
    attn = head.transformer.decoder.layers[2].attentions[1]
    out = attn(query, None, feats, query_pos=qpos, reference_points=refp,
               img_metas=metas)
    np.testing.assert_allclose(out.cpu().numpy()[::4], gold['out'], atol=5e-5, rtol=1e-5)


def _radar_inputs(tag):
    gold = g('g5_head_%s.npz' % tag)
    frame = synth.make_radar_frame(seed=2, n_per_radar=51,
                                   centres=gold['radar_centres'])
    return gold, frame


def test_radar_xattn_teacher_forced(T, sd, head):
    """One gated attention step on the oracle's own inputs (HEAD:549-581)."""
    gold, frame = _radar_inputs('tiny')
    feats = [torch.from_numpy(f) for f in synth.make_feats('tiny', seed=1, smooth=SMOOTH)]
    l2i = torch.from_numpy(synth.make_lidar2img()).float()[None]
    f36 = O.build_radar_features(frame)
    outs, dbg = O.head_forward(sd, feats, l2i, HW, f36, PCR, return_debug=True)
    tokens, fill_in = O.radar_tokens_from_features(f36)
    query = dbg['hs'][-1]                                   # [1,Q,C]
    ref = dbg['inter_refs'][-1]
    cxy = torch.stack([ref[..., 0] * (PCR[3] - PCR[0]) + PCR[0],
                       ref[..., 1] * (PCR[4] - PCR[1]) + PCR[1]], -1)
    box = dbg['tmp']
    radar_feat = dbg['radar_feat'].permute(1, 0, 2)         # [1,K,C]
    mask = O.circle_mask(cxy, box[..., 3], box[..., 6], box[..., 7],
                         tokens[:, :, :2], 1.0, 2.0)
    want_hits = (~mask).sum(1).numpy()
    rows = torch.where((~mask).any(1))[0]
    want = query[0].clone()
    tgt = O.multihead_attention(sd, 'rf_multihead_attn', query[0][rows][:, None],
                                radar_feat.permute(1, 0, 2), radar_feat.permute(1, 0, 2),
                                attn_mask=mask[rows])
    want[rows] += tgt[:, 0]
    # the HIP path sees T = fill_in + pad tokens, the tail folded into pad_mult
    from transcar_amd import radar as R
    tok_np, pad_mult = R.pack_tokens([f36])
    Tn = tok_np.shape[1]
    got, hits = T.ops.radar_gated_xattn(
        T.bricks.mha_view(head.rf_multihead_attn), gpu(query), gpu(cxy), gpu(box),
        gpu(radar_feat[:, :Tn]), gpu(tokens[:, :Tn, :2]), pad_mult, 1.0, 2.0)
    hits = hits[0].cpu().numpy()
    same = hits == want_hits
    assert (~same).sum() <= 2, 'gate decisions differ on %d queries' % (~same).sum()
    np.testing.assert_allclose(got[0].cpu().numpy()[same], want.numpy()[same],
                               atol=5e-5, rtol=1e-5)
    assert int(rows.numel()) == int(gold['Lq'][0])


@pytest.mark.parametrize('case', ['g5', 'ragged', 'empty', 'out_of_range', 'overflow'])
def test_radar_ingest_on_device(T, case):
    """tc_radar_build_tokens: raw devkit rows -> [T,36] tokens in one launch, against
    (i) the rows the REFERENCE built from the same raw frame (fixtures G5 / G4: HEAD:301-521 run
    unmodified by tests/golden/make_golden.py) and (ii) the host builder transcar_amd/radar.py.
    Columns without arithmetic (coordinates, ids, states, one-hots, time offsets) must be bit
    equal; the six rotated-velocity columns are float64 products rounded to float32 once on either
    side: at most 1 ulp(float32) apart."""
    from transcar_amd import radar as R
    gold_rows = None
    if case == 'g5':
        gold, frame = _radar_inputs('tiny')
        gold_rows = gold['radar_tokens']
    elif case == 'ragged':
        frame = synth.make_radar_frame(seed=5, n_per_radar=[7, 0, 3, 0, 12])
        gold_rows = g('g4_radar_ragged.npz')['radar_tokens']
    elif case == 'empty':
        frame = synth.make_radar_frame(seed=5, n_per_radar=[0, 0, 0, 0, 0])
    elif case == 'out_of_range':
        frame = synth.make_radar_frame(seed=9, n_per_radar=[40, 13, 0, 64, 70])
        rng = np.random.RandomState(3)
        for chan in R.RADAR_CHANNELS:                       # push a third of the points outside HEAD:304's box
            p = frame['points'][chan]
            if p.shape[1]:
                out = rng.rand(p.shape[1]) < 0.33
                p[rng.randint(0, 3), out] = 77.7
        frame['points'][R.RADAR_CHANNELS[0]][0, 0] = 51.2    # exactly on the bound: strict inequality drops it
    else:
        frame = synth.make_radar_frame(seed=4, n_per_radar=[90, 80, 70, 60, 50])    # 350 points > T - 1
    Tn = 256
    want_rows = R.build_radar_features(frame)
    tokens, count, pad_mult = T.ops.radar_build_tokens(frame, Tn, dev())
    assert pad_mult == 1500 - Tn + 1 and tokens.shape == (1, Tn, 36)
    n = int(count.item())
    assert n == want_rows.shape[0]
    got = tokens[0].cpu().numpy()
    k = min(n, Tn - 1)                                      # T < 1500: row T - 1 always stays a pad row (pad_mult)
    want = np.full((Tn, 36), 500.0, np.float32)
    want[:k] = want_rows[:k].astype(np.float32)
    exact = [c for c in range(36) if c not in (9, 10, 11, 12, 13, 14)]
    np.testing.assert_array_equal(got[:, exact], want[:, exact])
    vel = [9, 10, 11, 12, 13, 14]
    ulp = np.spacing(np.abs(want[:, vel]).astype(np.float32))
    assert np.all(np.abs(got[:, vel] - want[:, vel]) <= ulp)
    if gold_rows is not None:                               # the reference's own rows
        assert gold_rows.shape[0] == n
        np.testing.assert_array_equal(got[:n, exact], gold_rows[:, exact].astype(np.float32))
        assert np.all(np.abs(got[:n, vel] - gold_rows[:, vel]) <= np.spacing(np.abs(gold_rows[:, vel]).astype(np.float32)))
    if case == 'overflow':
        assert n > Tn - 1                                   # the caller sees that the frame did not fit
        np.testing.assert_array_equal(got[Tn - 1], np.full(36, 500.0, np.float32))
        with pytest.raises(T.TransCARHipError):             # ... and the checked form refuses it
            T.ops.radar_build_tokens(frame, Tn, dev(), check=True)
    elif case != 'empty':
        # and the head consumes them: same result as the host-built tokens
        tok_np, pm = R.pack_tokens([want_rows], T=Tn)
        assert pm == pad_mult
        np.testing.assert_array_equal(got[:, exact], tok_np[0][:, exact])


def test_radar_pad_token_multiplicity(T, head):
    """A query parked on the pad location (500,500) must see all pad tokens:
    folding them into one token with multiplicity equals materialising them."""
    rng = np.random.RandomState(9)
    Q, C = 64, 256
    query = gpu(rng.standard_normal((1, Q, C)))
    cxy = gpu(np.tile(np.array([[500.0, 500.0]], np.float32), (Q, 1)))[None]
    box = gpu(rng.standard_normal((1, Q, 10)) * 0.1)
    mha = T.bricks.mha_view(head.rf_multihead_attn2)

    def run(Tn, pad_mult):
        feat = gpu(np.tile(rng.standard_normal((1, 1, C)).astype(np.float32) * 0 + 0.37, (1, Tn, 1)))
        feat[0, :5] = gpu(np.arange(5 * C).reshape(5, C) % 7 * 0.1)
        xy = gpu(np.full((1, Tn, 2), 500.0, np.float32))
        xy[0, :5] = gpu(np.array([[500.5, 500.0]] * 5, np.float32))
        return T.ops.radar_gated_xattn(mha, query, cxy, box, feat, xy, pad_mult, 1.0, 2.0)
    full, hits_full = run(1500, 1)
    folded, hits_fold = run(64, 1500 - 64 + 1)
    assert torch.equal(hits_full, hits_fold) and int(hits_full[0, 0]) == 1500
    np.testing.assert_allclose(folded.cpu().numpy(), full.cpu().numpy(), atol=2e-5, rtol=1e-5)


# --------------------------------------------------------------------------
# end to end
# --------------------------------------------------------------------------
def _e2e_check(outs, want_cls, want_box, want_hits, aux, tol=E2E_TOL):
    hits = aux['radar_hit_counts'][:, 0].cpu().numpy()          # [3,Q]
    agree = np.all(hits == want_hits, axis=0)
    n_flip = int((~agree).sum())
    assert n_flip <= 6, 'radar gate decisions differ on %d queries' % n_flip
    cls = outs['all_cls_scores'][:, 0].cpu().numpy()
    box = outs['all_bbox_preds'][:, 0].cpu().numpy()
    np.testing.assert_allclose(cls[:, agree], want_cls[:, agree], atol=tol, rtol=0)
    np.testing.assert_allclose(box[:, agree], want_box[:, agree], atol=tol, rtol=0)
    return n_flip


# the paths a frame can take through the library: B = 1 with the automatic choice (4-row tiles, exact-fp32 MFMA), and the
# TIMED path of bench.py / FramePipeline / the trainer's look-ahead (two-plane f16 operands on the matrix cores) at every
# tile height it exists for, as a frame alone and as frame 1 of a three-frame launch (its neighbours: other seeds)
E2E_PATHS = ['auto', 'f16x2-16', 'f16x2-16-mid3', 'f16x2-32', 'f16x2-32-mid3']


def _run_head_path(head, feats_np, l2i, frame, path, tag):
    """-> outputs of the tested frame shaped as a B = 1 forward ([3,1,Q,10] ...)."""
    from transcar_amd.detr3d_head import head_options
    if path == 'auto':
        return head([gpu(f) for f in feats_np], synth.make_img_metas(1, l2i, radar=frame), aux=True)
    rows = int(path.split('-')[1])
    head.forward_options = head_options(tile_rows=rows, matrix_path='f16x2')
    try:
        if not path.endswith('mid3'):
            return head([gpu(f) for f in feats_np], synth.make_img_metas(1, l2i, radar=frame), aux=True)
        frames = [synth.make_radar_frame(seed=21, n_per_radar=40), frame, synth.make_radar_frame(seed=22, n_per_radar=51)]
        if tag == 'vovnet':
            # three VoVNet frames are 2.3 GB of seeded maps: the neighbours are made ON THE DEVICE from the tested frame's
            # own maps (shifted / mirrored and re-scaled: other frames as far as the tested one is concerned)
            feats = []
            for f in feats_np:
                mid = gpu(f)
                feats.append(torch.cat([torch.roll(mid, shifts=(3, 5), dims=(-2, -1)) * 0.9, mid,
                                        torch.flip(mid, dims=(-1,)) * 1.1], 0).contiguous())
                del mid
        else:
            others = [synth.make_feats(tag, seed=s, smooth=SMOOTH) for s in (11, 12)]
            feats = [gpu(np.concatenate([others[0][l], feats_np[l], others[1][l]], 0)) for l in range(len(feats_np))]
        metas = synth.make_img_metas(3, l2i, radar=frames)
        outs = head(feats, metas, aux=True)
    finally:
        head.forward_options = None
    aux = outs['aux']
    one = {k: (v[:, 1:2] if v is not None else None) for k, v in outs.items() if k != 'aux'}
    one['aux'] = dict(inter_states=aux['inter_states'][:, 1:2], init_reference=aux['init_reference'][1:2],
                      inter_references=aux['inter_references'][:, 1:2], radar_hit_counts=aux['radar_hit_counts'][:, 1:2],
                      last_box=aux['last_box'][1:2], sample_pairs=aux['sample_pairs'])
    return one


@pytest.mark.parametrize('path', E2E_PATHS)
@pytest.mark.parametrize('tag', ['tiny', 'res101', 'vovnet'])
def test_head_end_to_end(T, sd, head, tag, path):
    """Detr3DHead.forward: HIP vs the CPU oracle AND vs the reference's own
    outputs (golden G5), same seeded inputs -- FREE-RUNNING through all nine layers on every path."""
    gold, frame = _radar_inputs(tag)
    feats_np = synth.make_feats(tag, seed=1, smooth=SMOOTH)
    l2i = synth.make_lidar2img()
    outs = _run_head_path(head, feats_np, l2i, frame, path, tag)
    aux = outs['aux']
    # --- vs oracle (run here, on the host CPU)
    feats = [torch.from_numpy(f) for f in feats_np]
    f36 = O.build_radar_features(frame)
    want, dbg = O.head_forward(sd, feats, torch.from_numpy(l2i).float()[None], HW,
                               f36, PCR, return_debug=True)
    np.testing.assert_allclose(aux['init_reference'].cpu().numpy(),
                               dbg['init_ref'].numpy(), atol=1e-6, rtol=0)
    np.testing.assert_allclose(aux['inter_references'].cpu().numpy(),
                               dbg['inter_refs'].numpy(), atol=5e-5, rtol=0)
    np.testing.assert_allclose(aux['inter_states'].cpu().numpy(),
                               dbg['hs'].numpy(), atol=E2E_TOL, rtol=0)
    want_hits = np.stack([h.numpy() for h in dbg['hit_counts']])
    _e2e_check(outs, want['all_cls_scores'][:, 0].numpy(),
               want['all_bbox_preds'][:, 0].numpy(), want_hits, aux)
    # --- vs the reference itself (fixture)
    gh = np.zeros((3, 900), np.int64)
    # the fixture stores hit counts of the selected rows only; rebuild [3,Q]
    for i in range(3):
        rows = np.where(want_hits[i] > 0)[0]
        if len(rows) == int(gold['Lq'][i]):
            gh[i, rows] = gold['hit_counts%d' % i]
        else:
            gh[i] = want_hits[i]
    _e2e_check(outs, gold['all_cls_scores'][:, 0], gold['all_bbox_preds'][:, 0], gh, aux)
    np.testing.assert_allclose(aux['inter_references'].cpu().numpy(),
                               gold['inter_refs'], atol=5e-5, rtol=0)
    # algorithmic gather count: visible (query, cam) pairs over 6 layers
    nb = 3 if path.endswith('mid3') else 1
    assert 0 < int(aux['sample_pairs']) <= nb * 6 * 900 * 6


def test_head_end_to_end_vovnet_shapes(T, sd, head):
    """BASELINE.json configs[4]: the VoVNet FPN levels (232x400 ... 29x50; 757 MB of maps),
    whole head against the CPU oracle."""
    _, frame = _radar_inputs('res101')
    feats_np = synth.make_feats('vovnet', seed=4, smooth=SMOOTH)
    l2i = synth.make_lidar2img()
    outs = head([gpu(f) for f in feats_np], synth.make_img_metas(1, l2i, radar=frame), aux=True)
    aux = outs['aux']
    want, dbg = O.head_forward(sd, [torch.from_numpy(f) for f in feats_np],
                               torch.from_numpy(l2i).float()[None], HW,
                               O.build_radar_features(frame), PCR, return_debug=True)
    del feats_np
    np.testing.assert_allclose(aux['inter_references'].cpu().numpy(), dbg['inter_refs'].numpy(),
                               atol=5e-5, rtol=0)
    np.testing.assert_allclose(aux['inter_states'].cpu().numpy(), dbg['hs'].numpy(), atol=E2E_TOL, rtol=0)
    want_hits = np.stack([x.numpy() for x in dbg['hit_counts']])
    _e2e_check(outs, want['all_cls_scores'][:, 0].numpy(), want['all_bbox_preds'][:, 0].numpy(),
               want_hits, aux)


@pytest.mark.parametrize('nq', [777, 130])
def test_head_other_query_counts(T, nq):
    """num_query that is no multiple of the row tiles (4), the attention query tiles (32) or
    the key tiles: the tails of every kernel, against the CPU oracle."""
    sd_np = synth.make_state_dict(seed=5, num_query=nq)
    sd_t = O.to_torch_sd(sd_np)
    h = T.build_head(configs.head_cfg(num_query=nq))
    h.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    h = h.to(dev()).eval()
    _, frame = _radar_inputs('tiny')
    feats_np = synth.make_feats('tiny', seed=2, smooth=SMOOTH)
    l2i = synth.make_lidar2img()
    outs = h([gpu(f) for f in feats_np], synth.make_img_metas(1, l2i, radar=frame), aux=True)
    aux = outs['aux']
    want, dbg = O.head_forward(sd_t, [torch.from_numpy(f) for f in feats_np],
                               torch.from_numpy(l2i).float()[None], HW,
                               O.build_radar_features(frame), PCR, return_debug=True)
    assert outs['all_cls_scores'].shape == (3, 1, nq, 10)
    np.testing.assert_allclose(aux['inter_references'].cpu().numpy(), dbg['inter_refs'].numpy(),
                               atol=5e-5, rtol=0)
    np.testing.assert_allclose(aux['inter_states'].cpu().numpy(), dbg['hs'].numpy(), atol=E2E_TOL, rtol=0)
    want_hits = np.stack([x.numpy() for x in dbg['hit_counts']])
    _e2e_check(outs, want['all_cls_scores'][:, 0].numpy(), want['all_bbox_preds'][:, 0].numpy(),
               want_hits, aux)
    # two frames per step: a sample boundary inside a row tile and inside a transposed-V store
    feats2 = [gpu(np.concatenate([f, f], 0)) for f in feats_np]
    metas2 = synth.make_img_metas(2, l2i, radar=frame)
    outs2 = h(feats2, metas2)
    for k_ in ('all_cls_scores', 'all_bbox_preds'):
        a_ = outs2[k_].cpu().numpy()
        np.testing.assert_array_equal(a_[:, 0], a_[:, 1])
        np.testing.assert_allclose(a_[:, 0], outs[k_][:, 0].cpu().numpy(), atol=2e-4, rtol=0)   # 8- vs 4-row tiles
    # decode: top-k of nq * 10 scores
    boxes, scores, labels = h.get_bboxes(outs, synth.make_img_metas(1))[0]
    assert boxes.shape[0] == scores.shape[0] == labels.shape[0] <= 300
    s_ = scores.cpu().numpy()
    assert np.all(s_[:-1] >= s_[1:])


def test_head_ragged_and_empty_radar(T, sd, head):
    """Ragged (7/0/3/0/12 points per radar) and empty radar frames against the reference's own
    outputs (G4), compared hit-set aware: every query whose three gate decisions agree with the
    reference's must match within the end-to-end tolerance on every level; at most 2 queries may
    disagree on a gate decision (the gate is discontinuous)."""
    gold = g('g4_radar_ragged.npz')
    feats_np = synth.make_feats('tiny', seed=1, smooth=SMOOTH)
    feats = [torch.from_numpy(f) for f in feats_np]
    l2i_t = torch.from_numpy(synth.make_lidar2img()).float()[None]
    for n_per, check_gold in (([7, 0, 3, 0, 12], True), ([0, 0, 0, 0, 0], False)):
        frame = synth.make_radar_frame(seed=5, n_per_radar=n_per)
        metas = synth.make_img_metas(1, radar=frame)
        outs = head([gpu(f) for f in feats_np], metas, aux=True)
        assert torch.isfinite(outs['all_bbox_preds']).all() and torch.isfinite(outs['all_cls_scores']).all()
        # the reference's per-query hit counts are not in the fixture (only Lq): the oracle,
        # which equals the reference on this very fixture (tests/test_oracle_golden.py), gives them
        want, dbg = O.head_forward(sd, feats, l2i_t, HW, O.build_radar_features(frame), PCR,
                                   return_debug=True)
        want_hits = np.stack([h.numpy() for h in dbg['hit_counts']])
        if check_gold:
            assert [int((want_hits[i] > 0).sum()) for i in range(3)] == [int(v) for v in gold['Lq']]
            n_flip = _e2e_check(outs, gold['all_cls_scores'][:, 0], gold['all_bbox_preds'][:, 0],
                                want_hits, outs['aux'])
            assert n_flip <= 2
        else:
            assert int(outs['aux']['radar_hit_counts'].sum()) == 0 and int(want_hits.sum()) == 0
        _e2e_check(outs, want['all_cls_scores'][:, 0].numpy(), want['all_bbox_preds'][:, 0].numpy(),
                   want_hits, outs['aux'])


@pytest.mark.parametrize('nb', [2, 3])
def test_head_batch_equals_singles(T, head, nb):
    """Batch > 1 (not supported by the reference's radar part) = per-sample runs.
    nb = 2 runs the row chains on 8-row tiles, nb = 3 on 16-row tiles (B = 1: 4 rows).
    (i) At the batch's own tile height every sample is BIT-IDENTICAL to its single run; (ii) against the
    single run at ITS automatic height (4 rows: another summation order) the rows whose radar gates made
    the same decisions agree to 3e-4 and at most 2 queries per sample and layer decide differently."""
    from transcar_amd import ops
    from transcar_amd.detr3d_head import head_options
    l2i = synth.make_lidar2img()
    feats, frames = [], []
    for i in range(nb):
        f = synth.make_feats('tiny', seed=1 + 6 * i, smooth=SMOOTH)
        r = synth.make_radar_frame(seed=2, n_per_radar=51, centres=g('g5_head_tiny.npz')['radar_centres']) \
            if i == 0 else synth.make_radar_frame(seed=2 + i, n_per_radar=20 + 7 * i)
        feats.append(f)
        frames.append(r)
    both = [gpu(np.concatenate([f[l] for f in feats], 0)) for l in range(4)]
    metas = synth.make_img_metas(nb, l2i, radar=frames)
    ob = head(both, metas, aux=True)                         # the plugin entry, automatic tile height
    R = 8 if nb == 2 else 16
    nhwc = ops.to_nhwc_levels(both)
    l2i_t = ops.lidar2img_tensor(metas, dev())
    tokens, pm = head.radar_tokens(metas, dev())
    hw = metas[0]['img_shape'][0][:2]
    for i in range(nb):
        args = ([f[6 * i:6 * i + 6] for f in nhwc], l2i_t[i:i + 1], hw, tokens[i:i + 1], pm)
        same = head.forward_nhwc(*args, options=head_options(tile_rows=R))
        for k in ('all_cls_scores', 'all_bbox_preds'):
            assert torch.equal(same[k][:, 0], ob[k][:, i]), (k, i)
        own = head.forward_nhwc(*args, aux=True)             # 4-row tiles
        agree = (own['aux']['radar_hit_counts'][:, 0] == ob['aux']['radar_hit_counts'][:, i]).cpu().numpy()
        assert int((~agree).sum(1).max()) <= 2, (~agree).sum(1)
        ok = np.logical_and.accumulate(agree, 0)             # a flipped gate also changes the later layers' inputs
        for k in ('all_cls_scores', 'all_bbox_preds'):
            d = np.abs(own[k][:, 0].cpu().numpy() - ob[k][:, i].cpu().numpy()).max(-1)
            assert float(d[ok].max()) < 3e-4, (k, i, float(d[ok].max()))      # 16x16x4 vs 4x4x1 summation order, 9 layers


def test_module_api_matches_fused_head(T, head):
    """Detr3DTransformer.forward (operator-by-operator through the C ABI, the
    reference's module structure) = the fused tc_head_forward path."""
    feats = [gpu(f) for f in synth.make_feats('tiny', seed=1, smooth=SMOOTH)]
    frame = synth.make_radar_frame(seed=5, n_per_radar=[7, 0, 3, 0, 12])
    metas = synth.make_img_metas(1, radar=frame)
    outs = head(feats, metas, aux=True)
    hs, init_ref, inter_refs = head.transformer(
        feats, head.query_embedding.weight, reg_branches=head.reg_branches,
        img_metas=metas)
    np.testing.assert_allclose(init_ref.cpu().numpy(),
                               outs['aux']['init_reference'].cpu().numpy(), atol=1e-6)
    np.testing.assert_allclose(inter_refs.cpu().numpy(),
                               outs['aux']['inter_references'].cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(hs.permute(0, 2, 1, 3).cpu().numpy(),
                               outs['aux']['inter_states'].cpu().numpy(), atol=E2E_TOL)


def test_fused_chains_match_operator_path(T, head):
    """tc_head_forward's fused row-chain kernels (16 launches) against the same
    forward run operator by operator (TRANSCAR_UNFUSED=1, ~160 launches)."""
    gold, frame = _radar_inputs('res101')
    feats = [gpu(f) for f in synth.make_feats('res101', seed=1, smooth=SMOOTH)]
    metas = synth.make_img_metas(1, radar=frame)
    fused = head(feats, metas, aux=True)
    os.environ['TRANSCAR_UNFUSED'] = '1'
    try:
        plain = head(feats, metas, aux=True)
    finally:
        os.environ.pop('TRANSCAR_UNFUSED')
    np.testing.assert_allclose(fused['aux']['inter_references'].cpu().numpy(),
                               plain['aux']['inter_references'].cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(fused['aux']['inter_states'].cpu().numpy(),
                               plain['aux']['inter_states'].cpu().numpy(), atol=E2E_TOL)
    hf = fused['aux']['radar_hit_counts'][:, 0].cpu().numpy()
    hp = plain['aux']['radar_hit_counts'][:, 0].cpu().numpy()
    agree = np.all(hf == hp, axis=0)
    assert (~agree).sum() <= 4
    for k in ('all_cls_scores', 'all_bbox_preds'):
        np.testing.assert_allclose(fused[k][:, 0].cpu().numpy()[:, agree],
                                   plain[k][:, 0].cpu().numpy()[:, agree], atol=E2E_TOL)
    assert int(fused['aux']['sample_pairs']) == int(plain['aux']['sample_pairs'])


def test_fused_sampling_with_many_visible_cameras(T, head):
    """Queries seen by 3 and by 6 cameras at once (cameras duplicated): the loop over the
    visible cameras of the fused decoder chain against the operator path."""
    _, frame = _radar_inputs('res101')
    feats = [gpu(f) for f in synth.make_feats('res101', seed=3, smooth=SMOOTH)]
    for dup in ((0, 0, 0, 3, 3, 3), (2, 2, 2, 2, 2, 2)):
        metas = synth.make_img_metas(1, radar=frame)
        l2i = metas[0]['lidar2img']
        metas[0]['lidar2img'] = [np.array(l2i[i], copy=True) for i in dup]
        fused = head(feats, metas, aux=True)
        os.environ['TRANSCAR_UNFUSED'] = '1'
        try:
            plain = head(feats, metas, aux=True)
        finally:
            os.environ.pop('TRANSCAR_UNFUSED')
        pairs = int(fused['aux']['sample_pairs'])
        assert pairs == int(plain['aux']['sample_pairs'])
        assert pairs % 3 == 0 and pairs > 0            # every visible query is seen by 3 (or 6) cameras
        # layer 0: identical inputs on both paths
        np.testing.assert_allclose(fused['aux']['inter_states'][0].cpu().numpy(),
                                   plain['aux']['inter_states'][0].cpu().numpy(), atol=2e-4)
        np.testing.assert_allclose(fused['aux']['inter_references'].cpu().numpy(),
                                   plain['aux']['inter_references'].cpu().numpy(), atol=1e-4)


def test_frames_in_flight_equal_sequential(T, head):
    """transcar_amd.pipeline.FramePipeline: three frames in flight on three HIP streams (one
    hipGraph, one workspace each) give bit for bit what one frame at a time gives."""
    import bench
    bench._imports()
    from transcar_amd.pipeline import FramePipeline
    lanes = [bench.make_inputs(head, dev(), 'tiny', 1, seed=11 + i) for i in range(3)]
    want = []
    for inp in lanes:
        outs, dec = bench.one_step(head, inp)
        want.append([outs['all_cls_scores'].clone(), outs['all_bbox_preds'].clone()] + [d.clone() for d in dec])
    torch.cuda.synchronize()
    pipe = FramePipeline(head, lanes)
    assert pipe.lanes == 3
    for rnd in range(4):                                   # 12 launches, round robin, no waits in between
        for _ in range(3):
            pipe.launch()
    pipe.synchronize()
    for i in range(3):
        outs, dec = pipe.outputs[i]
        got = [outs['all_cls_scores'], outs['all_bbox_preds']] + list(dec)
        for a_, b_ in zip(got, want[i]):
            assert torch.equal(a_, b_)
    # new data in a lane's static inputs -> new result from the same graph
    lanes[0]['tokens'].copy_(lanes[1]['tokens'])
    for f0, f1 in zip(lanes[0]['nhwc'], lanes[1]['nhwc']):
        f0.copy_(f1)
    lane, (outs, dec) = pipe.launch(0)
    pipe.wait(lane)
    assert torch.equal(outs['all_bbox_preds'], want[1][1])
    # 8-row tiles in this pipeline's graphs only (tc_head_options.chain_tile_rows): same results up to rounding
    pipe8 = FramePipeline(head, lanes[1:], tile_rows=8)
    lane, (outs8, _) = pipe8.launch(0)
    pipe8.wait(lane)
    np.testing.assert_allclose(outs8['all_bbox_preds'].cpu().numpy(), want[1][1].cpu().numpy(), atol=2e-4, rtol=0)
    # 16-row tiles (one weight buffer refilled in place, two workgroups per CU; v_mfma_f32_16x16x4 on the
    # P16 weight copy: another summation order over k than the 4x4x1 tiles', so rounding-level differences
    # go through the rig's loop gain, DESIGN.md section 3 -- measured 3e-4 on 1-3 of 27 000 values; against the
    # oracle the 16-row tiles hold the same per-layer tolerances as the others, test_gpu_teacher_forced.py)
    pipe16 = FramePipeline(head, lanes[1:], tile_rows=16)
    lane, (outs16, _) = pipe16.launch(0)
    pipe16.wait(lane)
    np.testing.assert_allclose(outs16['all_bbox_preds'].cpu().numpy(), want[1][1].cpu().numpy(), atol=5e-4, rtol=0)
    np.testing.assert_allclose(outs16['all_cls_scores'].cpu().numpy(), want[1][0].cpu().numpy(), atol=5e-4, rtol=0)
    outs4, _ = bench.one_step(head, lanes[1])                 # per-call option: nothing process-wide changed
    assert torch.equal(outs4['all_bbox_preds'], want[1][1])


def test_pipeline_pairs_frames_per_launch(T, head):
    """frames_per_launch = 2: the caller still submits one frame at a time (per-slot input writes);
    a lane is replayed when both of its slots are filled, flush() launches a half-filled lane as a
    PARTIAL launch over its one filled slot.  Every frame of a full launch is bit-identical to the same
    frame alone at the launch's tile height (8 rows); so is the odd last frame: with two lanes the partial launch
    keeps the full launch's tile height (a single-lane pipeline uses the one-frame launch's own 4 rows)."""
    import bench
    from transcar_amd.detr3d_head import head_options
    from transcar_amd.pipeline import FramePipeline
    nframes = 7
    frames = [bench.make_inputs(head, dev(), 'tiny', 1, seed=51 + i) for i in range(nframes)]
    want = []
    for f in frames:
        o8 = head.forward_nhwc(f['nhwc'], f['l2i'], f['hw'], f['tokens'], f['pad_mult'], options=head_options(tile_rows=8))
        o4, _ = bench.one_step(head, f)
        want.append(((o8['all_bbox_preds'].clone(), o8['all_cls_scores'].clone()),
                     (o4['all_bbox_preds'].clone(), o4['all_cls_scores'].clone())))
    lanes = [bench.make_inputs(head, dev(), 'tiny', 2, seed=71 + i) for i in range(2)]
    pipe = FramePipeline(head, lanes)
    assert pipe.frames_per_launch == 2 and pipe.lanes == 2
    got, where = {}, {}
    launches = 0
    for i, f in enumerate(frames):
        def write(p_, lane, slot, f=f):
            p_.write_inputs(lane, slot=slot, nhwc=f['nhwc'], l2i=f['l2i'], tokens=f['tokens'],
                            pad_mult=f['pad_mult'])
        lane, slot, launched = pipe.submit(write)
        where[i] = (lane, slot)
        launches += int(launched)
        if launched or i == nframes - 1:
            partial = not launched
            if partial:
                assert pipe.flush() == 1 and pipe.flush() == 0          # the odd last frame
                assert pipe.last_flush[:2] == (lane, 1)
                launches += 1
            outs, _ = pipe.last_flush[2] if partial else pipe.outputs[lane]
            with torch.cuda.stream(pipe.streams[lane]):
                for j in [k for k, (l_, _) in where.items() if l_ == lane and k not in got]:
                    s_ = where[j][1]
                    got[j] = (partial, outs['all_bbox_preds'][:, s_].clone(), outs['all_cls_scores'][:, s_].clone())
    pipe.synchronize()
    assert launches == 4 and len(got) == nframes
    for i in range(nframes):
        partial = got[i][0]
        assert partial == (i == nframes - 1)
        # two lanes: the partial launch keeps the full launch's 8-row tiles (FramePipeline.tile_rows_of)
        for a_, b_ in zip(got[i][1:], want[i][0]):
            assert torch.equal(a_, b_[:, 0]), i
    assert pipe.tile_rows_of() == 8 and pipe.tile_rows_of(1) == 8
    solo = FramePipeline(head, lanes[:1])
    assert solo.tile_rows_of(1) == 4                    # a single lane: the partial launch is alone on the chip
    # the full graph for a partly filled lane (partial_graphs=False): the filled slot's result is the same frame
    # at the full launch's tile height
    pipe2 = FramePipeline(head, lanes, partial_graphs=False)
    f = frames[0]
    pipe2.submit(lambda p_, lane, slot: p_.write_inputs(lane, slot=slot, nhwc=f['nhwc'], l2i=f['l2i'],
                                                        tokens=f['tokens'], pad_mult=f['pad_mult']))
    assert pipe2.flush() == 1
    pipe2.synchronize()
    assert torch.equal(pipe2.last_flush[2][0]['all_bbox_preds'][:, 0], want[0][0][0][:, 0])


def test_pipeline_producer_rewrites_lane_inputs(T, head):
    """The pipeline's ordering contract: a producer refills a lane's static inputs on
    the current stream from pinned host memory (H2D) between replays -- write_inputs waits for the
    lane's previous replay, launch makes the lane wait for the producer -- and every frame's result
    equals its sequential result bit for bit, with no host synchronisation inside the loop."""
    import bench
    from transcar_amd import radar as R
    from transcar_amd.pipeline import FramePipeline
    nframes, nl = 12, 3
    frames = [bench.make_inputs(head, dev(), 'tiny', 1, seed=31 + i) for i in range(nframes)]
    assert len({tuple(f['tokens'].shape) for f in frames}) == 1 and len({f['pad_mult'] for f in frames}) == 1
    want = []
    for f in frames:
        outs, dec = bench.one_step(head, f)
        want.append((outs['all_bbox_preds'].clone(), outs['all_cls_scores'].clone(), dec[0].clone()))
    host = [dict(nhwc=[x.cpu().pin_memory() for x in f['nhwc']], l2i=f['l2i'].cpu().pin_memory(),
                 tokens=f['tokens'].cpu().pin_memory()) for f in frames]
    torch.cuda.synchronize()
    # the lanes start with frame 0's data everywhere: every result must come from the refill
    lanes = [dict(nhwc=[x.clone() for x in frames[0]['nhwc']], l2i=frames[0]['l2i'].clone(), hw=frames[0]['hw'],
                  tokens=frames[0]['tokens'].clone(), pad_mult=frames[0]['pad_mult']) for _ in range(nl)]
    pipe = FramePipeline(head, lanes)
    got = [None] * nframes
    for rnd in range(3):                       # the same 12 frames three times over, no host sync
        for fidx in range(nframes):
            lane = fidx % nl
            pipe.write_inputs(lane, nhwc=host[fidx]['nhwc'], l2i=host[fidx]['l2i'],
                              tokens=host[fidx]['tokens'], pad_mult=frames[fidx]['pad_mult'])
            _, (outs, dec) = pipe.launch(lane)
            with torch.cuda.stream(pipe.streams[lane]):      # consume on the lane's stream, before its next replay
                got[fidx] = (outs['all_bbox_preds'].clone(), outs['all_cls_scores'].clone(), dec[0].clone())
    pipe.synchronize()
    torch.cuda.synchronize()
    for fidx in range(nframes):
        for a_, b_ in zip(got[fidx], want[fidx]):
            assert torch.equal(a_, b_), 'frame %d' % fidx
    # a frame packed to another token count is refused, not silently mis-read
    tok_np, pm = R.pack_tokens([frames[0]['radar_feats'][0]], T=320)
    with pytest.raises(T.TransCARHipError):
        pipe.write_inputs(0, tokens=torch.from_numpy(tok_np).to(dev()), pad_mult=pm)
    # reloading a checkpoint re-packs IN PLACE (same buffer, same generation): the graphs stay valid
    gen = head.buffers_generation
    head.load_state_dict(head.state_dict())
    head.head_weights()
    assert head.buffers_generation == gen
    lane, (outs, _) = pipe.launch(1)
    pipe.wait(lane)
    last = [f for f in range(nframes) if f % nl == 1][-1]
    assert torch.equal(outs['all_bbox_preds'], want[last][0])


def test_box_decode_vs_oracle(T, head):
    gold = g('g5_head_res101.npz')
    outs = {'all_cls_scores': gpu(gold['all_cls_scores']),
            'all_bbox_preds': gpu(gold['all_bbox_preds'])}
    got = head.get_bboxes(outs, synth.make_img_metas(1))[0]
    np.testing.assert_allclose(got[1].cpu().numpy(), gold['dec_scores'], atol=1e-6, rtol=0)
    mine = np.concatenate([got[0].cpu().numpy(), got[1].cpu().numpy()[:, None],
                           got[2].cpu().numpy()[:, None].astype(np.float32)], 1)
    want = np.concatenate([gold['dec_boxes'], gold['dec_scores'][:, None],
                           gold['dec_labels'][:, None].astype(np.float32)], 1)
    assert_rows_match(mine, want, atol=2e-5, what='decoded boxes')
    # scores come out sorted (torch.topk order)
    s = got[1].cpu().numpy()
    assert np.all(s[:-1] >= s[1:])


@pytest.mark.parametrize('case', ['ties', 'all_equal', 'random_batch', 'tiny_scores', 'saturated', 'extremes', 'few_candidates'])
def test_box_decode_ties_and_batches(T, case):
    """The radix select stops early when a whole bin is wanted; ties between equal
    scores (which keep it going through the index digits) must still give exactly
    max_num results: the top scores, ties resolved to the lower flat index."""
    from transcar_amd import ops
    rng = np.random.RandomState(7)
    B = 3 if case == 'random_batch' else 1
    cls = rng.standard_normal((B, 900, 10)).astype(np.float32)
    if case == 'ties':
        cls = np.round(cls * 2) / 2            # ~15 distinct values, 9000 keys
    elif case == 'all_equal':
        cls[:] = 0.25
    elif case == 'tiny_scores':                # every score below 2^-16: the select's lower catch-all bucket
        cls = (cls * 2 - 20).astype(np.float32)
    elif case == 'saturated':                  # scores 1 - 2^-24 .. 1 and ties at exactly 1.0: the upper catch-all
        cls = (np.abs(cls) * 3 + 12).astype(np.float32)
    elif case == 'extremes':                   # a few hundred saturated, the rest tiny, the 300th in between
        cls = (cls - 18).astype(np.float32)
        cls.reshape(-1)[rng.permutation(9000)[:170]] = 25.0
        cls.reshape(-1)[rng.permutation(9000)[:100]] = rng.standard_normal(100).astype(np.float32)
    Qn = 20 if case == 'few_candidates' else 900      # 200 candidates < max_num: rows 200.. are empty
    cls = np.ascontiguousarray(cls[:, :Qn])
    box = rng.standard_normal((B, Qn, 10)).astype(np.float32) * 0.3
    pcr = configs.pts_bbox_head['bbox_coder']['post_center_range']
    boxes, scores, labels, valid = ops.box_decode_topk(gpu(cls), gpu(box), pcr, 300)
    sg = 1.0 / (1.0 + np.exp(-cls.astype(np.float64)))
    for b in range(B):
        flat = sg[b].reshape(-1)
        order = np.lexsort((np.arange(flat.size), -flat))[:300]      # score desc, index asc
        kk = len(order)
        np.testing.assert_allclose(scores[b].cpu().numpy()[:kk], flat[order], atol=1e-6, rtol=0)
        if case in ('ties', 'all_equal'):     # (distinct logits may round to one fp32 sigmoid: order free)
            np.testing.assert_array_equal(labels[b].cpu().numpy(), order % 10)
            np.testing.assert_allclose(boxes[b].cpu().numpy()[:, 0], box[b][order // 10, 0], atol=1e-6)
        else:
            # the fp32 scores the kernel saw: ties between them resolve to the lower flat index
            s32 = (1.0 / (1.0 + np.exp(-cls[b].astype(np.float32).reshape(-1)))).astype(np.float32)
            got_s = scores[b].cpu().numpy()[:kk]
            assert np.all(got_s[:-1] >= got_s[1:])
            lab = labels[b].cpu().numpy()[:kk]
            bx = boxes[b].cpu().numpy()[:kk, 0]
            # each output row is a real (query, class) pair with that score, no pair twice
            cand = {}
            for i in np.argsort(-s32, kind='stable')[:kk + 200]:
                cand.setdefault((np.float32(box[b][i // 10, 0]).item(), int(i % 10)), []).append(i)
            seen = set()
            for r in range(kk):
                ids = [i for i in cand.get((np.float32(bx[r]).item(), int(lab[r])), []) if i not in seen and abs(s32[i] - got_s[r]) <= 1e-6]
                assert ids, (case, r)
                seen.add(ids[0])
        if kk < 300:
            assert np.all(labels[b].cpu().numpy()[kk:] == -1) and not valid[b].cpu().numpy()[kk:].any()
            assert np.all(scores[b].cpu().numpy()[kk:] == 0)
    # tc_box_decode_kept: what NMSFreeCoder.decode_single returns (CODER:62-84) -- the rows inside post_center_range
    # (and above the threshold, strictly) compacted in score order, their number, int64 labels; bit-identical to a
    # mask select over the fixed-size rows.  z_shift 0 = the coder's own gravity-centre z
    for thr in (None, 0.0, float(np.median(scores.cpu().numpy()))):
        for z_shift in (True, False):
            kb, ks, kl, kc = ops.box_decode_kept(gpu(cls), gpu(box), pcr, 300, score_threshold=thr, z_shift=z_shift)
            assert kl.dtype == torch.int64 and kc.dtype == torch.int32
            for b in range(B):
                m = valid[b].bool()
                if thr:
                    m = m & (scores[b] > thr)
                n = int(kc[b])
                assert n == int(m.sum())
                want = boxes[b][m].clone()
                if not z_shift:
                    want[:, 2] = boxes[b][m][:, 2] + boxes[b][m][:, 5] * 0.5
                    assert float((kb[b, :n] - want).abs().max() if n else 0.0) < 2e-6
                else:
                    assert torch.equal(kb[b, :n], want)
                assert torch.equal(ks[b, :n], scores[b][m]) and torch.equal(kl[b, :n], labels[b][m].long())


def test_missing_gpu_inputs_fail_loudly(T, head):
    feats = [torch.from_numpy(f) for f in synth.make_feats('tiny', seed=1)]
    with pytest.raises(T.TransCARHipError):
        head(feats, synth.make_img_metas(1, radar=synth.make_radar_frame()))


@pytest.mark.parametrize('matrix', ['f32', 'f16x2'])
@pytest.mark.parametrize('case', ['ramp_up', 'ramp_down', 'huge_negative_start', 'spikes', 'short_ragged'])
def test_sdpa_lazy_recentring_extreme_scores(T, case, matrix):
    """The attention core re-centres its running reference only when a score exceeds it by 2^8
    (self_attn.hip): score sequences built to stress that -- monotone ramps over the keys (every tile
    above / below the last), a first tile hundreds of octaves below the rest, isolated spikes, a key
    count that leaves several waves without a tile -- against softmax(S) V in float64."""
    from transcar_amd import ops
    rng = np.random.RandomState(7)
    B, H, D = 2, 8, 32
    Q = 37 if case == 'short_ragged' else 900
    C = H * D
    q = rng.standard_normal((B, Q, C)).astype(np.float32)
    k = rng.standard_normal((B, Q, C)).astype(np.float32)
    v = rng.standard_normal((B, Q, C)).astype(np.float32)
    # one channel per head carries a key-dependent offset: q[..., 0] = 1 and k[..., 0] = f(key)
    f = {'ramp_up': np.linspace(-150.0, 150.0, Q), 'ramp_down': np.linspace(150.0, -150.0, Q),
         'huge_negative_start': np.where(np.arange(Q) < 16, -400.0, rng.uniform(-3, 3, Q)),
         'spikes': np.where(rng.uniform(size=Q) < 0.01, 120.0, 0.0),
         'short_ragged': np.linspace(-40.0, 40.0, Q)}[case].astype(np.float32)
    for h in range(H):
        q[:, :, h * D] = 1.0
        k[:, :, h * D] = f[None, :]
    # the kernel's q is pre-scaled by log2(e) / sqrt(D) and its softmax is 2^x: S = q k^T / sqrt(D) in nats
    qs = torch.from_numpy(q) * (1.4426950408889634 / np.sqrt(D))
    qpad = ((Q + 15) // 16) * 16
    vt = torch.zeros((B, C, qpad), dtype=torch.float32)
    vt[:, :, :Q] = torch.from_numpy(v).permute(0, 2, 1)
    got = ops.sdpa(gpu(qs), gpu(k), gpu(vt), matrix_path=matrix).cpu().double()
    qd = torch.from_numpy(q).double().view(B, Q, H, D).permute(0, 2, 1, 3)
    kd = torch.from_numpy(k).double().view(B, Q, H, D).permute(0, 2, 1, 3)
    vd = torch.from_numpy(v).double().view(B, Q, H, D).permute(0, 2, 1, 3)
    p = torch.softmax(qd @ kd.transpose(-1, -2) / np.sqrt(D), -1)
    want = (p @ vd).permute(0, 2, 1, 3).reshape(B, Q, C)
    assert torch.isfinite(got).all()
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize('tile_rows', [0, 16, 32])
def test_radar_row_order_is_invisible_in_the_outputs(T, head, tile_rows):
    """tc_head_options.radar_row_order: with the queries that have a radar return inside their first gate
    processed first (row tiles without any skip the q projection / gated attention / out_proj), scores,
    boxes and hit counts are bit for bit those of the queries' own order -- two samples per launch (a tile
    straddles the sample boundary), 4-row and 16-row tiles."""
    import bench
    bench._imports()
    from transcar_amd.detr3d_head import head_options
    inp = bench.make_inputs(head, dev(), 'tiny', 2, seed=23)
    outs = {}
    for name, compact in (('own', False), ('hits_first', True)):
        o = head.forward_nhwc(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'], inp['pad_mult'], aux=True,
                              options=head_options(tile_rows=tile_rows or None, radar_compact=compact))
        torch.cuda.synchronize()
        outs[name] = (o['all_cls_scores'].clone(), o['all_bbox_preds'].clone(), o['aux']['radar_hit_counts'].clone())
    hits = outs['own'][2]
    assert int((hits > 0).sum()) > 100 and int((hits == 0).sum()) > 1000        # both kinds of rows exist
    for a_, b_ in zip(outs['own'], outs['hits_first']):
        assert torch.equal(a_, b_)


def test_gate_predicate_without_square_roots_is_the_reference_comparison(T):
    """GateGeom::hit compares squared distances with the smallest float whose square root reaches the radius
    (rowdev.hpp sqrt_threshold) instead of `cdist < radius` (HEAD:568-571).  On the device: 4096 radii of the
    clamp range x (every float within 256 ulps of radius^2 + 4096 random ones) -- not one disagreement."""
    from transcar_amd import _lib as L
    bad = torch.zeros(1, dtype=torch.int64, device=dev())
    L.check(L.lib().tc_radar_gate_selfcheck(4096, 12345, bad.data_ptr(), None), 'selfcheck')
    torch.cuda.synchronize()
    assert int(bad.item()) == 0


def test_instruction_count_forms_of_the_row_arithmetic_give_the_same_bits(T):
    """Round 6 rewrote the LayerNorm of a wave's four rows (one packed reduction tree, one 1 / sqrt), the split of a value
    into two f16 planes (v_fma_mix forms) and the way back (rowdev.hpp ln_rows<4>, chain.hip split_t2 / act_ld4) for fewer
    vector instructions.  On the device: 2 048 workgroups x 4 waves of random and special rows / values (constant rows, a
    mean far above the deviation, signed zeros, subnormals, f16 overflow, infinities, NaN) through both forms -- not one
    differing bit."""
    from transcar_amd import _lib as L
    bad = torch.zeros(3, dtype=torch.int64, device=dev())
    L.check(L.lib().tc_rowops_selfcheck(2048, 20261004, bad.data_ptr(), None), 'rowops selfcheck')
    torch.cuda.synchronize()
    assert bad.tolist() == [0, 0, 0], 'LayerNorm / split / un-split mismatches: %s' % bad.tolist()


def test_pipelines_can_share_streams(T, head):
    """FramePipeline(streams=other.streams): a second pipeline on the (idle) first one's HIP streams -- the
    lanes of a third pipeline with streams of its own can end up on shared hardware queues -- gives the
    same results; too few streams are refused."""
    import bench
    bench._imports()
    from transcar_amd.pipeline import FramePipeline
    lanes = [bench.make_inputs(head, dev(), 'tiny', 1, seed=41 + i) for i in range(2)]
    a = FramePipeline(head, lanes)
    for _ in range(2):
        a.launch()
    a.synchronize()
    want = [a.outputs[i][0]['all_bbox_preds'].clone() for i in range(2)]
    b = FramePipeline(head, lanes, streams=a.streams)
    assert all(x is y for x, y in zip(a.streams, b.streams))
    for _ in range(2):
        b.launch()
    b.synchronize()
    for i in range(2):
        assert torch.equal(b.outputs[i][0]['all_bbox_preds'], want[i])
    with pytest.raises(ValueError):
        FramePipeline(head, lanes, streams=a.streams[:1])


@pytest.mark.parametrize('tile_rows,matrix', [(8, None), (16, 'f16x2'), (16, 'f32'), (32, 'f16x2')])
def test_a_frame_in_a_batch_equals_the_frame_alone(T, head, tile_rows, matrix):
    """Size independence of the whole path at a fixed tile height: every kernel is row-local except the
    per-(sample, head) attention, so frame b of a three-frame launch (tiles straddle the sample boundaries,
    the radar rows are re-ordered per sample) is BIT-IDENTICAL to the same frame launched alone."""
    import bench
    bench._imports()
    from transcar_amd.detr3d_head import head_options
    B = 3
    inp = bench.make_inputs(head, dev(), 'tiny', B, seed=61)
    opt = head_options(tile_rows=tile_rows, radar_compact=True, matrix_path=matrix)
    full = head.forward_nhwc(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'], inp['pad_mult'], aux=True, options=opt)
    torch.cuda.synchronize()
    full = {k: full[k].clone() for k in ('all_cls_scores', 'all_bbox_preds')}
    for b in range(B):
        one = head.forward_nhwc([f[6 * b:6 * b + 6] for f in inp['nhwc']], inp['l2i'][b:b + 1], inp['hw'],
                                inp['tokens'][b:b + 1], inp['pad_mult'], aux=True, options=opt)
        torch.cuda.synchronize()
        assert torch.equal(one['all_cls_scores'][:, 0], full['all_cls_scores'][:, b])
        assert torch.equal(one['all_bbox_preds'][:, 0], full['all_bbox_preds'][:, b])


def test_f16x2_chains_are_exact_and_repeatable_at_two_workgroups_per_cu(T, head):
    """Round 4 regression (profiles/r4_f16x2_hazard.txt): the first f16x2 item loop refilled its weight registers in
    place, as the f32 loop does; with TWO workgroups per CU (8 frames = 450 row tiles) the radar chain then came out
    wrong on a few queries per launch, differently from run to run, while one frame (one workgroup per CU) was always
    exact.  Eight frames through tc_radar_fusion_fwd on both matrix paths, both row orders: f16x2 equals f32 to fp32
    rounding on EVERY query (same gate decisions), and three runs are bit-identical."""
    import ctypes as C
    import bench
    bench._imports()
    from transcar_amd import _lib as L, ops
    from transcar_amd.detr3d_head import head_options
    B = 8
    inp = bench.make_inputs(head, dev(), 'tiny', B, seed=71, host_feats=False)
    o = head.forward_nhwc(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'], inp['pad_mult'], aux=True,
                          options=head_options(tile_rows=16, matrix_path='f32'))
    torch.cuda.synchronize()
    hs5 = o['aux']['inter_states'][-1].contiguous().clone()
    ref5 = o['aux']['inter_references'][-1].contiguous().clone()
    lbox = o['aux']['last_box'].contiguous().clone()
    ws = torch.empty(L.lib().tc_head_workspace_bytes(C.byref(head._packed_view), B, inp['tokens'].shape[1]),
                     dtype=torch.uint8, device=dev())

    def run(mp, compact):
        opt = head_options(tile_rows=16, matrix_path=mp, radar_compact=compact)
        c, b, h = ops.radar_fusion(head, hs5, ref5, lbox, inp['tokens'], inp['pad_mult'], 0, 3, options=opt, ws=ws)
        torch.cuda.synchronize()
        return c.clone(), b.clone(), h.clone()
    for compact in (None, False):
        want = run('f32', compact)
        assert int((want[2] > 0).sum()) > 200, 'the rig must exercise the gated attention'
        first = None
        for rep in range(3):
            got = run('f16x2', compact)
            assert torch.equal(got[2], want[2]), 'gate decisions differ'
            assert float((got[0] - want[0]).abs().max()) < 1e-4 and float((got[1] - want[1]).abs().max()) < 1e-4
            if first is None:
                first = got
            else:
                assert all(torch.equal(a, b) for a, b in zip(got, first)), 'f16x2 is not repeatable (run %d)' % rep
    # ... and the whole forward (decoder chains included), run to run
    outs = []
    for rep in range(3):
        o = head.forward_nhwc(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'], inp['pad_mult'],
                              options=head_options(tile_rows=16, matrix_path='f16x2'))
        torch.cuda.synchronize()
        outs.append((o['all_cls_scores'].clone(), o['all_bbox_preds'].clone()))
    assert all(torch.equal(outs[0][0], x[0]) and torch.equal(outs[0][1], x[1]) for x in outs[1:])


@pytest.mark.parametrize('tile_rows', [16, 32])
def test_full_size_launch_of_eight_frames_is_frame_by_frame_the_single_frame_path(T, tile_rows):
    """The bench's own configuration (ResNet-101 FPN shapes, iid-noise maps, 8 frames per launch, 16-row tiles
    on the 16x16x4 MFMA, two workgroups per CU, radar rows compacted): frames 0, 3 and 7 of the launch are
    bit-identical to the same frames launched alone at the same tile height, and the decoded boxes with them --
    a size-independent property at BASELINE.json's full size, where the oracle takes minutes per frame."""
    import bench
    bench._imports()
    from transcar_amd import ops
    from transcar_amd.detr3d_head import head_options
    head, _ = bench.build_head(dev())
    B = 8
    inp = bench.make_inputs(head, dev(), 'res101', B, seed=71, host_feats=False)
    opt = head_options(tile_rows=tile_rows)
    full = head.forward_nhwc(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'], inp['pad_mult'], options=opt)
    torch.cuda.synchronize()
    cls, box = full['all_cls_scores'].clone(), full['all_bbox_preds'].clone()
    assert torch.isfinite(cls).all() and torch.isfinite(box).all()
    dec_full = ops.box_decode_topk(cls[-1], box[-1], head.bbox_coder.post_center_range, head.bbox_coder.max_num)
    dec_full = [d.clone() for d in dec_full]
    for b in (0, 3, 7):
        one = head.forward_nhwc([f[6 * b:6 * b + 6] for f in inp['nhwc']], inp['l2i'][b:b + 1], inp['hw'],
                                inp['tokens'][b:b + 1], inp['pad_mult'], options=opt)
        torch.cuda.synchronize()
        assert torch.equal(one['all_cls_scores'][:, 0], cls[:, b])
        assert torch.equal(one['all_bbox_preds'][:, 0], box[:, b])
        dec = ops.box_decode_topk(one['all_cls_scores'][-1], one['all_bbox_preds'][-1],
                                  head.bbox_coder.post_center_range, head.bbox_coder.max_num)
        for a_, b_ in zip(dec, dec_full):
            assert torch.equal(a_[0], b_[b])


@pytest.mark.parametrize('fpl', ['resident', 10])
def test_timed_geometry_is_frame_by_frame_the_single_frame_path(T, fpl):
    """EXACTLY what `python bench.py --steps 20` launches: ResNet-101 FPN shapes, iid-noise maps,
    the default options (automatic tile height: 32 rows since round 5, radar rows compacted), three lanes in flight,
    frames per launch = the most whose tiles are resident at once (9 = 254 workgroups of 32 rows, one per CU), a window of
    20 submits = 9 + 9 + a PARTIAL launch of 2 (at the full launch's tile height: FramePipeline.tile_rows_of) -- and, second
    case, 10 frames per launch = 282 workgroups on 256 slots (a second scheduling round; the driver's round-2 geometry).  Every one of the 20 frames, and its decoded boxes,
    is bit-identical to the same frame launched alone at the same tile height."""
    import bench
    bench._imports()
    from transcar_amd import ops
    from transcar_amd.detr3d_head import head_options
    from transcar_amd.pipeline import FramePipeline, resident_frames_per_launch
    head, _ = bench.build_head(dev())
    P = resident_frames_per_launch(head.num_query, dev()) if fpl == 'resident' else fpl
    assert fpl != 'resident' or P == 9                  # MI355X: 256 CUs x 2 workgroups x 16 rows // 900
    K, nl = 20, 3
    lanes = [bench.make_inputs(head, dev(), 'res101', P, seed=81 + 7 * i, host_feats=False) for i in range(nl)]
    pipe = FramePipeline(head, lanes, options=head_options())
    got = []                                            # (lane, slot) in submit order
    for _ in range(K):
        lane, slot, _ = pipe.submit()
        got.append((lane, slot))
    n_part = pipe.flush()
    assert n_part == K % P and (n_part == 0 or pipe.last_flush[1] == n_part)
    pipe.synchronize()
    torch.cuda.synchronize()
    part_lane = pipe.last_flush[0] if n_part else -1
    for lane, slot in got:
        partial = lane == part_lane
        outs, dec = pipe.last_flush[2] if partial else pipe.outputs[lane]
        opt = head_options(tile_rows=pipe.tile_rows_of(n_part if partial else P))       # what that launch used
        assert opt.chain_tile_rows == 32                 # more than 4096 rows per full launch: one 32-row workgroup per CU
        inp = lanes[lane]
        one = head.forward_nhwc([f[6 * slot:6 * slot + 6] for f in inp['nhwc']], inp['l2i'][slot:slot + 1], inp['hw'],
                                inp['tokens'][slot:slot + 1], inp['pad_mult'], options=opt)
        torch.cuda.synchronize()
        assert torch.isfinite(one['all_cls_scores']).all()
        assert torch.equal(one['all_cls_scores'][:, 0], outs['all_cls_scores'][:, slot]), (lane, slot)
        assert torch.equal(one['all_bbox_preds'][:, 0], outs['all_bbox_preds'][:, slot]), (lane, slot)
        d1 = ops.box_decode_topk(one['all_cls_scores'][-1], one['all_bbox_preds'][-1],
                                 head.bbox_coder.post_center_range, head.bbox_coder.max_num)
        for a_, b_ in zip(d1, dec):
            assert torch.equal(a_[0], b_[slot]), (lane, slot)


@pytest.mark.parametrize('case', ['res101-32', 'res101-16', 'tiny-16-ragged'])
def test_cam_pregather_is_bit_identical_to_the_in_chain_gather(T, case):
    """Round 6 (tc_head_options.cam_pregather, ABI 12, opt-in): on the f16x2 path the camera taps of decoder
    layers 1..5 are gathered and bilinearly reduced by extra workgroups of the attention-core launch IN FRONT of the
    layer's chain (rowdev.hpp cam_pregather_rows), and the chain's sampling step only weighs and sums the stored level
    values (XFMR:367-373).  Same projection, same taps, same products in the same order: every output of the head --
    class scores, boxes, decoder states, reference points, hit counts, the visible-pair count -- is torch.equal to
    the forward whose chains gather for themselves (cam_pregather = 0: the default, every launch until round 5), on the bench's
    iid-noise ResNet-101 maps at both tile heights and on tiny maps with a ragged last tile (777 queries)."""
    import bench
    bench._imports()
    from transcar_amd.detr3d_head import head_options
    shapes, rows = case.split('-')[0], int(case.split('-')[1])
    if case.endswith('ragged'):
        sd_np = synth.make_state_dict(seed=5, num_query=777)
        head = T.build_head(configs.head_cfg(num_query=777))
        head.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
        head = head.to(dev()).eval()
        B = 3
    else:
        head, _ = bench.build_head(dev())
        B = 9 if rows == 32 else 5
    inp = bench.make_inputs(head, dev(), shapes, B, seed=31, host_feats=False)
    outs = {}
    for mode in ('pre', 'direct', 'pre2'):
        o = head.forward_nhwc(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'], inp['pad_mult'], aux=True,
                              options=head_options(tile_rows=rows, matrix_path='f16x2', cam_pregather=mode != 'direct'))
        torch.cuda.synchronize()
        outs[mode] = o
    assert torch.isfinite(outs['direct']['all_cls_scores']).all()
    for other in ('pre', 'pre2'):
        for k in ('all_cls_scores', 'all_bbox_preds'):
            assert torch.equal(outs[other][k], outs['direct'][k]), (other, k)
        for k in ('inter_states', 'inter_references', 'radar_hit_counts', 'last_box'):
            assert torch.equal(outs[other]['aux'][k], outs['direct']['aux'][k]), (other, k)
        assert int(outs[other]['aux']['sample_pairs']) == int(outs['direct']['aux']['sample_pairs']) > 0


@pytest.mark.parametrize('tile_rows', [16, 32])
def test_soak_of_the_timed_geometry_is_bit_identical_launch_to_launch(T, tile_rows):
    """>= 2 000 replays of bench.py's nine-frame launch (ResNet-101 FPN shapes, iid-noise maps,
    two-plane f16 operands on the matrix cores, radar rows compacted, three lanes in flight so that launches of
    different lanes overlap on the CUs) -- every launch's outputs and decoded boxes compared ON THE DEVICE with the
    first launch of its lane: no element may ever differ.  Round 4's flaky rows (root cause: tools/pk_hazard_probe.hip,
    profiles/r5_refill_hazard.txt) showed as 1-8 differing rows per launch in this geometry."""
    import time
    import bench
    bench._imports()
    from transcar_amd.detr3d_head import head_options
    from transcar_amd.pipeline import FramePipeline
    head, _ = bench.build_head(dev())
    P, nl, replays = 9, 3, 2010
    lanes = [bench.make_inputs(head, dev(), 'res101', P, seed=91 + 5 * i, host_feats=False) for i in range(nl)]
    pipe = FramePipeline(head, lanes, options=head_options(tile_rows=tile_rows, matrix_path='f16x2'))
    for _ in range(P * nl):                              # launch 0 of every lane
        pipe.submit()
    pipe.synchronize()
    torch.cuda.synchronize()
    ref = []
    for lane in range(nl):
        outs, dec = pipe.outputs[lane]
        assert torch.isfinite(outs['all_cls_scores']).all()
        ref.append([outs['all_cls_scores'].clone(), outs['all_bbox_preds'].clone()] + [d.clone() for d in dec])
    bad = [torch.zeros(1, dtype=torch.int64, device=dev()) for _ in range(nl)]      # one counter per lane's stream
    torch.cuda.synchronize()
    t0 = time.time()
    for rep in range(replays):
        lane = rep % nl
        for _ in range(P):
            got_lane, _, launched = pipe.submit()
        assert got_lane == lane and launched
        # the comparison is enqueued behind the lane's launch on the lane's own stream order (outputs are static tensors)
        with torch.cuda.stream(pipe.streams[lane]):
            outs, dec = pipe.outputs[lane]
            cur = [outs['all_cls_scores'], outs['all_bbox_preds']] + list(dec)
            for a_, b_ in zip(cur, ref[lane]):
                bad[lane] += (a_ != b_).sum()
    pipe.synchronize()
    torch.cuda.synchronize()
    nbad = sum(int(b) for b in bad)
    assert nbad == 0, '%d elements differed from the first launch over %d replays' % (nbad, replays)
    assert time.time() - t0 < 60.0


@pytest.mark.parametrize('rows', [0, 32])
@pytest.mark.parametrize('case', ['weight', 'activation'])
def test_f16_range_guard_flags_an_overflow_and_falls_back_to_f32(T, sd_np, case, rows):
    """an operand beyond the f16 planes' range on the DEFAULT (f16x2) path of a batched launch --
    a weight of 1e5 (> 65504) or feature maps of ~1e9 (a sampled activation > 65504 * 2^6 = 4.19e6) -- turns into
    inf / NaN in the outputs (never a wrong finite number), tc_head_options.range_status says so, get_bboxes reads the
    word with its counts and the head runs its next automatic forwards on the exact-fp32 kernels, whose results are
    finite and bit-identical to a head that was on the f32 path all along."""
    import warnings
    from transcar_amd.detr3d_head import head_options
    sdm = {k: v.copy() for k, v in sd_np.items()}
    scale = 1.0
    if case == 'weight':
        sdm['transformer.decoder.layers.3.ffns.0.layers.0.0.weight'][5, 7] = 1.0e5
    else:
        scale = 1.0e9

    def make():
        h = T.build_head(configs.head_cfg())
        h.load_state_dict({k: torch.from_numpy(v) for k, v in sdm.items()}, strict=True)
        return h.to(dev()).eval()
    h = make()
    if rows:                 # the 32-row tiles' epilogue (lin_epilogue32) carries its own copy of the test; on the fall-back the
        h.forward_options = head_options(tile_rows=rows)       # head lowers a forced 32 to the f32 kernels' 16 rows
    feats = [gpu(np.concatenate([synth.make_feats('tiny', seed=s_, smooth=SMOOTH)[l] for s_ in (1, 11, 12)], 0) * scale)
             for l in range(4)]
    frames = [synth.make_radar_frame(seed=2 + i) for i in range(3)]
    metas = synth.make_img_metas(3, synth.make_lidar2img(), radar=frames)
    assert h.last_range_status == 0 and not h.matrix_fallback
    outs = h(feats, metas)                                  # three frames: 16-row tiles on the f16 matrix cores
    assert not torch.isfinite(outs['all_cls_scores']).all(), 'the overflow must be visible in the values too'
    with pytest.warns(UserWarning, match='f16x2'):
        h.get_bboxes(outs, metas)                           # ... and in the status word, read with the decode's counts
    assert h.matrix_fallback and h.last_range_status == 0  # (cleared)
    again = h(feats, metas)                                 # automatic -> f32 now
    ref = make()
    ref.forward_options = head_options(matrix_path='f32')
    want = ref(feats, metas)
    assert torch.isfinite(again['all_cls_scores']).all() and torch.isfinite(again['all_bbox_preds']).all()
    assert torch.equal(again['all_cls_scores'], want['all_cls_scores']) and torch.equal(again['all_bbox_preds'], want['all_bbox_preds'])
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        h.get_bboxes(again, metas)                          # nothing to report
    assert ref.last_range_status == 0                       # the f32 path never sets it
    # an explicit f16x2 request is honoured (and flags again): only `auto` falls back
    h.forward_options = head_options(matrix_path='f16x2', tile_rows=16)
    h(feats, metas)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter('always')
        assert h.last_range_status == 1
    assert len(rec) == 1 and 'pinned' in str(rec[0].message)   # a pinned path cannot fall back: every overflow warns


def test_f16_range_guard_reaches_a_frame_pipeline(T, sd_np):
    """the main inference path replays captured graphs with the f16x2 kernels and the 32-row tiles
    baked in and never calls get_bboxes -- the guard must work THERE.  A lane's graph ends with a copy of the status word
    into a pinned host word; `wait` reads it, the head falls back (generation bump) and the next launch re-captures on
    the exact-fp32 kernels: finite outputs, equal to a pipeline that was on the f32 path all along."""
    import warnings
    from transcar_amd.detr3d_head import head_options
    from transcar_amd.pipeline import FramePipeline
    from transcar_amd import ops, radar as R
    sdm = {k: v.copy() for k, v in sd_np.items()}
    sdm['transformer.decoder.layers.3.ffns.0.layers.0.0.weight'][5, 7] = 1.0e5

    def make():
        h = T.build_head(configs.head_cfg())
        h.load_state_dict({k: torch.from_numpy(v) for k, v in sdm.items()}, strict=True)
        return h.to(dev()).eval()
    P = 5                                                   # 4 500 rows per launch: the automatic rule picks 32-row tiles
    feats = [gpu(np.concatenate([synth.make_feats('tiny', seed=10 + j, smooth=SMOOTH)[l] for j in range(P)], 0)) for l in range(4)]
    l2i = gpu(np.stack([synth.make_lidar2img()] * P))
    tok_np, pm = R.pack_tokens([R.build_radar_features(synth.make_radar_frame(seed=2 + j)) for j in range(P)], T=256)

    def lane_inputs():
        return dict(nhwc=[t.clone() for t in ops.to_nhwc_levels(feats)], l2i=l2i.clone(), hw=HW, tokens=gpu(tok_np), pad_mult=pm)
    h = make()
    pipe = FramePipeline(h, [lane_inputs(), lane_inputs()])
    assert pipe.tile_rows_of() == 32
    lane, (outs, dec) = pipe.launch()
    with pytest.warns(UserWarning, match='f16x2'):
        pipe.wait(lane)
    assert pipe.range_status(lane) == 1 and h.matrix_fallback
    assert not torch.isfinite(outs['all_cls_scores']).all()
    lane2, (outs2, dec2) = pipe.launch()                    # re-captures: f32 kernels, 16-row tiles
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        pipe.wait(lane2)
    assert pipe.range_status(lane2) == 0 and pipe.tile_rows_of() == 16
    ref = make()
    rp = FramePipeline(ref, [lane_inputs()], options=head_options(matrix_path='f32'))
    l3, (want, _) = rp.launch()
    rp.wait(l3)
    assert torch.isfinite(outs2['all_cls_scores']).all()
    assert torch.equal(outs2['all_cls_scores'], want['all_cls_scores']) and torch.equal(outs2['all_bbox_preds'], want['all_bbox_preds'])


@pytest.mark.parametrize('layout', ['channels_last', 'nchw'])
def test_plugin_entry_graphs_are_bit_identical_to_the_eager_entry(T, sd_np, layout):
    """Round 6: `head(mlvl_feats, img_metas)` + `get_bboxes` replay two captured hipGraphs once a call
    signature (feature-map addresses, shapes, options) is seen for the second time (transcar_amd/plugin_graph.py).
    Five calls on the same feature tensors with DIFFERENT frames' radar sweeps / lidar2img and maps refilled in place:
    call 1 is eager, call 2 captures, calls 3-5 replay -- every call's outputs and decoded boxes are torch.equal to a
    head whose graphs are switched off; results handed out earlier are not overwritten by later calls; new feature
    tensors (other addresses) take the eager path again; an in-place weight update is picked up by the next replay."""
    def make():
        h = T.build_head(configs.head_cfg())
        h.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
        return h.to(dev()).eval()
    hg, he = make(), make()
    he.plugin_graphs = False
    B = 2
    shapes = configs.LEVEL_SHAPES['tiny']

    def fresh(seed):
        g = torch.Generator(device=dev())
        g.manual_seed(seed)
        fs = [torch.randn((B, 6, 256, h_, w_), device=dev(), generator=g) for (h_, w_) in shapes]
        if layout == 'channels_last':
            fs = [f.view(B * 6, 256, *f.shape[-2:]).to(memory_format=torch.channels_last).view(B, 6, 256, *f.shape[-2:]) for f in fs]
        return fs
    feats = fresh(5)
    kept = []
    # (the very first forward of a head packs its weights and allocates its workspace: the buffer generation -- part of a
    # call's signature -- settles with it)
    hg(fresh(3), [synth.make_img_metas(1, radar=synth.make_radar_frame(seed=39, n_per_radar=30 + b))[0] for b in range(B)])
    base = dict(hg._plugin_graphs.stats)
    for it in range(5):
        for f in feats:                                  # the backbone writes the next frames into the same tensors
            f.mul_(0.9).add_(0.01 * (it + 1))
        l2i = synth.make_lidar2img()
        l2i = [m.copy() for m in l2i]
        l2i[0][0, 3] += 0.05 * it                         # another rig every call
        metas = [synth.make_img_metas(1, np.stack(l2i), radar=synth.make_radar_frame(seed=40 + 2 * it + b, n_per_radar=30 + b))[0]
                 for b in range(B)]
        og = hg(feats, metas)
        bg = hg.get_bboxes(og, metas)
        oe = he(feats, metas)
        be = he.get_bboxes(oe, metas)
        torch.cuda.synchronize()
        for k in ('all_cls_scores', 'all_bbox_preds'):
            assert torch.equal(og[k], oe[k]), (it, k)
        for sg, se in zip(bg, be):
            for x, y in zip(sg, se):
                assert x.shape == y.shape and torch.equal(x, y), it
        kept.append((og['all_cls_scores'], oe['all_cls_scores'].clone(), bg[0][0], be[0][0].clone()))
    st = {k: v - base[k] for k, v in hg._plugin_graphs.stats.items()}
    assert st['captures'] == 1 and st['replays'] == 4 and st['eager'] == 1, st
    for a_, b_, c_, d_ in kept:                          # what a call handed out still holds that call's values
        assert torch.equal(a_, b_) and torch.equal(c_, d_)
    # other tensors: eager again (first sighting of the new signature) ...
    feats2 = fresh(6)
    o2 = hg(feats2, metas)
    assert hg._plugin_graphs.stats['eager'] - base['eager'] == 2 and '_decoded' not in o2
    assert torch.equal(o2['all_cls_scores'], he(feats2, metas)['all_cls_scores'])
    # ... and an in-place change of a trainable weight reaches the next replay (re-pack in front of it)
    with torch.no_grad():
        for h in (hg, he):
            h.final_cls3[0].weight.mul_(1.5)
            h.mark_trainable_dirty()
    og, oe = hg(feats, metas), he(feats, metas)
    assert hg._plugin_graphs.stats['replays'] - base['replays'] == 5
    assert torch.equal(og['all_cls_scores'], oe['all_cls_scores']) and not torch.equal(og['all_cls_scores'][2], kept[-1][0][2])
    # a dict that is not the entry's latest is decoded again, correctly
    first = hg(feats, metas)
    second = hg(feats, metas)
    b1, b2 = hg.get_bboxes(first, metas), hg.get_bboxes(second, metas)
    for sg, se in zip(b1, b2):
        for x, y in zip(sg, se):
            assert torch.equal(x, y)


def test_plugin_entry_stages_lidar2img_per_call(T, head):
    """`Detr3DHead.forward` stages img_metas' projection matrices through a ring of eight pinned buffers and skips the
    copy when they are the ones staged last (ops._Lidar2ImgStaging): 20 calls that alternate fresh rigs, a repeated
    rig and a rig whose numpy arrays were modified IN PLACE since the last call all equal the device forward on a
    freshly uploaded tensor, bit for bit -- also from a second stream."""
    from transcar_amd import ops
    feats = [gpu(f) for f in synth.make_feats('tiny', seed=1, smooth=SMOOTH)]
    frame = synth.make_radar_frame(seed=2)
    tok, pm = head.radar_tokens(synth.make_img_metas(1, synth.make_lidar2img(), radar=frame), dev())
    nhwc = ops.to_nhwc_levels(feats)
    rng = np.random.RandomState(3)
    base = synth.make_lidar2img()
    l2i = base.copy()
    side = torch.cuda.Stream()
    for it in range(20):
        kind = it % 4
        if kind == 0:
            l2i = base * (1.0 + 0.01 * rng.standard_normal(base.shape))       # a new rig (new arrays)
        elif kind == 2:
            l2i[:, 0, 3] += 0.25                                                # the SAME arrays, modified in place
        # kind 1, 3: unchanged since the last call (no copy)
        metas = synth.make_img_metas(1, l2i, radar=frame)
        if it >= 12:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                got = head(feats, metas)
            torch.cuda.current_stream().wait_stream(side)
        else:
            got = head(feats, metas)
        want = head.forward_nhwc(nhwc, ops.lidar2img_tensor(metas, dev()), configs.IMG_SHAPE[:2], tok, pm)
        torch.cuda.synchronize()
        for k in ('all_cls_scores', 'all_bbox_preds'):
            assert torch.equal(got[k], want[k]), (it, k)
    st = ops._l2i_staging
    assert st.last[str(dev())][0].dtype == np.float32 and len(next(iter(st.slots.values()))[0]) == 8


def test_ragged_radar_batch_through_the_plugin_entry(T, head):
    """One `head(mlvl_feats, img_metas)` call with three samples whose radar frames differ in size (255 points, a
    ragged handful, none at all): the batch ingest (tc_radar_build_tokens_batch, one launch) writes every sample's
    token matrix exactly as the single-sample operator does at the batch's token count, and the call's outputs are
    those of the device forward on these tokens, bit for bit."""
    from transcar_amd import ops, radar as R
    l2i = synth.make_lidar2img()
    gold = g('g5_head_tiny.npz')
    frames = [synth.make_radar_frame(seed=2, n_per_radar=51, centres=gold['radar_centres']),
              synth.make_radar_frame(seed=11, n_per_radar=[7, 0, 33, 0, 12]),
              synth.make_radar_frame(seed=5, n_per_radar=[0, 0, 0, 0, 0])]
    B = len(frames)
    feats = [gpu(np.concatenate([f] * B, 0)) for f in synth.make_feats('tiny', seed=1, smooth=SMOOTH)]
    metas = synth.make_img_metas(B, l2i, radar=frames)
    got = head(feats, metas)
    tok, pm = head.radar_tokens(metas, dev())
    Tn = tok.shape[1]
    assert tok.shape[0] == B and Tn == ops.radar_tokens_T(255) == 256 and pm == 1500 - Tn + 1
    for b, fr in enumerate(frames):
        one, cnt, pm1 = ops.radar_build_tokens(fr, Tn, dev(), check=True)
        assert pm1 == pm and torch.equal(one[0], tok[b]), b
        n = sum(int(np.asarray(fr['points'][c]).shape[1]) for c in R.RADAR_CHANNELS)
        assert int(cnt) <= n
        assert torch.all(tok[b, int(cnt):, 0] == 500.0)           # pad rows behind the kept points (HEAD:526-530)
    want = head.forward_nhwc(ops.to_nhwc_levels(feats), ops.lidar2img_tensor(metas, dev()), configs.IMG_SHAPE[:2], tok, pm)
    for k in ('all_cls_scores', 'all_bbox_preds'):
        assert torch.equal(got[k], want[k]), k
    assert torch.isfinite(got['all_bbox_preds']).all()


@pytest.mark.parametrize('batch', [1, 3])
def test_forward_in_two_phases_around_the_token_build(T, head, batch):
    """tc_head_options.phase: phase 1 (prologue + decoder layers 0 .. L-3) is enqueued while the tokens do not exist
    yet, the caller fills them (here: a copy on the side stream forward_nhwc uses), phase 2 does the rest --
    every output and aux tensor bit-identical to the one-call forward (4- and 8-row tiles)."""
    from transcar_amd import ops
    l2i = synth.make_lidar2img()
    feats = [gpu(np.concatenate([f] * batch, 0)) for f in synth.make_feats('tiny', seed=1, smooth=SMOOTH)]
    nhwc = ops.to_nhwc_levels(feats)
    hw = configs.IMG_SHAPE[:2]
    metas = synth.make_img_metas(batch, l2i, radar=synth.make_radar_frame(seed=2, n_per_radar=40))
    tok, pm = head.radar_tokens(metas, dev())
    l2i_t = ops.lidar2img_tensor(metas, dev())
    want = head.forward_nhwc(nhwc, l2i_t, hw, tok, pm, aux=True)
    empty = torch.full_like(tok, float('nan'))
    calls = []

    def fill():
        calls.append(1)
        empty.copy_(tok)
    got = head.forward_nhwc(nhwc, l2i_t, hw, empty, pm, aux=True, fill_tokens=fill)
    assert calls == [1]
    for k in ('all_cls_scores', 'all_bbox_preds'):
        assert torch.equal(got[k], want[k]), k
    for k, v in want['aux'].items():
        assert torch.equal(got['aux'][k], v), k
    # the phases by hand, and a bad phase is refused
    o1 = T.detr3d_head.head_options(phase=1)
    with pytest.raises(T.TransCARHipError):
        head.forward_nhwc(nhwc, l2i_t, hw, tok, pm, options=T.detr3d_head.head_options(phase=3))
    with pytest.raises(T.TransCARHipError):
        head.forward_nhwc(nhwc, l2i_t, hw, tok, pm, options=T.detr3d_head.head_options(phase=1, unfused=True))
    assert o1.phase == 1


def test_device_radar_ingest_inside_forward_and_as_a_graph_node(T, head):
    """VERDICT r2 (missing 4): tc_radar_build_tokens is no island any more.
    (i) ``head(mlvl_feats, img_metas)`` with RAW sweeps in img_metas builds the tokens on the device
    (tc_radar_build_tokens_batch, one launch for the batch): bit-identical to handing the head the tokens of
    the single-sample device op, and equal to the host-token route up to the 1-ulp velocity columns.
    (ii) FramePipeline(radar_raw_capacity=...): the ingest is the first node of a lane's graph; frames arrive
    as raw sweeps (write_inputs(radar_frame=...)), results bit-identical to (i)'s route, frame after frame."""
    from transcar_amd import ops, radar as R
    from transcar_amd.pipeline import FramePipeline
    l2i = synth.make_lidar2img()
    gold = g('g5_head_tiny.npz')
    frames = [synth.make_radar_frame(seed=2, n_per_radar=51, centres=gold['radar_centres']),
              synth.make_radar_frame(seed=11, n_per_radar=[7, 0, 33, 0, 12]),
              synth.make_radar_frame(seed=5, n_per_radar=[0, 0, 0, 0, 0])]
    feats = [gpu(f) for f in synth.make_feats('tiny', seed=1, smooth=SMOOTH)]
    nhwc = ops.to_nhwc_levels(feats)
    hw = configs.IMG_SHAPE[:2]
    assert head.radar_ingest == 'device'
    for fr in frames:
        metas = synth.make_img_metas(1, l2i, radar=fr)
        got = head(feats, metas, aux=True)
        tok_dev, pm = head.radar_tokens(metas, dev())
        Tn = tok_dev.shape[1]
        tok_one, cnt, pm1 = ops.radar_build_tokens(fr, Tn, dev(), check=True)
        assert pm1 == pm and torch.equal(tok_one, tok_dev)
        l2i_t = ops.lidar2img_tensor(metas, dev())
        want = head.forward_nhwc(nhwc, l2i_t, hw, tok_one, pm)
        for k in ('all_cls_scores', 'all_bbox_preds'):
            assert torch.equal(got[k], want[k]), k
        # the reference's numpy route (HEAD:311-521 restated in transcar_amd/radar.py)
        tok_host, pmh = head.radar_tokens(metas, dev(), T=Tn, ingest='host')
        assert pmh == pm
        th, td = tok_host[0].cpu().numpy(), tok_dev[0].cpu().numpy()
        vel = [9, 10, 11, 12, 13, 14]
        exact = [c for c in range(36) if c not in vel]
        np.testing.assert_array_equal(td[:, exact], th[:, exact])
        assert np.all(np.abs(td[:, vel] - th[:, vel]) <= np.spacing(np.abs(th[:, vel]).astype(np.float32)))
        host = head.forward_nhwc(nhwc, l2i_t, hw, tok_host, pm, aux=True)
        agree = (host['aux']['radar_hit_counts'] == got['aux']['radar_hit_counts']).all(0)[0].cpu().numpy()
        assert agree.mean() > 0.995
        for k in ('all_cls_scores', 'all_bbox_preds'):
            d = (host[k] - got[k]).abs()[:, 0].max(-1).values.cpu().numpy()
            assert float(d[:, agree].max()) < 2e-5, (k, float(d[:, agree].max()))
    # (ii) the ingest as the first node of a lane's graph, two frame slots per lane
    Tn = 256
    P = 2
    lane_in = dict(nhwc=[torch.cat([x, x], 0) for x in nhwc],
                   l2i=ops.lidar2img_tensor(synth.make_img_metas(P, l2i), dev()), hw=hw,
                   tokens=torch.full((P, Tn, 36), 500.0, device=dev()), pad_mult=R.NUM_RADAR_TOKENS - Tn + 1)
    pipe = FramePipeline(head, [lane_in], radar_raw_capacity=512)
    for a_, b_ in ((frames[0], frames[1]), (frames[2], frames[0]), (frames[1], frames[1])):
        pipe.write_inputs(0, radar_frame=a_, slot=0)
        pipe.write_inputs(0, radar_frame=b_, slot=1)
        _, (outs, _) = pipe.launch(0)
        pipe.wait(0)
        assert not pipe.radar_overflow(0)
        for s_, fr in enumerate((a_, b_)):
            tok_one, _, pm1 = ops.radar_build_tokens(fr, Tn, dev())
            want = head.forward_nhwc(nhwc, lane_in['l2i'][:1], hw, tok_one, pm1,
                                     options=T.detr3d_head.head_options(tile_rows=8))
            assert torch.equal(outs['all_bbox_preds'][:, s_], want['all_bbox_preds'][:, 0])
            assert torch.equal(outs['all_cls_scores'][:, s_], want['all_cls_scores'][:, 0])
    # a frame that does not fit the captured token count is reported, not silently mis-weighted
    big = synth.make_radar_frame(seed=4, n_per_radar=[90, 80, 70, 60, 50])
    pipe.write_inputs(0, radar_frame=big, slot=0)
    pipe.launch(0)
    pipe.wait(0)
    assert pipe.radar_overflow(0)
