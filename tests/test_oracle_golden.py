"""The CPU oracle against the fixtures produced by the reference itself
(tests/golden/make_golden.py).  Runs on CPU, no reference needed."""
import os

import numpy as np
import pytest
import torch

from oracle import transcar_oracle as O
from transcar_amd import configs, synth
from parity_util import assert_rows_match



@pytest.fixture(autouse=True)
def _no_grad():
    with torch.no_grad():
        yield


PCR = configs.point_cloud_range
HW = configs.IMG_SHAPE[:2]
#: oracle vs reference, end to end through 6 decoder + 3 radar layers, fp32
#: both: half of the 1e-3 box tolerance BASELINE.json's north_star states
E2E_TOL = 5e-4


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.fixture(scope='module')
def sd():
    return O.to_torch_sd(synth.make_state_dict(seed=3))


def test_g1_feature_sampling(golden_dir):
    g = _g(golden_dir, 'g1_feature_sampling.npz')
    feats = [torch.from_numpy(f) for f in
             synth.make_feats('tiny', seed=12, channels=8)]
    l2i = torch.from_numpy(synth.make_lidar2img()).float()[None]
    sampled, mask = O.feature_sampling(feats, torch.from_numpy(g['ref_points']),
                                       PCR, l2i, HW)
    assert np.array_equal(mask.numpy(), g['mask'])
    np.testing.assert_allclose(sampled.numpy(), g['sampled'], atol=1e-6, rtol=0)
    # edge rows: the point at the rig origin is invisible everywhere
    assert not mask[0, 0, 0].any()


def test_g2_cross_atten(golden_dir, sd):
    g = _g(golden_dir, 'g2_cross_atten.npz')
    rng = np.random.RandomState(21)
    feats = [torch.from_numpy(f) for f in synth.make_feats('tiny', seed=22)]
    l2i = torch.from_numpy(synth.make_lidar2img()).float()[None]
    query = torch.from_numpy(rng.standard_normal((900, 1, 256)).astype(np.float32))
    qpos = torch.from_numpy(rng.standard_normal((900, 1, 256)).astype(np.float32))
    refp = torch.from_numpy(rng.uniform(0.02, 0.98, (1, 900, 3)).astype(np.float32))
    out = O.cross_atten(sd, 'transformer.decoder.layers.2.attentions.1',
                        query, qpos, feats, refp, PCR, l2i, HW)
    np.testing.assert_allclose(out.numpy()[::4], g['out'], atol=2e-5, rtol=0)


def _radar_frame_like_golden(g):
    """The radar frame of make_golden.g345_head, pass 2 (centres are stored
    in the fixture as an input)."""
    return synth.make_radar_frame(seed=2, n_per_radar=51,
                                  centres=g['radar_centres'])


@pytest.mark.parametrize('tag', ['tiny', 'res101', 'vovnet'])
def test_g5_full_head(golden_dir, sd, tag):
    g = _g(golden_dir, 'g5_head_%s.npz' % tag)
    feats = [torch.from_numpy(f) for f in synth.make_feats(tag, seed=1, smooth=(4, 6))]
    l2i = torch.from_numpy(synth.make_lidar2img()).float()[None]
    # radar tokens: the fixture holds the reference's own [n,36] rows; the
    # oracle's builder must reproduce them from the raw frame
    frame = _radar_frame_like_golden(g)
    f36 = O.build_radar_features(frame)
    assert f36.shape[0] == int(g['fill_in'])
    np.testing.assert_allclose(f36.astype(np.float32), g['radar_tokens'],
                               atol=1e-6, rtol=1e-6)
    outs, dbg = O.head_forward(sd, feats, l2i, HW, f36, PCR, return_debug=True)
    np.testing.assert_allclose(dbg['inter_refs'].numpy(), g['inter_refs'],
                               atol=2e-5, rtol=0)
    np.testing.assert_allclose(dbg['init_ref'].numpy(), g['init_ref'],
                               atol=1e-6, rtol=0)
    hs = dbg['hs'].permute(0, 2, 1, 3).numpy()          # [6,Q,1,C]
    np.testing.assert_allclose(hs[:, ::16, 0, :], g['hs_rows'], atol=5e-5, rtol=0)
    # hit sets: same rows selected, same number of radar hits per row
    for i in range(3):
        assert len(dbg['hit_rows'][i]) == int(g['Lq'][i])
        hc = dbg['hit_counts'][i][dbg['hit_rows'][i]].numpy()
        assert np.array_equal(hc, g['hit_counts%d' % i])
    np.testing.assert_allclose(outs['all_cls_scores'].numpy(),
                               g['all_cls_scores'], atol=E2E_TOL, rtol=0)
    np.testing.assert_allclose(outs['all_bbox_preds'].numpy(),
                               g['all_bbox_preds'], atol=E2E_TOL, rtol=0)
    # post-processing (CODER:39-111, HEAD:1018)
    b, s, l = O.get_bboxes(outs, configs.pts_bbox_head['bbox_coder']
                           ['post_center_range'])[0]
    # top-300 order may swap near-tied neighbours: compare up to permutation
    np.testing.assert_allclose(s.numpy(), g['dec_scores'], atol=1e-5, rtol=0)
    mine = np.concatenate([b.numpy(), s.numpy()[:, None],
                           l.numpy()[:, None].astype(np.float32)], 1)
    gold = np.concatenate([g['dec_boxes'], g['dec_scores'][:, None],
                           g['dec_labels'][:, None].astype(np.float32)], 1)
    assert_rows_match(mine, gold, atol=2e-4, what='decoded boxes')


def test_g4_ragged_radar(golden_dir, sd):
    g = _g(golden_dir, 'g4_radar_ragged.npz')
    feats = [torch.from_numpy(f) for f in synth.make_feats('tiny', seed=1, smooth=(4, 6))]
    l2i = torch.from_numpy(synth.make_lidar2img()).float()[None]
    frame = synth.make_radar_frame(seed=5, n_per_radar=[7, 0, 3, 0, 12])
    f36 = O.build_radar_features(frame)
    assert f36.shape[0] == int(g['fill_in'])
    np.testing.assert_allclose(f36.astype(np.float32), g['radar_tokens'],
                               atol=1e-6, rtol=1e-6)
    outs = O.head_forward(sd, feats, l2i, HW, f36, PCR)
    np.testing.assert_allclose(outs['all_bbox_preds'].numpy(),
                               g['all_bbox_preds'], atol=E2E_TOL, rtol=0)
    np.testing.assert_allclose(outs['all_cls_scores'].numpy(),
                               g['all_cls_scores'], atol=E2E_TOL, rtol=0)


def test_no_radar_points(sd):
    """Empty radar: every token is padding, no query is updated by radar."""
    feats = [torch.from_numpy(f) for f in synth.make_feats('tiny', seed=1, smooth=(4, 6))]
    l2i = torch.from_numpy(synth.make_lidar2img()).float()[None]
    frame = synth.make_radar_frame(seed=5, n_per_radar=[0, 0, 0, 0, 0])
    f36 = O.build_radar_features(frame)
    assert f36.shape == (0, 36)
    outs, dbg = O.head_forward(sd, feats, l2i, HW, f36, PCR, return_debug=True)
    assert all(len(r) == 0 for r in dbg['hit_rows'])
    assert torch.isfinite(outs['all_bbox_preds']).all()
