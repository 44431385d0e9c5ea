"""Radar ingest as a pipeline stage: the nuScenes radar .pcd
reader, the devkit's default filters, multi-sweep aggregation and the two
transforms.  The devkit is not part of the reference checkout, so there is no
golden vector here (parity unpinned, see transcar_amd/radar_pipeline.py): the
checks are a byte-level cross-decode with ``struct`` and transforms computed by
hand."""
import struct

import numpy as np
import pytest

from transcar_amd import radar as R
from transcar_amd import radar_pipeline as P


def synth_points(rs, n, all_valid=True):
    p = np.zeros((18, n))
    p[0:3] = rs.uniform(-40, 40, (3, n))
    p[2] = rs.uniform(-1, 1, n)
    p[3] = rs.randint(0, 7, n) if all_valid else rs.randint(0, 8, n)      # dyn_prop
    p[4] = rs.randint(0, 1000, n)                                         # id (int16)
    p[5] = rs.uniform(-5, 30, n)
    p[6:10] = rs.uniform(-10, 10, (4, n))
    p[10] = rs.randint(0, 2, n)
    p[11] = 3 if all_valid else rs.choice([1, 3, 4], n)                   # ambig_state
    p[12:14] = rs.randint(0, 20, (2, n))
    p[14] = 0 if all_valid else rs.choice([0, 0, 0, 4, 17], n)            # invalid_state
    p[15] = rs.randint(0, 8, n)
    p[16:18] = rs.randint(0, 20, (2, n))
    p[[0, 1, 2, 5, 6, 7, 8, 9]] = p[[0, 1, 2, 5, 6, 7, 8, 9]].astype(np.float32)   # what the file can hold
    return p


def test_pcd_round_trip_and_struct_cross_decode():
    rs = np.random.RandomState(0)
    pts = synth_points(rs, 57)
    raw = P.write_radar_pcd(pts)
    got = P.read_radar_pcd(raw)
    assert got.shape == (18, 57) and got.dtype == np.float64
    np.testing.assert_array_equal(got, pts)
    # decode three points the way the devkit does (struct, field by field)
    body = raw[raw.index(b'DATA binary\n') + len(b'DATA binary\n'):]
    fmt = 'fffbhfffffbbbbbbbb'
    assert struct.calcsize('<' + fmt) == 43
    for i in (0, 13, 56):
        one = struct.unpack('<' + fmt, body[43 * i:43 * (i + 1)])
        np.testing.assert_array_equal(np.array(one, dtype=np.float64), got[:, i])


def test_pcd_default_filters_and_disable():
    rs = np.random.RandomState(1)
    pts = synth_points(rs, 400, all_valid=False)
    raw = P.write_radar_pcd(pts)
    keep = np.isin(pts[14], [0]) & np.isin(pts[3], range(7)) & np.isin(pts[11], [3])
    assert 0 < keep.sum() < 400
    np.testing.assert_array_equal(P.read_radar_pcd(raw), pts[:, keep])
    np.testing.assert_array_equal(P.read_radar_pcd(raw, None, None, None), pts)


def test_pcd_empty_sweep_and_bad_header(tmp_path):
    pts = np.zeros((18, 1))
    pts[:3] = np.nan
    pts[11] = 3
    assert P.read_radar_pcd(P.write_radar_pcd(pts)).shape == (18, 0)
    raw = P.write_radar_pcd(synth_points(np.random.RandomState(2), 4))
    f = tmp_path / 'a.pcd'
    f.write_bytes(raw)
    assert P.read_radar_pcd(str(f)).shape == (18, 4)
    with pytest.raises(ValueError):
        P.read_radar_pcd(raw.replace(b'DATA binary', b'DATA ascii'))
    with pytest.raises(ValueError):
        P.read_radar_pcd(raw[:-10])


def rot_z(a):
    return np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])


def quat_z(a):
    return [np.cos(a / 2), 0.0, 0.0, np.sin(a / 2)]


def test_multisweep_transforms_by_hand():
    rs = np.random.RandomState(3)
    a_ref, a_cur, a_ego0, a_ego1 = 0.3, -1.1, 0.7, 0.75
    ref = dict(timestamp=1_000_000_000, sensor2ego_translation=[0.9, 0.0, 1.8], sensor2ego_rotation=quat_z(a_ref),
               ego2global_translation=[100.0, 50.0, 0.0], ego2global_rotation=quat_z(a_ego0))
    sweeps = []
    for i, (ts, ego_t, a_ego) in enumerate(((999_990_000, [100.1, 50.0, 0.0], a_ego0),
                                            (999_920_000, [99.2, 49.6, 0.0], a_ego1))):
        sweeps.append(dict(points=synth_points(rs, 20 + i), timestamp=ts,
                           sensor2ego_translation=[3.4, 0.2, 0.5], sensor2ego_rotation=quat_z(a_cur),
                           ego2global_translation=ego_t, ego2global_rotation=quat_z(a_ego)))
    pts, times = P.multisweep(sweeps, ref, nsweeps=5, min_distance=1.0)
    exp_pts, exp_t = [], []
    for sw, a_ego in zip(sweeps, (a_ego0, a_ego1)):
        p = sw['points'].copy()
        p = p[:, ~((np.abs(p[0]) < 1.0) & (np.abs(p[1]) < 1.0))]
        xyz = rot_z(a_cur) @ p[:3] + np.array(sw['sensor2ego_translation'])[:, None]          # sensor -> ego
        xyz = rot_z(a_ego) @ xyz + np.array(sw['ego2global_translation'])[:, None]             # ego -> global
        xyz = rot_z(a_ego0).T @ (xyz - np.array(ref['ego2global_translation'])[:, None])       # global -> ref ego
        xyz = rot_z(a_ref).T @ (xyz - np.array(ref['sensor2ego_translation'])[:, None])        # ego -> ref sensor
        p[:3] = xyz
        exp_pts.append(p)
        exp_t.append(np.full((1, p.shape[1]), 1e-6 * (ref['timestamp'] - sw['timestamp'])))
    np.testing.assert_allclose(pts, np.hstack(exp_pts), rtol=0, atol=1e-9)
    np.testing.assert_allclose(times, np.hstack(exp_t), rtol=0, atol=1e-12)
    np.testing.assert_array_equal(pts[3:], np.hstack(exp_pts)[3:])       # only xyz moves
    assert P.multisweep(sweeps, ref, nsweeps=1)[0].shape[1] == exp_pts[0].shape[1]


def test_remove_close():
    p = synth_points(np.random.RandomState(4), 6)
    p[0] = [0.5, 0.5, 2.0, -0.9, 0.0, 30.0]
    p[1] = [0.5, 2.0, 0.5, -0.9, 0.99, 0.0]
    ident = dict(timestamp=0, sensor2ego_translation=[0, 0, 0], sensor2ego_rotation=[1, 0, 0, 0],
                 ego2global_translation=[0, 0, 0], ego2global_rotation=[1, 0, 0, 0])
    out, _ = P.multisweep([dict(points=p, **ident)], ident)
    np.testing.assert_array_equal(out[0], [0.5, 2.0, 30.0])


class FakeNusc:
    """The four devkit tables radar_info_from_nusc walks."""

    def __init__(self, root, n_prev):
        self.dataroot = str(root)
        self.tables = {'sample': {}, 'sample_data': {}, 'calibrated_sensor': {}, 'ego_pose': {}}
        data = {}
        for ci, chan in enumerate(('LIDAR_TOP',) + R.RADAR_CHANNELS):
            self.tables['calibrated_sensor'][chan] = dict(translation=[0.5 * ci, 0.1, 0.4], rotation=quat_z(0.2 * ci))
            prev = ''
            for k in range(n_prev, -1, -1):                    # oldest first so that prev links exist
                tok = '%s_%d' % (chan, k)
                self.tables['ego_pose'][tok] = dict(translation=[10.0 - 0.5 * k, 0.1 * k, 0.0], rotation=quat_z(0.01 * k))
                self.tables['sample_data'][tok] = dict(calibrated_sensor_token=chan, ego_pose_token=tok, prev=prev,
                                                       filename='sweeps/%s.pcd' % tok, timestamp=2_000_000 - 70_000 * k)
                prev = tok
            data[chan] = '%s_0' % chan
        self.tables['sample']['s0'] = dict(data=data)

    def get(self, table, token):
        return self.tables[table][token]


def test_pipeline_end_to_end(tmp_path):
    rs = np.random.RandomState(5)
    nusc = FakeNusc(tmp_path, n_prev=2)                        # 3 sweeps exist, 5 asked for
    (tmp_path / 'sweeps').mkdir()
    raw_pts = {}
    for tok, sd in nusc.tables['sample_data'].items():
        if tok.startswith('RADAR'):
            raw_pts[tok] = synth_points(rs, rs.randint(5, 40), all_valid=False)
            (tmp_path / sd['filename']).write_bytes(P.write_radar_pcd(raw_pts[tok]))
    info = P.radar_info_from_nusc(nusc, 's0', nsweeps=5)
    assert [len(info['radars'][c]) for c in R.RADAR_CHANNELS] == [3] * 5
    assert info['radars']['RADAR_FRONT'][0]['data_path'].endswith('sweeps/RADAR_FRONT_0.pcd')
    res = P.LoadRadarPointsMultiSweep(nsweeps=5)({'radar_info': info})
    radar = res['radar']
    assert set(radar['points']) == set(R.RADAR_CHANNELS)
    for chan in R.RADAR_CHANNELS:
        n = radar['points'][chan].shape[1]
        assert radar['times'][chan].shape == (1, n)
        lags = np.unique(np.round(radar['times'][chan], 6))
        assert set(lags) <= {0.0, 0.07, 0.14}
        assert radar['radar_rot'][chan] == nusc.tables['calibrated_sensor'][chan]['rotation']
    assert radar['lidar_rot'] == nusc.tables['calibrated_sensor']['LIDAR_TOP']['rotation']
    feats = P.BuildRadarFeatures()(dict(radar=radar))['radar']
    assert feats.ndim == 2 and feats.shape[1] == R.NUM_FEATURES
    np.testing.assert_array_equal(feats, R.build_radar_features(radar))
    tokens, pad_mult = R.pack_tokens([feats])
    assert tokens.shape[0] == 1 and tokens.shape[2] == 36 and pad_mult == R.NUM_RADAR_TOKENS - tokens.shape[1] + 1
    # filters off: more points survive
    res2 = P.LoadRadarPointsMultiSweep(nsweeps=5, disable_filters=True)({'radar_info': info})
    assert sum(v.shape[1] for v in res2['radar']['points'].values()) > sum(v.shape[1] for v in radar['points'].values())
    assert 'LoadRadarPointsMultiSweep' in P.PIPELINES.module_dict and 'BuildRadarFeatures' in P.PIPELINES.module_dict
