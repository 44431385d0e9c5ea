"""Radar ingest as a data-pipeline stage.

The reference reads the radar sweeps INSIDE ``Detr3DHead.forward`` on the main
thread: five ``RadarPointCloud.from_file_multisweep(nusc, sample, chan,
ref_chan='LIDAR_TOP', nsweeps=5)`` calls per frame (HEAD:301-309), i.e. 25 file
reads, the nuScenes table look-ups and the numpy feature build, while the GPU
waits.  Here that work is a pipeline transform that runs in the data-loader
workers and hands ``img_metas['radar']`` to the head:

    LoadRadarPointsMultiSweep   info['radars'] -> results['radar'] (raw sweeps)
    BuildRadarFeatures          results['radar'] -> [n,36] features (HEAD:311-524)

``from_file`` / ``from_file_multisweep`` belong to the nuScenes devkit
(``nuscenes-devkit``, python-sdk/nuscenes/utils/data_classes.py; the reference
pins no version -- behaviour restated from v1.1.x), which is not part of
/root/reference: PARITY UNPINNED for this file, it is tested against synthetic
PCD files and hand-computed transforms (tests/test_radar_pipeline.py).  The
file format is the devkit's radar ``.pcd``: PCD v0.7, ``DATA binary``, 18 fields

    x y z dyn_prop id rcs vx vy vx_comp vy_comp is_quality_valid ambig_state
    x_rms y_rms invalid_state pdh0 vx_rms vy_rms
    F F F I        I  F   F  F  F       F       I                I
    I     I     I             I    I      I            (43 bytes per point)
"""
import os

import numpy as np

from . import radar as R
from .registry import Registry

PIPELINES = Registry('pipeline', mm_path=('mmdet.datasets.builder', 'PIPELINES'))

RADAR_FIELDS = ('x', 'y', 'z', 'dyn_prop', 'id', 'rcs', 'vx', 'vy', 'vx_comp', 'vy_comp',
                'is_quality_valid', 'ambig_state', 'x_rms', 'y_rms', 'invalid_state', 'pdh0',
                'vx_rms', 'vy_rms')
_NP_TYPES = {('F', 2): 'f2', ('F', 4): 'f4', ('F', 8): 'f8',
             ('I', 1): 'i1', ('I', 2): 'i2', ('I', 4): 'i4', ('I', 8): 'i8',
             ('U', 1): 'u1', ('U', 2): 'u2', ('U', 4): 'u4', ('U', 8): 'u8'}
#: the devkit's default filters (RadarPointCloud.default_filters)
INVALID_STATES = (0,)
DYNPROP_STATES = tuple(range(7))
AMBIG_STATES = (3,)


def read_radar_pcd(src, invalid_states=INVALID_STATES, dynprop_states=DYNPROP_STATES,
                   ambig_states=AMBIG_STATES):
    """One radar sweep -> [18, n] float64 (devkit ``RadarPointCloud.from_file``).

    src: path or bytes.  The three state filters are applied in the devkit's
    order; ``None`` disables a filter (``RadarPointCloud.disable_filters``)."""
    if isinstance(src, (bytes, bytearray, memoryview)):
        raw = bytes(src)
    else:
        with open(src, 'rb') as f:
            raw = f.read()
    meta, pos = [], 0
    while True:
        end = raw.find(b'\n', pos)
        if end < 0:
            raise ValueError('radar pcd: no DATA line')
        line = raw[pos:end].strip().decode('utf-8')
        pos = end + 1
        meta.append(line)
        if line.startswith('DATA'):
            break
    if not (meta[0].startswith('#') and meta[1].startswith('VERSION')):
        raise ValueError('radar pcd: unexpected header')
    hdr = {m.split(' ')[0]: m.split(' ')[1:] for m in meta[1:]}
    sizes = [int(s) for s in hdr['SIZE']]
    types = hdr['TYPE']
    counts = [int(c) for c in hdr['COUNT']]
    width, height = int(hdr['WIDTH'][0]), int(hdr['HEIGHT'][0])
    if any(c != 1 for c in counts):
        raise ValueError('radar pcd: COUNT != 1 is not supported')
    if height != 1 or width <= 0:
        raise ValueError('radar pcd: WIDTH=%d HEIGHT=%d' % (width, height))
    if hdr['DATA'][0] != 'binary':
        raise ValueError('radar pcd: DATA %s (binary expected)' % hdr['DATA'][0])
    nf = len(types)
    dt = np.dtype({'names': ['f%d' % i for i in range(nf)],
                   'formats': ['<' + _NP_TYPES[(t, s)] for t, s in zip(types, sizes)],
                   'offsets': list(np.cumsum([0] + sizes[:-1])), 'itemsize': int(sum(sizes))})
    body = raw[pos:]
    if len(body) < width * dt.itemsize:
        raise ValueError('radar pcd: %d bytes of data for %d points' % (len(body), width))
    rec = np.frombuffer(body, dtype=dt, count=width)
    points = np.stack([rec['f%d' % i].astype(np.float64) for i in range(nf)], axis=0)
    if np.any(np.isnan(points[:, 0])):           # a NaN in the first point: empty sweep
        return np.zeros((nf, 0))
    for row, allowed in ((nf - 4, invalid_states), (3, dynprop_states), (11, ambig_states)):
        if allowed is not None:
            points = points[:, np.isin(points[row, :], np.asarray(allowed, dtype=np.float64))]
    return points


def write_radar_pcd(points):
    """[18, n] -> bytes of a devkit-style radar .pcd (tests, synthetic data)."""
    points = np.asarray(points, dtype=np.float64)
    assert points.shape[0] == len(RADAR_FIELDS)
    types = 'F F F I I F F F F F I I I I I I I I'.split()
    sizes = [4, 4, 4, 1, 2, 4, 4, 4, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1]
    n = points.shape[1]
    head = ['# .PCD v0.7 - Point Cloud Data file format', 'VERSION 0.7',
            'FIELDS ' + ' '.join(RADAR_FIELDS), 'SIZE ' + ' '.join(str(s) for s in sizes),
            'TYPE ' + ' '.join(types), 'COUNT ' + ' '.join('1' for _ in sizes),
            'WIDTH %d' % n, 'HEIGHT 1', 'VIEWPOINT 0 0 0 1 0 0 0', 'POINTS %d' % n, 'DATA binary']
    dt = np.dtype({'names': list(RADAR_FIELDS),
                   'formats': ['<' + _NP_TYPES[(t, s)] for t, s in zip(types, sizes)],
                   'offsets': list(np.cumsum([0] + sizes[:-1])), 'itemsize': int(sum(sizes))})
    rec = np.zeros(n, dtype=dt)
    for i, name in enumerate(RADAR_FIELDS):
        rec[name] = points[i].astype(dt[name])
    return ('\n'.join(head) + '\n').encode('utf-8') + rec.tobytes()


def transform_matrix(translation, rotation_wxyz, inverse=False):
    """4x4 homogeneous transform (devkit ``geometry_utils.transform_matrix``)."""
    rot = R.quaternion_rotation_matrix(rotation_wxyz)
    tr = np.asarray(translation, dtype=np.float64)
    tm = np.eye(4)
    if inverse:
        tm[:3, :3] = rot.T
        tm[:3, 3] = rot.T.dot(-tr)
    else:
        tm[:3, :3] = rot
        tm[:3, 3] = tr
    return tm


def multisweep(sweeps, ref, read=read_radar_pcd, nsweeps=5, min_distance=1.0):
    """Aggregate the sweeps of one radar channel in the reference sensor's frame
    (devkit ``PointCloud.from_file_multisweep``).

    sweeps: newest first, each ``dict(data_path | points, timestamp [us],
            sensor2ego_translation, sensor2ego_rotation (wxyz),
            ego2global_translation, ego2global_rotation)``
    ref:    the same keys for the reference sensor (LIDAR_TOP) at the key frame
    Returns (points [18, n] float64, times [1, n]) -- xyz in the reference
    frame at the reference time, ``times`` = lag of the sweep in seconds.
    Velocities are NOT rotated here (the head does that, HEAD:317-327)."""
    ref_from_car = transform_matrix(ref['sensor2ego_translation'], ref['sensor2ego_rotation'], inverse=True)
    car_from_global = transform_matrix(ref['ego2global_translation'], ref['ego2global_rotation'], inverse=True)
    ref_time = 1e-6 * ref['timestamp']
    all_pts = np.zeros((len(RADAR_FIELDS), 0))
    all_times = np.zeros((1, 0))
    for sw in sweeps[:nsweeps]:
        pts = np.array(sw['points'], dtype=np.float64) if 'points' in sw else read(sw['data_path'])
        close = (np.abs(pts[0, :]) < min_distance) & (np.abs(pts[1, :]) < min_distance)
        pts = pts[:, ~close]                                      # PointCloud.remove_close
        global_from_car = transform_matrix(sw['ego2global_translation'], sw['ego2global_rotation'])
        car_from_cur = transform_matrix(sw['sensor2ego_translation'], sw['sensor2ego_rotation'])
        tm = ref_from_car.dot(car_from_global).dot(global_from_car).dot(car_from_cur)
        pts[:3, :] = tm[:3, :3].dot(pts[:3, :]) + tm[:3, 3:4]
        lag = ref_time - 1e-6 * sw['timestamp']
        all_times = np.hstack((all_times, lag * np.ones((1, pts.shape[1]))))
        all_pts = np.hstack((all_pts, pts))
    return all_pts, all_times


def radar_info_from_nusc(nusc, sample_token, nsweeps=5, ref_chan='LIDAR_TOP'):
    """Data-converter helper: the sweep lists ``LoadRadarPointsMultiSweep`` needs,
    from a nuScenes devkit object (anything with the devkit's ``get`` and
    ``dataroot``).  Walks ``sample_data.prev`` like from_file_multisweep."""
    sample = nusc.get('sample', sample_token)

    def sd_info(sd):
        cs = nusc.get('calibrated_sensor', sd['calibrated_sensor_token'])
        pose = nusc.get('ego_pose', sd['ego_pose_token'])
        return dict(data_path=os.path.join(nusc.dataroot, sd['filename']), timestamp=sd['timestamp'],
                    sensor2ego_translation=cs['translation'], sensor2ego_rotation=cs['rotation'],
                    ego2global_translation=pose['translation'], ego2global_rotation=pose['rotation'])
    info = dict(ref=sd_info(nusc.get('sample_data', sample['data'][ref_chan])), radars={})
    for chan in R.RADAR_CHANNELS:
        sd = nusc.get('sample_data', sample['data'][chan])
        sweeps = []
        for _ in range(nsweeps):
            sweeps.append(sd_info(sd))
            if sd['prev'] == '':
                break
            sd = nusc.get('sample_data', sd['prev'])
        info['radars'][chan] = sweeps
    return info


@PIPELINES.register_module(export=True)
class LoadRadarPointsMultiSweep:
    """results['radar_info'] (see radar_info_from_nusc) -> results['radar'], the raw
    multi-sweep arrays ``transcar_amd.radar.build_radar_features`` consumes."""

    def __init__(self, nsweeps=5, min_distance=1.0, disable_filters=False):
        self.nsweeps, self.min_distance = nsweeps, min_distance
        self.disable_filters = disable_filters

    def _read(self, path):
        if self.disable_filters:
            return read_radar_pcd(path, None, None, None)
        return read_radar_pcd(path)

    def __call__(self, results):
        info = results['radar_info']
        out = dict(points={}, times={}, radar_rot={}, lidar_rot=info['ref']['sensor2ego_rotation'])
        for chan in R.RADAR_CHANNELS:
            sweeps = info['radars'][chan]
            pts, times = multisweep(sweeps, info['ref'], self._read, self.nsweeps, self.min_distance)
            out['points'][chan], out['times'][chan] = pts, times
            out['radar_rot'][chan] = sweeps[0]['sensor2ego_rotation']     # HEAD:311-315: key-frame calibration
        results['radar'] = out
        return results


@PIPELINES.register_module(export=True)
class BuildRadarFeatures:
    """results['radar'] raw sweeps -> the [n,36] feature array (HEAD:311-524), so the
    head's forward only pads, uploads and attends."""

    def __init__(self, point_range=R.POINT_RANGE):
        self.point_range = tuple(point_range)

    def __call__(self, results):
        results['radar'] = R.build_radar_features(results['radar'], self.point_range)
        return results
