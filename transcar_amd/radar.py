"""Radar ingest of the fusion decoder -- host side, as in the reference, which
runs it in numpy on the main thread inside ``Detr3DHead.forward``
(HEAD:301-536).  The reference pulls the sweeps from the nuScenes devkit by
``sample_idx``; here the same raw arrays arrive through
``img_metas[i]['radar']``, produced by the data
pipeline:

    radar = dict(points={chan: [18,n] float64}, times={chan: [1,n]},
                 radar_rot={chan: wxyz}, lidar_rot=wxyz)

or an already-built ``[n,36]`` feature array.  The 36-column layout is the
contract of ``radar_feat_encoder`` (HEAD:183, 499-510).
"""
import numpy as np

RADAR_CHANNELS = ('RADAR_FRONT', 'RADAR_FRONT_LEFT', 'RADAR_FRONT_RIGHT',
                  'RADAR_BACK_LEFT', 'RADAR_BACK_RIGHT')
POINT_RANGE = (-51.2, -51.2, -5.0, 51.2, 51.2, 3.0)      # HEAD:304
NUM_RADAR_TOKENS = 1500                                   # HEAD:526
PAD_VALUE = 500.0                                         # HEAD:527
#: raw-row columns copied verbatim: x y z id rcs is_quality_valid invalid_state
BASE_COLUMNS = (0, 1, 2, 4, 5, 10, 14)                    # HEAD:499
NUM_FEATURES = 36


def quaternion_rotation_matrix(q):
    """Rotation matrix of a (w, x, y, z) quaternion (pyquaternion semantics:
    the quaternion is normalised first)."""
    q = np.asarray(q, dtype=np.float64)
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _one_hot(idx, width):
    out = np.zeros((idx.shape[0], width))
    out[np.arange(idx.shape[0]), idx] = 1.0
    return out


def _channel_features(points, times, rot_radar, rot_ref):
    n = points.shape[1]

    def to_lidar(v_xy):                                   # HEAD:317-327
        v = np.vstack((v_xy, np.zeros(n)))
        v = rot_ref.T @ (rot_radar @ v)
        v[2, :] = 0.0
        return v.T[:, :2]
    v_comp = to_lidar(points[8:10, :])
    v_raw = to_lidar(points[6:8, :])
    rows = points.T
    if times.shape[1] != 0:                               # HEAD:453-455
        times = times - np.max(times)
    t2 = np.repeat(times.T, 2, axis=1)
    return np.concatenate(
        (rows[:, list(BASE_COLUMNS)], t2, v_comp * t2, v_comp, v_raw,
         _one_hot(rows[:, 3].astype(int), 8),             # dyn_prop
         _one_hot(rows[:, 11].astype(int), 5),            # ambig_state
         _one_hot(rows[:, 15].astype(int), 8)), axis=1)   # pdh0


def build_radar_features(radar, point_range=POINT_RANGE):
    """Raw multi-sweep radar of one sample -> [n_kept, 36] float64."""
    if isinstance(radar, np.ndarray):
        assert radar.ndim == 2 and radar.shape[1] == NUM_FEATURES
        return radar
    rot_ref = quaternion_rotation_matrix(radar['lidar_rot'])
    per_chan = []
    for chan in RADAR_CHANNELS:
        pts = np.asarray(radar['points'][chan], dtype=np.float64)
        tms = np.asarray(radar['times'][chan], dtype=np.float64)
        per_chan.append(_channel_features(
            pts, tms, quaternion_rotation_matrix(radar['radar_rot'][chan]),
            rot_ref))
    allp = np.concatenate(per_chan, axis=0)
    lo, hi = np.asarray(point_range[:3]), np.asarray(point_range[3:])
    keep = np.all(allp[:, :3] > lo, axis=1) & np.all(allp[:, :3] < hi, axis=1)
    return allp[keep]


def pack_tokens(feature_list, granule=64, T=None):
    """Batch of [n_i,36] arrays -> (tokens [B,T,36] float32, pad_mult).

    T: fixed token count (a frame pipeline's static ``tokens`` tensor has one
    shape for every frame: T and pad_mult are baked into its captured graph);
    None picks the smallest multiple of ``granule`` that holds the frame.

    The reference always attends over 1500 tokens, the unused ones filled
    with 500.0 (HEAD:526-530).  Identical pad tokens have identical keys and
    values, so only T = round_up(max n_i + 1, granule) tokens are
    materialised and the last one carries the multiplicity of the missing
    1500 - T pad tokens (see tc_radar_gated_xattn_fwd)."""
    n_max = max(min(f.shape[0], NUM_RADAR_TOKENS) for f in feature_list)
    if T is None:
        T = min(NUM_RADAR_TOKENS, ((n_max + 1 + granule - 1) // granule) * granule)
    elif not (1 <= T <= NUM_RADAR_TOKENS) or (n_max + 1 > T and T < NUM_RADAR_TOKENS):
        raise ValueError('T=%d cannot hold %d radar points + one pad token (max %d)'
                         % (T, n_max, NUM_RADAR_TOKENS))
    tokens = np.full((len(feature_list), T, NUM_FEATURES), PAD_VALUE,
                     dtype=np.float32)
    for b, f in enumerate(feature_list):
        n = min(f.shape[0], T)
        tokens[b, :n] = f[:n].astype(np.float32)
    return tokens, NUM_RADAR_TOKENS - T + 1
