"""Seeded synthetic weights and inputs for the fusion-decoder hot path.

Everything is drawn from ``numpy.random.RandomState`` (MT19937: bit-stable
across machines and numpy versions), so the authoring container, the CPU
tests and the GPU box regenerate identical tensors from seeds and only small
OUTPUT fixtures need to be committed.

The state-dict key list below is the checkpoint contract of the reference
head; tests/golden/make_golden.py asserts it equals
the key set / shapes of the reference's own ``Detr3DHead`` instance.
"""
import math

import numpy as np

from .configs import IMG_SHAPE, LEVEL_SHAPES, point_cloud_range

RADAR_CHANNELS = ['RADAR_FRONT', 'RADAR_FRONT_LEFT', 'RADAR_FRONT_RIGHT',
                  'RADAR_BACK_LEFT', 'RADAR_BACK_RIGHT']


# --------------------------------------------------------------------------
# state dict
# --------------------------------------------------------------------------
def _linear(spec, name, out_f, in_f):
    spec.append((name + '.weight', (out_f, in_f), 'w'))
    spec.append((name + '.bias', (out_f,), 'b'))


def _ln(spec, name, c):
    spec.append((name + '.weight', (c,), 'g'))
    spec.append((name + '.bias', (c,), 'b'))


def _mha(spec, name, c):
    spec.append((name + '.in_proj_weight', (3 * c, c), 'w'))
    spec.append((name + '.in_proj_bias', (3 * c,), 'b'))
    _linear(spec, name + '.out_proj', c, c)


def _cls_branch(spec, name, c, ncls):
    _linear(spec, name + '.0', c, c)
    _ln(spec, name + '.1', c)
    _linear(spec, name + '.3', c, c)
    _ln(spec, name + '.4', c)
    _linear(spec, name + '.6', ncls, c)


def _reg_branch(spec, name, c, code):
    _linear(spec, name + '.0', c, c)
    _linear(spec, name + '.2', c, c)
    _linear(spec, name + '.4', code, c)


def _pos_encoder(spec, name, c):
    _linear(spec, name + '.0', c, 3)
    _ln(spec, name + '.1', c)
    _linear(spec, name + '.3', c, c)
    _ln(spec, name + '.4', c)


def state_dict_spec(num_query=900, embed=256, ffn=512, num_layers=6,
                    num_classes=10, code_size=10, num_cams=6, num_levels=4,
                    radar_in=36):
    """[(key, shape, kind)] for ``pts_bbox_head.*`` (HEAD:43-238, XFMR:61-63,
    XFMR:280-292; mmcv brick names per SURVEY.md section 8(b))."""
    c = embed
    s = []
    s.append(('code_weights', (code_size,), 'code_weights'))
    s.append(('query_embedding.weight', (num_query, 2 * c), 'embed'))
    _linear(s, 'transformer.reference_points', 3, c)
    for i in range(num_layers):
        p = 'transformer.decoder.layers.%d.' % i
        _mha(s, p + 'attentions.0.attn', c)
        _linear(s, p + 'attentions.1.attention_weights',
                num_cams * num_levels, c)
        _linear(s, p + 'attentions.1.output_proj', c, c)
        _pos_encoder(s, p + 'attentions.1.position_encoder', c)
        _linear(s, p + 'ffns.0.layers.0.0', ffn, c)
        _linear(s, p + 'ffns.0.layers.1', c, ffn)
        for n in range(3):
            _ln(s, p + 'norms.%d' % n, c)
    for i in range(num_layers):
        _cls_branch(s, 'cls_branches.%d' % i, c, num_classes)
    for i in range(num_layers):
        _reg_branch(s, 'reg_branches.%d' % i, c, code_size)
    for sfx in ('', '2', '3'):
        _cls_branch(s, 'final_cls' + sfx, c, num_classes)
        _reg_branch(s, 'final_reg' + sfx, c, code_size)
        _mha(s, 'rf_multihead_attn' + sfx, c)
    for sfx in ('', '_2', '_3'):
        _linear(s, 'rf_linear1' + sfx, ffn, c)
        _linear(s, 'rf_linear2' + sfx, c, ffn)
        for n in (1, 2, 3):
            _ln(s, 'rf_norm%d%s' % (n, sfx), c)
    _pos_encoder(s, 'radar_position_encoder', c)
    _linear(s, 'radar_feat_encoder.0', 64, radar_in)
    _linear(s, 'radar_feat_encoder.2', 128, 64)
    _linear(s, 'radar_feat_encoder.4', c, 128)
    for n in (2, 3):
        _linear(s, 'attention_weights%d' % n, num_cams * num_levels, c)
        _linear(s, 'output_proj%d' % n, c, c)
    return s


def make_state_dict(seed=3, **dims):
    """Seeded weights as {key: float32 ndarray}.

    Xavier-uniform matrices, small random biases, LN gamma around 1.
    ``attention_weights`` is NOT zero (its reference init, XFMR:299, would
    null the whole sampling path and hide errors in it).  The last layer of
    every box-regression MLP is scaled by ``reg_out_scale`` (0.1): trained
    DETR3D refinements are small deltas, and with full xavier scale the
    ref-point -> sampling -> ref-point feedback has loop gain ~4 per decoder
    layer, which makes any two fp32 implementations drift apart by >1e-3
    (DESIGN.md "Conditioning of the parity rig")."""
    reg_out_scale = dims.pop('reg_out_scale', 0.1)
    rng = np.random.RandomState(seed)
    sd = {}
    for key, shape, kind in state_dict_spec(**dims):
        if kind == 'w':
            fan_out, fan_in = shape
            a = math.sqrt(6.0 / (fan_in + fan_out))
            if fan_in == 3:            # position encoders see O(1..50) inputs
                a = 0.5
            v = rng.uniform(-a, a, size=shape)
        elif kind == 'b':
            v = rng.standard_normal(shape) * 0.05
        elif kind == 'g':
            v = 1.0 + rng.standard_normal(shape) * 0.1
        elif kind == 'embed':
            v = rng.standard_normal(shape)
        elif kind == 'code_weights':
            v = np.array([1.0] * 8 + [0.2, 0.2])[:shape[0]]
        else:
            raise KeyError(kind)
        if ('reg_branches.' in key or key.startswith('final_reg')) \
                and '.4.' in key:
            v = v * reg_out_scale
        sd[key] = np.ascontiguousarray(v, dtype=np.float32)
    return sd


# --------------------------------------------------------------------------
# cameras / feature maps
# --------------------------------------------------------------------------
CAM_YAWS_DEG = (0.0, -55.0, 55.0, 180.0, 110.0, -110.0)


def make_lidar2img(focal=1266.0, pp=(800.0, 464.0), yaws_deg=CAM_YAWS_DEG,
                   jitter_seed=None):
    """[6,4,4] float64 pinhole cameras in a ring (nuScenes-like rig).

    Lidar frame: z up; camera ``i`` looks along yaw_i in the xy plane from a
    mount point 0.5..1.5 m off the origin."""
    K = np.eye(4)
    K[0, 0] = K[1, 1] = focal
    K[0, 2], K[1, 2] = pp
    rng = np.random.RandomState(jitter_seed) if jitter_seed is not None \
        else None
    out = []
    for i, yaw in enumerate(yaws_deg):
        a = math.radians(yaw)
        fwd = np.array([math.cos(a), math.sin(a), 0.0])
        right = np.array([math.sin(a), -math.cos(a), 0.0])
        down = np.array([0.0, 0.0, -1.0])
        R = np.stack([right, down, fwd])          # lidar -> camera axes
        t = fwd * (0.5 + 0.2 * i) + np.array([0.0, 0.0, 1.5])
        if rng is not None:
            t = t + rng.uniform(-0.1, 0.1, 3)
        E = np.eye(4)
        E[:3, :3] = R
        E[:3, 3] = -R @ t
        out.append(K @ E)
    return np.stack(out).astype(np.float64)


def _interp_field(ctrl, h, w):
    """Bilinear resampling (corner-aligned) of control grids [..., gh, gw]
    to [..., h, w]; float64 mult/add only, so bit-stable across machines."""
    gh, gw = ctrl.shape[-2:]
    ys = np.linspace(0.0, gh - 1.0, h) if h > 1 else np.array([(gh - 1) / 2.0])
    xs = np.linspace(0.0, gw - 1.0, w) if w > 1 else np.array([(gw - 1) / 2.0])
    y0 = np.minimum(np.floor(ys).astype(np.int64), gh - 2)
    x0 = np.minimum(np.floor(xs).astype(np.int64), gw - 2)
    wy = (ys - y0)[:, None]
    wx = (xs - x0)[None, :]
    a = ctrl[..., y0, :][..., :, x0]
    b = ctrl[..., y0, :][..., :, x0 + 1]
    c = ctrl[..., y0 + 1, :][..., :, x0]
    d = ctrl[..., y0 + 1, :][..., :, x0 + 1]
    return (a * (1 - wx) + b * wx) * (1 - wy) + (c * (1 - wx) + d * wx) * wy


def make_feats(level_shapes='res101', seed=1, batch=1, num_cams=6,
               channels=256, smooth=None):
    """list of [B, N, C, H, W] float32 maps (NCHW like the FPN output).

    ``smooth=None``: iid N(0,1) per pixel (the BASELINE.md bench input).
    ``smooth=(gh, gw)``: every level samples ONE smooth random field per
    (camera, channel) -- a gh x gw grid of N(0,1) control points spanning the
    image, bilinearly interpolated.  White-noise maps make the 6-layer
    decoder chaotic (fp32 vs fp64 of the same algorithm drift apart by >0.1
    in the reference points at the res101 shapes, DESIGN.md "Conditioning"),
    so the end-to-end parity rigs use the smooth fields, which is also what
    real FPN features look like."""
    if isinstance(level_shapes, str):
        level_shapes = LEVEL_SHAPES[level_shapes]
    rng = np.random.RandomState(seed)
    feats = []
    if smooth is None:
        for (h, w) in level_shapes:
            feats.append(rng.standard_normal(
                (batch, num_cams, channels, h, w)).astype(np.float32))
        return feats
    gh, gw = smooth
    ctrl = rng.standard_normal((batch, num_cams, channels, gh, gw))
    for (h, w) in level_shapes:
        lvl = np.empty((batch, num_cams, channels, h, w), dtype=np.float32)
        for b in range(batch):
            for n in range(num_cams):
                lvl[b, n] = _interp_field(ctrl[b, n], h, w)
        feats.append(lvl)
    return feats


def make_img_metas(batch=1, lidar2img=None, radar=None):
    if lidar2img is None:
        lidar2img = make_lidar2img()
    metas = []
    for b in range(batch):
        m = dict(lidar2img=[lidar2img[i] for i in range(lidar2img.shape[0])],
                 img_shape=[IMG_SHAPE] * lidar2img.shape[0],
                 sample_idx='synthetic-%d' % b,
                 box_type_3d=None)
        if radar is not None:
            m['radar'] = radar[b] if isinstance(radar, (list, tuple)) \
                else radar
        metas.append(m)
    return metas


# --------------------------------------------------------------------------
# radar
# --------------------------------------------------------------------------
def _rand_quat(rng):
    q = rng.standard_normal(4)
    return q / np.linalg.norm(q)


def make_radar_frame(seed=2, n_per_radar=51, centres=None, near_frac=0.8,
                     near_sigma=0.6):
    """Raw radar input of one frame in the nuScenes-devkit layout the
    reference consumes (HEAD:305-309, :498): per channel ``points`` [18,n]
    float64 and ``times`` [1,n]; sensor->ego rotation quaternions (wxyz).

    rows: x y z dyn_prop id rcs vx vy vx_comp vy_comp is_quality_valid
    ambig_state x_rms y_rms invalid_state pdh0 vx_rms vy_rms.
    If ``centres`` ([m,2] metres) is given, ``near_frac`` of the points are
    placed within ~``near_sigma`` m of randomly chosen centres so the
    distance-gated attention has hits to work on."""
    rng = np.random.RandomState(seed)
    frame = dict(points={}, times={}, radar_rot={}, lidar_rot=_rand_quat(rng))
    n_list = n_per_radar if isinstance(n_per_radar, (list, tuple)) \
        else [n_per_radar] * len(RADAR_CHANNELS)
    for chan, n in zip(RADAR_CHANNELS, n_list):
        p = np.zeros((18, n), dtype=np.float64)
        p[0] = rng.uniform(-50, 50, n)
        p[1] = rng.uniform(-50, 50, n)
        p[2] = rng.uniform(-1, 1, n)
        if centres is not None and n > 0:
            k = int(round(near_frac * n))
            pick = rng.randint(0, centres.shape[0], k)
            p[0, :k] = centres[pick, 0] + rng.standard_normal(k) * near_sigma
            p[1, :k] = centres[pick, 1] + rng.standard_normal(k) * near_sigma
        p[3] = rng.randint(0, 8, n)            # dyn_prop
        p[4] = rng.randint(0, 100, n)          # id
        p[5] = rng.uniform(-5, 30, n)          # rcs
        p[6:10] = rng.standard_normal((4, n)) * 3.0
        p[10] = rng.randint(0, 2, n)           # is_quality_valid
        p[11] = rng.randint(0, 5, n)           # ambig_state
        p[12:14] = rng.randint(0, 20, (2, n))
        p[14] = rng.randint(0, 18, n)          # invalid_state
        p[15] = rng.randint(0, 8, n)           # pdh0
        p[16:18] = rng.randint(0, 20, (2, n))
        t = rng.choice([0.0, 0.07, 0.14, 0.21, 0.28], size=(1, n))
        frame['points'][chan] = p
        frame['times'][chan] = t
        frame['radar_rot'][chan] = _rand_quat(rng)
    return frame


def make_gt(seed=7, n=24, num_classes=10):
    """Synthetic ground truth of one frame: boxes [n,9] = (x, y, z_bottom, w, l, h,
    yaw, vx, vy) in the LiDAR-box convention of mmdet3d, labels [n]."""
    rng = np.random.RandomState(seed)
    b = np.zeros((n, 9), dtype=np.float32)
    b[:, 0:2] = rng.uniform(-50, 50, (n, 2))
    b[:, 2] = rng.uniform(-2.5, 0.5, n)
    b[:, 3:6] = np.exp(rng.normal(0.5, 0.3, (n, 3)))
    b[:, 6] = rng.uniform(-np.pi, np.pi, n)
    b[:, 7:9] = rng.normal(0, 2, (n, 2))
    return b, rng.randint(0, num_classes, n).astype(np.int64)


def pc_range():
    return list(point_cloud_range)
