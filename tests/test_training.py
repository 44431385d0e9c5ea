"""Training row: Hungarian assignment + focal / L1
losses against the reference's own `Detr3DHead.loss` (fixture G7, produced by
tests/golden/make_golden.py::g7_loss) -- CPU, host/PyTorch code as in the
reference."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import transcar_amd as T
from oracle import transcar_oracle as O
from transcar_amd import configs, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ['loss_cls', 'loss_bbox', 'd0.loss_cls', 'd0.loss_bbox', 'd1.loss_cls', 'd1.loss_bbox']


def _outs(golden_dir):
    g5 = np.load(os.path.join(golden_dir, 'g5_head_tiny.npz'))
    return {'all_cls_scores': torch.from_numpy(g5['all_cls_scores']),
            'all_bbox_preds': torch.from_numpy(g5['all_bbox_preds'])}


def _head():
    cfg = configs.head_cfg()
    cfg['train_cfg'] = configs.train_cfg_pts
    h = T.build_head(cfg)
    h.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(3).items()})
    return h


def test_oracle_loss_matches_reference(golden_dir):
    g7 = np.load(os.path.join(golden_dir, 'g7_loss.npz'))
    boxes, labels = synth.make_gt(seed=7, n=24)
    cw = torch.tensor([1.0] * 8 + [0.2, 0.2])
    res, matches = O.loss(_outs(golden_dir), torch.from_numpy(boxes), torch.from_numpy(labels), cw)
    for k in KEYS:
        assert abs(float(res[k]) - float(g7[k.replace('.', '_')])) <= 1e-4 * max(1, abs(float(g7[k.replace('.', '_')]))), k
    assert np.array_equal(np.stack([m.numpy() for m in matches]), g7['gt_inds'])


def test_head_loss_matches_reference(golden_dir):
    g7 = np.load(os.path.join(golden_dir, 'g7_loss.npz'))
    boxes, labels = synth.make_gt(seed=7, n=24)
    gt = torch.from_numpy(boxes).clone()
    gt[:, 2] += gt[:, 5] * 0.5                     # gravity centre, HEAD:963-965
    h = _head()
    out = h.loss([gt], [torch.from_numpy(labels)], _outs(golden_dir))
    assert sorted(out) == sorted(KEYS)
    for k in KEYS:
        ref = float(g7[k.replace('.', '_')])
        assert abs(float(out[k]) - ref) <= 1e-5 * max(1, abs(ref)), (k, float(out[k]), ref)
    # the assigner itself (ASSIGN:52-134)
    res = h.assigner.assign(_outs(golden_dir)['all_bbox_preds'][2, 0],
                            _outs(golden_dir)['all_cls_scores'][2, 0], gt, torch.from_numpy(labels))
    assert np.array_equal(res.gt_inds.numpy(), g7['gt_inds'][2])
    assert int((res.gt_inds > 0).sum()) == 24


def test_head_loss_without_ground_truth(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g7_loss_empty.npz'))
    h = _head()
    out = h.loss([torch.zeros(0, 9)], [torch.zeros(0, dtype=torch.long)], _outs(golden_dir))
    for k in KEYS:
        ref = float(g[k.replace('.', '_')])
        assert abs(float(out[k]) - ref) <= 1e-5 * max(1, abs(ref)), k
    assert float(out['loss_bbox']) == 0.0


def test_accepts_lidar_box_objects(golden_dir):
    class Boxes:                                    # what mmdet3d hands over
        def __init__(self, t):
            self.tensor = t

        @property
        def gravity_center(self):
            c = self.tensor[:, :3].clone()
            c[:, 2] += self.tensor[:, 5] * 0.5
            return c
    boxes, labels = synth.make_gt(seed=7, n=24)
    g7 = np.load(os.path.join(golden_dir, 'g7_loss.npz'))
    out = _head().loss([Boxes(torch.from_numpy(boxes))], [torch.from_numpy(labels)], _outs(golden_dir))
    assert abs(float(out['loss_cls']) - float(g7['loss_cls'])) < 1e-3


_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
import transcar_amd as T
from transcar_amd import configs, synth, dist as D
rank, world = D.init_process_group('gloo')
cfg = configs.head_cfg(); cfg['train_cfg'] = configs.train_cfg_pts
h = T.build_head(cfg)
g5 = np.load(os.path.join(%r, 'g5_head_tiny.npz'))
outs = {'all_cls_scores': torch.from_numpy(g5['all_cls_scores']), 'all_bbox_preds': torch.from_numpy(g5['all_bbox_preds'])}
n = 24 if rank == 0 else 8                       # different number of positives per rank
b, l = synth.make_gt(seed=7, n=24)
gt = torch.from_numpy(b[:n]).clone(); gt[:, 2] += gt[:, 5] * 0.5
out = h.loss([gt], [torch.from_numpy(l[:n])], outs)
# single-process value with the averaged normaliser (HEAD:889-902: reduce_mean)
h.sync_cls_avg_factor = False
solo = h.loss([gt], [torch.from_numpy(l[:n])], outs)
ratio = float(out['loss_cls']) / float(solo['loss_cls'])
assert abs(ratio - n / 16.0) < 1e-4, ratio          # mean positives over ranks = 16
dist.destroy_process_group()
print('ok', rank)
'''


def test_loss_normalisers_all_reduce_gloo_world2(tmp_path, golden_dir):
    script = tmp_path / 'w.py'
    script.write_text(_WORKER % (ROOT, golden_dir))
    port = 31000 + os.getpid() % 2000
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out.decode()


# ---------------------------------------------------------------------------
# gradients of one training iteration (fixture G8 = the reference's backward)
# ---------------------------------------------------------------------------
FROZEN = ('transformer.', 'cls_branches.', 'reg_branches.', 'query_embedding.', 'code_weights')


def trainable(name):
    """tools/train.py:245-252 freezes the DETR3D part of the head."""
    return not name.startswith(FROZEN)


def g8_name(tag='tiny'):
    return 'g8_train_grads.npz' if tag == 'tiny' else 'g8_train_grads_%s.npz' % tag


def g8_inputs(golden_dir, tag='tiny'):
    """tag: 'tiny' FPN shapes, or 'res101' = the ResNet-101 shapes of BASELINE.json configs[2]"""
    g5 = np.load(os.path.join(golden_dir, 'g5_head_%s.npz' % tag))
    feats = synth.make_feats(tag, seed=1, smooth=(4, 6))
    l2i = synth.make_lidar2img()
    frame = synth.make_radar_frame(seed=2, n_per_radar=51, centres=g5['radar_centres'])
    boxes, labels = synth.make_gt(seed=7, n=24)
    return feats, l2i, frame, boxes, labels


def check_grads_against_g8(grads, g8, rtol, what):
    """grads: {state_dict key: tensor or None}."""
    checked = 0
    for k, g in grads.items():
        key = k.replace('.', '__')
        if key + '__none' in g8.files:
            assert g is None or float(g.abs().max()) == 0.0, k
            continue
        ref = g8[key + '__stats']
        gd = g.detach().double().flatten().cpu()
        got = np.array([gd.sum(), gd.abs().sum(), gd.norm()])
        scale = ref[1]                      # sum |g|: the natural magnitude for all three
        assert abs(got[1] - ref[1]) <= rtol * scale, (what, k, got, ref)
        assert abs(got[2] - ref[2]) <= rtol * max(ref[2], 1e-12), (what, k, got, ref)
        assert abs(got[0] - ref[0]) <= rtol * scale, (what, k, got, ref)
        head = g8[key + '__head']
        tol = rtol * max(np.abs(head).max(), ref[2] / np.sqrt(gd.numel()))
        assert np.abs(gd[:16].float().numpy() - head).max() <= 4 * tol, (what, k)
        checked += 1
    assert checked == 98
    return checked


@pytest.mark.parametrize('tag', ['tiny', 'res101', 'vovnet'])
def test_oracle_backward_matches_reference(golden_dir, tag):
    g8 = np.load(os.path.join(golden_dir, g8_name(tag)))
    feats, l2i, frame, boxes, labels = g8_inputs(golden_dir, tag)
    sd = O.to_torch_sd(synth.make_state_dict(3))
    for k, v in sd.items():
        if trainable(k):
            v.requires_grad_(True)
    f36 = O.build_radar_features(frame)
    outs = O.head_forward(sd, [torch.from_numpy(f) for f in feats],
                          torch.from_numpy(l2i).float()[None], configs.IMG_SHAPE[:2], f36,
                          configs.point_cloud_range)
    assert np.abs(outs['all_cls_scores'].detach().numpy() - g8['all_cls_scores']).max() < 5e-4
    res, _ = O.loss(outs, torch.from_numpy(boxes), torch.from_numpy(labels), sd['code_weights'])
    total = sum(res.values())
    assert abs(float(total) - float(g8['total_loss'])) < 1e-4 * float(g8['total_loss'])
    total.backward()
    check_grads_against_g8({k: v.grad for k, v in sd.items() if trainable(k)}, g8, 2e-3, 'oracle')


# ---------------------------------------------------------------------------
# flat gradient bucket + schedule (host logic of transcar_amd/trainer.py)
# ---------------------------------------------------------------------------
_BUCKET_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from transcar_amd import dist as D
from transcar_amd.trainer import FlatBucket
rank, world = D.init_process_group('gloo')
torch.manual_seed(0)
m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
b = FlatBucket(list(m.named_parameters()))
assert b.numel == 8 * 16 + 16 + 16 * 4 + 4
x = torch.full((3, 8), float(rank + 1))
b.zero_grad()
m(x).sum().backward()
local = b.grads.clone()
assert float(local.abs().sum()) > 0                  # autograd accumulated INTO the flat views
for p, off in zip(b.items, b.offsets):
    assert p.grad.data_ptr() == b.grads.data_ptr() + 4 * off
    assert p.data_ptr() == b.params.data_ptr() + 4 * off
assert b.all_reduce() == 2
parts = [torch.zeros_like(local) for _ in range(2)]
dist.all_gather(parts, local)
assert torch.allclose(b.grads, parts[0] + parts[1])
dist.destroy_process_group()
'''


def test_flat_bucket_single_all_reduce_gloo_world2(tmp_path):
    script = tmp_path / 'b.py'
    script.write_text(_BUCKET_WORKER % ROOT)
    port = 33000 + os.getpid() % 2000
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out.decode()


_CHUNK_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from transcar_amd import dist as D
from transcar_amd.trainer import FlatBucket, exchange_chunk_of
rank, world = D.init_process_group('gloo')
# the names of the fusion stack's trainable modules (HEAD:74-189), in named_parameters order: NOT grouped by layer
names = ['final_cls.0.weight', 'final_cls2.0.weight', 'final_cls3.0.weight', 'final_reg.4.bias', 'final_reg3.4.bias',
         'rf_multihead_attn.in_proj_weight', 'rf_multihead_attn2.in_proj_weight', 'rf_multihead_attn3.out_proj.bias',
         'rf_linear1.weight', 'rf_linear1_2.weight', 'rf_linear2_3.bias', 'rf_norm2.weight', 'rf_norm3_2.bias', 'rf_norm3_3.bias',
         'radar_position_encoder.0.weight', 'radar_feat_encoder.4.bias']
assert [exchange_chunk_of(n) for n in names] == [2, 1, 0, 2, 0, 2, 1, 0, 2, 1, 0, 2, 1, 0, 3, 3]
torch.manual_seed(0)
params = [(n, torch.nn.Parameter(torch.randn(3 + i))) for i, n in enumerate(names)]
b = FlatBucket(params, chunk_of=exchange_chunk_of)
# four contiguous chunks that tile the bucket, chunk 0 (fusion layer 3) first; offsets stay in the caller's order
assert len(b.chunk_ranges) == 4 and b.chunk_ranges[0][0] == 0 and b.chunk_ranges[-1][1] == b.numel
for (a0, a1), (b0, b1) in zip(b.chunk_ranges, b.chunk_ranges[1:]):
    assert a1 == b0 and a1 > a0
for (n, p), off in zip(params, b.offsets):
    ch = exchange_chunk_of(n)
    assert b.chunk_ranges[ch][0] <= off and off + p.numel() <= b.chunk_ranges[ch][1]
    assert p.grad.data_ptr() == b.grads.data_ptr() + 4 * off and p.data_ptr() == b.params.data_ptr() + 4 * off
b.zero_grad()
b.grads += float(rank + 1)
calls = []
real = dist.all_reduce
def counting(t, *a, **kw):
    calls.append((int(t.numel()), bool(kw.get('async_op', False))))
    return real(t, *a, **kw)
dist.all_reduce = counting
pending = []
for i in range(4):
    pending.append(b.all_reduce_chunk_begin(i))
    if i < 3:                                       # chunk i is summed (or on its way), the later chunks are untouched so far
        nxt = b.chunk_ranges[i + 1]
        assert float(b.grads[nxt[0]:nxt[1]].max()) == float(rank + 1)
assert b.all_reduce_end(pending) == 2
dist.all_reduce = real
assert [c[0] for c in calls] == [e - a for a, e in b.chunk_ranges] and all(c[1] for c in calls)
assert float(b.grads.min()) == 3.0 and float(b.grads.max()) == 3.0      # 1 + 2 everywhere, every element exactly once
dist.destroy_process_group()
'''


def test_flat_bucket_exchange_in_chunks_gloo_world2(tmp_path):
    """Round 6: the bucket is laid out in exchange chunks (fusion layer 3, 2, 1, encoders) and
    `all_reduce_chunk_begin` sums one chunk at a time, asynchronously -- four collectives that together are the one
    all-reduce of the flat bucket (tools/train.py:253-260: DDP's buckets)."""
    script = tmp_path / 'c.py'
    script.write_text(_CHUNK_WORKER % ROOT)
    port = 35000 + os.getpid() % 2000
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out.decode()


def test_cosine_schedule_with_linear_warmup():
    from transcar_amd.trainer import cosine_lr
    base = 1.5e-5                                               # CFG:208
    assert abs(cosine_lr(base, 0, 0, 24) - base / 3) < 1e-12    # warmup_ratio 1/3
    assert abs(cosine_lr(base, 4000, 0, 24) - base) < 1e-12
    assert abs(cosine_lr(base, 10 ** 6, 24, 24) - base * 1e-3) < 1e-12
    assert cosine_lr(base, 10 ** 6, 12, 24) == pytest.approx(base * (1e-3 + 0.5 * (1 - 1e-3)))
