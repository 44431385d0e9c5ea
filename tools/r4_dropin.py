"""Where a plugin-entry frame's time goes: head(mlvl_feats, img_metas) + get_bboxes, one frame at a time."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench
from transcar_amd import configs, synth, ops
dev = torch.device('cuda:0')
class A: shapes = 'res101'
bench._imports()
head, _sd = bench.build_head(dev)
shapes = configs.LEVEL_SHAPES['res101']
frame = synth.make_radar_frame(seed=2)
metas = synth.make_img_metas(1, synth.make_lidar2img(), radar=frame)
g = torch.Generator(device=dev); g.manual_seed(77)
feats = [torch.randn((6, 256, h, w), device=dev, generator=g).to(memory_format=torch.channels_last).unsqueeze(0) for (h, w) in shapes]
def sync(): torch.cuda.synchronize()
def once():
    outs = head(feats, metas); b = head.get_bboxes(outs, metas); sync(); return b
for _ in range(10): once()
def med(f, n=60):
    ts = []
    for _ in range(n):
        sync(); t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3
print('frame_once            %.3f ms' % med(once))
print('forward + sync        %.3f ms' % med(lambda: (head(feats, metas), sync())))
outs = head(feats, metas); sync()
print('get_bboxes + sync     %.3f ms' % med(lambda: (head.get_bboxes(outs, metas), sync())))
print('to_nhwc_levels (host) %.3f ms' % med(lambda: ops.to_nhwc_levels(feats)))
print('lidar2img staged      %.3f ms' % med(lambda: ops.lidar2img_tensor(metas, dev, staged=True)))
print('lidar2img plain+sync  %.3f ms' % med(lambda: (ops.lidar2img_tensor(metas, dev), sync())))
raws = [m['radar'] for m in metas]
def plan():
    tokens, pm, fill = head._radar_tokens_plan(raws, dev)
    if fill is not None: fill()
    sync()
print('radar plan+fill+sync  %.3f ms' % med(plan))
print('radar plan only       %.3f ms' % med(lambda: head._radar_tokens_plan(raws, dev)))
nhwc = ops.to_nhwc_levels(feats); l2i = ops.lidar2img_tensor(metas, dev); tokens, pm, fill = head._radar_tokens_plan(raws, dev)
if fill is not None: fill()
hw = metas[0]['img_shape'][0][:2]
print('forward_nhwc + sync   %.3f ms' % med(lambda: (head.forward_nhwc(nhwc, l2i, hw, tokens, pm), sync())))
def host_only():
    t0 = time.perf_counter(); head.forward_nhwc(nhwc, l2i, hw, tokens, pm); return time.perf_counter() - t0
sync(); hs = []
for _ in range(30):
    sync(); hs.append(host_only())
print('forward_nhwc host enqueue %.3f ms' % (np.median(hs) * 1e3))
def host_fwd():
    t0 = time.perf_counter(); head(feats, metas); return time.perf_counter() - t0
hs = []
for _ in range(30):
    sync(); hs.append(host_fwd())
print('head() host enqueue   %.3f ms' % (np.median(hs) * 1e3))
cls, box = outs['all_cls_scores'][-1], outs['all_bbox_preds'][-1]
pcr = head.bbox_coder.post_center_range
print('decode kept + sync    %.3f ms' % med(lambda: (ops.box_decode_kept(cls, box, pcr, 300), sync())))
print('decode + tolist       %.3f ms' % med(lambda: ops.box_decode_kept(cls, box, pcr, 300)[3].tolist()))
# device time of the forward alone
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ds = []
for _ in range(30):
    sync(); e0.record(); head.forward_nhwc(nhwc, l2i, hw, tokens, pm); e1.record(); sync(); ds.append(e0.elapsed_time(e1))
print('forward_nhwc device span %.3f ms' % np.median(ds))
