#!/usr/bin/env python3
"""Host-side profile of the plugin-swap call head(mlvl_feats, img_metas) + get_bboxes (cProfile)."""
import cProfile, io, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from transcar_amd import configs, synth
dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
head, sd = bench.build_head(dev)
shapes = configs.LEVEL_SHAPES['res101']
frame = synth.make_radar_frame(seed=2)
metas = synth.make_img_metas(1, synth.make_lidar2img(), radar=frame)
feats = [torch.randn((6, 256, h, w), device=dev).to(memory_format=torch.channels_last).unsqueeze(0) for (h, w) in shapes]
def once():
    outs = head(feats, metas)
    b = head.get_bboxes(outs, metas)
    torch.cuda.synchronize()
    return b
for _ in range(20): once()
t0 = time.perf_counter()
for _ in range(200): once()
print('ms per frame: %.3f' % ((time.perf_counter() - t0) / 200 * 1e3))
# host time until everything is enqueued (no sync)
t0 = time.perf_counter()
for _ in range(200):
    outs = head(feats, metas)
torch.cuda.synchronize()
print('forward only, no per-frame sync: %.3f ms' % ((time.perf_counter() - t0) / 200 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(200): once()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(30); print(s.getvalue()[:6000])
