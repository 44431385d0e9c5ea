"""cProfile of the plugin entry's host side: head(mlvl_feats, img_metas) + get_bboxes, one frame."""
import cProfile, pstats, sys, torch
sys.path.insert(0, '.')
import bench; bench._imports()
from transcar_amd import configs, synth
dev = torch.device('cuda:0')
head, _ = bench.build_head(dev)
shapes = configs.LEVEL_SHAPES['res101']
metas = synth.make_img_metas(1, synth.make_lidar2img(), radar=synth.make_radar_frame(seed=2))
g = torch.Generator(device=dev); g.manual_seed(77)
feats = [torch.randn((6, 256, h, w), device=dev, generator=g).to(memory_format=torch.channels_last).unsqueeze(0) for (h, w) in shapes]
def once():
    o = head(feats, metas); b = head.get_bboxes(o, metas); torch.cuda.synchronize()
for _ in range(20): once()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): once()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
