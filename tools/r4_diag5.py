import os, sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from test_gpu_training import train_head, frame_inputs, dev, g8_name
from transcar_amd import ops, device_loss
from transcar_amd.trainer import FusionTrainer
gd = os.path.join('tests', 'golden')
tag = sys.argv[1] if len(sys.argv) > 1 else 'res101'
g8 = np.load(os.path.join(gd, g8_name(tag)))
h = train_head(gd)
feats, metas, gt, labels = frame_inputs(gd, tag)
nhwc = [ops.to_nhwc(f) for f in feats]
l2i = ops.lidar2img_tensor(metas, dev())
img_hw = metas[0]['img_shape'][0][:2]
tokens, pad_mult = h.radar_tokens(metas, dev())
print('Q', h.num_query, 'T', tokens.shape)
tr = FusionTrainer(h, dropout=0.0)
losses = tr.step_fused_nhwc(nhwc, l2i, img_hw, tokens, pad_mult, [gt], [labels], update=False)
fused = tr.bucket.grads.clone()
tr.bucket.zero_grad()
outs = h.train()(feats, metas)
ls = h.loss([gt], [labels], outs)
total = sum(v for k, v in ls.items() if 'loss' in k)
total.backward()
auto = tr.bucket.grads.clone()
for k in losses:
    print(k, float(losses[k]), float(ls[k]), float(g8['loss__' + k.replace('.', '_')]))
rows = []
for (n, p), off in zip(h.trainable_parameters(), tr.bucket.offsets):
    a = auto[off:off + p.numel()]; f = fused[off:off + p.numel()]
    key = 'grad__' + n.replace('.', '_')
    ref = None
    for cand in g8.files:
        if cand.endswith(n.replace('.', '_')) and 'grad' in cand:
            ref = torch.from_numpy(g8[cand]).to(a.device).flatten(); break
    ea = float((a - ref).abs().max() / ref.abs().max()) if ref is not None else -1
    ef = float((f - ref).abs().max() / ref.abs().max()) if ref is not None else -1
    rows.append((float((a - f).abs().max() / (a.abs().max() + 1e-30)), n, float(a.abs().max()), ea, ef))
rows.sort(reverse=True)
for r in rows[:14]:
    print('%.3e %-44s max %.3e  auto-vs-g8 %.2e fused-vs-g8 %.2e' % r)
names = [n for n, _ in h.trainable_parameters()]
def grad_of(buf, name):
    i = names.index(name); p = dict(h.trainable_parameters())[name]
    off = tr.bucket.offsets[i]
    return buf[off:off + p.numel()].view_as(p)
for nm in ['final_cls2.4.bias', 'final_cls2.4.weight', 'final_cls2.1.bias', 'final_cls.4.bias', 'final_cls3.4.bias']:
    dd = (grad_of(fused, nm) - grad_of(auto, nm)).abs()
    print(nm, 'max diff', float(dd.max()), 'n > 1%% of max: %d' % int((dd > 0.01 * dd.max()).sum()), 'argmax', int(dd.argmax()), 'second', float(dd.flatten().topk(2).values[1]))
D = (grad_of(fused, 'final_cls2.3.weight') - grad_of(auto, 'final_cls2.3.weight')).double()
sv = torch.linalg.svdvals(D)
print('final_cls2.3.weight diff singular values', [float(x) for x in sv[:4]])
