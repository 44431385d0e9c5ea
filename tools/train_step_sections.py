import sys, time, torch
sys.path.insert(0, '.')
import bench
import transcar_amd as T
from transcar_amd import configs, synth
from transcar_amd.trainer import FusionTrainer
dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
head, sd = bench.build_head(dev)
inp = bench.make_inputs(head, dev, 'res101', 1, seed=1)
cfg = configs.head_cfg(); cfg['train_cfg'] = configs.train_cfg_pts
th = T.build_head(cfg); th.load_state_dict(head.state_dict()); th = th.to(dev)
boxes, labels = synth.make_gt(seed=7, n=24)
gt = torch.from_numpy(boxes).clone(); gt[:, 2] += gt[:, 5] * 0.5
gts, lbs = [gt.to(dev)], [torch.from_numpy(labels).to(dev)]
torch.set_grad_enabled(True)
tr = FusionTrainer(th)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(8):
    t0 = sync()
    th.train()
    outs = th.forward_train_nhwc(inp['nhwc'], inp['l2i'], inp['hw'], inp['tokens'], inp['pad_mult'])
    t1 = sync()
    losses = th.loss(gts, lbs, outs)
    t2 = sync()
    total = sum(v for k, v in losses.items() if 'loss' in k)
    tr.bucket.zero_grad()
    total.backward()
    t3 = sync()
    tr.bucket.all_reduce()
    import ctypes as C
    from transcar_amd import _lib as L
    tr.iter += 1; tr.sq.zero_()
    L.lib().tc_sq_norm(tr.bucket.grads.data_ptr(), tr.bucket.numel, tr.sq.data_ptr(), tr._stream())
    L.lib().tc_adamw_step(tr.bucket.params.data_ptr(), tr.bucket.grads.data_ptr(), tr.m.data_ptr(), tr.v.data_ptr(), tr.bucket.numel, 1e-5, 0.9, 0.999, 1e-8, 0.01, tr.iter, 1.0, 35.0, tr.sq.data_ptr(), tr._stream())
    th.repack_weights()
    t4 = sync()
    if it >= 3: print('fwd %.2f  loss %.2f  bwd %.2f  opt+repack %.2f ms' % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3))
