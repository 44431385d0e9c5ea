import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
torch.cuda.set_device(0)
dist.init_process_group(backend='nccl', rank=0, world_size=1)
t = torch.ones(4, device='cuda'); dist.all_reduce(t); dist.barrier()
import bench
from transcar_amd.pipeline import FramePipeline
torch.set_grad_enabled(False)
head, _ = bench.build_head(torch.device('cuda:0'))
lanes = [bench.make_inputs(head, torch.device('cuda:0'), 'tiny', 1, seed=1 + i) for i in range(3)]
pipe = FramePipeline(head, lanes)
for _ in range(30): pipe.launch()
dist.barrier(); torch.cuda.synchronize(); dist.barrier()
t = torch.tensor([1.0], dtype=torch.float64, device='cuda'); dist.all_reduce(t, op=dist.ReduceOp.MAX)
print('nccl + graph capture ok', float(t.item()))
dist.destroy_process_group()
