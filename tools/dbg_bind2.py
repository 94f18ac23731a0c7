import sys, os
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_gpu_exchange as T
if len(sys.argv) > 1 and sys.argv[1] == "first":
    T.test_native_rccl_gather_single_rank(); print("first ok")
import torch, torch.distributed as dist
from pbrpathtracer_amd.distributed import gather_accumulator
os.environ.setdefault('MASTER_ADDR','127.0.0.1'); os.environ.setdefault('MASTER_PORT','29812')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda',0))
c = T._ctx()
c.reset(); c.render(0, 5, 11)
own = c.read_accum()
accum = torch.zeros(48 * 80 * 3, dtype=torch.float32, device="cuda")
print("stream", torch.cuda.current_stream().cuda_stream)
c.set_stream(torch.cuda.current_stream().cuda_stream)
c.bind_accum(accum.data_ptr())
c.reset(); c.render(0, 5, 11)
out = gather_accumulator(accum, dst=0)
dist.barrier(); torch.cuda.synchronize()
o = out.cpu().numpy().reshape(48, 80, 3); a = accum.cpu().numpy().reshape(48, 80, 3)
print("out==own", np.array_equal(o, own), "accum==own", np.array_equal(a, own), "nonzero", (o != 0).sum(), (a != 0).sum(), (own != 0).sum(), "err", c.L.ptk_last_error(c.h))
dist.destroy_process_group()
