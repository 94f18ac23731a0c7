import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR','127.0.0.1'); os.environ.setdefault('MASTER_PORT','29811')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda',0))
from conftest import load_golden, scene_from_golden
from pbrpathtracer_amd import ptk
z = load_golden("tier_s_cornell.npz")
c = ptk.Context(0)
c.upload_scene(scene_from_golden(z))
cam, proj = z["cam"], z["proj"]
c.set_camera(cam[0:3], cam[3:6], cam[6:9], float(proj[0]), float(proj[1]), float(z["focal_dist"]), float(z["aperture"]))
c.set_frame(80, 48, 4)
c.reset(); c.render(0, 5, 11)
own = c.read_accum()
accum = torch.zeros(48 * 80 * 3, dtype=torch.float32, device="cuda")
print("torch stream", torch.cuda.current_stream().cuda_stream)
c.set_stream(torch.cuda.current_stream().cuda_stream)
c.bind_accum(accum.data_ptr())
c.reset(); c.render(0, 5, 11)
c.synchronize(); torch.cuda.synchronize()
a = accum.cpu().numpy().reshape(48, 80, 3)
print("equal", np.array_equal(a, own), "sum", a.sum(), own.sum(), "nonzero", (a != 0).sum(), (own != 0).sum(), "maxdiff", np.abs(a - own).max())
b = c.read_accum()
print("read_accum equal own", np.array_equal(b, own), "equal a", np.array_equal(a, b))

from pbrpathtracer_amd.distributed import gather_accumulator
c.reset(); c.render(0, 5, 11)
out = gather_accumulator(accum, dst=0)
dist.barrier(); torch.cuda.synchronize()
o = out.cpu().numpy().reshape(48,80,3)
print("gathered equal", np.array_equal(o, own), "nonzero", (o!=0).sum(), "maxdiff", np.abs(o-own).max(), "accum equal", np.array_equal(accum.cpu().numpy().reshape(48,80,3), own))
dist.destroy_process_group()
