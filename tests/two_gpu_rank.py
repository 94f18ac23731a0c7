"""One rank of tests/test_gpu_exchange.py::test_native_gather_between_two_gpus: `python two_gpu_rank.py RANK WORLD DIR`.
Renders this rank's tiles on GPU `RANK`, joins the library's own RCCL communicator (the 128-byte id travels through a file
in DIR), gathers twice - once with the next render already queued behind the exchange - and, on the root, saves the gathered
image."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    rank, world, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    from conftest import load_golden, scene_from_golden
    from pbrpathtracer_amd import ptk
    z = load_golden("tier_s_cornell.npz")
    c = ptk.Context(rank)
    c.upload_scene(scene_from_golden(z))
    cam, proj = z["cam"], z["proj"]
    c.set_camera(cam[0:3], cam[3:6], cam[6:9], float(proj[0]), float(proj[1]), float(z["focal_dist"]), float(z["aperture"]))
    c.set_frame(200, 136, 4)
    c.set_tile(rank, world)
    idfile = os.path.join(out, "rccl_id.bin")
    if rank == 0:
        uid = ptk.comm_unique_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 120:
                raise RuntimeError("no RCCL id from rank 0")
            time.sleep(0.05)
        uid = open(idfile, "rb").read()
    c.comm_init(uid, rank, world)
    c.reset()
    c.render(0, 6, 17)
    c.gather_accum(0)
    c.render(6, 4, 17)              # queued at once: its trace kernel overlaps the transfer
    c.gather_wait()
    c.gather_accum(0)
    c.gather_wait()
    if rank == 0:
        np.save(os.path.join(out, "gathered.npy"), c.read_gathered())
    c.comm_destroy()
    c.close()


if __name__ == "__main__":
    main()
