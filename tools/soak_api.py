"""GPU box, one-off: random sequences of host-API calls on the drop-in PathTracer (camera, projection, lens, resolution, trace
depth, seed, material edits after BuildBVH, another scene file into the same tracer, tile splits, Exit() with nothing in flight,
persistent waves forced on and off, textures set on any slot after the build, BuildBVH again, further objects loaded into the scene, hand-off buffers of every kind coming and going, sample batching, a second tracer on the same GPU) - after every stage the
accumulator must be the oracle's for the state the calls left behind.  python tools/soak_api.py [first_seed] [count]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from pbrpathtracer_amd import scenes as S
from pbrpathtracer_amd.pathtracer import PathTracer, camera_from_scene
from pbrpathtracer_amd import distributed as Dm
from oracle import oracle_binding as OB
OB.build()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
tmp = tempfile.mkdtemp()
cfgs = [("C1", dict(width=96, height=64)), ("C4", dict(width=96, height=64, grid=14)), ("C3", dict(width=80, height=56))]
built = {}
for name, kw in cfgs:
    d = os.path.join(tmp, name); os.makedirs(d)
    built[name] = S.build_config(name, d, **kw)
tex_files = []
for k, img in enumerate((S.tex_checker(16, 4), S.tex_noise(32, 5, 0, 255, 4), S.tex_dots(8, 2, 0.4))):
    tex_files.append(os.path.join(tmp, f"t{k}.ppm")); S.write_ppm(tex_files[-1], img)
tex_files.append(os.path.join(tmp, "missing.ppm"))
bad = 0; t0 = time.time(); stages = 0
for k in range(count):
    seed = first + k
    rng = np.random.default_rng(seed)
    name = cfgs[k % len(cfgs)][0]
    pts, scene, _ = built[name]
    pt = PathTracer(0); pt.LoadSceneFile(pts)
    other = PathTracer(0); other.LoadSceneFile(built["C1"][0])
    cam = camera_from_scene(scene)
    st = dict(W=pt.GetResolution()[0], H=pt.GetResolution()[1], D=pt.GetTraceDepth(), seed=0, tile=(0, 1))
    out_img = None
    for stage in range(int(rng.integers(3, 8))):
        op = int(rng.integers(0, 15))
        # the hand-off buffer comes and goes, pageable (page-locked in place) or from AllocOutImage
        if rng.uniform() < 0.5:
            kind = int(rng.integers(0, 4))                # (3: a device buffer - a torch tensor - through SetOutDeviceImage)
            out_img = None if kind == 0 else "pending-%d" % kind
        if op == 0:
            cam["pos"] = (np.asarray(cam["pos"], np.float32) + rng.uniform(-0.3, 0.3, 3).astype(np.float32)); pt.SetCamera(cam["pos"], cam["dir"], cam["up"])
        elif op == 1:
            cam["aperture"] = float(rng.choice([0.0, 0.03])); pt.SetCameraAperture(cam["aperture"])
        elif op == 2:
            st["W"], st["H"] = int(rng.integers(1, 130)), int(rng.integers(1, 90)); pt.SetResolution((st["W"], st["H"]))
        elif op == 3:
            st["D"] = int(rng.integers(1, 9)); pt.SetTraceDepth(st["D"])
        elif op == 4:
            st["seed"] = int(rng.integers(0, 1 << 30)); pt.SetSeed(st["seed"])
        elif op == 5:
            cam["fovy"] = float(rng.uniform(30, 80)); pt.SetProjection(cam["focal"], cam["fovy"])
        elif op == 6:
            m = np.array([int(rng.integers(0, 2)), *rng.uniform(0.1, 0.9, 3), *rng.uniform(0.2, 1, 3), *(rng.uniform(0, 1, 3) * (rng.uniform() < 0.3)),
                          float(rng.uniform(1, 5)), float(rng.choice([0, 0.4, 1])), float(rng.choice([0, 0.5, 1])), float(rng.choice([0, 1])), 1.5], np.float32)
            objs = pt.GetLoadedObjects()
            ob = int(rng.integers(0, len(objs)))
            pt.SetMaterial(ob, int(rng.integers(0, max(1, objs[ob]))), m)
        elif op == 7:
            cam["focal_dist"] = float(cam["focal_dist"] * rng.uniform(0.8, 1.2)); pt.SetCameraFocalDist(cam["focal_dist"])
        elif op == 8:
            # another scene file into the same tracer (ClearScene + the loader's whole call sequence): resolution, depth and camera are the file's
            name = cfgs[int(rng.integers(0, len(cfgs)))][0]
            pts, scene, _ = built[name]
            pt.ClearScene(); pt.LoadSceneFile(pts)
            cam = camera_from_scene(scene)
            st.update(W=pt.GetResolution()[0], H=pt.GetResolution()[1], D=pt.GetTraceDepth())
        elif op == 9:
            w = int(rng.integers(1, 6)); st["tile"] = (int(rng.integers(0, w)), w); pt.SetTile(*st["tile"])
        elif op == 10:
            pt.Exit()                                        # nothing in flight: must not disturb what follows
        elif op == 12:
            # a texture set (or re-set: the reference reloads the Image in place, pathtracer.cpp:147-241) after the scene has been
            # built and rendered: any slot, files of three sizes and one that does not exist (samples as 0)
            objs = pt.GetLoadedObjects(); ob = int(rng.integers(0, len(objs)))
            if objs[ob]:
                el = int(rng.integers(0, objs[ob])); slot = int(rng.integers(0, 6))
                tf = str(rng.choice(tex_files)); pt._set_tex(slot, ob, el, tf)
                if os.environ.get("SOAK_VERBOSE"): print(f"    texture: object {ob} element {el} slot {slot} <- {os.path.basename(tf)}", flush=True)
        elif op == 13:
            pt.BuildBVH()                                    # again: the light list follows the materials as they are now (pathtracer.cpp:267-273)
        elif op == 14:
            # one more object into the loaded scene (LoadObject + SetMaterial + BuildBVH, previewer.cpp:770-817): a quad or a small fan, sometimes a light
            k = len(pt.GetLoadedObjects())
            extra = os.path.join(tmp, f"extra_{seed}_{stage}.obj")
            cx, cy, cz = rng.uniform(-0.5, 0.5, 3)
            open(extra, "w").write("".join(f"v {cx + dx:.4f} {cy + dy:.4f} {cz + 0.1 * dx * dy:.4f}\n" for dx, dy in ((-.2, -.2), (.2, -.2), (.25, .2), (-.2, .25), (0, .4)))
                                   + "vn 0 0 -1\n" + ("f 1//1 2//1 3//1 4//1\n" if rng.uniform() < 0.5 else "f 1//1 2//1 3//1 4//1 5//1\n"))
            pt.LoadObject(extra, np.eye(4, dtype=np.float32))
            mm = np.array([0, *rng.uniform(0.2, 0.9, 3), 0.5, 0.5, 0.5, *((1.0, 0.9, 0.8) if rng.uniform() < 0.4 else (0, 0, 0)), float(rng.uniform(1, 4)), 1.0, 0.0, 0.0, 1.5], np.float32)
            pt.SetMaterial(k, 0, mm)
            pt.BuildBVH()
        else:
            c = pt.context(); forced = bool(rng.integers(0, 2))
            c.set_option("persistent", 1 if forced else -1)
        if isinstance(out_img, str):
            Wc, Hc = st["W"], st["H"]
            if out_img.endswith("3"): out_img = torch.full((Hc, Wc, 3), 9, dtype=torch.uint8, device="cuda:0"); torch.cuda.synchronize()
            else: out_img = pt.AllocOutImage() if out_img.endswith("2") and pt.GetResolution() == (Wc, Hc) else np.full((Hc, Wc, 3), 9, np.uint8)
        if out_img is not None and tuple(out_img.shape[:2]) != (st["H"], st["W"]): out_img = np.full((st["H"], st["W"], 3), 9, np.uint8)
        if os.environ.get("SOAK_VERBOSE"): print(f"  seed {seed} stage {stage} op {op} st {st} out {None if out_img is None else (type(out_img).__name__, tuple(out_img.shape))}", flush=True)
        if torch.is_tensor(out_img): pt.SetOutDeviceImage(out_img.data_ptr())
        else: pt.SetOutDeviceImage(None); pt.SetOutImage(out_img)
        pt.ResetImage()
        total = int(rng.integers(1, 9)); done = 0
        while done < total:
            b = int(rng.integers(1, total - done + 1)); pt.RenderFrames(b); done += b
            if rng.uniform() < 0.3: other.RenderFrames(1)            # another tracer on the same GPU in between
        err = pt.LastError()
        got = pt.ReadAccumulation()
        o = OB.Oracle(pt.StagedScene())
        ocam = OB.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
        ref, ref8 = o.render(ocam, st["W"], st["H"], st["D"], 0, total, st["seed"], rank=st["tile"][0], world=st["tile"][1])
        o.close()
        stages += 1
        if st["tile"][1] > 1:                                # a rank of a split: only its own tiles are its business
            own = Dm.tile_owner_mask(st["W"], st["H"], *st["tile"])[::-1]
            got = np.where(own[..., None], got, 0); ref = np.where(own[..., None], ref, 0)
            if out_img is not None:
                h8 = out_img.cpu().numpy() if torch.is_tensor(out_img) else np.asarray(out_img)
                if not np.array_equal(h8[own], ref8[own]):
                    bad += 1; print(f"HAND-OFF MISMATCH (split) seed {seed} {name} stage {stage} op {op} state {st}", flush=True)
        elif out_img is not None and not np.array_equal(out_img.cpu().numpy() if torch.is_tensor(out_img) else np.asarray(out_img), ref8):
            bad += 1
            print(f"HAND-OFF MISMATCH seed {seed} {name} stage {stage} op {op} state {st} buffer {type(out_img).__name__}", flush=True)
        if err or got.shape != ref.shape or not np.array_equal(got, ref, equal_nan=True):       # (NaN where the reference has NaN: a normal map on a mesh without uvs)
            bad += 1
            print(f"MISMATCH seed {seed} {name} stage {stage} op {op} state {st} aperture {cam['aperture']} err '{err}'", flush=True)
    pt.SetOutDeviceImage(None); pt.SetOutImage(None); pt.close(); other.close(); out_img = None
    if k % 20 == 0: print(f"seed {seed} done [{time.time() - t0:.0f} s]", flush=True)
print("stages", stages, "mismatches:", bad)
sys.exit(1 if bad else 0)
