"""The contracted builds of the trace kernels (ptk_set_option "contract" 1: -ffp-contract=fast; 2: ... with the hardware's
1-ulp reciprocal / square root) against the oracle at the north star's tolerance: per-channel RMSE of the MEAN image
<= 1e-3 (BASELINE.json north_star), all five BASELINE configs at their full size and sample count.  The exact kernels
(option 0, the default) remain the bit-exact product and the test oracle; these variants only have to stay inside the
tolerance, and the test reports how many accumulator words still agree exactly.

Reference arithmetic concerned: IntersectTriangle pathtracer.cpp:373-409, Trace :545-732 (every a * b + c in them may fuse)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RMSE_TOL = 1e-3      # north_star tolerance, per-channel float, on the mean image


@pytest.mark.parametrize("cfg,world", [("C1", 1), ("C2", 23), ("C3", 149), ("C4", 499), ("C5", 1999)])
def test_contracted_kernels_stay_inside_the_tolerance(tmp_path, oracle_mod, cfg, world):
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd import distributed as D
    from pbrpathtracer_amd.pathtracer import PathTracer, camera_from_scene
    pts, scene, spp = S.build_config(cfg, str(tmp_path))
    pt = PathTracer(0); pt.LoadSceneFile(pts); pt.SetSeed(21)
    cam = camera_from_scene(scene)
    if scene.pinhole:
        pt.SetCameraAperture(0.0)               # as bench.py does (the .pts carries F = 1e9)
        cam["aperture"] = 0.0
    W, H = pt.GetResolution(); Dp = pt.GetTraceDepth()
    o = oracle_mod.Oracle(pt.StagedScene())
    ocam = oracle_mod.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
    rank = world // 3
    ref, _ = o.render(ocam, W, H, Dp, 0, spp, 21, rank=rank, world=world, want_rgb8=False)
    mask = D.tile_owner_mask(W, H, rank, world)[::-1]           # accumulator rows are bottom-up
    assert mask.sum() >= 16 * 16 * 4 and ref[mask].any()
    ctx = pt.context()
    exact = None
    for level in (0, 1, 2):
        ctx.set_option("contract", level)
        pt.ResetImage(); pt.RenderFrames(spp)
        assert pt.LastError() == "" and pt.GetSamples() == spp
        got = pt.ReadAccumulation()
        assert np.isfinite(got).all()
        d = (got[mask] - ref[mask]) / spp
        rmse = np.sqrt((d.astype(np.float64) ** 2).mean(axis=0))            # per channel
        frac = float(np.mean(got[mask] == ref[mask]))
        print(f"{cfg} contract={level}: per-channel RMSE of the mean image {rmse}, exact words {frac:.4f}, pixels {int(mask.sum())}")
        assert (rmse <= RMSE_TOL).all()
        if level == 0:
            assert np.array_equal(got[mask], ref[mask])                     # the default stays bit-exact
            exact = got
        else:
            # the whole frame against the exact kernels too (every pixel, not only the oracle's tiles)
            dw = (got - exact) / spp
            rm = np.sqrt((dw.astype(np.float64) ** 2).mean(axis=(0, 1)))
            print(f"   whole frame vs exact kernels: RMSE {rm}, max |d| {np.abs(dw).max():.3e}")
            assert (rm <= RMSE_TOL).all()
    ctx.set_option("contract", 0)
    pt.close()
