"""The pooled BVH kernel (trace_pool_kernel: more paths than lanes, shade / trace phases over a wave-private pool, walks
suspended across the shade phase) against the oracle and against trace_kernel<BVH>, bit for bit: random scenes through
every branch of PathTracer::Trace (pathtracer.cpp:551-727) with thin-lens and pinhole cameras, opacity textures (the walk's
stochastic test needs the path's key and ray number from the job record), with every pool size, with thresholds that make
lanes fetch at once or wait long, and with a switch threshold of zero, which suspends and resumes walks at every opportunity.
Small frames are forced onto persistent waves (option "persistent" 1), which the pooled kernel requires."""
import numpy as np
import pytest

from test_gpu_random_scenes import random_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from pbrpathtracer_amd import ptk
    c = ptk.Context(0)
    yield c
    for k, v in (("persistent", -1), ("pool", 0), ("fetch_threshold", 3), ("switch_threshold", 16), ("flat", 1), ("chunk", 0)):
        c.set_option(k, v)
    c.close()


@pytest.mark.parametrize("seed,n_tris,tex", [(21, 40, False), (22, 300, True), (23, 6000, True), (24, 6000, False), (25, 1500, True), (26, 12, True)])
def test_pooled_kernel_matches_oracle_and_megakernel(ctx, oracle_mod, seed, n_tris, tex):
    arrays, cam = random_scene(seed, n_tris, tex)
    W, H, D, spp = 72, 56, 7, 12
    o = oracle_mod.Oracle(arrays)
    ocam = oracle_mod.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
    ref, ref8 = o.render(ocam, W, H, D, 0, spp, seed)
    o.close()
    assert (ref != 0).any(axis=2).mean() > 0.5
    ctx.set_option("flat", 0)                   # (tiny scenes too: through a tree)
    ctx.set_option("persistent", 1)
    ctx.upload_scene(arrays); ctx.set_camera(**cam); ctx.set_frame(W, H, D); ctx.set_tile(0, 1)
    settings = [dict(pool=0), dict(pool=256), dict(pool=64), dict(pool=128, fetch_threshold=0), dict(pool=192, fetch_threshold=64),
                dict(pool=256, switch_threshold=0), dict(pool=64, switch_threshold=0, fetch_threshold=0), dict(pool=256, switch_threshold=4096),
                dict(pool=128, chunk=1), dict(pool=256, chunk=3)]
    for opts in settings:
        for k, v in dict(dict(fetch_threshold=3, switch_threshold=16, chunk=0), **opts).items():
            ctx.set_option(k, v)
        ctx.reset(); ctx.render(0, spp, seed)
        got, got8 = ctx.read_accum(), ctx.resolve_rgb8()
        assert np.array_equal(ref, got), (seed, opts, float(np.abs(ref - got).max()), float(np.mean(ref != got)))
        assert np.array_equal(ref8, got8), (seed, opts)
    # the counters of the pooled kernel count the same rays, node visits and hits as the megakernel's
    ctx.set_option("pool", 0); a = ctx.collect_stats(0, spp, seed)
    ctx.set_option("pool", 256); b = ctx.collect_stats(0, spp, seed)
    for k in ("rays", "shadow_rays", "hits_shaded", "tex_fetches", "paths_started"):
        assert a[k] == b[k], (k, a[k], b[k])
    print(f"seed {seed}: rays/sample {b['rays'] / b['samples']:.2f}; lanes per shade execution {a['shade_lanes'] / max(1, a['shade_wave_execs']):.1f} -> "
          f"{b['shade_lanes'] / max(1, b['shade_wave_execs']):.1f}, walking lanes per iteration {a['walk_lane_iters'] / max(1, a['walk_wave_iters']):.1f} -> "
          f"{b['walk_lane_iters'] / max(1, b['walk_wave_iters']):.1f}")


def test_pooled_kernel_under_a_tile_split_and_sample_batches(ctx, oracle_mod):
    """Ranks of a 3-way tile split, each rendered in two batches of samples, sum to the oracle's frame (multi-generation
    launches: waves retire at their quota with paths of later items never started)."""
    arrays, cam = random_scene(31, 2500, True)
    W, H, D, spp = 96, 64, 6, 10
    o = oracle_mod.Oracle(arrays)
    ocam = oracle_mod.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
    ref, _ = o.render(ocam, W, H, D, 0, spp, 5)
    o.close()
    ctx.set_option("flat", 0); ctx.set_option("persistent", 1); ctx.set_option("pool", 256)
    ctx.upload_scene(arrays); ctx.set_camera(**cam); ctx.set_frame(W, H, D)
    total = np.zeros_like(ref)
    for gens in (1, 2):
        ctx.set_option("generations", gens)
        total[:] = 0
        for r in range(3):
            ctx.set_tile(r, 3); ctx.reset()
            ctx.render(0, 4, 5); ctx.render(4, spp - 4, 5)
            total += ctx.read_accum()
        assert np.array_equal(total, ref), gens
    ctx.set_option("generations", 0)
    ctx.set_tile(0, 1)
