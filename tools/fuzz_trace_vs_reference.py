"""THIS CONTAINER ONLY (needs oracle/_ref built from /root/reference): PathTracer::Trace of the REAL reference against the oracle on
random scenes - the Cornell shell plus two to four spheres, every material field drawn at random (type, colours, emission, roughness,
reflectiveness, translucency, index of refraction, smoothing) and any of the six texture slots filled at random - with the
reference's own draws replayed from a tape (tier T's mechanism, oracle/gen_golden.py): the radiance must agree to 1e-5 relative and
exactly as many draws must be consumed.   python3 tools/fuzz_trace_vs_reference.py [--libm] [first_seed] [scenes] [paths_per_scene]
Round 3: 80 000 paths - with --libm every one bit-identical (NaN paths included); with the oracle's own sin/cos polynomial 17 part ways where a
last-bit change of a bounce direction lands in the neighbouring texel of a nearest-texel lookup."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pbrpathtracer_amd import scenes as S
from oracle.ref_binding import Ref
from oracle.ref_scene import arrays_from_ref
from oracle import oracle_binding as OB
OB.build()
if "--libm" in sys.argv:
    # the oracle with libm's sinf / cosf in place of its polynomial (oracle/pt_oracle.c ORC_LIBM_SINCOS): the one arithmetic difference
    # to the reference taken away, the replay must then agree BIT FOR BIT
    import subprocess
    sys.argv.remove("--libm")
    alt = os.path.join(tempfile.gettempdir(), "libptoracle_libm.so")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-fopenmp", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-DORC_LIBM_SINCOS", "-shared", "-o", alt,
                           os.path.join(ROOT, "oracle", "pt_oracle.c"), "-lm"])
    OB.LIB_PATH = alt
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
npaths = int(sys.argv[3]) if len(sys.argv) > 3 else 200
ref = Ref()
bad = 0; total = 0; worst = 0.0; nan_paths = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    tmp = tempfile.mkdtemp()
    sc = S.SceneDesc(trace_depth=int(rng.integers(1, 9)), width=64, height=64, focal_dist=3.5, camera_f=1.0e9)
    groups, mats = S.cornell_groups(uv=bool(seed % 4))                  # (every fourth scene: a shell without texture coordinates - NaN tangent frames, NaN radiance behind a normal map)
    texs = {"chk": S.tex_checker(32, 4), "nrm": S.tex_normal_waves(32, 2, 0.8), "noise": S.tex_noise(32, seed % 50, 0, 255, 4),
            "dots": S.tex_dots(32, 4, 0.35), "dim": (S.tex_dots(32, 4, 0.3) // 4).astype(np.uint8)}
    paths = {}
    for k, img in texs.items():
        paths[k] = os.path.join(tmp, k + ".ppm"); S.write_ppm(paths[k], img)
    for k in range(int(rng.integers(2, 5))):
        groups.append(S.uv_sphere(f"ball{k}", (float(rng.uniform(-0.6, 0.6)), float(rng.uniform(-0.7, 0.3)), float(rng.uniform(-0.4, 0.4))),
                                  float(rng.uniform(0.2, 0.4)), 10, 7, smooth=bool(rng.integers(0, 2))))
        mats.append(S.MaterialDesc())
    for i, m in enumerate(mats):
        if i < 6 and rng.uniform() < 0.4: continue                       # some of the shell's own materials stay (incl. the light)
        m.type = S.TRANSLUCENT if rng.uniform() < 0.35 else S.OPAQUE
        m.diffuse = tuple(float(x) for x in rng.uniform(0.05, 1.0, 3)); m.specular = tuple(float(x) for x in rng.uniform(0.05, 1.0, 3))
        if rng.uniform() < 0.15: m.emissive = tuple(float(x) for x in rng.uniform(0.0, 1.0, 3)); m.emissive_intensity = float(rng.uniform(0.5, 6.0))
        m.roughness = float(rng.choice([0.0, 1.0, rng.uniform(0.0, 1.0)])); m.reflectiveness = float(rng.choice([0.0, 1.0, rng.uniform(0.0, 1.0)]))
        m.translucency = float(rng.choice([0.0, 1.0, rng.uniform(0.0, 1.0)])); m.ior = float(rng.choice([1.0, 1.33, 1.5, rng.uniform(1.0, 2.4)]))
        for slot, pool in (("diffuse", ("chk", "noise")), ("normal", ("nrm",)), ("emissive", ("dim",)), ("roughness", ("noise", "chk")),
                           ("metallic", ("noise", "dots"))):          # (no opacity maps: the reference draws once per LEAF VISITED, in the order of its per-run random tree - DESIGN.md section 2, difference 4 - which no tape can replay)
            if rng.uniform() < 0.2: m.textures[slot] = paths[str(rng.choice(pool))] if rng.uniform() < 0.8 else os.path.join(tmp, "missing.ppm")      # (a file that is not there samples as 0)
    obj = os.path.join(tmp, "scene.obj")
    S.write_obj(obj, groups)
    sc.objects.append(S.ObjectDesc(obj, "scene", [S.ElementDesc(g.name, m) for g, m in zip(groups, mats)]))
    ref.load_scene(sc)
    arr = arrays_from_ref(ref, sc)
    o = OB.Oracle(arr)
    ro = np.array([0.0, 0.0, -3.5], np.float32)
    scene_bad = 0
    for i in range(npaths):
        rd = np.array([rng.uniform(-.46, .46) + 0.00137, rng.uniform(-.46, .46) + 0.00071, 1.0])
        rd = (rd / np.linalg.norm(rd)).astype(np.float32)
        ref.lib.ref_seed(900000 + seed * 1000 + i)
        tape = ref.peek_tape(8192)
        ref.lib.ref_mark()
        want = ref.trace(ro, rd)
        nd = ref.lib.ref_draws_since_mark(8192)
        if nd < 0: continue                                           # (a longer tape than we peeked)
        if not np.isfinite(want).all():
            # a NaN of the reference's own: the oracle must have it in the same channels after the same number of draws
            got, n = o.trace_tape(ro, rd, sc.trace_depth, tape[: nd + 4], mode=2)
            total += 1; nan_paths += 1
            if n != nd or not np.array_equal(np.isnan(got), np.isnan(want)) or not np.array_equal(got, want, equal_nan=True) and not np.allclose(got, want, rtol=1e-5, equal_nan=True):
                scene_bad += 1
                if scene_bad <= 3: print(f"MISMATCH (NaN path) seed {seed} path {i}: draws {n} vs {nd}, radiance {got} vs {want}", flush=True)
            continue
        got, n = o.trace_tape(ro, rd, sc.trace_depth, tape[: nd + 4], mode=2)
        total += 1
        err = float(np.abs(got - want).max() / max(1.0, float(np.abs(want).max())))
        worst = max(worst, err if np.isfinite(err) else 0.0)
        if n != nd or not (err <= 1e-5):
            scene_bad += 1
            if scene_bad <= 3: print(f"MISMATCH seed {seed} path {i} depth {sc.trace_depth}: draws {n} vs {nd}, radiance {got} vs {want}", flush=True)
    bad += scene_bad
    o.close()
    if scene_bad: print(f"  seed {seed}: {scene_bad} of {npaths} paths differ; materials: " + "; ".join(f"{g.name}: type {m.type} rough {m.roughness:.2f} refl {m.reflectiveness:.2f} transl {m.translucency:.2f} ior {m.ior:.2f} emis {m.emissive_intensity if any(m.emissive) else 0:.1f} tex {sorted(m.textures)}" for g, m in zip(groups, mats)), flush=True)
    if (seed - first) % 10 == 0: print(f"seed {seed}: {len(groups)} elements, depth {sc.trace_depth}, mismatches so far {bad} of {total} paths, worst rel. error {worst:.2e}", flush=True)
print("paths", total, "of which NaN in the reference", nan_paths, "mismatches", bad, "worst relative error %.2e" % worst)
sys.exit(1 if bad else 0)
