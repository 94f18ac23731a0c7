"""Headless driver: render a reference `.pts` scene file on the GPU and export a PNG.

    python -m pbrpathtracer_amd.render scene.pts --spp 256 --out image.png [--seed S] [--device D]

The headless equivalent of the reference's Start button + Export (main.cpp:3563-3618, :760-771):
LoadScene -> SendObjectsToPathTracer -> RenderFrame() x spp -> PNG (flipped to top-down)."""
from __future__ import annotations

import argparse
import sys
import time

import numpy as np


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("scene")
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--out", default="render.png")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--pinhole", action="store_true", help="SetCameraAperture(0) after loading")
    a = ap.parse_args(argv)
    from .pathtracer import PathTracer, export_png
    pt = PathTracer(a.device)
    t0 = time.time()
    pt.LoadSceneFile(a.scene)
    if a.pinhole:
        pt.SetCameraAperture(0.0)
    w, h = pt.GetResolution()
    out = np.zeros((h, w, 3), np.uint8)
    pt.SetOutImage(out)
    pt.SetSeed(a.seed)
    t1 = time.time()
    pt.RenderFrames(a.spp)
    t2 = time.time()
    if pt.LastError():
        print("error:", pt.LastError(), file=sys.stderr)
        return 1
    export_png(a.out, out)
    print(f"{a.scene}: {pt.GetTriangleCount()} triangles, {w}x{h}, {a.spp} spp: load {t1 - t0:.2f} s, "
          f"render {t2 - t1:.3f} s ({w * h * a.spp / (t2 - t1) / 1e6:.0f} Msamples/s) -> {a.out}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
