"""Throughput of the three configurations whose trees are scanned (cornell, perlin, cornell_smoke) at reduced sample counts,
with a checksum of the f64 frame (the same bytes are expected from every build of the library)."""
import sys, time, zlib
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import raytracing_amd as rt
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
out = []
for name, w, h, spp in [('cornell', 600, 600, 1000), ('perlin', 1920, 1080, 1000), ('cornell_smoke', 800, 800, 4000)]:
    s = max(1, int(spp * scale))
    sc, co = rt.Scene.example(name, 1); cam = rt.Camera(co)
    sc.render(cam, rt.ImageOptions(h, w, min(s, 64), 50), rt.RenderOptions(seed=1))
    best = 1e9
    for _ in range(2):
        t0 = time.time(); rgb, mean = sc.render(cam, rt.ImageOptions(h, w, s, 50), rt.RenderOptions(seed=1), want_mean=True); best = min(best, time.time() - t0)
    out.append('%s %d spp %.1f Msamples/s crc %08x' % (name, s, w * h * s / best / 1e6, zlib.crc32(mean.tobytes())))
print(' | '.join(out), flush=True)
