"""Minimal workload for rocprofv3: one frame of a BASELINE scene at a few samples per pixel, every bounce as one k_traverse and
one k_shade launch on one stream (RT_RENDER_NO_TAIL | RT_RENDER_ONE_STREAM), so that counters summed per kernel name cover
every path segment exactly once.  A first frame of one sample per pixel lets the library regroup the tree by sampled rays as
it does in any longer run (tools/pmc_summarise.py leaves that frame's launches out; the json holds the second frame's segments).
argv: config, spp, output json (segments and samples of the run).
Used by tools/pmc_collect.sh; nothing here is a measurement by itself."""
import json
import sys
sys.path.insert(0, '.')
import raytracing_amd as rt
from raytracing_amd import _abi
SIZES = {"cornell": (600, 600), "perlin": (1920, 1080), "cornell_smoke": (800, 800), "final_scene": (800, 800)}
name, spp = sys.argv[1], int(sys.argv[2])
w, h = SIZES[name]
sc, co = rt.Scene.example(name, 1, rt.earthmap() if name == "final_scene" else None)
cam = rt.Camera(co)
flags = _abi.RT_RENDER_NO_TAIL | _abi.RT_RENDER_ONE_STREAM
sc.render(cam, rt.ImageOptions(h, w, 1, 50), rt.RenderOptions(seed=2, flags=flags))
sc.render(cam, rt.ImageOptions(h, w, spp, 50), rt.RenderOptions(seed=1, flags=flags))
p = sc.profile()
if len(sys.argv) > 3:
    with open(sys.argv[3], "w") as fh:
        json.dump({"config": name, "width": w, "height": h, "spp": spp, "samples": p["samples"], "segments": p["segments"]}, fh)
