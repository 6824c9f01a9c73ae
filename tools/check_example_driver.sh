#!/bin/bash
# Runs examples/rt_example on the GPU (as a process of its own) and compares its image with the library's through Python.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 120 ./examples/rt_example cornell_smoke --width 48 --height 40 --spp 8 --seed 7 --out gpurun_out/example_a.ppm
timeout -k 10 120 ./examples/rt_example cornell_smoke --width 48 --height 40 --spp 8 --seed 7 --devices 0,0 --out gpurun_out/example_b.png
timeout -k 10 120 ./examples/rt_example final_scene --spp 16 --out gpurun_out/example_final_scene.png
timeout -k 10 120 python3 - <<'PY'
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import raytracing_amd as rt
from test_host_and_core_cpu import _decode_png
sc, co = rt.Scene.example("cornell_smoke", 1)
want = sc.render(rt.Camera(co), rt.ImageOptions(40, 48, 8, 50), rt.RenderOptions(seed=7))
assert np.array_equal(rt.load_ppm("gpurun_out/example_a.ppm"), want)
assert _decode_png(open("gpurun_out/example_b.png", "rb").read()) == (48, 40, want.tobytes())
w, h, px = _decode_png(open("gpurun_out/example_final_scene.png", "rb").read())
assert (w, h) == (500, 500) and np.frombuffer(px, np.uint8).mean() > 5
print("example driver: PPM and PNG match rt_render byte for byte; final_scene 500x500 written")
PY
