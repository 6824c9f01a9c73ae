"""MI355X-native path tracer: drop-in for the render loop of loewenheim/raytracing.

`raytracing_amd.api` mirrors the reference's public API; the work is done by hand-written
HIP kernels in csrc/librt_hip.so behind the C ABI of include/rt_abi.h.
"""
import os

import numpy as np

from .api import (AXIS_X, AXIS_Y, AXIS_Z, Camera, CameraOptions, ImageOptions, Material, Object, RenderOptions, Scene, Shape, Texture, pixels)

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


def load_ppm(path):
    """Binary PPM (P6, 8 bit) -> (h, w, 3) uint8 array."""
    with open(path, "rb") as f:
        data = f.read()
    parts, pos = [], 0
    while len(parts) < 4:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        parts.append(data[pos:end])
        pos = end
    assert parts[0] == b"P6" and parts[3] == b"255"
    w, h = int(parts[1]), int(parts[2])
    return np.frombuffer(data, dtype=np.uint8, count=3 * w * h, offset=pos + 1).reshape(h, w, 3).copy()


def earthmap():
    """The decoded earth map of the reference's final_scene example."""
    return load_ppm(os.path.join(ASSETS, "earthmap.ppm"))


__all__ = ["AXIS_X", "AXIS_Y", "AXIS_Z", "Camera", "CameraOptions", "ImageOptions", "Material", "Object", "RenderOptions", "Scene", "Shape", "Texture",
           "pixels", "earthmap", "load_ppm", "ASSETS"]
