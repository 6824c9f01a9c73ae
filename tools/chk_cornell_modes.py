import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, raytracing_amd as rt
from raytracing_amd import _abi
sc, co = rt.Scene.example("cornell", 1, None)
cam = rt.Camera(co); io = rt.ImageOptions(96, 96, 256, 50)
_, base = sc.render(cam, io, rt.RenderOptions(seed=3), want_mean=True)
for name, fl in (("two_kernels", _abi.RT_RENDER_TWO_KERNELS), ("tail_only", _abi.RT_RENDER_TAIL_ONLY), ("no_tail", _abi.RT_RENDER_NO_TAIL), ("one_stream", _abi.RT_RENDER_ONE_STREAM)):
    _, m = sc.render(cam, io, rt.RenderOptions(seed=3, flags=fl), want_mean=True)
    print(name, "identical" if np.array_equal(m, base) else "DIFFERENT max %g" % np.nanmax(np.abs(m - base)))
