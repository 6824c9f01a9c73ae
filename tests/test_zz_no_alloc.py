"""rt_render_device_noalloc (include/rt_abi.h): the render out of the library's static device workspace must give the
bytes of rt_render_device, call after call, whatever the previous call left in the workspace."""
import numpy as np
import pytest
import torch

import raytracing_amd as rt
from raytracing_amd import _abi

pytestmark = pytest.mark.gpu


def _both(sc, cam, io, seed, **kw):
    side = torch.cuda.Stream()
    outs = []
    for fn in (sc.render_device, sc.render_device_noalloc):
        buf = torch.full((io.height, io.width, 3), float("nan"), dtype=torch.float64, device="cuda")
        side.wait_stream(torch.cuda.current_stream())
        fn(cam, io, rt.RenderOptions(seed=seed, stream=side.cuda_stream, **kw), buf.data_ptr())
        side.synchronize()
        outs.append(buf.cpu().numpy())
    return outs


def test_noalloc_render_matches_render_device(earth):
    sc, co = rt.Scene.example("final_scene", 1, earth)
    cam = rt.Camera(co)
    # first call: 160 x 120 pixels at 40 spp = several passes of the static workspace's 262 144 paths, each long enough
    # for the wavefront kernels before the tail kernel; second call, right behind it: a smaller frame, one short pass
    for (h, w, spp, seed) in ((120, 160, 40, 3), (48, 40, 5, 4)):
        ref, got = _both(sc, cam, rt.ImageOptions(h, w, spp, 50), seed)
        assert np.isfinite(ref).all() and ref.any()
        assert np.array_equal(got, ref), (h, w, spp)
    # a rank's share of a split frame, and another scene handle on the same GPU (the array is per GPU, not per scene)
    sm, cs = rt.Scene.example("cornell_smoke", 1)
    ref, got = _both(sm, rt.Camera(cs), rt.ImageOptions(64, 64, 9, 50), 5, rank=1, world_size=3, tile_w=8, tile_h=4)
    assert np.array_equal(got, ref) and ref.any()


def test_noalloc_render_rejects_what_does_not_fit(earth):
    sc, co = rt.Scene.example("cornell", 1)
    cam = rt.Camera(co)
    buf = torch.zeros((600, 600, 3), dtype=torch.float64, device="cuda")
    for io, kw in ((rt.ImageOptions(600, 600, 1, 50), {}),                      # 360 000 pixels on one rank
                   (rt.ImageOptions(64, 64, 1, 255), {}),                       # depth beyond the counter slots
                   (rt.ImageOptions(64, 64, 1, 50), {"flags": _abi.RT_RENDER_PROFILE})):
        with pytest.raises(_abi.RtError) as e:
            sc.render_device_noalloc(cam, io, rt.RenderOptions(seed=1, **kw), buf.data_ptr())
        assert e.value.status in (_abi.RT_ERR_UNSUPPORTED, _abi.RT_ERR_INVALID_ARGUMENT)
    # the same frame fits once it is split over two ranks
    sc.render_device_noalloc(cam, rt.ImageOptions(600, 600, 1, 50), rt.RenderOptions(seed=1, rank=0, world_size=2), buf.data_ptr())
    torch.cuda.synchronize()
    assert buf.any()
