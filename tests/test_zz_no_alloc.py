"""rt_render_device_noalloc (include/rt_abi.h): the render out of the library's static device workspace must give the
bytes of rt_render_device, call after call, whatever the previous call left in the workspace; it must take no device memory;
and, captured into a HIP graph, every replay must give those bytes again."""
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
    # a scene whose first launch of a pass runs two bounces (the static workspace's schedule is fixed: nothing is read back)
    pl, cp = rt.Scene.example("perlin", 1)
    ref, got = _both(pl, rt.Camera(cp), rt.ImageOptions(120, 160, 40, 50), 6)
    assert np.array_equal(got, ref) and ref.any() and pl.profile()["two_bounce_passes"] == 4


def test_noalloc_render_takes_no_device_memory(earth):
    """hipMemGetInfo's free bytes are the same before and after calls of the entry point (torch's caching allocator holds on to
    what it has, so the library's own hipMalloc / hipFree would be the only thing to move them): frames of different sizes on a
    handle whose allocated workspace (grown by rt_render_device) is smaller than what they need."""
    sc, co = rt.Scene.example("final_scene", 1, earth)
    cam = rt.Camera(co)
    side = torch.cuda.Stream()
    buf = torch.zeros((200, 200, 3), dtype=torch.float64, device="cuda")
    sc.render_device(cam, rt.ImageOptions(16, 16, 2, 50), rt.RenderOptions(seed=1, stream=side.cuda_stream), buf.data_ptr())   # (a small allocated workspace)
    sc.render_device_noalloc(cam, rt.ImageOptions(16, 16, 2, 50), rt.RenderOptions(seed=1, stream=side.cuda_stream), buf.data_ptr())
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for (h, w, spp) in ((200, 200, 12), (64, 96, 3), (200, 160, 30)):
        sc.render_device_noalloc(cam, rt.ImageOptions(h, w, spp, 50), rt.RenderOptions(seed=2, stream=side.cuda_stream), buf.data_ptr())
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free0, (h, w, spp)
    assert buf.any()


def test_noalloc_render_under_graph_capture(earth):
    """The call joins the library's own stream to the caller's with an event on either side and creates nothing, so a capture of
    the caller's stream takes the whole render in: nothing runs during the capture, and every replay of the graph writes the
    bytes of rt_render_device."""
    sc, co = rt.Scene.example("final_scene", 1, earth)
    cam = rt.Camera(co)
    io = rt.ImageOptions(96, 128, 20, 50)
    side = torch.cuda.Stream()
    ref = torch.full((io.height, io.width, 3), float("nan"), dtype=torch.float64, device="cuda")
    side.wait_stream(torch.cuda.current_stream())
    sc.render_device(cam, io, rt.RenderOptions(seed=9, stream=side.cuda_stream), ref.data_ptr())
    side.synchronize()
    ref = ref.cpu().numpy()
    assert np.isfinite(ref).all() and ref.any()
    buf = torch.full((io.height, io.width, 3), float("nan"), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side, capture_error_mode="relaxed"):
        sc.render_device_noalloc(cam, io, rt.RenderOptions(seed=9, stream=side.cuda_stream), buf.data_ptr())
    torch.cuda.synchronize()
    assert torch.isnan(buf).all()          # captured, not run
    for _ in range(2):
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(buf.cpu().numpy(), ref)
        buf.fill_(float("nan"))
        torch.cuda.synchronize()


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
