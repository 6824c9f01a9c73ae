"""The N > 1 path on the GPU as far as one GPU can take it: the RCCL process group with a single rank (communicator set-up, the
f64 sum-reduce of a device frame on RCCL's stream behind the library's own streams, twice on one handle), the schedule the
library learns per pass size, tiles of odd shapes, and concurrent calls on one scene handle."""
import datetime
import socket
import threading

import numpy as np
import pytest
import torch
import torch.distributed as dist

import raytracing_amd as rt
from raytracing_amd import _abi
from raytracing_amd.distributed import owned_mask, render_distributed

pytestmark = pytest.mark.gpu


def _example(name, earth, **kw):
    return rt.Scene.example(name, 1, earth if name == "final_scene" else None, **kw)


def test_rccl_group_of_one_renders_the_single_process_frame_twice(earth):
    """render_distributed under backend "nccl" (RCCL on ROCm), world size 1: the reduce is the identity, so both frames must be
    the bytes of rt_render; the second call on the same handle reuses the learnt schedule and the side stream."""
    assert not dist.is_initialized()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0),
                            timeout=datetime.timedelta(seconds=120))
    try:
        assert dist.get_backend() == "nccl"
        for name, w, h, spp in (("final_scene", 96, 64, 24), ("cornell_smoke", 64, 48, 16)):
            sc, co = _example(name, earth)
            cam = rt.Camera(co)
            io = rt.ImageOptions(h, w, spp, 50)
            want_rgb, want_mean = sc.render(cam, io, rt.RenderOptions(seed=5), want_mean=True)
            for _ in range(2):
                rgb, mean = render_distributed(sc, cam, io, seed=5)
                torch.cuda.current_stream().synchronize()
                assert np.array_equal(mean.cpu().numpy(), want_mean), name
                assert np.array_equal(rgb.cpu().numpy(), want_rgb), name
        t = torch.tensor([3.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)      # (what bench.py does with the step time)
        assert float(t.item()) == 3.5
        gathered = [None]
        dist.all_gather_object(gathered, {"rank": 0})  # (and with the per-rank device records)
        assert gathered == [{"rank": 0}]
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_schedule_is_learnt_per_pass_size(earth):
    """A frame of one sample per pixel hands its short queue to the tail kernel after a bounce; the same frame in large passes
    must not inherit that (nor the reverse): the learnt hand-over belongs to the pass size it was seen at."""
    sc, co = _example("final_scene", earth)
    cam = rt.Camera(co)
    w, h = 256, 256
    sc.render(cam, rt.ImageOptions(h, w, 1, 50), rt.RenderOptions(seed=1))
    small = sc.profile()
    assert small["pass_paths"] == w * h and 0 <= small["tail_from"] <= 2, small
    sc.render(cam, rt.ImageOptions(h, w, 64, 50), rt.RenderOptions(seed=1))
    large = sc.profile()
    assert large["pass_paths"] == 64 * w * h
    assert large["tail_from"] > small["tail_from"] + 2, (small, large)     # 4 M paths stay above the hand-over length for several bounces
    # both are remembered: each pass size gets its own schedule back, and the frames do not depend on any of it
    a = sc.render(cam, rt.ImageOptions(h, w, 1, 50), rt.RenderOptions(seed=1))
    assert sc.profile()["tail_from"] == small["tail_from"]
    b = sc.render(cam, rt.ImageOptions(h, w, 1, 50), rt.RenderOptions(seed=1, flags=_abi.RT_RENDER_RECALIBRATE))
    assert np.array_equal(a, b)
    sc.render(cam, rt.ImageOptions(h, w, 64, 50), rt.RenderOptions(seed=1))
    again = sc.profile()
    assert again["tail_from"] == large["tail_from"] and again["passes"] == 1, again


def test_tiles_of_any_shape_cover_the_rank_s_pixels(earth):
    """The rank's pixel list is laid out tile by tile on the device (k_owned_list); whatever the tiles' shape, the ranks' frames
    are disjoint, follow owned_mask and add up to the single-rank frame."""
    sc, co = _example("cornell", earth)
    cam = rt.Camera(co)
    w, h = 50, 37
    io = rt.ImageOptions(h, w, 6, 50)
    _, base = sc.render(cam, io, rt.RenderOptions(seed=2), want_mean=True)
    assert base.any()
    for (tw, th, world) in ((16, 16, 3), (1, 1, 4), (50, 37, 2), (64, 64, 2), (7, 37, 3), (50, 5, 2), (3, 2, 8)):
        parts = [sc.render(cam, io, rt.RenderOptions(seed=2, rank=r, world_size=world, tile_w=tw, tile_h=th), want_mean=True)[1] for r in range(world)]
        assert np.array_equal(np.sum(parts, axis=0), base), (tw, th, world)
        for r in range(world):
            mask = owned_mask(w, h, r, world, min(tw, w), min(th, h))
            assert not parts[r][~mask].any(), (tw, th, world, r)


def test_calls_from_several_threads_on_one_handle(earth):
    """Every entry point holds the handle's lock: renders of different frames issued from four threads on one scene handle
    (workspace regrown, schedule relearnt in between) each give the bytes of the same render made alone."""
    sc, co = _example("cornell_smoke", earth)
    cam = rt.Camera(co)
    jobs = [(rt.ImageOptions(40, 56, 8, 50), 11), (rt.ImageOptions(96, 96, 24, 50), 12), (rt.ImageOptions(24, 24, 3, 50), 13), (rt.ImageOptions(64, 80, 12, 50), 14)]
    want = [sc.render(cam, io, rt.RenderOptions(seed=seed)) for io, seed in jobs]
    got = [[None] * 3 for _ in jobs]
    errors = []

    def work(k):
        try:
            for rep in range(3):
                got[k][rep] = sc.render(cam, jobs[k][0], rt.RenderOptions(seed=jobs[k][1]))
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    for k in range(len(jobs)):
        for rep in range(3):
            assert np.array_equal(got[k][rep], want[k]), (k, rep)


def test_a_stream_of_raised_priority_is_refused(earth):
    """A high-priority stream made to wait for normal-priority streams that still have work never gets released on this runtime
    (what hung gloo's staging of device tensors, which uses such a stream, behind a frame still being rendered): the library
    refuses to be ordered with one instead of hanging the caller."""
    sc, co = _example("cornell", earth)
    cam = rt.Camera(co)
    buf = torch.zeros((32, 32, 3), dtype=torch.float64, device="cuda")
    high = torch.cuda.Stream(priority=-1)
    with pytest.raises(_abi.RtError) as e:
        sc.render_device(cam, rt.ImageOptions(32, 32, 4, 50), rt.RenderOptions(seed=1, stream=high.cuda_stream), buf.data_ptr())
    assert e.value.status == _abi.RT_ERR_UNSUPPORTED
    normal = torch.cuda.Stream(priority=0)
    sc.render_device(cam, rt.ImageOptions(32, 32, 4, 50), rt.RenderOptions(seed=1, stream=normal.cuda_stream), buf.data_ptr())
    normal.synchronize()
    assert buf.any()
