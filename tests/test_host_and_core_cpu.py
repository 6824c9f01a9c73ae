"""CPU-side tests of the product's host logic and of the kernel arithmetic (tests/twin builds
raytracing_amd/csrc/rt_core.h with g++): flattening, BVH order, camera, and hit / shading /
texture / image parity with the oracle.  No GPU needed; the GPU runs of the same comparisons are
in test_gpu_parity.py."""
import ctypes as C
import re
import zlib

import numpy as np
import pytest

import raytracing_amd as rt
from raytracing_amd import _abi
from oracle_lib import (OracleScene, OracleShape, oracle, oracle_camera_rays, quantise_image, rmse, twin, twin_intersect, twin_render)
from scenes_util import EXAMPLES, compare_hits, degenerate_rays, path_rays, rays_at_box, shape_zoo

INF = float("inf")
WHITE = rt.Material.Lambertian(rt.Texture.SolidColor((0.7, 0.7, 0.7)))


def _example(name, earth, **kw):
    return rt.Scene.example(name, 1, earth if name == "final_scene" else None, **kw)


# ------------------------------------------------------------------ the boundary
def test_abi_header_symbols_are_exported():
    """Every function include/rt_abi.h declares is exported by librt_hip.so and bound (no compute calls)."""
    import os
    hdr = open(os.path.join(os.path.dirname(_abi.__file__), "..", "include", "rt_abi.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z_]+)\s*\(", hdr)) - {"rt_scene"}
    assert declared == set(_abi.HIP_SYMBOLS), declared ^ set(_abi.HIP_SYMBOLS)
    lib = _abi.hip_lib()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.rt_abi_version() == 3
    _abi.host_lib()


def test_struct_sizes_match_the_header():
    sizes = {_abi.rt_prim: 32, _abi.rt_sphere: 64, _abi.rt_rect: 48, _abi.rt_box: 48, _abi.rt_xform: 32, _abi.rt_material: 32, _abi.rt_texture: 32,
             _abi.rt_image: 16, _abi.rt_bvh_node: 32, _abi.rt_ray: 28, _abi.rt_hit: 56, _abi.rt_camera: 168}
    for t, n in sizes.items():
        assert C.sizeof(t) == n, t


def test_no_device_is_an_error_not_a_fallback():
    lib = _abi.hip_lib()
    if lib.rt_device_count() > 0:
        pytest.skip("a GPU is present")
    sc, _ = rt.Scene.example("cornell", 1)
    with pytest.raises(_abi.RtError) as e:
        sc.device()
    assert e.value.status == _abi.RT_ERR_NO_DEVICE


def test_scene_validation_errors():
    with pytest.raises(ValueError):
        rt.Scene([], (0, 0, 0))                     # lib.rs:161: assert!(!objects.is_empty())
    nested = rt.Shape.ConstantMedium(rt.Shape.ConstantMedium(rt.Shape.new_box((0, 0, 0), (1, 1, 1)), 1.0), 1.0)
    with pytest.raises(ValueError):
        rt.Scene([rt.Object(nested, WHITE)], (0, 0, 0))
    with pytest.raises(ValueError):
        rt.Camera(rt.CameraOptions(0.0, 1.0, 10.0, (0, 0, 0), (0, 0, 1), 0.5, 0.5, 40.0, (0, 1, 0)))  # gen_range panics on an empty range
    lib = _abi.hip_lib()
    out = C.c_void_p()
    assert lib.rt_scene_create(None, -1, C.byref(out)) == _abi.RT_ERR_INVALID_ARGUMENT
    sc, _ = rt.Scene.example("cornell", 1)
    d = sc.desc
    saved = d.bvh[0].skip
    d.bvh[0].skip = 0   # a skip link that does not move forward would never terminate on the device
    assert lib.rt_scene_create(C.byref(d), -1, C.byref(out)) == _abi.RT_ERR_INVALID_ARGUMENT
    d.bvh[0].skip = saved
    saved = d.prims[0].material
    d.prims[0].material = 999
    assert lib.rt_scene_create(C.byref(d), -1, C.byref(out)) == _abi.RT_ERR_INVALID_ARGUMENT
    d.prims[0].material = saved
    d.n_prims, n = 0, d.n_prims
    assert lib.rt_scene_create(C.byref(d), -1, C.byref(out)) == _abi.RT_ERR_EMPTY_SCENE
    d.n_prims = n


# ------------------------------------------------------------------ flattener / BVH / camera
def test_bounds_match_the_oracle():
    be = rt.api.host_backend()
    for name, shape in shape_zoo():
        h = shape.build(be)
        mn, mx = (C.c_double * 3)(), (C.c_double * 3)()
        be.shape_bound(h, mn, mx)
        be.shape_free(h)
        omn, omx = OracleShape(shape).bound()
        assert np.array_equal(np.array(mn), omn) and np.array_equal(np.array(mx), omx), name


def test_flattened_examples(earth):
    expect = {"random_spheres": None, "cornell": 8, "perlin": 4, "cornell_smoke": 8, "final_scene": 1410}
    for name in EXAMPLES:
        sc, co = _example(name, earth)
        osc, oco = OracleScene.example(name, 1, earth if name == "final_scene" else None)
        d = sc.desc
        assert d.n_prims == osc.num_objects()
        if expect[name]:
            assert d.n_prims == expect[name]
        for f in ("aperture", "aspect_ratio", "focus_distance", "shutter_close", "shutter_open", "vfov"):
            assert getattr(co, f) == getattr(oco, f)
        inert = [i for i in range(d.n_prims) if d.prims[i].flags & _abi.RT_PRIM_INERT]
        assert len(inert) == (2 if name == "final_scene" else 0)   # the two sphere-bounded media of final_scene.rs:113-137
        # the leaves, left to right, are the oracle's (= the reference's) leaves minus the inert ones
        leaves = [d.bvh[i].prim & ~_abi.RT_BVH_EXACT for i in range(d.n_bvh_nodes) if d.bvh[i].prim != _abi.RT_BVH_NONE]
        assert leaves == [int(p) for p in osc.leaf_order() if int(p) not in inert]
        sck, _ = _example(name, earth, keep_inert=True)
        dk = sck.desc
        assert [dk.bvh[i].prim & ~_abi.RT_BVH_EXACT for i in range(dk.n_bvh_nodes) if dk.bvh[i].prim != _abi.RT_BVH_NONE] == [int(p) for p in osc.leaf_order()]
        assert dk.n_bvh_nodes == 2 * dk.n_prims - 1
        # structure: subtrees nest, boxes of inner nodes contain their leaves
        for i in range(d.n_bvh_nodes):
            n = d.bvh[i]
            assert i < n.skip <= d.n_bvh_nodes
            for j in range(i + 1, n.skip):
                assert d.bvh[j].skip <= n.skip
                assert all(n.min[c] <= d.bvh[j].min[c] and n.max[c] >= d.bvh[j].max[c] for c in range(3))


def test_camera_frame_by_hand():
    """Camera::new (lib.rs:347-384) on numbers worked out by hand: vfov 90 degrees (h = tan 45 = 1), aspect 2, focus 3, looking
    down -z from (1, 2, 3) with vup = +y: w = (0, 0, 1), u = vup x w = (1, 0, 0), v = w x u = (0, 1, 0); the viewport is
    2h * aspect = 4 wide and 2h = 2 high, times the focus distance; aperture 0.5 -> lens radius 0.25.  Rays (lib.rs:55-60,
    :389-401): direction = lower_left + horizontal * (i + U) / (w - 1) + vertical * (j + U) / (h - 1) - origin - lens offset."""
    co = rt.CameraOptions(aperture=0.5, aspect_ratio=2.0, focus_distance=3.0, looking_at=(1, 2, -7), origin=(1, 2, 3), shutter_close=2.0, shutter_open=1.0,
                          vfov=90.0, vup=(0, 1, 0))
    cam = rt.Camera(co)
    a = cam.abi
    assert np.allclose(list(a.horizontal), [12, 0, 0], atol=1e-12) and np.allclose(list(a.vertical), [0, 6, 0], atol=1e-12)
    assert np.allclose(list(a.lower_left_corner), [1 - 6, 2 - 3, 3 - 3], atol=1e-12) and list(a.origin) == [1, 2, 3]
    assert np.allclose(list(a.u), [1, 0, 0], atol=1e-15) and np.allclose(list(a.v), [0, 1, 0], atol=1e-15) and a.lens_radius == 0.25
    w, h, n = 5, 3, 400
    rng = np.random.default_rng(1)
    pi, pj, ss = rng.integers(0, w, n), rng.integers(0, h, n), rng.integers(0, 50, n)
    for o, d, t in (oracle_camera_rays(co.to_c(), w, h, 77, pi, pj, ss),):
        off = o - [1, 2, 3]                                           # the lens offset: inside the disc of radius 0.25 in the u-v plane
        assert (np.hypot(off[:, 0], off[:, 1]) < 0.25).all() and (off[:, 2] == 0).all() and np.hypot(off[:, 0], off[:, 1]).max() > 0.2
        target = o + d                                                # origin' + direction = the point on the focus plane
        s = (target[:, 0] - (-5)) / 12 * (w - 1) - pi                 # = U1 in [0, 1)
        v = (target[:, 1] - (-1)) / 6 * (h - 1) - pj
        assert (s > -1e-9).all() and (s < 1 + 1e-9).all() and (v > -1e-9).all() and (v < 1 + 1e-9).all() and np.allclose(target[:, 2], 0.0, atol=1e-12)
        assert (t >= 1.0).all() and (t < 2.0).all() and t.std() > 0.2  # random_time: uniform in [shutter_open, shutter_close)
    out = np.zeros((n, 7), np.float32)
    pi32, pj32, ss32 = (x.astype(np.uint32) for x in (pi, pj, ss))
    twin().twin_camera_rays(C.byref(a), w, h, 77, n, pi32.ctypes.data, pj32.ctypes.data, ss32.ctypes.data, out.ctypes.data)
    assert np.array_equal(out[:, 0:3], o.astype(np.float32)) and np.abs(out[:, 3:6] - d).max() < 1e-6


def test_camera_frame_matches_the_oracle(earth):
    for name in EXAMPLES:
        _, co = _example(name, earth)
        cam = rt.Camera(co)
        out = (C.c_double * 19)()
        occ = co.to_c()
        oracle().orc_camera_frame(C.byref(occ), out)
        got = list(cam.abi.origin) + list(cam.abi.lower_left_corner) + list(cam.abi.horizontal) + list(cam.abi.vertical) + list(cam.abi.u) + list(cam.abi.v) + [cam.abi.lens_radius]
        assert np.allclose(got, list(out), rtol=0, atol=1e-12)


def test_perlin_table_and_noise():
    a, b = (C.c_uint8 * 256)(), (C.c_uint8 * 256)()
    _abi.host_lib().rth_perm_table(0, a)
    oracle().orc_perm_table(0, b)
    assert bytes(a) == bytes(b) and sorted(bytes(a)) == list(range(256))
    rng = np.random.default_rng(3)
    pts = (rng.random((3000, 3)) * 2 - 1) * 40
    ref = np.array([oracle().orc_perlin3(a, *p) for p in pts.astype(np.float32).astype(np.float64)])
    got = np.array([twin().twin_perlin3(a, *p) for p in pts.astype(np.float32)])
    assert np.abs(ref).max() <= 1.0 and np.abs(ref).max() > 0.5
    assert np.abs(got - ref).max() < 2e-5
    # lattice points are zeros of gradient noise
    assert oracle().orc_perlin3(a, 3.0, -7.0, 11.0) == 0.0


# ------------------------------------------------------------------ hit-record parity (CPU build of the kernel code)
@pytest.mark.parametrize("name,shape", shape_zoo(), ids=[n for n, _ in shape_zoo()])
def test_shape_intersect_parity(name, shape):
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    osh = OracleShape(shape)
    sc = rt.Scene([rt.Object(shape, WHITE)], (0, 0, 0), keep_inert=True)
    mn, mx = osh.bound()
    o, d, t = rays_at_box(rng, mn, mx, 6000)
    mu = rng.random(6000).astype(np.float32)
    for (tmin, tmax) in [(0.001, INF), (-INF, INF), (0.5, 40.0)]:
        ref = osh.intersect(o, d, t, tmin, tmax, medium_u=mu)
        got = twin_intersect(sc, o, d, t, tmin, tmax, scope=0, medium_u=mu)
        nsame, _ = compare_hits(got, ref, f"{name} ({tmin},{tmax})", allow_flag_mismatch=2e-3, origins=o)
        if name != "medium_sphere":
            assert nsame > 50, (name, nsame)
        else:
            assert ref["hit"].sum() == 0 and got["hit"].sum() == 0


@pytest.mark.parametrize("name,shape", shape_zoo(), ids=[n for n, _ in shape_zoo()])
def test_shape_intersect_parity_on_degenerate_rays(name, shape):
    """Axis-parallel rays (zero and negative-zero direction components), origins exactly on faces, the zero direction and
    a NaN direction: infinities and NaNs must flow as in the reference (shape.rs:321-329, :580-599; geometry.rs:340-346)."""
    osh = OracleShape(shape)
    sc = rt.Scene([rt.Object(shape, WHITE)], (0, 0, 0), keep_inert=True)
    mn, mx = osh.bound()
    o, d, t = degenerate_rays(mn, mx)
    mu = np.full(len(o), 0.37, np.float32)
    for (tmin, tmax) in [(0.001, INF), (-INF, INF), (0.0, 10.0)]:
        ref = osh.intersect(o, d, t, tmin, tmax, medium_u=mu)
        got = twin_intersect(sc, o, d, t, tmin, tmax, scope=0, medium_u=mu)
        # exact agreement for shapes in axis-aligned position; a rotated shape's corners touch its bounding box, so rays
        # through the box's corners and edges graze it, where fp32 and f64 may fall on different sides
        turned = "rot" in name or name in ("medium_wrapped", "medium_box")
        compare_hits(got, ref, f"{name} ({tmin},{tmax})", allow_flag_mismatch=2e-2 if turned else 0.0, allow_record_mismatch=2e-2 if turned else 0.0, origins=o)


def test_scene_closest_hit_parity_on_degenerate_rays(earth):
    """The same crafted rays through the BVH of every example scene (slab tests with zero, negative-zero and NaN direction
    components: shape.rs:580-599), grid laid over the scene's finite extent; in the Cornell rooms its outer planes are the
    walls' planes, so many rays lie in the plane of a wall, where the rectangle's plane parameter is 0 / 0, accepted by
    the reference as a hit with t = NaN (shape.rs:321-329) unless the wall's own bounding-box test turns the ray away first
    (lib.rs:236-240): plain rectangles are exact leaves for that."""
    for name in EXAMPLES:
        sc, co = _example(name, earth)
        osc, oco = OracleScene.example(name, 1, earth if name == "final_scene" else None)
        mn, mx = {"random_spheres": ((-11, 0, -11), (11, 1.2, 11)), "cornell": ((0, 0, 0), (555, 555, 555)), "perlin": ((-4, 0, -4), (5, 4, 4)),
                  "cornell_smoke": ((0, 0, 0), (555, 555, 555)), "final_scene": ((-100, 0, 0), (560, 560, 560))}[name]
        o, d, t = degenerate_rays(mn, mx)
        ref = osc.intersect(o, d, t, seed=9)
        got = twin_intersect(sc, o, d, t, seed=9)
        compare_hits(got, ref, name, allow_flag_mismatch=2e-2, allow_prim_mismatch=2e-2, allow_record_mismatch=2e-2, origins=o)


def test_scene_closest_hit_parity_along_paths(earth):
    """Closest hit over the whole scene (BVH) for rays as the renderer meets them."""
    rng = np.random.default_rng(11)
    for name in EXAMPLES:
        sc, co = _example(name, earth)
        osc, oco = OracleScene.example(name, 1, earth if name == "final_scene" else None)
        total = 0
        for o, d, t in path_rays(osc, oco, 40, 30, 5, 1500, 5, rng):
            ref = osc.intersect(o, d, t, seed=9)
            got = twin_intersect(sc, o, d, t, seed=9)
            nsame, _ = compare_hits(got, ref, name, allow_flag_mismatch=1e-3, allow_prim_mismatch=1e-3, origins=o)
            total += nsame
        assert total > 1500, (name, total)


def test_material_scatter_parity(earth):
    lib, tw = oracle(), twin()
    rng = np.random.default_rng(4)
    for name in EXAMPLES:
        sc, co = _example(name, earth)
        osc, oco = OracleScene.example(name, 1, earth if name == "final_scene" else None)
        desc = sc._host.rth_scene_desc(sc._h)
        kinds = set()
        for b, (o, d, t) in enumerate(path_rays(osc, oco, 40, 30, 6, 600, 3, rng)):
            ho = osc.intersect(o, d, t, seed=6)
            ht = twin_intersect(sc, o, d, t, seed=6)
            for k in np.where((ho["hit"] == 1) & (ht["hit"] == 1) & (ho["prim"] == ht["prim"]))[0][:400]:
                h32 = _abi.rt_hit.from_buffer_copy(ht[k].tobytes())
                s1, c1, o1, d1 = C.c_int32(), (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
                tw.twin_material_scatter(desc, C.byref(h32), 6, int(k), 2, b, C.byref(s1), c1, o1, d1)
                hk = ho[k:k + 1].copy()
                s2, c2, o2, d2 = C.c_int32(), (C.c_double * 3)(), (C.c_double * 3)(), (C.c_double * 3)()
                lib.orc_scene_scatter(osc.h, hk.ctypes.data, 6, int(k), 2, b, C.byref(s2), c2, o2, d2)
                mat = desc.contents.materials[desc.contents.prims[int(ht[k]["prim"])].material].kind
                kinds.add(mat)
                if s1.value != s2.value:
                    # Metal: the sign of scattered . normal decides (materials.rs:52); fp32 may disagree within rounding of zero
                    assert mat == 1
                    continue
                assert np.allclose(np.array(c1), np.array(c2), rtol=2e-5, atol=2e-5), (name, mat)
                if s1.value:
                    assert np.allclose(np.array(d1), np.array(d2), rtol=1e-5, atol=2e-5), (name, mat)
                    assert np.allclose(np.array(o1), np.array(o2), rtol=1e-5, atol=1e-5), (name, mat)
        assert kinds, name


def test_camera_ray_parity(earth):
    rng = np.random.default_rng(8)
    for name in EXAMPLES:
        _, co = _example(name, earth)
        cam = rt.Camera(co)
        n, w, h = 2000, 64, 36
        pi, pj, ss = (rng.integers(0, m, n).astype(np.uint32) for m in (w, h, 1000))
        occ = co.to_c()
        o, d, t = oracle_camera_rays(occ, w, h, 12345678901, pi, pj, ss)
        out = np.zeros((n, 7), np.float32)
        twin().twin_camera_rays(C.byref(cam.abi), w, h, 12345678901, n, pi.ctypes.data, pj.ctypes.data, ss.ctypes.data, out.ctypes.data)
        assert np.array_equal(out[:, 0:3], o.astype(np.float32)) and np.array_equal(out[:, 6], t.astype(np.float32))
        assert np.abs(out[:, 3:6] - d).max() < 1e-6


def test_image_parity_with_common_random_numbers(earth):
    """Same Philox streams on both sides: the fp32 image stays far inside the oracle's own
    seed-to-seed noise (tolerance: RMSE on the 8-bit image <= 0.5 x that noise floor)."""
    for name, spp in [("random_spheres", 16), ("cornell", 32), ("perlin", 16), ("cornell_smoke", 32), ("final_scene", 8)]:
        sc, co = _example(name, earth)
        osc, oco = OracleScene.example(name, 1, earth if name == "final_scene" else None)
        cam = rt.Camera(co)
        w, h = (48, 27) if co.aspect_ratio > 1 else (32, 32)
        a, _, _, _ = osc.render(oco, w, h, spp, 50, seed=1, threads=8)
        b, _, _, _ = osc.render(oco, w, h, spp, 50, seed=2, threads=8)
        mean = twin_render(sc, cam, w, h, spp, 50, seed=1, threads=8)
        assert not np.isnan(mean).any(), name
        floor, err = rmse(a, b), rmse(quantise_image(mean), a)
        assert err <= 0.5 * floor, (name, err, floor)


def _decode_png(data):
    """Minimal PNG reader for 8-bit RGB without interlace: returns (width, height, rgb bytes); checks every CRC."""
    import struct
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    at, idat, head = 8, b"", None
    while at < len(data):
        n, kind = struct.unpack(">I4s", data[at:at + 8])
        body = data[at + 8:at + 8 + n]
        assert struct.unpack(">I", data[at + 8 + n:at + 12 + n])[0] == zlib.crc32(kind + body)
        if kind == b"IHDR":
            head = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat += body
        at += 12 + n
    w, h, depth, colour, _, filt, interlace = head
    assert (depth, colour, filt, interlace) == (8, 2, 0, 0)
    raw = zlib.decompress(idat)          # checks the Adler-32 of the stored blocks
    rows = [raw[y * (3 * w + 1):(y + 1) * (3 * w + 1)] for y in range(h)]
    assert all(r[0] == 0 for r in rows)  # filter type None on every scanline
    return w, h, b"".join(r[1:] for r in rows)


def test_image_writers(tmp_path):
    """write_png / write_ppm, which stand in for ImageBuffer::from_raw(..).save("test.png") of the examples (cornell.rs:109-116).
    The PNG is large enough to need more than one stored deflate block (65 535 bytes each)."""
    host = _abi.host_lib()
    rng = np.random.default_rng(5)
    for w, h in [(5, 3), (1, 1), (211, 173)]:
        px = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for ext in ("png", "ppm"):
            path = str(tmp_path / f"img_{w}x{h}.{ext}")
            assert host.rth_write_image(path.encode(), w, h, px.ctypes.data) == 0
            data = open(path, "rb").read()
            if ext == "png":
                assert _decode_png(data) == (w, h, px.tobytes())
            else:
                assert data == b"P6\n%d %d\n255\n" % (w, h) + px.tobytes()
                assert np.array_equal(rt.load_ppm(path), px)
    assert host.rth_write_image(b"/nonexistent-dir/x.png", 1, 1, px.ctypes.data) != 0


def test_example_driver_is_built_and_refuses_to_run_without_a_gpu():
    """examples/rt_example (the reference's five example programs as one driver): usage errors exit with 2; on a machine
    without a HIP device a render exits with 1 and says so -- there is no CPU path to fall back to."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "rt_example")
    assert os.path.exists(exe), "run `make -C raytracing_amd/csrc`"
    if _abi.hip_lib().rt_device_count() == 0:  # (no child processes from a test process that may hold a GPU)
        assert subprocess.run([exe], capture_output=True).returncode == 2
        assert subprocess.run([exe, "teapot"], capture_output=True).returncode == 2
        r = subprocess.run([exe, "cornell", "--width", "8", "--spp", "1", "--out", "/dev/null"], capture_output=True, text=True)
        assert r.returncode == 1 and "no HIP device" in r.stderr


def _walk_nodes(sc):
    """The library's own node array for a host scene (rt::make_walk_nodes through the CPU twin): a structured array of
    32-byte slots {x0, x1, y0, y1, z0, z1, skip, prim}, the last one being the sentinel."""
    tw = twin()
    desc = sc._host.rth_scene_desc(sc._h)
    n = tw.twin_walk_nodes(desc, None, 0)
    raw = np.zeros((n, 8), dtype=np.uint32)
    assert tw.twin_walk_nodes(desc, raw.ctypes.data, n) == n
    return raw


def test_walk_node_array_keeps_the_reference_leaf_order(earth):
    """The node array the kernels walk (regrouped inner nodes, boxes as centre and half-widths, payload slots, sentinel): same leaves in the same
    order as the tree that crosses the ABI, nested subtrees, inner boxes containing everything below them, fewer nodes."""
    PRIM_MASK, PLAIN_RECT, LEAF, PLAIN_BOX, PLAIN_SPHERE = 0x07FFFFFF, 0x08000000, 0x10000000, 0x40000000, 0x20000000
    for name in EXAMPLES:
        sc, _ = _example(name, earth)
        d = sc.desc
        raw = _walk_nodes(sc)
        # a node's box: centre and half-width per axis, slots (hx, hy, cx, cy, hz, cz); [c - h, c + h] evaluated exactly in f64
        with np.errstate(invalid="ignore"):   # (payload slots read as floats are anything)
            ch = raw[:, :6].view(np.float32).astype(np.float64)
        ctr, half = ch[:, [2, 3, 5]], ch[:, [0, 1, 4]]
        f = np.empty((len(raw), 6))
        f[:, 0::2], f[:, 1::2] = ctr - half, ctr + half
        skip, prim = raw[:, 6].copy(), raw[:, 7].copy()
        n = len(raw) - 1
        byte_skip, byte_prim = skip.copy(), prim.copy()                  # node positions are byte offsets: slot number * 32
        skip //= 32
        prim[prim < LEAF] //= 32                                         # (an inner node's word is the position of its first child)
        assert skip[n] == n and prim[n] == n                            # sentinel: both links to itself
        is_payload = np.zeros(n + 2, bool)
        tight = f.copy()   # the boxes that have to nest: a node's own box, for exact leaves the exact box of the payload slot
        leaves, i = [], 0
        while i < n:                                                     # every slot is a node or the payload of the sphere leaf before it
            if prim[i] >= LEAF:
                leaves.append(int(prim[i] & PRIM_MASK))
                P = d.prims[leaves[-1]]
                if prim[i] & PLAIN_SPHERE:
                    moving = bool(prim[i] & PLAIN_RECT)                      # (the two bits together: an unwrapped MovingSphere, two payload slots)
                    assert P.geom == (_abi.RT_GEOM_MOVING_SPHERE if moving else _abi.RT_GEOM_SPHERE) and not (P.flags & _abi.RT_PRIM_MEDIUM)
                    is_payload[i + 1] = True
                    c = raw[i + 1].view(np.float64)
                    sp = d.spheres[P.geom_index]
                    assert c[3] == sp.radius
                    if P.xf_count == 0:
                        assert list(c[:3]) == list(sp.center)
                    if moving:
                        assert P.xf_count == 0
                        is_payload[i + 2] = True
                        m = raw[i + 2].view(np.float64)
                        assert list(m[:3]) == list(sp.motion) and m[3] == sp.start_time
                    assert skip[i] == i + (3 if moving else 2)
                    i += 2 if moving else 1
                elif prim[i] & _abi.RT_BVH_EXACT:
                    # exact leaves: the exact box in the next slot, inside the (widened) box of the node itself
                    is_payload[i + 1] = True
                    ex = raw[i + 1, :6].view(np.float32)
                    assert (f[i, 0::2] < ex[:3]).all() and (f[i, 1::2] > ex[3:]).all()
                    assert (f[i, 0::2] > ex[:3] - 0.02).all() and (f[i, 1::2] < ex[3:] + 0.02).all()   # ... and not much wider
                    tight[i, 0::2], tight[i, 1::2] = ex[:3], ex[3:]
                    n_slots = 1
                    if prim[i] & PLAIN_BOX:
                        assert P.geom == _abi.RT_GEOM_BOX and P.xf_count == 0
                        is_payload[i + 2] = True
                        n_slots = 2
                        bx = d.boxes[P.geom_index]
                        full = ex.astype(np.float64) + raw[i + 2, :6].view(np.float32).astype(np.float64)   # value + remainder
                        assert np.allclose(full, list(bx.min) + list(bx.max), rtol=2e-14, atol=0)
                    elif prim[i] & PLAIN_RECT:
                        # a plain rectangle: its bound() (shape.rs:457-481) as value + remainder, the plane, the axis
                        assert P.geom == _abi.RT_GEOM_RECT and P.xf_count == 0
                        is_payload[i + 2] = True
                        n_slots = 2
                        rc = d.rects[P.geom_index]
                        p1, p2, oo = [(1, 2, 0), (2, 0, 1), (0, 1, 2)][rc.axis]
                        mn, mx = [0.0] * 3, [0.0] * 3
                        mn[p1], mn[p2], mn[oo] = rc.x0, rc.y0, rc.height - 0.0001
                        mx[p1], mx[p2], mx[oo] = rc.x1, rc.y1, rc.height + 0.0001
                        full = ex.astype(np.float64) + raw[i + 2, :6].view(np.float32).astype(np.float64)
                        assert np.allclose(full, mn + mx, rtol=2e-14, atol=0)
                        h = raw[i + 1, 6:8].view(np.float32).astype(np.float64).sum()
                        assert abs(h - rc.height) <= 2e-14 * max(1.0, abs(rc.height)) and raw[i + 2, 6] == rc.axis
                    assert skip[i] == i + 1 + n_slots
                    i += n_slots
                else:
                    assert not (prim[i] & PLAIN_BOX)
                    assert skip[i] == i + 1
            i += 1
        assert leaves == [d.bvh[k].prim & 0x1FFFFFFF for k in range(d.n_bvh_nodes) if d.bvh[k].prim != _abi.RT_BVH_NONE]
        nodes = [k for k in range(n) if not is_payload[k]]
        assert len(nodes) <= d.n_bvh_nodes
        if name == "final_scene":
            assert len(nodes) < 0.85 * d.n_bvh_nodes                    # 2 815 in the caller's binary tree
        for k in nodes:
            assert byte_skip[k] % 32 == 0 and (byte_prim[k] >= LEAF or byte_prim[k] % 32 == 0)
            assert k < skip[k] <= n
            keep = ~is_payload[k + 1:skip[k]]
            if keep.any():
                assert prim[k] == k + 1                                 # inner: the index of its first child
                assert skip[k + 1:skip[k]][keep].max() <= skip[k]
                sub = tight[k + 1:skip[k]][keep]
                # (a box is rounded to centre and half-width on its own, outwards: a child's rounding may stick out of its
                # parent's by an fp32 ulp or two, which is far inside the padding every box carries)
                tol = 2.0 ** -21 * max(1.0, np.abs(f[:n][np.isfinite(f[:n])]).max())
                assert (f[k, 0::2] <= sub[:, 0::2].min(axis=0) + tol).all() and (f[k, 1::2] >= sub[:, 1::2].max(axis=0) - tol).all()


def test_wrapped_sphere_leaves_carry_the_centre_in_the_outer_frame():
    """A Sphere behind Translated / Rotated wrappers is tested as a sphere around the wrapped centre carried out through the
    wrappers (rotations keep lengths): the payload slot holds that centre, and hits agree with the oracle's frame-by-frame
    evaluation (shape.rs:356-384) to the hit-record tolerance."""
    shape = rt.Shape.Sphere((3.0, 1.0, -2.0), 1.5).rotated(1, 30.0).translated((10.0, -4.0, 7.0)).rotated(0, -70.0)
    sc = rt.Scene([rt.Object(shape, rt.Material.Lambertian(rt.Texture.SolidColor((0.5, 0.5, 0.5))))], (0, 0, 0))
    raw = _walk_nodes(sc)
    k = [i for i in range(len(raw) - 1) if raw[i, 7] >= 0x10000000 and (raw[i, 7] & 0x07FFFFFF) == 0][0]
    c = raw[k + 1].view(np.float64)
    p = np.array([3.0, 1.0, -2.0])
    a = np.radians(30.0); p = np.array([p[2] * np.sin(a) + p[0] * np.cos(a), p[1], p[2] * np.cos(a) - p[0] * np.sin(a)])   # rotate Y (geometry.rs:33-50)
    p = p + [10.0, -4.0, 7.0]
    a = np.radians(-70.0); p = np.array([p[0], p[1] * np.cos(a) - p[2] * np.sin(a), p[1] * np.sin(a) + p[2] * np.cos(a)])  # rotate X
    assert np.allclose(c[:3], p, rtol=0, atol=1e-12) and c[3] == 1.5
    osh = OracleShape(shape)
    rng = np.random.default_rng(11)
    mn, mx = osh.bound()
    o, dd, t = rays_at_box(rng, mn, mx, 4000)
    ref = osh.intersect(o, dd, t, 0.001, float("inf"))
    got = twin_intersect(sc, o, dd, t)
    nsame, _ = compare_hits(got, ref, "wrapped sphere", allow_flag_mismatch=2e-3, origins=o)
    assert nsame > 500


def test_noise_kinds_fbm_and_turbulence():
    """The two noise functions beyond Perlin that a Noise texture can name (rt_abi.h RT_NOISE_*): their definitions by hand on the
    oracle, the flattened description, and the kernel code (CPU build) against the oracle at random points."""
    from oracle_lib import oracle_backend
    from scenes_util import noise_kinds_scene
    lib, tw, be = oracle(), twin(), oracle_backend()
    T = rt.Texture
    tables = []
    for seed in range(8):
        buf = (C.c_uint8 * 256)()
        lib.orc_perm_table(seed, buf)
        tables.append(buf)

    def colour(tex, p):
        h = tex.build(be)
        out = (C.c_double * 3)()
        lib.orc_texture_color(h, 0.0, 0.0, (C.c_double * 3)(*p), out)
        be.tex_free(h)
        assert out[0] == out[1] == out[2]
        return out[0]

    def perlin(seed, p):
        return lib.orc_perlin3(tables[seed], *p)

    p = np.array([0.37, -1.21, 2.6])
    assert colour(T.Fbm(1.0, seed=3, octaves=1), p) == perlin(3, p)                                  # one octave is the Perlin of that seed
    lac, pers = 2.0, 0.6
    assert colour(T.Fbm(1.0, seed=0, octaves=2, lacunarity=lac, persistence=pers), p) == perlin(0, p) + perlin(1, p * lac) * pers
    three = perlin(2, p * 0.5) + perlin(3, p * 0.5 * lac) * pers + perlin(4, p * 0.5 * lac * lac) * (pers * pers)
    assert abs(colour(T.Fbm(0.5, seed=2, octaves=3, lacunarity=lac, persistence=pers), p) - three) < 1e-15       # scale multiplies the point first
    assert colour(T.Turbulence(1.0, seed=4, power=0.0), p) == perlin(0, p)                            # no power: the source alone
    u = 1.0 / 65536.0
    off = np.array([[12414, 65124, 31337], [26519, 18128, 60493], [53820, 11213, 44845]]) * u
    q = p + np.array([perlin(4 + c, p + off[c]) for c in range(3)]) * 0.75                             # roughness 1: each distortion is one Perlin
    assert abs(colour(T.Turbulence(1.0, seed=4, roughness=1, power=0.75), p) - perlin(0, q)) < 1e-15
    with pytest.raises(ValueError):
        T.Fbm(1.0, octaves=0).build(be)
    with pytest.raises(ValueError):
        T.Fbm(1.0, octaves=33).build(_abi_backend())

    # the flattened description: function and octaves in `odd`, consecutive tables from `resource` on, shared between functions
    objects, background, _ = noise_kinds_scene()
    sc = rt.Scene(objects, background)
    d = sc.desc
    noises = [d.textures[i] for i in range(d.n_textures) if d.textures[i].kind == 2]
    assert sorted(t.odd & 255 for t in noises) == [_abi.RT_NOISE_PERLIN, _abi.RT_NOISE_FBM, _abi.RT_NOISE_FBM, _abi.RT_NOISE_TURBULENCE]
    for t in noises:
        fn, octaves = t.odd & 255, (t.odd >> 8) & 255
        n_tables = {0: 1, 1: octaves, 2: octaves + 2}[fn]
        assert t.resource + n_tables <= d.n_perm_tables
        seed = {0: 0, 1: 0 if octaves == 6 else 2, 2: 5}[fn]
        for k in range(n_tables):
            buf = (C.c_uint8 * 256)()
            lib.orc_perm_table(seed + k, buf)
            assert bytes(d.perm_tables[256 * (t.resource + k):256 * (t.resource + k + 1)]) == bytes(buf)
    assert d.n_perm_tables == 1 + 6 + 6                                # Perlin's, seeds 0..5, seeds 5..10; the Fbm of seeds 2..4 reuses the second run

    # kernel code against the oracle: emitted colour of a light that wears each texture, at random points near the origin
    texs = [T.Fbm(2.0), T.Turbulence(1.5, seed=5, roughness=4, power=0.75), T.Fbm(3.0, seed=2, octaves=3, lacunarity=2.0, persistence=0.6)]
    sc2 = rt.Scene([rt.Object(rt.Shape.Sphere((3.0 * i, 0.0, 0.0), 1.0), rt.Material.DiffuseLight(t)) for i, t in enumerate(texs)], (0, 0, 0))
    desc = sc2._host.rth_scene_desc(sc2._h)
    rng = np.random.default_rng(21)
    worst = 0.0
    for i, t in enumerate(texs):
        for pt in ((rng.random((300, 3)) * 2 - 1) * 6).astype(np.float32):
            h32 = _abi.rt_hit(1, i, 1.0, (C.c_float * 3)(*pt), (C.c_float * 3)(0, 1, 0), (C.c_float * 3)(0, -1, 0), 0.0, 0.0)
            s1, c1, o1, d1 = C.c_int32(), (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
            tw.twin_material_scatter(desc, C.byref(h32), 6, 0, 0, 0, C.byref(s1), c1, o1, d1)
            ref = colour(t, pt.astype(np.float64))
            assert s1.value == 0
            worst = max(worst, abs(c1[0] - ref))
    assert worst < 5e-5, worst


def _abi_backend():
    from raytracing_amd.api import host_backend
    return host_backend()


def test_final_scene_book_variant_flattens_with_the_switches(earth):
    """final_scene_book (not a reference example): final_scene's objects with every sphere a far-root sphere, every material with the
    facing normal, the book's camera; the media bounded by spheres are no longer inert and stand in the tree."""
    a, _ = rt.Scene.example("final_scene", 1, earth)
    b, co = rt.Scene.example("final_scene_book", 1, earth)
    da, db = a.desc, b.desc
    assert db.n_prims == da.n_prims == 1410 and tuple(co.origin) == (478.0, 278.0, -600.0) and co.vfov == 40.0 and co.aperture == 0.0
    for i in range(db.n_prims):
        pa, pb = da.prims[i], db.prims[i]
        assert pb.geom == pa.geom and pb.geom_index == pa.geom_index
        sphere = pb.geom in (_abi.RT_GEOM_SPHERE, _abi.RT_GEOM_MOVING_SPHERE)
        assert bool(pb.flags & _abi.RT_PRIM_BOTH_ROOTS) == sphere and not (pb.flags & _abi.RT_PRIM_INERT)
        assert db.materials[pb.material].reserved[1] == _abi.RT_MATF_FACE_NORMAL and da.materials[pa.material].reserved[1] == 0
    assert sum(1 for i in range(da.n_prims) if da.prims[i].flags & _abi.RT_PRIM_INERT) == 2
    leaves = lambda d: sum(1 for k in range(d.n_bvh_nodes) if d.bvh[k].prim != _abi.RT_BVH_NONE)
    assert leaves(db) == 1410 and leaves(da) == 1408


def test_regrouping_by_sample_rays_keeps_leaves_and_hits(earth):
    """The inner nodes of the walk's tree regrouped by the share of sample rays that meet a box instead of by box areas
    (SampleRays in rt_core.h; the renderer samples its own queues): the same leaves in the same order with the same payloads,
    hit records equal bit for bit, a fifth fewer box tests on the rays of final_scene."""
    from oracle_lib import rays_to_c
    tw = twin()
    sc, co = _example("final_scene", earth)
    osc, oco = OracleScene.example("final_scene", 1, earth)
    desc = sc._host.rth_scene_desc(sc._h)
    LEAF = 0x10000000

    def rays_of(seed, n):
        parts = list(path_rays(osc, oco, 64, 64, 5, n, 8, np.random.default_rng(seed)))
        return tuple(np.concatenate([p[k] for p in parts]) for k in range(3))

    def leaves_and_payloads(raw):
        out, i, n = [], 0, len(raw) - 1
        while i < n:
            if raw[i, 7] >= LEAF:
                nxt = raw[i, 6] // 32
                out.append((int(raw[i, 7]), raw[i + 1:nxt].tobytes()))
                i = nxt
            else:
                i += 1
        return out

    def work(o, d, t):
        rays = rays_to_c(o, d, t)
        out = np.zeros((len(rays), 3), dtype=np.uint32)
        tw.twin_traverse_stats(desc, rays.ctypes.data, len(rays), 1, out.ctypes.data, None, 0)
        return out.sum(axis=0) / len(rays)

    test = rays_of(3, 1200)
    train = rays_to_c(*(a[:300] for a in rays_of(11, 60)))
    raw0, hits0, work0 = _walk_nodes(sc), twin_intersect(sc, *test), work(*test)
    tw.twin_set_sample_rays(train.ctypes.data, len(train))
    try:
        raw1, hits1, work1 = _walk_nodes(sc), twin_intersect(sc, *test), work(*test)
    finally:
        tw.twin_set_sample_rays(None, 0)
    assert leaves_and_payloads(raw0) == leaves_and_payloads(raw1) and len(leaves_and_payloads(raw0)) == 1408
    assert hits0.tobytes() == hits1.tobytes()
    assert work1[0] < 0.85 * work0[0] and work1[1] == work0[1] and work1[2] == work0[2], (work0, work1)   # box tests; leaves reached and primitive tests as before
    assert np.array_equal(_walk_nodes(sc), raw0)                                                         # without sample rays: by areas again


def test_origins_are_rounded_to_the_side_a_ray_leaves_on():
    """side_of_surface (rt_core.h): the fp32 origin of a ray that leaves a plane off the fp32 grid lies on the leaving side of
    the f64 point in every component the normal has, at most one fp32 step from the nearest rounding; components the normal
    lacks, exact values, zeros and NaNs keep the nearest rounding."""
    tw = twin()
    rng = np.random.default_rng(11)
    n = 4000
    pts = rng.uniform(-600.0, 600.0, (n, 3))
    pts[:50] = np.round(pts[:50])                # exactly representable
    pts[50:60, 0] = 0.0                          # zeros
    pts[60:70, 1] = rng.uniform(-1e-14, 1e-14, 10)   # residues around zero: representable in fp32 as they are
    for normal in ([0.9659258, 0.0, -0.2588190], [-0.30901699, 0.0, 0.95105652], [0.0, 1.0, 0.0]):
        nv = np.array(normal, np.float32)
        for along in (1, 0):
            out = np.zeros((n, 3), np.float32)
            p = np.ascontiguousarray(pts)
            tw.twin_side_of_surface(p.ctypes.data, nv.ctypes.data, along, n, out.ctypes.data)
            near = pts.astype(np.float32)
            s = np.sign(nv.astype(np.float64)) * (1.0 if along else -1.0)
            for c in range(3):
                moved = out[:, c] != near[:, c]
                if s[c] == 0.0:
                    assert not moved.any()
                    continue
                assert ((out[:, c].astype(np.float64) - pts[:, c]) * s[c] >= 0.0).all()          # on the leaving side
                step = np.abs(np.nextafter(near[:, c], np.float32(np.inf) * np.float32(s[c])) - near[:, c])
                assert (np.abs(out[:, c] - near[:, c]) <= step).all()                               # one step at most
                exact = near[:, c].astype(np.float64) == pts[:, c]
                assert not (moved & exact).any()
                assert moved.sum() > n // 4                                                         # about half of the inexact ones
    bad = np.array([[np.nan, 1.0, np.inf]])
    out = np.zeros((1, 3), np.float32)
    nv = np.array([1.0, 1.0, 1.0], np.float32)
    tw.twin_side_of_surface(bad.ctypes.data, nv.ctypes.data, 1, 1, out.ctypes.data)
    assert np.isnan(out[0, 0]) and out[0, 1] == 1.0 and np.isinf(out[0, 2])
