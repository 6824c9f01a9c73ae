"""GPU parity tests: everything goes through the C ABI of librt_hip.so and is compared with the oracle.

Tolerances
  hit records: scenes_util.TOL (1e-5 relative; see there for the point);
  images: RMSE of the 8-bit image against the oracle rendered with the same Philox streams must stay
    below 0.5 x the oracle's own seed-to-seed RMSE at that sample count (the stated bound for converged
    images is 1.5 x that floor; common random numbers make the actual distance far smaller);
  invariance: byte-identical mean-radiance buffers across pass sizes, kernel splits and rank counts.
"""
import ctypes as C
import zlib

import numpy as np
import pytest

import raytracing_amd as rt
from raytracing_amd import _abi
from oracle_lib import OracleScene, OracleShape, oracle, oracle_camera_rays, rmse
from scenes_util import EXAMPLES, compare_hits, degenerate_rays, path_rays, rays_at_box, shape_zoo

pytestmark = pytest.mark.gpu
INF = float("inf")
WHITE = rt.Material.Lambertian(rt.Texture.SolidColor((0.7, 0.7, 0.7)))


def _example(name, earth, **kw):
    return rt.Scene.example(name, 1, earth if name == "final_scene" else None, **kw)


def test_library_is_loaded_and_sees_the_gpu():
    assert _abi.hip_lib().rt_device_count() >= 1


@pytest.mark.parametrize("name,shape", shape_zoo(), ids=[n for n, _ in shape_zoo()])
def test_shape_intersect_parity(name, shape):
    """Intersection::intersect (shape.rs:14) for every Shape variant, three intervals, injected medium variates."""
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    osh = OracleShape(shape)
    sc = rt.Scene([rt.Object(shape, WHITE)], (0, 0, 0), keep_inert=True)
    mn, mx = osh.bound()
    o, d, t = rays_at_box(rng, mn, mx, 20000)
    mu = rng.random(20000).astype(np.float32)
    for (tmin, tmax) in [(0.001, INF), (-INF, INF), (0.5, 40.0)]:
        ref = osh.intersect(o, d, t, tmin, tmax, medium_u=mu)
        got = sc.intersect_batch(o, d, t, tmin, tmax, scope=0, medium_u=mu)
        nsame, _ = compare_hits(got, ref, f"{name} ({tmin},{tmax})", allow_flag_mismatch=1e-4, origins=o)
        if name != "medium_sphere":
            assert nsame > 150, (name, nsame)
        else:
            assert got["hit"].sum() == 0


@pytest.mark.parametrize("name,shape", shape_zoo(), ids=[n for n, _ in shape_zoo()])
def test_shape_intersect_parity_on_degenerate_rays(name, shape):
    """Axis-parallel rays (zero and negative-zero direction components), origins exactly on faces, the zero direction and
    a NaN direction: infinities and NaNs must flow as in the reference (shape.rs:321-329, :580-599; geometry.rs:340-346).
    Exact agreement on hit / miss for shapes in axis-aligned position; a rotated shape's corners touch its bounding box,
    so the grid's rays through them graze it, and 2 % of the records may fall on the other side there."""
    osh = OracleShape(shape)
    sc = rt.Scene([rt.Object(shape, WHITE)], (0, 0, 0), keep_inert=True)
    mn, mx = osh.bound()
    o, d, t = degenerate_rays(mn, mx)
    mu = np.full(len(o), 0.37, np.float32)
    turned = "rot" in name or name in ("medium_wrapped", "medium_box")
    for (tmin, tmax) in [(0.001, INF), (-INF, INF), (0.0, 10.0)]:
        ref = osh.intersect(o, d, t, tmin, tmax, medium_u=mu)
        got = sc.intersect_batch(o, d, t, tmin, tmax, scope=0, medium_u=mu)
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
        got = sc.intersect_batch(o, d, t, seed=9)
        compare_hits(got, ref, name, allow_flag_mismatch=2e-2, allow_prim_mismatch=2e-2, allow_record_mismatch=2e-2, origins=o)


def test_zero_and_negative_zero_direction_components():
    """The hand-derived cases of tests/test_oracle.py (same name) through the C ABI: BoundingBox::hit's division by a zero
    direction component, its `d < 0.0` swap that -0.0 escapes (shape.rs:580-599), Rectangle::intersect's NaN-blind
    rejections (shape.rs:321-329)."""
    S = rt.Shape
    ball = rt.Scene([rt.Object(S.Sphere((0, 0, 0), 1.0), WHITE)], (0, 0, 0))
    one = lambda sc, o, d, **kw: sc.intersect_batch([o], [d], [0.0], **kw)[0]
    h = one(ball, [-5, 0, 0], [1, 0.0, 0.0])
    assert h["hit"] == 1 and h["t"] == 4.0
    assert one(ball, [-5, 2, 0], [1, 0.0, 0.0])["hit"] == 0
    assert one(ball, [-5, 0, 0], [1, -0.0, 0.0], scope=0)["hit"] == 1      # the sphere alone: no bounding-box test
    assert one(ball, [-5, 0, 0], [1, -0.0, 0.0])["hit"] == 0               # behind its bounding box: missed
    wall = rt.Scene([rt.Object(S.rectangle((0, 0, 1), (2, 2, 1)), WHITE)], (0, 0, 0))
    for scope in (0, -1):                                                  # the rectangle alone, and behind its leaf's box
        h = one(wall, [-1, 1, 1], [1, 0, 0.0], scope=scope)
        assert h["hit"] == 1 and np.isnan(h["t"]) and np.isnan(h["point"]).all() and list(h["normal"]) == [0, 0, 1]
        assert one(wall, [-1, 1, 2], [1, 0, 0.0], scope=scope)["hit"] == 0
        assert one(wall, [-1, 1, 0], [1, 0, 0.0], scope=scope)["hit"] == 0
    z = one(wall, [1, 1, 0], [0.0, 0.0, 0.0], scope=0)
    assert z["hit"] == 1 and z["t"] == INF and np.isnan(z["point"]).all()
    assert one(wall, [1, 1, 0], [0.0, 0.0, 0.0], scope=0, tmax=1e30)["hit"] == 0
    assert one(wall, [-1, 1, 1], [0.0, 0.0, 0.0])["hit"] == 0


def _gpu_scatter(shape, material, o, d, seed=5, key=(3, 1, 0)):
    sc = rt.Scene([rt.Object(shape, material)], (0, 0, 0))
    h = sc.intersect_batch([o], [d], [0.0], seed=seed)
    assert h[0]["hit"] == 1
    s = sc.scatter_batch(h, seed=seed, keys=[key])[0]
    return h[0], int(s["scattered"]), s["color"].astype(np.float64), s["origin"].astype(np.float64), s["direction"].astype(np.float64)


def test_material_rules_by_hand():
    """The hand-derived cases of tests/test_oracle.py (same name) through rt_intersect_batch + rt_scatter_batch:
    Material::scatter (materials.rs:28-113) where the outcome does not hang on a random draw, or only within bounds.
    fp32 tolerances: 1e-6."""
    S, M, T = rt.Shape, rt.Material, rt.Texture
    floor = S.rectangle((-5, 0, -5), (5, 0, 5))
    down, up = [1.0, -1.0, 0.0], [1.0, 1.0, 0.0]
    r2 = np.sqrt(0.5)
    for o, d in (([-2, 2, 0], down), ([-2, -2, 0], up)):
        _, sc, col, _, _ = _gpu_scatter(floor, M.DiffuseLight(T.SolidColor((15, 7, 4))), o, d)
        assert sc == 0 and list(col) == [15, 7, 4]
    _, sc, col, ro, rd = _gpu_scatter(floor, M.Metal((0.8, 0.6, 0.2), 0.0), [-2, 2, 0], down)
    assert sc == 1 and np.allclose(col, [0.8, 0.6, 0.2], rtol=1e-7) and np.allclose(ro, [0, 0, 0], atol=1e-6) and np.allclose(rd, [r2, r2, 0], atol=1e-6)
    _, sc, col, _, _ = _gpu_scatter(floor, M.Metal((0.8, 0.6, 0.2), 0.0), [-2, -2, 0], up)
    assert sc == 0 and list(col) == [0, 0, 0]
    _, sc, col, ro, rd = _gpu_scatter(floor, M.glass(), [-2, -2, 0], up)
    assert sc == 1 and list(col) == [1, 1, 1] and np.allclose(rd, [r2, -r2, 0], atol=1e-6)
    for sample in range(8):
        _, sc, col, ro, rd = _gpu_scatter(floor, M.glass(), [0, 3, 0], [0.0, -2.0, 0.0], key=(3, sample, 0))
        assert sc == 1 and rd[0] == 0 and rd[2] == 0 and abs(abs(rd[1]) - 1.0) < 1e-6
    for sample in range(8):
        _, sc, col, ro, rd = _gpu_scatter(floor, M.Lambertian(T.SolidColor((0.2, 0.4, 0.6))), [-2, 2, 0], down, key=(9, sample, 2))
        assert sc == 1 and abs(np.linalg.norm(rd - [0, 1, 0]) - 1.0) < 1e-6 and rd[1] > 0 and np.allclose(col, [0.2, 0.4, 0.6], rtol=1e-7)
    _, sc, col, ro, rd = _gpu_scatter(floor, M.Lambertian(T.SolidColor((0.2, 0.4, 0.6))), [-2, -2, 0], up)
    assert sc == 1 and rd[1] > 0 and np.allclose(col, [0.2, 0.4, 0.6], rtol=1e-7)
    _, sc, col, ro, rd = _gpu_scatter(floor, M.Isotropic(T.SolidColor((0.5, 0.5, 0.5))), [-2, 2, 0], down)
    assert sc == 1 and abs(np.linalg.norm(rd) - 1.0) < 1e-6 and list(col) == [0.5, 0.5, 0.5] and np.allclose(ro, [0, 0, 0], atol=1e-6)
    chk = M.DiffuseLight(T.Checkered(T.SolidColor((1, 0, 0)), T.SolidColor((0, 1, 0))))
    wall = S.rectangle((0, 0, 0.1), (1, 1, 0.1))
    for (x, y), want in (((0.1, 0.1), [0, 1, 0]), ((0.4, 0.1), [1, 0, 0])):
        _, sc, col, _, _ = _gpu_scatter(wall, chk, [x, y, -1], [0.0, 0.0, 1.0])
        assert sc == 0 and list(col) == want

    # Image (textures.rs:59-76): u clamped, v flipped and clamped, texel (min(u W, W-1), min((1 - v) H, H-1)), value / 255;
    # on an Axis::Z rectangle u = (x - x0) / (x1 - x0), v = (y - y0) / (y1 - y0) (shape.rs:323-324)
    img = np.arange(4 * 2 * 3, dtype=np.uint8).reshape(2, 4, 3) * 10          # rows = j (top row first), columns = i
    card = S.rectangle((0, 0, 0), (4, 2, 0))
    for (x, y), (i, j) in (((0.5, 1.9), (0, 0)), ((3.5, 1.9), (3, 0)), ((1.5, 0.5), (1, 1)), ((4.0, 0.0), (3, 1)), ((2.0, 1.0), (2, 1))):
        h, sc, col, _, _ = _gpu_scatter(card, M.DiffuseLight(T.Image(img)), [x, y, -1], [0.0, 0.0, 1.0])
        assert sc == 0 and np.allclose(col, img[j, i] / 255.0, rtol=1e-6), (x, y)


def test_a_tree_too_large_for_lds():
    """6 000 spheres and boxes: the node array (over 144 KB) stays in global memory, so the walk kernels run their other
    variant by size and not by flag.  Closest hits against the oracle (same tree: same seed), and the rendered image
    the same bytes whichever way a path is traced."""
    rng = np.random.default_rng(12)
    objs = []
    for k in range(6000):
        c = rng.random(3) * 200.0
        if k % 3:
            objs.append(rt.Object(rt.Shape.Sphere(tuple(c), 1.0 + 2.0 * rng.random()), WHITE))
        else:
            e = 1.0 + 3.0 * rng.random(3)
            objs.append(rt.Object(rt.Shape.new_box(tuple(c - e), tuple(c + e)), WHITE))
    sc = rt.Scene(objs, (1, 1, 1), seed=5)
    osc = OracleScene(objs, (1, 1, 1), bvh_seed=5)
    o, d, t = rays_at_box(rng, (0, 0, 0), (200, 200, 200), 6000)
    ref = osc.intersect(o, d, t, seed=3)
    got = sc.intersect_batch(o, d, t, seed=3)
    nsame, _ = compare_hits(got, ref, "large tree", allow_flag_mismatch=1e-3, allow_prim_mismatch=1e-3, origins=o)
    assert nsame > 3000
    co = rt.CameraOptions(aperture=0.0, aspect_ratio=1.0, focus_distance=10.0, looking_at=(100, 100, 100), origin=(-150, 120, -180), shutter_close=1.0,
                          shutter_open=0.0, vfov=40.0, vup=(0, 1, 0))
    cam = rt.Camera(co)
    io = rt.ImageOptions(64, 64, 16, 12)
    _, base = sc.render(cam, io, rt.RenderOptions(seed=2), want_mean=True)
    for flags in (_abi.RT_RENDER_NO_TAIL, _abi.RT_RENDER_TAIL_ONLY, _abi.RT_RENDER_ONE_STREAM, _abi.RT_RENDER_ONE_KERNEL):   # (ONE_KERNEL is refused silently: the nodes are not in LDS)
        _, m = sc.render(cam, io, rt.RenderOptions(seed=2, flags=flags), want_mean=True)
        assert np.array_equal(m, base, equal_nan=True), flags
    mean_o = osc.render(co.to_c(), 64, 64, 16, 12, seed=2, threads=8)[1]
    assert rmse(np.clip(base, 0, 1), np.clip(mean_o, 0, 1)) < 0.02   # same Philox streams: the fp32 image stays next to the oracle's


def test_a_scene_of_nothing_but_an_inert_medium():
    """A medium bounded by a sphere can never be hit (shape.rs:386-410: both boundary queries find the same near root), so the
    library leaves it out of its tree: the tree is empty, every ray misses and every pixel is the background."""
    fog = rt.Object(rt.Shape.ConstantMedium(rt.Shape.Sphere((0, 0, 0), 5.0), 0.3), rt.Material.Isotropic(rt.Texture.SolidColor((1, 1, 1))))
    co = rt.CameraOptions(aperture=0.0, aspect_ratio=1.0, focus_distance=10.0, looking_at=(0, 0, 0), origin=(0, 0, -20), shutter_close=1.0, shutter_open=0.0,
                          vfov=40.0, vup=(0, 1, 0))
    for keep in (False, True):
        sc = rt.Scene([fog], (0.25, 0.5, 1.0), keep_inert=keep)
        h = sc.intersect_batch([[0, 0, -20]], [[0, 0, 1]], [0.0])
        assert h[0]["hit"] == 0
        rgb, mean = sc.render(rt.Camera(co), rt.ImageOptions(16, 16, 4, 5), rt.RenderOptions(seed=1), want_mean=True)
        assert np.allclose(mean, [0.25, 0.5, 1.0]) and (rgb.reshape(-1, 3) == [127, 181, 255]).all()   # sqrt(x) * 255.999 truncated (lib.rs:68-72)


def test_empty_and_ragged_batches():
    sc = rt.Scene([rt.Object(rt.Shape.Sphere((0, 0, 0), 1.0), WHITE)], (0, 0, 0))
    assert len(sc.intersect_batch(np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(0))) == 0
    for n in (1, 63, 64, 65, 257):
        o = np.tile([[-3.0, 0.0, 0.0]], (n, 1))
        d = np.tile([[1.0, 0.0, 0.0]], (n, 1))
        h = sc.intersect_batch(o, d, np.zeros(n))
        assert (h["hit"] == 1).all() and np.allclose(h["t"], 2.0)
    with pytest.raises(_abi.RtError):
        sc.intersect_batch(o, d, np.zeros(n), scope=5)


def test_scene_closest_hit_parity_along_paths(earth):
    """BvhNode::scatter_ (lib.rs:228-257): closest hit over the five example scenes for camera rays and
    their scattered successors, including the leaf-order dependent cases of final_scene."""
    rng = np.random.default_rng(21)
    for name in EXAMPLES:
        sc, co = _example(name, earth)
        osc, oco = OracleScene.example(name, 1, earth if name == "final_scene" else None)
        total = 0
        for o, d, t in path_rays(osc, oco, 64, 48, 5, 4000, 6, rng):
            ref = osc.intersect(o, d, t, seed=9)
            got = sc.intersect_batch(o, d, t, seed=9)
            nsame, _ = compare_hits(got, ref, name, allow_flag_mismatch=1e-4, allow_prim_mismatch=0.0, origins=o)
            total += nsame
        assert total > 4000, (name, total)


@pytest.mark.parametrize("name", EXAMPLES)
def test_scene_closest_hit_parity_on_a_large_sample(name, earth):
    """The same comparison on 40 000 camera rays per scene and five generations of their scattered successors (over 600 000
    rays in all): no ray may name another primitive than the oracle's, at most 1 in 100 000 may differ in hit / miss, and
    every record must hold the 1e-5 tolerance of scenes_util.TOL (t purely relative)."""
    e = earth if name == "final_scene" else None
    sc, co = _example(name, earth)
    osc, oco = OracleScene.example(name, 1, e)
    rays = flag = 0
    worst = {}
    for o, d, t in path_rays(osc, oco, 200, 150, 5, 40000, 5, np.random.default_rng(17)):
        ref = osc.intersect(o, d, t, seed=9)
        got = sc.intersect_batch(o, d, t, seed=9)
        rays += len(ref)
        flag += int((got["hit"] != ref["hit"]).sum())
        _, w = compare_hits(got, ref, name, allow_flag_mismatch=1e-3, allow_prim_mismatch=0.0, origins=o)   # (hit / miss: summed below)
        for k, v in w.items():
            worst[k] = max(worst.get(k, 0.0), v)
    print(f"\n{name}: {rays} rays, {flag} hit/miss mismatches, worst relative errors " + ", ".join(f"{k} {v:.1e}" for k, v in worst.items()))
    assert rays > 60000 and flag <= 1e-5 * rays, (name, rays, flag)


def test_material_scatter_parity(earth):
    """Material::scatter (materials.rs:28-113) with Texture::color_at (textures.rs:46-78) on the device, record by record:
    the kind of RayHit, the attenuation or emitted colour, and the scattered ray, for hits along the paths of the five
    examples.  Tolerances: colours 2e-5, ray origins and directions 1e-5 relative (2e-5 absolute on directions)."""
    lib = oracle()
    rng = np.random.default_rng(4)
    for name in EXAMPLES:
        sc, co = _example(name, earth)
        osc, oco = OracleScene.example(name, 1, earth if name == "final_scene" else None)
        desc = sc._host.rth_scene_desc(sc._h)
        kinds, compared = set(), 0
        for b, (o, d, t) in enumerate(path_rays(osc, oco, 40, 30, 6, 1500, 4, rng)):
            ho = osc.intersect(o, d, t, seed=6)
            hg = sc.intersect_batch(o, d, t, seed=6)
            idx = np.where((ho["hit"] == 1) & (hg["hit"] == 1) & (ho["prim"] == hg["prim"]))[0][:600]
            if len(idx) == 0:
                continue
            keys = np.stack([idx, np.full(len(idx), 2), np.full(len(idx), b)], axis=1).astype(np.uint32)
            got = sc.scatter_batch(hg[idx], seed=6, keys=keys)
            for j, k in enumerate(idx):
                hk = ho[k:k + 1].copy()
                s2, c2, o2, d2 = C.c_int32(), (C.c_double * 3)(), (C.c_double * 3)(), (C.c_double * 3)()
                lib.orc_scene_scatter(osc.h, hk.ctypes.data, 6, int(k), 2, b, C.byref(s2), c2, o2, d2)
                mat = desc.contents.materials[desc.contents.prims[int(hg[k]["prim"])].material].kind
                kinds.add(mat)
                if got[j]["scattered"] != s2.value:
                    # Metal: the sign of scattered . normal decides (materials.rs:52); fp32 may disagree within rounding of zero
                    assert mat == 1
                    continue
                compared += 1
                assert np.allclose(got[j]["color"], np.array(c2), rtol=2e-5, atol=2e-5, equal_nan=True), (name, mat)
                if s2.value:
                    assert np.allclose(got[j]["direction"], np.array(d2), rtol=1e-5, atol=2e-5), (name, mat)
                    assert np.allclose(got[j]["origin"], np.array(o2), rtol=1e-5, atol=1e-5), (name, mat)
        assert kinds and compared > 500, (name, kinds, compared)
    with pytest.raises(_abi.RtError):
        bad = np.zeros(1, dtype=hg.dtype); bad["prim"] = 10 ** 6
        sc.scatter_batch(bad)
    assert sc.scatter_batch(hg[:0]).shape == (0,)


def test_keeping_inert_primitives_changes_nothing(earth):
    rng = np.random.default_rng(2)
    a, co = _example("final_scene", earth)
    b, _ = _example("final_scene", earth, keep_inert=True)
    osc, oco = OracleScene.example("final_scene", 1, earth)
    for o, d, t in path_rays(osc, oco, 32, 32, 5, 3000, 3, rng):
        ha, hb = a.intersect_batch(o, d, t, seed=1), b.intersect_batch(o, d, t, seed=1)
        assert np.array_equal(ha.view(np.uint8), hb.view(np.uint8))


def test_camera_rays(earth):
    rng = np.random.default_rng(8)
    for name in EXAMPLES:
        sc, co = _example(name, earth)
        cam = rt.Camera(co)
        io = rt.ImageOptions(36, 64, 1, 50)
        n = 5000
        pi, pj, ss = (rng.integers(0, m, n).astype(np.uint32) for m in (64, 36, 1000))
        got = sc.camera_rays(cam, io, 987654321012, pi, pj, ss)
        o, d, t = oracle_camera_rays(co.to_c(), 64, 36, 987654321012, pi, pj, ss)
        assert np.array_equal(got[:, 0:3], o.astype(np.float32)) and np.array_equal(got[:, 6], t.astype(np.float32))
        assert np.abs(got[:, 3:6] - d).max() < 1e-6


@pytest.mark.parametrize("name,spp", [("random_spheres", 32), ("cornell", 64), ("perlin", 32), ("cornell_smoke", 64), ("final_scene", 16)])
def test_image_parity(name, spp, earth):
    sc, co = _example(name, earth)
    osc, oco = OracleScene.example(name, 1, earth if name == "final_scene" else None)
    cam = rt.Camera(co)
    w, h = (96, 54) if co.aspect_ratio > 1 else (64, 64)
    rgb, mean = sc.render(cam, rt.ImageOptions(h, w, spp, 50), rt.RenderOptions(seed=1), want_mean=True)
    a, amean, _, _ = osc.render(oco, w, h, spp, 50, seed=1, threads=16)
    b, _, _, _ = osc.render(oco, w, h, spp, 50, seed=2, threads=16)
    floor, err = rmse(a, b), rmse(rgb, a)
    assert not np.isnan(mean).any()
    assert err <= 0.5 * floor, (name, err, floor)
    assert abs(mean.mean() - amean.mean()) <= 0.02 * abs(amean.mean()) + 1e-3


def test_result_does_not_depend_on_passes_kernel_split_or_ranks(earth):
    # (cornell, cornell_smoke and perlin between them run every build of the one-kernel bounce: with and without side rounding,
    # with and without the queue key by long-test leaves, and the first bounce that makes its own camera rays)
    for name, spp, w, h in [("cornell_smoke", 24, 70, 50), ("final_scene", 12, 64, 64), ("random_spheres", 16, 80, 45), ("cornell", 16, 48, 48), ("perlin", 16, 64, 36)]:
        sc, co = _example(name, earth)
        cam = rt.Camera(co)
        io = rt.ImageOptions(h, w, spp, 50)
        _, base = sc.render(cam, io, rt.RenderOptions(seed=3), want_mean=True)
        for ro in (rt.RenderOptions(seed=3, samples_per_pass=5), rt.RenderOptions(seed=3, flags=_abi.RT_RENDER_NO_TAIL),
                   rt.RenderOptions(seed=3, flags=_abi.RT_RENDER_TAIL_ONLY), rt.RenderOptions(seed=3, flags=_abi.RT_RENDER_PROFILE),
                   rt.RenderOptions(seed=3, tile_w=8, tile_h=4), rt.RenderOptions(seed=3, flags=_abi.RT_RENDER_ONE_STREAM),
                   rt.RenderOptions(seed=3, flags=_abi.RT_RENDER_ONE_KERNEL), rt.RenderOptions(seed=3, flags=_abi.RT_RENDER_TWO_KERNELS),
                   rt.RenderOptions(seed=3, flags=_abi.RT_RENDER_ONE_KERNEL | _abi.RT_RENDER_NO_TAIL), rt.RenderOptions(seed=3, flags=_abi.RT_RENDER_TWO_KERNELS | _abi.RT_RENDER_NO_TAIL),
                   rt.RenderOptions(seed=3, samples_per_pass=5, flags=_abi.RT_RENDER_ONE_STREAM)):
            _, m = sc.render(cam, io, ro, want_mean=True)
            assert np.array_equal(m, base), (name, ro)
        for world in (2, 4, 8):
            parts = [sc.render(cam, io, rt.RenderOptions(seed=3, rank=r, world_size=world), want_mean=True)[1] for r in range(world)]
            assert np.array_equal(np.sum(parts, axis=0), base), (name, world)
            covered = np.sum([np.any(p != 0, axis=2) for p in parts], axis=0)
            assert covered.max() <= 1
        _, other = sc.render(cam, io, rt.RenderOptions(seed=4), want_mean=True)
        assert not np.array_equal(other, base)


def test_two_bounces_in_the_first_launch_change_no_byte(earth):
    """Scenes of a few leaves without long-test leaves (perlin) run bounces 0 and 1 of a pass in one launch once the schedule is
    known (rt_profile.two_bounce_passes); passes that read queue lengths back never do.  Same bytes either way, also when the
    second bounce is the last one allowed (max_reflections 2) and when there is only one (1)."""
    sc, co = _example("perlin", earth)
    cam = rt.Camera(co)
    for depth, expect in ((50, 3), (3, 3), (2, 3), (1, 0)):
        io = rt.ImageOptions(144, 256, 64, depth)
        _, got = sc.render(cam, io, rt.RenderOptions(seed=9, samples_per_pass=16), want_mean=True)   # 4 passes of 590 k paths: the first learns the schedule
        prof = sc.profile()
        assert prof["passes"] == 4 and prof["two_bounce_passes"] == expect, (depth, prof)
        _, ref = sc.render(cam, io, rt.RenderOptions(seed=9, samples_per_pass=16, flags=_abi.RT_RENDER_NO_TAIL), want_mean=True)
        assert sc.profile()["two_bounce_passes"] == 0
        assert np.array_equal(got, ref) and ref.any(), depth
    # the queue of the Cornell rooms is ordered by long-test leaves: one bounce per launch
    sc, co = _example("cornell", earth)
    sc.render(rt.Camera(co), rt.ImageOptions(144, 256, 64, 50), rt.RenderOptions(seed=9, samples_per_pass=16))
    assert sc.profile()["passes"] == 4 and sc.profile()["two_bounce_passes"] == 0


def test_pixels_entry_point_matches_rt_render(earth):
    """`pixels(&camera, &scene, image_options)` through the C++ host entry: same bytes as rt_render, row-major, top row first."""
    sc, co = _example("cornell", earth)
    cam = rt.Camera(co)
    io = rt.ImageOptions(40, 56, 16, 50)
    buf = rt.pixels(cam, sc, io, seed=1)
    rgb = sc.render(cam, io, rt.RenderOptions(seed=1))
    assert len(buf) == 3 * 40 * 56 and buf == rgb.tobytes()
    # the light is in the ceiling: the top rows are brighter than the bottom rows
    img = np.frombuffer(buf, np.uint8).reshape(40, 56, 3).astype(float)
    assert img[:10].mean() > img[-10:].mean()


def test_render_argument_errors(earth):
    sc, co = _example("cornell", earth)
    cam = rt.Camera(co)
    for io in (rt.ImageOptions(1, 8, 4, 50), rt.ImageOptions(8, 1, 4, 50), rt.ImageOptions(8, 8, 0, 50)):
        with pytest.raises(_abi.RtError) as e:
            sc.render(cam, io)
        assert e.value.status == _abi.RT_ERR_INVALID_ARGUMENT
    with pytest.raises(_abi.RtError):
        sc.render(cam, rt.ImageOptions(8, 8, 4, 50), rt.RenderOptions(rank=2, world_size=2))
    black = sc.render(cam, rt.ImageOptions(8, 8, 4, 0))          # max_reflections = 0: lib.rs:103 returns black
    assert not black.any()


def test_full_size_properties(earth):
    """BASELINE-size frame (800x800 final_scene) at low spp: properties that do not need the oracle."""
    sc, co = _example("final_scene", earth)
    cam = rt.Camera(co)
    io = rt.ImageOptions(800, 800, 4, 50)
    rgb, mean = sc.render(cam, io, rt.RenderOptions(seed=1, flags=_abi.RT_RENDER_PROFILE), want_mean=True)
    prof = sc.profile()
    assert prof["samples"] == 800 * 800 * 4 and prof["segments"] >= prof["samples"]
    assert np.isfinite(mean).all()   # (means may be negative: the noise texture returns values in [-1, 1], textures.rs:57)
    assert np.array_equal(rgb, np.minimum(np.floor(np.sqrt(np.clip(mean, 0, 1)) * 255.999), 255).astype(np.uint8))
    halves = [sc.render(cam, io, rt.RenderOptions(seed=1, rank=r, world_size=2), want_mean=True)[1] for r in range(2)]
    assert np.array_equal(halves[0] + halves[1], mean)
    # more samples of the same streams: the 8-spp mean is the average of two 4-spp halves only if sample s keys its own stream
    _, m8 = sc.render(cam, rt.ImageOptions(800, 800, 8, 50), rt.RenderOptions(seed=1), want_mean=True)
    assert abs(m8.mean() - mean.mean()) < 0.05 * mean.mean()


@pytest.mark.parametrize("name,w,h", [("cornell", 600, 600), ("perlin", 1920, 1080), ("cornell_smoke", 800, 800)])
def test_full_size_properties_of_the_other_configurations(name, w, h):
    """The remaining GPU configurations at their full frame size (examples/cornell.rs, perlin.rs, cornell_smoke.rs at the sizes
    the benchmark quotes) and 4 spp: properties that hold at any size.  perlin at 1920 x 1080 is the largest pixel count
    (2 073 600 owned pixels, 32-sample passes)."""
    sc, co = rt.Scene.example(name, 1)
    cam = rt.Camera(co)
    io = rt.ImageOptions(h, w, 4, 50)
    rgb, mean = sc.render(cam, io, rt.RenderOptions(seed=1, flags=_abi.RT_RENDER_PROFILE), want_mean=True)
    prof = sc.profile()
    assert prof["samples"] == w * h * 4 and prof["segments"] >= prof["samples"]
    assert np.isfinite(mean).all() and mean.any()
    assert np.array_equal(rgb, np.minimum(np.floor(np.sqrt(np.clip(mean, 0, 1)) * 255.999), 255).astype(np.uint8))    # lib.rs:68-72
    # the frame does not depend on the schedule: unprofiled (cached calibration, two streams where they pay), one stream, short passes
    # (these scenes have a handful of leaves: by default a bounce is one kernel that scans them; RT_RENDER_TWO_KERNELS is the traversal / shading pair)
    for ro in (rt.RenderOptions(seed=1), rt.RenderOptions(seed=1, flags=_abi.RT_RENDER_ONE_STREAM), rt.RenderOptions(seed=1, samples_per_pass=3),
               rt.RenderOptions(seed=1, flags=_abi.RT_RENDER_TWO_KERNELS)):
        _, m = sc.render(cam, io, ro, want_mean=True)
        assert np.array_equal(m, mean), ro
    thirds = [sc.render(cam, io, rt.RenderOptions(seed=1, rank=r, world_size=3), want_mean=True)[1] for r in range(3)]
    assert np.array_equal(thirds[0] + thirds[1] + thirds[2], mean)
    # sample s keys its own stream: 8 spp is 4 spp plus four more samples per pixel, so the frame means stay close
    _, m8 = sc.render(cam, rt.ImageOptions(h, w, 8, 50), rt.RenderOptions(seed=1), want_mean=True)
    assert abs(m8.mean() - mean.mean()) < 0.05 * abs(mean.mean())


def test_quantise_device_matches_reference_rule():
    import torch
    sc = rt.Scene([rt.Object(rt.Shape.Sphere((0, 0, 0), 1.0), WHITE)], (0, 0, 0))
    vals = torch.tensor([0.25, 1.0, 4.0, -0.5, float("nan"), 0.0, 1e-9, 0.9999999], dtype=torch.float64, device="cuda")
    out = torch.zeros(8, dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())   # the fill of `out` is queued on the current stream
    sc.quantise_device(vals.data_ptr(), 8, out.data_ptr(), side.cuda_stream)
    with pytest.raises(ValueError):                 # handle 0 would mean the library's own, unordered stream
        sc.quantise_device(vals.data_ptr(), 8, out.data_ptr(), 0)
    torch.cuda.synchronize()
    assert out.cpu().tolist() == [127, 255, 255, 0, 0, 0, 0, 255]


def test_rank_frames_are_ordered_on_torchs_stream(earth):
    """The multi-GPU path's stream order (raytracing_amd/distributed.py): each rank's frame is rendered on a side stream
    and handed to torch's current stream by an event, so a consumer queued there -- the RCCL reduce in a real run, a
    tensor sum here -- reads finished frames without any device-wide synchronisation.  Two ranks' frames rendered back to
    back (same scene handle, same workspace) and summed at once must be the single-rank frame, byte for byte."""
    import torch
    from raytracing_amd.distributed import quantise, render_local
    sc, co = _example("final_scene", earth)
    cam = rt.Camera(co)
    io = rt.ImageOptions(200, 200, 48, 50)
    with pytest.raises(ValueError):
        rt.RenderOptions(stream=torch.cuda.current_stream().cuda_stream).to_c()   # handle 0 cannot be named
    for _ in range(2):
        parts = [render_local(sc, cam, io, seed=7, rank=r, world_size=2) for r in range(2)]
        total = parts[0] + parts[1]                        # queued on torch's stream right behind the renders
        rgb = quantise(sc, total)
        busy = torch.randn(2048, 2048, device="cuda") @ torch.randn(2048, 2048, device="cuda")   # the current stream stays busy
        full = render_local(sc, cam, io, seed=7)
        assert torch.equal(total, full) and busy.shape[0] == 2048
        ref = sc.render(cam, io, rt.RenderOptions(seed=7))
        assert np.array_equal(rgb.cpu().numpy(), ref)


def test_pixels_split_over_device_contexts(earth):
    """RenderOptions::devices of the C++ host API (what `rt_example --gpus N` uses): the frame split over three device
    contexts (all on GPU 0 here), means added and quantised on the host -- the bytes of the single-device render."""
    sc, co = _example("final_scene", earth)
    cam = rt.Camera(co)
    io = rt.ImageOptions(50, 70, 6, 50)
    assert rt.pixels(cam, sc, io, seed=5, devices=[0, 0, 0]) == rt.pixels(cam, sc, io, seed=5)
    with pytest.raises(RuntimeError):
        rt.pixels(cam, sc, io, seed=5, devices=[0, 999])


def test_noise_kinds_beyond_perlin():
    """Noise textures that name an Fbm or a Turbulence (rt_abi.h RT_NOISE_*; an extension, no reference example uses them):
    shading records along the paths of a scene that wears them, and its image with shared random streams, against the oracle."""
    from scenes_util import noise_kinds_scene
    objects, background, co_kw = noise_kinds_scene()
    sc, osc = rt.Scene(objects, background), OracleScene(objects, background)
    co = rt.CameraOptions(**co_kw)
    oco = co.to_c()
    lib = oracle()
    rng = np.random.default_rng(12)
    compared, seen = 0, set()
    for b, (o, d, t) in enumerate(path_rays(osc, oco, 48, 32, 6, 1500, 4, rng)):
        ho = osc.intersect(o, d, t, seed=6)
        hg = sc.intersect_batch(o, d, t, seed=6)
        compare_hits(hg, ho, "noise kinds scene", allow_flag_mismatch=1e-3, origins=o)
        idx = np.where((ho["hit"] == 1) & (hg["hit"] == 1) & (ho["prim"] == hg["prim"]))[0][:500]
        keys = np.stack([idx, np.full(len(idx), 2), np.full(len(idx), b)], axis=1).astype(np.uint32)
        got = sc.scatter_batch(hg[idx], seed=6, keys=keys)
        for j, k in enumerate(idx):
            hk = ho[k:k + 1].copy()
            s2, c2, o2, d2 = C.c_int32(), (C.c_double * 3)(), (C.c_double * 3)(), (C.c_double * 3)()
            lib.orc_scene_scatter(osc.h, hk.ctypes.data, 6, int(k), 2, b, C.byref(s2), c2, o2, d2)
            assert got[j]["scattered"] == s2.value
            # (the texture is looked up at the fp32 hit point: the colour carries the point's rounding times the noise's slope)
            assert np.allclose(got[j]["color"], np.array(c2), rtol=1e-4, atol=1e-4, equal_nan=True), int(hg[k]["prim"])
            seen.add(int(hg[k]["prim"]))
            compared += 1
    assert seen == {0, 1, 2, 3} and compared > 800
    cam = rt.Camera(co)
    w, h, spp = 96, 64, 32
    rgb, mean = sc.render(cam, rt.ImageOptions(h, w, spp, 50), rt.RenderOptions(seed=1), want_mean=True)
    a, amean, _, _ = osc.render(oco, w, h, spp, 50, seed=1, threads=16)
    bb, _, _, _ = osc.render(oco, w, h, spp, 50, seed=2, threads=16)
    floor, err = rmse(a, bb), rmse(rgb, a)
    assert not np.isnan(mean).any()
    assert err <= 0.5 * floor, (err, floor)
    assert abs(mean.mean() - amean.mean()) <= 0.02 * abs(amean.mean()) + 1e-3
    # a description that asks for more tables than it brings is refused
    bad = rt.Scene([rt.Object(rt.Shape.Sphere((0, 0, 0), 1.0), rt.Material.Lambertian(rt.Texture.Fbm(1.0, octaves=4)))], (0, 0, 0))
    bad.desc.n_perm_tables = 3
    with pytest.raises(_abi.RtError):
        bad.device()


def test_multi_scatter_media_variant():
    """Media flagged RT_PRIM_MULTI_SCATTER (NOT the reference's behaviour): closest hits along the scene's paths (many of which
    start inside the smoke) and the image with shared random streams, against the oracle built with the same switch; the same
    scene with the reference's media traces fewer segments (its paths scatter once per entry and leave)."""
    from scenes_util import multi_scatter_scene
    objects, background, co_kw = multi_scatter_scene()
    sc, osc = rt.Scene(objects, background), OracleScene(objects, background)
    assert sum(1 for i in range(sc.desc.n_prims) if sc.desc.prims[i].flags & _abi.RT_PRIM_MULTI_SCATTER) == 2
    co = rt.CameraOptions(**co_kw)
    oco = co.to_c()
    rng = np.random.default_rng(21)
    inside_hits = 0
    for b, (o, d, t) in enumerate(path_rays(osc, oco, 48, 48, 6, 2000, 6, rng)):
        ho = osc.intersect(o, d, t, seed=6)
        hg = sc.intersect_batch(o, d, t, seed=6)
        compare_hits(hg, ho, "multi-scatter scene", allow_flag_mismatch=1e-3, origins=o)
        if b > 0:   # rays that start on a scattering point inside a medium (normal (1, 0, 0), shape.rs:413) and scatter there again
            inside_hits += int(((ho["hit"] == 1) & (ho["prim"] >= 6)).sum())
    assert inside_hits > 200
    cam = rt.Camera(co)
    w, h, spp = 80, 80, 48
    rgb, mean = sc.render(cam, rt.ImageOptions(h, w, spp, 50), rt.RenderOptions(seed=1), want_mean=True)
    a, amean, _, _ = osc.render(oco, w, h, spp, 50, seed=1, threads=16)
    b2, _, _, _ = osc.render(oco, w, h, spp, 50, seed=2, threads=16)
    floor = rmse(a, b2)
    assert rmse(rgb, a) <= 0.5 * floor, (rmse(rgb, a), floor)
    # the reference's media in the same room: another picture
    plain = [rt.Object(rt.Shape.ConstantMedium(o.shape.kw["inner"], o.shape.kw["density"]), o.material) if o.shape.kind == "constant_medium" else o for o in objects]
    sc1 = rt.Scene(plain, background)
    segments = []
    for scene in (sc, sc1):
        scene.render(cam, rt.ImageOptions(h, w, spp, 50), rt.RenderOptions(seed=1, flags=_abi.RT_RENDER_PROFILE))
        segments.append(scene.profile()["segments"])
    assert segments[0] > 1.1 * segments[1], segments      # paths go on scattering inside the smoke instead of once per entry


def test_far_root_and_facing_normal_variants():
    """A scene made of the switches that are NOT the reference's behaviour (RT_PRIM_BOTH_ROOTS, RT_MATF_FACE_NORMAL): closest hits
    and shading records along its paths and its image with shared random streams against the oracle built with the same switches;
    and the same scene without them differs, as it must (rays get out of the glass ball, the ball of smoke scatters)."""
    from scenes_util import far_root_variant_scene
    objects, background, co_kw = far_root_variant_scene()
    sc, osc = rt.Scene(objects, background), OracleScene(objects, background)
    co = rt.CameraOptions(**co_kw)
    oco = co.to_c()
    lib = oracle()
    rng = np.random.default_rng(14)
    compared, seen = 0, set()
    for b, (o, d, t) in enumerate(path_rays(osc, oco, 48, 32, 6, 2000, 5, rng)):
        ho = osc.intersect(o, d, t, seed=6)
        hg = sc.intersect_batch(o, d, t, seed=6)
        compare_hits(hg, ho, "far-root scene", allow_flag_mismatch=1e-3, origins=o)
        idx = np.where((ho["hit"] == 1) & (hg["hit"] == 1) & (ho["prim"] == hg["prim"]))[0][:500]
        keys = np.stack([idx, np.full(len(idx), 2), np.full(len(idx), b)], axis=1).astype(np.uint32)
        got = sc.scatter_batch(hg[idx], seed=6, keys=keys)
        for j, k in enumerate(idx):
            hk = ho[k:k + 1].copy()
            s2, c2, o2, d2 = C.c_int32(), (C.c_double * 3)(), (C.c_double * 3)(), (C.c_double * 3)()
            lib.orc_scene_scatter(osc.h, hk.ctypes.data, 6, int(k), 2, b, C.byref(s2), c2, o2, d2)
            prim = int(hg[k]["prim"])
            if got[j]["scattered"] != s2.value:
                assert prim == 4           # Metal: the sign of scattered . normal decides (materials.rs:52); fp32 may disagree within rounding of zero
                continue
            assert np.allclose(got[j]["color"], np.array(c2), rtol=2e-5, atol=2e-5, equal_nan=True), prim
            if s2.value and prim != 1:     # (glass near the critical angle amplifies the fp32 rounding of the normal: checked through the image)
                assert np.allclose(got[j]["direction"], np.array(d2), rtol=1e-5, atol=3e-5), prim
            seen.add(prim)
            compared += 1
    assert seen == {0, 1, 2, 3, 4} and compared > 1000
    cam = rt.Camera(co)
    w, h, spp = 96, 64, 32
    rgb, mean = sc.render(cam, rt.ImageOptions(h, w, spp, 50), rt.RenderOptions(seed=1), want_mean=True)
    a, amean, _, _ = osc.render(oco, w, h, spp, 50, seed=1, threads=16)
    bb, _, _, _ = osc.render(oco, w, h, spp, 50, seed=2, threads=16)
    floor, err = rmse(a, bb), rmse(rgb, a)
    assert not np.isnan(mean).any()
    assert err <= 0.5 * floor, (err, floor)
    assert abs(mean.mean() - amean.mean()) <= 0.02 * abs(amean.mean()) + 1e-3
    # the reference's own behaviour on the same geometry is another image
    for ob in objects:
        ob.shape = rt.Shape.ConstantMedium(rt.Shape.Sphere(ob.shape.kw["inner"].kw["center"], 1.0), 1.5) if ob.shape.kind == "constant_medium" else rt.Shape.Sphere(ob.shape.kw["center"], ob.shape.kw["radius"])
        ob.material = rt.Material(ob.material.kind, ob.material.texture, ob.material.albedo, ob.material.param)
    _, ref_mean = rt.Scene(objects, background).render(cam, rt.ImageOptions(h, w, spp, 50), rt.RenderOptions(seed=1), want_mean=True)
    assert rmse(np.clip(ref_mean, 0, 1) * 255, np.clip(mean, 0, 1) * 255) > 5.0   # grey levels, on the linear means


def test_final_scene_book_variant_renders(earth):
    """final_scene_book (not a reference example; its switches are tested against the oracle above): renders, finite, the same
    bytes whatever the kernel split, and its two balls of smoke, inert in the reference's scene, are hit."""
    sc, co = rt.Scene.example("final_scene_book", 1, earth)
    cam = rt.Camera(co)
    io = rt.ImageOptions(64, 64, 8, 50)
    _, base = sc.render(cam, io, rt.RenderOptions(seed=2), want_mean=True)
    assert np.isfinite(base).all() and base.mean() > 0.01
    for flags in (_abi.RT_RENDER_NO_TAIL, _abi.RT_RENDER_TAIL_ONLY, _abi.RT_RENDER_NODES_IN_GLOBAL):
        assert np.array_equal(sc.render(cam, io, rt.RenderOptions(seed=2, flags=flags), want_mean=True)[1], base)
    rng = np.random.default_rng(5)
    rays = sc.camera_rays(cam, io, 2, rng.integers(0, 64, 4000), rng.integers(0, 64, 4000), rng.integers(0, 8, 4000))
    hits = sc.intersect_batch(rays[:, 0:3], rays[:, 3:6], rays[:, 6], seed=3)
    d = sc.desc
    media = {i for i in range(d.n_prims) if d.prims[i].flags & _abi.RT_PRIM_MEDIUM}
    assert len(media) == 2 and media & set(int(p) for p in hits["prim"][hits["hit"] == 1])


def test_tree_regrouped_by_sampled_rays_keeps_every_byte(earth):
    """The first sizeable pass on a scene handle lends rays to the library, which regroups the inner nodes of its copy of the
    tree by them (rt_profile.tree_tests_*): fewer box tests, the same leaves in the same order, so the same hit records and the
    same frame, bit for bit; RT_RENDER_KEEP_TREE holds it off."""
    sc, co = _example("final_scene", earth)
    cam = rt.Camera(co)
    io = rt.ImageOptions(400, 400, 8, 50)
    rng = np.random.default_rng(9)
    rays = sc.camera_rays(cam, io, 1, rng.integers(0, 400, 20000), rng.integers(0, 400, 20000), rng.integers(0, 8, 20000))
    hits0 = sc.intersect_batch(rays[:, 0:3], rays[:, 3:6], rays[:, 6], seed=3)
    _, kept = sc.render(cam, io, rt.RenderOptions(seed=4, flags=_abi.RT_RENDER_KEEP_TREE), want_mean=True)
    p = sc.profile()
    assert p["tree_tests_by_area"] == 0.0 and p["tree_tests_sampled"] == 0.0
    _, small = sc.render(cam, rt.ImageOptions(64, 64, 8, 50), rt.RenderOptions(seed=4), want_mean=True)   # too small a pass to lend rays
    assert sc.profile()["tree_tests_sampled"] == 0.0
    _, tuned = sc.render(cam, io, rt.RenderOptions(seed=4), want_mean=True)
    p = sc.profile()
    assert 0.0 < p["tree_tests_sampled"] < 0.9 * p["tree_tests_by_area"], p
    assert np.array_equal(kept, tuned)
    _, after = sc.render(cam, io, rt.RenderOptions(seed=4), want_mean=True)
    assert np.array_equal(kept, after) and sc.profile()["tree_tests_sampled"] == p["tree_tests_sampled"]          # regrouped once
    for flags in (_abi.RT_RENDER_NODES_IN_GLOBAL, _abi.RT_RENDER_TAIL_ONLY, _abi.RT_RENDER_ONE_STREAM):
        assert np.array_equal(sc.render(cam, io, rt.RenderOptions(seed=4, flags=flags), want_mean=True)[1], kept)
    hits1 = sc.intersect_batch(rays[:, 0:3], rays[:, 3:6], rays[:, 6], seed=3)
    assert hits0.tobytes() == hits1.tobytes()
