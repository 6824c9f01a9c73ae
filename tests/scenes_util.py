"""Shared helpers for the parity tests: shape zoo, ray generators, hit-record comparison."""
import numpy as np

import raytracing_amd as rt

EXAMPLES = ["random_spheres", "cornell", "perlin", "cornell_smoke", "final_scene"]
# Hit-record tolerance.  t: purely relative, |got - oracle| <= TOL * |oracle|.  normal, in_vec (unit vectors) and u, v
# (in [0, 1]): |got - oracle| <= TOL * max(1, |oracle|), i.e. relative to their range.  The point is origin + t * direction
# with t held in fp32, so its error scales with the ray, not with the coordinate it lands on:
# |got - oracle| <= TOL * max(1, |oracle point|_inf, |ray origin|_inf).
TOL = 1e-5


def shape_zoo():
    """(name, Shape) pairs covering every Shape variant and wrapper nesting (shape.rs:62-106)."""
    S = rt.Shape
    box = S.new_box((0, 0, 0), (165, 330, 165))
    return [
        ("sphere", S.Sphere((1.0, 2.0, -3.0), 2.5)),
        ("sphere_big", S.Sphere((0.0, -1000.0, 0.0), 1000.0)),
        ("moving_sphere", S.sphere(0.7, (0.0, (0.0, 0.5, 0.0)), (1.0, (0.3, 1.1, -0.2)))),
        ("rect_x", S.rectangle((2.0, -1.0, -2.0), (2.0, 3.0, 4.0))),
        ("rect_y", S.rectangle((-3.0, 0.5, -1.0), (4.0, 0.5, 2.0))),
        ("rect_z", S.rectangle((-1.0, -2.0, 1.5), (3.0, 2.0, 1.5))),
        ("rect_flipped", S.rectangle((0.0, 0.0, 5.0), (5.0, 5.0, 5.0)).flipped()),
        ("box_cube", S.new_box((-1, -1, -1), (2, 2, 2))),
        ("box_slab", S.new_box((0, 0, -100), (100, 41.25, 0))),  # x and z ranges differ: y sides carry swapped extents
        ("box_rot_y_trans", box.rotated(rt.AXIS_Y, 15.0).translated((265, 0, 295))),
        ("box_rot_x", S.new_box((0, 0, 0), (3, 2, 1)).rotated(rt.AXIS_X, -33.0)),
        ("box_rot_z_rot_y", S.new_box((-1, 0, 0), (2, 2, 3)).rotated(rt.AXIS_Z, 70.0).rotated(rt.AXIS_Y, 10.0).translated((0.5, -0.5, 2))),
        ("sphere_rot_y", S.Sphere((0, 0, 0), 2.0).rotate(rt.AXIS_Y, 45.0)),
        ("sphere_trans_flipped", S.Sphere((1, 1, 1), 1.5).translated((-2, 0, 3)).flipped()),
        ("medium_box", S.ConstantMedium(S.new_box((0, 0, 0), (165, 165, 165)).rotated(rt.AXIS_Y, -18.0).translated((130, 0, 65)), 0.01)),
        ("medium_cube", S.ConstantMedium(S.new_box((-2, -2, -2), (2, 2, 2)), 0.4)),
        ("medium_sphere", S.ConstantMedium(S.Sphere((0, 0, 0), 3.0), 0.5)),  # never hit (near root twice)
        ("medium_wrapped", S.ConstantMedium(S.new_box((-1, -1, -1), (1, 1, 1)).rotated(rt.AXIS_Z, 30.0), 0.8).rotated(rt.AXIS_X, 20.0).translated((1, 2, 3)).flipped()),
        # the far-root variant (RT_PRIM_BOTH_ROOTS; not the reference's behaviour): hit from within as well, and a medium inside one scatters
        ("sphere_both_roots", S.Sphere((1.0, 2.0, -3.0), 2.5, both_roots=True)),
        ("sphere_both_roots_rot_y_trans", S.Sphere((0, 0, 0), 2.0, both_roots=True).rotate(rt.AXIS_Y, 45.0).translated((1, -1, 2))),
        ("medium_sphere_both_roots", S.ConstantMedium(S.Sphere((0, 0, 0), 3.0, both_roots=True), 0.5)),
        # media that scatter again from within (RT_PRIM_MULTI_SCATTER; not the reference's behaviour either)
        ("medium_cube_multi", S.ConstantMedium(S.new_box((-2, -2, -2), (2, 2, 2)), 0.4, multi_scatter=True)),
        ("medium_wrapped_multi", S.ConstantMedium(S.new_box((-1, -1, -1), (1, 1, 1)).rotated(rt.AXIS_Z, 30.0), 0.8, multi_scatter=True).rotated(rt.AXIS_X, 20.0).translated((1, 2, 3))),
    ]


def bound_of(shape):
    from oracle_lib import OracleShape
    return OracleShape(shape).bound()


def rays_at_box(rng, mn, mx, n, spread=1.5):
    """n rays with origins around (and partly inside) the box [mn, mx], aimed at random points in it."""
    mn, mx = np.asarray(mn, float), np.asarray(mx, float)
    mn, mx = np.maximum(mn, -5000.0), np.minimum(mx, 5000.0)
    c, e = 0.5 * (mn + mx), np.maximum(0.5 * (mx - mn), 0.5)
    o = c + (rng.random((n, 3)) * 2 - 1) * e * spread * 2
    inside = rng.random(n) < 0.25
    o[inside] = (c + (rng.random((n, 3)) * 2 - 1) * e)[inside]
    target = c + (rng.random((n, 3)) * 2 - 1) * e * 1.1
    d = target - o
    d *= (0.2 + 2.0 * rng.random((n, 1))) / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-9)
    # a share of axis-parallel rays: zeros in the direction exercise the inf / NaN paths
    ax = rng.random(n) < 0.05
    k = rng.integers(0, 3, n)
    for i in np.where(ax)[0]:
        d[i, k[i]] = 0.0
    t = rng.random(n)
    return o.astype(np.float32), d.astype(np.float32), t.astype(np.float32)


def degenerate_rays(mn, mx):
    """Rays that exercise the infinities, NaNs and ties of the reference's arithmetic (shape.rs:321-329, :580-599,
    geometry.rs:340-346): origins on a 5 x 5 x 5 grid that includes the exact face coordinates of the box [mn, mx], its
    centre and points outside; directions along the axes (zeros and negative zeros in the other components), in the
    coordinate planes, the zero vector and one with a NaN."""
    mn, mx = np.maximum(np.asarray(mn, float), -5000.0), np.minimum(np.asarray(mx, float), 5000.0)
    c = 0.5 * (mn + mx)
    axes = [np.array([mn[k] - 1.0, mn[k], c[k], mx[k], mx[k] + 1.0]) for k in range(3)]
    o = np.array([[x, y, z] for x in axes[0] for y in axes[1] for z in axes[2]])
    nz = -0.0
    dirs = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [1, 0, 1], [0, 1, -1], [-1, 1, 0],
                     [nz, 1, 0.5], [1, nz, nz], [nz, nz, -1], [0, 0, 0], [np.nan, 1, 0]], dtype=float)
    oo = np.repeat(o, len(dirs), axis=0)
    dd = np.tile(dirs, (len(o), 1))
    return oo.astype(np.float32), dd.astype(np.float32), np.full(len(oo), 0.5, np.float32)


def compare_hits(got, ref, what="", tol=TOL, allow_flag_mismatch=0.0, allow_prim_mismatch=0.0, origins=None, allow_record_mismatch=0.0):
    """Asserts hit records `got` (rt_hit, fp32) agree with the oracle's `ref` (f64).
    allow_record_mismatch: share of the records that may differ beyond `tol` in some field (for crafted rays that
    graze edges: which side of a box an edge hit names, and what a medium makes of it, hangs on rounding)."""
    n = len(ref)
    # An accepted NaN t (a side whose plane holds the origin of a ray parallel to it: 0 / 0 passes every comparison,
    # shape.rs:321-329) against a number counts as a hit / miss mismatch: which of the two a grazing ray ends with
    # hangs on the rounding of the sides tried after it.
    nan_mis = (got["hit"] == 1) & (ref["hit"] == 1) & (np.isnan(got["t"]) != np.isnan(ref["t"]))
    flag_mis = (got["hit"] != ref["hit"]) | nan_mis
    assert flag_mis.sum() <= allow_flag_mismatch * n, f"{what}: {int(flag_mis.sum())}/{n} hit/miss mismatches, first at {np.where(flag_mis)[0][:5]}"
    both = (got["hit"] == 1) & (ref["hit"] == 1) & ~nan_mis
    prim_mis = both & (got["prim"] != ref["prim"])
    assert prim_mis.sum() <= allow_prim_mismatch * n, f"{what}: {int(prim_mis.sum())}/{n} primitive mismatches, first at {np.where(prim_mis)[0][:5]}"
    same = both & ~prim_mis
    idx = np.where(same)[0]
    errs = {}
    for f in ("t", "point", "normal", "in_vec", "u", "v"):
        a, b = got[f][same].astype(np.float64), ref[f][same]
        scale = np.maximum(1.0, np.abs(b))
        if f == "t":
            scale = np.maximum(np.abs(b), 1e-300)
        if f == "point":   # relative to the largest finite coordinate (a point may have NaN or infinite components on both sides)
            fin = lambda x: np.where(np.isfinite(x), np.abs(x), 0.0).max(axis=1, keepdims=True)
            scale = np.maximum(1.0, fin(b))
            if origins is not None:
                scale = np.maximum(scale, fin(np.asarray(origins, np.float64)[same]))
        with np.errstate(invalid="ignore"):
            err = np.abs(a - b) / scale
        err[(np.isnan(a) & np.isnan(b)) | (a == b)] = 0.0   # NaN on both sides, or the same infinity
        if f == "u":   # at a sphere's pole (v = 0 or 1) the longitude is the angle of a vector of rounding errors
            v_ref = ref["v"][same]
            err[(np.abs(v_ref) < 1e-6) | (np.abs(v_ref - 1.0) < 1e-6)] = 0.0
        err = np.where(np.isnan(err), np.inf, err)   # NaN on one side only
        errs[f] = err.reshape(len(idx), -1).max(axis=1) if len(idx) else np.zeros(0)
    off = np.zeros(len(idx), bool)
    for f in errs:
        off |= errs[f] > tol
    if off.sum() > allow_record_mismatch * n:
        f = max(errs, key=lambda k: errs[k].max())
        k = int(np.argmax(errs[f]))
        raise AssertionError(f"{what}: {int(off.sum())}/{n} records off; worst: {f} off by {errs[f][k]:.3e} (> {tol}) at ray {idx[k]}")
    worst = {f: float(errs[f][~off].max()) if (~off).any() else 0.0 for f in errs}
    return int(len(idx) - off.sum()), worst


def path_rays(oscene, oco, width, height, seed, n, bounces, rng):
    """Rays as the renderer meets them: camera rays and their scattered successors for a few
    bounces, produced by the oracle (so both sides of a parity test see the same fp32 rays)."""
    import ctypes as C
    from oracle_lib import oracle, oracle_camera_rays
    lib = oracle()
    pi, pj, ss = rng.integers(0, width, n), rng.integers(0, height, n), rng.integers(0, 64, n)
    o, d, t = oracle_camera_rays(oco, width, height, seed, pi, pj, ss)
    o, d, t = o.astype(np.float32), d.astype(np.float32), t.astype(np.float32)
    out = [(o, d, t)]
    ids = np.arange(n)
    for b in range(bounces):
        h = oscene.intersect(o, d, t, seed=seed)
        idx = np.where(h["hit"] == 1)[0]
        no, nd, keep = np.zeros((len(idx), 3), np.float32), np.zeros((len(idx), 3), np.float32), []
        for j, k in enumerate(idx):
            hk = h[k:k + 1].copy()
            sc, col, ro, rd = C.c_int32(), (C.c_double * 3)(), (C.c_double * 3)(), (C.c_double * 3)()
            lib.orc_scene_scatter(oscene.h, hk.ctypes.data, seed, int(ids[k]), 0, b, C.byref(sc), col, ro, rd)
            if sc.value and np.isfinite(np.array(rd)).all():
                keep.append(j)
                no[j], nd[j] = ro, rd
        keep = np.array(keep, dtype=int)
        if len(keep) == 0:
            break
        o, d, t, ids = no[keep], nd[keep], t[idx][keep], ids[idx][keep]
        out.append((o, d, t))
    return out


def noise_kinds_scene():
    """Three spheres that wear the noise functions beyond Perlin (an Fbm, a Turbulence, an Fbm under a checker) and a Perlin floor,
    lit by the background: `Texture::Noise` takes any `NoiseFn` (textures.rs:14-17); the reference's examples only use Perlin."""
    import raytracing_amd as rt
    T, M, S = rt.Texture, rt.Material, rt.Shape
    objects = [
        rt.Object(S.Sphere((0.0, -100.0, 0.0), 100.0), M.Lambertian(T.Noise(1.0))),
        rt.Object(S.Sphere((-2.2, 1.0, 0.0), 1.0), M.Lambertian(T.Fbm(2.0))),
        rt.Object(S.Sphere((0.0, 1.0, 0.0), 1.0), M.Lambertian(T.Turbulence(1.5, seed=5, roughness=4, power=0.75))),
        rt.Object(S.Sphere((2.2, 1.0, 0.0), 1.0), M.DiffuseLight(T.Checkered(T.Fbm(3.0, seed=2, octaves=3, lacunarity=2.0, persistence=0.6), T.SolidColor((0.9, 0.8, 0.7))))),
    ]
    co = dict(aperture=0.0, aspect_ratio=1.5, focus_distance=10.0, looking_at=(0.0, 1.0, 0.0), origin=(0.0, 2.5, 9.0), shutter_close=1.0, shutter_open=0.0,
              vfov=30.0, vup=(0.0, 1.0, 0.0))
    return objects, (0.7, 0.8, 1.0), co


def far_root_variant_scene():
    """A scene built from the variants that are NOT the reference's behaviour (RT_PRIM_BOTH_ROOTS spheres, RT_MATF_FACE_NORMAL
    materials): a glass ball that refracts on the way out, a ball of smoke, a matt ball and a floor, under a sky."""
    import raytracing_amd as rt
    T, M, S = rt.Texture, rt.Material, rt.Shape
    objects = [
        rt.Object(S.Sphere((0.0, -100.0, 0.0), 100.0, both_roots=True), M.Lambertian(T.Checkered(T.SolidColor((0.2, 0.3, 0.1)), T.SolidColor((0.9, 0.9, 0.9)))).facing_normal()),
        rt.Object(S.Sphere((-2.2, 1.0, 0.0), 1.0, both_roots=True), M.glass().facing_normal()),
        rt.Object(S.ConstantMedium(S.Sphere((0.0, 1.0, 0.0), 1.0, both_roots=True), 1.5), M.Isotropic(T.SolidColor((0.9, 0.5, 0.3)))),
        rt.Object(S.Sphere((2.2, 1.0, 0.0), 1.0, both_roots=True), M.Lambertian(T.SolidColor((0.4, 0.2, 0.6))).facing_normal()),
        rt.Object(S.Sphere((0.0, 1.0, -2.5), 1.0, both_roots=True), M.Metal((0.8, 0.8, 0.8), 0.1).facing_normal()),
    ]
    co = dict(aperture=0.05, aspect_ratio=1.5, focus_distance=9.0, looking_at=(0.0, 1.0, 0.0), origin=(0.0, 2.0, 9.0), shutter_close=1.0, shutter_open=0.0,
              vfov=30.0, vup=(0.0, 1.0, 0.0))
    return objects, (0.7, 0.8, 1.0), co


def multi_scatter_scene():
    """Cornell's room with two boxes of smoke that scatter a path as often as its free flights end inside them
    (RT_PRIM_MULTI_SCATTER: NOT the reference's behaviour, whose media scatter at most once per entry): one unwrapped, one
    behind a rotation and a translation, dense enough for several scatterings per crossing."""
    import raytracing_amd as rt
    T, M, S = rt.Texture, rt.Material, rt.Shape
    white, red, green = (M.Lambertian(T.SolidColor(c)) for c in ((0.73, 0.73, 0.73), (0.65, 0.05, 0.05), (0.12, 0.45, 0.15)))
    objects = [
        rt.Object(S.rectangle((0, 0, 0), (0, 555, 555)), red),
        rt.Object(S.rectangle((555, 0, 0), (555, 555, 555)).flipped(), green),
        rt.Object(S.rectangle((0, 0, 0), (555, 0, 555)), white),
        rt.Object(S.rectangle((0, 555, 0), (555, 555, 555)).flipped(), white),
        rt.Object(S.rectangle((0, 0, 555), (555, 555, 555)).flipped(), white),
        rt.Object(S.rectangle((113, 554, 127), (443, 554, 432)), M.DiffuseLight(T.SolidColor((7, 7, 7)))),
        rt.Object(S.ConstantMedium(S.new_box((100, 50, 100), (260, 330, 260)), 0.02, multi_scatter=True), M.Isotropic(T.SolidColor((0.9, 0.9, 0.9)))),
        rt.Object(S.ConstantMedium(S.new_box((0, 0, 0), (165, 165, 165)).rotated(1, -18.0).translated((300, 20, 165)), 0.03, multi_scatter=True),
                  M.Isotropic(T.SolidColor((0.2, 0.4, 0.9)))),
    ]
    co = dict(aperture=0.0, aspect_ratio=1.0, focus_distance=10.0, looking_at=(278.0, 278.0, 0.0), origin=(278.0, 278.0, -800.0), shutter_close=1.0, shutter_open=0.0,
              vfov=40.0, vup=(0.0, 1.0, 0.0))
    return objects, (0.0, 0.0, 0.0), co


def zoo_room():
    """A Cornell room holding the wrapped shapes and the materials side by side (not one of the reference's scenes): boxes behind
    rotations about each axis and behind a translation off the fp32 grid, a tilted two-sided panel, a glass box, a metal box, a
    medium in a rotated box, a moving sphere, a metal sphere.  For the many-seed comparison of GPU and oracle."""
    import raytracing_amd as rt
    T, M, S = rt.Texture, rt.Material, rt.Shape
    X, Y, Z = 0, 1, 2
    white, red, green = (M.Lambertian(T.SolidColor(c)) for c in ((0.73, 0.73, 0.73), (0.65, 0.05, 0.05), (0.12, 0.45, 0.15)))
    objects = [
        rt.Object(S.rectangle((0, 0, 0), (0, 555, 555)), red),
        rt.Object(S.rectangle((555, 0, 0), (555, 555, 555)).flipped(), green),
        rt.Object(S.rectangle((0, 0, 0), (555, 0, 555)), white),
        rt.Object(S.rectangle((0, 555, 0), (555, 555, 555)).flipped(), white),
        rt.Object(S.rectangle((0, 0, 555), (555, 555, 555)).flipped(), white),
        rt.Object(S.rectangle((163, 554, 177), (393, 554, 382)), M.DiffuseLight(T.SolidColor((12, 12, 12)))),
        rt.Object(S.new_box((0, 0, 0), (120, 250, 120)).rotated(Y, 22.0).translated((330.3, 0.0, 340.7)), white),              # the cornell kind, off the grid
        rt.Object(S.new_box((0, 0, 0), (100, 100, 100)).rotated(X, 17.0).translated((60.1, 40.2, 300.3)), M.Metal((0.8, 0.8, 0.9), 0.3)),
        rt.Object(S.new_box((0, 0, 0), (90, 90, 90)).rotated(Z, -12.0).rotated(Y, 31.0).translated((390.5, 30.25, 120.6)), M.glass()),
        rt.Object(S.new_box((0, 0, 0), (110, 60, 110)).translated((200.33, 0.0, 90.77)), white),                               # translated only
        rt.Object(S.rectangle((0, 0, 0), (140, 180, 0)).rotated(Y, 40.0).rotated(X, -10.0).translated((150.2, 120.4, 380.6)), green),   # a tilted panel, hit from both sides
        rt.Object(S.ConstantMedium(S.new_box((0, 0, 0), (100, 140, 100)).rotated(Y, -25.0).translated((230.7, 0.0, 230.9)), 0.02), M.Isotropic(T.SolidColor((0.9, 0.9, 0.9)))),
        rt.Object(S.sphere(35.0, (0.0, (470.0, 300.0, 250.0)), (1.0, (470.0, 330.0, 250.0))), red),
        rt.Object(S.Sphere((110.0, 260.0, 160.0), 45.0), M.Metal((0.9, 0.8, 0.7), 0.05)),
    ]
    co = dict(aperture=0.0, aspect_ratio=1.0, focus_distance=10.0, looking_at=(278.0, 278.0, 0.0), origin=(278.0, 278.0, -800.0), shutter_close=1.0, shutter_open=0.0,
              vfov=40.0, vup=(0.0, 1.0, 0.0))
    return objects, (0.0, 0.0, 0.0), co
