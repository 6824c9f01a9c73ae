"""Converged-image parity: the GPU renderer against committed frames of the CPU oracle at 4 096 samples per pixel.

The fixtures (tests/golden/converged_<scene>.npz, made by tests/golden/make_converged.py) hold two oracle renders of a
small frame of each GPU configuration's scene, with independent render seeds A and B.  The GPU renders the same frame
with a third seed, so no random stream is shared with either: what is compared is the converged image, pixel() / ray_color()
of src/lib.rs:40-104 as a whole, and a bias anywhere in the kernels' sampling shows up as a distance from A that the
distance between A and B does not explain.

Stated tolerance (per channel, on the 8-bit image of lib.rs:68-72): RMSE(GPU, A) <= 1.5 x RMSE(A, B).  Two images of the
same distribution are expected at 1.0 x; the regression bound asserted here is 1.15 x, and per channel the frame's mean of
the clamped linear radiance must agree within 3 standard errors of the difference of two such renders (estimated from A - B).
A second pass over 8 x 8 block means (noise 8 x lower, so a bias has 8 x the weight) holds the same ratio below 1.25 x.
"""
import os

import numpy as np
import pytest

import raytracing_amd as rt
from oracle_lib import quantise_image

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEED_GPU = 3003
CONTRACT, REGRESSION, BLOCK_REGRESSION = 1.5, 1.15, 1.25


def _rmse_c(a, b):
    d = np.asarray(a, np.float64) - np.asarray(b, np.float64)
    return np.sqrt((d ** 2).reshape(-1, 3).mean(axis=0))


def _blocks(x, k=8):
    h, w = x.shape[0] // k * k, x.shape[1] // k * k
    return x[:h, :w].reshape(h // k, k, w // k, k, 3).mean(axis=(1, 3))


def converged_stats(mean_gpu, mean_a, mean_b):
    """Per-channel figures of one scene: 8-bit RMSEs, their ratio, block-mean ratio, frame means and the 3-sigma bound."""
    qa, qb, qg = (quantise_image(m).astype(np.float64) for m in (mean_a, mean_b, mean_gpu))
    floor, err = _rmse_c(qa, qb), _rmse_c(qg, qa)
    ca, cb, cg = (np.clip(np.nan_to_num(m, nan=0.0), 0.0, 1.0) for m in (mean_a, mean_b, mean_gpu))
    bfloor, berr = _rmse_c(_blocks(ca), _blocks(cb)), _rmse_c(_blocks(cg), _blocks(ca))
    n = ca.shape[0] * ca.shape[1]
    sigma = (ca - cb).reshape(-1, 3).std(axis=0) / np.sqrt(n)   # standard error of the difference of two renders' frame means
    dmean = np.abs(cg.reshape(-1, 3).mean(axis=0) - ca.reshape(-1, 3).mean(axis=0))
    return {"floor": floor, "err": err, "ratio": err / floor, "block_ratio": berr / bfloor, "dmean": dmean, "sigma": sigma}


@pytest.mark.parametrize("name", ["cornell", "perlin", "cornell_smoke", "final_scene"])
def test_converged_image_matches_the_oracle(name, earth):
    fx = np.load(os.path.join(GOLDEN, f"converged_{name}.npz"))
    w, h, spp, depth = (int(fx[k]) for k in ("width", "height", "spp", "depth"))
    assert spp >= 4096
    sc, co = rt.Scene.example(name, int(fx["scene_seed"]), earth if name == "final_scene" else None)
    _, mean = sc.render(rt.Camera(co), rt.ImageOptions(h, w, spp, depth), rt.RenderOptions(seed=SEED_GPU), want_mean=True)
    assert SEED_GPU not in (int(fx["seed_a"]), int(fx["seed_b"]))
    st = converged_stats(mean, fx["mean_a"].astype(np.float64), fx["mean_b"].astype(np.float64))
    print(f"\n{name} {w}x{h} {spp} spp: RMSE(gpu, A) {np.round(st['err'], 3)} vs floor RMSE(A, B) {np.round(st['floor'], 3)} -> ratio {np.round(st['ratio'], 3)}, "
          f"block ratio {np.round(st['block_ratio'], 3)}, |mean diff| / sigma {np.round(st['dmean'] / st['sigma'], 2)}")
    assert (st["floor"] > 0).all()
    assert (st["ratio"] <= CONTRACT).all() and (st["ratio"] <= REGRESSION).all(), st
    assert (st["block_ratio"] <= BLOCK_REGRESSION).all(), st
    assert (st["dmean"] <= 3.0 * st["sigma"]).all(), st


# ---- many seeds on both sides -----------------------------------------------------------------------------------------------
# One GPU render against one oracle render resolves a bias of about a per cent of the frame mean.  The fixtures
# tests/golden/frame_means/<scene>_oracle.npz (made by tools/bias_check.py) hold, for many oracle renders of the same small
# frame with independent seeds, the means of the frame's 8 x 8 blocks (per channel, of the per-pixel means clamped to [0, 1]);
# the GPU renders the frame with as many seeds of its own (the five example scenes, and a room of wrapped shapes and all
# materials that the examples do not contain, at 2 048 spp).  Asserted: the two sides' average frame means agree within 4
# standard errors of their difference (estimated from the scatter between seeds; a standard error is at most 0.05 % of the
# mean), and block by block the difference of the averages, in units of its standard error (variance pooled over both sides),
# behaves like noise: mean square at most 1.4 over the frame's blocks (1.0 expected, 0.07 its own scatter) and no block beyond
# 5.5.  This is the test that sees a 0.3 % loss of the frame, 3 % on the sides of a box, from rays meeting the surface they
# left (ray origins in fp32 on Cornell's rotated boxes, side_of_surface in rt_core.h).
FRAME_MEANS = os.path.join(GOLDEN, "frame_means")
GPU_SEEDS = 24


def _block_means(m, k):
    c = np.clip(np.nan_to_num(np.asarray(m, np.float64), nan=0.0), 0.0, 1.0)
    h, w = c.shape[:2]
    return c.reshape(h // k, k, w // k, k, 3).mean(axis=(1, 3))


@pytest.mark.parametrize("name", ["cornell", "perlin", "cornell_smoke", "final_scene", "random_spheres", "zoo_room"])
def test_many_seeds_on_both_sides_agree_over_the_frame_and_block_by_block(name, earth):
    fx = np.load(os.path.join(FRAME_MEANS, f"{name}_oracle.npz"))
    o = fx["blocks"].astype(np.float64)
    assert len(o) >= 6, "too few oracle renders in the fixture"
    w, h, k, spp, depth = (int(fx[key]) for key in ("width", "height", "block", "spp", "depth"))
    if name == "zoo_room":   # (not an example of the reference: the wrapped shapes and the materials side by side, tests/scenes_util.py)
        from scenes_util import zoo_room
        objects, background, co_kw = zoo_room()
        sc, co = rt.Scene(objects, background), rt.CameraOptions(**co_kw)
    else:
        sc, co = rt.Scene.example(name, int(fx["scene_seed"]), earth if name == "final_scene" else None)
    cam, io = rt.Camera(co), rt.ImageOptions(h, w, spp, depth)
    g = []
    for seed in range(7001, 7001 + GPU_SEEDS):
        assert seed not in fx["seeds"]
        _, mean = sc.render(cam, io, rt.RenderOptions(seed=seed), want_mean=True)
        g.append(_block_means(mean, k))
    g = np.array(g)
    fo, fg = o.mean(axis=(1, 2)), g.mean(axis=(1, 2))
    diff = fg.mean(axis=0) - fo.mean(axis=0)
    se = np.sqrt(fo.var(axis=0, ddof=1) / len(fo) + fg.var(axis=0, ddof=1) / len(fg))
    pooled = ((len(o) - 1) * o.var(axis=0, ddof=1) + (len(g) - 1) * g.var(axis=0, ddof=1)) / (len(o) + len(g) - 2)
    z = np.where(pooled > 0, (g.mean(axis=0) - o.mean(axis=0)) / np.sqrt(np.maximum(pooled, 1e-300) * (1.0 / len(o) + 1.0 / len(g))), 0.0)
    print(f"\n{name}: {len(g)} gpu against {len(o)} oracle renders: frame mean gpu - oracle {np.round(diff, 6)} = {np.round(diff / se, 2)} standard errors, "
          f"{np.round(100 * diff / fo.mean(axis=0), 3)} %; blocks: mean square z {np.mean(z ** 2):.3f}, largest |z| {np.abs(z).max():.2f}")
    assert (np.abs(diff) <= 4.0 * se).all(), (diff, se)
    assert (se <= 5e-4 * fo.mean(axis=0)).all(), (se, fo.mean(axis=0))   # (the comparison resolves 0.2 % of the mean or better)
    assert np.mean(z ** 2) <= 1.4 and np.abs(z).max() <= 5.5, (np.mean(z ** 2), np.abs(z).max())
