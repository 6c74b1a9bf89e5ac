"""Convex-hull LINK geometry: GJK / EPA on support functions (include/tmx_gjk.h, tmx_geom.h
tmx_hull_closest_to_obstacle), the contact data of hull links against every obstacle primitive, discrete and cast (swept).
The reference gets these contacts from tesseract / Bullet (absent here: unpinned by nature), so the geometry is pinned
against brute force - the exact distance / penetration depth of two polytopes from the convex hull of their Minkowski difference
(scipy / Qhull) - and against the reference's own KAT for a swept box link, trajopt/test/cast_cost_unit.cpp:64-117 with its fixture
trajopt_common/data/config/box_cast_test.json (tests/golden/json/box_cast_test.json) + boxbot.urdf restated below: "the initial
trajectory is in collision, the optimized one is collision free"."""
import os

import numpy as np
import pytest

import parity_checks as pc
from trajopt_amd import abi, configs, json_io, runtime
from trajopt_amd.problem import Robot, _tf12

HERE = os.path.dirname(os.path.abspath(__file__))
CUBE = np.array([[sx, sy, sz] for sx in (-.5, .5) for sy in (-.5, .5) for sz in (-.5, .5)], dtype=np.float64)


def _rot(rng):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _signed_distance_exact(VA, VB):
    """signed distance of two polytopes (vertex clouds) from the hull of their Minkowski difference A - B: the origin's distance to
    it when outside (closest point over the facets, by projection), minus the smallest facet offset when inside"""
    from scipy.spatial import ConvexHull
    M = (VA[:, None, :] - VB[None, :, :]).reshape(-1, 3)
    H = ConvexHull(M)
    offs = H.equations[:, 3]
    if (offs < 0).all():
        return -float((-offs).min())
    # outside: distance to the hull = min over facets (triangles) of the point-triangle distance
    best = np.inf
    for simp in H.simplices:
        a, b, c = M[simp]
        best = min(best, _pt_tri(np.zeros(3), a, b, c))
    return float(best)


def _pt_tri(p, a, b, c):
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = ab @ ap, ac @ ap
    if d1 <= 0 and d2 <= 0:
        return np.linalg.norm(ap)
    bp = p - b
    d3, d4 = ab @ bp, ac @ bp
    if d3 >= 0 and d4 <= d3:
        return np.linalg.norm(bp)
    vc = d1 * d4 - d3 * d2
    if vc <= 0 and d1 >= 0 and d3 <= 0:
        return np.linalg.norm(p - (a + ab * d1 / (d1 - d3)))
    cp = p - c
    d5, d6 = ab @ cp, ac @ cp
    if d6 >= 0 and d5 <= d6:
        return np.linalg.norm(cp)
    vb = d5 * d2 - d1 * d6
    if vb <= 0 and d2 >= 0 and d6 <= 0:
        return np.linalg.norm(p - (a + ac * d2 / (d2 - d6)))
    va = d3 * d6 - d5 * d4
    if va <= 0 and (d4 - d3) >= 0 and (d5 - d6) >= 0:
        return np.linalg.norm(p - (b + (c - b) * (d4 - d3) / ((d4 - d3) + (d5 - d6))))
    den = 1.0 / (va + vb + vc)
    return np.linalg.norm(p - (a + ab * vb * den + ac * vc * den))


def _hull_tris(V):
    from scipy.spatial import ConvexHull
    H = ConvexHull(V)
    tris = []
    for simp, eq in zip(H.simplices, H.equations):
        a, b, c = V[simp]
        if np.cross(b - a, c - a) @ eq[:3] < 0:
            b, c = c, b
        tris.append([a, b, c])
    return np.array(tris)


def test_gjk_epa_against_the_minkowski_hull(orc):
    """random hull links against mesh / box / capsule / sphere obstacles, separated and overlapping: distance, penetration depth and
    the witness points (p in the link hull, q on the obstacle, q - p along the contact direction)"""
    rng = np.random.default_rng(7)
    n_sep = n_pen = 0
    worst = 0.0
    for case in range(160):
        nv = int(rng.integers(4, 12))
        hv = rng.standard_normal((nv, 3)) * 0.3
        R, t = _rot(rng), rng.standard_normal(3) * rng.choice([0.15, 0.5, 1.1])
        VA = hv @ R.T + t
        kind = case % 4
        if kind == 0:     # convex mesh obstacle (a random hull, triangles counter-clockwise seen from outside)
            vb = rng.standard_normal((int(rng.integers(4, 10)), 3)) * 0.3
            tris = _hull_tris(vb)
            VB = tris.reshape(-1, 3)
            ins, p, q, _ = orc.hull_contact(hv, R, t, np.zeros(3), mesh=tris)
        elif kind == 1:   # box
            h, Rb, oc = rng.uniform(0.1, 0.4, 3), _rot(rng), rng.standard_normal(3) * 0.2
            VB = np.array([[sx * h[0], sy * h[1], sz * h[2]] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]) @ Rb.T + oc
            ins, p, q, _ = orc.hull_contact(hv, R, t, oc, box=np.concatenate([h, Rb.reshape(-1)]))
        elif kind == 2:   # capsule core = segment
            oc, ax = rng.standard_normal(3) * 0.2, rng.standard_normal(3) * 0.4
            VB = np.array([oc, oc + ax, oc + 0.5 * ax + 1e-9 * rng.standard_normal(3), oc + 0.3 * ax - 1e-9 * rng.standard_normal(3)])
            ins, p, q, _ = orc.hull_contact(hv, R, t, oc, axis=ax)
        else:             # sphere core = point
            oc = rng.standard_normal(3) * 0.3
            VB = oc[None, :] + 1e-9 * rng.standard_normal((4, 3))
            ins, p, q, _ = orc.hull_contact(hv, R, t, oc)
        ref = _signed_distance_exact(VA, VB)
        got = np.linalg.norm(q - p) * (-1.0 if ins else 1.0)
        tol = 1e-7 if kind < 2 else 1e-6    # (the segment / point stand-ins of the reference are slivers of width 1e-9)
        assert abs(got - ref) <= tol * (1.0 + abs(ref)), (case, kind, got, ref)
        worst = max(worst, abs(got - ref))
        n_pen += int(ins)
        n_sep += int(not ins)
        # p lies in the link hull: no vertex direction separates it
        from scipy.spatial import ConvexHull
        eqs = ConvexHull(VA).equations
        assert (eqs[:, :3] @ p + eqs[:, 3]).max() <= 1e-9
    assert n_sep >= 40 and n_pen >= 20, (n_sep, n_pen)
    print(f"GJK / EPA vs Minkowski hull: {n_sep} separated, {n_pen} overlapping, worst |difference| {worst:.2e}")


def test_swept_hull_is_the_hull_of_both_placements(orc):
    rng = np.random.default_rng(11)
    for case in range(60):
        hv = rng.standard_normal((6, 3)) * 0.2
        R0, R1 = _rot(rng), _rot(rng)
        t0 = rng.standard_normal(3) * 0.7
        t1 = t0 + rng.standard_normal(3) * 0.5
        vb = rng.standard_normal((6, 3)) * 0.25
        tris = _hull_tris(vb)
        ins, p, q, tau = orc.hull_contact(hv, R0, t0, np.zeros(3), R1=R1, t1=t1, mesh=tris)
        VA = np.vstack([hv @ R0.T + t0, hv @ R1.T + t1])
        ref = _signed_distance_exact(VA, tris.reshape(-1, 3))
        got = np.linalg.norm(q - p) * (-1.0 if ins else 1.0)
        assert abs(got - ref) <= 1e-7 * (1.0 + abs(ref)), (case, got, ref)
        assert 0.0 <= tau <= 1.0
    # a pure translation of a cube past a point: the contact belongs to the start pose before the point, to the end pose after it,
    # and to the fraction in between while the point faces the swept side
    for x, want in ((-2.0, 0.0), (3.0, 1.0), (0.5, None)):
        ins, p, q, tau = orc.hull_contact(CUBE, np.eye(3), np.zeros(3), np.array([x, 2.0, 0.0]), R1=np.eye(3), t1=np.array([1.0, 0.0, 0.0]))
        assert not ins and abs(np.linalg.norm(q - p) - (1.5 if want is None else np.hypot(1.5, abs(x - (0.0 if want == 0.0 else 1.0)) - 0.5))) < 1e-9
        if want is not None:
            assert tau == want
        else:
            assert 0.0 < tau < 1.0


# ---- the reference's KAT: a swept BOX link ---------------------------------------------------------------------------------------
def _boxbot_problem():
    # boxbot.urdf: two prismatic joints (x, then y) carry a 1 x 1 x 1 box; a static 1 x 1 x 1 box sits at the origin
    rob = Robot(joint_types=[1, 1], origins=[_tf12(), _tf12()], axes=[np.array([1.0, 0, 0]), np.array([0, 1.0, 0])],
                lower=np.array([-20.0, -20.0]), upper=np.array([20.0, 20.0]))
    rob.link_spheres = [(1, (0.0, 0.0, 0.0), 0.0, ("hull", CUBE))]
    box = ((0.0, 0.0, 0.0), 0.0, ("box", (0.5, 0.5, 0.5), None))
    env = json_io.Environment(manipulators={"manipulator": rob}, tip_links={"manipulator": "boxbot_link"},
                              joint_state={"manipulator": [-1.9, 0.0]}, obstacles=[box])
    text = open(os.path.join(HERE, "golden", "json", "box_cast_test.json")).read()
    return json_io.construct_problem(text, env)


def _boxes_overlap_along(traj, n=4001):
    """checkTrajectory with a CONTINUOUS config and margin 0 (cast_cost_unit.cpp:79-88): does the swept box meet the obstacle box?
    Both are axis-aligned unit cubes and the link only translates in x / y: they overlap iff |dx| < 1 and |dy| < 1 somewhere."""
    for a, b in zip(traj[:-1], traj[1:]):
        s = np.linspace(0.0, 1.0, n)[:, None]
        pts = a[None, :] * (1 - s) + b[None, :] * s
        if ((np.abs(pts[:, 0]) < 1.0 - 1e-9) & (np.abs(pts[:, 1]) < 1.0 - 1e-9)).any():
            return True
    return False


def _check_boxbot(x, status, init):
    assert _boxes_overlap_along(init)                   # "Initial trajectory is in collision"
    assert status == abi.OPT_CONVERGED
    assert not _boxes_overlap_along(x)                  # "Final trajectory is collision free"
    assert np.abs(x[0] - init[0]).max() < 1e-9 and np.abs(x[-1] - np.array([1.9, 3.8])).max() < 1e-3


def test_box_cast_kat_oracle(orc):
    pp = _boxbot_problem()
    o = orc.sqp_batch(pp.pci.to_desc(), pp.init_traj[None, :, :])
    _check_boxbot(o["x"][0], o["status"][0], pp.init_traj)


def test_box_cast_kat_kernel_sources_on_host(hostemu_lib, orc):
    pp = _boxbot_problem()
    opt = runtime.BatchedTrustRegionSQP(pp.pci, lib_path=hostemu_lib)
    opt.setParameters(pp.sqp_params)
    opt.initialize(pp.init_traj[None, :, :])
    opt.optimize()
    r = opt.results()
    _check_boxbot(r["x"][0], r["status"][0], pp.init_traj)
    o = orc.sqp_batch(pp.pci.to_desc(), pp.init_traj[None, :, :])
    assert (r["n_qp_solves"] == o["n_qp_solves"]).all() and np.abs(r["x"] - o["x"]).max() < 1e-5
    opt.ctx.close()


@pytest.mark.gpu
def test_box_cast_kat_device(orc):
    pp = _boxbot_problem()
    opt = runtime.BatchedTrustRegionSQP(pp.pci)
    opt.setParameters(pp.sqp_params)
    opt.initialize(np.repeat(pp.init_traj[None, :, :], 3, axis=0))
    opt.optimize()
    r = opt.results()
    for b in range(3):
        _check_boxbot(r["x"][b], r["status"][b], pp.init_traj)
    o = orc.sqp_batch(pp.pci.to_desc(), pp.init_traj[None, :, :])
    assert (r["n_qp_solves"] == o["n_qp_solves"][0]).all() and np.abs(r["x"] - o["x"][0][None]).max() < 1e-5
    opt.ctx.close()


# ---- hull links on the 4-DOF test arm: every stage against the oracle -------------------------------------------------------------
def _stages(ctx, orc, cid, B=3):
    pci, s, g = pc.cfg(cid)
    x0 = configs.seeds_for(9, pci, s, g, B, sigma=0.05)
    desc = pc.make_ctx_inputs(ctx, pci, x0)
    pc.check_evaluate(ctx, orc, desc, x0, tol=1e-12)
    for b in range(B):
        pc.check_first_qp_structure(ctx, orc, desc, x0, b, val_tol=1e-12)
    assert all(same for same, _ in pc.check_first_qp_solve(ctx, orc, desc, x0))
    ctx.set_x0(x0)
    r, o, same, dx = pc.check_full_sqp(ctx, orc, desc, x0, exact=False)
    assert (r["status"] == o["status"]).all() and (dx[same] <= 1e-5).all() and same.sum() >= B - 1, (same, dx)


@pytest.mark.parametrize("cid", [45, 46, 47])
def test_hull_links_stage_by_stage_on_host_build(hostemu_lib, orc, cid):
    ctx = runtime.Context(0, hostemu_lib)
    _stages(ctx, orc, cid)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cid", [45, 46, 47])
def test_hull_links_stage_by_stage_on_device(gpu_ctx_factory, orc, cid):
    ctx = gpu_ctx_factory()
    _stages(ctx, orc, cid, B=8)
    ctx.close()


# ---- the PR2's own convex collision mesh as a link hull ---------------------------------------------------------------------------
def _config1_with_pr2_forearm():
    """BASELINE config 1 (the PR2 right arm) with the robot's forearm collision geometry - trajopt_common/data/pr2/meshes/forearm_v0/
    convex/forearm_convex.stla, 97 vertices (tests/golden/pr2_forearm_convex_vertices.npy, tests/tools/make_pr2_hull.py) - as a hull
    on the forearm link (the child of the forearm-roll joint, joint 4) instead of that link's sphere"""
    from trajopt_amd import meshes
    pci, s, g = configs.config1()
    v = np.load(os.path.join(HERE, "golden", "pr2_forearm_convex_vertices.npy"))
    rob = pci.robot
    rob.link_spheres = [p for p in rob.link_spheres if p[0] != 4] + [meshes.hull_link(4, meshes.convex_hull_vertices(v))]
    return pci, s, g


def test_mesh_loader_reads_the_reference_meshes():
    """ASCII STL, binary STL and OBJ variants of one PR2 mesh give the same vertex cloud (build container only: needs /root/reference)"""
    from trajopt_amd import meshes
    base = "/root/reference/trajopt_common/data/pr2/meshes/forearm_v0/convex/forearm_convex"
    if not os.path.exists(base + ".stla"):
        pytest.skip("reference checkout not present")
    a, b, c = (meshes.load_mesh_vertices(base + ext) for ext in (".stla", ".stlb", ".obj"))
    gold = np.load(os.path.join(HERE, "golden", "pr2_forearm_convex_vertices.npy"))
    assert np.array_equal(a, gold)
    for other in (b, c):    # (float32 / other print precision: the same points to 1e-5)
        assert len(other) >= 0.9 * len(a)
        d = np.linalg.norm(other[:, None, :] - a[None, :, :], axis=2).min(axis=1)
        assert d.max() < 1e-4
    hv = meshes.convex_hull_vertices(a)
    assert 20 <= len(hv) <= len(a)
    assert len(meshes.convex_hull_vertices(a, max_vertices=16)) == 16


def test_pr2_forearm_hull_on_config1_host_build(hostemu_lib, orc):
    pci, s, g = _config1_with_pr2_forearm()
    x0 = configs.seeds_for(1, pci, s, g, 2)
    ctx = runtime.Context(0, hostemu_lib)
    desc = pc.make_ctx_inputs(ctx, pci, x0)
    pc.check_evaluate(ctx, orc, desc, x0, tol=1e-12)
    pc.check_first_qp_structure(ctx, orc, desc, x0, 0, val_tol=1e-12)
    assert all(same for same, _ in pc.check_first_qp_solve(ctx, orc, desc, x0))
    ctx.close()


@pytest.mark.gpu
def test_pr2_forearm_hull_on_config1_device(gpu_ctx_factory, orc):
    pci, s, g = _config1_with_pr2_forearm()
    B = 8
    x0 = configs.seeds_for(1, pci, s, g, B)
    ctx = gpu_ctx_factory()
    desc = pc.make_ctx_inputs(ctx, pci, x0)
    pc.check_evaluate(ctx, orc, desc, x0, tol=1e-12)
    pc.check_first_qp_structure(ctx, orc, desc, x0, 0, val_tol=1e-12)
    assert all(same for same, _ in pc.check_first_qp_solve(ctx, orc, desc, x0))
    ctx.set_x0(x0)
    r, o, same, dx = pc.check_full_sqp(ctx, orc, desc, x0, exact=False)
    print(f"config 1 with the PR2 forearm hull: same history {same.sum()}/{B}, within 1e-5: {(dx <= 1e-5).sum()}/{B}, worst {dx.max():.2e}")
    assert (r["status"] == o["status"]).all() and (dx[same] <= 1e-5).all() and same.sum() >= B // 2
    ctx.close()
