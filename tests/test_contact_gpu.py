"""GPU parity: inside test (K4) and contact loss (K2+K4+K5) through the C-ABI vs the reference's
golden vectors and the CPU oracle.  Hit counts / masks are integer work: bit-exact except where a
ray grazes a triangle edge within fp32 round-off (both the reference and the kernel are then
arbitrary); such points are identified with an fp64 margin and excluded."""
import numpy as np
import pytest
import torch

from oracle import contact as ocontact
from obman_train_amd.contactzones import hand_template, load_contacts
from obman_train_amd.icosphere import icosphere, multi_patch
from tests.golden.common import unpack_bits

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _blob(subdiv, B, seed, radius=40.0, patches=1):
    rng = np.random.RandomState(seed)
    v, f = multi_patch(subdiv, patches)
    scale = radius * (1.0 + 0.3 * np.sin(4.0 * v[:, :1]) * np.cos(3.0 * v[:, 1:2]))
    pts = v[None] * scale[None] * rng.uniform(0.6, 1.4, size=(B, 1, 3))
    if patches > 1:  # spread the patches out so the union is not degenerate
        n = v.shape[0] // patches
        for p in range(patches):
            pts[:, p * n:(p + 1) * n] += rng.normal(0, 25.0, size=(B, 1, 3))
    return T(pts.astype(np.float32)), f.astype(np.int32)


GRAZE = 1e-5  # r05: was 1e-4.  fp32 round-off of u, v, t at these scales is ~1e-6 (three 3-term dot products of O(1) factors)


def _margin_ok(origins, verts, faces, margin=GRAZE):
    """fp64 mask of points whose ray does not graze any triangle border (|u|,|v|,|1-u-v|,|t| > margin)."""
    o = origins.double()
    tri = verts.double()[:, T(faces.astype(np.int64))]
    a, e1, e2 = tri[:, :, 0], tri[:, :, 1] - tri[:, :, 0], tri[:, :, 2] - tri[:, :, 0]
    d = torch.tensor(ocontact.RAY_DIRECTION, dtype=torch.float64)
    pvec = torch.cross(d.expand_as(e2), e2, dim=2)
    det = (e1 * pvec).sum(2)
    inv = 1.0 / det
    tvec = o[:, :, None, :] - a[:, None]
    u = (tvec * pvec[:, None]).sum(3) * inv[:, None]
    q = torch.cross(tvec, e1[:, None].expand_as(tvec), dim=3)
    v = (q * d).sum(3) * inv[:, None]
    t = (q * e2[:, None]).sum(3) * inv[:, None]
    near_plane_hit = (u > -1e-3) & (u < 1 + 1e-3) & (v > -1e-3) & (u + v < 1 + 1e-3) & (t > -1e-3)
    graze = near_plane_hit & ((u.abs() < margin) | (v.abs() < margin) | ((1 - u - v).abs() < margin) | (t.abs() < margin))
    return ~graze.any(2)


def test_contains_matches_reference_golden(golden):
    from obman_train_amd.networks.branches.contactutils import batch_mesh_contains_points, mesh_exterior

    g = golden("contains")
    ext, hits = mesh_exterior(T(g["origins"]).cuda(), T(g["obj_verts"]).cuda(), T(g["faces"]).cuda())
    np.testing.assert_array_equal(ext.cpu().numpy(), g["exterior"])
    tri = T(g["obj_verts"])[:, T(g["faces"].astype(np.int64))].cuda()
    ext2 = batch_mesh_contains_points(T(g["origins"]).cuda(), tri)  # the reference's triangle signature
    np.testing.assert_array_equal(ext2.cpu().numpy(), g["exterior"])


@pytest.mark.parametrize("B,P,subdiv,patches", [(1, 1, 0, 1), (2, 100, 1, 1), (3, 778, 2, 1), (2, 1500, 2, 3), (2, 70, 3, 1)])
def test_contains_matches_oracle(B, P, subdiv, patches):
    from obman_train_amd import ops

    verts, faces = _blob(subdiv, B, 3, patches=patches)
    rng = np.random.RandomState(4)
    origins = T(rng.normal(0, 35, size=(B, P, 3)).astype(np.float32))
    hits = ops.mesh_contains_hits(origins.cuda(), verts.cuda(), T(faces).cuda()).cpu()
    tri = verts[:, T(faces.astype(np.int64))]
    want = ocontact.mesh_contains_points(origins, tri)
    ok = _margin_ok(origins, verts, faces)
    got = (hits & 1) == 0
    assert ok.float().mean() > 0.995
    np.testing.assert_array_equal(got[ok].numpy(), want[ok].numpy())
    # inside the margin both sides are arbitrary - but COUNT what differs there: the kernel's arithmetic is the reference's up to
    # the association of the triple products, so even grazing points almost always agree
    from tests.conftest import record_measurement

    n_in = int((~ok).sum())
    n_diff = int((got != want)[~ok].sum())
    record_measurement("contains_graze[%d,%d,%d,%d]" % (B, P, subdiv, patches), {"margin": GRAZE, "points": int(ok.numel()),
                                                                                "in_margin": n_in, "in_margin_disagree": n_diff})
    assert n_diff <= max(1, ok.numel() // 1000), (n_diff, n_in)
    assert 0.02 < (~want).float().mean() < 0.98 or P == 1  # both classes present
    if patches > 1:  # grouped mode: interior = inside ANY patch (OR of the per-patch parities)
        inside_any = ops.mesh_contains_hits(origins.cuda(), verts.cuda(), T(faces).cuda(), patches=patches).cpu()
        assert set(np.unique(inside_any.numpy()).tolist()) <= {0, 1}
        want_any = torch.zeros_like(want)
        for grp in T(faces.astype(np.int64)).chunk(patches, 0):
            want_any |= ~ocontact.mesh_contains_points(origins, verts[:, grp])
        np.testing.assert_array_equal((inside_any != 0)[ok].numpy(), want_any[ok].numpy())


def _pair_classes(origins, verts, faces, fixed=GRAZE, K=8.0):
    """Per POINT, in fp64 from the fp32 inputs (exact to ~1e-16: the "true" u, v, t, det of every pair):
    ``exact``    the crossing count exact arithmetic gives (contactutils.py:62-159's tests on the true values),
    ``fixed``    some pair's true u / v / 1-u-v lies within ``fixed`` of its threshold (or t within fixed * |tvec| of tol, or |det|
                 within fixed * |e1| |e2| of tol) while its other tests are near-passing - the 1e-5 margin of VERDICT r05,
    ``cond``     some pair's true value lies within the fp32 FORWARD-ERROR BOUND of its threshold: the bound of a 3-term fp32 dot
                 product of rounded differences times 1 / det, K * eps * Q * |tvec| / |e| with Q = |e1| |e2| / |det| (K = 8: four times what the
                 reference's own fp32 arithmetic needs on these scenes - K = 2 already separates it; covers the kernel's re-associated triple products too).  A point with no such pair is
                 DECIDED: every correct fp32 evaluation must give ``exact``."""
    eps = 2.0 ** -24
    tol = ocontact.TOL
    o = origins.double()
    tri = verts.double()[:, T(faces.astype(np.int64))]
    a, e1, e2 = tri[:, :, 0], tri[:, :, 1] - tri[:, :, 0], tri[:, :, 2] - tri[:, :, 0]
    d = torch.tensor(ocontact.RAY_DIRECTION, dtype=torch.float64)
    pvec = torch.cross(d.expand_as(e2), e2, dim=2)
    det = (e1 * pvec).sum(2)                       # [B,F]
    inv = 1.0 / (det + 0.1 * tol)
    l1, l2 = e1.norm(dim=2), e2.norm(dim=2)
    Q = (l1 * l2 * inv.abs())[:, None]             # [B,1,F]
    tvec = o[:, :, None, :] - a[:, None]           # [B,P,F,3]
    lt = tvec.norm(dim=3)  # fl(o - a) of two fp32 inputs is correctly rounded: relative error eps / 2, no cancellation term
    u = (tvec * pvec[:, None]).sum(3) * inv[:, None]
    q = torch.cross(tvec, e1[:, None].expand_as(tvec), dim=3)
    v = (q * d).sum(3) * inv[:, None]
    t = (q * e2[:, None]).sum(3) * inv[:, None]
    w = 1.0 - u - v
    par = det.abs()[:, None] < tol
    exact = ((u > 0) & (u < 1) & (v > 0) & (w > 0) & (t >= tol) & ~par).sum(2)
    len1, len2 = l1[:, None].clamp_min(1e-300), l2[:, None].clamp_min(1e-300)
    mu = K * eps * Q * (lt / len1 + u.abs())
    mv = K * eps * Q * (lt / len2 + v.abs())
    mt = K * eps * Q * (lt + t.abs())
    md = (K * eps * l1 * l2)[:, None]

    def classes(bu, bv, bw, bt, bd):
        """a pair is undecided when one test sits inside its band and every OTHER test passes with its band's benefit of doubt"""
        pu, pv, pw = (u > -bu) & (u < 1 + bu), v > -bv, w > -bw
        pt, pd = t >= tol - bt, det.abs()[:, None] >= tol - bd
        near = pu & pv & pw & pt & pd
        on = (u.abs() < bu) | ((u - 1).abs() < bu) | (v.abs() < bv) | (w.abs() < bw) | ((t - tol).abs() < bt) | \
             ((det.abs()[:, None] - tol).abs() < bd)
        return (near & on).any(2)

    fx = torch.full_like(u, fixed)
    return exact, classes(fx, fx, 2 * fx, fixed * lt, (fixed * l1 * l2)[:, None].expand_as(u)), classes(mu, mv, mu + mv, mt, md.expand_as(u))


@pytest.mark.parametrize("seed,B,P,F", [(0, 3000, 24, 32), (1, 2500, 48, 20), (2, 4000, 8, 64)])
def test_contains_adversarial_scenes_vs_oracle(seed, B, P, F):
    """VERDICT r05 weak #2 / task 2a: the inside test's arithmetic is NOT the reference's (1 / det folded into three per-triangle
    vectors, triple products re-associated), so bit-exactness against contactutils.py:62-159 is not by construction.  The adversarial
    generator of tests/test_contains_binned_gpu.py (query points within a few ulp of projected triangle borders, slivers,
    edge-on and near-parallel triangles, tiny triangles around the parallel threshold, far outliers; finite scenes) goes through
    the fp32 ORACLE, the all-pairs kernel and the product (grid-culled) kernel, and every point is classified against EXACT
    arithmetic (fp64 on the fp32 inputs):
      * a DECIDED point (no pair within the fp32 forward-error bound of a threshold) must get the exact crossing count from the
        kernels AND from the oracle - zero tolerance;
      * inside the bound both sides are legitimately arbitrary: disagreements are counted and their rate is frozen.
    The fixed 1e-5 margin of the smooth-blob tests is reported beside it (on ill-conditioned triangles - Q = |e1||e2|/|det| up to
    1e4 here - fp32 moves u, v by more than 1e-5, so that margin alone cannot separate the classes)."""
    from obman_train_amd import ops
    from tests.conftest import record_measurement
    from tests.test_contains_binned_gpu import _random_scenes

    rng = np.random.RandomState(100 + seed)
    pts, verts, faces = _random_scenes(rng, B, P, F, Nv=max(8, F // 2))
    assert np.isfinite(pts).all() and np.isfinite(verts).all()
    o, vv = T(pts), T(verts)
    counts = {}
    for name, kw in (("all_pairs", {"all_pairs": True}), ("product", {})):
        counts[name] = ops.mesh_contains_hits(o.cuda(), vv.cuda(), T(faces).cuda(), **kw).cpu().long()
    want = torch.cat([ocontact.mesh_contains_points(o[i:i + 500], vv[i:i + 500][:, T(faces.astype(np.int64))], return_counts=True)
                      for i in range(0, B, 500)]).long()
    exact, in_fixed, in_cond = [], [], []
    for i in range(0, B, 500):
        e, f, c = _pair_classes(o[i:i + 500], vv[i:i + 500], faces)
        exact.append(e); in_fixed.append(f); in_cond.append(c)
    exact, in_fixed, in_cond = torch.cat(exact), torch.cat(in_fixed), torch.cat(in_cond)
    got = counts["all_pairs"]
    assert torch.equal(counts["product"], got)            # the culled kernel is the all-pairs kernel, bit for bit
    decided = ~in_cond
    rec = {"points": int(exact.numel()), "with_a_crossing": int((exact > 0).sum()),
           "in_fixed_1e-5_margin": int(in_fixed.sum()), "in_error_bound": int(in_cond.sum()),
           "kernel_vs_oracle_disagree_decided": int((got != want)[decided].sum()),
           "kernel_vs_exact_disagree_decided": int((got != exact)[decided].sum()),
           "oracle_vs_exact_disagree_decided": int((want != exact)[decided].sum()),
           "kernel_vs_oracle_disagree_in_bound": int((got != want)[in_cond].sum()),
           "kernel_vs_exact_disagree_in_bound": int((got != exact)[in_cond].sum()),
           "oracle_vs_exact_disagree_in_bound": int((want != exact)[in_cond].sum()),
           "kernel_vs_oracle_disagree_outside_fixed_margin": int((got != want)[~in_fixed].sum()),
           "kernel_vs_oracle_parity_disagree_total": int(((got & 1) != (want & 1)).sum())}
    record_measurement("contains_adversarial[%d,%d,%d,%d]" % (seed, B, P, F), rec)
    assert rec["in_error_bound"] > 0.01 * rec["points"], rec       # the generator does reach the border region (r05: 0 points)
    assert rec["with_a_crossing"] > 0.02 * rec["points"], rec
    assert rec["kernel_vs_exact_disagree_decided"] == 0, rec       # bit-exact wherever fp32 can decide
    assert rec["oracle_vs_exact_disagree_decided"] == 0, rec       # (and the bound is a bound for the reference's arithmetic too)
    assert rec["kernel_vs_oracle_disagree_decided"] == 0, rec
    # frozen after the first measured run (profiles/r06_parity_measured.md): inside the bound the two fp32 evaluations differ on
    # at most this fraction of the undecided points
    assert rec["kernel_vs_oracle_disagree_in_bound"] <= IN_BOUND_RATE * rec["in_error_bound"] + 2, rec


IN_BOUND_RATE = 0.2  # measured 0.120 - 0.128 on the three parametrisations (profiles/r06_parity_measured.md)


def test_grouped_inside_test_on_overlapping_patches():
    """Two concentric closed spheres as one 2-patch mesh (what duplicated or overlapping AtlasNet patches look like): a point
    inside both crosses an even number of triangles in total - the plain parity calls it exterior - but is interior in
    grouped mode; between the shells both agree; far away both say exterior.  Also with a triangle split (large B x P)."""
    from obman_train_amd import ops

    v, f = icosphere(2)
    verts = np.concatenate([v * 30.0, v * 60.0], 0)[None].astype(np.float32)
    faces = np.concatenate([f, f + v.shape[0]], 0).astype(np.int32)
    pts = np.array([[[1.0, 2.0, 3.0], [40.0, 5.0, -3.0], [500.0, 0.0, 0.0]]], dtype=np.float32)
    for B in (1, 70):
        vv, pp = T(np.repeat(verts, B, 0)).cuda(), T(np.repeat(pts, B, 0)).cuda()
        plain = ops.mesh_contains_hits(pp, vv, T(faces).cuda()).cpu().numpy()
        grouped = ops.mesh_contains_hits(pp, vv, T(faces).cuda(), patches=2).cpu().numpy()
        assert (plain[:, 0] % 2 == 0).all() and (grouped[:, 0] == 1).all()   # inside both shells
        assert (plain[:, 1] % 2 == 1).all() and (grouped[:, 1] == 1).all()   # between the shells
        assert (plain[:, 2] == 0).all() and (grouped[:, 2] == 0).all()       # outside
    # the loss sees it: penetration mask through compute_contact_loss(obj_patches=2)
    from obman_train_amd.networks.branches.contactloss import compute_contact_loss

    hand = T(np.repeat(pts, 2, 0)).cuda()
    _, _, info, _ = compute_contact_loss(hand, None, T(np.repeat(verts, 2, 0)).cuda(), faces, obj_patches=2)
    np.testing.assert_array_equal(info["repulsion_masks"].cpu().numpy(), np.array([[True, True, False]] * 2))


def test_contains_full_size_properties():
    """bs 64 x 778 points x 1280 triangles (config 2) and a 25-patch mesh: far points are outside, the
    centre of a star-shaped blob is inside, triangle order does not matter (integer counts), repeat
    launches are identical (atomic merge of integer counts)."""
    from obman_train_amd import ops

    for subdiv, patches, B in ((3, 1, 64), (3, 25, 4)):
        verts, faces = _blob(subdiv, B, 7, patches=1 if patches == 1 else patches)
        P = 778
        rng = np.random.RandomState(8)
        origins = T(rng.normal(0, 30, size=(B, P, 3)).astype(np.float32))
        origins[:, 0] = 1e4
        if patches == 1:
            origins[:, 1] = 0.0
        fc = T(faces).cuda()
        hits = ops.mesh_contains_hits(origins.cuda(), verts.cuda(), fc)
        assert torch.all(hits[:, 0] == 0)
        if patches == 1:
            assert torch.all(hits[:, 1] % 2 == 1)
        perm = torch.randperm(fc.shape[0]).cuda()
        hits_p = ops.mesh_contains_hits(origins.cuda(), verts.cuda(), fc[perm].contiguous())
        assert torch.equal(hits, hits_p)
        assert torch.equal(hits, ops.mesh_contains_hits(origins.cuda(), verts.cuda(), fc))
        # spot-check 2 samples against the oracle
        tri = verts[:2][:, T(faces.astype(np.int64))]
        want = ocontact.mesh_contains_points(origins[:2], tri)
        ok = _margin_ok(origins[:2], verts[:2], faces)
        np.testing.assert_array_equal(((hits[:2].cpu() & 1) == 0)[ok].numpy(), want[ok].numpy())


def test_contact_loss_all_modes_match_reference_golden(golden):
    from obman_train_amd.networks.branches.contactloss import compute_contact_loss

    g = golden("contact")
    hand_faces = hand_template()[1]
    combos = [str(c).split("|") for c in g["combos"]]
    for ci, (zone_mode, cmode, kmode, target) in enumerate(combos):
        tag = "c%02d_" % ci
        hand = T(g["hand"]).cuda().requires_grad_()
        obj = T(g["obj"]).cuda().requires_grad_()
        missed, penetr, info, metrics = compute_contact_loss(
            hand, hand_faces, obj, g["faces"], contact_thresh=10, contact_mode=cmode, collision_thresh=20,
            collision_mode=kmode, contact_target=target, contact_zones=zone_mode)
        np.testing.assert_allclose(float(missed), float(g[tag + "missed"][0]), rtol=1e-4, err_msg=str(combos[ci]))
        np.testing.assert_allclose(float(penetr), float(g[tag + "penetr"][0]), rtol=1e-4, err_msg=str(combos[ci]))
        np.testing.assert_allclose(float(metrics["max_penetr"]), float(g[tag + "max_penetr"]), rtol=1e-4)
        np.testing.assert_allclose(float(metrics["mean_penetr"]), float(g[tag + "mean_penetr"]), rtol=1e-4)
        shape = tuple(info["repulsion_masks"].shape)
        np.testing.assert_array_equal(info["attraction_masks"].cpu().numpy() != 0, unpack_bits(g[tag + "attr_mask"], shape))
        np.testing.assert_array_equal(info["repulsion_masks"].cpu().numpy(), unpack_bits(g[tag + "rep_mask"], shape))
        assert str(info["attraction_masks"].dtype) == str(g[tag + "attr_dtype"])
        (missed.sum() + 2.0 * penetr.sum()).backward()
        for got, want in ((hand.grad, g[tag + "grad_hand"]), (obj.grad, g[tag + "grad_obj"])):
            got = got.cpu().numpy() if got is not None else np.zeros_like(want)
            err = np.abs(got - want).max()
            assert err <= 1e-3 * max(np.abs(want).max(), 1e-6), (combos[ci], err, np.abs(want).max())
        if ci == 0:
            np.testing.assert_allclose(info["min_dists"].cpu().numpy(), g["min_dists"], rtol=1e-4, atol=2e-3)
            np.testing.assert_allclose(info["contact_points"].cpu().numpy(), g["contact_points"], rtol=1e-6)


def test_meshiou_matches_reference_golden(golden):
    from obman_train_amd.networks.branches.contactloss import meshiou

    g = golden("contact")
    ious, auc = meshiou(T(g["iou_gt_dists"]).cuda(), T(g["min_dists"]).cuda())
    np.testing.assert_allclose(ious.cpu().numpy(), g["iou_batch"], rtol=1e-6)
    np.testing.assert_allclose(float(auc), float(g["iou_auc"]), rtol=1e-6)


def test_contact_empty_masks_give_zero_loss_and_zero_grad():
    """Object far away: nothing penetrates (penetr mask empty), nothing within dist_sq threshold."""
    from obman_train_amd.networks.branches.contactloss import compute_contact_loss

    tv, tf = hand_template()
    hand = (T(tv) * 1000).unsqueeze(0).repeat(2, 1, 1).cuda().requires_grad_()
    v, f = icosphere(1)
    obj = (T(v.astype(np.float32)) * 10 + 5000.0).unsqueeze(0).repeat(2, 1, 1).cuda().requires_grad_()
    missed, penetr, info, metrics = compute_contact_loss(hand, tf, obj, f, contact_thresh=10, contact_mode="dist_sq",
                                                         collision_thresh=20, collision_mode="dist_sq")
    assert float(missed) == 0.0 and float(penetr) == 0.0 and float(metrics["max_penetr"]) == 0.0
    (missed + penetr).backward()
    assert torch.all(hand.grad == 0) and torch.all(obj.grad == 0)
    with pytest.raises(ValueError):
        compute_contact_loss(hand, tf, obj, f, contact_mode="nope")
    with pytest.raises(ValueError):
        compute_contact_loss(hand, tf, obj, f, contact_zones="palm")


@pytest.mark.parametrize("subdiv,patches", [(1, 1), (3, 1), (3, 25)])
def test_contact_backward_object_side_is_the_ordered_scatter_of_the_hand_side(subdiv, patches):
    """d/d(obj) of the contact tail = for every object point the sum, in ASCENDING hand-vertex order, of the g_delta of the vertices
    whose closest point it is (``contactloss.py:173-180``: ``batch_index_select`` backward), and d/d(hand) = -g_delta.  The kernel's
    owner walk must reproduce a sequential fp32 scatter of the hand-side gradient bit for bit - 42 points (every point shared by
    ~19 vertices), 642, and 16 050 (eight 2 048-point slices per sample)."""
    from obman_train_amd import ops
    from obman_train_amd.networks.branches.contactloss import compute_contact_loss

    B = 3
    tv, tf = hand_template()
    rng = np.random.RandomState(5)
    hand0 = (T(tv) * 1000).unsqueeze(0).repeat(B, 1, 1) + T(rng.normal(0, 2.0, size=(B, 1, 3)).astype(np.float32))
    obj0, faces = _blob(subdiv, B, seed=11, patches=patches)
    grads = {}
    for target in ("all", "obj", "hand"):
        hand = hand0.clone().cuda().requires_grad_()
        obj = obj0.clone().cuda().requires_grad_()
        missed, penetr, info, _ = compute_contact_loss(hand, tf, obj, faces, contact_mode="dist_tanh", collision_mode="dist_tanh",
                                                       contact_target=target, obj_patches=patches)
        (1.7 * missed + 0.6 * penetr).backward()
        grads[target] = (hand.grad.cpu().numpy(), obj.grad.cpu().numpy())
        if target == "all":
            assert bool(info["repulsion_masks"].any()) and bool(info["attraction_masks"].any())
    _, idx, _, _ = ops.pairmin(hand0.cuda(), obj0.cuda(), want_y=False)
    idx = idx.cpu().numpy()
    gh, go = grads["all"]
    want = np.zeros_like(go)
    for b in range(B):
        for v in range(gh.shape[1]):
            want[b, idx[b, v]] += -gh[b, v]  # float32 adds, ascending v
    assert np.count_nonzero(want.any(axis=2)) < gh.shape[0] * gh.shape[1]  # several vertices share a closest point
    assert np.array_equal(go.view(np.uint32), want.view(np.uint32))
    # contact_target only selects the side that receives the gradient (contactloss.py:187-196)
    assert np.array_equal(grads["obj"][1].view(np.uint32), go.view(np.uint32)) and not grads["obj"][0].any()
    assert np.array_equal(grads["hand"][0].view(np.uint32), gh.view(np.uint32)) and not grads["hand"][1].any()
    # one side only asks for its gradient (the launcher then owns every vertex with one block per sample / skips the hand writes)
    for side in ("hand", "obj"):
        hand = hand0.clone().cuda().requires_grad_(side == "hand")
        obj = obj0.clone().cuda().requires_grad_(side == "obj")
        missed, penetr, _, _ = compute_contact_loss(hand, tf, obj, faces, contact_mode="dist_tanh", collision_mode="dist_tanh",
                                                    obj_patches=patches)
        (1.7 * missed + 0.6 * penetr).backward()
        got, ref = (hand.grad, gh) if side == "hand" else (obj.grad, go)
        assert np.array_equal(got.cpu().numpy().view(np.uint32), ref.view(np.uint32)), side
