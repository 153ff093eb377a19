"""Property tests of the ORCA restatement.  rvo2 itself is unavailable here; its arithmetic is pinned end to end by the
reference's shipped ORCA-robot evaluation log (tests/test_reference_eval_log.py) -- these properties and the bit-exact
agreement of the two independent implementations are the fine-grained complement."""
import numpy as np
import pytest

from oracle import oracle as O

EPS = 2e-5


def _det(a, b):
    return a[0] * b[1] - a[1] * b[0]


def _scene(rs, n, spread):
    s = np.zeros(8, np.float32)
    s[0:2] = rs.uniform(-spread, spread, 2); s[2:4] = rs.uniform(-1, 1, 2); s[4] = 0.46; s[5] = rs.uniform(0.5, 1.5)
    p = rs.uniform(-1, 1, 2); s[6:8] = p / max(np.linalg.norm(p), 1.0)
    o = np.zeros((n, 5), np.float32)
    o[:, 0:2] = rs.uniform(-spread, spread, (n, 2)); o[:, 2:4] = rs.uniform(-1, 1, (n, 2)); o[:, 4] = rs.uniform(0.46, 0.66, n)
    return s, o


def test_speed_limit_and_halfplanes_satisfied_when_feasible():
    rs = np.random.RandomState(0)
    n_feasible = 0
    for _ in range(400):
        s, o = _scene(rs, 19, 6.0)
        (vx, vy), lines, fail = O.orca_velocity(s, o, want_lines=True)
        assert np.hypot(vx, vy) <= s[5] * (1 + 1e-4) + 1e-5
        if fail == len(lines):   # LP2 succeeded: every constraint holds
            n_feasible += 1
            for pt_x, pt_y, dx, dy in lines:
                assert _det((dx, dy), (pt_x - vx, pt_y - vy)) <= EPS
    assert n_feasible > 150


def test_free_agent_takes_preferred_velocity():
    s = np.array([0, 0, 0.3, -0.2, 0.46, 1.0, 0.6, 0.3], np.float32)
    far = np.array([[30, 30, 0, 0, 0.46], [-25, 4, 1, 0, 0.46]], np.float32)
    assert O.orca_velocity(s, far, neighbor_dist=10.0) == (np.float32(0.6), np.float32(0.3))
    s[6:8] = (3.0, 4.0)   # longer than max speed: clipped onto the speed disc along the same direction
    vx, vy = O.orca_velocity(s, far, neighbor_dist=10.0)
    assert np.hypot(vx, vy) == pytest.approx(1.0, abs=1e-6) and vx / vy == pytest.approx(0.75, rel=1e-6)


def test_head_on_pair_is_point_symmetric_and_reciprocal():
    a = np.array([-2, 0, 1, 0, 0.46, 1.0, 1, 0], np.float32)
    b = np.array([2, 0, -1, 0, 0.46, 1.0, -1, 0], np.float32)
    va = O.orca_velocity(a, np.array([[b[0], b[1], b[2], b[3], b[4]]], np.float32))
    vb = O.orca_velocity(b, np.array([[a[0], a[1], a[2], a[3], a[4]]], np.float32))
    assert va[0] == pytest.approx(-vb[0], abs=1e-6) and va[1] == pytest.approx(-vb[1], abs=1e-6)
    assert va != (1.0, 0.0)            # they do react
    # reciprocity: with both new velocities the pair stays collision free over the time horizon
    pa, pb = a[0:2].astype(np.float64), b[0:2].astype(np.float64)
    for _ in range(20):
        pa = pa + 0.25 * np.array(va); pb = pb + 0.25 * np.array(vb)
        assert np.linalg.norm(pa - pb) >= 0.92 - 1e-4


# ---- a third, independent implementation of RVO2's linear programs in exact-ish arithmetic (python floats = fp64) ----
def _lp1_64(lines, i, r, opt, dir_opt):
    p, d = lines[i]
    dot = p[0] * d[0] + p[1] * d[1]
    disc = dot * dot + r * r - (p[0] ** 2 + p[1] ** 2)
    if disc < 0:
        return None
    sq = disc ** 0.5
    tl, tr = -dot - sq, -dot + sq
    for j in range(i):
        pj, dj = lines[j]
        den, num = _det(d, dj), _det(dj, (p[0] - pj[0], p[1] - pj[1]))
        if abs(den) <= 1e-5:
            if num < 0:
                return None
            continue
        t = num / den
        if den >= 0:
            tr = min(tr, t)
        else:
            tl = max(tl, t)
        if tl > tr:
            return None
    if dir_opt:
        t = tr if opt[0] * d[0] + opt[1] * d[1] > 0 else tl
    else:
        t = min(max(d[0] * (opt[0] - p[0]) + d[1] * (opt[1] - p[1]), tl), tr)
    return (p[0] + t * d[0], p[1] + t * d[1])


def _lp2_64(lines, r, opt, dir_opt):
    if dir_opt:
        res = (opt[0] * r, opt[1] * r)
    elif opt[0] ** 2 + opt[1] ** 2 > r * r:
        n = (opt[0] ** 2 + opt[1] ** 2) ** 0.5
        res = (opt[0] / n * r, opt[1] / n * r)
    else:
        res = opt
    for i, (p, d) in enumerate(lines):
        if _det(d, (p[0] - res[0], p[1] - res[1])) > 0:
            nr = _lp1_64(lines, i, r, opt, dir_opt)
            if nr is None:
                return i, res
            res = nr
    return len(lines), res


def _lp3_64(lines, begin, r, res):
    dist = 0.0
    for i in range(begin, len(lines)):
        p, d = lines[i]
        if _det(d, (p[0] - res[0], p[1] - res[1])) > dist:
            proj = []
            for j in range(i):
                pj, dj = lines[j]
                dt = _det(d, dj)
                if abs(dt) <= 1e-5:
                    if d[0] * dj[0] + d[1] * dj[1] > 0:
                        continue
                    pt = (0.5 * (p[0] + pj[0]), 0.5 * (p[1] + pj[1]))
                else:
                    sc = _det(dj, (p[0] - pj[0], p[1] - pj[1])) / dt
                    pt = (p[0] + sc * d[0], p[1] + sc * d[1])
                dd = (dj[0] - d[0], dj[1] - d[1])
                n = (dd[0] ** 2 + dd[1] ** 2) ** 0.5
                proj.append((pt, (dd[0] / n, dd[1] / n)))
            f, nr = _lp2_64(proj, r, (-d[1], d[0]), True)
            if f == len(proj):
                res = nr
            dist = _det(d, (p[0] - res[0], p[1] - res[1]))
    return res


def test_infeasible_scenes_lp3_minimises_max_violation_and_fp32_tracks_fp64():
    """LP3 (taken when the half-planes are infeasible) minimises the maximum penetration: checked on the fp64
    re-implementation against 400 sampled velocities per scene; the fp32 oracle must track the fp64 result except
    where fp32 cancellation in a far-away projected line trips RVO2's own 'should in principle not happen' fallback
    (`discriminant < 0` in linearProgram1) -- behaviour of the published fp32 algorithm, not of this restatement."""
    rs = np.random.RandomState(5)
    found = close = 0
    for _ in range(300):
        s, o = _scene(rs, 19, 1.0)       # overlapping crowd: LP2 fails, LP3 takes over
        (vx, vy), lines, fail = O.orca_velocity(s, o, want_lines=True)
        if fail == len(lines):
            continue
        found += 1
        assert np.hypot(vx, vy) <= s[5] * (1 + 1e-2) + 1e-5
        L = [((float(l[0]), float(l[1])), (float(l[2]), float(l[3]))) for l in lines]
        r = float(s[5])
        f64_fail, res = _lp2_64(L, r, (float(s[6]), float(s[7])), False)
        assert f64_fail == fail
        res = _lp3_64(L, f64_fail, r, res)
        viol64 = max(_det(d, (p[0] - res[0], p[1] - res[1])) for p, d in L)
        ang = rs.uniform(0, 2 * np.pi, 400); rad = r * np.sqrt(rs.uniform(0, 1, 400))
        for cx, cy in zip(rad * np.cos(ang), rad * np.sin(ang)):
            assert max(_det(d, (p[0] - cx, p[1] - cy)) for p, d in L) >= viol64 - 1e-6
        viol32 = max(_det(d, (p[0] - vx, p[1] - vy)) for p, d in L)
        close += abs(viol32 - viol64) <= 2e-3
    assert found > 50 and close >= 0.93 * found


def test_neighbor_range_and_max_neighbors_filter():
    s = np.array([0, 0, 0, 0, 0.46, 1.0, 1, 0], np.float32)
    near = [1.2, 0.05, -1, 0, 0.46]
    far = [6.0, 0.0, -1, 0, 0.46]
    v_both = O.orca_velocity(s, np.array([near, far], np.float32), neighbor_dist=10.0)
    v_near = O.orca_velocity(s, np.array([near], np.float32), neighbor_dist=10.0)
    v_cut = O.orca_velocity(s, np.array([near, far], np.float32), neighbor_dist=5.0)       # far one out of range
    v_max1 = O.orca_velocity(s, np.array([far, near], np.float32), neighbor_dist=10.0, max_neighbors=1)  # keeps the nearest
    assert v_cut == v_near and v_max1 == v_near
    assert isinstance(v_both[0], float)


@pytest.mark.gpu
def test_hip_orca_has_the_same_properties():
    torch = pytest.importorskip("torch")
    from crowdnav_prediction_attngraph_amd.hip import orca_solve
    rs = np.random.RandomState(1)
    B = 512
    S, Os = zip(*[_scene(rs, 19, 6.0) for _ in range(B)])
    S, Os = np.stack(S), np.stack(Os)
    v = orca_solve(torch.from_numpy(S).cuda(), torch.from_numpy(Os).cuda()).cpu().numpy()
    assert np.all(np.hypot(v[:, 0], v[:, 1]) <= S[:, 5] * (1 + 1e-2) + 1e-5)
    for b in range(B):
        (vx, vy), lines, fail = O.orca_velocity(S[b], Os[b], want_lines=True)
        assert (v[b, 0], v[b, 1]) == (np.float32(vx), np.float32(vy))
