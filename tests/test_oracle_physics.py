"""Closed-form known-answer tests for the oracle's mini-Chipmunk (oracle/cp_lite.c).  The reference has no golden
vectors for the physics (pymunk/Chipmunk2D are not vendored), so these pin the restatement against analytic results any
correct sequential-impulse solver with Chipmunk's e/slop/bias conventions must reproduce."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol

CIRCLE, CAPSULE, BOX = 0, 1, 2


def two_body(a, b, steps):
    l = ol.lib()
    inp = np.array([a, b], np.float64)
    out = np.zeros((2, 6))
    l.oracle_sandbox_two_body.restype = C.c_int
    n = l.oracle_sandbox_two_body(inp.ctypes.data_as(C.c_void_p), steps, out.ctypes.data_as(C.c_void_p))
    return out, n


def body(kind, sx, sy, m, px, py, vx=0.0, vy=0.0, ang=0.0, w=0.0, e=0.0, u=0.0):
    return [kind, sx, sy, m, px, py, vx, vy, ang, w, e, u]


@pytest.fixture(scope="module", autouse=True)
def _built(oracle_built):
    return oracle_built


def test_bias_constants_match_device_header():
    l = ol.lib()
    l.oracle_bias_coef.restype = C.c_double
    assert l.oracle_bias_coef(0) == 0.061259621561307376  # DE_CONTACT_BIAS_COEF in dynenv_amd/csrc/dev_common.h
    assert l.oracle_bias_coef(1) == 0.022762779044189331  # DE_PIVOT_BIAS_COEF


def test_elastic_equal_circles_swap_velocities():
    out, n = two_body(body(CIRCLE, 5, 0, 10, 0, 0, vx=50, e=1.0), body(CIRCLE, 5, 0, 10, 30, 0, vx=0, e=1.0), 80)
    assert n > 0
    np.testing.assert_allclose(out[0, 2], 0.0, atol=1e-9)
    np.testing.assert_allclose(out[1, 2], 50.0, rtol=1e-12)


def test_inelastic_equal_circles_share_velocity_and_conserve_momentum():
    out, n = two_body(body(CIRCLE, 5, 0, 10, 0, 0, vx=50, e=0.0), body(CIRCLE, 5, 0, 30, 30, 0, vx=-10, e=0.0), 60)
    assert n > 0
    p0 = 10 * 50 + 30 * (-10)
    np.testing.assert_allclose(10 * out[0, 2] + 30 * out[1, 2], p0, rtol=1e-12)
    np.testing.assert_allclose(out[0, 2], out[1, 2], atol=1e-9)  # e = e_a*e_b = 0 -> common velocity


def test_restitution_is_product_of_elasticities():
    # approach speed 40, e = 0.5*0.8 -> separation speed 16 for equal masses
    out, _ = two_body(body(CIRCLE, 5, 0, 10, 0, 0, vx=40, e=0.5), body(CIRCLE, 5, 0, 10, 30, 0, e=0.8), 80)
    np.testing.assert_allclose(out[1, 2] - out[0, 2], 0.4 * 40, rtol=1e-9)


def test_box_against_static_wall_rests_at_collision_slop():
    # a box pushed into a wall ends with penetration -> slop (0.1): bias velocity removes anything deeper
    out, n = two_body(body(BOX, 10, 5, 1000, 0, 0, vx=30, e=0.0), body(BOX, 10, 50, -1, 40, 0), 400)
    assert n > 300  # persistent contact
    pen = (out[0, 0] + 10) - (40 - 10)
    assert 0.0 <= pen <= 0.1 + 1e-6
    np.testing.assert_allclose(out[0, 2], 0.0, atol=1e-6)
    np.testing.assert_allclose(out[0, 5], 0.0, atol=1e-9)  # symmetric 2-point manifold: no spin


def test_oblique_box_box_conserves_linear_and_angular_momentum():
    a = body(BOX, 10, 5, 1200, 0, 0, vx=50, vy=10, ang=0.3, e=0.05)
    b = body(BOX, 15, 6, 1800, 40, 12, vx=-30, vy=-5, ang=2.0, e=0.05)
    out, n = two_body(a, b, 60)
    assert n > 0
    px0, py0 = 1200 * 50 + 1800 * -30, 1200 * 10 + 1800 * -5
    np.testing.assert_allclose(1200 * out[0, 2] + 1800 * out[1, 2], px0, rtol=1e-11)
    np.testing.assert_allclose(1200 * out[0, 3] + 1800 * out[1, 3], py0, rtol=1e-11)
    assert abs(out[0, 5]) > 1e-4 or abs(out[1, 5]) > 1e-4  # off-centre hit spins the boxes
    l = ol.lib()
    Ia, Ib = l.oracle_moment_for_box(1200.0, 10.0, 5.0), l.oracle_moment_for_box(1800.0, 15.0, 6.0)
    # total angular momentum about the origin: before (w=0) vs after
    L0 = 1200 * (0 * 10 - 0 * 50) + 1800 * (40 * -5 - 12 * -30)
    L1 = (1200 * (out[0, 0] * out[0, 3] - out[0, 1] * out[0, 2]) + Ia * out[0, 5] +
          1800 * (out[1, 0] * out[1, 3] - out[1, 1] * out[1, 2]) + Ib * out[1, 5])
    # velocity impulses conserve L exactly; Chipmunk's position-correction pseudo-velocities (v_bias) shift the
    # bodies without momentum, which perturbs r x mv at the percent level
    np.testing.assert_allclose(L1, L0, rtol=2e-2)


def test_circle_hits_box_face_and_corner():
    # face hit: pedestrian circle into a parked car box, equal e=0.05 -> e_arb = 0.0025
    out, n = two_body(body(CIRCLE, 5, 0, 90, -30, 0, vx=40, e=0.05), body(BOX, 10, 5, 1200, 0, 0, e=0.05), 80)
    assert n > 0
    np.testing.assert_allclose(90 * out[0, 2] + 1200 * out[1, 2], 90 * 40, rtol=1e-11)
    np.testing.assert_allclose(out[1, 2] - out[0, 2], 0.0025 * 40, rtol=1e-6)
    # corner hit: normal points along the diagonal, so the circle picks up lateral velocity
    out, n = two_body(body(CIRCLE, 5, 0, 90, -22, 9.5, vx=40, e=0.05), body(BOX, 10, 5, 1200, 0, 0, e=0.05), 60)
    assert n > 0 and abs(out[0, 3]) > 1e-3


def test_no_tunnelling_at_car_speeds():
    # 60 px/s = 0.6 px per substep against a 20 px obstacle: the box never passes through
    out, n = two_body(body(BOX, 10, 5, 1200, 0, 0, vx=60, e=0.05), body(BOX, 10, 10, -1, 50, 0, e=0.05), 300)
    assert n > 0 and out[0, 0] + 10 <= 40 + 0.11


def test_capsule_circle_and_capsule_capsule_contacts():
    out, n = two_body(body(CIRCLE, 10, 0, 10, -40, 0, vx=60, e=0.98), body(CAPSULE, 10, 7.5, 4000, 0, 0, ang=np.pi / 2, e=0.3), 60)
    assert n > 0
    np.testing.assert_allclose(10 * out[0, 2] + 4000 * out[1, 2], 600.0, rtol=1e-10)
    assert out[0, 2] < 0  # the light ball bounces back off the heavy foot
    out, n = two_body(body(CAPSULE, 10, 7.5, 4000, -40, 3, vx=50, e=0.3), body(CAPSULE, 10, 7.5, 4000, 0, 0, ang=0.4, e=0.3), 80)
    assert n > 0
    np.testing.assert_allclose(4000 * (out[0, 2] + out[1, 2]), 4000 * 50, rtol=1e-11)


def test_robot_joint_keeps_feet_together():
    """PivotJoint(error_bias=0.1) + RotaryLimitJoint(0,0) of Robot.py:58-60: the two feet move as one body."""
    l = ol.lib()
    out = np.zeros((2, 6))
    l.oracle_sandbox_robot_joint.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p]
    l.oracle_sandbox_robot_joint(100.0, 0.0, 0.0, 50, out.ctypes.data_as(C.c_void_p))
    # momentum of the kicked foot is shared by both (no friction in the sandbox): v -> 50 each
    np.testing.assert_allclose(out[0, 2] + out[1, 2], 100.0, rtol=1e-12)
    # the 1 px the kicked foot moved before the joint reacted is bled off at 1-0.1^dt per step (error_bias=0.1)
    gap = out[0, 0] - out[1, 0]
    np.testing.assert_allclose(gap, (1.0 - 0.022762779044189331) ** 49, rtol=1e-6)
    np.testing.assert_allclose(out[1, 2] - out[0, 2], 0.022762779044189331 * gap / 0.01, rtol=1e-6)
    # quirk C14: a rotary-limit joint with min=max=0 is inactive while the angle error is exactly 0
    l.oracle_sandbox_robot_joint(0.0, 0.0, 20.0, 1, out.ctypes.data_as(C.c_void_p))
    np.testing.assert_allclose(out[0, 4], 0.2, rtol=1e-12)  # left foot integrated w*dt before the joint reacts
    assert out[1, 4] == 0.0
    l.oracle_sandbox_robot_joint(0.0, 0.0, 20.0, 200, out.ctypes.data_as(C.c_void_p))
    np.testing.assert_allclose(out[0, 4], out[1, 4], atol=2e-3)  # ...and afterwards drags the right foot along
    np.testing.assert_allclose(out[0, 5], out[1, 5], atol=1e-3)


# ------------------------------------------------------------------------------------------------------------------
# Round 2: known answers whose contact point, normal and impulse can be derived by hand.  Conventions of the derivations:
# n = contact normal from body a to body b; r_a, r_b = contact point relative to the bodies' centres; approach speed
# s = -(v_b + w_b x r_b - v_a - w_a x r_a) . n > 0; k_n = 1/m_a + 1/m_b + (r_a x n)^2 / I_a + (r_b x n)^2 / I_b; a first contact
# has no warm start and a single constraint converges in the first of the 10 iterations, so after the substep that detects it
# j_n = (1 + e) s / k_n,   v_a -= j_n n / m_a,   w_a -= (r_a x n) j_n / I_a   (b with the opposite sign), e = e_a e_b.
BIAS = 0.061259621561307376  # 1 - collisionBias ** dt, collisionBias = (1 - 0.1) ** 60 (Space defaults), dt = 0.01
SLOP = 0.1


def _first_contact(a, b, limit=400):
    """-> (k, state after k substeps, state after k - 1): k = the substep whose collision phase first has an active arbiter"""
    prev = None
    for k in range(1, limit):
        out, n = two_body(a, b, k)
        if n >= 1:
            assert n == 1
            return k, out, prev
        prev = out
    raise AssertionError("no contact")


def _box_inertia(m, hx, hy):
    return m * ((2 * hx) ** 2 + (2 * hy) ** 2) / 12.0


def test_kat_tilted_box_corner_against_wall_impulse():
    """One vertex of a tilted box meets a static wall's face: n = (1, 0), r_a = the vertex relative to the centre,
    j = (1 + e) v_x / (1/m + (r_a x n)^2 / I);  v_x' = v_x - j / m,  w' = -(r_a x n) j / I  (r_a x n = -r_a.y)."""
    m, hx, hy, th, vx, e = 1000.0, 10.0, 5.0, 0.5, 30.0, 0.6
    a = body(BOX, hx, hy, m, 0.0, 0.0, vx=vx, ang=th, e=e)
    b = body(BOX, 10, 60, -1, 60.0, 0.0, e=1.0)
    k, out, prev = _first_contact(a, b)
    c, s = np.cos(th), np.sin(th)
    r = np.array([hx * c + hy * s, hx * s - hy * c])  # the vertex (hx, -hy) turned by th: the one with the largest x
    assert abs((prev[0, 0] + vx * 0.01 + r[0]) - 50.0) < vx * 0.01 + 1e-9, "the vertex has just crossed the wall's face x = 50"
    I = _box_inertia(m, hx, hy)
    rxn = -r[1]                       # r x (1, 0)
    j = (1 + e) * vx / (1 / m + rxn * rxn / I)
    np.testing.assert_allclose(out[0, 2], vx - j / m, rtol=1e-9)
    np.testing.assert_allclose(out[0, 3], 0.0, atol=1e-9)          # frictionless: nothing tangential
    np.testing.assert_allclose(out[0, 5], -rxn * j / I, rtol=1e-9)


def test_kat_face_on_box_bounces_with_a_two_point_manifold():
    """A box meets a wall face-on: two contacts at the face's end points with r_a = (hx, +-hy) share the impulse, the torques
    cancel: v_x' = -e v_x exactly like a point mass, w' = 0."""
    vx, e = 25.0, 0.5
    k, out, _ = _first_contact(body(BOX, 10, 5, 1200, 0, 0, vx=vx, e=e), body(BOX, 10, 50, -1, 45, 0, e=1.0))
    np.testing.assert_allclose(out[0, 2], -e * vx, rtol=1e-9)
    # (the two contacts are relaxed one after the other: 10 Gauss-Seidel iterations leave the spin at rounding level, not at 0)
    assert abs(out[0, 5]) < 1e-9 and abs(out[0, 3]) < 1e-9


def test_kat_oblique_circles_exchange_normal_velocity_only():
    """Frictionless discs: n = (c_b - c_a) / |c_b - c_a| at the detecting substep, tangential components unchanged,
    v_an' = (m_a v_an + m_b v_bn - m_b e (v_an - v_bn)) / (m_a + m_b),  v_bn' = (m_a v_an + m_b v_bn + m_a e (v_an - v_bn)) / (m_a + m_b)."""
    ma, mb, e = 10.0, 35.0, 0.8 * 0.9
    va, vb = np.array([60.0, 5.0]), np.array([-10.0, 0.0])
    a = body(CIRCLE, 5, 0, ma, 0.0, 0.0, vx=va[0], vy=va[1], e=0.8)
    b = body(CIRCLE, 8, 0, mb, 40.0, 9.0, vx=vb[0], vy=vb[1], e=0.9)
    k, out, prev = _first_contact(a, b)
    pa, pb = prev[0, :2] + va * 0.01, prev[1, :2] + vb * 0.01      # positions the collision phase of substep k saw
    n = (pb - pa) / np.linalg.norm(pb - pa)
    t = np.array([-n[1], n[0]])
    van, vbn = va @ n, vb @ n
    van2 = (ma * van + mb * vbn - mb * e * (van - vbn)) / (ma + mb)
    vbn2 = (ma * van + mb * vbn + ma * e * (van - vbn)) / (ma + mb)
    np.testing.assert_allclose(out[0, 2:4], van2 * n + (va @ t) * t, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(out[1, 2:4], vbn2 * n + (vb @ t) * t, rtol=1e-9, atol=1e-9)
    assert abs(out[0, 5]) < 1e-12 and abs(out[1, 5]) < 1e-12          # central normals: no spin without friction (r x n = rounding)


def test_kat_friction_brings_a_disc_to_rolling_when_the_cone_allows_it():
    """A disc (I = m r^2 / 2) strikes a static face at an angle with a large friction coefficient: the tangential impulse
    j_t = -v_t / (1/m + r^2 / I) = -m v_t / 3 is inside the cone u j_n, so the contact point stops sliding:
    v_t' = 2 v_t / 3 and w' = -+ v_t' / r (the sign that makes the contact point's tangential velocity zero)."""
    m, r, vn, vt = 10.0, 5.0, 40.0, 9.0
    a = body(CIRCLE, r, 0, m, 0.0, 0.0, vx=vn, vy=vt, e=0.0, u=10.0)
    k, out, _ = _first_contact(a, body(BOX, 10, 200, -1, 30.0, 0.0, e=1.0, u=1.0))
    np.testing.assert_allclose(out[0, 2], 0.0, atol=1e-9)               # e = 0
    np.testing.assert_allclose(out[0, 3], 2.0 * vt / 3.0, rtol=1e-9)
    np.testing.assert_allclose(abs(out[0, 5]) * r, 2.0 * vt / 3.0, rtol=1e-9)
    # the contact point (centre + r n) no longer slides: v_t + w r = 0 with n = (1, 0)
    np.testing.assert_allclose(out[0, 3] + out[0, 5] * r, 0.0, atol=1e-9)


def test_kat_friction_is_clamped_to_the_coulomb_cone():
    """Same strike with a small coefficient: |j_t| = u j_n with u = u_a u_b and j_n = (1 + e) m v_n:
    v_t' = v_t - u (1 + e) v_n,  |w'| = r u j_n / I = 2 u (1 + e) v_n / r."""
    m, r, vn, vt, e, ua, ub = 10.0, 5.0, 40.0, 30.0, 0.5, 0.4, 0.25
    a = body(CIRCLE, r, 0, m, 0.0, 0.0, vx=vn, vy=vt, e=e, u=ua)
    k, out, _ = _first_contact(a, body(BOX, 10, 200, -1, 30.0, 0.0, e=1.0, u=ub))
    u = ua * ub
    assert u * (1 + e) * m * vn < m * vt / 3.0, "the cone must be the binding limit in this scene"
    np.testing.assert_allclose(out[0, 2], -e * vn, rtol=1e-9)
    np.testing.assert_allclose(out[0, 3], vt - u * (1 + e) * vn, rtol=1e-9)
    np.testing.assert_allclose(abs(out[0, 5]), 2.0 * u * (1 + e) * vn / r, rtol=1e-9)


def test_kat_capsule_end_cap_head_on():
    """A disc runs into the rounded end of a capsule along the capsule's axis: the contact is disc-vs-end-circle, the normal
    passes through both centres of mass, so it is the 1-D two-body formula with e = e_a e_b and no rotation."""
    ma, mb, e = 10.0, 4000.0, 0.98 * 0.3
    v = 80.0
    a = body(CIRCLE, 10, 0, ma, -60.0, 0.0, vx=v, e=0.98)
    b = body(CAPSULE, 10, 7.5, mb, 0.0, 0.0, e=0.3)
    k, out, prev = _first_contact(a, b)
    assert abs((prev[0, 0] + v * 0.01) - (-10.0 - 7.5 - 10.0)) < v * 0.01 + 1e-9   # end point -10, cap radius 7.5, disc radius 10
    va2 = (ma * v - mb * e * v) / (ma + mb)
    vb2 = (ma * v + ma * e * v) / (ma + mb)
    np.testing.assert_allclose(out[0, 2], va2, rtol=1e-9)
    np.testing.assert_allclose(out[1, 2], vb2, rtol=1e-9)
    assert abs(out[1, 5]) < 1e-12 and abs(out[0, 3]) < 1e-12


def test_kat_parallel_capsules_touch_with_two_points_and_do_not_spin():
    """Two parallel capsules meet side on (offset along the axis by less than their length): a two-point manifold, normals
    through neither centre but symmetric impulses - with equal masses and e = e_a e_b the normal velocities follow the 1-D
    formula and the pair's total angular momentum about the common centre stays zero."""
    m, e, v = 4000.0, 0.3 * 0.3, 50.0
    a = body(CAPSULE, 10, 7.5, m, 0.0, -40.0, vy=v, e=0.3)
    b = body(CAPSULE, 10, 7.5, m, 0.0, 0.0, e=0.3)
    k, out, _ = _first_contact(a, b)
    np.testing.assert_allclose(out[0, 3], (1 - e) * v / 2, rtol=1e-9)
    np.testing.assert_allclose(out[1, 3], (1 + e) * v / 2, rtol=1e-9)
    assert abs(out[0, 5]) < 1e-9 and abs(out[1, 5]) < 1e-9


def test_kat_elastic_boxes_swap_velocities():
    """Equal boxes, face to face, e = 1: they swap their velocities (two-point manifold, no spin)."""
    k, out, _ = _first_contact(body(BOX, 10, 5, 1200, 0, 0, vx=40, e=1.0), body(BOX, 10, 5, 1200, 50, 0, vx=-5, e=1.0))
    np.testing.assert_allclose(out[0, 2], -5.0, rtol=1e-9)
    np.testing.assert_allclose(out[1, 2], 40.0, rtol=1e-9)
    assert abs(out[0, 5]) < 1e-9 and abs(out[1, 5]) < 1e-9


def test_kat_penetration_decays_geometrically_to_the_slop():
    """A box at rest overlapping a static wall: only the position correction acts - bias velocity = biasCoef (pen - slop) / dt,
    applied by the next position update - so the depth follows pen_k - slop = (pen_0 - slop) (1 - biasCoef)^k (k = position
    updates after the detecting substep), with biasCoef = 1 - collisionBias^dt = 0.06126 and slop = 0.1."""
    pen0 = 2.0
    a = body(BOX, 10, 5, 1200, 0.0, 0.0, e=0.0)
    b = body(BOX, 10, 50, -1, 20.0 - pen0, 0.0)
    for steps in (1, 2, 5, 20, 60):
        out, n = two_body(a, b, steps)
        assert n == steps
        pen = (out[0, 0] + 10.0) - (20.0 - pen0 - 10.0)
        np.testing.assert_allclose(pen - SLOP, (pen0 - SLOP) * (1.0 - BIAS) ** (steps - 1), rtol=1e-9)
        assert out[0, 2] == 0.0 and out[0, 5] == 0.0   # the real velocities never move


def test_kat_circle_on_box_corner_normal_and_impulse():
    """A disc meets a free box at a corner: n = (corner - centre_disc) / |...| at the detecting substep, r_box = corner - centre_box,
    j = (1 + e) s / (1/m_a + 1/m_b + (r_b x n)^2 / I_b);  the disc's velocity changes along n only, the box also spins."""
    ma, mb, hx, hy, e = 90.0, 1200.0, 10.0, 5.0, 0.05 * 0.05
    va = np.array([40.0, -12.0])
    a = body(CIRCLE, 5, 0, ma, -21.0, 12.0, vx=va[0], vy=va[1], e=0.05)
    b = body(BOX, hx, hy, mb, 0.0, 0.0, e=0.05)
    k, out, prev = _first_contact(a, b)
    ca = prev[0, :2] + va * 0.01
    corner = np.array([-hx, hy])
    assert ca[0] < -hx and ca[1] > hy, "the disc's centre is in the corner's Voronoi region"
    n = (corner - ca) / np.linalg.norm(corner - ca)
    rb = corner
    rbxn = rb[0] * n[1] - rb[1] * n[0]
    Ib = _box_inertia(mb, hx, hy)
    s = va @ n
    j = (1 + e) * s / (1 / ma + 1 / mb + rbxn * rbxn / Ib)
    np.testing.assert_allclose(out[0, 2:4], va - j * n / ma, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(out[1, 2:4], j * n / mb, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(out[1, 5], rbxn * j / Ib, rtol=1e-9)


def test_kat_rotary_limit_joint_is_one_sided():
    """RotaryLimitJoint(a, b, 0, 0) (Robot.py:60) between two free bodies, no friction: in the substep that first sees an angle
    error d_1 = 0.2 the solve sets the relative angular velocity to bias = biasCoef d_1 / dt (default errorBias (1 - 0.1)^60:
    biasCoef 0.06126 - the pivot's own error_bias = 0.1 is a different number).  A LIMIT only ever pushes towards the range:
    in the following substeps less closing speed would be asked for (the error has shrunk), the accumulated impulse clamps
    at zero, and the feet coast at that velocity - the error falls LINEARLY, d_k = d_1 (1 - (k - 1) biasCoef), not
    geometrically, until it is used up."""
    l = ol.lib()
    out = np.zeros((2, 6))
    l.oracle_sandbox_robot_joint.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p]
    for k in (2, 3, 10, 16):
        l.oracle_sandbox_robot_joint(0.0, 0.0, 20.0, k, out.ctypes.data_as(C.c_void_p))
        np.testing.assert_allclose(out[0, 4] - out[1, 4], 0.2 * (1.0 - (k - 1) * BIAS), rtol=1e-9)
        np.testing.assert_allclose(out[1, 5] - out[0, 5], BIAS * 0.2 / 0.01, rtol=1e-9)
        # equal inertias and no friction in the sandbox: the angular momentum I * 20 is shared, the angles' sum grows by 0.2 a step
        np.testing.assert_allclose(out[0, 4] + out[1, 4], 0.2 * k, rtol=1e-12)
