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
