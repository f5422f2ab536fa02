// robocup_rpl.hip — the RoboCup step with the common substep in REGISTERS (included by robocup_kernels.hip).
//
// One lane per ROBOT: lane r < R of the environment's lane group (Grp<EPW>: the wave, or a half wave with two environments per
// wave) owns robot r with BOTH its feet (bodies 2r, 2r+1), lane 10 owns the ball (in the registers of "foot 0"), lanes 16..25
// lend a hand with the second foot's sincos.  Everything a robot's tick, its feet's position and
// velocity updates and its two joints touch then lives in that lane's registers for the whole step: the common substep
// (no cross-robot event in the game logic, no contact candidate, ball inside the field - ~95 % of all substeps) runs
// without an LDS round trip for the state; what crosses lanes is the ball's position, the per-robot distances of the
// "closest robot" searches and the bounding boxes of the contact prefilter.  Measured motivation (tools/probe/exec_probe.hip):
// a wave64 VALU instruction costs a lone gfx950 wave the same whatever the number of active lanes (6.5 cycles fp64, 3.75 fp32),
// a dependent LDS round trip ~110 cycles; the foot-per-lane kernel spent 37 % of a wave's life in s_waitcnt.
//
// Every rare case flushes the registers to the LDS tile, runs the proven foot-per-lane code of robocup_kernels.hip on it
// (rc_game_logic<2>: serial ticks / processAction with the fall dice; rc_physics_inl<2>: contacts, the general solve,
// callbacks) and reloads.  The arithmetic per body / robot / joint is the same expression sequence as there (oracle:
// oracle/robocup.c, cp_lite.c), so results stay bit-identical.
// One dynamic body in registers.  Not here, because they are zero in every common substep: the forces (only fall() applies any)
// and the bias velocities (only the contact solver produces any) - see `hasForce` / `vbLive` in rc_step_rpl_body.
struct RplBody {
  double px, py, vx, vy, a, w;
  double rc, rs;  // (cos, sin) of a ...
  bool rv;        // ... if valid (a changes in the position update only: a teleport goes through a flush / reload)
};
struct RplRobot {
  double head, headmov, moveT, prevx, prevy, fallT, penalT, jx, jy, jr, rrew, rposrew;
  int flags, fallc;
};
struct RplEnv {  // per-environment scalars, kept identically by every lane of the half wave
  double bprevx, bprevy, grace, freecnt, team0, team1;
  int owned, close0, close1, def0, def1, pivfirst, nlk, lk;  // lk: the lastKicked list, 8 bits per entry
};

DE_DEV void rpl_load_body(const RcLds& L, int i, RplBody& b) {
  b.px = L.px[i]; b.py = L.py[i]; b.vx = L.vx[i]; b.vy = L.vy[i]; b.a = L.ang[i]; b.w = L.w[i];
  b.rc = 1.0; b.rs = 0.0; b.rv = false;
}
// cacheToo: a position update ran on the registers since the last flush (it zeroed the bias velocities and refreshed the pose
// the shapes were last indexed at); the forces in LDS are only ever written by fall() and consumed by the LDS velocity update
DE_DEV void rpl_store_body(RcLds& L, int i, const RplBody& b, bool cacheToo, bool isFoot) {
  L.px[i] = b.px; L.py[i] = b.py; L.vx[i] = b.vx; L.vy[i] = b.vy; L.ang[i] = b.a; L.w[i] = b.w;
  if (cacheToo) {
    L.vbx[i] = 0.0; L.vby[i] = 0.0; L.wb[i] = 0.0;  // the shape cache = the pose at the last position update (done in registers since the last flush)
    L.cpx[i] = b.px; L.cpy[i] = b.py;
    if (isFoot) { L.crc[i] = b.rc; L.crs[i] = b.rs; }
  }
}
// The robot's own fields: persistent registers where the budget allows (256 VGPRs at two environments per wave); at 128 VGPRs
// they stay in LDS and visit registers for the tick (`logic` fields) resp. the joint solve (`joint` fields) only.
#define RPL_ROBOT_REGS(EPW) ((EPW) == 2)
DE_DEV void rpl_load_robot(const RcLds& L, int lane, int R, RplRobot& r, bool logic, bool joint) {
  const int q = lane < R ? lane : 0;
  r.flags = L.rflags[q];
  if (logic) {
    r.head = L.head[q]; r.headmov = L.headmov[q]; r.moveT = L.moveT[q]; r.prevx = L.prevx[q]; r.prevy = L.prevy[q];
    r.fallT = L.fallT[q]; r.penalT = L.penalT[q]; r.rrew = L.rrew[q]; r.rposrew = L.rposrew[q]; r.fallc = L.fallc[q];
  }
  if (joint) { r.jx = L.jx[q]; r.jy = L.jy[q]; r.jr = L.jrot[q]; }
}
DE_DEV void rpl_store_robot(RcLds& L, int lane, int R, const RplRobot& r, bool logic, bool joint) {
  if (lane < R) {
    if (logic) {
      L.head[lane] = r.head; L.headmov[lane] = r.headmov; L.moveT[lane] = r.moveT; L.prevx[lane] = r.prevx; L.prevy[lane] = r.prevy;
      L.fallT[lane] = r.fallT; L.penalT[lane] = r.penalT; L.rrew[lane] = r.rrew; L.rposrew[lane] = r.rposrew;
      L.rflags[lane] = r.flags; L.fallc[lane] = r.fallc;
    }
    if (joint) { L.jx[lane] = r.jx; L.jy[lane] = r.jy; L.jrot[lane] = r.jr; }
  }
}
template <int EPW>
DE_DEV void rpl_reload(const RcLds& L, int lane, int R, RplBody& b0, RplBody& b1, RplRobot& r, RplEnv& v) {
  typedef Grp<EPW> G;
  const bool isRobot = lane < R, isBall = lane == 10;
  const int i0 = isBall ? RC_BALL : (isRobot ? 2 * lane : 0);
  rpl_load_body(L, i0, b0);
  rpl_load_body(L, isRobot ? 2 * lane + 1 : 0, b1);
  if (RPL_ROBOT_REGS(EPW)) rpl_load_robot(L, lane, R, r, true, true);
  v.bprevx = L.envd[RD_BPREVX]; v.bprevy = L.envd[RD_BPREVY]; v.grace = L.envd[RD_GRACE]; v.freecnt = L.envd[RD_FREECNT];
  v.team0 = L.teamRew[0]; v.team1 = L.teamRew[1];
  // (the integer scalars are uniform over the group: scalar registers when the group is the wave)
  v.owned = G::uniform_i(L.envi[RE_OWNED]); v.close0 = G::uniform_i(L.envi[RE_CLOSE0]); v.close1 = G::uniform_i(L.envi[RE_CLOSE1]);
  v.def0 = G::uniform_i(L.envi[RE_DEF0]); v.def1 = G::uniform_i(L.envi[RE_DEF1]); v.pivfirst = G::uniform_i(L.envi[RE_PIVFIRST]);
  v.nlk = G::uniform_i(L.envi[RE_NLK]);
  v.lk = G::uniform_i((L.envi[RE_LK0] & 0xFF) | ((L.envi[RE_LK1] & 0xFF) << 8) | ((L.envi[RE_LK2] & 0xFF) << 16) | ((L.envi[RE_LK3] & 0xFF) << 24));
}
template <int EPW>
DE_DEV void rpl_flush(RcLds& L, int lane, int R, const RplBody& b0, const RplBody& b1, const RplRobot& r, const RplEnv& v, bool cacheToo) {
  const bool isRobot = lane < R, isBall = lane == 10;
  if (isRobot) {
    rpl_store_body(L, 2 * lane, b0, cacheToo, true);
    rpl_store_body(L, 2 * lane + 1, b1, cacheToo, true);
    if (RPL_ROBOT_REGS(EPW)) rpl_store_robot(L, lane, R, r, true, true);
  }
  if (isBall) rpl_store_body(L, RC_BALL, b0, cacheToo, false);
  if (lane == 0) {
    L.envd[RD_BPREVX] = v.bprevx; L.envd[RD_BPREVY] = v.bprevy; L.envd[RD_GRACE] = v.grace; L.envd[RD_FREECNT] = v.freecnt;
    L.teamRew[0] = v.team0; L.teamRew[1] = v.team1;
    L.envi[RE_OWNED] = v.owned; L.envi[RE_CLOSE0] = v.close0; L.envi[RE_CLOSE1] = v.close1;
  }
}

// cpBodyUpdatePosition (rc_step_body: "cpBodyUpdatePosition + shape cache"); returns "the rotation must be recomputed"
DE_DEV bool rpl_update_position(RplBody& b, double vbx, double vby, double wb) {
  b.px = b.px + (b.vx + vbx) * DE_DT;
  b.py = b.py + (b.vy + vby) * DE_DT;
  const double na = b.a + (b.w + wb) * DE_DT;
  const bool stale = na != b.a || !b.rv;
  b.a = na;
  return stale;
}
// rc_velocity_update (cutils.py:102-140 apply_friction incl. Body.update_velocity) on registers; the per-lane constants
// (foot or ball) are passed in
struct RplFric {
  double minv, iinv, factor, rotFactor, spin;
};
DE_DEV void rpl_velocity_update(RplBody& b, const RplFric& k) {  // with zero forces (see RplBody)
  double vx = b.vx, vy = b.vy, w = b.w;
  // Body.update_velocity with f = t = +0 and the positive finite 1/m, 1/i: v * 1.0 + (0.0 + 0.0 * m_inv) * dt is v + (+0)
  // (which turns a -0 into +0 like the full expression does)
  vx = vx + 0.0;
  vy = vy + 0.0;
  w = w + 0.0;
  double x = vx, y = vy;
  const double length = 1.0 / (dm_abs(x) + dm_abs(y) + 1e-5);
  double theta = w;
  double a0 = x * k.factor * length;
  double a1 = y * k.factor * length;
  a0 += a1 * k.spin * theta;
  a1 -= a0 * k.spin * theta;
  if (dm_abs(x) < k.factor) x = 0.0; else x -= a0;
  if (dm_abs(y) < k.factor) y = 0.0; else y -= a1;
  if (dm_abs(theta) < k.rotFactor) theta = 0.0; else theta -= (theta > 0.0 ? k.rotFactor : -k.rotFactor);
  b.vx = x; b.vy = y; b.w = theta;
}
DE_DEV void rpl_seg_world(const RplBody& b, bool right, SegW& o) {  // seg_world() from the registers (= the shape cache right after the position update)
  const double ly = right ? -10.0 : 10.0;
  const double c = b.rc, sn = b.rs, x = b.px, y = b.py;
  o.ta = v2(c * -10.0 - sn * ly + x, sn * -10.0 + c * ly + y);
  o.tb = v2(c * 10.0 - sn * ly + x, sn * 10.0 + c * ly + y);
  o.tn = v2(c * 0.0 - sn * -1.0, sn * 0.0 + c * -1.0);
}
DE_DEV V2 rpl_robot_pos(const RplBody& b0, const RplBody& b1) { return v2((b0.px + b1.px) / 2.0, (b0.py + b1.py) / 2.0); }

// rc_tick_has_event() + "the ball left the field" from the registers
DE_DEV bool rpl_tick_has_event(const RplBody& b0, const RplBody& b1, const RplRobot& r, const RplEnv& v, int lane) {
  const double time = RC_TIME;
  const int f = r.flags;
  bool ev = false;
  if (r.moveT > 0.0 && (f & RF_KICK)) {
    const double mt = r.moveT - time;
    if ((mt + time > 500.0 && mt <= 500.0 && !(f & RF_JREM)) || mt <= 300.0) ev = true;
  }
  if ((f & RF_FALLEN) && r.fallT - time < 0.0) ev = true;
  const V2 p = rpl_robot_pos(b0, b1);
  if (f & RF_PENAL) {
    if (r.penalT - time <= 0.0) ev = true;
  } else {
    const int teamIdx = (f & RF_TEAMPOS) ? 0 : 1;
    const double robX = teamIdx ? RC_W - p.x : p.x;
    const double penX = RC_SIDE + 60.0 + 5.0 / 2.0;
    const bool isDef = ((teamIdx ? v.def1 : v.def0) & (1 << lane)) != 0;
    const bool inArea = robX < penX && p.y > (RC_H / 2.0 - 110.0) && p.y < (RC_H / 2.0 + 110.0);
    if (inArea != isDef) ev = true;
  }
  if (p.y < 0.0 || p.x < 0.0 || p.y > RC_H || p.x > RC_W) ev = true;
  return ev;
}
// rc_tick() for a robot without an event, on registers (RoboCupEnvironment.py:862-1007; every branch that rc_tick_has_event
// does not flag: timers, head, the kick's forward / backward foot velocities, the stop at the end of a move, the
// approach-the-ball reward)
DE_DEV void rpl_tick(RplBody& b0, RplBody& b1, RplRobot& r, const RplEnv& v, int lane, V2 ballPos) {
  const double time = RC_TIME;
  if (r.moveT > 0.0) {
    r.moveT -= time;
    if (r.headmov != 0.0) {
      const double h = r.head + r.headmov;
      r.head = dm_max(-ROBOT_HEAD_MAX, dm_min(ROBOT_HEAD_MAX, h));
    }
    const int f = r.flags;
    if (f & RF_KICK) {
      const bool right = (f & RF_FOOT) != 0;
      const double mt = r.moveT;
      const bool fwd = mt + time > 500.0 && mt <= 500.0, back = mt + time > 400.0 && mt <= 400.0;
      if (fwd || back) {  // (the joint's removal at the 500 ms mark and the end of the kick are events: not here)
        const DevSC sc = dev_sincos(right ? b1.a : b0.a);
        double nvx, nvy;
        if (fwd) { const double vxl = ROBOT_VELOCITY * 3.0; nvx = vxl * sc.c - 0.0 * sc.s; nvy = vxl * sc.s + 0.0 * sc.c; }
        else { const double vxl = ROBOT_VELOCITY * 2.5; nvx = -(vxl * sc.c - 0.0 * sc.s); nvy = -(vxl * sc.s + 0.0 * sc.c); }
        if (right) { b1.vx = nvx; b1.vy = nvy; } else { b0.vx = nvx; b0.vy = nvy; }
      }
    }
    if (r.moveT <= 0.0) {
      r.moveT = 0.0; r.headmov = 0.0;
      b0.vx = 0.0; b0.vy = 0.0; b0.w = 0.0;
      b1.vx = 0.0; b1.vy = 0.0; b1.w = 0.0;
    }
  }
  if (r.flags & RF_FALLEN) r.fallT -= time;  // (running out is an event)
  if (r.flags & RF_PENAL) r.penalT -= time;  // (likewise; the defender bookkeeping of the other branch changes nothing without an event)
  const V2 pos = rpl_robot_pos(b0, b1);
  if (pos.x != r.prevx || pos.y != r.prevy) {
    if ((lane == v.close0 || lane == v.close1) && !(r.flags & RF_PENAL)) {
      const double diff = vlen(vsub(pos, ballPos)) - vlen(vsub(v2(r.prevx, r.prevy), ballPos));
      r.rrew -= diff * 0.05;
      r.rposrew += dm_max(0.0, -diff * 0.05);
    }
    r.prevx = pos.x; r.prevy = pos.y;
  }
}
DE_DEV bool rpl_ball_outside(V2 pos) {
  const double outMin = RC_SIDE - 5.0, outMaxX = RC_W - RC_SIDE + 5.0, outMaxY = RC_H - RC_SIDE + 5.0;
  return pos.y < outMin || pos.x < outMin || pos.y > outMaxY || pos.x > outMaxX;
}
// rc_ball_logic() for a ball inside the field, on registers: the progress reward with its per-robot shares, the free-kick
// counters, the closest robot of each team.  `L.u.rq.q` carries the per-robot squared distances across the lanes.
template <int EPW>
DE_DEV void rpl_ball_logic(const RcCtx& c, RcLds& L, int lane, const RplBody& b0, const RplBody& b1, RplRobot& r, RplEnv& v, V2 pos) {
  typedef Grp<EPW> G;
  const int n = c.n;
  double cr0 = 0.0, cr1 = 0.0;
  {
    const double dx = pos.x - v.bprevx;
    const double d = dx == 0.0 ? dx : dx / 20.0;
    cr0 += d;
    cr1 -= d;
  }
  const V2 rp = rpl_robot_pos(b0, b1);
  const bool anyTeamTerm = !(cr0 == 0.0 && cr1 == 0.0);
  if (anyTeamTerm && lane < c.R) {
    bool inLk = false;
    double disc = 1.0;
    for (int i = 0; i < v.nlk; ++i) {
      if (((v.lk >> (8 * i)) & 0xFF) == lane) {
        inLk = true;
        const double rew = (lane < n ? cr0 : cr1) * disc;
        r.rrew += rew;
        r.rposrew += dm_max(0.0, rew);
      }
      disc *= 0.5;
    }
    const bool cond1 = (lane == v.close0 || lane == v.close1);
    const bool cond2 = vlen(vsub(rp, pos)) < 150.0;
    if ((cond1 || cond2) && !inLk) r.rrew += dm_min(0.0, (lane < n ? cr0 : cr1) * 0.5);
  }
  // closest robot of each team: the reference's ascending strict-< loops over all robots' distances
  if (lane < c.R) {
    const V2 d = vsub(pos, rp);
    L.u.rq.q[lane] = d.x * d.x + d.y * d.y;
  }
  __syncthreads();
  int best0 = 0, best1 = 0;
  double d0 = INFINITY, d1 = INFINITY;
#pragma unroll
  for (int i = 0; i < RC_MAXR / 2; ++i) {
    if (i < n) {
      const double qa = L.u.rq.q[i], qb = L.u.rq.q[n + i];
      if (qa < d0) { d0 = qa; best0 = i; }
      if (qb < d1) { d1 = qb; best1 = i; }
    }
  }
  // ballFreeKickProcess(0) (:600-619), the progress bookkeeping and the team rewards (every lane keeps the same copies)
  if (v.grace > 0.0) {
    v.grace -= RC_TIME;
    if (v.grace < 0.0) { v.grace = 0.0; v.freecnt = 9999.0; }
  } else if (v.freecnt > 0.0) {
    v.freecnt -= RC_TIME;
    if (v.freecnt < 0.0) { v.freecnt = 0.0; v.owned = 0; }  // (uniform: every lane holds the same counters)
  }
  v.bprevx = pos.x; v.bprevy = pos.y;
  v.team0 += cr0 * 0.1;
  v.team1 += cr1 * 0.1;
  v.close0 = G::uniform_i(best0);
  v.close1 = G::uniform_i(n + best1);
}
// the two joints of my robot: joint_prestep() + joints_solve() (cpPivotJoint / cpRotaryLimitJoint) on registers
DE_DEV void rpl_joints(RplBody& b0, RplBody& b1, RplRobot& r, const RplEnv& v, int lane) {
  RcJoint J;
  J.hasPivot = !(r.flags & RF_JREM);
  J.pivotFirst = (v.pivfirst >> lane) & 1;
  double jx = r.jx, jy = r.jy, jr = r.jr;
  J.m = RC.footMinv; J.i = RC.footIinv;
  J.kk0 = J.kk1 = J.kk2 = J.kk3 = 0.0; J.pbx = J.pby = 0.0;
  if (J.hasPivot) {
    J.kk0 = RC.jkk0; J.kk1 = RC.jkk1; J.kk2 = RC.jkk2; J.kk3 = RC.jkk3;
    const V2 pr1 = v2(0.0, 0.0), pr2 = v2(0.0, 0.0);
    const V2 delta = vsub(vadd(v2(b1.px, b1.py), pr2), vadd(v2(b0.px, b0.py), pr1));
    J.pbx = delta.x * (-DE_PIVOT_BIAS_COEF / DE_DT); J.pby = delta.y * (-DE_PIVOT_BIAS_COEF / DE_DT);
  }
  {
    const double dist = b1.a - b0.a;
    double pdist = 0.0;
    if (dist > 0.0) pdist = 0.0 - dist; else if (dist < 0.0) pdist = 0.0 - dist;
    J.iSum = RC.jiSum;
    J.rbias = -DE_JOINT_BIAS_COEF * pdist / DE_DT;
    if (J.rbias == 0.0) jr = 0.0;
  }
  RcFeet f;
  f.vx0 = b0.vx; f.vy0 = b0.vy; f.w0 = b0.w; f.vx1 = b1.vx; f.vy1 = b1.vy; f.w1 = b1.w;
  const double jx0 = jx, jy0 = jy, jr0 = jr;
  bool clean = feet_clean(f) && is_finite(jx) && is_finite(jy);
  if (clean) {
    joints_solve<true>(J, f, jx, jy, jr);
    clean = is_finite(jx) && is_finite(jy) && is_finite(f.vx0) && is_finite(f.vy0) && is_finite(f.vx1) && is_finite(f.vy1) &&
            is_finite(f.w0) && is_finite(f.w1);
    if (!clean) {  // never seen: redo from the unchanged inputs with the reference's full arithmetic
      jx = jx0; jy = jy0; jr = jr0;
      f.vx0 = b0.vx; f.vy0 = b0.vy; f.w0 = b0.w; f.vx1 = b1.vx; f.vy1 = b1.vy; f.w1 = b1.w;
    }
  }
  if (!clean) joints_solve<false>(J, f, jx, jy, jr);
  b0.vx = f.vx0; b0.vy = f.vy0; b0.w = f.w0; b1.vx = f.vx1; b1.vy = f.vy1; b1.w = f.w1;
  r.jx = jx; r.jy = jy; r.jr = jr;
}
// Does the broadphase find a candidate pair besides the robots' own feet pairs?  Two tiers, the second one exact:
//  1. boxes around whole robots: every shape of robot i lies in the box of half extent R_i = 21.65 + |p_left - p_right|_1 / 2
//     (+ margin) around the robot's position (capsule end points at local (+-10, +-10), radius 7.5: 14.15 + 7.5 from the
//     foot's body position, which is half the feet's separation from the robot's), the ball / a goalpost in a box of half
//     extent 10 (+ margin).  Entities whose boxes do not overlap have no shapes with overlapping bounding boxes.
//  2. for the entity pairs tier 1 could not separate: the broadphase's own test (cpBBIntersects of the shapes' exact bounding
//     boxes, as rc_step_body computes them) for each of their shape pairs.
// The answer is the one the foot-per-lane broadphase gives (its fp32 prefilter only ever rejects pairs the exact test rejects).
struct RplBox {
  double l, b, r, t;
};
DE_DEV RplBox rpl_seg_box(const SegW& s) {
  RplBox o;
  if (s.ta.x < s.tb.x) { o.l = s.ta.x; o.r = s.tb.x; } else { o.l = s.tb.x; o.r = s.ta.x; }
  if (s.ta.y < s.tb.y) { o.b = s.ta.y; o.t = s.tb.y; } else { o.b = s.tb.y; o.t = s.ta.y; }
  o.l = o.l - FOOT_RADIUS; o.b = o.b - FOOT_RADIUS; o.r = o.r + FOOT_RADIUS; o.t = o.t + FOOT_RADIUS;
  return o;
}
DE_DEV bool rpl_box_hit(const RplBox& a, double bl, double bb, double br, double bt) { return a.l <= br && bl <= a.r && a.b <= bt && bb <= a.t; }
template <int EPW>
DE_DEV bool rpl_no_candidates(RcLds& L, int lane, int R, const RplBody& b0, const RplBody& b1, const SegW& s1, const SegW& s2) {
  typedef Grp<EPW> G;
  const bool isRobot = lane < R, isBall = lane == 10;
  float cx = 0.0f, cy = 0.0f, rad = 0.0f;
  if (isRobot) {
    cx = (float)(0.5 * (b0.px + b1.px)); cy = (float)(0.5 * (b0.py + b1.py));
    rad = 21.65f + 0.5f * (float)(dm_abs(b0.px - b1.px) + dm_abs(b0.py - b1.py)) + 1.0f;
  } else if (isBall) {
    cx = (float)b0.px; cy = (float)b0.py; rad = 11.0f;
  }
  if (isRobot || isBall) { L.u.rq.bx[lane] = cx; L.u.rq.by[lane] = cy; L.u.rq.br[lane] = rad; }
  __syncthreads();
  int nearMask = 0;  // bit j < 10: robot j, bit 10: the ball, bits 11..14: the goalposts
  if (isRobot || isBall) {
#pragma unroll
    for (int j = 0; j <= 10; ++j) {
      if (j < R || j == 10) {
        const float dx = __builtin_fabsf(cx - L.u.rq.bx[j]), dy = __builtin_fabsf(cy - L.u.rq.by[j]), rr = rad + L.u.rq.br[j];
        if (j != lane && dx <= rr && dy <= rr) nearMask |= 1 << j;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const V2 pc = post_pos(RC_POST + k);
      const float dx = __builtin_fabsf(cx - (float)pc.x), dy = __builtin_fabsf(cy - (float)pc.y), rr = rad + 11.0f;
      if (dx <= rr && dy <= rr) nearMask |= 1 << (11 + k);
    }
  }
  if (G::ballot(nearMask != 0) == 0ull) return true;
  // tier 2 (rare): the shapes' bounding boxes through LDS, then the exact test for my unresolved entity pairs
  RplBox m0, m1;
  m0.l = m0.b = m0.r = m0.t = 0.0; m1 = m0;
  if (isRobot) {
    m0 = rpl_seg_box(s1); m1 = rpl_seg_box(s2);
    L.aabb[2 * lane][0] = m0.l; L.aabb[2 * lane][1] = m0.b; L.aabb[2 * lane][2] = m0.r; L.aabb[2 * lane][3] = m0.t;
    L.aabb[2 * lane + 1][0] = m1.l; L.aabb[2 * lane + 1][1] = m1.b; L.aabb[2 * lane + 1][2] = m1.r; L.aabb[2 * lane + 1][3] = m1.t;
  } else if (isBall) {
    m0.l = b0.px - BALL_R; m0.b = b0.py - BALL_R; m0.r = b0.px + BALL_R; m0.t = b0.py + BALL_R;
    L.aabb[RC_BALL][0] = m0.l; L.aabb[RC_BALL][1] = m0.b; L.aabb[RC_BALL][2] = m0.r; L.aabb[RC_BALL][3] = m0.t;
  }
  __syncthreads();
  bool hit = false;
  for (int mm = nearMask; mm; mm &= mm - 1) {
    const int j = __builtin_ctz(mm);
    for (int k = 0; k < 2; ++k) {  // the other entity's shapes: a robot's two feet, the ball, a goalpost
      if (k == 1 && j >= 10) break;
      double bl, bb, br, bt;
      if (j > 10) { const V2 pc = post_pos(RC_POST + (j - 11)); bl = pc.x - POST_R; bb = pc.y - POST_R; br = pc.x + POST_R; bt = pc.y + POST_R; }
      else { const int sh = j == 10 ? RC_BALL : 2 * j + k; bl = L.aabb[sh][0]; bb = L.aabb[sh][1]; br = L.aabb[sh][2]; bt = L.aabb[sh][3]; }
      hit = hit || rpl_box_hit(m0, bl, bb, br, bt) || (isRobot && rpl_box_hit(m1, bl, bb, br, bt));
    }
  }
  return G::ballot(hit) == 0ull;
}

// Full observation of one snapshot from the registers (rc_write_obs' staging, then its output loop)
DE_DEV void rpl_write_obs(RcLds& L, int lane, int R, int obs_dim, const RplBody& b0, const RplBody& b1, const RplRobot& r, const RplEnv& v,
                          float* __restrict__ out, int W) {
  RcObsStage& O = L.u.ob;
  __syncthreads();
  if (lane < R) {
    const V2 p = rpl_robot_pos(b0, b1);
    const double ang = (b0.a + b1.a) / 2.0;
    O.rx[lane] = (float)norm_after_scale(p.x, RC_STD_NORM, RC_W / 2.0);
    O.ry[lane] = (float)norm_after_scale(p.y, RC_STD_NORM, RC_H / 2.0);
    const DevSC a = dev_sincos(ang);
    O.rcs[lane] = (float)a.c; O.rsn[lane] = (float)a.s;
    const DevSC ah = dev_sincos(ang + r.head);
    O.ahc[lane] = (float)ah.c; O.ahs[lane] = (float)ah.s;
    const DevSC h = dev_sincos(r.head);
    O.hc[lane] = (float)h.c; O.hs[lane] = (float)h.s;
    O.team[lane] = (r.flags & RF_TEAMPOS) ? 1.0f : -1.0f;
    O.down[lane] = (r.flags & (RF_FALLEN | RF_PENAL)) ? 1.0f : 0.0f;
  }
  if (lane == 10) {
    O.bx = (float)norm_after_scale(b0.px, RC_STD_NORM, RC_W / 2.0);
    O.by = (float)norm_after_scale(b0.py, RC_STD_NORM, RC_H / 2.0);
  }
  __syncthreads();
  const float owned = (float)v.owned;
  const int c0 = v.close0, c1 = v.close1;
  for (int idx = lane; idx < R * obs_dim; idx += W) {
    const int a = idx / obs_dim, ff = idx - a * obs_dim;
    const float team = O.team[a];
    float x;
    if (ff == 0) x = O.bx * team;
    else if (ff == 1) x = O.by * team;
    else if (ff == 2) x = owned * team;
    else if (ff == 3) x = (a == c0 || a == c1) ? 1.0f : 0.0f;
    else if (ff == 4) x = O.rx[a] * team;
    else if (ff == 5) x = O.ry[a] * team;
    else if (ff == 6) x = O.ahc[a];
    else if (ff == 7) x = O.ahs[a];
    else if (ff == 8) x = O.hc[a];
    else if (ff == 9) x = O.hs[a];
    else if (ff == 10) x = team;
    else if (ff == 11) x = O.down[a];
    else {
      int k = (ff - 12) / 6, q = (ff - 12) - k * 6;
      k += (k >= a);
      x = q == 0 ? O.rx[k] * team : q == 1 ? O.ry[k] * team : q == 2 ? O.rcs[k] : q == 3 ? O.rsn[k] : q == 4 ? O.team[k] * team : O.down[k];
    }
    out[idx] = x;
  }
  __syncthreads();
}

// the foot-per-lane broadphase of rc_step_body on the LDS tile (AABBs + fp32 prefilter, then the pair rounds): used when the
// register prefilter cannot rule candidates out
template <int NROUNDS>
DE_DEV int rpl_full_broadphase(RcLds& L, int lane, int R, uint64_t pairLo, uint64_t pairHi, uint64_t pairTop, int feetPairs) {
  const bool isBody = lane == RC_BALL || lane < 2 * R;
  if (isBody) {
    if (lane != RC_BALL) {
      SegW s;
      seg_world(L, lane, s);
      double l, r, b, t;
      if (s.ta.x < s.tb.x) { l = s.ta.x; r = s.tb.x; } else { l = s.tb.x; r = s.ta.x; }
      if (s.ta.y < s.tb.y) { b = s.ta.y; t = s.tb.y; } else { b = s.tb.y; t = s.ta.y; }
      L.aabb[lane][0] = l - FOOT_RADIUS; L.aabb[lane][1] = b - FOOT_RADIUS; L.aabb[lane][2] = r + FOOT_RADIUS; L.aabb[lane][3] = t + FOOT_RADIUS;
    } else {
      const double npx = L.cpx[lane], npy = L.cpy[lane];
      L.aabb[lane][0] = npx - BALL_R; L.aabb[lane][1] = npy - BALL_R; L.aabb[lane][2] = npx + BALL_R; L.aabb[lane][3] = npy + BALL_R;
    }
    const double al = L.aabb[lane][0], ab = L.aabb[lane][1], ar = L.aabb[lane][2], at = L.aabb[lane][3];
    L.u.pf.cx[lane] = (float)(0.5 * (al + ar)); L.u.pf.cy[lane] = (float)(0.5 * (ab + at));
    L.u.pf.hx[lane] = (float)(0.5 * (ar - al)) + 1.0f; L.u.pf.hy[lane] = (float)(0.5 * (at - ab)) + 1.0f;
  }
  __syncthreads();
  int cand = feetPairs, pre = 0;
#pragma unroll
  for (int t = 0; t < NROUNDS; ++t) {  // fp32 prefilter of all my pairs first: their LDS reads are in flight together
    const int pr = RC_MY_PAIR(t);
    if (pr != 0xFFFF && !((feetPairs >> t) & 1)) {
      const int i = pr >> 8, j = pr & 0xFF;
      float bx, by, bhx, bhy;
      if (j <= RC_BALL) { bx = L.u.pf.cx[j]; by = L.u.pf.cy[j]; bhx = L.u.pf.hx[j]; bhy = L.u.pf.hy[j]; }
      else { const V2 pc = post_pos(j); bx = (float)pc.x; by = (float)pc.y; bhx = 11.0f; bhy = 11.0f; }
      const float dx = L.u.pf.cx[i] - bx, dy = L.u.pf.cy[i] - by;
      if (__builtin_fabsf(dx) <= L.u.pf.hx[i] + bhx && __builtin_fabsf(dy) <= L.u.pf.hy[i] + bhy) pre |= 1 << t;
    }
  }
#pragma unroll 1
  for (int mm = pre; mm; mm &= mm - 1) {  // the exact test (cpBBIntersects) of the pairs that passed
    const int t = __builtin_ctz(mm);
    const int pr = RC_MY_PAIR(t);
    const int i = pr >> 8, j = pr & 0xFF;
    const double al = L.aabb[i][0], ab = L.aabb[i][1], ar = L.aabb[i][2], at = L.aabb[i][3];
    double bl, bb, br, bt;
    if (j <= RC_BALL) { bl = L.aabb[j][0]; bb = L.aabb[j][1]; br = L.aabb[j][2]; bt = L.aabb[j][3]; }
    else { const V2 pc = post_pos(j); bl = pc.x - POST_R; bb = pc.y - POST_R; br = pc.x + POST_R; bt = pc.y + POST_R; }
    if (al <= br && bl <= ar && ab <= bt && bb <= at) cand |= (1 << t);
  }
  __syncthreads();
  return cand;
}

// ---- the rare paths: out of line, entered with the registers flushed and left with a reload, so that nothing of the
// register-resident state is live across the call (no callee-saved traffic, and their own register allocation)
template <int EPW>
__device__ __noinline__ void rpl_slow_logic(RcCtx c, int it, const int* __restrict__ actions, const double* __restrict__ headAct, int lane) {
  rc_game_logic<EPW>(c, it, actions, headAct, lane);
  __syncthreads();
}
struct RplPairs {  // my canonical pairs of every broadphase round (16 bits each) + which of them are a robot's own feet
  uint64_t lo, hi, top;
  int feet;
};
template <int EPW>
DE_DEV RplPairs rpl_my_pairs(int lane, int R) {
  typedef Grp<EPW> G;
  constexpr int W = G::W, NROUNDS = (RC_NPAIR_ROUNDS * 64) / W;
  RplPairs p;
  p.lo = p.hi = p.top = 0ull; p.feet = 0;
#pragma unroll
  for (int t = 0; t < NROUNDS; ++t) {
    int pr = RC.pairs[t * W + lane];
    int i = pr >> 8, j = pr & 0xFF;
    bool ok = pr != 0xFFFF;
    if (ok && i < RC_BALL) ok = i < 2 * R;
    if (ok && j < RC_BALL) ok = j < 2 * R;
    uint64_t v = (uint64_t)(ok ? pr : 0xFFFF);
    if (t < 4) p.lo |= v << (16 * t); else if (t < 8) p.hi |= v << (16 * (t - 4)); else p.top |= v << (16 * (t - 8));
    if (ok && j < RC_BALL && j == i + 1 && !(i & 1)) p.feet |= 1 << t;
  }
  return p;
}
template <int EPW>
__device__ __noinline__ RcStepRet rpl_general_physics(RcCtx c, int lane, uint64_t occ, uint64_t pairLo, uint64_t pairHi, uint64_t pairTop, int feetPairs) {
  typedef Grp<EPW> G;
  constexpr int NROUNDS = (RC_NPAIR_ROUNDS * 64) / G::W;
  if (EPW != 2) {  // the caller does not keep a pair table for this lane layout: fetch it now
    const RplPairs p = rpl_my_pairs<EPW>(lane, c.R);
    pairLo = p.lo; pairHi = p.hi; pairTop = p.top; feetPairs = p.feet;
  }
  const int cand = rpl_full_broadphase<NROUNDS>(G::tile(), lane, c.R, pairLo, pairHi, pairTop, feetPairs);
  const RcStepRet sr = rc_physics_inl<EPW == 1 ? 1 : EPW>(c, lane, cand, pairLo, pairHi, pairTop, occ);
  __syncthreads();
  return sr;
}
// does the LDS tile hold a force (fall() pushes its neighbours) that the next velocity update has to consume?
template <int EPW>
DE_DEV bool rpl_any_force(const RcLds& L, int lane, int R) {
  bool f = false;
  if (lane == RC_BALL || lane < 2 * R) f = L.fx[lane] != 0.0 || L.fy[lane] != 0.0 || L.tq[lane] != 0.0;
  return Grp<EPW>::ballot(f) != 0ull;
}
// (cos, sin) of both feet's angles in ONE pass of dm_sincos: foot 1's angle visits lane 16 + r (idle in this layout)
template <int EPW>
DE_DEV void rpl_rotations(RplBody& b0, RplBody& b1, bool need0, bool need1, int lane, bool isRobot) {
  const int base = (int)threadIdx.x & (64 - Grp<EPW>::W);
  const bool helper = lane >= 16 && lane < 26;
  const double a1 = __shfl(b1.a, base | (lane & 15), 64);  // helper lane 16 + r reads robot r's foot-1 angle
  const bool n1 = __shfl((int)need1, base | (lane & 15), 64) != 0;
  const double x = helper ? a1 : b0.a;
  DevSC sc;
  sc.s = 0.0; sc.c = 1.0;
  if (helper ? n1 : (isRobot && need0)) sc = dev_sincos(x);
  const double c1 = __shfl(sc.c, base | ((lane & 15) + 16), 64), s1 = __shfl(sc.s, base | ((lane & 15) + 16), 64);
  if (isRobot && need0) { b0.rc = sc.c; b0.rs = sc.s; b0.rv = true; }
  if (isRobot && need1) { b1.rc = c1; b1.rs = s1; b1.rv = true; }
}

template <int EPW>
DE_DEV void rc_step_rpl_body(const RcState& S, const int e, const int* __restrict__ actions, const double* __restrict__ headActions, float* __restrict__ obs,
                             double* __restrict__ rewards, uint8_t* __restrict__ dones) {
  typedef Grp<EPW> G;
  constexpr int W = G::W;
  const int lane = G::lane(), R = S.R;
  if (e < 0 || e >= S.E) return;  // (two environments per wave: a half without an environment)
  RcLds& L = G::tile();
  uint64_t occ = (uint64_t)(uint32_t)G::uniform_i(S.envi[(size_t)e * RE_COUNT + RE_OCC]);
  if (occ != 0ull) __builtin_amdgcn_s_setprio(3);
  rc_load_env(S, L, e, lane, occ, W);
  RcCtx c;
  c.seed = S.seed; c.genv = (uint32_t)(S.env_id_offset + e); c.n = S.n; c.R = R;
  c.canFall = (S.flags & DYNENV_FLAG_CAN_FALL) != 0; c.allowHead = (S.flags & DYNENV_FLAG_ALLOW_HEAD_TURN) != 0;
  c.detTurn = (S.flags & DYNENV_FLAG_DETERMINISTIC_TURN) != 0;
  int err = 0;
  __syncthreads();
  c.episode = (uint32_t)G::uniform_i(L.envi[RE_EPISODE]);
  if (lane == 0) refresh_pivot_first(L);
  __syncthreads();
  const bool isRobot = lane < R, isBall = lane == 10;
  const int* myActions = actions + (size_t)e * R * 4;
  const double* myHead = headActions ? headActions + (size_t)e * R : nullptr;
  RplFric k0, k1;  // friction constants of my two bodies (foot / foot, or ball / -)
  {
    const double m0 = isBall ? 10.0 : ROBOT_MASS;
    k0.minv = isBall ? 1.0 / 10.0 : 1.0 / ROBOT_MASS; k0.iinv = isBall ? RC.ballIinv : RC.footIinv;
    k0.factor = (isBall ? 2.8e-2 : 1e-3) * m0; k0.rotFactor = (isBall ? 1e-3 : 1e-2) * m0; k0.spin = isBall ? 5e-2 : 0.0;
    k1.minv = 1.0 / ROBOT_MASS; k1.iinv = RC.footIinv; k1.factor = 1e-3 * ROBOT_MASS; k1.rotFactor = 1e-2 * ROBOT_MASS; k1.spin = 0.0;
  }
  RplPairs pairs;
  pairs.lo = pairs.hi = pairs.top = 0ull; pairs.feet = 0;
  if (EPW == 2) pairs = rpl_my_pairs<EPW>(lane, R);  // (256 VGPRs: kept for the whole step)
  RplBody b0, b1;
  RplRobot r;
  RplEnv v;
  rpl_reload<EPW>(L, lane, R, b0, b1, r, v);
  // posInRegs: the last position update ran on registers (the LDS bias velocities and shape cache are stale until the next flush).
  // vbLive: the LDS tile holds bias velocities of a contact solve that the next position update has to apply.
  // hasForce: ... forces of a fall() that the next velocity update has to consume (then that substep takes the LDS path).
  bool posInRegs = false, vbLive = true, hasForce = rpl_any_force<EPW>(L, lane, R);
  int snap = 0;
  const bool bothHalves = EPW == 2 && __popcll(__ballot(true)) == 64;  // (an odd batch leaves the last wave's second half empty)
RC_PROF(if (lane < 12 && e < 4096) g_rcprof[e * 12 + lane] = 0ull; const unsigned long long K0 = __builtin_amdgcn_s_memtime(); unsigned long long tG = 0, tP = 0, tQ = 0, tS = 0, nSlow = 0, nGen = 0;)
  for (int it = 0; it < 50; ++it) {
RC_PROF(const unsigned long long A0 = __builtin_amdgcn_s_memtime();)
    // ---- game logic: for robot in agents: [processAction]; tick; then the ball --------------------------------------
    if (isBall) { L.px[RC_BALL] = b0.px; L.py[RC_BALL] = b0.py; }
    __syncthreads();
    const V2 ballPos = v2(L.px[RC_BALL], L.py[RC_BALL]);
    bool slowLogic = it == 0;  // processAction draws the fall dice and may knock other robots over: the reference's order
    if (!RPL_ROBOT_REGS(EPW) && !slowLogic) rpl_load_robot(L, lane, R, r, true, false);
    if (!slowLogic) slowLogic = G::ballot((isRobot && rpl_tick_has_event(b0, b1, r, v, lane)) || rpl_ball_outside(ballPos)) != 0ull;
RC_PROF(nSlow += slowLogic;)
    if (slowLogic) {
      rpl_flush<EPW>(L, lane, R, b0, b1, r, v, posInRegs);
      posInRegs = false;
      __syncthreads();
      rpl_slow_logic<EPW>(c, it, myActions, myHead, lane);
      rpl_reload<EPW>(L, lane, R, b0, b1, r, v);
      hasForce = rpl_any_force<EPW>(L, lane, R);
    } else {
      if (isRobot) rpl_tick(b0, b1, r, v, lane, ballPos);
      rpl_ball_logic<EPW>(c, L, lane, b0, b1, r, v, ballPos);
      if (!RPL_ROBOT_REGS(EPW)) rpl_store_robot(L, lane, R, r, true, false);
    }
RC_PROF(const unsigned long long A1 = __builtin_amdgcn_s_memtime();)
    // ---- cpBodyUpdatePosition ------------------------------------------------------------------------------------------
    {
      double vbx0 = 0.0, vby0 = 0.0, wb0 = 0.0, vbx1 = 0.0, vby1 = 0.0, wb1 = 0.0;
      if (vbLive) {
        const int i0 = isBall ? RC_BALL : (isRobot ? 2 * lane : 0), i1 = isRobot ? 2 * lane + 1 : 0;
        vbx0 = L.vbx[i0]; vby0 = L.vby[i0]; wb0 = L.wb[i0]; vbx1 = L.vbx[i1]; vby1 = L.vby[i1]; wb1 = L.wb[i1];
      }
      bool need0 = false, need1 = false;
      if (isRobot || isBall) need0 = rpl_update_position(b0, vbx0, vby0, wb0);
      if (isRobot) need1 = rpl_update_position(b1, vbx1, vby1, wb1);
      if (G::ballot(isRobot && (need0 || need1)) != 0ull) rpl_rotations<EPW>(b0, b1, need0, need1, lane, isRobot);
      posInRegs = true; vbLive = false;
    }
    // ---- is this a quiet substep?  no cached arbiter, no force to consume, the robots' own feet apart, no other candidate pair
    bool quiet = occ == 0ull && !hasForce;
    if (quiet) {
      SegW s1, s2;
      rpl_seg_world(b0, false, s1);
      rpl_seg_world(b1, true, s2);
      const bool far = isRobot ? capsules_far_apart(s1, s2) : true;
      quiet = G::ballot(!far) == 0ull;
      if (quiet) quiet = rpl_no_candidates<EPW>(L, lane, R, b0, b1, s1, s2);
    }
RC_PROF(const unsigned long long A2 = __builtin_amdgcn_s_memtime(); nGen += !quiet;)
    if (quiet) {  // velocity update, then every robot's joints: all on registers
      if (isRobot || isBall) rpl_velocity_update(b0, k0);
      if (isRobot) rpl_velocity_update(b1, k1);
      if (!RPL_ROBOT_REGS(EPW)) rpl_load_robot(L, lane, R, r, false, true);
      if (isRobot) rpl_joints(b0, b1, r, v, lane);
      if (!RPL_ROBOT_REGS(EPW)) rpl_store_robot(L, lane, R, r, false, true);
    }
    if (EPW == 2) {
      // Two environments per wave: the general path of ONE of them runs with the whole wave on that environment's tile - the
      // foot-per-lane code with wave-uniform control flow (Grp<3> / Grp<4>), twice as fast as its half-wave form.  Both
      // halves flush first (so that no register of the resident layout is live across the out-of-line call) and reload after.
      const uint64_t needW = __ballot(!quiet);
      const bool need0 = (needW & 0xFFFFFFFFull) != 0ull, need1 = (needW >> 32) != 0ull;
      if ((need0 && need1) || (needW != 0ull && !bothHalves)) {
        // both environments have contact work: side by side in their halves (the half-wave form costs one what it costs two)
        if (!quiet) {
          __builtin_amdgcn_s_setprio(3);
          rpl_flush<EPW>(L, lane, R, b0, b1, r, v, true);
          posInRegs = false;
          __syncthreads();
          const RcStepRet sr = rpl_general_physics<EPW>(c, lane, occ, pairs.lo, pairs.hi, pairs.top, pairs.feet);
          occ = sr.occ; err |= sr.err;
          rpl_reload<EPW>(L, lane, R, b0, b1, r, v);
          vbLive = true;
          hasForce = rpl_any_force<EPW>(L, lane, R);
        }
      } else if (needW != 0ull) {
        __builtin_amdgcn_s_setprio(3);  // (an environment with contact work is on the launch's critical path: issue it first)
        rpl_flush<EPW>(L, lane, R, b0, b1, r, v, true);
        posInRegs = false;
        __syncthreads();
        const int wl = (int)threadIdx.x;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if ((needW >> (32 * h)) & 0xFFFFFFFFull) {
            RcCtx ch = c;
            ch.genv = (uint32_t)__builtin_amdgcn_readlane((int)c.genv, 32 * h);
            ch.episode = (uint32_t)__builtin_amdgcn_readlane((int)c.episode, 32 * h);
            const uint64_t occh = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)occ, 32 * h);
            const RcStepRet sr = h == 0 ? rpl_general_physics<3>(ch, wl, occh, 0ull, 0ull, 0ull, 0) : rpl_general_physics<4>(ch, wl, occh, 0ull, 0ull, 0ull, 0);
            if (G::id() == h) { occ = sr.occ; err |= sr.err; }
          }
        }
        rpl_reload<EPW>(L, lane, R, b0, b1, r, v);
        vbLive = true;  // (a half that did not solve finds zeros there)
        hasForce = rpl_any_force<EPW>(L, lane, R);  // a post-solve callback may have made a robot fall
      }
    } else if (!quiet) {  // (an environment with contact work is on the launch's critical path: issue it first)
      __builtin_amdgcn_s_setprio(3);
      rpl_flush<EPW>(L, lane, R, b0, b1, r, v, true);
      posInRegs = false;
      __syncthreads();
      const RcStepRet sr = rpl_general_physics<EPW>(c, lane, occ, pairs.lo, pairs.hi, pairs.top, pairs.feet);
      occ = G::uniform_u64(sr.occ); err |= sr.err;
      rpl_reload<EPW>(L, lane, R, b0, b1, r, v);
      vbLive = true;
      hasForce = rpl_any_force<EPW>(L, lane, R);  // a post-solve callback may have made a robot fall
    }
RC_PROF(const unsigned long long A3 = __builtin_amdgcn_s_memtime(); tG += A1 - A0; tP += A2 - A1; if (quiet) tQ += A3 - A2; else tS += A3 - A2;)
    if (lane == 0) L.envi[RE_ELAPSED] += 1;
    if (it % 10 == 9) {
      if (obs) {
        if (!RPL_ROBOT_REGS(EPW)) rpl_load_robot(L, lane, R, r, true, false);
        rpl_write_obs(L, lane, R, S.obs_dim, b0, b1, r, v, obs + ((size_t)e * 5 + snap) * R * S.obs_dim, W);
      }
      ++snap;
    }
  }
RC_PROF(if (lane == 0 && e < 4096) { unsigned long long* d = g_rcprof + e * 12; d[0] = tG; d[1] = tP; d[2] = tQ; d[4] = tS; d[5] = nSlow * 100 + nGen; d[11] = __builtin_amdgcn_s_memtime() - K0; })
  // ---- end of env step :497-524 --------------------------------------------------------------------------------------------
  if (!RPL_ROBOT_REGS(EPW)) rpl_load_robot(L, lane, R, r, true, false);
  if (isRobot) {
    const double tr = lane < S.n ? v.team0 : v.team1;
    double rew = r.rrew + tr;
    double prew = r.rposrew + dm_max(0.0, tr);
    rew += 0.0;   // obsRewards are zero for Full observations (processSeens returns early)
    prew += 0.0;
    double* er = S.epr + (size_t)e * 16 + lane;
    double* ep = S.epr + (size_t)S.E * 16 + (size_t)e * 16 + lane;
    *er = *er + rew;
    *ep = *ep + prew;
    rewards[(size_t)e * R + lane] = rew;
  }
  rpl_flush<EPW>(L, lane, R, b0, b1, r, v, posInRegs);
  __syncthreads();
  if (lane == 0) {
    dones[e] = (uint8_t)(L.envi[RE_ELAPSED] >= RC_MAX_TIME);
    L.envi[RE_OCC] = (int)(uint32_t)occ;
    if (err) L.envi[RE_ERR] |= 1;
  }
  __syncthreads();
  rc_store_env(S, L, e, lane, occ, W);
}
