/* cp_lite.c — ORACLE (test infrastructure; see cp_lite.h header for scope and parity status).
 * Step order and formulas follow SURVEY.md Appendix A (Chipmunk2D 7.0.x cpSpaceStep / cpArbiter / cpCollision /
 * cpPivotJoint / cpRotaryLimitJoint as reached from pymunk `Space.step`, DrivingEnvironment.py:278,
 * RoboCupEnvironment.py:482). */
#include "cp_lite.h"

#include <float.h>
#include <math.h>
#include <string.h>

#include "dynenv_math.h"

static inline double cpfmax(double a, double b) { return (a > b) ? a : b; }
static inline double cpfmin(double a, double b) { return (a < b) ? a : b; }
static inline double cpfclamp(double f, double lo, double hi) { return cpfmin(cpfmax(f, lo), hi); }
static inline double cpfclamp01(double f) { return cpfmax(0.0, cpfmin(f, 1.0)); }

/* ---------------------------------------------------------------- bodies / shapes */

void cpBodyInit(cpBody* b, double m, double i, int type) {
  memset(b, 0, sizeof(*b));
  b->type = type;
  if (type == CP_BODY_STATIC) {
    b->m = INFINITY; b->i = INFINITY; b->m_inv = 0.0; b->i_inv = 0.0;
  } else {
    b->m = m; b->i = i; b->m_inv = 1.0 / m; b->i_inv = 1.0 / i;
  }
  b->rot = cpv_(1.0, 0.0);
}

void cpBodySetAngle(cpBody* b, double a) {
  double s, c;
  b->a = a;
  dm_sincos(a, &s, &c);
  b->rot = cpv_(c, s);
}

static inline cpv xform_point(const cpBody* b, cpv p) {
  /* cpTransformPoint with a=rot.x, c=-rot.y, b=rot.y, d=rot.x, t=p (cog = 0) */
  return cpv_(b->rot.x * p.x - b->rot.y * p.y + b->p.x, b->rot.y * p.x + b->rot.x * p.y + b->p.y);
}
static inline cpv xform_vect(const cpBody* b, cpv v) {
  return cpv_(b->rot.x * v.x - b->rot.y * v.y, b->rot.y * v.x + b->rot.x * v.y);
}

static void shape_common(cpShape* s, cpBody* body, int type, int slot) {
  memset(s, 0, sizeof(*s));
  s->type = type; s->body = body; s->slot = slot;
}

void cpCircleInit(cpShape* s, cpBody* body, double r, int slot) {
  shape_common(s, body, CP_SHAPE_CIRCLE, slot);
  s->r = r; s->c = cpv_(0.0, 0.0);
}

void cpSegmentInit(cpShape* s, cpBody* body, cpv a, cpv b, double r, int slot) {
  cpv d; double len;
  shape_common(s, body, CP_SHAPE_SEGMENT, slot);
  s->sa = a; s->sb = b; s->r = r;
  /* n = cpvrperp(cpvnormalize(b - a)) ; rperp(v) = (v.y, -v.x) */
  d = cpvsub(b, a);
  len = dm_sqrt(cpvdot(d, d));
  d = cpvmult(d, 1.0 / (len + DBL_MIN));
  s->sn = cpv_(d.y, -d.x);
}

void cpBoxInit(cpShape* s, cpBody* body, double hx, double hy, int slot) {
  /* hull order of pymunk Poly([(h,w),(-h,w),(-h,-w),(h,-w)]): CCW starting at the min-x/min-y vertex (cpConvexHull) */
  shape_common(s, body, CP_SHAPE_POLY, slot);
  s->count = 4; s->r = 0.0;
  s->local[0].v0 = cpv_(-hx, -hy); s->local[0].n = cpv_(-1.0, 0.0);
  s->local[1].v0 = cpv_(hx, -hy);  s->local[1].n = cpv_(0.0, -1.0);
  s->local[2].v0 = cpv_(hx, hy);   s->local[2].n = cpv_(1.0, 0.0);
  s->local[3].v0 = cpv_(-hx, hy);  s->local[3].n = cpv_(0.0, 1.0);
}

void cpShapeCacheBB(cpShape* s) {
  const cpBody* b = s->body;
  if (s->type == CP_SHAPE_CIRCLE) {
    s->tc = xform_point(b, s->c);
    s->bb_l = s->tc.x - s->r; s->bb_b = s->tc.y - s->r; s->bb_r = s->tc.x + s->r; s->bb_t = s->tc.y + s->r;
  } else if (s->type == CP_SHAPE_SEGMENT) {
    double l, r, bt, t;
    s->ta = xform_point(b, s->sa); s->tb = xform_point(b, s->sb); s->tn = xform_vect(b, s->sn);
    if (s->ta.x < s->tb.x) { l = s->ta.x; r = s->tb.x; } else { l = s->tb.x; r = s->ta.x; }
    if (s->ta.y < s->tb.y) { bt = s->ta.y; t = s->tb.y; } else { bt = s->tb.y; t = s->ta.y; }
    s->bb_l = l - s->r; s->bb_b = bt - s->r; s->bb_r = r + s->r; s->bb_t = t + s->r;
  } else {
    double l = INFINITY, r = -INFINITY, bt = INFINITY, t = -INFINITY;
    int i;
    for (i = 0; i < s->count; ++i) {
      /* polygon vertices: both multiply-adds fused (dms_xform_x/y, dynenv_math.h), as the kernels' boxp_vertex / box_world */
      cpv v = cpv_(dms_xform_x(b->rot.x, b->rot.y, s->local[i].v0.x, s->local[i].v0.y, b->p.x),
                   dms_xform_y(b->rot.x, b->rot.y, s->local[i].v0.x, s->local[i].v0.y, b->p.y));
      s->world[i].v0 = v;
      s->world[i].n = xform_vect(b, s->local[i].n);
      l = cpfmin(l, v.x); r = cpfmax(r, v.x); bt = cpfmin(bt, v.y); t = cpfmax(t, v.y);
    }
    s->bb_l = l - s->r; s->bb_b = bt - s->r; s->bb_r = r + s->r; s->bb_t = t + s->r;
  }
}

double cpMomentForBox(double m, double hx, double hy) {
  /* cpMomentForPoly over Car.points order (Car.py:21-23) */
  cpv verts[4];
  double sum1 = 0.0, sum2 = 0.0;
  int i;
  verts[0] = cpv_(hx, hy); verts[1] = cpv_(-hx, hy); verts[2] = cpv_(-hx, -hy); verts[3] = cpv_(hx, -hy);
  for (i = 0; i < 4; ++i) {
    cpv v1 = verts[i], v2 = verts[(i + 1) % 4];
    double a = cpvcross(v2, v1);
    double b = cpvdot(v1, v1) + cpvdot(v1, v2) + cpvdot(v2, v2);
    sum1 += a * b; sum2 += a;
  }
  return (m * sum1) / (6.0 * sum2);
}
double cpMomentForCircle(double m, double r1, double r2) { return m * (0.5 * (r1 * r1 + r2 * r2) + 0.0); }
double cpMomentForSegment(double m, cpv a, cpv b, double r) {
  cpv offset = cpvlerp(a, b, 0.5);
  cpv d = cpvsub(b, a);
  double length = dm_sqrt(cpvdot(d, d)) + 2.0 * r;
  return m * ((length * length + 4.0 * r * r) / 12.0 + cpvlengthsq(offset));
}

/* ---------------------------------------------------------------- space bookkeeping */

static int default_begin(cpArbiter* a, cpSpace* s, void* d) { (void)a; (void)s; (void)d; return 1; }

void cpSpaceInit(cpSpace* s) {
  memset(s, 0, sizeof(*s));
  s->iterations = 10;
  s->collisionSlop = 0.1;
  s->collisionBias = pow((double)(1.0f - 0.1f), 60.0);
  s->collisionPersistence = 3;
  s->damping = 1.0;
  s->gravity = cpv_(0.0, 0.0);
  s->default_handler.typeA = -1; s->default_handler.typeB = -1;
  s->default_handler.begin = default_begin;
}

void cpSpaceAddBody(cpSpace* s, cpBody* b) {
  if (b->type != CP_BODY_DYNAMIC) return;
  if (s->n_bodies >= CP_MAX_SHAPES) { s->overflow = 1; return; }
  s->bodies[s->n_bodies++] = b;
}

void cpSpaceAddShape(cpSpace* s, cpShape* sh) {
  int i;
  if (s->n_shapes >= CP_MAX_SHAPES) { s->overflow = 1; return; }
  cpShapeCacheBB(sh);
  i = s->n_shapes++;
  while (i > 0 && s->shapes[i - 1]->slot > sh->slot) { s->shapes[i] = s->shapes[i - 1]; --i; }
  s->shapes[i] = sh;
}

void cpSpaceAddConstraint(cpSpace* s, cpConstraint* c) {
  if (c->in_space) return;
  if (s->n_constraints >= CP_MAX_CONSTRAINTS) { s->overflow = 1; return; }
  s->constraints[s->n_constraints++] = c;
  c->in_space = 1;
}

void cpSpaceRemoveConstraint(cpSpace* s, cpConstraint* c) {
  int i, j;
  if (!c->in_space) return;
  for (i = 0; i < s->n_constraints; ++i) {
    if (s->constraints[i] == c) {
      /* Chipmunk's cpArrayDeleteObj moves the LAST element into the hole */
      j = --s->n_constraints;
      s->constraints[i] = s->constraints[j];
      break;
    }
  }
  c->in_space = 0;
}

cpHandler* cpSpaceAddHandler(cpSpace* s, int typeA, int typeB) {
  cpHandler* h;
  if (s->n_handlers >= CP_MAX_HANDLERS) { s->overflow = 1; return &s->default_handler; }
  h = &s->handlers[s->n_handlers++];
  memset(h, 0, sizeof(*h));
  h->typeA = typeA; h->typeB = typeB; h->begin = default_begin;
  return h;
}

static cpHandler* lookup_handler(cpSpace* s, int ta, int tb) {
  int i;
  for (i = 0; i < s->n_handlers; ++i) {
    cpHandler* h = &s->handlers[i];
    if ((h->typeA == ta && h->typeB == tb) || (h->typeA == tb && h->typeB == ta)) return h;
  }
  return &s->default_handler;
}

void cpArbiterGetShapes(const cpArbiter* arb, cpShape** a, cpShape** b) {
  if (arb->swapped) { *a = arb->b; *b = arb->a; } else { *a = arb->a; *b = arb->b; }
}

void cpBodyUpdateVelocity(cpBody* b, cpv gravity, double damping, double dt) {
  b->v = cpvadd(cpvmult(b->v, damping), cpvmult(cpvadd(gravity, cpvmult(b->f, b->m_inv)), dt));
  b->w = b->w * damping + b->t * b->i_inv * dt;
  b->f = cpv_(0.0, 0.0);
  b->t = 0.0;
}

void cpBodyApplyForceAtWorldPoint(cpBody* b, cpv force, cpv point) {
  cpv r = cpvsub(point, b->p); /* cog = 0 */
  b->f = cpvadd(b->f, force);
  b->t += cpvcross(r, force);
}

/* ---------------------------------------------------------------- narrowphase */

typedef struct { cpShape *a, *b; cpv n; int count; cpv p1[2], p2[2]; uint32_t hash[2]; int degenerate; /* the normal is a fallback, not a contact normal */ } cpCollisionInfo;

static void push_contact(cpCollisionInfo* info, cpv p1, cpv p2, uint32_t hash) {
  info->p1[info->count] = p1; info->p2[info->count] = p2; info->hash[info->count] = hash;
  info->count++;
}

typedef struct { cpv p; uint32_t hash; } cpEdgePoint;
typedef struct { cpEdgePoint a, b; double r; cpv n; } cpEdge;

static int poly_support_index(const cpShape* p, cpv n) {
  double max = -INFINITY; int index = 0, i;
  for (i = 0; i < p->count; ++i) {
    double d = dms_dot(p->world[i].v0.x, p->world[i].v0.y, n.x, n.y);
    if (d > max) { max = d; index = i; }
  }
  return index;
}

static cpEdge support_edge_poly(const cpShape* p, cpv n) {
  int count = p->count;
  int i1 = poly_support_index(p, n);
  int i0 = (i1 - 1 + count) % count;
  int i2 = (i1 + 1) % count;
  uint32_t h = (uint32_t)p->slot * 4u;
  cpEdge e;
  if (cpvdot(n, p->world[i1].n) > cpvdot(n, p->world[i2].n)) {
    e.a.p = p->world[i0].v0; e.a.hash = h + (uint32_t)i0;
    e.b.p = p->world[i1].v0; e.b.hash = h + (uint32_t)i1;
    e.r = p->r; e.n = p->world[i1].n;
  } else {
    e.a.p = p->world[i1].v0; e.a.hash = h + (uint32_t)i1;
    e.b.p = p->world[i2].v0; e.b.hash = h + (uint32_t)i2;
    e.r = p->r; e.n = p->world[i2].n;
  }
  return e;
}

static cpEdge support_edge_segment(const cpShape* s, cpv n) {
  uint32_t h = (uint32_t)s->slot * 4u;
  cpEdge e;
  if (cpvdot(s->tn, n) > 0.0) {
    e.a.p = s->ta; e.a.hash = h + 0u; e.b.p = s->tb; e.b.hash = h + 1u; e.r = s->r; e.n = s->tn;
  } else {
    e.a.p = s->tb; e.a.hash = h + 1u; e.b.p = s->ta; e.b.hash = h + 0u; e.r = s->r; e.n = cpvneg(s->tn);
  }
  return e;
}

static inline uint32_t hash_pair(uint32_t a, uint32_t b) { return 1u + ((a << 8) | b); }

/* Chipmunk cpCollision.c ContactPoints(): clip the two support edges against each other */
static void contact_points(cpEdge e1, cpEdge e2, cpv n, cpCollisionInfo* info) {
  double d_e1_a = cpvcross(e1.a.p, n);
  double d_e1_b = cpvcross(e1.b.p, n);
  double d_e2_a = cpvcross(e2.a.p, n);
  double d_e2_b = cpvcross(e2.b.p, n);
  double e1_denom = 1.0 / (d_e1_b - d_e1_a + DBL_MIN);
  double e2_denom = 1.0 / (d_e2_b - d_e2_a + DBL_MIN);
  info->n = n;
  {
    cpv p1 = cpvadd(cpvmult(n, e1.r), cpvlerp(e1.a.p, e1.b.p, cpfclamp01((d_e2_b - d_e1_a) * e1_denom)));
    cpv p2 = cpvadd(cpvmult(n, -e2.r), cpvlerp(e2.a.p, e2.b.p, cpfclamp01((d_e1_a - d_e2_a) * e2_denom)));
    double dist = cpvdot(cpvsub(p2, p1), n);
    if (dist <= 0.0) push_contact(info, p1, p2, hash_pair(e1.a.hash, e2.b.hash));
  }
  {
    cpv p1 = cpvadd(cpvmult(n, e1.r), cpvlerp(e1.a.p, e1.b.p, cpfclamp01((d_e2_a - d_e1_a) * e1_denom)));
    cpv p2 = cpvadd(cpvmult(n, -e2.r), cpvlerp(e2.a.p, e2.b.p, cpfclamp01((d_e1_b - d_e2_a) * e2_denom)));
    double dist = cpvdot(cpvsub(p2, p1), n);
    if (dist <= 0.0) push_contact(info, p1, p2, hash_pair(e1.b.hash, e2.a.hash));
  }
}

/* max over planes of `a` of (min over verts of `b` of plane distance) */
static double sat_max_sep(const cpShape* a, const cpShape* b, int* best) {
  double maxsep = -INFINITY; int i, j;
  *best = 0;
  for (i = 0; i < a->count; ++i) {
    cpv n = a->world[i].n;
    double d0 = dms_dot(n.x, n.y, a->world[i].v0.x, a->world[i].v0.y);
    double minv = INFINITY;
    for (j = 0; j < b->count; ++j) {
      double d = dms_dot(n.x, n.y, b->world[j].v0.x, b->world[j].v0.y) - d0;
      if (d < minv) minv = d;
    }
    if (minv > maxsep) { maxsep = minv; *best = i; }
  }
  return maxsep;
}

static void poly_to_poly(cpShape* p1, cpShape* p2, cpCollisionInfo* info) {
  int ia, ib;
  double sa = sat_max_sep(p1, p2, &ia);
  double sb;
  cpv n;
  if (sa > 0.0) return;
  sb = sat_max_sep(p2, p1, &ib);
  if (sb > 0.0) return;
  if (sa >= sb) n = p1->world[ia].n; else n = cpvneg(p2->world[ib].n);
  contact_points(support_edge_poly(p1, n), support_edge_poly(p2, cpvneg(n)), n, info);
}

static void circle_to_circle(cpShape* c1, cpShape* c2, cpCollisionInfo* info) {
  double mindist = c1->r + c2->r;
  cpv delta = cpvsub(c2->tc, c1->tc);
  double distsq = cpvlengthsq(delta);
  if (distsq < mindist * mindist) {
    double dist = dm_sqrt(distsq);
    cpv n = info->n = (dist != 0.0 ? cpvmult(delta, 1.0 / dist) : cpv_(1.0, 0.0));
    push_contact(info, cpvadd(c1->tc, cpvmult(n, c1->r)), cpvadd(c2->tc, cpvmult(n, -c2->r)), 0);
  }
}

static void circle_to_segment(cpShape* circle, cpShape* seg, cpCollisionInfo* info) {
  cpv seg_a = seg->ta, seg_b = seg->tb, center = circle->tc;
  cpv seg_delta = cpvsub(seg_b, seg_a);
  double closest_t = cpfclamp01(cpvdot(seg_delta, cpvsub(center, seg_a)) / cpvlengthsq(seg_delta));
  cpv closest = cpvadd(seg_a, cpvmult(seg_delta, closest_t));
  double mindist = circle->r + seg->r;
  cpv delta = cpvsub(closest, center);
  double distsq = cpvlengthsq(delta);
  if (distsq < mindist * mindist) {
    double dist = dm_sqrt(distsq);
    cpv n = info->n = (dist != 0.0 ? cpvmult(delta, 1.0 / dist) : seg->tn);
    /* a_tangent = b_tangent = 0 (pymunk default) so the endcap rejection never fires */
    push_contact(info, cpvadd(center, cpvmult(n, circle->r)), cpvadd(closest, cpvmult(n, -seg->r)), 0);
  }
}

/* circle vs convex poly (radius 0): closest point on the boundary (outside) or minimum-penetration face (inside) */
static void circle_to_poly(cpShape* circle, cpShape* poly, cpCollisionInfo* info) {
  cpv c = circle->tc;
  double maxsep = -INFINITY; int best = 0, i;
  for (i = 0; i < poly->count; ++i) {
    double d = cpvdot(poly->world[i].n, cpvsub(c, poly->world[i].v0));
    if (d > maxsep) { maxsep = d; best = i; }
  }
  if (maxsep > circle->r) return; /* cheap reject: farther than r from one face plane */
  if (maxsep <= 0.0) {
    /* centre inside: n points from the circle into the poly = -face normal, d = maxsep (<0) */
    cpv fn = poly->world[best].n;
    cpv n = info->n = cpvneg(fn);
    cpv pb = cpvsub(c, cpvmult(fn, maxsep));
    push_contact(info, cpvadd(c, cpvmult(n, circle->r)), pb, 0);
  } else {
    /* centre outside: closest point over the 4 edges (edge i runs v[i-1] -> v[i]) */
    double bestd = INFINITY; cpv bestp = c;
    for (i = 0; i < poly->count; ++i) {
      cpv a = poly->world[(i + poly->count - 1) % poly->count].v0, b = poly->world[i].v0;
      cpv d = cpvsub(b, a);
      double t = cpfclamp01(cpvdot(d, cpvsub(c, a)) / cpvlengthsq(d));
      cpv q = cpvadd(a, cpvmult(d, t));
      double dsq = cpvlengthsq(cpvsub(q, c));
      if (dsq < bestd) { bestd = dsq; bestp = q; }
    }
    if (bestd <= circle->r * circle->r) { /* points.d - mindist <= 0 */
      double dist = dm_sqrt(bestd);
      cpv delta = cpvsub(bestp, c);
      cpv n = info->n = (dist != 0.0 ? cpvmult(delta, 1.0 / dist) : cpvneg(poly->world[best].n));
      push_contact(info, cpvadd(c, cpvmult(n, circle->r)), bestp, 0);
    }
  }
}

/* closest points between two segment cores (replaces GJK for capsule-capsule) */
static void closest_seg_seg(cpv p1, cpv q1, cpv p2, cpv q2, cpv* c1, cpv* c2) {
  cpv d1 = cpvsub(q1, p1), d2 = cpvsub(q2, p2), r = cpvsub(p1, p2);
  double a = cpvdot(d1, d1), e = cpvdot(d2, d2), f = cpvdot(d2, r);
  double c = cpvdot(d1, r), b = cpvdot(d1, d2);
  double denom = a * e - b * b;
  double s, t;
  if (denom != 0.0) s = cpfclamp01((b * f - c * e) / denom); else s = 0.0;
  t = (b * s + f) / e;
  if (t < 0.0) { t = 0.0; s = cpfclamp01(-c / a); }
  else if (t > 1.0) { t = 1.0; s = cpfclamp01((b - c) / a); }
  *c1 = cpvadd(p1, cpvmult(d1, s));
  *c2 = cpvadd(p2, cpvmult(d2, t));
}

/* Capsule cores that touch or CROSS (closest distance below 1e-6: an exact 0, or the rounding noise of the crossing point computed twice).
 * Two feet of ONE robot get there in play once the robot has been knocked about (the rotary limit joint is soft: feet a radian apart,
 * RoboCup seed 42, environment 1193, step 119 - found by error bit 4 in round 6; the `d == 0` counter of round 5 never saw it because a
 * crossing comes out as ~1e-15, not 0).  The closest-point formula has no normal there.  Chipmunk's GJK finds the origin INSIDE the
 * Minkowski difference B - A of the two cores - a parallelogram with two edges parallel to each core - and its EPA answers with the
 * inward normal of the edge closest to the origin and minus that distance (cpCollision.c EPA / ClosestPointsNew).  The four edges are the
 * four "bring one end point back onto the other core's line" translations:
 *   edge {b_k - a(t)}: distance |s_k|, s_k = cross(uA, b_k - a_0), inward normal -sign(s_k) perp(uA)      (end k of B against A's line)
 *   edge {b(t) - a_k}: distance |t_k|, t_k = cross(uB, a_k - b_0), inward normal +sign(t_k) perp(uB)      (end k of A against B's line)
 * (n points from shape 1 to shape 2: pushing shape 2 along n by the distance separates the cores).  Candidates in the order s_0, s_1, t_0,
 * t_1, the first strictly smallest wins (which of two EQUAL edges Chipmunk's EPA takes depends on its hull order: unknowable, measure zero).
 * Returns 0 when the cores are close without crossing (the end points of one core on ONE side of the other's line): then the closest
 * points have a direction, however small the distance.  *exact0 = the winning distance is exactly 0 (collinear or exactly touching
 * cores: the normal's SIGN is a convention) -> cpSpace.degenerate, error bit 4. */
long cp_lite_cores_cross = 0; /* diagnostics (tools / tests): narrowphase calls that took the crossing branch */
#define CP_CORES_TOUCH_DSQ 1e-12 /* (1e-6 px)^2; the kernels use the same constant (robocup_kernels.hip rc_narrowphase) */
static int cores_crossing_normal(const cpShape* s1, const cpShape* s2, cpv* n, int* exact0) {
  cpv dA = cpvsub(s1->tb, s1->ta), dB = cpvsub(s2->tb, s2->ta);
  cpv uA = cpvmult(dA, 1.0 / dm_sqrt(cpvlengthsq(dA))), uB = cpvmult(dB, 1.0 / dm_sqrt(cpvlengthsq(dB)));
  double s0 = cpvcross(uA, cpvsub(s2->ta, s1->ta)), s1_ = cpvcross(uA, cpvsub(s2->tb, s1->ta));
  double t0 = cpvcross(uB, cpvsub(s1->ta, s2->ta)), t1 = cpvcross(uB, cpvsub(s1->tb, s2->ta));
  double best;
  if (!(s0 * s1_ <= 0.0 && t0 * t1 <= 0.0)) return 0;
  best = fabs(s0); *n = (s0 > 0.0 ? cpvneg(cpvperp(uA)) : cpvperp(uA));
  if (fabs(s1_) < best) { best = fabs(s1_); *n = (s1_ > 0.0 ? cpvneg(cpvperp(uA)) : cpvperp(uA)); }
  if (fabs(t0) < best) { best = fabs(t0); *n = (t0 > 0.0 ? cpvperp(uB) : cpvneg(cpvperp(uB))); }
  if (fabs(t1) < best) { best = fabs(t1); *n = (t1 > 0.0 ? cpvperp(uB) : cpvneg(cpvperp(uB))); }
  *exact0 = best == 0.0;
  return 1;
}

static void segment_to_segment(cpShape* s1, cpShape* s2, cpCollisionInfo* info) {
  cpv a, b, delta, n;
  double dsq, d, mind = s1->r + s2->r;
  int crossing = 0, exact0 = 0;
  closest_seg_seg(s1->ta, s1->tb, s2->ta, s2->tb, &a, &b);
  delta = cpvsub(b, a);
  dsq = cpvlengthsq(delta);
  if (dsq > mind * mind) return;
  d = dm_sqrt(dsq);
  n = (d != 0.0 ? cpvmult(delta, 1.0 / d) : s1->tn);
  if (dsq < CP_CORES_TOUCH_DSQ) {
    cpv nx;
    crossing = cores_crossing_normal(s1, s2, &nx, &exact0);
    if (crossing) {
      n = nx;
#pragma omp atomic
      cp_lite_cores_cross++;
    }
    if (exact0 || (!crossing && d == 0.0)) info->degenerate = 1; /* -> cpSpace.degenerate, error bit 4 of both sides (include/dynenv.h) */
  }
  contact_points(support_edge_segment(s1, n), support_edge_segment(s2, cpvneg(n)), n, info);
}

static void collide(cpShape* a, cpShape* b, cpCollisionInfo* info) {
  info->a = a; info->b = b; info->count = 0; info->n = cpv_(0.0, 0.0); info->degenerate = 0;
  if (a->type > b->type) { info->a = b; info->b = a; }
  a = info->a; b = info->b;
  if (a->type == CP_SHAPE_CIRCLE && b->type == CP_SHAPE_CIRCLE) circle_to_circle(a, b, info);
  else if (a->type == CP_SHAPE_CIRCLE && b->type == CP_SHAPE_SEGMENT) circle_to_segment(a, b, info);
  else if (a->type == CP_SHAPE_CIRCLE && b->type == CP_SHAPE_POLY) circle_to_poly(a, b, info);
  else if (a->type == CP_SHAPE_SEGMENT && b->type == CP_SHAPE_SEGMENT) segment_to_segment(a, b, info);
  else if (a->type == CP_SHAPE_POLY && b->type == CP_SHAPE_POLY) poly_to_poly(a, b, info);
  /* segment-poly never occurs in DynEnv */
}


/* differential tests only (tests/test_kat_general.py): the narrowphase of one shape pair as cpSpaceStep would run it - bounding
 * boxes refreshed, QueryReject, collide() in shape-type order.  out[0] = bounding boxes overlap, out[1] = contact count,
 * out[2] = 1 if the shapes were swapped into type order, out[3..4] = n, then per contact p1.x p1.y p2.x p2.y hash */
void cpTestCollide(cpShape* a, cpShape* b, double* out) {
  cpCollisionInfo info;
  int i;
  cpShapeCacheBB(a); cpShapeCacheBB(b);
  for (i = 0; i < 15; ++i) out[i] = 0.0;
  if (!(a->bb_l <= b->bb_r && b->bb_l <= a->bb_r && a->bb_b <= b->bb_t && b->bb_b <= a->bb_t)) return;
  out[0] = 1.0;
  collide(a, b, &info);
  out[1] = (double)info.count; out[2] = (info.a != a) ? 1.0 : 0.0; out[3] = info.n.x; out[4] = info.n.y;
  for (i = 0; i < info.count; ++i) {
    out[5 + 5 * i + 0] = info.p1[i].x; out[5 + 5 * i + 1] = info.p1[i].y;
    out[5 + 5 * i + 2] = info.p2[i].x; out[5 + 5 * i + 3] = info.p2[i].y; out[5 + 5 * i + 4] = (double)info.hash[i];
  }
}

/* ---------------------------------------------------------------- arbiters */

static cpArbiter* arbiter_find_or_create(cpSpace* s, cpShape* a, cpShape* b) {
  int i, free_i = -1;
  for (i = 0; i < CP_MAX_ARBITERS; ++i) {
    cpArbiter* arb = &s->pool[i];
    if (!arb->used) { if (free_i < 0) free_i = i; continue; }
    if ((arb->a == a && arb->b == b) || (arb->a == b && arb->b == a)) return arb;
  }
  if (free_i < 0) { s->overflow = 1; return 0; }
  {
    cpArbiter* arb = &s->pool[free_i];
    memset(arb, 0, sizeof(*arb));
    arb->used = 1; arb->a = a; arb->b = b; arb->body_a = a->body; arb->body_b = b->body;
    arb->state = CP_ARB_FIRST; arb->stamp = 0;
    return arb;
  }
}

static void arbiter_update(cpArbiter* arb, cpCollisionInfo* info, cpSpace* s) {
  cpShape *a = info->a, *b = info->b;
  cpContact fresh[2];
  int i, j;
  arb->a = a; arb->body_a = a->body; arb->b = b; arb->body_b = b->body;
  for (i = 0; i < info->count; ++i) {
    cpContact* con = &fresh[i];
    memset(con, 0, sizeof(*con));
    con->r1 = cpvsub(info->p1[i], a->body->p);
    con->r2 = cpvsub(info->p2[i], b->body->p);
    con->hash = info->hash[i];
    con->jnAcc = con->jtAcc = 0.0;
    for (j = 0; j < arb->count; ++j) {
      cpContact* old = &arb->contacts[j];
      if (con->hash == old->hash) { con->jnAcc = old->jnAcc; con->jtAcc = old->jtAcc; }
    }
  }
  for (i = 0; i < info->count; ++i) arb->contacts[i] = fresh[i];
  arb->count = info->count;
  arb->n = info->n;
  arb->e = a->e * b->e;
  arb->u = a->u * b->u;
  arb->handler = lookup_handler(s, a->collision_type, b->collision_type);
  arb->swapped = (a->collision_type != arb->handler->typeA && arb->handler->typeA != -1);
  if (arb->state == CP_ARB_CACHED) arb->state = CP_ARB_FIRST;
}

static void collide_shapes(cpSpace* s, cpShape* a, cpShape* b) {
  cpCollisionInfo info;
  cpArbiter* arb;
  /* QueryReject */
  if (!(a->bb_l <= b->bb_r && b->bb_l <= a->bb_r && a->bb_b <= b->bb_t && b->bb_b <= a->bb_t)) return;
  if (a->body == b->body) return;
  collide(a, b, &info);
  if (info.degenerate) s->degenerate = 1;
  if (info.count == 0) return;
  arb = arbiter_find_or_create(s, info.a, info.b);
  if (!arb) return;
  arbiter_update(arb, &info, s);
  if (arb->state == CP_ARB_FIRST && !arb->handler->begin(arb, s, arb->handler->data)) arb->state = CP_ARB_IGNORE;
  if (arb->state != CP_ARB_IGNORE && !(a->body->m == INFINITY && b->body->m == INFINITY)) {
    s->active[s->n_active++] = arb;
  }
  arb->stamp = s->stamp;
}

static inline double k_scalar_body(const cpBody* b, cpv r, cpv n) {
  double rcn = cpvcross(r, n);
  return b->m_inv + b->i_inv * rcn * rcn;
}
static inline cpv relative_velocity(const cpBody* a, const cpBody* b, cpv r1, cpv r2) {
  cpv v1_sum = cpvadd(a->v, cpvmult(cpvperp(r1), a->w));
  cpv v2_sum = cpvadd(b->v, cpvmult(cpvperp(r2), b->w));
  return cpvsub(v2_sum, v1_sum);
}
static inline void apply_impulse(cpBody* b, cpv j, cpv r) {
  b->v = cpvadd(b->v, cpvmult(j, b->m_inv));
  b->w += b->i_inv * cpvcross(r, j);
}
static inline void apply_impulses(cpBody* a, cpBody* b, cpv r1, cpv r2, cpv j) {
  apply_impulse(a, cpvneg(j), r1);
  apply_impulse(b, j, r2);
}
static inline void apply_bias_impulse(cpBody* b, cpv j, cpv r) {
  b->v_bias = cpvadd(b->v_bias, cpvmult(j, b->m_inv));
  b->w_bias += b->i_inv * cpvcross(r, j);
}

/* The ARBITER solver's arithmetic with its multiply-adds fused (include/dynenv_math.h, "fused multiply-add ..."): the same dms_*
 * functions the kernels call.  Operation order is Chipmunk's (cpArbiterPreStep, cpArbiterApplyCachedImpulse,
 * cpArbiterApplyImpulse); only the rounding of a * b + c differs from the unfused helpers above, which the joints keep. */
static inline double k_scalar_body_f(const cpBody* b, cpv r, cpv n) { return dms_k_scalar(b->m_inv, b->i_inv, r.x, r.y, n.x, n.y); }
static inline cpv relative_velocity_f(const cpBody* a, const cpBody* b, cpv r1, cpv r2) {
  cpv v1_sum = cpv_(dms_point_vx(a->v.x, r1.y, a->w), dms_point_vy(a->v.y, r1.x, a->w));
  cpv v2_sum = cpv_(dms_point_vx(b->v.x, r2.y, b->w), dms_point_vy(b->v.y, r2.x, b->w));
  return cpvsub(v2_sum, v1_sum);
}
static inline void apply_impulse_f(cpBody* b, cpv j, cpv r) {
  b->v = cpv_(dm_fma(j.x, b->m_inv, b->v.x), dm_fma(j.y, b->m_inv, b->v.y));
  b->w = dm_fma(b->i_inv, dms_cross(r.x, r.y, j.x, j.y), b->w);
}
static inline void apply_impulses_f(cpBody* a, cpBody* b, cpv r1, cpv r2, cpv j) {
  apply_impulse_f(a, cpvneg(j), r1);
  apply_impulse_f(b, j, r2);
}
static inline void apply_bias_impulse_f(cpBody* b, cpv j, cpv r) {
  b->v_bias = cpv_(dm_fma(j.x, b->m_inv, b->v_bias.x), dm_fma(j.y, b->m_inv, b->v_bias.y));
  b->w_bias = dm_fma(b->i_inv, dms_cross(r.x, r.y, j.x, j.y), b->w_bias);
}
static inline cpv cpvrotate_f(cpv n, cpv j) { return cpv_(dms_rotate_x(n.x, n.y, j.x, j.y), dms_rotate_y(n.x, n.y, j.x, j.y)); }

static void arbiter_prestep(cpArbiter* arb, double dt, double slop, double bias) {
  cpBody *a = arb->body_a, *b = arb->body_b;
  cpv n = arb->n;
  cpv body_delta = cpvsub(b->p, a->p);
  int i;
  for (i = 0; i < arb->count; ++i) {
    cpContact* con = &arb->contacts[i];
    double dist;
    cpv d, vr;
    con->nMass = 1.0 / (k_scalar_body_f(a, con->r1, n) + k_scalar_body_f(b, con->r2, n));
    con->tMass = 1.0 / (k_scalar_body_f(a, con->r1, cpvperp(n)) + k_scalar_body_f(b, con->r2, cpvperp(n)));
    d = cpvadd(cpvsub(con->r2, con->r1), body_delta);
    dist = dms_dot(d.x, d.y, n.x, n.y);
    con->bias = -bias * cpfmin(0.0, dist + slop) / dt;
    con->jBias = 0.0;
    vr = relative_velocity_f(a, b, con->r1, con->r2);
    con->bounce = dms_dot(vr.x, vr.y, n.x, n.y) * arb->e;
  }
}

static void arbiter_apply_cached(cpArbiter* arb, double dt_coef) {
  int i;
  if (arb->state == CP_ARB_FIRST) return;
  for (i = 0; i < arb->count; ++i) {
    cpContact* con = &arb->contacts[i];
    cpv j = cpvrotate_f(arb->n, cpv_(con->jnAcc, con->jtAcc));
    apply_impulses_f(arb->body_a, arb->body_b, con->r1, con->r2, cpvmult(j, dt_coef));
  }
}

static void arbiter_apply_impulse(cpArbiter* arb) {
  cpBody *a = arb->body_a, *b = arb->body_b;
  cpv n = arb->n;
  double friction = arb->u;
  int i;
  for (i = 0; i < arb->count; ++i) {
    cpContact* con = &arb->contacts[i];
    double nMass = con->nMass;
    cpv r1 = con->r1, r2 = con->r2;
    cpv vb1 = cpv_(dms_point_vx(a->v_bias.x, r1.y, a->w_bias), dms_point_vy(a->v_bias.y, r1.x, a->w_bias));
    cpv vb2 = cpv_(dms_point_vx(b->v_bias.x, r2.y, b->w_bias), dms_point_vy(b->v_bias.y, r2.x, b->w_bias));
    cpv vr = relative_velocity_f(a, b, r1, r2); /* surface_vr = 0 */
    cpv dvb = cpvsub(vb2, vb1);
    double vbn = dms_dot(dvb.x, dvb.y, n.x, n.y);
    double vrn = dms_dot(vr.x, vr.y, n.x, n.y);
    double vrt = dms_dot(vr.x, vr.y, -n.y, n.x); /* vr . perp(n) */
    double jbnOld = con->jBias;
    double jnOld, jtMax, jtOld;
    con->jBias = dms_acc_clamp0(con->bias - vbn, nMass, jbnOld);
    jnOld = con->jnAcc;
    con->jnAcc = dms_acc_clamp0(-(con->bounce + vrn), nMass, jnOld);
    jtMax = friction * con->jnAcc;
    jtOld = con->jtAcc;
    con->jtAcc = cpfclamp(dm_fma(-vrt, con->tMass, jtOld), -jtMax, jtMax);
    {
      cpv jb = cpvmult(n, con->jBias - jbnOld);
      apply_bias_impulse_f(a, cpvneg(jb), r1);
      apply_bias_impulse_f(b, jb, r2);
    }
    apply_impulses_f(a, b, r1, r2, cpvrotate_f(n, cpv_(con->jnAcc - jnOld, con->jtAcc - jtOld)));
  }
}

/* ---------------------------------------------------------------- constraints */

static const double CP_DEFAULT_ERROR_BIAS_BASE = (double)(1.0f - 0.1f);

void cpPivotJointInit(cpConstraint* c, cpBody* a, cpBody* b, cpv pivot) {
  memset(c, 0, sizeof(*c));
  c->type = CP_JOINT_PIVOT; c->a = a; c->b = b;
  c->errorBias = pow(CP_DEFAULT_ERROR_BIAS_BASE, 60.0);
  /* cpBodyWorldToLocal = inverse rigid transform */
  {
    cpv da = cpvsub(pivot, a->p), db = cpvsub(pivot, b->p);
    c->anchorA = cpv_(a->rot.x * da.x + a->rot.y * da.y, -a->rot.y * da.x + a->rot.x * da.y);
    c->anchorB = cpv_(b->rot.x * db.x + b->rot.y * db.y, -b->rot.y * db.x + b->rot.x * db.y);
  }
}

void cpRotaryLimitJointInit(cpConstraint* c, cpBody* a, cpBody* b, double mn, double mx) {
  memset(c, 0, sizeof(*c));
  c->type = CP_JOINT_ROTARY_LIMIT; c->a = a; c->b = b; c->min = mn; c->max = mx;
  c->errorBias = pow(CP_DEFAULT_ERROR_BIAS_BASE, 60.0);
}

static void constraint_prestep(cpConstraint* c, double dt) {
  cpBody *a = c->a, *b = c->b;
  double coef = 1.0 - pow(c->errorBias, dt);
  if (c->type == CP_JOINT_PIVOT) {
    double m_sum, k11, k12, k21, k22, det, det_inv;
    cpv delta;
    c->r1 = xform_vect(a, c->anchorA);
    c->r2 = xform_vect(b, c->anchorB);
    m_sum = a->m_inv + b->m_inv;
    k11 = m_sum; k12 = 0.0; k21 = 0.0; k22 = m_sum;
    {
      double r1xsq = c->r1.x * c->r1.x * a->i_inv, r1ysq = c->r1.y * c->r1.y * a->i_inv;
      double r1nxy = -c->r1.x * c->r1.y * a->i_inv;
      k11 += r1ysq; k12 += r1nxy; k21 += r1nxy; k22 += r1xsq;
    }
    {
      double r2xsq = c->r2.x * c->r2.x * b->i_inv, r2ysq = c->r2.y * c->r2.y * b->i_inv;
      double r2nxy = -c->r2.x * c->r2.y * b->i_inv;
      k11 += r2ysq; k12 += r2nxy; k21 += r2nxy; k22 += r2xsq;
    }
    det = k11 * k22 - k12 * k21;
    det_inv = 1.0 / det;
    c->k[0] = k22 * det_inv; c->k[1] = -k12 * det_inv; c->k[2] = -k21 * det_inv; c->k[3] = k11 * det_inv;
    delta = cpvsub(cpvadd(b->p, c->r2), cpvadd(a->p, c->r1));
    c->bias2 = cpvmult(delta, -coef / dt); /* maxBias = INFINITY: no clamp */
  } else {
    double dist = b->a - a->a, pdist = 0.0;
    if (dist > c->max) pdist = c->max - dist; else if (dist < c->min) pdist = c->min - dist;
    c->iSum = 1.0 / (a->i_inv + b->i_inv);
    c->bias = -coef * pdist / dt;
    if (c->bias == 0.0) c->jAcc = 0.0;
  }
}

/* warm start and iteration of the two joints with their multiply-adds fused, like the arbiters' (dynenv_math.h); the prestep
 * (K tensor, biases) is unfused: it runs once per substep, and the kernels take the pivot's K^-1 as a host-computed constant */
static void constraint_apply_cached(cpConstraint* c, double dt_coef) {
  if (c->type == CP_JOINT_PIVOT) {
    apply_impulses_f(c->a, c->b, c->r1, c->r2, cpvmult(c->jAcc2, dt_coef));
  } else {
    double j = c->jAcc * dt_coef;
    c->a->w = dm_fma(-j, c->a->i_inv, c->a->w);
    c->b->w = dm_fma(j, c->b->i_inv, c->b->w);
  }
}

static void constraint_apply_impulse(cpConstraint* c, double dt) {
  cpBody *a = c->a, *b = c->b;
  (void)dt;
  if (c->type == CP_JOINT_PIVOT) {
    cpv vr = relative_velocity_f(a, b, c->r1, c->r2);
    cpv d = cpvsub(c->bias2, vr);
    cpv j = cpv_(dm_fma(d.x, c->k[0], d.y * c->k[1]), dm_fma(d.x, c->k[2], d.y * c->k[3]));
    cpv jOld = c->jAcc2;
    c->jAcc2 = cpvadd(c->jAcc2, j); /* maxForce = INFINITY: no clamp */
    j = cpvsub(c->jAcc2, jOld);
    apply_impulses_f(a, b, c->r1, c->r2, j);
  } else {
    double wr, s, j, jOld;
    if (c->bias == 0.0) return;
    wr = b->w - a->w;
    jOld = c->jAcc;
    s = dm_fma(-(c->bias + wr), c->iSum, jOld);
    if (c->bias < 0.0) c->jAcc = cpfmax(s, 0.0); else c->jAcc = cpfmin(s, 0.0);
    j = c->jAcc - jOld;
    a->w = dm_fma(-j, a->i_inv, a->w);
    b->w = dm_fma(j, b->i_inv, b->w);
  }
}

/* ---------------------------------------------------------------- step */

static int pair_key(const cpArbiter* arb) {
  int sa = arb->a->slot, sb = arb->b->slot;
  return sa < sb ? sa * 64 + sb : sb * 64 + sa;
}

/* ---------------------------------------------------------------- test modes (tools / tests only; both off in every parity run)
 * cp_lite_test_reverse_order: colliding pairs are found in DESCENDING slot order instead of ascending - the same pair set, another
 *   arbiter (= callback and solver) order.  Chipmunk's own order comes out of its BB-tree and nobody here can know it; this is the
 *   other extreme, to measure what the order is worth (tools/pair_order_cost.py).
 * cp_lite_test_nudge: every dynamic body's velocity is multiplied by 1 + nudge * u, u in [-1, 1) a hash of (body, step), before the
 *   step - what another rounding of the same arithmetic does (nudge = 1e-15): the yardstick beside the reversed order.
 * cpSpace.trace_fn: called at the end of every step (tools/pymunk_crosscheck.py records per-substep states and the arbiter order). */
int cp_lite_test_reverse_order = 0;
double cp_lite_test_nudge = 0.0;
static double nudge_unit(unsigned a, unsigned b, unsigned c) {
  unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu ^ (c + 0x165667B1u) * 0xC2B2AE35u;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return (double)h / 2147483648.0 - 1.0;
}

void cpSpaceStep(cpSpace* s, double dt) {
  int i, j, it;
  double prev_dt;
  if (dt == 0.0) return;
  s->stamp++;
  if (cp_lite_test_nudge != 0.0)
    for (i = 0; i < s->n_bodies; ++i) {
      cpBody* b = s->bodies[i];
      b->v.x *= 1.0 + cp_lite_test_nudge * nudge_unit((unsigned)s->stamp, (unsigned)i, 0u);
      b->v.y *= 1.0 + cp_lite_test_nudge * nudge_unit((unsigned)s->stamp, (unsigned)i, 1u);
      b->w *= 1.0 + cp_lite_test_nudge * nudge_unit((unsigned)s->stamp, (unsigned)i, 2u);
    }
  prev_dt = s->curr_dt;
  s->curr_dt = dt;
  for (i = 0; i < s->n_active; ++i) s->active[i]->state = CP_ARB_NORMAL;
  s->n_active = 0;

  /* integrate positions (cpBodyUpdatePosition) */
  for (i = 0; i < s->n_bodies; ++i) {
    cpBody* b = s->bodies[i];
    b->p = cpvadd(b->p, cpvmult(cpvadd(b->v, b->v_bias), dt));
    cpBodySetAngle(b, b->a + (b->w + b->w_bias) * dt);
    b->v_bias = cpv_(0.0, 0.0);
    b->w_bias = 0.0;
  }
  /* refresh world geometry of dynamic shapes, then collide all pairs in canonical slot order */
  for (i = 0; i < s->n_shapes; ++i)
    if (s->shapes[i]->body->type == CP_BODY_DYNAMIC) cpShapeCacheBB(s->shapes[i]);
  if (s->test_free_flight) { /* golden tests: the generator's Space stand-in detects and solves nothing */
    for (i = 0; i < s->n_bodies; ++i) {
      cpBody* b = s->bodies[i];
      if (b->velocity_func) b->velocity_func(b, s->gravity, pow(s->damping, dt), dt);
      else cpBodyUpdateVelocity(b, s->gravity, pow(s->damping, dt), dt);
    }
    return;
  }
  if (!cp_lite_test_reverse_order) {
    for (i = 0; i < s->n_shapes; ++i) {
      for (j = i + 1; j < s->n_shapes; ++j) {
        cpShape *a = s->shapes[i], *b = s->shapes[j];
        if (a->body->type != CP_BODY_DYNAMIC && b->body->type != CP_BODY_DYNAMIC) continue;
        collide_shapes(s, a, b);
      }
    }
  } else { /* test mode: the same pairs (a = lower slot), last pair first */
    for (i = s->n_shapes - 1; i >= 0; --i) {
      for (j = s->n_shapes - 1; j > i; --j) {
        cpShape *a = s->shapes[i], *b = s->shapes[j];
        if (a->body->type != CP_BODY_DYNAMIC && b->body->type != CP_BODY_DYNAMIC) continue;
        collide_shapes(s, a, b);
      }
    }
  }
  /* cpSpaceArbiterSetFilter in canonical pair order: separate callbacks + expiry */
  {
    int order[CP_MAX_ARBITERS], n = 0, k;
    for (i = 0; i < CP_MAX_ARBITERS; ++i)
      if (s->pool[i].used) {
        int key = pair_key(&s->pool[i]);
        k = n++;
        while (k > 0 && pair_key(&s->pool[order[k - 1]]) > key) { order[k] = order[k - 1]; --k; }
        order[k] = i;
      }
    for (k = 0; k < n; ++k) {
      cpArbiter* arb = &s->pool[order[cp_lite_test_reverse_order ? n - 1 - k : k]];
      int ticks = s->stamp - arb->stamp;
      if (ticks >= 1 && arb->state != CP_ARB_CACHED) {
        arb->state = CP_ARB_CACHED;
        if (arb->handler && arb->handler->separate) arb->handler->separate(arb, s, arb->handler->data);
      }
      if (ticks >= s->collisionPersistence) arb->used = 0;
    }
  }
  /* prestep */
  {
    double slop = s->collisionSlop;
    double biasCoef = 1.0 - pow(s->collisionBias, dt);
    for (i = 0; i < s->n_active; ++i) arbiter_prestep(s->active[i], dt, slop, biasCoef);
    for (i = 0; i < s->n_constraints; ++i) constraint_prestep(s->constraints[i], dt);
  }
  /* integrate velocities */
  {
    double damping = pow(s->damping, dt);
    for (i = 0; i < s->n_bodies; ++i) {
      cpBody* b = s->bodies[i];
      if (b->velocity_func) b->velocity_func(b, s->gravity, damping, dt);
      else cpBodyUpdateVelocity(b, s->gravity, damping, dt);
    }
  }
  /* warm start */
  {
    double dt_coef = (prev_dt == 0.0 ? 0.0 : dt / prev_dt);
    for (i = 0; i < s->n_active; ++i) arbiter_apply_cached(s->active[i], dt_coef);
    for (i = 0; i < s->n_constraints; ++i) constraint_apply_cached(s->constraints[i], dt_coef);
  }
  /* sequential impulses */
  for (it = 0; it < s->iterations; ++it) {
    for (j = 0; j < s->n_active; ++j) arbiter_apply_impulse(s->active[j]);
    for (j = 0; j < s->n_constraints; ++j) constraint_apply_impulse(s->constraints[j], dt);
  }
  /* post-solve callbacks */
  for (i = 0; i < s->n_active; ++i) {
    cpArbiter* arb = s->active[i];
    if (arb->handler->post_solve) arb->handler->post_solve(arb, s, arb->handler->data);
  }
  if (s->trace_fn) s->trace_fn(s, s->trace_data);
}

/* ---------------------------------------------------------------- golden-test hook */
int cpSpaceTestCallback(cpSpace* s, int slotA, int slotB, int which) {
  cpShape *a = 0, *b = 0, *t;
  cpArbiter arb;
  int i;
  for (i = 0; i < s->n_shapes; ++i) {
    if (s->shapes[i]->slot == slotA) a = s->shapes[i];
    if (s->shapes[i]->slot == slotB) b = s->shapes[i];
  }
  if (!a || !b) return -1;
  if (a->type > b->type) { t = a; a = b; b = t; } /* cpCollide: shape types ascending */
  memset(&arb, 0, sizeof(arb));
  arb.a = a; arb.b = b; arb.body_a = a->body; arb.body_b = b->body;
  for (i = 0; i < s->n_handlers; ++i) {
    cpHandler* h = &s->handlers[i];
    if ((h->typeA == a->collision_type && h->typeB == b->collision_type) ||
        (h->typeA == b->collision_type && h->typeB == a->collision_type)) {
      arb.handler = h;
      arb.swapped = (a->collision_type != h->typeA);
      if (which == 0) return h->begin ? h->begin(&arb, s, h->data) : 1;
      if (which == 1 && h->post_solve) h->post_solve(&arb, s, h->data);
      if (which == 2 && h->separate) h->separate(&arb, s, h->data);
      return 1;
    }
  }
  return 1;
}

/* ---------------------------------------------------------------- queries */

int cpSpacePointQuery(cpSpace* s, cpv p, double maxDist, cpShape** out, int cap) {
  int i, n = 0;
  for (i = 0; i < s->n_shapes; ++i) {
    cpShape* sh = s->shapes[i];
    double d;
    if (sh->type == CP_SHAPE_CIRCLE) {
      cpv delta = cpvsub(p, sh->tc);
      d = dm_sqrt(cpvdot(delta, delta)) - sh->r;
    } else if (sh->type == CP_SHAPE_SEGMENT) {
      cpv seg_delta = cpvsub(sh->tb, sh->ta);
      double t = cpfclamp01(cpvdot(seg_delta, cpvsub(p, sh->ta)) / cpvlengthsq(seg_delta));
      cpv closest = cpvadd(sh->ta, cpvmult(seg_delta, t));
      cpv delta = cpvsub(p, closest);
      d = dm_sqrt(cpvdot(delta, delta)) - sh->r;
    } else {
      continue; /* polys are never queried in DynEnv (RoboCup has none) */
    }
    if (d < maxDist && n < cap) out[n++] = sh;
  }
  return n;
}
