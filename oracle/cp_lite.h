/* cp_lite.h — ORACLE (test infrastructure, never shipped, never on the product path).
 *
 * Scalar fp64 restatement of the subset of pymunk 5.x / Chipmunk2D 7.0.x that DynEnv exercises
 * through `space.step(1/100)` (DrivingEnvironment.py:278, RoboCupEnvironment.py:482,
 * environment_base.py:126-128,179-188).  pymunk/Chipmunk are NOT vendored in /root/reference and are not
 * installed in the build image, so this follows the published Chipmunk2D 7 algorithm as laid out in
 * SURVEY.md Appendix A:  PARITY UNPINNED against pymunk / Chipmunk itself (no golden vectors exist anywhere in the
 * reference, neither library is in this pipeline).  What stands behind it instead: closed-form KATs; tests/kat_general.py, a second,
 * independent restatement by Chipmunk's own route (GJK / EPA), which agrees with this file to 1e-9 on 5 x 10^5 shape pairs and 4500
 * scenes or differs in a catalogued way (tests/test_kat_general.py); and the reference's own step() code run on that
 * restatement through collisions (tests/golden/gen_golden_contacts.py).  The DynEnv game logic layered on top is pinned by
 * tests/golden (fixtures computed by the reference's own Python).
 *
 * Deliberate deviations (documented in DESIGN.md):
 *  - broadphase = all pairs in canonical (slot_lo, slot_hi) order instead of Chipmunk's BB-tree order
 *    (same pair SET, order differs; the order only permutes Gauss-Seidel sweeps / callback order);
 *  - poly-poly / circle-poly narrowphase = SAT min-penetration axis + Chipmunk's ContactPoints edge clip
 *    instead of GJK/EPA (equivalent for the radius-0 convex boxes DynEnv uses);
 *  - contact hashes are (slot,feature) pairs instead of pointer hashes (same matching semantics).
 */
#ifndef CP_LITE_H
#define CP_LITE_H

#include <stdint.h>

typedef struct { double x, y; } cpv;

enum { CP_BODY_DYNAMIC = 0, CP_BODY_STATIC = 2 };
enum { CP_SHAPE_CIRCLE = 0, CP_SHAPE_SEGMENT = 1, CP_SHAPE_POLY = 2 };
enum { CP_ARB_FIRST = 0, CP_ARB_NORMAL = 1, CP_ARB_IGNORE = 2, CP_ARB_CACHED = 3 };
enum { CP_JOINT_PIVOT = 0, CP_JOINT_ROTARY_LIMIT = 1 };

#define CP_MAX_SHAPES 64
#define CP_MAX_ARBITERS 256
#define CP_MAX_CONSTRAINTS 32
#define CP_MAX_HANDLERS 8

struct cpBody;
struct cpSpace;
struct cpArbiter;
typedef void (*cpVelocityFunc)(struct cpBody* body, cpv gravity, double damping, double dt);

typedef struct cpBody {
  int type;
  double m, i, m_inv, i_inv;
  cpv p, v, f;
  double a, w, t;
  cpv rot; /* (cos a, sin a) */
  cpv v_bias;
  double w_bias;
  cpVelocityFunc velocity_func; /* NULL = cpBodyUpdateVelocity */
  void* user;
} cpBody;

typedef struct { cpv v0, n; } cpPlane;

typedef struct cpShape {
  int type;
  cpBody* body;
  double bb_l, bb_b, bb_r, bb_t;
  double e, u;
  int collision_type;
  int slot; /* canonical order + hash id */
  void* user;
  double r;           /* circle / capsule radius (0 for polys) */
  cpv c, tc;          /* circle centre local / world */
  cpv sa, sb, sn;     /* segment local a, b, normal */
  cpv ta, tb, tn;     /* segment world */
  int count;          /* poly */
  cpPlane local[4], world[4];
} cpShape;

typedef struct {
  cpv r1, r2;
  double nMass, tMass, bounce, jnAcc, jtAcc, jBias, bias;
  uint32_t hash;
} cpContact;

typedef int (*cpBeginFunc)(struct cpArbiter* arb, struct cpSpace* space, void* data);
typedef void (*cpArbFunc)(struct cpArbiter* arb, struct cpSpace* space, void* data);

typedef struct cpHandler {
  int typeA, typeB;
  cpBeginFunc begin;
  cpArbFunc post_solve;
  cpArbFunc separate;
  void* data;
} cpHandler;

typedef struct cpArbiter {
  int used;
  cpShape *a, *b; /* narrowphase order (shape type ascending) */
  cpBody *body_a, *body_b;
  double e, u;
  cpv n;
  int count;
  cpContact contacts[2];
  int state;
  int stamp;
  cpHandler* handler;
  int swapped;
} cpArbiter;

typedef struct cpConstraint {
  int type;
  cpBody *a, *b;
  double errorBias; /* default pow(1-0.1f, 60) */
  int in_space;
  /* pivot */
  cpv anchorA, anchorB, r1, r2, jAcc2, bias2;
  double k[4];
  /* rotary limit */
  double min, max, iSum, bias, jAcc;
} cpConstraint;

typedef struct cpSpace {
  int iterations;
  double collisionSlop, collisionBias;
  int collisionPersistence;
  int stamp;
  double curr_dt;
  cpv gravity;
  double damping;
  int n_bodies;
  cpBody* bodies[CP_MAX_SHAPES]; /* dynamic bodies, add order */
  int n_shapes;
  cpShape* shapes[CP_MAX_SHAPES]; /* kept sorted by slot */
  cpArbiter pool[CP_MAX_ARBITERS];
  int n_active;
  cpArbiter* active[CP_MAX_ARBITERS];
  int n_constraints;
  cpConstraint* constraints[CP_MAX_CONSTRAINTS];
  int n_handlers;
  cpHandler handlers[CP_MAX_HANDLERS];
  cpHandler default_handler;
  int overflow; /* set if a fixed capacity was exceeded */
  int degenerate; /* set (until the space is rebuilt) when two capsule CORES were exactly collinear / exactly touching: the sign of the
                     minimum-translation normal segment_to_segment takes for crossing cores is a convention there (cores_crossing_normal in
                     cp_lite.c); error bit 4 of dynenv_error_flags */
  void (*trace_fn)(struct cpSpace*, void*); /* tools only: called at the end of every cpSpaceStep (NULL: nothing) */
  void* trace_data;
  int test_free_flight; /* golden tests only: cpSpaceStep = position update + velocity functions, nothing collides, no
                           constraint is solved (what the generator scripts' Space stand-in does, tests/golden/gen_golden.py) */
} cpSpace;

/* vector helpers */
static inline cpv cpv_(double x, double y) { cpv v; v.x = x; v.y = y; return v; }
static inline cpv cpvadd(cpv a, cpv b) { return cpv_(a.x + b.x, a.y + b.y); }
static inline cpv cpvsub(cpv a, cpv b) { return cpv_(a.x - b.x, a.y - b.y); }
static inline cpv cpvneg(cpv a) { return cpv_(-a.x, -a.y); }
static inline cpv cpvmult(cpv a, double s) { return cpv_(a.x * s, a.y * s); }
static inline double cpvdot(cpv a, cpv b) { return a.x * b.x + a.y * b.y; }
static inline double cpvcross(cpv a, cpv b) { return a.x * b.y - a.y * b.x; }
static inline cpv cpvperp(cpv a) { return cpv_(-a.y, a.x); }
static inline cpv cpvrotate(cpv a, cpv b) { return cpv_(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
static inline double cpvlengthsq(cpv a) { return cpvdot(a, a); }
static inline cpv cpvlerp(cpv a, cpv b, double t) { return cpvadd(cpvmult(a, 1.0 - t), cpvmult(b, t)); }

void cpSpaceInit(cpSpace* s);
void cpBodyInit(cpBody* b, double m, double i, int type);
void cpBodySetAngle(cpBody* b, double a);
void cpCircleInit(cpShape* s, cpBody* body, double r, int slot);
void cpSegmentInit(cpShape* s, cpBody* body, cpv a, cpv b, double r, int slot);
void cpBoxInit(cpShape* s, cpBody* body, double hx, double hy, int slot); /* verts (+-hx, +-hy) */
void cpShapeCacheBB(cpShape* s);
void cpSpaceAddBody(cpSpace* s, cpBody* b);
void cpSpaceAddShape(cpSpace* s, cpShape* sh);
void cpSpaceAddConstraint(cpSpace* s, cpConstraint* c);
void cpSpaceRemoveConstraint(cpSpace* s, cpConstraint* c);
cpHandler* cpSpaceAddHandler(cpSpace* s, int typeA, int typeB);
void cpPivotJointInit(cpConstraint* c, cpBody* a, cpBody* b, cpv pivot_world);
void cpRotaryLimitJointInit(cpConstraint* c, cpBody* a, cpBody* b, double min, double max);
void cpSpaceStep(cpSpace* s, double dt);
void cpBodyUpdateVelocity(cpBody* b, cpv gravity, double damping, double dt);
void cpBodyApplyForceAtWorldPoint(cpBody* b, cpv force, cpv point);
/* handler-ordered shapes (arbiter.shapes in pymunk) */
void cpArbiterGetShapes(const cpArbiter* arb, cpShape** a, cpShape** b);
double cpMomentForBox(double m, double hx, double hy); /* moment_for_poly on Car.points order */
double cpMomentForCircle(double m, double r1, double r2);
double cpMomentForSegment(double m, cpv a, cpv b, double r);
/* space.point_query(p, maxDist, filter=all): shapes whose surface distance < maxDist */
int cpSpacePointQuery(cpSpace* s, cpv p, double maxDist, cpShape** out, int cap);
/* golden tests only: run one handler callback of the shape pair (slots) the way cpSpaceStep would hand it over:
 * which = 0 begin (returns its value), 1 post_solve, 2 separate; -1 if a shape is missing, 1 without a handler */
int cpSpaceTestCallback(cpSpace* s, int slotA, int slotB, int which);

/* differential tests only: one pair through the narrowphase (out[15], layout at the definition) */
void cpTestCollide(cpShape* a, cpShape* b, double* out);

#endif
