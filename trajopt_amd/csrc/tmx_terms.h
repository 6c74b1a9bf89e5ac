// tmx_terms.h — device kinematics, exact term evaluation (K6) and convexification (K1 FD-Jacobian of the
// Cartesian-pose error, K3 collision linearisation).  One workgroup per problem; threads stride over cart-pose
// instances / contact slots.  The waypoint's DOF vector is read coalesced from HBM (D contiguous doubles).
//
// Reference behaviour restated (paths relative to the reference checkout):
//   CartPoseErrCalculator / CartPoseJacCalculator   trajopt/src/kinematic_terms.cpp:250-263,348-366 (FD, eps 1e-5)
//   ConstraintFromErrFunc / CostFromErrFunc::convex trajopt_sco/src/modeling_utils.cpp:168-211,247-269
//   CollisionEvaluator::GetGradient / CollisionsToDistanceExpressions  trajopt/src/collision_terms.cpp:203-250,343-383
//   CollisionCost::convex / value                   trajopt/src/collision_terms.cpp:1283-1327
//   JointVelEqCost / JointPosEqConstraint           trajopt/src/trajectory_costs.cpp:139-183,257-301
#pragma once
#include "tmx_types.h"
#include "../../include/tmx_detmath.h"
#include "../../include/tmx_expr.h"     // tmx_expr programs (function terms), shared with the oracle
#include "../../include/tmx_geom.h"     // sphere / capsule obstacle contacts, shared with the oracle  // sin / cos / atan2 with a fixed IEEE operation sequence, shared with the oracle

#define TMX_EPS_FD 1e-5       // sco DEFAULT_EPSILON, trajopt_sco/src/modeling_utils.cpp:13
#define TMX_CLEANUP_TOL 1e-7  // sco::cleanupAff, trajopt_sco/src/expr_ops.cpp:91

struct Tf3
{
  double R[9];
  double t[3];
};

TMX_DEVFN void tf_from12(const double* a, Tf3& T)
{
  for (int r = 0; r < 3; ++r)
  {
    for (int c = 0; c < 3; ++c)
      T.R[3 * r + c] = a[4 * r + c];
    T.t[r] = a[4 * r + 3];
  }
}
TMX_DEVFN void tf_mul(const Tf3& A, const Tf3& B, Tf3& C)
{
  for (int r = 0; r < 3; ++r)
  {
    for (int c = 0; c < 3; ++c)
      C.R[3 * r + c] = A.R[3 * r + 0] * B.R[0 + c] + A.R[3 * r + 1] * B.R[3 + c] + A.R[3 * r + 2] * B.R[6 + c];
    C.t[r] = A.R[3 * r + 0] * B.t[0] + A.R[3 * r + 1] * B.t[1] + A.R[3 * r + 2] * B.t[2] + A.t[r];
  }
}
TMX_DEVFN void tf_inv(const Tf3& A, Tf3& C)
{
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      C.R[3 * r + c] = A.R[3 * c + r];
  for (int r = 0; r < 3; ++r)
    C.t[r] = -(C.R[3 * r + 0] * A.t[0] + C.R[3 * r + 1] * A.t[1] + C.R[3 * r + 2] * A.t[2]);
}
TMX_DEVFN void tf_joint_motion(const double* ax, int type, double q, Tf3& T)
{
  for (int k = 0; k < 9; ++k)
    T.R[k] = 0.0;
  T.R[0] = T.R[4] = T.R[8] = 1.0;
  T.t[0] = T.t[1] = T.t[2] = 0.0;
  if (type == 0)
  {
    double c, s;
    tmx_sincos(q, &s, &c);
    const double v = 1.0 - c;
    const double x = ax[0], y = ax[1], z = ax[2];
    T.R[0] = c + x * x * v;
    T.R[1] = x * y * v - z * s;
    T.R[2] = x * z * v + y * s;
    T.R[3] = y * x * v + z * s;
    T.R[4] = c + y * y * v;
    T.R[5] = y * z * v - x * s;
    T.R[6] = z * x * v - y * s;
    T.R[7] = z * y * v + x * s;
    T.R[8] = c + z * z * v;
  }
  else
  {
    T.t[0] = ax[0] * q;
    T.t[1] = ax[1] * q;
    T.t[2] = ax[2] * q;
  }
}

// world transform of link `upto` (child of joint `upto`).  The joint values come from a callable (joint index -> value)
// and the joint frames before motion go to a visitor: no per-thread array is ever indexed with a run-time subscript,
// so nothing of the kinematics lives in scratch memory (a Tf3 jf[TMX_MAX_DOF] alone was 1.5 KB per lane).
template <class QAt, class Visit>
TMX_DEVFN void fk_link_visit(const DevProblem* P, QAt&& qat, int upto, Tf3& out, Visit&& visit)
{
  Tf3 T, O, M, U;
  tf_from12(P->base, T);
  for (int k = 0; k <= upto; ++k)
  {
    tf_from12(P->origin[k], O);
    tf_mul(T, O, U);
    visit(k, U);
    tf_joint_motion(P->axis[k], P->jtype[k], qat(k), M);
    tf_mul(U, M, T);
  }
  out = T;
}
template <class QAt>
TMX_DEVFN void fk_link_at(const DevProblem* P, QAt&& qat, int upto, Tf3& out)
{
  fk_link_visit(P, qat, upto, out, [](int, const Tf3&) {});
}
TMX_DEVFN void fk_link(const DevProblem* P, const double* q, int upto, Tf3& out)
{
  fk_link_at(P, [q](int k) { return q[k]; }, upto, out);
}
template <class QAt>
TMX_DEVFN void fk_tool_at(const DevProblem* P, QAt&& qat, Tf3& out)
{
  Tf3 L, Tl;
  fk_link_at(P, qat, P->D - 1, L);
  tf_from12(P->tool, Tl);
  tf_mul(L, Tl, out);
}
TMX_DEVFN void fk_tool(const DevProblem* P, const double* q, Tf3& out)
{
  fk_tool_at(P, [q](int k) { return q[k]; }, out);
}
// -n . (translational Jacobian column of joint k at world point p), joint frame F before motion
// (trajopt_common::getGradient: gradient of the signed distance w.r.t. joint k, collision_utils.cpp:116-221)
TMX_DEVFN double contact_grad_col(const DevProblem* P, int k, const Tf3& F, const double p[3], const double n[3])
{
  double z[3];
  for (int rr = 0; rr < 3; ++rr)
    z[rr] = F.R[3 * rr + 0] * P->axis[k][0] + F.R[3 * rr + 1] * P->axis[k][1] + F.R[3 * rr + 2] * P->axis[k][2];
  double col[3];
  if (P->jtype[k] == 0)
  {
    const double dd[3] = { p[0] - F.t[0], p[1] - F.t[1], p[2] - F.t[2] };
    col[0] = z[1] * dd[2] - z[2] * dd[1];
    col[1] = z[2] * dd[0] - z[0] * dd[2];
    col[2] = z[0] * dd[1] - z[1] * dd[0];
  }
  else
  {
    col[0] = z[0];
    col[1] = z[1];
    col[2] = z[2];
  }
  return -1.0 * (n[0] * col[0] + n[1] * col[1] + n[2] * col[2]);
}

// tesseract::common::calcRotationalErrorDecomposed [NOT IN REFERENCE] — same statement as oracle/trajprob.hpp
TMX_DEVFN void rot_err_decomposed(const double* R, double axis[3], double& angle)
{
  double qw, qx, qy, qz;
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0.0)
  {
    double t = sqrt(tr + 1.0);
    qw = 0.5 * t;
    t = 0.5 / t;
    qx = (R[7] - R[5]) * t;
    qy = (R[2] - R[6]) * t;
    qz = (R[3] - R[1]) * t;
  }
  else
  {
    // largest diagonal entry i, then (i, j, k) cyclic: spelled out per case so that R is only indexed with constants
    int i = 0;
    if (R[4] > R[0])
      i = 1;
    if (R[8] > (i == 1 ? R[4] : R[0]))
      i = 2;
    if (i == 0)
    {
      double t = sqrt(R[0] - R[4] - R[8] + 1.0);
      qx = 0.5 * t;
      t = 0.5 / t;
      qw = (R[7] - R[5]) * t;
      qy = (R[3] + R[1]) * t;
      qz = (R[6] + R[2]) * t;
    }
    else if (i == 1)
    {
      double t = sqrt(R[4] - R[8] - R[0] + 1.0);
      qy = 0.5 * t;
      t = 0.5 / t;
      qw = (R[2] - R[6]) * t;
      qz = (R[7] + R[5]) * t;
      qx = (R[1] + R[3]) * t;
    }
    else
    {
      double t = sqrt(R[8] - R[0] - R[4] + 1.0);
      qz = 0.5 * t;
      t = 0.5 / t;
      qw = (R[3] - R[1]) * t;
      qx = (R[2] + R[6]) * t;
      qy = (R[5] + R[7]) * t;
    }
  }
  double n = sqrt(qx * qx + qy * qy + qz * qz);
  double ang, ax[3];
  if (n != 0.0)
  {
    ang = 2.0 * tmx_atan2(n, fabs(qw));
    if (qw < 0)
      n = -n;
    ax[0] = qx / n;
    ax[1] = qy / n;
    ax[2] = qz / n;
  }
  else
  {
    ang = 0.0;
    ax[0] = 1.0;
    ax[1] = 0.0;
    ax[2] = 0.0;
  }
  const double s = ((qx * ax[0] + qy * ax[1] + qz * ax[2]) < 0) ? -1.0 : 1.0;
  const double two_pi = 2.0 * M_PI;
  double a = s * ang;
  a = copysign(fmod(fabs(a), two_pi), a);
  if (a < -M_PI)
    a += two_pi;
  else if (a > M_PI)
    a -= two_pi;
  axis[0] = s * ax[0];
  axis[1] = s * ax[1];
  axis[2] = s * ax[2];
  angle = a;
}
// tesseract::common::calcTransformError(target, source)
TMX_DEVFN void transform_error(const Tf3& tinv, const Tf3& src, double err[6], double ax[3], double& ang)
{
  Tf3 pe;
  tf_mul(tinv, src, pe);
  rot_err_decomposed(pe.R, ax, ang);
  err[0] = pe.t[0];
  err[1] = pe.t[1];
  err[2] = pe.t[2];
  err[3] = ax[0] * ang;
  err[4] = ax[1] * ang;
  err[5] = ax[2] * ang;
}

// column k of the translational geometric Jacobian of world point p (rigidly attached to a link behind joint k), joint frame F
// before motion: z x (p - o) for a revolute joint, z for a prismatic one (same statement as oracle Chain::jacobianPoint)
TMX_DEVFN void jac_point_col(const DevProblem* P, int k, const Tf3& F, const double p[3], double col[3])
{
  double z[3];
  for (int rr = 0; rr < 3; ++rr)
    z[rr] = F.R[3 * rr + 0] * P->axis[k][0] + F.R[3 * rr + 1] * P->axis[k][1] + F.R[3 * rr + 2] * P->axis[k][2];
  if (P->jtype[k] == 0)
  {
    const double dd[3] = { p[0] - F.t[0], p[1] - F.t[1], p[2] - F.t[2] };
    col[0] = z[1] * dd[2] - z[2] * dd[1];
    col[1] = z[2] * dd[0] - z[0] * dd[2];
    col[2] = z[0] * dd[1] - z[1] * dd[0];
  }
  else
  {
    col[0] = z[0];
    col[1] = z[1];
    col[2] = z[2];
  }
}
#if TMX_LINK_ROWS
// CartVelErrCalculator (trajopt/src/kinematic_terms.cpp:411-426), row i of segment (q0, q1):
//   i < 3: (p1 - p0)[i] - limit      i >= 3: (p0 - p1)[i - 3] - limit      (p = origin of the tool frame)
TMX_DEVFN double cart_vel_value(const DevProblem* P, const double* q0, const double* q1, int i, double limit, Tf3& s0, Tf3& s1)
{
  fk_tool(P, q0, s0);
  fk_tool(P, q1, s1);
  // (select chains, not t[c]: a run-time subscript would put both transforms in scratch memory)
  const int c = i < 3 ? i : i - 3;
  const double p0 = (c == 0) ? s0.t[0] : (c == 1) ? s0.t[1] : s0.t[2];
  const double p1 = (c == 0) ? s1.t[0] : (c == 1) ? s1.t[1] : s1.t[2];
  return (i < 3) ? (p1 - p0) - limit : (p0 - p1) - limit;
}
#endif

// the capsule-link / box-obstacle contact functions of include/tmx_geom.h out of line on the device: they are cold, and inlined
// their golden-section search grows the persistent kernel's collision code for every problem (config 1 -1 % with it inline)
#if TMX_IS_DEVICE
__device__ __attribute__((noinline)) static int link_closest_b_nl(const double* c, const double* e, const double* oc, const double* oa, const double* ob,
                                                                  const double* mesh, double* p, double* q)
{
  return tmx_link_closest_to_obstacle_b(c, e, oc, oa, ob, mesh, p, q);
}
__device__ __attribute__((noinline)) static double swept_closest_b_nl(const double* ca, const double* e, const double* oc, const double* oa,
                                                                      const double* ob, const double* mesh, double* q, int* inside)
{
  return tmx_swept_closest_to_obstacle_b(ca, e, oc, oa, ob, mesh, q, inside);
}
// convex-hull link against an obstacle primitive (GJK / EPA, include/tmx_gjk.h): cold, a few KB of private arrays
__device__ __attribute__((noinline)) static int hull_closest_nl(const double* hv, int nv, const double* R0, const double* t0, const double* R1,
                                                                const double* t1, const double* oc, const double* oa, const double* ob,
                                                                const double* mesh, double* p, double* q, double* tau)
{
  return tmx_hull_closest_to_obstacle(hv, nv, R0, t0, R1, t1, oc, oa, ob, mesh, p, q, tau);
}
#else
#define link_closest_b_nl tmx_link_closest_to_obstacle_b
#define swept_closest_b_nl tmx_swept_closest_to_obstacle_b
#define hull_closest_nl tmx_hull_closest_to_obstacle
#endif

// sphere-vs-sphere signed distance for contact slot (link sphere s, obstacle o) at joint values q
// HULL: the instantiation carries the convex-hull link code (piecewise kernels of ST problems only)
template <bool HULL = false>
TMX_DEVFN double contact_distance(const DevProblem* P, const double* q, int s, int o, double n[3], double pw[3])
{
  Tf3 L;
  const int link = P->ls_link[s];
  fk_link(P, q, link, L);
  if constexpr (HULL)
    if (P->n_ls_hull > 0 && P->ls_hull[2 * s + 1] > 0)
    {
      double pc[3], oq[3];
      const int inside = hull_closest_nl(P->hull + 3 * P->ls_hull[2 * s], P->ls_hull[2 * s + 1], L.R, L.t, nullptr, nullptr, P->ob_center + 3 * o,
                                         P->ob_axis + 3 * o, P->n_ob_box > 0 ? P->ob_box + 12 * o : nullptr, P->mesh, pc, oq, nullptr);
      const double len = tmx_contact_normal(pc, oq, inside, n);
      const double rs = P->ls_radius[s];
      for (int r = 0; r < 3; ++r)
        pw[r] = pc[r] + rs * n[r];
      return len - rs - P->ob_radius[o];
    }
  double c[3], d[3];
  for (int r = 0; r < 3; ++r)
    c[r] = L.R[3 * r + 0] * P->ls_center[3 * s + 0] + L.R[3 * r + 1] * P->ls_center[3 * s + 1] +
           L.R[3 * r + 2] * P->ls_center[3 * s + 2] + L.t[r];
  double oq[3];  // closest point of the obstacle primitive to the sphere centre (the centre itself for a sphere)
  int inside = 0;  // the link core point lies inside a box obstacle's core
  if (P->n_ls_capsule > 0 || P->n_ob_box > 0)
  {
    // capsule link: the point of the link's segment closest to the obstacle takes the place of the centre; box obstacles: signed
    // distance with penetration (include/tmx_geom.h)
    double e[3], pc[3];
    bool capsule = false;
    if (P->n_ls_capsule > 0)
    {
      for (int r = 0; r < 3; ++r)
        e[r] = L.R[3 * r + 0] * P->ls_axis[3 * s + 0] + L.R[3 * r + 1] * P->ls_axis[3 * s + 1] + L.R[3 * r + 2] * P->ls_axis[3 * s + 2];
      capsule = P->ls_axis[3 * s + 0] != 0.0 || P->ls_axis[3 * s + 1] != 0.0 || P->ls_axis[3 * s + 2] != 0.0;
    }
    inside = link_closest_b_nl(c, capsule ? e : nullptr, P->ob_center + 3 * o, P->ob_axis + 3 * o,
                                            P->n_ob_box > 0 ? P->ob_box + 12 * o : nullptr, P->mesh, pc, oq);
    for (int r = 0; r < 3; ++r)
      c[r] = pc[r];
  }
  else
    tmx_obstacle_closest_to_point(P->ob_center + 3 * o, P->ob_axis + 3 * o, c, oq);
  (void)d;
  const double len = tmx_contact_normal(c, oq, inside, n);
  const double rs = P->ls_radius[s];
  for (int r = 0; r < 3; ++r)
    pw[r] = c[r] + rs * n[r];
  return len - rs - P->ob_radius[o];
}

// ---- LVS_DISCRETE / (LVS_)CONTINUOUS collision on a segment (q0 = x[t], q1 = x[t+1]) --------------------------------
// Same statement, operation by operation, as oracle/trajprob.hpp LvsEvaluator (DiscreteCollisionEvaluator /
// CastCollisionEvaluator::CalcCollisions trajopt/src/collision_terms.cpp:823-905, :1071-1173 on sphere geometry).
// Slot (t, sphere s, obstacle o, sub-state / sub-segment index i).
struct LvsContact
{
  double distance, n[3], p_local[3], cc_time;
  double R0[9], R1[9];  // rotation of the link in `transform` / `cc_transform`
};
TMX_DEVFN double lin_spaced_at(int size, double low, double high, int i)
{
  const int size1 = size - 1;
  const double step = (high - low) / (double)size1;
  const bool flip = fabs(high) < fabs(low);
  if (flip)
    return (i == 0) ? low : (high - (double)(size1 - i) * step);
  return (i == size1) ? high : (low + (double)i * step);
}
// returns true if the slot holds a contact of the (filtered) result vector
template <bool HULL = false>
TMX_DEVFN bool lvs_contact(const DevProblem* P, const double* q0, const double* q1, int r, LvsContact& c)
{
  const int s = P->slot_sub[r], o = P->slot_sub2[r], flags = P->slot_sub3[r];
  const int i = (flags >> 3) & 0x1FFF, kmax = flags >> 16;  // sub-state index; sub-state capacity of the slot's term
  const bool fixed0 = flags & 1, fixed1 = flags & 2, cast = flags & 4;
  const double lvs = P->slot_aux3[r], margin = P->slot_aux1[r], buffer = P->slot_aux2[r];
  double d2 = 0.0;
  for (int j = 0; j < P->DK; ++j)  // the joints (a time-parameterised problem's time column is not part of the state distance)
    d2 += (q1[j] - q0[j]) * (q1[j] - q0[j]);
  const double dist = sqrt(d2);
  int cnt = 2;
  if (dist > lvs)
    cnt = (int)ceil(dist / lvs) + 1;
  if (cnt > kmax)
    cnt = kmax;
  const bool split = dist > lvs;
  const int last = cnt - 1;
  const int n_sub = cast ? last : cnt;
  if (i >= n_sub)
    return false;
  const double dt = 1.0 / (double)last;
  const int link = P->ls_link[s];
  // sub-state joint values on the fly (no per-thread arrays): start and end state of sub-segment i
  auto qa = [=](int j) { return (!split && cast) ? q0[j] : lin_spaced_at(cnt, q0[j], q1[j], i); };
  auto qb = [=](int j) { return (!split && cast) ? q1[j] : lin_spaced_at(cnt, q0[j], q1[j], cast ? i + 1 : i); };
  Tf3 Ta, Tb;
  fk_link_at(P, qa, link, Ta);
  double ca[3], p[3];
  for (int rr = 0; rr < 3; ++rr)
    ca[rr] = Ta.R[3 * rr + 0] * P->ls_center[3 * s + 0] + Ta.R[3 * rr + 1] * P->ls_center[3 * s + 1] + Ta.R[3 * rr + 2] * P->ls_center[3 * s + 2] + Ta.t[rr];
  double tau = 0.0;
  double oq[3];  // closest point of the obstacle primitive (include/tmx_geom.h)
  int inside = 0;  // the link core point lies inside a box obstacle's core
  bool hull = false;
  if constexpr (HULL)
    hull = P->n_ls_hull > 0 && P->ls_hull[2 * s + 1] > 0;
  if (hull)
  {
    // convex-hull link: at the sub-state (discrete) or swept over the sub-segment (cast: the convex hull of both placements)
    if (cast)
      fk_link_at(P, qb, link, Tb);
    else
      Tb = Ta;
    inside = hull_closest_nl(P->hull + 3 * P->ls_hull[2 * s], P->ls_hull[2 * s + 1], Ta.R, Ta.t, cast ? Tb.R : nullptr, cast ? Tb.t : nullptr,
                             P->ob_center + 3 * o, P->ob_axis + 3 * o, P->n_ob_box > 0 ? P->ob_box + 12 * o : nullptr, P->mesh, p, oq,
                             cast ? &tau : nullptr);
  }
  else if (cast)
  {
    fk_link_at(P, qb, link, Tb);
    double cb[3];
    for (int rr = 0; rr < 3; ++rr)
      cb[rr] = Tb.R[3 * rr + 0] * P->ls_center[3 * s + 0] + Tb.R[3 * rr + 1] * P->ls_center[3 * s + 1] + Tb.R[3 * rr + 2] * P->ls_center[3 * s + 2] + Tb.t[rr];
    const double e[3] = { cb[0] - ca[0], cb[1] - ca[1], cb[2] - ca[2] };
    if (P->n_ob_box > 0)
      tau = swept_closest_b_nl(ca, e, P->ob_center + 3 * o, P->ob_axis + 3 * o, P->ob_box + 12 * o, P->mesh, oq, &inside);
    else
      tau = tmx_swept_closest_to_obstacle(ca, e, P->ob_center + 3 * o, P->ob_axis + 3 * o, oq);
    for (int rr = 0; rr < 3; ++rr)
      p[rr] = ca[rr] + tau * e[rr];
  }
  else
  {
    Tb = Ta;
    if (P->n_ls_capsule > 0 || P->n_ob_box > 0)
    {
      double ea[3];
      bool capsule = false;
      if (P->n_ls_capsule > 0)
      {
        for (int rr = 0; rr < 3; ++rr)
          ea[rr] = Ta.R[3 * rr + 0] * P->ls_axis[3 * s + 0] + Ta.R[3 * rr + 1] * P->ls_axis[3 * s + 1] + Ta.R[3 * rr + 2] * P->ls_axis[3 * s + 2];
        capsule = P->ls_axis[3 * s + 0] != 0.0 || P->ls_axis[3 * s + 1] != 0.0 || P->ls_axis[3 * s + 2] != 0.0;
      }
      inside = link_closest_b_nl(ca, capsule ? ea : nullptr, P->ob_center + 3 * o, P->ob_axis + 3 * o,
                                              P->n_ob_box > 0 ? P->ob_box + 12 * o : nullptr, P->mesh, p, oq);
    }
    else
    {
      for (int rr = 0; rr < 3; ++rr)
        p[rr] = ca[rr];
      tmx_obstacle_closest_to_point(P->ob_center + 3 * o, P->ob_axis + 3 * o, p, oq);
    }
  }
  const double len = tmx_contact_normal(p, oq, inside, c.n);
  const double rs = P->ls_radius[s];
  c.distance = len - rs - P->ob_radius[o];
  double pw[3];
  for (int rr = 0; rr < 3; ++rr)
    pw[rr] = p[rr] + rs * c.n[rr];
  for (int rr = 0; rr < 3; ++rr)
    c.p_local[rr] = Ta.R[0 + rr] * (pw[0] - Ta.t[0]) + Ta.R[3 + rr] * (pw[1] - Ta.t[1]) + Ta.R[6 + rr] * (pw[2] - Ta.t[2]);
  for (int k = 0; k < 9; ++k)
  {
    c.R0[k] = Ta.R[k];
    c.R1[k] = Tb.R[k];
  }
  // cc_type: 0 none, 1 Time0, 2 Time1, 3 Between (ContactResultMap::addInterpolatedCollisionResults)
  int raw = 0, type;
  if (cast)
    raw = (tau == 0.0) ? 1 : ((tau == 1.0) ? 2 : 3);
  if (!cast || split)
  {
    c.cc_time = cast ? ((double)i * dt) + (tau * dt) : ((double)i * dt);
    if (i == 0 && (raw == 0 || raw == 1))
      type = 1;
    else if (i == last && (raw == 0 || raw == 2))
      type = 2;
    else
      type = 3;
  }
  else
  {
    c.cc_time = tau;
    type = raw;
  }
  // trajopt_common::removeInvalidContactResults (collision_utils.cpp:71-114), link 1 static
  if (c.distance > (margin + buffer))
    return false;
  if (!fixed0 && !fixed1)
    return true;
  if (fixed0 && type != 0 && type != 1)
    return true;
  if (fixed1 && type != 0 && type != 2)
    return true;
  return false;
}
// GetGradient(dofvals, contact, isTimestep1) of the link sphere at end state q: scale * grad goes to sg[k * stride]
// (the caller's row in HBM: the raw gradient is staged there and post-processed in place) and scale * -(grad . q) is returned
TMX_DEVFN void lvs_end_gradient(const DevProblem* P, const double* q, int s, const LvsContact& c, bool is1, double* sg, double& sconst)
{
  const int D = P->D, link = P->ls_link[s];
  const double scale = is1 ? c.cc_time : (1 - c.cc_time);
  const double* Rl = is1 ? c.R1 : c.R0;
  // pass 1: the link frame -> world position of the contact point; pass 2: the joint frames again, one gradient entry each
  Tf3 L;
  fk_link(P, q, link, L);
  double p[3];
  for (int rr = 0; rr < 3; ++rr)
    p[rr] = L.t[rr] + (Rl[3 * rr + 0] * c.p_local[0] + Rl[3 * rr + 1] * c.p_local[1] + Rl[3 * rr + 2] * c.p_local[2]);
  double gq = 0.0;
  fk_link_visit(P, [q](int k) { return q[k]; }, link, L, [&](int k, const Tf3& F) {
    const double g = contact_grad_col(P, k, F, p, c.n);
    sg[k] = scale * g;
    gq += g * q[k];
  });
  for (int k = link + 1; k < D; ++k)
  {
    const double g = -1.0 * (c.n[0] * 0.0 + c.n[1] * 0.0 + c.n[2] * 0.0);
    sg[k] = scale * g;
    gq += g * q[k];
  }
  sconst = scale * -gq;
}

// ---- function terms (tmx_expr programs): sco::CostFromFunc / ConstraintFromErrFunc, trajopt_sco/src/modeling_utils.cpp ----------
// fx_nops[inst] < 0: a built-in kinematic function instead of a program (fx_op0 = link, parameters behind the row weights)
#define FX_AVOID_SINGULARITY (-1)  // consts: lambda, first joint of the subset + 1 (0: all joints)
#define FX_DYN_CART_POSE (-2)      // consts: link_T_target (12; world_T_target when fx_op0 < 0), row indices (6), tolerance flag, lower (6), upper (6)
TMX_DEVFN void fx_builtin_eval(const DevProblem* P, int inst, const double* x, double* out);
TMX_DEVFN void fx_eval(const DevProblem* P, int inst, const double* x, double* out)
{
  if (P->fx_nops[inst] < 0)
  {
    fx_builtin_eval(P, inst, x, out);
    return;
  }
  tmx_expr_eval(P->fx_ops + 2 * P->fx_op0[inst], P->fx_nops[inst], P->fx_consts + P->fx_c0[inst], x, out);
}
TMX_DEVFN double fx_eval1(const DevProblem* P, int inst, const double* x)
{
  double o[TMX_EXPR_MAX_OUT];
  fx_eval(P, inst, x, o);
  return o[0];
}
// forwardNumGrad(f, eps)(x)  (num_diff.cpp:41-54, sco/num_diff.hpp): g_i = (f(x + eps e_i) - f(x)) / eps; x is restored
TMX_DEVFN void fx_forward_grad(const DevProblem* P, int inst, double* x, int k, double* g)
{
  const double y = fx_eval1(P, inst, x);
  for (int i = 0; i < k; ++i)
  {
    const double xi = x[i];
    x[i] = xi + TMX_EPS_FD;
    const double yp = fx_eval1(P, inst, x);
    g[i] = (yp - y) / TMX_EPS_FD;
    x[i] = xi;
  }
}
// eigen-decomposition of the symmetric k x k matrix A (row-major, overwritten by its diagonal form), eigenvectors in the columns
// of V: cyclic Jacobi rotations.  Stands in for Eigen::SelfAdjointEigenSolver (modeling_utils.cpp:77): the projection on the
// positive eigenspace computed from it is a matrix function of A and does not depend on the method beyond rounding.
TMX_DEVFN void sym_eig_jacobi(double* A, double* V, int k)
{
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j)
      V[i * k + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep)
  {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < k; ++i)
      for (int j = 0; j < k; ++j)
        if (i != j)
          off += A[i * k + j] * A[i * k + j];
        else
          diag += A[i * k + j] * A[i * k + j];
    if (!(off > 1e-32 * (diag + off)) || off == 0.0)
      break;
    for (int p = 0; p < k - 1; ++p)
      for (int q = p + 1; q < k; ++q)
      {
        const double apq = A[p * k + q];
        if (apq == 0.0)
          continue;
        const double theta = (A[q * k + q] - A[p * k + p]) / (2.0 * apq);
        const double t = ((theta >= 0.0) ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
        for (int r = 0; r < k; ++r)  // A <- A J
        {
          const double arp = A[r * k + p], arq = A[r * k + q];
          A[r * k + p] = c * arp - sn * arq;
          A[r * k + q] = sn * arp + c * arq;
        }
        for (int r = 0; r < k; ++r)  // A <- J' A
        {
          const double apr = A[p * k + r], aqr = A[q * k + r];
          A[p * k + r] = c * apr - sn * aqr;
          A[q * k + r] = sn * apr + c * aqr;
        }
        for (int r = 0; r < k; ++r)
        {
          const double vrp = V[r * k + p], vrq = V[r * k + q];
          V[r * k + p] = c * vrp - sn * vrq;
          V[r * k + q] = sn * vrp + c * vrq;
        }
      }
  }
}
// ---- built-in kinematic functions of the function-term machinery -----------------------------------------------------------
// 6 x D geometric Jacobian (row-major; linear rows on top, angular below; reference point = origin of the link frame, base
// coordinates) of moving link `link`: what tesseract's JointGroup::calcJacobian(q, link_name) returns for a serial chain
TMX_DEVFN void link_jacobian6(const DevProblem* P, const double* q, int link, double* J)
{
  const int D = P->D;
  Tf3 L;
  fk_link(P, q, link, L);
  const double p[3] = { L.t[0], L.t[1], L.t[2] };
  for (int e = 0; e < 6 * D; ++e)
    J[e] = 0.0;
  fk_link_visit(P, [q](int k) { return q[k]; }, link, L, [&](int k, const Tf3& F) {
    double col[3];
    jac_point_col(P, k, F, p, col);
    for (int r = 0; r < 3; ++r)
      J[r * D + k] = col[r];
    if (P->jtype[k] == 0)
      for (int r = 0; r < 3; ++r)
        J[(3 + r) * D + k] = F.R[3 * r + 0] * P->axis[k][0] + F.R[3 * r + 1] * P->axis[k][1] + F.R[3 * r + 2] * P->axis[k][2];
  });
}
// smallest singular value of the 6 x D matrix J with its left / right singular vectors u (6), v (D) - the last triplet of
// Eigen::JacobiSVD(J, ComputeThinU | ComputeThinV) (kinematic_terms.cpp:589-593): one-sided Jacobi rotations (Hestenes) on the
// columns of the tall orientation, the same sequence of rotations as the oracle's thinSvd (a Gram-matrix eigen-solve would lose
// s_max^2 / s_min digits exactly where the term matters, near a singular posture).  u' dJ v does not depend on the common sign.
TMX_DEVFN double smallest_singular(const double* J, int D, double* u, double* v)
{
  const bool flip = 6 < D;  // tall orientation: m >= n
  const int m = flip ? D : 6, n = flip ? 6 : D;
  double G[TMX_MAX_DOF * 6], W[36];
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j)
      G[i * n + j] = flip ? J[j * D + i] : J[i * D + j];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      W[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 80; ++sweep)
  {
    bool rotated = false;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q)
      {
        double alpha = 0.0, beta = 0.0, gamma = 0.0;
        for (int i = 0; i < m; ++i)
        {
          alpha += G[i * n + p] * G[i * n + p];
          beta += G[i * n + q] * G[i * n + q];
          gamma += G[i * n + p] * G[i * n + q];
        }
        if (gamma == 0.0 || fabs(gamma) <= 1e-17 * sqrt(alpha * beta))
          continue;
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = ((zeta >= 0.0) ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
        for (int i = 0; i < m; ++i)
        {
          const double gp = G[i * n + p], gq = G[i * n + q];
          G[i * n + p] = c * gp - sn * gq;
          G[i * n + q] = sn * gp + c * gq;
        }
        for (int i = 0; i < n; ++i)
        {
          const double wp = W[i * n + p], wq = W[i * n + q];
          W[i * n + p] = c * wp - sn * wq;
          W[i * n + q] = sn * wp + c * wq;
        }
      }
    if (!rotated)
      break;
  }
  int jm = 0;
  double sv = 0.0;
  for (int j = 0; j < n; ++j)
  {
    double nn = 0.0;
    for (int i = 0; i < m; ++i)
      nn += G[i * n + j] * G[i * n + j];
    nn = sqrt(nn);
    if (j == 0 || nn <= sv)  // (the last of equal minima: where a stable descending sort leaves it)
    {
      sv = nn;
      jm = j;
    }
  }
  // J' = L S R' (flip) or J = L S R': left factor of the tall orientation = normalised column of G, right factor = column of W
  double* lv = flip ? v : u;  // m values
  double* rv = flip ? u : v;  // n values
  for (int i = 0; i < m; ++i)
    lv[i] = (sv > 0.0) ? G[i * n + jm] / sv : 0.0;
  for (int i = 0; i < n; ++i)
    rv[i] = W[i * n + jm];
  return sv;
}
// the Jacobian AvoidSingularity decomposes: all n_dof columns (j0p1 == 0, the problem's joint group) or the columns of joints
// j0 .. link, j0 = j0p1 - 1 (the subset group of AvoidSingularitySubsetErrCalculator: a rigid motion of everything upstream does not
// change the singular values, joints downstream of the link do not move it); returns the number of columns
TMX_DEVFN int sing_jacobian(const DevProblem* P, const double* q, int link, int j0p1, double* J)
{
  const int D = P->D;
  link_jacobian6(P, q, link, J);
  if (j0p1 == 0 && P->DK == D)
    return D;
  // (a time-parameterised problem, D = DK + 1: the time column is no joint of the group - the 6 x DK matrix the reference
  //  decomposes, not one with a zero column behind it, whose smallest singular value would be 0 for DK < 6)
  const int j0 = j0p1 ? j0p1 - 1 : 0, nc = j0p1 ? link - j0 + 1 : P->DK;
  for (int r = 0; r < 6; ++r)  // compaction in place: (r, c) moves down to a smaller index
    for (int c = 0; c < nc; ++c)
      J[r * nc + c] = J[r * D + j0 + c];
  return nc;
}
// target and source frames of a pose instance at q: link * offset (or the static world frame), tool
TMX_DEVFN void dyn_pose_frames(const DevProblem* P, int inst, const double* q, Tf3& tinv, Tf3& src)
{
  Tf3 L, off, tgt;
  tf_from12(P->fx_consts + P->fx_c0[inst], off);
  if (P->fx_op0[inst] >= 0)
  {
    fk_link(P, q, P->fx_op0[inst], L);
    tf_mul(L, off, tgt);
    tf_inv(tgt, tinv);
  }
  else
    tf_inv(off, tinv);
  fk_tool(P, q, src);
}
// tesseract::common::applyTolerances [NOT IN REFERENCE; call sites kinematic_terms.cpp:92, :234, :243]: the part of the error
// outside the band [lower, upper], zero inside
TMX_DEVFN void pose_apply_tolerances(const double* par, double err[6])
{
  if (par[18] == 0.0)
    return;
  for (int i = 0; i < 6; ++i)
  {
    const double lo = par[19 + i], up = par[25 + i];
    err[i] = (err[i] < lo) ? err[i] - lo : ((err[i] > up) ? err[i] - up : 0.0);
  }
}
TMX_DEVFN void fx_builtin_eval(const DevProblem* P, int inst, const double* x, double* out)
{
  const double* par = P->fx_consts + P->fx_c0[inst];
  if (P->fx_nops[inst] == FX_AVOID_SINGULARITY)
  {
    // AvoidSingularityErrCalculator::operator()  kinematic_terms.cpp:586-603 (subset form :644-653: the Jacobian of the joint subset)
    double J[6 * TMX_MAX_DOF], u[6], v[TMX_MAX_DOF];
    const int nc = sing_jacobian(P, x, P->fx_op0[inst], (int)par[1], J);
    const double sv = smallest_singular(J, nc, u, v);
    const double lambda = par[0];
    out[0] = 1.0 / (sv + lambda) - 1.0 / (0.1 + lambda);
    return;
  }
  // DynamicCartPoseErrCalculator::operator()  kinematic_terms.cpp:98-111
  Tf3 tinv, src;
  dyn_pose_frames(P, inst, x, tinv, src);
  double err[6], ax[3], ang;
  transform_error(tinv, src, err, ax, ang);
  pose_apply_tolerances(par, err);
  for (int i = 0; i < P->fx_nout[inst]; ++i)
    out[i] = err[(int)par[12 + i]];
}
// Jacobian of a built-in at x (perturbed in place and restored), rows J[o][.]
TMX_DEVFN void fx_builtin_jac(const DevProblem* P, int inst, double* x, double (*Jo)[TMX_MAX_DOF])
{
  const int D = P->D;
  const double* par = P->fx_consts + P->fx_c0[inst];
  if (P->fx_nops[inst] == FX_AVOID_SINGULARITY)
  {
    // AvoidSingularityJacCalculator::operator() / jacobianPartialDerivative  kinematic_terms.cpp:605-642 (eps_ = 1e-6,
    // kinematic_terms.hpp:371): d s_min / d q_k = u' (dJ / dq_k) v, Jacobian differenced forward
    // Subset form (:655-680): the gradient of the subset's joints, zero for the others.
    const double eps = 1.0e-6;
    double J0[6 * TMX_MAX_DOF], J1[6 * TMX_MAX_DOF], u[6], v[TMX_MAX_DOF];
    const int link = P->fx_op0[inst], j0p1 = (int)par[1];
    const int nc = sing_jacobian(P, x, link, j0p1, J0);
    const double sv = smallest_singular(J0, nc, u, v);
    const double lambda = par[0];
    const double scale = -1.0 / ((sv + lambda) * (sv + lambda));
    const int k0 = j0p1 ? j0p1 - 1 : 0, k1 = j0p1 ? link : P->DK - 1;  // (the time column of a time-parameterised problem: zero)
    for (int k = 0; k < D; ++k)
    {
      if (k < k0 || k > k1)
      {
        Jo[0][k] = 0.0;
        continue;
      }
      const double xk = x[k];
      x[k] = xk + eps;
      sing_jacobian(P, x, link, j0p1, J1);
      x[k] = xk;
      double acc = 0.0;
      for (int c = 0; c < nc; ++c)
      {
        double uc = 0.0;
        for (int r = 0; r < 6; ++r)
          uc += u[r] * ((J1[r * nc + c] - J0[r * nc + c]) / eps);
        acc += uc * v[c];
      }
      Jo[0][k] = acc * scale;
    }
    return;
  }
  // DynamicCartPoseJacCalculator::operator()  kinematic_terms.cpp:158-185: calcJacobianTransformErrorDiff(target, target',
  // source, source', lower, upper) / eps - both frames perturbed, the +-pi handling of the static CartPose Jacobian
  // (convexify_terms); with a tolerance band both errors pass through it before the difference
  Tf3 tinv, src, pe, pp;
  dyn_pose_frames(P, inst, x, tinv, src);
  tf_mul(tinv, src, pe);
  double ax0[3], a0;
  rot_err_decomposed(pe.R, ax0, a0);
  double e0[6] = { pe.t[0], pe.t[1], pe.t[2], ax0[0] * a0, ax0[1] * a0, ax0[2] * a0 };
  pose_apply_tolerances(par, e0);
  for (int k = 0; k < D; ++k)
  {
    const double xk = x[k];
    x[k] = xk + TMX_EPS_FD;
    dyn_pose_frames(P, inst, x, tinv, src);
    x[k] = xk;
    tf_mul(tinv, src, pp);
    double ax1[3], a1;
    rot_err_decomposed(pp.R, ax1, a1);
    double a1c = a1;
    if (a1 > M_PI_2 && a0 < -M_PI_2)
      a1c = a1 - 2.0 * M_PI;
    else if (a1 < -M_PI_2 && a0 > M_PI_2)
      a1c = a1 + 2.0 * M_PI;
    double e1[6] = { pp.t[0], pp.t[1], pp.t[2], ax1[0] * a1c, ax1[1] * a1c, ax1[2] * a1c };
    pose_apply_tolerances(par, e1);
    for (int i = 0; i < P->fx_nout[inst]; ++i)
    {
      const int r = (int)par[12 + i];
      Jo[i][k] = (e1[r] - e0[r]) / TMX_EPS_FD;
    }
  }
}

// CostFromFunc::convex (modeling_utils.cpp:52-113) for cost instance `inst` at the waypoint values q: quadratic model
//   c + g . x + sum_i (H_ii / 2) x_i^2 + sum_{i<j} H_ij x_i x_j   ->  H (k x k, diagonal form: off-diagonals zero), g, c
TMX_DEVFN void fx_convexify_cost(const DevProblem* P, int inst, const double* q, int k, double* H, double* g, double* cst, double* W)
{
  double x[TMX_MAX_DOF];
  for (int i = 0; i < k; ++i)
    x[i] = q[i];
  if (P->fx_kind[inst] == 3)
  {
    // CostFromErrFunc::convex, SQUARED (modeling_utils.cpp:166-190): per output i the linearised row aff = k + a . x
    // (affFromValGrad with the 1e-7 clean-up), quad = exprSquare(aff) scaled by the weight: constant w k^2, linear 2 w k a_j,
    // (j, j) coefficient w a_j^2, (j < l) coefficient 2 w a_j a_l - accumulated in the (H, g, c) form of this model
    double y[TMX_EXPR_MAX_OUT], yp[TMX_EXPR_MAX_OUT];
    double* J = W;  // n_out x k
    const int no = P->fx_nout[inst];
    fx_eval(P, inst, x, y);
    for (int i = 0; i < k; ++i)
    {
      const double xi = x[i];
      x[i] = xi + TMX_EPS_FD;
      fx_eval(P, inst, x, yp);
      for (int o = 0; o < no; ++o)
        J[o * k + i] = (yp[o] - y[o]) / TMX_EPS_FD;
      x[i] = xi;
    }
    for (int i = 0; i < k * k; ++i)
      H[i] = 0.0;
    for (int i = 0; i < k; ++i)
      g[i] = 0.0;
    double cacc = 0.0;
    const double* wts = P->fx_consts + P->fx_c0[inst] - TMX_EXPR_MAX_OUT;  // the instance's weights sit in front of its constants
    for (int o = 0; o < no; ++o)
    {
      const double w = wts[o];
      if (w == 0)
        continue;  // :175-176
      double dot = 0.0;
      for (int j = 0; j < k; ++j)
        dot += J[o * k + j] * x[j];
      const double kc = y[o] - dot;
      cacc += (kc * kc) * w;
      for (int j = 0; j < k; ++j)
      {
        const double aj = (fabs(J[o * k + j]) > TMX_CLEANUP_TOL) ? J[o * k + j] : 0.0;
        g[j] += (2 * kc * aj) * w;
        H[j * k + j] += 2.0 * ((aj * aj) * w);
        for (int l = j + 1; l < k; ++l)
        {
          const double al = (fabs(J[o * k + l]) > TMX_CLEANUP_TOL) ? J[o * k + l] : 0.0;
          const double v = (2 * aj * al) * w;
          H[j * k + l] += v;
          H[l * k + j] += v;
        }
      }
    }
    *cst = cacc;
    return;
  }
  if (P->fx_kind[inst] == 0)
  {
    // calcGradAndDiagHess (num_diff.cpp:70-91), hess = max(hess, 0)
    const double y = fx_eval1(P, inst, x);
    double gx = 0.0, xhx = 0.0;
    for (int i = 0; i < k * k; ++i)
      H[i] = 0.0;
    for (int i = 0; i < k; ++i)
    {
      const double xi = x[i];
      x[i] = xi + TMX_EPS_FD / 2;
      const double yplus = fx_eval1(P, inst, x);
      x[i] = xi - TMX_EPS_FD / 2;
      const double yminus = fx_eval1(P, inst, x);
      x[i] = xi;
      const double gi = (yplus - yminus) / TMX_EPS_FD;
      double hi = (yplus + yminus - 2 * y) / (TMX_EPS_FD * TMX_EPS_FD / 4);
      hi = fmax(hi, 0.0);
      H[i * k + i] = hi;
      g[i] = gi;
    }
    for (int i = 0; i < k; ++i)
    {
      gx += g[i] * x[i];
      xhx += x[i] * (H[i * k + i] * x[i]);
    }
    *cst = y - gx + .5 * xhx;
    for (int i = 0; i < k; ++i)
      g[i] = g[i] - H[i * k + i] * x[i];
    return;
  }
  // calcGradHess (num_diff.cpp:93-105): grad = forward gradient, hess = forward Jacobian of the forward gradient, symmetrised
  double* Hn = W;          // k x k
  double* V = W + k * k;   // k x k
  double g0[TMX_MAX_DOF], g1[TMX_MAX_DOF];
  const double y = fx_eval1(P, inst, x);
  fx_forward_grad(P, inst, x, k, g0);
  for (int i = 0; i < k; ++i)
  {
    const double xi = x[i];
    x[i] = xi + TMX_EPS_FD;
    fx_forward_grad(P, inst, x, k, g1);
    for (int r = 0; r < k; ++r)
      Hn[r * k + i] = (g1[r] - g0[r]) / TMX_EPS_FD;
    x[i] = xi;
  }
  for (int i = 0; i < k; ++i)
    for (int j = i; j < k; ++j)
    {
      const double sy = (Hn[i * k + j] + Hn[j * k + i]) / 2;
      H[i * k + j] = sy;  // (H as scratch for the symmetric matrix; rebuilt below)
      H[j * k + i] = sy;
    }
  for (int i = 0; i < k * k; ++i)
    Hn[i] = H[i];
  sym_eig_jacobi(Hn, V, k);
  for (int i = 0; i < k * k; ++i)
    H[i] = 0.0;
  for (int e = 0; e < k; ++e)
    if (Hn[e * k + e] > 0)
      for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j)
          H[i * k + j] += Hn[e * k + e] * V[i * k + e] * V[j * k + e];
  double gx = 0.0, xhx = 0.0;
  for (int i = 0; i < k; ++i)
  {
    double hx = 0.0;
    for (int j = 0; j < k; ++j)
      hx += H[i * k + j] * x[j];
    g1[i] = hx;
    gx += g0[i] * x[i];
    xhx += x[i] * hx;
  }
  *cst = y - gx + .5 * xhx;
  for (int i = 0; i < k; ++i)
    g[i] = g0[i] - g1[i];
}
// QuadExpr::value of that model at the waypoint values xq (the (i, i) terms carry H_ii / 2, the (i < j) terms H_ij)
TMX_DEVFN double fx_model_value(const double* H, const double* g, double cst, const double* xq, int k)
{
  double v = cst;
  for (int i = 0; i < k; ++i)
    v += g[i] * xq[i];
  for (int i = 0; i < k; ++i)
  {
    v += (H[i * k + i] / 2) * xq[i] * xq[i];
    for (int j = i + 1; j < k; ++j)
      v += H[i * k + j] * xq[i] * xq[j];
  }
  return v;
}

// number of entries per joint of a squared joint cost minus (last_step - first_step): position 1, velocity 0, acc -1, jerk -2
// (run-time in every instantiation: the banded structured path evaluates acceleration / jerk COSTS in the fused kernels too)
TMX_DEVFN bool vel_is_ifopt_kind(int pk) { return pk == 4 || pk == 5; }  // (trajopt_ifopt accel / jerk sets: n rows per joint)
template <bool ST>
TMX_DEVFN int vel_len_adj(int pk)
{
  return vel_is_ifopt_kind(pk) ? 1 : ((pk >= 2) ? 1 - pk : pk);
}
// ---- finite-difference rows of order 2 / 3 (JointAcc / JointJerk, trajectory_costs.cpp:502-1016) ------------------------
// order of a SLOT_JOINTVEL / SLOT_JOINTVEL_INEQ row: 1 (velocity: x[t], x[t+1]) unless slot_sub3 says 2 or 3
TMX_DEVFN int diff_row_order(const DevProblem* P, int r) { return P->slot_sub3[r] >= 2 ? P->slot_sub3[r] : 1; }
// stencil entry k of the order's difference expression as the reference writes it (:277-279, :522-525, :775-779)
TMX_DEVFN double diff_stencil(int ord, int k)
{
  if (ord == 1)
    return k == 0 ? -1.0 : 1.0;
  if (ord == 2)
    return k == 1 ? -2.0 : 1.0;
  return (k == 0) ? -1.0 : (k == 1) ? 3.0 : (k == 2) ? -3.0 : 1.0;
}
// coefficient of row r on x[t + k][j]: exprMult(diff, coeff) (EQ), (upper_tol - diff) * -coeff, (lower_tol - diff) * coeff
TMX_DEVFN double diff_row_coef(const DevProblem* P, int r, int k)
{
  const double s = diff_stencil(diff_row_order(P, r), k), c = P->slot_scale[r];
  if (P->slot_kind[r] == SLOT_JOINTVEL)
    return (1.0 * s) * c;
  return (P->slot_sub2[r] == 0) ? (0.0 - (1.0 * s)) * -c : (0.0 - (1.0 * s)) * c;
}
// entry (i, j) of diffAxis0 applied `ord` times (trajectory_costs.cpp:17-20): differences of differences, in that order
TMX_DEVFN double diff_value(const double* xv, int D, int i, int j, int ord)
{
  const double d0 = xv[(i + 1) * D + j] - xv[i * D + j];
  if (ord == 1)
    return d0;
  const double d1 = xv[(i + 2) * D + j] - xv[(i + 1) * D + j];
  if (ord == 2)
    return d1 - d0;
  const double d2 = xv[(i + 3) * D + j] - xv[(i + 2) * D + j];
  return (d2 - d1) - (d1 - d0);
}
// ---- trajopt_ifopt JointAccelConstraint / JointJerkConstraint as squared cost sets of the trajopt_sqp flavour (vel_kind 4 / 5, round 5;
// joint_acceleration_constraint.cpp:90-175, joint_jerk_constraint.cpp:90-180).  A set over n = last - first + 1 waypoints has n rows per
// joint: row i takes the FORWARD stencil on waypoints i .. i + ord while i < n - ord and the BACKWARD one on i - ord .. i for the last
// `ord` rows (which repeat the stencils of rows n - 2 ord .. n - ord - 1, with another order of the additions).
TMX_DEVFN int ifo_ord(int pk) { return pk - 2; }
TMX_DEVFN int ifo_start(int n, int ord, int i) { return (i < n - ord) ? i : i - ord; }  // first waypoint of row i's stencil (set-local)
// getValues() of row (i, j): xs = the set's first waypoint
TMX_DEVFN double ifo_value(const double* xs, int D, int n, int ord, int i, int j)
{
  if (ord == 2)
  {
    if (i < n - 2)
      return (xs[(i + 2) * D + j] - 2.0 * xs[(i + 1) * D + j]) + xs[i * D + j];       // q2 - 2.0 * q1 + q0
    return (xs[(i - 2) * D + j] - 2.0 * xs[(i - 1) * D + j]) + xs[i * D + j];         // (q2 = q_{i-2}, q1 = q_{i-1}, q0 = q_i)
  }
  if (i < n - 3)
    return ((-xs[i * D + j] + 3.0 * xs[(i + 1) * D + j]) - 3.0 * xs[(i + 2) * D + j]) + xs[(i + 3) * D + j];  // -q0 + 3.0 * q1 - 3.0 * q2 + q3
  return ((xs[i * D + j] - 3.0 * xs[(i - 1) * D + j]) + 3.0 * xs[(i - 2) * D + j]) - xs[(i - 3) * D + j];    // q0 - 3.0 * q1 + 3.0 * q2 - q3
}
// row (i, j) of the squared set at the convexification point x0 (AffExprs::create / square, expressions.cpp:28-112, as the JointVel set
// below): a = target - (value - J x0), the row scale sr = 2 (a w); entries of the FLIPPED Jacobian in ascending columns: -stencil
TMX_DEVFN double ifo_row_a(const double* xs0, int D, int n, int ord, int i, int j, double targ)
{
  const int a0 = ifo_start(n, ord, i);
  double jx = diff_stencil(ord, 0) * xs0[a0 * D + j];
  for (int k = 1; k <= ord; ++k)
    jx += diff_stencil(ord, k) * xs0[(a0 + k) * D + j];
  double cst = ifo_value(xs0, D, n, ord, i, j);
  cst += -1.0 * jx;
  return targ - cst;
}


// ---- TIME-PARAMETERISED TERMS (DevProblem::use_time) ---------------------------------------------------------------------------
// JointVelErrCalculator / JointVelJacCalculator (trajopt/src/kinematic_terms.cpp:427-470) for segment (t, t+1) of joint j:
//   vel = (x[t+1][j] - x[t][j]) * tau[t+1];  upper error -(upper_tol - (vel - target)),  lower error lower_tol - (vel - target);
//   Jacobian of the upper row: -tau on x[t][j], +tau on x[t+1][j], x[t+1][j] - x[t][j] on tau[t+1]; the lower row is its negative.
// affFromValGrad (trajopt_sco/src/modeling_utils.cpp:31-39): constant = y - sum_k J_k x_k over the term's variables in their order
// (the joint's column, then the time column: x[t][j], x[t+1][j], tau[t+1]), coefficients with |J| <= 1e-7 dropped (cleanupAff).
struct TvSeg
{
  double ja, jb, jc;    // cleaned Jacobian of the UPPER row
  double y_up, y_lo;    // error values
  double k_up, k_lo;    // constants of the two affine rows
};
TMX_DEVFN void tv_segment(const double* xv, int D, int t, int j, double target, double up, double lo, TvSeg& s)
{
  const double x0 = xv[t * D + j], x1 = xv[(t + 1) * D + j], tau = xv[(t + 1) * D + D - 1];
  const double vel = (x1 - x0) * tau;
  s.y_up = -(up - (vel - target));
  s.y_lo = lo - (vel - target);
  const double ja = -1.0 * tau, jb = 1.0 * tau, jc = x1 - x0;
  double du = 0.0, dl = 0.0;
  du += ja * x0;
  du += jb * x1;
  du += jc * tau;
  dl += (-ja) * x0;
  dl += (-jb) * x1;
  dl += (-jc) * tau;
  s.k_up = s.y_up - du;
  s.k_lo = s.y_lo - dl;
  s.ja = (fabs(ja) > TMX_CLEANUP_TOL) ? ja : 0.0;
  s.jb = (fabs(jb) > TMX_CLEANUP_TOL) ? jb : 0.0;
  s.jc = (fabs(jc) > TMX_CLEANUP_TOL) ? jc : 0.0;
}
// TimeCostCalculator (kinematic_terms.cpp:572-577): var_vals.cwiseInverse().sum() over tau[1 .. T-1] in the order of Eigen's
// vectorised linear reduction with two-lane packets (oracle/trajprob.hpp timeInverseSum: the same statement)
TMX_DEVFN double time_inverse_sum(const double* xv, int D, int T)
{
  const int n = T - 1;
  auto inv = [&](int k) { return 1.0 / xv[(k + 1) * D + D - 1]; };
  if (n <= 0)
    return 0.0;
  const int aligned2 = (n / 4) * 4, aligned = (n / 2) * 2;
  double res;
  if (aligned)
  {
    double p00 = inv(0), p01 = inv(1);
    if (aligned > 2)
    {
      double p10 = inv(2), p11 = inv(3);
      for (int k = 4; k < aligned2; k += 4)
      {
        p00 += inv(k);
        p01 += inv(k + 1);
        p10 += inv(k + 2);
        p11 += inv(k + 3);
      }
      p00 += p10;
      p01 += p11;
      if (aligned > aligned2)
      {
        p00 += inv(aligned2);
        p01 += inv(aligned2 + 1);
      }
    }
    res = p00 + p01;
    for (int k = aligned; k < n; ++k)
      res += inv(k);
  }
  else
  {
    res = inv(0);
    for (int k = 1; k < n; ++k)
      res += inv(k);
  }
  return res;
}
// Convexification of the time-parameterised terms at xv (piecewise kernels; one thread per row / record):
//   SLOT_JOINTVEL_TIME rows: coefficients on both waypoints + right-hand side (exprScale(aff, coeff), modeling_utils.cpp:175-204, :258-268)
//   tv_aff: the linearised rows of the SQUARED velocity costs;  tt_aff: gradient and constant of every TotalTime term
//   SLOT_TOTAL_TIME rows: right-hand side (their entries live in tt_aff)
TMX_DEVFN void convexify_time_terms(const DevProblem* P, const double* xv, int* active, double* coef, double* coef2, double* rhs, double* tv_aff,
                                    double* tt_aff, int tid, int NT)
{
  const int D = P->D, T = P->T;
  for (int r = tid; r < P->R; r += NT)
  {
    const int kind = P->slot_kind[r];
    if (kind == SLOT_JOINTVEL_TIME)
    {
      const int t = P->slot_t[r], j = P->slot_sub[r];
      const double w = P->slot_scale[r];
      TvSeg sg;
      // slot_aux1 = target; slot_aux2 = the row's own tolerance (upper or lower): the other one does not enter this row
      tv_segment(xv, D, t, j, P->slot_aux1[r], P->slot_aux2[r], P->slot_aux2[r], sg);
      double* a0 = coef + (size_t)r * D;
      double* a1 = coef2 + (size_t)P->slot_c2[r] * D;
      for (int k = 0; k < D; ++k)
        a0[k] = a1[k] = 0.0;
      const bool upper = P->slot_sub2[r] == 0;
      // (the lower row's Jacobian is the negated upper one: jac.bottomRows = -jac.topRows, kinematic_terms.cpp:466)
      a0[j] = (upper ? sg.ja : -sg.ja) * w;
      a1[j] = (upper ? sg.jb : -sg.jb) * w;
      a1[D - 1] = (upper ? sg.jc : -sg.jc) * w;
      rhs[r] = -((upper ? sg.k_up : sg.k_lo) * w);
      active[r] = 1;
    }
  }
  for (int item = tid; item < P->n_tv * (T - 1); item += NT)
  {
    const int k = item / (T - 1), t = item % (T - 1);
    double* rec = tv_aff + ((size_t)k * T + t) * TMX_TV_REC;
    if (t < P->tv_first[k] || t >= P->tv_last[k])
    {
      for (int q = 0; q < TMX_TV_REC; ++q)
        rec[q] = 0.0;
      continue;
    }
    TvSeg sg;
    tv_segment(xv, D, t, P->tv_joint[k], P->tv_target[k], P->tv_up[k], P->tv_lo[k], sg);
    rec[0] = sg.ja;
    rec[1] = sg.jb;
    rec[2] = sg.jc;
    rec[3] = sg.k_up;
    rec[4] = sg.k_lo;
  }
  for (int k = tid; k < P->n_tt; k += NT)
  {
    // TimeCostJacCalculator (kinematic_terms.cpp:579-584): -1 / tau^2; affFromValGrad over tau[1 .. T-1]
    double* g = tt_aff + (size_t)k * (T + 1);
    double dot = 0.0;
    g[0] = 0.0;
    for (int t = 1; t < T; ++t)
    {
      const double tau = xv[t * D + D - 1];
      const double jv = -1 * (1.0 / (tau * tau));
      dot += jv * tau;
      g[t] = (fabs(jv) > TMX_CLEANUP_TOL) ? jv : 0.0;
    }
    const double y = time_inverse_sum(xv, D, T) - P->tt_limit[k];
    g[T] = y - dot;
    const int r = P->tt_slot[k];
    if (r >= 0)
    {
      for (int q = 0; q < D; ++q)
        coef[(size_t)r * D + q] = 0.0;
      rhs[r] = -(g[T] * P->tt_coeff[k]);
      active[r] = 1;
    }
  }
}
// entry of the global row of TotalTime term k on tau[t] (exprScale(aff, coeff))
TMX_DEVFN double tt_row_entry(const DevProblem* P, const double* tt_aff, int k, int t) { return tt_aff[(size_t)k * (P->T + 1) + t] * P->tt_coeff[k]; }

// ---------------------------------------------------------------------------------------------------
// K6: exact costs / constraint violations at trajectory xv -> cost_out[n_costs], viol_out[n_cnts]
// (BasicTrustRegionSQP::evaluateCosts / evaluateConstraintViols, trajopt_sco/src/optimizers.cpp:176-192)
// `scratch` : LDS, >= tmx_eval_scratch_doubles(P) doubles
// ---------------------------------------------------------------------------------------------------
// LDS doubles needed by evaluate_terms / sqp_update_block: slot values, slot keys, velocity terms + sums
TMX_HOSTDEVFN size_t tmx_eval_scratch_doubles(int R, int NX, int n_vel, int n_costs, int n_cnts)
{
  return (size_t)R + (size_t)(R + 1) / 2 + (size_t)n_vel * NX + (size_t)n_vel + (size_t)n_costs + (size_t)n_cnts + 16;
}
// ST: the problem may hold rows / costs of difference order 2 and 3 (DevProblem::n_stencil, vel_kind 2 / 3)
template <bool ST = false, bool HULL = false>
TMX_DEVFN void evaluate_terms(const DevProblem* P, const double* xv, double* cost_out, double* viol_out, double* scratch,
                              int tid, int NT)
{
  const int D = P->D;
  // per-slot scalar: collision hinge value / |cart-pose row| / joint-pos value ; fixed rows contribute nothing
  for (int r = tid; r < P->R; r += NT)
  {
    const int kind = P->slot_kind[r];
    const int t = P->slot_t[r];
    double v = 0.0;
    if (kind == SLOT_COLLISION)
    {
      double n[3], pw[3];
      const double dist = contact_distance<HULL>(P, xv + t * D, P->slot_sub[r], P->slot_sub2[r], n, pw);
      const double margin = P->slot_aux1[r];
      if (!(dist > margin + P->slot_aux2[r]))
      {
        const double pv = margin - dist;
        v = ((pv > 0) ? pv : 0.0) * P->slot_objc[r];
      }
    }
#if TMX_LINK_ROWS
    else if (kind == SLOT_COLLISION_LVS)
    {
      // CollisionCost::value / CollisionConstraint::value over the segment's filtered contacts (collision_terms.cpp:1306-1327)
      LvsContact c;
      if (lvs_contact<HULL>(P, xv + t * D, xv + (t + 1) * D, r, c))
      {
        const double pv = P->slot_aux1[r] - c.distance;
        // flavour 1: calcBoundsViolations of the value margin - distance against (-inf, 0], UNWEIGHTED (the exact penalty
        // costs / constraint violations of TrajOptQPProblem are plain sums, trajopt_qp_problem.cpp:1003-1019, :1030-1046)
        v = (P->flavor == 1) ? ((pv > 0.0) ? fabs(pv - 0.0) : 0.0) : ((pv > 0) ? pv : 0.0) * P->slot_objc[r];
      }
    }
#endif
    else if (kind == SLOT_JOINTPOS_INEQ)
    {
      // JointPosIneqConstraint::value (trajectory_costs.cpp:227-242) then IneqConstraint::violations = pospart
      const double pos = xv[t * D + P->slot_sub[r]] - P->slot_aux1[r];
      const double e = (P->slot_sub2[r] == 0) ? (pos - P->slot_aux2[r]) * P->slot_scale[r] : ((pos * -1) + P->slot_aux2[r]) * P->slot_scale[r];
      v = (e > 0) ? e : 0.0;
    }
    else if (kind == SLOT_JOINTPOS)
    {
      // quirk Q5: JointPosEqConstraint::value returns coeff * diff^2 (trajectory_costs.cpp:165-174)
      const double d = xv[t * D + P->slot_sub[r]] - P->slot_aux1[r];
      // flavour 1: |x - target| (calcBoundsViolations with equality bounds, ifopt_utils.cpp:122-145)
      v = (P->flavor == 1) ? ((d != 0.0) ? fabs(d) : 0.0) : fabs((d * d) * P->slot_scale[r]);
    }
#if TMX_LINK_ROWS
    else if (kind == SLOT_JOINTVEL)
    {
      // JointVelEqConstraint::value: coeff * diff^2 as well (trajectory_costs.cpp:403-413)
      const int j = P->slot_sub[r];
      const double d = (ST ? diff_value(xv, D, t, j, diff_row_order(P, r)) : (xv[(t + 1) * D + j] - xv[t * D + j])) - P->slot_aux1[r];
      v = fabs((d * d) * P->slot_scale[r]);
    }
    else if (kind == SLOT_JOINTVEL_INEQ)
    {
      // JointVelIneqCost::value / JointVelIneqConstraint::value (trajectory_costs.cpp:349-361, 472-487)
      const int j = P->slot_sub[r];
      const double d0 = (ST ? diff_value(xv, D, t, j, diff_row_order(P, r)) : (xv[(t + 1) * D + j] - xv[t * D + j])) - P->slot_aux1[r];
      const double e = (P->slot_sub2[r] == 0) ? (d0 - P->slot_aux2[r]) * P->slot_scale[r] : ((d0 * -1) + P->slot_aux2[r]) * P->slot_scale[r];
      v = (e > 0) ? e : 0.0;
    }
#endif
#if TMX_LINK_ROWS
    else if (kind == SLOT_CARTVEL)
    {
      // ABS cost without coefficients: |err_i| (modeling_utils.cpp:143-167); INEQ constraint: pospart(err_i)
      Tf3 s0, s1;
      const double e = cart_vel_value(P, xv + t * D, xv + (t + 1) * D, P->slot_sub[r], P->slot_aux1[r], s0, s1);
      v = P->slot_iscnt[r] ? ((e > 0) ? e : 0.0) : fabs(e);
    }
#endif
    if constexpr (ST)
    {
      if (kind == SLOT_JOINTVEL_TIME)
      {
        // CostFromErrFunc::value, HINGE: pospart(err) * coeff; ConstraintFromErrFunc::value: err * coeff, violation |.| (EQ) / pospart
        // (INEQ)  (modeling_utils.cpp:143-165, :238-245)
        TvSeg sg;
        tv_segment(xv, D, t, P->slot_sub[r], P->slot_aux1[r], P->slot_aux2[r], P->slot_aux2[r], sg);
        const double e = (P->slot_sub2[r] == 0) ? sg.y_up : sg.y_lo;
        if (P->slot_iscnt[r])
        {
          const double ec = e * P->slot_scale[r];
          v = P->slot_eq[r] ? fabs(ec) : ((ec > 0) ? ec : 0.0);
        }
        else
          v = ((e > 0) ? e : 0.0) * P->slot_scale[r];
      }
      else if (kind == SLOT_TOTAL_TIME)
      {
        const int k = P->slot_sub[r];
        const double e = time_inverse_sum(xv, D, P->T) - P->tt_limit[k];
        if (P->slot_iscnt[r])
        {
          const double ec = e * P->tt_coeff[k];
          v = P->slot_eq[r] ? fabs(ec) : ((ec > 0) ? ec : 0.0);
        }
        else
          v = ((e > 0) ? e : 0.0) * P->tt_coeff[k];
      }
    }
    // cart-pose (and function) slots are written by the instance loops below (another thread, no barrier in between): never store here
    if (kind != SLOT_CARTPOSE && kind != SLOT_FUNC)
      scratch[r] = v;
  }
  // cart-pose instances: |coeff_i * err_i| (constraint violation) or abs cost
  for (int c = tid; c < P->n_cp; c += NT)
  {
    Tf3 tgt, tinv, src;
    tf_from12(P->cp_target + 12 * c, tgt);
    tf_inv(tgt, tinv);
    fk_tool(P, xv + P->cp_t[c] * D, src);
    double err[6], ax[3], ang;
    transform_error(tinv, src, err, ax, ang);
    const int s0 = P->cp_slot0[c];
    for (int i = 0; i < P->cp_nrows[c]; ++i)
    {
      // constraint: violation = |err*coeff| (modeling_utils.cpp:238-245, modeling.cpp:150-167);
      // ABS cost: |err|*coeff (modeling_utils.cpp:143-167)
      // (select chain, not err[idx]: a run-time subscript would put err[] in scratch memory)
      const int ix = P->cp_idx[6 * c + i];
      const double e = (ix == 0) ? err[0] : (ix == 1) ? err[1] : (ix == 2) ? err[2] : (ix == 3) ? err[3] : (ix == 4) ? err[4] : err[5];
      const double cc = P->cp_coeff[6 * c + i];
      // flavour 1: calcBoundsViolations of the row value against BoundZero, unweighted (getExactConstraintViolations)
      scratch[s0 + i] = (P->flavor == 1) ? ((e != 0.0) ? fabs(e - 0.0) : 0.0) : (P->cp_iscnt[c] ? fabs(e * cc) : fabs(e) * cc);
    }
  }
  if constexpr (ST)
    for (int c = tid; c < P->n_fx; c += NT)
      if (fx_is_rows(P->fx_kind[c]))
      {
        // ConstraintFromErrFunc::value (modeling_utils.cpp:238-245): err * coeffs; violation |.| (EQ) / pospart (INEQ).
        // CostFromErrFunc::value, ABS / HINGE (:143-165): |err| or pospart(err), THEN times the coefficient
        double o[TMX_EXPR_MAX_OUT];
        fx_eval(P, c, xv + P->fx_t[c] * D, o);
        int r = P->fx_slot0[c];
        for (int i = 0; i < P->fx_nout[c] && r < P->R; ++i)
          if (P->slot_kind[r] == SLOT_FUNC && P->slot_sub2[r] == c && P->slot_sub[r] == i)
          {
            if (P->slot_iscnt[r])
            {
              const double e = o[i] * P->slot_scale[r];
              scratch[r] = P->slot_eq[r] ? fabs(e) : ((e > 0) ? e : 0.0);
            }
            else
              scratch[r] = (P->slot_eq[r] ? fabs(o[i]) : ((o[i] > 0) ? o[i] : 0.0)) * P->slot_scale[r];
            ++r;
          }
      }
  // owner key of every slot (cost owners first, then constraint owners; -1 = contributes nothing)
  int* keys = reinterpret_cast<int*>(scratch + P->R);
  for (int r = tid; r < P->R; r += NT)
    keys[r] = (P->slot_kind[r] == SLOT_FIXED) ? -1 : (P->slot_iscnt[r] ? P->n_costs + P->slot_owner[r] : P->slot_owner[r]);
  // JointVelEqCost::value — (diff^2 * diag(coeffs)).sum(): the terms in parallel, summed below in column-major order
  double* vterm = scratch + P->R + (P->R + 1) / 2;  // n_vel x (D * (T-1))
  double* vsum = vterm + (size_t)P->n_vel * P->NX;
  for (int v = 0; v < P->n_vel; ++v)
  {
    // 0: difference of consecutive steps (JointVelEqCost), 1: position (JointPosEqCost); ST: 2 / 3 = second / third difference
    // (JointAccEqCost / JointJerkEqCost, trajectory_costs.cpp:532-545, :786-801)
    const int pk = P->vel_kind[v];
    const int first = P->vel_first[v], len = P->vel_last[v] - first + vel_len_adj<ST>(pk);
    for (int e = tid; e < D * len; e += NT)
    {
      // summation order: the reference's column-major Eigen array (joint-major) for trajopt_sco; row order of the constraint
      // set (segment-major) for the trajopt_sqp flavour (getExactCosts, trajopt_qp_problem.cpp:986-1001)
      const int j = (P->flavor == 1) ? e % D : e / len, i = first + ((P->flavor == 1) ? e / D : e % len);
      const double dv = vel_is_ifopt_kind(pk) ? ifo_value(xv + first * D, D, len, ifo_ord(pk), i - first, j)
                                              : ((pk >= 2) ? diff_value(xv, D, i, j, pk) : (pk ? xv[i * D + j] : (xv[(i + 1) * D + j] - xv[i * D + j])));
      const double d = dv - P->vel_targets[v * TMX_MAX_DOF + j];
      vterm[(size_t)v * P->NX + e] = (d * d) * P->vel_coeffs[v * TMX_MAX_DOF + j];
    }
  }
  TMX_SYNC();
  for (int v = tid; v < P->n_vel; v += NT)
  {
    const int cnt = D * (P->vel_last[v] - P->vel_first[v] + vel_len_adj<ST>(P->vel_kind[v]));
    double sacc = 0;
    for (int e = 0; e < cnt; ++e)
      sacc += vterm[(size_t)v * P->NX + e];
    vsum[v] = sacc;
  }
  TMX_SYNC();
  // one thread per owner accumulates its slots in slot order (sequential per owner => reference summation order)
  for (int k = tid; k < P->n_costs + P->n_cnts; k += NT)
  {
    double acc = 0.0;
    for (int r = P->own_lo[k]; r <= P->own_hi[k]; ++r)
      if (keys[r] == k)
        acc += scratch[r];
    if (k < P->n_costs)
    {
      for (int v = 0; v < P->n_vel; ++v)
        if (P->vel_cost[v] == k)
          acc += vsum[v];
      if constexpr (ST)
        for (int c = 0; c < P->n_fx; ++c)
          if (fx_is_quad(P->fx_kind[c]) && P->fx_owner[c] == k)
          {
            if (P->fx_kind[c] != 3)
              acc += fx_eval1(P, c, xv + P->fx_t[c] * D);  // CostFromFunc::value (modeling_utils.cpp:46-50)
            else
            {
              // CostFromErrFunc::value, SQUARED: (err^2 * coeffs).sum()  (:143-165)
              double o[TMX_EXPR_MAX_OUT];
              fx_eval(P, c, xv + P->fx_t[c] * D, o);
              const double* wts = P->fx_consts + P->fx_c0[c] - TMX_EXPR_MAX_OUT;
              double sacc = 0.0;
              for (int i = 0; i < P->fx_nout[c]; ++i)
                sacc += (o[i] * o[i]) * wts[i];
              acc += sacc;
            }
          }
      if constexpr (ST)
      {
        // SQUARED time-parameterised costs, CostFromErrFunc::value (:143-165): err^2, times the coefficient, summed in row order
        for (int c = 0; c < P->n_tv; ++c)
          if (P->tv_owner[c] == k)
          {
            double sacc = 0.0;
            for (int half = 0; half < 2; ++half)  // the upper rows, then the lower rows
              for (int t = P->tv_first[c]; t < P->tv_last[c]; ++t)
              {
                TvSeg sg;
                tv_segment(xv, D, t, P->tv_joint[c], P->tv_target[c], P->tv_up[c], P->tv_lo[c], sg);
                const double e = half ? sg.y_lo : sg.y_up;
                sacc += (e * e) * P->tv_coeff[c];
              }
            acc += sacc;
          }
        for (int c = 0; c < P->n_tt; ++c)
          if (P->tt_form[c] == 0 && P->tt_owner[c] == k)
          {
            const double e = time_inverse_sum(xv, D, P->T) - P->tt_limit[c];
            acc += (e * e) * P->tt_coeff[c];
          }
      }
      cost_out[k] = acc;
    }
    else
      viol_out[k - P->n_costs] = acc;
  }
  TMX_SYNC();
}

// Function terms at the convexification point (piecewise kernels of qp_dense problems only): ConstraintFromErrFunc::convex rows
// (modeling_utils.cpp:247-269: forward-difference Jacobian, affFromValGrad, optional row scale) and the quadratic models of the
// CostFromFunc instances.  One thread per instance: the evaluations of an instance are sequential by definition
// (x is perturbed in place), the instances are independent.
TMX_DEVFN void convexify_func_terms(const DevProblem* P, const double* xv, int* active, double* coef, double* rhs, double* fxH, double* fxg,
                                    double* fxc, double* fxW, int tid, int NT)
{
  const int D = P->D;
  // the function sees the JOINT values of its waypoint: prob.GetVarRow(s, 0, n_dof) (problem_description.cpp:611-660) - in a
  // time-parameterised problem (D = DK + 1) the time column is not one of its variables: zero Jacobian / Hessian entries there (round 5)
  const int KV = P->DK;
  for (int c = tid; c < P->n_fx; c += NT)
  {
    const double* q = xv + P->fx_t[c] * D;
    if (fx_is_quad(P->fx_kind[c]))
    {
      const int ci = P->fx_ci[c];
      double* Hc = fxH + (size_t)ci * D * D;
      double* gc = fxg + (size_t)ci * D;
      fx_convexify_cost(P, c, q, KV, Hc, gc, fxc + ci, fxW + (size_t)ci * 2 * D * D);
      if (KV < D)
      {
        // the KV x KV model in the D x D storage every consumer indexes: rows moved apart from the last entry down, padding zeroed
        for (int i = KV - 1; i >= 0; --i)
          for (int j = KV - 1; j >= 0; --j)
            Hc[i * D + j] = Hc[i * KV + j];
        for (int i = 0; i < D; ++i)
          for (int j = 0; j < D; ++j)
            if (i >= KV || j >= KV)
              Hc[i * D + j] = 0.0;
        for (int j = KV; j < D; ++j)
          gc[j] = 0.0;
      }
      continue;
    }
    double x[TMX_MAX_DOF], y[TMX_EXPR_MAX_OUT], yp[TMX_EXPR_MAX_OUT];
    double J[TMX_EXPR_MAX_OUT][TMX_MAX_DOF];
    for (int i = 0; i < D; ++i)
      x[i] = q[i];
    for (int o = 0; o < TMX_EXPR_MAX_OUT; ++o)
      for (int i = KV; i < D; ++i)
        J[o][i] = 0.0;
    const int no = P->fx_nout[c];
    fx_eval(P, c, x, y);  // calcForwardNumJac evaluates f(x) first (num_diff.cpp:57), convex() once more for y (:252): same value
    if (P->fx_nops[c] < 0)
      fx_builtin_jac(P, c, x, J);  // the calculators' own dfdx (modeling_utils.cpp:171, :250)
    else
      for (int i = 0; i < KV; ++i)
      {
        const double xi = x[i];
        x[i] = xi + TMX_EPS_FD;
        fx_eval(P, c, x, yp);
        for (int o = 0; o < no; ++o)
          J[o][i] = (yp[o] - y[o]) / TMX_EPS_FD;
        x[i] = xi;
      }
    int r = P->fx_slot0[c];
    for (int o = 0; o < no && r < P->R; ++o)
    {
      if (!(P->slot_kind[r] == SLOT_FUNC && P->slot_sub2[r] == c && P->slot_sub[r] == o))
        continue;  // row dropped at upload (zero coefficient, :258-259)
      double dot = 0.0;
      for (int k = 0; k < KV; ++k)
        dot += J[o][k] * x[k];
      const double cc = P->slot_scale[r];
      const double constant = (y[o] - dot) * cc;
      for (int k = 0; k < D; ++k)
        coef[r * D + k] = (fabs(J[o][k]) > TMX_CLEANUP_TOL) ? J[o][k] * cc : 0.0;
      rhs[r] = -constant;
      active[r] = 1;
      ++r;
    }
  }
  TMX_SYNC();
}

// ---------------------------------------------------------------------------------------------------
// K1 + K3: convexify at xv -> rows (active, coef[R][D], rhs[R]) for the dynamic slots.
// FIXED and JOINTPOS rows are constant (written once by init_static_rows).
// ---------------------------------------------------------------------------------------------------
TMX_DEVFN void init_static_rows(const DevProblem* P, const double* x0, int* active, double* coef, double* coef2, double* rhs, int tid, int NT)
{
  (void)coef2;
  const int D = P->D;
  for (int r = tid; r < P->R; r += NT)
  {
    const int kind = P->slot_kind[r];
#if TMX_LINK_ROWS
    if (kind == SLOT_JOINTVEL || kind == SLOT_JOINTVEL_INEQ)
    {
      // home coefficient on x[t][j], second-block coefficient on x[t+1][j]
      double* c2r = coef2 + (size_t)P->slot_c2[r] * D;
      for (int k = 0; k < D; ++k)
      {
        coef[r * D + k] = 0.0;
        c2r[k] = 0.0;
      }
      const int j = P->slot_sub[r];
      const double c = P->slot_scale[r], targ = P->slot_aux1[r], tol = P->slot_aux2[r];
      if (kind == SLOT_JOINTVEL)
      {
        // exprMult(vel, coeff), vel = -1*x[t] + 1*x[t+1] - target   (trajectory_costs.cpp:392-400)
        coef[r * D + j] = (1.0 * -1) * c;
        c2r[j] = (0.0 + (1.0 * 1)) * c;
        rhs[r] = -((0.0 - targ) * c);
      }
      else if (P->slot_sub2[r] == 0)
      {
        // expr = upper_tol - vel, scaled by -coeff   (:334-338, :458-462)
        coef[r * D + j] = (0.0 - (1.0 * -1)) * -c;
        c2r[j] = (0.0 - (1.0 * 1)) * -c;
        rhs[r] = -((tol - (0.0 - targ)) * -c);
      }
      else
      {
        // expr_neg = lower_tol - vel, scaled by coeff   (:340-344, :464-468)
        coef[r * D + j] = (0.0 - (1.0 * -1)) * c;
        c2r[j] = (0.0 - (1.0 * 1)) * c;
        rhs[r] = -((tol - (0.0 - targ)) * c);
      }
      if (P->slot_sub3[r] >= 2)
      {
        // acceleration / jerk rows: the same three forms over the longer stencil; the coefficients beyond waypoint t + 1 are
        // not stored (diff_row_coef)
        coef[r * D + j] = diff_row_coef(P, r, 0);
        c2r[j] = diff_row_coef(P, r, 1);
      }
      active[r] = 1;
      continue;
    }
#endif
    if (kind == SLOT_FIXED || kind == SLOT_JOINTPOS || kind == SLOT_JOINTPOS_INEQ)
    {
      for (int k = 0; k < D; ++k)
        coef[r * D + k] = 0.0;
      const int j = P->slot_sub[r];
      if (kind == SLOT_JOINTPOS_INEQ)
      {
        // rows "aff <= 0" with aff built exactly as the reference does (trajectory_costs.cpp:206-223):
        //   upper: ((1*x - target) - upper_tol) * coeff          lower: (lower_tol - (1*x - target)) * coeff
        const double c = P->slot_scale[r], targ = P->slot_aux1[r], tol = P->slot_aux2[r];
        if (P->slot_sub2[r] == 0)
        {
          coef[r * D + j] = 1.0 * c;
          rhs[r] = -(((0.0 - targ) - tol) * c);
        }
        else
        {
          coef[r * D + j] = (0.0 + (1.0 * -1)) * c;   // exprDec(expr_neg, pos): pos scaled by -1 and added
          rhs[r] = -((tol + ((0.0 - targ) * -1)) * c);
        }
      }
      else if (kind == SLOT_FIXED)
      {
        // exprSub(AffExpr(var), init): x - init == 0
        coef[r * D + j] = 1.0;
        rhs[r] = -(0.0 - x0[P->slot_t[r] * D + j]);
      }
      else if (P->flavor == 1)
      {
        // JointPosConstraint row of TrajOptQPProblem::convexify (trajopt_qp_problem.cpp:752-792): Jacobian entry 1, constant
        // value - J x0 = x - 1 x = 0, bounds target - constant
        coef[r * D + j] = 1.0;
        rhs[r] = P->slot_aux1[r] - 0.0;
      }
      else
      {
        // exprMult(pos, coeff) with pos = 1*x - target   (trajectory_costs.cpp:151-161)
        const double c = P->slot_scale[r];
        coef[r * D + j] = 1.0 * c;
        rhs[r] = -((0.0 - P->slot_aux1[r]) * c);
      }
      active[r] = 1;
    }
  }
}

// LDS doubles needed by convexify_terms: per cart-pose instance (D+1) pose records of 7, the error vector (6) and the
// finite-difference Jacobian (6 x D)
TMX_HOSTDEVFN size_t tmx_cvx_scratch_doubles(int n_cp, int D) { return (size_t)n_cp * ((size_t)(D + 1) * 7 + 6 + 6 * (size_t)D) + 8; }
// HULL: the instantiation carries the convex-hull link contacts (k_convexify of ST problems only)
template <bool HULL = false>
TMX_DEVFN void convexify_terms(const DevProblem* P, const double* xv, int* active, double* coef, double* coef2, double* rhs, double* scratch,
                               int tid, int NT, double* rowc = nullptr, double* qdyn = nullptr)
{
  (void)coef2;
  (void)rowc;
#if TMX_LINK_ROWS
  if (P->flavor == 1)
  {
    // trajopt_sqp flavour: linear objective of the squared cost sets at the convexification point
    // (TrajOptQPProblem::convexify, trajopt_qp_problem.cpp:861-927 with AffExprs::create / square, expressions.cpp:28-112):
    //   per row r = (segment i, joint j):  constants = (x1 - x0) - ((-1) x0 + (1) x1) ;  a = target - constants ;
    //   flipped Jacobian entries (+1 at x0, -1 at x1) ;  sr = 2 (a w) ;  objective_linear[col] += entry * sr   (row order)
    const int D = P->D;
    for (int v = tid; v < P->NX; v += NT)
    {
      const int t = v / D, j = v % D;
      double acc = 0.0;
      for (int k = 0; k < P->n_vel; ++k)
      {
        double part = 0.0;  // objective_linear_coeffs of this set, accumulated over its rows in row order
        const double w = P->vel_coeffs[k * TMX_MAX_DOF + j], targ = P->vel_targets[k * TMX_MAX_DOF + j];
        if (vel_is_ifopt_kind(P->vel_kind[k]))
        {
          // JointAccelConstraint / JointJerkConstraint set: every row i whose stencil covers waypoint t, in row order
          const int first = P->vel_first[k], n = P->vel_last[k] - first + 1, ord = ifo_ord(P->vel_kind[k]);
          for (int i = 0; i < n; ++i)
          {
            const int a0 = ifo_start(n, ord, i);
            if (t - first < a0 || t - first > a0 + ord)
              continue;
            const double a = ifo_row_a(xv + first * D, D, n, ord, i, j, targ);
            const double sr = 2.0 * (a * w);
            part += (diff_stencil(ord, t - first - a0) * -1) * sr;
          }
          acc += part;
          continue;
        }
        if (P->vel_kind[k] != 0)
          continue;
        for (int side = 1; side >= 0; --side)  // the row of segment t-1 (entry -1 at x1 = this var) comes before segment t's
        {
          const int i = side ? t - 1 : t;      // segment index
          if (i < P->vel_first[k] || i > P->vel_last[k] - 1)
            continue;
          const double x0 = xv[i * D + j], x1 = xv[(i + 1) * D + j];
          double cst = x1 - x0;
          cst += -1.0 * ((-1 * x0) + (1 * x1));
          const double a = targ - cst;
          const double sr = 2.0 * (a * w);
          part += (side ? (1.0 * -1) : (-1.0 * -1)) * sr;
        }
        acc += part;
      }
      qdyn[v] = acc;
    }
  }
#endif
  const int D = P->D;
  // ---- K1: cart-pose rows by forward finite differences over full FK.  One thread per (instance, perturbed joint | base
  //      pose): the D+1 forward-kinematics chains of an instance run side by side instead of one after the other
  double* rec = scratch;                                    // [n_cp][D+1][7] : translation, rotation axis, angle
  double* errv = rec + (size_t)P->n_cp * (D + 1) * 7;        // [n_cp][6]
  double* Jm = errv + (size_t)P->n_cp * 6;                   // [n_cp][6][D]
  for (int item = tid; item < P->n_cp * (D + 1); item += NT)
  {
    const int c = item / (D + 1), k = item % (D + 1);
    const double* q = xv + P->cp_t[c] * D;
    Tf3 tgt, tinv, src, pp;
    tf_from12(P->cp_target + 12 * c, tgt);
    tf_inv(tgt, tinv);
    fk_tool_at(P, [q, k](int j) { return (j == k) ? q[j] + TMX_EPS_FD : q[j]; }, src);  // k == D: the unperturbed pose
    tf_mul(tinv, src, pp);
    double ax[3], ang;
    if (k < D)
      rot_err_decomposed(pp.R, ax, ang);
    else
    {
      double err[6];
      transform_error(tinv, src, err, ax, ang);
      for (int r = 0; r < 6; ++r)
        errv[c * 6 + r] = err[r];
    }
    double* o = rec + (size_t)item * 7;
    for (int r = 0; r < 3; ++r)
    {
      o[r] = pp.t[r];
      o[3 + r] = ax[r];
    }
    o[6] = ang;
  }
  TMX_SYNC();
  for (int item = tid; item < P->n_cp * D; item += NT)
  {
    const int c = item / D, k = item % D;
    const double* o0 = rec + ((size_t)c * (D + 1) + D) * 7;  // base pose
    const double* o1 = rec + ((size_t)c * (D + 1) + k) * 7;
    const double a0 = o0[6], a1 = o1[6];
    // calcJacobianTransformErrorDiff(target, source, source_perturbed)
    double a1c = a1;
    if (a1 > M_PI_2 && a0 < -M_PI_2)
      a1c = a1 - 2.0 * M_PI;
    else if (a1 < -M_PI_2 && a0 > M_PI_2)
      a1c = a1 + 2.0 * M_PI;
    for (int i = 0; i < P->cp_nrows[c]; ++i)
    {
      const int r = P->cp_idx[6 * c + i];
      const double diff = (r < 3) ? o1[r] - o0[r] : o1[r] * a1c - o0[r] * a0;
      Jm[((size_t)c * 6 + i) * D + k] = diff / TMX_EPS_FD;
    }
  }
  TMX_SYNC();
  for (int item = tid; item < P->n_cp * 6; item += NT)
  {
    const int c = item / 6, i = item % 6;
    if (i >= P->cp_nrows[c])
      continue;
    // affFromValGrad: constant = y - J.x ; coeffs = J with |c| <= 1e-7 dropped ; then exprScale(aff, coeff)
    const double* q = xv + P->cp_t[c] * D;
    const double* Jr = Jm + ((size_t)c * 6 + i) * D;
    const double y = errv[c * 6 + P->cp_idx[6 * c + i]];
    double dot = 0.0;
    for (int k = 0; k < D; ++k)
      dot += Jr[k] * q[k];
    const double cc = P->cp_coeff[6 * c + i];
    const int r = P->cp_slot0[c] + i;
#if TMX_LINK_ROWS
    if (P->flavor == 1)
    {
      // trajopt_ifopt::CartPosConstraint row in TrajOptQPProblem::convexify (trajopt_qp_problem.cpp:752-792; round 5): the Jacobian
      // block as the set returns it (forward differences / eps, unscaled), entries with |v| < 1e-7 stored as 0, constant = value - J x0
      // with the UNPRUNED Jacobian, equality bounds 0 - constant; the coefficient weighs the slack pair only (aux_cost)
      const double constant = y - dot;
      for (int k = 0; k < D; ++k)
        coef[r * D + k] = (fabs(Jr[k]) < TMX_CLEANUP_TOL) ? 0.0 : Jr[k];
      rhs[r] = P->slot_aux1[r] - constant;
      rowc[r] = constant;
      active[r] = 1;
      continue;
    }
#endif
    const double constant = (y - dot) * cc;
    for (int k = 0; k < D; ++k)
    {
      const double jv = Jr[k];
      coef[r * D + k] = (fabs(jv) > TMX_CLEANUP_TOL) ? jv * cc : 0.0;
    }
    rhs[r] = -constant;
    active[r] = 1;
  }
#if TMX_LINK_ROWS
  // ---- CartVel rows (pair rows, analytic Jacobian): CartVelJacCalculator kinematic_terms.cpp:376-401, affFromValGrad
  for (int r = tid; r < P->R; r += NT)
  {
    if (P->slot_kind[r] != SLOT_CARTVEL)
      continue;
    const int t = P->slot_t[r], i = P->slot_sub[r], c = i < 3 ? i : i - 3;
    const double* q0 = xv + t * D;
    const double* q1 = xv + (t + 1) * D;
    double* a0 = coef + (size_t)r * D;
    double* a1 = coef2 + (size_t)P->slot_c2[r] * D;
    Tf3 s0, s1, L;
    const double y = cart_vel_value(P, q0, q1, i, P->slot_aux1[r], s0, s1);
    // component c of the Jacobian columns at both waypoints, staged in the row's own coefficient arrays
    fk_link_visit(P, [q0](int k) { return q0[k]; }, D - 1, L, [&](int k, const Tf3& F) {
      double col[3];
      jac_point_col(P, k, F, s0.t, col);
      a0[k] = (c == 0) ? col[0] : (c == 1) ? col[1] : col[2];
    });
    fk_link_visit(P, [q1](int k) { return q1[k]; }, D - 1, L, [&](int k, const Tf3& F) {
      double col[3];
      jac_point_col(P, k, F, s1.t, col);
      a1[k] = (c == 0) ? col[0] : (c == 1) ? col[1] : col[2];
    });
    // out.block(0, 0) = -jac0, out.block(0, n) = jac1, out.block(3, 0) = jac0, out.block(3, n) = -jac1
    double dot = 0.0;
    for (int k = 0; k < D; ++k)
    {
      a0[k] = (i < 3) ? -a0[k] : a0[k];
      dot += a0[k] * q0[k];
    }
    for (int k = 0; k < D; ++k)
    {
      a1[k] = (i < 3) ? a1[k] : -a1[k];
      dot += a1[k] * q1[k];
    }
    const double constant = y - dot;
    for (int k = 0; k < D; ++k)
    {
      a0[k] = (fabs(a0[k]) > TMX_CLEANUP_TOL) ? a0[k] : 0.0;
      a1[k] = (fabs(a1[k]) > TMX_CLEANUP_TOL) ? a1[k] : 0.0;
    }
    rhs[r] = -constant;
    active[r] = 1;
  }
#endif
#if TMX_LINK_ROWS
  // ---- K3': collision rows of the segment evaluators (pair rows: gradient on both waypoints)
  for (int r = tid; r < P->R; r += NT)
  {
    if (P->slot_kind[r] != SLOT_COLLISION_LVS)
      continue;
    const int t = P->slot_t[r], s = P->slot_sub[r], flags = P->slot_sub3[r];
    const bool fixed0 = flags & 1, fixed1 = flags & 2;
    const double* q0 = xv + t * D;
    const double* q1 = xv + (t + 1) * D;
    double* c2r = coef2 + (size_t)P->slot_c2[r] * D;
    LvsContact c;
    if (!lvs_contact<HULL>(P, q0, q1, r, c))
    {
      active[r] = 0;
      for (int k = 0; k < D; ++k)
      {
        coef[r * D + k] = 0.0;
        c2r[k] = 0.0;
      }
      rhs[r] = 0.0;
      continue;
    }
    // dist_expr = distance + [scale0 (grad0 . x0 - grad0 . q0)] + [scale1 (grad1 . x1 - grad1 . q1)], cleanupAff (1e-7)
    // the raw scaled gradients are staged in the row's own coefficient arrays and post-processed in place
    double *g0 = coef + (size_t)r * D, *g1 = c2r, k0 = 0.0, k1 = 0.0;
    for (int k = 0; k < D; ++k)
      g0[k] = g1[k] = 0.0;
    if (!fixed0)
      lvs_end_gradient(P, q0, s, c, false, g0, k0);
    if (!fixed1)
      lvs_end_gradient(P, q1, s, c, true, g1, k1);
    const double margin = P->slot_aux1[r];
    if (P->flavor == 1)
    {
      // SegmentCollisionConstraint row in TrajOptQPProblem::convexify: value = margin - distance, Jacobian = -(scaled
      // gradients) on both waypoints, constant = value - J x0, entries with |v| < 1e-7 stored as 0, upper bound 0 - constant
      double sdot = 0.0;
      for (int k = 0; k < D; ++k)
        if (!fixed0)
          sdot += (-g0[k]) * q0[k];
      for (int k = 0; k < D; ++k)
        if (!fixed1)
          sdot += (-g1[k]) * q1[k];
      double cc = margin - c.distance;
      cc += -1.0 * sdot;
      for (int k = 0; k < D; ++k)
      {
        const double a0 = -g0[k], a1 = -g1[k];
        coef[r * D + k] = (fabs(a0) < TMX_CLEANUP_TOL) ? 0.0 : a0;
        c2r[k] = (fabs(a1) < TMX_CLEANUP_TOL) ? 0.0 : a1;
      }
      rhs[r] = 0.0 - cc;
      rowc[r] = cc;
      active[r] = 1;
      continue;
    }
    double cst = c.distance;
    if (!fixed0)
      cst += (0.0 + k0);
    if (!fixed1)
      cst += (0.0 + k1);
    const double viol_const = margin - cst;
    const double cc = P->slot_iscnt[r] ? P->slot_scale[r] : 1.0;
    for (int k = 0; k < D; ++k)
    {
      const double a0 = (fabs(g0[k]) > TMX_CLEANUP_TOL) ? g0[k] : 0.0, a1 = (fabs(g1[k]) > TMX_CLEANUP_TOL) ? g1[k] : 0.0;
      // viol = margin - dist_expr: coefficients -a; CollisionConstraint scales the row by the collision coefficient
      coef[r * D + k] = P->slot_iscnt[r] ? (-a0) * cc : -a0;
      c2r[k] = P->slot_iscnt[r] ? (-a1) * cc : -a1;
    }
    rhs[r] = P->slot_iscnt[r] ? -(viol_const * cc) : -viol_const;
    active[r] = 1;
  }
#endif
  // ---- K3: collision rows
  for (int r = tid; r < P->R; r += NT)
  {
    if (P->slot_kind[r] != SLOT_COLLISION)
      continue;
    const double* q = xv + P->slot_t[r] * D;
    const int s = P->slot_sub[r], o = P->slot_sub2[r];
    const int link = P->ls_link[s];
    double n[3], pw[3];
    const double dist = contact_distance<HULL>(P, q, s, o, n, pw);
    const double margin = P->slot_aux1[r];
    if (dist > margin + P->slot_aux2[r])
    {
      active[r] = 0;
      for (int k = 0; k < D; ++k)
        coef[r * D + k] = 0.0;
      rhs[r] = 0.0;
      continue;
    }
    // gradient = -n' * J_trans(nearest point), object 0 = robot link (collision_terms.cpp:203-250).  Second pass over the
    // chain (the joint frames are recomputed instead of being kept in a per-thread array); the raw gradient is staged in
    // the row's coefficient array
    double* grad = coef + (size_t)r * D;
    double gq = 0.0;
    {
      Tf3 L;
      fk_link_visit(P, [q](int k) { return q[k]; }, link, L, [&](int k, const Tf3& F) {
        const double g = contact_grad_col(P, k, F, pw, n);
        grad[k] = g;
        gq += g * q[k];
      });
      for (int k = link + 1; k < D; ++k)
      {
        const double g = -1.0 * (n[0] * 0.0 + n[1] * 0.0 + n[2] * 0.0);
        grad[k] = g;
        gq += g * q[k];
      }
    }
    // dist_expr = grad.x + (-(grad.q) + dist) ; viol = margin - dist_expr ; row: viol - hinge <= 0
    const double c_dist = (0.0 + (-gq)) + dist;
    const double viol_const = margin - c_dist;
    if (P->slot_iscnt[r])
    {
      // CollisionConstraint: exprMult(viol, coeff)  (collision_terms.cpp:1387-1391)
      const double cc = P->slot_scale[r];
      for (int k = 0; k < D; ++k)
        coef[r * D + k] = (-grad[k]) * cc;
      rhs[r] = -(viol_const * cc);
    }
    else
    {
      for (int k = 0; k < D; ++k)
        coef[r * D + k] = -grad[k];
      rhs[r] = -viol_const;
    }
    active[r] = 1;
  }
  TMX_SYNC();
}
