/*
 * tmx_geom.h — closest points between a link sphere (or the segment its centre sweeps over a sub-segment of the
 * trajectory) and a world obstacle.  ONE statement of the arithmetic, included by the device kernels
 * (trajopt_amd/csrc/tmx_terms.h) and by the CPU oracle (oracle/trajprob.hpp), so that the contact data - distance,
 * normal, nearest point, cc_time - are bit-identical on both sides (no FMA contraction in either build).
 *
 * Obstacle primitives: SPHERE (centre, radius), CAPSULE = the sphere swept from `centre` to `centre + axis` and (round 3) the
 * rounded BOX (tmx_problem_desc::obstacle_boxes) and CONVEX TRIANGLE MESH (obstacle_mesh / mesh_triangles), both with signed
 * distance and penetration; link primitives: sphere and capsule.
 * (tmx_problem_desc::obstacle_axes; a zero axis is a sphere).  In the reference these contacts come from
 * tesseract / Bullet (trajopt/src/collision_terms.cpp:655-691 discrete, :1064-1173 cast); for sphere-vs-sphere and
 * sphere-vs-capsule the signed distance, the normal and the nearest points have the closed forms below.
 * The sphere branches are literally the round-1 / round-2 formulas (existing fixtures keep their bits).
 */
#ifndef TMX_GEOM_H_
#define TMX_GEOM_H_

#if defined(__HIPCC__)
#define TMX_GM_FN __host__ __device__ static inline
#else
#define TMX_GM_FN static inline
#endif

#include <math.h>
#define TMX_GM_EPS 1e-24 /* squared length under which a segment counts as a point */

TMX_GM_FN double tmx_gm_clamp01(double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }

/* point of the obstacle (centre oc, axis oa - may be a null pointer or zero: sphere) closest to the point c */
TMX_GM_FN void tmx_obstacle_closest_to_point(const double oc[3], const double* oa, const double c[3], double q[3])
{
  double aa = 0.0;
  if (oa)
    aa = oa[0] * oa[0] + oa[1] * oa[1] + oa[2] * oa[2];
  if (!(aa > TMX_GM_EPS))
  {
    q[0] = oc[0];
    q[1] = oc[1];
    q[2] = oc[2];
    return;
  }
  const double f = oa[0] * (c[0] - oc[0]) + oa[1] * (c[1] - oc[1]) + oa[2] * (c[2] - oc[2]);
  const double t = tmx_gm_clamp01(f / aa);
  q[0] = oc[0] + t * oa[0];
  q[1] = oc[1] + t * oa[1];
  q[2] = oc[2] + t * oa[2];
}

/* closest points between the swept centre  P(tau) = ca + tau e, tau in [0, 1]  and the obstacle; returns tau and the
 * obstacle point q.  Sphere obstacle: tau = clamp(e.(oc - ca) / e.e), 0 for a link that does not move.  Capsule: the
 * closest points of two segments (the standard clamped solution of the 2 x 2 normal equations). */
TMX_GM_FN double tmx_swept_closest_to_obstacle(const double ca[3], const double e[3], const double oc[3], const double* oa, double q[3])
{
  double aa = 0.0;
  if (oa)
    aa = oa[0] * oa[0] + oa[1] * oa[1] + oa[2] * oa[2];
  const double ee = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
  if (!(aa > TMX_GM_EPS))
  {
    const double eo = e[0] * (oc[0] - ca[0]) + e[1] * (oc[1] - ca[1]) + e[2] * (oc[2] - ca[2]);
    double tau = (ee > TMX_GM_EPS) ? eo / ee : 0.0; /* a link that does not move over the sub-segment: contact at its start */
    tau = tau < 0.0 ? 0.0 : (tau > 1.0 ? 1.0 : tau);
    q[0] = oc[0];
    q[1] = oc[1];
    q[2] = oc[2];
    return tau;
  }
  /* segments  ca + s e  and  oc + t oa */
  const double r[3] = { ca[0] - oc[0], ca[1] - oc[1], ca[2] - oc[2] };
  const double f = oa[0] * r[0] + oa[1] * r[1] + oa[2] * r[2];
  double s, t;
  if (!(ee > TMX_GM_EPS))
  {
    s = 0.0;
    t = tmx_gm_clamp01(f / aa);
  }
  else
  {
    const double c = e[0] * r[0] + e[1] * r[1] + e[2] * r[2];
    const double b = e[0] * oa[0] + e[1] * oa[1] + e[2] * oa[2];
    const double denom = ee * aa - b * b;
    s = (denom > TMX_GM_EPS * ee * aa) ? tmx_gm_clamp01((b * f - c * aa) / denom) : 0.0; /* parallel: start of the sweep */
    t = (b * s + f) / aa;
    if (t < 0.0)
    {
      t = 0.0;
      s = tmx_gm_clamp01(-c / ee);
    }
    else if (t > 1.0)
    {
      t = 1.0;
      s = tmx_gm_clamp01((b - c) / ee);
    }
  }
  q[0] = oc[0] + t * oa[0];
  q[1] = oc[1] + t * oa[1];
  q[2] = oc[2] + t * oa[2];
  return s;
}

/* ---- BOX obstacles: centre oc, ob = 12 doubles (half extents hx hy hz, then the rotation world_R_box row-major); the obstacle's
 * `radius` rounds the box (core + radius, like the sphere / capsule primitives).  A null pointer or zero extents: not a box. ---- */
TMX_GM_FN int tmx_is_box(const double* ob) { return ob != 0 && (ob[0] > 0.0 || ob[1] > 0.0 || ob[2] > 0.0); }

/* signed distance of the point c to the box (negative inside) and the surface point q it is measured to: outside the clamped
 * point, inside the nearest point of the nearest face (ties: the lowest axis) */
TMX_GM_FN double tmx_box_sdf(const double oc[3], const double* ob, const double c[3], double q[3])
{
  const double* R = ob + 3;
  const double r[3] = { c[0] - oc[0], c[1] - oc[1], c[2] - oc[2] };
  double l[3], k[3];
  int inside = 1;
  for (int i = 0; i < 3; ++i)
  {
    l[i] = R[0 + i] * r[0] + R[3 + i] * r[1] + R[6 + i] * r[2]; /* box frame: R' r */
    k[i] = l[i] < -ob[i] ? -ob[i] : (l[i] > ob[i] ? ob[i] : l[i]);
    if (k[i] != l[i])
      inside = 0;
  }
  double sd;
  if (inside)
  {
    int ax = 0;
    double best = ob[0] - (l[0] < 0.0 ? -l[0] : l[0]);
    for (int i = 1; i < 3; ++i)
    {
      const double g = ob[i] - (l[i] < 0.0 ? -l[i] : l[i]);
      if (g < best)
      {
        best = g;
        ax = i;
      }
    }
    k[ax] = l[ax] < 0.0 ? -ob[ax] : ob[ax];
    sd = -best;
  }
  else
  {
    const double dx = l[0] - k[0], dy = l[1] - k[1], dz = l[2] - k[2];
    sd = sqrt(dx * dx + dy * dy + dz * dz);
  }
  for (int i = 0; i < 3; ++i)
    q[i] = oc[i] + R[3 * i + 0] * k[0] + R[3 * i + 1] * k[1] + R[3 * i + 2] * k[2];
  return sd;
}

/* ---- CONVEX TRIANGLE-MESH obstacles: nt triangles of 9 doubles (three world-frame vertices, counter-clockwise seen from
 * outside), the boundary of a convex polytope (a convex hull).  Internally a mesh obstacle is a 12-double record like a box with the
 * tag ob[0] = -1, ob[1] = number of triangles, ob[2] = offset (in doubles) into the mesh array. ---- */
TMX_GM_FN int tmx_is_mesh(const double* ob) { return ob != 0 && ob[0] == -1.0; }

/* closest point of the triangle (a, b, c) to p (Voronoi-region walk: vertices, edges, face) */
TMX_GM_FN void tmx_tri_closest(const double* a, const double* b, const double* c, const double p[3], double q[3])
{
  const double ab[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] }, ac[3] = { c[0] - a[0], c[1] - a[1], c[2] - a[2] };
  const double ap[3] = { p[0] - a[0], p[1] - a[1], p[2] - a[2] };
  const double d1 = ab[0] * ap[0] + ab[1] * ap[1] + ab[2] * ap[2], d2 = ac[0] * ap[0] + ac[1] * ap[1] + ac[2] * ap[2];
  double u = 0.0, v = 0.0; /* q = a + u ab + v ac */
  if (d1 <= 0.0 && d2 <= 0.0)
  {
    u = 0.0;
    v = 0.0;
  }
  else
  {
    const double bp[3] = { p[0] - b[0], p[1] - b[1], p[2] - b[2] };
    const double d3 = ab[0] * bp[0] + ab[1] * bp[1] + ab[2] * bp[2], d4 = ac[0] * bp[0] + ac[1] * bp[1] + ac[2] * bp[2];
    const double cp[3] = { p[0] - c[0], p[1] - c[1], p[2] - c[2] };
    const double d5 = ab[0] * cp[0] + ab[1] * cp[1] + ab[2] * cp[2], d6 = ac[0] * cp[0] + ac[1] * cp[1] + ac[2] * cp[2];
    const double vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    if (d3 >= 0.0 && d4 <= d3)
    {
      u = 1.0; /* vertex b */
      v = 0.0;
    }
    else if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0)
    {
      u = d1 / (d1 - d3); /* edge ab */
      v = 0.0;
    }
    else if (d6 >= 0.0 && d5 <= d6)
    {
      u = 0.0; /* vertex c */
      v = 1.0;
    }
    else if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0)
    {
      u = 0.0; /* edge ac */
      v = d2 / (d2 - d6);
    }
    else if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0)
    {
      const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); /* edge bc */
      u = 1.0 - w;
      v = w;
    }
    else
    {
      const double denom = 1.0 / (va + vb + vc); /* inside the face */
      u = vb * denom;
      v = vc * denom;
    }
  }
  for (int i = 0; i < 3; ++i)
    q[i] = a[i] + u * ab[i] + v * ac[i];
}

/* signed distance of p to the convex mesh (negative inside) and the surface point q it is measured to: outside the closest point
 * over all triangles (first minimum), inside the projection on the nearest face plane */
TMX_GM_FN double tmx_mesh_sdf(const double* tri, int nt, const double p[3], double q[3])
{
  double best_plane = -1e300, best_d2 = 1e300;
  int kin = 0;
  double qb[3] = { p[0], p[1], p[2] };
  for (int k = 0; k < nt; ++k)
  {
    const double *a = tri + 9 * k, *b = a + 3, *c = a + 6;
    const double ab[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] }, ac[3] = { c[0] - a[0], c[1] - a[1], c[2] - a[2] };
    double n[3] = { ab[1] * ac[2] - ab[2] * ac[1], ab[2] * ac[0] - ab[0] * ac[2], ab[0] * ac[1] - ab[1] * ac[0] };
    const double nl = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (!(nl > 0.0))
      continue; /* degenerate triangle */
    const double pd = (n[0] * (p[0] - a[0]) + n[1] * (p[1] - a[1]) + n[2] * (p[2] - a[2])) / nl;
    if (pd > best_plane)
    {
      best_plane = pd;
      kin = k;
    }
    double qq[3];
    tmx_tri_closest(a, b, c, p, qq);
    const double dx = p[0] - qq[0], dy = p[1] - qq[1], dz = p[2] - qq[2];
    const double d2 = dx * dx + dy * dy + dz * dz;
    if (d2 < best_d2)
    {
      best_d2 = d2;
      qb[0] = qq[0];
      qb[1] = qq[1];
      qb[2] = qq[2];
    }
  }
  if (best_plane <= 0.0 && nt > 0)
  {
    /* inside (or on) every face plane: the nearest boundary point is the projection on the plane of face kin */
    const double *a = tri + 9 * kin, *b = a + 3, *c = a + 6;
    const double ab[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] }, ac[3] = { c[0] - a[0], c[1] - a[1], c[2] - a[2] };
    double n[3] = { ab[1] * ac[2] - ab[2] * ac[1], ab[2] * ac[0] - ab[0] * ac[2], ab[0] * ac[1] - ab[1] * ac[0] };
    const double nl = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    for (int i = 0; i < 3; ++i)
      q[i] = p[i] - best_plane * (n[i] / nl);
    return best_plane;
  }
  q[0] = qb[0];
  q[1] = qb[1];
  q[2] = qb[2];
  return sqrt(best_d2);
}

/* signed distance to a box or mesh core */
TMX_GM_FN double tmx_shape_sdf(const double oc[3], const double* ob, const double* mesh, const double p[3], double q[3])
{
  if (tmx_is_mesh(ob))
    return tmx_mesh_sdf(mesh + (long long)ob[2], (int)ob[1], p, q);
  return tmx_box_sdf(oc, ob, p, q);
}

/* point of the obstacle core closest to c for any primitive; returns 1 when c lies inside a box core (q is then the nearest face
 * point: the caller's distance is negative and its normal points from q to c) */
TMX_GM_FN int tmx_obstacle_closest_to_point_b(const double oc[3], const double* oa, const double* ob, const double* mesh, const double c[3],
                                              double q[3])
{
  if (tmx_is_box(ob) || tmx_is_mesh(ob))
    return tmx_shape_sdf(oc, ob, mesh, c, q) < 0.0 ? 1 : 0;
  tmx_obstacle_closest_to_point(oc, oa, c, q);
  return 0;
}

/* swept centre / capsule-link segment  P(tau) = ca + tau e  against any obstacle primitive.  Box: the signed distance of a convex
 * set is convex along the segment - a fixed 64-step golden-section search (deterministic: the same operations in the oracle and in
 * the kernels), then the end points take over when they are at least as close (tau is exactly 0 or 1 there, as the evaluators'
 * cc_type tests expect; a tie goes to the start of the sweep). */
TMX_GM_FN double tmx_swept_closest_to_obstacle_b(const double ca[3], const double e[3], const double oc[3], const double* oa, const double* ob,
                                                 const double* mesh, double q[3], int* inside)
{
  *inside = 0;
  if (!tmx_is_box(ob) && !tmx_is_mesh(ob))
    return tmx_swept_closest_to_obstacle(ca, e, oc, oa, q);
  const double ee = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
  double p[3], qq[3];
  double tau = 0.0;
  double f0 = tmx_shape_sdf(oc, ob, mesh, ca, q);
  if (ee > TMX_GM_EPS)
  {
    const double gr = 0.6180339887498949; /* (sqrt(5) - 1) / 2 */
    double a = 0.0, b = 1.0;
    double x1 = b - gr * (b - a), x2 = a + gr * (b - a);
    for (int i = 0; i < 3; ++i)
      p[i] = ca[i] + x1 * e[i];
    double f1 = tmx_shape_sdf(oc, ob, mesh, p, qq);
    for (int i = 0; i < 3; ++i)
      p[i] = ca[i] + x2 * e[i];
    double f2 = tmx_shape_sdf(oc, ob, mesh, p, qq);
    for (int it = 0; it < 64; ++it)
    {
      if (f1 <= f2)
      {
        b = x2;
        x2 = x1;
        f2 = f1;
        x1 = b - gr * (b - a);
        for (int i = 0; i < 3; ++i)
          p[i] = ca[i] + x1 * e[i];
        f1 = tmx_shape_sdf(oc, ob, mesh, p, qq);
      }
      else
      {
        a = x1;
        x1 = x2;
        f1 = f2;
        x2 = a + gr * (b - a);
        for (int i = 0; i < 3; ++i)
          p[i] = ca[i] + x2 * e[i];
        f2 = tmx_shape_sdf(oc, ob, mesh, p, qq);
      }
    }
    const double tm = 0.5 * (a + b);
    for (int i = 0; i < 3; ++i)
      p[i] = ca[i] + tm * e[i];
    const double fm = tmx_shape_sdf(oc, ob, mesh, p, qq);
    for (int i = 0; i < 3; ++i)
      p[i] = ca[i] + e[i];
    double q1[3];
    const double fe = tmx_shape_sdf(oc, ob, mesh, p, q1);
    if (f0 <= fm && f0 <= fe)
      tau = 0.0; /* q already holds the start point's contact */
    else if (fe <= fm)
    {
      tau = 1.0;
      f0 = fe;
      for (int i = 0; i < 3; ++i)
        q[i] = q1[i];
    }
    else
    {
      tau = tm;
      f0 = fm;
      for (int i = 0; i < 3; ++i)
        q[i] = qq[i];
    }
  }
  *inside = f0 < 0.0 ? 1 : 0;
  return tau;
}

/* normal and signed core distance of a contact between the link core point p and the obstacle core point q: n points from the link
 * towards the obstacle (into it when the link point is inside a box), len = |q - p| with the sign of the penetration */
TMX_GM_FN double tmx_contact_normal(const double p[3], const double q[3], int inside, double n[3])
{
  const double d[3] = { q[0] - p[0], q[1] - p[1], q[2] - p[2] };
  const double len = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  const double sg = inside ? -1.0 : 1.0;
  for (int r = 0; r < 3; ++r)
    n[r] = (len > 0) ? sg * (d[r] / len) : (r == 2 ? 1.0 : 0.0);
  return sg * len;
}

/* a LINK primitive against an obstacle primitive at one configuration: the link sphere (centre c, world frame) or, with a non-zero
 * world axis e, the link CAPSULE swept by that sphere from c to c + e.  p = the point of the link's core (centre / segment) closest
 * to the obstacle's core, q = the obstacle's closest core point; the caller subtracts both radii.  (A capsule link against a capsule
 * obstacle is the two-segment problem of tmx_swept_closest_to_obstacle with the link's own axis in place of the sweep.) */
TMX_GM_FN int tmx_link_closest_to_obstacle_b(const double c[3], const double* e, const double oc[3], const double* oa, const double* ob,
                                             const double* mesh, double p[3], double q[3])
{
  double ee = 0.0;
  if (e)
    ee = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
  if (!(ee > TMX_GM_EPS))
  {
    p[0] = c[0];
    p[1] = c[1];
    p[2] = c[2];
    return tmx_obstacle_closest_to_point_b(oc, oa, ob, mesh, c, q);
  }
  int inside = 0;
  const double s = tmx_swept_closest_to_obstacle_b(c, e, oc, oa, ob, mesh, q, &inside);
  p[0] = c[0] + s * e[0];
  p[1] = c[1] + s * e[1];
  p[2] = c[2] + s * e[2];
  return inside;
}
TMX_GM_FN void tmx_link_closest_to_obstacle(const double c[3], const double* e, const double oc[3], const double* oa, double p[3], double q[3])
{
  double ee = 0.0;
  if (e)
    ee = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
  if (!(ee > TMX_GM_EPS))
  {
    p[0] = c[0];
    p[1] = c[1];
    p[2] = c[2];
    tmx_obstacle_closest_to_point(oc, oa, c, q);
    return;
  }
  const double s = tmx_swept_closest_to_obstacle(c, e, oc, oa, q);
  p[0] = c[0] + s * e[0];
  p[1] = c[1] + s * e[1];
  p[2] = c[2] + s * e[2];
}

/* ---- CONVEX-HULL LINKS (round 4): a link primitive given as a vertex cloud hv[nv][3] in the LINK frame, placed at (R0, t0) -
 * or, for the cast evaluators, swept from (R0, t0) to (R1, t1): the convex hull of both placements, as tesseract's cast shapes are
 * (trajopt/src/collision_terms.cpp:1064-1173; trajopt/test/cast_cost_unit.cpp:64-117 sweeps a box link) - against any obstacle
 * primitive, by GJK / EPA on support functions (tmx_gjk.h).  Same outputs as tmx_link_closest_to_obstacle_b: p on the link's core,
 * q on the obstacle's core, return 1 when the cores overlap (q - p then has the length of the penetration and points back out of the
 * obstacle).  *tau (swept only) = the fraction of the sweep the contact belongs to, by the rule of tesseract's cast contacts: the
 * support vertices v0, v1 of the link at the two poses in the contact direction n; the pose with the larger support owns the contact
 * (tau = 0 / 1), equal supports (a contact on a face spanned between the poses) interpolate by the distances of p to v0 and v1. */
#include "tmx_gjk.h"
#define TMX_GM_SUP_TOL 1e-9
TMX_GM_FN void tmx_obstacle_cvx(const double oc[3], const double* oa, const double* ob, const double* mesh, tmx_cvx* B)
{
  B->a = oc;
  B->b = 0;
  B->n = 0;
  if (tmx_is_mesh(ob))
  {
    B->kind = 3;
    B->a = mesh + (long)ob[2];
    B->n = 3 * (int)ob[1];
  }
  else if (tmx_is_box(ob))
  {
    B->kind = 2;
    B->b = ob;
  }
  else if (oa && (oa[0] * oa[0] + oa[1] * oa[1] + oa[2] * oa[2]) > TMX_GM_EPS)
  {
    B->kind = 1;
    B->b = oa;
  }
  else
    B->kind = 0;
}
TMX_GM_FN int tmx_hull_closest_to_obstacle(const double* hv, int nv, const double* R0, const double* t0, const double* R1, const double* t1,
                                           const double oc[3], const double* oa, const double* ob, const double* mesh, double p[3], double q[3],
                                           double* tau)
{
  tmx_cvx A, B;
  A.kind = R1 ? 5 : 4;
  A.a = hv;
  A.b = 0;
  A.n = nv;
  for (int k = 0; k < 9; ++k)
  {
    A.R[k] = R0[k];
    A.R2[k] = R1 ? R1[k] : R0[k];
  }
  for (int k = 0; k < 3; ++k)
  {
    A.t[k] = t0[k];
    A.t2[k] = R1 ? t1[k] : t0[k];
  }
  tmx_obstacle_cvx(oc, oa, ob, mesh, &B);
  const int inside = tmx_gjk_epa(&A, &B, p, q);
  if (tau)
  {
    *tau = 0.0;
    if (R1)
    {
      double n[3];
      tmx_contact_normal(p, q, inside, n);
      double v0[3], v1[3];
      const double s0 = tmx_gjk_cloud_support(hv, nv, R0, t0, n, v0), s1 = tmx_gjk_cloud_support(hv, nv, R1, t1, n, v1);
      if (s0 - s1 > TMX_GM_SUP_TOL)
        *tau = 0.0;
      else if (s1 - s0 > TMX_GM_SUP_TOL)
        *tau = 1.0;
      else
      {
        const double d0[3] = { p[0] - v0[0], p[1] - v0[1], p[2] - v0[2] }, d1[3] = { p[0] - v1[0], p[1] - v1[1], p[2] - v1[2] };
        const double l0 = sqrt(d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2]), l1 = sqrt(d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2]);
        *tau = (l0 + l1 > 0.0) ? l0 / (l0 + l1) : 0.5;
      }
    }
  }
  return inside;
}

#endif /* TMX_GEOM_H_ */
