/*
 * tmx_geom.h — closest points between a link sphere (or the segment its centre sweeps over a sub-segment of the
 * trajectory) and a world obstacle.  ONE statement of the arithmetic, included by the device kernels
 * (trajopt_amd/csrc/tmx_terms.h) and by the CPU oracle (oracle/trajprob.hpp), so that the contact data - distance,
 * normal, nearest point, cc_time - are bit-identical on both sides (no FMA contraction in either build).
 *
 * Obstacle primitives: SPHERE (centre, radius), CAPSULE = the sphere swept from `centre` to `centre + axis` and (round 3) the
 * rounded BOX (tmx_problem_desc::obstacle_boxes, signed distance with penetration); link primitives: sphere and capsule.
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

/* point of the obstacle core closest to c for any primitive; returns 1 when c lies inside a box core (q is then the nearest face
 * point: the caller's distance is negative and its normal points from q to c) */
TMX_GM_FN int tmx_obstacle_closest_to_point_b(const double oc[3], const double* oa, const double* ob, const double c[3], double q[3])
{
  if (tmx_is_box(ob))
    return tmx_box_sdf(oc, ob, c, q) < 0.0 ? 1 : 0;
  tmx_obstacle_closest_to_point(oc, oa, c, q);
  return 0;
}

/* swept centre / capsule-link segment  P(tau) = ca + tau e  against any obstacle primitive.  Box: the signed distance of a convex
 * set is convex along the segment - a fixed 64-step golden-section search (deterministic: the same operations in the oracle and in
 * the kernels), then the end points take over when they are at least as close (tau is exactly 0 or 1 there, as the evaluators'
 * cc_type tests expect; a tie goes to the start of the sweep). */
TMX_GM_FN double tmx_swept_closest_to_obstacle_b(const double ca[3], const double e[3], const double oc[3], const double* oa, const double* ob,
                                                 double q[3], int* inside)
{
  *inside = 0;
  if (!tmx_is_box(ob))
    return tmx_swept_closest_to_obstacle(ca, e, oc, oa, q);
  const double ee = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
  double p[3], qq[3];
  double tau = 0.0;
  double f0 = tmx_box_sdf(oc, ob, ca, q);
  if (ee > TMX_GM_EPS)
  {
    const double gr = 0.6180339887498949; /* (sqrt(5) - 1) / 2 */
    double a = 0.0, b = 1.0;
    double x1 = b - gr * (b - a), x2 = a + gr * (b - a);
    for (int i = 0; i < 3; ++i)
      p[i] = ca[i] + x1 * e[i];
    double f1 = tmx_box_sdf(oc, ob, p, qq);
    for (int i = 0; i < 3; ++i)
      p[i] = ca[i] + x2 * e[i];
    double f2 = tmx_box_sdf(oc, ob, p, qq);
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
        f1 = tmx_box_sdf(oc, ob, p, qq);
      }
      else
      {
        a = x1;
        x1 = x2;
        f1 = f2;
        x2 = a + gr * (b - a);
        for (int i = 0; i < 3; ++i)
          p[i] = ca[i] + x2 * e[i];
        f2 = tmx_box_sdf(oc, ob, p, qq);
      }
    }
    const double tm = 0.5 * (a + b);
    for (int i = 0; i < 3; ++i)
      p[i] = ca[i] + tm * e[i];
    const double fm = tmx_box_sdf(oc, ob, p, qq);
    for (int i = 0; i < 3; ++i)
      p[i] = ca[i] + e[i];
    double q1[3];
    const double fe = tmx_box_sdf(oc, ob, p, q1);
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
                                             double p[3], double q[3])
{
  double ee = 0.0;
  if (e)
    ee = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
  if (!(ee > TMX_GM_EPS))
  {
    p[0] = c[0];
    p[1] = c[1];
    p[2] = c[2];
    return tmx_obstacle_closest_to_point_b(oc, oa, ob, c, q);
  }
  int inside = 0;
  const double s = tmx_swept_closest_to_obstacle_b(c, e, oc, oa, ob, q, &inside);
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

#endif /* TMX_GEOM_H_ */
