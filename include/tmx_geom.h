/*
 * tmx_geom.h — closest points between a link sphere (or the segment its centre sweeps over a sub-segment of the
 * trajectory) and a world obstacle.  ONE statement of the arithmetic, included by the device kernels
 * (trajopt_amd/csrc/tmx_terms.h) and by the CPU oracle (oracle/trajprob.hpp), so that the contact data - distance,
 * normal, nearest point, cc_time - are bit-identical on both sides (no FMA contraction in either build).
 *
 * Obstacle primitives: SPHERE (centre, radius) and CAPSULE = the sphere swept from `centre` to `centre + axis`
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

/* a LINK primitive against an obstacle primitive at one configuration: the link sphere (centre c, world frame) or, with a non-zero
 * world axis e, the link CAPSULE swept by that sphere from c to c + e.  p = the point of the link's core (centre / segment) closest
 * to the obstacle's core, q = the obstacle's closest core point; the caller subtracts both radii.  (A capsule link against a capsule
 * obstacle is the two-segment problem of tmx_swept_closest_to_obstacle with the link's own axis in place of the sweep.) */
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
