/*
 * tmx_gjk.h — distance and penetration between two CONVEX sets given by support functions: GJK (Gilbert-Johnson-Keerthi, closest
 * points of separated sets) and EPA (expanding polytope, penetration depth and witness points of overlapping sets).
 *
 * Why it exists: convex LINK geometry.  The reference gets its contacts from tesseract / Bullet, whose convex-convex narrow phase is
 * GJK + EPA (trajopt/src/collision_terms.cpp:655-691 discrete, :1064-1173 cast: a cast link is the convex hull of the link shape at two
 * poses - here a support function that takes the better of the two poses; trajopt/test/cast_cost_unit.cpp:64-117 sweeps a BOX link).
 * Link spheres / capsules against sphere / capsule / box / mesh obstacles keep their closed forms (tmx_geom.h); a link given as a
 * convex HULL (vertex cloud in the link frame, optionally rounded by a radius) goes through this file, against every obstacle
 * primitive.  ONE statement of the arithmetic, included by the kernels and by the CPU oracle (bit-identical contact data on both
 * sides; fixed iteration caps, no recursion, no allocation).  tesseract / Bullet are absent third-party code: the contact data of
 * hull links are "parity unpinned" against them like the rest of the collision geometry - they are pinned
 * against brute-force geometry instead (tests/test_hull_geometry.py).
 */
#ifndef TMX_GJK_H_
#define TMX_GJK_H_

#if defined(__HIPCC__)
#define TMX_GJK_FN __host__ __device__ static inline
#else
#define TMX_GJK_FN static inline
#endif
#include <math.h>

/* a convex set by its support function */
typedef struct
{
  int kind;         /* 0 point a | 1 segment a .. a + b | 2 box (centre a, b = half extents hx hy hz + rotation world_R_box row-major)
                       | 3 vertex cloud v[n][3] in the world frame | 4 vertex cloud in a local frame at pose R, t
                       | 5 the same swept between two poses (R, t) and (R2, t2): the convex hull of both placements */
  const double* a;
  const double* b;
  int n;
  double R[9], t[3], R2[9], t2[3];
} tmx_cvx;

TMX_GJK_FN double tmx_gjk_dot(const double* x, const double* y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; }

/* farthest vertex of a cloud placed at (R, t) in direction d: the direction goes to the local frame (R' d), ties go to the first */
TMX_GJK_FN double tmx_gjk_cloud_support(const double* v, int n, const double* R, const double* t, const double d[3], double out[3])
{
  const double dl[3] = { R[0] * d[0] + R[3] * d[1] + R[6] * d[2], R[1] * d[0] + R[4] * d[1] + R[7] * d[2], R[2] * d[0] + R[5] * d[1] + R[8] * d[2] };
  int best = 0;
  double bs = v[0] * dl[0] + v[1] * dl[1] + v[2] * dl[2];
  for (int i = 1; i < n; ++i)
  {
    const double s = v[3 * i] * dl[0] + v[3 * i + 1] * dl[1] + v[3 * i + 2] * dl[2];
    if (s > bs)
    {
      bs = s;
      best = i;
    }
  }
  const double* p = v + 3 * best;
  for (int r = 0; r < 3; ++r)
    out[r] = R[3 * r] * p[0] + R[3 * r + 1] * p[1] + R[3 * r + 2] * p[2] + t[r];
  return tmx_gjk_dot(out, d);
}

TMX_GJK_FN void tmx_cvx_support(const tmx_cvx* s, const double d[3], double out[3])
{
  if (s->kind == 0)
  {
    out[0] = s->a[0];
    out[1] = s->a[1];
    out[2] = s->a[2];
  }
  else if (s->kind == 1)
  {
    const double f = tmx_gjk_dot(s->b, d) > 0.0 ? 1.0 : 0.0;
    for (int r = 0; r < 3; ++r)
      out[r] = s->a[r] + f * s->b[r];
  }
  else if (s->kind == 2)
  {
    const double* h = s->b;
    const double* Rb = s->b + 3;
    for (int r = 0; r < 3; ++r)
      out[r] = s->a[r];
    for (int k = 0; k < 3; ++k)
    {
      const double ax[3] = { Rb[k], Rb[3 + k], Rb[6 + k] }; /* column k of world_R_box */
      const double sg = tmx_gjk_dot(ax, d) >= 0.0 ? h[k] : -h[k];
      for (int r = 0; r < 3; ++r)
        out[r] += sg * ax[r];
    }
  }
  else if (s->kind == 3)
  {
    int best = 0;
    double bs = tmx_gjk_dot(s->a, d);
    for (int i = 1; i < s->n; ++i)
    {
      const double v = tmx_gjk_dot(s->a + 3 * i, d);
      if (v > bs)
      {
        bs = v;
        best = i;
      }
    }
    for (int r = 0; r < 3; ++r)
      out[r] = s->a[3 * best + r];
  }
  else
  {
    const double s0 = tmx_gjk_cloud_support(s->a, s->n, s->R, s->t, d, out);
    if (s->kind == 5)
    {
      double o2[3];
      const double s1 = tmx_gjk_cloud_support(s->a, s->n, s->R2, s->t2, d, o2);
      if (s1 > s0) /* a tie goes to the start of the sweep */
      {
        out[0] = o2[0];
        out[1] = o2[1];
        out[2] = o2[2];
      }
    }
  }
}

/* ---- closest point of a simplex of the Minkowski difference to the origin --------------------------------------------------------
 * Simplex vertices w[i] = sa[i] - sb[i] (i < n <= 4).  Returns the barycentric weights lam[] of the closest point (zero for the
 * vertices that do not support it) - the Voronoi-region walk of Ericson, "Real-Time Collision Detection", 5.1.5 / 5.1.6, applied to
 * the origin. */
TMX_GJK_FN void tmx_gjk_closest_segment(const double* a, const double* b, double lam[2])
{
  const double ab[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] };
  const double den = tmx_gjk_dot(ab, ab);
  double t = den > 0.0 ? -tmx_gjk_dot(a, ab) / den : 0.0;
  t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
  lam[0] = 1.0 - t;
  lam[1] = t;
}
TMX_GJK_FN void tmx_gjk_closest_triangle(const double* a, const double* b, const double* c, double lam[3])
{
  const double ab[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] }, ac[3] = { c[0] - a[0], c[1] - a[1], c[2] - a[2] };
  const double d1 = -tmx_gjk_dot(ab, a), d2 = -tmx_gjk_dot(ac, a); /* ap = -a */
  lam[0] = lam[1] = lam[2] = 0.0;
  if (d1 <= 0.0 && d2 <= 0.0)
  {
    lam[0] = 1.0;
    return;
  }
  const double d3 = -tmx_gjk_dot(ab, b), d4 = -tmx_gjk_dot(ac, b);
  if (d3 >= 0.0 && d4 <= d3)
  {
    lam[1] = 1.0;
    return;
  }
  const double vc = d1 * d4 - d3 * d2;
  if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0)
  {
    const double v = d1 / (d1 - d3);
    lam[0] = 1.0 - v;
    lam[1] = v;
    return;
  }
  const double d5 = -tmx_gjk_dot(ab, c), d6 = -tmx_gjk_dot(ac, c);
  if (d6 >= 0.0 && d5 <= d6)
  {
    lam[2] = 1.0;
    return;
  }
  const double vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0)
  {
    const double w = d2 / (d2 - d6);
    lam[0] = 1.0 - w;
    lam[2] = w;
    return;
  }
  const double va = d3 * d6 - d5 * d4;
  if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0)
  {
    const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    lam[1] = 1.0 - w;
    lam[2] = w;
    return;
  }
  const double den = 1.0 / (va + vb + vc);
  lam[1] = vb * den;
  lam[2] = vc * den;
  lam[0] = 1.0 - lam[1] - lam[2];
}
/* origin outside the plane of (a, b, c) on the side away from d?  (degenerate: counts as outside) */
TMX_GJK_FN int tmx_gjk_outside(const double* a, const double* b, const double* c, const double* d)
{
  const double ab[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] }, ac[3] = { c[0] - a[0], c[1] - a[1], c[2] - a[2] };
  const double n[3] = { ab[1] * ac[2] - ab[2] * ac[1], ab[2] * ac[0] - ab[0] * ac[2], ab[0] * ac[1] - ab[1] * ac[0] };
  const double ad[3] = { d[0] - a[0], d[1] - a[1], d[2] - a[2] };
  const double so = -tmx_gjk_dot(n, a), sd = tmx_gjk_dot(n, ad);
  return so * sd <= 0.0;
}
/* returns 1 when the origin lies inside the tetrahedron (lam then untouched) */
TMX_GJK_FN int tmx_gjk_closest_tetra(const double* w, double lam[4])
{
  const double *a = w, *b = w + 3, *c = w + 6, *d = w + 9;
  double best = 1e300;
  int any = 0;
  const int F[4][4] = { { 0, 1, 2, 3 }, { 0, 2, 3, 1 }, { 0, 3, 1, 2 }, { 1, 3, 2, 0 } };
  const double* P[4] = { a, b, c, d };
  for (int f = 0; f < 4; ++f)
    if (tmx_gjk_outside(P[F[f][0]], P[F[f][1]], P[F[f][2]], P[F[f][3]]))
    {
      double l3[3];
      tmx_gjk_closest_triangle(P[F[f][0]], P[F[f][1]], P[F[f][2]], l3);
      double q[3];
      for (int r = 0; r < 3; ++r)
        q[r] = l3[0] * P[F[f][0]][r] + l3[1] * P[F[f][1]][r] + l3[2] * P[F[f][2]][r];
      const double dd = tmx_gjk_dot(q, q);
      if (dd < best)
      {
        best = dd;
        lam[0] = lam[1] = lam[2] = lam[3] = 0.0;
        lam[F[f][0]] = l3[0];
        lam[F[f][1]] = l3[1];
        lam[F[f][2]] = l3[2];
      }
      any = 1;
    }
  return !any;
}

#define TMX_GJK_MAX_ITER 64
#ifndef TMX_EPA_MAX_VERT
#define TMX_EPA_MAX_VERT 40
#define TMX_EPA_MAX_FACE 80
#define TMX_EPA_MAX_ITER 32
#endif
#define TMX_GJK_REL_TOL 1e-13

/* EPA face bookkeeping */
typedef struct
{
  int v[3];
  double n[3], dist;
  int alive;
} tmx_epa_face;

TMX_GJK_FN int tmx_epa_make_face(const double* W, int i, int j, int k, tmx_epa_face* f)
{
  const double *a = W + 3 * i, *b = W + 3 * j, *c = W + 3 * k;
  const double ab[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] }, ac[3] = { c[0] - a[0], c[1] - a[1], c[2] - a[2] };
  double n[3] = { ab[1] * ac[2] - ab[2] * ac[1], ab[2] * ac[0] - ab[0] * ac[2], ab[0] * ac[1] - ab[1] * ac[0] };
  const double len = sqrt(tmx_gjk_dot(n, n));
  f->v[0] = i;
  f->v[1] = j;
  f->v[2] = k;
  f->alive = 1;
  if (!(len > 0.0))
  {
    f->n[0] = f->n[1] = f->n[2] = 0.0;
    f->dist = 1e300; /* degenerate: never the closest face */
    return 0;
  }
  for (int r = 0; r < 3; ++r)
    f->n[r] = n[r] / len;
  f->dist = tmx_gjk_dot(f->n, a);
  if (f->dist < 0.0) /* outward orientation (the origin is inside the polytope) */
  {
    f->v[1] = k;
    f->v[2] = j;
    for (int r = 0; r < 3; ++r)
      f->n[r] = -f->n[r];
    f->dist = -f->dist;
  }
  return 1;
}

/* Closest points (separated: returns 0, pa on A and pb on B at minimum distance) or penetration witnesses (overlapping: returns 1,
 * pa = the point of A deepest inside B, pb = the point of B's boundary it has to be moved to, |pa - pb| = the penetration depth). */
TMX_GJK_FN int tmx_gjk_epa(const tmx_cvx* A, const tmx_cvx* B, double pa[3], double pb[3])
{
  double W[4 * 3], SA[4 * 3], SB[4 * 3], lam[4] = { 1.0, 0.0, 0.0, 0.0 };
  int n = 0;
  double d[3] = { 1.0, 0.0, 0.0 }, v[3] = { 0.0, 0.0, 0.0 };
  {
    double sa[3], sb[3];
    const double nd[3] = { -d[0], -d[1], -d[2] };
    tmx_cvx_support(A, d, sa);
    tmx_cvx_support(B, nd, sb);
    for (int r = 0; r < 3; ++r)
    {
      SA[r] = sa[r];
      SB[r] = sb[r];
      W[r] = sa[r] - sb[r];
      v[r] = W[r];
    }
    n = 1;
  }
  int inside = 0;
  for (int it = 0; it < TMX_GJK_MAX_ITER; ++it)
  {
    const double vv = tmx_gjk_dot(v, v);
    double scale = 0.0; /* size of the simplex: a closest point that is round-off of its vertices is the origin itself */
    for (int i = 0; i < n; ++i)
    {
      const double ww = tmx_gjk_dot(W + 3 * i, W + 3 * i);
      scale = ww > scale ? ww : scale;
    }
    if (!(vv > 1e-28 * scale))
    {
      inside = 1; /* the origin lies on the current simplex: touching or overlapping */
      break;
    }
    const double nd[3] = { -v[0], -v[1], -v[2] };
    double sa[3], sb[3], w[3];
    tmx_cvx_support(A, nd, sa);
    tmx_cvx_support(B, v, sb);
    for (int r = 0; r < 3; ++r)
      w[r] = sa[r] - sb[r];
    /* no progress towards the origin: v is the closest point of the Minkowski difference */
    if (vv - tmx_gjk_dot(v, w) <= TMX_GJK_REL_TOL * vv)
      break;
    /* a vertex already in the simplex cannot be added again (cycling through round-off) */
    int dup = 0;
    for (int i = 0; i < n; ++i)
      if (W[3 * i] == w[0] && W[3 * i + 1] == w[1] && W[3 * i + 2] == w[2])
        dup = 1;
    if (dup)
      break;
    for (int r = 0; r < 3; ++r)
    {
      W[3 * n + r] = w[r];
      SA[3 * n + r] = sa[r];
      SB[3 * n + r] = sb[r];
    }
    ++n;
    if (n == 2)
      tmx_gjk_closest_segment(W, W + 3, lam);
    else if (n == 3)
      tmx_gjk_closest_triangle(W, W + 3, W + 6, lam);
    else if (tmx_gjk_closest_tetra(W, lam))
    {
      inside = 1;
      break;
    }
    /* keep the supporting vertices only */
    int m = 0;
    for (int i = 0; i < n; ++i)
      if (lam[i] > 0.0)
      {
        if (m != i)
          for (int r = 0; r < 3; ++r)
          {
            W[3 * m + r] = W[3 * i + r];
            SA[3 * m + r] = SA[3 * i + r];
            SB[3 * m + r] = SB[3 * i + r];
          }
        lam[m] = lam[i];
        ++m;
      }
    n = m;
    for (int r = 0; r < 3; ++r)
    {
      v[r] = 0.0;
      for (int i = 0; i < n; ++i)
        v[r] += lam[i] * W[3 * i + r];
    }
  }
  if (!inside)
  {
    for (int r = 0; r < 3; ++r)
    {
      pa[r] = pb[r] = 0.0;
      for (int i = 0; i < n; ++i)
      {
        pa[r] += lam[i] * SA[3 * i + r];
        pb[r] += lam[i] * SB[3 * i + r];
      }
    }
    return 0;
  }
  /* ---- EPA: grow the simplex to a tetrahedron that contains the origin, then expand towards the boundary ---------------------- */
  double EW[TMX_EPA_MAX_VERT * 3], EA[TMX_EPA_MAX_VERT * 3]; /* (the B points are not kept: pb follows from pa, depth and direction) */
  int nv = n;
  for (int i = 0; i < 3 * n; ++i)
  {
    EW[i] = W[i];
    EA[i] = SA[i];
  }
  const double AX[6][3] = { { 1, 0, 0 }, { -1, 0, 0 }, { 0, 1, 0 }, { 0, -1, 0 }, { 0, 0, 1 }, { 0, 0, -1 } };
  for (int tries = 0; nv < 4 && tries < 12; ++tries)
  {
    /* a direction that leaves the affine hull of the current simplex: coordinate axes / normals, both signs */
    double dir[3];
    if (nv == 1)
      for (int r = 0; r < 3; ++r)
        dir[r] = AX[tries % 6][r];
    else if (nv == 2)
    {
      const double e[3] = { EW[3] - EW[0], EW[4] - EW[1], EW[5] - EW[2] };
      const double* ax = AX[2 * ((tries / 2) % 3)];
      dir[0] = e[1] * ax[2] - e[2] * ax[1];
      dir[1] = e[2] * ax[0] - e[0] * ax[2];
      dir[2] = e[0] * ax[1] - e[1] * ax[0];
      if (tries & 1)
        for (int r = 0; r < 3; ++r)
          dir[r] = -dir[r];
    }
    else
    {
      const double e1[3] = { EW[3] - EW[0], EW[4] - EW[1], EW[5] - EW[2] }, e2[3] = { EW[6] - EW[0], EW[7] - EW[1], EW[8] - EW[2] };
      dir[0] = e1[1] * e2[2] - e1[2] * e2[1];
      dir[1] = e1[2] * e2[0] - e1[0] * e2[2];
      dir[2] = e1[0] * e2[1] - e1[1] * e2[0];
      if (tries & 1)
        for (int r = 0; r < 3; ++r)
          dir[r] = -dir[r];
    }
    if (!(tmx_gjk_dot(dir, dir) > 0.0))
      continue;
    const double nd[3] = { -dir[0], -dir[1], -dir[2] };
    double sa[3], sb[3], w[3];
    tmx_cvx_support(A, dir, sa);
    tmx_cvx_support(B, nd, sb);
    for (int r = 0; r < 3; ++r)
      w[r] = sa[r] - sb[r];
    /* accept the point if it is affinely independent of the simplex */
    double indep = 0.0;
    if (nv == 1)
    {
      const double e[3] = { w[0] - EW[0], w[1] - EW[1], w[2] - EW[2] };
      indep = tmx_gjk_dot(e, e);
    }
    else if (nv == 2)
    {
      const double e[3] = { EW[3] - EW[0], EW[4] - EW[1], EW[5] - EW[2] }, f[3] = { w[0] - EW[0], w[1] - EW[1], w[2] - EW[2] };
      const double c[3] = { e[1] * f[2] - e[2] * f[1], e[2] * f[0] - e[0] * f[2], e[0] * f[1] - e[1] * f[0] };
      indep = tmx_gjk_dot(c, c);
    }
    else
    {
      const double f[3] = { w[0] - EW[0], w[1] - EW[1], w[2] - EW[2] };
      indep = fabs(tmx_gjk_dot(dir, f));
    }
    if (indep > 1e-24)
    {
      for (int r = 0; r < 3; ++r)
      {
        EW[3 * nv + r] = w[r];
        EA[3 * nv + r] = sa[r];
      }
      ++nv;
    }
  }
  if (nv < 4)
  {
    /* flat Minkowski difference (touching contact of lower-dimensional sets): zero depth at the simplex point nearest the origin */
    for (int r = 0; r < 3; ++r)
      pa[r] = pb[r] = 0.0;
    const double l0 = 1.0 / (double)(n > 0 ? n : 1);
    for (int r = 0; r < 3; ++r)
      for (int i = 0; i < n; ++i)
      {
        pa[r] += l0 * SA[3 * i + r];
        pb[r] += l0 * SB[3 * i + r];
      }
    return 1;
  }
  tmx_epa_face F[TMX_EPA_MAX_FACE];
  int nf = 0;
  tmx_epa_make_face(EW, 0, 1, 2, &F[nf++]);
  tmx_epa_make_face(EW, 0, 3, 1, &F[nf++]);
  tmx_epa_make_face(EW, 0, 2, 3, &F[nf++]);
  tmx_epa_make_face(EW, 1, 3, 2, &F[nf++]);
  int bestf = 0;
  for (int it = 0; it < TMX_EPA_MAX_ITER; ++it)
  {
    bestf = -1;
    for (int f = 0; f < nf; ++f)
      if (F[f].alive && (bestf < 0 || F[f].dist < F[bestf].dist))
        bestf = f;
    if (bestf < 0)
      break;
    const double* nn = F[bestf].n;
    const double nd[3] = { -nn[0], -nn[1], -nn[2] };
    double sa[3], sb[3], w[3];
    tmx_cvx_support(A, nn, sa);
    tmx_cvx_support(B, nd, sb);
    for (int r = 0; r < 3; ++r)
      w[r] = sa[r] - sb[r];
    const double sd = tmx_gjk_dot(w, nn);
    if (sd - F[bestf].dist <= 1e-12 * (1.0 + fabs(sd)) || nv >= TMX_EPA_MAX_VERT)
      break;
    /* remove the faces the new point sees; the horizon = edges of removed faces that are not shared with another removed face */
    int E[3 * TMX_EPA_MAX_FACE][2], ne = 0;
    for (int f = 0; f < nf; ++f)
      if (F[f].alive)
      {
        const double* a = EW + 3 * F[f].v[0];
        const double aw[3] = { w[0] - a[0], w[1] - a[1], w[2] - a[2] };
        if (tmx_gjk_dot(F[f].n, aw) > 0.0)
        {
          F[f].alive = 0;
          for (int e = 0; e < 3; ++e)
          {
            const int p = F[f].v[e], q = F[f].v[(e + 1) % 3];
            int found = -1;
            for (int k = 0; k < ne; ++k)
              if (E[k][0] == q && E[k][1] == p)
                found = k;
            if (found >= 0)
            {
              E[found][0] = E[ne - 1][0];
              E[found][1] = E[ne - 1][1];
              --ne;
            }
            else
            {
              E[ne][0] = p;
              E[ne][1] = q;
              ++ne;
            }
          }
        }
      }
    if (ne == 0)
      break;
    for (int r = 0; r < 3; ++r)
    {
      EW[3 * nv + r] = w[r];
      EA[3 * nv + r] = sa[r];
    }
    const int wi = nv++;
    int full = 0;
    for (int k = 0; k < ne; ++k)
    {
      int slot = -1;
      for (int f = 0; f < nf; ++f)
        if (!F[f].alive)
        {
          slot = f;
          break;
        }
      if (slot < 0)
      {
        if (nf >= TMX_EPA_MAX_FACE)
        {
          full = 1;
          break;
        }
        slot = nf++;
      }
      tmx_epa_make_face(EW, E[k][0], E[k][1], wi, &F[slot]);
    }
    if (full)
      break;
  }
  /* The loop can also end with the face it chose already removed - the vertex / face caps, an empty horizon - and, at the face cap, with
   * that face's slot reused by a new horizon face: take the closest face that is ALIVE now (the regular exit left through the
   * convergence test before anything was removed: this selects the same face again). */
  {
    int alive_best = -1;
    for (int f = 0; f < nf; ++f)
      if (F[f].alive && (alive_best < 0 || F[f].dist < F[alive_best].dist))
        alive_best = f;
    if (alive_best >= 0)
      bestf = alive_best;
  }
  if (bestf < 0)
    bestf = 0;
  /* Witness points.  Depth and direction are those of the closest face (its plane supports the Minkowski difference: exact for
   * polytopes).  The face is one TRIANGLE of a facet that may be a larger polygon, and the projection of the origin can fall outside
   * the triangle (inside the facet): pa = the point of A the triangle's barycentric coordinates of the nearest triangle point give
   * (a point of the contact region, as any narrow phase picks one for a face-face contact), pb = pa moved out of B along the
   * normal - so |pa - pb| is the exact depth and pb - pa the exact direction. */
  {
    const int i = F[bestf].v[0], j = F[bestf].v[1], k = F[bestf].v[2];
    const double q[3] = { F[bestf].n[0] * F[bestf].dist, F[bestf].n[1] * F[bestf].dist, F[bestf].n[2] * F[bestf].dist };
    double ta[3], tb[3], tc[3], l3[3];
    for (int r = 0; r < 3; ++r)
    {
      ta[r] = EW[3 * i + r] - q[r];
      tb[r] = EW[3 * j + r] - q[r];
      tc[r] = EW[3 * k + r] - q[r];
    }
    tmx_gjk_closest_triangle(ta, tb, tc, l3);
    for (int r = 0; r < 3; ++r)
    {
      pa[r] = l3[0] * EA[3 * i + r] + l3[1] * EA[3 * j + r] + l3[2] * EA[3 * k + r];
      pb[r] = pa[r] - q[r];
    }
  }
  return 1;
}

#endif /* TMX_GJK_H_ */
