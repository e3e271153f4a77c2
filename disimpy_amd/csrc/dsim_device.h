/* dsim_device.h - per-walker device functions of the random-walk step (gfx950).
 *
 * Register-resident restatement of disimpy/simulations.py:23-343, :616-679 and
 * numba/cuda/random.py:79-98, :128-167, :199-221.  Walker state lives in
 * scalars (SoA in HBM, registers inside the fused time loop); there are no
 * per-thread arrays in global memory as in the reference.
 *
 * Arithmetic convention (must match oracle/dsim_oracle.c so that the two agree
 * bit for bit; this TU is compiled with -ffp-contract=off):
 *      a*b + c*d + e*f  ->  fma(e, f, fma(a, b, c*d))
 *      a*b - c*d        ->  fma(a, b, -(c*d))
 *      x - y*z          ->  fma(-y, z, x)
 */
#ifndef DSIM_DEVICE_H
#define DSIM_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dsim_math.h"

#define DSD __device__ __forceinline__

struct V3 {
  double x, y, z;
};

struct Rng {
  uint64_t s0, s1;
};

DSD uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

/* numba/cuda/random.py:79-98.  Written on 32-bit halves: gfx950 has no 64-bit
 * rotate and its 64-bit shifts issue at the fp64 rate, while v_alignbit_b32
 * (funnel shift) is a full-rate 32-bit op: rotl(x,55) = rotr(x,9) is two of
 * them, rotl(x,36) is a half swap plus a 4-bit funnel, x<<14 one funnel + one
 * shift.  Pure integer arithmetic: bit-identical to the 64-bit formulation. */
DSD uint64_t rng_next(Rng &r) {
  uint32_t a0 = (uint32_t)r.s0, a1 = (uint32_t)(r.s0 >> 32);
  uint32_t b0 = (uint32_t)r.s1, b1 = (uint32_t)(r.s1 >> 32);
  uint64_t result = r.s0 + r.s1;
  b0 ^= a0;
  b1 ^= a1;
  /* rotl(s0, 55): lo = (lo >> 9) | (hi << 23), hi = (hi >> 9) | (lo << 23) */
  uint32_t r0 = __builtin_amdgcn_alignbit(a1, a0, 9), r1 = __builtin_amdgcn_alignbit(a0, a1, 9);
  /* s1 << 14 */
  uint32_t t0 = b0 << 14, t1 = __builtin_amdgcn_alignbit(b1, b0, 18);
  uint32_t n0 = r0 ^ b0 ^ t0, n1 = r1 ^ b1 ^ t1;
  /* rotl(s1, 36) = rotl(swap halves, 4): hi = (lo << 4) | (hi >> 28), lo = (hi << 4) | (lo >> 28) */
  uint32_t m1 = __builtin_amdgcn_alignbit(b0, b1, 28), m0 = __builtin_amdgcn_alignbit(b1, b0, 28);
  r.s0 = ((uint64_t)n1 << 32) | n0;
  r.s1 = ((uint64_t)m1 << 32) | m0;
  return result;
}

/* numba/cuda/random.py:128-139: (double)(x >> 11) * 2^-53.  The 53-bit integer is split into its
 * top 21 and low 32 bits; hi * 2^-21 and lo * 2^-53 are exact and do not overlap, so the fma is
 * exact as well - the same double as the 64-bit shift + u64 -> f64 conversion (two conversions,
 * two ldexp and an add on gfx950) in 2 integer and 4 floating-point operations. */
DSD double u64_to_unit_f64(uint64_t x) {
  /* hi from the upper dword alone, behind a barrier the optimiser cannot see through: as a shift of the 64-bit value it
   * is converted like a 64-bit integer - ldexp(0, 32) plus an addition of zero per draw */
  uint32_t xh = (uint32_t)(x >> 32);
  asm volatile("" : "+v"(xh));
  const uint32_t hi = xh >> 11, lo = (uint32_t)(x >> 11);
  return __builtin_fma((double)lo, 0x1p-53, (double)hi * 0x1p-21);
}

/* sqrt for the Box-Muller radius and the vector lengths.  For 2^-767 <= x < inf this is the
 * instruction sequence of the compiler's own correctly rounded fp64 sqrt (v_rsq_f64 + two
 * Newton steps with exact residuals) WITHOUT the range scaling and special-case selects around
 * it (8 of 18 instructions) - there they are the identity, so the result is the same bit for
 * bit; everything else (0, tiny, negative, inf, NaN) takes the generic expansion. */
DSD double ds_sqrt_dev(double x) {
  if (!(x >= 0x1p-767 && x < DS_INF)) return __builtin_sqrt(x);
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = y * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  double d = __builtin_fma(-g, g, x);
  h = __builtin_fma(h, r, h);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  return __builtin_fma(d, h, g);
}
DSD double rng_uniform_f64(Rng &r) { return u64_to_unit_f64(rng_next(r)); }
DSD float rng_uniform_f32(Rng &r) { return (float)u64_to_unit_f64(rng_next(r)); }

/* numba/cuda/random.py:199-221: float32 uniforms, float32 log, float64 rest */
DSD double rng_normal_f64(Rng &r, const double *log_tab, const double *cos_tab) {
  float u1 = rng_uniform_f32(r);
  float u2 = rng_uniform_f32(r);
  double lg = (double)ds_logf(u1, log_tab);
  return ds_sqrt_dev(-2.0 * lg) * ds_cos_0_2pi(6.283185307179586 * (double)u2, cos_tab);
}

/* simulations.py:23-36 */
DSD double dot3(const V3 &a, const V3 &b) { return __builtin_fma(a.z, b.z, __builtin_fma(a.x, b.x, a.y * b.y)); }

/* simulations.py:39-56 */
DSD V3 cross3(const V3 &a, const V3 &b) {
  V3 c;
  c.x = __builtin_fma(a.y, b.z, -(a.z * b.y));
  c.y = __builtin_fma(a.z, b.x, -(a.x * b.z));
  c.z = __builtin_fma(a.x, b.y, -(a.y * b.x));
  return c;
}

/* simulations.py:59-74: three divisions by the same length, evaluated with one
 * division + exact residual corrections (ds_div3, same quotients) */
DSD void normalize3(V3 &v) {
  double length = ds_sqrt_dev(dot3(v, v));
  ds_div3(&v.x, &v.y, &v.z, length);
}

/* simulations.py:121-138 without the final normalisation (the caller's: the cooperative mesh kernel normalises the new
 * directions of the walkers that start a step and of those that were reflected in the previous pass in ONE place) */
DSD V3 random_step_raw(Rng &r, const double *log_tab, const double *cos_tab) {
  V3 s;
  s.x = rng_normal_f64(r, log_tab, cos_tab);
  s.y = rng_normal_f64(r, log_tab, cos_tab);
  s.z = rng_normal_f64(r, log_tab, cos_tab);
  return s;
}

/* simulations.py:121-138 */
DSD V3 random_step(Rng &r, const double *log_tab, const double *cos_tab) {
  V3 s = random_step_raw(r, log_tab, cos_tab);
  normalize3(s);
  return s;
}

/* simulations.py:141-160, R row-major */
DSD V3 mat_mul(const double *R, const V3 &v) {
  V3 o;
  o.x = __builtin_fma(R[2], v.z, __builtin_fma(R[0], v.x, R[1] * v.y));
  o.y = __builtin_fma(R[5], v.z, __builtin_fma(R[3], v.x, R[4] * v.y));
  o.z = __builtin_fma(R[8], v.z, __builtin_fma(R[6], v.x, R[7] * v.y));
  return o;
}

/* simulations.py:163-182 on the (y, z) components */
DSD double line_circle_intersection(double r0, double r1, double s0, double s1, double radius) {
  double A = __builtin_fma(s0, s0, s1 * s1);
  double B = 2 * __builtin_fma(r0, s0, r1 * s1);
  double C = __builtin_fma(-radius, radius, __builtin_fma(r0, r0, r1 * r1));
  return (-B + ds_sqrt_dev(__builtin_fma(B, B, -(4 * A * C)))) / (2 * A);
}

/* simulations.py:185-202 */
DSD double line_sphere_intersection(const V3 &r0, const V3 &step, double radius) {
  double dp = dot3(step, r0);
  double inner = __builtin_fma(-radius, radius, dot3(r0, r0));
  return -dp + ds_sqrt_dev(__builtin_fma(dp, dp, -inner)); /* correctly rounded, as the IEEE sqrt it replaces (ds_sqrt_dev) */
}

/* simulations.py:205-231 */
DSD double line_ellipsoid_intersection(const V3 &r0, const V3 &step, double a, double b, double c) {
  double sa = step.x / a, sb = step.y / b, sc = step.z / c;
  double A = __builtin_fma(sc, sc, __builtin_fma(sa, sa, sb * sb));
  double ia = 1.0 / (a * a), ib = 1.0 / (b * b), ic = 1.0 / (c * c);
  double B = 2 * __builtin_fma(ic * step.z, r0.z, __builtin_fma(ia * step.x, r0.x, ib * step.y * r0.y));
  double ra = r0.x / a, rb = r0.y / b, rc = r0.z / c;
  double C = __builtin_fma(rc, rc, __builtin_fma(ra, ra, rb * rb)) - 1;
  return (-B + ds_sqrt_dev(__builtin_fma(B, B, -(4 * A * C)))) / (2 * A);
}

/* simulations.py:234-275 on a pre-gathered record {A, E1 = B - A, E2 = C - A}
 * (the subtractions are done once on the host in the same IEEE arithmetic). */
DSD double ray_triangle_intersection_check(const V3 &A, const V3 &E1, const V3 &E2, const V3 &r0, const V3 &step) {
  V3 T = {r0.x - A.x, r0.y - A.y, r0.z - A.z};
  V3 P = cross3(step, E2);
  V3 Q = cross3(T, E1);
  double det = dot3(P, E1);
  if (det != 0) {
    double inv = 1 / det;
    double t = inv * dot3(Q, E2);
    double u = inv * dot3(P, T);
    double v = inv * dot3(Q, step);
    if (u >= 0 && u <= 1 && v >= 0 && v <= 1 && u + v <= 1) return t;
    return DS_NAN;
  }
  return DS_NAN;
}

/* simulations.py:278-311; updates r0, step and the (possibly flipped) normal.  NORM = false: the reflected direction is
 * left as computed, the caller normalises it (normalize3) before anything reads it */
template <bool NORM = true>
DSD void reflection(V3 &r0, V3 &step, double d, V3 &n, double epsilon) {
  V3 is = {__builtin_fma(d, step.x, r0.x), __builtin_fma(d, step.y, r0.y), __builtin_fma(d, step.z, r0.z)};
  V3 v = {is.x - r0.x, is.y - r0.y, is.z - r0.z};
  double dp = dot3(v, n);
  if (dp > 0) {
    n.x = -n.x;
    n.y = -n.y;
    n.z = -n.z;
    dp = dot3(v, n);
  }
  double m2dp = -(2 * dp);
  step.x = (__builtin_fma(m2dp, n.x, v.x) + is.x) - is.x;
  step.y = (__builtin_fma(m2dp, n.y, v.y) + is.y) - is.y;
  step.z = (__builtin_fma(m2dp, n.z, v.z) + is.z) - is.z;
  if (NORM) normalize3(step);
  r0.x = __builtin_fma(epsilon, n.x, is.x);
  r0.y = __builtin_fma(epsilon, n.y, is.y);
  r0.z = __builtin_fma(epsilon, n.z, is.z);
}

/* simulations.py:314-343 */
DSD void crossing(V3 &r0, const V3 &step, double d, V3 &n, double epsilon) {
  V3 is = {__builtin_fma(d, step.x, r0.x), __builtin_fma(d, step.y, r0.y), __builtin_fma(d, step.z, r0.z)};
  V3 v = {is.x - r0.x, is.y - r0.y, is.z - r0.z};
  double dp = dot3(v, n);
  if (dp < 0) {
    n.x = -n.x;
    n.y = -n.y;
    n.z = -n.z;
  }
  r0.x = __builtin_fma(epsilon, n.x, is.x);
  r0.y = __builtin_fma(epsilon, n.y, is.y);
  r0.z = __builtin_fma(epsilon, n.z, is.z);
}

/* Subvoxel lookups.  The reference scans the boundary array linearly
 * (simulations.py:616-651); the boundaries are np.linspace(0, L, n+1)
 * (substrates.py:506-508), so a multiply gives a first guess that is then
 * corrected against the ACTUAL boundary values, reproducing the reference's
 * strict/non-strict comparisons exactly.  xs points at n+1 doubles in LDS. */
DSD int ll_subvoxel_overlap(const double *xs, int n, double inv_h, double xmin) {
  if (xmin <= xs[0]) return 0;
  if (xmin >= xs[n]) return n;
  int i = (int)(xmin * inv_h);
  i = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
  while (xs[i] > xmin) --i;      /* largest i with xs[i] <= xmin ... */
  while (xs[i + 1] <= xmin) ++i; /* ... = (first i with xs[i] > xmin) - 1 */
  return i;
}

DSD int ul_subvoxel_overlap(const double *xs, int n, double inv_h, double xmax) {
  if (xmax != xmax) return 0; /* NaN: the reference's scan stops at i = 0 */
  if (xmax >= xs[n]) return n;
  if (xmax <= xs[0]) return 0;
  int i = (int)(xmax * inv_h) + 1;
  i = i < 1 ? 1 : (i > n ? n : i);
  while (xs[i - 1] >= xmax) --i; /* smallest i with xs[i] >= xmax */
  while (xs[i] < xmax) ++i;
  return i;
}

/* floor(x / voxel) with the IEEE quotient of the reference (simulations.py:661,
 * :675) but without paying an fp64 division on the common paths:
 *  - 0 <= x < voxel: the correctly rounded quotient is in [0, 1), floor is 0;
 *  - otherwise q = x * (1/voxel) is within 4 ulp of the IEEE quotient, so the
 *    two floors can only differ when q lies within 1e-9 of an integer (for
 *    |q| < 1e6); only then is the real division evaluated. */
DSD double floor_quotient(double x, double voxel, double inv_voxel) {
  if (x >= 0 && x < voxel) return 0.0;
  double q = x * inv_voxel;
  double k = __builtin_floor(q);
  double f = q - k;
  if (!(f > 1e-9 && f < 1.0 - 1e-9 && __builtin_fabs(q) < 1e6)) k = __builtin_floor(x / voxel);
  return k;
}

/* simulations.py:654-679; inv_voxel = 1 / voxel is a first guess only */
DSD int ll_subvoxel_overlap_periodic(const double *xs, int n, double inv_h, double inv_voxel, double x1, double x2) {
  double xmin = x1 < x2 ? x1 : x2;
  double voxel = __builtin_fabs(xs[n] - xs[0]);
  double k = floor_quotient(xmin, voxel, inv_voxel);
  double shifted = __builtin_fma(-k, voxel, xmin);
  return ll_subvoxel_overlap(xs, n, inv_h, shifted) + (int)k * n;
}

DSD int ul_subvoxel_overlap_periodic(const double *xs, int n, double inv_h, double inv_voxel, double x1, double x2) {
  double xmax = x1 > x2 ? x1 : x2;
  double voxel = __builtin_fabs(xs[n] - xs[0]);
  double k = floor_quotient(xmax, voxel, inv_voxel);
  double shifted = __builtin_fma(-k, voxel, xmax);
  return ul_subvoxel_overlap(xs, n, inv_h, shifted) + (int)k * n;
}

/* Both periodic subvoxel limits of one axis for the segment x1 -> x2 without the search
 * loops: the multiply's guess is checked against the two boundaries around it (read from
 * LDS independently of each other and of the other axes).  Returns false when a guess
 * does not satisfy the conditions the reference's scans end on - the caller then uses
 * ll/ul_subvoxel_overlap_periodic.  lo = xs[0], hi = xs[n]. */
DSD bool subvoxel_limits_fast(const double *xs, int n, double lo, double hi, double inv_h, double inv_voxel, double x1, double x2,
                              int &ll, int &ul) {
  const double xmin = x1 < x2 ? x1 : x2, xmax = x1 > x2 ? x1 : x2;
  const double voxel = __builtin_fabs(hi - lo);
  const double kmin = floor_quotient(xmin, voxel, inv_voxel), kmax = floor_quotient(xmax, voxel, inv_voxel);
  const double smin = __builtin_fma(-kmin, voxel, xmin), smax = __builtin_fma(-kmax, voxel, xmax);
  const bool in_min = smin > lo && smin < hi, in_max = smax > lo && smax < hi;
  int gi = in_min ? (int)(smin * inv_h) : 0;
  gi = gi < 0 ? 0 : (gi > n - 1 ? n - 1 : gi);
  int gj = in_max ? (int)(smax * inv_h) + 1 : 1;
  gj = gj < 1 ? 1 : (gj > n ? n : gj);
  const double a = xs[gi], b = xs[gi + 1], c = xs[gj - 1], d = xs[gj];
  /* outside (lo, hi): the early returns of ll/ul_subvoxel_overlap; NaN ends in 0 there as well */
  const int li = in_min ? gi : (smin >= hi ? n : 0);
  const int ui = in_max ? gj : (smax >= hi ? n : 0);
  ll = li + (int)kmin * n;
  ul = ui + (int)kmax * n;
  return (!in_min || (a <= smin && smin < b)) && (!in_max || (c < smax && smax <= d));
}

/* fine-cell coordinate of x on one axis of the clearance map: position inside the periodic voxel
 * (any image) times cells per voxel, clamped.  A rounding error here moves a walker that sits within
 * ~1e-12 m of a cell face into the neighbouring cell; the map is built with a margin of 1e-3 cell
 * sizes for exactly that. */
DSD int clearance_cell(double x, double inv_l, int n_fine) {
  double f = x * inv_l;
  f -= __builtin_floor(f);
  int i = (int)(f * (double)n_fine);
  return i < 0 ? 0 : (i > n_fine - 1 ? n_fine - 1 : i);
}

/* floor division for the periodic image index (Python: math.floor(x / n)) */
DSD int floor_div(int x, int n) {
  int q = x / n;
  return (x % n != 0 && ((x < 0) != (n < 0))) ? q - 1 : q;
}

#endif /* DSIM_DEVICE_H */
