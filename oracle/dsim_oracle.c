/* dsim_oracle.c - CPU restatement of disimpy's Monte Carlo hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load this library; the product
 * (disimpy_amd/) never does and fails loudly without its HIP extension.
 *
 * What is restated (reference = /root/reference/disimpy, numba-src =
 * numba/cuda/random.py, numba 0.54.1 as read in the survey; the reference pins
 * numba>=0.58.1 (setup.py:18) - the algorithm is the public xoroshiro128+ /
 * splitmix64 / Box-Muller one, which has not changed):
 *   RNG            numba-src random.py:45-68 (splitmix64 seeding), :71-98
 *                  (rotl, next: the 55/14/36 variant), :101-125 (jump),
 *                  :128-167 (u64 -> float64/float32), :199-221
 *                  (normal_float64: FLOAT32 uniforms, log on a float32),
 *                  :224-240 (state i = jump(state i-1))
 *   device fns     simulations.py:23-343, :616-679
 *   step kernels   simulations.py:682-1013 (free, sphere, cylinder, ellipsoid, mesh)
 *   fill_mesh      simulations.py:421-502 (+ host loop :552-579)
 *   signal         simulations.py:1413-1421
 * The layout is the reference's: positions (N,3) row-major, phases (M,N),
 * g_x/g_y/g_z (M,T), RNG states {s0,s1}, int64 mesh index arrays, linear scans
 * over the subvoxel boundaries, one walker-step at a time.
 *
 * Pinning (tests/test_oracle_*.py): the default build reproduces the
 * reference's golden trajectory tests/golden/traj_free_seed123.npy
 * (= disimpy/tests/test_traj.txt: 10 walkers x 999 free steps, seed 123) to
 * <= 6e-8 per-step direction error (1-ulp differences between glibc logf and
 * CUDA logf), and the known answers of disimpy/tests/test_simulations.py:162-360.
 * For reflecting geometries the reference ships no golden trajectory:
 * walker-level parity with the reference's own kernel is UNPINNED there and is
 * replaced by the reference's statistical fixtures (MISST, containment).
 *
 * Floating-point convention.  The reference's kernels are compiled by NVVM,
 * which contracts a*b+c into fma; which products fuse is not observable from
 * the fixture (positions agree to 5e-13 m either way).  This file fixes ONE
 * convention, written out with explicit fma() and compiled with
 * -ffp-contract=off: in a sum of products the LEFT product of the innermost
 * add/sub is fused first, i.e.
 *      a*b + c*d + e*f  ->  fma(e, f, fma(a, b, c*d))
 *      a*b - c*d        ->  fma(a, b, -(c*d))
 *      x - y*z          ->  fma(-y, z, x)
 * (LLVM's DAG combiner order).  The HIP kernels use the same convention, so
 * that the build with -DDSO_SHARED_MATH (log/cos taken from
 * disimpy_amd/csrc/dsim_math.h instead of glibc) is bit-identical to them.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef DSO_SHARED_MATH
#include "../disimpy_amd/csrc/dsim_math.h"
#define DSO_LOGF(x) ds_logf((x), ds_log_table_host)
#define DSO_COS(x) ds_cos_0_2pi((x), ds_cos_table_host)
#else
#define DSO_LOGF(x) logf(x)
#define DSO_COS(x) cos(x)
#endif

#define DSO_GAMMA 267.513e6 /* gradients.py:13 */

typedef struct {
  uint64_t s0, s1;
} dso_state; /* numba-src random.py:32-34 */

/* 0: log evaluated in float32 (real GPU: math.log on a float32 -> logf);
 * 1: log evaluated in float64 on the float32 value (NUMBA_ENABLE_CUDASIM). */
static int g_log_mode = 0;
void dso_set_log_mode(int mode) { g_log_mode = mode; }
int dso_shared_math(void) {
#ifdef DSO_SHARED_MATH
  return 1;
#else
  return 0;
#endif
}

/* ------------------------------------------------------------------ RNG --- */

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

/* numba-src random.py:79-98 */
static inline uint64_t xoro_next(dso_state *s) {
  uint64_t s0 = s->s0, s1 = s->s1;
  uint64_t result = s0 + s1;
  s1 ^= s0;
  s->s0 = rotl64(s0, 55) ^ s1 ^ (s1 << 14);
  s->s1 = rotl64(s1, 36);
  return result;
}

/* numba-src random.py:101-125 */
static void xoro_jump(dso_state *s) {
  static const uint64_t jump[2] = {0xbeac0467eba5facbULL, 0xd86b048b86aa9922ULL};
  uint64_t a = 0, c = 0;
  for (int i = 0; i < 2; i++)
    for (int b = 0; b < 64; b++) {
      if (jump[i] & ((uint64_t)1 << b)) {
        a ^= s->s0;
        c ^= s->s1;
      }
      xoro_next(s);
    }
  s->s0 = a;
  s->s1 = c;
}

/* numba-src random.py:45-68 + :224-240 */
void dso_rng_init_states(dso_state *st, int64_t n, uint64_t seed, uint64_t subsequence_start) {
  if (n < 1) return;
  uint64_t z = seed + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z = z ^ (z >> 31);
  st[0].s0 = z;
  st[0].s1 = z;
  for (uint64_t k = 0; k < subsequence_start; k++) xoro_jump(&st[0]);
  for (int64_t i = 1; i < n; i++) {
    st[i] = st[i - 1];
    xoro_jump(&st[i]);
  }
}

/* numba-src random.py:128-167 */
static inline double u64_to_unit_f64(uint64_t x) { return (double)(x >> 11) * (1.0 / 9007199254740992.0); }
static inline double uniform_f64(dso_state *s) { return u64_to_unit_f64(xoro_next(s)); }
static inline float uniform_f32(dso_state *s) { return (float)u64_to_unit_f64(xoro_next(s)); }

/* numba-src random.py:199-221 */
static inline double normal_f64(dso_state *s) {
  float u1 = uniform_f32(s);
  float u2 = uniform_f32(s);
  double lg = g_log_mode ? log((double)u1) : (double)DSO_LOGF(u1);
  return sqrt(-2.0 * lg) * DSO_COS(6.283185307179586 * (double)u2);
}

/* --------------------------------------------------- device functions --- */

/* simulations.py:23-36 */
static inline double dot3(const double *a, const double *b) { return fma(a[2], b[2], fma(a[0], b[0], a[1] * b[1])); }

/* simulations.py:39-56 */
static inline void cross3(const double *a, const double *b, double *c) {
  c[0] = fma(a[1], b[2], -(a[2] * b[1]));
  c[1] = fma(a[2], b[0], -(a[0] * b[2]));
  c[2] = fma(a[0], b[1], -(a[1] * b[0]));
}

/* simulations.py:59-74 */
static inline void normalize3(double *v) {
  double length = sqrt(dot3(v, v));
#ifdef DSO_SHARED_MATH
  ds_div3(&v[0], &v[1], &v[2], length); /* the product's evaluation of the same three quotients */
#else
  for (int i = 0; i < 3; i++) v[i] = v[i] / length;
#endif
}

/* simulations.py:77-97; triangle is (3,3) row-major */
static inline void triangle_normal(const double *tri, double *normal) {
  double v[3], k[3];
  for (int i = 0; i < 3; i++) {
    v[i] = tri[0 * 3 + i] - tri[1 * 3 + i];
    k[i] = tri[0 * 3 + i] - tri[2 * 3 + i];
  }
  cross3(v, k, normal);
  normalize3(normal);
}

/* simulations.py:100-118 */
static inline void get_triangle(int64_t i, const double *vertices, const int64_t *faces, double *tri) {
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) tri[a * 3 + b] = vertices[faces[i * 3 + a] * 3 + b];
}

/* simulations.py:121-138 */
static inline void random_step(double *step, dso_state *s) {
  for (int i = 0; i < 3; i++) step[i] = normal_f64(s);
  normalize3(step);
}

/* simulations.py:141-160; R is (3,3) row-major */
static inline void mat_mul(const double *R, double *v) {
  double r[3];
  for (int i = 0; i < 3; i++) r[i] = fma(R[i * 3 + 2], v[2], fma(R[i * 3 + 0], v[0], R[i * 3 + 1] * v[1]));
  for (int i = 0; i < 3; i++) v[i] = r[i];
}

/* simulations.py:163-182 (r0, step are 2-vectors) */
static inline double line_circle_intersection(const double *r0, const double *step, double radius) {
  double A = fma(step[0], step[0], step[1] * step[1]);
  double B = 2 * fma(r0[0], step[0], r0[1] * step[1]);
  double C = fma(-radius, radius, fma(r0[0], r0[0], r0[1] * r0[1]));
  return (-B + sqrt(fma(B, B, -(4 * A * C)))) / (2 * A);
}

/* simulations.py:185-202 */
static inline double line_sphere_intersection(const double *r0, const double *step, double radius) {
  double dp = dot3(step, r0);
  double inner = fma(-radius, radius, dot3(r0, r0));
  return -dp + sqrt(fma(dp, dp, -inner));
}

/* simulations.py:205-231 */
static inline double line_ellipsoid_intersection(const double *r0, const double *step, const double *semiaxes) {
  double a = semiaxes[0], b = semiaxes[1], c = semiaxes[2];
  double sa = step[0] / a, sb = step[1] / b, sc = step[2] / c;
  double A = fma(sc, sc, fma(sa, sa, sb * sb));
  double ia = 1.0 / (a * a), ib = 1.0 / (b * b), ic = 1.0 / (c * c); /* a ** (-2) */
  double B = 2 * fma(ic * step[2], r0[2], fma(ia * step[0], r0[0], ib * step[1] * r0[1]));
  double ra = r0[0] / a, rb = r0[1] / b, rc = r0[2] / c;
  double C = fma(rc, rc, fma(ra, ra, rb * rb)) - 1;
  return (-B + sqrt(fma(B, B, -(4 * A * C)))) / (2 * A);
}

/* simulations.py:234-275; returns NaN when there is no intersection */
static inline double ray_triangle_intersection_check(const double *tri, const double *r0, const double *step) {
  const double *A = tri, *B = tri + 3, *C = tri + 6;
  double T[3], E1[3], E2[3], P[3], Q[3];
  for (int i = 0; i < 3; i++) {
    T[i] = r0[i] - A[i];
    E1[i] = B[i] - A[i];
    E2[i] = C[i] - A[i];
  }
  cross3(step, E2, P);
  cross3(T, E1, Q);
  double det = dot3(P, E1);
  if (det != 0) {
    double inv = 1 / det;
    double t = inv * dot3(Q, E2);
    double u = inv * dot3(P, T);
    double v = inv * dot3(Q, step);
    if (u >= 0 && u <= 1 && v >= 0 && v <= 1 && u + v <= 1) return t;
    return NAN;
  }
  return NAN;
}

/* simulations.py:278-311; mutates r0, step AND normal */
static inline void reflection(double *r0, double *step, double d, double *normal, double epsilon) {
  double isect[3], v[3];
  for (int i = 0; i < 3; i++) {
    isect[i] = fma(d, step[i], r0[i]);
    v[i] = isect[i] - r0[i];
  }
  double dp = dot3(v, normal);
  if (dp > 0) {
    for (int i = 0; i < 3; i++) normal[i] *= -1;
    dp = dot3(v, normal);
  }
  for (int i = 0; i < 3; i++) step[i] = (fma(-(2 * dp), normal[i], v[i]) + isect[i]) - isect[i];
  normalize3(step);
  for (int i = 0; i < 3; i++) r0[i] = fma(epsilon, normal[i], isect[i]);
}

/* simulations.py:314-343 */
static inline void crossing(double *r0, const double *step, double d, double *normal, double epsilon) {
  double isect[3], v[3];
  for (int i = 0; i < 3; i++) {
    isect[i] = fma(d, step[i], r0[i]);
    v[i] = isect[i] - r0[i];
  }
  double dp = dot3(v, normal);
  if (dp < 0)
    for (int i = 0; i < 3; i++) normal[i] *= -1;
  for (int i = 0; i < 3; i++) r0[i] = fma(epsilon, normal[i], isect[i]);
}

/* simulations.py:616-632; n_xs = len(xs) */
static inline int64_t ll_subvoxel_overlap(const double *xs, int64_t n_xs, double x1, double x2) {
  double xmin = x1 < x2 ? x1 : x2; /* min(x1, x2) */
  if (xmin <= xs[0]) return 0;
  if (xmin >= xs[n_xs - 1]) return n_xs - 1;
  for (int64_t i = 0; i < n_xs; i++)
    if (xs[i] > xmin) return i - 1;
  return 0;
}

/* simulations.py:635-651 */
static inline int64_t ul_subvoxel_overlap(const double *xs, int64_t n_xs, double x1, double x2) {
  double xmax = x1 > x2 ? x1 : x2;
  if (xmax >= xs[n_xs - 1]) return n_xs - 1;
  if (xmax <= xs[0]) return 0;
  for (int64_t i = 0; i < n_xs; i++)
    if (!(xs[i] < xmax)) return i;
  return n_xs - 1;
}

/* simulations.py:654-665 */
static inline int64_t ll_subvoxel_overlap_periodic(const double *xs, int64_t n_xs, double x1, double x2) {
  double xmin = x1 < x2 ? x1 : x2;
  double voxel_size = fabs(xs[n_xs - 1] - xs[0]);
  double n = floor(xmin / voxel_size);
  double xmin_shifted = fma(-n, voxel_size, xmin);
  int64_t ll_shifted = ll_subvoxel_overlap(xs, n_xs, xmin_shifted, xmin_shifted);
  return ll_shifted + (int64_t)n * (n_xs - 1);
}

/* simulations.py:668-679 */
static inline int64_t ul_subvoxel_overlap_periodic(const double *xs, int64_t n_xs, double x1, double x2) {
  double xmax = x1 > x2 ? x1 : x2;
  double voxel_size = fabs(xs[n_xs - 1] - xs[0]);
  double n = floor(xmax / voxel_size);
  double xmax_shifted = fma(-n, voxel_size, xmax);
  int64_t ul_shifted = ul_subvoxel_overlap(xs, n_xs, xmax_shifted, xmax_shifted);
  return ul_shifted + (int64_t)n * (n_xs - 1);
}

/* phase accumulation shared by all step kernels, e.g. simulations.py:692-701 */
static inline void accumulate_phase(int64_t tid, int64_t N, int M, int T, int t, const double *gx, const double *gy,
                                    const double *gz, const double *pos, double *phases, double dt) {
  double gdt = DSO_GAMMA * dt;
  for (int m = 0; m < M; m++) {
    double inner = fma(gz[(int64_t)m * T + t], pos[2], fma(gx[(int64_t)m * T + t], pos[0], gy[(int64_t)m * T + t] * pos[1]));
    phases[(int64_t)m * N + tid] = fma(gdt, inner, phases[(int64_t)m * N + tid]);
  }
}

/* ------------------------------------------------------ step kernels --- */

typedef struct {
  int type; /* 0 free, 1 sphere, 2 cylinder, 3 ellipsoid, 4 mesh */
  double radius;
  double semiaxes[3];
  double R[9], R_inv[9];
  /* mesh */
  const double *vertices;
  const int64_t *faces;
  const double *xs, *ys, *zs;
  int64_t n_xs, n_ys, n_zs; /* lengths of xs, ys, zs (= n_sv + 1) */
  const int64_t *subvoxel_indices; /* (prod n_sv, 2) */
  const int64_t *triangle_indices;
  int64_t n_sv[3];
  double perm_prob;
  int64_t max_iter;
  double epsilon;
} dso_geom;

/* simulations.py:682-702 */
static void step_free(int64_t tid, double *positions, dso_state *st, double step_l) {
  double step[3];
  random_step(step, &st[tid]);
  for (int i = 0; i < 3; i++) positions[tid * 3 + i] = fma(step[i], step_l, positions[tid * 3 + i]);
}

/* simulations.py:705-745 */
static void step_sphere(int64_t tid, double *positions, dso_state *st, double step_l, const dso_geom *g,
                        uint8_t *iter_exc) {
  double step[3];
  random_step(step, &st[tid]);
  double *r0 = &positions[tid * 3];
  int64_t iter_idx = 0;
  int check_intersection = 1;
  while (check_intersection && step_l > 0 && iter_idx < g->max_iter) {
    iter_idx++;
    double d = line_sphere_intersection(r0, step, g->radius);
    if (d > 0 && d < step_l) {
      double normal[3];
      for (int i = 0; i < 3; i++) normal[i] = -fma(d, step[i], r0[i]);
      normalize3(normal);
      reflection(r0, step, d, normal, g->epsilon);
      step_l -= d + g->epsilon;
    } else {
      check_intersection = 0;
    }
  }
  if (iter_idx >= g->max_iter) iter_exc[tid] = 1;
  for (int i = 0; i < 3; i++) positions[tid * 3 + i] = fma(step[i], step_l, r0[i]);
}

/* simulations.py:759-805 */
static void step_cylinder(int64_t tid, double *positions, dso_state *st, double step_l, const dso_geom *g,
                          uint8_t *iter_exc) {
  double step[3];
  random_step(step, &st[tid]);
  double *r0 = &positions[tid * 3];
  mat_mul(g->R, r0); /* to the cylinder frame, in place like the reference */
  int64_t iter_idx = 0;
  int check_intersection = 1;
  while (check_intersection && step_l > 0 && iter_idx < g->max_iter) {
    iter_idx++;
    double d = line_circle_intersection(r0 + 1, step + 1, g->radius);
    if (d > 0 && d < step_l) {
      double normal[3];
      normal[0] = 0;
      for (int i = 1; i < 3; i++) normal[i] = -fma(d, step[i], r0[i]);
      normalize3(normal);
      reflection(r0, step, d, normal, g->epsilon);
      step_l -= d + g->epsilon;
    } else {
      check_intersection = 0;
    }
  }
  if (iter_idx >= g->max_iter) iter_exc[tid] = 1;
  mat_mul(g->R_inv, step);
  mat_mul(g->R_inv, r0);
  for (int i = 0; i < 3; i++) positions[tid * 3 + i] = fma(step[i], step_l, r0[i]);
}

/* simulations.py:819-864 */
static void step_ellipsoid(int64_t tid, double *positions, dso_state *st, double step_l, const dso_geom *g,
                           uint8_t *iter_exc) {
  double step[3];
  random_step(step, &st[tid]);
  double *r0 = &positions[tid * 3];
  mat_mul(g->R, r0);
  int64_t iter_idx = 0;
  int check_intersection = 1;
  while (check_intersection && step_l > 0 && iter_idx < g->max_iter) {
    iter_idx++;
    double d = line_ellipsoid_intersection(r0, step, g->semiaxes);
    if (d > 0 && d < step_l) {
      double normal[3];
      for (int i = 0; i < 3; i++) normal[i] = -fma(d, step[i], r0[i]) / (g->semiaxes[i] * g->semiaxes[i]);
      normalize3(normal);
      reflection(r0, step, d, normal, g->epsilon);
      step_l -= d + g->epsilon;
    } else {
      check_intersection = 0;
    }
  }
  if (iter_idx >= g->max_iter) iter_exc[tid] = 1;
  mat_mul(g->R_inv, step);
  mat_mul(g->R_inv, r0);
  for (int i = 0; i < 3; i++) positions[tid * 3 + i] = fma(step[i], step_l, r0[i]);
}

/* Python's math.floor(x / n) for integers */
static inline int64_t floor_div_as_float(int64_t x, int64_t n) { return (int64_t)floor((double)x / (double)n); }

/* simulations.py:924-997: the reflect loop of the mesh kernel for a given start r0, direction `step` and remaining
 * length *step_l_io, as many iterations as it takes (max_passes <= 0) or at most max_passes of them.  Returns 1
 * when the loop is over (no triangle within the remaining length, or the while-condition no longer holds), 0 when
 * it was cut short by max_passes.  r0, step, *step_l_io, *iter_idx_io and the RNG state are left as the loop
 * leaves them; the final advance of the position (simulations.py:1001-1002) is the caller's. */
static int mesh_reflect_loop(double *r0, double *step, double *step_l_io, dso_state *st, const dso_geom *g,
                             int64_t *iter_idx_io, int64_t max_passes) {
  double triangle[9], normal[3], shifts[3], temp_r0[3];
  int64_t lls[3], uls[3];
  double step_l = *step_l_io;
  int64_t iter_idx = *iter_idx_io, passes = 0;
  int64_t closest_triangle_index = 0;
  int over = 1;
  const int64_t nx = g->n_xs - 1, ny = g->n_ys - 1, nz = g->n_zs - 1;
  while (step_l > 0 && iter_idx < g->max_iter) { /* check_intersection leaves through `break` */
    if (max_passes > 0 && passes == max_passes) {
      over = 0;
      break;
    }
    passes++;
    iter_idx++;
    double min_d = INFINITY;
    lls[0] = ll_subvoxel_overlap_periodic(g->xs, g->n_xs, r0[0], fma(step[0], step_l, r0[0]));
    lls[1] = ll_subvoxel_overlap_periodic(g->ys, g->n_ys, r0[1], fma(step[1], step_l, r0[1]));
    lls[2] = ll_subvoxel_overlap_periodic(g->zs, g->n_zs, r0[2], fma(step[2], step_l, r0[2]));
    uls[0] = ul_subvoxel_overlap_periodic(g->xs, g->n_xs, r0[0], fma(step[0], step_l, r0[0]));
    uls[1] = ul_subvoxel_overlap_periodic(g->ys, g->n_ys, r0[1], fma(step[1], step_l, r0[1]));
    uls[2] = ul_subvoxel_overlap_periodic(g->zs, g->n_zs, r0[2], fma(step[2], step_l, r0[2]));
    for (int64_t xi = lls[0]; xi < uls[0]; xi++) {
      int64_t x = xi;
      if (x < 0 || x > g->n_xs - 2) {
        int64_t shift_n = floor_div_as_float(x, nx);
        x -= shift_n * nx;
        shifts[0] = (double)shift_n * g->xs[g->n_xs - 1];
      } else {
        shifts[0] = 0;
      }
      for (int64_t yi = lls[1]; yi < uls[1]; yi++) {
        int64_t y = yi;
        if (y < 0 || y > g->n_ys - 2) {
          int64_t shift_n = floor_div_as_float(y, ny);
          y -= shift_n * ny;
          shifts[1] = (double)shift_n * g->ys[g->n_ys - 1];
        } else {
          shifts[1] = 0;
        }
        for (int64_t zi = lls[2]; zi < uls[2]; zi++) {
          int64_t z = zi;
          if (z < 0 || z > g->n_zs - 2) {
            int64_t shift_n = floor_div_as_float(z, nz);
            z -= shift_n * nz;
            shifts[2] = (double)shift_n * g->zs[g->n_zs - 1];
          } else {
            shifts[2] = 0;
          }
          int64_t sv = x * g->n_sv[1] * g->n_sv[2] + y * g->n_sv[2] + z;
          for (int i = 0; i < 3; i++) temp_r0[i] = r0[i] - shifts[i];
          for (int64_t i = g->subvoxel_indices[sv * 2]; i < g->subvoxel_indices[sv * 2 + 1]; i++) {
            get_triangle(g->triangle_indices[i], g->vertices, g->faces, triangle);
            double d = ray_triangle_intersection_check(triangle, temp_r0, step);
            if (d > 0 && d < min_d) {
              closest_triangle_index = g->triangle_indices[i];
              min_d = d;
            }
          }
        }
      }
    }
    if (min_d > step_l) {
      break; /* check_intersection = False */
    } else if (g->perm_prob < uniform_f64(st)) {
      get_triangle(closest_triangle_index, g->vertices, g->faces, triangle);
      triangle_normal(triangle, normal);
      reflection(r0, step, min_d, normal, g->epsilon);
      step_l -= min_d;
    } else {
      get_triangle(closest_triangle_index, g->vertices, g->faces, triangle);
      triangle_normal(triangle, normal);
      crossing(r0, step, min_d, normal, g->epsilon);
      step_l -= min_d;
    }
  }
  *step_l_io = step_l;
  *iter_idx_io = iter_idx;
  return over;
}

/* simulations.py:878-1002 */
static void step_mesh(int64_t tid, double *positions, dso_state *st, double step_l, const dso_geom *g,
                      uint8_t *iter_exc) {
  double step[3];
  double *r0 = &positions[tid * 3];
  random_step(step, &st[tid]);
  int64_t iter_idx = 0;
  mesh_reflect_loop(r0, step, &step_l, &st[tid], g, &iter_idx, 0);
  if (iter_idx >= g->max_iter) iter_exc[tid] = 1;
  for (int i = 0; i < 3; i++) positions[tid * 3 + i] = fma(step[i], step_l, r0[i]);
}

static inline void one_walker_step(int64_t tid, const dso_geom *g, double *positions, dso_state *st, double step_l,
                                   uint8_t *iter_exc) {
  switch (g->type) {
    case 0: step_free(tid, positions, st, step_l); break;
    case 1: step_sphere(tid, positions, st, step_l, g, iter_exc); break;
    case 2: step_cylinder(tid, positions, st, step_l, g, iter_exc); break;
    case 3: step_ellipsoid(tid, positions, st, step_l, g, iter_exc); break;
    default: step_mesh(tid, positions, st, step_l, g, iter_exc); break;
  }
}

/* The reference's host loop (simulations.py:1198-1216 and siblings): for each
 * time step t in [t_begin, t_end) launch one step for every walker.  Walkers
 * are independent, so the loop nest is walker-major here (same results);
 * n_threads > 1 splits the walkers with OpenMP.  traj (optional, may be NULL)
 * receives positions after every step: (t_end - t_begin, N, 3). */
void dso_run(const dso_geom *g, int64_t N, int M, int T, const double *gx, const double *gy, const double *gz,
             double dt, double step_l, double *positions, double *phases, dso_state *st, uint8_t *iter_exc,
             int t_begin, int t_end, double *traj, int n_threads) {
  (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
#endif
  for (int64_t tid = 0; tid < N; tid++) {
    for (int t = t_begin; t < t_end; t++) {
      one_walker_step(tid, g, positions, st, step_l, iter_exc);
      accumulate_phase(tid, N, M, T, t, gx, gy, gz, &positions[tid * 3], phases, dt);
      if (traj)
        for (int i = 0; i < 3; i++) traj[((int64_t)(t - t_begin) * N + tid) * 3 + i] = positions[tid * 3 + i];
    }
  }
}

/* simulations.py:1413-1421: real(nansum(exp(1j*phases))) with exceeded walkers
 * set to NaN; also returns the imaginary part (sum of sines). */
void dso_signal(const double *phases, const uint8_t *iter_exc, int M, int64_t N, double *out_cos, double *out_sin) {
  for (int m = 0; m < M; m++) {
    double c = 0, s = 0;
    for (int64_t i = 0; i < N; i++) {
      double p = phases[(int64_t)m * N + i];
      if (iter_exc[i] || isnan(p)) continue;
      double cv = cos(p), sv = sin(p);
      if (isnan(cv) || isnan(sv)) continue; /* nansum over the complex exponentials */
      c += cv;
      s += sv;
    }
    out_cos[m] = c;
    out_sin[m] = s;
  }
}

/* ---------------------------------------------------------- fill_mesh --- */

/* One launch of _cuda_fill_mesh (simulations.py:421-502) over all threads. */
static void fill_mesh_attempt(double *points, int64_t n_points, dso_state *st, int intra, const dso_geom *g,
                              const double *voxel_size) {
  for (int64_t tid = 0; tid < n_points; tid++) {
    if (points[tid * 3] != INFINITY) continue;
    double point[3], ray[3] = {1.0, 0.0, 0.0}, triangle[9];
    for (int i = 0; i < 3; i++) point[i] = uniform_f64(&st[tid]) * voxel_size[i];
    int64_t lls[3], uls[3];
    lls[0] = ll_subvoxel_overlap(g->xs, g->n_xs, point[0], point[0] + ray[0]);
    lls[1] = ll_subvoxel_overlap(g->ys, g->n_ys, point[1], point[1] + ray[1]);
    lls[2] = ll_subvoxel_overlap(g->zs, g->n_zs, point[2], point[2] + ray[2]);
    uls[0] = ul_subvoxel_overlap(g->xs, g->n_xs, point[0], point[0] + ray[0]);
    uls[1] = ul_subvoxel_overlap(g->ys, g->n_ys, point[1], point[1] + ray[1]);
    uls[2] = ul_subvoxel_overlap(g->zs, g->n_zs, point[2], point[2] + ray[2]);
    int n_intersections = 0;
    int aborted = 0;
    int64_t triangles[1000];
    for (int64_t x = lls[0]; x < uls[0] && !aborted; x++)
      for (int64_t y = lls[1]; y < uls[1] && !aborted; y++)
        for (int64_t z = lls[2]; z < uls[2] && !aborted; z++) {
          int64_t sv = x * g->n_sv[1] * g->n_sv[2] + y * g->n_sv[2] + z;
          for (int64_t i = g->subvoxel_indices[sv * 2]; i < g->subvoxel_indices[sv * 2 + 1]; i++) {
            if (n_intersections >= 1000) {
              aborted = 1; /* the reference returns from the kernel here */
              break;
            }
            get_triangle(g->triangle_indices[i], g->vertices, g->faces, triangle);
            double d = ray_triangle_intersection_check(triangle, point, ray);
            if (d > 0) {
              int already = 0;
              for (int j = 0; j < n_intersections; j++)
                if (triangles[j] == g->triangle_indices[i]) {
                  already = 1;
                  break;
                }
              if (!already) triangles[n_intersections++] = g->triangle_indices[i];
            }
          }
        }
    if (aborted) continue;
    if (intra ? (n_intersections % 2 == 1) : (n_intersections % 2 == 0))
      for (int i = 0; i < 3; i++) points[tid * 3 + i] = point[i];
  }
}

/* _fill_mesh host loop (simulations.py:521-579).  The mesh arrays passed in are
 * the ones the reference uploads (already stripped of the 12 wall triangles
 * for non-periodic substrates).  out must hold n_points*3 doubles.  Returns
 * the number of kernel launches used, or -1 if max_launches was hit. */
int dso_fill_mesh(int64_t n_points, const dso_geom *g, const double *voxel_size, int intra, uint64_t seed,
                  int64_t cuda_bs, double *out, int max_launches) {
  int64_t gs = (n_points + cuda_bs - 1) / cuda_bs;
  int64_t n_states = gs * cuda_bs;
  dso_state *st = (dso_state *)malloc(sizeof(dso_state) * (size_t)n_states);
  double *pts = (double *)malloc(sizeof(double) * 3 * (size_t)n_points);
  dso_rng_init_states(st, n_states, seed, 0);
  int64_t have = 0;
  int launches = 0;
  while (have < n_points) {
    if (launches >= max_launches) {
      launches = -1;
      break;
    }
    for (int64_t i = 0; i < n_points * 3; i++) pts[i] = INFINITY;
    fill_mesh_attempt(pts, n_points, st, intra, g, voxel_size);
    launches++;
    for (int64_t i = 0; i < n_points && have < n_points; i++)
      if (!isinf(pts[i * 3])) {
        memcpy(&out[have * 3], &pts[i * 3], 3 * sizeof(double));
        have++;
      }
  }
  free(st);
  free(pts);
  return launches;
}

/* ------------------------------------- unit wrappers for known answers --- */

double dso_u_dot(const double *a, const double *b) { return dot3(a, b); }
void dso_u_cross(const double *a, const double *b, double *c) { cross3(a, b, c); }
void dso_u_normalize(double *v) { normalize3(v); }
void dso_u_triangle_normal(const double *tri, double *n) { triangle_normal(tri, n); }
void dso_u_mat_mul(const double *R, double *v) { mat_mul(R, v); }
double dso_u_line_circle(const double *r0, const double *s, double r) { return line_circle_intersection(r0, s, r); }
double dso_u_line_sphere(const double *r0, const double *s, double r) { return line_sphere_intersection(r0, s, r); }
double dso_u_line_ellipsoid(const double *r0, const double *s, const double *ax) {
  return line_ellipsoid_intersection(r0, s, ax);
}
double dso_u_ray_triangle(const double *tri, const double *r0, const double *s) {
  return ray_triangle_intersection_check(tri, r0, s);
}
void dso_u_reflection(double *r0, double *step, double d, double *normal, double eps) {
  reflection(r0, step, d, normal, eps);
}
void dso_u_crossing(double *r0, const double *step, double d, double *normal, double eps) {
  crossing(r0, step, d, normal, eps);
}
void dso_u_random_steps(dso_state *st, int64_t n, double *steps) {
  for (int64_t i = 0; i < n; i++) random_step(&steps[i * 3], &st[i]);
}
void dso_u_subvoxel_range(const double *xs, int64_t n_xs, double x1, double x2, int periodic, int64_t *ll,
                          int64_t *ul) {
  if (periodic) {
    *ll = ll_subvoxel_overlap_periodic(xs, n_xs, x1, x2);
    *ul = ul_subvoxel_overlap_periodic(xs, n_xs, x1, x2);
  } else {
    *ll = ll_subvoxel_overlap(xs, n_xs, x1, x2);
    *ul = ul_subvoxel_overlap(xs, n_xs, x1, x2);
  }
}
uint64_t dso_u_next(dso_state *s) { return xoro_next(s); }
double dso_u_uniform_f64(dso_state *s) { return uniform_f64(s); }
float dso_u_logf(float x) { return DSO_LOGF(x); }
#ifdef DSO_SHARED_MATH
void dso_u_div3(double *v, double length) { ds_div3(&v[0], &v[1], &v[2], length); }
/* the product's integer form of `perm_prob < u` (dsim_math.h, ds_perm_threshold), handed out so that tests/test_dsim_math.py
 * can hold it against the reference's own comparison (simulations.py:988) */
int64_t dso_u_perm_threshold(double perm_prob) { return ds_perm_threshold(perm_prob); }
#endif
double dso_u_cos(double x) { return DSO_COS(x); }
size_t dso_sizeof_geom(void) { return sizeof(dso_geom); }

/* EXHAUSTIVE pins of the two table functions of the random step (tests/test_dsim_math.py), over every float32 with bit
 * pattern in [bits_lo, bits_hi) - a float32 uniform is one of the 4.4e8 floats of [2^-53, 1] (numba/cuda/random.py:199-221:
 * u1 feeds the log, 2 pi u2 the cosine).  The yardstick is the x87 long double function (64-bit significand) rounded once.
 *   dso_u_logf_sweep: how many inputs give something else than RN_float32(logl(x)); the first such input in *first_bad
 *   dso_u_cos_sweep:  max |cos(6.283185307179586 * u) - cosl(same argument)| over the range, and the largest such error
 *                     in units of the result's ulp where |cos| >= 1/2 */
int64_t dso_u_logf_sweep(uint32_t bits_lo, uint32_t bits_hi, uint32_t *first_bad) {
  int64_t bad = 0;
  uint32_t first = 0xffffffffu;
#pragma omp parallel for schedule(static) reduction(+ : bad) reduction(min : first)
  for (int64_t b = (int64_t)bits_lo; b < (int64_t)bits_hi; b++) {
    union {
      uint32_t u;
      float f;
    } in;
    in.u = (uint32_t)b;
    const float got = DSO_LOGF(in.f), want = (float)logl((long double)in.f);
    if (!(got == want)) {
      bad++;
      if ((uint32_t)b < first) first = (uint32_t)b;
    }
  }
  if (first_bad) *first_bad = first;
  return bad;
}

void dso_u_cos_sweep(uint32_t bits_lo, uint32_t bits_hi, double *max_abs_err, double *max_ulp_err_big) {
  double ea = 0, eu = 0;
#pragma omp parallel for schedule(static) reduction(max : ea) reduction(max : eu)
  for (int64_t b = (int64_t)bits_lo; b < (int64_t)bits_hi; b++) {
    union {
      uint32_t u;
      float f;
    } in;
    in.u = (uint32_t)b;
    const double x = 6.283185307179586 * (double)in.f;
    const double got = DSO_COS(x);
    const long double want = cosl((long double)x);
    const double err = (double)fabsl((long double)got - want);
    if (err > ea) ea = err;
    if (fabsl(want) >= 0.5L) {
      const double w = (double)want;
      const double ulp = nextafter(fabs(w), INFINITY) - fabs(w);
      if (err / ulp > eu) eu = err / ulp;
    }
  }
  *max_abs_err = ea;
  *max_ulp_err_big = eu;
}

/* The mesh kernel's reflect loop (simulations.py:924-997) on caller-supplied rays - the CPU twin of dsu_mesh_pass()
 * in disimpy_amd/csrc/dsim_unit.hip, for the adversarial traversal tests (rays through shared edges and vertices,
 * rays in subvoxel boundary planes, hits at d == step_l, far periodic images).  Walker i starts at r0[i] with
 * direction step[i], remaining length step_l[i] and RNG state st[i]; n_pass > 0: at most that many iterations of
 * the loop, nothing else (flags bit 0 = the loop is over); n_pass == 0: the whole loop, then the final advance
 * r0 += step * step_l (simulations.py:1001-1002; flags bit 1 = iter_exc).  Arrays are updated in place. */
void dso_u_mesh_pass(const dso_geom *g, int64_t n, int64_t n_pass, double *r0, double *step, double *step_l,
                     dso_state *st, int32_t *flags) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {
    int64_t iter_idx = 0;
    double sl = step_l[i];
    const int over = mesh_reflect_loop(&r0[i * 3], &step[i * 3], &sl, &st[i], g, &iter_idx, n_pass);
    int f = over ? 1 : 0;
    if (n_pass == 0) {
      if (iter_idx >= g->max_iter) f |= 2;
      for (int k = 0; k < 3; k++) r0[i * 3 + k] = fma(step[i * 3 + k], sl, r0[i * 3 + k]);
    }
    step_l[i] = sl;
    flags[i] = f;
  }
}

/* How adversarial is a ray?  For the FIRST iteration of the reflect loop: ties[i] = number of DISTINCT triangles that
 * attain the smallest positive d over the lists of all subvoxels in range (0 when that d is beyond step_l[i], i.e. no
 * hit), lists[i] = number of list entries visited, subs[i] = number of subvoxels in range.  Test statistics only. */
void dso_u_mesh_pass_stats(const dso_geom *g, int64_t n, const double *r0v, const double *stepv, const double *step_lv,
                           int32_t *ties, int32_t *lists, int32_t *subs) {
  const int64_t nx = g->n_xs - 1, ny = g->n_ys - 1, nz = g->n_zs - 1;
#pragma omp parallel for schedule(static)
  for (int64_t w = 0; w < n; w++) {
    const double *r0 = &r0v[w * 3], *step = &stepv[w * 3];
    const double step_l = step_lv[w];
    double triangle[9], temp_r0[3], sh[3];
    int64_t lls[3], uls[3], at_min[64];
    int n_at_min = 0, n_lists = 0, n_subs = 0;
    double min_d = INFINITY;
    const double *b3[3] = {g->xs, g->ys, g->zs};
    const int64_t nb[3] = {g->n_xs, g->n_ys, g->n_zs}, ns[3] = {nx, ny, nz};
    for (int k = 0; k < 3; k++) {
      lls[k] = ll_subvoxel_overlap_periodic(b3[k], nb[k], r0[k], fma(step[k], step_l, r0[k]));
      uls[k] = ul_subvoxel_overlap_periodic(b3[k], nb[k], r0[k], fma(step[k], step_l, r0[k]));
    }
    for (int64_t xi = lls[0]; xi < uls[0]; xi++)
      for (int64_t yi = lls[1]; yi < uls[1]; yi++)
        for (int64_t zi = lls[2]; zi < uls[2]; zi++) {
          int64_t c[3] = {xi, yi, zi};
          for (int k = 0; k < 3; k++) {
            sh[k] = 0;
            if (c[k] < 0 || c[k] > nb[k] - 2) {
              const int64_t sn = floor_div_as_float(c[k], ns[k]);
              c[k] -= sn * ns[k];
              sh[k] = (double)sn * b3[k][nb[k] - 1];
            }
            temp_r0[k] = r0[k] - sh[k];
          }
          n_subs++;
          const int64_t sv = c[0] * g->n_sv[1] * g->n_sv[2] + c[1] * g->n_sv[2] + c[2];
          for (int64_t i = g->subvoxel_indices[sv * 2]; i < g->subvoxel_indices[sv * 2 + 1]; i++) {
            n_lists++;
            get_triangle(g->triangle_indices[i], g->vertices, g->faces, triangle);
            const double d = ray_triangle_intersection_check(triangle, temp_r0, step);
            if (!(d > 0)) continue;
            if (d < min_d) {
              min_d = d;
              n_at_min = 0;
            }
            if (d == min_d) {
              int seen = 0;
              for (int q = 0; q < n_at_min; q++) seen |= at_min[q] == g->triangle_indices[i];
              if (!seen && n_at_min < 64) at_min[n_at_min++] = g->triangle_indices[i];
            }
          }
        }
    ties[w] = min_d > step_l ? 0 : n_at_min;
    lists[w] = n_lists;
    subs[w] = n_subs;
  }
}

/* ---- array form of the unit functions: the CPU side of tests/test_gpu_device_units.py ----
 * One call evaluates one device function of the reference on n independent inputs (column
 * arrays in[c][i], results out[c][i]), mirroring disimpy_amd/csrc/dsim_unit.hip op for op so
 * that the HIP device functions can be compared with this restatement bit for bit on millions
 * of inputs.  Integer results (subvoxel indices) and raw 64-bit patterns travel as doubles /
 * as the bit pattern of a double. */
static inline uint64_t bits_of(double x) {
  uint64_t b;
  memcpy(&b, &x, 8);
  return b;
}
static inline double double_of(uint64_t b) {
  double x;
  memcpy(&x, &b, 8);
  return x;
}

int dso_ua_eval(int op, int64_t n, const double *const *in, double *const *out, const double *xs, int64_t n_xs) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {
    switch (op) {
      case 0: out[0][i] = (double)DSO_LOGF((float)in[0][i]); break;
      case 1: out[0][i] = DSO_COS(in[0][i]); break;
      case 2: {
        double v[3] = {in[0][i], in[1][i], in[2][i]};
#ifdef DSO_SHARED_MATH
        ds_div3(&v[0], &v[1], &v[2], in[3][i]);
#else
        for (int k = 0; k < 3; k++) v[k] = v[k] / in[3][i];
#endif
        out[0][i] = v[0], out[1][i] = v[1], out[2][i] = v[2];
      } break;
      case 3: out[0][i] = sqrt(in[0][i]); break;
      case 4: out[0][i] = u64_to_unit_f64(bits_of(in[0][i])); break;
      case 5: { /* {A, E1, E2} pre-gathered as the library stores them: rebuild B, C is NOT possible exactly, so
                   the test passes the vertices and this side subtracts like simulations.py:259-262 */
        double tri[9], r0[3], st[3];
        for (int k = 0; k < 9; k++) tri[k] = in[k][i];
        for (int k = 0; k < 3; k++) r0[k] = in[9 + k][i], st[k] = in[12 + k][i];
        out[0][i] = ray_triangle_intersection_check(tri, r0, st);
      } break;
      case 6:
      case 7: {
        double r0[3], st[3], nn[3];
        for (int k = 0; k < 3; k++) r0[k] = in[k][i], st[k] = in[3 + k][i], nn[k] = in[7 + k][i];
        if (op == 6)
          reflection(r0, st, in[6][i], nn, in[10][i]);
        else
          crossing(r0, st, in[6][i], nn, in[10][i]);
        for (int k = 0; k < 3; k++) out[k][i] = r0[k], out[3 + k][i] = st[k], out[6 + k][i] = nn[k];
      } break;
      case 8:
        out[0][i] = (double)ll_subvoxel_overlap(xs, n_xs, in[0][i], in[1][i]);
        out[1][i] = (double)ul_subvoxel_overlap(xs, n_xs, in[0][i], in[1][i]);
        break;
      case 9:
        out[0][i] = (double)ll_subvoxel_overlap_periodic(xs, n_xs, in[0][i], in[1][i]);
        out[1][i] = (double)ul_subvoxel_overlap_periodic(xs, n_xs, in[0][i], in[1][i]);
        break;
      case 11: {
        double r0[2] = {in[0][i], in[1][i]}, st[2] = {in[2][i], in[3][i]};
        out[0][i] = line_circle_intersection(r0, st, in[4][i]);
      } break;
      case 12: {
        double r0[3] = {in[0][i], in[1][i], in[2][i]}, st[3] = {in[3][i], in[4][i], in[5][i]};
        out[0][i] = line_sphere_intersection(r0, st, in[6][i]);
      } break;
      case 13: {
        double r0[3] = {in[0][i], in[1][i], in[2][i]}, st[3] = {in[3][i], in[4][i], in[5][i]};
        double ax[3] = {in[6][i], in[7][i], in[8][i]};
        out[0][i] = line_ellipsoid_intersection(r0, st, ax);
      } break;
      case 14: {
        double v[3] = {in[0][i], in[1][i], in[2][i]};
        normalize3(v);
        out[0][i] = v[0], out[1][i] = v[1], out[2][i] = v[2];
      } break;
      case 15: {
        dso_state s = {bits_of(in[0][i]), bits_of(in[1][i])};
        double st[3];
        random_step(st, &s);
        out[0][i] = st[0], out[1][i] = st[1], out[2][i] = st[2];
        out[3][i] = double_of(s.s0), out[4][i] = double_of(s.s1);
      } break;
      case 16: {
        dso_state s = {bits_of(in[0][i]), bits_of(in[1][i])};
        out[0][i] = (double)uniform_f32(&s);
        out[1][i] = uniform_f64(&s);
        out[2][i] = double_of(s.s0), out[3][i] = double_of(s.s1);
      } break;
      case 17: {
        double R[9], v[3] = {in[9][i], in[10][i], in[11][i]};
        for (int k = 0; k < 9; k++) R[k] = in[k][i];
        mat_mul(R, v);
        out[0][i] = v[0], out[1][i] = v[1], out[2][i] = v[2];
      } break;
      case 18: {
        double a[3] = {in[0][i], in[1][i], in[2][i]}, b[3] = {in[3][i], in[4][i], in[5][i]}, c[3];
        cross3(a, b, c);
        out[0][i] = dot3(a, b);
        out[1][i] = c[0], out[2][i] = c[1], out[3][i] = c[2];
      } break;
      default: break;
    }
  }
  return (op >= 0 && op <= 18 && op != 10) ? 0 : -1;
}
