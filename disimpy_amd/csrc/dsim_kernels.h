/* dsim_kernels.h - the fused random-walk kernels (gfx950, wave64).
 *
 * One launch advances every walker over a whole range of time steps
 * [t_begin, t_end) (the reference launches one kernel per step and round-trips
 * positions, RNG state and the (M,N) phase matrix through HBM each time,
 * simulations.py:1198-1216).  Per walker, position + xoroshiro128+ state + MB
 * phase accumulators stay in registers for the whole range; HBM is touched
 * once at the start and once at the end (SoA, one 8-byte lane per wavefront
 * lane => fully coalesced).  LDS holds (a) the 1 KB log table of ds_logf and
 * the 4 KB cos/sin table of ds_cos_0_2pi,
 * (b) the subvoxel boundaries of the mesh index, (c) the current tile of the
 * gradient waveform, laid out [step][measurement][x y z pad] so that every lane
 * of a wavefront reads the same address (LDS broadcast, no bank conflict).
 */
#ifndef DSIM_KERNELS_H
#define DSIM_KERNELS_H

#include <limits.h>

#include "dsim_device.h"

/* Development build (-DDSIM_STATS): event counters for the mesh traversal.  STAT_LANE
 * counts once per lane, STAT_WAVE once per wavefront that reaches the point. */
#ifdef DSIM_STATS
__device__ unsigned long long dsim_stats[32];

#define STAT_LANE(i) atomicAdd(&dsim_stats[i], 1ULL)
#define STAT_WAVE(i)                                                                         \
  do {                                                                                       \
    if ((int)(__ffsll((long long)__ballot(1)) - 1) == (int)(threadIdx.x & 63)) atomicAdd(&dsim_stats[16 + (i)], 1ULL); \
  } while (0)
#else
#define STAT_LANE(i)
#define STAT_WAVE(i)
#endif

/* Development build (-DDSIM_TIMERS, tools/mesh_timers.py): shader-clock time per phase of the
 * cooperative traversal, accumulated per wavefront in LDS by lane 0 (T_MARK closes phase i). */
#ifdef DSIM_TIMERS
__device__ unsigned long long dsim_timers[16];
#define T_MARK(W, lane, i)                                        \
  do {                                                            \
    const unsigned long long t_ = __builtin_readcyclecounter();   \
    if ((lane) == 0) {                                            \
      (W)->tacc[i] += t_ - (W)->tlast;                            \
      (W)->tlast = t_;                                            \
    }                                                             \
  } while (0)
#else
#define T_MARK(W, lane, i)
#endif

/* Timing-ablation switch (-DDSIM_ABLATE, tools/ablate_coop.py; results become wrong, only the
 * timing is of interest): 1 = every subvoxel looks empty, 2 = cells are enumerated but nothing
 * is queued or scanned, 3 = items are culled but survivors are dropped, 4 = as 3 and every lane
 * reads the same four records (stage 2 without its gather traffic). */
#ifdef DSIM_ABLATE
__device__ int dsim_ablate = 0;
#define ABLATE(n) (dsim_ablate == (n))
#else
#define ABLATE(n) false
#endif

/* native vectors: a load of one is ONE instruction of that width (HIP's uint4 / float2 are structs of scalars that the
 * compiler takes apart and regroups as it sees fit - the 32-byte records came back as 12 + 16 (misaligned) + 4 bytes) */
typedef unsigned ds_uint4 __attribute__((ext_vector_type(4)));
typedef float ds_float2 __attribute__((ext_vector_type(2)));
typedef short ds_short2 __attribute__((ext_vector_type(2)));

enum { GEOM_FREE = 0, GEOM_SPHERE = 1, GEOM_CYLINDER = 2, GEOM_ELLIPSOID = 3, GEOM_MESH = 4 };

#define DS_TRI_STRIDE 12 /* doubles per triangle record */
#define DS_STATIC_LDS_BYTES ((DS_LOG_TABLE_DOUBLES + DS_COS_TABLE_DOUBLES) * 8) /* walk_body's static LDS (the tables of the random step) */

struct MeshDev {
  const double *tri;       /* [n_faces][12] {A, E1 = B - A, E2 = C - A, unit normal}: the normal is
                              normalize(E1 x E2) evaluated once on the host with the kernel's own operations
                              (same fma convention, IEEE sqrt, ds_div3) instead of after every hit
                              (simulations.py:989-990 recomputes it per reflection) */
  const int2 *sv_range;    /* [prod n_sv]   [start, end) into tri_idx    */
  const int32_t *tri_idx;  /* [n_refs]                                   */
  /* fine grid (dsim_mesh_host.h): every non-empty subvoxel is cut into k^3 cells, k = 1 << lk for the whole mesh */
  const int32_t *sv_base;     /* [prod n_sv]  first cell of the subvoxel in cell_start, -1 = empty (lk == 0: unused,
                                 cell_start is indexed by the subvoxel itself) */
  const uint32_t *cell_start; /* [n_cells + 1] the records of cells c .. c+r-1 (a run along z inside one subvoxel) are
                                 mrec[cell_start[c] .. cell_start[c + r]) */
  /* [r5] lk > 0: the same table with a slot block for EVERY subvoxel, empty ones included (slot (s << (3 lk + 2)) + cell;
   * an empty subvoxel's slots all hold the record count before it): the cooperative traversal finds a run's records
   * with ONE level of look-ups instead of sv_base -> cell_start.  Built on the device from the two compact arrays
   * (expand_cells_kernel); nullptr when the table would exceed DSIM_DENSE_CELLS_MB (the two-level path then) */
  const uint32_t *cell_dense;
  /* [r6] 16-byte records, one load each, and the triangle's plane by a second load from a table indexed by the triangle
   * (round 5: 32-byte records that carried the plane; rounds 2-4: 48 bytes, float32 boxes):
   *   mrec[r]   = {lo_x | lo_y << 16,  lo_z | nhi_x << 16,  nhi_y | nhi_z << 16,  triangle}
   *   planes[t] = {nx, ny, nz, c}   float32 supporting plane n.x = c of triangle t - the same in every cell that lists it
   * (the neuron refined to 50 000 triangles: 1.28 M records = 20 MB + 0.8 MB of planes instead of 41 MB; a wavefront's walkers
   * share their triangles, so the planes come out of L1 / L2).
   * The box - of the triangle's part inside the cell - in 15-bit fixed point: coordinate x -> x * qs with ONE scale for the
   * three axes, qs = 32767 / (the largest of xs[nx], ys[ny], zs[nz]) - the reference's grids start at 0
   * (substrates.py:506-508), so the voxel maps into [0, 32767]; lo rounded down, hi rounded up and stored as
   * nhi = 32767 - hi, all clamped into [0, 32767] (a grid handed in through the C ABI that lies elsewhere clamps and is
   * culled by the plane test alone - slower, never wrong).  A segment's box comes in the same form,
   * {hi_x | hi_y << 16, hi_z | nlo_x << 16, nlo_y | nlo_z << 16} with nlo = 32767 - lo (quantise_box): the boxes overlap
   * iff no 16-bit half of the three packed differences box - record is negative (three v_pk_sub_i16, one v_or3, one
   * v_and).  Rounding outwards on both sides and clamping are monotone, so an overlap of the float32 boxes is never lost:
   * what passes is a superset of what passed the float32 comparison, looser by up to 2 quanta (6.8 nm on the neuron's
   * 223 um voxel, cells of 0.5 um) per side */
  const ds_uint4 *mrec;
  const ds_uint4 *planes;
  float qs;
  int lk;
  const double *bounds;    /* xs | ys | zs, (nx+1)+(ny+1)+(nz+1) doubles  */
  /* clearance map (dsim_hip.hip, build_clearance_map): every subvoxel is cut into F^3 fine cells, F = 1 << clear_lf;
   * a set bit = some triangle (of any periodic image) comes within one step length (+ margins) of some point of
   * that cell.  clear_idx[sv] = -1: no bit set in the whole subvoxel, else the subvoxel's block of
   * max(1, F^3 / 32) dwords in clear_bits.  nullptr = no map (every walker takes the full pass). */
  const int32_t *clear_idx;
  const uint32_t *clear_bits;
  int clear_lf;
  int nx, ny, nz;
  float inv_nx, inv_ny, inv_nz;  /* [r6] 1 / n as float32: the periodic image of a subvoxel coordinate by one multiplication (wrap_subvoxel) */
  float pad_;
  double inv_hx, inv_hy, inv_hz; /* n / voxel_size: first guess only      */
  double inv_lx, inv_ly, inv_lz; /* 1 / voxel_size: first guess only      */
  double x0, y0, z0, x1, y1, z1; /* xs[0], ys[0], zs[0], xs[nx], ys[ny], zs[nz] */
};

struct WalkParams {
  int64_t n;
  const double *x_in, *y_in, *z_in;
  const uint64_t *s0_in, *s1_in;
  double *x_out, *y_out, *z_out;
  uint64_t *s0_out, *s1_out;
  double *phase;      /* (M, phase_stride): row m starts at phase + m * phase_stride */
  int64_t phase_stride; /* = n, except when this launch covers a sub-range of the handle's walkers */
  uint8_t *exc;       /* (n,)   */
  const double *g4;   /* [T][M][4] = gx, gy, gz, 0 */
  int M, m0, mc;      /* this pass handles measurements [m0, m0+mc), mc <= MB */
  int t_begin, t_end, tile_steps;
  int write_state;    /* last pass writes positions / RNG state / exc */
  double step_l, gdt; /* gdt = GAMMA * dt (gradients.py:13) */
  double radius;
  double ax[3];
  double R[9], Rinv[9];
  int64_t max_iter;
  double eps, perm_prob;
  int64_t perm_thresh; /* ds_perm_threshold(perm_prob): the cooperative mesh kernel decides reflection / crossing on the draw's integer bits */
  MeshDev mesh;
  double *traj; /* walk_kernel_coop4_traj: [t - t_begin][3][n], the positions after every time step of the launch */
  /* > 0: WALKER-major instead, [slot][traj_row_steps][x y z 0]: a walker's step is ONE 32-byte store into its own row.
   * Runs whose phases come from recorded positions (more than four measurements on a mesh) use it: with per-lane
   * clocks the lanes of a wavefront sit at different steps, three 8-byte stores per lane into three rows of the
   * step-major layout cost the walk kernel a fifth of its rate (1.19e10 with them, 1.51e10 without, 1.35e10 this
   * way: profiles/history/r04_ab_m32_recorded_positions_layout.txt) */
  int traj_row_steps;
  /* recording for the caller (dsim_set_recording) while the walkers sit in sorted slots: slot i records into column
   * rec_col[i] - its walker's index - so that the recording comes out in walker order; nullptr: column i */
  const int32_t *rec_col;
  /* cooperative mesh kernel with per-walker clocks (dsim_hip.hip, run_own_clock): tcur[i] = the next time step of walker i
   * on entry and on exit; the launch ends when every wavefront has run iter_budget passes (walkers inside a step
   * finish it) or all its walkers have reached t_end.  nullptr: every walker runs [t_begin, t_end). */
  int32_t *tcur;
  int iter_budget;
  /* with tcur: {walkers not yet at t_end, time steps they still have to take, the most any of them has left}, added up
   * by the kernel itself at the end of the launch (one atomic each per wavefront); nullptr: not asked */
  unsigned long long *progress;
  int xcd_run_log2; /* mesh: an XCD takes runs of 1 << xcd_run_log2 consecutive walker groups (the grid is a multiple of 8 such runs) */
};

/* ---- one walker, one time step, by substrate -------------------------------- */

/* simulations.py:705-745 */
DSD void step_sphere(V3 &r0, V3 &step, double step_l, const WalkParams &p, bool &exc) {
  int64_t iter_idx = 0;
  bool check = true;
  while (check && step_l > 0 && iter_idx < p.max_iter) {
    iter_idx++;
    double d = line_sphere_intersection(r0, step, p.radius);
    if (d > 0 && d < step_l) {
      V3 n = {-__builtin_fma(d, step.x, r0.x), -__builtin_fma(d, step.y, r0.y), -__builtin_fma(d, step.z, r0.z)};
      normalize3(n);
      reflection(r0, step, d, n, p.eps);
      step_l -= d + p.eps;
    } else {
      check = false;
    }
  }
  if (iter_idx >= p.max_iter) exc = true;
  r0.x = __builtin_fma(step.x, step_l, r0.x);
  r0.y = __builtin_fma(step.y, step_l, r0.y);
  r0.z = __builtin_fma(step.z, step_l, r0.z);
}

/* simulations.py:759-805 */
DSD void step_cylinder(V3 &r0, V3 &step, double step_l, const WalkParams &p, bool &exc) {
  r0 = mat_mul(p.R, r0);
  int64_t iter_idx = 0;
  bool check = true;
  while (check && step_l > 0 && iter_idx < p.max_iter) {
    iter_idx++;
    double d = line_circle_intersection(r0.y, r0.z, step.y, step.z, p.radius);
    if (d > 0 && d < step_l) {
      V3 n = {0.0, -__builtin_fma(d, step.y, r0.y), -__builtin_fma(d, step.z, r0.z)};
      normalize3(n);
      reflection(r0, step, d, n, p.eps);
      step_l -= d + p.eps;
    } else {
      check = false;
    }
  }
  if (iter_idx >= p.max_iter) exc = true;
  step = mat_mul(p.Rinv, step);
  r0 = mat_mul(p.Rinv, r0);
  r0.x = __builtin_fma(step.x, step_l, r0.x);
  r0.y = __builtin_fma(step.y, step_l, r0.y);
  r0.z = __builtin_fma(step.z, step_l, r0.z);
}

/* simulations.py:819-864 */
DSD void step_ellipsoid(V3 &r0, V3 &step, double step_l, const WalkParams &p, bool &exc) {
  r0 = mat_mul(p.R, r0);
  int64_t iter_idx = 0;
  bool check = true;
  while (check && step_l > 0 && iter_idx < p.max_iter) {
    iter_idx++;
    double d = line_ellipsoid_intersection(r0, step, p.ax[0], p.ax[1], p.ax[2]);
    if (d > 0 && d < step_l) {
      V3 n = {-__builtin_fma(d, step.x, r0.x) / (p.ax[0] * p.ax[0]), -__builtin_fma(d, step.y, r0.y) / (p.ax[1] * p.ax[1]),
              -__builtin_fma(d, step.z, r0.z) / (p.ax[2] * p.ax[2])};
      normalize3(n);
      reflection(r0, step, d, n, p.eps);
      step_l -= d + p.eps;
    } else {
      check = false;
    }
  }
  if (iter_idx >= p.max_iter) exc = true;
  step = mat_mul(p.Rinv, step);
  r0 = mat_mul(p.Rinv, r0);
  r0.x = __builtin_fma(step.x, step_l, r0.x);
  r0.y = __builtin_fma(step.y, step_l, r0.y);
  r0.z = __builtin_fma(step.z, step_l, r0.z);
}

DSD V3 load_v3(const double *p) {
  V3 v = {p[0], p[1], p[2]};
  return v;
}

/* Conservative cull box of the remaining segment r0 -> r0 + step*step_l, in
 * float32.  A candidate triangle whose (outward-rounded) float32 AABB does not
 * meet this box cannot return 0 < d <= step_l from the Moller-Trumbore test, so
 * skipping it changes neither the decision `min_d > step_l` nor the closest
 * triangle when there is a hit (simulations.py:981-986): results stay
 * bit-identical to testing every candidate.
 *
 * The box is DERIVED from seven numbers - the segment's midpoint m and half vector h rounded to
 * float32 and the margin tau = 1e-6 step_l + 2e-6 of the largest coordinate magnitude (16 float32
 * ulps of it): the bounds are m -+ (|h| + tau).  That margin covers the half-ulp roundings of m and h
 * many times over and dwarfs every fp64 rounding error of the test itself (~1e-16 relative); it is
 * also the plane-side margin (the float32 evaluation of n.m - c carries an absolute error below
 * ~8 ulp of the largest coordinate).  Only the seven numbers travel between lanes. */
struct CullBox {
  float lox, loy, loz, hix, hiy, hiz;
  ds_float2 mhx, mhy, mhz;      /* {midpoint, half vector} of the segment per axis, float32 (pairs: the plane test runs packed) */
  float tau;                    /* box and plane-side margin */
};
struct QBox { /* the box in the records' fixed point (quantise_box) */
  unsigned b0, b1, b2;
};

DSD void box_from_segment(CullBox &c) {
  const float ex = __builtin_fabsf(c.mhx.y) + c.tau, ey = __builtin_fabsf(c.mhy.y) + c.tau, ez = __builtin_fabsf(c.mhz.y) + c.tau;
  c.lox = c.mhx.x - ex, c.hix = c.mhx.x + ex;
  c.loy = c.mhy.x - ey, c.hiy = c.mhy.x + ey;
  c.loz = c.mhz.x - ez, c.hiz = c.mhz.x + ez;
}

DSD unsigned pk_i16(int a, int b) { /* both saturated into [0, 32767] and packed a | b << 16 */
  const ds_short2 p = __builtin_amdgcn_cvt_pk_i16(a, b); /* v_cvt_pk_i16_i32: signed saturation */
  const ds_short2 z = {0, 0};
  const ds_short2 m = __builtin_elementwise_max(p, z);   /* v_pk_max_i16 */
  return __builtin_bit_cast(unsigned, m);
}

/* the float32 box of c in the records' fixed point (MeshDev::mrec): hi = ceil, lo = floor of x * qs, both by
 * truncation of a value pushed up by 1.001 quanta (the float -> int conversion truncates, saturates and sends NaN to 0;
 * a result that is one quantum too wide only keeps a record the exact box would have dropped).  18 VALU operations. */
DSD QBox quantise_box(const CullBox &c, const MeshDev &ms) {
  const float qs = ms.qs, up = 1.001f, top = 32767.0f + 1.001f;
  const int hx = (int)__builtin_fmaf(c.hix, qs, up), hy = (int)__builtin_fmaf(c.hiy, qs, up), hz = (int)__builtin_fmaf(c.hiz, qs, up);
  const int lx = (int)__builtin_fmaf(-c.lox, qs, top), ly = (int)__builtin_fmaf(-c.loy, qs, top), lz = (int)__builtin_fmaf(-c.loz, qs, top);
  QBox q;
  q.b0 = pk_i16(hx, hy);
  q.b1 = pk_i16(hz, lx);
  q.b2 = pk_i16(ly, lz);
  return q;
}

/* Can the record be skipped?  No when (a) no 16-bit half of the three packed differences box - record is negative (the
 * fixed-point boxes overlap) and (b) the segment does not lie wholly on one side of the triangle's plane: with the midpoint
 * m and the half vector h of the segment the end points evaluate to (n.m - c) -+ n.h, so both are beyond the margin tau on
 * the same side iff |n.m - c| - |n.h| > tau.  mh_x = {m.x, h.x} etc.: the two dot products run as three packed
 * operations.  Every rounding / flush-to-zero / NaN case falls on the "keep" side or concerns a walker whose fp64 test
 * cannot hit anyway (a NaN box quantises to an empty one).  record_tests hands out the two comparisons separately: the
 * cooperative traversal takes the wave's mask of each (a comparison IS its mask) and combines them on the scalar unit. */
DSD void record_tests(const ds_uint4 &q0, const ds_uint4 &q1, unsigned b0, unsigned b1, unsigned b2, const ds_float2 &mh_x,
                      const ds_float2 &mh_y, const ds_float2 &mh_z, float tau, bool &box_ok, bool &plane_ok) {
  /* (elements copied out first: __builtin_bit_cast applied to a vector ELEMENT, `q0.y`, reads element 0 with this compiler) */
  const unsigned rx = q0.x, ry = q0.y, rz = q0.z;
  const ds_short2 d0 = __builtin_bit_cast(ds_short2, b0) - __builtin_bit_cast(ds_short2, rx);
  const ds_short2 d1 = __builtin_bit_cast(ds_short2, b1) - __builtin_bit_cast(ds_short2, ry);
  const ds_short2 d2 = __builtin_bit_cast(ds_short2, b2) - __builtin_bit_cast(ds_short2, rz);
  const unsigned neg = (__builtin_bit_cast(unsigned, d0) | __builtin_bit_cast(unsigned, d1) | __builtin_bit_cast(unsigned, d2)) & 0x80008000u;
  const float nx = __uint_as_float(q1.x), ny = __uint_as_float(q1.y), nz = __uint_as_float(q1.z), pc = __uint_as_float(q1.w);
  ds_float2 t = (ds_float2){ny, ny} * mh_y;
  t = __builtin_elementwise_fma((ds_float2){nx, nx}, mh_x, t);
  t = __builtin_elementwise_fma((ds_float2){nz, nz}, mh_z, t); /* {n.m, n.h} */
  const float side = __builtin_fabsf(t.x - pc) - __builtin_fabsf(t.y);
  box_ok = neg == 0u;
  plane_ok = side <= tau;
}
DSD bool record_survives(const ds_uint4 &q0, const ds_uint4 &q1, unsigned b0, unsigned b1, unsigned b2, const ds_float2 &mh_x,
                         const ds_float2 &mh_y, const ds_float2 &mh_z, float tau) {
  bool box_ok, plane_ok;
  record_tests(q0, q1, b0, b1, b2, mh_x, mh_y, mh_z, tau, box_ok, plane_ok);
  return box_ok && plane_ok;
}
DSD CullBox make_cull_box(const V3 &a, const V3 &step, double step_l) {
  /* float32 arithmetic from the start: start point and direction rounded once (half an ulp each), the midpoint one
   * more rounding - about one ulp of the largest coordinate in all, against a margin of sixteen */
  const float ax = (float)a.x, ay = (float)a.y, az = (float)a.z;
  const float sx = (float)step.x, sy = (float)step.y, sz = (float)step.z;
  const float sl = (float)step_l, hl = 0.5f * sl;
  CullBox c;
  c.mhx.y = sx * hl;
  c.mhy.y = sy * hl;
  c.mhz.y = sz * hl;
  c.mhx.x = ax + c.mhx.y;
  c.mhy.x = ay + c.mhy.y;
  c.mhz.x = az + c.mhz.y;
  /* magnitude of the largest coordinate of either end point: |a| + step_l bounds both (|step| = 1) */
  const float big = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(ax), __builtin_fabsf(ay)), __builtin_fabsf(az)) + sl;
  c.tau = __builtin_fmaf(2.0001e-6f, big, 1.0001e-6f * sl);
  box_from_segment(c);
  return c;
}

struct SubvoxelWalk { /* the subvoxel ranges of one reflect-loop iteration (simulations.py:929-934) */
  int llx, lly, llz, ulx, uly, ulz;
};

struct BestHit {
  double d;
  int tri;
};

/* The launch parameters, read afresh.  A kernel argument struct of ~100 dwords that is read through its by-value
 * parameter is fetched into SGPRs once and kept there for the whole kernel; the cooperative mesh kernel, whose pass
 * loop needs most of the scalar file for itself, then spills them through VGPR lanes (v_writelane / v_readlane are
 * VALU instructions in a VALU-bound kernel: round 2 shipped with 124 spilled SGPRs).  With KA = 1 the kernel instead
 * re-reads what a stage needs from the kernarg segment (scalar loads out of the constant cache) through a pointer the
 * compiler cannot see through, so no value lives longer than the stage that uses it.  KA = 0 (every other kernel, and
 * the test wrappers): the by-value parameter as it is.  WalkParams must be the kernel's first and only argument and
 * trivially copyable (asserted here; dsu_kernarg_check in the unit library compares the segment with the parameter
 * dword by dword on the device: tests/test_gpu_device_units.py). */
static_assert(__is_trivially_copyable(WalkParams), "fresh_params reads WalkParams straight out of the kernarg segment");
template <int KA>
DSD const WalkParams &fresh_params(const WalkParams &p) {
  if (KA) {
    auto k = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return *(const WalkParams *)k;
  }
  return p;
}

/* One iteration of the reflect loop exactly as the reference does it (simulations.py:936-983): every triangle of
 * every subvoxel in range, in the reference's order, no culling, the first smallest distance wins.  The fall-back
 * of both traversals - for a walker whose work does not fit the wavefront's queues, and whenever two different
 * triangles tie for the smallest distance, where only the reference's visiting order decides.  Kept small in
 * registers: the occupancy of a kernel is set by its largest code path. */
DSD void iteration_serial(const MeshDev &ms, const SubvoxelWalk &w, const V3 &r0, const V3 &step, const double *xs,
                          const double *ys, const double *zs, BestHit &best) {
  const int nx = ms.nx, ny = ms.ny, nz = ms.nz;
  const double Lx = xs[nx], Ly = ys[ny], Lz = zs[nz];
  for (int xi = w.llx; xi < w.ulx; xi++) {
    int x = xi, snx = 0;
    if (x < 0 || x > nx - 1) {
      snx = floor_div(x, nx);
      x -= snx * nx;
    }
    for (int yi = w.lly; yi < w.uly; yi++) {
      int y = yi, sny = 0;
      if (y < 0 || y > ny - 1) {
        sny = floor_div(y, ny);
        y -= sny * ny;
      }
      for (int zi = w.llz; zi < w.ulz; zi++) {
        int z = zi, snz = 0;
        if (z < 0 || z > nz - 1) {
          snz = floor_div(z, nz);
          z -= snz * nz;
        }
        const int2 rg = ms.sv_range[(x * ny + y) * nz + z];
        if (rg.y <= rg.x) continue;
        STAT_LANE(8); /* subvoxel scanned serially by its owner */
        STAT_WAVE(8);
        const V3 tr0 = {r0.x - (double)snx * Lx, r0.y - (double)sny * Ly, r0.z - (double)snz * Lz};
        for (int i = rg.x; i < rg.y; i++) {
          const int ti = ms.tri_idx[i];
          const double *rec = ms.tri + (int64_t)ti * DS_TRI_STRIDE;
          const double d = ray_triangle_intersection_check(load_v3(rec), load_v3(rec + 3), load_v3(rec + 6), tr0, step);
          if (d > 0 && d < best.d) {
            best.d = d;
            best.tri = ti;
          }
        }
      }
    }
  }
}

/* Candidate scan of a run of cells: 4 records (128 bytes) are requested before the first is
 * examined, so that the L2 round trips overlap instead of queueing.  The cells are not visited in
 * the reference's order, so a tie - two DIFFERENT triangles at the same smallest distance, of which
 * the reference keeps the one its loops meet first (strict <, simulations.py:981) - is only
 * detected here (`tie`); the caller then redoes the pass in the reference's order.  The same
 * triangle met again (another cell, another subvoxel of the same image) returns the same d. */
DSD void scan_candidates(const MeshDev &ms, int begin, int end, const CullBox &cb, const QBox &qb, const V3 &tr0, const V3 &step,
                         double &min_d, int &closest, bool &tie) {
  const ds_float2 mh_x = cb.mhx, mh_y = cb.mhy, mh_z = cb.mhz;
  const float tau = cb.tau;
  const unsigned b0 = qb.b0, b1 = qb.b1, b2 = qb.b2;
  for (int i = begin; i < end; i += 4) {
    STAT_LANE(4);
    STAT_WAVE(4);
    ds_uint4 q0[4], q1[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      int j = min(i + u, end - 1);
      q0[u] = ms.mrec[j];
      q1[u] = ms.planes[q0[u].w];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (i + u >= end) break;
      STAT_LANE(5);
      /* (1) box of the triangle's part inside the cell vs cull box of the segment, (2) both end points strictly on one
       * side of the triangle's plane (margin tau): the ray cannot meet the plane at 0 < d <= step_l.  Degenerate
       * triangles have n = 0. */
      if (!record_survives(q0[u], q1[u], b0, b1, b2, mh_x, mh_y, mh_z, tau)) continue;
      STAT_LANE(6);
      STAT_WAVE(6);
      int ti = (int)q0[u].w;
      const double *rec = ms.tri + (int64_t)ti * DS_TRI_STRIDE;
      double d = ray_triangle_intersection_check(load_v3(rec), load_v3(rec + 3), load_v3(rec + 6), tr0, step);
      if (d > 0) {
        if (d < min_d) {
          closest = ti;
          min_d = d;
          tie = false;
        } else if (d == min_d && ti != closest) {
          tie = true;
        }
      }
    }
  }
}

/* simulations.py:878-1002.  xs/ys/zs: LDS copies of the subvoxel boundaries. */
template <int KA>
DSD void step_mesh(V3 &r0, V3 &step, double step_l, const WalkParams &p_arg, const double *xs, const double *ys,
                   const double *zs, Rng &rng, bool &exc) {
  int64_t iter_idx = 0;
  bool check = true;
  int closest = 0;
  while (check && step_l > 0 && iter_idx < fresh_params<KA>(p_arg).max_iter) {
    /* KA = 1: the launch parameters are re-read per pass (and once more per subvoxel below): none of the ~40 dwords of
     * MeshDev occupies an SGPR for the whole kernel */
    const WalkParams &p = fresh_params<KA>(p_arg);
    const MeshDev &ms = p.mesh;
    const int nx = ms.nx, ny = ms.ny, nz = ms.nz;
    const double Lx = xs[nx], Ly = ys[ny], Lz = zs[nz];
    iter_idx++;
    STAT_LANE(0);
    STAT_WAVE(0);
    double min_d = DS_INF;
    bool tie = false;
    double ex = __builtin_fma(step.x, step_l, r0.x);
    double ey = __builtin_fma(step.y, step_l, r0.y);
    double ez = __builtin_fma(step.z, step_l, r0.z);
    int llx = ll_subvoxel_overlap_periodic(xs, nx, ms.inv_hx, ms.inv_lx, r0.x, ex);
    int lly = ll_subvoxel_overlap_periodic(ys, ny, ms.inv_hy, ms.inv_ly, r0.y, ey);
    int llz = ll_subvoxel_overlap_periodic(zs, nz, ms.inv_hz, ms.inv_lz, r0.z, ez);
    int ulx = ul_subvoxel_overlap_periodic(xs, nx, ms.inv_hx, ms.inv_lx, r0.x, ex);
    int uly = ul_subvoxel_overlap_periodic(ys, ny, ms.inv_hy, ms.inv_ly, r0.y, ey);
    int ulz = ul_subvoxel_overlap_periodic(zs, nz, ms.inv_hz, ms.inv_lz, r0.z, ez);
    /* the subvoxels of the range in the reference's order (x outermost, z innermost) as ONE loop with a carrying
     * counter, the columns of cells of a subvoxel likewise: every level of nested divergent loops costs a saved
     * execution mask in the scalar registers, and eight levels of them spilled */
    const bool some = ulx > llx && uly > lly && ulz > llz;
    for (int xi = llx, yi = lly, zi = llz; some && xi < ulx;) {
      const MeshDev &ms = fresh_params<KA>(p_arg).mesh;
      const int nx = ms.nx, ny = ms.ny, nz = ms.nz;
      int x = xi, y = yi, z = zi;
      double shx = 0, shy = 0, shz = 0;
      if (x < 0 || x > nx - 1) {
        int sn = floor_div(x, nx);
        x -= sn * nx;
        shx = (double)sn * Lx;
      }
      if (y < 0 || y > ny - 1) {
        int sn = floor_div(y, ny);
        y -= sn * ny;
        shy = (double)sn * Ly;
      }
      if (z < 0 || z > nz - 1) {
        int sn = floor_div(z, nz);
        z -= sn * nz;
        shz = (double)sn * Lz;
      }
      const int sv = (x * ny + y) * nz + z;
      STAT_LANE(1);
      STAT_WAVE(1);
      const int lk = ms.lk;
      int base = sv;
      if (lk > 0) base = ms.sv_base[sv];
      if (base >= 0) { /* not an empty subvoxel */
        STAT_LANE(2);
        STAT_WAVE(2);
        V3 tr0 = {r0.x - shx, r0.y - shy, r0.z - shz};
        const CullBox cb = make_cull_box(tr0, step, step_l);
        const QBox qb = quantise_box(cb, ms);
        if (lk == 0) {
          STAT_LANE(3);
          STAT_WAVE(3);
          scan_candidates(ms, (int)ms.cell_start[base], (int)ms.cell_start[base + 1], cb, qb, tr0, step, min_d, closest, tie);
        } else {
          /* x, y: 2k half-cell units per subvoxel, a cell of two units at every unit but the last (dsim_mesh_host.h):
           * the units u0 .. u1 of the box are covered by the cells at u0, u0 + 2, ... */
          const int k = 1 << lk, k2 = 2 * k;
          const double sx = (double)k2 * ms.inv_hx, sy = (double)k2 * ms.inv_hy, sz = (double)k * ms.inv_hz;
          const int a0 = micro_cell((double)cb.lox, xs[x], sx, k2), a1 = micro_cell((double)cb.hix, xs[x], sx, k2);
          const int b0 = micro_cell((double)cb.loy, ys[y], sy, k2), b1 = micro_cell((double)cb.hiy, ys[y], sy, k2);
          const int c0 = micro_cell((double)cb.loz, zs[z], sz, k), c1 = micro_cell((double)cb.hiz, zs[z], sz, k);
          for (int a = a0, b = b0;;) {
            STAT_LANE(3);
            STAT_WAVE(3);
            const int c = base + (min(a, k2 - 2) * k2 + min(b, k2 - 2)) * k; /* the cells c0 .. c1 of this column: one run of records */
            scan_candidates(ms, (int)ms.cell_start[c + c0], (int)ms.cell_start[c + c1 + 1], cb, qb, tr0, step, min_d, closest, tie);
            if (b + 1 < b1) {
              b += 2;
            } else if (a + 1 < a1) {
              a += 2, b = b0;
            } else {
              break;
            }
          }
        }
      }
      if (++zi == ulz) {
        zi = llz;
        if (++yi == uly) yi = lly, xi++;
      }
    }
    if (tie && min_d <= step_l) {
      /* two different triangles at the same smallest distance: the reference keeps the one its loops meet first
       * (simulations.py:981) - redo this pass in the reference's own order */
      BestHit best = {DS_INF, 0};
      SubvoxelWalk sw = {llx, lly, llz, ulx, uly, ulz};
      iteration_serial(ms, sw, r0, step, xs, ys, zs, best);
      min_d = best.d;
      closest = best.tri;
    }
    if (min_d > step_l) {
      check = false;
    } else {
      STAT_LANE(7);
      STAT_WAVE(7);
      bool reflect = p.perm_prob < rng_uniform_f64(rng);
      const double *rec = ms.tri + (int64_t)closest * DS_TRI_STRIDE;
      V3 n = load_v3(rec + 9); /* normalize((A-B) x (A-C)), precomputed: (A-B)x(A-C) == E1 x E2 bit for bit */
      if (reflect)
        reflection(r0, step, min_d, n, p.eps);
      else
        crossing(r0, step, min_d, n, p.eps);
      step_l -= min_d;
    }
  }
  if (iter_idx >= fresh_params<KA>(p_arg).max_iter) exc = true;
  r0.x = __builtin_fma(step.x, step_l, r0.x);
  r0.y = __builtin_fma(step.y, step_l, r0.y);
  r0.z = __builtin_fma(step.z, step_l, r0.z);
}

/* the dense cell table of MeshDev::cell_dense from the compact one: entry (s << lcs) + c = cell_start[sv_base[s] + c], or - for
 * an empty subvoxel - recs_before[s], the number of records of the subvoxels before s (= the first record of the next
 * non-empty one: the table stays a CSR over all slots).  n_slots = n_sub << lcs; the last entry (the end of all records) too */
__global__ void expand_cells_kernel(const int32_t *sv_base, const uint32_t *cell_start, const uint32_t *recs_before, int lcs,
                                    uint64_t n_slots, uint32_t *dense) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n_slots) return;
  const uint64_t s = i >> lcs;
  if (i == n_slots) {
    dense[i] = recs_before[s];
    return;
  }
  const int32_t base = sv_base[s];
  dense[i] = base >= 0 ? cell_start[(uint32_t)base + (uint32_t)(i & ((1ull << lcs) - 1ull))] : recs_before[s];
}

#include "dsim_mesh_coop.h"

/* lane J of every row of 16 lanes, broadcast to its row - as an operand modifier of the two fp64 VALU operations
 * that accept one on gfx90a and later (DPP row_newbcast on v_mov_b64 and v_fmac_f64).  Every lane of the wavefront has to be
 * active, and the broadcast register must not have been written by a VALU instruction two issue slots earlier (the
 * callers load it from memory and fence with dpp_fence). */
template <int J>
DSD double row_bcast(double v) {
  double r;
  asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(J));
  return r;
}
/* acc = fma(lane J of g in this row, v, acc) */
template <int J>
DSD void row_fmac(double &acc, double g, double v) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(g), "v"(v), "n"(J));
}
DSD void dpp_fence(double &a, double &b, double &c) { asm volatile("s_nop 1" : "+v"(a), "+v"(b), "+v"(c)); }

/* measurements M0 .. M0 + 15 of one time step: gx, gy, gz hold, in lane j of every row, the gradient sample of
 * measurement M0 + j; the same operations in the same order as the fused kernels (walk_body):
 * inner = fma(gz, z, fma(gx, x, gy * y)), phase = fma(gdt, inner, phase).
 * [r6] Four measurements at a time, their chains interleaved: a DPP operation must not read a register a vector instruction
 * wrote within the last two issue slots, and with ONE chain every second instruction waited for that (three s_nop per
 * measurement in the shipped code of rounds 4-5); and gy * y as fma(gy, y, +0) on the DPP form itself instead of a
 * broadcast move followed by a multiplication.  The fma returns the product's own rounding; only an exact zero product
 * can come out as +0 where the multiplication gives -0, and a zero of either sign leaves the next fma's result - and in
 * the end the phase, which is never -0 (it starts at +0 and x + (-x) = +0) - bit for bit the same. */
template <int M0, int J, int MB>
DSD void phase_step16(double (&ph)[MB], double gx, double gy, double gz, double x, double y, double z, double gdt) {
  if constexpr (J < 16 && M0 + J < MB && MB % 4 == 0) {
    double i0 = 0.0, i1 = 0.0, i2 = 0.0, i3 = 0.0;
    row_fmac<J>(i0, gy, y), row_fmac<J + 1>(i1, gy, y), row_fmac<J + 2>(i2, gy, y), row_fmac<J + 3>(i3, gy, y);
    row_fmac<J>(i0, gx, x), row_fmac<J + 1>(i1, gx, x), row_fmac<J + 2>(i2, gx, x), row_fmac<J + 3>(i3, gx, x);
    row_fmac<J>(i0, gz, z), row_fmac<J + 1>(i1, gz, z), row_fmac<J + 2>(i2, gz, z), row_fmac<J + 3>(i3, gz, z);
    ph[M0 + J] = __builtin_fma(gdt, i0, ph[M0 + J]);
    ph[M0 + J + 1] = __builtin_fma(gdt, i1, ph[M0 + J + 1]);
    ph[M0 + J + 2] = __builtin_fma(gdt, i2, ph[M0 + J + 2]);
    ph[M0 + J + 3] = __builtin_fma(gdt, i3, ph[M0 + J + 3]);
    phase_step16<M0, J + 4, MB>(ph, gx, gy, gz, x, y, z, gdt);
  } else if constexpr (J < 16 && M0 + J < MB) { /* (instantiated, never run: fewer than four accumulators take other paths) */
    double inner = row_bcast<J>(gy) * y;
    row_fmac<J>(inner, gx, x);
    row_fmac<J>(inner, gz, z);
    ph[M0 + J] = __builtin_fma(gdt, inner, ph[M0 + J]);
    phase_step16<M0, J + 1, MB>(ph, gx, gy, gz, x, y, z, gdt);
  }
}

/* ---- the fused kernel --------------------------------------------------------- */

template <int GEOM, int MB, int GSRC, int COOP, int TRAJ = 0, int KA = 0>
DSD void walk_body(const WalkParams &p_arg) {
  const WalkParams &p = fresh_params<KA>(p_arg);
  constexpr bool ROWG = GSRC == 0 && MB >= 16 && GEOM != GEOM_MESH; /* gradient samples through DPP row broadcasts (analytic substrates) */
  /* [r6] the two tables of the random step as STATIC arrays: their LDS addresses are compile-time constants that fold into
   * the offset field of the reads (behind the dynamic array every table look-up first added the array's base, a
   * relocation the compiler cannot fold: six vector instructions per time step).  DS_STATIC_LDS_BYTES on the host. */
  __shared__ __attribute__((aligned(16))) double log_tab[DS_LOG_TABLE_DOUBLES]; /* 128 doubles */
  __shared__ __attribute__((aligned(16))) double cos_tab[DS_COS_TABLE_DOUBLES]; /* 514 doubles */
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double *bounds = lds; /* mesh only */
  int n_bounds = 0;
  if (GEOM == GEOM_MESH) n_bounds = p.mesh.nx + p.mesh.ny + p.mesh.nz + 3;
  double *gt = bounds + ((n_bounds + 1) & ~1);     /* gradient tile, 16-byte aligned  */
  /* mesh, cooperative traversal: one CoopWave per wavefront behind the gradient tile */
  CoopWave *coop = reinterpret_cast<CoopWave *>(gt + (GSRC == 0 ? (size_t)p.tile_steps * MB * 4 : 0)) + (threadIdx.x >> 6);

  for (int i = threadIdx.x; i < DS_LOG_TABLE_DOUBLES; i += blockDim.x) log_tab[i] = ds_log_table_dev[i];
  for (int i = threadIdx.x; i < DS_COS_TABLE_DOUBLES; i += blockDim.x) cos_tab[i] = ds_cos_table_dev[i];
  if (GEOM == GEOM_MESH)
    for (int i = threadIdx.x; i < n_bounds; i += blockDim.x) bounds[i] = p.mesh.bounds[i];
  const double *xs = bounds, *ys = bounds + p.mesh.nx + 1, *zs = bounds + p.mesh.nx + p.mesh.ny + 2;

  /* Workgroups are handed to the 8 XCDs round-robin (workgroup b runs on XCD b % 8) and every
   * XCD has its own L2.  Mesh walkers are sorted in space: an XCD takes runs of 1 << xcd_run_log2 consecutive
   * walker groups (default 8 groups = 2048 walkers, a compact region whose part of the index stays in that
   * XCD's L2), the runs are dealt to the XCDs in turn so that dense and sparse regions spread
   * evenly (giving every XCD one contiguous eighth was 11 % slower on the neuron).  The host
   * rounds the grid up to a multiple of 8 runs; groups beyond the last walker are empty. */
  int64_t group = blockIdx.x;
  if (GEOM == GEOM_MESH) {
    const unsigned slot = blockIdx.x >> 3; /* position in the XCD's own sequence */
    const int rl = p.xcd_run_log2;
    group = (((int64_t)(slot >> rl) * 8 + (blockIdx.x & 7)) << rl) + (slot & ((1u << rl) - 1u));
  }
  const int64_t i = group * blockDim.x + threadIdx.x;
  const bool live = i < p.n;
  V3 r = {0, 0, 0};
  Rng rng = {1, 1};
  bool exc = false;
  double ph[MB];
#pragma unroll
  for (int m = 0; m < MB; m++) ph[m] = 0;
  int64_t rec_i = i;
  if (TRAJ && live && p.rec_col != nullptr) rec_i = p.rec_col[i];
  if (live || ROWG) {
    const int64_t ii = live ? i : p.n - 1; /* ROWG: a surplus lane walks a copy of the last walker and stores nothing */
    r.x = p.x_in[ii];
    r.y = p.y_in[ii];
    r.z = p.z_in[ii];
    rng.s0 = p.s0_in[ii];
    rng.s1 = p.s1_in[ii];
    exc = p.exc[ii] != 0;
#pragma unroll
    for (int m = 0; m < MB; m++)
      if (m < p.mc) ph[m] = p.phase[(int64_t)(p.m0 + m) * p.phase_stride + ii];
  }

  for (int t0 = p.t_begin; t0 < p.t_end; t0 += p.tile_steps) {
    const int kt = min(p.tile_steps, p.t_end - t0);
    if (GSRC == 0) {
      __syncthreads(); /* previous tile fully consumed */
      /* tile rows are contiguous runs of mc*4 doubles inside g4[t][M][4]; measurement slots
       * beyond mc are zero-filled so that the accumulation loop needs no per-slot guard */
      const int row = p.mc * 2; /* double2 per step */
      const double2 *src = reinterpret_cast<const double2 *>(p.g4);
      double2 *dst = reinterpret_cast<double2 *>(gt);
      for (int e = threadIdx.x; e < kt * (MB * 2); e += blockDim.x) {
        int k = e / (MB * 2), c = e - k * (MB * 2);
        dst[e] = c < row ? src[((int64_t)(t0 + k) * p.M + p.m0) * 2 + c] : make_double2(0.0, 0.0);
      }
    }
    __syncthreads(); /* tile (and, the first time, the tables) visible */
    if (GEOM == GEOM_MESH && COOP) {
      /* Cooperative mesh traversal: every lane of the wavefront takes part in every pass, but each
       * walker runs through the kt time steps of the tile at its own pace (k_lane): a walker that
       * reflects needs further passes for the same step while its neighbours already start the
       * next one.  Per walker the sequence of operations (RNG draws included) is unchanged. */
      const int lane = (int)(threadIdx.x & 63);
#ifdef DSIM_TIMERS
      if (lane == 0) {
        for (int q = 0; q < 16; q++) coop->tacc[q] = 0;
        coop->tlast = __builtin_readcyclecounter();
      }
#endif
      /* k_lane: the walker's own clock relative to t0.  With p.tcur the clocks persist across launches: a launch
       * is then a budget of passes per wavefront instead of a range of time steps, so that no lane waits at the
       * end of a launch for neighbours that needed more passes (walkers in thin processes reflect on most steps,
       * walkers in the soma hardly ever). */
      int k_lane = 0;
      if (p.tcur != nullptr && live) k_lane = p.tcur[i] - t0;
      bool in_step = false, clear = false;
      /* one accumulator, gradient samples from global memory: the sample of a step is loaded when the step STARTS (6 VGPRs
       * more: 125 of the 128) - loaded when the step ends it is needed at once, and the wavefront waited for it */
      constexpr bool PREG = MB == 1 && GSRC == 1 && !TRAJ;
      double2 gpre_xy = make_double2(0.0, 0.0);
      double gpre_z = 0.0;
      V3 step = {0, 0, 0};
      double sl = 0;
      int64_t iter_idx = 0;
      bool need_norm = false; /* `step` holds a direction that has not been normalised yet */
      for (int passes = 0;; passes++) {
        const WalkParams &p = fresh_params<KA>(p_arg); /* nothing of the launch parameters stays in SGPRs across passes */
        const bool want = live && k_lane < kt;
        const bool may_start = p.tcur == nullptr || passes < p.iter_budget; /* wave-uniform */
        if (__ballot(want && (in_step || may_start)) == 0ULL) break;
        if (want && !in_step && may_start) {
          clear = walker_is_clear(p.mesh, r); /* look-ups issued before, consumed after the random step */
          if (PREG) { /* the step's gradient sample, asked for now: it is needed when the step ends, one to several passes from here */
            const double *g = p.g4 + ((int64_t)(t0 + k_lane) * p.M + p.m0) * 4;
            gpre_xy = *reinterpret_cast<const double2 *>(g);
            gpre_z = g[2];
          }
          step = DSIM_DEFER_NORM ? random_step_raw(rng, log_tab, cos_tab) : random_step(rng, log_tab, cos_tab);
          need_norm = DSIM_DEFER_NORM;
          sl = p.step_l;
          iter_idx = 0;
          in_step = true;
        }
        if (DSIM_DEFER_NORM && need_norm) { /* the new directions: of a started step, or of the reflection of the previous pass */
          normalize3(step);
          need_norm = false;
        }
        T_MARK(coop, lane, 0);
        const bool finished = coop_iteration<KA>(r, step, sl, iter_idx, in_step, clear, p, xs, ys, zs, rng, coop, lane, need_norm);
        clear = false; /* only the first pass of a step starts a full step length away from its start */
        if (in_step && finished) {
          if (iter_idx >= p.max_iter) exc = true; /* simulations.py:999-1002 */
          r.x = __builtin_fma(step.x, sl, r.x);
          r.y = __builtin_fma(step.y, sl, r.y);
          r.z = __builtin_fma(step.z, sl, r.z);
          if (TRAJ) {
            if (p.traj_row_steps > 0) {
              double *q = p.traj + ((int64_t)rec_i * p.traj_row_steps + (t0 + k_lane - p.t_begin)) * 4;
              *reinterpret_cast<double4 *>(q) = make_double4(r.x, r.y, r.z, 0.0);
            } else {
              double *q = p.traj + (int64_t)(t0 + k_lane - p.t_begin) * 3 * p.n + rec_i;
              q[0] = r.x, q[p.n] = r.y, q[2 * p.n] = r.z;
            }
          }
          if (TRAJ) {
            /* phases are accumulated from the recorded positions */
          } else if (GSRC == 0) {
            const double *g = gt + k_lane * (MB * 4);
#pragma unroll
            for (int m = 0; m < MB; m++) {
              double2 gxy = *reinterpret_cast<const double2 *>(g + m * 4);
              double gz = g[m * 4 + 2];
              double inner = __builtin_fma(gz, r.z, __builtin_fma(gxy.x, r.x, gxy.y * r.y));
              ph[m] = __builtin_fma(p.gdt, inner, ph[m]);
            }
          } else if (PREG) {
            double inner = __builtin_fma(gpre_z, r.z, __builtin_fma(gpre_xy.x, r.x, gpre_xy.y * r.y));
            ph[0] = __builtin_fma(p.gdt, inner, ph[0]);
          } else {
            const double *g = p.g4 + ((int64_t)(t0 + k_lane) * p.M + p.m0) * 4;
#pragma unroll
            for (int m = 0; m < MB; m++) {
              double inner = __builtin_fma(g[m * 4 + 2], r.z, __builtin_fma(g[m * 4], r.x, g[m * 4 + 1] * r.y));
              ph[m] = __builtin_fma(p.gdt, inner, ph[m]);
            }
          }
          k_lane++;
          in_step = false;
        }
        T_MARK(coop, lane, 9);
      }
      {
        const WalkParams &q = fresh_params<KA>(p_arg);
        if (q.tcur != nullptr) {
          if (live) q.tcur[i] = t0 + k_lane;
          if (q.progress != nullptr) {
            int left = live ? max(0, q.t_end - (t0 + k_lane)) : 0;
            const unsigned long long unfinished = __ballot(left > 0);
            int sum = left, mx = left;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
              sum += __shfl_down(sum, off, 64);
              mx = max(mx, __shfl_down(mx, off, 64));
            }
            if (lane == 0 && unfinished) {
              atomicAdd(&q.progress[0], (unsigned long long)__popcll(unfinished));
              atomicAdd(&q.progress[1], (unsigned long long)sum);
              atomicMax(&q.progress[2], (unsigned long long)mx);
            }
          }
        }
      }
#ifdef DSIM_TIMERS
      if (lane == 0)
        for (int q = 0; q < 16; q++) atomicAdd(&dsim_timers[q], coop->tacc[q]);
#endif
    } else if (live || ROWG) { /* ROWG: the row broadcasts want every lane active */
      for (int k = 0; k < kt; k++) {
        const WalkParams &p = fresh_params<KA>(p_arg); /* mesh (KA = 1): no launch parameter stays in SGPRs across time steps */
        V3 step = random_step(rng, log_tab, cos_tab);
        if (GEOM == GEOM_FREE) {
          /* simulations.py:690-691 */
          r.x = __builtin_fma(step.x, p.step_l, r.x);
          r.y = __builtin_fma(step.y, p.step_l, r.y);
          r.z = __builtin_fma(step.z, p.step_l, r.z);
        } else if (GEOM == GEOM_SPHERE) {
          step_sphere(r, step, p.step_l, p, exc);
        } else if (GEOM == GEOM_CYLINDER) {
          step_cylinder(r, step, p.step_l, p, exc);
        } else if (GEOM == GEOM_ELLIPSOID) {
          step_ellipsoid(r, step, p.step_l, p, exc);
        } else {
          step_mesh<KA>(r, step, p.step_l, p_arg, xs, ys, zs, rng, exc);
        }
        if (TRAJ && live) {
          double *q = p.traj + (int64_t)(t0 + k - p.t_begin) * 3 * p.n + rec_i;
          q[0] = r.x, q[p.n] = r.y, q[2 * p.n] = r.z;
        }
        /* phases[m, i] += GAMMA*dt*(gx*x + gy*y + gz*z), e.g. simulations.py:692-701 */
        if (ROWG) {
          /* 16 or more accumulators: read at a wave-uniform address, the samples of a step cost 2 LDS reads per
           * measurement - 12 cycles of the CU's one LDS pipe for every 4 fp64 operations of one of its four SIMDs.
           * Lane j of every row of 16 reads the sample of measurement j instead and the FMAs take it from there with
           * a DPP row broadcast (phase_step16): 2 reads per 16 measurements, one more VALU operation per measurement */
          const double *g = gt + k * (MB * 4) + (int)(threadIdx.x & 15u) * 4;
          double2 axy = *reinterpret_cast<const double2 *>(g);
          double az = g[2];
          dpp_fence(axy.x, axy.y, az);
          phase_step16<0, 0, MB>(ph, axy.x, axy.y, az, r.x, r.y, r.z, p.gdt);
          if (MB > 16) {
            double2 bxy = *reinterpret_cast<const double2 *>(g + 64);
            double bz = g[66];
            dpp_fence(bxy.x, bxy.y, bz);
            phase_step16<16, 0, MB>(ph, bxy.x, bxy.y, bz, r.x, r.y, r.z, p.gdt);
          }
        } else if (GSRC == 0) {
          /* every lane reads the same LDS address: broadcast */
          const double *g = gt + k * (MB * 4);
#pragma unroll
          for (int m = 0; m < MB; m++) {
            double2 gxy = *reinterpret_cast<const double2 *>(g + m * 4);
            double gz = g[m * 4 + 2];
            double inner = __builtin_fma(gz, r.z, __builtin_fma(gxy.x, r.x, gxy.y * r.y));
            ph[m] = __builtin_fma(p.gdt, inner, ph[m]);
          }
        } else {
          /* wave-uniform address: the compiler emits scalar (SMEM) loads and the gradient
           * samples feed the fp64 FMAs straight from SGPRs (g4 is padded by 32 samples) */
          const double *g = p.g4 + ((int64_t)(t0 + k) * p.M + p.m0) * 4;
#pragma unroll
          for (int m = 0; m < MB; m++) {
            double inner = __builtin_fma(g[m * 4 + 2], r.z, __builtin_fma(g[m * 4], r.x, g[m * 4 + 1] * r.y));
            ph[m] = __builtin_fma(p.gdt, inner, ph[m]);
          }
        }
      }
    }
  }

  if (live) {
    const WalkParams &p = fresh_params<KA>(p_arg);
#pragma unroll
    for (int m = 0; m < MB; m++)
      if (m < p.mc) p.phase[(int64_t)(p.m0 + m) * p.phase_stride + i] = ph[m];
    if (p.write_state) {
      p.x_out[i] = r.x;
      p.y_out[i] = r.y;
      p.z_out[i] = r.z;
      p.s0_out[i] = rng.s0;
      p.s1_out[i] = rng.s1;
      p.exc[i] = exc ? 1 : 0;
    }
  }
}

template <int GEOM, int MB, int GSRC, int COOP>
__global__ __launch_bounds__(256) void walk_kernel(const WalkParams p) {
  /* the mesh kernels (per-lane traversal, and the cooperative one with more than four accumulators) re-read their launch
   * parameters where they use them (fresh_params): read once, MeshDev alone is ~40 SGPRs that the traversal's loops
   * then spill through VGPR lanes */
  walk_body<GEOM, MB, GSRC, COOP, 0, (GEOM == GEOM_MESH || MB >= 16) ? 1 : 0>(p);
}

/* Cooperative mesh traversal with few phase accumulators: capped at 128 VGPRs (a handful of
 * spills) so that four wavefronts share a SIMD - the traversal is a chain of dependent look-ups
 * and every resident wavefront hides some of its latency (measured: +10..14 % over three). */
template <int MB, int GSRC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void walk_kernel_coop4(const WalkParams p) {
  walk_body<GEOM_MESH, MB, GSRC, 1, 0, 1>(p);
}

/* 32 measurements in the kernel and the positions of every step recorded in p.traj: the first pass
 * of a run with M > 32; the other measurements are accumulated by phase_from_traj_kernel instead
 * of replaying the walk */
template <int GEOM>
__global__ __launch_bounds__(256) void walk_kernel_traj32(const WalkParams p) {
  walk_body<GEOM, 32, 0, 0, 1, 1>(p);
}

/* the same, recording the position after every step in p.traj (phases are then accumulated by
 * phase_from_traj_kernel: mesh substrates with more than two measurements) */
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void walk_kernel_coop4_traj(const WalkParams p) {
  walk_body<GEOM_MESH, 1, 1, 1, 1, 1>(p);
}

/* Phase accumulation from recorded positions: phases[m, i] += GAMMA*dt * g[m, t] . r_i(t) for the
 * time steps of one launch, in the same order and with the same roundings as the fused kernels.
 * Used for mesh substrates with more than two measurements: the cooperative walk kernel then
 * carries no phase accumulators at all (128 VGPRs, 4 waves/SIMD whatever M is), writes 24 bytes
 * per walker and step, and this kernel - streaming + FMAs - does the rest.  traj = [step][3][n].
 * [r4] The gradient samples of a step are the same for every walker.  Read from an LDS tile at a wave-uniform address
 * (rounds 1-3: 2 LDS reads per measurement and step, 12 LDS cycles for 4 fp64 operations) the LDS pipe of the CU, not
 * its four SIMDs, set the pace: 19 ms per launch of 5e6 walkers x 143 steps x 32 measurements, a fifth of the HBM rate.
 * Now lane j of every row of 16 loads the sample of measurement j (three loads per 16 measurements and step, out of
 * L1) and the FMAs take it from there with a DPP row broadcast - no LDS, no barrier, 5 VALU operations per measurement
 * and step.  tile_steps is not used any more. */
template <int MB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void phase_from_traj_kernel(const double *traj, int64_t n, int steps, int tile_steps, const double *g4,
                                                              int M, int m0, int mc, int t_first, double gdt, double *phase,
                                                              int64_t phase_stride, const int32_t *rec_col) {
  (void)tile_steps;
  const int64_t i_raw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i_raw < n;
  const int64_t i = live ? i_raw : n - 1; /* every lane stays active (row broadcasts); the surplus lanes redo the last walker and store nothing */
  const int64_t rec_i = rec_col != nullptr ? (int64_t)rec_col[i] : i; /* where slot i's walker was recorded (WalkParams::rec_col) */
  const int j = (int)(threadIdx.x & 15u);
  double ph[MB];
#pragma unroll
  for (int m = 0; m < MB; m++) ph[m] = m < mc ? phase[(int64_t)(m0 + m) * phase_stride + i] : 0.0;
  /* lane j reads the samples of measurements m0 + j and m0 + 16 + j; beyond mc they are somebody else's samples or the
   * padding behind the table (g4 is padded by 32 samples) and feed accumulators that are never stored */
  const double *gp = g4 + ((int64_t)t_first * M + m0 + j) * 4;
  const double *q = traj + rec_i;
  double x = q[0], y = q[n], z = q[2 * n];
  double2 axy = *reinterpret_cast<const double2 *>(gp), bxy = make_double2(0.0, 0.0);
  double az = gp[2], bz = 0.0;
  if (MB > 16) bxy = *reinterpret_cast<const double2 *>(gp + 64), bz = gp[66];
  for (int k = 0; k < steps; k++) {
    /* the next step's operands are on their way while this one's are used (the last step reads its own again) */
    const int kn = min(k + 1, steps - 1);
    const double *gn = gp + (int64_t)kn * M * 4; /* never beyond the last step: m0 + 31 <= M + 30 stays inside the padding */
    const double *qn = traj + (int64_t)kn * 3 * n + rec_i;
    const double xn = qn[0], yn = qn[n], zn = qn[2 * n];
    const double2 axy_n = *reinterpret_cast<const double2 *>(gn);
    const double az_n = gn[2];
    double2 bxy_n = make_double2(0.0, 0.0);
    double bz_n = 0.0;
    if (MB > 16) bxy_n = *reinterpret_cast<const double2 *>(gn + 64), bz_n = gn[66];
    dpp_fence(axy.x, axy.y, az);
    phase_step16<0, 0, MB>(ph, axy.x, axy.y, az, x, y, z, gdt);
    if (MB > 16) {
      dpp_fence(bxy.x, bxy.y, bz);
      phase_step16<16, 0, MB>(ph, bxy.x, bxy.y, bz, x, y, z, gdt);
    }
    x = xn, y = yn, z = zn;
    axy = axy_n, az = az_n, bxy = bxy_n, bz = bz_n;
  }
  if (live) {
#pragma unroll
    for (int m = 0; m < MB; m++)
      if (m < mc) phase[(int64_t)(m0 + m) * phase_stride + i] = ph[m];
  }
}

/* The same accumulation from WALKER-major positions, rows[slot][row_steps][x y z 0] (WalkParams::traj_row_steps; row_steps
 * even).  Read by its own lane a row costs a 32-byte request per lane and step - 64 cache lines per load instruction, and
 * the address unit of the CU, not its SIMDs, sets the pace (measured: the whole job gains 3 % where the walk kernel gained
 * 10 %).  So a wavefront fetches the rows of its 64 walkers TOGETHER, two steps (64 bytes of every row) at a time, straight
 * into LDS (global_load_lds_dwordx4: four per chunk, each 16 rows x 64 bytes, no VGPRs in between), one chunk ahead of the
 * arithmetic, and every lane then reads its own row out of LDS.  The 16-byte pieces of a row sit at position
 * piece ^ ((row >> 1) & 3), which spreads the 64 lanes of a 16-byte read evenly over the banks.  4 KB per chunk, two chunks
 * in flight per wavefront: 32 KB per workgroup - what phase_lds_reserve asked for without using it is now used. */
#define PHASE_ROWS_LDS_PER_WAVE 8192
template <int MB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void phase_from_rows_kernel(const double *rows, int64_t n, int steps, int row_steps,
                                                                                                      const double *g4, int M, int m0, int mc, int t_first,
                                                                                                      double gdt, double *phase, int64_t phase_stride) {
  extern __shared__ __attribute__((aligned(16))) char stage[];
  const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
  char *buf = stage + wave * PHASE_ROWS_LDS_PER_WAVE;
  const int64_t i_raw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i_raw < n;
  const int64_t i = live ? i_raw : n - 1; /* every lane stays active (row broadcasts); surplus lanes redo the last walker and store nothing */
  const int j = lane & 15;
  double ph[MB];
#pragma unroll
  for (int m = 0; m < MB; m++) ph[m] = m < mc ? phase[(int64_t)(m0 + m) * phase_stride + i] : 0.0;
  /* what this lane fetches for the wavefront: in fetch q (0..3) the piece `pos ^ ((r >> 1) & 3)` of row r = 16 q + lane / 4 */
  const int64_t w0 = (int64_t)blockIdx.x * blockDim.x + wave * 64;
  const char *src[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int r = 16 * q + (lane >> 2), pos = lane & 3;
    const int64_t slot = w0 + r < n ? w0 + r : n - 1;
    src[q] = reinterpret_cast<const char *>(rows) + (slot * row_steps) * 32 + ((pos ^ ((r >> 1) & 3)) << 4);
  }
  auto fetch = [&](int c) { /* chunk c = steps 2c, 2c + 1 of the 64 rows -> buffer c & 1 */
    char *dst = buf + (c & 1) * 4096;
#pragma unroll
    for (int q = 0; q < 4; q++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[q] + (int64_t)c * 64),
                                       (__attribute__((address_space(3))) void *)(dst + q * 1024), 16, 0, 0);
  };
  const int n_chunks = (steps + 1) >> 1;
  fetch(0);
  const double *gp = g4 + ((int64_t)t_first * M + m0 + j) * 4;
  double2 axy = *reinterpret_cast<const double2 *>(gp), bxy = make_double2(0.0, 0.0);
  double az = gp[2], bz = 0.0;
  if (MB > 16) bxy = *reinterpret_cast<const double2 *>(gp + 64), bz = gp[66];
  const int sw = (lane >> 1) & 3;
  for (int c = 0; c < n_chunks; c++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* chunk c has landed in LDS (it was asked for a whole chunk of arithmetic ago) */
    __builtin_amdgcn_wave_barrier();
    if (c + 1 < n_chunks) fetch(c + 1);
    const char *mine = buf + (c & 1) * 4096 + lane * 64;
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++) {
      const int k = 2 * c + s2;
      if (k < steps) { /* wave-uniform */
        const double2 xy = *reinterpret_cast<const double2 *>(mine + (((2 * s2) ^ sw) << 4));
        const double z = *reinterpret_cast<const double *>(mine + (((2 * s2 + 1) ^ sw) << 4));
        const int kn = min(k + 1, steps - 1);
        const double *gn = gp + (int64_t)kn * M * 4; /* never beyond the last step: m0 + 31 <= M + 30 stays inside the padding */
        const double2 axy_n = *reinterpret_cast<const double2 *>(gn);
        const double az_n = gn[2];
        double2 bxy_n = make_double2(0.0, 0.0);
        double bz_n = 0.0;
        if (MB > 16) bxy_n = *reinterpret_cast<const double2 *>(gn + 64), bz_n = gn[66];
        dpp_fence(axy.x, axy.y, az);
        phase_step16<0, 0, MB>(ph, axy.x, axy.y, az, xy.x, xy.y, z, gdt);
        if (MB > 16) {
          dpp_fence(bxy.x, bxy.y, bz);
          phase_step16<16, 0, MB>(ph, bxy.x, bxy.y, bz, xy.x, xy.y, z, gdt);
        }
        axy = axy_n, az = az_n, bxy = bxy_n, bz = bz_n;
      }
    }
  }
  if (live) {
#pragma unroll
    for (int m = 0; m < MB; m++)
      if (m < mc) phase[(int64_t)(m0 + m) * phase_stride + i] = ph[m];
  }
}

/* ---- signal: sum over walkers of cos/sin(phase), wavefront shuffles ----------- */

DSD double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

/* The sum over walkers is taken EXACTLY: every term cos(phase), sin(phase) is rounded once to a multiple of 2^-60
 * (exact for |term| >= 2^-8, an absolute error below 2^-61 otherwise) and the multiples are added as integers, split
 * into a high and a low limb of 30 bits so that 2^33 terms fit 64-bit accumulators: q = hi * 2^30 + lo, 0 <= lo < 2^30.
 * Integer addition is associative, so the signal does not depend on how the walkers are cut into threads, workgroups,
 * shards, devices or ranks - an 8-GPU job returns the bytes of the 1-GPU job (the reference adds the doubles up with
 * numpy's pairwise sum on the host, simulations.py:1413-1421; the exact sum is within N * 2^-61 of the true one).
 * Walkers with iter_exc set or a NaN phase contribute nothing (the reference's NaN mask + nansum). */
#define DS_SIG_SCALE 0x1p60
#define DS_SIG_LIMB_BITS 30

DSD long long wave_sum_i64(long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int lo = __shfl_down((int)(v & 0xffffffffLL), off, 64), hi = __shfl_down((int)(v >> 32), off, 64);
    v += ((long long)hi << 32) | (long long)(unsigned)lo;
  }
  return v;
}

/* grid = (blocks, M); partial[(m*gridDim.x + b)*4 + {0..3}] = block sums of the limbs {cos hi, cos lo, sin hi, sin lo};
 * the m == 0 row also counts the exceeded walkers into nexc_partial[b].
 * Replaces phases D2H + np.nansum(np.exp(1j*phases)) (simulations.py:1413-1421). */
__global__ __launch_bounds__(256) void signal_partial_kernel(const double *phase, const uint8_t *exc, int64_t n,
                                                             long long *partial, long long *nexc_partial) {
  __shared__ long long sm[4][5];
  const int m = blockIdx.y;
  long long ch = 0, cl = 0, sh = 0, sl = 0, e = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double ph = phase[(int64_t)m * n + i];
    const bool x = exc[i] != 0;
    e += x ? 1 : 0;
    if (!x && ph == ph) {
      double sv, cv;
      sincos(ph, &sv, &cv);
      if (sv == sv && cv == cv) {
        const long long qc = (long long)__builtin_rint(cv * DS_SIG_SCALE), qs = (long long)__builtin_rint(sv * DS_SIG_SCALE);
        ch += qc >> DS_SIG_LIMB_BITS, cl += qc & ((1LL << DS_SIG_LIMB_BITS) - 1);
        sh += qs >> DS_SIG_LIMB_BITS, sl += qs & ((1LL << DS_SIG_LIMB_BITS) - 1);
      }
    }
  }
  ch = wave_sum_i64(ch), cl = wave_sum_i64(cl), sh = wave_sum_i64(sh), sl = wave_sum_i64(sl), e = wave_sum_i64(e);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) sm[wave][0] = ch, sm[wave][1] = cl, sm[wave][2] = sh, sm[wave][3] = sl, sm[wave][4] = e;
  __syncthreads();
  if (threadIdx.x < 5) {
    long long t = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += sm[w][threadIdx.x];
    if (threadIdx.x < 4)
      partial[((int64_t)m * gridDim.x + blockIdx.x) * 4 + threadIdx.x] = t;
    else if (m == 0)
      nexc_partial[blockIdx.x] = t;
  }
}

/* one wavefront per measurement adds the block partials up (integers: any order gives the same sums); block M adds
 * the exceeded-walker counts.  out: [4 M + 1] = per measurement {cos hi, cos lo, sin hi, sin lo}, then the count */
__global__ __launch_bounds__(64) void signal_final_kernel(const long long *partial, const long long *nexc_partial, int n_blocks,
                                                          int M, long long *out /* [4M + 1] */) {
  const int m = blockIdx.x;
  if (m < M) {
    long long t[4] = {0, 0, 0, 0};
    for (int b = threadIdx.x; b < n_blocks; b += 64)
#pragma unroll
      for (int q = 0; q < 4; q++) t[q] += partial[((int64_t)m * n_blocks + b) * 4 + q];
#pragma unroll
    for (int q = 0; q < 4; q++) t[q] = wave_sum_i64(t[q]);
    if (threadIdx.x == 0)
      for (int q = 0; q < 4; q++) out[4 * m + q] = t[q];
  } else {
    long long cnt = 0;
    for (int b = threadIdx.x; b < n_blocks; b += 64) cnt += nexc_partial[b];
    cnt = wave_sum_i64(cnt);
    if (threadIdx.x == 0) out[4 * M] = cnt;
  }
}

/* all_signals (simulations.py:1414-1417): out[e] = cos(phase[first + e]) for the elements [first, first + count) of
 * the flattened (M, n) phase matrix, NaN where the walker's iter_exc flag is set.  grid = (walkers / 256, rows the
 * range touches): blockIdx.y names the measurement, so the walker index needs no 64-bit modulo per element. */
__global__ __launch_bounds__(256) void all_signals_kernel(const double *phase, const uint8_t *exc, int64_t n, int64_t first,
                                                          int64_t count, int64_t first_row, double *out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t flat = (first_row + blockIdx.y) * n + i;
  if (i >= n || flat < first || flat >= first + count) return;
  out[flat - first] = exc[i] != 0 ? DS_NAN : cos(phase[flat]);
}

/* ---- layout conversion ---------------------------------------------------------- */

__global__ void aos_to_soa_kernel(const double *aos, int64_t n, double *x, double *y, double *z) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    x[i] = aos[i * 3];
    y[i] = aos[i * 3 + 1];
    z[i] = aos[i * 3 + 2];
  }
}

__global__ void soa_to_aos_kernel(const double *x, const double *y, const double *z, int64_t n, double *aos) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    aos[i * 3] = x[i];
    aos[i * 3 + 1] = y[i];
    aos[i * 3 + 2] = z[i];
  }
}

__global__ void states_to_soa_kernel(const uint64_t *st, int64_t n, uint64_t *s0, uint64_t *s1) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    s0[i] = st[i * 2];
    s1[i] = st[i * 2 + 1];
  }
}

__global__ void soa_to_states_kernel(const uint64_t *s0, const uint64_t *s1, int64_t n, uint64_t *st) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    st[i * 2] = s0[i];
    st[i * 2 + 1] = s1[i];
  }
}

/* ---- spatial re-sorting of walker slots (mesh substrates) -------------------------- */

DSD uint32_t spread10(uint32_t v) { /* 10 bits -> every third bit */
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

/* ---- the re-sort: a counting sort on the top bits of the Morton code, hand-written (round 2 called hipcub's radix
 * sort: 21 dispatches of its own per re-sort, each of which had to find a free wavefront slot between the other walker
 * group's workgroups).  Three kernels per re-sort:
 *   sort_count_kernel    bin of every walker (kept in bin_of[]), histogram with ONE atomic per wavefront and distinct
 *                        bin (the walkers arrive nearly sorted: a wavefront holds one to three bins), and - in the same
 *                        pass over the positions - how many walkers sit in a clear cell of the clearance map;
 *   sort_scan_kernel     exclusive prefix sum over the bins, one workgroup of 1024 (the histogram is zeroed for the next
 *                        time, as are the counters the NEXT launch will use);
 *   sort_scatter_kernel  every walker takes a slot of its bin (one atomic per wavefront and bin again) and moves there
 *                        with everything it owns: position, RNG state, iter_exc, clock, original index, phases.
 * The order inside a bin is whatever the atomics make it - walkers do not interact, results cannot depend on it, and
 * the order is undone before anything leaves the run.  Bins are 2^bits cells of the periodic voxel in Morton order
 * (bits = 18: 64^3 cells of about 3.4 um on the neuron - a walker moves 2.7 um in the 256 steps between two re-sorts),
 * plus one last bin for walkers that have reached the end of the call. */
DSD uint32_t walker_morton30(double x, double y, double z, double inv_lx, double inv_ly, double inv_lz) {
  double fx = x * inv_lx, fy = y * inv_ly, fz = z * inv_lz;
  fx -= __builtin_floor(fx);
  fy -= __builtin_floor(fy);
  fz -= __builtin_floor(fz);
  /* only a sort key: approximate arithmetic is fine, results never depend on it */
  uint32_t ix = fx == fx ? (uint32_t)min(1023, max(0, (int)(fx * 1024.0))) : 0u;
  uint32_t iy = fy == fy ? (uint32_t)min(1023, max(0, (int)(fy * 1024.0))) : 0u;
  uint32_t iz = fz == fz ? (uint32_t)min(1023, max(0, (int)(fz * 1024.0))) : 0u;
  return (spread10(ix) << 2) | (spread10(iy) << 1) | spread10(iz);
}

struct SortParams {
  const double *x, *y, *z;
  int64_t n;
  double inv_lx, inv_ly, inv_lz;
  const int32_t *tcur; /* per-walker clocks: walkers at t_end go to the last bin (nullptr: no clocks) */
  int32_t t_end;
  int by_steps_left;   /* last launch of a call: 64 classes of steps left first, space inside a class */
  int bits;            /* bins = (1 << bits) + 1 */
  uint32_t *bin_of;    /* [n] */
  uint32_t *hist;      /* [(1 << bits) + 1], zero on entry */
  unsigned long long *clear_count; /* += walkers in a clear cell (nullptr: not asked) */
};

/* one atomic per distinct value of `bin` among the valid lanes of the wavefront; returns what the atomic returned plus
 * the lane's rank among the lanes with the same bin (the lane's own slot when `counter` is a cursor) */
DSD uint32_t wave_bin_add(uint32_t *counter, uint32_t bin, bool valid) {
  const int lane = (int)(threadIdx.x & 63);
  uint32_t slot = 0;
  unsigned long long todo = __ballot(valid);
  while (todo) {
    const int leader = (int)__ffsll((long long)todo) - 1;
    const uint32_t b = (uint32_t)__shfl((int)bin, leader, 64);
    const unsigned long long same = __ballot(valid && bin == b) & todo;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&counter[b], (uint32_t)__popcll(same));
    base = (uint32_t)__shfl((int)base, leader, 64);
    if (valid && bin == b) slot = base + (uint32_t)__popcll(same & ((1ULL << lane) - 1ULL));
    todo &= ~same;
  }
  return slot;
}

__global__ __launch_bounds__(256) void sort_count_kernel(const SortParams p, const MeshDev clear_map) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < p.n;
  uint32_t bin = 0;
  bool clear = false;
  if (valid) {
    const double x = p.x[i], y = p.y[i], z = p.z[i];
    if (p.tcur != nullptr && p.tcur[i] >= p.t_end) {
      bin = 1u << p.bits;
    } else {
      const uint32_t m = walker_morton30(x, y, z, p.inv_lx, p.inv_ly, p.inv_lz);
      if (p.by_steps_left && p.tcur != nullptr) {
        /* walkers with a similar number of steps left share wavefronts, so that no lane idles long for its neighbours */
        const uint32_t cls = (uint32_t)min(63, (p.t_end - p.tcur[i]) >> 5);
        bin = (cls << (p.bits - 6)) | (m >> (36 - p.bits));
      } else {
        bin = m >> (30 - p.bits);
      }
    }
    p.bin_of[i] = bin;
    if (p.clear_count != nullptr) {
      const V3 r = {x, y, z};
      clear = walker_is_clear(clear_map, r);
    }
  }
  (void)wave_bin_add(p.hist, bin, valid);
  if (p.clear_count != nullptr) {
    const unsigned long long m = __ballot(clear);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(p.clear_count, (unsigned long long)__popcll(m));
  }
}

/* exclusive prefix sum of hist[0 .. n_bins) into offs, hist zeroed; ONE workgroup of 256 threads walks the bins in tiles
 * of 4096: a thread takes 16 consecutive bins with four 16-byte loads issued together (the first version - 4 bins per
 * thread, 257 dependent tile iterations for 2^18 bins - took 0.87 ms, 8 % of a launch cycle of a 6.25e5-walker group;
 * a workgroup of 1024 threads is no answer: it has to wait until ONE CU has sixteen wavefront slots free between the
 * other group's walk kernel, measured -0.3 %).  zero4: four counters to clear for the launch after this one. */
#define SORT_SCAN_THREADS 256
__global__ __launch_bounds__(SORT_SCAN_THREADS) void sort_scan_kernel(uint32_t *hist, uint32_t *offs, int n_bins, unsigned long long *zero4) {
  __shared__ uint32_t wsum[SORT_SCAN_THREADS / 64];
  const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
  if (zero4 != nullptr && t < 4) zero4[t] = 0ULL;
  uint32_t carry = 0;
  for (int base = 0; base < n_bins; base += 16 * SORT_SCAN_THREADS) {
    const int b0 = base + 16 * t;
    const bool full = b0 + 16 <= n_bins;
    uint32_t v[16];
    if (full) {
      uint4 *src = reinterpret_cast<uint4 *>(hist + b0);
      const uint4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
      v[0] = q0.x, v[1] = q0.y, v[2] = q0.z, v[3] = q0.w, v[4] = q1.x, v[5] = q1.y, v[6] = q1.z, v[7] = q1.w;
      v[8] = q2.x, v[9] = q2.y, v[10] = q2.z, v[11] = q2.w, v[12] = q3.x, v[13] = q3.y, v[14] = q3.z, v[15] = q3.w;
      const uint4 z = make_uint4(0u, 0u, 0u, 0u);
      src[0] = z, src[1] = z, src[2] = z, src[3] = z;
    } else {
#pragma unroll
      for (int q = 0; q < 16; q++) {
        v[q] = b0 + q < n_bins ? hist[b0 + q] : 0u;
        if (b0 + q < n_bins) hist[b0 + q] = 0u;
      }
    }
    uint32_t mine = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) mine += v[q];
    const uint32_t incl = (uint32_t)wave_inclusive_sum((int)mine);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t before = carry, total = 0;
#pragma unroll
    for (int w = 0; w < SORT_SCAN_THREADS / 64; w++) {
      const uint32_t sw = wsum[w];
      total += sw;
      if (w < wave) before += sw;
    }
    uint32_t run = before + incl - mine;
    if (full) {
      uint32_t o[16];
#pragma unroll
      for (int q = 0; q < 16; q++) {
        o[q] = run;
        run += v[q];
      }
      uint4 *dst = reinterpret_cast<uint4 *>(offs + b0);
      dst[0] = make_uint4(o[0], o[1], o[2], o[3]), dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
      dst[2] = make_uint4(o[8], o[9], o[10], o[11]), dst[3] = make_uint4(o[12], o[13], o[14], o[15]);
    } else {
#pragma unroll
      for (int q = 0; q < 16; q++) {
        if (b0 + q < n_bins) offs[b0 + q] = run;
        run += v[q];
      }
    }
    carry += total;
    __syncthreads();
  }
}

struct ScatterParams {
  int64_t n, phase_stride;
  int M;
  const uint32_t *bin_of;
  uint32_t *offs; /* cursors: first free slot of every bin */
  const double *x, *y, *z;
  const uint64_t *s0, *s1;
  const uint8_t *exc;
  const int32_t *orig; /* nullptr: the walkers are in their original order (orig = index) */
  const int32_t *tcur; /* nullptr: no clocks */
  const double *phase;
  double *xo, *yo, *zo;
  uint64_t *s0o, *s1o;
  uint8_t *exco;
  int32_t *origo, *tcuro;
  double *phaseo;
};

__global__ __launch_bounds__(256) void sort_scatter_kernel(const ScatterParams p) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < p.n;
  const uint32_t bin = valid ? p.bin_of[i] : 0u;
  const uint32_t j = wave_bin_add(p.offs, bin, valid);
  if (!valid) return;
  p.xo[j] = p.x[i];
  p.yo[j] = p.y[i];
  p.zo[j] = p.z[i];
  p.s0o[j] = p.s0[i];
  p.s1o[j] = p.s1[i];
  p.exco[j] = p.exc[i];
  p.origo[j] = p.orig != nullptr ? p.orig[i] : (int32_t)i;
  if (p.tcur != nullptr) p.tcuro[j] = p.tcur[i];
  for (int m = 0; m < p.M; m++) p.phaseo[(int64_t)m * p.phase_stride + j] = p.phase[(int64_t)m * p.phase_stride + i];
}

__global__ void fill_i32_kernel(int32_t *v, int64_t n, int32_t value) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = value;
}

/* slot i returns to walker index orig[i] */
__global__ void scatter_state_kernel(const int32_t *orig, int64_t n, const double *x, const double *y, const double *z,
                                     const uint64_t *s0, const uint64_t *s1, const uint8_t *exc, double *xo, double *yo,
                                     double *zo, uint64_t *s0o, uint64_t *s1o, uint8_t *exco) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t j = orig[i];
  xo[j] = x[i];
  yo[j] = y[i];
  zo[j] = z[i];
  s0o[j] = s0[i];
  s1o[j] = s1[i];
  exco[j] = exc[i];
}

__global__ void scatter_rows_kernel(const int32_t *orig, int64_t n, int64_t stride, const double *src, double *dst) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[(int64_t)blockIdx.y * stride + orig[i]] = src[(int64_t)blockIdx.y * stride + i];
}

/* ---- _cuda_fill_mesh (simulations.py:421-502) ------------------------------------ */

struct FillParams {
  /* The fill kernel's own index of the (possibly "stripped") subvoxel lists, built by dsim_fill_mesh:
   *  - per (y, z) column the NON-EMPTY subvoxels in ascending x: col_ent[col_start[c] .. col_start[c + 1]) =
   *    {x, first entry}; the entries of one subvoxel end where those of the next element begin (sentinel at the
   *    end).  A +x ray only ever walks along one column, and on a sparse mesh (the neuron: 1.8 % of the subvoxels
   *    hold triangles) almost all of the ~nx/2 subvoxels it crosses are empty: skipping them changes neither the
   *    triangles met nor their order;
   *  - per entry {triangle, prev} and the triangle's (y, z) extent.  `prev` = the largest x' <= x at which the same
   *    triangle occurs EARLIER in this column's visiting order (an earlier position of the same list counts as
   *    x' = x), -1 if none.  The reference keeps the triangles it has hit in a 1000-slot local array and skips a
   *    triangle that is already there (simulations.py:484-492); the same point, ray and triangle give the same
   *    answer at every occurrence, and a ray that starts in subvoxel llx visits all of the column's subvoxels
   *    from llx on, so "already hit" == "occurs earlier at some x' >= llx" == "prev >= llx".  No per-thread
   *    array: the kernel needs no scratch memory and runs at full occupancy. */
  const int32_t *col_start;
  const int2 *col_ent;
  const int2 *ent;
  const float4 *ent_yz;
  /* The same entries once more, bucketed by where in the column's (y, z) cross-section the triangle lies: a +x
   * ray keeps its (y, z), so only the triangles whose (y, z) extent holds the point can be met, and a triangle
   * of a fine mesh covers a few per cent of a subvoxel's cross-section.  mcol[c] = {first bucket, G}: the
   * column is cut into G x G buckets, bucket b holds ment[mstart[b] .. mstart[b + 1]) = {triangle,
   * x << 16 | (prev + 1)} in DESCENDING x (the ray's subvoxels are the column's tail from llx on, so the walk
   * starts at the far end and stops at the first x < llx), ment_yz the extents.  Only the NUMBER of distinct
   * triangles met matters (its parity), not the order - except for the reference's give-up rule at 1000 hits,
   * for which the kernel falls back to the reference's own order (fill_count_reference_order). */
  const int2 *mcol;
  const int32_t *mstart;
  const int2 *ment;
  const float4 *ment_yz;
  int64_t n_points;
  uint64_t *s0, *s1;
  /* one dispatch = a BATCH of the reference's launches (fill_batch_kernel): */
  int n_launches;          /* launches of the batch, <= FILL_MAX_BATCH */
  int stage_cap;           /* accepted candidates a workgroup can park (FillEntry each) */
  struct FillEntry *stage; /* [workgroups][stage_cap], in the order the workgroup's threads got there */
  int32_t *stage_count;    /* [workgroups] entries parked */
  int32_t *cnt;            /* [n_launches][workgroups] candidates accepted per launch and workgroup */
  int32_t *overflow;       /* set when a workgroup ran out of parking room: the batch is void, the host redoes it */
  int intra;
  double voxel[3];
  MeshDev mesh;
};

#define FILL_MAX_HITS 1000
#define FILL_MAX_BATCH 64

/* an accepted candidate on its way to its row: key = launch << 8 | thread inside the workgroup */
struct FillEntry {
  double x, y, z;
  uint32_t key, pad;
};

/* The reference's loop (simulations.py:461-492) over one column in its own order; returns the number of distinct
 * triangles met, or -1 when the sample is abandoned (:478-479: the 1000-slot list is full and another list entry
 * follows). */
DSD int fill_count_reference_order(const FillParams &p, const V3 &pt, int llx, int ulx, int lly, int uly, int llz, int ulz) {
  const MeshDev &ms = p.mesh;
  const V3 ray = {1.0, 0.0, 0.0};
  const float fy = (float)pt.y, fz = (float)pt.z;
  int n_hits = 0;
  for (int y = lly; y < uly; y++)
    for (int z = llz; z < ulz; z++) {
      const int col = y * ms.nz + z;
      for (int e = p.col_start[col]; e < p.col_start[col + 1]; e++) {
        const int2 ce = p.col_ent[e];
        if (ce.x < llx) continue;
        if (ce.x >= ulx) break;
        const int end = p.col_ent[e + 1].y;
        for (int i = ce.y; i < end; i++) {
          if (n_hits >= FILL_MAX_HITS) return -1;
          const int2 en = p.ent[i];
          const float4 yz = p.ent_yz[i];
          if (fy < yz.x || fy > yz.z || fz < yz.y || fz > yz.w) continue;
          if (en.y >= llx) continue; /* met before on this ray: counted there if it was a hit */
          const double *rec = ms.tri + (int64_t)en.x * DS_TRI_STRIDE;
          double d = ray_triangle_intersection_check(load_v3(rec), load_v3(rec + 3), load_v3(rec + 6), pt, ray);
          if (d > 0) n_hits++;
        }
      }
    }
  return n_hits;
}

/* bucket of the point pt in its (y, z) column lly, llz (FillParams::mcol): {first entry, end} of ment, or {0, 0} when the
 * column holds no triangle at all */
DSD int2 fill_bucket(const FillParams &p, const double *ys, const double *zs, const V3 &pt, int lly, int llz) {
  const MeshDev &ms = p.mesh;
  const int2 mc = p.mcol[lly * ms.nz + llz];
  if (mc.x < 0) return make_int2(0, 0);
  const int G = mc.y;
  int iy = (int)((pt.y - ys[lly]) * ((double)G * ms.inv_hy)), iz = (int)((pt.z - zs[llz]) * ((double)G * ms.inv_hz));
  iy = min(max(iy, 0), G - 1), iz = min(max(iz, 0), G - 1);
  const int b = mc.x + iy * G + iz;
  return make_int2(p.mstart[b], p.mstart[b + 1]);
}

/* The triangles a +x ray from pt meets among the entries [begin, end) of its bucket: the number of distinct ones,
 * FILL_MAX_HITS or more when the reference's give-up rule (simulations.py:478-479) has to decide */
DSD int fill_count_bucket(const FillParams &p, const V3 &pt, int begin, int end, int llx, int ulx) {
  const MeshDev &ms = p.mesh;
  const V3 ray = {1.0, 0.0, 0.0};
  const float fy = (float)pt.y, fz = (float)pt.z;
  int n_hits = 0;
  for (int i = begin; i < end; i++) {
    const int2 en = p.ment[i];
    const int x = en.y >> 16, prev = (en.y & 0xffff) - 1;
    if (x < llx) break; /* descending x: the rest of the bucket lies behind the point */
    if (x >= ulx) continue;
    const float4 yz = p.ment_yz[i];
    if (fy < yz.x || fy > yz.z || fz < yz.y || fz > yz.w) continue;
    if (prev >= llx) continue; /* the same triangle is listed at a nearer x of this ray as well: counted there */
    const double *rec = ms.tri + (int64_t)en.x * DS_TRI_STRIDE;
    double d = ray_triangle_intersection_check(load_v3(rec), load_v3(rec + 3), load_v3(rec + 6), pt, ray);
    if (d > 0) n_hits++;
  }
  return n_hits;
}

/* the launch parameters of the fill kernel read afresh from the kernarg segment (see fresh_params: FillParams is the
 * kernel's first and only argument) */
DSD const FillParams &fresh_fill(const FillParams &p) {
  auto k = __builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(k));
  (void)p;
  return *(const FillParams *)k;
}

/* candidates that have a bucket to walk, waiting for a full trip of the wavefront (fill_batch_kernel) */
#define FILL_QUEUE_CAP 128
struct FillQueue {
  double x[FILL_QUEUE_CAP], y[FILL_QUEUE_CAP], z[FILL_QUEUE_CAP];
  uint32_t key[FILL_QUEUE_CAP]; /* launch << 8 | thread inside the workgroup */
  int32_t llx[FILL_QUEUE_CAP], ulx[FILL_QUEUE_CAP], col[FILL_QUEUE_CAP]; /* col = lly << 16 | llz */
  int32_t begin[FILL_QUEUE_CAP], end[FILL_QUEUE_CAP];                     /* the bucket's entries */
};

/* A BATCH of the reference's launches in one dispatch.  The reference launches _cuda_fill_mesh again and again - every
 * thread draws one candidate per launch from its own RNG state (simulations.py:560-579, a fresh all-inf `points` array
 * each time) - and appends the accepted points of a launch in thread order.  Thread t only ever needs state t, so it
 * runs its part of `n_launches` launches back to back with the state in registers (round 3: one dispatch per launch
 * that read and wrote 32 bytes of state per thread, then a library scan and a scatter - 16 000 dispatches for 1e7
 * points inside the neuron).
 *
 * Most candidates are decided at once - their (y, z) column of the index holds no triangle at all (the neuron fills
 * 0.025 % of its voxel) - while one in a few has a bucket of triangles to walk, a loop of dependent look-ups whose
 * length differs from lane to lane.  Left in place that loop runs on nearly every launch for the few lanes that need
 * it; here those candidates are QUEUED per wavefront (point, key, x range, column, bucket: 48 bytes in LDS, slots from a
 * ballot) and walked 64 at a time, whoever drew them.  An accepted candidate is parked in the workgroup's staging rows
 * with its (launch, thread) key; fill_scan_kernel and fill_place_kernel then put the batch in (launch, thread) order -
 * the order the reference's appends produce.  Candidates, counts and order are the reference's bit for bit
 * (tests/test_gpu_parity.py, the fill tests; two ranks: tests/test_gpu_two_ranks.py). */
__global__ __launch_bounds__(256) void fill_batch_kernel(const FillParams p_arg) {
  extern __shared__ double lds[];
  __shared__ int s_parked, s_cnt[FILL_MAX_BATCH];
  __shared__ FillQueue s_queue[4];
  const FillParams &p = fresh_fill(p_arg);
  const MeshDev &ms = p.mesh;
  const int n_bounds = ms.nx + ms.ny + ms.nz + 3;
  for (int i = threadIdx.x; i < n_bounds; i += blockDim.x) lds[i] = ms.bounds[i];
  if (threadIdx.x < FILL_MAX_BATCH) s_cnt[threadIdx.x] = 0;
  if (threadIdx.x == 0) s_parked = 0;
  __syncthreads();
  const double *xs = lds, *ys = lds + ms.nx + 1, *zs = lds + ms.nx + ms.ny + 2;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = tid < p.n_points;
  const int lane = (int)(threadIdx.x & 63);
  FillQueue &Q = s_queue[threadIdx.x >> 6];
  FillEntry *stage = p.stage + (int64_t)blockIdx.x * p.stage_cap;
  Rng rng = {1, 1};
  if (live) rng.s0 = p.s0[tid], rng.s1 = p.s1[tid];
  auto park = [&](const V3 &pt, uint32_t key) {
    atomicAdd(&s_cnt[key >> 8], 1);
    const int slot = atomicAdd(&s_parked, 1);
    if (slot < p.stage_cap) {
      FillEntry e = {pt.x, pt.y, pt.z, key, 0u};
      stage[slot] = e;
    } else {
      *p.overflow = 1;
    }
  };
  /* one trip over the queued candidates [base, base + count), count <= 64: every lane walks one bucket */
  auto walk_trip = [&](int base, int count) {
    if (lane < count) {
      const int q = base + lane;
      const V3 pt = {Q.x[q], Q.y[q], Q.z[q]};
      const FillParams &p = fresh_fill(p_arg);
      const int lly = Q.col[q] >> 16, llz = Q.col[q] & 0xffff, llx = Q.llx[q], ulx = Q.ulx[q];
      int n_hits = fill_count_bucket(p, pt, Q.begin[q], Q.end[q], llx, ulx);
      if (n_hits >= FILL_MAX_HITS) n_hits = fill_count_reference_order(p, pt, llx, ulx, lly, lly + 1, llz, llz + 1);
      const bool inside = (n_hits & 1) == 1;
      if (n_hits >= 0 && (p.intra ? inside : !inside)) park(pt, Q.key[q]);
    }
  };
  int n_q = 0; /* candidates waiting in the queue (wave-uniform) */
  for (int l = 0; l < fresh_fill(p_arg).n_launches; l++) {
    const FillParams &p = fresh_fill(p_arg); /* re-read per launch: ~70 dwords of parameters do not stay in SGPRs */
    const MeshDev &ms = p.mesh;
    bool need = false, accept = false;
    V3 pt = {0, 0, 0};
    int llx = 0, ulx = 0, col = 0;
    int2 bucket = make_int2(0, 0);
    if (live) {
      pt.x = rng_uniform_f64(rng) * p.voxel[0];
      pt.y = rng_uniform_f64(rng) * p.voxel[1];
      pt.z = rng_uniform_f64(rng) * p.voxel[2];
      /* non-periodic lookups (:455-460): ll on min(p, p+ray) = p, ul on max = p+ray (ray = +x) */
      llx = ll_subvoxel_overlap(xs, ms.nx, ms.inv_hx, pt.x);
      const int lly = ll_subvoxel_overlap(ys, ms.ny, ms.inv_hy, pt.y);
      const int llz = ll_subvoxel_overlap(zs, ms.nz, ms.inv_hz, pt.z);
      ulx = ul_subvoxel_overlap(xs, ms.nx, ms.inv_hx, pt.x + 1.0);
      const int uly = ul_subvoxel_overlap(ys, ms.ny, ms.inv_hy, pt.y);
      const int ulz = ul_subvoxel_overlap(zs, ms.nz, ms.inv_hz, pt.z);
      /* The y and z ranges of a +x ray hold ONE subvoxel each, or none when the point sits exactly on a boundary
       * (ul <= ll + 1 for a degenerate interval): one column of the index, or no triangle at all. */
      int n_hits = 0;
      if (uly - lly == 1 && ulz - llz == 1) {
        col = (lly << 16) | llz; /* n_sv <= 4096 per axis */
        /* the point's bucket of its column holds triangles: it is walked with the next full trip (an empty bucket - four
         * in five of a column that a dendrite crosses - decides the candidate here and now: no triangle, outside) */
        bucket = fill_bucket(p, ys, zs, pt, lly, llz);
        need = bucket.y > bucket.x;
#ifdef DSIM_FILL_ABLATE /* development build: DSIM_FILL_ABLATE=1 no bucket is walked (timing of the draws and look-ups alone), 2: nothing but the draws */
        if (DSIM_FILL_ABLATE >= 1) need = false;
#endif
      } else if (uly > lly && ulz > llz) {
        n_hits = fill_count_reference_order(p, pt, llx, ulx, lly, uly, llz, ulz); /* cannot happen (see above); the reference's loop decides */
      }
      if (!need) accept = n_hits >= 0 && (p.intra ? (n_hits & 1) == 1 : (n_hits & 1) == 0);
    }
    if (accept) park(pt, ((uint32_t)l << 8) | threadIdx.x);
    const unsigned long long m = __ballot(need);
    if (m != 0ULL) {
      if (need) {
        const int q = n_q + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        Q.x[q] = pt.x, Q.y[q] = pt.y, Q.z[q] = pt.z;
        Q.key[q] = ((uint32_t)l << 8) | threadIdx.x;
        Q.llx[q] = llx, Q.ulx[q] = ulx, Q.col[q] = col;
        Q.begin[q] = bucket.x, Q.end[q] = bucket.y;
      }
      n_q += (int)__popcll(m);
      wave_sync();
      if (n_q >= 64) {
        n_q -= 64;
        walk_trip(n_q, 64);
        wave_sync(); /* the trip's slots may be written again by the next launch */
      }
    }
  }
  if (n_q > 0) walk_trip(0, n_q);
  if (live) {
    p.s0[tid] = rng.s0;
    p.s1[tid] = rng.s1;
  }
  __syncthreads();
  if ((int)threadIdx.x < p.n_launches) p.cnt[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = s_cnt[threadIdx.x];
  if (threadIdx.x == 0) p.stage_count[blockIdx.x] = min(s_parked, p.stage_cap);
}

/* workgroup l: exclusive prefix sum of cnt[l][0 .. n_blocks) in place (what the workgroups before b accepted in launch
 * l) and the launch's total; one workgroup of 256 threads walks the row in tiles of 4096 (as sort_scan_kernel does) */
__global__ __launch_bounds__(256) void fill_scan_kernel(int32_t *cnt, int n_blocks, int64_t *totals) {
  __shared__ uint32_t wsum[4];
  int32_t *row = cnt + (int64_t)blockIdx.x * n_blocks;
  const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
  uint32_t carry = 0;
  for (int base = 0; base < n_blocks; base += 16 * 256) {
    const int b0 = base + 16 * t;
    uint32_t v[16], mine = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) {
      v[q] = b0 + q < n_blocks ? (uint32_t)row[b0 + q] : 0u;
      mine += v[q];
    }
    const uint32_t incl = (uint32_t)wave_inclusive_sum((int)mine);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t before = carry, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const uint32_t sw = wsum[w];
      total += sw;
      if (w < wave) before += sw;
    }
    uint32_t run = before + incl - mine;
#pragma unroll
    for (int q = 0; q < 16; q++) {
      if (b0 + q < n_blocks) row[b0 + q] = (int32_t)run;
      run += v[q];
    }
    carry += total;
    __syncthreads();
  }
  if (t == 0) totals[blockIdx.x] = (int64_t)carry;
}

/* workgroup b: its parked candidates go to their rows.  Row of the candidate of thread t in launch l =
 * rows before the batch + what launches < l accepted + what the workgroups before b accepted in launch l (pref) + the
 * accepted threads below t in this workgroup and launch (a 256-bit map per launch in LDS).  Rows >= capacity are
 * dropped (the caller sizes its batches so that none it needs is). */
__global__ __launch_bounds__(256) void fill_place_kernel(const FillEntry *stage, const int32_t *stage_count, int stage_cap, const int32_t *pref,
                                                         const int64_t *totals, int n_launches, int64_t rows_before, int64_t capacity,
                                                         double *out) {
  __shared__ unsigned long long bits[FILL_MAX_BATCH][4];
  __shared__ long long base[FILL_MAX_BATCH];
  const int t = (int)threadIdx.x;
  if (t < FILL_MAX_BATCH) bits[t][0] = bits[t][1] = bits[t][2] = bits[t][3] = 0ULL;
  if (t < n_launches) {
    long long before = rows_before;
    for (int l = 0; l < t; l++) before += totals[l];
    base[t] = before + pref[(int64_t)t * gridDim.x + blockIdx.x];
  }
  __syncthreads();
  const int n_e = stage_count[blockIdx.x];
  const FillEntry *mine = stage + (int64_t)blockIdx.x * stage_cap;
  for (int e = t; e < n_e; e += 256) {
    const uint32_t key = mine[e].key;
    atomicOr(&bits[key >> 8][(key & 255u) >> 6], 1ULL << (key & 63u));
  }
  __syncthreads();
  for (int e = t; e < n_e; e += 256) {
    const FillEntry en = mine[e];
    const int l = (int)(en.key >> 8), th = (int)(en.key & 255u), w = th >> 6;
    int rank = __popcll(bits[l][w] & ((1ULL << (th & 63)) - 1ULL));
    for (int q = 0; q < w; q++) rank += __popcll(bits[l][q]);
    const long long row = base[l] + rank;
    if (row < capacity) {
      out[row * 3] = en.x;
      out[row * 3 + 1] = en.y;
      out[row * 3 + 2] = en.z;
    }
  }
}

#endif /* DSIM_KERNELS_H */
