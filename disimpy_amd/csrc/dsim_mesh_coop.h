/* dsim_mesh_coop.h - wavefront-cooperative traversal of the mesh index.
 *
 * Per-lane traversal (step_mesh in dsim_kernels.h) pays, at every loop level,
 * for the slowest of the 64 lanes: measured on the neuron (tools/mesh_stats.py)
 * a wavefront makes 47 record-batch trips per step with 8.6 lanes active and 45
 * Moller-Trumbore trips with 2.2 lanes active.  Here ONE pass of the reference's
 * reflect loop (simulations.py:924-997) is done for the 64 walkers of a
 * wavefront together, as flat loops over work queues that ALL lanes drain:
 *
 *   stage 1   every walker ("owner"): subvoxel limits, float32 cull box and -
 *             by arithmetic alone, the fine grid has the same k for the whole
 *             mesh - the fine cells its segment touches, as RUNS of cells along
 *             z (the records of a run are contiguous).  The first run goes to
 *             the queue slot of the owner's own lane, further runs behind;
 *   stage 1c  any lane per run: the record range from the cell table (one level:
 *             a slot block for every subvoxel), cut into items of COOP_ITEM_RECS
 *             records (slots from a wave prefix sum);
 *   stage 2   any lane per item: its records tested (fixed-point box + float32 plane)
 *             against the OWNER's cull box -> survivor pool, slots from wave
 *             ballots;
 *   stage 3   any lane per survivor: fp64 Moller-Trumbore test with the OWNER's
 *             ray, minimum per owner by 64-bit LDS atomics on the bit pattern
 *             of d; a second atomic pass names the winner's triangle and
 *             notices when two DIFFERENT triangles share the minimum;
 *   stage 4   the owner fetches the unit normal of a hit within the step:
 *             reflection / crossing per walker, then the next pass for the
 *             walkers that hit something.
 *
 * Stages 1c-3 run in rounds (runs are expanded while the item queue has room,
 * items are drained in full trips of 64, the survivor pool is flushed in full
 * trips of 64), so the queue sizes bound the LDS footprint, not the work per
 * pass.  The owners' segments (midpoint, half vector and margin: 7 floats, from
 * which the working lane rebuilds the cull box) and rays stay in the owners'
 * registers and are read with ds_bpermute.
 *
 * Exactness.  Which (triangle, periodic image) pairs reach the fp64 test is a
 * subset of what the reference tests (only list entries of subvoxels inside the
 * segment's range are looked at, through the cells of THAT subvoxel) and a
 * superset of the pairs that can return 0 < d <= step_l (culling is
 * conservative with margins far above the rounding of the test).  The minimum
 * over them is therefore the reference's min_d whenever that is <= step_l, and
 * "beyond step_l" otherwise.  WHICH triangle attains it is unambiguous unless
 * two different triangles return the same d bit for bit (a ray through a shared
 * edge or vertex); the reference then keeps the one its loops meet first
 * (strict <, simulations.py:981).  The queues do not preserve that order, so a
 * tie is only DETECTED here and the pass of that one walker is redone by its own
 * lane exactly as the reference does it (iteration_serial) - as is the work of
 * a walker that does not fit the queues (more than 64 runs, image shifts beyond
 * +-31 voxels, run queue full, a single run with more than 1024 records, more
 * than 128 survivors from one trip).  tests/test_gpu_mesh_adversarial.py aims
 * rays at exactly these cases.
 */
#ifndef DSIM_MESH_COOP_H
#define DSIM_MESH_COOP_H

/* Wave priorities (s_setprio): the SIMD's arbiter prefers the wavefront with the higher priority when several are
 * ready to issue.  A wavefront runs the stages that look things up (1c: sv_base -> cell_start, 2: records, 3: triangles -
 * chains of dependent loads whose latency is what the other wavefronts have to cover) at priority 1 and everything that is
 * pure arithmetic - the random step and the loop head in walk_body, the fp64 Moller-Trumbore tests, stage 4 - at
 * priority 0: a wavefront about to issue loads gets there first, the arithmetic of the others fills the gaps.  Measured
 * on the same box (profiles/history/r03_ab_a18_a21_wave_priorities.txt): C5b +1.8 %, C4 +2.4 %; the stage-2
 * arithmetic at low priority as well, or only the look-ups of 1c at high priority, lose most of it again. */
#define COOP_PRIO_LOOKUPS() __builtin_amdgcn_s_setprio(1)
#define COOP_PRIO_ARITHMETIC() __builtin_amdgcn_s_setprio(0)

#define COOP_ITEM_CAP 320
/* records per item (stage 2 tests them in one trip).  [r5] 5 records ([r6]: 16 bytes + the 16 bytes of the triangle's plane: 40 VGPRs; four 48-byte records took
 * 44): a pass of the refined neuron then needs 1.92 items per walker instead of 2.31 (the mean run holds 8 records) and
 * its items are two trips of stage 2 more often than three.  Same-box A/B (profiles/history/r05_ab_mesh_kernel_steps.txt): C5b
 * +3.7 % at 1e7 and at 1.25e6 walkers, C4 +0.4 %; 6 records per item need 5 spilled VGPRs at four wavefronts per SIMD and
 * lose what they gain (0 / -1 % / -4.6 %).  -DDSIM_ITEM_RECS=4 / 6 / 8 for comparison (at most 8: three bits of an item
 * hold the count) */
#ifndef DSIM_ITEM_RECS
#define DSIM_ITEM_RECS 5
#endif
#define COOP_ITEM_RECS DSIM_ITEM_RECS
#define COOP_ITEM_FIRST_MASK 0x1fffffffu /* 29 bits of first record, 3 bits of count - 1 */
#define COOP_SURV_CAP 128
#define COOP_RUNQ_CAP 320 /* slots 0..63: the first run of owner 0..63; the other runs behind */
#define COOP_MAX_RUNS 64  /* per owner and pass */
#define COOP_NO_SHIFT 0x20820u /* (tag >> 6) & 0x3ffff when all three image numbers are 0 (stored + 32, 6 bits each) */

/* Per-wavefront scratch in LDS (7.5 KB).  The owners' cull boxes and rays are NOT here: they
 * stay in the owners' registers and the lane that works on an item reads them with ds_bpermute
 * (__shfl), which costs LDS bandwidth but no LDS space. */
struct CoopWave {
  unsigned long long best_d[64];   /* smallest positive d (bit pattern orders like the value)                 0.5 KB */
  uint2 hit[64];                   /* {a triangle that attains it (the smallest index among those seen), flags: byte 0 = some work of this
                                    * owner did not fit a queue (redo serially), byte 1 = two different triangles share its smallest d} 0.5 KB */
  uint2 first[64];                 /* every owner's first run {owner | shifts << 6 | (cells - 1) << 24, subvoxel * k^3 + first cell} */
  unsigned more[COOP_RUNQ_CAP - 64]; /* the other runs, by name only: owner | number of the run << 6              1.5 KB */
  int4 plan_a[64];                 /* what the runs of an owner are numbered over (RunPlan): a0, b0, c0, c1 ...          */
  int2 plan_b[64];                 /*   ... first z unit, columns in y | z units << 8 | n0a << 16 | n0b << 24      1.5 KB */
  uint2 items[COOP_ITEM_CAP];      /* items {first record | (count - 1) << 29, owner | shifts << 6} (1c -> 2)  2.5 KB */
  unsigned surv_tri[COOP_SURV_CAP]; /* survivors (2 -> 3): triangle,                                                  */
  unsigned surv_tag[COOP_SURV_CAP]; /*   owner | shifts << 6 (directly behind surv_tri: stage 2 writes both through one pointer) 1.0 KB */
  int pad[4];
#ifdef DSIM_TIMERS
  unsigned long long tacc[16], tlast;
  unsigned long long tpad;
#endif
};

#define COOP_BEST_TRI(W, i) ((W)->hit[i].x)
#define COOP_OVERFLOW(W, i) (reinterpret_cast<unsigned char *>(&(W)->hit[i].y)[0])
#define COOP_TIE(W, i) (reinterpret_cast<unsigned char *>(&(W)->hit[i].y)[1])

static_assert(offsetof(CoopWave, surv_tag) == offsetof(CoopWave, surv_tri) + COOP_SURV_CAP * sizeof(unsigned),
              "stage 2 writes a survivor's triangle and tag through one pointer");

/* inclusive prefix sum over the 64 lanes with DPP moves (VALU only, no LDS crossbar): Kogge-Stone
 * inside every row of 16 lanes, then the row totals are carried into the following rows */
DSD int wave_inclusive_sum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); /* row_shr:1 */
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); /* row_shr:2 */
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); /* row_shr:4 */
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); /* row_shr:8 */
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); /* row_bcast:15 into rows 1 and 3 */
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); /* row_bcast:31 into rows 2 and 3 */
  return v;
}

/* the wave's mask of a condition.  HIP's __ballot takes an int: the condition is first turned into a 0 / 1 per lane and
 * compared with 0 again (two vector instructions); the builtin takes the condition as it is - one s_and with exec */
DSD unsigned long long wave_ballot(bool c) { return __builtin_amdgcn_ballot_w64(c); }

DSD double shfl_f64(double v, int src) {
  const long long b = __double_as_longlong(v);
  const int lo = __shfl((int)(b & 0xffffffffLL), src, 64), hi = __shfl((int)(b >> 32), src, 64);
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}

DSD void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* segment moved into the frame of a periodic image (shift subtracted from its midpoint), margin grown by
 * the float32 rounding of the subtraction; shift = 0 returns the box unchanged */
DSD CullBox shift_cull_box(const CullBox &c, double shx, double shy, double shz) {
  if (shx == 0 && shy == 0 && shz == 0) return c;
  const float sx = (float)shx, sy = (float)shy, sz = (float)shz;
  CullBox o = c;
  o.mhx.x = c.mhx.x - sx;
  o.mhy.x = c.mhy.x - sy;
  o.mhz.x = c.mhz.x - sz;
  /* rounding of the shifts and of the three subtractions: a few float32 ulps of the larger operand each */
  o.tau = c.tau + 2e-6f * (__builtin_fabsf(sx) + __builtin_fabsf(sy) + __builtin_fabsf(sz) + __builtin_fabsf(c.mhx.x) +
                           __builtin_fabsf(c.mhy.x) + __builtin_fabsf(c.mhz.x));
  box_from_segment(o);
  return o;
}

/* the fine grid coordinates of a pass, any periodic image, both ends inclusive, already clamped into the first / last
 * subvoxel of the reference's range: along z cells (subvoxel * k + cell); along x and y - where the cells of a k > 1
 * grid overlap, dsim_mesh_host.h - half-cell UNITS (subvoxel * 2k + unit; k == 1: the subvoxels themselves) */
struct FineRange {
  int a0, a1, b0, b1, c0, c1;
};

/* Both periodic subvoxel limits of one axis in float32, for the common case that decides itself: the boundaries
 * are np.linspace(0, L, n + 1), image k of boundary i sits at (k n + i) h, so for a coordinate that is NOT
 * within 1e-4 subvoxel widths of any boundary the reference's scans (simulations.py:616-679) end at
 * ll = floor(t_min), ul = floor(t_max) + 1 with t = x / h - whatever arithmetic produced t, as long as its
 * error stays below that distance.  t is the fp64 product rounded once to float32 (relative error 6e-8).  The
 * caller accepts when `edge` < 0.4999 on every axis (both fractional parts farther than 1e-4 from 0 and 1) and
 * the sum of all six |t| is below 1500 (so every |t| < 1500 and its error < 9e-5); otherwise - too close to a
 * boundary, too far out, or not a number - it takes the exact fp64 path.  tmin / tmax (subvoxel units) also
 * place the segment on the fine grid. */
DSD void subvoxel_limits_f32(double x0, double inv_h, double x1, double x2, int &ll, int &ul, float &edge, float &reach, float &tmin,
                             float &tmax) {
  const float t1 = (float)((x1 - x0) * inv_h), t2 = (float)((x2 - x0) * inv_h);
  tmin = __builtin_fminf(t1, t2), tmax = __builtin_fmaxf(t1, t2);
  const float fmin = __builtin_floorf(tmin), fmax = __builtin_floorf(tmax);
  ll = (int)fmin;
  ul = (int)fmax + 1;
  /* distance of the fractional parts from 1/2: below 0.4999 = farther than 1e-4 from both neighbouring boundaries */
  edge = __builtin_fmaxf(__builtin_fabsf((tmin - fmin) - 0.5f), __builtin_fabsf((tmax - fmax) - 0.5f));
  reach = __builtin_fabsf(t1) + __builtin_fabsf(t2); /* an addition: NaN survives it (min / max would drop it) */
}
DSD void subvoxel_limits_f32(double x0, double inv_h, double x1, double x2, int &ll, int &ul, float &edge, float &reach) {
  float tmin, tmax;
  subvoxel_limits_f32(x0, inv_h, x1, x2, ll, ul, edge, reach, tmin, tmax);
}

/* The fine cells of one axis: the segment's extent [tmin, tmax] in subvoxel units times k, widened by far more than
 * the error of t (half a float32 ulp of t, i.e. <= 6e-8 `reach` subvoxels) and of the index's own cell assignment
 * (micro_cell() in fp64; cells registered with an inflation of 1e-6), then clamped into the first and the last
 * subvoxel of the reference's range [ll, ul): together with "border cells extend to infinity" of the index this
 * visits, in EVERY subvoxel of the range, the cells of the segment's box clamped into that subvoxel.  Any superset of
 * the exact cells is valid (an extra cell only costs its records a look). */
DSD void fine_limits(float tmin, float tmax, float reach, int ll, int ul, int lk, int &f0, int &f1) {
  const int k = 1 << lk;
  const float kf = (float)k;
  const float mk = __builtin_fmaf(reach, 1.2e-7f * kf, __builtin_fmaf(1e-4f, kf, 1e-3f));
  const int lo = ll * k, hi = (ul - 1) * k;
  /* float -> int conversions saturate and send NaN to 0; the clamps make any value safe */
  f0 = min(max((int)__builtin_floorf(__builtin_fmaf(tmin, kf, -mk)), lo), lo + k - 1);
  f1 = min(max((int)__builtin_floorf(__builtin_fmaf(tmax, kf, mk)), hi), hi + k - 1);
  f1 = max(f1, f0); /* only ever changes a value that NaN input produced inside a single subvoxel */
}

/* subvoxel ranges (simulations.py:929-934), cull box and fine cells of the remaining segment */
DSD void segment_query(const MeshDev &ms, const V3 &r0, const V3 &step, double step_l, const double *xs, const double *ys,
                       const double *zs, SubvoxelWalk &sw, CullBox &cb0, FineRange &fr) {
  const double ex = __builtin_fma(step.x, step_l, r0.x);
  const double ey = __builtin_fma(step.y, step_l, r0.y);
  const double ez = __builtin_fma(step.z, step_l, r0.z);
  cb0 = make_cull_box(r0, step, step_l);
  float e0, e1, e2, q0, q1, q2, tn0, tx0, tn1, tx1, tn2, tx2;
  subvoxel_limits_f32(ms.x0, ms.inv_hx, r0.x, ex, sw.llx, sw.ulx, e0, q0, tn0, tx0);
  subvoxel_limits_f32(ms.y0, ms.inv_hy, r0.y, ey, sw.lly, sw.uly, e1, q1, tn1, tx1);
  subvoxel_limits_f32(ms.z0, ms.inv_hz, r0.z, ez, sw.llz, sw.ulz, e2, q2, tn2, tx2);
  /* decided on all three axes: 99.9 % of the segments (NaN fails both comparisons) */
  if (!(__builtin_fmaxf(__builtin_fmaxf(e0, e1), e2) < 0.4999f && (q0 + q1) + q2 < 1500.0f)) {
    const bool fx = subvoxel_limits_fast(xs, ms.nx, ms.x0, ms.x1, ms.inv_hx, ms.inv_lx, r0.x, ex, sw.llx, sw.ulx);
    const bool fy = subvoxel_limits_fast(ys, ms.ny, ms.y0, ms.y1, ms.inv_hy, ms.inv_ly, r0.y, ey, sw.lly, sw.uly);
    const bool fz = subvoxel_limits_fast(zs, ms.nz, ms.z0, ms.z1, ms.inv_hz, ms.inv_lz, r0.z, ez, sw.llz, sw.ulz);
    if (!(fx && fy && fz)) {
      sw.llx = ll_subvoxel_overlap_periodic(xs, ms.nx, ms.inv_hx, ms.inv_lx, r0.x, ex);
      sw.lly = ll_subvoxel_overlap_periodic(ys, ms.ny, ms.inv_hy, ms.inv_ly, r0.y, ey);
      sw.llz = ll_subvoxel_overlap_periodic(zs, ms.nz, ms.inv_hz, ms.inv_lz, r0.z, ez);
      sw.ulx = ul_subvoxel_overlap_periodic(xs, ms.nx, ms.inv_hx, ms.inv_lx, r0.x, ex);
      sw.uly = ul_subvoxel_overlap_periodic(ys, ms.ny, ms.inv_hy, ms.inv_ly, r0.y, ey);
      sw.ulz = ul_subvoxel_overlap_periodic(zs, ms.nz, ms.inv_hz, ms.inv_lz, r0.z, ez);
    }
  }
  const int lu = ms.lk > 0 ? ms.lk + 1 : 0; /* x, y: 2k units per subvoxel */
  fine_limits(tn0, tx0, q0, sw.llx, sw.ulx, lu, fr.a0, fr.a1);
  fine_limits(tn1, tx1, q1, sw.lly, sw.uly, lu, fr.b0, fr.b1);
  fine_limits(tn2, tx2, q2, sw.llz, sw.ulz, ms.lk, fr.c0, fr.c1);
}

/* wrapped subvoxel coordinate and image number of an absolute subvoxel coordinate: sn = floor(s / n) (Python: math.floor,
 * simulations.py:661).  [r6] By one float32 multiplication instead of an integer division (~20 instructions, four of them
 * quarter-rate 32-bit multiplies; three divisions per run on the packed cylinders, whose walkers are nearly all in a
 * periodic image along z after a few hundred steps - positions are never wrapped, like the reference's): (s + 1/2) / n is
 * at least 1 / (2 n) >= 1.2e-4 away from every integer (n <= 4096, upload_mesh), |s| <= 33 n wherever the result is used
 * (plan_runs sends anything beyond +-32 voxels to the serial pass), so s + 1/2 is exact and the product is off by less than
 * 33 x 2^-22 = 8e-6: the floor is the exact one.  inv_n = 1.0f / n from the host (MeshDev). */
DSD void wrap_subvoxel(int s, int n, float inv_n, int &x, int &sn) {
  x = s, sn = 0;
  if (s < 0 || s > n - 1) {
    sn = (int)__builtin_floorf(((float)s + 0.5f) * inv_n);
    x = s - __mul24(sn, n);
  }
}

/* The runs of a pass.  A run is a stretch of fine cells along z whose records are contiguous in mrec: inside one
 * subvoxel (lk > 0: its k cells of one (x, y) column of cells) or, when the subvoxels themselves are the cells (lk == 0),
 * inside one periodic image of the z axis (cell_start is indexed by the subvoxel, z-fastest).  The "z unit" is that
 * stretch - a subvoxel or an image.  The runs of an owner are numbered j = (ia * nb + ib) * nc + iz over its na x nb
 * columns and nc z units; any lane can work out run j from the plan alone.
 *
 * Columns along x (y alike), lk > 0: the box covers the units a0 .. a1 (2k units per subvoxel) and the cells are two
 * units wide, one at every unit but the last of a subvoxel.  In the first subvoxel the columns are the cells at
 * a0, a0 + 2, ... (n0a of them; a cell that would start at the subvoxel's last unit is the one before it), every
 * further subvoxel of the range is entered at its unit 0 with the cells 0, 2, 4, ...  A box narrower than a unit - the
 * rule: units are at least 0.75 step lengths wide - covers at most two units: ONE column unless it straddles a subvoxel. */
struct RunPlan {
  int a0, b0, c0, c1; /* first unit along x and y, fine z range */
  int zu0;            /* first z unit */
  int nb, nc;         /* columns along y, z units */
  int n0a, n0b;       /* lk > 0: columns inside the first subvoxel along x, y */
};

/* columns of the unit range [u0, u1] on an axis with 1 << lu units per subvoxel, and how many lie in the first subvoxel */
DSD int plan_columns(int u0, int u1, int lu, int &n_first) {
  const int next = ((u0 >> lu) + 1) << lu; /* first unit of the following subvoxel */
  n_first = ((min(u1, next - 1) - u0) >> 1) + 1;
  return n_first + (u1 >= next ? ((u1 - next) >> 1) + 1 : 0);
}

/* number of runs of the pass (0: the reference's range is empty, nothing is visited), or -1 when the walker cannot be
 * queued: image shifts must fit 6 bits each, a run holds at most 16 cells, and no more than COOP_MAX_RUNS runs.
 * Written without early returns: every field of the plan is computed whatever the outcome (integer arithmetic that is safe
 * on any input), and the outcome is a chain of selects - a return in the middle makes the compiler materialise a default
 * for every field on every path (some forty v_mov per pass, [r5]) */
DSD int plan_runs(const MeshDev &ms, const SubvoxelWalk &w, const FineRange &f, RunPlan &rp) {
  const bool empty = (w.ulx <= w.llx) | (w.uly <= w.lly) | (w.ulz <= w.llz);
  const int nx = ms.nx, ny = ms.ny, nz = ms.nz;
  bool bad = (w.llx < -32 * nx) | (w.ulx > 32 * nx) | (w.lly < -32 * ny) | (w.uly > 32 * ny) | (w.llz < -32 * nz) | (w.ulz > 32 * nz);
  int zu1, na;
  if (ms.lk > 0) { /* wave-uniform */
    rp.zu0 = w.llz, zu1 = w.ulz - 1;
    na = plan_columns(f.a0, f.a1, ms.lk + 1, rp.n0a);
    rp.nb = plan_columns(f.b0, f.b1, ms.lk + 1, rp.n0b);
  } else {
    bad |= w.ulz - w.llz > 16;
    int z;
    wrap_subvoxel(w.llz, nz, ms.inv_nz, z, rp.zu0);
    wrap_subvoxel(w.ulz - 1, nz, ms.inv_nz, z, zu1);
    na = f.a1 - f.a0 + 1;
    rp.nb = f.b1 - f.b0 + 1;
    rp.n0a = rp.n0b = 0;
  }
  rp.a0 = f.a0, rp.b0 = f.b0, rp.c0 = f.c0, rp.c1 = f.c1;
  rp.nc = zu1 - rp.zu0 + 1;
  /* each factor is checked on its own, so the product of three accepted factors cannot overflow */
  /* (24-bit multiplies - full rate; v_mul_lo_u32 runs at a quarter of it: a factor beyond 2^24 is rejected by its own check) */
  const unsigned n = __umul24(__umul24((unsigned)na, (unsigned)rp.nb), (unsigned)rp.nc);
  bad |= ((unsigned)na > (unsigned)COOP_MAX_RUNS) | ((unsigned)rp.nb > (unsigned)COOP_MAX_RUNS) | ((unsigned)rp.nc > (unsigned)COOP_MAX_RUNS) |
         (n > (unsigned)COOP_MAX_RUNS);
  return empty ? 0 : (bad ? -1 : (int)n);
}

/* column i of an axis (plan_columns): the subvoxel (absolute, any image) and the cell inside it */
DSD void plan_column(int u0, int n_first, int i, int lk, int &sv, int &cell) {
  const int lu = lk + 1, um = (1 << lu) - 1;
  if (i < n_first) {
    const int g = u0 + 2 * i;
    sv = g >> lu;
    cell = min(g & um, um - 1);
  } else {
    const int r = i - n_first;
    sv = (u0 >> lu) + 1 + (r >> lk);
    cell = 2 * (r & ((1 << lk) - 1));
  }
}

/* run j of an owner's plan as a queue entry {owner | shifts << 6 | (cells - 1) << 24, subvoxel * cell slots + first cell} */
DSD uint2 make_run(const MeshDev &ms, int owner, const RunPlan &rp, int j) {
  const int lk = ms.lk, k = 1 << lk, km = k - 1;
  int ia = 0, ib = 0, iz = j;
  if (j > 0) { /* j < 64: (j + 1/2) / n is at least 1/128 away from every integer, the reciprocal is good to 1e-6 */
    const int bc = __mul24(rp.nb, rp.nc);
    ia = (int)(((float)j + 0.5f) * __builtin_amdgcn_rcpf((float)bc));
    const int rem = j - __mul24(ia, bc);
    ib = (int)(((float)rem + 0.5f) * __builtin_amdgcn_rcpf((float)rp.nc));
    iz = rem - __mul24(ib, rp.nc);
  }
  const int zu = rp.zu0 + iz;
  int sa, sb, ca = 0, cb = 0;
  if (lk > 0) {
    plan_column(rp.a0, rp.n0a, ia, lk, sa, ca);
    plan_column(rp.b0, rp.n0b, ib, lk, sb, cb);
  } else {
    sa = rp.a0 + ia, sb = rp.b0 + ib;
  }
  int x, snx, y, sny, z, snz, cs, ce;
  wrap_subvoxel(sa, ms.nx, ms.inv_nx, x, snx);
  wrap_subvoxel(sb, ms.ny, ms.inv_ny, y, sny);
  if (lk > 0) {
    wrap_subvoxel(zu, ms.nz, ms.inv_nz, z, snz);
    cs = max(rp.c0, zu * k), ce = min(rp.c1, zu * k + km);
  } else {
    snz = zu;
    const int lo = __mul24(zu, ms.nz);
    cs = max(rp.c0, lo), ce = min(rp.c1, lo + ms.nz - 1);
    z = cs - lo;
  }
  /* x < nx, y < ny <= 4096 each and fewer than 2^31 subvoxels in all: 24-bit multiplies (full rate) give the exact index */
  const unsigned sv = __umul24(__umul24((unsigned)x, (unsigned)ms.ny) + (unsigned)y, (unsigned)ms.nz) + (unsigned)z;
  /* cell slots of a subvoxel: (2k)^2 k, column-major with z fastest (lk == 0: the subvoxel is the cell) */
  const unsigned cell = lk > 0 ? (unsigned)(((((ca << (lk + 1)) | cb) << lk) | (cs & km))) : 0u;
  return make_uint2((unsigned)owner | ((unsigned)(snx + 32) << 6) | ((unsigned)(sny + 32) << 12) | ((unsigned)(snz + 32) << 18) |
                        ((unsigned)(ce - cs) << 24),
                    lk > 0 ? (sv << (3 * lk + 2)) + cell : sv);
}

/* Stage 3: one trip of fp64 tests over the survivors [base, base + count) of the pool, count <= 64:
 * Moller-Trumbore test with the owner's ray, minimum per owner by 64-bit LDS atomics on the bit
 * pattern of d; a second atomic pass records a triangle that attains the minimum and flags the owner
 * when two different triangles do.  Runs several times per pass (a full trip whenever the pool holds 64
 * survivors, the remainder at the end); seen_d is best_d[lane] as of the previous call.  Every
 * lane of the wavefront must call this; the caller's fill level of the pool becomes `base`. */
DSD void flush_survivors(const MeshDev &ms, const V3 &r0, const V3 &step, CoopWave *W, int lane, unsigned long long &seen_d,
                         int base, int count) {
  const double Lx = ms.x1, Ly = ms.y1, Lz = ms.z1;
  const int idx = base + lane;
  const bool valid = lane < count;
  unsigned tri = 0u, tag = 0u;
  if (valid) tri = W->surv_tri[idx], tag = W->surv_tag[idx];
  /* the triangle first: its address only needs the survivor, so the nine loads are in flight while the owner's ray
   * comes through the shuffles (a lane without a survivor reads triangle 0) */
  /* buffer loads: the triangle by its 32-bit byte offset (96 bytes each: 2^25 triangles at most, DevMesh::coop_ok) */
  const __amdgpu_buffer_rsrc_t tri_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)ms.tri, 0, -1, 0x00020000);
  /* an UNSIGNED 32-bit byte offset: 2^25 triangles x 96 bytes pass 2^31 (a signed product would overflow from 22 369 622 triangles on) */
  const int toff = (int)(tri * (unsigned)(DS_TRI_STRIDE * 8));
  typedef double ds_double2 __attribute__((ext_vector_type(2)));
  const ds_uint4 t0 = __builtin_amdgcn_raw_buffer_load_b128(tri_rsrc, toff, 0, 0), t1 = __builtin_amdgcn_raw_buffer_load_b128(tri_rsrc, toff + 16, 0, 0);
  const ds_uint4 t2 = __builtin_amdgcn_raw_buffer_load_b128(tri_rsrc, toff + 32, 0, 0), t3 = __builtin_amdgcn_raw_buffer_load_b128(tri_rsrc, toff + 48, 0, 0);
  const uint2 t4 = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(tri_rsrc, toff + 64, 0, 0));
  const ds_double2 d0 = __builtin_bit_cast(ds_double2, t0), d1 = __builtin_bit_cast(ds_double2, t1), d2 = __builtin_bit_cast(ds_double2, t2),
                   d3 = __builtin_bit_cast(ds_double2, t3);
  const V3 tA = {d0.x, d0.y, d1.x}, tE1 = {d1.y, d2.x, d2.y}, tE2 = {d3.x, d3.y, __builtin_bit_cast(double, t4)};
  const int owner = (int)(tag & 63u);
  V3 tr0 = {shfl_f64(r0.x, owner), shfl_f64(r0.y, owner), shfl_f64(r0.z, owner)};
  const V3 st = {shfl_f64(step.x, owner), shfl_f64(step.y, owner), shfl_f64(step.z, owner)};
  double d = 0;
  COOP_PRIO_ARITHMETIC(); /* the loads and shuffles are issued: the test itself is arithmetic */
  if (valid) { /* pass A: distances, atomic minimum of d */
    STAT_LANE(6);
    STAT_WAVE(6);
    if ((tag >> 6) != COOP_NO_SHIFT) { /* x - 0.0 == x bit for bit, so the unshifted case skips the arithmetic */
      const int snx = (int)((tag >> 6) & 63u) - 32, sny = (int)((tag >> 12) & 63u) - 32, snz = (int)((tag >> 18) & 63u) - 32;
      tr0.x -= (double)snx * Lx, tr0.y -= (double)sny * Ly, tr0.z -= (double)snz * Lz;
    }
    d = ray_triangle_intersection_check(tA, tE1, tE2, tr0, st);
    if (d > 0) atomicMin(&W->best_d[owner], (unsigned long long)__double_as_longlong(d));
  }
  COOP_PRIO_LOOKUPS();
  wave_sync();
  T_MARK(W, lane, 5);
  { /* a smaller minimum than at the last flush: what was known about the old one is void */
    const unsigned long long cur = W->best_d[lane];
    if (cur != seen_d) {
      COOP_BEST_TRI(W, lane) = 0xffffffffu;
      COOP_TIE(W, lane) = 0;
      seen_d = cur;
    }
  }
  wave_sync();
  if (valid && d > 0 && (unsigned long long)__double_as_longlong(d) == W->best_d[owner]) {
    /* pass B: the same triangle again (another cell, another subvoxel or image) is no tie - it IS the winner */
    const unsigned old = atomicMin(&COOP_BEST_TRI(W, owner), tri);
    if (old != 0xffffffffu && old != tri) COOP_TIE(W, owner) = 1;
  }
  wave_sync();
  T_MARK(W, lane, 6);
}

/* Is the walker farther than one step length from every triangle (clearance map, MeshDev)?  Then the
 * first pass of its reflect loop cannot find a hit (`min_d > step_l`, simulations.py:986) and the whole
 * traversal is skipped.  Two dependent 4-byte look-ups; called before the random step is drawn so that
 * their latency hides behind it. */
DSD bool walker_is_clear(const MeshDev &ms, const V3 &r) {
  if (ms.clear_idx == nullptr) return false;
  const int lf = ms.clear_lf, m = (1 << lf) - 1;
  const int ix = clearance_cell(r.x - ms.x0, ms.inv_lx, ms.nx << lf);
  const int iy = clearance_cell(r.y - ms.y0, ms.inv_ly, ms.ny << lf);
  const int iz = clearance_cell(r.z - ms.z0, ms.inv_lz, ms.nz << lf);
  const int blk = ms.clear_idx[__umul24(__umul24((unsigned)(ix >> lf), (unsigned)ms.ny) + (unsigned)(iy >> lf), (unsigned)ms.nz) + (unsigned)(iz >> lf)];
  if (blk < 0) return true; /* nothing within reach of the whole subvoxel */
  const unsigned bit = (unsigned)(((((ix & m) << lf) | (iy & m)) << lf) | (iz & m));
  const int lw = lf >= 2 ? 3 * lf - 5 : 0; /* log2 of the dwords per block */
  const uint32_t w = ms.clear_bits[((size_t)blk << lw) + (bit >> 5)];
  return ((w >> (bit & 31u)) & 1u) == 0u;
}

/* the one-level table of a run's records: lk > 0 the dense cell table (a slot block for every subvoxel; nullptr when it was not
 * built), lk == 0 cell_start itself - indexed by the subvoxel, [r6] read through the same paired buffer loads (entries named by
 * 32-bit byte offsets: fewer than 2^30 subvoxels) */
DSD const uint32_t *cell_table(const MeshDev &ms) {
  return ms.lk > 0 ? ms.cell_dense : ((unsigned)(ms.nx * ms.ny * ms.nz) < (1u << 30) ? ms.cell_start : nullptr);
}

/* ONE pass of the reflect loop of simulations.py:924-997 for the 64 walkers of a wavefront
 * together.  EVERY lane of the wavefront must call this; in_step marks the lanes whose walker
 * is inside a time step.  Returns true for the lanes whose step is over after this pass (no
 * hit within the remaining length, or the loop condition `step_l > 0 and iter_idx < max_iter`
 * no longer holds).  The caller lets every lane run through its own time steps at its own pace
 * (a wavefront then needs about mean-iterations-per-step passes per step, not the maximum over
 * its 64 lanes). */
/* [r6] need_norm (out): this lane's walker was reflected and goes on - its new direction is NOT yet normalised; the caller
 * does that at the head of the next pass, in the same instructions that normalise the directions of the walkers that start
 * a step there (the two sets are disjoint: the normalisation ran twice per pass, at 44 and at 15 of 64 lanes, now once). */
#ifndef DSIM_DEFER_NORM
#define DSIM_DEFER_NORM 1
#endif
#ifndef DSIM_LOOKUPS_TOGETHER
#define DSIM_LOOKUPS_TOGETHER 1
#endif
template <int KA>
DSD bool coop_iteration(V3 &r0, V3 &step, double &step_l, int64_t &iter_idx, bool in_step, bool clear, const WalkParams &p_arg,
                        const double *xs, const double *ys, const double *zs, Rng &rng, CoopWave *W, int lane, bool &need_norm) {
  const WalkParams &p = fresh_params<KA>(p_arg);
  /* (plain boolean algebra, no branches: the conditions stay lane masks on the scalar unit)
   * looping: the while-condition holds on entry.  skip: no triangle within the step length (clearance map) - this pass of the
   * loop finds min_d > step_l whatever the direction (simulations.py:986-987): counted, but not traversed */
  const bool looping = in_step & (step_l > 0) & (iter_idx < p.max_iter);
  const bool skip = looping & clear;
  const bool act = looping & !clear;
  bool finished = in_step & !act; /* while-condition false on entry, or skipped */
  iter_idx += skip ? 1 : 0;
  if (wave_ballot(act) == 0ULL) return finished;
  COOP_PRIO_LOOKUPS();
  W->best_d[lane] = 0x7ff0000000000000ULL; /* +inf */
  W->hit[lane] = make_uint2(0xffffffffu, 0u); /* no triangle, no flag: one 8-byte store */
  T_MARK(W, lane, 1);
#ifdef DSIM_TIMERS
  if (lane == 0) W->tacc[10] += 1;
#endif
  bool serial = false;
  /* ---- stage 1: owners work out their cull box and list the runs of fine cells their segment touches.  Box and
   * ray stay in the owner's registers (cb0, r0, step: unchanged until stage 4); the other lanes read
   * them with __shfl, so every lane has to be active wherever a shuffle is executed. */
  /* segment, box and plan of a lane that is NOT in the pass are never used (no run, no item, no survivor names it as the
   * owner; a lane without an item reads lane 0's and throws the result away): their registers are left as they are
   * ("set", for the compiler, by an asm that emits nothing; zeroing them was 19 v_mov per pass) */
  CullBox cb0;
  QBox qb0;
  RunPlan rp;
  asm volatile("" : "=v"(cb0.lox), "=v"(cb0.loy), "=v"(cb0.loz), "=v"(cb0.hix), "=v"(cb0.hiy), "=v"(cb0.hiz), "=v"(cb0.mhx), "=v"(cb0.mhy),
               "=v"(cb0.mhz), "=v"(cb0.tau), "=v"(qb0.b0), "=v"(qb0.b1), "=v"(qb0.b2));
  asm volatile("" : "=v"(rp.a0), "=v"(rp.b0), "=v"(rp.c0), "=v"(rp.c1), "=v"(rp.zu0), "=v"(rp.nb), "=v"(rp.nc), "=v"(rp.n0a), "=v"(rp.n0b));
  int n_runs = 0;
  const MeshDev &ms = p.mesh;
  if (act) {
    iter_idx++;
    STAT_LANE(0);
    STAT_WAVE(0);
    SubvoxelWalk sw;
    FineRange fr;
    segment_query(ms, r0, step, step_l, xs, ys, zs, sw, cb0, fr);
    qb0 = quantise_box(cb0, ms);
    n_runs = plan_runs(ms, sw, fr, rp);
    serial = n_runs < 0;
    n_runs = max(n_runs, 0);
#ifdef DSIM_STATS
    if (n_runs > 0) atomicAdd(&dsim_stats[1], (unsigned long long)((sw.ulx - sw.llx) * (sw.uly - sw.lly) * (sw.ulz - sw.llz)));
    atomicAdd(&dsim_stats[2], (unsigned long long)n_runs);
#endif
  }
  T_MARK(W, lane, 2);
  int n_runq; /* runs queued, the 64 first-run slots included (wave-uniform) */
  {
    /* the first run goes, complete, into the slot of the owner's own lane; the others are only NAMED in the queue
     * (a loop of one 4-byte store per run) and worked out from the owner's plan by whichever lane takes them in
     * stage 1c - an owner with many runs does not hold up its wavefront */
    const int extra_n = max(n_runs - 1, 0);
    const int incl = wave_inclusive_sum(extra_n);
    const int extra = incl - extra_n;
    n_runq = 64 + __builtin_amdgcn_readlane(incl, 63);
    if (extra_n > 0 && extra + extra_n > COOP_RUNQ_CAP - 64) {
      /* queue full: neutralise the part of the reservation that lies inside the queue */
      for (int j = extra; j < COOP_RUNQ_CAP - 64; j++) W->more[j] = 0xffffffffu;
      serial = true;
      n_runs = 0;
    }
    uint2 e0 = make_uint2(0xffffffffu, 0u);
    if (n_runs > 0) {
      e0 = make_run(ms, lane, rp, 0);
      if (n_runs > 1)
      {
        W->plan_a[lane] = make_int4(rp.a0, rp.b0, rp.c0, rp.c1);
        W->plan_b[lane] = make_int2(rp.zu0, rp.nb | (rp.nc << 8) | (rp.n0a << 16) | (rp.n0b << 24));
      }
      for (int j = 1; j < n_runs; j++) W->more[extra + j - 1] = (unsigned)lane | ((unsigned)j << 6);
    }
    W->first[lane] = e0;
    n_runq = min(n_runq, COOP_RUNQ_CAP);
  }
  T_MARK(W, lane, 12);
  wave_sync();
  /* ---- stages 1c, 2, 3 in rounds: the runs are cut into items of COOP_ITEM_RECS records until the item queue
   * is nearly full, the items are drained into the survivor pool, the pool is flushed (fp64 tests,
   * per-owner minimum) whenever it holds 64 - so no queue size limits the amount of work per pass */
  {
    const MeshDev &ms = fresh_params<KA>(p_arg).mesh;
    const double Lx = ms.x1, Ly = ms.y1, Lz = ms.z1; /* = xs[nx], ys[ny], zs[nz] */
    const int lk = ms.lk;
    const int lcs = 3 * lk + 2;                 /* log2 of the cell slots per subvoxel, lk > 0 */
    const unsigned cmask = (1u << lcs) - 1u;
    const uint32_t *const dense = cell_table(ms);
    const __amdgpu_buffer_rsrc_t rec_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)ms.mrec, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t plane_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)ms.planes, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t dense_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)dense, 0, -1, 0x00020000);
    unsigned long long seen_d = 0x7ff0000000000000ULL; /* best_d[lane] at the last flush */
    /* two cursors: over the 64 first-run slots and over the named extra runs behind them.  A trip of stage 1c gives
     * every lane one of each - the extras are few (a box that straddles a subvoxel face: one walker in ten), so a trip
     * of their own would run at a tenth of its width; riding along, they cost the decoding of their names in a trip
     * that has to run anyway */
    int cur_a = 0, cur_b = 0;
    const int n_extra = n_runq - 64;
    int n_items = 0;        /* items waiting in the queue (wave-uniform) */
    int n_surv = 0;         /* survivors waiting in the pool (wave-uniform) */
    bool drain_all = false; /* the next run does not fit behind the waiting items */
    /* records of a run: {first record, number of records} */
    auto look_up = [&](const uint2 &e) -> int2 {
      int first = 0, cnt = 0;
      if (e.x != 0xffffffffu) {
        STAT_LANE(5);
        STAT_WAVE(5);
        if (dense != nullptr) { /* one level: the run's slots in the table that has a block for every subvoxel (wave-uniform branch);
                                 * buffer loads by 32-bit byte offset - the table has fewer than 2^30 slots (upload_mesh_host) */
          const unsigned r0i = __builtin_amdgcn_raw_buffer_load_b32(dense_rsrc, (int)(e.y << 2), 0, 0);
          const unsigned r1i = __builtin_amdgcn_raw_buffer_load_b32(dense_rsrc, (int)((e.y + ((e.x >> 24) & 15u) + 1u) << 2), 0, 0);
          first = (int)r0i;
          cnt = (int)(r1i - r0i);
        } else {
          int cell = (int)e.y;
          bool some = true;
          if (lk > 0) {
            const int base = ms.sv_base[e.y >> lcs];
            some = base >= 0;
            cell = base + (int)(e.y & cmask);
          }
          if (some && !ABLATE(2)) {
            const unsigned r0i = ms.cell_start[cell], r1i = ms.cell_start[cell + (int)((e.x >> 24) & 15u) + 1];
            first = (int)r0i;
            cnt = (int)(r1i - r0i);
          }
        }
      }
      return make_int2(first, cnt);
    };
    /* the records of a run as items of COOP_ITEM_RECS; the last one holds what is left (its count goes in after the loop: the loop
     * itself is one store and two additions per item - its trip count is the longest run of the 64) */
    auto write_items = [&](uint2 *q, int first, int cnt, int nb, unsigned tag) {
      unsigned x = (unsigned)first | ((unsigned)(COOP_ITEM_RECS - 1) << 29);
      for (uint2 *const end = q + nb - 1; q < end; q++, x += (unsigned)COOP_ITEM_RECS) *q = make_uint2(x, tag);
      *q = make_uint2((x & COOP_ITEM_FIRST_MASK) | ((unsigned)(cnt - COOP_ITEM_RECS * (nb - 1) - 1) << 29), tag);
    };
    /* items of a run of n records (n <= 1024: longer runs go to their owner's serial pass - COOP_ITEM_CAP) */
    auto n_items_of = [](int n) -> int {
      if (COOP_ITEM_RECS == 4) return (n + 3) >> 2;
      if (COOP_ITEM_RECS == 8) return (n + 7) >> 3;
      return (int)(__umul24((unsigned)(n + COOP_ITEM_RECS - 1), (unsigned)(65536 / COOP_ITEM_RECS + 1)) >> 16); /* exact below 2^14 */
    };
    while (cur_a < 64 || cur_b < n_extra) {
      /* -- 1c: up to 64 first runs and 64 extra runs per trip, one of each per lane */
      while ((cur_a < 64 || cur_b < n_extra) && n_items <= COOP_ITEM_CAP - 64) {
        uint2 ea = make_uint2(0xffffffffu, 0u), eb = make_uint2(0xffffffffu, 0u);
        if (cur_a + lane < 64) ea = W->first[cur_a + lane];
        const bool extras = cur_b < n_extra; /* wave-uniform */
        if (extras && cur_b + lane < n_extra) {
          const unsigned m = W->more[cur_b + lane];
          if (m != 0xffffffffu) {
            const int o = (int)(m & 63u);
            const int4 pa = W->plan_a[o];
            const int2 pb = W->plan_b[o];
            const RunPlan q = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y & 255, (pb.y >> 8) & 255, (pb.y >> 16) & 255, (pb.y >> 24) & 255};
            eb = make_run(ms, o, q, (int)(m >> 6));
          }
        }
        int2 ra, rb = make_int2(0, 0);
#if DSIM_LOOKUPS_TOGETHER
        if (dense != nullptr) { /* (wave-uniform) [r6] the slots of both runs asked for before either answer is waited for, and
                                 * without a branch per lane: a lane without a run reads slot 0 twice - an empty range */
          const bool va = ea.x != 0xffffffffu, vb = eb.x != 0xffffffffu;
          const unsigned a0 = va ? ea.y : 0u, a1 = va ? ea.y + ((ea.x >> 24) & 15u) + 1u : 0u;
          const unsigned r0a = __builtin_amdgcn_raw_buffer_load_b32(dense_rsrc, (int)(a0 << 2), 0, 0);
          const unsigned r1a = __builtin_amdgcn_raw_buffer_load_b32(dense_rsrc, (int)(a1 << 2), 0, 0);
          if (extras) {
            const unsigned b0 = vb ? eb.y : 0u, b1 = vb ? eb.y + ((eb.x >> 24) & 15u) + 1u : 0u;
            const unsigned r0b = __builtin_amdgcn_raw_buffer_load_b32(dense_rsrc, (int)(b0 << 2), 0, 0);
            const unsigned r1b = __builtin_amdgcn_raw_buffer_load_b32(dense_rsrc, (int)(b1 << 2), 0, 0);
            rb = make_int2((int)r0b, (int)(r1b - r0b));
          }
          ra = make_int2((int)r0a, (int)(r1a - r0a));
#ifdef DSIM_STATS
          if (va) { STAT_LANE(5); STAT_WAVE(5); }
          if (extras && vb) { STAT_LANE(5); STAT_WAVE(5); }
#endif
        } else
#endif
        {
          ra = look_up(ea);
          if (extras) rb = look_up(eb);
        }
        const int nba = n_items_of(min(ra.y, 8192)), nbb = n_items_of(min(rb.y, 8192)), nb = nba + nbb;
        if (nb > 0) {
          STAT_LANE(3);
          STAT_WAVE(3);
        }
        const int incl = wave_inclusive_sum(nb);
        const unsigned long long fits = wave_ballot(n_items + incl <= COOP_ITEM_CAP);
        const int n_take = fits == ~0ULL ? 64 : (int)__ffsll((long long)~fits) - 1; /* the lanes that fit form a prefix */
        if (n_take == 0) {
          if (n_items > 0) { /* drain first, then retry with an empty queue */
            drain_all = true;
            break;
          }
          /* the runs of lane 0 alone hold more records than the whole queue: their owners work serially */
          if (lane == 0) {
            if (ea.x != 0xffffffffu) COOP_OVERFLOW(W, (int)(ea.x & 63u)) = 1;
            if (eb.x != 0xffffffffu) COOP_OVERFLOW(W, (int)(eb.x & 63u)) = 1;
          }
          cur_a = min(64, cur_a + 1);
          cur_b = min(n_extra, cur_b + 1);
          continue;
        }
        if (lane < n_take) {
          uint2 *q = &W->items[n_items + incl - nb];
          if (nba > 0) write_items(q, ra.x, ra.y, nba, ea.x & 0xffffffu);
          if (nbb > 0) write_items(q + nba, rb.x, rb.y, nbb, eb.x & 0xffffffu);
        }
        n_items += __builtin_amdgcn_readlane(incl, n_take - 1);
        cur_a = min(64, cur_a + n_take);
        cur_b = min(n_extra, cur_b + n_take);
        if (n_take < 64) break;
      }
      T_MARK(W, lane, 13);
      /* -- 2: all lanes drain the item queue: fixed-point box + float32 plane culling against the owner's segment.
       * While more runs are coming only full trips of 64 items are drained (the tail of the queue);
       * the remainder waits at the head for the next round */
      const int start = ((cur_a >= 64 && cur_b >= n_extra) || drain_all) ? 0 : (n_items & 63);
      drain_all = false;
      if (start == n_items) continue;
      wave_sync();
      for (int first = start; first < n_items; first += 64) {
        const int idx = first + lane;
        const bool valid = idx < n_items;
        uint2 it = make_uint2(0u, 0u);
        if (valid) it = W->items[idx];
        /* the records first: their addresses only need the item, so the 8 loads are in flight while the owner's
         * segment comes through the shuffles (a lane without an item reads record 0) */
        /* a lane without an item: count 0, no record of it counts */
        const int first_rec = valid ? (int)(it.x & COOP_ITEM_FIRST_MASK) : 0;
        int cnt = (valid && !ABLATE(3) && !ABLATE(4)) ? (int)(it.x >> 29) + 1 : 0;
        asm volatile("" : "+v"(cnt)); /* (opaque: `cnt > u` below stays ONE comparison instead of valid & (count > u)) */
        /* buffer loads: the index as a scalar resource, a record named by its 32-bit byte offset (one shift per record
         * instead of a 64-bit address in two VGPRs; 28 bits of record number x 16 bytes: the 2^28 records upload_mesh admits).
         * [r6] The five 16-byte records of the item from one address register at immediate offsets, then - as they
         * arrive - the planes of the triangles they name from the table by triangle */
        ds_uint4 q0[COOP_ITEM_RECS], q1[COOP_ITEM_RECS];
        const int rec_off = (ABLATE(4) ? 0 : first_rec) << 4; /* 4: every lane reads the first records of the index */
#pragma unroll
        for (int u = 0; u < COOP_ITEM_RECS; u++) q0[u] = __builtin_amdgcn_raw_buffer_load_b128(rec_rsrc, rec_off + u * 16, 0, 0);
#pragma unroll
        for (int u = 0; u < COOP_ITEM_RECS; u++) q1[u] = __builtin_amdgcn_raw_buffer_load_b128(plane_rsrc, (int)(q0[u].w << 4), 0, 0);
        const int owner = (int)(it.y & 63u);
        /* the owner's segment (plane test: midpoint and half vector as the pairs {m, h} per axis) and its box in the
         * records' fixed point: ten dwords; the float32 box itself is only rebuilt for a periodic image */
        ds_float2 mh_x = {__shfl(cb0.mhx.x, owner, 64), __shfl(cb0.mhx.y, owner, 64)};
        ds_float2 mh_y = {__shfl(cb0.mhy.x, owner, 64), __shfl(cb0.mhy.y, owner, 64)};
        ds_float2 mh_z = {__shfl(cb0.mhz.x, owner, 64), __shfl(cb0.mhz.y, owner, 64)};
        float tau = __shfl(cb0.tau, owner, 64);
        unsigned b0 = (unsigned)__shfl((int)qb0.b0, owner, 64), b1 = (unsigned)__shfl((int)qb0.b1, owner, 64);
        unsigned b2 = (unsigned)__shfl((int)qb0.b2, owner, 64);
        if (valid) {
          STAT_LANE(4);
          STAT_WAVE(4);
        }
        if (valid && (it.y >> 6) != COOP_NO_SHIFT) { /* periodic image: move the box into its frame */
          const int snx = (int)((it.y >> 6) & 63u) - 32, sny = (int)((it.y >> 12) & 63u) - 32, snz = (int)((it.y >> 18) & 63u) - 32;
          CullBox cb;
          cb.mhx = mh_x, cb.mhy = mh_y, cb.mhz = mh_z, cb.tau = tau;
          box_from_segment(cb);
          cb = shift_cull_box(cb, (double)snx * Lx, (double)sny * Ly, (double)snz * Lz);
          const QBox qb = quantise_box(cb, ms);
          mh_x.x = cb.mhx.x, mh_y.x = cb.mhy.x, mh_z.x = cb.mhz.x, tau = cb.tau;
          b0 = qb.b0, b1 = qb.b1, b2 = qb.b2;
        }
        /* the tests without a branch between or around them (with 64 lanes x 5 records somebody always passes
         * the box, so a branch only costs mask juggling); a lane without an item tests record 0 for nothing.  Slots beyond
         * the item's count hold the records that follow in the index (the loads stay unconditional and share one address
         * register; the index ends with COOP_ITEM_RECS - 1 records of padding, upload_mesh_host) and do not count.  The results never become
         * a per-lane word: a comparison IS the wave's ballot of it */
        bool alive[COOP_ITEM_RECS];
        unsigned long long mask[COOP_ITEM_RECS], any = 0ULL;
#pragma unroll
        for (int u = 0; u < COOP_ITEM_RECS; u++) {
          bool box_ok, plane_ok;
          record_tests(q0[u], q1[u], b0, b1, b2, mh_x, mh_y, mh_z, tau, box_ok, plane_ok);
          /* the mask of every comparison on its own, combined on the scalar unit (the mask of a combined condition would
           * first be turned into a 0 / 1 per lane and compared again: two vector instructions per record) */
          mask[u] = wave_ballot(cnt > u) & wave_ballot(box_ok) & wave_ballot(plane_ok);
          alive[u] = __builtin_amdgcn_inverse_ballot_w64(mask[u]); /* the mask itself selects the lanes that write */
          any |= mask[u];
        }
        /* the survivors' slots from the ballots (survivors in lower lanes first): the pool's fill level stays in a
         * scalar register, there is no LDS atomic to wait for and no counter to read back before the flush check
         * (round 2 / early round 3: one LDS atomic per lane and an LDS read per trip; C5b +1.4 %, C4 +1.7 %) */
        if (any != 0ULL) {
          unsigned below = 0u;
          int total = 0;
#pragma unroll
          for (int u = 0; u < COOP_ITEM_RECS; u++) {
            below = __builtin_amdgcn_mbcnt_hi((unsigned)(mask[u] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask[u], below));
            total += (int)__popcll(mask[u]);
          }
          int slot = n_surv + (int)below;
          if (n_surv + total <= COOP_SURV_CAP) { /* wave-uniform, and all but always: nobody has to look at its slot */
            unsigned *st = &W->surv_tri[slot];
#pragma unroll
            for (int u = 0; u < COOP_ITEM_RECS; u++) {
              if (!alive[u]) continue;
              st[0] = q0[u].w;
              st[COOP_SURV_CAP] = it.y; /* surv_tag follows surv_tri */
              st++;
            }
          } else {
#pragma unroll
            for (int u = 0; u < COOP_ITEM_RECS; u++) {
              if (!alive[u]) continue;
              if (slot < COOP_SURV_CAP) {
                W->surv_tri[slot] = q0[u].w;
                W->surv_tag[slot] = it.y;
              } else
                COOP_OVERFLOW(W, owner) = 1;
              slot++;
            }
          }
          n_surv = min(n_surv + total, COOP_SURV_CAP);
        }
        T_MARK(W, lane, 4);
        wave_sync();
        for (; n_surv >= 64; n_surv -= 64) flush_survivors(ms, r0, step, W, lane, seen_d, n_surv - 64, 64);
      }
      n_items = start;
    }
    if (n_surv > 0) flush_survivors(ms, r0, step, W, lane, seen_d, 0, n_surv);
  }
  /* ---- stage 4: resolve (simulations.py:986-997) */
  COOP_PRIO_ARITHMETIC();
  if (act) {
    const WalkParams &p = fresh_params<KA>(p_arg);
    const MeshDev &ms = p.mesh;
    /* straight-line: the minimum as it stands (its initial pattern IS +inf: nothing hit), the triangle that goes with it, and
     * whether the walker has to go through the pass on its own - queues overflowed, or two triangles at the same distance
     * within the step (the reference's visiting order decides between them) */
    double d = __longlong_as_double((long long)W->best_d[lane]);
    const uint2 hv = W->hit[lane]; /* triangle and flags: one 8-byte load */
    int tri = (int)hv.x;
    const bool redo = serial | ((hv.y & 0xffu) != 0u) | ((d <= step_l) & ((hv.y & 0xff00u) != 0u));
    if (redo) {
      /* this walker's whole iteration again, by its own lane, exactly as the reference does it */
      STAT_LANE(9);
      STAT_WAVE(9);
      BestHit best = {DS_INF, 0};
      SubvoxelWalk sw; /* recomputed rather than kept in registers through the drain stages */
      FineRange fr;
      segment_query(ms, r0, step, step_l, xs, ys, zs, sw, cb0, fr);
      iteration_serial(ms, sw, r0, step, xs, ys, zs, best);
      d = best.d;
      tri = best.tri;
    }
    T_MARK(W, lane, 7);
    if (d > step_l) {
      finished = true; /* check_intersection = False */
    } else {
      V3 n = load_v3(ms.tri + (int64_t)tri * DS_TRI_STRIDE + 9);
      /* perm_prob < u on the integer bits of the draw (ds_perm_threshold): a shift and a comparison instead of two conversions,
       * a multiplication, an fma and a comparison - the same decision for every draw */
      const bool reflect = (int64_t)(rng_next(rng) >> 11) > p.perm_thresh;
      if (reflect)
        reflection<!DSIM_DEFER_NORM>(r0, step, d, n, p.eps);
      else
        crossing(r0, step, d, n, p.eps);
      step_l -= d;
      if (!(step_l > 0 && iter_idx < p.max_iter)) finished = true; /* the while-condition of the next pass */
      if (DSIM_DEFER_NORM && reflect) {
        if (finished) /* (rare: the hit used the step up, or max_iter) the caller moves the walker along `step` right away */
          normalize3(step);
        else
          need_norm = true;
      }
    }
  }
  T_MARK(W, lane, 8);
  return finished;
}

#endif /* DSIM_MESH_COOP_H */
