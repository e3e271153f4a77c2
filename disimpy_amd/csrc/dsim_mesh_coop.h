/* dsim_mesh_coop.h - wavefront-cooperative traversal of the mesh index.
 *
 * Per-lane traversal (step_mesh in dsim_kernels.h) pays, at every loop level,
 * for the slowest of the 64 lanes: measured on the neuron (tools/mesh_stats.py)
 * a wavefront makes 47 record-batch trips per step with 8.6 lanes active and 45
 * Moller-Trumbore trips with 2.2 lanes active.  Here ONE pass of the reference's
 * reflect loop (simulations.py:924-997) is done for the 64 walkers of a
 * wavefront together, as flat loops over work queues that ALL lanes drain:
 *
 *   stage 1a  every walker ("owner"): subvoxel limits, float32 cull box, one
 *             task per subvoxel in range (reference order = visit number);
 *   stage 1b  any lane per subvoxel task: sv_micro look-up, the micro-cells the
 *             owner's box reaches -> cell tasks;
 *   stage 1c  any lane per cell task: cell_range look-up, the record range cut
 *             into 4-record items (slots from a wave prefix sum);
 *   stage 2   any lane per item: 4 records tested (float32 AABB + plane)
 *             against the OWNER's cull box -> survivor pool;
 *   stage 3   any lane per survivor: fp64 Moller-Trumbore test with the OWNER's
 *             ray, minimum per owner by 64-bit LDS atomics on the bit pattern
 *             of d, ties resolved by a second atomic pass whose key also names
 *             the winner's triangle;
 *   stage 4   the owner fetches the unit normal of a hit within the step:
 *             reflection / crossing per walker, then the next pass for the
 *             walkers that hit something.
 *
 * Stages 1c-3 run in rounds (cells are expanded while the item queue has room,
 * items are drained in full trips of 64, the survivor pool is flushed in full
 * trips of 64), so the queue sizes bound the LDS footprint, not the work per
 * pass.  The owners' segments (midpoint, half vector and margin: 7 floats, from which the
 * working lane rebuilds the cull box) and rays stay in the owners' registers and
 * are read with ds_bpermute; the kernel then fits 128 VGPRs and 7 KB of LDS per
 * wavefront, i.e. 4 wavefronts per SIMD.
 *
 * Which triangles are tested, in fp64, against which ray is unchanged, and the
 * minimum keeps the reference's "first visited wins a tie" rule through the
 * explicit key (subvoxel visit number, position in the reference's list), so
 * results stay bit-identical to step_mesh and to the oracle.  A walker whose
 * work cannot be queued (more than 256 subvoxels in range, image shifts beyond
 * +-31 voxels, cell-task queue full, a single cell with more than 1024 records,
 * more than 128 survivors from one trip) is redone by its own lane exactly as
 * the reference does it (iteration_serial).
 */
#ifndef DSIM_MESH_COOP_H
#define DSIM_MESH_COOP_H

#define COOP_ITEM_CAP 256
#define COOP_SURV_CAP 128
#define COOP_SVQ_CAP 448
#define COOP_CELLQ_CAP 256
#define COOP_NO_SHIFT 0x20820u /* tag >> 14 when all three image numbers are 0 (stored + 32, 6 bits each) */

/* Per-wavefront scratch in LDS (6.6 KB).  The owners' cull boxes and rays are NOT here: they
 * stay in the owners' registers and the lane that works on an item reads them with ds_bpermute
 * (__shfl), which costs LDS bandwidth but no LDS space. */
struct CoopWave {
  unsigned long long best_d[64];        /* smallest positive d (bit pattern orders like the value)      0.5 KB */
  unsigned long long best_key[64];      /* visit << 48 | list position << 24 | triangle of the winner   0.5 KB */
  uint2 cellq[COOP_CELLQ_CAP];          /* cell tasks {owner|visit|shifts, index into cell_range  or
                                           0x80000000 | subvoxel (range stored inline)} (1b -> 1c)      2.0 KB */
  /* the next four arrays (3.5 KB) also hold, between stages 1a and 1b, the subvoxel tasks
   * {owner|visit|shifts, x | y << 10 | z << 20} */
  uint2 items[COOP_ITEM_CAP];           /* items {first record | (count-1) << 30, owner|visit|shifts} (1c -> 2) */
  unsigned surv_tri[COOP_SURV_CAP];     /* survivors (2 -> 3): triangle,                                       */
  unsigned surv_pos[COOP_SURV_CAP];     /*   position in the reference's list,                                 */
  unsigned surv_tag[COOP_SURV_CAP];     /*   owner|visit|shifts                                                */
  unsigned char overflow[64];           /* some work of this owner did not fit a queue: redo serially          */
  int n_surv, n_svq, pad[2];
#ifdef DSIM_STATS
  unsigned filt[64][4]; /* development: (owner, list position) pairs already among the survivors of this pass */
#endif
#ifdef DSIM_TIMERS
  unsigned long long tacc[16], tlast;
  unsigned long long tpad;
#endif
};
static_assert(COOP_SVQ_CAP * 8 <= COOP_ITEM_CAP * 8 + COOP_SURV_CAP * 12, "subvoxel tasks share the item and survivor queues");

DSD uint2 *coop_items(CoopWave *W) { return W->items; }
DSD uint2 *coop_svq(CoopWave *W) { return W->items; }
DSD uint2 *coop_cellq(CoopWave *W) { return W->cellq; }

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
  o.mx = c.mx - sx;
  o.my = c.my - sy;
  o.mz = c.mz - sz;
  /* rounding of the shifts and of the three subtractions: a few float32 ulps of the larger operand each */
  o.tau = c.tau + 2e-6f * (__builtin_fabsf(sx) + __builtin_fabsf(sy) + __builtin_fabsf(sz) + __builtin_fabsf(c.mx) +
                           __builtin_fabsf(c.my) + __builtin_fabsf(c.mz));
  box_from_segment(o);
  return o;
}

/* (visit, position) of the current best candidate; a candidate with an equal d wins only
 * if it comes earlier in the reference's visiting order (simulations.py:937-983) */
struct BestHit {
  double d;
  int visit, pos, tri;
  bool have_n; /* n holds E1 x E2 of tri (published by the lane that tested it) */
  V3 n;
};

DSD void offer_hit(BestHit &b, double d, int visit, int pos, int tri) {
  if (d > 0 && (d < b.d || (d == b.d && (visit < b.visit || (visit == b.visit && pos < b.pos))))) {
    b.d = d;
    b.visit = visit;
    b.pos = pos;
    b.tri = tri;
    b.have_n = false;
  }
}

/* > 0 when the record can be skipped - ONE number, no branches or mask arithmetic: the max over (a) the six box
 * gaps (a positive gap on any axis separates the triangle's box from the segment's) and (b) how far beyond the
 * margin tau the whole segment lies on one side of the triangle's plane: with the midpoint m and the half vector
 * h of the segment, the end points evaluate to (n.m - c) -+ n.h, so both are beyond tau on the same side iff
 * |n.m - c| - |n.h| > tau.  Every rounding / flush-to-zero / NaN case falls on the "keep" side or concerns a
 * walker whose fp64 test cannot hit anyway. */
DSD float cull_margin(const float4 &q0, const float4 &q1, const float4 &q2, const CullBox &cb) {
  const float g1 = __builtin_fmaxf(__builtin_fmaxf(q0.x - cb.hix, q0.y - cb.hiy), q0.z - cb.hiz);
  const float g2 = __builtin_fmaxf(__builtin_fmaxf(cb.lox - q0.w, cb.loy - q1.x), cb.loz - q1.y);
  const float sm = __builtin_fmaf(q2.x, cb.mz, __builtin_fmaf(q1.z, cb.mx, __builtin_fmaf(q1.w, cb.my, -q2.y)));
  const float sh = __builtin_fmaf(q2.x, cb.hz, __builtin_fmaf(q1.z, cb.hx, q1.w * cb.hy));
  const float side = (__builtin_fabsf(sm) - __builtin_fabsf(sh)) - cb.tau;
  return __builtin_fmaxf(__builtin_fmaxf(g1, g2), side);
}

struct SubvoxelWalk { /* the subvoxel ranges of one reflect-loop iteration (simulations.py:929-934) */
  int llx, lly, llz, ulx, uly, ulz;
};

/* Fallback: the whole iteration of one walker done by its own lane exactly as the reference does
 * it (simulations.py:936-983): every triangle of every subvoxel in range, no culling.  Kept
 * deliberately small in registers - the occupancy of the kernel is set by its largest code path
 * and this one runs only when a walker's work does not fit the wavefront's queues. */
DSD void iteration_serial(const MeshDev &ms, const SubvoxelWalk &w, const V3 &r0, const V3 &step, const double *xs,
                          const double *ys, const double *zs, BestHit &best) {
  const int nx = ms.nx, ny = ms.ny, nz = ms.nz;
  const double Lx = xs[nx], Ly = ys[ny], Lz = zs[nz];
  int visit = 0;
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
        const int v = visit++;
        const int2 rg = ms.sv_range[(x * ny + y) * nz + z];
        if (rg.y <= rg.x) continue;
        STAT_LANE(8); /* subvoxel scanned serially by its owner */
        STAT_WAVE(8);
        const V3 tr0 = {r0.x - (double)snx * Lx, r0.y - (double)sny * Ly, r0.z - (double)snz * Lz};
        for (int i = rg.x; i < rg.y; i++) {
          const int ti = ms.tri_idx[i];
          const double *rec = ms.tri + (int64_t)ti * DS_TRI_STRIDE;
          const double d = ray_triangle_intersection_check(load_v3(rec), load_v3(rec + 3), load_v3(rec + 6), tr0, step);
          offer_hit(best, d, v, i, ti);
        }
      }
    }
  }
}

/* Stage 1a: the owner lists the subvoxels of its iteration (reference order = visit number) as
 * tasks for the wavefront.  Returns false when they cannot be encoded or queued (the caller
 * then does the iteration serially). */
DSD bool push_subvoxels(const MeshDev &ms, const SubvoxelWalk &w, int nsub, CoopWave *W, int lane) {
  const int nx = ms.nx, ny = ms.ny, nz = ms.nz;
  if (nx > 1024 || ny > 1024 || nz > 1024) return false;
  /* image shifts must fit 6 bits each */
  if (w.llx < -32 * nx || w.ulx > 32 * nx || w.lly < -32 * ny || w.uly > 32 * ny || w.llz < -32 * nz || w.ulz > 32 * nz) return false;
  const int base = atomicAdd(&W->n_svq, nsub);
  uint2 *q = coop_svq(W);
  if (base + nsub > COOP_SVQ_CAP) {
    /* queue full: neutralise the part of the reservation that lies inside the queue */
    for (int j = base; j < COOP_SVQ_CAP; j++) q[j] = make_uint2(0xffffffffu, 0u);
    return false;
  }
  /* wrapped index and image number of the first subvoxel on every axis; the loop then steps
   * through the range z-fastest (simulations.py:936-938) without further divisions */
  int x = w.llx, snx = 0, y0 = w.lly, sny0 = 0, z0 = w.llz, snz0 = 0;
  if (x < 0 || x > nx - 1) {
    snx = floor_div(x, nx);
    x -= snx * nx;
  }
  if (y0 < 0 || y0 > ny - 1) {
    sny0 = floor_div(y0, ny);
    y0 -= sny0 * ny;
  }
  if (z0 < 0 || z0 > nz - 1) {
    snz0 = floor_div(z0, nz);
    z0 -= snz0 * nz;
  }
  int y = y0, sny = sny0, z = z0, snz = snz0, yi = w.lly, zi = w.llz;
  for (unsigned v = 0; v < (unsigned)nsub; v++) {
    q[base + v] = make_uint2((unsigned)lane | (v << 6) | ((unsigned)(snx + 32) << 14) | ((unsigned)(sny + 32) << 20) |
                                 ((unsigned)(snz + 32) << 26),
                             (unsigned)x | ((unsigned)y << 10) | ((unsigned)z << 20));
    zi++, z++;
    if (z == nz) z = 0, snz++;
    if (zi == w.ulz) {
      zi = w.llz, z = z0, snz = snz0;
      yi++, y++;
      if (y == ny) y = 0, sny++;
      if (yi == w.uly) {
        yi = w.lly, y = y0, sny = sny0;
        x++;
        if (x == nx) x = 0, snx++;
      }
    }
  }
  return true;
}

/* Both periodic subvoxel limits of one axis in float32, for the common case that decides itself: the boundaries
 * are np.linspace(x0, x0 + L, n + 1), image k of boundary i sits at (k n + i) h, so for a coordinate that is NOT
 * within 1e-4 subvoxel widths of any boundary the reference's scans (simulations.py:616-679) end at
 * ll = floor(t_min), ul = floor(t_max) + 1 with t = (x - x0) / h - whatever arithmetic produced t, as long as its
 * error stays below that distance.  t is the fp64 product rounded once to float32 (relative error 6e-8).  The
 * caller accepts when `edge` < 0.4999 on every axis (both fractional parts farther than 1e-4 from 0 and 1) and
 * the sum of all six |t| is below 1500 (so every |t| < 1500 and its error < 9e-5); otherwise - too close to a
 * boundary, too far out, or not a number - it takes the exact fp64 path. */
DSD void subvoxel_limits_f32(double x0, double inv_h, double x1, double x2, int &ll, int &ul, float &edge, float &reach) {
  const float t1 = (float)((x1 - x0) * inv_h), t2 = (float)((x2 - x0) * inv_h);
  const float tmin = __builtin_fminf(t1, t2), tmax = __builtin_fmaxf(t1, t2);
  const float fmin = __builtin_floorf(tmin), fmax = __builtin_floorf(tmax);
  ll = (int)fmin;
  ul = (int)fmax + 1;
  /* distance of the fractional parts from 1/2: below 0.4999 = farther than 1e-4 from both neighbouring boundaries */
  edge = __builtin_fmaxf(__builtin_fabsf((tmin - fmin) - 0.5f), __builtin_fabsf((tmax - fmax) - 0.5f));
  reach = __builtin_fabsf(t1) + __builtin_fabsf(t2); /* an addition: NaN survives it (min / max would drop it) */
}

/* subvoxel ranges (simulations.py:929-934) and cull box of the remaining segment */
DSD void segment_query(const MeshDev &ms, const V3 &r0, const V3 &step, double step_l, const double *xs, const double *ys,
                       const double *zs, SubvoxelWalk &sw, CullBox &cb0) {
  const double ex = __builtin_fma(step.x, step_l, r0.x);
  const double ey = __builtin_fma(step.y, step_l, r0.y);
  const double ez = __builtin_fma(step.z, step_l, r0.z);
  cb0 = make_cull_box(r0, step, step_l);
  {
    float e0, e1, e2, q0, q1, q2;
    subvoxel_limits_f32(ms.x0, ms.inv_hx, r0.x, ex, sw.llx, sw.ulx, e0, q0);
    subvoxel_limits_f32(ms.y0, ms.inv_hy, r0.y, ey, sw.lly, sw.uly, e1, q1);
    subvoxel_limits_f32(ms.z0, ms.inv_hz, r0.z, ez, sw.llz, sw.ulz, e2, q2);
    /* decided on all three axes: 99.9 % of the segments (NaN fails both comparisons) */
    if (__builtin_fmaxf(__builtin_fmaxf(e0, e1), e2) < 0.4999f && (q0 + q1) + q2 < 1500.0f) return;
  }
  const bool fx = subvoxel_limits_fast(xs, ms.nx, ms.x0, ms.x1, ms.inv_hx, ms.inv_lx, r0.x, ex, sw.llx, sw.ulx);
  const bool fy = subvoxel_limits_fast(ys, ms.ny, ms.y0, ms.y1, ms.inv_hy, ms.inv_ly, r0.y, ey, sw.lly, sw.uly);
  const bool fz = subvoxel_limits_fast(zs, ms.nz, ms.z0, ms.z1, ms.inv_hz, ms.inv_lz, r0.z, ez, sw.llz, sw.ulz);
  if (fx && fy && fz) return;
  sw.llx = ll_subvoxel_overlap_periodic(xs, ms.nx, ms.inv_hx, ms.inv_lx, r0.x, ex);
  sw.lly = ll_subvoxel_overlap_periodic(ys, ms.ny, ms.inv_hy, ms.inv_ly, r0.y, ey);
  sw.llz = ll_subvoxel_overlap_periodic(zs, ms.nz, ms.inv_hz, ms.inv_lz, r0.z, ez);
  sw.ulx = ul_subvoxel_overlap_periodic(xs, ms.nx, ms.inv_hx, ms.inv_lx, r0.x, ex);
  sw.uly = ul_subvoxel_overlap_periodic(ys, ms.ny, ms.inv_hy, ms.inv_ly, r0.y, ey);
  sw.ulz = ul_subvoxel_overlap_periodic(zs, ms.nz, ms.inv_hz, ms.inv_lz, r0.z, ez);
}

/* Stage 3: one trip of fp64 tests over the survivors [base, base + count) of the pool, count <= 64:
 * Moller-Trumbore test with the owner's ray, minimum per owner by 64-bit LDS atomics on the bit
 * pattern of d, ties resolved by a second atomic pass on (visit, list position, triangle).
 * Runs several times per pass (a full trip whenever the pool holds 64
 * survivors, the remainder at the end); seen_d is best_d[lane] as of the previous call.  Every
 * lane of the wavefront must call this.  Leaves W->n_surv = base. */
DSD void flush_survivors(const MeshDev &ms, const V3 &r0, const V3 &step, CoopWave *W, int lane, unsigned long long &seen_d,
                         int base, int count) {
  const double Lx = ms.x1, Ly = ms.y1, Lz = ms.z1;
  const int idx = base + lane;
  const bool valid = lane < count;
  uint4 sv = make_uint4(0u, 0u, 0u, 0u);
  if (valid) sv = make_uint4(W->surv_tri[idx], W->surv_pos[idx], W->surv_tag[idx], 0u);
  const int owner = (int)(sv.z & 63u);
  /* (visit, list position, triangle): the first two order equal distances as the reference's loop does, the triangle
   * rides along so that the minimum names the winner (24 bits each: the host sends meshes beyond that to the
   * per-lane traversal) */
  const unsigned long long key = ((unsigned long long)((sv.z >> 6) & 255u) << 48) | ((unsigned long long)sv.y << 24) | (unsigned long long)sv.x;
  V3 tr0 = {shfl_f64(r0.x, owner), shfl_f64(r0.y, owner), shfl_f64(r0.z, owner)};
  const V3 st = {shfl_f64(step.x, owner), shfl_f64(step.y, owner), shfl_f64(step.z, owner)};
  double d = 0;
  if (valid) { /* pass A: distances, atomic minimum of d */
    STAT_LANE(6);
    STAT_WAVE(6);
    if ((sv.z >> 14) != COOP_NO_SHIFT) { /* x - 0.0 == x bit for bit, so the unshifted case skips the arithmetic */
      const int snx = (int)((sv.z >> 14) & 63u) - 32, sny = (int)((sv.z >> 20) & 63u) - 32, snz = (int)(sv.z >> 26) - 32;
      tr0.x -= (double)snx * Lx, tr0.y -= (double)sny * Ly, tr0.z -= (double)snz * Lz;
    }
    const double *rec = ms.tri + (int64_t)(int)sv.x * DS_TRI_STRIDE;
    d = ray_triangle_intersection_check(load_v3(rec), load_v3(rec + 3), load_v3(rec + 6), tr0, st);
    if (d > 0) atomicMin(&W->best_d[owner], (unsigned long long)__double_as_longlong(d));
  }
  if (lane == 0) W->n_surv = base;
  wave_sync();
  T_MARK(W, lane, 5);
  { /* a smaller minimum than at the last flush invalidates the tie-break key of the earlier one */
    const unsigned long long cur = W->best_d[lane];
    if (cur != seen_d) {
      W->best_key[lane] = ~0ULL;
      seen_d = cur;
    }
  }
  wave_sync();
  const bool is_min = valid && d > 0 && (unsigned long long)__double_as_longlong(d) == W->best_d[owner];
  if (is_min) atomicMin(&W->best_key[owner], key); /* pass B: among equal d the earliest (visit, position) */
  /* the owner reads the winner's triangle out of the key and fetches its precomputed unit normal in stage 4, and
   * only if the hit lies within the step: most minima of a flush are beaten later or lie beyond the step's end */
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
  const int blk = ms.clear_idx[((ix >> lf) * ms.ny + (iy >> lf)) * ms.nz + (iz >> lf)];
  if (blk < 0) return true; /* nothing within reach of the whole subvoxel */
  const unsigned bit = (unsigned)(((((ix & m) << lf) | (iy & m)) << lf) | (iz & m));
  const unsigned words = lf >= 2 ? 1u << (3 * lf - 5) : 1u;
  const uint32_t w = ms.clear_bits[(size_t)blk * words + (bit >> 5)];
  return ((w >> (bit & 31u)) & 1u) == 0u;
}

/* ONE pass of the reflect loop of simulations.py:924-997 for the 64 walkers of a wavefront
 * together.  EVERY lane of the wavefront must call this; in_step marks the lanes whose walker
 * is inside a time step.  Returns true for the lanes whose step is over after this pass (no
 * hit within the remaining length, or the loop condition `step_l > 0 and iter_idx < max_iter`
 * no longer holds).  The caller lets every lane run through its own time steps at its own pace
 * (a wavefront then needs about mean-iterations-per-step passes per step, not the maximum over
 * its 64 lanes). */
DSD bool coop_iteration(V3 &r0, V3 &step, double &step_l, int64_t &iter_idx, bool in_step, bool clear, const WalkParams &p,
                        const double *xs, const double *ys, const double *zs, Rng &rng, CoopWave *W, int lane) {
  const MeshDev &ms = p.mesh;
  const int ny = ms.ny, nz = ms.nz;
  const double Lx = ms.x1, Ly = ms.y1, Lz = ms.z1; /* = xs[nx], ys[ny], zs[nz] */
  bool finished = false;
  bool act = in_step && step_l > 0 && iter_idx < p.max_iter;
  if (in_step && !act) finished = true; /* while-condition false on entry */
  if (act && clear) {
    /* no triangle within the step length (clearance map): this pass of the loop finds min_d > step_l
     * whatever the direction (simulations.py:986-987) - counted, but not traversed */
    iter_idx++;
    finished = true;
    act = false;
  }
  if (__ballot(act) == 0ULL) return finished;
  if (lane == 0) {
    W->n_surv = 0;
    W->n_svq = 0;
  }
  W->best_d[lane] = 0x7ff0000000000000ULL; /* +inf */
#ifdef DSIM_STATS
  W->filt[lane][0] = W->filt[lane][1] = W->filt[lane][2] = W->filt[lane][3] = 0xffffffffu;
#endif
  W->best_key[lane] = ~0ULL;
  W->overflow[lane] = 0;
  wave_sync();
  T_MARK(W, lane, 1);
#ifdef DSIM_TIMERS
  if (lane == 0) W->tacc[10] += 1;
#endif
  bool serial = false;
  /* ---- stage 1a: owners work out their cull box and list their subvoxels in range.  Box and ray
   * stay in the owner's registers (cb0, r0, step: unchanged until stage 4); the other lanes read
   * them with __shfl, so every lane has to be active wherever a shuffle is executed. */
  CullBox cb0 = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  if (act) {
    iter_idx++;
    STAT_LANE(0);
    STAT_WAVE(0);
    SubvoxelWalk sw;
    segment_query(ms, r0, step, step_l, xs, ys, zs, sw, cb0);
    const long long nsub = (long long)max(0, sw.ulx - sw.llx) * max(0, sw.uly - sw.lly) * max(0, sw.ulz - sw.llz);
    serial = nsub > 256 || !push_subvoxels(ms, sw, (int)nsub, W, lane);
  }
  T_MARK(W, lane, 2);
  wave_sync();
  /* ---- stage 1b: all lanes look the subvoxels up and list the micro-cells the cull box reaches */
  int n_cellq = 0; /* cell tasks queued (wave-uniform; may exceed the capacity: the excess was not written) */
  {
    /* a counter read from LDS is the same in every lane, but only readfirstlane tells the compiler so: the queue
     * loops then run on scalar compares and branches instead of exec-mask arithmetic, and their counters live
     * in SGPRs */
    const int n = __builtin_amdgcn_readfirstlane(min(W->n_svq, COOP_SVQ_CAP));
    const uint2 *svq = coop_svq(W);
    uint2 *cq = coop_cellq(W);
    for (int first = 0; first < n; first += 64) {
      const int idx = first + lane;
      uint2 e = make_uint2(0xffffffffu, 0u);
      if (idx < n) e = svq[idx];
      const bool valid = e.x != 0xffffffffu;
      const int owner = (int)(e.x & 63u);
      const int x = (int)(e.y & 1023u), y = (int)((e.y >> 10) & 1023u), z = (int)(e.y >> 20);
      const int sv = (x * ny + y) * nz + z;
      int2 mg = make_int2(0, 0);
      if (valid) {
        STAT_LANE(1);
        STAT_WAVE(1);
        mg = ms.sv_micro[sv];
      }
      /* the owner's segment (7 floats through ds_bpermute); the box is rebuilt from it on this side */
      CullBox cb = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, __shfl(cb0.mx, owner, 64), __shfl(cb0.my, owner, 64), __shfl(cb0.mz, owner, 64),
                    __shfl(cb0.hx, owner, 64), __shfl(cb0.hy, owner, 64), __shfl(cb0.hz, owner, 64), __shfl(cb0.tau, owner, 64)};
      box_from_segment(cb);
      const bool has = mg.x != 0 && !ABLATE(1); /* not an empty subvoxel (or no task for this lane) */
      if (has) {
        STAT_LANE(2);
        STAT_WAVE(2);
      }
      int a0 = 0, a1 = 0, b0 = 0, b1 = 0, c0 = 0, c1 = 0, k = 1;
      if (mg.x > 0 && !ABLATE(1)) {
        if ((e.x >> 14) != COOP_NO_SHIFT) { /* periodic image: move the box into its frame */
          const int snx = (int)((e.x >> 14) & 63u) - 32, sny = (int)((e.x >> 20) & 63u) - 32, snz = (int)(e.x >> 26) - 32;
          cb = shift_cull_box(cb, (double)snx * Lx, (double)sny * Ly, (double)snz * Lz);
        }
        k = 1 << mg.y;
        /* The cells the box reaches, in float32 and WIDENED: any superset of the exact cells of micro_cell() is
         * valid (an extra cell only costs its records a look).  The float32 evaluation of (c - x0) * scale is off
         * by < 0.5 ulp of the subvoxel origin times scale + a few ulps of the result, i.e. < 3e-7 k n_sv + 1e-6 k
         * cells; the ranges are widened by dl >= 1e-3 + 5e-7 k n_sv cells on both sides. */
        const float kf = (float)k;
        const float dl = __builtin_fmaf(5e-7f * kf, (float)max(ms.nx, max(ms.ny, ms.nz)), 1e-3f);
        const float sx = kf * (float)ms.inv_hx, sy = kf * (float)ms.inv_hy, sz = kf * (float)ms.inv_hz;
        const float ox = (float)xs[x], oy = (float)ys[y], oz = (float)zs[z];
        a0 = min(max((int)((cb.lox - ox) * sx - dl), 0), k - 1), a1 = min(max((int)((cb.hix - ox) * sx + dl), 0), k - 1);
        b0 = min(max((int)((cb.loy - oy) * sy - dl), 0), k - 1), b1 = min(max((int)((cb.hiy - oy) * sy + dl), 0), k - 1);
        c0 = min(max((int)((cb.loz - oz) * sz - dl), 0), k - 1), c1 = min(max((int)((cb.hiz - oz) * sz + dl), 0), k - 1);
      }
      /* queue slots from a prefix sum over the lanes (DPP; no LDS atomic) */
      const int ncell = has ? (a1 - a0 + 1) * (b1 - b0 + 1) * (c1 - c0 + 1) : 0;
      const int incl = wave_inclusive_sum(ncell);
      const int base = n_cellq + incl - ncell;
      n_cellq += __builtin_amdgcn_readlane(incl, 63);
      if (!has) continue;
      if (base + ncell > COOP_CELLQ_CAP) {
        /* queue full: neutralise the part of the reservation that lies inside the queue */
        for (int j = base; j < COOP_CELLQ_CAP; j++) cq[j] = make_uint2(0xffffffffu, 0u);
        W->overflow[owner] = 1;
        continue;
      }
      if (mg.x < 0) { /* single-cell subvoxel: its record range is stored inline in sv_micro */
        cq[base] = make_uint2(e.x, 0x80000000u | (unsigned)sv);
        continue;
      }
      int j = base;
      for (int a = a0; a <= a1; a++)
        for (int b = b0; b <= b1; b++)
          for (int c = c0; c <= c1; c++) cq[j++] = make_uint2(e.x, (unsigned)(mg.x + (a * k + b) * k + c));
    }
  }
  T_MARK(W, lane, 12);
  wave_sync();
  /* ---- stages 1c, 2, 3 in rounds: the cell tasks are cut into 4-record items until the item queue
   * is nearly full, the items are drained into the survivor pool, the pool is flushed (fp64 tests,
   * per-owner minimum) when it is half full - so no queue size limits the amount of work per pass */
  {
    const int n_cells = min(n_cellq, COOP_CELLQ_CAP);
    const uint2 *cq = coop_cellq(W);
    uint2 *items = coop_items(W);
    unsigned long long seen_d = 0x7ff0000000000000ULL; /* best_d[lane] at the last flush */
    int cursor = 0;
    int n_items = 0;        /* items waiting in the queue (wave-uniform) */
    bool drain_all = false; /* the next cell does not fit behind the waiting items */
    while (cursor < n_cells) {
      /* -- 1c: up to 64 cell tasks per round, one per lane */
      while (cursor < n_cells && n_items <= COOP_ITEM_CAP - 64) {
        const int idx = cursor + lane;
        uint2 e = make_uint2(0xffffffffu, 0u);
        if (idx < n_cells) e = cq[idx];
        int2 cr = make_int2(0, 0);
        if (e.x != 0xffffffffu) {
          const bool inl = (e.y & 0x80000000u) != 0;
          const int2 *src = inl ? ms.sv_micro + (e.y & 0x7fffffffu) : ms.cell_range + e.y;
          cr = *src;
          if (inl) cr = make_int2(-cr.x - 1, -cr.x - 1 + cr.y);
        }
        if (e.x != 0xffffffffu) {
          STAT_LANE(5);
          STAT_WAVE(5);
        }
        const int cnt = ABLATE(2) ? 0 : max(0, cr.y - cr.x);
        const int nb = (cnt + 3) >> 2;
        if (nb > 0) {
          STAT_LANE(3);
          STAT_WAVE(3);
        }
        const int incl = wave_inclusive_sum(nb);
        const unsigned long long fits = __ballot(n_items + incl <= COOP_ITEM_CAP);
        const int n_take = fits == ~0ULL ? 64 : (int)__ffsll((long long)~fits) - 1; /* the lanes that fit form a prefix */
        if (n_take == 0) {
          if (n_items > 0) { /* drain first, then retry with an empty queue */
            drain_all = true;
            break;
          }
          /* a single cell with more records than the whole queue holds: its owner works serially */
          if (lane == 0) W->overflow[(int)(e.x & 63u)] = 1;
          cursor += 1;
          continue;
        }
        if (lane < n_take) {
          const int base = n_items + incl - nb;
          for (int j = 0; j < nb; j++) {
            const int left = cnt - 4 * j;
            items[base + j] = make_uint2((unsigned)(cr.x + 4 * j) | ((unsigned)((left < 4 ? left : 4) - 1) << 30), e.x);
          }
        }
        n_items += __builtin_amdgcn_readlane(incl, n_take - 1);
        cursor += n_take;
        if (n_take < 64) break;
      }
      T_MARK(W, lane, 13);
      /* -- 2: all lanes drain the item queue: float32 AABB + plane culling against the owner's box.
       * While more cells are coming only full trips of 64 items are drained (the tail of the queue);
       * the remainder waits at the head for the next round */
      const int start = (cursor >= n_cells || drain_all) ? 0 : (n_items & 63);
      drain_all = false;
      if (start == n_items) continue;
      wave_sync();
      for (int first = start; first < n_items; first += 64) {
        const int idx = first + lane;
        const bool valid = idx < n_items;
        uint2 it = make_uint2(0u, 0u);
        if (valid) it = items[idx];
        const int owner = (int)(it.y & 63u);
        CullBox cb;
        cb.mx = __shfl(cb0.mx, owner, 64), cb.my = __shfl(cb0.my, owner, 64), cb.mz = __shfl(cb0.mz, owner, 64);
        cb.hx = __shfl(cb0.hx, owner, 64), cb.hy = __shfl(cb0.hy, owner, 64), cb.hz = __shfl(cb0.hz, owner, 64);
        cb.tau = __shfl(cb0.tau, owner, 64);
        box_from_segment(cb);
        if (valid) {
        STAT_LANE(4);
        STAT_WAVE(4);
        const int first_rec = (int)(it.x & 0x3fffffffu), cnt = (int)(it.x >> 30) + 1;
        if ((it.y >> 14) != COOP_NO_SHIFT) { /* periodic image: move the box into its frame */
          const int snx = (int)((it.y >> 14) & 63u) - 32, sny = (int)((it.y >> 20) & 63u) - 32, snz = (int)(it.y >> 26) - 32;
          cb = shift_cull_box(cb, (double)snx * Lx, (double)sny * Ly, (double)snz * Lz);
        }
        float4 q0[4], q1[4], q2[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int j = ABLATE(4) ? u : first_rec + min(u, cnt - 1); /* 4: every lane reads records 0..3 */
          q0[u] = ms.mrec[3 * j];
          q1[u] = ms.mrec[3 * j + 1];
          q2[u] = ms.mrec[3 * j + 2];
        }
        /* the four tests without a branch between them (the plane test used to sit behind the box test, but
         * with 64 lanes x 4 records somebody always passes the box, so the branch only cost mask juggling);
         * one LDS atomic per lane reserves the slots of all its survivors */
        unsigned alive = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) alive |= (cull_margin(q0[u], q1[u], q2[u], cb) <= 0.0f ? 1u : 0u) << u;
        alive &= (1u << cnt) - 1u; /* slots beyond the item's count repeat its last record (loads stay unconditional) */
        if (ABLATE(3) || ABLATE(4)) alive = 0;
        if (alive) {
          int slot = atomicAdd(&W->n_surv, (int)__builtin_popcount(alive));
#pragma unroll
          for (int u = 0; u < 4; u++) {
            if (!(alive >> u & 1u)) continue;
#ifdef DSIM_STATS
            {
              const unsigned pos_ = (unsigned)__float_as_int(q2[u].w);
              const unsigned prev_ = atomicExch(&W->filt[owner][(pos_ ^ (pos_ >> 2)) & 3u], pos_);
              if (prev_ == pos_) STAT_LANE(7); /* the same list entry of the same owner survived from another cell */
            }
#endif
            if (slot < COOP_SURV_CAP) {
              W->surv_tri[slot] = (unsigned)__float_as_int(q2[u].z);
              W->surv_pos[slot] = (unsigned)__float_as_int(q2[u].w);
              W->surv_tag[slot] = it.y;
            } else
              W->overflow[owner] = 1;
            slot++;
          }
        }
        }
        T_MARK(W, lane, 4);
        /* a trip adds 15 survivors on average and 256 at most: whenever the pool holds 64, its last
         * 64 entries make one full trip of fp64 tests */
        wave_sync();
        for (int n = __builtin_amdgcn_readfirstlane(min(W->n_surv, COOP_SURV_CAP)); n >= 64; n -= 64)
          flush_survivors(ms, r0, step, W, lane, seen_d, n - 64, 64);
      }
      n_items = start;
    }
    {
      const int n = __builtin_amdgcn_readfirstlane(min(W->n_surv, COOP_SURV_CAP)); /* < 64 */
      if (n > 0) flush_survivors(ms, r0, step, W, lane, seen_d, 0, n);
    }
  }
  /* ---- stage 4: resolve (simulations.py:986-997) */
  if (act) {
    double d = DS_INF;
    V3 n = {0, 0, 0};
    if (serial || W->overflow[lane]) {
      /* this walker's work did not fit the queues: its whole iteration again, by its own lane */
      STAT_LANE(9);
      STAT_WAVE(9);
      BestHit best = {DS_INF, 0x7fffffff, 0x7fffffff, 0, false, {0, 0, 0}};
      SubvoxelWalk sw; /* recomputed rather than kept in registers through the drain stages */
      CullBox cbx;
      segment_query(ms, r0, step, step_l, xs, ys, zs, sw, cbx);
      iteration_serial(ms, sw, r0, step, xs, ys, zs, best);
      d = best.d;
      if (d <= step_l) {
        const double *rec = ms.tri + (int64_t)best.tri * DS_TRI_STRIDE;
        n = load_v3(rec + 9);
      }
    } else {
      const unsigned long long bd = W->best_d[lane];
      if (bd != 0x7ff0000000000000ULL) {
        d = __longlong_as_double((long long)bd);
        if (d <= step_l) n = load_v3(ms.tri + (int64_t)(W->best_key[lane] & 0xffffffULL) * DS_TRI_STRIDE + 9);
      }
    }
    T_MARK(W, lane, 7);
    if (d > step_l) {
      finished = true; /* check_intersection = False */
    } else {
      const bool reflect = p.perm_prob < rng_uniform_f64(rng);
      if (reflect)
        reflection(r0, step, d, n, p.eps);
      else
        crossing(r0, step, d, n, p.eps);
      step_l -= d;
      if (!(step_l > 0 && iter_idx < p.max_iter)) finished = true; /* the while-condition of the next pass */
    }
  }
  T_MARK(W, lane, 8);
  return finished;
}

#endif /* DSIM_MESH_COOP_H */
