/* dsim_mesh_coop.h - wavefront-cooperative traversal of the mesh index.
 *
 * Per-lane traversal (step_mesh in dsim_kernels.h) pays, at every loop level,
 * for the slowest of the 64 lanes: measured on the neuron (tools/mesh_stats.py)
 * a wavefront makes 47 record-batch trips per step with 8.6 lanes active and 45
 * Moller-Trumbore trips with 2.2 lanes active.  Here the 64 walkers of a
 * wavefront pool their candidate work in LDS:
 *
 *   stage 1  every walker ("owner") enumerates the micro-cells its segment
 *            reaches and appends 4-record work items to the wave's item queue;
 *   stage 2  all 64 lanes drain the queue round-robin - each item is 4 records
 *            tested (float32 AABB + plane) against the OWNER's cull box read
 *            from LDS; survivors are appended to the owner's survivor slots;
 *   stage 3  all lanes drain the survivor pool: fp64 Moller-Trumbore test with
 *            the OWNER's ray (LDS), minimum per owner by 64-bit LDS atomics on
 *            the bit pattern of d, ties resolved by a second atomic pass;
 *   stage 4  reflection / crossing per walker, then the next iteration for
 *            the walkers that hit something (simulations.py:924-997).
 *
 * Which triangles are tested, in fp64, against which ray is unchanged, and the
 * minimum keeps the reference's "first visited wins a tie" rule through the
 * explicit key (subvoxel visit number, position in the reference's list), so
 * results stay bit-identical to step_mesh and to the oracle.  Anything that
 * does not fit the queues (more than 255 subvoxels in range, image shifts
 * beyond +-31 voxels, item queue or survivor pool overflow) is handled by the
 * owner itself with the serial scan.
 */
#ifndef DSIM_MESH_COOP_H
#define DSIM_MESH_COOP_H

#define COOP_ITEM_CAP 512
#define COOP_SURV_CAP 192
#define COOP_SVQ_CAP 128
#define COOP_CELLQ_CAP 256

/* Per-wavefront scratch in LDS.  Two byte pools are reused along the pass:
 *   pool_a: item queue (stage 1c -> 2), afterwards the survivors' distances and the winners' normals (stage 3);
 *   pool_b: subvoxel and cell task queues (stage 1a -> 1c), afterwards the survivor pool (stage 2 -> 3). */
struct CoopWave {
  float box[64][13];                    /* owner's cull box: lo xyz, hi xyz, a xyz, e xyz, tau      3.3 KB */
  double ray[64][6];                    /* owner's r0 xyz, step xyz (fp64, for the exact test)      3.0 KB */
  unsigned long long best_d[64];        /* smallest positive d (bit pattern orders like the value)  0.5 KB */
  unsigned long long best_key[64];      /* (visit << 32) | list position of the winner              0.5 KB */
  int best_tri[64];                     /* its triangle                                                    */
  int overflow[64];                     /* some work of this owner did not fit a queue: redo serially      */
  unsigned long long pool_a[COOP_ITEM_CAP];                   /*                                    4.0 KB */
  unsigned long long pool_b[COOP_SURV_CAP * 2];               /*                                    3.0 KB */
  int n_items, n_surv, n_svq, n_cellq;
#ifdef DSIM_TIMERS
  unsigned long long tacc[16], tlast;
  unsigned long long tpad;
#endif
};
static_assert(COOP_SURV_CAP * 8 + 64 * 3 * 8 <= COOP_ITEM_CAP * 8, "pool_a: distances + normals");
static_assert((COOP_SVQ_CAP + COOP_CELLQ_CAP) * 8 <= COOP_SURV_CAP * 16, "pool_b: task queues");

/* {first record | (count-1) << 30, owner|visit|shifts} */
DSD uint2 *coop_items(CoopWave *W) { return reinterpret_cast<uint2 *>(W->pool_a); }
/* survivor's distance, kept between the reduction passes */
DSD double *coop_surv_d(CoopWave *W) { return reinterpret_cast<double *>(W->pool_a); }
/* E1 x E2 of each owner's winning triangle (un-normalised normal), [64][3] */
DSD double *coop_best_n(CoopWave *W) { return reinterpret_cast<double *>(W->pool_a) + COOP_SURV_CAP; }
/* {triangle, list position, owner|visit|shifts, 0} */
DSD uint4 *coop_surv(CoopWave *W) { return reinterpret_cast<uint4 *>(W->pool_b); }
/* {owner|visit|shifts, x | y << 10 | z << 20} of a subvoxel in range */
DSD uint2 *coop_svq(CoopWave *W) { return reinterpret_cast<uint2 *>(W->pool_b); }
/* {owner|visit|shifts, index into cell_range  or  0x80000000 | subvoxel (range stored inline)} */
DSD uint2 *coop_cellq(CoopWave *W) { return reinterpret_cast<uint2 *>(W->pool_b) + COOP_SVQ_CAP; }

DSD void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* cull box moved into the frame of a periodic image (shift subtracted), inflated by the
 * float32 rounding of the subtraction; shift = 0 returns the box unchanged */
DSD CullBox shift_cull_box(const CullBox &c, double shx, double shy, double shz) {
  if (shx == 0 && shy == 0 && shz == 0) return c;
  const float sx = (float)shx, sy = (float)shy, sz = (float)shz;
  const float mx = 5e-7f * (__builtin_fabsf(sx) + __builtin_fmaxf(__builtin_fabsf(c.lox), __builtin_fabsf(c.hix)));
  const float my = 5e-7f * (__builtin_fabsf(sy) + __builtin_fmaxf(__builtin_fabsf(c.loy), __builtin_fabsf(c.hiy)));
  const float mz = 5e-7f * (__builtin_fabsf(sz) + __builtin_fmaxf(__builtin_fabsf(c.loz), __builtin_fabsf(c.hiz)));
  CullBox o;
  o.lox = c.lox - sx - mx;
  o.loy = c.loy - sy - my;
  o.loz = c.loz - sz - mz;
  o.hix = c.hix - sx + mx;
  o.hiy = c.hiy - sy + my;
  o.hiz = c.hiz - sz + mz;
  o.ax = c.ax - sx;
  o.ay = c.ay - sy;
  o.az = c.az - sz;
  o.ex = c.ex - sx;
  o.ey = c.ey - sy;
  o.ez = c.ez - sz;
  o.tau = c.tau + 4.0f * (mx + my + mz);
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

DSD bool cull_record(const float4 &q0, const float4 &q1, const float4 &q2, const CullBox &cb) {
  if (q0.x > cb.hix || q0.w < cb.lox || q0.y > cb.hiy || q1.x < cb.loy || q0.z > cb.hiz || q1.y < cb.loz) return true;
  const float s0 = __builtin_fmaf(q2.x, cb.az, __builtin_fmaf(q1.z, cb.ax, q1.w * cb.ay)) - q2.y;
  const float s1 = __builtin_fmaf(q2.x, cb.ez, __builtin_fmaf(q1.z, cb.ex, q1.w * cb.ey)) - q2.y;
  return (s0 > cb.tau && s1 > cb.tau) || (s0 < -cb.tau && s1 < -cb.tau);
}

/* serial scan of one micro-cell by its owner (fallback path) */
DSD void scan_cell_serial(const MeshDev &ms, int begin, int end, const CullBox &cb, const V3 &tr0, const V3 &step, int visit,
                          BestHit &best) {
  for (int i = begin; i < end; i++) {
    const float4 q0 = ms.mrec[3 * i], q1 = ms.mrec[3 * i + 1], q2 = ms.mrec[3 * i + 2];
    if (cull_record(q0, q1, q2, cb)) continue;
    const int ti = __float_as_int(q2.z), pos = __float_as_int(q2.w);
    const double *rec = ms.tri + (int64_t)ti * 9;
    double d = ray_triangle_intersection_check(load_v3(rec), load_v3(rec + 3), load_v3(rec + 6), tr0, step);
    offer_hit(best, d, visit, pos, ti);
  }
}

struct SubvoxelWalk { /* the subvoxel ranges of one reflect-loop iteration (simulations.py:929-934) */
  int llx, lly, llz, ulx, uly, ulz;
};

/* The whole iteration of one walker done by its own lane, in the reference's order (fallback). */
DSD void enumerate_cells_serial(const MeshDev &ms, const SubvoxelWalk &w, const CullBox &cb0, const V3 &r0, const V3 &step,
                                const double *xs, const double *ys, const double *zs, BestHit &best) {
  const int nx = ms.nx, ny = ms.ny, nz = ms.nz;
  const double Lx = xs[nx], Ly = ys[ny], Lz = zs[nz];
  int visit = 0;
  for (int xi = w.llx; xi < w.ulx; xi++) {
    int x = xi, snx = 0;
    if (x < 0 || x > nx - 1) {
      snx = floor_div(x, nx);
      x -= snx * nx;
    }
    const double shx = (double)snx * Lx;
    for (int yi = w.lly; yi < w.uly; yi++) {
      int y = yi, sny = 0;
      if (y < 0 || y > ny - 1) {
        sny = floor_div(y, ny);
        y -= sny * ny;
      }
      const double shy = (double)sny * Ly;
      for (int zi = w.llz; zi < w.ulz; zi++) {
        int z = zi, snz = 0;
        if (z < 0 || z > nz - 1) {
          snz = floor_div(z, nz);
          z -= snz * nz;
        }
        const double shz = (double)snz * Lz;
        const int v = visit++;
        const int sv = (x * ny + y) * nz + z;
        const int2 mg = ms.sv_micro[sv];
        if (mg.x == 0) continue; /* empty subvoxel */
        STAT_LANE(8); /* subvoxel scanned serially by its owner */
        STAT_WAVE(8);
        const CullBox cb = shift_cull_box(cb0, shx, shy, shz);
        const V3 tr0 = {r0.x - shx, r0.y - shy, r0.z - shz};
        if (mg.x < 0) { /* single-cell subvoxel: its record range is stored inline */
          scan_cell_serial(ms, -mg.x - 1, -mg.x - 1 + mg.y, cb, tr0, step, v, best);
          continue;
        }
        const int k = 1 << mg.y;
        const double sx = (double)k * ms.inv_hx, sy = (double)k * ms.inv_hy, sz = (double)k * ms.inv_hz;
        const int a0 = micro_cell((double)cb.lox, xs[x], sx, k), a1 = micro_cell((double)cb.hix, xs[x], sx, k);
        const int b0 = micro_cell((double)cb.loy, ys[y], sy, k), b1 = micro_cell((double)cb.hiy, ys[y], sy, k);
        const int c0 = micro_cell((double)cb.loz, zs[z], sz, k), c1 = micro_cell((double)cb.hiz, zs[z], sz, k);
        for (int a = a0; a <= a1; a++)
          for (int b = b0; b <= b1; b++)
            for (int c = c0; c <= c1; c++) {
              const int2 cr = ms.cell_range[mg.x + (a * k + b) * k + c];
              scan_cell_serial(ms, cr.x, cr.y, cb, tr0, step, v, best);
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
  int j = base;
  unsigned visit = 0;
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
      const unsigned tag_xy = (unsigned)lane | ((unsigned)(snx + 32) << 14) | ((unsigned)(sny + 32) << 20);
      const unsigned cell_xy = (unsigned)x | ((unsigned)y << 10);
      for (int zi = w.llz; zi < w.ulz; zi++) {
        int z = zi, snz = 0;
        if (z < 0 || z > nz - 1) {
          snz = floor_div(z, nz);
          z -= snz * nz;
        }
        q[j++] = make_uint2(tag_xy | (visit++ << 6) | ((unsigned)(snz + 32) << 26), cell_xy | ((unsigned)z << 20));
      }
    }
  }
  return true;
}

/* subvoxel ranges (simulations.py:929-934) and cull box of the remaining segment */
DSD void segment_query(const MeshDev &ms, const V3 &r0, const V3 &step, double step_l, const double *xs, const double *ys,
                       const double *zs, SubvoxelWalk &sw, CullBox &cb0) {
  const double ex = __builtin_fma(step.x, step_l, r0.x);
  const double ey = __builtin_fma(step.y, step_l, r0.y);
  const double ez = __builtin_fma(step.z, step_l, r0.z);
  sw.llx = ll_subvoxel_overlap_periodic(xs, ms.nx, ms.inv_hx, ms.inv_lx, r0.x, ex);
  sw.lly = ll_subvoxel_overlap_periodic(ys, ms.ny, ms.inv_hy, ms.inv_ly, r0.y, ey);
  sw.llz = ll_subvoxel_overlap_periodic(zs, ms.nz, ms.inv_hz, ms.inv_lz, r0.z, ez);
  sw.ulx = ul_subvoxel_overlap_periodic(xs, ms.nx, ms.inv_hx, ms.inv_lx, r0.x, ex);
  sw.uly = ul_subvoxel_overlap_periodic(ys, ms.ny, ms.inv_hy, ms.inv_ly, r0.y, ey);
  sw.ulz = ul_subvoxel_overlap_periodic(zs, ms.nz, ms.inv_hz, ms.inv_lz, r0.z, ez);
  cb0 = make_cull_box(r0, step, step_l);
}

/* ONE pass of the reflect loop of simulations.py:924-997 for the 64 walkers of a wavefront
 * together.  EVERY lane of the wavefront must call this; in_step marks the lanes whose walker
 * is inside a time step.  Returns true for the lanes whose step is over after this pass (no
 * hit within the remaining length, or the loop condition `step_l > 0 and iter_idx < max_iter`
 * no longer holds).  The caller lets every lane run through its own time steps at its own pace
 * (a wavefront then needs about mean-iterations-per-step passes per step, not the maximum over
 * its 64 lanes). */
DSD bool coop_iteration(V3 &r0, V3 &step, double &step_l, int64_t &iter_idx, bool in_step, const WalkParams &p, const double *xs,
                        const double *ys, const double *zs, Rng &rng, CoopWave *W, int lane) {
  const MeshDev &ms = p.mesh;
  const int nx = ms.nx, ny = ms.ny, nz = ms.nz;
  const double Lx = xs[nx], Ly = ys[ny], Lz = zs[nz];
  bool finished = false;
  {
    const bool act = in_step && step_l > 0 && iter_idx < p.max_iter;
    if (in_step && !act) finished = true; /* while-condition false on entry */
    if (__ballot(act) == 0ULL) return finished;
    if (lane == 0) {
      W->n_items = 0;
      W->n_surv = 0;
      W->n_svq = 0;
      W->n_cellq = 0;
    }
    W->best_d[lane] = 0x7ff0000000000000ULL; /* +inf */
    W->best_key[lane] = ~0ULL;
    W->overflow[lane] = 0;
    wave_sync();
    T_MARK(W, lane, 1);
#ifdef DSIM_TIMERS
    if (lane == 0) W->tacc[10] += 1;
#endif
    BestHit best = {DS_INF, 0x7fffffff, 0x7fffffff, 0, false, {0, 0, 0}};
    bool serial_all = false;
    /* ---- stage 1a: owners publish their cull box, ray and subvoxels in range */
    if (act) {
      iter_idx++;
      STAT_LANE(0);
      STAT_WAVE(0);
      SubvoxelWalk sw;
      CullBox cb0;
      segment_query(ms, r0, step, step_l, xs, ys, zs, sw, cb0);
      const long long nsub = (long long)max(0, sw.ulx - sw.llx) * max(0, sw.uly - sw.lly) * max(0, sw.ulz - sw.llz);
      serial_all = nsub > 256 || !push_subvoxels(ms, sw, (int)nsub, W, lane);
      if (serial_all) {
        STAT_LANE(10);
        STAT_WAVE(10);
        enumerate_cells_serial(ms, sw, cb0, r0, step, xs, ys, zs, best);
      } else {
        float *bx = W->box[lane];
        bx[0] = cb0.lox, bx[1] = cb0.loy, bx[2] = cb0.loz, bx[3] = cb0.hix, bx[4] = cb0.hiy, bx[5] = cb0.hiz;
        bx[6] = cb0.ax, bx[7] = cb0.ay, bx[8] = cb0.az, bx[9] = cb0.ex, bx[10] = cb0.ey, bx[11] = cb0.ez, bx[12] = cb0.tau;
        double *ry = W->ray[lane];
        ry[0] = r0.x, ry[1] = r0.y, ry[2] = r0.z, ry[3] = step.x, ry[4] = step.y, ry[5] = step.z;
      }
    }
    T_MARK(W, lane, 2);
    wave_sync();
    /* ---- stage 1b: all lanes look the subvoxels up and list the micro-cells the cull box reaches */
    {
      const int n = min(W->n_svq, COOP_SVQ_CAP);
      const uint2 *svq = coop_svq(W);
      uint2 *cq = coop_cellq(W);
      for (int idx = lane; idx < n; idx += 64) {
        const uint2 e = svq[idx];
        if (e.x == 0xffffffffu) continue;
        const int x = (int)(e.y & 1023u), y = (int)((e.y >> 10) & 1023u), z = (int)(e.y >> 20);
        const int sv = (x * ny + y) * nz + z;
        STAT_LANE(1);
        STAT_WAVE(1);
        const int2 mg = ms.sv_micro[sv];
        if (mg.x == 0 || ABLATE(1)) continue; /* empty subvoxel */
        const int owner = (int)(e.x & 63u);
        int a0 = 0, a1 = 0, b0 = 0, b1 = 0, c0 = 0, c1 = 0, k = 1;
        if (mg.x > 0) {
          const int snx = (int)((e.x >> 14) & 63u) - 32, sny = (int)((e.x >> 20) & 63u) - 32, snz = (int)(e.x >> 26) - 32;
          const float *bx = W->box[owner];
          CullBox cb = {bx[0], bx[1], bx[2], bx[3], bx[4], bx[5], 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
          cb = shift_cull_box(cb, (double)snx * Lx, (double)sny * Ly, (double)snz * Lz);
          k = 1 << mg.y;
          const double sx = (double)k * ms.inv_hx, sy = (double)k * ms.inv_hy, sz = (double)k * ms.inv_hz;
          a0 = micro_cell((double)cb.lox, xs[x], sx, k), a1 = micro_cell((double)cb.hix, xs[x], sx, k);
          b0 = micro_cell((double)cb.loy, ys[y], sy, k), b1 = micro_cell((double)cb.hiy, ys[y], sy, k);
          c0 = micro_cell((double)cb.loz, zs[z], sz, k), c1 = micro_cell((double)cb.hiz, zs[z], sz, k);
        }
        const int ncell = (a1 - a0 + 1) * (b1 - b0 + 1) * (c1 - c0 + 1);
        const int base = atomicAdd(&W->n_cellq, ncell);
        if (base + ncell > COOP_CELLQ_CAP) {
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
    /* ---- stage 1c: all lanes look the cells up and cut their record ranges into 4-record items */
    {
      const int n = min(W->n_cellq, COOP_CELLQ_CAP);
      const uint2 *cq = coop_cellq(W);
      uint2 *items = coop_items(W);
      for (int idx = lane; idx < n; idx += 64) {
        const uint2 e = cq[idx];
        if (e.x == 0xffffffffu) continue;
        const bool inl = (e.y & 0x80000000u) != 0;
        const int2 *src = inl ? ms.sv_micro + (e.y & 0x7fffffffu) : ms.cell_range + e.y;
        int2 cr = *src;
        if (inl) cr = make_int2(-cr.x - 1, -cr.x - 1 + cr.y);
        const int cnt = cr.y - cr.x;
        if (cnt <= 0 || ABLATE(2)) continue;
        STAT_LANE(3);
        STAT_WAVE(3);
        const int nb = (cnt + 3) >> 2;
        const int base = atomicAdd(&W->n_items, nb);
        if (base + nb > COOP_ITEM_CAP) {
          for (int j = base; j < COOP_ITEM_CAP; j++) items[j] = make_uint2(0xffffffffu, 0u);
          W->overflow[(int)(e.x & 63u)] = 1;
          continue;
        }
        for (int j = 0; j < nb; j++) {
          const int left = cnt - 4 * j;
          items[base + j] = make_uint2((unsigned)(cr.x + 4 * j) | ((unsigned)((left < 4 ? left : 4) - 1) << 30), e.x);
        }
      }
    }
    T_MARK(W, lane, 13);
    wave_sync();
    T_MARK(W, lane, 3);
    /* ---- stage 2: all lanes drain the item queue (skipped by wavefronts in open space) */
    const int n_queued = W->n_items;
    if (n_queued > 0) {
      const int n = min(n_queued, COOP_ITEM_CAP);
      for (int idx = lane; idx < n; idx += 64) {
        STAT_LANE(4);
        STAT_WAVE(4);
        const uint2 it = coop_items(W)[idx];
        if (it.x == 0xffffffffu) continue;
        const int first = (int)(it.x & 0x3fffffffu), cnt = (int)(it.x >> 30) + 1;
        const int owner = (int)(it.y & 63u);
        const int snx = (int)((it.y >> 14) & 63u) - 32, sny = (int)((it.y >> 20) & 63u) - 32, snz = (int)(it.y >> 26) - 32;
        const float *bx = W->box[owner];
        CullBox cb;
        cb.lox = bx[0], cb.loy = bx[1], cb.loz = bx[2], cb.hix = bx[3], cb.hiy = bx[4], cb.hiz = bx[5];
        cb.ax = bx[6], cb.ay = bx[7], cb.az = bx[8], cb.ex = bx[9], cb.ey = bx[10], cb.ez = bx[11], cb.tau = bx[12];
        cb = shift_cull_box(cb, (double)snx * Lx, (double)sny * Ly, (double)snz * Lz);
        float4 q0[4], q1[4], q2[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int j = first + min(u, cnt - 1);
          q0[u] = ms.mrec[3 * j];
          q1[u] = ms.mrec[3 * j + 1];
          q2[u] = ms.mrec[3 * j + 2];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          if (u < cnt && !cull_record(q0[u], q1[u], q2[u], cb) && !ABLATE(3)) {
            const int slot = atomicAdd(&W->n_surv, 1);
            if (slot < COOP_SURV_CAP)
              coop_surv(W)[slot] = make_uint4((unsigned)__float_as_int(q2[u].z), (unsigned)__float_as_int(q2[u].w), it.y, 0u);
            else
              W->overflow[owner] = 1;
          }
        }
      }
    }
    if (n_queued > 0) wave_sync();
    T_MARK(W, lane, 4);
    /* ---- stage 3: all lanes drain the survivor pool (fp64 tests), minimum per owner */
    const int n_surviving = n_queued > 0 ? W->n_surv : 0;
    if (n_surviving > 0) {
      const int n = min(n_surviving, COOP_SURV_CAP);
      for (int idx = lane; idx < n; idx += 64) { /* pass A: distances, atomic minimum of d */
        STAT_LANE(6);
        STAT_WAVE(6);
        const uint4 sv = coop_surv(W)[idx];
        const int owner = (int)(sv.z & 63u);
        const int snx = (int)((sv.z >> 14) & 63u) - 32, sny = (int)((sv.z >> 20) & 63u) - 32, snz = (int)(sv.z >> 26) - 32;
        const double *ry = W->ray[owner];
        const V3 tr0 = {ry[0] - (double)snx * Lx, ry[1] - (double)sny * Ly, ry[2] - (double)snz * Lz};
        const V3 st = {ry[3], ry[4], ry[5]};
        const double *rec = ms.tri + (int64_t)(int)sv.x * 9;
        const double d = ray_triangle_intersection_check(load_v3(rec), load_v3(rec + 3), load_v3(rec + 6), tr0, st);
        coop_surv_d(W)[idx] = d;
        if (d > 0) atomicMin(&W->best_d[owner], (unsigned long long)__double_as_longlong(d));
      }
      wave_sync();
      T_MARK(W, lane, 5);
      for (int idx = lane; idx < n; idx += 64) { /* pass B: among equal d the earliest (visit, position) */
        const uint4 sv = coop_surv(W)[idx];
        const int owner = (int)(sv.z & 63u);
        const double d = coop_surv_d(W)[idx];
        if (d > 0 && (unsigned long long)__double_as_longlong(d) == W->best_d[owner])
          atomicMin(&W->best_key[owner], ((unsigned long long)((sv.z >> 6) & 255u) << 32) | (unsigned long long)sv.y);
      }
      wave_sync();
      for (int idx = lane; idx < n; idx += 64) { /* pass C: the winner publishes its triangle */
        const uint4 sv = coop_surv(W)[idx];
        const int owner = (int)(sv.z & 63u);
        const double d = coop_surv_d(W)[idx];
        if (d > 0 && (unsigned long long)__double_as_longlong(d) == W->best_d[owner] &&
            ((((unsigned long long)((sv.z >> 6) & 255u)) << 32) | (unsigned long long)sv.y) == W->best_key[owner]) {
          W->best_tri[owner] = (int)sv.x;
          /* the record is still in the vector L1: publish E1 x E2 so that the owner does not
           * have to fetch the triangle again before it can reflect */
          const double *rec = ms.tri + (int64_t)(int)sv.x * 9;
          const V3 nn = cross3(load_v3(rec + 3), load_v3(rec + 6));
          double *bn = coop_best_n(W) + 3 * owner;
          bn[0] = nn.x, bn[1] = nn.y, bn[2] = nn.z;
        }
      }
      wave_sync();
      T_MARK(W, lane, 6);
    }
    if (act && !serial_all) {
      if (W->overflow[lane]) {
        /* a survivor was dropped: redo this walker's iteration serially (keeps every candidate) */
        STAT_LANE(9);
        STAT_WAVE(9);
        best.d = DS_INF;
        best.visit = 0x7fffffff;
        best.pos = 0x7fffffff;
        SubvoxelWalk sw; /* recomputed rather than kept in registers through the drain stages */
        CullBox cb0;
        segment_query(ms, r0, step, step_l, xs, ys, zs, sw, cb0);
        enumerate_cells_serial(ms, sw, cb0, r0, step, xs, ys, zs, best);
      } else {
        const unsigned long long bd = W->best_d[lane];
        if (bd != 0x7ff0000000000000ULL) {
          const unsigned long long key = W->best_key[lane];
          const double d_pool = __longlong_as_double((long long)bd);
          const int v_pool = (int)(key >> 32), p_pool = (int)(key & 0xffffffffu);
          offer_hit(best, d_pool, v_pool, p_pool, W->best_tri[lane]);
          if (best.d == d_pool && best.visit == v_pool && best.pos == p_pool) { /* the pool's winner is the overall winner */
            best.have_n = true;
            const double *bn = coop_best_n(W) + 3 * lane;
            best.n.x = bn[0], best.n.y = bn[1], best.n.z = bn[2];
          }
        }
      }
    }
    T_MARK(W, lane, 7);
    /* ---- stage 4: resolve (simulations.py:986-997) */
    if (act) {
      if (best.d > step_l) {
        finished = true; /* check_intersection = False */
      } else {
        bool reflect = p.perm_prob < rng_uniform_f64(rng);
        V3 n = best.n;
        if (!best.have_n) {
          const double *rec = ms.tri + (int64_t)best.tri * 9;
          n = cross3(load_v3(rec + 3), load_v3(rec + 6)); /* (A-B)x(A-C) == E1 x E2 bit for bit */
        }
        normalize3(n);
        if (reflect)
          reflection(r0, step, best.d, n, p.eps);
        else
          crossing(r0, step, best.d, n, p.eps);
        step_l -= best.d;
        if (!(step_l > 0 && iter_idx < p.max_iter)) finished = true; /* the while-condition of the next pass */
      }
    }
    T_MARK(W, lane, 8);
  }
  return finished;
}

#endif /* DSIM_MESH_COOP_H */
