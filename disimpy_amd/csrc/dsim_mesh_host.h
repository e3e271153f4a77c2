/* dsim_mesh_host.h - host side of the mesh index: the device copy of the reference's subvoxel lists, the
 * micro-grid built on top of them and the clearance map.  Shared by the product library (dsim_hip.hip) and the
 * test-only wrapper library (dsim_unit.hip, which runs single passes of the traversal on caller-supplied rays).
 * Include after dsim_kernels.h; host code only.
 */
#ifndef DSIM_MESH_HOST_H
#define DSIM_MESH_HOST_H

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dsim.h"

/* ------------------------------------------------------------------ errors --- */

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                             \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return fail(DSIM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

struct DevMesh {
  double *tri = nullptr;
  int2 *sv_range = nullptr;
  int32_t *tri_idx = nullptr;
  ds_uint4 *mrec = nullptr, *planes = nullptr;
  float qs = 0; /* fixed point of the record boxes: x -> x * qs (MeshDev) */
  int32_t *sv_base = nullptr;
  uint32_t *cell_start = nullptr;
  uint32_t *cell_dense = nullptr; /* lk > 0: a slot block for every subvoxel (MeshDev::cell_dense), or nullptr */
  int lk = 0; /* log2 of the fine cells per subvoxel and axis */
  int32_t *clear_idx = nullptr;
  uint32_t *clear_bits = nullptr;
  int clear_lf = 0;
  double clear_frac = 0; /* share of the voxel's fine cells that are clear (diagnostic) */
  double *bounds = nullptr;
  int nx = 0, ny = 0, nz = 0;
  double inv_h[3] = {0, 0, 0}, inv_l[3] = {0, 0, 0};
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}; /* first and last boundary of every axis */
  bool set = false;
  /* the subvoxel boundaries are what the reference builds - np.linspace(0, L, n + 1) on every axis (substrates.py:506-508) -
   * to within 1e-6 of a subvoxel width: the cooperative traversal's float32 shortcut for the subvoxel ranges, its
   * arithmetic fine-cell coordinates and the clearance map assume it.  Any other grid handed in through the C ABI is
   * walked by the per-lane traversal, which checks every guess against the boundary values themselves. */
  bool coop_ok = false;
};

static void free_mesh(DevMesh &m) {
  if (m.tri) (void)hipFree(m.tri);
  if (m.sv_range) (void)hipFree(m.sv_range);
  if (m.tri_idx) (void)hipFree(m.tri_idx);
  if (m.mrec) (void)hipFree(m.mrec);
  if (m.planes) (void)hipFree(m.planes);
  if (m.sv_base) (void)hipFree(m.sv_base);
  if (m.cell_start) (void)hipFree(m.cell_start);
  if (m.cell_dense) (void)hipFree(m.cell_dense);
  if (m.clear_idx) (void)hipFree(m.clear_idx);
  if (m.clear_bits) (void)hipFree(m.clear_bits);
  if (m.bounds) (void)hipFree(m.bounds);
  m = DevMesh();
}

/* Does the triangle reach the cell that spans the units [abc, abc + span) of a subvoxel cut into nun units per axis
 * (x, y: cells of two half-cell units at every unit, i.e. overlapping; z: one unit per cell)?  Conservative separating-axis test
 * (Akenine-Moller) in double precision against the cell inflated by 1e-6 of its size - far more
 * than any rounding of the test or of micro_cell().  Cells on the border of the grid extend to
 * infinity outwards, because the clamped cell map sends every coordinate beyond the subvoxel to
 * the border cell; for a well-conditioned test the (possibly infinite) cell is first clipped to
 * the triangle's own float32 bounding box, which cannot change the answer.  The bounding box
 * alone lists a triangle in every cell its box touches: on the neuron that is twice as many
 * records per cell as the triangles that actually pass through it. */
static bool triangle_reaches_cell(const double *const tv[3], const float *tlo, const float *thi, const double org[3],
                                  const double scale[3], const int abc[3], const int span[3], const int nun[3]) {
  double c[3], h[3];
  for (int ax = 0; ax < 3; ax++) {
    const double w = 1.0 / scale[ax], m = 1e-6 * w * span[ax];
    double lo = org[ax] + abc[ax] * w - m, hi = org[ax] + (abc[ax] + span[ax]) * w + m;
    const double blo = (double)tlo[ax] - m, bhi = (double)thi[ax] + m;
    if (abc[ax] == 0 || lo < blo) lo = blo;         /* border cell: unbounded below; clip to the triangle's box */
    if (abc[ax] + span[ax] == nun[ax] || hi > bhi) hi = bhi;
    if (!(hi > lo)) return true; /* cannot happen for a cell of the triangle's range; keep the record */
    c[ax] = 0.5 * (lo + hi);
    h[ax] = 0.5 * (hi - lo);
  }
  double v[3][3], e[3][3];
  for (int q = 0; q < 3; q++)
    for (int ax = 0; ax < 3; ax++) v[q][ax] = tv[q][ax] - c[ax];
  for (int ax = 0; ax < 3; ax++) {
    e[0][ax] = v[1][ax] - v[0][ax];
    e[1][ax] = v[2][ax] - v[1][ax];
    e[2][ax] = v[0][ax] - v[2][ax];
  }
  /* the 9 axes unit_i x e_j */
  for (int j = 0; j < 3; j++)
    for (int ax = 0; ax < 3; ax++) {
      double a[3] = {0, 0, 0};
      const int u = (ax + 1) % 3, w = (ax + 2) % 3;
      a[u] = -e[j][w]; /* unit_ax x e = (0, -e_w, e_u) in the (ax, u, w) frame */
      a[w] = e[j][u];
      const double p0 = a[u] * v[0][u] + a[w] * v[0][w], p1 = a[u] * v[1][u] + a[w] * v[1][w], p2 = a[u] * v[2][u] + a[w] * v[2][w];
      const double r = h[u] * std::fabs(a[u]) + h[w] * std::fabs(a[w]);
      const double mn = std::min(p0, std::min(p1, p2)), mx = std::max(p0, std::max(p1, p2));
      if (mn > r || mx < -r) return false;
    }
  /* the box faces */
  for (int ax = 0; ax < 3; ax++) {
    const double mn = std::min(v[0][ax], std::min(v[1][ax], v[2][ax])), mx = std::max(v[0][ax], std::max(v[1][ax], v[2][ax]));
    if (mn > h[ax] || mx < -h[ax]) return false;
  }
  /* the triangle's plane */
  const double n[3] = {e[0][1] * e[1][2] - e[0][2] * e[1][1], e[0][2] * e[1][0] - e[0][0] * e[1][2], e[0][0] * e[1][1] - e[0][1] * e[1][0]};
  const double d = n[0] * v[0][0] + n[1] * v[0][1] + n[2] * v[0][2];
  const double r = h[0] * std::fabs(n[0]) + h[1] * std::fabs(n[1]) + h[2] * std::fabs(n[2]);
  if (std::fabs(d) > r) return false;
  return true;
}

/* Triangle (v0, v1, v2) against the axis-aligned box of centre c and half-widths h: separating-axis
 * test (Akenine-Moller), in double precision on coordinates relative to the box centre.  Used with boxes
 * already inflated by far more than its own rounding, so "false" is certain. */
static bool triangle_meets_box(const double *v0, const double *v1, const double *v2, const double c[3], const double h[3]) {
  double v[3][3], e[3][3];
  for (int ax = 0; ax < 3; ax++) {
    v[0][ax] = v0[ax] - c[ax];
    v[1][ax] = v1[ax] - c[ax];
    v[2][ax] = v2[ax] - c[ax];
  }
  for (int ax = 0; ax < 3; ax++) {
    const double mn = std::min(v[0][ax], std::min(v[1][ax], v[2][ax])), mx = std::max(v[0][ax], std::max(v[1][ax], v[2][ax]));
    if (mn > h[ax] || mx < -h[ax]) return false;
  }
  for (int ax = 0; ax < 3; ax++) {
    e[0][ax] = v[1][ax] - v[0][ax];
    e[1][ax] = v[2][ax] - v[1][ax];
    e[2][ax] = v[0][ax] - v[2][ax];
  }
  for (int j = 0; j < 3; j++)
    for (int ax = 0; ax < 3; ax++) {
      const int u = (ax + 1) % 3, w = (ax + 2) % 3;
      const double au = -e[j][w], aw = e[j][u]; /* unit_ax x e_j */
      const double p0 = au * v[0][u] + aw * v[0][w], p1 = au * v[1][u] + aw * v[1][w], p2 = au * v[2][u] + aw * v[2][w];
      const double r = h[u] * std::fabs(au) + h[w] * std::fabs(aw);
      if (std::min(p0, std::min(p1, p2)) > r || std::max(p0, std::max(p1, p2)) < -r) return false;
    }
  const double n[3] = {e[0][1] * e[1][2] - e[0][2] * e[1][1], e[0][2] * e[1][0] - e[0][0] * e[1][2], e[0][0] * e[1][1] - e[0][1] * e[1][0]};
  const double d = n[0] * v[0][0] + n[1] * v[0][1] + n[2] * v[0][2];
  const double r = h[0] * std::fabs(n[0]) + h[1] * std::fabs(n[1]) + h[2] * std::fabs(n[2]);
  return std::fabs(d) <= r;
}

/* Clearance map.  The reference's reflect loop (simulations.py:924-997) ends a time step without a
 * reflection - and without drawing from the RNG - whenever no triangle is met within the step length:
 * `min_d > step_l`.  For a walker farther than step_l from EVERY triangle that is certain before any
 * look-up.  The map makes that test one bit: every subvoxel is cut into F^3 fine cells (F a power of two
 * <= 16, cells not smaller than half a step); a cell's bit is set when some triangle - in any periodic
 * image: the reference's look-ups wrap around the voxel whatever `periodic` says - meets the cell's box
 * inflated on all sides by
 *     reach = step_l * (1 + 1e-6)    (a hit at fp64 distance d <= step_l lies within that of the start)
 *           + 2e-3 * cell size       (the kernel's cell assignment near a face; rounding of this test).
 * The cube contains the ball, so a clear bit proves `min_d > step_l` for every walker in the cell and
 * every direction; a set bit proves nothing and the walker takes the full pass.  Results are therefore
 * bit-identical with and without the map (tests/test_gpu_production.py compares against the oracle,
 * which knows nothing of it). */
static void build_clearance_map(const double *vertices, const int64_t *faces, int64_t n_faces, const double *const b3[3],
                                const int64_t n_sv[3], double step_l, std::vector<int32_t> &clear_idx,
                                std::vector<uint32_t> &clear_bits, int &lf_out, double &clear_frac) {
  clear_idx.clear();
  clear_bits.clear();
  lf_out = 0;
  clear_frac = 0;
  double L[3], hsub_min = 1e300;
  for (int k = 0; k < 3; k++) {
    L[k] = b3[k][n_sv[k]] - b3[k][0];
    if (!(L[k] > 0) || !std::isfinite(L[k])) return;
    hsub_min = std::min(hsub_min, L[k] / (double)n_sv[k]);
  }
  if (!(step_l >= 0) || !std::isfinite(step_l)) return;
  const int64_t n_sub = n_sv[0] * n_sv[1] * n_sv[2];
  int lf = 4;
  while (lf > 0 && hsub_min / (double)(1 << lf) < 0.5 * step_l) lf--;
  while (lf > 0 && (double)n_sub * (double)(1 << (3 * lf)) / 8.0 > 768e6) lf--; /* dense scratch bitmap <= 768 MB */
  if ((double)n_sub * (double)(1 << (3 * lf)) / 8.0 > 768e6) return;
  const int F = 1 << lf;
  const size_t words = (size_t)std::max(1, (F * F * F) >> 5);
  double hf[3], reach[3];
  int64_t nf[3];
  for (int k = 0; k < 3; k++) {
    nf[k] = n_sv[k] * F;
    hf[k] = L[k] / (double)nf[k];
    reach[k] = step_l * (1.0 + 1e-6) + 2e-3 * hf[k];
  }
  /* work estimate: cells in the inflated bounding boxes; a mesh of huge triangles on a fine grid is not worth it */
  double work = 0;
  for (int64_t f = 0; f < n_faces; f++) {
    double cells = 1;
    for (int k = 0; k < 3; k++) {
      const double a = vertices[faces[f * 3] * 3 + k], b = vertices[faces[f * 3 + 1] * 3 + k], c = vertices[faces[f * 3 + 2] * 3 + k];
      const double ext = std::max(a, std::max(b, c)) - std::min(a, std::min(b, c)) + 2 * reach[k];
      cells *= std::floor(ext / hf[k]) + 3;
    }
    work += cells;
  }
  if (work > 4e9) return;
  std::unique_ptr<std::atomic<uint32_t>[]> dense(new std::atomic<uint32_t>[(size_t)n_sub * words]());
  const unsigned n_thr = (unsigned)std::max<int64_t>(1, std::min<int64_t>(std::min(32u, std::max(1u, std::thread::hardware_concurrency())), n_faces / 64));
  std::atomic<int64_t> next(0);
  auto worker = [&]() {
    for (;;) {
      const int64_t f0 = next.fetch_add(64);
      if (f0 >= n_faces) break;
      for (int64_t f = f0; f < std::min(n_faces, f0 + 64); f++) {
        const double *v0 = &vertices[faces[f * 3] * 3], *v1 = &vertices[faces[f * 3 + 1] * 3], *v2 = &vertices[faces[f * 3 + 2] * 3];
        int64_t lo[3], hi[3];
        for (int k = 0; k < 3; k++) {
          const double mn = std::min(v0[k], std::min(v1[k], v2[k])) - reach[k], mx = std::max(v0[k], std::max(v1[k], v2[k])) + reach[k];
          /* one cell of slack: the SAT test decides.  A range longer than the axis visits a wrapped cell through
           * several of its images - each has to be tested, the triangle may reach only one of them */
          lo[k] = (int64_t)std::floor((mn - b3[k][0]) / hf[k]) - 1;
          hi[k] = (int64_t)std::floor((mx - b3[k][0]) / hf[k]) + 1;
        }
        for (int64_t i = lo[0]; i <= hi[0]; i++)
          for (int64_t j = lo[1]; j <= hi[1]; j++)
            for (int64_t k = lo[2]; k <= hi[2]; k++) {
              const int64_t ijk[3] = {i, j, k};
              double c[3], h[3];
              int64_t w[3];
              for (int a = 0; a < 3; a++) {
                c[a] = b3[a][0] + ((double)ijk[a] + 0.5) * hf[a]; /* the image cell, in the triangle's frame */
                h[a] = 0.5 * hf[a] + reach[a];
                w[a] = ((ijk[a] % nf[a]) + nf[a]) % nf[a];
              }
              if (!triangle_meets_box(v0, v1, v2, c, h)) continue;
              const int64_t sv = ((w[0] >> lf) * n_sv[1] + (w[1] >> lf)) * n_sv[2] + (w[2] >> lf);
              const unsigned bit = (unsigned)(((((w[0] & (F - 1)) << lf) | (w[1] & (F - 1))) << lf) | (w[2] & (F - 1)));
              dense[(size_t)sv * words + (bit >> 5)].fetch_or(1u << (bit & 31), std::memory_order_relaxed);
            }
      }
    }
  };
  if (n_thr <= 1) {
    worker();
  } else {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < n_thr; t++) th.emplace_back(worker);
    for (auto &t : th) t.join();
  }
  clear_idx.assign((size_t)n_sub, -1);
  const uint32_t used_mask = F * F * F >= 32 ? 0xffffffffu : ((1u << (F * F * F)) - 1u);
  double set_bits = 0;
  for (int64_t s = 0; s < n_sub; s++) {
    bool any = false;
    for (size_t w = 0; w < words; w++) any |= (dense[(size_t)s * words + w].load(std::memory_order_relaxed) & used_mask) != 0;
    if (!any) continue;
    clear_idx[(size_t)s] = (int32_t)(clear_bits.size() / words);
    for (size_t w = 0; w < words; w++) {
      const uint32_t v = dense[(size_t)s * words + w].load(std::memory_order_relaxed);
      clear_bits.push_back(v);
      set_bits += __builtin_popcount(v & used_mask);
    }
  }
  if (clear_bits.empty()) clear_bits.push_back(0u);
  lf_out = lf;
  clear_frac = 1.0 - set_bits / ((double)n_sub * F * F * F);
}

/* Bounding box of the part of a triangle that lies inside micro-cell (a, b, c) of a k^3 grid - the cell inflated by
 * 1e-6 of its size and, on the border of the grid, extended to infinity, exactly as triangle_reaches_cell() takes it.
 * A record only has to answer for hits that happen inside ITS cell (a hit point lies in a visited cell in which the
 * triangle is registered, and it is that cell's record that must survive the box test), so the box of the clipped
 * triangle is a valid - and for the long slivers of a dendrite's skin far tighter - replacement of the triangle's
 * own box.  Sutherland-Hodgman clipping in double precision; the result is pushed outwards by another 1e-6 cell
 * sizes (rounding of the clipping) and intersected with the triangle's box.  Returns false (keep the triangle's box)
 * when the clipped polygon comes out empty, which only rounding can cause for a registered triangle. */
static bool clipped_triangle_box(const double *const tv[3], const double org[3], const double scale[3], const int abc[3],
                                 const int span[3], const int nun[3], double lo[3], double hi[3]) {
  double poly[16][3], tmp[16][3];
  int n = 3;
  for (int q = 0; q < 3; q++)
    for (int ax = 0; ax < 3; ax++) poly[q][ax] = tv[q][ax];
  for (int ax = 0; ax < 3 && n > 0; ax++) {
    const double w = 1.0 / scale[ax], m = 1e-6 * w * span[ax];
    for (int side = 0; side < 2 && n > 0; side++) {
      if ((side == 0 && abc[ax] == 0) || (side == 1 && abc[ax] + span[ax] == nun[ax])) continue; /* border cell: unbounded */
      const double bound = side == 0 ? org[ax] + abc[ax] * w - m : org[ax] + (abc[ax] + span[ax]) * w + m;
      int out = 0;
      for (int q = 0; q < n; q++) {
        const double *p0 = poly[q], *p1 = poly[(q + 1) % n];
        const double d0 = side == 0 ? p0[ax] - bound : bound - p0[ax], d1 = side == 0 ? p1[ax] - bound : bound - p1[ax];
        if (d0 >= 0) {
          for (int e = 0; e < 3; e++) tmp[out][e] = p0[e];
          out++;
        }
        if ((d0 >= 0) != (d1 >= 0)) {
          const double t = d0 / (d0 - d1);
          for (int e = 0; e < 3; e++) tmp[out][e] = p0[e] + t * (p1[e] - p0[e]);
          tmp[out][ax] = bound;
          out++;
        }
      }
      n = out;
      for (int q = 0; q < n; q++)
        for (int e = 0; e < 3; e++) poly[q][e] = tmp[q][e];
    }
  }
  if (n == 0) return false;
  for (int ax = 0; ax < 3; ax++) {
    const double m = 1e-6 * span[ax] / scale[ax];
    lo[ax] = hi[ax] = poly[0][ax];
    for (int q = 1; q < n; q++) {
      lo[ax] = std::min(lo[ax], poly[q][ax]);
      hi[ax] = std::max(hi[ax], poly[q][ax]);
    }
    lo[ax] -= m;
    hi[ax] += m;
  }
  return true;
}

/* a record of the fine grid before it is packed: float32 box {min x y z, max x y z} of the triangle's part inside the
 * cell, float32 supporting plane n.x = c, triangle, position in the reference's list */
struct HostRecord {
  float lo[3], hi[3];
  float n[3], c;
  int32_t tri, pos;
};

/* fixed point of the record boxes (MeshDev::mrec): quanta per metre as the FLOAT32 constant the kernels use */
static float record_quantum(const double *const b3[3], const int64_t n_sv[3]) {
  double far = 0;
  for (int k = 0; k < 3; k++) far = std::max(far, b3[k][n_sv[k]]);
  const float qs = far > 0 && std::isfinite(far) ? (float)(32767.0 / far) : 0.0f;
  return std::isfinite(qs) ? qs : 0.0f; /* 0: every box becomes [0, 0] and nothing is culled by it */
}

/* [r6] Device records from the float32 ones: 16 bytes per record {box, triangle} - lo rounded down, hi rounded up in the EXACT
 * value of x * qs (double arithmetic on the float32 constant; 1e-6 quanta of slack for its own rounding), clamped into
 * [0, 32767] - and, ONCE per triangle, its float32 supporting plane {nx, ny, nz, c} (rounds 2-5 carried the plane in every
 * record: the same 16 bytes in each of the dozen cells that list a triangle) */
static void pack_records(const std::vector<HostRecord> &recs, float qs, size_t n_triangles, std::vector<uint4> &out, std::vector<uint4> &planes) {
  /* + 8 records of padding (they name triangle 0): the cooperative traversal loads an item's full width whatever its count (dsim_mesh_coop.h, stage 2) */
  out.assign(recs.size() + 8, make_uint4(0u, 0u, 0u, 0u));
  planes.assign(std::max<size_t>(n_triangles, 1), make_uint4(0u, 0u, 0u, 0u));
  auto q = [&](float x, bool up) -> uint32_t {
    const double v = (double)x * (double)qs;
    double r = up ? std::ceil(v + 1e-6) : std::floor(v - 1e-6);
    if (!(r >= 0.0)) r = 0.0; /* NaN too */
    if (r > 32767.0) r = 32767.0;
    return (uint32_t)r;
  };
  for (size_t i = 0; i < recs.size(); i++) {
    const HostRecord &r = recs[i];
    const uint32_t lx = q(r.lo[0], false), ly = q(r.lo[1], false), lz = q(r.lo[2], false);
    const uint32_t hx = 32767u - q(r.hi[0], true), hy = 32767u - q(r.hi[1], true), hz = 32767u - q(r.hi[2], true);
    uint32_t nb[4];
    memcpy(nb, r.n, 12);
    memcpy(nb + 3, &r.c, 4);
    out[i] = make_uint4(lx | (ly << 16), lz | (hx << 16), hy | (hz << 16), (uint32_t)r.tri);
    planes[(size_t)r.tri] = make_uint4(nb[0], nb[1], nb[2], nb[3]); /* the same for every record of the triangle */
  }
}

/* the fine grid as built on the host (dsim_mesh_fine_grid hands it out so that its completeness can be tested without a GPU) */
struct HostFineGrid {
  int lk = 0;
  std::vector<int32_t> sv_base;
  std::vector<uint32_t> cell_start;
  std::vector<HostRecord> recs;
  std::vector<uint4> mrec, planes; /* recs as the device gets them: 16-byte records and the planes by triangle */
  float qs = 0;
};

/* everything dsim_set_mesh puts on a device, as built on the host: built ONCE per call and uploaded to every shard of a
 * multi-device handle (round 4 had every shard build its own: G times the host work, G x 32 threads, G temporary copies) */
struct HostMesh {
  std::vector<double> rec;
  std::vector<int2> ranges;
  std::vector<int32_t> refs, sv_base, cidx;
  std::vector<uint32_t> cell_start, recs_before, cbits;
  std::vector<uint4> packed, planes; /* 16-byte records (+ padding), float32 planes by triangle */
  std::vector<double> bounds;
  int lk = 0, clear_lf = 0;
  double clear_frac = 0, step_l = 0;
  float qs = 0;
  int64_t n_sub = 0;
  int n[3] = {0, 0, 0};
  double inv_h[3] = {0, 0, 0}, inv_l[3] = {0, 0, 0}, lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  bool coop_ok = false;
};

static int upload_mesh_host(DevMesh &dm, hipStream_t stream, const HostMesh &hm);

static int upload_mesh(DevMesh &dm, hipStream_t stream, const double *vertices, int64_t n_vertices, const int64_t *faces,
                       int64_t n_faces, const double *xs, const double *ys, const double *zs, const int64_t n_sv[3],
                       const int64_t *subvoxel_indices, const int64_t *triangle_indices, int64_t n_refs, double step_l,
                       HostFineGrid *host_only = nullptr, HostMesh *build_only = nullptr, bool want_clearance = true) {
  if (!vertices || !faces || !xs || !ys || !zs || !n_sv || !subvoxel_indices || (!triangle_indices && n_refs))
    return fail(DSIM_ERR_INVALID, "mesh: null argument");
  if (n_faces <= 0 || n_vertices <= 0 || n_faces > INT32_MAX || n_refs > INT32_MAX || n_refs < 0)
    return fail(DSIM_ERR_INVALID, "mesh: bad sizes");
  for (int k = 0; k < 3; k++)
    if (n_sv[k] < 1 || n_sv[k] > 4096) return fail(DSIM_ERR_INVALID, "mesh: n_sv[%d] = %lld out of range", k, (long long)n_sv[k]);
  const int64_t n_sub = n_sv[0] * n_sv[1] * n_sv[2];
  if (n_sub > INT32_MAX) return fail(DSIM_ERR_INVALID, "mesh: too many subvoxels");
  if (!host_only && !build_only) free_mesh(dm);
  HostMesh local_hm;
  HostMesh &hm = build_only ? *build_only : local_hm;
  hm = HostMesh();
  /* pre-gathered triangle records {A, B - A, C - A} (simulations.py:259-262 does these subtractions per candidate) */
  std::vector<double> &rec = hm.rec;
  rec.assign((size_t)n_faces * DS_TRI_STRIDE, 0.0);
  for (int64_t f = 0; f < n_faces; f++) {
    int64_t ia = faces[f * 3], ib = faces[f * 3 + 1], ic = faces[f * 3 + 2];
    if (ia < 0 || ib < 0 || ic < 0 || ia >= n_vertices || ib >= n_vertices || ic >= n_vertices)
      return fail(DSIM_ERR_INVALID, "mesh: face %lld references a vertex out of range", (long long)f);
    for (int k = 0; k < 3; k++) {
      double a = vertices[ia * 3 + k];
      rec[f * DS_TRI_STRIDE + k] = a;
      rec[f * DS_TRI_STRIDE + 3 + k] = vertices[ib * 3 + k] - a;
      rec[f * DS_TRI_STRIDE + 6 + k] = vertices[ic * 3 + k] - a;
    }
    /* unit normal = normalize((A-B) x (A-C)) (simulations.py:77-97) with the device's own operations: cross3 and
     * dot3 in the fma convention of dsim_device.h, correctly rounded sqrt (= ds_sqrt_dev, tested over the whole
     * range), ds_div3 (same source on host and device).  A degenerate triangle gets the same NaNs as on the device. */
    {
      const double *e1 = &rec[f * DS_TRI_STRIDE + 3], *e2 = &rec[f * DS_TRI_STRIDE + 6];
      double nx = std::fma(e1[1], e2[2], -(e1[2] * e2[1]));
      double ny = std::fma(e1[2], e2[0], -(e1[0] * e2[2]));
      double nz = std::fma(e1[0], e2[1], -(e1[1] * e2[0]));
      const double len = std::sqrt(std::fma(nz, nz, std::fma(nx, nx, ny * ny)));
      ds_div3(&nx, &ny, &nz, len);
      rec[f * DS_TRI_STRIDE + 9] = nx;
      rec[f * DS_TRI_STRIDE + 10] = ny;
      rec[f * DS_TRI_STRIDE + 11] = nz;
    }
  }
  std::vector<int2> &ranges = hm.ranges;
  ranges.assign((size_t)n_sub, make_int2(0, 0));
  for (int64_t s = 0; s < n_sub; s++) {
    int64_t a = subvoxel_indices[s * 2], b = subvoxel_indices[s * 2 + 1];
    if (a < 0 || b < a || b > n_refs) return fail(DSIM_ERR_INVALID, "mesh: subvoxel %lld has a bad range", (long long)s);
    ranges[s] = make_int2((int)a, (int)b);
  }

  std::vector<int32_t> &refs = hm.refs;
  refs.assign((size_t)std::max<int64_t>(n_refs, 1), 0);
  for (int64_t i = 0; i < n_refs; i++) {
    if (triangle_indices[i] < 0 || triangle_indices[i] >= n_faces) return fail(DSIM_ERR_INVALID, "mesh: triangle index out of range");
    refs[i] = (int32_t)triangle_indices[i];
  }
  /* Fine grid.  The reference's candidate set of a subvoxel is its whole list (up to 88 triangles per 4.3 um
   * subvoxel on the neuron), and a wavefront pays for its worst lane.  Every NON-EMPTY reference subvoxel is
   * therefore cut into cells, k = 2^lk per axis the same for the whole mesh, so that a walker can name the cells its
   * segment touches by arithmetic alone (no look-up per subvoxel before the cells are known).  Cell c of subvoxel s
   * lists the references OF s whose triangle reaches c; the border cells of a subvoxel extend to infinity outwards
   * (the query clamps its cell range into the subvoxel), so whatever the reference lists in s is found through s -
   * the exactness argument never leans on the reference's index being geometrically complete.
   *
   * [r4] Along x and y the cells OVERLAP: an axis of a subvoxel is cut into 2k half-cell "units" and there is a cell
   * of two units at EVERY unit (cell i = units i and i + 1, i = 0 .. 2k - 2).  A segment's box is narrower than a unit
   * on almost every pass (units are >= 0.75 step lengths and the box of a segment is |direction cosine| x what is left
   * of the step), so it covers at most two units per axis and lies inside ONE cell: one run of records per pass instead
   * of the 1.6 the disjoint cells of round 3 needed (a box that straddles a cell face had to read both neighbours'
   * lists: 13.8 records and 1.6 look-ups per pass on the refined neuron, 2.2 runs per step; now the list of the one
   * cell around it).  Boxes wider than two units take cells i, i + 2, ...  The price is memory: (2k)^2 k cell slots per
   * non-empty subvoxel (slot 2k - 1 of an axis is never used: shifts instead of multiplications) and about 2.5 times
   * the records.  Along z the cells are disjoint, stored fastest with their records back to back, so a run of cells
   * along z is ONE contiguous range of records.  Records (packed to 16 bytes by pack_records, the plane kept once per triangle)
   * {AABB of the triangle's part inside the cell, supporting plane in float32, triangle, position in the reference's list}:
   *     sv_base[s]           first cell of subvoxel s in cell_start, -1 for an empty subvoxel
   *                          (lk == 0: not used, cell_start is indexed by s itself, nothing overlaps)
   *     cell                 sv_base[s] + ((i * 2k + j) * k + c)
   *     cell_start[c .. c+r] records of the r cells c .. c+r-1: [cell_start[c], cell_start[c+r]) of mrec */
  double inv_h3[3];
  {
    const double *b3[3] = {xs, ys, zs};
    for (int k = 0; k < 3; k++) {
      double L = b3[k][n_sv[k]] - b3[k][0];
      inv_h3[k] = L > 0 ? (double)n_sv[k] / L : 0.0;
    }
  }
  double sub_min = 1e300; /* smallest subvoxel edge */
  for (int k = 0; k < 3; k++)
    if (inv_h3[k] > 0) sub_min = std::min(sub_min, 1.0 / inv_h3[k]);
  /* k by the density of the lists a walker meets (the mean list length seen from a list entry) with the thresholds
   * tuned in round 2 on the neuron and the packed cylinders (more than 2 / 4 / 16 entries: k = 2 / 4 / 8), but no
   * cell smaller than `min_cell` step lengths (a segment that spans several cells per axis multiplies the runs)
   * and no more than 2^30 cells in all.  DSIM_MICRO="t1,t2,t3,t4,min_cell" overrides thresholds and min_cell in
   * tuning runs - results never depend on them. */
  int micro_t[4] = {2, 4, 16, 1 << 30};
  double micro_min_cell = 1.5;
  if (const char *env = getenv("DSIM_MICRO")) {
    int a, b, c, d;
    double m;
    if (sscanf(env, "%d,%d,%d,%d,%lf", &a, &b, &c, &d, &m) == 5 && a >= 0 && b >= a && c >= b && d >= c && m > 0) {
      micro_t[0] = a, micro_t[1] = b, micro_t[2] = c, micro_t[3] = d;
      micro_min_cell = m;
    }
  }
  int64_t n_nonempty = 0;
  double sum_n = 0, sum_n2 = 0;
  for (int64_t s = 0; s < n_sub; s++) {
    const double n_s = ranges[s].y - ranges[s].x;
    if (n_s <= 0) continue;
    n_nonempty++;
    sum_n += n_s;
    sum_n2 += n_s * n_s;
  }
  const double met = sum_n > 0 ? sum_n2 / sum_n : 0.0;
  int lk = met <= micro_t[0] ? 0 : (met <= micro_t[1] ? 1 : (met <= micro_t[2] ? 2 : (met <= micro_t[3] ? 3 : 4)));
  while (lk > 0 && sub_min / (double)(1 << lk) < micro_min_cell * step_l) lk--;
  /* cell slots per subvoxel: 4 k^3 (x and y in 2k units) */
  while (lk > 0 && ((double)n_nonempty * 4.0 * (double)(1 << (3 * lk)) > (double)(1 << 30) || (double)n_sub * 4.0 * (double)(1 << (3 * lk)) >= 4294967296.0)) lk--;
  const int k = 1 << lk;
  /* units, cell span and cell slots per axis */
  const int nun[3] = {lk > 0 ? 2 * k : 1, lk > 0 ? 2 * k : 1, k}, span[3] = {lk > 0 ? 2 : 1, lk > 0 ? 2 : 1, 1};
  const size_t slots = (size_t)nun[0] * nun[1] * nun[2];
  std::vector<int32_t> &sv_base = hm.sv_base;
  sv_base.assign((size_t)(lk > 0 ? n_sub : 1), -1);
  std::vector<uint32_t> &cell_start = hm.cell_start;
  std::vector<HostRecord> mrec;
  std::vector<uint32_t> &recs_before = hm.recs_before;
  recs_before.assign((size_t)(lk > 0 ? n_sub + 1 : 1), 0u); /* records of the subvoxels before s (dense table) */
  cell_start.reserve((size_t)(lk > 0 ? n_nonempty * slots : n_sub) + 1);
  {
    /* per non-empty subvoxel: the record count of every slot and the records, built by a few host threads (the subvoxels are
     * independent), then strung together in subvoxel order */
    struct Built {
      std::vector<uint32_t> count; /* [slots] */
      std::vector<HostRecord> recs;
    };
    std::vector<int64_t> todo;
    for (int64_t s = 0; s < n_sub; s++)
      if (ranges[s].y > ranges[s].x) todo.push_back(s);
    std::vector<Built> built(todo.size());
    auto build_one = [&](size_t tn, std::vector<std::vector<int32_t>> &cells, std::vector<float> &lo3, std::vector<float> &hi3) {
      const int64_t s = todo[tn];
      Built &out = built[tn];
      const int begin = ranges[s].x, end = ranges[s].y, n_s = end - begin;
      const int64_t sx = s / (n_sv[1] * n_sv[2]), sy = (s / n_sv[2]) % n_sv[1], sz = s % n_sv[2];
      const double org[3] = {xs[sx], ys[sy], zs[sz]};
      const double scale[3] = {(double)nun[0] * inv_h3[0], (double)nun[1] * inv_h3[1], (double)nun[2] * inv_h3[2]}; /* units per metre */
      cells.assign(slots, std::vector<int32_t>());
      lo3.assign((size_t)n_s * 3, 0.f);
      hi3.assign((size_t)n_s * 3, 0.f);
      for (int i = begin; i < end; i++) {
        const int64_t *fc = &faces[(size_t)refs[i] * 3];
        int c0[3], c1[3];
        for (int ax = 0; ax < 3; ax++) {
          double a = vertices[fc[0] * 3 + ax], b = vertices[fc[1] * 3 + ax], c = vertices[fc[2] * 3 + ax];
          double mn = std::min(a, std::min(b, c)), mx = std::max(a, std::max(b, c));
          float lo = std::nextafter((float)mn, -std::numeric_limits<float>::infinity());
          float hi = std::nextafter((float)mx, std::numeric_limits<float>::infinity());
          lo3[(size_t)(i - begin) * 3 + ax] = lo;
          hi3[(size_t)(i - begin) * 3 + ax] = hi;
          /* units the triangle's box touches -> the cells that hold one of them: cell q spans units q .. q + span - 1 */
          const int u0 = micro_cell((double)lo, org[ax], scale[ax], nun[ax]), u1 = micro_cell((double)hi, org[ax], scale[ax], nun[ax]);
          c0[ax] = std::max(0, u0 - (span[ax] - 1));
          c1[ax] = std::min(u1, nun[ax] - span[ax]);
        }
        const double *tv[3] = {&vertices[fc[0] * 3], &vertices[fc[1] * 3], &vertices[fc[2] * 3]};
        const float *tlo = &lo3[(size_t)(i - begin) * 3], *thi = &hi3[(size_t)(i - begin) * 3];
        for (int a = c0[0]; a <= c1[0]; a++)
          for (int b = c0[1]; b <= c1[1]; b++)
            for (int c = c0[2]; c <= c1[2]; c++) {
              const int abc[3] = {a, b, c};
              if (k > 1 && !triangle_reaches_cell(tv, tlo, thi, org, scale, abc, span, nun)) continue;
              cells[((size_t)a * nun[1] + b) * nun[2] + c].push_back(i);
            }
      }
      out.count.assign(slots, 0u);
      for (size_t ci = 0; ci < cells.size(); ci++) {
        auto &cl = cells[ci];
        const int abc[3] = {(int)(ci / ((size_t)nun[1] * nun[2])), (int)((ci / nun[2]) % nun[1]), (int)(ci % nun[2])};
        out.count[ci] = (uint32_t)cl.size();
        for (int32_t i : cl) {
          float lo[3], hi[3];
          for (int ax = 0; ax < 3; ax++) lo[ax] = lo3[(size_t)(i - begin) * 3 + ax], hi[ax] = hi3[(size_t)(i - begin) * 3 + ax];
          if (k > 1) { /* the box of the triangle's part inside this cell, float32 rounded outwards, never beyond the triangle's own */
            const int64_t *fc3 = &faces[(size_t)refs[i] * 3];
            const double *tv[3] = {&vertices[fc3[0] * 3], &vertices[fc3[1] * 3], &vertices[fc3[2] * 3]};
            double clo[3], chi[3];
            if (clipped_triangle_box(tv, org, scale, abc, span, nun, clo, chi))
              for (int ax = 0; ax < 3; ax++) {
                const float l = std::nextafter((float)clo[ax], -std::numeric_limits<float>::infinity());
                const float h2 = std::nextafter((float)chi[ax], std::numeric_limits<float>::infinity());
                lo[ax] = std::max(lo[ax], l);
                hi[ax] = std::min(hi[ax], h2);
              }
          }
          int32_t ti = refs[i];
          /* supporting plane n.x = c in float32 (unit normal; c from the ROUNDED normal so that the
           * float32 plane is self-consistent); a degenerate triangle gets n = 0, c = 0 and is never culled */
          const double *r = &rec[(size_t)ti * DS_TRI_STRIDE];
          double nx = r[4] * r[8] - r[5] * r[7], ny = r[5] * r[6] - r[3] * r[8], nz = r[3] * r[7] - r[4] * r[6];
          double nn = std::sqrt(nx * nx + ny * ny + nz * nz);
          float fx = 0.f, fy = 0.f, fz = 0.f, fc = 0.f;
          if (nn > 0 && std::isfinite(nn)) {
            fx = (float)(nx / nn);
            fy = (float)(ny / nn);
            fz = (float)(nz / nn);
            fc = (float)((double)fx * r[0] + (double)fy * r[1] + (double)fz * r[2]);
          }
          out.recs.push_back(HostRecord{{lo[0], lo[1], lo[2]}, {hi[0], hi[1], hi[2]}, {fx, fy, fz}, fc, ti, i});
        }
      }
    };
    {
      const unsigned n_thr = (unsigned)std::max<int64_t>(1, std::min<int64_t>(std::min(32u, std::max(1u, std::thread::hardware_concurrency())), (int64_t)todo.size() / 16));
      std::atomic<size_t> next(0);
      auto worker = [&]() {
        std::vector<std::vector<int32_t>> cells;
        std::vector<float> lo3, hi3;
        for (;;) {
          const size_t tn = next.fetch_add(1);
          if (tn >= todo.size()) break;
          build_one(tn, cells, lo3, hi3);
        }
      };
      if (n_thr <= 1) {
        worker();
      } else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < n_thr; t++) th.emplace_back(worker);
        for (auto &t : th) t.join();
      }
    }
    size_t tn = 0;
    for (int64_t s = 0; s < n_sub; s++) {
      if (lk > 0) recs_before[(size_t)s] = (uint32_t)mrec.size();
      if (ranges[s].y <= ranges[s].x) {
        if (lk == 0) cell_start.push_back((uint32_t)mrec.size()); /* dense over the subvoxels: an empty range */
        continue;
      }
      const Built &bu = built[tn++];
      if (lk > 0) sv_base[s] = (int32_t)cell_start.size();
      uint32_t at = (uint32_t)mrec.size();
      for (size_t ci = 0; ci < slots; ci++) {
        cell_start.push_back(at);
        at += bu.count[ci];
      }
      mrec.insert(mrec.end(), bu.recs.begin(), bu.recs.end());
    }
    if (mrec.size() + 8 >= 0x10000000ull) return fail(DSIM_ERR_INVALID, "mesh: too many fine-grid records"); /* the cooperative traversal names a record by a 32-bit byte offset (16 bytes each) */
    cell_start.push_back((uint32_t)mrec.size()); /* end of the last cell */
    if (lk > 0) recs_before[(size_t)n_sub] = (uint32_t)mrec.size();
  }
  float qs;
  {
    const double *b3[3] = {xs, ys, zs};
    qs = record_quantum(b3, n_sv);
  }
  std::vector<uint4> &packed = hm.packed;
  pack_records(mrec, qs, (size_t)n_faces, packed, hm.planes);
  if (host_only) { /* no device involved */
    host_only->lk = lk;
    host_only->sv_base.swap(sv_base);
    host_only->cell_start.swap(cell_start);
    host_only->recs.swap(mrec);
    host_only->mrec.swap(packed);
    host_only->planes.swap(hm.planes);
    host_only->qs = qs;
    return DSIM_OK;
  }
  std::vector<double> &bounds = hm.bounds;
  bounds.insert(bounds.end(), xs, xs + n_sv[0] + 1);
  bounds.insert(bounds.end(), ys, ys + n_sv[1] + 1);
  bounds.insert(bounds.end(), zs, zs + n_sv[2] + 1);
  hm.step_l = step_l;
  hm.lk = lk;
  hm.qs = qs;
  hm.n_sub = n_sub;
  if (step_l > 0 && want_clearance) { /* the walk's clearance map (not used by the fill kernel, which passes step_l = 0, nor by the cost model) */
    const double *bb[3] = {xs, ys, zs};
    build_clearance_map(vertices, faces, n_faces, bb, n_sv, step_l, hm.cidx, hm.cbits, hm.clear_lf, hm.clear_frac);
  }
  const double *b3[3] = {xs, ys, zs};
  for (int k = 0; k < 3; k++) {
    double L = b3[k][n_sv[k]] - b3[k][0];
    hm.n[k] = (int)n_sv[k];
    hm.inv_h[k] = L > 0 ? (double)n_sv[k] / L : 0.0;
    hm.inv_l[k] = L > 0 ? 1.0 / L : 0.0;
    hm.lo[k] = b3[k][0];
    hm.hi[k] = b3[k][n_sv[k]];
  }
  hm.coop_ok = true;
  for (int k = 0; k < 3; k++) {
    const double L = b3[k][n_sv[k]] - b3[k][0], h = L / (double)n_sv[k];
    if (!(L > 0) || !std::isfinite(L) || b3[k][0] != 0.0) hm.coop_ok = false;
    for (int64_t i = 0; i <= n_sv[k] && hm.coop_ok; i++)
      if (!(std::fabs(b3[k][i] - (double)i * h) <= 1e-6 * h)) hm.coop_ok = false;
  }
  if (build_only) return DSIM_OK;
  return upload_mesh_host(dm, stream, hm);
}

/* the host arrays of one mesh onto one device (the current device is the caller's business) */
static int upload_mesh_host(DevMesh &dm, hipStream_t stream, const HostMesh &hm) {
  free_mesh(dm);
  dm.clear_lf = hm.clear_lf;
  dm.clear_frac = hm.clear_frac;
  if (!hm.cidx.empty()) {
    HIP_TRY(hipMalloc(&dm.clear_idx, hm.cidx.size() * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&dm.clear_bits, hm.cbits.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpyAsync(dm.clear_idx, hm.cidx.data(), hm.cidx.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(dm.clear_bits, hm.cbits.data(), hm.cbits.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
  }
  HIP_TRY(hipMalloc(&dm.tri, hm.rec.size() * sizeof(double)));
  HIP_TRY(hipMalloc(&dm.sv_range, hm.ranges.size() * sizeof(int2)));
  HIP_TRY(hipMalloc(&dm.tri_idx, hm.refs.size() * sizeof(int32_t)));
  HIP_TRY(hipMalloc(&dm.mrec, hm.packed.size() * sizeof(uint4)));
  HIP_TRY(hipMalloc(&dm.planes, hm.planes.size() * sizeof(uint4)));
  HIP_TRY(hipMemcpyAsync(dm.planes, hm.planes.data(), hm.planes.size() * sizeof(uint4), hipMemcpyHostToDevice, stream));
  HIP_TRY(hipMalloc(&dm.sv_base, hm.sv_base.size() * sizeof(int32_t)));
  HIP_TRY(hipMalloc(&dm.cell_start, hm.cell_start.size() * sizeof(uint32_t)));
  HIP_TRY(hipMemcpyAsync(dm.mrec, hm.packed.data(), hm.packed.size() * sizeof(uint4), hipMemcpyHostToDevice, stream));
  dm.qs = hm.qs;
  HIP_TRY(hipMemcpyAsync(dm.sv_base, hm.sv_base.data(), hm.sv_base.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
  HIP_TRY(hipMemcpyAsync(dm.cell_start, hm.cell_start.data(), hm.cell_start.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
  dm.lk = hm.lk;
  if (hm.lk > 0 && hm.step_l > 0) { /* (the fill session passes step_l = 0: it never walks the fine grid)
     * [r5] the cooperative traversal's one-level table: a slot block for EVERY subvoxel (C5b: 125 000 subvoxels x 2 048
     * slots x 4 bytes = 1 GB of which the 2 280 non-empty blocks are ever read; 288 GB of HBM per GPU afford it), expanded
     * on the device from the compact arrays.  DSIM_DENSE_CELLS_MB caps it (default 4096, 0 = never: two-level look-ups) */
    const int lcs = 3 * hm.lk + 2;
    const uint64_t n_slots = (uint64_t)hm.n_sub << lcs;
    double cap_mb = 4096;
    if (const char *e = getenv("DSIM_DENSE_CELLS_MB")) cap_mb = atof(e);
    if ((double)(n_slots + 1) * 4.0 <= cap_mb * 1048576.0 && n_slots + 1 < (1ull << 30)) { /* (slots are named by 32-bit byte offsets) */
      /* The table is an optimisation: the two-level look-up (cell_dense == nullptr) is complete without it.  When its memory
       * is not to be had - several shards or ranks on one device multiply it - the error is cleared and the shard goes on
       * with the compact arrays; before_d is freed on every path. */
      uint32_t *before_d = nullptr;
      hipError_t e = hipMalloc(&dm.cell_dense, (size_t)(n_slots + 1) * sizeof(uint32_t));
      if (e == hipSuccess) e = hipMalloc(&before_d, hm.recs_before.size() * sizeof(uint32_t));
      if (e == hipSuccess) e = hipMemcpyAsync(before_d, hm.recs_before.data(), hm.recs_before.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream);
      if (e == hipSuccess) {
        expand_cells_kernel<<<(unsigned)((n_slots + 256) / 256), 256, 0, stream>>>(dm.sv_base, dm.cell_start, before_d, lcs, n_slots, dm.cell_dense);
        e = hipGetLastError();
      }
      if (e == hipSuccess) e = hipStreamSynchronize(stream);
      if (before_d) (void)hipFree(before_d);
      if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError(); /* clear the sticky error: the run continues on the two-level path */
        if (dm.cell_dense) (void)hipFree(dm.cell_dense);
        dm.cell_dense = nullptr;
      } else if (e != hipSuccess) {
        return fail(DSIM_ERR_HIP, "dense cell table: %s", hipGetErrorString(e));
      }
    }
  }
  HIP_TRY(hipMalloc(&dm.bounds, hm.bounds.size() * sizeof(double)));
  HIP_TRY(hipMemcpyAsync(dm.tri, hm.rec.data(), hm.rec.size() * sizeof(double), hipMemcpyHostToDevice, stream));
  HIP_TRY(hipMemcpyAsync(dm.sv_range, hm.ranges.data(), hm.ranges.size() * sizeof(int2), hipMemcpyHostToDevice, stream));
  HIP_TRY(hipMemcpyAsync(dm.tri_idx, hm.refs.data(), hm.refs.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
  HIP_TRY(hipMemcpyAsync(dm.bounds, hm.bounds.data(), hm.bounds.size() * sizeof(double), hipMemcpyHostToDevice, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  dm.nx = hm.n[0];
  dm.ny = hm.n[1];
  dm.nz = hm.n[2];
  for (int k = 0; k < 3; k++) dm.inv_h[k] = hm.inv_h[k], dm.inv_l[k] = hm.inv_l[k], dm.lo[k] = hm.lo[k], dm.hi[k] = hm.hi[k];
  dm.set = true;
  dm.coop_ok = hm.coop_ok && hm.rec.size() / DS_TRI_STRIDE < (1u << 25); /* the cooperative traversal names a triangle by a 32-bit byte offset */
  return DSIM_OK;
}

static MeshDev mesh_dev(const DevMesh &dm) {
  MeshDev m;
  m.tri = dm.tri;
  m.sv_range = dm.sv_range;
  m.tri_idx = dm.tri_idx;
  m.mrec = dm.mrec;
  m.planes = dm.planes;
  m.qs = dm.qs;
  m.sv_base = dm.sv_base;
  m.cell_start = dm.cell_start;
  m.cell_dense = dm.cell_dense;
  m.lk = dm.lk;
  m.bounds = dm.bounds;
  m.clear_idx = dm.clear_idx;
  m.clear_bits = dm.clear_bits;
  m.clear_lf = dm.clear_lf;
  m.nx = dm.nx;
  m.ny = dm.ny;
  m.nz = dm.nz;
  m.inv_nx = 1.0f / (float)std::max(1, dm.nx), m.inv_ny = 1.0f / (float)std::max(1, dm.ny), m.inv_nz = 1.0f / (float)std::max(1, dm.nz);
  m.pad_ = 0.f;
  m.inv_hx = dm.inv_h[0];
  m.inv_hy = dm.inv_h[1];
  m.inv_hz = dm.inv_h[2];
  m.inv_lx = dm.inv_l[0];
  m.inv_ly = dm.inv_l[1];
  m.inv_lz = dm.inv_l[2];
  m.x0 = dm.lo[0], m.y0 = dm.lo[1], m.z0 = dm.lo[2];
  m.x1 = dm.hi[0], m.y1 = dm.hi[1], m.z1 = dm.hi[2];
  return m;
}

#endif /* DSIM_MESH_HOST_H */
