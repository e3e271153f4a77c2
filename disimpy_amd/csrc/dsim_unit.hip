/* dsim_unit.hip - TEST-ONLY build (libdsim_unit.so, not part of the product ABI of
 * include/dsim.h): wrapper kernels that evaluate the per-walker DEVICE functions of
 * dsim_device.h / dsim_math.h on arrays, one thread per input, so that
 * tests/test_gpu_device_units.py can compare them on the GPU (i) bit for bit with the CPU
 * oracle's restatement on millions of random and edge inputs and (ii) with the known
 * answers of the reference's own wrapper-kernel tests (disimpy/tests/test_simulations.py:23-360).
 *
 * The op numbers are shared with dso_ua_eval() in oracle/dsim_oracle.c.  Inputs and outputs
 * are column arrays of doubles (in[c * n + i]); integers travel as doubles, 64-bit patterns
 * as the bit pattern of a double.  xs = subvoxel boundaries (ops 8-10).
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>
#include <vector>

#include "dsim_kernels.h"
#include "dsim_mesh_host.h"

#define DSU_MAX_COLS 16

struct UnitArgs {
  int op;
  long long n;
  const double *in;
  double *out;
  const double *xs;
  int n_xs;
};

__global__ __launch_bounds__(256) void unit_kernel(const UnitArgs a) {
  __shared__ double log_tab[DS_LOG_TABLE_DOUBLES];
  __shared__ double cos_tab[DS_COS_TABLE_DOUBLES];
  for (int i = threadIdx.x; i < DS_LOG_TABLE_DOUBLES; i += blockDim.x) log_tab[i] = ds_log_table_dev[i];
  for (int i = threadIdx.x; i < DS_COS_TABLE_DOUBLES; i += blockDim.x) cos_tab[i] = ds_cos_table_dev[i];
  __syncthreads();
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const long long n = a.n;
#define IN(c) a.in[(long long)(c) * n + i]
#define OUT(c) a.out[(long long)(c) * n + i]
  switch (a.op) {
    case 0: OUT(0) = (double)ds_logf((float)IN(0), log_tab); break;
    case 1: OUT(0) = ds_cos_0_2pi(IN(0), cos_tab); break;
    case 2: {
      double x = IN(0), y = IN(1), z = IN(2);
      ds_div3(&x, &y, &z, IN(3));
      OUT(0) = x, OUT(1) = y, OUT(2) = z;
    } break;
    case 3: OUT(0) = ds_sqrt_dev(IN(0)); break;
    case 4: OUT(0) = u64_to_unit_f64((uint64_t)__double_as_longlong(IN(0))); break;
    case 5: { /* vertices A, B, C; the library pre-gathers {A, B - A, C - A} with these very subtractions */
      const V3 A = {IN(0), IN(1), IN(2)};
      const V3 E1 = {IN(3) - A.x, IN(4) - A.y, IN(5) - A.z};
      const V3 E2 = {IN(6) - A.x, IN(7) - A.y, IN(8) - A.z};
      const V3 r0 = {IN(9), IN(10), IN(11)}, st = {IN(12), IN(13), IN(14)};
      OUT(0) = ray_triangle_intersection_check(A, E1, E2, r0, st);
    } break;
    case 6:
    case 7: {
      V3 r0 = {IN(0), IN(1), IN(2)}, st = {IN(3), IN(4), IN(5)}, nn = {IN(7), IN(8), IN(9)};
      if (a.op == 6)
        reflection(r0, st, IN(6), nn, IN(10));
      else
        crossing(r0, st, IN(6), nn, IN(10));
      OUT(0) = r0.x, OUT(1) = r0.y, OUT(2) = r0.z;
      OUT(3) = st.x, OUT(4) = st.y, OUT(5) = st.z;
      OUT(6) = nn.x, OUT(7) = nn.y, OUT(8) = nn.z;
    } break;
    case 8: {
      const int ns = a.n_xs - 1;
      const double inv_h = (double)ns / (a.xs[ns] - a.xs[0]);
      const double x1 = IN(0), x2 = IN(1);
      OUT(0) = (double)ll_subvoxel_overlap(a.xs, ns, inv_h, x1 < x2 ? x1 : x2);
      OUT(1) = (double)ul_subvoxel_overlap(a.xs, ns, inv_h, x1 > x2 ? x1 : x2);
    } break;
    case 9:
    case 10: {
      const int ns = a.n_xs - 1;
      const double L = a.xs[ns] - a.xs[0];
      const double inv_h = (double)ns / L, inv_l = 1.0 / L;
      if (a.op == 9) {
        OUT(0) = (double)ll_subvoxel_overlap_periodic(a.xs, ns, inv_h, inv_l, IN(0), IN(1));
        OUT(1) = (double)ul_subvoxel_overlap_periodic(a.xs, ns, inv_h, inv_l, IN(0), IN(1));
      } else { /* the loop-free variant of the cooperative kernel: (ll, ul, usable) */
        int ll = 0, ul = 0;
        const bool ok = subvoxel_limits_fast(a.xs, ns, a.xs[0], a.xs[ns], inv_h, inv_l, IN(0), IN(1), ll, ul);
        OUT(0) = (double)ll, OUT(1) = (double)ul, OUT(2) = ok ? 1.0 : 0.0;
      }
    } break;
    case 19: { /* the float32 shortcut of the cooperative kernel: (ll, ul, decided) */
      const int ns = a.n_xs - 1;
      int ll = 0, ul = 0;
      float edge, reach;
      subvoxel_limits_f32(a.xs[0], (double)ns / (a.xs[ns] - a.xs[0]), IN(0), IN(1), ll, ul, edge, reach);
      const bool ok = edge < 0.4999f && reach < 1500.0f; /* the acceptance rule of segment_query() */
      OUT(0) = (double)ll, OUT(1) = (double)ul, OUT(2) = ok ? 1.0 : 0.0;
    } break;
    case 11: OUT(0) = line_circle_intersection(IN(0), IN(1), IN(2), IN(3), IN(4)); break;
    case 12: {
      const V3 r0 = {IN(0), IN(1), IN(2)}, st = {IN(3), IN(4), IN(5)};
      OUT(0) = line_sphere_intersection(r0, st, IN(6));
    } break;
    case 13: {
      const V3 r0 = {IN(0), IN(1), IN(2)}, st = {IN(3), IN(4), IN(5)};
      OUT(0) = line_ellipsoid_intersection(r0, st, IN(6), IN(7), IN(8));
    } break;
    case 14: {
      V3 v = {IN(0), IN(1), IN(2)};
      normalize3(v);
      OUT(0) = v.x, OUT(1) = v.y, OUT(2) = v.z;
    } break;
    case 15: {
      Rng r = {(uint64_t)__double_as_longlong(IN(0)), (uint64_t)__double_as_longlong(IN(1))};
      const V3 s = random_step(r, log_tab, cos_tab);
      OUT(0) = s.x, OUT(1) = s.y, OUT(2) = s.z;
      OUT(3) = __longlong_as_double((long long)r.s0), OUT(4) = __longlong_as_double((long long)r.s1);
    } break;
    case 16: {
      Rng r = {(uint64_t)__double_as_longlong(IN(0)), (uint64_t)__double_as_longlong(IN(1))};
      OUT(0) = (double)rng_uniform_f32(r);
      OUT(1) = rng_uniform_f64(r);
      OUT(2) = __longlong_as_double((long long)r.s0), OUT(3) = __longlong_as_double((long long)r.s1);
    } break;
    case 17: {
      double R[9];
      for (int k = 0; k < 9; k++) R[k] = IN(k);
      const V3 v = {IN(9), IN(10), IN(11)};
      const V3 o = mat_mul(R, v);
      OUT(0) = o.x, OUT(1) = o.y, OUT(2) = o.z;
    } break;
    case 18: {
      const V3 p = {IN(0), IN(1), IN(2)}, q = {IN(3), IN(4), IN(5)};
      const V3 c = cross3(p, q);
      OUT(0) = dot3(p, q);
      OUT(1) = c.x, OUT(2) = c.y, OUT(3) = c.z;
    } break;
    default: break;
  }
#undef IN
#undef OUT
}

static std::string g_unit_err;
extern "C" const char *dsu_last_error(void) { return g_unit_err.c_str(); }

/* in: n_in columns of n doubles (column-major block), out: n_out columns.  Returns 0 or -1. */
extern "C" int dsu_eval(int op, long long n, int n_in, const double *in, int n_out, double *out, const double *xs, int n_xs) {
  if (n <= 0 || n_in < 1 || n_in > DSU_MAX_COLS || n_out < 1 || n_out > DSU_MAX_COLS || !in || !out) {
    g_unit_err = "dsu_eval: bad arguments";
    return -1;
  }
  double *d_in = nullptr, *d_out = nullptr, *d_xs = nullptr;
  auto fail = [&](const char *what, hipError_t e) {
    g_unit_err = std::string(what) + ": " + hipGetErrorString(e);
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (d_xs) (void)hipFree(d_xs);
    return -1;
  };
  hipError_t e;
  if ((e = hipMalloc(&d_in, (size_t)n_in * n * sizeof(double))) != hipSuccess) return fail("hipMalloc", e);
  if ((e = hipMalloc(&d_out, (size_t)n_out * n * sizeof(double))) != hipSuccess) return fail("hipMalloc", e);
  if ((e = hipMemcpy(d_in, in, (size_t)n_in * n * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy", e);
  if ((e = hipMemset(d_out, 0, (size_t)n_out * n * sizeof(double))) != hipSuccess) return fail("hipMemset", e);
  if (xs && n_xs > 0) {
    if ((e = hipMalloc(&d_xs, (size_t)n_xs * sizeof(double))) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMemcpy(d_xs, xs, (size_t)n_xs * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy", e);
  }
  UnitArgs a = {op, n, d_in, d_out, d_xs, n_xs};
  unit_kernel<<<(unsigned)((n + 255) / 256), 256>>>(a);
  if ((e = hipGetLastError()) != hipSuccess) return fail("launch", e);
  if ((e = hipDeviceSynchronize()) != hipSuccess) return fail("sync", e);
  if ((e = hipMemcpy(out, d_out, (size_t)n_out * n * sizeof(double), hipMemcpyDeviceToHost)) != hipSuccess) return fail("hipMemcpy", e);
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  if (d_xs) (void)hipFree(d_xs);
  return 0;
}

/* ---- the mesh traversal on caller-supplied rays (tests/test_gpu_mesh_adversarial.py) --------------------------
 * Walker i starts at r0[i] with direction step[i], remaining length step_l[i] and RNG state st[i] - no random
 * step is drawn, so a test can aim rays at shared edges and vertices, along subvoxel boundary planes, at hits
 * with d == step_l and into far periodic images.  mode 0: the wavefront-cooperative traversal, n_pass > 0 calls
 * of coop_iteration() (flags bit 0 = the walker's reflect loop is over) or, n_pass == 0, as many as it takes,
 * followed by the final advance of the position (flags bit 1 = iter_exc); mode 1: step_mesh(), the per-lane
 * traversal (whole loop + advance only).  The CPU twin is dso_u_mesh_pass() in oracle/dsim_oracle.c. */
struct MeshPassArgs {
  long long n;
  int n_pass, mode;
  const double *r0, *step, *step_l;
  const uint64_t *st;
  double *r0_out, *step_out, *step_l_out;
  uint64_t *st_out;
  int32_t *flags;
};

__global__ __launch_bounds__(256) void mesh_pass_kernel(const WalkParams p, const MeshPassArgs a) {
  extern __shared__ double lds[];
  const int n_bounds = p.mesh.nx + p.mesh.ny + p.mesh.nz + 3;
  for (int i = threadIdx.x; i < n_bounds; i += blockDim.x) lds[i] = p.mesh.bounds[i];
  __syncthreads();
  const double *xs = lds, *ys = lds + p.mesh.nx + 1, *zs = lds + p.mesh.nx + p.mesh.ny + 2;
  CoopWave *coop = reinterpret_cast<CoopWave *>(lds + ((n_bounds + 1) & ~1)) + (threadIdx.x >> 6);
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < a.n;
  const int lane = (int)(threadIdx.x & 63);
  V3 r = {0, 0, 0}, step = {1, 0, 0};
  double sl = 0;
  Rng rng = {1, 1};
  if (live) {
    r = load_v3(a.r0 + i * 3);
    step = load_v3(a.step + i * 3);
    sl = a.step_l[i];
    rng.s0 = a.st[2 * i];
    rng.s1 = a.st[2 * i + 1];
  }
  int flags = 0;
  if (a.mode == 1) {
    bool exc = false;
    if (live) step_mesh<0>(r, step, sl, p, xs, ys, zs, rng, exc);
    flags = 1 | (exc ? 2 : 0);
    sl = DS_NAN; /* step_mesh keeps the remaining length to itself */
  } else {
#ifdef DSIM_TIMERS
    if (lane == 0) coop->tlast = 0;
#endif
    int64_t iter_idx = 0;
    bool done = false;
    for (int pass = 0; a.n_pass == 0 || pass < a.n_pass; pass++) {
      if (__ballot(live && !done) == 0ULL) break;
      bool need_norm = false; /* the caller's part of the contract: a reflected direction is normalised before anything reads it */
      const bool fin = coop_iteration<0>(r, step, sl, iter_idx, live && !done, false, p, xs, ys, zs, rng, coop, lane, need_norm);
      if (need_norm) normalize3(step);
      if (live && !done && fin) done = true;
    }
    flags = done ? 1 : 0;
    if (a.n_pass == 0) {
      if (iter_idx >= p.max_iter) flags |= 2;
      r.x = __builtin_fma(step.x, sl, r.x);
      r.y = __builtin_fma(step.y, sl, r.y);
      r.z = __builtin_fma(step.z, sl, r.z);
    }
  }
  if (live) {
    a.r0_out[i * 3] = r.x, a.r0_out[i * 3 + 1] = r.y, a.r0_out[i * 3 + 2] = r.z;
    a.step_out[i * 3] = step.x, a.step_out[i * 3 + 1] = step.y, a.step_out[i * 3 + 2] = step.z;
    a.step_l_out[i] = sl;
    a.st_out[2 * i] = rng.s0, a.st_out[2 * i + 1] = rng.s1;
    a.flags[i] = flags;
  }
}

/* fresh_params<1> takes the kernarg segment for the kernel's WalkParams (nothing enforced that it is the
 * first and only argument at offset 0).  This kernel has exactly that signature; it compares the two dword by dword
 * and reports through p.exc. */
__global__ void kernarg_check_kernel(const WalkParams p) {
  const WalkParams &f = fresh_params<1>(p);
  const uint32_t *a = reinterpret_cast<const uint32_t *>(&f);
  uint32_t diff = 0;
  WalkParams copy = p;
  const uint32_t *b = reinterpret_cast<const uint32_t *>(&copy);
  for (size_t i = 0; i < sizeof(WalkParams) / 4; i++) diff |= a[i] ^ b[i];
  p.exc[0] = diff == 0 ? 1 : 2;
}

extern "C" int dsu_kernarg_check(void) {
  WalkParams p;
  uint8_t *bytes = reinterpret_cast<uint8_t *>(&p);
  for (size_t i = 0; i < sizeof(WalkParams); i++) bytes[i] = (uint8_t)(i * 37u + 11u); /* every field a pattern of its own */
  uint8_t *out = nullptr;
  if (hipMalloc(&out, 1) != hipSuccess || hipMemset(out, 0, 1) != hipSuccess) return -1;
  p.exc = out;
  kernarg_check_kernel<<<1, 1>>>(p);
  uint8_t got = 0;
  const bool ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(&got, out, 1, hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(out);
  return ok && got == 1 ? 0 : -1;
}

/* index_step_l: the step length the mesh index is laid out for (cell sizes; as dsim_create's step_l).  Returns 0 or -1. */
extern "C" int dsu_mesh_pass(const double *vertices, long long n_vertices, const int64_t *faces, long long n_faces, const double *xs,
                             const double *ys, const double *zs, const int64_t n_sv[3], const int64_t *subvoxel_indices,
                             const int64_t *triangle_indices, long long n_refs, double index_step_l, long long max_iter, double eps,
                             double perm_prob, int mode, int n_pass, long long n, const double *r0, const double *step,
                             const double *step_l, const uint64_t *st, double *r0_out, double *step_out, double *step_l_out,
                             uint64_t *st_out, int32_t *flags) {
  if (n <= 0 || !r0 || !step || !step_l || !st || !r0_out || !step_out || !step_l_out || !st_out || !flags || (mode != 0 && mode != 1) ||
      n_pass < 0 || (mode == 1 && n_pass != 0)) {
    g_unit_err = "dsu_mesh_pass: bad arguments";
    return -1;
  }
  DevMesh dm;
  std::vector<void *> dev;
  auto fail_with = [&](const std::string &what) {
    g_unit_err = what;
    for (void *q : dev) (void)hipFree(q);
    free_mesh(dm);
    return -1;
  };
  if (upload_mesh(dm, nullptr, vertices, n_vertices, faces, n_faces, xs, ys, zs, n_sv, subvoxel_indices, triangle_indices, n_refs,
                  index_step_l) != DSIM_OK)
    return fail_with("upload_mesh: " + g_err);
  auto to_dev = [&](const void *src, size_t bytes, void **out) -> bool {
    void *q = nullptr;
    if (hipMalloc(&q, bytes) != hipSuccess) return false;
    dev.push_back(q);
    if (src && hipMemcpy(q, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return false;
    *out = q;
    return true;
  };
  MeshPassArgs a = {};
  a.n = n;
  a.n_pass = n_pass;
  a.mode = mode;
  void *q[9];
  const size_t v3 = (size_t)n * 3 * sizeof(double), v1 = (size_t)n * sizeof(double), vs = (size_t)n * 2 * sizeof(uint64_t);
  if (!to_dev(r0, v3, &q[0]) || !to_dev(step, v3, &q[1]) || !to_dev(step_l, v1, &q[2]) || !to_dev(st, vs, &q[3]) ||
      !to_dev(nullptr, v3, &q[4]) || !to_dev(nullptr, v3, &q[5]) || !to_dev(nullptr, v1, &q[6]) || !to_dev(nullptr, vs, &q[7]) ||
      !to_dev(nullptr, (size_t)n * sizeof(int32_t), &q[8]))
    return fail_with("dsu_mesh_pass: device memory");
  a.r0 = (const double *)q[0], a.step = (const double *)q[1], a.step_l = (const double *)q[2], a.st = (const uint64_t *)q[3];
  a.r0_out = (double *)q[4], a.step_out = (double *)q[5], a.step_l_out = (double *)q[6], a.st_out = (uint64_t *)q[7];
  a.flags = (int32_t *)q[8];
  WalkParams p = {};
  p.n = n;
  p.max_iter = max_iter;
  p.eps = eps;
  p.perm_prob = perm_prob;
  p.perm_thresh = ds_perm_threshold(perm_prob);
  p.xcd_run_log2 = 3;
  p.mesh = mesh_dev(dm);
  p.mesh.clear_idx = nullptr;
  const int n_bounds = dm.nx + dm.ny + dm.nz + 3;
  const size_t lds = (size_t)((n_bounds + 1) & ~1) * sizeof(double) + 4 * sizeof(CoopWave);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(mesh_pass_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return fail_with(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
  mesh_pass_kernel<<<(unsigned)((n + 255) / 256), 256, lds>>>(p, a);
  if ((e = hipGetLastError()) != hipSuccess) return fail_with(std::string("launch: ") + hipGetErrorString(e));
  if ((e = hipDeviceSynchronize()) != hipSuccess) return fail_with(std::string("sync: ") + hipGetErrorString(e));
  if (hipMemcpy(r0_out, a.r0_out, v3, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(step_out, a.step_out, v3, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(step_l_out, a.step_l_out, v1, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(st_out, a.st_out, vs, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(flags, a.flags, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess)
    return fail_with("dsu_mesh_pass: copy back");
  for (void *d : dev) (void)hipFree(d);
  free_mesh(dm);
  return 0;
}
