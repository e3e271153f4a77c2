/* dsim_hip.hip - host side of the C ABI declared in include/dsim.h.
 *
 * Owns device memory (SoA walker state, phases, gradient tiles, compact mesh
 * index), one HIP stream per simulation and the launch logic.  Compiled for
 * gfx950 only, with -ffp-contract=off (see dsim_device.h).
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dsim.h"
#include "dsim_kernels.h"
#include "dsim_mesh_host.h"

#define DSIM_GAMMA 267.513e6 /* gradients.py:13 */

extern "C" const char *dsim_last_error(void) { return g_err.c_str(); }
extern "C" int dsim_abi_version(void) { return 1; }

extern "C" int dsim_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

/* ----------------------------------------------------------------- RNG (host) --- */

struct HostRng {
  uint64_t s0, s1;
};

static inline uint64_t h_rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

/* numba/cuda/random.py:79-98 */
static inline uint64_t h_next(HostRng &r) {
  uint64_t s0 = r.s0, s1 = r.s1, res = s0 + s1;
  s1 ^= s0;
  r.s0 = h_rotl(s0, 55) ^ s1 ^ (s1 << 14);
  r.s1 = h_rotl(s1, 36);
  return res;
}

/* numba/cuda/random.py:101-125 */
static void h_jump(HostRng &r) {
  static const uint64_t J[2] = {0xbeac0467eba5facbULL, 0xd86b048b86aa9922ULL};
  uint64_t a = 0, c = 0;
  for (int i = 0; i < 2; i++)
    for (int b = 0; b < 64; b++) {
      if (J[i] & (1ULL << b)) {
        a ^= r.s0;
        c ^= r.s1;
      }
      h_next(r);
    }
  r.s0 = a;
  r.s1 = c;
}

/* numba/cuda/random.py:45-68 */
static HostRng h_seed_state(uint64_t seed) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z = z ^ (z >> 31);
  HostRng r = {z, z};
  return r;
}

/* The xoroshiro128+ state transition is linear over GF(2), so "jump 2^64 draws"
 * is a 128x128 bit matrix; column b is the image of basis vector e_b.  Level k
 * holds the matrix of 2^k jumps, which lets any state index be reached with
 * popcount(index) matrix-vector products instead of `index` sequential jumps
 * (numba/cuda/random.py:224-240 is sequential). */
struct JumpTables {
  std::vector<HostRng> cols; /* [levels][128] */
  int levels = 0;
};

static HostRng matvec(const HostRng *cols, HostRng v) {
  HostRng o = {0, 0};
  for (int b = 0; b < 64; b++)
    if (v.s0 >> b & 1) {
      o.s0 ^= cols[b].s0;
      o.s1 ^= cols[b].s1;
    }
  for (int b = 0; b < 64; b++)
    if (v.s1 >> b & 1) {
      o.s0 ^= cols[64 + b].s0;
      o.s1 ^= cols[64 + b].s1;
    }
  return o;
}

static void build_jump_tables(JumpTables &jt) {
  const int L = 48;
  jt.cols.resize((size_t)L * 128);
  /* level 0: apply the reference's jump() to each basis vector */
  for (int b = 0; b < 128; b++) {
    HostRng e = {b < 64 ? 1ULL << b : 0, b >= 64 ? 1ULL << (b - 64) : 0};
    h_jump(e);
    jt.cols[b] = e;
  }
  for (int k = 1; k < L; k++)
    for (int b = 0; b < 128; b++) jt.cols[(size_t)k * 128 + b] = matvec(&jt.cols[(size_t)(k - 1) * 128], jt.cols[(size_t)(k - 1) * 128 + b]);
  jt.levels = L;
}

static const JumpTables &jump_tables() {
  static JumpTables jt;
  static std::once_flag once;
  std::call_once(once, build_jump_tables, std::ref(jt));
  return jt;
}

static HostRng h_state_at(uint64_t seed, uint64_t index) {
  const JumpTables &jt = jump_tables();
  HostRng r = h_seed_state(seed);
  for (int k = 0; k < jt.levels && (index >> k); k++)
    if (index >> k & 1) r = matvec(&jt.cols[(size_t)k * 128], r);
  return r;
}

extern "C" int dsim_rng_init(uint64_t seed, uint64_t first_index, uint64_t n, uint64_t *out) {
  if (!out && n) return fail(DSIM_ERR_INVALID, "dsim_rng_init: out is NULL");
  if (n == 0) return DSIM_OK;
  if ((first_index + n) >> 47) return fail(DSIM_ERR_INVALID, "dsim_rng_init: index range too large");
  /* first state through the jump matrices, the rest by the reference's own
   * sequential recurrence state[i] = jump(state[i-1]) */
  HostRng r = h_state_at(seed, first_index);
  for (uint64_t i = 0; i < n; i++) {
    out[2 * i] = r.s0;
    out[2 * i + 1] = r.s1;
    if (i + 1 < n) h_jump(r);
  }
  return DSIM_OK;
}

/* device-side parallel seeding */
__global__ void rng_init_kernel(uint64_t seed_s0, uint64_t seed_s1, uint64_t first, const int64_t *ids, int64_t n, const uint64_t *cols,
                                int levels, uint64_t *s0, uint64_t *s1) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t idx = first + (uint64_t)(ids != nullptr ? ids[i] : i); /* slot i holds walker first + ids[i] (a shard cut by space) */
  uint64_t a = seed_s0, c = seed_s1;
  for (int k = 0; k < levels && (idx >> k); k++) {
    if (!(idx >> k & 1)) continue;
    const uint64_t *m = cols + (size_t)k * 256;
    uint64_t na = 0, nc = 0;
    for (int b = 0; b < 64; b++) {
      uint64_t mask = 0 - (a >> b & 1);
      na ^= m[2 * b] & mask;
      nc ^= m[2 * b + 1] & mask;
    }
    for (int b = 0; b < 64; b++) {
      uint64_t mask = 0 - (c >> b & 1);
      na ^= m[128 + 2 * b] & mask;
      nc ^= m[128 + 2 * b + 1] & mask;
    }
    a = na;
    c = nc;
  }
  s0[i] = a;
  s1[i] = c;
}

/* ------------------------------------------------------------------ handle --- */

#define DSIM_MAX_GROUPS 4 /* walker groups of a mesh run, each with a stream of its own */

struct dsim_sim {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int64_t n = 0, first = 0;
  int64_t phase_stride = 0; /* row stride of phase / phase_alt: n, also in a view of a sub-range (run_own_clock) */
  int M = 0, T = 0;
  double dt = 0, step_l = 0;
  double *x[2] = {nullptr, nullptr}, *y[2] = {nullptr, nullptr}, *z[2] = {nullptr, nullptr};
  uint64_t *s0[2] = {nullptr, nullptr}, *s1[2] = {nullptr, nullptr};
  int cur = 0; /* which of the two state buffers is current */
  double *phase = nullptr;
  uint8_t *exc = nullptr;
  double *g4 = nullptr;
  long long *partial = nullptr, *nexc_partial = nullptr, *sig_out = nullptr; /* exact signal sums: integer limbs (dsim_kernels.h) */
  int sig_blocks = 0;
  void *staging = nullptr;
  size_t staging_bytes = 0;
  DevMesh mesh;
  double last_ms = 0;
  int last_launches = 0;
  /* events around every launch of the walk kernel itself (the run also contains sort / gather /
   * phase kernels): bench.py's roofline is computed from these */
  std::vector<hipEvent_t> walk_ev;
  int walk_ev_used = 0;
  double last_walk_ms = 0;
  int block_size = 256, tile_steps = 0, meas_per_pass = 0;
  int resort_steps = 0; /* mesh: 0 = automatic, < 0 = never re-sort the walkers */
  int grad_src = 0;     /* gradient samples from 0 = LDS tiles, 1 = scalar (SMEM) loads */
  int mesh_traversal = 0; /* 0/2 = wavefront-cooperative (clearance map when it pays), 1 = per-lane, 3 = cooperative, never the map, 4 = always */
  unsigned long long *clear_count = nullptr; /* one-off count before the first launch of a call */
  /* per-walker clocks of the cooperative mesh kernel (run_own_clock) */
  int32_t *tcur[2] = {nullptr, nullptr};
  /* per group two sets of {walkers not yet at t_end, time steps they still have to take, most left, walkers in a clear
   * cell}: launch k fills set k & 1 while the host still reads the other one; [group][set][4] */
  unsigned long long *counters = nullptr;
  /* second walker group of a grouped run (run_own_clock): its own stream, sort scratch and counters */
  hipStream_t grp_stream[DSIM_MAX_GROUPS - 1] = {};
  std::vector<hipEvent_t> grp_walk_ev[DSIM_MAX_GROUPS - 1];
  int grp_walk_ev_used[DSIM_MAX_GROUPS - 1] = {};
  int last_groups = 1;
  double last_walk_busy_ms = 0; /* time during which at least one walk kernel was running */
  double last_clear_frac = 0;
  /* spatial re-sorting (mesh): alternates of phase/exc, original-index maps, sort scratch */
  bool sort_ready = false;
  double *phase_alt = nullptr;
  double *traj = nullptr; /* [steps][3][n] positions of the last launch (mesh with M > 2: phases from positions) */
  int traj_steps = 0;
  size_t traj_doubles = 0; /* capacity of traj */
  /* hand-over of recorded positions: a copy stream of its own and two pinned staging buffers, so that the
   * D2H transfer of one block of steps overlaps the host's work (copy-out / text formatting) on the previous one */
  hipStream_t copy_stream = nullptr;
  void *pin[2] = {nullptr, nullptr};
  size_t pin_bytes = 0;
  hipEvent_t pin_ev[2] = {nullptr, nullptr};
  bool recording = false;        /* dsim_set_recording: keep the positions of every step of a run */
  int rec_t0 = 0, rec_t1 = 0;    /* time range held in traj after a recorded run */
  uint8_t *exc_alt = nullptr;
  int32_t *orig[2] = {nullptr, nullptr};
  /* counting sort (sort_count / sort_scan / sort_scatter kernels): bin of every walker, and per walker group a
   * histogram and the bins' cursors */
  uint32_t *bin_of = nullptr;
  uint32_t *sort_hist[DSIM_MAX_GROUPS] = {}, *sort_offs[DSIM_MAX_GROUPS] = {};
  int sort_bits = 0;
  int orig_cur = 0;
  /* multi-device handle (dsim_create_multi): the walkers [first, first + n) cut into balanced contiguous ranges, one
   * single-device handle per range (shards[k] owns the walkers [shard_off[k], shard_off[k] + shards[k]->n) of this
   * handle); a handle with shards owns no device memory itself - every entry point hands the call on to the shards,
   * each on a host thread of its own */
  std::vector<dsim_sim *> shards;
  std::vector<int64_t> shard_off;
  /* [r5] shards cut by SPACE (dsim_partition_by_position): shard k holds the walkers shard_ids[k] (indices inside the
   * handle, ascending), not the range that starts at shard_off[k]; empty: the contiguous ranges.  Host arrays of the
   * handle stay in walker order - rows are gathered / scattered through the lists */
  std::vector<std::vector<int64_t>> shard_ids;
  /* a shard of such a handle: the walker of slot i is first + walker_ids[i] (device array; the RNG seeding needs it) */
  int64_t *walker_ids = nullptr;
  bool seeded = false, states_given = false, has_run = false; /* what a re-partition would have to redo */
  uint64_t seed_value = 0;
  /* what it takes to create a shard again with another number of walkers (dsim_partition_by_position cuts a mesh into
   * parts of equal estimated COST, not of equal counts): the gradient, the tuning, the recording switch */
  std::vector<double> keep_g[3];
  int keep_tune[6] = {0, 0, 0, 0, 0, 0};
  bool keep_recording = false;
  std::shared_ptr<const HostMesh> host_mesh; /* the index of dsim_set_mesh as built on the host: the cost model looks cells up in it */
};

/* f(k, shard, first walker of the shard inside the handle) for every shard of a multi-device handle, concurrently (one
 * host thread per shard: HIP's current device is per thread, and the calls block); the first failure is reported with
 * the shard's own message */
template <class F>
static int multi_each(dsim_sim *h, F f) {
  const int n = (int)h->shards.size();
  std::vector<int> rc((size_t)n, DSIM_OK);
  std::vector<std::string> msg((size_t)n);
  auto work = [&](int k) {
    rc[(size_t)k] = f(k, h->shards[(size_t)k], h->shard_off[(size_t)k]);
    if (rc[(size_t)k] != DSIM_OK) msg[(size_t)k] = g_err; /* the worker thread's own message */
  };
  if (n == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int k = 0; k < n; k++) th.emplace_back(work, k);
    for (auto &t : th) t.join();
  }
  for (int k = 0; k < n; k++)
    if (rc[(size_t)k] != DSIM_OK) return fail(rc[(size_t)k], "shard %d (device %d): %s", k, h->shards[(size_t)k]->device, msg[(size_t)k].c_str());
  return DSIM_OK;
}
#define IS_MULTI(h) ((h) != nullptr && !(h)->shards.empty())
/* hand a call on to every shard: `call` sees the shard as s_ and its first walker inside the handle as off_ */
#define MULTI_FORWARD(h, ...)                                                         \
  do {                                                                                \
    if (IS_MULTI(h))                                                                  \
      return multi_each((h), [&](int k_, dsim_sim *s_, int64_t off_) -> int {         \
        (void)k_, (void)off_;                                                         \
        return (__VA_ARGS__);                                                         \
      });                                                                             \
  } while (0)

/* rows of a per-walker array: the shard's (n_k, row) block out of / into the handle's (n, row) array in walker order */
static const void *shard_rows_in(const dsim_sim *h, int k, int64_t off, const void *whole, size_t row, std::vector<char> &tmp) {
  if (h->shard_ids.empty()) return (const char *)whole + (size_t)off * row;
  const std::vector<int64_t> &ids = h->shard_ids[(size_t)k];
  tmp.resize(ids.size() * row);
  for (size_t j = 0; j < ids.size(); j++) memcpy(tmp.data() + j * row, (const char *)whole + (size_t)ids[j] * row, row);
  return tmp.data();
}
static void shard_rows_out(const dsim_sim *h, int k, int64_t off, void *whole, size_t row, const void *part, int64_t n_k) {
  if (h->shard_ids.empty()) {
    memcpy((char *)whole + (size_t)off * row, part, (size_t)n_k * row);
    return;
  }
  const std::vector<int64_t> &ids = h->shard_ids[(size_t)k];
  if (row == 8) {
    for (size_t j = 0; j < ids.size(); j++) ((uint64_t *)whole)[ids[j]] = ((const uint64_t *)part)[j];
    return;
  }
  for (size_t j = 0; j < ids.size(); j++) memcpy((char *)whole + (size_t)ids[j] * row, (const char *)part + j * row, row);
}

static int ensure_staging(dsim_sim *h, size_t bytes) {
  if (h->staging_bytes >= bytes) return DSIM_OK;
  if (h->staging) (void)hipFree(h->staging);
  h->staging = nullptr;
  h->staging_bytes = 0;
  HIP_TRY(hipMalloc(&h->staging, bytes));
  h->staging_bytes = bytes;
  return DSIM_OK;
}

extern "C" int dsim_destroy(dsim_handle h) {
  if (!h) return DSIM_OK;
  if (IS_MULTI(h)) {
    for (dsim_sim *s : h->shards) (void)dsim_destroy(s);
    delete h;
    return DSIM_OK;
  }
  (void)hipSetDevice(h->device);
  for (int b = 0; b < 2; b++) {
    if (h->x[b]) (void)hipFree(h->x[b]);
    if (h->y[b]) (void)hipFree(h->y[b]);
    if (h->z[b]) (void)hipFree(h->z[b]);
    if (h->s0[b]) (void)hipFree(h->s0[b]);
    if (h->s1[b]) (void)hipFree(h->s1[b]);
  }
  if (h->phase) (void)hipFree(h->phase);
  if (h->exc) (void)hipFree(h->exc);
  if (h->g4) (void)hipFree(h->g4);
  if (h->partial) (void)hipFree(h->partial);
  if (h->nexc_partial) (void)hipFree(h->nexc_partial);
  if (h->sig_out) (void)hipFree(h->sig_out);
  if (h->staging) (void)hipFree(h->staging);
  if (h->clear_count) (void)hipFree(h->clear_count);
  for (int b = 0; b < 2; b++)
    if (h->tcur[b]) (void)hipFree(h->tcur[b]);
  if (h->counters) (void)hipFree(h->counters);
  for (int g = 0; g < DSIM_MAX_GROUPS - 1; g++) {
    for (hipEvent_t e : h->grp_walk_ev[g]) (void)hipEventDestroy(e);
    if (h->grp_stream[g]) (void)hipStreamDestroy(h->grp_stream[g]);
  }
  if (h->phase_alt) (void)hipFree(h->phase_alt);
  if (h->traj) (void)hipFree(h->traj);
  if (h->exc_alt) (void)hipFree(h->exc_alt);
  for (int b = 0; b < 2; b++)
    if (h->orig[b]) (void)hipFree(h->orig[b]);
  for (int g = 0; g < DSIM_MAX_GROUPS; g++) {
    if (h->sort_hist[g]) (void)hipFree(h->sort_hist[g]);
    if (h->sort_offs[g]) (void)hipFree(h->sort_offs[g]);
  }
  if (h->bin_of) (void)hipFree(h->bin_of);
  if (h->walker_ids) (void)hipFree(h->walker_ids);
  free_mesh(h->mesh);
  for (hipEvent_t e : h->walk_ev) (void)hipEventDestroy(e);
  for (int b = 0; b < 2; b++) {
    if (h->pin[b]) (void)hipHostFree(h->pin[b]);
    if (h->pin_ev[b]) (void)hipEventDestroy(h->pin_ev[b]);
  }
  if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return DSIM_OK;
}

static int create_impl(dsim_sim *h, const double *gx, const double *gy, const double *gz) {
  const int64_t n = h->n;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreate(&h->ev0));
  HIP_TRY(hipEventCreate(&h->ev1));
  for (int b = 0; b < 2; b++) {
    HIP_TRY(hipMalloc(&h->x[b], n * sizeof(double)));
    HIP_TRY(hipMalloc(&h->y[b], n * sizeof(double)));
    HIP_TRY(hipMalloc(&h->z[b], n * sizeof(double)));
    HIP_TRY(hipMalloc(&h->s0[b], n * sizeof(uint64_t)));
    HIP_TRY(hipMalloc(&h->s1[b], n * sizeof(uint64_t)));
  }
  HIP_TRY(hipMalloc(&h->phase, (size_t)h->M * n * sizeof(double)));
  HIP_TRY(hipMalloc(&h->exc, n));
  HIP_TRY(hipMemsetAsync(h->x[0], 0, n * sizeof(double), h->stream));
  HIP_TRY(hipMemsetAsync(h->y[0], 0, n * sizeof(double), h->stream));
  HIP_TRY(hipMemsetAsync(h->z[0], 0, n * sizeof(double), h->stream));
  HIP_TRY(hipMemsetAsync(h->s0[0], 0, n * sizeof(uint64_t), h->stream));
  HIP_TRY(hipMemsetAsync(h->s1[0], 0, n * sizeof(uint64_t), h->stream));
  HIP_TRY(hipMemsetAsync(h->phase, 0, (size_t)h->M * n * sizeof(double), h->stream));
  HIP_TRY(hipMemsetAsync(h->exc, 0, n, h->stream));
  /* gradient re-laid-out [t][m][gx gy gz 0]: one contiguous, 16-byte aligned LDS tile per range of steps */
  std::vector<double> g4((size_t)h->T * h->M * 4 + 32 * 4, 0.0); /* + 32 zero samples: scalar-path over-read */
  for (int m = 0; m < h->M; m++)
    for (int t = 0; t < h->T; t++) {
      size_t o = ((size_t)t * h->M + m) * 4;
      g4[o] = gx[(size_t)m * h->T + t];
      g4[o + 1] = gy[(size_t)m * h->T + t];
      g4[o + 2] = gz[(size_t)m * h->T + t];
      g4[o + 3] = 0.0;
    }
  HIP_TRY(hipMalloc(&h->g4, g4.size() * sizeof(double)));
  HIP_TRY(hipMemcpyAsync(h->g4, g4.data(), g4.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  h->sig_blocks = (int)std::min<int64_t>(1024, (n + 255) / 256);
  HIP_TRY(hipMalloc(&h->partial, (size_t)h->M * h->sig_blocks * 4 * sizeof(long long)));
  HIP_TRY(hipMalloc(&h->nexc_partial, (size_t)h->sig_blocks * sizeof(long long)));
  HIP_TRY(hipMalloc(&h->sig_out, (size_t)(4 * h->M + 1) * sizeof(long long)));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return DSIM_OK;
}

extern "C" int dsim_create(dsim_handle *out, int device, int64_t n_walkers, int64_t first_walker, int32_t M, int32_t T,
                           const double *gx, const double *gy, const double *gz, double dt, double step_l) {
  if (!out) return fail(DSIM_ERR_INVALID, "dsim_create: out is NULL");
  *out = nullptr;
  if (n_walkers <= 0 || first_walker < 0) return fail(DSIM_ERR_INVALID, "dsim_create: bad walker range");
  if (M <= 0 || T <= 0 || !gx || !gy || !gz) return fail(DSIM_ERR_INVALID, "dsim_create: bad gradient");
  if (!(dt > 0) || !(step_l >= 0)) return fail(DSIM_ERR_INVALID, "dsim_create: bad dt / step_l");
  int ndev = dsim_device_count();
  if (ndev <= 0) return fail(DSIM_ERR_NO_DEVICE, "no HIP device visible (need an AMD GPU with ROCm)");
  if (device < 0 || device >= ndev) return fail(DSIM_ERR_INVALID, "dsim_create: device %d out of range (%d)", device, ndev);
  dsim_sim *h = new dsim_sim();
  h->device = device;
  h->n = n_walkers;
  h->phase_stride = n_walkers;
  h->first = first_walker;
  h->M = M;
  h->T = T;
  h->dt = dt;
  h->step_l = step_l;
  int rc = create_impl(h, gx, gy, gz);
  if (rc != DSIM_OK) {
    std::string keep = g_err;
    dsim_destroy(h);
    g_err = keep;
    return rc;
  }
  *out = h;
  return DSIM_OK;
}

extern "C" int dsim_create_multi(dsim_handle *out, const int32_t *device_ids, int32_t n_devices, int64_t n_walkers, int64_t first_walker,
                                 int32_t M, int32_t T, const double *gx, const double *gy, const double *gz, double dt, double step_l) {
  if (!out) return fail(DSIM_ERR_INVALID, "dsim_create_multi: out is NULL");
  *out = nullptr;
  if (!device_ids || n_devices < 1 || n_devices > 1024) return fail(DSIM_ERR_INVALID, "dsim_create_multi: bad device list");
  if (n_walkers <= 0 || first_walker < 0) return fail(DSIM_ERR_INVALID, "dsim_create_multi: bad walker range");
  dsim_sim *h = new dsim_sim();
  h->device = device_ids[0];
  h->n = n_walkers;
  h->phase_stride = n_walkers;
  h->first = first_walker;
  h->M = M;
  h->T = T;
  h->dt = dt;
  h->step_l = step_l;
  if (M > 0 && T > 0 && gx && gy && gz) {
    h->keep_g[0].assign(gx, gx + (size_t)M * T);
    h->keep_g[1].assign(gy, gy + (size_t)M * T);
    h->keep_g[2].assign(gz, gz + (size_t)M * T);
  }
  /* balanced contiguous ranges: the first n % G shards own one walker more (disimpy_amd/sharding.py, shard_range) */
  const int64_t base = n_walkers / n_devices, extra = n_walkers % n_devices;
  for (int k = 0; k < n_devices; k++) {
    const int64_t a = k * base + std::min<int64_t>(k, extra), cnt = base + (k < extra ? 1 : 0);
    if (cnt == 0) continue; /* more devices than walkers */
    dsim_sim *sh = nullptr;
    const int rc = dsim_create(&sh, device_ids[k], cnt, first_walker + a, M, T, gx, gy, gz, dt, step_l);
    if (rc != DSIM_OK) {
      const std::string keep = g_err;
      for (dsim_sim *q : h->shards) (void)dsim_destroy(q);
      delete h;
      return fail(rc, "dsim_create_multi: shard %d on device %d: %s", k, device_ids[k], keep.c_str());
    }
    h->shards.push_back(sh);
    h->shard_off.push_back(a);
  }
  *out = h;
  return DSIM_OK;
}

extern "C" int dsim_shard_count(dsim_handle h) { return h ? (IS_MULTI(h) ? (int)h->shards.size() : 1) : 0; }

extern "C" int dsim_set_tuning(dsim_handle h, int32_t block_size, int32_t tile_steps, int32_t meas_per_pass, int32_t resort_steps,
                               int32_t gradient_source, int32_t mesh_traversal) {
  if (IS_MULTI(h)) {
    const int v[6] = {block_size, tile_steps, meas_per_pass, resort_steps, gradient_source, mesh_traversal};
    for (int q = 0; q < 6; q++)
      if (v[q]) h->keep_tune[q] = v[q];
  }
  MULTI_FORWARD(h, dsim_set_tuning(s_, block_size, tile_steps, meas_per_pass, resort_steps, gradient_source, mesh_traversal));
  if (!h) return fail(DSIM_ERR_INVALID, "null handle");
  if (block_size) {
    if (block_size < 64 || block_size > 256 || block_size % 64) return fail(DSIM_ERR_INVALID, "block_size must be 64, 128, 192 or 256");
    h->block_size = block_size;
  }
  if (tile_steps > 0) h->tile_steps = tile_steps;
  if (meas_per_pass > 0) h->meas_per_pass = meas_per_pass;
  if (resort_steps != 0) h->resort_steps = resort_steps;
  if (gradient_source == 1 || gradient_source == 2) {
#ifndef DSIM_SMEM_GRADIENT
    if (gradient_source == 2) return fail(DSIM_ERR_INVALID, "scalar-load gradient path not built (compile with -DDSIM_SMEM_GRADIENT)");
#endif
    h->grad_src = gradient_source - 1;
  }
  if (mesh_traversal >= 1 && mesh_traversal <= 5) h->mesh_traversal = mesh_traversal;
  return DSIM_OK;
}

static inline int grid_for(int64_t n, int bs) { return (int)((n + bs - 1) / bs); }

extern "C" int dsim_set_positions(dsim_handle h, const double *pos) {
  if (IS_MULTI(h)) {
    if (!pos) return fail(DSIM_ERR_INVALID, "dsim_set_positions: null argument");
    return multi_each(h, [&](int k, dsim_sim *sh, int64_t off) -> int {
      std::vector<char> tmp;
      return dsim_set_positions(sh, (const double *)shard_rows_in(h, k, off, pos, 24, tmp));
    });
  }
  if (!h || !pos) return fail(DSIM_ERR_INVALID, "dsim_set_positions: null argument");
  HIP_TRY(hipSetDevice(h->device));
  int rc = ensure_staging(h, (size_t)h->n * 3 * sizeof(double));
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(h->staging, pos, (size_t)h->n * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  aos_to_soa_kernel<<<grid_for(h->n, 256), 256, 0, h->stream>>>((const double *)h->staging, h->n, h->x[h->cur], h->y[h->cur], h->z[h->cur]);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  return DSIM_OK;
}

/* Estimated relative cost of a walker-step at the walker's position in a mesh: 1 + c x (records of the fine cell the
 * position lies in, at most 40).  The records are what a pass of the traversal has to test; a walker in an empty
 * cell - the inside of a soma - finishes its steps in one cheap pass.  c from the measured rates of the eight parts of the
 * refined neuron: 0.065 in round 5 (1.43e10 in the dendrites, 2.05e10 in the soma; profiles/history/r05_spatial_parts.txt);
 * [r6] 0.056 - the traversal has become cheaper per record since (1.55e10 / 2.25e10): with 0.065 the soma part was the
 * longest of the eight (0.731 s per pass against a mean of 0.690), with 0.055-0.058 the longest is 0.710 s; a lower cap than
 * 40 records only does worse (profiles/history/r06_spatial_cut_cost_weight.txt).  DSIM_COST_PER_RECORD overrides it in
 * tuning runs.  Only the CUT depends on it, never a result. */
static void mesh_walker_cost(const HostMesh &hm, const double *pos, int64_t n, double *w) {
  double c = 0.056;
  if (const char *e = getenv("DSIM_COST_PER_RECORD")) c = atof(e);
  const int lk = hm.lk, k = 1 << lk, ku = lk > 0 ? 2 * k : 1;
  const int per[3] = {ku, ku, k};
  const unsigned n_thr = (unsigned)std::max<int64_t>(1, std::min<int64_t>(std::min(32u, std::max(1u, std::thread::hardware_concurrency())), n / 65536));
  auto work = [&](int64_t a, int64_t b) {
    for (int64_t i = a; i < b; i++) {
      int sv[3], cell[3];
      bool ok = true;
      for (int ax = 0; ax < 3 && ok; ax++) {
        const double t = (pos[3 * i + ax] - hm.lo[ax]) * hm.inv_h[ax]; /* subvoxel units */
        if (!std::isfinite(t) || std::fabs(t) > 1e9) {
          ok = false;
          break;
        }
        const double ft = std::floor(t);
        int64_t s = (int64_t)ft;
        const int nn = hm.n[ax];
        int64_t sw = s % nn;
        if (sw < 0) sw += nn; /* a periodic image */
        sv[ax] = (int)sw;
        int u = (int)((t - ft) * per[ax]);
        u = std::min(std::max(u, 0), per[ax] - 1);
        cell[ax] = (lk > 0 && ax < 2) ? std::min(u, ku - 2) : u;
      }
      uint32_t rec = 0;
      if (ok) {
        const int64_t s = ((int64_t)sv[0] * hm.n[1] + sv[1]) * hm.n[2] + sv[2];
        if (lk > 0) {
          const int32_t base = hm.sv_base[(size_t)s];
          if (base >= 0) {
            const size_t q = (size_t)base + ((size_t)cell[0] * ku + cell[1]) * k + cell[2];
            rec = hm.cell_start[q + 1] - hm.cell_start[q];
          }
        } else {
          rec = hm.cell_start[(size_t)s + 1] - hm.cell_start[(size_t)s];
        }
      }
      w[i] = 1.0 + c * (double)std::min<uint32_t>(rec, 40u);
    }
  };
  std::vector<std::thread> th;
  for (unsigned t = 1; t < n_thr; t++) th.emplace_back(work, n * (int64_t)t / n_thr, n * (int64_t)(t + 1) / n_thr);
  work(0, n / n_thr);
  for (auto &t : th) t.join();
}

/* host only: the weights dsim_partition_by_position uses on this mesh, for a process that cuts the walkers itself
 * (dsim_partition_walkers): out_cost[i] for the n positions */
extern "C" int dsim_mesh_walker_cost(const double *vertices, int64_t n_vertices, const int64_t *faces, int64_t n_faces, const double *xs,
                                     const double *ys, const double *zs, const int64_t n_sv[3], const int64_t *subvoxel_indices,
                                     const int64_t *triangle_indices, int64_t n_refs, double step_l, const double *pos, int64_t n, double *out_cost) {
  if (!pos || !out_cost || n < 0) return fail(DSIM_ERR_INVALID, "dsim_mesh_walker_cost: bad argument");
  HostMesh hm;
  DevMesh none;
  /* the fine grid only (its k follows from step_l): the cost model never looks at the clearance map */
  int rc = upload_mesh(none, nullptr, vertices, n_vertices, faces, n_faces, xs, ys, zs, n_sv, subvoxel_indices, triangle_indices, n_refs, step_l,
                       nullptr, &hm, false);
  if (rc) return rc;
  mesh_walker_cost(hm, pos, n, out_cost);
  return DSIM_OK;
}

/* [r5] Walkers cut into n_parts by SPACE: ordered along the Morton curve of `pos` (21 bits per axis, cubic cells over the
 * bounding box of the finite coordinates; ties by walker index), part k is the k-th stretch of that order.  weights ==
 * NULL: the balanced sizes of the contiguous cut (the first n % n_parts parts hold one walker more); else stretches of
 * equal total weight (every part keeps at least one walker while n >= n_parts).  ids_out (n): the walkers of part 0 in
 * ascending order, then those of part 1, ...; part_start (n_parts + 1): where each part begins in ids_out.  Host only,
 * deterministic - every rank of a job computes the same cut from the same positions. */
extern "C" int dsim_partition_walkers(const double *pos, int64_t n, int32_t n_parts, const double *weights, int64_t *ids_out, int64_t *part_start) {
  if (!pos || !ids_out || !part_start || n < 0 || n_parts < 1) return fail(DSIM_ERR_INVALID, "dsim_partition_walkers: bad argument");
  const size_t G = (size_t)n_parts;
  const int64_t base = n / n_parts, extra = n % n_parts;
  for (size_t k = 0; k <= G; k++) part_start[k] = (int64_t)k * base + std::min<int64_t>((int64_t)k, extra);
  if (n == 0) return DSIM_OK;
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int64_t i = 0; i < n; i++)
    for (int c = 0; c < 3; c++) {
      const double v = pos[3 * i + c];
      if (std::isfinite(v)) lo[c] = std::min(lo[c], v), hi[c] = std::max(hi[c], v);
    }
  /* ONE scale for the three axes: the cells of the curve are cubes in space (a thin substrate keeps its thin axis' top bits at 0) */
  double ext = 0;
  for (int c = 0; c < 3; c++) ext = std::max(ext, hi[c] - lo[c]);
  const double scale = ext > 0 ? 2097151.0 / ext : 0.0;
  auto spread = [](uint64_t v) { /* 21 bits -> every third bit */
    v &= 0x1fffffULL;
    v = (v | v << 32) & 0x1f00000000ffffULL;
    v = (v | v << 16) & 0x1f0000ff0000ffULL;
    v = (v | v << 8) & 0x100f00f00f00f00fULL;
    v = (v | v << 4) & 0x10c30c30c30c30c3ULL;
    v = (v | v << 2) & 0x1249249249249249ULL;
    return v;
  };
  struct Key {
    uint64_t key;
    int64_t idx;
    bool operator<(const Key &o) const { return key != o.key ? key < o.key : idx < o.idx; }
  };
  std::vector<Key> keys((size_t)n);
  const unsigned n_thr = (unsigned)std::max<int64_t>(1, std::min<int64_t>(std::min(32u, std::max(1u, std::thread::hardware_concurrency())), n / 65536));
  {
    auto work = [&](int64_t a, int64_t b) {
      for (int64_t i = a; i < b; i++) {
        uint64_t q[3];
        for (int c = 0; c < 3; c++) {
          const double v = pos[3 * i + c];
          const double t = std::isfinite(v) ? (v - lo[c]) * scale : 0.0;
          q[c] = (uint64_t)std::min(2097151.0, std::max(0.0, t));
        }
        keys[(size_t)i] = Key{spread(q[0]) << 2 | spread(q[1]) << 1 | spread(q[2]), i};
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < n_thr; t++) th.emplace_back(work, n * (int64_t)t / n_thr, n * (int64_t)(t + 1) / n_thr);
    work(0, n / n_thr);
    for (auto &t : th) t.join();
  }
  /* bisection with nth_element into chunks [cs[a], cs[b]): O(n log chunks), the halves on threads of their own */
  std::vector<int64_t> cs;
  std::function<void(size_t, size_t)> cut = [&](size_t a, size_t b) {
    if (b - a < 2) return;
    const size_t m = (a + b) / 2;
    std::nth_element(keys.begin() + cs[a], keys.begin() + cs[m], keys.begin() + cs[b]);
    std::thread left(cut, a, m);
    cut(m, b);
    left.join();
  };
  if (!weights) { /* only the cut points matter, not the order inside a part */
    cs.assign(part_start, part_start + G + 1);
    cut(0, G);
  } else {
    /* the whole order: 16 chunks by count, each sorted on its own thread; then the running weight along the curve */
    const size_t P = (size_t)std::max<int64_t>(1, std::min<int64_t>(16, n / 4096));
    cs.resize(P + 1);
    for (size_t q = 0; q <= P; q++) cs[q] = n * (int64_t)q / (int64_t)P;
    cut(0, P);
    {
      std::vector<std::thread> th;
      for (size_t q = 1; q < P; q++) th.emplace_back([&, q]() { std::sort(keys.begin() + cs[q], keys.begin() + cs[q + 1]); });
      std::sort(keys.begin() + cs[0], keys.begin() + cs[1]);
      for (auto &t : th) t.join();
    }
    long double total = 0;
    for (int64_t i = 0; i < n; i++) total += (long double)std::max(0.0, weights[i]);
    long double run = 0;
    size_t k = 1;
    for (int64_t j = 0; j < n && k < G; j++) {
      run += (long double)std::max(0.0, weights[keys[(size_t)j].idx]);
      while (k < G && run >= total * (long double)k / (long double)G) part_start[k++] = j + 1;
    }
    while (k < G) part_start[k++] = n;
    if (n >= (int64_t)G) { /* nobody stays empty */
      for (size_t q = 1; q < G; q++) part_start[q] = std::max(part_start[q], part_start[q - 1] + 1);
      for (size_t q = G - 1; q >= 1; q--) part_start[q] = std::min(part_start[q], part_start[q + 1] - 1);
    }
  }
  std::vector<std::thread> th;
  auto finish = [&](size_t k) {
    int64_t *ids = ids_out + part_start[k];
    const int64_t cnt = part_start[k + 1] - part_start[k];
    for (int64_t j = 0; j < cnt; j++) ids[j] = keys[(size_t)(part_start[k] + j)].idx;
    std::sort(ids, ids + cnt); /* walker order inside the part */
  };
  for (size_t k = 1; k < G; k++) th.emplace_back(finish, k);
  finish(0);
  for (auto &t : th) t.join();
  return DSIM_OK;
}

/* Single-device handle: slot i holds walker first_walker + ids[i] instead of first_walker + i (ids ascending, n of
 * them; NULL: back to the range).  Only the RNG seeding looks at it (dsim_rng_init_device: walker i <- state i) - a
 * process that owns a part of dsim_partition_walkers sets it before seeding.  Not for multi-device handles, which keep
 * their own lists (dsim_partition_by_position). */
extern "C" int dsim_set_walker_ids(dsim_handle h, const int64_t *ids) {
  if (!h || IS_MULTI(h)) return fail(DSIM_ERR_INVALID, "dsim_set_walker_ids: needs a single-device handle");
  HIP_TRY(hipSetDevice(h->device));
  if (!ids) {
    if (h->walker_ids) (void)hipFree(h->walker_ids);
    h->walker_ids = nullptr;
    return DSIM_OK;
  }
  for (int64_t i = 0; i < h->n; i++)
    if (ids[i] < 0 || (i > 0 && ids[i] <= ids[i - 1])) return fail(DSIM_ERR_INVALID, "dsim_set_walker_ids: ids must be ascending and >= 0");
  if (!h->walker_ids) HIP_TRY(hipMalloc(&h->walker_ids, (size_t)h->n * sizeof(int64_t)));
  HIP_TRY(hipMemcpy(h->walker_ids, ids, (size_t)h->n * sizeof(int64_t), hipMemcpyHostToDevice));
  return DSIM_OK;
}

/* [r5] Cut the walkers of a multi-device handle into its shards by SPACE (dsim_partition_walkers) and set their
 * positions.  A mesh shard then works in ITS part of the substrate: at an eighth of the walkers per region a GPU of an
 * 8-GPU job dragged the whole index through its caches (the fill emits positions in (launch, thread) order: a range of
 * indices is a random sample of every dendrite).  When the handle knows its mesh (dsim_set_mesh came first) the parts
 * are of equal estimated COST (mesh_walker_cost), not of equal counts - the walkers of a soma are cheap, those of thin
 * dendrites dear, and the job ends with its slowest shard: the shards are created again with their new sizes (the
 * device copies of the index move over).  Nothing else changes: walker i keeps RNG state i (rng_init_kernel takes the
 * index list) and position i, every host array of the handle stays in walker order (rows are gathered / scattered
 * through the lists), the exact integer signal sums do not depend on the cut.  Call it instead of dsim_set_positions,
 * before the first run; RNG states seeded with dsim_rng_init_device are re-seeded under the new cut, states given with
 * dsim_set_rng_states are moved.  On a single-device handle: dsim_set_positions. */
extern "C" int dsim_partition_by_position(dsim_handle h, const double *pos) {
  if (!h || !pos) return fail(DSIM_ERR_INVALID, "dsim_partition_by_position: null argument");
  if (!IS_MULTI(h) || h->shards.size() < 2) return dsim_set_positions(h, pos);
  if (h->has_run) return fail(DSIM_ERR_STATE, "dsim_partition_by_position: the handle has already run");
  const int64_t n = h->n;
  const size_t G = h->shards.size();
  if ((int64_t)G > n) return dsim_set_positions(h, pos);
  std::vector<uint64_t> states;
  if (h->states_given) { /* explicit RNG states: collect them under the old cut */
    states.resize((size_t)n * 2);
    const int rc = dsim_get_rng_states(h, states.data());
    if (rc) return rc;
  }
  std::vector<double> weights;
  if (h->host_mesh && !h->keep_g[0].empty()) {
    weights.resize((size_t)n);
    mesh_walker_cost(*h->host_mesh, pos, n, weights.data());
  }
  std::vector<int64_t> ids((size_t)n), start(G + 1);
  int rc = dsim_partition_walkers(pos, n, (int32_t)G, weights.empty() ? nullptr : weights.data(), ids.data(), start.data());
  if (rc) return rc;
  bool resize = false;
  for (size_t k = 0; k < G; k++) resize = resize || start[k + 1] - start[k] != h->shards[k]->n;
  /* Prepared aside, committed at the end: the new shards - or, with unchanged sizes, the new device copies of
   * the index lists - are made first, and the handle (shards, shard_ids, shard_off) changes only when EVERY shard has succeeded.
   * A failure on one device (an allocation, say) frees what was prepared and leaves the handle as it was. */
  std::vector<std::vector<int64_t>> new_ids(G);
  std::vector<dsim_sim *> fresh(G, nullptr); /* resize: the shards with their new numbers of walkers (no mesh yet) */
  std::vector<int64_t *> fresh_ids(G, nullptr); /* else: the device arrays the old shards will take */
  rc = multi_each(h, [&](int k, dsim_sim *old, int64_t) -> int {
    std::vector<int64_t> &mine = new_ids[(size_t)k];
    mine.assign(ids.begin() + start[(size_t)k], ids.begin() + start[(size_t)k + 1]);
    if (resize) {
      dsim_sim *sh = nullptr;
      int rc1 = dsim_create(&sh, old->device, (int64_t)mine.size(), h->first, h->M, h->T, h->keep_g[0].data(), h->keep_g[1].data(),
                            h->keep_g[2].data(), h->dt, h->step_l);
      if (rc1) return rc1;
      fresh[(size_t)k] = sh;
      const int *t = h->keep_tune;
      rc1 = dsim_set_tuning(sh, t[0], t[1], t[2], t[3], t[4], t[5]);
      if (rc1 == DSIM_OK && h->keep_recording) rc1 = dsim_set_recording(sh, 1);
      if (rc1) return rc1;
      sh->first = h->first; /* slot i holds walker first + ids[i] */
      return dsim_set_walker_ids(sh, mine.data());
    }
    HIP_TRY(hipSetDevice(old->device));
    HIP_TRY(hipMalloc(&fresh_ids[(size_t)k], std::max<size_t>(1, mine.size()) * sizeof(int64_t)));
    HIP_TRY(hipMemcpy(fresh_ids[(size_t)k], mine.data(), mine.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    return DSIM_OK;
  });
  if (rc) { /* nothing of the handle has been touched */
    const std::string why = g_err;
    for (size_t k = 0; k < G; k++) {
      if (fresh[k]) (void)dsim_destroy(fresh[k]);
      if (fresh_ids[k]) {
        (void)hipSetDevice(h->shards[k]->device);
        (void)hipFree(fresh_ids[k]);
      }
    }
    return fail(rc, "%s", why.c_str());
  }
  for (size_t k = 0; k < G; k++) { /* commit: cannot fail */
    dsim_sim *old = h->shards[k];
    if (resize) {
      fresh[k]->mesh = old->mesh; /* the device copy of the index moves over */
      old->mesh = DevMesh();
      (void)dsim_destroy(old);
      h->shards[k] = fresh[k];
    } else {
      (void)hipSetDevice(old->device);
      if (old->walker_ids) (void)hipFree(old->walker_ids);
      old->walker_ids = fresh_ids[k];
      old->first = h->first;
    }
    h->shard_off[k] = start[k];
  }
  h->shard_ids = std::move(new_ids);
  rc = dsim_set_positions(h, pos);
  if (rc) return rc;
  if (h->seeded) return dsim_rng_init_device(h, h->seed_value);
  if (h->states_given) return dsim_set_rng_states(h, states.data());
  return DSIM_OK;
}

extern "C" int dsim_get_positions(dsim_handle h, double *pos) {
  if (IS_MULTI(h)) {
    if (!pos) return fail(DSIM_ERR_INVALID, "dsim_get_positions: null argument");
    return multi_each(h, [&](int k, dsim_sim *sh, int64_t off) -> int {
      if (h->shard_ids.empty()) return dsim_get_positions(sh, pos + 3 * off);
      std::vector<double> part((size_t)sh->n * 3);
      const int rc = dsim_get_positions(sh, part.data());
      if (rc == DSIM_OK) shard_rows_out(h, k, off, pos, 24, part.data(), sh->n);
      return rc;
    });
  }
  if (!h || !pos) return fail(DSIM_ERR_INVALID, "dsim_get_positions: null argument");
  HIP_TRY(hipSetDevice(h->device));
  int rc = ensure_staging(h, (size_t)h->n * 3 * sizeof(double));
  if (rc) return rc;
  soa_to_aos_kernel<<<grid_for(h->n, 256), 256, 0, h->stream>>>(h->x[h->cur], h->y[h->cur], h->z[h->cur], h->n, (double *)h->staging);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(pos, h->staging, (size_t)h->n * 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return DSIM_OK;
}

extern "C" int dsim_set_rng_states(dsim_handle h, const uint64_t *st) {
  if (IS_MULTI(h)) {
    if (!st) return fail(DSIM_ERR_INVALID, "dsim_set_rng_states: null argument");
    h->states_given = true, h->seeded = false;
    return multi_each(h, [&](int k, dsim_sim *sh, int64_t off) -> int {
      std::vector<char> tmp;
      return dsim_set_rng_states(sh, (const uint64_t *)shard_rows_in(h, k, off, st, 16, tmp));
    });
  }
  if (!h || !st) return fail(DSIM_ERR_INVALID, "dsim_set_rng_states: null argument");
  HIP_TRY(hipSetDevice(h->device));
  int rc = ensure_staging(h, (size_t)h->n * 2 * sizeof(uint64_t));
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(h->staging, st, (size_t)h->n * 2 * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
  states_to_soa_kernel<<<grid_for(h->n, 256), 256, 0, h->stream>>>((const uint64_t *)h->staging, h->n, h->s0[h->cur], h->s1[h->cur]);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  return DSIM_OK;
}

extern "C" int dsim_get_rng_states(dsim_handle h, uint64_t *st) {
  if (IS_MULTI(h)) {
    if (!st) return fail(DSIM_ERR_INVALID, "dsim_get_rng_states: null argument");
    return multi_each(h, [&](int k, dsim_sim *sh, int64_t off) -> int {
      if (h->shard_ids.empty()) return dsim_get_rng_states(sh, st + 2 * off);
      std::vector<uint64_t> part((size_t)sh->n * 2);
      const int rc = dsim_get_rng_states(sh, part.data());
      if (rc == DSIM_OK) shard_rows_out(h, k, off, st, 16, part.data(), sh->n);
      return rc;
    });
  }
  if (!h || !st) return fail(DSIM_ERR_INVALID, "dsim_get_rng_states: null argument");
  HIP_TRY(hipSetDevice(h->device));
  int rc = ensure_staging(h, (size_t)h->n * 2 * sizeof(uint64_t));
  if (rc) return rc;
  soa_to_states_kernel<<<grid_for(h->n, 256), 256, 0, h->stream>>>(h->s0[h->cur], h->s1[h->cur], h->n, (uint64_t *)h->staging);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(st, h->staging, (size_t)h->n * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return DSIM_OK;
}

extern "C" int dsim_rng_init_device(dsim_handle h, uint64_t seed) {
  if (IS_MULTI(h)) h->seeded = true, h->states_given = false, h->seed_value = seed;
  MULTI_FORWARD(h, dsim_rng_init_device(s_, seed));
  if (!h) return fail(DSIM_ERR_INVALID, "null handle");
  if (((uint64_t)h->first + (uint64_t)h->n) >> 47) return fail(DSIM_ERR_INVALID, "walker index range too large");
  HIP_TRY(hipSetDevice(h->device));
  const JumpTables &jt = jump_tables();
  size_t bytes = jt.cols.size() * sizeof(HostRng);
  int rc = ensure_staging(h, bytes);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(h->staging, jt.cols.data(), bytes, hipMemcpyHostToDevice, h->stream));
  HostRng s = h_seed_state(seed);
  rng_init_kernel<<<grid_for(h->n, 256), 256, 0, h->stream>>>(s.s0, s.s1, (uint64_t)h->first, h->walker_ids, h->n, (const uint64_t *)h->staging, jt.levels,
                                                             h->s0[h->cur], h->s1[h->cur]);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  return DSIM_OK;
}

extern "C" int dsim_get_phases(dsim_handle h, double *phases) {
  if (IS_MULTI(h)) {
    if (!phases) return fail(DSIM_ERR_INVALID, "dsim_get_phases: null argument");
    return multi_each(h, [&](int k_, dsim_sim *sh, int64_t off) -> int {
      std::vector<double> part((size_t)sh->M * (size_t)sh->n);
      const int rc = dsim_get_phases(sh, part.data());
      if (rc) return rc;
      for (int m = 0; m < sh->M; m++) shard_rows_out(h, k_, off, phases + (size_t)m * (size_t)h->n, 8, part.data() + (size_t)m * (size_t)sh->n, sh->n);
      return DSIM_OK;
    });
  }
  if (!h || !phases) return fail(DSIM_ERR_INVALID, "dsim_get_phases: null argument");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipMemcpyAsync(phases, h->phase, (size_t)h->M * h->n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return DSIM_OK;
}

/* all_signals: cos(phase) with the NaN mask, computed on the device block by block; block k is written by the compute
 * stream into one of two device buffers, crosses PCIe on the copy stream into one of two pinned buffers, and is copied
 * out by the host while block k + 1 is already on its way. */
extern "C" int dsim_get_all_signals(dsim_handle h, double *signals) {
  if (IS_MULTI(h)) {
    if (!signals) return fail(DSIM_ERR_INVALID, "dsim_get_all_signals: null argument");
    return multi_each(h, [&](int k_, dsim_sim *sh, int64_t off) -> int {
      std::vector<double> part((size_t)sh->M * (size_t)sh->n);
      const int rc = dsim_get_all_signals(sh, part.data());
      if (rc) return rc;
      for (int m = 0; m < sh->M; m++) shard_rows_out(h, k_, off, signals + (size_t)m * (size_t)h->n, 8, part.data() + (size_t)m * (size_t)sh->n, sh->n);
      return DSIM_OK;
    });
  }
  if (!h || !signals) return fail(DSIM_ERR_INVALID, "dsim_get_all_signals: null argument");
  HIP_TRY(hipSetDevice(h->device));
  const int64_t total = (int64_t)h->M * h->n;
  const int64_t per = std::min<int64_t>(total, (int64_t)4 << 20); /* 32 MB blocks */
  const size_t need = (size_t)per * sizeof(double);
  if (!h->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
  for (int b = 0; b < 2; b++)
    if (!h->pin_ev[b]) HIP_TRY(hipEventCreateWithFlags(&h->pin_ev[b], hipEventDisableTiming));
  if (h->pin_bytes < need) {
    for (int b = 0; b < 2; b++) {
      if (h->pin[b]) (void)hipHostFree(h->pin[b]);
      h->pin[b] = nullptr;
    }
    h->pin_bytes = 0;
    for (int b = 0; b < 2; b++) HIP_TRY(hipHostMalloc(&h->pin[b], need, hipHostMallocDefault));
    h->pin_bytes = need;
  }
  int rc = ensure_staging(h, 2 * need);
  if (rc) return rc;
  hipEvent_t done[2] = {nullptr, nullptr};
  for (int b = 0; b < 2; b++) HIP_TRY(hipEventCreateWithFlags(&done[b], hipEventDisableTiming));
  auto finish = [&](int code) {
    (void)hipStreamSynchronize(h->stream);
    (void)hipStreamSynchronize(h->copy_stream);
    for (int b = 0; b < 2; b++) (void)hipEventDestroy(done[b]);
    return code;
  };
  int64_t prev_first = 0, prev_count = 0;
  int k = 0;
  for (int64_t first = 0; first < total; first += per, k++) {
    const int b = k & 1;
    const int64_t count = std::min(per, total - first);
    double *dev = (double *)h->staging + (size_t)b * per;
    const int64_t row0 = first / h->n, row1 = (first + count - 1) / h->n;
    all_signals_kernel<<<dim3((unsigned)grid_for(h->n, 256), (unsigned)(row1 - row0 + 1)), 256, 0, h->stream>>>(h->phase, h->exc, h->n, first, count, row0, dev);
    if (hipGetLastError() != hipSuccess) return finish(fail(DSIM_ERR_HIP, "all_signals_kernel launch failed"));
    if (hipEventRecord(done[b], h->stream) != hipSuccess || hipStreamWaitEvent(h->copy_stream, done[b], 0) != hipSuccess ||
        hipMemcpyAsync(h->pin[b], dev, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, h->copy_stream) != hipSuccess ||
        hipEventRecord(h->pin_ev[b], h->copy_stream) != hipSuccess)
      return finish(fail(DSIM_ERR_HIP, "dsim_get_all_signals: copy failed"));
    if (prev_count) { /* the previous block is consumed before its buffers are reused two blocks later */
      if (hipEventSynchronize(h->pin_ev[1 - b]) != hipSuccess) return finish(fail(DSIM_ERR_HIP, "dsim_get_all_signals: copy failed"));
      memcpy(signals + prev_first, h->pin[1 - b], (size_t)prev_count * sizeof(double));
    }
    prev_first = first, prev_count = count;
  }
  if (prev_count) {
    const int b = (k - 1) & 1;
    if (hipEventSynchronize(h->pin_ev[b]) != hipSuccess) return finish(fail(DSIM_ERR_HIP, "dsim_get_all_signals: copy failed"));
    memcpy(signals + prev_first, h->pin[b], (size_t)prev_count * sizeof(double));
  }
  return finish(DSIM_OK);
}

extern "C" int dsim_get_iter_exc(dsim_handle h, uint8_t *exc) {
  if (IS_MULTI(h)) {
    if (!exc) return fail(DSIM_ERR_INVALID, "dsim_get_iter_exc: null argument");
    return multi_each(h, [&](int k, dsim_sim *sh, int64_t off) -> int {
      if (h->shard_ids.empty()) return dsim_get_iter_exc(sh, exc + off);
      std::vector<uint8_t> part((size_t)sh->n);
      const int rc = dsim_get_iter_exc(sh, part.data());
      if (rc == DSIM_OK) shard_rows_out(h, k, off, exc, 1, part.data(), sh->n);
      return rc;
    });
  }
  if (!h || !exc) return fail(DSIM_ERR_INVALID, "dsim_get_iter_exc: null argument");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipMemcpyAsync(exc, h->exc, (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return DSIM_OK;
}

/* exact sum q * 2^-60 of a signal from its limbs (dsim_kernels.h), rounded once to double */
static double signal_from_limbs(int64_t hi, int64_t lo) {
  const __int128 q = ((__int128)hi << DS_SIG_LIMB_BITS) + (__int128)lo;
  return (double)q * (1.0 / DS_SIG_SCALE); /* int128 -> double is correctly rounded, the power of two exact */
}

extern "C" int dsim_get_signal_limbs(dsim_handle h, int64_t *limbs, int64_t *n_exceeded) {
  if (!h || !limbs) return fail(DSIM_ERR_INVALID, "dsim_get_signal_limbs: null argument");
  if (IS_MULTI(h)) {
    /* integer sums: the shards' limbs simply add */
    const size_t G = h->shards.size(), L = (size_t)4 * (size_t)h->M;
    std::vector<int64_t> part(G * L);
    std::vector<int64_t> ne(G, 0);
    const int rc = multi_each(h, [&](int k, dsim_sim *sh, int64_t) -> int { return dsim_get_signal_limbs(sh, &part[(size_t)k * L], &ne[(size_t)k]); });
    if (rc) return rc;
    int64_t total = 0;
    for (size_t q = 0; q < L; q++) {
      int64_t t = 0;
      for (size_t k = 0; k < G; k++) t += part[k * L + q];
      limbs[q] = t;
    }
    for (size_t k = 0; k < G; k++) total += ne[k];
    if (n_exceeded) *n_exceeded = total;
    return DSIM_OK;
  }
  HIP_TRY(hipSetDevice(h->device));
  dim3 grid(h->sig_blocks, h->M);
  signal_partial_kernel<<<grid, 256, 0, h->stream>>>(h->phase, h->exc, h->n, h->partial, h->nexc_partial);
  HIP_TRY(hipGetLastError());
  signal_final_kernel<<<h->M + 1, 64, 0, h->stream>>>(h->partial, h->nexc_partial, h->sig_blocks, h->M, h->sig_out);
  HIP_TRY(hipGetLastError());
  std::vector<long long> out((size_t)4 * h->M + 1);
  HIP_TRY(hipMemcpyAsync(out.data(), h->sig_out, out.size() * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (size_t q = 0; q < (size_t)4 * h->M; q++) limbs[q] = (int64_t)out[q];
  if (n_exceeded) *n_exceeded = (int64_t)out[(size_t)4 * h->M];
  return DSIM_OK;
}

extern "C" int dsim_signal_from_limbs(int32_t M, const int64_t *limbs, double *out_cos, double *out_sin) {
  if (M < 1 || !limbs || !out_cos) return fail(DSIM_ERR_INVALID, "dsim_signal_from_limbs: bad argument");
  for (int m = 0; m < M; m++) {
    out_cos[m] = signal_from_limbs(limbs[4 * m], limbs[4 * m + 1]);
    if (out_sin) out_sin[m] = signal_from_limbs(limbs[4 * m + 2], limbs[4 * m + 3]);
  }
  return DSIM_OK;
}

extern "C" int dsim_get_signal(dsim_handle h, double *out_cos, double *out_sin, int64_t *n_exceeded) {
  if (!h || !out_cos) return fail(DSIM_ERR_INVALID, "dsim_get_signal: null argument");
  std::vector<int64_t> limbs((size_t)4 * (size_t)h->M);
  const int rc = dsim_get_signal_limbs(h, limbs.data(), n_exceeded);
  if (rc) return rc;
  return dsim_signal_from_limbs(h->M, limbs.data(), out_cos, out_sin);
}

/* ------------------------------------------------------------------- mesh --- */


extern "C" int dsim_mesh_clearance_map(const double *vertices, int64_t n_vertices, const int64_t *faces, int64_t n_faces, const double *xs,
                                       const double *ys, const double *zs, const int64_t n_sv[3], double step_l, int32_t *clear_idx,
                                       uint32_t *clear_bits, int64_t *n_words, int32_t *log2_cells) {
  if (!vertices || !faces || !xs || !ys || !zs || !n_sv || !n_words || !log2_cells) return fail(DSIM_ERR_INVALID, "dsim_mesh_clearance_map: null argument");
  for (int64_t i = 0; i < n_faces * 3; i++)
    if (faces[i] < 0 || faces[i] >= n_vertices) return fail(DSIM_ERR_INVALID, "dsim_mesh_clearance_map: face out of range");
  std::vector<int32_t> cidx;
  std::vector<uint32_t> cbits;
  int lf = 0;
  double frac = 0;
  const double *bb[3] = {xs, ys, zs};
  build_clearance_map(vertices, faces, n_faces, bb, n_sv, step_l, cidx, cbits, lf, frac);
  if (cidx.empty()) { /* no map for this mesh / step length: nothing is ever skipped */
    *n_words = 0;
    *log2_cells = -1;
    return DSIM_OK;
  }
  if (clear_idx && clear_bits) {
    if (*n_words < (int64_t)cbits.size()) return fail(DSIM_ERR_INVALID, "dsim_mesh_clearance_map: clear_bits too small");
    memcpy(clear_idx, cidx.data(), cidx.size() * sizeof(int32_t));
    memcpy(clear_bits, cbits.data(), cbits.size() * sizeof(uint32_t));
  }
  *n_words = (int64_t)cbits.size();
  *log2_cells = lf;
  return DSIM_OK;
}

/* host only: the fine grid dsim_set_mesh builds for this mesh and step length */
extern "C" int dsim_mesh_fine_grid(const double *vertices, int64_t n_vertices, const int64_t *faces, int64_t n_faces, const double *xs,
                                   const double *ys, const double *zs, const int64_t n_sv[3], const int64_t *subvoxel_indices,
                                   const int64_t *triangle_indices, int64_t n_refs, double step_l, int32_t *log2_cells, int64_t *n_cells,
                                   int64_t *n_records, int32_t *sv_base, uint32_t *cell_start, int32_t *rec_triangle, int32_t *rec_position,
                                   float *rec_box, uint32_t *rec_packed, float *quantum) {
  if (!log2_cells || !n_cells || !n_records) return fail(DSIM_ERR_INVALID, "dsim_mesh_fine_grid: null argument");
  HostFineGrid fg;
  DevMesh none;
  int rc = upload_mesh(none, nullptr, vertices, n_vertices, faces, n_faces, xs, ys, zs, n_sv, subvoxel_indices, triangle_indices, n_refs, step_l, &fg);
  if (rc) return rc;
  const int64_t cells = (int64_t)fg.cell_start.size() - 1, recs = (int64_t)fg.cell_start.back();
  if (cell_start) { /* second call: the caller's buffers hold the sizes of the first */
    if (*n_cells < cells || *n_records < recs) return fail(DSIM_ERR_INVALID, "dsim_mesh_fine_grid: buffers too small");
    if (sv_base && fg.lk > 0) memcpy(sv_base, fg.sv_base.data(), fg.sv_base.size() * sizeof(int32_t));
    memcpy(cell_start, fg.cell_start.data(), fg.cell_start.size() * sizeof(uint32_t));
    for (int64_t r = 0; r < recs; r++) {
      const HostRecord &q = fg.recs[(size_t)r];
      if (rec_triangle) rec_triangle[r] = q.tri;
      if (rec_position) rec_position[r] = q.pos;
      if (rec_box) {
        const float b[6] = {q.lo[0], q.lo[1], q.lo[2], q.hi[0], q.hi[1], q.hi[2]}; /* min x y z, max x y z */
        memcpy(rec_box + 6 * r, b, sizeof(b));
      }
      if (rec_packed) { /* the record and the plane of its triangle */
        memcpy(rec_packed + 8 * r, &fg.mrec[(size_t)r], 16);
        memcpy(rec_packed + 8 * r + 4, &fg.planes[(size_t)q.tri], 16);
      }
    }
    if (quantum) *quantum = fg.qs;
  }
  *log2_cells = fg.lk;
  *n_cells = cells;
  *n_records = recs;
  return DSIM_OK;
}

extern "C" int dsim_set_mesh(dsim_handle h, const double *vertices, int64_t n_vertices, const int64_t *faces, int64_t n_faces,
                             const double *xs, const double *ys, const double *zs, const int64_t n_sv[3],
                             const int64_t *subvoxel_indices, const int64_t *triangle_indices, int64_t n_refs) {
  if (!h) return fail(DSIM_ERR_INVALID, "null handle");
  if (IS_MULTI(h)) {
    /* [r5] the index is built ONCE on the host and handed to every shard, which only allocates and copies on its own
     * device (round 4: every shard built it - G times the host work, G x 32 threads, G temporary copies) */
    std::shared_ptr<HostMesh> built(new HostMesh());
    DevMesh none;
    int rc = upload_mesh(none, nullptr, vertices, n_vertices, faces, n_faces, xs, ys, zs, n_sv, subvoxel_indices, triangle_indices, n_refs,
                         h->step_l, nullptr, built.get());
    if (rc) return rc;
    h->host_mesh = built;
    const HostMesh &hm = *built;
    return multi_each(h, [&](int, dsim_sim *sh, int64_t) -> int {
      HIP_TRY(hipSetDevice(sh->device));
      return upload_mesh_host(sh->mesh, sh->stream, hm);
    });
  }
  HIP_TRY(hipSetDevice(h->device));
  return upload_mesh(h->mesh, h->stream, vertices, n_vertices, faces, n_faces, xs, ys, zs, n_sv, subvoxel_indices, triangle_indices, n_refs,
                     h->step_l);
}

/* ------------------------------------------------------------------ launch --- */

/* The scalar-load (SMEM) gradient path was measured at half the speed of the LDS tiles
 * (C2: 3.1e10 vs 6.0e10 spin-steps/s); its kernels are only instantiated when the library is
 * built with -DDSIM_SMEM_GRADIENT (halves the build time otherwise). */
/* cooperative mesh kernels with at most this many phase accumulators run at 4 waves/SIMD (4 accumulators: 125 VGPRs,
 * nothing spilled; 8 would not fit 128) */
#define COOP4_MAX_MB 4

template <int GEOM, int MB>
static hipError_t launch_one(const WalkParams &p, int grid, int block, size_t lds, hipStream_t s, int gsrc, int coop) {
  if (GEOM == GEOM_MESH && coop) {
    /* the wave queues need more than the default 64 KB of dynamic LDS */
    /* MB <= COOP4_MAX_MB: the 4-waves/SIMD kernel; it reads the gradient samples from global memory
     * with the lane's own time index (32 or 64 bytes per step out of the L1), so there is no
     * LDS tile and no workgroup barrier in the time loop */
    auto k = MB <= COOP4_MAX_MB ? walk_kernel_coop4<(MB <= COOP4_MAX_MB ? MB : 1), 1> : walk_kernel<GEOM, MB, 0, GEOM == GEOM_MESH>;
    if (MB == 1 && p.traj) k = walk_kernel_coop4_traj;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    k<<<grid, block, lds, s>>>(p);
  } else {
#ifdef DSIM_SMEM_GRADIENT
    if (gsrc)
      walk_kernel<GEOM, MB, 1, 0><<<grid, block, lds, s>>>(p);
    else
#endif
        if (MB == 32 && p.traj)
      walk_kernel_traj32<GEOM><<<grid, block, lds, s>>>(p);
    else
      walk_kernel<GEOM, MB, 0, 0><<<grid, block, lds, s>>>(p);
  }
  (void)gsrc;
  return hipGetLastError();
}

template <int GEOM>
static hipError_t launch_mb(int mb, const WalkParams &p, int grid, int block, size_t lds, hipStream_t s, int gsrc, int coop) {
  switch (mb) {
    case 1: return launch_one<GEOM, 1>(p, grid, block, lds, s, gsrc, coop);
    case 2: return launch_one<GEOM, 2>(p, grid, block, lds, s, gsrc, coop);
    case 4: return launch_one<GEOM, 4>(p, grid, block, lds, s, gsrc, coop);
    case 8: return launch_one<GEOM, 8>(p, grid, block, lds, s, gsrc, coop);
    case 16: return launch_one<GEOM, 16>(p, grid, block, lds, s, gsrc, coop);
    default: return launch_one<GEOM, 32>(p, grid, block, lds, s, gsrc, coop);
  }
}

static int pick_mb(int M, int forced) {
  int cap = forced > 0 ? forced : 32;
  int mb = 1;
  while (mb < M && mb < cap && mb < 32) mb <<= 1;
  return mb;
}

/* ---- spatial re-sorting of walkers (mesh substrates) ------------------------------
 * Walkers are independent, so their slots may be permuted freely.  Every
 * `resort` time steps the walkers are sorted by the Morton code of their
 * (periodically wrapped) position, so that the lanes of a wavefront sit in the
 * same few subvoxels: their index / triangle gathers then hit the same cache
 * lines (the vector L1 serves a wave-wide load at one line per cycle, so 64
 * scattered addresses cost ~64x a coherent one) and their candidate loops have
 * similar trip counts.  At the end of the run the original walker order is
 * restored, so no other entry point sees the permutation. */

static int ensure_sort_buffers(dsim_sim *h) {
  if (h->sort_ready) return DSIM_OK;
  const int64_t n = h->n;
  HIP_TRY(hipMalloc(&h->phase_alt, (size_t)h->M * n * sizeof(double)));
  HIP_TRY(hipMalloc(&h->exc_alt, n));
  for (int b = 0; b < 2; b++) HIP_TRY(hipMalloc(&h->orig[b], n * sizeof(int32_t)));
  HIP_TRY(hipMalloc(&h->bin_of, n * sizeof(uint32_t)));
  /* 64^3 bins of the periodic voxel; measured on C5b at 1e7 walkers: 32^3 ... 256^3 bins give the same walk-kernel rate
   * (1.24e10 +- 0.3 %) - what matters is that a wavefront's walkers share a neighbourhood, not their order inside it -
   * and the single-workgroup scan of 2^24 bins costs 3.5 ms (DSIM_SORT_BITS: tuning runs) */
  /* [r4] re-measured with the overlapping cells: 21 bits gain 0.2 % for M = 1 (A/B) and cost a small shard's recorded mode,
   * which sorts at every launch, 9 % (the 3.6 ms scan of 2^21 bins): profiles/history/r04_sweeps_xcd_run_sort_bits_groups.txt */
  h->sort_bits = 18;
  if (const char *e = getenv("DSIM_SORT_BITS")) {
    const int v = atoi(e);
    if (v >= 6 && v <= 24) h->sort_bits = v;
  }
  const size_t n_bins = ((size_t)1 << h->sort_bits) + 1;
  for (int g = 0; g < DSIM_MAX_GROUPS; g++) {
    HIP_TRY(hipMalloc(&h->sort_hist[g], n_bins * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&h->sort_offs[g], n_bins * sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(h->sort_hist[g], 0, n_bins * sizeof(uint32_t), h->stream));
  }
  for (int b = 0; b < 2; b++) HIP_TRY(hipMalloc(&h->tcur[b], n * sizeof(int32_t)));
  HIP_TRY(hipMalloc(&h->counters, DSIM_MAX_GROUPS * 2 * 4 * sizeof(unsigned long long)));
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->sort_ready = true;
  return DSIM_OK;
}

/* Sort the walkers of the current state buffer into Morton bins (three kernels, see dsim_kernels.h); flips h->cur.
 * `grp`: which histogram / cursor arrays to use (a walker group of run_own_clock is a view of the handle with its own).
 * count_clear: also count the walkers in a clear cell of `clear_map` into *count_clear; zero4: four counters the scan
 * kernel clears on the way (for the launch after this one). */
static int resort_walkers(dsim_sim *h, bool first, bool clocks = false, int32_t t_end = 0, bool by_steps_left = false, int grp = 0,
                          const MeshDev *clear_map = nullptr, unsigned long long *count_clear = nullptr, unsigned long long *zero4 = nullptr) {
  const int64_t n = h->n;
  const int g = grid_for(n, 256);
  const int c = h->cur, o = 1 - h->cur;
  const DevMesh &dm = h->mesh;
  /* the clocks travel with the state buffers: tcur[cur] belongs to x[cur] */
  SortParams sp = {};
  sp.x = h->x[c], sp.y = h->y[c], sp.z = h->z[c];
  sp.n = n;
  sp.inv_lx = dm.inv_l[0], sp.inv_ly = dm.inv_l[1], sp.inv_lz = dm.inv_l[2];
  sp.tcur = clocks ? h->tcur[c] : nullptr;
  sp.t_end = t_end;
  sp.by_steps_left = by_steps_left ? 1 : 0;
  sp.bits = h->sort_bits;
  sp.bin_of = h->bin_of;
  sp.hist = h->sort_hist[grp];
  sp.clear_count = clear_map != nullptr ? count_clear : nullptr;
  const MeshDev no_map = {};
  sort_count_kernel<<<g, 256, 0, h->stream>>>(sp, clear_map != nullptr ? *clear_map : no_map);
  HIP_TRY(hipGetLastError());
  sort_scan_kernel<<<1, SORT_SCAN_THREADS, 0, h->stream>>>(h->sort_hist[grp], h->sort_offs[grp], (1 << h->sort_bits) + 1, zero4);
  HIP_TRY(hipGetLastError());
  ScatterParams q = {};
  q.n = n, q.phase_stride = h->phase_stride, q.M = h->M;
  q.bin_of = h->bin_of, q.offs = h->sort_offs[grp];
  q.x = h->x[c], q.y = h->y[c], q.z = h->z[c], q.s0 = h->s0[c], q.s1 = h->s1[c];
  q.exc = h->exc;
  q.orig = first ? nullptr : h->orig[h->orig_cur];
  q.tcur = clocks ? h->tcur[c] : nullptr;
  q.phase = h->phase;
  q.xo = h->x[o], q.yo = h->y[o], q.zo = h->z[o], q.s0o = h->s0[o], q.s1o = h->s1[o];
  q.exco = h->exc_alt;
  q.origo = h->orig[1 - h->orig_cur];
  q.tcuro = h->tcur[o];
  q.phaseo = h->phase_alt;
  sort_scatter_kernel<<<g, 256, 0, h->stream>>>(q);
  HIP_TRY(hipGetLastError());
  std::swap(h->phase, h->phase_alt);
  std::swap(h->exc, h->exc_alt);
  h->orig_cur = 1 - h->orig_cur;
  h->cur = o;
  return DSIM_OK;
}

/* undo the permutation: slot i goes back to walker orig[i]; flips h->cur */
static int restore_walker_order(dsim_sim *h) {
  const int64_t n = h->n;
  const int g = grid_for(n, 256);
  const int c = h->cur, o = 1 - h->cur;
  const int32_t *orig = h->orig[h->orig_cur];
  scatter_state_kernel<<<g, 256, 0, h->stream>>>(orig, n, h->x[c], h->y[c], h->z[c], h->s0[c], h->s1[c], h->exc, h->x[o], h->y[o], h->z[o],
                                                 h->s0[o], h->s1[o], h->exc_alt);
  HIP_TRY(hipGetLastError());
  dim3 pg(g, h->M);
  scatter_rows_kernel<<<pg, 256, 0, h->stream>>>(orig, n, h->phase_stride, h->phase, h->phase_alt);
  HIP_TRY(hipGetLastError());
  std::swap(h->phase, h->phase_alt);
  std::swap(h->exc, h->exc_alt);
  h->cur = o;
  return DSIM_OK;
}

/* how many walkers currently sit in a clear cell of the clearance map */
__global__ void clear_count_kernel(const double *x, const double *y, const double *z, int64_t n, const MeshDev ms, unsigned long long *count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool c = false;
  if (i < n) {
    const V3 r = {x[i], y[i], z[i]};
    c = walker_is_clear(ms, r);
  }
  const unsigned long long m = __ballot(c);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, (unsigned long long)__popcll(m));
}

/* ---- mesh runs with per-walker clocks (cooperative kernel, re-sorted walkers, nothing recorded) ---------------
 * A launch is a budget of passes per wavefront, not a range of time steps: a lane that needs fewer passes than its
 * neighbours (no reflections) simply gets further instead of idling until the slowest lane of its wavefront has
 * finished the range.  tcur[i] holds the next time step of walker i between launches; walkers that have reached
 * t_end are sorted to the end, where their wavefronts exit at once.  The call returns when every walker is at
 * t_end, so nothing outside sees the clocks.
 *
 * Walker groups.  All wavefronts of a launch get the same budget, so they finish together, round after round of
 * the 4096 wavefront slots of the GPU - and the last round is as full as the walker count happens to make it
 * (1.25e6 walkers = 4.77 rounds: 5 % of the slots idle), after which the re-sort and the read-back of the
 * progress counters run on an otherwise empty device.  With enough walkers the shard is therefore cut
 * into two halves that are sorted and launched independently of each other on two streams (walkers do not
 * interact; every array of the handle is simply addressed at an offset): while one half drains, sorts and
 * reports, the other half's launch has the machine.  Same-box A/B on C5b: +4.5 % at 1.25e6 walkers, +0.6 % at 1e7
 * (C4, 1e6 walkers: +8 %).  The blocks of the launch that was issued first are dispatched first, so the halves
 * take turns by themselves; shortening the second half's first launch to put them half a launch apart cost 1-3 %.
 * The host drives both halves from this one thread: issue A, issue B, then wait for whichever was issued first,
 * read its counters and issue its next launch. */
struct WalkGroup {
  dsim_sim v; /* the handle's arrays narrowed to this group's walkers (a view: owns nothing) */
  int64_t off = 0;
  std::vector<hipEvent_t> *ev = nullptr;
  int *ev_used = nullptr;
  unsigned long long *counters = nullptr; /* [2][4] on the device: see dsim_sim::counters */
  int launch_no = 0;                      /* launch k uses counter set k & 1 */
  double *traj = nullptr;                 /* recorded mode: this group's rows[slot][step][x y z 0] of the handle's buffer */
  int32_t t_pos = 0;                      /* recorded mode: the time step the group has reached */
  double clear_frac = -1;                 /* share of walkers in clear cells as of the last launch's sort (-1: unknown) */
  long long unfinished = 0, steps_left = 0, max_left = 0;
  double passes_per_step = 1.45; /* running estimate: passes a wavefront spends per time step of its walkers */
  int budget = INT_MAX, launches_left = 1 << 20;
  bool permuted = false, inflight = false;
};

/* Dynamic LDS the phase kernel asks for WITHOUT using it: with none, its workgroups (112 VGPRs, no LDS of their own) slip
 * into every CU next to the other walker group's walk workgroups and stream the recorded positions - 17 GB per launch of
 * 5e6 walkers - through the L2s the traversal's index lives in; the walk kernel slows by more than the overlap is worth
 * (C5b, 1e7 walkers, M = 32: 1.025e10 against 1.04e10 for the round-3 kernel, which never found a slot and ran in the gap
 * between two walk kernels).  Asking for a little more LDS than a CU has left beside a full set of walk workgroups lets a
 * phase workgroup in only where a walk workgroup has retired - one per CU while the walk kernel runs, several once it
 * drains: 1.08e10, and 9.35e9 against 8.83e9 at 1.25e6 walkers (profiles/history/r04_ab_m32_phase_kernel_dpp.txt: 0 / 20 /
 * 28 / 32 / 40 / 48 / 64 KB asked for give 1.025-1.037 / 1.025 / 1.077 / 1.081 / 1.065 / 1.001 / 1.05e10).
 * walk_lds: LDS of one walk workgroup (static + dynamic).  DSIM_PHASE_LDS (bytes) overrides it in tuning runs. */
static size_t phase_lds_reserve(size_t walk_lds) {
  if (const char *e = getenv("DSIM_PHASE_LDS")) return (size_t)std::min<long>(std::max<long>(atol(e), 0L), 64L * 1024L); /* a launch cannot ask for more */
  const size_t cu = 160 * 1024, granule = 1280;
  const size_t per_wg = std::max<size_t>((walk_lds + granule - 1) / granule * granule, granule);
  const size_t max_wg = std::min<size_t>(4, cu / per_wg); /* 4 waves per SIMD: at most four workgroups of 256 */
  if (max_wg == 0) return 0;
  const size_t left = cu - max_wg * per_wg;
  return std::min<size_t>(std::min<size_t>(left + 8 * 1024, cu - (max_wg - 1) * per_wg), 64 * 1024);
}

static int walker_groups(const dsim_sim *h) {
  int want = 0; /* development override: DSIM_GROUPS=1..4 (see include/dsim.h, environment) */
  if (const char *e = getenv("DSIM_GROUPS")) want = atoi(e);
  if (want >= 1 && want <= DSIM_MAX_GROUPS) return h->n >= 4096 * (int64_t)want ? want : 1;
  return h->n >= (1 << 18) ? 2 : 1;
}

static int run_own_clock(dsim_sim *h, WalkParams p, const MeshDev &clear_map, bool clear_auto, int mb, int block, size_t lds, int resort,
                         int32_t t_begin, int32_t t_end, int mb_phase = 0) {
  /* mb_phase > 0: more than two measurements - the walk kernel carries no accumulators and records the positions of a
   * range of steps (h->traj, cut into one region per group), phase_from_traj_kernel<mb_phase> then adds the phases up on
   * the same stream.  A launch then is a range of time steps common to the walkers of a group (no per-walker clocks: the
   * accumulation wants one gradient tile per workgroup), but the groups still run on their own streams: one group's
   * sort, walk and phase kernels fill what the other's leaves free. */
  const bool recorded = mb_phase > 0;
  const int n_groups = walker_groups(h);
  if (!h->clear_count) HIP_TRY(hipMalloc(&h->clear_count, sizeof(unsigned long long)));
  /* Walker groups are index ranges of the handle's arrays.  Cut out of the caller's order every group is a random
   * sample of the whole substrate - half the walkers per fine cell, wavefronts that straddle twice as many cells - so
   * the whole shard is sorted in space ONCE first: a group then is a region of the substrate at the shard's full
   * density (walkers mix across the cut only by diffusion).  Every later sort is a group's own. */
  bool presorted = false;
  if (n_groups > 1) {
    int rc = resort_walkers(h, true);
    if (rc) return rc;
    presorted = true;
  }
  WalkGroup grp[DSIM_MAX_GROUPS];
  h->last_groups = n_groups;
  const int64_t part = ((h->n + n_groups - 1) / n_groups + 255) & ~(int64_t)255; /* whole workgroups */
  for (int k = 0; k < n_groups; k++) {
    WalkGroup &g = grp[k];
    g.v = *h;
    g.off = std::min<int64_t>(h->n, k * part);
    g.v.n = std::min<int64_t>(h->n, (k + 1) * part) - g.off;
    g.ev = k == 0 ? &h->walk_ev : &h->grp_walk_ev[k - 1];
    g.ev_used = k == 0 ? &h->walk_ev_used : &h->grp_walk_ev_used[k - 1];
    g.counters = h->counters + k * 8;
    if (k >= 1) {
      if (!h->grp_stream[k - 1]) HIP_TRY(hipStreamCreateWithFlags(&h->grp_stream[k - 1], hipStreamNonBlocking));
      g.v.stream = h->grp_stream[k - 1];
      for (int b = 0; b < 2; b++) {
        g.v.x[b] += g.off, g.v.y[b] += g.off, g.v.z[b] += g.off, g.v.s0[b] += g.off, g.v.s1[b] += g.off;
        g.v.tcur[b] += g.off, g.v.orig[b] += g.off;
      }
      g.v.phase += g.off, g.v.phase_alt += g.off, g.v.exc += g.off, g.v.exc_alt += g.off, g.v.bin_of += g.off;
    }
    g.permuted = presorted; /* orig[] already holds the walkers' original indices */
    g.traj = recorded ? h->traj + (size_t)h->traj_steps * 4 * g.off : nullptr; /* walker-major rows of 32 bytes per step */
    g.t_pos = t_begin;
    g.unfinished = g.v.n;
    g.steps_left = (long long)g.v.n * (t_end - t_begin);
    g.max_left = t_end - t_begin;
  }
  for (int k = 0; k < DSIM_MAX_GROUPS - 1; k++) h->grp_walk_ev_used[k] = 0;
  HIP_TRY(hipMemsetAsync(h->counters, 0, DSIM_MAX_GROUPS * 2 * 4 * sizeof(unsigned long long), h->stream));
  HIP_TRY(hipEventRecord(h->ev0, h->stream));
  for (int k = 1; k < n_groups; k++) HIP_TRY(hipStreamWaitEvent(grp[k].v.stream, h->ev0, 0)); /* after whatever the handle's stream still holds */
  for (int k = 0; k < n_groups; k++) {
    fill_i32_kernel<<<grid_for(grp[k].v.n, 256), 256, 0, grp[k].v.stream>>>(grp[k].v.tcur[grp[k].v.cur], grp[k].v.n, t_begin);
    HIP_TRY(hipGetLastError());
  }
  /* One launch of a group = four kernels: the three of the re-sort and the walk kernel, which adds up the progress
   * counters itself; the counters come back with one small copy.  (Round 2: ~30 dispatches - keys, hipcub's radix
   * sort, two gathers, two memsets, progress - each of which had to find a wavefront slot between the other group's
   * workgroups.) */
  auto issue = [&](WalkGroup &g) -> int {
    dsim_sim *v = &g.v;
    if (--g.launches_left < 0) return fail(DSIM_ERR_STATE, "mesh run makes no progress");
    if (recorded) {
      const int32_t ta = g.t_pos, tb = std::min<int32_t>(t_end, ta + std::min(resort, h->traj_steps));
      /* a launch is as long as the buffer of recorded positions allows (1e7 walkers: 143 steps) and every launch starts
       * with a re-sort: measured on C5b at 1e7 walkers with 32 measurements, sorting only every `resort` steps and
       * recording 286 steps per launch (64 GB) ran at 9.2e9 against 1.05e10 (profiles/history/r04_ab_m32_*) - without
       * per-walker clocks the wavefronts of a launch stay in step, and freshly sorted walkers keep their passes alike */
      {
        int rc = resort_walkers(v, !g.permuted, false, 0, false, (int)(&g - grp));
        if (rc) return rc;
        g.permuted = true;
      }
      WalkParams q = p;
      const int c = v->cur;
      q.n = v->n;
      q.x_in = q.x_out = v->x[c], q.y_in = q.y_out = v->y[c], q.z_in = q.z_out = v->z[c];
      q.s0_in = q.s0_out = v->s0[c], q.s1_in = q.s1_out = v->s1[c];
      q.phase = v->phase, q.exc = v->exc;
      q.t_begin = ta, q.t_end = tb;
      q.tile_steps = tb - ta; /* gradient from global memory: one "tile" per launch */
      q.traj = g.traj;
      q.traj_row_steps = h->traj_steps; /* walker-major: one 32-byte store per walker and step */
      q.rec_col = nullptr;
      q.tcur = nullptr, q.progress = nullptr, q.iter_budget = INT_MAX;
      q.m0 = 0, q.mc = 0, q.write_state = true;
      if (*g.ev_used + 2 > (int)g.ev->size()) {
        hipEvent_t a = nullptr, b = nullptr;
        HIP_TRY(hipEventCreate(&a));
        g.ev->push_back(a);
        HIP_TRY(hipEventCreate(&b));
        g.ev->push_back(b);
      }
      const int grid_unit = 8 << q.xcd_run_log2;
      const int grid = (grid_for(v->n, block) + grid_unit - 1) / grid_unit * grid_unit;
      HIP_TRY(hipEventRecord((*g.ev)[*g.ev_used], v->stream));
      hipError_t e = launch_mb<GEOM_MESH>(1, q, grid, block, lds, v->stream, h->grad_src, 1);
      if (e != hipSuccess) return fail(DSIM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
      HIP_TRY(hipEventRecord((*g.ev)[*g.ev_used + 1], v->stream));
      *g.ev_used += 2;
      h->last_launches++;
      const size_t lds_p = std::max<size_t>(4 * PHASE_ROWS_LDS_PER_WAVE, n_groups > 1 ? phase_lds_reserve(lds + DS_STATIC_LDS_BYTES) : 0); /* 32 KB of staging; see phase_lds_reserve */
      const int grid_p = grid_for(v->n, 256);
      for (int m0 = 0; m0 < h->M; m0 += mb_phase) {
        const int mc = std::min(mb_phase, h->M - m0);
#define DSIM_PHASE_LAUNCH_G(MBV)                                                                                                    \
  phase_from_rows_kernel<MBV><<<grid_p, 256, lds_p, v->stream>>>(g.traj, v->n, tb - ta, h->traj_steps, h->g4, h->M, m0, mc, ta, p.gdt, \
                                                                 v->phase, h->phase_stride)
        switch (mb_phase) {
          case 4: DSIM_PHASE_LAUNCH_G(4); break;
          case 8: DSIM_PHASE_LAUNCH_G(8); break;
          case 16: DSIM_PHASE_LAUNCH_G(16); break;
          default: DSIM_PHASE_LAUNCH_G(32); break;
        }
#undef DSIM_PHASE_LAUNCH_G
        HIP_TRY(hipGetLastError());
      }
      g.t_pos = tb;
      g.inflight = true;
      return DSIM_OK;
    }
    /* aim at `resort` time steps per walker and launch; once nobody has much more than that left, the launch runs
     * every walker to t_end (wavefronts end when their last lane is done, as without clocks) */
    g.budget = g.max_left <= (long long)resort + resort / 4 ? INT_MAX : (int)std::ceil(g.passes_per_step * resort) + 2;
    WalkParams q = p;
    if (clear_auto && g.clear_frac < 0) {
      /* The map pays when whole wavefronts are clear (they then run at the cost of free diffusion); with a
       * minority of clear walkers scattered over the wavefronts its two look-ups per step cost more than the
       * skipped passes save (neuron, walkers inside: 21 % clear, -1.5 %).  Decided per launch from the share of
       * walkers that were clear at the PREVIOUS launch's sort (counted there in passing; the share moves slowly and
       * results do not depend on the choice) - only the first launch of a call has to count and wait. */
      HIP_TRY(hipMemsetAsync(h->clear_count, 0, sizeof(unsigned long long), v->stream));
      clear_count_kernel<<<grid_for(v->n, 256), 256, 0, v->stream>>>(v->x[v->cur], v->y[v->cur], v->z[v->cur], v->n, clear_map, h->clear_count);
      HIP_TRY(hipGetLastError());
      unsigned long long cnt = 0;
      HIP_TRY(hipMemcpyAsync(&cnt, h->clear_count, sizeof(cnt), hipMemcpyDeviceToHost, v->stream));
      HIP_TRY(hipStreamSynchronize(v->stream));
      g.clear_frac = (double)cnt / (double)v->n;
    }
    unsigned long long *set = g.counters + 4 * (g.launch_no & 1), *other = g.counters + 4 * ((g.launch_no + 1) & 1);
    int rc = resort_walkers(v, !g.permuted, true, t_end, g.budget == INT_MAX && g.max_left < t_end - t_begin, (int)(&g - grp),
                            clear_auto ? &clear_map : nullptr, set + 3, other);
    if (rc) return rc;
    g.permuted = true;
    if (clear_auto) {
      h->last_clear_frac = g.clear_frac;
      q.mesh.clear_idx = g.clear_frac >= 0.4 ? clear_map.clear_idx : nullptr;
    }
    const int c = v->cur;
    q.n = v->n; /* q.phase_stride stays the handle's */
    q.x_in = q.x_out = v->x[c];
    q.y_in = q.y_out = v->y[c];
    q.z_in = q.z_out = v->z[c];
    q.s0_in = q.s0_out = v->s0[c];
    q.s1_in = q.s1_out = v->s1[c];
    q.phase = v->phase;
    q.exc = v->exc;
    q.t_begin = t_begin;
    q.t_end = t_end;
    q.traj = nullptr;
    q.tcur = v->tcur[c];
    q.iter_budget = g.budget;
    q.progress = set;
    q.m0 = 0;
    q.mc = std::min(mb, h->M);
    q.write_state = true;
    if (*g.ev_used + 2 > (int)g.ev->size()) {
      hipEvent_t a = nullptr, b = nullptr;
      HIP_TRY(hipEventCreate(&a));
      g.ev->push_back(a);
      HIP_TRY(hipEventCreate(&b));
      g.ev->push_back(b);
    }
    const int grid_unit = 8 << q.xcd_run_log2;
    const int grid = (grid_for(v->n, block) + grid_unit - 1) / grid_unit * grid_unit; /* XCD-aware group mapping in the kernel */
    HIP_TRY(hipEventRecord((*g.ev)[*g.ev_used], v->stream));
    hipError_t e = launch_mb<GEOM_MESH>(mb, q, grid, block, lds, v->stream, h->grad_src, 1);
    if (e != hipSuccess) return fail(DSIM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipEventRecord((*g.ev)[*g.ev_used + 1], v->stream));
    *g.ev_used += 2;
    h->last_launches++;
    g.inflight = true;
    return DSIM_OK;
  };
  /* wait for the group's launch and read how far its walkers got */
  auto collect = [&](WalkGroup &g) -> int {
    if (recorded) {
      HIP_TRY(hipStreamSynchronize(g.v.stream));
      g.inflight = false;
      g.unfinished = g.t_pos < t_end ? g.v.n : 0;
      return DSIM_OK;
    }
    unsigned long long pr[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(pr, g.counters + 4 * (g.launch_no & 1), sizeof(pr), hipMemcpyDeviceToHost, g.v.stream));
    HIP_TRY(hipStreamSynchronize(g.v.stream));
    g.inflight = false;
    g.launch_no++;
    if (clear_auto) g.clear_frac = (double)pr[3] / (double)g.v.n;
    const long long done = g.steps_left - (long long)pr[1];
    if (done > 0 && g.budget != INT_MAX) {
      const double measured = (double)g.budget * (double)g.unfinished / (double)done; /* an upper estimate: finished lanes idle */
      g.passes_per_step = std::min(4.0, std::max(1.0, 0.5 * g.passes_per_step + 0.5 * measured));
    }
    g.unfinished = (long long)pr[0];
    g.steps_left = (long long)pr[1];
    g.max_left = (long long)pr[2];
    return DSIM_OK;
  };
  auto drive = [&]() -> int {
    for (int k = 0; k < n_groups; k++) {
      int rc = issue(grp[k]);
      if (rc) return rc;
    }
    for (bool any = true; any;) {
      any = false;
      for (int k = 0; k < n_groups; k++) {
        WalkGroup &g = grp[k];
        if (!g.inflight) continue;
        int rc = collect(g);
        if (rc) return rc;
        if (g.unfinished > 0) {
          rc = issue(g);
          if (rc) return rc;
          any = true;
        }
      }
    }
    return DSIM_OK;
  };
  const int loop_rc = drive();
  /* whatever happened, the walker order is put back before anything is reported (getters must never see Morton
   * order).  Every sort flips a group between the two sets of state buffers; groups that ended in the other set
   * than group 0 are copied across (their own region of that set is free), then ONE scatter over the whole shard
   * takes every walker back to its original index - the groups' orig[] hold indices of the whole shard. */
  const std::string keep = g_err;
  int rc = DSIM_OK;
  for (int k = 0; k < n_groups; k++) (void)hipStreamSynchronize(grp[k].v.stream);
  if (grp[0].permuted) {
    const dsim_sim &v0 = grp[0].v;
    for (int k = 1; k < n_groups && !rc; k++) {
      const dsim_sim &v = grp[k].v;
      if (v.cur == v0.cur) continue;
      const int64_t off = grp[k].off, cnt = v.n;
      auto move = [&](void *dst, const void *src, size_t bytes) -> hipError_t { return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, h->stream); };
      hipError_t e = move(h->x[v0.cur] + off, v.x[v.cur], cnt * sizeof(double));
      if (e == hipSuccess) e = move(h->y[v0.cur] + off, v.y[v.cur], cnt * sizeof(double));
      if (e == hipSuccess) e = move(h->z[v0.cur] + off, v.z[v.cur], cnt * sizeof(double));
      if (e == hipSuccess) e = move(h->s0[v0.cur] + off, v.s0[v.cur], cnt * sizeof(uint64_t));
      if (e == hipSuccess) e = move(h->s1[v0.cur] + off, v.s1[v.cur], cnt * sizeof(uint64_t));
      for (int m = 0; m < h->M && e == hipSuccess; m++)
        e = move(v0.phase + (int64_t)m * h->phase_stride + off, v.phase + (int64_t)m * h->phase_stride, cnt * sizeof(double));
      if (e == hipSuccess) e = move(v0.exc + off, v.exc, cnt);
      if (e == hipSuccess) e = move(h->orig[v0.orig_cur] + off, v.orig[v.orig_cur], cnt * sizeof(int32_t));
      if (e != hipSuccess) rc = fail(DSIM_ERR_HIP, "merging the walker groups failed: %s", hipGetErrorString(e));
    }
    if (!rc) {
      h->cur = v0.cur;
      h->orig_cur = v0.orig_cur;
      h->phase = v0.phase, h->phase_alt = v0.phase_alt;
      h->exc = v0.exc, h->exc_alt = v0.exc_alt;
      rc = restore_walker_order(h);
    }
  }
  for (int k = 0; k < n_groups; k++) (void)hipStreamSynchronize(grp[k].v.stream);
  if (loop_rc) {
    g_err = keep;
    return loop_rc;
  }
  if (rc) return rc;
  HIP_TRY(hipEventRecord(h->ev1, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->last_ms = ms;
  /* walk-kernel time: the sum over the launches (what a kernel trace adds up) and the time during which at
   * least one of them was running (what the throughput of the kernel follows from when two groups overlap) */
  std::vector<std::pair<float, float>> spans;
  for (int k = 0; k < n_groups; k++)
    for (int i = 0; i + 1 < *grp[k].ev_used; i += 2) {
      float a = 0, b = 0;
      HIP_TRY(hipEventElapsedTime(&a, h->ev0, (*grp[k].ev)[i]));
      HIP_TRY(hipEventElapsedTime(&b, h->ev0, (*grp[k].ev)[i + 1]));
      h->last_walk_ms += b - a;
      spans.emplace_back(a, b);
    }
  std::sort(spans.begin(), spans.end());
  float busy = 0, hi = -1e30f;
  for (auto &sp : spans) {
    if (sp.second <= hi) continue;
    busy += sp.second - std::max(sp.first, hi);
    hi = sp.second;
  }
  h->last_walk_busy_ms = busy;
  return DSIM_OK;
}

static int run_impl(dsim_sim *h, int geom, WalkParams p, int32_t t_begin, int32_t t_end) {
  if (!h) return fail(DSIM_ERR_INVALID, "null handle");
  if (t_begin < 0 || t_end > h->T || t_begin > t_end) return fail(DSIM_ERR_INVALID, "time range [%d, %d) outside [0, %d)", t_begin, t_end, h->T);
  HIP_TRY(hipSetDevice(h->device));
  h->last_ms = 0;
  h->last_launches = 0;
  h->last_walk_ms = 0;
  h->last_walk_busy_ms = 0;
  h->walk_ev_used = 0;
  for (int k = 0; k < DSIM_MAX_GROUPS - 1; k++) h->grp_walk_ev_used[k] = 0;
  if (t_begin == t_end) return DSIM_OK;
  const int mb_all = pick_mb(h->M, h->meas_per_pass);
  const int passes_all = (h->M + mb_all - 1) / mb_all;
  /* Mesh, cooperative traversal, more than two measurements: the walk kernel carries no phase
   * accumulators (it stays at 128 VGPRs / 4 waves per SIMD), records the positions of every
   * step and phase_from_traj_kernel accumulates the phases from them - same operations, same
   * order, no replay of the walk for M > 32. */
  const bool split_phase = geom == GEOM_MESH && h->mesh_traversal != 1 && h->mesh.coop_ok && (mb_all > COOP4_MAX_MB || h->recording);
  /* Everything else with more than 32 measurements: the first 32 in the walk kernel, which also
   * records the positions; the other passes accumulate from the positions (128 FMAs per walker
   * and step) instead of replaying RNG and geometry (480). */
  if (h->recording && h->grad_src != 0) return fail(DSIM_ERR_STATE, "recording needs the LDS gradient path (gradient_source 1)");
  const bool record_extra = !split_phase && (h->recording || (passes_all > 1 && mb_all == 32)) && h->grad_src == 0;
  const bool record = split_phase || record_extra;
  /* a recording run carries up to 32 measurements in the 32-accumulator kernel (the only one with a
   * recording variant) */
  const int mb = split_phase ? 1 : (record_extra ? 32 : mb_all);
  const int passes = record ? 1 : passes_all;
  /* measurements the walk kernel itself accumulates in a recording run: none when the phases come
   * from the recorded positions, otherwise the first min(32, M) whatever meas_per_pass says (the
   * 32-accumulator kernel is the only one with a recording variant) */
  const int walk_meas = split_phase ? 0 : (record_extra ? std::min(32, h->M) : 0);
  p.n = h->n;
  p.phase_stride = h->phase_stride;
  p.g4 = h->g4;
  p.M = h->M;
  p.step_l = h->step_l;
  p.gdt = DSIM_GAMMA * h->dt;
  /* LDS budget: 1 KB log table + subvoxel boundaries + <= 16 KB of gradient tile */
  int tile = h->tile_steps > 0 ? h->tile_steps : std::max(1, 16384 / (mb * 32));
  tile = std::min(tile, t_end - t_begin);
  tile = std::min(tile, std::max(1, (int)((60 * 1024) / (mb * 32))));
  p.tile_steps = tile;
  /* the cooperative traversal needs the reference's own uniform subvoxel grid (DevMesh::coop_ok) */
  const int coop = geom == GEOM_MESH && h->mesh_traversal != 1 && h->mesh.coop_ok;
  const int block = h->block_size;
  int n_bounds = geom == GEOM_MESH ? h->mesh.nx + h->mesh.ny + h->mesh.nz + 3 : 0;
  /* (the whole LDS of a workgroup: the static tables of the random step + the dynamic part asked for at the launch) */
  const size_t lds_fixed = DS_STATIC_LDS_BYTES + (size_t)((n_bounds + 1) & ~1) * sizeof(double) +
                           (coop ? (size_t)(block / 64) * sizeof(CoopWave) : 0);
  const bool coop4 = coop && mb <= COOP4_MAX_MB;
  if (coop4) {
    tile = t_end - t_begin; /* gradient from global memory: one "tile" per launch, no LDS */
  } else if (coop) {
    /* three workgroups per CU (3 x 51 KB of the 160 KB LDS, allocated in 1280-byte granules): the
     * gradient tile gets what the wave queues leave; when that is less than 16 time steps, two
     * workgroups per CU (2 x 80 KB) and a tile of at most 4 KB */
    const size_t third = 41 * 1280, half = 80 * 1024, per_step = (size_t)mb * 32;
    size_t room = lds_fixed < third ? third - lds_fixed : 0;
    if (room < 16 * per_step) room = std::min<size_t>(lds_fixed < half ? half - lds_fixed : 0, 4096);
    tile = std::min(tile, std::max(1, (int)(room / per_step)));
  }
  p.tile_steps = tile;
  size_t lds = lds_fixed + (coop4 ? 0 : (size_t)tile * mb * 4 * sizeof(double));
  if (lds > (coop ? 160u : 64u) * 1024u) return fail(DSIM_ERR_INVALID, "LDS request %zu bytes too large (n_sv too fine?)", lds);
  lds -= DS_STATIC_LDS_BYTES; /* what the launches ask for: the dynamic part */
  int grid = grid_for(h->n, block);
  /* runs of 8 walker groups (2048 walkers) per XCD: measured on C5b with the counting sort's 64^3 bins, 1.25e6 walkers:
   * runs of 1 / 2 / 4 / 8 / 16 / 32 / 64 / 128 groups give 1.085 / 1.13 / 1.18 / 1.20 / 1.18 / 1.14 / 1.10 / 0.91e10
   * (round 2 used 32 with the full Morton sort); 1e7 walkers: 1.235 ... 1.249 ... 1.225e10 */
  p.xcd_run_log2 = 3;
  if (const char *e = getenv("DSIM_XCD_RUN")) { /* tuning runs */
    const int v = atoi(e);
    if (v >= 0 && v <= 10) p.xcd_run_log2 = v;
  }
  if (geom == GEOM_MESH) {
    const int grid_unit = 8 << p.xcd_run_log2;
    grid = (grid + grid_unit - 1) / grid_unit * grid_unit; /* XCD-aware group mapping in the kernel */
  }
  /* mesh: re-sort every `resort` steps (auto: the time a walker needs to diffuse ~half a subvoxel) */
  int resort = 0;
  /* [r4] also while recording for the caller: the kernels then record through the slot -> walker map (rec_col) */
  if (geom == GEOM_MESH && h->resort_steps >= 0 && h->n >= 4096 && h->n < (int64_t)INT32_MAX) {
    resort = h->resort_steps;
    if (resort == 0) {
      double hmin = 1e300;
      for (int k = 0; k < 3; k++)
        if (h->mesh.inv_h[k] > 0) hmin = std::min(hmin, 1.0 / h->mesh.inv_h[k]);
      double r = h->step_l > 0 ? hmin / h->step_l : 1e3;
      /* measured (C4, C5): a sort costs about as much as 10 time steps, the lanes of a wavefront stay
       * in neighbouring subvoxels for several hundred steps, and every launch boundary idles the
       * lanes that are ahead of their neighbours: 512..1024 steps, longer where walkers move slowly */
      resort = (int)std::min(1024.0, std::max(512.0, 8.0 * r * r));
      /* A launch that is only a few rounds of the GPU's 4096 wavefront slots long pays for long workgroups twice:
       * its last round is partly empty for a whole workgroup duration, and the short kernels between two launches
       * of a group (keys, radix sort, gathers, counters: ~30 dispatches) each wait for a workgroup of the other
       * group's launch to end before they find a free slot.  Shorter launches shorten both waits; with many rounds
       * per launch the sorts themselves weigh more.  Measured on C5b with two walker groups (resort 1024 -> 256):
       * +3.6 % at 1.25e6 walkers, +0.9 % on C4 at 1e6; at 1e7 walkers 1024 -> 512 costs 0.8 %. */
      const double rounds = (double)h->n / (double)walker_groups(h) / 64.0 / 4096.0;
      /* [r4] With the fine grid of overlapping cells (2.5 times the records of round 3's, every walker reading the one
       * list around it) a wavefront whose walkers have drifted apart pays more, and sorting more often pays: C5b at 1e7
       * walkers 1.47 / 1.53 / 1.53 / 1.50 / 1.43e10 for 1024 / 512 / 256 / 128 / 64 steps, at 1.25e6 walkers
       * 1.285 / 1.32 / 1.34 / 1.33 / 1.27e10 for 512 / 256 / 128 / 64 / 32, C4 at 1e6 walkers 1.13 / 1.15 / 1.175e10 for
       * 512 / 256 / 128 (profiles/history/r04_sweeps_resort_interval.txt) */
      /* [r5] with 32-byte records in items of five a launch costs less per step and the sorts weigh more: 1e7 walkers
       * 1.679 / 1.733 / 1.742 / 1.748e10 for 128 / 256 / 384 / 512 steps, 1.25e6 walkers 1.616 / 1.645 / 1.642 / 1.639e10 for
       * 64 / 128 / 192 / 256 (profiles/history/r05_sweeps_resort_interval.txt) */
      if (coop4) resort = std::min(resort, rounds < 6.0 ? 128 : (rounds < 15.0 ? 256 : 512));
    }
    int rc = ensure_sort_buffers(h);
    if (rc) return rc;
    /* the counting sort leaves its histograms zeroed for the next time; a run that failed half-way may not have */
    for (int g = 0; g < DSIM_MAX_GROUPS; g++)
      HIP_TRY(hipMemsetAsync(h->sort_hist[g], 0, (((size_t)1 << h->sort_bits) + 1) * sizeof(uint32_t), h->stream));
  }
  h->rec_t0 = h->rec_t1 = 0;
  /* the branch further down that lets the walker groups record per range of steps and adds the phases up from that */
  const bool own_clock_recorded = coop4 && split_phase && !h->recording && resort > 0 && h->mesh_traversal != 5;
  if (record) {
    /* up to 32 GB of recorded positions (24 bytes per walker and step; MI355X: 288 GB), at most 1024 steps
     * per launch; a recording run must hold its whole range.  The walker groups' own recording is walker-major with
     * 32 bytes per walker and step (WalkParams::traj_row_steps): 43 GB, the same 143 steps at 1e7 walkers */
    const int width = own_clock_recorded ? 4 : 3;
    const int64_t fit = own_clock_recorded ? (int64_t)(43ull << 30) / (h->n * 32) : (int64_t)(32ull << 30) / (h->n * 24);
    int steps = (int)std::min<int64_t>(1024, std::max<int64_t>(16, fit));
    steps = std::min(steps, std::max(16, t_end - t_begin)); /* no more than this run needs */
    if (h->recording) {
      if (t_end - t_begin > fit) return fail(DSIM_ERR_INVALID, "recording: %d steps of %lld walkers do not fit 32 GB", t_end - t_begin, (long long)h->n);
      steps = std::max(steps, t_end - t_begin);
    }
    if (h->traj && h->traj_doubles < (size_t)steps * width * h->n) {
      HIP_TRY(hipFree(h->traj));
      h->traj = nullptr;
      h->traj_doubles = 0;
    }
    while (!h->traj) {
      /* not a recording run: shorter launches if the device cannot spare that much memory */
      if (hipMalloc(&h->traj, (size_t)steps * width * h->n * sizeof(double)) == hipSuccess) {
        h->traj_doubles = (size_t)steps * width * h->n;
        break;
      }
      h->traj = nullptr;
      (void)hipGetLastError();
      if (h->recording || steps <= 16) return fail(DSIM_ERR_HIP, "no device memory for %d steps of recorded positions", steps);
      steps = std::max(16, steps / 2);
    }
    h->traj_steps = (int)std::min<int64_t>(1 << 20, (int64_t)(h->traj_doubles / ((size_t)width * h->n))); /* steps the buffer holds in this layout */
    if (own_clock_recorded) h->traj_steps &= ~1; /* rows of an even number of steps: phase_from_rows_kernel reads two at a time */
  }
  /* clearance map: mesh_traversal 2 = decide per launch, 4 = always, 3 (and the per-lane traversal) = never */
  const MeshDev clear_map = p.mesh;
  if (geom != GEOM_MESH || !coop || h->mesh_traversal == 3) p.mesh.clear_idx = nullptr;
  const bool clear_auto = p.mesh.clear_idx != nullptr && h->mesh_traversal != 4;
  /* mesh, cooperative kernel, re-sorted walkers, nothing recorded: per-walker clocks (run_own_clock) */
  if (coop4 && resort > 0 && !record && passes == 1 && h->mesh_traversal != 5)
    return run_own_clock(h, p, clear_map, clear_auto, mb, block, lds, resort, t_begin, t_end);
  /* the same with more than two measurements: positions recorded per range of steps, phases added up from them */
  if (own_clock_recorded) {
    if (clear_auto) p.mesh.clear_idx = nullptr;
    return run_own_clock(h, p, clear_map, false, mb, block, lds, resort, t_begin, t_end, mb_all);
  }
  HIP_TRY(hipEventRecord(h->ev0, h->stream));
  bool permuted = false;
  /* the time loop; any failure inside leaves through `return`, and the walker order is put back
   * below before the error is reported (getters must never see Morton order) */
  auto time_loop = [&]() -> int {
  for (int ca = t_begin; ca < t_end;) {
    const int cb = resort > 0 ? std::min(t_end, ca + resort) : t_end;
    if (resort > 0) {
      int rc = resort_walkers(h, !permuted);
      if (rc) return rc;
      permuted = true;
    }
    if (clear_auto) {
      /* The map pays when whole wavefronts are clear (they then run at the cost of free diffusion); with a
       * minority of clear walkers scattered over the wavefronts its two look-ups per step cost more than the
       * skipped passes save (neuron, walkers inside: 21 % clear, -1.5 %).  Decide per launch from the share
       * of walkers that are clear right now - results do not depend on the choice. */
      if (!h->clear_count) HIP_TRY(hipMalloc(&h->clear_count, sizeof(unsigned long long)));
      HIP_TRY(hipMemsetAsync(h->clear_count, 0, sizeof(unsigned long long), h->stream));
      clear_count_kernel<<<grid_for(h->n, 256), 256, 0, h->stream>>>(h->x[h->cur], h->y[h->cur], h->z[h->cur], h->n, clear_map, h->clear_count);
      HIP_TRY(hipGetLastError());
      unsigned long long cnt = 0;
      HIP_TRY(hipMemcpyAsync(&cnt, h->clear_count, sizeof(cnt), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(hipStreamSynchronize(h->stream));
      h->last_clear_frac = (double)cnt / (double)h->n;
      p.mesh.clear_idx = h->last_clear_frac >= 0.4 ? clear_map.clear_idx : nullptr;
    }
   for (int ta = ca; ta < cb;) {
    const int tb = record ? std::min(cb, ta + h->traj_steps) : cb; /* recording: traj_steps covers the whole range */
    const int src = h->cur, dst = passes > 1 ? 1 - h->cur : h->cur;
    p.x_in = h->x[src];
    p.y_in = h->y[src];
    p.z_in = h->z[src];
    p.s0_in = h->s0[src];
    p.s1_in = h->s1[src];
    p.x_out = h->x[dst];
    p.y_out = h->y[dst];
    p.z_out = h->z[dst];
    p.s0_out = h->s0[dst];
    p.s1_out = h->s1[dst];
    p.phase = h->phase;
    p.exc = h->exc;
    p.t_begin = ta;
    p.t_end = tb;
    /* a recording for the caller holds the whole range of the call, in walker order: launch [ta, tb) writes its part,
     * every slot into the column of the walker it holds */
    double *traj_launch = record ? h->traj + (h->recording ? (size_t)(ta - t_begin) * 3 * (size_t)h->n : 0) : nullptr;
    const int32_t *rec_col = (h->recording && permuted) ? h->orig[h->orig_cur] : nullptr;
    p.traj = traj_launch;
    p.rec_col = rec_col;
    p.tcur = nullptr;
    p.iter_budget = INT_MAX;
    for (int pass = 0; pass < passes; pass++) {
      p.m0 = pass * mb;
      p.mc = split_phase ? 0 : std::min(mb, h->M - p.m0);
      p.write_state = pass == passes - 1;
      hipError_t e;
      if (h->walk_ev_used + 2 > (int)h->walk_ev.size()) {
        hipEvent_t a = nullptr, b = nullptr;
        HIP_TRY(hipEventCreate(&a));
        h->walk_ev.push_back(a);
        HIP_TRY(hipEventCreate(&b));
        h->walk_ev.push_back(b);
      }
      HIP_TRY(hipEventRecord(h->walk_ev[h->walk_ev_used], h->stream));
      switch (geom) {
        case GEOM_FREE: e = launch_mb<GEOM_FREE>(mb, p, grid, block, lds, h->stream, h->grad_src, coop); break;
        case GEOM_SPHERE: e = launch_mb<GEOM_SPHERE>(mb, p, grid, block, lds, h->stream, h->grad_src, coop); break;
        case GEOM_CYLINDER: e = launch_mb<GEOM_CYLINDER>(mb, p, grid, block, lds, h->stream, h->grad_src, coop); break;
        case GEOM_ELLIPSOID: e = launch_mb<GEOM_ELLIPSOID>(mb, p, grid, block, lds, h->stream, h->grad_src, coop); break;
        default: e = launch_mb<GEOM_MESH>(mb, p, grid, block, lds, h->stream, h->grad_src, coop); break;
      }
      if (e != hipSuccess) return fail(DSIM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
      HIP_TRY(hipEventRecord(h->walk_ev[h->walk_ev_used + 1], h->stream));
      h->walk_ev_used += 2;
      h->last_launches++;
    }
    h->cur = dst;
    if (record) {
      const int tile_p = std::max(1, 16384 / (mb_all * 32));
      const size_t lds_p = 0; /* one stream: nothing to keep out of the way of (the samples come through L1 and DPP row broadcasts) */
      const int grid_p = grid_for(h->n, 256);
      for (int m0 = walk_meas; m0 < h->M; m0 += mb_all) {
        const int mc = std::min(mb_all, h->M - m0);
#define DSIM_PHASE_LAUNCH(MBV)                                                                                             \
  phase_from_traj_kernel<MBV><<<grid_p, 256, lds_p, h->stream>>>(traj_launch, h->n, tb - ta, tile_p, h->g4, h->M, m0, mc, ta, \
                                                                 p.gdt, h->phase, h->phase_stride, rec_col)
        switch (mb_all) {
          case 1: DSIM_PHASE_LAUNCH(1); break;
          case 2: DSIM_PHASE_LAUNCH(2); break;
          case 4: DSIM_PHASE_LAUNCH(4); break;
          case 8: DSIM_PHASE_LAUNCH(8); break;
          case 16: DSIM_PHASE_LAUNCH(16); break;
          default: DSIM_PHASE_LAUNCH(32); break;
        }
#undef DSIM_PHASE_LAUNCH
        HIP_TRY(hipGetLastError());
      }
    }
    ta = tb;
   }
    ca = cb;
  }
  return DSIM_OK;
  };
  const int loop_rc = time_loop();
  if (permuted) {
    const std::string keep = g_err;
    int rc = restore_walker_order(h);
    if (loop_rc) {
      (void)hipStreamSynchronize(h->stream);
      g_err = keep;
      return loop_rc;
    }
    if (rc) return rc;
  }
  if (loop_rc) return loop_rc;
  if (h->recording) {
    h->rec_t0 = t_begin;
    h->rec_t1 = t_end;
  }
  HIP_TRY(hipEventRecord(h->ev1, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->last_ms = ms;
  for (int i = 0; i + 1 < h->walk_ev_used; i += 2) {
    float w = 0;
    HIP_TRY(hipEventElapsedTime(&w, h->walk_ev[i], h->walk_ev[i + 1]));
    h->last_walk_ms += w;
  }
  h->last_walk_busy_ms = h->last_walk_ms; /* one stream: the launches do not overlap */
  return DSIM_OK;
}

#ifdef DSIM_ABLATE
extern "C" int dsim_debug_ablate(int mode) {
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(dsim_ablate), &mode, sizeof(int)));
  return DSIM_OK;
}
#endif

#ifdef DSIM_STATS
/* development build only: read and clear the traversal counters */
extern "C" int dsim_debug_stats(unsigned long long *out32) {
  HIP_TRY(hipMemcpyFromSymbol(out32, HIP_SYMBOL(dsim_stats), 32 * sizeof(unsigned long long)));
  unsigned long long zero[32] = {0};
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(dsim_stats), zero, sizeof(zero)));
  return DSIM_OK;
}
#endif

#ifdef DSIM_TIMERS
/* development build only: read and clear the phase timers of the cooperative traversal */
extern "C" int dsim_debug_timers(unsigned long long *out16) {
  HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(dsim_timers), 16 * sizeof(unsigned long long)));
  unsigned long long zero[16] = {0};
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(dsim_timers), zero, sizeof(zero)));
  return DSIM_OK;
}
#endif

extern "C" int dsim_set_recording(dsim_handle h, int32_t enable) {
  if (IS_MULTI(h)) h->keep_recording = enable != 0;
  MULTI_FORWARD(h, dsim_set_recording(s_, enable));
  if (!h) return fail(DSIM_ERR_INVALID, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  h->recording = enable != 0;
  h->rec_t0 = h->rec_t1 = 0;
  if (!h->recording && h->traj) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipFree(h->traj));
    h->traj = nullptr;
    h->traj_steps = 0;
    h->traj_doubles = 0;
  }
  return DSIM_OK;
}

/* ---- recorded positions -> host, double buffered ------------------------------------------------
 * Steps [t_begin, t_end) of the last recorded run are handed to `sink` block by block.  Blocks go
 * through two pinned staging buffers on a dedicated copy stream: while the host consumes block k
 * (copies it out, or formats it as text), the DMA engine is already filling the other buffer with
 * block k+1.  sink(first_step, n_steps, data) sees the device layout [step][3][n]. */
static int fetch_recorded(dsim_sim *h, int32_t t_begin, int32_t t_end, const std::function<int(int, int, const double *)> &sink) {
  const size_t row = (size_t)3 * h->n * sizeof(double);
  const int per = (int)std::max<size_t>(1, std::min<size_t>((size_t)(t_end - t_begin), ((size_t)32 << 20) / row));
  const size_t need = (size_t)per * row;
  if (!h->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
  for (int b = 0; b < 2; b++)
    if (!h->pin_ev[b]) HIP_TRY(hipEventCreateWithFlags(&h->pin_ev[b], hipEventDisableTiming));
  if (h->pin_bytes < need) {
    for (int b = 0; b < 2; b++) {
      if (h->pin[b]) (void)hipHostFree(h->pin[b]);
      h->pin[b] = nullptr;
    }
    h->pin_bytes = 0;
    for (int b = 0; b < 2; b++) HIP_TRY(hipHostMalloc(&h->pin[b], need, hipHostMallocDefault));
    h->pin_bytes = need;
  }
  auto issue = [&](int t0, int b) -> int {
    const int cnt = std::min(per, t_end - t0);
    HIP_TRY(hipMemcpyAsync(h->pin[b], h->traj + (size_t)(t0 - h->rec_t0) * 3 * h->n, (size_t)cnt * row, hipMemcpyDeviceToHost, h->copy_stream));
    HIP_TRY(hipEventRecord(h->pin_ev[b], h->copy_stream));
    return DSIM_OK;
  };
  int rc = issue(t_begin, 0);
  if (rc) return rc;
  int b = 0;
  for (int t0 = t_begin; t0 < t_end; t0 += per, b ^= 1) {
    if (t0 + per < t_end) {
      rc = issue(t0 + per, b ^ 1); /* the other buffer: its previous block has been consumed already */
      if (rc) return rc;
    }
    HIP_TRY(hipEventSynchronize(h->pin_ev[b]));
    rc = sink(t0, std::min(per, t_end - t0), (const double *)h->pin[b]);
    if (rc) {
      (void)hipStreamSynchronize(h->copy_stream);
      return rc;
    }
  }
  return DSIM_OK;
}

static int check_recorded_range(dsim_sim *h, const char *who, int32_t t_begin, int32_t t_end) {
  if (!h->recording || !h->traj) return fail(DSIM_ERR_STATE, "%s: recording is off", who);
  if (t_begin < h->rec_t0 || t_end > h->rec_t1 || t_begin > t_end)
    return fail(DSIM_ERR_INVALID, "%s: [%d, %d) outside the recorded range [%d, %d)", who, t_begin, t_end, h->rec_t0, h->rec_t1);
  return DSIM_OK;
}

/* steps [t_begin, t_end) of every shard's recording into one component-major block [step][3][n] of the whole handle */
static int multi_get_recorded(dsim_sim *h, int32_t t_begin, int32_t t_end, double *out) {
  return multi_each(h, [&](int k, dsim_sim *sh, int64_t off) -> int {
    std::vector<double> part((size_t)(t_end - t_begin) * 3 * (size_t)sh->n);
    const int rc = dsim_get_recorded(sh, t_begin, t_end, part.data());
    if (rc) return rc;
    for (size_t r = 0; r < (size_t)(t_end - t_begin) * 3; r++) shard_rows_out(h, k, off, out + r * (size_t)h->n, 8, part.data() + r * (size_t)sh->n, sh->n);
    return DSIM_OK;
  });
}

extern "C" int dsim_get_recorded(dsim_handle h, int32_t t_begin, int32_t t_end, double *out) {
  if (IS_MULTI(h)) {
    if (!out || t_begin > t_end) return fail(DSIM_ERR_INVALID, "dsim_get_recorded: bad argument");
    return t_begin == t_end ? DSIM_OK : multi_get_recorded(h, t_begin, t_end, out);
  }
  if (!h || !out) return fail(DSIM_ERR_INVALID, "dsim_get_recorded: null argument");
  int rc = check_recorded_range(h, "dsim_get_recorded", t_begin, t_end);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(h->device));
  if (t_begin == t_end) return DSIM_OK;
  const size_t row_doubles = (size_t)3 * h->n;
  return fetch_recorded(h, t_begin, t_end, [&](int t0, int cnt, const double *src) {
    memcpy(out + (size_t)(t0 - t_begin) * row_doubles, src, (size_t)cnt * row_doubles * sizeof(double));
    return DSIM_OK;
  });
}

/* ---- trajectories file (simulations.py:1043-1048) ------------------------------------------------
 * The reference writes `str(value) + " "` for every coordinate and "\n" per time step.  str() of a
 * float is the shortest decimal string that round-trips, laid out by CPython's float_repr_style
 * "short" rules: scientific notation iff the decimal exponent is < -4 or >= 16, at least two exponent
 * digits, ".0" appended to integral values in fixed notation.  std::to_chars gives the same shortest
 * digits; the layout is redone here.  Returns the number of characters written (<= 24). */
static size_t format_py_float(double x, char *out) {
  if (x != x) {
    memcpy(out, "nan", 3);
    return 3;
  }
  char *p = out;
  if (std::signbit(x)) {
    *p++ = '-';
    x = -x;
  }
  if (x == std::numeric_limits<double>::infinity()) {
    memcpy(p, "inf", 3);
    return (size_t)(p + 3 - out);
  }
  if (x == 0) {
    memcpy(p, "0.0", 3);
    return (size_t)(p + 3 - out);
  }
  char tmp[40], digits[24];
  const auto r = std::to_chars(tmp, tmp + sizeof(tmp), x, std::chars_format::scientific); /* d[.ddd]e[+-]XX, shortest */
  int nd = 0;
  const char *q = tmp;
  digits[nd++] = *q++;
  if (*q == '.')
    for (q++; *q != 'e'; q++) digits[nd++] = *q;
  q++;
  const bool eneg = *q++ == '-';
  int e = 0;
  while (q < r.ptr) e = e * 10 + (*q++ - '0');
  const int decpt = (eneg ? -e : e) + 1; /* value = 0.d1d2... x 10^decpt */
  if (decpt <= -4 || decpt > 16) {
    *p++ = digits[0];
    if (nd > 1) {
      *p++ = '.';
      memcpy(p, digits + 1, (size_t)nd - 1);
      p += nd - 1;
    }
    *p++ = 'e';
    int ex = decpt - 1;
    *p++ = ex < 0 ? '-' : '+';
    if (ex < 0) ex = -ex;
    if (ex >= 100) {
      *p++ = (char)('0' + ex / 100);
      ex %= 100;
    }
    *p++ = (char)('0' + ex / 10);
    *p++ = (char)('0' + ex % 10);
  } else if (decpt <= 0) {
    *p++ = '0';
    *p++ = '.';
    for (int i = 0; i < -decpt; i++) *p++ = '0';
    memcpy(p, digits, (size_t)nd);
    p += nd;
  } else if (decpt >= nd) {
    memcpy(p, digits, (size_t)nd);
    p += nd;
    for (int i = 0; i < decpt - nd; i++) *p++ = '0';
    *p++ = '.';
    *p++ = '0';
  } else {
    memcpy(p, digits, (size_t)decpt);
    p += decpt;
    *p++ = '.';
    memcpy(p, digits + decpt, (size_t)(nd - decpt));
    p += nd - decpt;
  }
  return (size_t)(p - out);
}

/* One line of the file = the n walkers of one time step, "x y z x y z ... \n"; x/y/z point at the
 * three coordinates of walker 0 and advance by `stride` doubles per walker (1 for the device's
 * component-major layout, 3 for (n,3) row-major host arrays).  Walker ranges are formatted by a few
 * host threads into private buffers and written in order. */
static int write_traj_lines(FILE *f, int64_t n, int n_lines, const double *base, size_t line_stride, const double *x_off,
                            const double *y_off, const double *z_off, size_t stride) {
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const int64_t values = n * n_lines;
  const int n_thr = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<unsigned>(hw, 32u), values / 4096));
  /* the block of lines is cut into n_thr contiguous pieces of (line, walker) pairs; every piece is formatted into a
   * buffer of its own by one thread - one spawn per block of lines, not per line - and the buffers are written in order */
  std::vector<std::vector<char>> bufs((size_t)n_thr);
  auto work = [&](int k) {
    const int64_t a = values * k / n_thr, b = values * (k + 1) / n_thr;
    std::vector<char> &buf = bufs[(size_t)k];
    buf.resize((size_t)(b - a) * 3 * 25 + (size_t)n_lines + 2);
    char *p = buf.data();
    int64_t line = a / n, i = a % n;
    for (int64_t v = a; v < b; v++) {
      const size_t off = (size_t)line * line_stride + (size_t)i * stride;
      p += format_py_float(x_off[off], p);
      *p++ = ' ';
      p += format_py_float(y_off[off], p);
      *p++ = ' ';
      p += format_py_float(z_off[off], p);
      *p++ = ' ';
      if (++i == n) {
        *p++ = '\n';
        i = 0;
        line++;
      }
    }
    buf.resize((size_t)(p - buf.data()));
  };
  if (n_thr == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int k = 0; k < n_thr; k++) th.emplace_back(work, k);
    for (auto &t : th) t.join();
  }
  for (int k = 0; k < n_thr; k++)
    if (fwrite(bufs[(size_t)k].data(), 1, bufs[(size_t)k].size(), f) != bufs[(size_t)k].size()) return fail(DSIM_ERR_STATE, "trajectories file: write failed");
  (void)base;
  return DSIM_OK;
}

extern "C" int dsim_write_traj_line(const char *path, const double *positions_aos, int64_t n_walkers, int32_t append) {
  if (!path || !positions_aos || n_walkers <= 0) return fail(DSIM_ERR_INVALID, "dsim_write_traj_line: bad argument");
  FILE *f = fopen(path, append ? "ab" : "wb");
  if (!f) return fail(DSIM_ERR_STATE, "cannot open %s", path);
  int rc = write_traj_lines(f, n_walkers, 1, positions_aos, 0, positions_aos, positions_aos + 1, positions_aos + 2, 3);
  if (fclose(f) != 0 && rc == DSIM_OK) rc = fail(DSIM_ERR_STATE, "trajectories file: close failed");
  return rc;
}

extern "C" int dsim_write_traj(dsim_handle h, const char *path, int32_t t_begin, int32_t t_end, int32_t append) {
  if (IS_MULTI(h)) {
    /* a line of the file holds every walker of the handle: blocks of steps (<= 64 MB) are collected from the shards */
    if (!path || t_begin > t_end) return fail(DSIM_ERR_INVALID, "dsim_write_traj: bad argument");
    FILE *f = fopen(path, append ? "ab" : "wb");
    if (!f) return fail(DSIM_ERR_STATE, "cannot open %s", path);
    const size_t row = (size_t)3 * (size_t)h->n;
    const int per = (int)std::max<size_t>(1, ((size_t)8 << 20) / row);
    std::vector<double> buf;
    int rc = DSIM_OK;
    for (int t0 = t_begin; t0 < t_end && rc == DSIM_OK; t0 += per) {
      const int cnt = std::min(per, t_end - t0);
      buf.resize((size_t)cnt * row);
      rc = multi_get_recorded(h, t0, t0 + cnt, buf.data());
      if (rc == DSIM_OK) rc = write_traj_lines(f, h->n, cnt, buf.data(), row, buf.data(), buf.data() + h->n, buf.data() + 2 * h->n, 1);
    }
    if (fclose(f) != 0 && rc == DSIM_OK) rc = fail(DSIM_ERR_STATE, "trajectories file: close failed");
    return rc;
  }
  if (!h || !path) return fail(DSIM_ERR_INVALID, "dsim_write_traj: null argument");
  int rc = check_recorded_range(h, "dsim_write_traj", t_begin, t_end);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(h->device));
  FILE *f = fopen(path, append ? "ab" : "wb");
  if (!f) return fail(DSIM_ERR_STATE, "cannot open %s", path);
  const int64_t n = h->n;
  if (t_begin < t_end)
    rc = fetch_recorded(h, t_begin, t_end, [&](int, int cnt, const double *src) {
      return write_traj_lines(f, n, cnt, src, (size_t)3 * n, src, src + n, src + 2 * n, 1);
    });
  if (fclose(f) != 0 && rc == DSIM_OK) rc = fail(DSIM_ERR_STATE, "trajectories file: close failed");
  return rc;
}

extern "C" int dsim_last_run_ms(dsim_handle h, double *ms, int32_t *n_launches) {
  if (IS_MULTI(h)) { /* the shards run concurrently: the slowest one's time, everybody's launches */
    double worst = 0;
    int launches = 0;
    for (dsim_sim *sh : h->shards) worst = std::max(worst, sh->last_ms), launches += sh->last_launches;
    if (ms) *ms = worst;
    if (n_launches) *n_launches = launches;
    return DSIM_OK;
  }
  if (!h) return fail(DSIM_ERR_INVALID, "null handle");
  if (ms) *ms = h->last_ms;
  if (n_launches) *n_launches = h->last_launches;
  return DSIM_OK;
}

extern "C" int dsim_last_walk_ms(dsim_handle h, double *ms, int32_t *n_launches) {
  if (IS_MULTI(h)) { /* the shards run concurrently: the slowest one's time (like dsim_last_run_ms), everybody's launches */
    double worst = 0;
    int launches = 0;
    for (dsim_sim *sh : h->shards) {
      double q = 0;
      int32_t nl = 0;
      (void)dsim_last_walk_ms(sh, &q, &nl);
      worst = std::max(worst, q), launches += nl;
    }
    if (ms) *ms = worst;
    if (n_launches) *n_launches = launches;
    return DSIM_OK;
  }
  if (!h) return fail(DSIM_ERR_INVALID, "null handle");
  if (ms) *ms = h->last_walk_ms;
  if (n_launches) {
    int used = h->walk_ev_used;
    for (int k = 0; k < DSIM_MAX_GROUPS - 1; k++) used += h->grp_walk_ev_used[k];
    *n_launches = used / 2;
  }
  return DSIM_OK;
}

extern "C" int dsim_last_walk_busy_ms(dsim_handle h, double *ms, int32_t *n_streams) {
  if (IS_MULTI(h)) {
    double worst = 0;
    int streams = 0;
    for (dsim_sim *sh : h->shards) {
      double q = 0;
      int32_t ns = 0;
      (void)dsim_last_walk_busy_ms(sh, &q, &ns);
      worst = std::max(worst, q), streams += ns;
    }
    if (ms) *ms = worst;
    if (n_streams) *n_streams = streams;
    return DSIM_OK;
  }
  if (!h) return fail(DSIM_ERR_INVALID, "null handle");
  if (ms) *ms = h->last_walk_busy_ms;
  if (n_streams) *n_streams = h->grp_walk_ev_used[0] > 0 ? h->last_groups : 1;
  return DSIM_OK;
}

static int check_iter(int64_t max_iter, double eps) {
  if (max_iter < 1) return fail(DSIM_ERR_INVALID, "max_iter must be >= 1");
  if (!(eps == eps)) return fail(DSIM_ERR_INVALID, "epsilon is NaN");
  return DSIM_OK;
}

extern "C" int dsim_run_free(dsim_handle h, int32_t t_begin, int32_t t_end) {
  if (IS_MULTI(h)) h->has_run = true;
  MULTI_FORWARD(h, dsim_run_free(s_, t_begin, t_end));
  WalkParams p = {};
  p.max_iter = 1;
  return run_impl(h, GEOM_FREE, p, t_begin, t_end);
}

extern "C" int dsim_run_sphere(dsim_handle h, double radius, int64_t max_iter, double epsilon, int32_t t_begin, int32_t t_end) {
  if (IS_MULTI(h)) h->has_run = true;
  MULTI_FORWARD(h, dsim_run_sphere(s_, radius, max_iter, epsilon, t_begin, t_end));
  if (!(radius > 0)) return fail(DSIM_ERR_INVALID, "radius must be > 0");
  int rc = check_iter(max_iter, epsilon);
  if (rc) return rc;
  WalkParams p = {};
  p.radius = radius;
  p.max_iter = max_iter;
  p.eps = epsilon;
  return run_impl(h, GEOM_SPHERE, p, t_begin, t_end);
}

extern "C" int dsim_run_cylinder(dsim_handle h, double radius, const double R[9], const double R_inv[9], int64_t max_iter,
                                 double epsilon, int32_t t_begin, int32_t t_end) {
  if (IS_MULTI(h)) h->has_run = true;
  MULTI_FORWARD(h, dsim_run_cylinder(s_, radius, R, R_inv, max_iter, epsilon, t_begin, t_end));
  if (!(radius > 0) || !R || !R_inv) return fail(DSIM_ERR_INVALID, "bad cylinder arguments");
  int rc = check_iter(max_iter, epsilon);
  if (rc) return rc;
  WalkParams p = {};
  p.radius = radius;
  memcpy(p.R, R, sizeof(p.R));
  memcpy(p.Rinv, R_inv, sizeof(p.Rinv));
  p.max_iter = max_iter;
  p.eps = epsilon;
  return run_impl(h, GEOM_CYLINDER, p, t_begin, t_end);
}

extern "C" int dsim_run_ellipsoid(dsim_handle h, const double semiaxes[3], const double R[9], const double R_inv[9],
                                  int64_t max_iter, double epsilon, int32_t t_begin, int32_t t_end) {
  if (IS_MULTI(h)) h->has_run = true;
  MULTI_FORWARD(h, dsim_run_ellipsoid(s_, semiaxes, R, R_inv, max_iter, epsilon, t_begin, t_end));
  if (!semiaxes || !R || !R_inv) return fail(DSIM_ERR_INVALID, "bad ellipsoid arguments");
  int rc = check_iter(max_iter, epsilon);
  if (rc) return rc;
  WalkParams p = {};
  memcpy(p.ax, semiaxes, sizeof(p.ax));
  memcpy(p.R, R, sizeof(p.R));
  memcpy(p.Rinv, R_inv, sizeof(p.Rinv));
  p.max_iter = max_iter;
  p.eps = epsilon;
  return run_impl(h, GEOM_ELLIPSOID, p, t_begin, t_end);
}

extern "C" int dsim_run_mesh(dsim_handle h, int64_t max_iter, double epsilon, double perm_prob, int32_t t_begin, int32_t t_end) {
  if (IS_MULTI(h)) h->has_run = true;
  MULTI_FORWARD(h, dsim_run_mesh(s_, max_iter, epsilon, perm_prob, t_begin, t_end));
  if (!h) return fail(DSIM_ERR_INVALID, "null handle");
  if (!h->mesh.set) return fail(DSIM_ERR_STATE, "dsim_run_mesh called before dsim_set_mesh");
  int rc = check_iter(max_iter, epsilon);
  if (rc) return rc;
  WalkParams p = {};
  p.max_iter = max_iter;
  p.eps = epsilon;
  p.perm_prob = perm_prob;
  p.perm_thresh = ds_perm_threshold(perm_prob);
  p.mesh = mesh_dev(h->mesh);
  return run_impl(h, GEOM_MESH, p, t_begin, t_end);
}

/* -------------------------------------------------------------- fill_mesh --- */

/* A fill session: the device state of the _fill_mesh loop (simulations.py:505-579) for a RANGE of its threads.
 * The reference launches _cuda_fill_mesh over n_points threads again and again - every thread draws one candidate
 * per launch from its own RNG state - and appends the accepted points of a launch in thread order until n_points
 * are there.  Thread t only ever needs its own state, so the threads can be split over sessions (one per GPU):
 * each runs the same launches over its range and reports, per launch, how many of its candidates were accepted;
 * the position of a point in the global sequence follows from those counts alone (disimpy_amd/sharding.py).
 * dsim_fill_mesh is one session over all threads. */
struct dsim_fill {
  int device = 0;
  hipStream_t stream = nullptr;
  DevMesh dm;
  int64_t n_threads = 0;
  uint64_t *s0 = nullptr, *s1 = nullptr;
  uint64_t *s0_keep = nullptr, *s1_keep = nullptr; /* the states as they were before the current batch (a batch that overflows is redone) */
  int32_t *d_col_start = nullptr, *d_mstart = nullptr;
  int2 *d_col_ent = nullptr, *d_ent = nullptr, *d_mcol = nullptr, *d_ment = nullptr;
  float4 *d_ent_yz = nullptr, *d_ment_yz = nullptr;
  /* a batch of launches = three dispatches (fill_batch_kernel, fill_scan_kernel, fill_place_kernel; dsim_kernels.h) */
  int n_blocks = 0, stage_cap = 0;
  FillEntry *d_stage = nullptr;     /* [n_blocks][stage_cap] */
  int32_t *d_stage_count = nullptr; /* [n_blocks] */
  int32_t *d_cnt = nullptr;         /* [FILL_MAX_BATCH][n_blocks] */
  int64_t *d_totals = nullptr;      /* [FILL_MAX_BATCH] accepted per launch of the batch, then [1] the overflow flag (as int32) */
  int32_t *d_overflow = nullptr;
  double *d_out = nullptr;          /* accepted points of the current run, in (launch, thread) order */
  int64_t out_cap = 0;
  FillParams fp;
  size_t lds = 0;
  int64_t launches = 0;
};

extern "C" int dsim_fill_close(dsim_fill *fs) {
  if (!fs) return DSIM_OK;
  (void)hipSetDevice(fs->device);
  if (fs->stream) (void)hipStreamSynchronize(fs->stream);
  free_mesh(fs->dm);
  void *ptrs[] = {fs->s0, fs->s1, fs->s0_keep, fs->s1_keep, fs->d_col_start, fs->d_mstart, fs->d_col_ent, fs->d_ent, fs->d_mcol, fs->d_ment,
                  fs->d_ent_yz, fs->d_ment_yz, fs->d_stage, fs->d_stage_count, fs->d_cnt, fs->d_totals, fs->d_overflow, fs->d_out};
  for (void *q : ptrs)
    if (q) (void)hipFree(q);
  if (fs->stream) (void)hipStreamDestroy(fs->stream);
  delete fs;
  return DSIM_OK;
}

/* buckets per axis over the cross-section of a fill column that lists n_e entries (see fill_open_impl) */
static int fill_buckets_per_axis(int n_e, int fine) { return n_e <= 8 ? 1 : fine * (n_e <= 48 ? 4 : (n_e <= 400 ? 8 : 16)); }

static int fill_open_impl(dsim_fill *fs, int64_t thread_begin, int intra, uint64_t seed, const double *vertices, int64_t n_vertices,
                          const int64_t *faces, int64_t n_faces, const double voxel_size[3], const double *xs, const double *ys,
                          const double *zs, const int64_t n_sv[3], const int64_t *subvoxel_indices, const int64_t *triangle_indices,
                          int64_t n_refs) {
  const int64_t n = fs->n_threads;
  HIP_TRY(hipSetDevice(fs->device));
  HIP_TRY(hipStreamCreateWithFlags(&fs->stream, hipStreamNonBlocking));
  int rc = upload_mesh(fs->dm, fs->stream, vertices, n_vertices, faces, n_faces, xs, ys, zs, n_sv, subvoxel_indices, triangle_indices, n_refs, 0.0);
  if (rc) return rc;
  HIP_TRY(hipMalloc(&fs->s0, n * sizeof(uint64_t)));
  HIP_TRY(hipMalloc(&fs->s1, n * sizeof(uint64_t)));
  HIP_TRY(hipMalloc(&fs->s0_keep, n * sizeof(uint64_t)));
  HIP_TRY(hipMalloc(&fs->s1_keep, n * sizeof(uint64_t)));
  /* own RNG states with the same seed (simulations.py:524): thread i <- state i, whatever the range */
  {
    const JumpTables &jt = jump_tables();
    uint64_t *cols = nullptr;
    HIP_TRY(hipMalloc(&cols, jt.cols.size() * sizeof(HostRng)));
    hipError_t e = hipMemcpyAsync(cols, jt.cols.data(), jt.cols.size() * sizeof(HostRng), hipMemcpyHostToDevice, fs->stream);
    HostRng sd = h_seed_state(seed);
    if (e == hipSuccess) {
      rng_init_kernel<<<grid_for(n, 256), 256, 0, fs->stream>>>(sd.s0, sd.s1, (uint64_t)thread_begin, nullptr, n, cols, jt.levels, fs->s0, fs->s1);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(fs->stream);
    (void)hipFree(cols);
    if (e != hipSuccess) return fail(DSIM_ERR_HIP, "fill: RNG seeding failed: %s", hipGetErrorString(e));
  }
  /* (y, z) extent of every triangle, float32 rounded outward by three ulps: cull box of the
   * +x rays of fill_mesh_kernel */
  std::vector<float4> h_tri_yz((size_t)n_faces);
  for (int64_t f = 0; f < n_faces; f++) {
    const int64_t *fc = &faces[(size_t)f * 3];
    float lo[2], hi[2];
    for (int ax = 1; ax < 3; ax++) {
      double a = vertices[fc[0] * 3 + ax], b = vertices[fc[1] * 3 + ax], c = vertices[fc[2] * 3 + ax];
      float l = (float)std::min(a, std::min(b, c)), h = (float)std::max(a, std::max(b, c));
      const float ninf = -std::numeric_limits<float>::infinity(), pinf = std::numeric_limits<float>::infinity();
      lo[ax - 1] = std::nextafter(std::nextafter(std::nextafter(l, ninf), ninf), ninf);
      hi[ax - 1] = std::nextafter(std::nextafter(std::nextafter(h, pinf), pinf), pinf);
    }
    h_tri_yz[f] = make_float4(lo[0], lo[1], hi[0], hi[1]);
  }
  /* the fill kernel's index (see FillParams): per (y, z) column the non-empty subvoxels in ascending x, and per
   * list entry the triangle, where it was last met in the column, and its (y, z) extent */
  {
    const int nx = fs->dm.nx, ny = fs->dm.ny, nz = fs->dm.nz;
    std::vector<int32_t> col_start((size_t)ny * nz + 1, 0), last_x((size_t)std::max<int64_t>(n_faces, 1), -1), touched;
    std::vector<int2> col_ent, ent;
    std::vector<float4> ent_yz;
    std::vector<int32_t> ent_x; /* subvoxel x of every entry */
    ent.reserve((size_t)n_refs);
    ent_yz.reserve((size_t)n_refs);
    for (int y = 0; y < ny; y++)
      for (int z = 0; z < nz; z++) {
        col_start[(size_t)y * nz + z] = (int32_t)col_ent.size();
        for (int x = 0; x < nx; x++) {
          const int64_t sv = ((int64_t)x * ny + y) * nz + z;
          const int64_t lo = subvoxel_indices[sv * 2], hi = subvoxel_indices[sv * 2 + 1];
          if (hi <= lo) continue;
          if (lo < 0 || hi > n_refs) {
            return fail(DSIM_ERR_INVALID, "dsim_fill_mesh: subvoxel %lld lists entries [%lld, %lld) of %lld", (long long)sv, (long long)lo,
                        (long long)hi, (long long)n_refs);
          }
          col_ent.push_back(make_int2(x, (int)ent.size()));
          for (int64_t i = lo; i < hi; i++) {
            const int64_t ti = triangle_indices[i];
            if (ti < 0 || ti >= n_faces) {
              return fail(DSIM_ERR_INVALID, "dsim_fill_mesh: list entry %lld names triangle %lld of %lld", (long long)i, (long long)ti,
                          (long long)n_faces);
            }
            ent.push_back(make_int2((int)ti, last_x[ti]));
            ent_yz.push_back(h_tri_yz[ti]);
            ent_x.push_back(x);
            if (last_x[ti] < 0) touched.push_back((int32_t)ti);
            last_x[ti] = x;
          }
        }
        for (int32_t ti : touched) last_x[ti] = -1;
        touched.clear();
      }
    col_start[(size_t)ny * nz] = (int32_t)col_ent.size();
    col_ent.push_back(make_int2(nx, (int)ent.size())); /* sentinel: end of the last list */
    if (ent.empty()) {
      ent.push_back(make_int2(0, -1));
      ent_yz.push_back(make_float4(0, 0, 0, 0));
    }
    /* the bucketed copy (FillParams::mcol): G x G buckets over the cross-section of every non-empty column,
     * entries in descending x; a triangle goes into the buckets its (y, z) PROJECTION reaches, among those of its float32
     * extent (rounded outward), every bucket widened by 1e-6 widths against the kernel's own bucket arithmetic */
    std::vector<int2> mcol((size_t)ny * nz, make_int2(-1, 0)), ment;
    std::vector<int32_t> mstart;
    std::vector<float4> ment_yz;
    {
      /* how fine: 16 / 32 / 64 buckets per axis by the column's entry count (measured on the C5b neuron, 640
       * launches of 1e7 threads: 4 / 8 / 16 per axis 0.29 ms per launch, 8 / 16 / 32 0.18 ms, 16 / 32 / 64 0.15 ms,
       * twice that again 0.13 ms and 0.1 s more to build), halved while the index would pass 4M buckets */
      int fine = 4;
      for (; fine > 1; fine >>= 1) {
        int64_t total = 0;
        for (size_t c = 0; c + 1 < col_start.size(); c++) {
          if (col_start[c + 1] == col_start[c]) continue;
          const int e0 = col_ent[col_start[c]].y;
          const int e1 = (size_t)col_start[c + 1] < col_ent.size() ? col_ent[col_start[c + 1]].y : (int)ent.size();
          const int64_t G = fill_buckets_per_axis(e1 - e0, fine);
          total += G * G;
        }
        if (total <= ((int64_t)4 << 20)) break;
      }
      std::vector<std::vector<int32_t>> buckets;
      for (int y = 0; y < ny; y++)
        for (int z = 0; z < nz; z++) {
          const size_t c = (size_t)y * nz + z;
          if (col_start[c + 1] == col_start[c]) continue;
          const int e0 = col_ent[col_start[c]].y;
          const int e1 = (size_t)col_start[c + 1] < col_ent.size() ? col_ent[col_start[c + 1]].y : (int)ent.size();
          const int n_e = e1 - e0;
          const int G = fill_buckets_per_axis(n_e, fine);
          buckets.assign((size_t)G * G, {});
          const double sy = (double)G * fs->dm.inv_h[1], sz = (double)G * fs->dm.inv_h[2];
          for (int i = e1 - 1; i >= e0; i--) { /* descending x (ascending order reversed) */
            const float4 q = ent_yz[i];
            auto cell = [&](double v, double o, double sc, double d) {
              const double t = std::floor((v - o) * sc + d);
              return (int)std::min((double)(G - 1), std::max(0.0, t));
            };
            const int y0 = cell(q.x, ys[y], sy, -1e-6), y1 = cell(q.z, ys[y], sy, 1e-6);
            const int z0 = cell(q.y, zs[z], sz, -1e-6), z1 = cell(q.w, zs[z], sz, 1e-6);
            if (y0 == y1 && z0 == z1) {
              buckets[(size_t)y0 * G + z0].push_back(i);
              continue;
            }
            /* of the buckets its box reaches, those its (y, z) PROJECTION reaches: the three edge normals as
             * separating axes, every bucket widened by 1e-6 of its width (a +x ray meets the triangle only above
             * its projection, and Moller-Trumbore's own rounding at an edge is 1e-10 of that margin) */
            const int64_t *fc = &faces[(size_t)ent[i].x * 3];
            double py[3], pz[3];
            for (int k = 0; k < 3; k++) py[k] = vertices[fc[k] * 3 + 1], pz[k] = vertices[fc[k] * 3 + 2];
            const double hby = 0.5 / sy * (1 + 2e-6), hbz = 0.5 / sz * (1 + 2e-6);
            for (int a = y0; a <= y1; a++)
              for (int b = z0; b <= z1; b++) {
                const double cy = ys[y] + (a + 0.5) / sy, cz = zs[z] + (b + 0.5) / sz;
                bool apart = false;
                for (int k = 0; k < 3 && !apart; k++) {
                  const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
                  const double ny_ = -(pz[k1] - pz[k]), nz_ = py[k1] - py[k]; /* normal of edge k -> k1 */
                  const double d_edge = ny_ * (py[k] - cy) + nz_ * (pz[k] - cz), d_far = ny_ * (py[k2] - cy) + nz_ * (pz[k2] - cz);
                  const double r = hby * std::fabs(ny_) + hbz * std::fabs(nz_);
                  const double slack = 1e-9 * (std::fabs(d_edge) + std::fabs(d_far) + r); /* of this arithmetic itself */
                  apart = std::min(d_edge, d_far) > r + slack || std::max(d_edge, d_far) < -r - slack;
                }
                if (!apart) buckets[(size_t)a * G + b].push_back(i);
              }
          }
          mcol[c] = make_int2((int)mstart.size(), G);
          for (auto &bk : buckets) {
            mstart.push_back((int32_t)ment.size());
            for (int32_t i : bk) {
              ment.push_back(make_int2(ent[i].x, (ent_x[i] << 16) | (ent[i].y + 1)));
              ment_yz.push_back(ent_yz[i]);
            }
          }
          mstart.push_back((int32_t)ment.size()); /* end of the column's last bucket */
        }
      if (getenv("DSIM_FILL_STATS") && !mstart.empty()) { /* development: the shape of the bucket index on stderr */
        /* buckets weighted by their area (a uniform candidate's chance to land in them), in units of one column */
        double w_all = 0, w_full = 0, w_ent = 0, w_max = 0;
        int64_t n_full = 0, e_max = 0;
        for (size_t c = 0; c < mcol.size(); c++) {
          if (mcol[c].x < 0) continue;
          const int G = mcol[c].y;
          const double w = 1.0 / ((double)G * G);
          for (int b = 0; b < G * G; b++) {
            const int64_t ne = mstart[(size_t)mcol[c].x + b + 1] - mstart[(size_t)mcol[c].x + b];
            w_all += w;
            if (ne > 0) w_full += w, w_ent += w * (double)ne, n_full++, e_max = std::max(e_max, ne);
          }
        }
        fprintf(stderr, "fill index: %zu columns (%.0f non-empty), %zu buckets (%lld non-empty, %.3f of the non-empty columns' area), "
                "%zu entries, %.1f per non-empty bucket by area, max %lld\n", mcol.size(), w_all, mstart.size(), (long long)n_full,
                w_full / std::max(w_all, 1.0), ment.size(), w_ent / std::max(w_full, 1e-30), (long long)e_max);
        (void)w_max;
      }
      if (mstart.empty()) mstart.push_back(0);
      if (ment.empty()) {
        ment.push_back(make_int2(0, 0));
        ment_yz.push_back(make_float4(0, 0, 0, 0));
      }
    }
    HIP_TRY(hipMalloc(&fs->d_mcol, mcol.size() * sizeof(int2)));
    HIP_TRY(hipMalloc(&fs->d_mstart, mstart.size() * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&fs->d_ment, ment.size() * sizeof(int2)));
    HIP_TRY(hipMalloc(&fs->d_ment_yz, ment_yz.size() * sizeof(float4)));
    HIP_TRY(hipMemcpy(fs->d_mcol, mcol.data(), mcol.size() * sizeof(int2), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(fs->d_mstart, mstart.data(), mstart.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(fs->d_ment, ment.data(), ment.size() * sizeof(int2), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(fs->d_ment_yz, ment_yz.data(), ment_yz.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&fs->d_col_start, col_start.size() * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&fs->d_col_ent, col_ent.size() * sizeof(int2)));
    HIP_TRY(hipMalloc(&fs->d_ent, ent.size() * sizeof(int2)));
    HIP_TRY(hipMalloc(&fs->d_ent_yz, ent_yz.size() * sizeof(float4)));
    HIP_TRY(hipMemcpy(fs->d_col_start, col_start.data(), col_start.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(fs->d_col_ent, col_ent.data(), col_ent.size() * sizeof(int2), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(fs->d_ent, ent.data(), ent.size() * sizeof(int2), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(fs->d_ent_yz, ent_yz.data(), ent_yz.size() * sizeof(float4), hipMemcpyHostToDevice));
  }
  FillParams &fp = fs->fp;
  fp.col_start = fs->d_col_start;
  fp.col_ent = fs->d_col_ent;
  fp.ent = fs->d_ent;
  fp.ent_yz = fs->d_ent_yz;
  fp.mcol = fs->d_mcol;
  fp.mstart = fs->d_mstart;
  fp.ment = fs->d_ment;
  fp.ment_yz = fs->d_ment_yz;
  fp.n_points = n;
  fp.s0 = fs->s0;
  fp.s1 = fs->s1;
  fp.intra = intra ? 1 : 0;
  for (int k = 0; k < 3; k++) fp.voxel[k] = voxel_size[k];
  fp.mesh = mesh_dev(fs->dm);
  fs->lds = (size_t)(fs->dm.nx + fs->dm.ny + fs->dm.nz + 3) * sizeof(double);
  /* staging rows per workgroup: a batch is sized (by the callers' batch rule) so that it accepts about as many points as
   * are still wanted, i.e. at most ~256 per workgroup; 1024 leaves room for an estimate that is off by a factor of four,
   * and a batch that overflows all the same is redone launch by launch (never more than 256 per workgroup) */
  fs->n_blocks = grid_for(n, 256);
  fs->stage_cap = 1024;
  HIP_TRY(hipMalloc(&fs->d_stage, (size_t)fs->n_blocks * fs->stage_cap * sizeof(FillEntry)));
  HIP_TRY(hipMalloc(&fs->d_stage_count, (size_t)fs->n_blocks * sizeof(int32_t)));
  HIP_TRY(hipMalloc(&fs->d_cnt, (size_t)FILL_MAX_BATCH * fs->n_blocks * sizeof(int32_t)));
  HIP_TRY(hipMalloc(&fs->d_totals, FILL_MAX_BATCH * sizeof(int64_t)));
  HIP_TRY(hipMalloc(&fs->d_overflow, sizeof(int32_t)));
  fp.stage = fs->d_stage;
  fp.stage_cap = fs->stage_cap;
  fp.stage_count = fs->d_stage_count;
  fp.cnt = fs->d_cnt;
  fp.overflow = fs->d_overflow;
  fp.n_launches = 1;
  return DSIM_OK;
}

extern "C" int dsim_fill_open(dsim_fill **out, int device, int64_t thread_begin, int64_t thread_end, int intra, uint64_t seed,
                              const double *vertices, int64_t n_vertices, const int64_t *faces, int64_t n_faces, const double voxel_size[3],
                              const double *xs, const double *ys, const double *zs, const int64_t n_sv[3], const int64_t *subvoxel_indices,
                              const int64_t *triangle_indices, int64_t n_refs) {
  if (!out) return fail(DSIM_ERR_INVALID, "dsim_fill_open: out is NULL");
  *out = nullptr;
  if (thread_begin < 0 || thread_end <= thread_begin || !voxel_size) return fail(DSIM_ERR_INVALID, "dsim_fill_open: bad thread range");
  if (thread_end - thread_begin >= (int64_t)INT32_MAX || ((uint64_t)thread_end >> 47)) return fail(DSIM_ERR_INVALID, "dsim_fill_open: too many threads");
  int ndev = dsim_device_count();
  if (ndev <= 0) return fail(DSIM_ERR_NO_DEVICE, "no HIP device visible (need an AMD GPU with ROCm)");
  if (device < 0 || device >= ndev) return fail(DSIM_ERR_INVALID, "device out of range");
  dsim_fill *fs = new dsim_fill();
  fs->device = device;
  fs->n_threads = thread_end - thread_begin;
  int rc = fill_open_impl(fs, thread_begin, intra, seed, vertices, n_vertices, faces, n_faces, voxel_size, xs, ys, zs, n_sv, subvoxel_indices,
                          triangle_indices, n_refs);
  if (rc) {
    std::string keep = g_err;
    dsim_fill_close(fs);
    g_err = keep;
    return rc;
  }
  *out = fs;
  return DSIM_OK;
}

/* One batch of <= FILL_MAX_BATCH launches: three dispatches.  counts[l] = accepted candidates of launch l of the batch;
 * their points go to d_out rows rows_before, rows_before + 1, ... in (launch, thread) order (rows >= capacity are dropped).
 * *accepted = all accepted candidates of the batch.  A batch in which a workgroup ran out of staging rows is redone
 * launch by launch from the RNG states it started with. */
static int fill_batch(dsim_fill *fs, int n_launches, int64_t *counts, int64_t rows_before, int64_t capacity, int64_t *accepted) {
  const int64_t n = fs->n_threads;
  HIP_TRY(hipMemcpyAsync(fs->s0_keep, fs->s0, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, fs->stream));
  HIP_TRY(hipMemcpyAsync(fs->s1_keep, fs->s1, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, fs->stream));
  HIP_TRY(hipMemsetAsync(fs->d_overflow, 0, sizeof(int32_t), fs->stream));
  fs->fp.n_launches = n_launches;
  fill_batch_kernel<<<fs->n_blocks, 256, fs->lds, fs->stream>>>(fs->fp);
  HIP_TRY(hipGetLastError());
  fill_scan_kernel<<<n_launches, 256, 0, fs->stream>>>(fs->d_cnt, fs->n_blocks, fs->d_totals);
  HIP_TRY(hipGetLastError());
  fill_place_kernel<<<fs->n_blocks, 256, 0, fs->stream>>>(fs->d_stage, fs->d_stage_count, fs->stage_cap, fs->d_cnt, fs->d_totals, n_launches,
                                                           rows_before, capacity, fs->d_out);
  HIP_TRY(hipGetLastError());
  int32_t overflow = 0;
  HIP_TRY(hipMemcpyAsync(&overflow, fs->d_overflow, sizeof(int32_t), hipMemcpyDeviceToHost, fs->stream));
  HIP_TRY(hipMemcpyAsync(counts, fs->d_totals, (size_t)n_launches * sizeof(int64_t), hipMemcpyDeviceToHost, fs->stream));
  HIP_TRY(hipStreamSynchronize(fs->stream));
  if (overflow) {
    if (n_launches == 1) return fail(DSIM_ERR_STATE, "dsim_fill_mesh: a single launch overflowed its staging rows"); /* cannot happen: 256 <= stage_cap */
    HIP_TRY(hipMemcpyAsync(fs->s0, fs->s0_keep, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, fs->stream));
    HIP_TRY(hipMemcpyAsync(fs->s1, fs->s1_keep, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, fs->stream));
    int64_t total = 0;
    for (int l = 0; l < n_launches; l++) {
      int64_t one = 0;
      int rc = fill_batch(fs, 1, counts + l, rows_before + total, capacity, &one);
      if (rc) return rc;
      total += one;
    }
    *accepted = total;
    return DSIM_OK;
  }
  int64_t total = 0;
  for (int l = 0; l < n_launches; l++) total += counts[l];
  *accepted = total;
  return DSIM_OK;
}

/* n_launches more launches over the session's threads.  counts[l] = accepted candidates of launch l; their points,
 * in (launch, thread) order, go to points[(row, 3)] - at most `capacity` rows, the rest is dropped (the caller sizes
 * its batches so that nothing it needs is; *n_rows = rows written).  The launches run in batches of up to 64 per dispatch
 * (fill_batch); accepted points are put in order ON THE DEVICE (simulations.py:573-576 keeps the finite rows of every
 * launch in thread order); only counts and rows cross PCIe, once per call. */
extern "C" int dsim_fill_run(dsim_fill *fs, int64_t n_launches, int64_t *counts, double *points, int64_t capacity, int64_t *n_rows) {
  if (!fs || n_launches <= 0 || !counts || capacity < 0 || (capacity > 0 && !points)) return fail(DSIM_ERR_INVALID, "dsim_fill_run: bad arguments");
  HIP_TRY(hipSetDevice(fs->device));
  if (fs->launches + n_launches > 1000000) return fail(DSIM_ERR_STATE, "dsim_fill_mesh: no progress after 1000000 launches (is the surface closed?)");
  const int64_t cap = std::max<int64_t>(capacity, 1);
  if (fs->out_cap < cap) {
    if (fs->d_out) HIP_TRY(hipFree(fs->d_out));
    fs->d_out = nullptr;
    HIP_TRY(hipMalloc(&fs->d_out, (size_t)cap * 3 * sizeof(double)));
    fs->out_cap = cap;
  }
  int64_t have = 0;
  for (int64_t k = 0; k < n_launches; k += FILL_MAX_BATCH) {
    const int nl = (int)std::min<int64_t>(FILL_MAX_BATCH, n_launches - k);
    int64_t got = 0;
    int rc = fill_batch(fs, nl, counts + k, have, capacity, &got);
    if (rc) return rc;
    have += got;
  }
  fs->launches += n_launches;
  const int64_t rows = std::min(have, capacity);
  if (rows > 0) {
    HIP_TRY(hipMemcpyAsync(points, fs->d_out, (size_t)rows * 3 * sizeof(double), hipMemcpyDeviceToHost, fs->stream));
    HIP_TRY(hipStreamSynchronize(fs->stream));
  }
  if (n_rows) *n_rows = rows;
  return DSIM_OK;
}

extern "C" int dsim_fill_mesh(int device, int64_t n_points, int intra, uint64_t seed, const double *vertices, int64_t n_vertices,
                              const int64_t *faces, int64_t n_faces, const double voxel_size[3], const double *xs,
                              const double *ys, const double *zs, const int64_t n_sv[3], const int64_t *subvoxel_indices,
                              const int64_t *triangle_indices, int64_t n_refs, double *out_points) {
  if (n_points <= 0 || !out_points || !voxel_size) return fail(DSIM_ERR_INVALID, "dsim_fill_mesh: bad arguments");
  dsim_fill *fs = nullptr;
  int rc = dsim_fill_open(&fs, device, 0, n_points, intra, seed, vertices, n_vertices, faces, n_faces, voxel_size, xs, ys, zs, n_sv,
                          subvoxel_indices, triangle_indices, n_refs);
  if (rc) return rc;
  /* The running count comes back after a BATCH of launches: a launch past the n-th accepted point only produces
   * rows that are dropped (and advances RNG states nobody reads again), so running a few too many is harmless;
   * the batch is sized from the acceptance rate seen so far so that the stream is not drained after each of the
   * ~1/rate launches a thin structure needs (the neuron: ~4000). */
  int64_t have = 0, launches = 0, batch = 1;
  std::vector<int64_t> counts;
  while (have < n_points && rc == DSIM_OK) {
    counts.assign((size_t)batch, 0);
    int64_t rows = 0;
    rc = dsim_fill_run(fs, batch, counts.data(), out_points + have * 3, n_points - have, &rows);
    if (rc) break;
    launches += batch;
    have += rows;
    if (have > 0 && have < n_points) {
      const double per_launch = (double)have / (double)launches;
      const double left = (double)(n_points - have) / per_launch;
      batch = (int64_t)std::min(64.0, std::max(1.0, std::floor(left * 0.9)));
    } else if (have == 0) {
      batch = std::min<int64_t>(batch * 2, 64);
    }
  }
  const std::string keep = g_err;
  dsim_fill_close(fs);
  g_err = keep;
  return rc;
}

/* [r5] dsim_fill_mesh with the threads of every launch split over several devices - one fill session per listed device
 * (a device may be named more than once), each on a host thread of its own, the per-launch accepted counts merged in
 * memory: what sharding.fill_mesh_sharded does across processes with an all-gather.  Thread t draws from RNG state t
 * whatever the split and the accepted candidates of a launch are appended in thread order (simulations.py:573-576), so
 * the rows are those of dsim_fill_mesh bit for bit; the device work per device is 1 / n_devices of it. */
extern "C" int dsim_fill_mesh_multi(const int32_t *device_ids, int32_t n_devices, int64_t n_points, int intra, uint64_t seed,
                                    const double *vertices, int64_t n_vertices, const int64_t *faces, int64_t n_faces,
                                    const double voxel_size[3], const double *xs, const double *ys, const double *zs, const int64_t n_sv[3],
                                    const int64_t *subvoxel_indices, const int64_t *triangle_indices, int64_t n_refs, double *out_points) {
  if (!device_ids || n_devices < 1 || n_devices > 1024) return fail(DSIM_ERR_INVALID, "dsim_fill_mesh_multi: bad device list");
  if (n_points <= 0 || !out_points || !voxel_size) return fail(DSIM_ERR_INVALID, "dsim_fill_mesh_multi: bad arguments");
  const int G = (int)std::min<int64_t>(n_devices, n_points);
  if (G == 1)
    return dsim_fill_mesh(device_ids[0], n_points, intra, seed, vertices, n_vertices, faces, n_faces, voxel_size, xs, ys, zs, n_sv,
                          subvoxel_indices, triangle_indices, n_refs, out_points);
  std::vector<dsim_fill *> fs((size_t)G, nullptr);
  std::vector<int64_t> t0((size_t)G + 1);
  for (int k = 0; k <= G; k++) t0[(size_t)k] = (int64_t)k * (n_points / G) + std::min<int64_t>(k, n_points % G);
  std::vector<int> rcs((size_t)G, DSIM_OK);
  std::vector<std::string> msgs((size_t)G);
  auto each = [&](const std::function<int(int)> &f) -> int {
    std::vector<std::thread> th;
    auto work = [&](int k) {
      rcs[(size_t)k] = f(k);
      if (rcs[(size_t)k] != DSIM_OK) msgs[(size_t)k] = g_err;
    };
    for (int k = 1; k < G; k++) th.emplace_back(work, k);
    work(0);
    for (auto &t : th) t.join();
    for (int k = 0; k < G; k++)
      if (rcs[(size_t)k] != DSIM_OK) return fail(rcs[(size_t)k], "dsim_fill_mesh_multi: session %d on device %d: %s", k, device_ids[k], msgs[(size_t)k].c_str());
    return DSIM_OK;
  };
  int rc = each([&](int k) {
    return dsim_fill_open(&fs[(size_t)k], device_ids[k], t0[(size_t)k], t0[(size_t)k + 1], intra, seed, vertices, n_vertices, faces, n_faces, voxel_size,
                          xs, ys, zs, n_sv, subvoxel_indices, triangle_indices, n_refs);
  });
  int64_t have = 0, launches = 0, batch = 1;
  std::vector<std::vector<int64_t>> counts((size_t)G);
  std::vector<std::vector<double>> pts((size_t)G);
  std::vector<int64_t> rows((size_t)G, 0);
  while (rc == DSIM_OK && have < n_points) {
    rc = each([&](int k) {
      const int64_t threads = t0[(size_t)k + 1] - t0[(size_t)k];
      /* a session's accepted points beyond the (n_points - have)-th cannot be needed; with an estimate of the rate the
       * buffer stays small (8 x the expectation - an overflow is reported below, never dropped silently) */
      int64_t cap = std::min<int64_t>(n_points - have, threads * batch);
      if (launches > 0 && have > 0) cap = std::min<int64_t>(cap, (int64_t)(8.0 * (double)have / (double)launches * (double)batch * (double)threads / (double)n_points) + 100000);
      counts[(size_t)k].assign((size_t)batch, 0);
      pts[(size_t)k].resize((size_t)std::max<int64_t>(cap, 1) * 3);
      int rc1 = dsim_fill_run(fs[(size_t)k], batch, counts[(size_t)k].data(), pts[(size_t)k].data(), cap, &rows[(size_t)k]);
      if (rc1) return rc1;
      int64_t got = 0;
      for (int64_t c : counts[(size_t)k]) got += c;
      if (got > rows[(size_t)k] && rows[(size_t)k] < n_points - have) return fail(DSIM_ERR_STATE, "%lld accepted points in one batch overflow the buffer of %lld", (long long)got, (long long)cap);
      return DSIM_OK;
    });
    if (rc) break;
    /* launch l of the batch: its accepted candidates in thread order = session 0's, then session 1's, ... */
    int64_t at = have;
    std::vector<int64_t> used((size_t)G, 0);
    for (int64_t l = 0; l < batch; l++)
      for (int k = 0; k < G; k++) {
        const int64_t c = counts[(size_t)k][(size_t)l];
        const int64_t avail = std::max<int64_t>(0, std::min<int64_t>(c, rows[(size_t)k] - used[(size_t)k]));
        const int64_t take = std::max<int64_t>(0, std::min<int64_t>(avail, n_points - at));
        if (take > 0) memcpy(out_points + at * 3, pts[(size_t)k].data() + used[(size_t)k] * 3, (size_t)take * 3 * sizeof(double));
        used[(size_t)k] += c;
        at += c;
      }
    launches += batch;
    have = at;
    if (launches > 1000000) rc = fail(DSIM_ERR_STATE, "dsim_fill_mesh: no progress after 1000000 launches (is the surface closed?)");
    if (have > 0 && have < n_points) { /* the batch rule of dsim_fill_mesh */
      const double left = (double)(n_points - have) / ((double)have / (double)launches);
      batch = (int64_t)std::min(64.0, std::max(1.0, std::floor(left * 0.9)));
    } else if (have == 0) {
      batch = std::min<int64_t>(batch * 2, 64);
    }
  }
  const std::string keep = g_err;
  for (dsim_fill *f : fs)
    if (f) (void)dsim_fill_close(f);
  g_err = keep;
  return rc;
}

/* -------------------------------------------- mesh subdivision (host only) --- */
/* substrates.py:272-536, restated operation by operation in IEEE double WITHOUT
 * fused multiply-adds (the reference runs this part as plain numpy / numba CPU
 * code), so that membership matches the reference's .npy fixtures exactly. */

static inline double sd_dot(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void sd_cross(const double *a, const double *b, double *c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

/* substrates.py:291-368 (Akenine-Moller SAT); tri (3,3), box (2,3) */
static bool triangle_box_overlap(const double tri[9], const double box[6]) {
  double c[3], hh[3], v[9];
  for (int i = 0; i < 3; i++) {
    c[i] = (box[i] + box[3 + i]) / 2.0; /* np.mean of two values */
    hh[i] = std::fabs(box[3 + i] - box[i]) / 2;
  }
  for (int k = 0; k < 3; k++)
    for (int i = 0; i < 3; i++) v[k * 3 + i] = tri[k * 3 + i] - c[i];
  /* triangle AABB against the box: the reference rejects only if ALL three axes are outside (:323) */
  bool all_above = true, all_below = true;
  for (int i = 0; i < 3; i++) {
    double mn = std::min(v[i], std::min(v[3 + i], v[6 + i]));
    double mx = std::max(v[i], std::max(v[3 + i], v[6 + i]));
    if (!(mn > hh[i])) all_above = false;
    if (!(mx < -hh[i])) all_below = false;
  }
  if (all_above || all_below) return false;
  double f[9];
  for (int i = 0; i < 3; i++) {
    f[i] = v[3 + i] - v[i];
    f[3 + i] = v[6 + i] - v[3 + i];
    f[6 + i] = v[i] - v[6 + i];
  }
  double normal[3];
  sd_cross(f, f + 3, normal);
  const double sg[8][3] = {{1, 1, 1}, {-1, -1, -1}, {-1, 1, 1}, {1, -1, -1}, {1, -1, 1}, {-1, 1, -1}, {1, 1, -1}, {-1, -1, 1}};
  bool in_plane = false, all_behind = true, none_behind = true;
  for (int k = 0; k < 8; k++) {
    double d[3];
    for (int i = 0; i < 3; i++) d[i] = v[i] - sg[k][i] * hh[i];
    double dp = sd_dot(normal, d);
    bool behind = false;
    if (dp == 0)
      in_plane = true;
    else
      behind = dp > 0;
    if (behind)
      none_behind = false;
    else
      all_behind = false;
  }
  if (!in_plane && (all_behind || none_behind)) return false;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double e[3] = {0, 0, 0};
      e[i] = 1.0;
      double a[3];
      sd_cross(e, f + 3 * j, a);
      double aa[3] = {std::fabs(a[0]), std::fabs(a[1]), std::fabs(a[2])};
      double r = sd_dot(hh, aa);
      double p0 = sd_dot(a, v), p1 = sd_dot(a, v + 3), p2 = sd_dot(a, v + 6);
      double mn = std::min(p0, std::min(p1, p2)), mx = std::max(p0, std::max(p1, p2));
      if (mn > r || mx < -r) return false;
    }
  return true;
}

/* substrates.py:371-419 */
static void interval_sv_overlap(const double *xs, int64_t len, double x1, double x2, int64_t *ll_out, int64_t *ul_out) {
  double xmin = std::min(x1, x2), xmax = std::max(x1, x2);
  int64_t ll, ul;
  if (xmin <= xs[0])
    ll = 0;
  else if (xmin >= xs[len - 1])
    ll = len - 1;
  else {
    ll = 0;
    for (int64_t i = 0; i < len; i++)
      if (xs[i] > xmin) {
        ll = i - 1;
        break;
      }
  }
  if (xmax >= xs[len - 1])
    ul = len - 1;
  else if (xmax <= xs[0])
    ul = 0;
  else {
    ul = len - 1;
    for (int64_t i = 0; i < len; i++)
      if (!(xs[i] < xmax)) {
        ul = i;
        break;
      }
  }
  if (ll != ul) {
    *ll_out = ll;
    *ul_out = ul;
  } else if (ll != len - 1) {
    *ll_out = ll;
    *ul_out = ul + 1;
  } else {
    *ll_out = ll - 1;
    *ul_out = ul;
  }
}

extern "C" int dsim_triangle_box_overlap(const double triangle[9], const double box[6]) {
  if (!triangle || !box) return fail(DSIM_ERR_INVALID, "dsim_triangle_box_overlap: null argument");
  return triangle_box_overlap(triangle, box) ? 1 : 0;
}

extern "C" int dsim_interval_sv_overlap(const double *xs, int64_t n_xs, double x1, double x2, int64_t *ll, int64_t *ul) {
  if (!xs || n_xs < 2 || !ll || !ul) return fail(DSIM_ERR_INVALID, "dsim_interval_sv_overlap: bad argument");
  interval_sv_overlap(xs, n_xs, x1, x2, ll, ul);
  return DSIM_OK;
}

extern "C" int dsim_mesh_subdivide(const double *vertices, int64_t n_vertices, const int64_t *faces, int64_t n_faces,
                                   const double *xs_in, const double *ys_in, const double *zs_in, const int64_t n_sv[3],
                                   int64_t *subvoxel_indices, int64_t *triangle_indices, int64_t *n_refs) {
  if (!vertices || !faces || !xs_in || !ys_in || !zs_in || !n_sv || !n_refs)
    return fail(DSIM_ERR_INVALID, "dsim_mesh_subdivide: null argument");
  for (int k = 0; k < 3; k++)
    if (n_sv[k] < 1) return fail(DSIM_ERR_INVALID, "dsim_mesh_subdivide: n_sv must be >= 1");
  const int64_t n_sub = n_sv[0] * n_sv[1] * n_sv[2];
  std::vector<double> xs(xs_in, xs_in + n_sv[0] + 1), ys(ys_in, ys_in + n_sv[1] + 1), zs(zs_in, zs_in + n_sv[2] + 1);
  /* two passes over (triangle, subvoxel) pairs in the reference's order: count, then fill (CSR) */
  std::vector<int64_t> count((size_t)n_sub, 0);
  std::vector<std::pair<int64_t, int64_t>> pairs; /* (subvoxel, triangle), generated in triangle order */
  for (int64_t f = 0; f < n_faces; f++) {
    double tri[9];
    for (int a = 0; a < 3; a++) {
      int64_t vi = faces[f * 3 + a];
      if (vi < 0 || vi >= n_vertices) return fail(DSIM_ERR_INVALID, "dsim_mesh_subdivide: face %lld out of range", (long long)f);
      for (int b = 0; b < 3; b++) tri[a * 3 + b] = vertices[vi * 3 + b];
    }
    int64_t lo[3], hi[3];
    const std::vector<double> *ax[3] = {&xs, &ys, &zs};
    for (int k = 0; k < 3; k++) {
      double mn = std::min(tri[k], std::min(tri[3 + k], tri[6 + k]));
      double mx = std::max(tri[k], std::max(tri[3 + k], tri[6 + k]));
      interval_sv_overlap(ax[k]->data(), (int64_t)ax[k]->size(), mn, mx, &lo[k], &hi[k]);
    }
    for (int64_t x = lo[0]; x < hi[0]; x++)
      for (int64_t y = lo[1]; y < hi[1]; y++)
        for (int64_t z = lo[2]; z < hi[2]; z++) {
          double box[6] = {xs[x], ys[y], zs[z], xs[x + 1], ys[y + 1], zs[z + 1]};
          if (triangle_box_overlap(tri, box)) {
            int64_t sv = x * n_sv[1] * n_sv[2] + y * n_sv[2] + z;
            count[sv]++;
            pairs.emplace_back(sv, f);
          }
        }
  }
  *n_refs = (int64_t)pairs.size();
  if (subvoxel_indices) {
    int64_t c = 0;
    for (int64_t s = 0; s < n_sub; s++) {
      subvoxel_indices[s * 2] = c;
      c += count[s];
      subvoxel_indices[s * 2 + 1] = c;
    }
  }
  if (triangle_indices) {
    if (!subvoxel_indices) return fail(DSIM_ERR_INVALID, "dsim_mesh_subdivide: triangle_indices needs subvoxel_indices");
    std::vector<int64_t> cursor((size_t)n_sub);
    for (int64_t s = 0; s < n_sub; s++) cursor[s] = subvoxel_indices[s * 2];
    for (auto &pr : pairs) triangle_indices[cursor[pr.first]++] = pr.second; /* triangle order kept within a subvoxel */
  }
  return DSIM_OK;
}
