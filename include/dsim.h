/* dsim.h - C ABI of the MI355X-native Monte Carlo diffusion-MRI hot path.
 *
 * The reference (kerkelae/disimpy) has no C ABI: its device boundary is Numba's
 * `kernel[gs, bs, stream](*args)` launch inside disimpy/simulations.py
 * simulation() (:1051-1429).  Each entry point below names the reference lines
 * it replaces; together they are exactly what a ctypes binding inside
 * simulation() needs (see INTEGRATION.md for that binding).
 *
 * Conventions: plain pointers and sizes, no C++/torch types.  Every function
 * returns DSIM_OK (0) or a negative error code and never throws; after a
 * failure dsim_last_error() returns a message (thread-local).  All functions
 * block the calling host thread until the work is done.  Host buffers are
 * caller-owned; device memory is library-owned behind the opaque handle.
 * Layouts follow the reference: positions (N,3) row-major float64, phases
 * (M,N) float64, gradient components (M,T) float64 each, RNG states
 * {uint64 s0, uint64 s1} per walker, mesh index arrays int64.  Inside the
 * library everything is re-laid-out SoA for coalesced HBM access.
 *
 * Hard limits (every one fails loudly with DSIM_ERR_INVALID / DSIM_ERR_STATE and a message,
 * none silently changes results):
 *   - walker indices: first_walker + n_walkers < 2^47 (depth of the RNG jump tables);
 *   - mesh: n_faces, n_refs and prod(n_sv) < 2^31, every n_sv[k] in [1, 4096]; the wavefront-cooperative
 *     traversal (the default, mesh_traversal 2) assumes the reference's own subvoxel grid, xs = linspace(0, L, n + 1)
 *     on every axis (substrates.py:506-508): any other boundaries handed in here are walked by the per-lane
 *     traversal (mesh_traversal 1; same results, checked against the boundary values themselves), as is a mesh of 2^25
 *     triangles or more (the cooperative traversal names a triangle by a 32-bit byte offset); the fine grid built on top of
 *     the subvoxel lists may hold at most 2^28 records (16 bytes each, named by byte offsets: dsim_set_mesh fails beyond).  A walker whose
 *     segment reaches more than 31 periodic images away from the voxel, spans more than 64 runs of fine cells or meets
 *     two different triangles at bit-equal distances is redone by its own lane in the reference's order (same
 *     results); the LDS request (subvoxel boundaries + tables + work queues) must stay below 160 KB,
 *     i.e. n_sv[0] + n_sv[1] + n_sv[2] below about 14 000;
 *   - spatial re-sorting of the walker slots (mesh substrates) needs 4096 <= n_walkers < 2^31
 *     per handle; outside that range the run proceeds unsorted (slower, same results); a recording run
 *     (dsim_set_recording) is sorted like any other - the kernels record slot i into the column of its walker;
 *   - recording (dsim_set_recording): the range of one dsim_run_* call must fit 32 GB of
 *     positions (24 bytes per walker and step); M > 32 or mesh runs with M > 2 record
 *     internally and cut their launches to what the device can spare instead;
 *   - dsim_fill_mesh: n_points < 2^31, and a sample that meets 1000 distinct triangles is
 *     abandoned like in the reference (simulations.py:478-479);
 *   - gradient: M >= 1, T >= 1, T * M * 32 bytes of device memory for the re-laid-out waveform.
 *
 * Environment.  The library reads eight variables, all of them tuning or diagnostic switches for measurement runs -
 * results never depend on them (the GPU parity tests run with and without):
 *   DSIM_GROUPS=1..4      walker groups (streams) of a mesh run with per-walker clocks; default 2 from 2^18 walkers;
 *   DSIM_SORT_BITS=6..24  log2 of the Morton bins of the walker re-sort; default 18;
 *   DSIM_XCD_RUN=0..10    log2 of the consecutive walker groups (of 256) an XCD takes in turn; default 3;
 *   DSIM_MICRO=t1,t2,t3,t4,min_cell  thresholds of the fine grid (list length -> cells per subvoxel) and the smallest
 *                         cell in step lengths; default 2,4,16,2^30,1.5;
 *   DSIM_PHASE_LDS=bytes  dynamic LDS the phase kernel of a recorded mesh run (more than four measurements) asks for
 *                         beyond what it uses, so that its workgroups enter a CU only where a walk workgroup has
 *                         retired; default: a rule on the walk kernel's own LDS (phase_lds_reserve);
 *   DSIM_FILL_STATS=1     prints the shape of the mesh fill's bucket index on stderr when a fill session opens.
 * The Python binding adds DSIM_LIB=<path> (disimpy_amd/_lib.py): load another build of this library, e.g. an older
 * kernel for a same-box A/B; every symbol of this header must be there, or loading fails - unless DSIM_LIB_OLD_ABI=1
 * says that the build is an OLDER one kept for such a comparison: entry points it lacks raise when they are called.
 * Two more switches belong to this round: DSIM_DENSE_CELLS_MB (cap of the one-level cell table of a mesh, default 4096;
 * 0 = two-level look-ups) and DSIM_COST_PER_RECORD (weight of the spatial cut's cost model, default 0.056).
 */
#ifndef DSIM_H
#define DSIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSIM_OK 0
#define DSIM_ERR_INVALID (-1)   /* bad argument                                   */
#define DSIM_ERR_NO_DEVICE (-2) /* no usable AMD GPU (reference: simulations.py:1116-1125) */
#define DSIM_ERR_HIP (-3)       /* a HIP runtime call failed                      */
#define DSIM_ERR_STATE (-4)     /* call order violated (e.g. run before set_mesh) */

typedef struct dsim_sim *dsim_handle;

/* Library version of this ABI (bumped on any signature change). */
int dsim_abi_version(void);

/* Message for the most recent failure on this thread ("" if none). */
const char *dsim_last_error(void);

/* Number of visible HIP devices; 0 when there is none.
 * Replaces: cuda.detect() at simulations.py:1116-1125. */
int dsim_device_count(void);

/* Host-side RNG state creation, bit-faithful to
 * numba.cuda.random.create_xoroshiro128p_states(n, seed) (call sites
 * simulations.py:524, :1171): state 0 = splitmix64(seed) in both words, state
 * i = state i-1 jumped 2^64 draws.  Writes the states of the global walker
 * indices [first_index, first_index + n) as n pairs {s0, s1} into out (2n
 * uint64).  Sequential in the index (128 draws per state), like the reference. */
int dsim_rng_init(uint64_t seed, uint64_t first_index, uint64_t n, uint64_t *out_states);

/* Same states, computed on the device in parallel with precomputed GF(2) jump
 * matrices and left in the handle (no host round trip). */
int dsim_rng_init_device(dsim_handle h, uint64_t seed);

/* Create a simulation of n_walkers walkers whose global indices start at
 * first_walker (multi-GPU sharding: walker i always uses RNG state i and
 * initial position i), M measurements and T time points.
 * gx, gy, gz: (M,T) row-major, the contiguous gradient[:, :, k] of
 * simulations.py:1174-1176.  step_l = sqrt(6*D*dt) (:1181).  Allocates
 * phases = 0 (:1177) and iter_exc = false (:1178), positions = 0.
 * RNG states are NOT initialised: call dsim_rng_init_device or dsim_set_rng_states. */
int dsim_create(dsim_handle *out, int device, int64_t n_walkers, int64_t first_walker, int32_t M, int32_t T,
                const double *gx, const double *gy, const double *gz, double dt, double step_l);

/* The same simulation spread over several devices of this node behind ONE handle (the reference's
 * simulation() is one function on one GPU, simulations.py:1051): the walkers [first_walker, first_walker + n_walkers)
 * are cut into n_devices balanced contiguous ranges (the first n_walkers % n_devices ranges hold one walker more),
 * range k lives on device device_ids[k] (a device may be named more than once: several shards on one GPU).  Every
 * other entry point of this header takes the handle as it takes a dsim_create handle: host arrays cover all
 * n_walkers walkers in order (each shard reads / fills its slice), dsim_run_* advance the shards concurrently (one
 * host thread per shard), dsim_get_signal adds the shards' exact integer sums (dsim_get_signal_limbs), dsim_write_traj
 * writes lines that hold every walker.  Walker i uses RNG state i and initial position i whatever the split, so every
 * output - per-walker arrays and the signal - is identical to a single-device run bit for bit.  No
 * torch, no process group: one process drives all devices. */
int dsim_create_multi(dsim_handle *out, const int32_t *device_ids, int32_t n_devices, int64_t n_walkers, int64_t first_walker,
                      int32_t M, int32_t T, const double *gx, const double *gy, const double *gz, double dt, double step_l);
/* number of shards behind a handle (1 for a dsim_create handle) */
int dsim_shard_count(dsim_handle h);

int dsim_destroy(dsim_handle h);

/* Initial positions, (n_walkers,3) row-major (cuda.to_device(positions), e.g. :1195). */
int dsim_set_positions(dsim_handle h, const double *positions_aos);

/* [r5] dsim_set_positions that also cuts the walkers of a multi-device handle (dsim_create_multi) into its shards by
 * SPACE: the walkers are ordered along the Morton curve of the positions and shard k takes the k-th stretch of that
 * order, so that every GPU works in its own part of a mesh substrate instead of at 1/G of the walker density all over
 * it.  When dsim_set_mesh was called BEFORE, the stretches are of equal estimated cost, not of equal counts (a
 * walker's cost: 1 + 0.056 x the records of the fine cell it starts in - walkers inside a soma are cheap, those in thin
 * dendrites dear, and the job ends with its slowest shard); the shards are then created again with their new sizes.
 * The reference has no counterpart (simulations.py:1051 is one GPU).  Walker i keeps RNG state i and position i,
 * host arrays stay in walker order, the exact signal sums do not depend on the cut: every output is the bytes of the
 * contiguous cut.  Before the first dsim_run_* call only; RNG states seeded or given earlier are re-seeded / moved.
 * Single-device handle: the same as dsim_set_positions. */
int dsim_partition_by_position(dsim_handle h, const double *positions_aos);

/* The cut itself, host only: the walkers ordered along the Morton curve of their positions (21 bits per axis, cubic
 * cells over the bounding box of the finite coordinates, ties by index), part k = the k-th stretch of that order -
 * with the balanced sizes of the contiguous cut (weights == NULL) or of equal total weight.  ids_out (n): the walkers
 * of part 0 ascending, then part 1, ...; part_start (n_parts + 1).  Deterministic: every process of a job launched as
 * one process per GPU computes the same cut. */
int dsim_partition_walkers(const double *positions_aos, int64_t n_walkers, int32_t n_parts, const double *weights, int64_t *ids_out,
                           int64_t *part_start);

/* The weights dsim_partition_by_position uses on a mesh (host only; the arguments of dsim_set_mesh plus the step
 * length the handle was created with): out_cost[i] = 1 + 0.056 x min(40, records of the fine cell of positions[i]). */
int dsim_mesh_walker_cost(const double *vertices, int64_t n_vertices, const int64_t *faces, int64_t n_faces, const double *xs,
                          const double *ys, const double *zs, const int64_t n_sv[3], const int64_t *subvoxel_indices,
                          const int64_t *triangle_indices, int64_t n_refs, double step_l, const double *positions_aos,
                          int64_t n_walkers, double *out_cost);

/* Single-device handle that holds such a part: slot i is walker first_walker + ids[i] (ascending, n of the handle;
 * NULL: the range again).  dsim_rng_init_device seeds slot i with the state of ITS walker; every other entry point
 * is per slot.  (Multi-device handles keep their own lists: dsim_partition_by_position.) */
int dsim_set_walker_ids(dsim_handle h, const int64_t *ids);

/* Upload RNG states produced by dsim_rng_init: n_walkers pairs {s0,s1}. */
int dsim_set_rng_states(dsim_handle h, const uint64_t *states);

/* Triangle mesh + voxelised triangle index of substrates._Substrate
 * (substrates.py:24-42; uploads at simulations.py:1358-1365).  vertices
 * (n_vertices,3) f64, faces (n_faces,3) i64, xs/ys/zs subvoxel boundaries of
 * length n_sv[k]+1, subvoxel_indices (prod(n_sv),2) i64 [start,end) into
 * triangle_indices (n_refs) i64.  The library rebuilds a compact device index
 * (int32 CSR + pre-gathered {A, B-A, C-A} triangle records) with identical
 * membership. */
int dsim_set_mesh(dsim_handle h, const double *vertices, int64_t n_vertices, const int64_t *faces, int64_t n_faces,
                  const double *xs, const double *ys, const double *zs, const int64_t n_sv[3],
                  const int64_t *subvoxel_indices, const int64_t *triangle_indices, int64_t n_refs);

/* Advance all walkers over time steps [t_begin, t_end) in ONE fused launch per
 * block of measurements (the reference launches one kernel per time step).
 * Each replaces the per-step launch loop named beside it. */
int dsim_run_free(dsim_handle h, int32_t t_begin, int32_t t_end); /* _cuda_step_free, :1198-1210 */
int dsim_run_sphere(dsim_handle h, double radius, int64_t max_iter, double epsilon, int32_t t_begin,
                    int32_t t_end); /* _cuda_step_sphere, :1268-1284 */
int dsim_run_cylinder(dsim_handle h, double radius, const double R[9], const double R_inv[9], int64_t max_iter,
                      double epsilon, int32_t t_begin, int32_t t_end); /* _cuda_step_cylinder, :1233-1251 */
int dsim_run_ellipsoid(dsim_handle h, const double semiaxes[3], const double R[9], const double R_inv[9],
                       int64_t max_iter, double epsilon, int32_t t_begin,
                       int32_t t_end); /* _cuda_step_ellipsoid, :1308-1326 */
int dsim_run_mesh(dsim_handle h, int64_t max_iter, double epsilon, double perm_prob, int32_t t_begin,
                  int32_t t_end); /* _cuda_step_mesh, :1369-1393 (without the 10 ms sleep of :1394) */

/* Signal assembly on the device (replaces the D2H of the (M,N) phase matrix and
 * the host numpy reduction of simulations.py:1406-1421): out_cos[m] = sum of
 * cos(phase[m,i]), out_sin[m] = sum of sin(phase[m,i]) over walkers that did
 * not exceed max_iter and whose phase is not NaN (nansum semantics);
 * n_exceeded = number of walkers with iter_exc set.  out_sin / n_exceeded may
 * be NULL. */
int dsim_get_signal(dsim_handle h, double *out_cos, double *out_sin, int64_t *n_exceeded);

/* The same sums before the final rounding.  The device adds every term up EXACTLY: cos / sin of a phase is rounded once
 * to a multiple of 2^-60 and the multiples are summed as integers, kept in two limbs per sum (q = hi * 2^30 + lo, each
 * limb an int64 that holds 2^33 terms).  limbs[4 m + {0, 1, 2, 3}] = {cos hi, cos lo, sin hi, sin lo} of measurement m.
 * Integer sums do not depend on the order of the additions: limbs of shards, devices or ranks simply ADD (one all-reduce
 * of 4 M + 1 int64 over RCCL), and dsim_signal_from_limbs turns the totals into the doubles dsim_get_signal returns
 * (exact sum, rounded once) - so a job cut over 8 GPUs returns the bytes of the 1-GPU job, whatever the cut.
 * dsim_signal_from_limbs needs no handle and no GPU. */
int dsim_get_signal_limbs(dsim_handle h, int64_t *limbs, int64_t *n_exceeded);
int dsim_signal_from_limbs(int32_t M, const int64_t *limbs, double *out_cos, double *out_sin);

/* Per-walker outputs for final_pos / traj / all_signals / the warning text. */
int dsim_get_positions(dsim_handle h, double *positions_aos);        /* (n_walkers,3), :1212, :1426 */
int dsim_get_phases(dsim_handle h, double *phases);                  /* (M,n_walkers), :1415          */
int dsim_get_iter_exc(dsim_handle h, uint8_t *iter_exc);             /* (n_walkers,), :1406           */
/* all_signals=True (simulations.py:1414-1417): signals[m, i] = real(exp(1j * phases[m, i])) = cos(phases[m, i]), NaN for
 * walkers with iter_exc set - evaluated on the device and handed over in blocks through pinned staging buffers on a
 * copy stream of its own (the reference copies the whole (M, N) phase matrix to the host and takes a complex
 * exponential of it on one core). */
int dsim_get_all_signals(dsim_handle h, double *signals);            /* (M,n_walkers)                 */
int dsim_get_rng_states(dsim_handle h, uint64_t *states);            /* n_walkers pairs               */

/* Initial positions inside ('intra' != 0) or outside the surface: the whole
 * _fill_mesh loop of simulations.py:505-579 (own RNG states from `seed`,
 * repeated _cuda_fill_mesh launches over n_points threads until n_points
 * points were accepted, accepted points kept in thread order).  The mesh
 * arrays are the ones the reference uploads there (wall triangles already
 * stripped for non-periodic substrates).  out: (n_points,3). */
int dsim_fill_mesh(int device, int64_t n_points, int intra, uint64_t seed, const double *vertices,
                   int64_t n_vertices, const int64_t *faces, int64_t n_faces, const double voxel_size[3],
                   const double *xs, const double *ys, const double *zs, const int64_t n_sv[3],
                   const int64_t *subvoxel_indices, const int64_t *triangle_indices, int64_t n_refs,
                   double *out_points);

/* [r5] The same fill with the threads of every launch split over several devices (one fill session per listed device,
 * host threads, the per-launch counts merged in memory): the rows of dsim_fill_mesh bit for bit, 1 / n_devices of the
 * device work each - what simulation() uses on a node with several GPUs (the reference's fill, simulations.py:505-579,
 * runs on its one GPU). */
int dsim_fill_mesh_multi(const int32_t *device_ids, int32_t n_devices, int64_t n_points, int intra, uint64_t seed,
                         const double *vertices, int64_t n_vertices, const int64_t *faces, int64_t n_faces,
                         const double voxel_size[3], const double *xs, const double *ys, const double *zs,
                         const int64_t n_sv[3], const int64_t *subvoxel_indices, const int64_t *triangle_indices,
                         int64_t n_refs, double *out_points);

/* The same loop for a RANGE of its threads, so that the GPUs of a node share it (disimpy_amd/sharding.py,
 * fill_mesh_sharded).  Thread t draws one candidate per launch from RNG state t whatever the range, and the reference
 * appends the accepted candidates of a launch in thread order (simulations.py:560-579): a session runs launches over
 * threads [thread_begin, thread_end) and reports per launch how many of them were accepted plus the accepted points in
 * (launch, thread) order; where a point stands in the global sequence follows from the counts of all sessions.
 * dsim_fill_mesh == one session over threads [0, n_points). */
typedef struct dsim_fill dsim_fill;
int dsim_fill_open(dsim_fill **out, int device, int64_t thread_begin, int64_t thread_end, int intra, uint64_t seed,
                   const double *vertices, int64_t n_vertices, const int64_t *faces, int64_t n_faces, const double voxel_size[3],
                   const double *xs, const double *ys, const double *zs, const int64_t n_sv[3], const int64_t *subvoxel_indices,
                   const int64_t *triangle_indices, int64_t n_refs);
/* n_launches further launches: counts[n_launches], points[(rows, 3)] with at most `capacity` rows (further accepted
 * points are dropped), *n_rows = rows written */
int dsim_fill_run(dsim_fill *session, int64_t n_launches, int64_t *counts, double *points, int64_t capacity, int64_t *n_rows);
int dsim_fill_close(dsim_fill *session);

/* Voxelised triangle index on the host (substrates.py:467-536,
 * _mesh_space_subdivision; the boundaries xs/ys/zs of :506-508 are passed in).
 * Call with triangle_indices == NULL to get the number of references in
 * *n_refs, then again with a buffer of that size.  subvoxel_indices:
 * (prod(n_sv),2) i64.  Needs no GPU. */
int dsim_mesh_subdivide(const double *vertices, int64_t n_vertices, const int64_t *faces, int64_t n_faces,
                        const double *xs, const double *ys, const double *zs, const int64_t n_sv[3],
                        int64_t *subvoxel_indices, int64_t *triangle_indices, int64_t *n_refs);

/* The clearance map dsim_set_mesh builds for the walk kernel, exported for tests (needs no GPU).
 * The reference's reflect loop ends a step without reflection and without an RNG draw when no
 * triangle is met within the step length (`min_d > step_l`, simulations.py:986-987); a walker farther
 * than step_l from every triangle (of every periodic image) can skip the look-ups that establish
 * this.  Every subvoxel is cut into F^3 cells, F = 1 << *log2_cells; bit ((a * F + b) * F + c) of a
 * subvoxel's block of max(1, F^3 / 32) words is SET when some triangle meets cell (a, b, c) inflated
 * by step_l * (1 + 1e-6) + 2e-3 cell sizes.  clear_idx[sv] = -1 (whole subvoxel clear) or the block
 * number.  Call with clear_idx == NULL to get *n_words (0: no map is used for this input), then
 * again with buffers of prod(n_sv) int32 and *n_words uint32. */
int dsim_mesh_clearance_map(const double *vertices, int64_t n_vertices, const int64_t *faces, int64_t n_faces,
                            const double *xs, const double *ys, const double *zs, const int64_t n_sv[3],
                            double step_l, int32_t *clear_idx, uint32_t *clear_bits, int64_t *n_words,
                            int32_t *log2_cells);

/* Host only: the fine grid dsim_set_mesh builds on top of the reference's subvoxel lists, k = 1 << *log2_cells.
 * k == 1: the cells are the subvoxels themselves (cell_start has prod(n_sv) + 1 entries, sv_base is not written).
 * k > 1: along x and y a non-empty subvoxel is cut into 2k units and there is a cell of two units at every unit but the
 * last (the cells overlap), along z into k cells; the subvoxel owns the slots sv_base[s] .. sv_base[s] + (2k)^2 k - 1
 * (sv_base[s] = -1: empty), slot ((i * 2k + j) * k + c) = the cell that starts at unit i along x, unit j along y, z cell c
 * (the slots with i or j = 2k - 1 stay empty).  The records of slot q are [cell_start[q], cell_start[q + 1]).  Per record
 * the triangle, its position in the reference's triangle_indices and the float32 box {min x, y, z, max x, y, z} of the
 * triangle's part inside the cell; rec_packed (8 dwords per record, may be NULL): the 16-byte record as the kernels read
 * it followed by the plane of its triangle, which the kernels keep once per triangle -
 * {lo_x | lo_y << 16, lo_z | nhi_x << 16, nhi_y | nhi_z << 16, triangle, plane nx, ny, nz, c as float32 bits}, the box
 * in 15-bit fixed point x -> x * *quantum (one float, may be NULL), lo rounded down, hi rounded up and stored as
 * nhi = 32767 - hi.  Exported so that the completeness of the grid can be tested without a GPU
 * (tests/test_fine_grid.py).  Call once with cell_start == NULL for the sizes, then with buffers of *n_cells + 1 and
 * *n_records elements. */
int dsim_mesh_fine_grid(const double *vertices, int64_t n_vertices, const int64_t *faces, int64_t n_faces, const double *xs,
                        const double *ys, const double *zs, const int64_t n_sv[3], const int64_t *subvoxel_indices,
                        const int64_t *triangle_indices, int64_t n_refs, double step_l, int32_t *log2_cells, int64_t *n_cells,
                        int64_t *n_records, int32_t *sv_base, uint32_t *cell_start, int32_t *rec_triangle, int32_t *rec_position,
                        float *rec_box, uint32_t *rec_packed, float *quantum);

/* The two predicates dsim_mesh_subdivide is built from, exported for unit tests
 * that mirror disimpy/tests/test_substrates.py:293-345: triangle (3,3) vs box
 * (2,3) overlap (substrates.py:291-368; returns 1/0), and the subvoxel index
 * range [ll, ul) overlapped by the interval [x1, x2] (substrates.py:371-419). */
int dsim_triangle_box_overlap(const double triangle[9], const double box[6]);
int dsim_interval_sv_overlap(const double *xs, int64_t n_xs, double x1, double x2, int64_t *ll, int64_t *ul);

/* Measurement hooks used by bench.py: device time (HIP events on the library's
 * own stream) of the most recent dsim_run_* call in milliseconds, the number of
 * kernel launches it made, and the name of the kernel. */
int dsim_last_run_ms(dsim_handle h, double *ms, int32_t *n_launches);

/* The same call's time inside the walk kernel alone (HIP events recorded on the library's
 * stream immediately before and after every launch of walk_kernel*, summed), and the number of
 * those launches: the run also contains the re-sort, gather and phase kernels.  bench.py's
 * roofline of the dominant kernel is computed from this figure. */
int dsim_last_walk_ms(dsim_handle h, double *ms, int32_t *n_launches);

/* Mesh runs with M <= 4 (the kernel with per-walker clocks) and at least 2^18 walkers sort and launch the two halves of the shard independently on
 * two streams, so that one half's launch fills the wavefront slots the other half leaves idle at the end of its
 * launch and during its re-sort (csrc/dsim_hip.hip, run_own_clock).  The launches of the two halves then overlap
 * and the sum above counts the overlap twice: *ms is the time during which at least one walk kernel was running
 * (equal to the sum when there is one stream), *n_streams the number of streams the run used. */
int dsim_last_walk_busy_ms(dsim_handle h, double *ms, int32_t *n_streams);

/* Trajectories (replaces the reference's per-step `positions.copy_to_host()` + file append of
 * simulations.py:1043-1048, :1211-1213 by one transfer per range of steps).
 * dsim_set_recording(h, 1): every following dsim_run_* call keeps the position of every walker
 * after every time step of its range in device memory (24 bytes per walker and step; the call
 * fails if the range does not fit in 32 GB); 0 switches it off and frees the buffer.
 * dsim_get_recorded copies steps [t_begin, t_end) of the LAST dsim_run_* call to the host,
 * component-major: out[((t - t_begin) * 3 + c) * n + i]. */
int dsim_set_recording(dsim_handle h, int32_t enable);
int dsim_get_recorded(dsim_handle h, int32_t t_begin, int32_t t_end, double *out);

/* The trajectories file itself (simulations.py:1043-1048: `str(value) + " "` per coordinate, one
 * line per time step).  dsim_write_traj appends (append != 0) or writes the lines of steps
 * [t_begin, t_end) of the last recorded dsim_run_* call to `path`; blocks of steps cross PCIe on
 * a copy stream of their own into two pinned buffers while host threads format the previous block
 * (the same double buffering serves dsim_get_recorded).  The text is byte-identical to what the
 * reference's Python writes: shortest round-trip decimals in CPython's float repr layout.
 * dsim_write_traj_line writes one such line from a host array (n_walkers,3) - the line of initial
 * positions that opens the file (:1194); it needs no GPU. */
int dsim_write_traj(dsim_handle h, const char *path, int32_t t_begin, int32_t t_end, int32_t append);
int dsim_write_traj_line(const char *path, const double *positions_aos, int64_t n_walkers, int32_t append);

/* Tuning knobs (0 keeps the default): threads per workgroup, time steps per
 * LDS gradient tile, measurements held in registers per pass, and - for mesh
 * substrates - the number of time steps between spatial re-sorts of the walker
 * slots (0 = automatic: 512 steps for the cooperative traversal, 256 / 128 when the shard is so small that a launch is only a
 * few rounds of the GPU's wavefront slots; negative = never), and where the kernels read the
 * gradient samples from (1 = LDS tiles, 2 = scalar loads), how the mesh index is
 * traversed (1 = per lane, 2 = wavefront-cooperative, using the clearance map -
 * walkers farther than a step from every triangle skip the traversal - in the
 * launches that start with >= 40 % of the walkers clear; 3 = never, 4 = always
 * the map; 5 = as 2, but every launch covers the same range of time steps for all
 * walkers instead of giving each wavefront a budget of passes - the round-1
 * scheduling, kept for comparison).  Results never depend on any of them. */
int dsim_set_tuning(dsim_handle h, int32_t block_size, int32_t tile_steps, int32_t meas_per_pass,
                    int32_t resort_steps, int32_t gradient_source, int32_t mesh_traversal);

#ifdef __cplusplus
}
#endif
#endif /* DSIM_H */
