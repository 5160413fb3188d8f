/* include/mpmhip.h — C ABI of the MI355X-native MLS-MPM time-stepping core (libmpmhip.so).
 *
 * This is the drop-in boundary for the hot path of yuanming-hu/taichi_mpm: the per-substep
 * sequence  sort -> P2G -> grid normalise + boundary -> G2P -> boundary cleanup  that the
 * reference runs on the CPU inside `MPM<3>::substep()` (src/mpm.cpp:452-575).  The reference has
 * no C ABI (it is a factory-registered C++ class reached through pybind11, src/mpm.cpp:983-988);
 * each entry point below names the reference interface it replaces, and INTEGRATION.md shows the
 * binding a maintainer of the reference would add on top of it.
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures; `void* stream` is a hipStream_t.
 *   - every function returns 0 on success or a negative MPMHIP_E* code; it never throws.
 *     `mpmhip_last_error()` returns a human-readable message for the last failure on that ctx
 *     (reference behaviour: TC_ASSERT/TC_ERROR abort, e.g. src/mpm.cpp:41-42,154,162,775).
 *   - host buffers are caller-owned; `*_dev` variants take device pointers.
 *   - one ctx per GPU; calls on one ctx must be serialised by the caller (reference: not
 *     re-entrant either).
 *   - 3x3 matrices are row-major float[9]; `B` is the reference's `apic_b` (sign/units of
 *     src/transfer.cpp:898-903: B = sum_i w_i v_i (x_p - x_i)^T / dx).
 */
#ifndef MPMHIP_H
#define MPMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPMHIP_ABI_VERSION 3  /* 2: mpmhip_async_config gained left_boundary; 3: the native data plane of tiled runs (mpmhip_tiled_*, mpmhip_comm_*) */

enum {
  MPMHIP_OK = 0,
  MPMHIP_EINVAL = -1,    /* bad argument / unsupported configuration */
  MPMHIP_ENOMEM = -2,    /* device or host allocation failed, or capacity exceeded */
  MPMHIP_EHIP = -3,      /* a HIP runtime call failed (message in last_error) */
  MPMHIP_ECAPACITY = -4, /* more particles / active blocks than the ctx was created for */
  MPMHIP_ENOTIMPL = -5
};

/* material ids == y component of MPMParticle::get_debug_info() (src/particles.cpp:157..839);
 * names are the factory aliases of TC_REGISTER_MPM_PARTICLE (src/particles.cpp:849-856). */
enum {
  MPMHIP_VISCO = 1,
  MPMHIP_SNOW = 2,
  MPMHIP_LINEAR = 3,
  MPMHIP_JELLY = 4,
  MPMHIP_WATER = 5,
  MPMHIP_SAND = 6,
  MPMHIP_VON_MISES = 7,
  MPMHIP_ELASTIC = 8
};

#define MPMHIP_NPARAM 16
/* per-group parameter row float[16] (a "group" = one add_particles() call of the reference,
 * src/mpm.cpp:77-148: same type, same material constants, mass = vol*density):
 *   [0] mass [1] vol
 *   snow      [2] mu_0 [3] lambda_0 [4] hardening [5] theta_c [6] theta_s [7] min_Jp [8] max_Jp ; aux = Jp
 *   linear    [2] mu   [3] lambda
 *   jelly     [2] mu   [3] lambda
 *   water     [2] k    [3] gamma                                                             ; aux = j
 *   sand      [2] mu_0 [3] lambda_0 [4] alpha [5] cohesion [6] beta                           ; aux = logJp
 *   von_mises [2] mu_0 [3] lambda_0 [4] yield_stress
 *   elastic   [2] mu_0 [3] lambda_0 [4] E (reported by get_debug_info() only, src/particles.cpp:838-840)
 *   visco     [2] mu_0 [3] lambda_0 [4] visco_nu [5] visco_kappa [6] base_delta_t                ; aux = visco_tau
 */

/* replaces: Config keys read by MPM<dim>::initialize (src/mpm.cpp:26-75) */
typedef struct {
  int32_t res[3];           /* "res": cells per axis (nodes = res+1) */
  float dx;                 /* "delta_x" (python default 1/res[0], scripts/async/async_mpm.py:40-41) */
  float dt;                 /* "base_delta_t" * "dt_multiplier" */
  float gravity[3];         /* "gravity" (default (0,-10,0)) */
  int32_t particle_gravity; /* "particle_gravity" (default 1): gravity added to particles before P2G */
  float apic_damping;       /* "apic_damping" */
  float rpic_damping;       /* "rpic_damping" */
  int32_t clean_boundary;   /* "clean_boundary" (default 1): src/mpm.cpp:563-565 */
  /* analytic level set (replaces set_levelset(DynamicLevelSet)): up to 8 half-spaces
   * phi(x) = n.x + d in world units (|n| = 1), combined with min(); phi<0 inside the solid */
  int32_t n_planes;
  float planes[8][4];
  float friction;           /* levelset friction code: -1 sticky, <=-2 slip, >=0 separate (README.md:326-330) */
  int64_t max_particles;    /* capacity of the SoA pools (reference: < 2^25, src/mpm.cpp:773-775) */
  int64_t max_blocks;       /* capacity of the active-block table; 0 = choose from max_particles */
  int32_t device;           /* HIP device ordinal */
  int32_t reorder_interval; /* "reorder_interval" (src/mpm.cpp:45,811-813): physical reorder of the particle records
                               into sorted order every this many substeps; 0 = never (the sorted INDEX is rebuilt
                               every substep regardless) */
  int32_t particle_collision; /* "particle_collision" (default 0): particle_collision_resolution after G2P, src/mpm.cpp:414-426,566-569 */
  int32_t discard_apic_b;   /* 1 = G2P does not store apic_b: it lives on folded into the P2G affine matrix
                               A = stress*(-4 inv_dx dt) + apic_b*(4 m) (src/transfer.cpp:521-522), which is the state
                               the next substep consumes (-48 of 180 stored bytes per particle-step).  download(B)
                               then recovers apic_b = (A - stress(F) S)/(4 m) on demand, to ~1e-5..1e-4 relative */
  int32_t generic_path;     /* config "optimized" = false: the reference's generic transfer path (src/transfer.cpp:193-278,
                             * 585-687).  Same arithmetic as the optimised path here (both go through the same kernels); what
                             * differs in the reference is the position clamp into [0, res - eps] after the advection
                             * (:668-670), which this flag turns on */
  int32_t deterministic;    /* 1 = bitwise reproducible runs: every cell's particles are put in ascending creation id behind the sort
                             * (the in-cell ranks of the counting sort come from atomics, so without it the summation order of P2G's
                             * per-cell sums — and the last bits of everything downstream — differs from run to run).  Two runs of one
                             * scene, the packed and the per-block G2P walk, the three- and the four-launch sort, and a tiled job over any
                             * wire then give identical bits; a K-rank job differs from the one-ctx run only by how the <= 8 block tiles of
                             * a halo node are grouped into rank partials.  Costs one more launch per sort (bench.py reports it).
                             * Not covered: the impulse / torque sums of CPIC rigid bodies and calculate_energy (float atomics).
                             * The reference's sort key is unique for the same purpose: (offset >> 5) << 25 | i, src/mpm.cpp:785-795.
                             * env MPMHIP_DETERMINISTIC=0/1 overrides */
  int32_t reserved[2];
} mpmhip_config;

typedef struct mpmhip_ctx mpmhip_ctx;

/* particle fields for upload/download; element = one particle's record of `width` scalars */
enum {
  MPMHIP_F_X = 0,   /* float[3]  position (world units)                MPMParticle::pos      */
  MPMHIP_F_V = 1,   /* float[3]  velocity                              get_velocity()        */
  MPMHIP_F_B = 2,   /* float[9]  apic_b                                MPMParticle::apic_b   */
  MPMHIP_F_F = 3,   /* float[9]  elastic deformation gradient          MPMParticle::dg_e     */
  MPMHIP_F_AUX = 4, /* float[1]  Jp | j | logJp (material state)                             */
  MPMHIP_F_GID = 5, /* int32[1]  group id                                                    */
  MPMHIP_F_ID = 6,  /* int32[1]  creation index (MPMParticle::id semantics for download ordering) */
  MPMHIP_F_STATES = 7 /* int32[1] CPIC colour word (MPMParticle::states, src/particles.h): 2 bits per rigid body */
};

uint32_t mpmhip_abi_version(void);

/* lifecycle — replaces create_instance<Simulation3D>("mpm") + MPM<3>::initialize (src/mpm.cpp:26-75) */
int mpmhip_create(const mpmhip_config *cfg, mpmhip_ctx **out);
void mpmhip_destroy(mpmhip_ctx *ctx);
const char *mpmhip_last_error(const mpmhip_ctx *ctx); /* ctx may be NULL: last create() failure */
int mpmhip_set_stream(mpmhip_ctx *ctx, void *hip_stream); /* NULL = the ctx's own stream */
/* mpmhip_config.deterministic of a live ctx (from the next sort on; no reference counterpart: its scalar path has one order) */
int mpmhip_set_deterministic(mpmhip_ctx *ctx, int32_t enabled);
/* replaces Simulation::set_levelset(DynamicLevelSet) (scripts/async/async_mpm.py:119-127) for analytic half-spaces */
int mpmhip_set_levelset(mpmhip_ctx *ctx, int32_t n_planes, const float *planes /* [n][4] */, float friction);

/* general analytic level set: union of up to MPMHIP_MAX_SHAPES solids, phi = min over shapes, negative inside a
 * solid; replaces the LevelSet the scene scripts build with add_plane / add_sphere / add_cuboid
 * (e.g. scripts/mls-cpic/goo_blocks.py:17-20, scripts/async/sand.py:34-37).  World units.
 *   type 0 plane  p = {nx, ny, nz, d}            phi = n.x + d            (|n| = 1)
 *   type 1 sphere p = {cx, cy, cz, r}            solid ball;  inside_out = 1: the ball is the FREE space
 *   type 2 cuboid p = {lo x,y,z, hi x,y,z}       solid box;   inside_out = 1: the box is a container */
#define MPMHIP_MAX_SHAPES 16
typedef struct {
  int32_t type, inside_out;
  float p[6];
} mpmhip_shape;
int mpmhip_set_levelset_shapes(mpmhip_ctx *ctx, int32_t n, const mpmhip_shape *shapes, float friction);
/* MPM<dim>::rigid_body_levelset_collision (src/mpm_rigid_body.cpp:347-387; config key rigid_body_levelset_collision, run between
 * normalize_grid and the grid boundary condition, src/mpm.cpp:535-538): every boundary particle of a rigid body below the level
 * set gives its body a normal impulse (restitution) and a Coulomb friction impulse (friction0) at once, so that the next boundary
 * particle sees the changed velocity — in the order of the reference's sorted particle list, which the library keeps for the
 * boundary particles (sort key of src/mpm.cpp:785-790, ties by the position in the previous order). */
int mpmhip_set_rigid_levelset_collision(mpmhip_ctx *ctx, int32_t enabled);
/* MPM<3>::apply_dirichlet_boundary_conditions (src/mpm.cpp:401-412; runs behind the grid boundary condition when the config key
 * dirichlet_boundary_radius is > 0, :541-544): grid nodes with y > 0.525 are held at rest (the reference's 3D form ignores the
 * radius and hard-codes the plane) */
int mpmhip_set_dirichlet(mpmhip_ctx *ctx, int32_t enabled);
/* time-dependent level set — replaces DynamicLevelSet::initialize(t0, t1, levelset(t0), levelset(t1)) +
 * Simulation::set_levelset as the python driver calls them before every frame (scripts/async/async_mpm.py:119-127,
 * e.g. the shrinking container of scripts/async/balls.py:38-49): two key frames blended linearly in time; at time t
 * phi = lerp(phi0, phi1), the normal is the normalised lerp of the two gradients, and the grid boundary condition
 * uses boundary_velocity = -(phi1 - phi0)/(t1 - t0) n delta_x (src/mpm.cpp:323-342).  t = the ctx's current time. */
int mpmhip_set_levelset_keyframes(mpmhip_ctx *ctx, float t0, float t1, int32_t n0, const mpmhip_shape *shapes0,
                                  int32_t n1, const mpmhip_shape *shapes1, float friction);

/* A ctx holds at most MPMHIP_MAX_GROUPS groups (k_g2p mirrors the whole group table in LDS). */
#define MPMHIP_MAX_GROUPS 64
/* particles — replaces MPM<3>::add_particles (src/mpm.cpp:77-270) with caller-generated samples.
 * add_group returns the group id (>=0) or a negative error. F/B/aux may be NULL (identity/0/material default). */
int mpmhip_add_group(mpmhip_ctx *ctx, int32_t material, const float params[MPMHIP_NPARAM]);
int mpmhip_add_particles(mpmhip_ctx *ctx, int32_t group, int64_t n, const float *x, const float *v,
                         const float *F, const float *B, const float *aux);
int64_t mpmhip_num_particles(mpmhip_ctx *ctx); /* synchronises; <0 on error */
int mpmhip_download(mpmhip_ctx *ctx, int32_t field, void *dst, int64_t n_capacity); /* returns n written */
int mpmhip_upload(mpmhip_ctx *ctx, int32_t field, const void *src, int64_t n);

/* time stepping — replaces MPM<3>::step / substep (src/mpm.cpp:428-450, 452-575) */
int mpmhip_substep(mpmhip_ctx *ctx);                 /* one substep, asynchronous on the ctx stream */
int mpmhip_run_substeps(mpmhip_ctx *ctx, int32_t n); /* n substeps back to back */
int mpmhip_step(mpmhip_ctx *ctx, float dt);          /* step(dt): dt<0 => one substep; else while (t + base_dt < request_t) substep */
double mpmhip_current_time(const mpmhip_ctx *ctx);   /* get_current_time() */
int mpmhip_synchronize(mpmhip_ctx *ctx);

/* phase-level entry points (parity tests) — each replaces the named reference function */
int mpmhip_sort(mpmhip_ctx *ctx);        /* sort_particles_and_populate_grid  src/mpm.cpp:770-918 (+ clear_boundary_particles :582-633) */
int mpmhip_p2g(mpmhip_ctx *ctx);         /* rasterize_optimized               src/transfer.cpp:361-581 */
int mpmhip_grid_update(mpmhip_ctx *ctx); /* normalize_grid_and_apply_external_force + apply_grid_boundary_conditions  src/mpm.cpp:277-372 */
int mpmhip_g2p(mpmhip_ctx *ctx);         /* resample_optimized                src/transfer.cpp:702-970 */

/* dense grid views, node-major float[(rx+1)(ry+1)(rz+1)][4], z fastest.
 *   which=0: (m*v, m) accumulated by P2G;  which=1: (v, m) after grid_update.
 * upload_grid installs a dense (v, m) field as the input of the next mpmhip_g2p (needs a prior sort). */
int mpmhip_download_grid(mpmhip_ctx *ctx, int32_t which, float *dst);
int mpmhip_upload_grid(mpmhip_ctx *ctx, const float *src);

/* whole-state snapshots — replaces the TC_IO serialization behind general_action "save" / "load"
 * (src/mpm.cpp:940-960, src/mpm.h:38-54,134-169).  The blob holds the raw particle records (with the P2G affine
 * matrices), the group table and the clocks; it loads into a ctx of the same grid with enough capacity, which then
 * continues exactly where the saved run stopped.  Level set and config are NOT part of it (the reference re-reads
 * them from the scene script too). 
 * A scene with rigid bodies: the blob also carries the bodies' records (pose, velocities, mass properties) and the joints;
 * meshes, boundary particles and scripts come from the scene again — add the same bodies, in the same order, before loading. */
int64_t mpmhip_snapshot_size(mpmhip_ctx *ctx);
int mpmhip_snapshot_save(mpmhip_ctx *ctx, void *dst, size_t capacity);
int mpmhip_snapshot_load(mpmhip_ctx *ctx, const void *src, size_t size);

/* replaces MPM<3>::calculate_energy (src/mpm.cpp:1078-1110; general_action "calculate_energy", :936-938): sorts and
 * rasterizes, then kinetic = sum over grid nodes of 1/2 m |v|^2, potential = sum of MPMParticle::potential_energy()
 * (defined for linear, jelly, elastic: src/particles.cpp:323-327,400-407,785-796).  Returns MPMHIP_ENOTIMPL (with a
 * valid *kinetic) if a live particle is of another type, as the reference aborts there.  Synchronises.
 * On a ctx of a tiled job (mpmhip_tiled_setup, wires RCCL / IPC) the call is COLLECTIVE and returns the energy of the whole job:
 * halo exchange after the rasterization, every node's kinetic energy counted by the lowest rank that holds mass on it, the ranks'
 * shares summed by mpmhip_tiled_reduce (MPMHIP_WIRE_LOCAL: mpmhip_calculate_energy_group). */
int mpmhip_calculate_energy(mpmhip_ctx *ctx, double *kinetic, double *potential);

/* replaces general_action "delete_particles_inside_level_set" (src/mpm.cpp:962-974): deletes every particle whose
 * level-set value at its position is negative (the level set last given to mpmhip_set_levelset[_shapes]);
 * *deleted = how many.  The next sort rebuilds the keys.  Synchronises. */
int mpmhip_delete_particles_inside_level_set(mpmhip_ctx *ctx, int64_t *deleted);

/* frame output — replaces MPM<dim>::write_bgeo / write_partio (src/mpm.h:333-337, src/visualize.cpp:17-100; called
 * by visualize(), src/visualize.cpp:156-164) including the Partio encoder behind it (external/partio/src/io/BGEO.cpp:
 * 57-194, Houdini .bgeo version 5, big-endian).  Byte-identical to the reference's file for the same particle state:
 * attributes position, type, index, limit[3], v and — verbose (config "verbose_bgeo") — m, boundary_normal[3], debug[3],
 * states, boundary_distance, near_boundary, apic_frobenius_norm; particles in ascending id.  The rows are gathered,
 * interleaved and byte-swapped on the device; the host adds header and trailer.  Fields this library does not track
 * (no rigid bodies, no async stepping) hold the reference's constructor values (src/particles.h:92-99).
 * mpmhip_bgeo_size: bytes of the file for the current state.  mpmhip_bgeo_encode: the file image into caller memory
 * (*written = bytes; MPMHIP_ECAPACITY if `capacity` is too small).  mpmhip_write_bgeo: the same to `path` (plain
 * .bgeo as the reference writes; a ".gz" suffix is MPMHIP_ENOTIMPL).  A tiled ctx encodes its own rank's particles.
 * All three synchronise. */
int mpmhip_bgeo_size(mpmhip_ctx *ctx, int32_t verbose, size_t *bytes);
int mpmhip_bgeo_encode(mpmhip_ctx *ctx, int32_t verbose, void *dst, size_t capacity, size_t *written);
int mpmhip_write_bgeo(mpmhip_ctx *ctx, const char *path, int32_t verbose);

/* profiling — replaces TC_PROFILE / TC_PROFILE_TPE scoped timers (src/mpm.cpp:464-572).
 * level 0: off.  level 1: hipEvents bracket each phase of every substep on the ctx stream (six records per
 * substep; each record costs ~5 us of idle GPU).  level 2 / 3: only k_g2p / only k_p2g is bracketed (two records),
 * for timing the dominant kernel inside a throughput measurement.  level 4: the three parts of a tiled substep are
 * bracketed — substep_begin (sort, boundary P2G, halo pack) -> "p2g", substep_interior -> "grid", substep_end -> "g2p" —
 * with no record INSIDE a part: what one rank computes per substep when nothing is being profiled.
 * mpmhip_profile writes a JSON object: {"substeps":N,"particles":n,"active_blocks":a,"rank_mode":m,
 *   "phases":{"sort":ms,"p2g":ms,"exchange":ms,"grid":ms,"g2p":ms}}  (totals since the last reset; rank_mode = the
 *   in-cell ranking path the next sort will take: 0 per-run atomics, 1 per-batch LDS hash; "exchange" is
 *   the gap between substep_begin and substep_end of a tiled run; phases not bracketed at the level stay 0). */
int mpmhip_set_profiling(mpmhip_ctx *ctx, int32_t level);
/* levels 2 / 3: bracket the kernel only in every `every`-th substep (default 1 = all).  "substeps" of mpmhip_profile then
 * counts the bracketed ones, so phase time / substeps stays the mean launch duration — with a quarter of the ~5 us event
 * bubbles inside a throughput measurement at every = 4. */
int mpmhip_set_profile_sampling(mpmhip_ctx *ctx, int32_t every);
int mpmhip_profile(mpmhip_ctx *ctx, char *json, size_t cap);
int mpmhip_profile_reset(mpmhip_ctx *ctx);

/* ---------------------------------------------------------------------------------------------------------
 * Multi-GPU tiling (one ctx per GPU, each over the WHOLE grid index space but holding only the particles of its
 * brick).  The reference has no multi-process code; this is the sharding SURVEY §8(e) derives from the stencil
 * support (base..base+2 nodes, src/kernel.h:119-121, src/transfer.cpp:59-63).  The library never communicates:
 * the caller moves the device buffers named here with any transport (RCCL, MPI, hipMemcpyPeer).
 *
 *   partition : `dims[a]` bricks per axis with cell cut planes `cuts[a][0..dims[a]]` (cuts[a][0] = 0,
 *               cuts[a][dims[a]] >= res[a]); rank of brick (px,py,pz) = (px*dims[1] + py)*dims[2] + pz.
 *               A particle belongs to the brick containing its base cell int(x/dx - 0.5) (src/kernel.h:119-121);
 *               it may sit up to `margin` cells outside its rank's brick between two migrations.
 *   halo      : node boxes shared with other ranks.  After P2G, mpmhip_halo_pack() writes this rank's partial
 *               (m v, m) sums on every box to `send`; the caller exchanges send<->recv with rank `peer`;
 *               grid_update then forms every node total as the sum over contributors IN RANK ORDER (own partial
 *               at its rank position), so all holders of a node compute bit-identical values.
 *   migration : mpmhip_leaver_counts -> per-destination counts (synchronises); mpmhip_export_leavers packs the
 *               MPMHIP_MIGRATE_FLOATS-float records grouped by destination rank (ascending) into a device buffer
 *               and removes them locally; mpmhip_import_particles appends received records.
 */
#define MPMHIP_MAX_PARTS 16       /* bricks per axis */
#define MPMHIP_MAX_HALO_BOXES 64
#define MPMHIP_MIGRATE_FLOATS 44  /* RecG(16) + RecP(16) + apic_b(12) : 176 bytes per migrating particle */

typedef struct {
  int32_t lo[3], hi[3]; /* node box [lo, hi) in global node coordinates */
  int32_t peer;         /* rank of the other contributor */
  int32_t reserved;
  void *send;           /* device float4[volume], x-major (z fastest): written by mpmhip_halo_pack */
  const void *recv;     /* device float4[volume]: the peer's partial sums, read by grid_update */
} mpmhip_halo_box;

int mpmhip_set_partition(mpmhip_ctx *ctx, int32_t rank, const int32_t dims[3], const int32_t *cuts_x,
                         const int32_t *cuts_y, const int32_t *cuts_z, int32_t margin);
/* boxes must be sorted by ascending peer; n = 0 disables the halo */
int mpmhip_set_halo(mpmhip_ctx *ctx, int32_t n, const mpmhip_halo_box *boxes);
int mpmhip_halo_pack(mpmhip_ctx *ctx);
/* tiled substep = substep_begin (sort, P2G, halo_pack) ; caller exchanges ; substep_end (grid, G2P) */
int mpmhip_substep_begin(mpmhip_ctx *ctx);
int mpmhip_substep_end(mpmhip_ctx *ctx);
/* exchange/compute overlap: with set_overlap(1), substep_begin only rasterizes the blocks that touch a halo box (then
 * packs); the caller starts the exchange asynchronously and calls substep_interior (P2G, grid and G2P of everything
 * that cannot touch a halo node) while it is on the wire, waits for it, and calls substep_end (the boundary part).
 * substep_interior is a no-op when the overlap is off; substep_end runs it if the caller skipped it. */
int mpmhip_set_overlap(mpmhip_ctx *ctx, int32_t enabled);
/* The per-substep loop of a tiled rank in native code: for each of n substeps
 *     substep_begin;  exchange(user, MPMHIP_EXCHANGE_START);  substep_interior;  exchange(user, MPMHIP_EXCHANGE_WAIT);  substep_end
 * — the caller supplies only the transport (START: launch the all-to-all of the halo boxes behind the work enqueued so far,
 * e.g. on a side stream; WAIT: make the ctx's stream wait for it).  A non-zero return of the callback ends the loop and is
 * returned (negative values are taken as they are, positive ones as MPMHIP_EINVAL).  `until_migration` > 0 stops after that
 * many substeps even if n is larger (the caller runs its migration and calls again); returns the number of substeps run. */
enum { MPMHIP_EXCHANGE_START = 0, MPMHIP_EXCHANGE_WAIT = 1 };
typedef int32_t (*mpmhip_exchange_fn)(void *user, int32_t phase);
int64_t mpmhip_tiled_run(mpmhip_ctx *ctx, int64_t n, int64_t until_migration, mpmhip_exchange_fn exchange, void *user);
int mpmhip_substep_interior(mpmhip_ctx *ctx);
/* counts[world]: live particles whose base cell now lies in another rank's brick.  Also raises MPMHIP_ECAPACITY
 * (sticky) if a particle is more than `margin` cells outside this rank's brick. */
int mpmhip_leaver_counts(mpmhip_ctx *ctx, int32_t world, int64_t *counts);
/* the same pass also yields the bounding box [lo, hi) of the base cells of all live particles (lo > hi: none) and
 * the speed of the fastest one in cells per substep (max |v|_inf dt / dx; may be NULL), so one synchronisation per
 * migration serves the counts, the halo-box clipping and the schedule of the next migration (a particle needs
 * margin / speed substeps to cross the margin) */
int mpmhip_migration_scan(mpmhip_ctx *ctx, int32_t world, int64_t *counts, int32_t lo[3], int32_t hi[3],
                          float *max_cells_per_substep);
/* packs every leaver (n_total = sum of the counts just returned) into dev_records, grouped by destination */
int mpmhip_export_leavers(mpmhip_ctx *ctx, int32_t world, const int64_t *counts, void *dev_records);
int mpmhip_import_particles(mpmhip_ctx *ctx, int64_t n, const void *dev_records);
/* ---- Multi-GPU tiling, the native data plane (csrc/tiled_api.h).  With the calls above the CALLER plans the halo boxes, owns
 * the buffers and moves them (the Python / gloo path of taichi_mpm_amd/tiled.py, kept for the CPU tests).  With the calls below
 * the LIBRARY does all of it: it derives the halo boxes from the partition (box R∩S = intersection of the two ranks' node
 * boxes, clipped to the occupied part of the grid — the same on both sides), owns send / receive buffers sized for the worst
 * case (no re-allocation when the boxes follow the particles), runs the per-substep loop, the halo exchange, the migration
 * (scan, table all-gather, record exchange, import), the re-planning and the migration schedule — nothing returns to the
 * host language between mpmhip_tiled_advance's first and last substep.  SURVEY section 8(e) collectives (1)-(3).
 * Wires:
 *   MPMHIP_WIRE_RCCL   ncclGroupStart / per-neighbour ncclSend + ncclRecv / ncclGroupEnd on a side stream of the ctx (fenced to
 *                      the ctx stream by events; on the ctx stream itself when the overlap split is off), ncclAllGather for
 *                      the migration table, grouped ncclSend / ncclRecv for the records.  librccl is dlopen'ed (env
 *                      MPMHIP_RCCL_LIB names it): libmpmhip does not link it.  The caller only carries the ncclUniqueId from
 *                      rank 0 to the other ranks (mpmhip_comm_unique_id / mpmhip_comm_init).
 *   MPMHIP_WIRE_IPC    no collective at all: every rank maps every other rank's receive arena (hipIpcGetMemHandle /
 *                      hipIpcOpenMemHandle); k_halo_pack writes each box straight into the peer's receive buffer (double
 *                      buffered by substep parity) and publishes the substep's epoch in the peer's flag word; the peer's
 *                      stream polls that word (a one-wave kernel with a bounded wait: sticky error instead of a hang) before
 *                      its grid kernel reads the box.  Migration rows and records travel the same way.  The 64-byte handles
 *                      are all-gathered by the caller (mpmhip_tiled_ipc_handle / _connect), or through the ctx's RCCL
 *                      communicator when it has one (_connect with handles == NULL).
 *   MPMHIP_WIRE_LOCAL  the IPC wire between ctx of ONE process (virtual ranks on one device: tests, bench.py --virtual):
 *                      plain pointers instead of mapped ones, driven by mpmhip_tiled_advance_group.
 *   MPMHIP_WIRE_LOCAL_RCCL  a LOCAL job whose HALO BOXES travel through RCCL all the same — the one-GPU pre-flight of the
 *                      MPMHIP_WIRE_RCCL exchange: every ctx of the job has its own ONE-rank communicator
 *                      (mpmhip_comm_init(ctx, id, 0, 1) before mpmhip_tiled_setup), every box is an ncclSend to the rank itself
 *                      whose ncclRecv is aimed at the box's place in the PEER ctx's receive buffer: the same ncclGroupStart / Send
 *                      + Recv per box / ncclGroupEnd, on the same side stream behind the same two event fences when the substep is
 *                      split, with the one receive buffer of that wire — under real RCCL kernels next to the substep's own.
 *                      Migration rows, records and reductions travel as on MPMHIP_WIRE_LOCAL.  The ranks must share one stream.
 * A ctx with a native plan refuses mpmhip_set_halo / mpmhip_tiled_run with a callback, and vice versa. */
enum { MPMHIP_WIRE_RCCL = 1, MPMHIP_WIRE_IPC = 2, MPMHIP_WIRE_LOCAL = 3, MPMHIP_WIRE_LOCAL_RCCL = 4 };
#define MPMHIP_COMM_ID_BYTES 128   /* sizeof(ncclUniqueId) */
#define MPMHIP_IPC_HANDLE_BYTES 64 /* sizeof(hipIpcMemHandle_t) */
typedef struct {
  int32_t rank, world;
  int32_t dims[3];             /* bricks per axis; world = dims[0] dims[1] dims[2] */
  int32_t margin;              /* cells a particle may sit outside its brick between two migrations */
  int32_t clip_lo[3], clip_hi[3]; /* node box the halo boxes are clipped to (occupied part of the grid + slack); the library
                                  * re-wraps it around the particles at every migration (slack max(8, 4 margin + 4) cells) AND, from a
                                  * planning scan in front of the job's first substep on, cuts every rank's node box to that rank's OWN
                                  * occupancy (every rank's particle bounds travel in the migration table): ranks whose particles cannot
                                  * meet before the next check exchange nothing */
  int32_t migrate_interval;    /* > 0: a migration every that many substeps (<= margin); 0: the CFL interval (= margin)
                                  * stretched by the measured top speed, up to migrate_cap substeps */
  int32_t migrate_cap;         /* 0: 64 */
  int32_t wire;                /* MPMHIP_WIRE_* */
  int32_t overlap;             /* boundary / interior split of the substep (mpmhip_set_overlap) */
  int64_t inbox_records;       /* capacity of this rank's migration inbox in records; 0: max(65536, capacity / 8).  Every rank learns
                                * every inbox's capacity with the migration table and all refuse together when one would overflow */
} mpmhip_tiled_config;
/* rank 0: a fresh ncclUniqueId; every rank: ncclCommInitRank on the ctx's device (collective over the ranks) */
int mpmhip_comm_unique_id(uint8_t id[MPMHIP_COMM_ID_BYTES]);
int mpmhip_comm_init(mpmhip_ctx *ctx, const uint8_t id[MPMHIP_COMM_ID_BYTES], int32_t rank, int32_t world);
int mpmhip_comm_destroy(mpmhip_ctx *ctx);
/* loopback check of the binding: world-sized all-gather + a grouped send / receive ring (to self when world == 1) on device
 * buffers, verified on the host; collective over the ranks */
int mpmhip_comm_selftest(mpmhip_ctx *ctx);
/* partition + plan + buffers; replaces mpmhip_set_partition / mpmhip_set_halo.  Synchronises. */
int mpmhip_tiled_setup(mpmhip_ctx *ctx, const mpmhip_tiled_config *cfg, const int32_t *cuts_x, const int32_t *cuts_y,
                       const int32_t *cuts_z);
int mpmhip_tiled_ipc_handle(mpmhip_ctx *ctx, uint8_t handle[MPMHIP_IPC_HANDLE_BYTES]);
int mpmhip_tiled_ipc_connect(mpmhip_ctx *ctx, const uint8_t *handles /* [world][64], own entry ignored; NULL: via RCCL */);
int mpmhip_tiled_connect_local(mpmhip_ctx *const *ctxs, int32_t n /* = world, ctxs[r] = rank r */);
/* n substeps of this rank, migrations included when they are due (collective over the ranks: every rank calls it with the
 * same n).  Returns n or a negative error code. */
int64_t mpmhip_tiled_advance(mpmhip_ctx *ctx, int64_t n);
/* the same for all ranks of a MPMHIP_WIRE_LOCAL job: per substep begin of every rank, [interior of every rank,] end of every rank */
int64_t mpmhip_tiled_advance_group(mpmhip_ctx *const *ctxs, int32_t n_ctx, int64_t n);
/* SURVEY section 8(e) collective (3) — a few scalars over all ranks of a tiled job, inside the library: in-place all-reduce of n <=
 * MPMHIP_REDUCE_MAX_VALUES doubles (MPMHIP_WIRE_RCCL: ncclAllReduce on a device row; MPMHIP_WIRE_IPC: every rank writes its row into
 * every rank's table, publishes an epoch, waits for the world's and reduces the table in rank order — the bit-identical result on
 * every rank).  Collective: every rank calls it with the same n and op.  Synchronises. */
enum { MPMHIP_REDUCE_SUM = 0, MPMHIP_REDUCE_MAX = 1, MPMHIP_REDUCE_MIN = 2 };
#define MPMHIP_REDUCE_MAX_VALUES 16
int mpmhip_tiled_reduce(mpmhip_ctx *ctx, double *values, int32_t n, int32_t op);
/* the same for all ranks of a MPMHIP_WIRE_LOCAL job: values = [n_ctx][n], every row receives the result */
int mpmhip_tiled_reduce_group(mpmhip_ctx *const *ctxs, int32_t n_ctx, double *values, int32_t n, int32_t op);
/* MPM<dim>::calculate_energy (src/mpm.cpp:1078-1110) of the WHOLE job of a MPMHIP_WIRE_LOCAL group (on the other wires
 * mpmhip_calculate_energy of a tiled ctx is itself the collective): sort + rasterize + halo exchange on every rank, grid kinetic
 * energy with every node counted by the lowest rank that holds mass on it, the particles' potential energy, summed over the ranks */
int mpmhip_calculate_energy_group(mpmhip_ctx *const *ctxs, int32_t n_ctx, double *kinetic, double *potential);
/* out = {live particles, active blocks, sticky error word (OR over the ranks: bit 0 block table full, 1 a particle beyond the margin,
 * 2 distance-field pages, 4 a peer's epoch timed out), particles migrated so far} of the whole job; collective; synchronises */
int mpmhip_tiled_totals(mpmhip_ctx *ctx, int64_t out[4]);
int mpmhip_tiled_totals_group(mpmhip_ctx *const *ctxs, int32_t n_ctx, int64_t out[4]);
/* out = {substeps run, substep of the next migration, particles migrated out so far, migrations, re-plans, halo boxes,
 *        halo nodes (float4) per exchange, wire} */
int mpmhip_tiled_state(mpmhip_ctx *ctx, int64_t out[8]);
/* the current plan: boxes sorted by peer (send / recv = the library's buffers); returns the number of boxes */
int32_t mpmhip_tiled_plan(mpmhip_ctx *ctx, int32_t capacity, mpmhip_halo_box *out);
/* cell bounding box [lo, hi) of the active 4^3-cell blocks of the last sort (lo > hi when there are none);
 * synchronises.  Lets the caller clip the halo boxes to the occupied part of the grid. */
int mpmhip_active_bounds(mpmhip_ctx *ctx, int32_t lo[3], int32_t hi[3]);
int64_t mpmhip_num_slots(mpmhip_ctx *ctx);       /* slots in use (live + dead) — capacity pressure */
int mpmhip_request_compaction(mpmhip_ctx *ctx);  /* physical reorder + drop of dead slots at the next sort */
/* Grows the particle capacity of a live ctx IN PLACE to at least max_particles (never shrinks): the record arrays are
 * re-allocated and copied, an auto-sized block table (max_blocks = 0 at creation) grows with them.  Everything else
 * stays — groups, level set, clocks, stream, rigid bodies and joints, the async block table, partition and halo boxes —
 * so a scene that keeps adding particles (the reference's ParticleAllocator pool grows on demand,
 * src/particle_allocator.h:36-60) needs no max_particles up front, with or without bodies.  Synchronises; the next
 * substep sorts from scratch.  Not between substep_begin and substep_end. */
int mpmhip_reserve(mpmhip_ctx *ctx, int64_t max_particles);
int64_t mpmhip_capacity(mpmhip_ctx *ctx);

/* ---- AsyncMPM, first half — replaces AsyncMPM<dim>::update_dt_limits (src/async/async_mpm.cpp:90-164) and the limit
 * attributes AsyncMPM<dim>::visualize writes (src/async/async_visualize.cpp:17-26).  A scheduler block is the reference's
 * SPGrid block of 4 x 4 x 8 nodes holding the particle's base node.  Per block: strength_dt_limit =
 * int(strength_dt_mul * min get_allowed_dt(dx) / unit_delta_t) (per-material sound-speed bound, src/particles.cpp),
 * cfl_dt_limit = int(cfl_dt_mul * dx / unit_delta_t / sqrt(max |v|^2)), continuous_dt_limit = the power of two that
 * tracks min(cfl, strength, max_units) under the reference's halving / doubling rule.  The per-block reduction runs on
 * the device, the block state machine on the host.  After mpmhip_async_update_dt_limits the `limit` attribute of
 * mpmhip_write_bgeo carries (continuous, strength, cfl) of each particle's block instead of (1, 1, 1).
 * The stepping itself (AsyncMPM<dim>::advance / step) is the second half, mpmhip_async_begin / _step below. */
typedef struct {
  float unit_delta_t;     /* config "unit_delta_t", default 1e-6 (src/async/async_mpm.cpp:24) */
  int64_t max_units;      /* "max_units", default 8192 */
  float cfl_dt_mul;       /* "cfl_dt_mul", default 1 */
  float strength_dt_mul;  /* "strength_dt_mul", default 1 */
  int32_t left_boundary;  /* "left_boundary", default 0: the scheduler blocks whose corner lies in x <= 0.2 of the domain follow
                           * the SMALLEST step in use (src/async/async_mpm.cpp:43-53, 155-163) */
} mpmhip_async_config;
int mpmhip_async_enable(mpmhip_ctx *ctx, const mpmhip_async_config *cfg);
int mpmhip_async_update_dt_limits(mpmhip_ctx *ctx);
int64_t mpmhip_async_blocks(mpmhip_ctx *ctx, int64_t capacity, int32_t *corner_node /* [n][3] */, int64_t *strength,
                            int64_t *cfl, int64_t *continuous, int64_t *count, int64_t min_max_delta_t_int[2]);
int mpmhip_async_set_time_int(mpmhip_ctx *ctx, int64_t current_t_int);
/* dense block table: nb[3] = blocks per axis, block b = (bx nb[1] + by) nb[2] + bz; returns the number of blocks (with
 * capacity < that number nothing is written: size query) */
int64_t mpmhip_async_table(mpmhip_ctx *ctx, int32_t nb[3], int64_t capacity, int64_t *strength, int64_t *cfl,
                           int64_t *continuous, int64_t *count);
/* ---- AsyncMPM, second half — replaces AsyncMPM<dim> itself (TC_IMPLEMENTATION(Simulation3D, AsyncMPM3D, "async_mpm"),
 * src/async/async_mpm.cpp:423-427): initialize (:13-55), add_particles (:57-75), update_dt_limits (:90-253), advance
 * (:255-373), step (:380-421), and the particle list of visualize (src/async/async_visualize.cpp:86-96).
 * The particle pools and backup pools of every scheduler block live in a device-resident store of containers; an advance
 * gathers its working set (first copy of an id wins: smaller-step neighbours' pools, then the pools of the level's blocks,
 * then larger-step neighbours' backups, each in the reference's block order) into the ctx's records with index kernels,
 * runs ONE ordinary substep with dt = unit_delta_t * limit, and files the results back.  Block tables and the level walk
 * are host code as in the reference; no particle data crosses the host boundary during stepping.  The ctx must keep apic_b
 * (discard_apic_b = 0) and hold neither rigid bodies nor a partition.
 *   begin            after mpmhip_create; takes the AsyncMPM config keys
 *   pool_particles   after every mpmhip_add_particles: the new particles move from the ctx's records to their blocks' pools
 *   step(dt)         AsyncMPM::step; dt < 0 (the synchronous single substep of the base class) is refused
 *   load_pools       every container of every particle pool becomes the ctx's records (with its pool block's limits for
 *                    the frame's `limit` attribute), so that mpmhip_write_bgeo / download / calculate_energy / snapshots
 *                    see the whole state; the next step() ignores the records (the pools are the state)
 *   state            out = {current_t_int, update_counter, min_delta_t_int, max_delta_t_int, live containers, store size
 *                    incl. freed containers, compactions so far, step_counter}
 *   block_times      particle_t / backup_t / local_min_dt_limit of every block of the dense table (size query as above)
 *   download_pools   rows of 27 floats {x3, v3, F9, apic_b9, aux, gid bits, id bits} + the pool block of every container
 *                    (an id can occur in more than one pool, as in the reference); rows == NULL: the count */
int mpmhip_async_begin(mpmhip_ctx *ctx, const mpmhip_async_config *cfg);
int mpmhip_async_pool_particles(mpmhip_ctx *ctx);
int mpmhip_async_step(mpmhip_ctx *ctx, float dt);
int mpmhip_async_load_pools(mpmhip_ctx *ctx);
int mpmhip_async_state(mpmhip_ctx *ctx, int64_t out[8]);
double mpmhip_async_current_time(const mpmhip_ctx *ctx);
int64_t mpmhip_async_block_times(mpmhip_ctx *ctx, int64_t capacity, int64_t *particle_t, int64_t *backup_t, int64_t *local_min);
int64_t mpmhip_async_download_pools(mpmhip_ctx *ctx, int64_t capacity, float *rows, int32_t *block);
/* host wall time spent so far, ms: {update_dt_limits, of which the neighbour lists, advance, of which the substep (meaningful
 * with sync != 0: the stream is then synchronised behind every substep), store compaction, number of advances} */
int mpmhip_async_profile(mpmhip_ctx *ctx, int32_t sync, double out[6]);
/* snapshots of the asynchronous stepper (the reference serialises every pool and the block table, src/async/async_mpm.h:120-172):
 * groups, block table, clocks and every live container of the store.  Loaded into a ctx of the same grid on which
 * mpmhip_async_begin has run with the same unit_delta_t; level set and configuration come from the scene again. */
int64_t mpmhip_async_snapshot_size(mpmhip_ctx *ctx);
int mpmhip_async_snapshot_save(mpmhip_ctx *ctx, void *dst, size_t capacity);
int mpmhip_async_snapshot_load(mpmhip_ctx *ctx, const void *src, size_t size);
/* bytes of particle data copied between host and device by this ctx so far (add_particles, download, upload, snapshots,
 * download_pools): a stepping call must leave it unchanged */
int64_t mpmhip_host_particle_bytes(const mpmhip_ctx *ctx);
/* plumbing of a host-driven stepper (the round-2 asynchronous stepper used these): drop all particles but keep groups /
 * level set / config; change base_delta_t (the P2G matrices are rebuilt) and the clock — AsyncMPM<dim>::advance / step set
 * both before every MPM<dim>::substep (src/async/async_mpm.cpp:405-408) */
int mpmhip_clear_particles(mpmhip_ctx *ctx);
int mpmhip_set_dt(mpmhip_ctx *ctx, float base_delta_t);
int mpmhip_set_time(mpmhip_ctx *ctx, double current_t);
/* the three clocks of a ctx — current_t, the request_t accumulator of step() (src/mpm.cpp:428-439), the substep counter
 * that phases the physical reorder (src/mpm.cpp:811-813): read / restored by a host layer that re-creates a ctx */
int mpmhip_get_clock(const mpmhip_ctx *ctx, double *current_t, double *request_t, int64_t *substeps);
int mpmhip_set_clock(mpmhip_ctx *ctx, double current_t, double request_t, int64_t substeps);
/* MPMParticle::get_allowed_dt(dx) of n particle states of one material (src/particles.cpp:136-155,254-278,480-490,...) */
int mpmhip_debug_allowed_dt(mpmhip_ctx *ctx, int32_t material, const float params[MPMHIP_NPARAM], int64_t n, const float *F,
                            const float *aux, const float *v, float dx, float *out);

/* ---- MPM<2>: the reference's 2D simulation — replaces the object tc_core.create_simulation2('mpm') returns
 * (TC_IMPLEMENTATION(Simulation2D, MPM2D, "mpm"), src/mpm.cpp:983-986).  MPM<2> runs the GENERIC transfer path
 * (rasterize_optimized = rasterize, resample_optimized = resample: src/transfer.cpp:280-283,697-700; bodies :193-278,
 * :585-687, including the position clamp of :668-670) with all eight particle types in their dim = 2 form.  Its own small
 * object: SoA particles, dense (res+1)^2 grid (2D scenes are the reference's small cases).  Matrices row-major float[4].
 * Level-set shapes are mpmhip_shape read in the plane (z ignored); n1 >= 0 adds the key frame at t1 (DynamicLevelSet). */
typedef struct mpmhip2d_ctx mpmhip2d_ctx;
typedef struct {
  int32_t res[2];
  float dx, dt;
  float gravity[2];
  int32_t particle_gravity;
  float apic_damping, rpic_damping;
  int32_t clean_boundary, particle_collision;
  int64_t max_particles;
  int32_t device;
  int32_t reserved[3];
} mpmhip2d_config;
int mpmhip2d_create(const mpmhip2d_config *cfg, mpmhip2d_ctx **out);
void mpmhip2d_destroy(mpmhip2d_ctx *ctx);
const char *mpmhip2d_last_error(const mpmhip2d_ctx *ctx);
/* MPM<2>::apply_dirichlet_boundary_conditions (src/mpm.cpp:374-399): grid nodes with x < distance_left move with
 * (velocity_left, 0), nodes with x > 1 - distance_right with (velocity_right, 0) — config keys dirichlet_boundary_radius,
 * dirichlet_distance_left / _right, dirichlet_boundary_velocity, dirichlet_boundary_left / _right */
int mpmhip2d_set_dirichlet(mpmhip2d_ctx *ctx, int32_t enabled, float distance_left, float distance_right, float velocity_left,
                           float velocity_right);
int mpmhip2d_set_levelset(mpmhip2d_ctx *ctx, int32_t n0, const mpmhip_shape *shapes0, int32_t n1, const mpmhip_shape *shapes1,
                          float t0, float t1, float friction);
int mpmhip2d_add_group(mpmhip2d_ctx *ctx, int32_t material, const float params[MPMHIP_NPARAM]);
int mpmhip2d_add_particles(mpmhip2d_ctx *ctx, int32_t group, int64_t n, const float *x, const float *v, const float *F,
                           const float *B, const float *aux);
int mpmhip2d_substep(mpmhip2d_ctx *ctx);          /* MPM<2>::substep, src/mpm.cpp:452-575; asynchronous */
int mpmhip2d_step(mpmhip2d_ctx *ctx, float dt);   /* MPM<dim>::step, src/mpm.cpp:428-439 (dt < 0: one substep) */
double mpmhip2d_current_time(const mpmhip2d_ctx *ctx);
int64_t mpmhip2d_num_particles(mpmhip2d_ctx *ctx); /* synchronises */
int64_t mpmhip2d_download(mpmhip2d_ctx *ctx, int64_t capacity, float *x, float *v, float *F, float *B, float *aux, int32_t *gid,
                          int32_t *id);            /* live particles in slot order; NULL outputs are skipped; returns n */
int mpmhip2d_download_grid(mpmhip2d_ctx *ctx, float *grid /* [(res0+1)(res1+1)][3] = (v.x, v.y, m) */);

/* frame output of the 2D simulation — replaces MPM<2>::write_partio (src/visualize.cpp:17-100): the same .bgeo as the 3D
 * entry points above (z = 0), boundary particles of rigid bodies as rows of type 1; with a resident asynchronous stepper the rows
 * are the containers of every particle pool with their block's limits (src/async/async_visualize.cpp:17-26,86-96) */
int mpmhip2d_bgeo_size(mpmhip2d_ctx *ctx, int32_t verbose, size_t *bytes);
int mpmhip2d_bgeo_encode(mpmhip2d_ctx *ctx, int32_t verbose, void *dst, size_t capacity, size_t *written);
int mpmhip2d_write_bgeo(mpmhip2d_ctx *ctx, const char *path, int32_t verbose);
/* snapshots of the 2D simulation — replaces MPM<2>::general_action(action = 'save' | 'load') (src/mpm.cpp:940-960; AsyncMPM<2>:
 * every pool and the block table, src/async/async_mpm.h:120-172): groups, particles, clocks, the rigid bodies' records and joints,
 * the asynchronous stepper's block tables and containers.  Loaded into an object of the same grid whose scene (level set,
 * configuration, the rigid bodies' outlines and scripts, mpmhip2d_async_begin) has been set up again, as for the 3D snapshots. */
int64_t mpmhip2d_snapshot_size(mpmhip2d_ctx *ctx);
int mpmhip2d_snapshot_save(mpmhip2d_ctx *ctx, void *dst, size_t capacity);
int mpmhip2d_snapshot_load(mpmhip2d_ctx *ctx, const void *src, size_t size);

/* ---- AsyncMPM<2> — replaces create_simulation2('async_mpm') (TC_IMPLEMENTATION(Simulation2D, AsyncMPM2D, "async_mpm"),
 * src/async/async_mpm.cpp:423-427): the asynchronous stepper of mpmhip_async_begin / _step above for the 2D simulation
 * object.  A scheduler block is the reference's 2D SPGrid block, 8 x 16 nodes (SPGrid_Mask<5, 5, 2>); the block scheduler
 * (limits, neighbour lists, the action table of an advance) is the same host code as in 3D (csrc/async_sched.h); the pools
 * and backup pools are a device-resident store of 64-byte containers (csrc/k_async2d.h), and no particle data crosses the
 * host boundary while stepping.
 *   _begin            initialize (:13-55); not with rigid bodies.  After it mpmhip2d_step IS the asynchronous step (the
 *                     reference's virtual Simulation::step) and mpmhip2d_substep is refused.
 *   _pool_particles   add_particles' tail (:62-75): the particles just added with mpmhip2d_add_particles move to the pools of
 *                     their blocks (also done by the next _step)
 *   _step             step (:380-421): update_dt_limits, then advance(level) for every power-of-two level due
 *   _load_pools       AsyncMPM::visualize's particle list (src/async/async_visualize.cpp:86-96): all containers of all particle
 *                     pools become the object's particles (mpmhip2d_download / _num_particles then see the whole state, each
 *                     particle at its block's time; an id can occur more than once, as in the reference); returns their number.
 *                     _view_blocks: the pool block of each, in download order.
 *   _state            {current_t_int, update_counter, min_delta_t_int, max_delta_t_int, live containers, store size,
 *                     compactions, steps}
 *   _table            dense block table (block b = bx nb[1] + by): limits, pool sizes, particle_t, backup_t, local_min_dt_limit;
 *                     capacity < number of blocks returns the number only */
int mpmhip2d_async_begin(mpmhip2d_ctx *ctx, const mpmhip_async_config *cfg);
int mpmhip2d_async_pool_particles(mpmhip2d_ctx *ctx);
int mpmhip2d_async_step(mpmhip2d_ctx *ctx, float dt);
int64_t mpmhip2d_async_load_pools(mpmhip2d_ctx *ctx);
int64_t mpmhip2d_async_view_blocks(mpmhip2d_ctx *ctx, int64_t capacity, int32_t *block);
int mpmhip2d_async_state(mpmhip2d_ctx *ctx, int64_t out[8]);
double mpmhip2d_async_current_time(const mpmhip2d_ctx *ctx);
int64_t mpmhip2d_async_table(mpmhip2d_ctx *ctx, int32_t nb[2], int64_t capacity, int64_t *strength, int64_t *cfl, int64_t *continuous,
                             int64_t *count, int64_t *particle_t, int64_t *backup_t, int64_t *local_min);

/* ---- the 2D dense-grid demo (BASELINE configs[0]) — replaces advance(dt) of mls-mpm88.cpp:16-69 (annotated twin
 * mls-mpm88-explained.cpp:64-197): (n+1)^2 grid, snow model inline (E = 1e4, nu = 0.2, hardening 10, sigma clamp
 * [0.975, 1.0075], Jp in [0.6, 20]), gravity -200 on the grid, sticky side/top walls and a separating floor at 0.05.
 * Particle state as the demo's `Particle` (:11-13): x[2], v[2], F[4] row-major, C[4], Jp.  NULL inputs of
 * mpmhip_mpm88_add take the demo's constructor values (v = 0, F = I, C = 0, Jp = 1); NULL outputs of download are
 * skipped.  `plastic` = the demo's `plastic` flag (:10).  Its own small object: no mpmhip_ctx involved. */
typedef struct mpmhip_mpm88 mpmhip_mpm88;
int mpmhip_mpm88_create(int32_t n_grid, float dt, int32_t plastic, int32_t device, mpmhip_mpm88 **out);
void mpmhip_mpm88_destroy(mpmhip_mpm88 *m);
const char *mpmhip_mpm88_last_error(const mpmhip_mpm88 *m);
int mpmhip_mpm88_add(mpmhip_mpm88 *m, int64_t n, const float *x, const float *v, const float *F, const float *C, const float *Jp);
int64_t mpmhip_mpm88_num_particles(const mpmhip_mpm88 *m);
int mpmhip_mpm88_advance(mpmhip_mpm88 *m, int32_t steps);  /* `steps` x advance(dt); asynchronous */
int mpmhip_mpm88_download(mpmhip_mpm88 *m, float *x, float *v, float *F, float *C, float *Jp);  /* synchronises */
int mpmhip_mpm88_download_grid(mpmhip_mpm88 *m, float *grid /* [(n+1)^2][3] = (v.x, v.y, m>0 ? 1 : 0) after advance */);

/* measurement helper: streams `bytes` (rounded down to 16; two scratch buffers of that size are allocated and freed)
 * through a plain copy kernel `iters` times on the ctx stream and returns the best rate in GB/s, counting the bytes
 * read plus the bytes written.  bench.py reports it next to the nominal HBM peak.  Synchronises. */
int mpmhip_debug_copy_bandwidth(mpmhip_ctx *ctx, size_t bytes, int32_t iters, double *gb_per_s);
/* measurement helper: 1 when the next substep's G2P is k_g2p_packed (chunks of 256 consecutive sorted positions; large
 * one-material problems without rigid bodies or tiling), 0 when it is k_g2p (chunks inside one block) */
int mpmhip_debug_g2p_is_packed(const mpmhip_ctx *ctx);
/* the launch bound of the single-pass (chained) scans of the sort as a function of the occupancy API's answer: `limit` = workgroups
 * the host launches at most, `resident` = workgroups the device certainly keeps resident (one per CU below the API's number, at most
 * 7).  Pure host arithmetic (no device): mpmhip_create checks limit <= resident for every such kernel; the waits are bounded besides. */
/* measurement helper: census of cond(F) = sigma_max / sigma_min over the live particles of the ctx (the state as stored: elastic
 * deformation gradients after the last return mapping).  out[0] live particles, [1] particles with cond > 8 — the ones the
 * eigen-solve of the transfer kernels finishes on F itself —, [2] waves of 64 consecutive slots, [3] waves
 * holding such a particle, [4] max cond, [8 + b] histogram over b = floor(8 log2 cond) (eighth-octave bins, clamped to 255).
 * Synchronises.  (What decides whether the device's fp32 tolerances hold: they do to cond 1e2) */
#define MPMHIP_COND_CENSUS_WORDS 264
int mpmhip_debug_cond_census(mpmhip_ctx *ctx, double out[MPMHIP_COND_CENSUS_WORDS]);
int mpmhip_debug_scan_grid(int32_t n_cus, int32_t per_cu, int32_t env_request, uint32_t *limit, uint32_t *resident);
/* 64-byte record gather of KNOWN size — the access pattern of k_p2g / k_g2p (a lane fetches one whole record with four
 * 16-byte loads through an index): n (a power of two) records, index pattern 0 identity / 1 shuffled runs of 8 /
 * 2 fully shuffled.  Reads exactly (64 + 4) n bytes per launch: the yardstick rocprofv3's FETCH_SIZE is calibrated
 * against for this access width (profiles/calibrate_fetch.py). */
int mpmhip_debug_gather_bandwidth(mpmhip_ctx *ctx, int64_t n_records, int32_t mode, int32_t iters, double *gb_per_s);

/* debug/parity helpers running the device math on host arrays (n items each) */
int mpmhip_debug_svd3(mpmhip_ctx *ctx, int64_t n, const float *F, float *U, float *S, float *V);
int mpmhip_debug_force(mpmhip_ctx *ctx, int32_t material, const float params[MPMHIP_NPARAM], int64_t n,
                       const float *F, const float *aux, float *out);
/* next_force == NULL: plasticity(cdg) alone; else the fused "plasticity + next substep's calculate_force" of k_g2p */
int mpmhip_debug_plasticity(mpmhip_ctx *ctx, int32_t material, const float params[MPMHIP_NPARAM], int64_t n,
                            const float *cdg, float *F, float *aux, float *next_force);

/* CPIC rigid coupling of MPM<2> (bodies made of segments; same semantics as the 3D entry points below).  A script
 * returns the position in out[0..1] resp. the angle in degrees in out[0]. */
typedef void (*mpmhip_script_fn)(void *user, float t, float out[3]);
typedef struct mpmhip2d_rigid_config {
  int32_t codimensional, recenter, reverse_vertices, reserved0;
  float density;            /* <= 0: 40 (codimensional) / 400 */
  float friction[2], restitution;
  float scale[2];
  float initial_position[2], initial_rotation /* degrees */, initial_velocity[2], initial_angular_velocity;
  float linear_damping, angular_damping;
  mpmhip_script_fn scripted_position; void *position_user;
  mpmhip_script_fn scripted_rotation; void *rotation_user;
} mpmhip2d_rigid_config;
int mpmhip2d_set_rigid_coupling(mpmhip2d_ctx *ctx, float penalty, float pushing_force);
/* MPM<2>::rigid_body_levelset_collision (src/mpm_rigid_body.cpp:347-387, config key rigid_body_levelset_collision): see
 * mpmhip_set_rigid_levelset_collision */
int mpmhip2d_set_rigid_levelset_collision(mpmhip2d_ctx *ctx, int32_t enabled);
int mpmhip2d_add_rigid_body(mpmhip2d_ctx *ctx, const mpmhip2d_rigid_config *cfg, int64_t n_segments, const float *segments /* n x 4 */);
/* joints in 2D: type = MPMHIP_JOINT_ROTATION only (scripts/mls-cpic/sand_wheel_2D.py:88); struct below, in the CPIC section */
struct mpmhip_joint_config;
int mpmhip2d_add_articulation(mpmhip2d_ctx *ctx, const struct mpmhip_joint_config *cfg);
int mpmhip2d_set_articulation_iterations(mpmhip2d_ctx *ctx, int32_t n);
int mpmhip2d_rigid_get_state(mpmhip2d_ctx *ctx, int32_t id, float *out /* [10]: pos 2, angle, vel 2, omega, mass, inv_mass, inertia, inv_inertia */);
int64_t mpmhip2d_rigid_get_samples(mpmhip2d_ctx *ctx, int32_t id, int64_t capacity, float *position);
int64_t mpmhip2d_rigid_get_mesh(mpmhip2d_ctx *ctx, int32_t id, int64_t capacity_segments, float *segments /* n x 4, world space */);
int mpmhip2d_cdf_phase(mpmhip2d_ctx *ctx); /* rasterize_rigid_boundary + gather_cdf (parity tests) */
int mpmhip2d_download_cdf(mpmhip2d_ctx *ctx, uint32_t *states, float *distance); /* dense (res+1)^2 */
int64_t mpmhip2d_download_colours(mpmhip2d_ctx *ctx, int64_t capacity, uint32_t *states, float *distance, float *normal, int32_t *near);

/* ---------------------------------------------------------------------------------------------------------------------
 * CPIC rigid coupling (3D) — replaces add_particles(type='rigid', ...) (src/mpm.cpp:80-83 -> MPM::add_rigid_particle,
 * src/mpm_rigid_body.cpp:130-252), rasterize_rigid_boundary / gather_cdf (src/rigid_transfer.cpp), the rigid branches of
 * the transfers (block_op_rigid, src/transfer.cpp:367-463,706-835) and advect_rigid_bodies (src/mpm_rigid_body.cpp:255-286).
 * Once a body exists, every substep runs: sort | rasterize_rigid_boundary | gather_cdf | P2G | grid | G2P | advect.
 * Not part of this library: rigid-rigid collisions (rigidify / libccd), rigid_body_levelset_collision (joints are:
 * mpmhip_add_articulation below).  Rigid bodies cannot be combined with the multi-GPU tiling or asynchronous stepping.
 * ------------------------------------------------------------------------------------------------------------------ */
/* scripted_position(t) -> world position; scripted_rotation(t) -> Euler angles in degrees (applied X * Y * Z):
 * tc.function13 objects in the scene scripts (scripts/mls-cpic/sand_paddles.py:27, src/mpm_rigid_body.cpp:79-92) */
typedef struct mpmhip_rigid_config {
  int32_t codimensional;     /* a shell (cuts the material) instead of a solid; mandatory key of the reference */
  int32_t recenter;          /* 1 (reference default): the mesh is moved so that its centre of mass is the body origin */
  int32_t reverse_vertices;  /* flip the orientation of every triangle */
  int32_t reserved0;
  float density;             /* <= 0: the reference's default, 40 (codimensional) / 400 */
  float friction[2];         /* friction0 / friction1: the two sides of the surface ('friction' sets both) */
  float restitution;
  float scale[3];            /* 0 = 1 */
  float initial_position[3], initial_rotation[3] /* Euler, degrees */, initial_velocity[3], initial_angular_velocity[3];
  float rotation_axis[3];    /* angular velocity restricted to this (world) axis when max |axis| > 0.1 */
  float linear_damping, angular_damping;
  mpmhip_script_fn scripted_position; void *position_user;  /* null: a free body */
  mpmhip_script_fn scripted_rotation; void *rotation_user;
} mpmhip_rigid_config;

/* 'penalty' and 'pushing_force' of MPM::initialize (src/mpm.cpp:35,40); defaults 0 and 20000 */
int mpmhip_set_rigid_coupling(mpmhip_ctx *ctx, float penalty, float pushing_force);
/* triangles: n_triangles x 9 floats (mesh space, before `scale`).  Returns the body's index (>= 1; 0 = background) —
 * the string add_particles returns for type='rigid' (src/mpm.cpp:82) */
int mpmhip_add_rigid_body(mpmhip_ctx *ctx, const mpmhip_rigid_config *cfg, int64_t n_triangles, const float *triangles);
int32_t mpmhip_num_rigid_bodies(const mpmhip_ctx *ctx); /* including the background body: rigids.size() */
/* out[33]: position 3, rotation quaternion (w,x,y,z) 4, velocity 3, angular velocity 3, mass, inv_mass,
 * inertia 9 (body frame, row-major), inv_inertia 9 */
int mpmhip_rigid_get_state(mpmhip_ctx *ctx, int32_t id, float *out);
int mpmhip_rigid_set_velocity(mpmhip_ctx *ctx, int32_t id, const float *velocity, const float *angular_velocity);
/* the boundary particles sampled on the body's triangles (RigidBoundaryParticle, src/boundary_particle.h): world
 * position, offset from the centre of mass in the body frame, body index; id < 0: all bodies.  Returns the count. */
int64_t mpmhip_rigid_get_samples(mpmhip_ctx *ctx, int32_t id, int64_t capacity, float *position, float *offset, int32_t *body);
/* the body's triangles in world space, 9 floats each (write_rigid_body, src/visualize.cpp:102-154: the rigid_%03d_%04d.obj
 * next to every .bgeo frame).  Returns the triangle count. */
int64_t mpmhip_rigid_get_mesh(mpmhip_ctx *ctx, int32_t id, int64_t capacity_triangles, float *triangles);
/* phases (parity tests; mpmhip_substep runs them itself): rasterize_rigid_boundary, gather_cdf (needs a sort),
 * advect_rigid_bodies(base_delta_t) */
int mpmhip_rasterize_rigid_boundary(mpmhip_ctx *ctx);
int mpmhip_gather_cdf(mpmhip_ctx *ctx);
int mpmhip_advect_rigid_bodies(mpmhip_ctx *ctx);
/* joints between rigid bodies — replaces general_action(action='add_articulation', type=..., obj0=..., obj1=...)
 * (src/mpm.cpp:923-933; the Articulation classes of src/articulation.cpp; mpm.add_articulation(...) in
 * scripts/mls-cpic/robot.py:103, water_wheel.py:81).  obj1 = 0 links to the background body.  The joint is set up on the
 * bodies' poses at the time of the call; MPM::articulate (src/mpm.h:278-319) then runs inside every substep between the
 * sort and rasterize_rigid_boundary: apply(dt), articulation_iterations (100) Gauss-Seidel sweeps of project(), penalize(dt). */
enum {
  MPMHIP_JOINT_ROTATION = 0,  /* 'rotation': both bodies share one angular velocity (their total angular momentum)        */
  MPMHIP_JOINT_FROZEN = 1,    /* 'frozen': obj0 keeps angular velocity z and velocity x, y only                           */
  MPMHIP_JOINT_DISTANCE = 2,  /* 'distance': anchors offset0 / offset1 (world-frame offsets from the centres) at target_distance */
  MPMHIP_JOINT_AXIAL_ROTATION = 3, /* 'axial_rotation': hinge about `axis` through obj0's centre + offset0                */
  MPMHIP_JOINT_MOTOR = 4,     /* 'motor': hinge + torque `power` about the axis                                           */
  MPMHIP_JOINT_STEPPER = 5    /* 'stepper': hinge + relative angular velocity about the axis held at `angular_velocity`   */
};
typedef struct mpmhip_joint_config {
  int32_t type, obj0, obj1;
  int32_t has_offset1;          /* 'offset1' given (mandatory for a distance joint to the background body)        */
  int32_t has_target_distance;  /* 'target_distance' given; otherwise the anchors' distance at the time of the call */
  float offset0[3], offset1[3];
  float target_distance;
  float penalty;                /* < 0: the reference's default 1e3 */
  float axis[3];
  float axis_length;            /* < 0: the reference's default 0.1 */
  float power, angular_velocity;
} mpmhip_joint_config;
int mpmhip_add_articulation(mpmhip_ctx *ctx, const mpmhip_joint_config *cfg);
int32_t mpmhip_num_articulations(const mpmhip_ctx *ctx);
int mpmhip_set_articulation_iterations(mpmhip_ctx *ctx, int32_t n); /* config key 'articulation_iterations', default 100 */
int mpmhip_articulate(mpmhip_ctx *ctx); /* phase (parity tests): MPM::articulate(base_delta_t) */
/* dense (res+1)^3 views of the grid's colored distance field: GridState::states (24 colour bits | body id + 1 << 24,
 * src/mpm_fwd.h:69-105) and GridState::distance */
int mpmhip_download_cdf(mpmhip_ctx *ctx, uint32_t *states, float *distance);
/* gather_cdf's per-particle results, live particles in slot order, 5 floats each: boundary_normal 3, boundary_distance,
 * near_boundary.  Valid between mpmhip_gather_cdf and the next G2P (a G2P moves the records). */
int64_t mpmhip_download_boundary(mpmhip_ctx *ctx, float *out, int64_t n_capacity);

#ifdef __cplusplus
}
#endif
#endif /* MPMHIP_H */
