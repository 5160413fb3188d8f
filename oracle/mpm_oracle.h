/* oracle/mpm_oracle.h — C ABI of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the reported CPU baseline.
 *
 * The oracle is a plain CPU restatement (C++17, fp32, no dependencies) of the
 * reference's MLS-MPM sub-step: yuanming-hu/taichi_mpm, files cited next to
 * each function as `file:line` relative to the reference root.
 *
 * PARITY STATUS: "parity unpinned" for P2G / G2P / constitutive models / svd /
 * polar_decomp — the reference cannot be compiled in this environment (needs
 * the un-vendored legacy taichi core) and ships no golden vectors for them.
 * Only the B-spline kernel weights are pinned by the reference's own tests
 * (src/tests.cpp:13-51, src/transfer.cpp:353-359,975-989), re-expressed in
 * tests/test_oracle_kernel.py.  Everything else is pinned by invariants
 * (conservation, SVD reconstruction vs numpy) and by an independent numpy
 * restatement (oracle/np_mpm.py).
 */
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Material ids = the y-component of MPMParticle::get_debug_info()
 * (src/particles.cpp:157,289,348,423,497,670,755,839). */
enum {
  ORC_VISCO = 1,
  ORC_SNOW = 2,
  ORC_LINEAR = 3,
  ORC_JELLY = 4,
  ORC_WATER = 5,
  ORC_SAND = 6,
  ORC_VON_MISES = 7,
  ORC_ELASTIC = 8
};

#define ORC_NPARAM 16
/* Per-group parameter table, float[ORC_NPARAM]:
 *  [0] mass  [1] vol
 *  snow     : [2] mu_0 [3] lambda_0 [4] hardening [5] theta_c [6] theta_s [7] min_Jp [8] max_Jp   aux = Jp
 *  linear   : [2] mu   [3] lambda
 *  jelly    : [2] mu   [3] lambda
 *  water    : [2] k    [3] gamma                                                                 aux = j
 *  sand     : [2] mu_0 [3] lambda_0 [4] alpha [5] cohesion [6] beta                               aux = logJp
 *  von_mises: [2] mu_0 [3] lambda_0 [4] yield_stress
 *  elastic  : [2] mu_0 [3] lambda_0
 *  visco    : [2] mu_0 [3] lambda_0 [4] visco_nu [5] visco_kappa [6] dt                           aux = visco_tau
 */

typedef struct {
  int32_t res[3];          /* cells per axis; nodes = res+1 (src/mpm.cpp:66) */
  float dx;                /* delta_x */
  float dt;                /* base_delta_t * dt_multiplier */
  float gravity[3];
  int32_t particle_gravity; /* src/mpm.cpp:47 (default true) */
  float apic_damping;
  float rpic_damping;
  int32_t clean_boundary;  /* src/mpm.cpp:563 */
  /* analytic level set: up to 8 half-spaces  phi(x) = n.x + d  (world units),
   * combined by min (union of solids = intersection of free space);
   * negative inside the solid.  (reference: taichi LevelSet, sampled) */
  int32_t n_planes;
  float planes[8][4];
  float friction;          /* levelset0->friction: -1 sticky, <=-2 slip, >=0 separate */
  /* further analytic solids, also combined by min (taichi LevelSet::add_sphere / add_cuboid as the scene
   * scripts use them, e.g. scripts/mls-cpic; the sampled implementation itself is in the un-vendored
   * taichi core): type 1 sphere {cx,cy,cz,r}; type 2 axis-aligned cuboid {lo xyz, hi xyz}; world units.
   * inside_out = 1: the FREE space is the inside of the shape (a container). */
  int32_t n_shapes;
  struct { int32_t type, inside_out; float p[6]; } shapes[8];
  int32_t particle_collision; /* src/mpm.cpp:566-569, :414-426: push particles out of the level set after G2P */
  /* DynamicLevelSet (scripts/async/async_mpm.py:119-127: levelset(t0), levelset(t1) blended linearly in time): when
   * `dynamic` != 0 the fields above are the key frame at t0 and these the key frame at t1; `t` = current_t of the
   * substep.  phi = lerp, gradient = normalised lerp of the two gradients, boundary velocity =
   * -(phi1 - phi0)/(t1 - t0) n dx (src/mpm.cpp:340-342). */
  int32_t dynamic;
  float t0, t1, t;
  int32_t n_planes1;
  float planes1[8][4];
  int32_t n_shapes1;
  struct { int32_t type, inside_out; float p[6]; } shapes1[8];
} orc_config;

/* --- kernel weights (src/kernel.h:103-135,168-210; src/transfer.cpp:162-191) */
/* MPMKernel<3,2>/MPMFastKernel32::get_dw_w for the 27 nodes: out[27][4] = (dw/dx,dw/dy,dw/dz,w) */
void orc_kernel3_dw_w(const float pos[3], float inv_dx, float out[27 * 4]);
/* same through MPMKernelBase::shuffle()+get_dw_w (the slow path), for the KAT of src/tests.cpp:35-51 */
void orc_kernel3_dw_w_slow(const float pos[3], float inv_dx, float out[27 * 4]);
/* MLSMPMFastKernel32: weights only, pos already relative to the base cell, in [0.5,1.5)^3 */
void orc_mls_kernel3_w(const float rel_pos[3], float out[27]);
/* MPMKernel<2,2>::get_dw_w : out[9][3] */
void orc_kernel2_dw_w(const float pos[2], float inv_dx, float out[9 * 3]);
/* MPMKernel<dim,3> (cubic) for the Σw / Σ∇w KAT: out[(dim==2?16:64)][dim+1] */
void orc_kernel_cubic_dw_w(int dim, const float* pos, float inv_dx, float* out);

/* --- 3x3 / 2x2 factorizations (taichi core svd()/polar_decomp(), un-vendored;
 * convention documented in DESIGN.md: U,V rotations, |sigma| descending, sign on last) */
void orc_svd3(const float F[9], float U[9], float S[3], float V[9]);
void orc_polar3(const float F[9], float R[9], float Ssym[9]);
void orc_svd2(const float F[4], float U[4], float S[2], float V[4]);
void orc_polar2(const float F[4], float R[4], float Ssym[4]);

/* --- constitutive models (src/particles.cpp) — all matrices row-major float[9] */
void orc_calculate_force(int type, const float* gp, const float F[9], float aux, float out[9]);
void orc_plasticity(int type, const float* gp, const float cdg[9], float F[9], float* aux);

/* --- friction_project (src/mpm_fwd.h:25-57) */
void orc_friction_project(const float v[3], const float vb[3], const float n[3], float mu, float out[3]);

/* --- phases of MPM<3>::substep (src/mpm.cpp:452-575); dense grid float[(rx+1)(ry+1)(rz+1)][4] */
/* P2G: src/transfer.cpp:467-569 (rasterize_optimized/block_op_normal).  v is updated in place
 * when particle_gravity (transfer.cpp:485-487). alive[i]==0 particles are skipped (may be NULL). */
void orc_p2g(const orc_config* c, int64_t n, const float* x, float* v, const float* B, const float* F,
             const float* aux, const int32_t* gid, const float* gparams, const int32_t* gtype,
             float* grid);
/* normalize_grid_and_apply_external_force + apply_grid_boundary_conditions
 * (src/mpm.cpp:277-294,296-372) */
void orc_grid_update(const orc_config* c, float* grid);
/* G2P: src/transfer.cpp:837-954 (resample_optimized/block_op_normal) */
void orc_g2p(const orc_config* c, int64_t n, float* x, float* v, float* B, float* F, float* aux,
             const int32_t* gid, const float* gparams, const int32_t* gtype, const float* grid);
/* clear_boundary_particles: src/mpm.cpp:582-633, src/mpm.h:269-276. keep[i]=1 if the particle survives.
 * (also drops particles whose stencil would leave the grid — the reference has UB there) */
int64_t orc_clear_boundary(const orc_config* c, int64_t n, const float* x, const float* v, uint8_t* keep);
/* general_action "delete_particles_inside_level_set" (src/mpm.cpp:962-974): keep[p] = 0 where phi(x_p) < 0 */
int64_t orc_delete_inside_levelset(const orc_config* c, int64_t n, const float* x, uint8_t* keep);
/* particle_collision_resolution: src/mpm.cpp:414-426 */
void orc_particle_collision(const orc_config* c, int64_t n, float* x, float* v);
/* one full substep on SoA arrays; particles are compacted in place (stable); returns new n.
 * ids (may be NULL) is permuted along. grid is scratch of the dense size. */
int64_t orc_substep(const orc_config* c, int64_t n, float* x, float* v, float* B, float* F, float* aux,
                    int32_t* gid, int32_t* ids, const float* gparams, const int32_t* gtype, float* grid);

/* --- 2D dense demo: mls-mpm88.cpp:16-69 advance(dt).  x,v: n*2; F,C: n*4; Jp: n; grid scratch (ng+1)^2*3 */
void orc_mpm88_advance(int n_grid, float dt, int64_t n, float* x, float* v, float* F, float* C, float* Jp,
                       float* grid, int plastic);

/* --- timed CPU baseline: block-sorted, 8-colour, scratch-tile restatement of
 * rasterize_optimized / resample_optimized with OpenMP threads.
 * Runs `steps` substeps, returns wall seconds; phase_ns[4] = sort,p2g,grid,g2p totals. */
double orc_opt_run(const orc_config* c, int64_t n, float* x, float* v, float* B, float* F, float* aux,
                   const int32_t* gid, const float* gparams, const int32_t* gtype,
                   int steps, int threads, double phase_s[4]);

#ifdef __cplusplus
}
#endif
