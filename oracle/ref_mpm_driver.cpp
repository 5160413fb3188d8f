// oracle/ref_mpm_driver.cpp — TEST INFRASTRUCTURE (never shipped, never on the product path).
//
// C-ABI driver around the REFERENCE's own solver, compiled from its sources where they lie
// (/root/reference/src/{mpm,transfer,visualize}.cpp as separate objects, particles.cpp included below because the
// material classes are defined in that .cpp) against oracle/taichi_shim — see the header of
// oracle/taichi_shim/taichi/common/util.h for what that stand-in provides and what it cannot pin (svd/polar_decomp
// and the sampled level set of the absent taichi core).  Built by `make -C oracle ref_mpm` into
// oracle/_ref/libmpm_ref.so.  Used (i) to check the restated oracle (oracle/mpm_oracle.cpp) and the HIP library
// against the reference's real arithmetic, (ii) to generate tests/golden/ref_*.npz, (iii) as the `cpu_baseline` of
// bench.py (kind "reference").
//
// Everything here only moves data in and out of the reference's own objects (MPM<dim>, MPMParticle<dim>):
// no arithmetic of the path is written in this file.  Matrices cross this ABI row-major (m[dim*r + c]); the
// reference stores columns (Matrix[i] = column i).
#include <taichi/common/util.h>

#include "particles.cpp"  // the reference's constitutive models (class definitions live in the .cpp)

#include "kernel.h"
#include "mpm.h"
#include "boundary_particle.h"
#include "async/async_mpm.h"  // AsyncMPM<dim>: this TU is compiled with -fno-access-control so that the checker can read
                              // the block table (`blocks`, `particle_pool`), which the class keeps private

TC_NAMESPACE_BEGIN

// CPIC rigid coupling: src/rigid_transfer.cpp (colored distance field), src/mpm_rigid_body.cpp (bodies, boundary
// particles, advection) and src/boundary_particle.cpp are compiled as separate objects like the other sources; the rigid
// body itself is the shim's (taichi/dynamics/rigid_body_shim.h).  Joints: src/articulation.cpp, compiled in place as well
// (ref_general_action with action=add_articulation; phase 10 = MPM::articulate).
namespace {

thread_local std::string g_err;

template <int dim>
MatrixND<dim, real> mat_in(const float *m) {
  MatrixND<dim, real> M;
  for (int r = 0; r < dim; r++)
    for (int c = 0; c < dim; c++) M[c][r] = m[dim * r + c];
  return M;
}
template <int dim>
void mat_out(const MatrixND<dim, real> &M, float *m) {
  for (int r = 0; r < dim; r++)
    for (int c = 0; c < dim; c++) m[dim * r + c] = M[c][r];
}

// the one scalar of material state outside dg_e: Jp (snow), j (water), logJp (sand), visco_tau (visco)
template <int dim>
real *aux_ptr(MPMParticle<dim> *p) {
  if (auto q = dynamic_cast<SnowParticle<dim> *>(p)) return &q->Jp;
  if (auto q = dynamic_cast<WaterParticle<dim> *>(p)) return &q->j;
  if (auto q = dynamic_cast<SandParticle<dim> *>(p)) return &q->logJp;
  if (auto q = dynamic_cast<ViscoParticle<dim> *>(p)) return &q->visco_tau;
  return nullptr;
}

struct Handle {
  int dim = 3;
  std::unique_ptr<MPM<2>> m2;
  std::unique_ptr<MPM<3>> m3;
  AsyncMPM<3> *async3 = nullptr;  // == m3.get() when the simulation is the "async_mpm" one
  AsyncMPM<2> *async2 = nullptr;  // == m2.get() likewise (create_simulation2('async_mpm'))
  // scripted motions of rigid bodies: the reference receives POINTERS to std::function objects through its config
  // (src/mpm_rigid_body.cpp:79-92) and copies them
  std::vector<std::unique_ptr<RigidBody<3>::PositionFunctionType>> pos_scripts;
  std::vector<std::unique_ptr<RigidBody<3>::RotationFunctionType>> rot_scripts;
  std::vector<std::unique_ptr<RigidBody<2>::PositionFunctionType>> pos_scripts2;
  std::vector<std::unique_ptr<RigidBody<2>::RotationFunctionType>> rot_scripts2;
};

template <int dim> MPM<dim> &sim(Handle *h);
template <> MPM<2> &sim<2>(Handle *h) { return *h->m2; }
template <> MPM<3> &sim<3>(Handle *h) { return *h->m3; }
template <int dim> AsyncMPM<dim> *async_sim(Handle *h);
template <> AsyncMPM<2> *async_sim<2>(Handle *h) { return h->async2; }
template <> AsyncMPM<3> *async_sim<3>(Handle *h) { return h->async3; }

template <int dim>
int add_particles(Handle *h, const Config &cfg, int64_t n, const float *x, const float *v, const float *F, const float *B,
                  const float *aux) {
  MPM<dim> &m = sim<dim>(h);
  const std::string type = cfg.get<std::string>("type");
  const real vol = cfg.get<real>("vol"), mass = cfg.get<real>("mass");
  // one prototype goes through initialize(config) (string lookups); the others are byte copies of its container with
  // their own id — the reference itself moves particles as container bytes (ParticleContainer's copy constructor,
  // sort_allocator src/mpm.cpp:752-768)
  ParticleContainer<dim> proto;
  {
    MPMParticle<dim> *pp = create_instance_placement<MPMParticle<dim>>(type, &proto);
    pp->initialize(cfg);
    pp->vol = vol;
    pp->set_mass(mass);
  }
  m.allocator.pool.reserve(m.allocator.pool.size() + (size_t)n);
  m.particles.reserve(m.particles.size() + (size_t)n);
  for (int64_t i = 0; i < n; i++) {  // = create_particle, src/mpm.cpp:101-147, with explicit state
    auto alloc = m.allocator.allocate_particle(type);
    MPMParticle<dim> *p = alloc.second;
    const int32 id = p->id;
    memcpy(&m.allocator.pool[alloc.first], &proto, sizeof proto);
    p->id = id;
    VectorND<dim, real> pos, vel(0.0f);
    for (int k = 0; k < dim; k++) { pos[k] = x[dim * i + k]; if (v) vel[k] = v[dim * i + k]; }
    p->pos = pos;
    p->set_velocity(vel);
    if (F) p->dg_e = mat_in<dim>(F + dim * dim * i);
    if (B) p->apic_b = mat_in<dim>(B + dim * dim * i);
    if (aux) if (real *a = aux_ptr<dim>(p)) *a = aux[i];
    m.particles.push_back(alloc.first);
  }
  if (async_sim<dim>(h)) {
    // the tail of AsyncMPM<dim>::add_particles (src/async/async_mpm.cpp:62-75): the new particles move from the
    // flat list into the pool of the scheduler block of their base node
    AsyncMPM<dim> &a = *async_sim<dim>(h);
    using Mask = typename MPM<dim>::SparseMask;
    for (auto p : a.particles) {
      uint64 grid_offset = Mask::Linear_Offset(a.MPM<dim>::get_grid_base_pos(
          (reinterpret_cast<MPMParticle<dim> *>(&a.allocator.pool[p]))->pos * a.inv_delta_x));
      uint64 offset = (grid_offset >> Mask::data_bits >> Mask::block_bits) & a.scheduler_mask;
      a.particle_pool[offset].push_back(a.allocator.pool[p]);
    }
    a.particles.clear();
  }
  return 0;
}

template <int dim>
int64_t download(Handle *h, float *x, float *v, float *F, float *B, float *aux, int32_t *id) {
  MPM<dim> &m = sim<dim>(h);
  int64_t i = 0;
  for (auto pi : m.particles) {
    MPMParticle<dim> *p = m.allocator[pi];
    if (p->is_rigid()) continue;  // the boundary particles of a rigid body: see ref_rigid_samples
    auto vel = p->get_velocity();
    for (int k = 0; k < dim; k++) { if (x) x[dim * i + k] = p->pos[k]; if (v) v[dim * i + k] = vel[k]; }
    if (F) mat_out<dim>(p->dg_e, F + dim * dim * i);
    if (B) mat_out<dim>(p->apic_b, B + dim * dim * i);
    if (aux) { real *a = aux_ptr<dim>(p); aux[i] = a ? *a : 0.0f; }
    if (id) id[i] = p->id;
    i++;
  }
  return i;
}

// dense (res+1)^dim x (dim+1) view of the reference's sparse grid: velocity_and_mass of every allocated node
template <int dim>
int grid_io(Handle *h, float *dense, bool upload) {
  MPM<dim> &m = sim<dim>(h);
  using Mask = typename MPM<dim>::SparseMask;
  auto blocks = m.fat_page_map->Get_Blocks();
  auto bs = m.grid_block_size();
  for (unsigned b = 0; b < blocks.second; b++) {
    VectorND<dim, int> base(Mask::LinearToCoord(blocks.first[b]));
    for (auto &ind : RegionND<dim>(VectorND<dim, int>(0), bs)) {
      VectorND<dim, int> g = base + ind.get_ipos();
      bool in = true;
      size_t lin = 0;
      for (int k = 0; k < dim; k++) { in = in && g[k] <= m.res[k]; lin = lin * (m.res[k] + 1) + g[k]; }
      if (!in) continue;
      auto &node = m.get_grid(g).velocity_and_mass;
      for (int k = 0; k <= dim; k++) {
        if (upload) node[k] = dense[lin * (dim + 1) + k];
        else dense[lin * (dim + 1) + k] = node[k];
      }
    }
  }
  return 0;
}

template <int dim>
std::shared_ptr<LevelSet<dim>> make_levelset(MPM<dim> &m, int n, const float *shapes, float friction) {
  auto L = std::make_shared<LevelSet<dim>>();
  L->friction = friction;
  L->delta_x = m.delta_x;
  for (int i = 0; i < n; i++) {  // rows of 8 floats: type, inside_out, p[6]
    const float *r = shapes + 8 * i;
    ShimShape s;
    s.type = (int)r[0]; s.inside_out = (int)r[1];
    for (int k = 0; k < 6; k++) s.p[k] = r[2 + k];
    L->shapes.push_back(s);
  }
  return L;
}
// static level set (n1 < 0) or DynamicLevelSet::initialize(t0, t1, levelset(t0), levelset(t1)) as
// scripts/async/async_mpm.py:119-127 update_levelset builds it every frame
template <int dim>
void set_levelset(Handle *h, int n0, const float *shapes0, int n1, const float *shapes1, float t0, float t1, float friction) {
  MPM<dim> &m = sim<dim>(h);
  DynamicLevelSet<dim> D;
  D.initialize(t0, t1, make_levelset<dim>(m, n0, shapes0, friction), n1 >= 0 ? make_levelset<dim>(m, n1, shapes1, friction) : nullptr);
  m.set_levelset(D);
}

template <int dim>
int phase(Handle *h, int which, int optimized) {
  MPM<dim> &m = sim<dim>(h);
  const real dt = m.base_delta_t;
  switch (which) {
    case 0: m.sort_particles_and_populate_grid(); break;                      // src/mpm.cpp:770-918
    case 1: if (optimized) m.rasterize_optimized(dt); else m.rasterize(dt); break;  // src/transfer.cpp:193-283,361-581
    case 2:                                                                    // src/mpm.cpp:524-533
      m.normalize_grid_and_apply_external_force(m.particle_gravity ? VectorND<dim, real>(0.0f) : m.gravity * dt);
      m.apply_grid_boundary_conditions(m.levelset, m.current_t);
      break;
    case 3: if (optimized) m.resample_optimized(); else m.resample(); break;  // src/transfer.cpp:585-700,702-970
    case 4: m.clear_boundary_particles(); break;                              // src/mpm.cpp:582-633
    case 5: m.particle_collision_resolution(m.current_t); break;              // src/mpm.cpp:414-426
    case 6: m.normalize_grid_and_apply_external_force(m.particle_gravity ? VectorND<dim, real>(0.0f) : m.gravity * dt); break;
    case 7: m.rasterize_rigid_boundary(); break;                               // src/rigid_transfer.cpp:17-115
    case 8: m.gather_cdf(); break;                                             // src/rigid_transfer.cpp:121-275
    case 9: m.advect_rigid_bodies(dt); break;                                  // src/mpm_rigid_body.cpp:255-286
    case 10: m.articulate(dt); break;                                          // src/mpm.h:278-319 (joints: src/articulation.cpp)
    default: return -1;
  }
  return 0;
}

template <int dim>
MPMParticle<dim> *scratch_particle(const Config &cfg, std::vector<uint8> &buf) {
  buf.assign(get_particle_size_upper_bound<dim>(), 0);
  MPMParticle<dim> *p = create_instance_placement<MPMParticle<dim>>(cfg.get<std::string>("type"), buf.data());
  p->initialize(cfg);
  p->vol = cfg.get<real>("vol");
  p->set_mass(cfg.get<real>("mass"));
  return p;
}

template <typename F>
int guarded(const F &f) {
  try {
    return f();
  } catch (const std::exception &e) {
    g_err = e.what();
    return -1;
  }
}

}  // namespace

TC_NAMESPACE_END

using namespace taichi;

#define DISPATCH(h, expr2, expr3) ((h)->dim == 2 ? (expr2) : (expr3))

// ---- AsyncMPM helpers, both dimensions (the extern "C" wrappers are below) ----
#define ASYNC_DISPATCH(h, call) ((h)->dim == 2 ? call<2>(h) : call<3>(h))
template <int dim>
AsyncMPM<dim> &async_ref(Handle *h) {
  if (!async_sim<dim>(h)) TC_ERROR("not an async_mpm simulation");
  return *async_sim<dim>(h);
}
template <int dim>
int64_t async_blocks(Handle *h, int64_t cap, int32_t *coord, int64_t *strength, int64_t *cfl, int64_t *continuous, int64_t *count, int64_t *mm) {
  AsyncMPM<dim> &a = async_ref<dim>(h);
  using Mask = typename MPM<dim>::SparseMask;
  int64_t n = 0;
  for (uint64 offset = 0; offset < a.scheduler_size; ++offset) {
    if (a.particle_pool[offset].empty()) continue;
    if (n >= cap) TC_ERROR("block buffer too small");
    auto c = Mask::LinearToCoord(uint64(offset) << Mask::data_bits << Mask::block_bits);
    for (int k = 0; k < 3; k++) coord[3 * n + k] = k < dim ? c[k] : 0;
    strength[n] = a.blocks[offset].strength_dt_limit;
    cfl[n] = a.blocks[offset].cfl_dt_limit;
    continuous[n] = a.blocks[offset].continuous_dt_limit;
    count[n] = (int64_t)a.particle_pool[offset].size();
    n++;
  }
  if (mm) { mm[0] = a.min_delta_t_int; mm[1] = a.max_delta_t_int; }
  return n;
}
template <int dim>
int64_t async_download(Handle *h, int64_t cap, float *x, float *v, float *F, float *B, float *aux, int32_t *id, int64_t *limits) {
  AsyncMPM<dim> &a = async_ref<dim>(h);
  int64_t n = 0;
  for (uint64 offset = 0; offset < a.scheduler_size; ++offset)
    for (auto &container : a.particle_pool[offset]) {
      if (n >= cap) TC_ERROR("particle buffer too small");
      MPMParticle<dim> *p = const_cast<MPMParticle<dim> *>(reinterpret_cast<const MPMParticle<dim> *>(&container));
      auto vel = p->get_velocity();
      for (int k = 0; k < dim; k++) { x[dim * n + k] = p->pos[k]; v[dim * n + k] = vel[k]; }
      if (F) mat_out<dim>(p->dg_e, F + dim * dim * n);
      if (B) mat_out<dim>(p->apic_b, B + dim * dim * n);
      if (aux) { real *q = aux_ptr<dim>(p); aux[n] = q ? *q : 0.0f; }
      id[n] = p->id;
      if (limits) {
        limits[4 * n + 0] = a.blocks[offset].continuous_dt_limit; limits[4 * n + 1] = a.blocks[offset].strength_dt_limit;
        limits[4 * n + 2] = a.blocks[offset].cfl_dt_limit; limits[4 * n + 3] = a.blocks[offset].particle_t;
      }
      n++;
    }
  return n;
}
// the scheduler's geometry: per block number (= position in the order the pools are walked in) the corner node, the
// cached_neighbours (src/async/async_mpm.h:255-301; 26 slots, -1 terminated) and whether it is a "left_boundary" block
template <int dim>
int64_t async_geometry(Handle *h, int64_t cap, int32_t *coord, int32_t *neighbours, int32_t *is_boundary) {
  AsyncMPM<dim> &a = async_ref<dim>(h);
  using Mask = typename MPM<dim>::SparseMask;
  if ((int64_t)a.scheduler_size > cap) return (int64_t)a.scheduler_size;
  for (uint64 offset = 0; offset < a.scheduler_size; ++offset) {
    auto c = Mask::LinearToCoord(uint64(offset) << Mask::data_bits << Mask::block_bits);
    for (int k = 0; k < 3; k++) coord[3 * offset + k] = k < dim ? c[k] : 0;
    int m = 0;
    for (auto q : a.cached_neighbours[offset]) neighbours[26 * offset + m++] = (int32_t)q;
    for (; m < 26; m++) neighbours[26 * offset + m] = -1;
    is_boundary[offset] = 0;
  }
  for (auto b : a.boundary) is_boundary[b] = 1;
  return (int64_t)a.scheduler_size;
}
template <int dim> int64_t async_num(Handle *h) {
  if (!async_sim<dim>(h)) return -1;
  int64_t n = 0;
  for (uint64 offset = 0; offset < async_sim<dim>(h)->scheduler_size; ++offset) n += (int64_t)async_sim<dim>(h)->particle_pool[offset].size();
  return n;
}
template <int dim> int64_t async_time_int(Handle *h) { return async_sim<dim>(h) ? (int64_t)async_sim<dim>(h)->current_t_int : -1; }
template <int dim> int64_t async_update_counter(Handle *h) { return async_sim<dim>(h) ? (int64_t)async_sim<dim>(h)->update_counter : -1; }

extern "C" {

const char *ref_last_error() { return g_err.c_str(); }

int ref_set_threads(int n) {
  ShimRuntime::get().threads = n > 0 ? n : omp_get_max_threads();
  return ShimRuntime::get().threads;
}

// cfg: "res=(32,32,32);delta_x=0.03125;base_delta_t=1e-4;gravity=(0,-10,0);..." — the keys of MPM<dim>::initialize
void *ref_create(int dim, const char *cfg) {
  Handle *h = nullptr;
  const int rc = guarded([&] {
    h = new Handle();
    h->dim = dim;
    Config c = Config::from_string(cfg);
    if (!c.has_key("num_threads")) c.set("num_threads", ShimRuntime::get().threads);
    if (dim == 2 && c.get("async", false)) {  // create_simulation2('async_mpm'), src/async/async_mpm.cpp:423-427
      h->async2 = new AsyncMPM<2>();
      h->m2.reset(h->async2);
      h->m2->initialize(c);
      set_levelset<2>(h, 0, nullptr, -1, nullptr, 0, 1, 1.0f);
    }
    else if (dim == 2) { h->m2 = std::make_unique<MPM<2>>(); h->m2->initialize(c); set_levelset<2>(h, 0, nullptr, -1, nullptr, 0, 1, 1.0f); }
    else if (dim == 3 && c.get("async", false)) {  // create_simulation3('async_mpm'), src/async/async_mpm.cpp:423-427
      h->async3 = new AsyncMPM<3>();
      h->m3.reset(h->async3);
      h->m3->initialize(c);
      set_levelset<3>(h, 0, nullptr, -1, nullptr, 0, 1, 1.0f);
    }
    else if (dim == 3) { h->m3 = std::make_unique<MPM<3>>(); h->m3->initialize(c); set_levelset<3>(h, 0, nullptr, -1, nullptr, 0, 1, 1.0f); }
    else TC_ERROR("dim must be 2 or 3");
    return 0;
  });
  if (rc) { delete h; return nullptr; }
  return h;
}
void ref_destroy(void *h) { delete (Handle *)h; }

int ref_set_levelset(void *hh, int n0, const float *shapes0, int n1, const float *shapes1, float t0, float t1, float friction) {
  Handle *h = (Handle *)hh;
  return guarded([&] {
    DISPATCH(h, set_levelset<2>(h, n0, shapes0, n1, shapes1, t0, t1, friction), set_levelset<3>(h, n0, shapes0, n1, shapes1, t0, t1, friction));
    return 0;
  });
}

// type_cfg: "type=sand;mass=..;vol=..;<the material's own config keys>"
int ref_add_particles(void *hh, const char *type_cfg, int64_t n, const float *x, const float *v, const float *F, const float *B,
                      const float *aux) {
  Handle *h = (Handle *)hh;
  return guarded([&] {
    Config c = Config::from_string(type_cfg);
    return DISPATCH(h, add_particles<2>(h, c, n, x, v, F, B, aux), add_particles<3>(h, c, n, x, v, F, B, aux));
  });
}
// the reference's own generator (src/mpm.cpp:149-186): cfg "type=..;benchmark=125|8000;<material keys>"
int ref_add_particles_cfg(void *hh, const char *cfg) {
  Handle *h = (Handle *)hh;
  return guarded([&] {
    Config c = Config::from_string(cfg);
    DISPATCH(h, h->m2->add_particles(c), h->m3->add_particles(c));
    return 0;
  });
}
int64_t ref_num_particles(void *hh) {  // material particles (the boundary particles of rigid bodies are not counted)
  Handle *h = (Handle *)hh;
  int64_t n = 0;
  if (h->dim == 2) { for (auto pi : h->m2->particles) n += !h->m2->allocator[pi]->is_rigid(); }
  else { for (auto pi : h->m3->particles) n += !h->m3->allocator[pi]->is_rigid(); }
  return n;
}
int64_t ref_download(void *hh, float *x, float *v, float *F, float *B, float *aux, int32_t *id) {
  Handle *h = (Handle *)hh;
  int64_t n = -1;
  guarded([&] { n = DISPATCH(h, download<2>(h, x, v, F, B, aux, id), download<3>(h, x, v, F, B, aux, id)); return 0; });
  return n;
}
int ref_download_grid(void *hh, float *dense) {
  Handle *h = (Handle *)hh;
  return guarded([&] { return DISPATCH(h, grid_io<2>(h, dense, false), grid_io<3>(h, dense, false)); });
}
int ref_upload_grid(void *hh, const float *dense) {
  Handle *h = (Handle *)hh;
  return guarded([&] { return DISPATCH(h, grid_io<2>(h, const_cast<float *>(dense), true), grid_io<3>(h, const_cast<float *>(dense), true)); });
}
int ref_substep(void *hh, int n) {
  Handle *h = (Handle *)hh;
  return guarded([&] {
    for (int i = 0; i < n; i++) DISPATCH(h, h->m2->substep(), h->m3->substep());
    return 0;
  });
}
int ref_step(void *hh, float dt) {
  Handle *h = (Handle *)hh;
  return guarded([&] { DISPATCH(h, h->m2->step(dt), h->m3->step(dt)); return 0; });
}
// which: 0 sort+populate, 1 P2G, 2 grid normalise + boundary, 3 G2P, 4 clear_boundary_particles, 5 particle collision,
// 6 grid normalise only, 7 rasterize_rigid_boundary, 8 gather_cdf, 9 advect_rigid_bodies, 10 articulate
int ref_phase(void *hh, int which, int optimized) {
  Handle *h = (Handle *)hh;
  return guarded([&] { return DISPATCH(h, phase<2>(h, which, optimized), phase<3>(h, which, optimized)); });
}
double ref_time(void *hh) {
  Handle *h = (Handle *)hh;
  return DISPATCH(h, (double)h->m2->current_t, (double)h->m3->current_t);
}
int ref_set_time(void *hh, double t) {
  Handle *h = (Handle *)hh;
  DISPATCH(h, h->m2->current_t = (real)t, h->m3->current_t = (real)t);
  return 0;
}
double ref_calculate_energy(void *hh) {
  Handle *h = (Handle *)hh;
  double e = std::nan("");
  guarded([&] { e = DISPATCH(h, (double)h->m2->calculate_energy(), (double)h->m3->calculate_energy()); return 0; });
  return e;
}
int ref_write_bgeo(void *hh, const char *path) {  // MPM<dim>::write_partio, src/visualize.cpp:17-100
  Handle *h = (Handle *)hh;
  return guarded([&] { DISPATCH(h, h->m2->write_partio(path), h->m3->write_partio(path)); return 0; });
}
int ref_general_action(void *hh, const char *cfg, char *out, size_t cap) {
  Handle *h = (Handle *)hh;
  return guarded([&] {
    Config c = Config::from_string(cfg);
    std::string r = DISPATCH(h, h->m2->general_action(c), h->m3->general_action(c));
    if (out && cap) { std::snprintf(out, cap, "%s", r.c_str()); }
    return 0;
  });
}

// ---- AsyncMPM (src/async/async_mpm.{h,cpp}), both dimensions ------------------------------------------------------------
int ref_async_update_dt_limits(void *hh) {
  Handle *h = (Handle *)hh;
  return guarded([&] { if (h->dim == 2) async_ref<2>(h).update_dt_limits(); else async_ref<3>(h).update_dt_limits(); return 0; });
}
// non-empty scheduler blocks after update_dt_limits(): node coordinates of the block's corner + its three limits +
// the number of particles in its pool.  Returns the number of blocks (-1 on error); min / max_delta_t_int in mm[2].
int64_t ref_async_blocks(void *hh, int64_t cap, int32_t *coord, int64_t *strength, int64_t *cfl, int64_t *continuous, int64_t *count,
                         int64_t *mm) {
  Handle *h = (Handle *)hh;
  int64_t n = 0;
  const int rc = guarded([&] {
    n = h->dim == 2 ? async_blocks<2>(h, cap, coord, strength, cfl, continuous, count, mm) : async_blocks<3>(h, cap, coord, strength, cfl, continuous, count, mm);
    return 0;
  });
  return rc ? -1 : n;
}
// every particle of every pool (state at its block's particle_t) with its block's limits
int64_t ref_async_download(void *hh, int64_t cap, float *x, float *v, float *F, float *B, float *aux, int32_t *id, int64_t *limits) {
  Handle *h = (Handle *)hh;
  int64_t n = 0;
  const int rc = guarded([&] {
    n = h->dim == 2 ? async_download<2>(h, cap, x, v, F, B, aux, id, limits) : async_download<3>(h, cap, x, v, F, B, aux, id, limits);
    return 0;
  });
  return rc ? -1 : n;
}
int64_t ref_async_geometry(void *hh, int64_t cap, int32_t *coord, int32_t *neighbours, int32_t *is_boundary) {
  Handle *h = (Handle *)hh;
  int64_t n = 0;
  const int rc = guarded([&] {
    n = h->dim == 2 ? async_geometry<2>(h, cap, coord, neighbours, is_boundary) : async_geometry<3>(h, cap, coord, neighbours, is_boundary);
    return 0;
  });
  return rc ? -1 : n;
}
int64_t ref_async_num_particles(void *hh) { Handle *h = (Handle *)hh; return ASYNC_DISPATCH(h, async_num); }
int64_t ref_async_time_int(void *hh) { Handle *h = (Handle *)hh; return ASYNC_DISPATCH(h, async_time_int); }
int64_t ref_async_update_counter(void *hh) { Handle *h = (Handle *)hh; return ASYNC_DISPATCH(h, async_update_counter); }


// ---- CPIC rigid coupling (3D) --------------------------------------------------------------------------------------
// add_particles(type='rigid', ...) (src/mpm.cpp:80-83 -> src/mpm_rigid_body.cpp:130-252) with the mesh handed over as
// n_tri triangles (9 floats each, mesh space).  cfg: the reference's own keys (codimensional, density, friction |
// friction0+friction1, restitution, scale, initial_position | initial_rotation | initial_velocity |
// initial_angular_velocity, rotation_axis, linear_damping, angular_damping, recenter).  script (18 floats, may be null):
// [has_pos, p0(3), vel(3), amp(3), omega | has_rot, e0(3) deg, rate(3) deg/s]:
//   scripted_position(t) = p0 + vel t + amp sin(omega t),  scripted_rotation(t) = e0 + rate t   (Euler angles, degrees)
// Returns the rigid body's index in MPM::rigids (>= 1; 0 is the background body).
int ref_add_rigid(void *hh, const char *cfg, int n_tri, const float *tri, const float *script) {
  Handle *h = (Handle *)hh;
  return guarded([&] {
    if (h->dim != 3) TC_ERROR("rigid bodies: 3D only in this driver");
    MPM<3> &m = *h->m3;
    Config c = Config::from_string(cfg);
    c.set("type", std::string("rigid"));
    c.set("shim_mesh_ptr", (unsigned long long)(uintptr_t)tri);
    c.set("shim_mesh_n", n_tri);
    if (script && script[0] != 0.0f) {
      const Vector3 p0(script[1], script[2], script[3]), vel(script[4], script[5], script[6]), amp(script[7], script[8], script[9]);
      const real omega = script[10];
      h->pos_scripts.push_back(std::make_unique<RigidBody<3>::PositionFunctionType>(
          [=](real t) { return p0 + vel * t + amp * std::sin(omega * t); }));
      c.set("scripted_position", (unsigned long long)(uintptr_t)h->pos_scripts.back().get());
      c.set("scripted_position_id", (int)h->pos_scripts.size() - 1);
    }
    if (script && script[11] != 0.0f) {
      const Vector3 e0(script[12], script[13], script[14]), rate(script[15], script[16], script[17]);
      h->rot_scripts.push_back(std::make_unique<RigidBody<3>::RotationFunctionType>([=](real t) { return e0 + rate * t; }));
      c.set("scripted_rotation", (unsigned long long)(uintptr_t)h->rot_scripts.back().get());
      c.set("scripted_rotation_id", (int)h->rot_scripts.size() - 1);
    }
    m.add_particles(c);
    m.rigids.back()->id = (int)m.rigids.size() - 1;  // (the legacy core numbered bodies with a process-wide counter)
    return (int)m.rigids.size() - 1;
  });
}
// out[33]: position 3, quaternion (w, x, y, z) 4, velocity 3, angular velocity 3, mass, inv_mass, inertia 9 (body frame,
// row-major), inv_inertia 9
int ref_rigid_state(void *hh, int id, float *out) {
  Handle *h = (Handle *)hh;
  return guarded([&] {
    MPM<3> &m = *h->m3;
    if (id < 0 || id >= (int)m.rigids.size()) TC_ERROR("no such rigid body");
    const RigidBody<3> &r = *m.rigids[id];
    int k = 0;
    for (int i = 0; i < 3; i++) out[k++] = r.position[i];
    out[k++] = r.rotation.value.qw; out[k++] = r.rotation.value.qx; out[k++] = r.rotation.value.qy; out[k++] = r.rotation.value.qz;
    for (int i = 0; i < 3; i++) out[k++] = r.velocity[i];
    for (int i = 0; i < 3; i++) out[k++] = r.angular_velocity.value[i];
    out[k++] = r.mass; out[k++] = r.inv_mass;
    mat_out<3>(r.inertia, out + k); k += 9;
    mat_out<3>(r.inv_inertia, out + k); k += 9;
    return 0;
  });
}
int ref_rigid_set_velocity(void *hh, int id, const float *v, const float *w) {
  Handle *h = (Handle *)hh;
  return guarded([&] {
    MPM<3> &m = *h->m3;
    if (id < 1 || id >= (int)m.rigids.size()) TC_ERROR("no such rigid body");
    if (v) m.rigids[id]->velocity = Vector3(v[0], v[1], v[2]);
    if (w) m.rigids[id]->angular_velocity.value = Vector3(w[0], w[1], w[2]);
    m.advect_rigid_bodies(0.0f);  // boundary particles follow (align_with_rigid_body); a zero step moves nothing
    return 0;
  });
}
// the boundary particles sampled on the body's triangles: world position, offset from the centre of mass (body frame),
// the untransformed triangle (9 floats, recentred mesh space).  Returns the count of body `id` (all bodies: id < 0).
int64_t ref_rigid_samples(void *hh, int id, int64_t cap, float *pos, float *offset, float *element, int32_t *body) {
  Handle *h = (Handle *)hh;
  int64_t n = 0;
  const int rc = guarded([&] {
    MPM<3> &m = *h->m3;
    for (auto pi : m.particles) {
      auto *p = dynamic_cast<RigidBoundaryParticle<3> *>(m.allocator[pi]);
      if (!p || (id >= 0 && p->rigid->id != id)) continue;
      if (n < cap) {
        for (int k = 0; k < 3; k++) { if (pos) pos[3 * n + k] = p->pos[k]; if (offset) offset[3 * n + k] = p->offset[k]; }
        if (element) for (int q = 0; q < 3; q++) for (int k = 0; k < 3; k++) element[9 * n + 3 * q + k] = p->untransformed_element.v[q][k];
        if (body) body[n] = p->rigid->id;
      }
      n++;
    }
    return 0;
  });
  return rc ? -1 : n;
}
// colored distance field of the grid: dense (res+1)^3 arrays of GridState::states (24 tag bits | rigid id + 1 << 24,
// src/mpm_fwd.h:69-105) and GridState::distance
int ref_download_cdf(void *hh, uint32_t *states, float *distance) {
  Handle *h = (Handle *)hh;
  return guarded([&] {
    MPM<3> &m = *h->m3;
    using Mask = typename MPM<3>::SparseMask;
    auto blocks = m.fat_page_map->Get_Blocks();
    auto bs = m.grid_block_size();
    for (unsigned b = 0; b < blocks.second; b++) {
      Vector3i base(Mask::LinearToCoord(blocks.first[b]));
      for (auto &ind : RegionND<3>(Vector3i(0), bs)) {
        Vector3i g = base + ind.get_ipos();
        if (g[0] > m.res[0] || g[1] > m.res[1] || g[2] > m.res[2]) continue;
        const size_t lin = ((size_t)g[0] * (m.res[1] + 1) + g[1]) * (m.res[2] + 1) + g[2];
        states[lin] = m.get_grid(g).states;
        distance[lin] = m.get_grid(g).distance;
      }
    }
    return 0;
  });
}
// per material particle, in the order of ref_download: MPMParticle::states, boundary_distance, boundary_normal,
// near_boundary_  (what gather_cdf leaves, src/rigid_transfer.cpp:121-275); upload = states only
int64_t ref_particle_cdf(void *hh, uint32_t *states, float *distance, float *normal, int32_t *near, const uint32_t *upload_states) {
  Handle *h = (Handle *)hh;
  int64_t n = 0;
  guarded([&] {
    MPM<3> &m = *h->m3;
    for (auto pi : m.particles) {
      MPMParticle<3> *p = m.allocator[pi];
      if (p->is_rigid()) continue;
      if (upload_states) p->states = upload_states[n];
      if (states) states[n] = p->states;
      if (distance) distance[n] = p->boundary_distance;
      if (normal) for (int k = 0; k < 3; k++) normal[3 * n + k] = p->boundary_normal[k];
      if (near) near[n] = p->near_boundary_;
      n++;
    }
    return 0;
  });
  return n;
}


// ---- CPIC in 2D (MPM<2>: generic transfers, src/transfer.cpp:193-278,585-687) -------------------------------------
// segments: n_seg x 4 floats (two end points each).  script (8 floats, may be null): [has_pos, p0(2), vel(2) | has_rot,
// a0 deg, rate deg/s]: scripted_position(t) = p0 + vel t, scripted_rotation(t) = a0 + rate t
int ref2_add_rigid(void *hh, const char *cfg, int n_seg, const float *seg, const float *script) {
  Handle *h = (Handle *)hh;
  return guarded([&] {
    if (h->dim != 2) TC_ERROR("ref2_add_rigid needs a 2D simulation");
    MPM<2> &m = *h->m2;
    Config c = Config::from_string(cfg);
    c.set("type", std::string("rigid"));
    c.set("shim_mesh_ptr", (unsigned long long)(uintptr_t)seg);
    c.set("shim_mesh_n", n_seg);
    if (script && script[0] != 0.0f) {
      const Vector2 p0(script[1], script[2]), vel(script[3], script[4]);
      h->pos_scripts2.push_back(std::make_unique<RigidBody<2>::PositionFunctionType>([=](real t) { return p0 + vel * t; }));
      c.set("scripted_position", (unsigned long long)(uintptr_t)h->pos_scripts2.back().get());
      c.set("scripted_position_id", (int)h->pos_scripts2.size() - 1);
    }
    if (script && script[5] != 0.0f) {
      const real a0 = script[6], rate = script[7];
      h->rot_scripts2.push_back(std::make_unique<RigidBody<2>::RotationFunctionType>([=](real t) { return a0 + rate * t; }));
      c.set("scripted_rotation", (unsigned long long)(uintptr_t)h->rot_scripts2.back().get());
      c.set("scripted_rotation_id", (int)h->rot_scripts2.size() - 1);
    }
    m.add_particles(c);
    m.rigids.back()->id = (int)m.rigids.size() - 1;
    return (int)m.rigids.size() - 1;
  });
}
// out[10]: position 2, angle (radians), velocity 2, angular velocity, mass, inv_mass, inertia, inv_inertia
int ref2_rigid_state(void *hh, int id, float *out) {
  Handle *h = (Handle *)hh;
  return guarded([&] {
    MPM<2> &m = *h->m2;
    if (id < 0 || id >= (int)m.rigids.size()) TC_ERROR("no such rigid body");
    const RigidBody<2> &r = *m.rigids[id];
    out[0] = r.position[0]; out[1] = r.position[1]; out[2] = r.rotation.value; out[3] = r.velocity[0]; out[4] = r.velocity[1];
    out[5] = r.angular_velocity.value; out[6] = r.mass; out[7] = r.inv_mass; out[8] = r.inertia; out[9] = r.inv_inertia;
    return 0;
  });
}
int64_t ref2_rigid_samples(void *hh, int id, int64_t cap, float *pos) {
  Handle *h = (Handle *)hh;
  int64_t n = 0;
  const int rc = guarded([&] {
    MPM<2> &m = *h->m2;
    for (auto pi : m.particles) {
      auto *p = dynamic_cast<RigidBoundaryParticle<2> *>(m.allocator[pi]);
      if (!p || (id >= 0 && p->rigid->id != id)) continue;
      if (n < cap && pos) { pos[2 * n] = p->pos[0]; pos[2 * n + 1] = p->pos[1]; }
      n++;
    }
    return 0;
  });
  return rc ? -1 : n;
}
int ref2_download_cdf(void *hh, uint32_t *states, float *distance) {  // dense (res+1)^2
  Handle *h = (Handle *)hh;
  return guarded([&] {
    MPM<2> &m = *h->m2;
    using Mask = typename MPM<2>::SparseMask;
    auto blocks = m.fat_page_map->Get_Blocks();
    auto bs = m.grid_block_size();
    for (unsigned b = 0; b < blocks.second; b++) {
      Vector2i base(Mask::LinearToCoord(blocks.first[b]));
      for (auto &ind : RegionND<2>(Vector2i(0), bs)) {
        Vector2i g = base + ind.get_ipos();
        if (g[0] > m.res[0] || g[1] > m.res[1]) continue;
        const size_t lin = (size_t)g[0] * (m.res[1] + 1) + g[1];
        states[lin] = m.get_grid(g).states;
        distance[lin] = m.get_grid(g).distance;
      }
    }
    return 0;
  });
}
int64_t ref2_particle_cdf(void *hh, uint32_t *states, float *distance, float *normal, int32_t *near) {
  Handle *h = (Handle *)hh;
  int64_t n = 0;
  guarded([&] {
    MPM<2> &m = *h->m2;
    for (auto pi : m.particles) {
      MPMParticle<2> *p = m.allocator[pi];
      if (p->is_rigid()) continue;
      if (states) states[n] = p->states;
      if (distance) distance[n] = p->boundary_distance;
      if (normal) { normal[2 * n] = p->boundary_normal[0]; normal[2 * n + 1] = p->boundary_normal[1]; }
      if (near) near[n] = p->near_boundary_;
      n++;
    }
    return 0;
  });
  return n;
}

// seconds per TC_PROFILE name since the last reset, as "name=seconds;..."
int ref_profile(char *out, size_t cap, int reset) {
  auto &r = ShimRuntime::get();
  std::string s;
  for (auto &kv : r.seconds) { char b[256]; std::snprintf(b, sizeof b, "%s=%.9g;", kv.first.c_str(), kv.second); s += b; }
  if (out && cap) std::snprintf(out, cap, "%s", s.c_str());
  if (reset) r.seconds.clear();
  return 0;
}

// ---- single-particle entry points (materials, kernels, friction) ------------------------------------------------
int ref_calculate_force(int dim, const char *type_cfg, int64_t n, const float *F, const float *aux, float *out) {
  return guarded([&] {
    Config c = Config::from_string(type_cfg);
    std::vector<uint8> buf;
    for (int64_t i = 0; i < n; i++) {
      if (dim == 3) {
        auto *p = scratch_particle<3>(c, buf);
        p->dg_e = mat_in<3>(F + 9 * i);
        if (real *a = aux_ptr<3>(p)) *a = aux[i];
        mat_out<3>(p->calculate_force(), out + 9 * i);
      } else {
        auto *p = scratch_particle<2>(c, buf);
        p->dg_e = mat_in<2>(F + 4 * i);
        if (real *a = aux_ptr<2>(p)) *a = aux[i];
        mat_out<2>(p->calculate_force(), out + 4 * i);
      }
    }
    return 0;
  });
}
int ref_plasticity(int dim, const char *type_cfg, int64_t n, const float *cdg, float *F, float *aux) {
  return guarded([&] {
    Config c = Config::from_string(type_cfg);
    std::vector<uint8> buf;
    for (int64_t i = 0; i < n; i++) {
      if (dim == 3) {
        auto *p = scratch_particle<3>(c, buf);
        p->dg_e = mat_in<3>(F + 9 * i);
        real *a = aux_ptr<3>(p);
        if (a) *a = aux[i];
        p->plasticity(mat_in<3>(cdg + 9 * i));
        mat_out<3>(p->dg_e, F + 9 * i);
        if (a) aux[i] = *a;
      } else {
        auto *p = scratch_particle<2>(c, buf);
        p->dg_e = mat_in<2>(F + 4 * i);
        real *a = aux_ptr<2>(p);
        if (a) *a = aux[i];
        p->plasticity(mat_in<2>(cdg + 4 * i));
        mat_out<2>(p->dg_e, F + 4 * i);
        if (a) aux[i] = *a;
      }
    }
    return 0;
  });
}
// get_allowed_dt(dx) and potential_energy() of one particle state (3D)
int ref_particle_scalars(const char *type_cfg, int64_t n, const float *F, const float *aux, const float *v, float dx,
                         float *allowed_dt, float *potential) {
  return guarded([&] {
    Config c = Config::from_string(type_cfg);
    std::vector<uint8> buf;
    for (int64_t i = 0; i < n; i++) {
      auto *p = scratch_particle<3>(c, buf);
      p->dg_e = mat_in<3>(F + 9 * i);
      if (real *a = aux_ptr<3>(p)) *a = aux[i];
      p->set_velocity(Vector3(v[3 * i], v[3 * i + 1], v[3 * i + 2]));
      if (allowed_dt) allowed_dt[i] = p->get_allowed_dt(dx);
      if (potential) {
        try { potential[i] = p->potential_energy(); } catch (const ShimError &) { potential[i] = std::nanf(""); }
      }
    }
    return 0;
  });
}
// MPMKernel<3,2> (kind 0), MPMFastKernel32 (kind 1): get_dw_w of the 27 stencil nodes -> out[27][4]
int ref_kernel3(int kind, const float *pos, float inv_dx, float *out) {
  return guarded([&] {
    const Vector3 p(pos[0], pos[1], pos[2]);
    int n = 0;
    if (kind == 0) {
      MPMKernel<3, 2> k(p, inv_dx);
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int l = 0; l < 3; l++, n++) {
        const Vector4 w = k.get_dw_w(Vector3i(i, j, l));
        for (int q = 0; q < 4; q++) out[4 * n + q] = w[q];
      }
    } else {
      MPMFastKernel32 k(p, inv_dx);
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int l = 0; l < 3; l++, n++) {
        const Vector4 w = k.get_dw_w(Vector3i(i, j, l));
        for (int q = 0; q < 4; q++) out[4 * n + q] = w[q];
      }
    }
    return 0;
  });
}
int ref_kernel2(const float *pos, float inv_dx, float *out) {  // MPMKernel<2,2>: out[9][3]
  return guarded([&] {
    MPMKernel<2, 2> k(Vector2(pos[0], pos[1]), inv_dx);
    int n = 0;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++, n++) {
      const Vector3 w = k.get_dw_w(Vector2i(i, j));
      for (int q = 0; q < 3; q++) out[3 * n + q] = w[q];
    }
    return 0;
  });
}
int ref_stencil_start(float x) { return MPMKernel<3, 2>::get_stencil_start(x); }
int ref_friction_project(const float *v, const float *vb, const float *n, float mu, float *out) {
  return guarded([&] {
    const Vector3 r = friction_project<3>(Vector3(v[0], v[1], v[2]), Vector3(vb[0], vb[1], vb[2]), Vector3(n[0], n[1], n[2]), mu);
    for (int k = 0; k < 3; k++) out[k] = r[k];
    return 0;
  });
}
// the shim's own svd / polar_decomp (NOT reference code): exposed so the tests can state how far the restated oracle's
// fp32 routines are from an exact decomposition
int ref_shim_svd3(const float *F, float *U, float *S, float *V) {
  Matrix3 u, s, v;
  svd(mat_in<3>(F), u, s, v);
  mat_out<3>(u, U); mat_out<3>(v, V);
  for (int i = 0; i < 3; i++) S[i] = s[i][i];
  return 0;
}

}  // extern "C"
