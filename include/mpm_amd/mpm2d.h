// include/mpm_amd/mpm2d.h — `MPM<2>` and `AsyncMPM<2>`: the reference's 2D simulation classes (src/mpm.h:56-489 with dim = 2;
// TC_IMPLEMENTATION(Simulation2D, MPM2D, "mpm"), src/mpm.cpp:983-986; TC_IMPLEMENTATION(Simulation2D, AsyncMPM2D, "async_mpm"),
// src/async/async_mpm.cpp:423-427) as a thin C++ host layer over the C ABI of libmpmhip (include/mpmhip.h: mpmhip2d_*).  Same
// method names and semantics as the 3D mirror of mpm.h, which explains the conventions (errors throw std::runtime_error with
// the library's message; no numerics here).
//   initialize(Config)  src/mpm.cpp:26-75    add_particles(Config) -> std::string  :77-270    step(real dt)  :428-439
//   substep()           :452-575             add_rigid_body(config, segments)      src/mpm_rigid_body.cpp:130-252 (dim = 2)
#pragma once
#include <algorithm>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "mpm.h"

namespace mpm_amd {

using Vector2 = VectorND<2, real>;
using Vector2i = VectorND<2, int>;

struct Particle2D {  // one row of MPM<2>::get_particles()
  Vector2 position, velocity;
  real F[4], apic_b[4], aux;
  int32_t group, id;
};

template <>
class MPM<2> {
 public:
  static constexpr int D = 2;
  using Vector = Vector2;
  using VectorI = Vector2i;
  using ScriptPosition = std::function<Vector(real)>;
  using ScriptRotation = std::function<real(real)>;  // degrees

  MPM() = default;
  MPM(const MPM &) = delete;
  MPM &operator=(const MPM &) = delete;
  virtual ~MPM() { if (ctx_) mpmhip2d_destroy(ctx_); }

  // --- MPM<2>::initialize (src/mpm.cpp:26-75)
  virtual void initialize(const Config &config) {
    if (config.has_key("delta_t")) throw std::runtime_error("Please use 'base_delta_t' instead of 'delta_t'");  // :41-42
    for (const char *k : {"gravity_cutting", "sand_climb", "sand_crawler", "stork_nod", "energy_experiment", "visualize_cdf",
                          "visualize_particle_cdf", "benchmark_rasterize", "benchmark_resample"})
      if (config.get(k, false)) throw std::runtime_error(std::string("config key '") + k + "' is not implemented by this library");
    if (config.get("expr_leaky_levelset", 0) != 0 || config.get("remove_particles", 0) != 0 ||
        config.get("coupling_iterations", 1) != 1 || config.get("cdf_expand", 0) != 0)
      throw std::runtime_error("expr_leaky_levelset / remove_particles / coupling_iterations / cdf_expand are not implemented");
    res = config.get_vec("res", VectorI(0, 0));
    if (res[0] <= 0) throw std::runtime_error("config key 'res' is required");
    delta_x = config.get("delta_x", 1.0f / res[0]);
    base_delta_t = config.get("base_delta_t", 1e-4f) * config.get("dt_multiplier", 1.0f);
    mpmhip2d_config c{};
    c.res[0] = res[0]; c.res[1] = res[1];
    c.dx = delta_x; c.dt = base_delta_t;
    const Vector g = config.get_vec("gravity", Vector(0.0f, -10.0f));
    c.gravity[0] = g[0]; c.gravity[1] = g[1];
    c.particle_gravity = config.get("particle_gravity", true);
    c.apic_damping = config.get("apic_damping", 0.0f);
    c.rpic_damping = config.get("rpic_damping", 0.0f);
    c.clean_boundary = config.get("clean_boundary", true);
    c.particle_collision = config.get("particle_collision", false);
    c.max_particles = (int64_t)config.get("max_particles", (double)(1 << 20));
    c.device = config.get("device", 0);
    verbose_bgeo = config.get("verbose_bgeo", false);   // src/visualize.cpp:22
    frame_directory = config.get("frame_directory", "");  // injected by the python driver, async_mpm.py:49
    frame_count = 0;
    check(mpmhip2d_create(&c, &ctx_), nullptr);
    check(mpmhip2d_set_rigid_coupling(ctx_, config.get("penalty", 0.0f), config.get("pushing_force", 20000.0f)), ctx_);
    check(mpmhip2d_set_articulation_iterations(ctx_, config.get("articulation_iterations", 100)), ctx_);
    check(mpmhip2d_set_rigid_levelset_collision(ctx_, config.get("rigid_body_levelset_collision", false) ? 1 : 0), ctx_);  // src/mpm.cpp:535-538
    const float d = config.get("dirichlet_boundary_radius", 0.0f), vel = config.get("dirichlet_boundary_velocity", 0.0f);  // :374-399
    check(mpmhip2d_set_dirichlet(ctx_, d > 0 ? 1 : 0, config.get("dirichlet_distance_left", d), config.get("dirichlet_distance_right", d),
                                 config.get("dirichlet_boundary_left", vel), config.get("dirichlet_boundary_right", vel)), ctx_);
  }

  // --- analytic level set in the plane (planes = lines, spheres = discs, cuboids = boxes); two key frames like the 3D mirror
  void set_levelset(const std::vector<mpmhip_shape> &shapes, real friction) {
    check(mpmhip2d_set_levelset(ctx_, (int32_t)shapes.size(), shapes.data(), -1, nullptr, 0.0f, 1.0f, friction), ctx_);
  }
  void set_levelset(real t0, real t1, const std::vector<mpmhip_shape> &shapes0, const std::vector<mpmhip_shape> &shapes1, real friction) {
    check(mpmhip2d_set_levelset(ctx_, (int32_t)shapes0.size(), shapes0.data(), (int32_t)shapes1.size(), shapes1.data(), t0, t1, friction), ctx_);
  }

  // --- MPM<2>::add_particles (src/mpm.cpp:77-270): a lattice "square_lo" / "square_hi" in cells (4 particles per cell at the
  // +-0.25 dx points), or explicit positions through the overload
  virtual std::string add_particles(const Config &config) {
    if (!config.has_key("square_lo")) throw std::runtime_error("add_particles(Config) needs 'square_lo'/'square_hi'; pass sampled positions to add_particles(config, n, x, v)");
    std::vector<float> x;
    for (int i = config.get("square_lo", 0); i < config.get("square_hi", 0); i++)
      for (int j = config.get("square_lo", 0); j < config.get("square_hi", 0); j++)
        for (int s = 0; s < 4; s++) {
          x.push_back((i + 0.5f + ((s & 1) ? 0.25f : -0.25f)) * delta_x);
          x.push_back((j + 0.5f + ((s & 2) ? 0.25f : -0.25f)) * delta_x);
        }
    return add_particles(config, (int64_t)x.size() / 2, x.data(), nullptr, 4.0f);
  }
  virtual std::string add_particles(const Config &config, int64_t n, const float *x, const float *v, float maximum = 0) {
    const std::string type = config.get("type", "");
    if (type == "rigid") throw std::runtime_error("type='rigid': hand the outline over with add_rigid_body(config, n_segments, segments)");
    if (maximum <= 0) maximum = config.get("ppc", config.get("maximum", 4.0f));
    const float vol = delta_x * delta_x / maximum;  // pow<dim>(delta_x) / maximum, :134
    const float mass = vol * config.get("density", 400.0f);
    const ParticleType t = create_particle_type(type, config, mass, vol);
    const int gid = mpmhip2d_add_group(ctx_, t.material, t.params);
    check(gid, ctx_);
    std::vector<float> xs, vs, Fs, auxs;
    const Vector v0 = config.get_vec("initial_velocity", Vector(0.0f, 0.0f));
    for (int64_t i = 0; i < n; i++) {  // "particle out of box or near boundary. Ignored." (:129-132)
      bool near = false;
      for (int k = 0; k < 2; k++) {
        const float X = x[2 * i + k] / delta_x;
        near = near || X < 7.0f || X - res[k] > -7.0f;
      }
      if (near) continue;
      for (int k = 0; k < 2; k++) { xs.push_back(x[2 * i + k]); vs.push_back(v ? v[2 * i + k] : v0[k]); }
      for (int k = 0; k < 4; k++) Fs.push_back(k % 3 == 0 ? t.initial_dg : 0.0f);
      auxs.push_back(t.initial_aux);
    }
    const int64_t m = (int64_t)auxs.size();
    if (m) check(mpmhip2d_add_particles(ctx_, gid, m, xs.data(), vs.data(), Fs.data(), nullptr, auxs.data()), ctx_);
    return "";
  }

  // --- add_particles(type='rigid') in 2D: the outline as n_segments x 4 floats (two end points each); keys as in 3D
  virtual std::string add_rigid_body(const Config &config, int64_t n_segments, const float *segments, ScriptPosition scripted_position = nullptr,
                                     ScriptRotation scripted_rotation = nullptr) {
    if (!config.has_key("codimensional")) throw std::runtime_error("rigid bodies need the key 'codimensional'");
    if (!scripted_position && !config.has_key("initial_position"))
      throw std::runtime_error("Please specify one (and only one) of 'scripted_position' and 'initial_position'.");
    mpmhip2d_rigid_config r{};
    r.codimensional = config.get("codimensional", true);
    r.recenter = config.get("recenter", true);
    r.reverse_vertices = config.get("reverse_vertices", false);
    r.density = config.get("density", 0.0f);
    r.friction[0] = config.get("friction0", config.get("friction", 0.0f));
    r.friction[1] = config.get("friction1", config.get("friction", 0.0f));
    r.restitution = config.get("restitution", 0.0f);
    auto vec = [&](const char *key, const Vector &d, float *out) { const Vector q = config.get_vec(key, d); out[0] = q[0]; out[1] = q[1]; };
    vec("scale", Vector(1.0f, 1.0f), r.scale);
    vec("initial_position", Vector(0.0f, 0.0f), r.initial_position);
    vec("initial_velocity", Vector(0.0f, 0.0f), r.initial_velocity);
    r.initial_rotation = config.get("initial_rotation", 0.0f);
    r.initial_angular_velocity = config.get("initial_angular_velocity", 0.0f);
    r.linear_damping = config.get("linear_damping", 0.0f);
    r.angular_damping = config.get("angular_damping", 0.0f);
    if (scripted_position) {
      pos_scripts_.push_back(std::make_unique<ScriptPosition>(std::move(scripted_position)));
      r.scripted_position = [](void *user, float t, float out[3]) { const Vector p = (*static_cast<ScriptPosition *>(user))(t); out[0] = p[0]; out[1] = p[1]; };
      r.position_user = pos_scripts_.back().get();
    }
    if (scripted_rotation) {
      rot_scripts_.push_back(std::make_unique<ScriptRotation>(std::move(scripted_rotation)));
      r.scripted_rotation = [](void *user, float t, float out[3]) { out[0] = (*static_cast<ScriptRotation *>(user))(t); };
      r.rotation_user = rot_scripts_.back().get();
    }
    const int id = mpmhip2d_add_rigid_body(ctx_, &r, n_segments, segments);
    check(id, ctx_);
    return std::to_string(id);
  }
  // general_action(action='add_articulation', type='rotation', obj0, obj1) — the joint of scripts/mls-cpic/sand_wheel_2D.py:88
  std::string add_articulation(const Config &config) {
    if (std::string(config.get("type", "")) != "rotation") throw std::runtime_error("the 2D simulation has the 'rotation' joint only");
    mpmhip_joint_config j{};
    j.type = 0;
    j.obj0 = config.get("obj0", 0);
    j.obj1 = config.get("obj1", 0);
    check(mpmhip2d_add_articulation(ctx_, &j), ctx_);
    return "";
  }
  // position 2, angle, velocity 2, angular velocity, mass, inv_mass, inertia, inv_inertia
  std::vector<float> get_rigid_state(int id) const {
    std::vector<float> o(10);
    check(mpmhip2d_rigid_get_state(ctx_, id, o.data()), ctx_);
    return o;
  }

  // --- time stepping
  virtual void step(real dt) { check(mpmhip2d_step(ctx_, dt), ctx_); frame++; }  // src/mpm.cpp:428-439 (dt < 0: one substep)
  virtual void substep() { check(mpmhip2d_substep(ctx_), ctx_); }         // :452-575
  virtual real get_current_time() const { return (real)mpmhip2d_current_time(ctx_); }
  virtual int64_t get_num_particles() const { const int64_t n = mpmhip2d_num_particles(ctx_); check((int)std::min<int64_t>(n, 0), ctx_); return n; }
  // every live particle, ordered by creation id (slots are not a stable handle)
  virtual std::vector<Particle2D> get_particles() const { return particles_of_arrays(); }
  std::vector<float> get_grid() const {  // (v.x, v.y, m) per node of the (res+1)^2 grid after the last substep
    std::vector<float> g((size_t)3 * (res[0] + 1) * (res[1] + 1));
    check(mpmhip2d_download_grid(ctx_, g.data()), ctx_);
    return g;
  }
  // --- frame output: visualize() -> write_bgeo() -> write_partio(file) (src/visualize.cpp:156-159, src/mpm.h:333-337,
  // src/visualize.cpp:17-100): the same .bgeo as the 3D simulation writes, z = 0; with the asynchronous stepper every container of
  // every particle pool with its block's limits
  void write_partio(const std::string &file_name) const { check(mpmhip2d_write_bgeo(ctx_, file_name.c_str(), verbose_bgeo), ctx_); }
  std::string write_bgeo() {
    if (frame_directory.empty()) throw std::runtime_error("write_bgeo() needs the config key 'frame_directory'");
    char name[32];
    std::snprintf(name, sizeof name, "/%04d.bgeo", ++frame_count);  // frames start at 1 (src/mpm.h:334-336)
    write_partio(frame_directory + name);
    return frame_directory + name;
  }
  void visualize() { write_bgeo(); }
  // --- MPM<2>::general_action (src/mpm.cpp:920-978): add_articulation, save / load (whole-state snapshot to / from "file_name";
  // the scene — level set, configuration, the rigid bodies — is set up again before a load, as in the reference)
  std::string general_action(const Config &config) {
    const std::string action = config.get("action", "");
    if (action == "add_articulation") return add_articulation(config);
    if (action == "save" || action == "load") {
      const std::string fn = config.get("file_name", "");
      if (action == "save") {
        const int64_t n = mpmhip2d_snapshot_size(ctx_);
        check((int)std::min<int64_t>(n, 0), ctx_);
        std::vector<char> buf((size_t)n);
        check(mpmhip2d_snapshot_save(ctx_, buf.data(), buf.size()), ctx_);
        FILE *f = std::fopen(fn.c_str(), "wb");
        if (!f || std::fwrite(buf.data(), 1, buf.size(), f) != buf.size()) throw std::runtime_error("cannot write " + fn);
        std::fclose(f);
      } else {
        FILE *f = std::fopen(fn.c_str(), "rb");
        if (!f) throw std::runtime_error("cannot read " + fn);
        std::fseek(f, 0, SEEK_END);
        std::vector<char> buf((size_t)std::ftell(f));
        std::fseek(f, 0, SEEK_SET);
        if (std::fread(buf.data(), 1, buf.size(), f) != buf.size()) throw std::runtime_error("cannot read " + fn);
        std::fclose(f);
        check(mpmhip2d_snapshot_load(ctx_, buf.data(), buf.size()), ctx_);
      }
      return "";
    }
    throw std::runtime_error("general_action(action='" + action + "') is outside the scope of this build");
  }
  bool test() const { return true; }
  virtual std::string get_name() const { return "mpm"; }
  mpmhip2d_ctx *ctx() const { return ctx_; }

  VectorI res;
  real delta_x = 0, base_delta_t = 0;
  int frame = 0, frame_count = 0;
  bool verbose_bgeo = false;
  std::string frame_directory;

 protected:
  std::vector<Particle2D> particles_of_arrays() const {
    const int64_t n = mpmhip2d_num_particles(ctx_);
    check((int)std::min<int64_t>(n, 0), ctx_);
    std::vector<float> x(2 * n), v(2 * n), F(4 * n), B(4 * n), aux(n);
    std::vector<int32_t> gid(n), id(n);
    const int64_t got = mpmhip2d_download(ctx_, n, x.data(), v.data(), F.data(), B.data(), aux.data(), gid.data(), id.data());
    check((int)std::min<int64_t>(got, 0), ctx_);
    std::vector<Particle2D> out((size_t)got);
    for (int64_t i = 0; i < got; i++) {
      out[i].position = Vector(x[2 * i], x[2 * i + 1]);
      out[i].velocity = Vector(v[2 * i], v[2 * i + 1]);
      for (int k = 0; k < 4; k++) { out[i].F[k] = F[4 * i + k]; out[i].apic_b[k] = B[4 * i + k]; }
      out[i].aux = aux[i]; out[i].group = gid[i]; out[i].id = id[i];
    }
    std::stable_sort(out.begin(), out.end(), [](const Particle2D &a, const Particle2D &b) { return a.id < b.id; });
    return out;
  }
  static void check(int rc, const mpmhip2d_ctx *c) {
    if (rc < 0) throw std::runtime_error(std::string("libmpmhip error ") + std::to_string(rc) + ": " + mpmhip2d_last_error(c));
  }
  mpmhip2d_ctx *ctx_ = nullptr;
  std::vector<std::unique_ptr<ScriptPosition>> pos_scripts_;  // scripted motions of rigid bodies (called back by the library)
  std::vector<std::unique_ptr<ScriptRotation>> rot_scripts_;
};

using MPM2D = MPM<2>;

// `AsyncMPM<2>`: the asynchronous stepper over the C ABI's "AsyncMPM<2>" (include/mpmhip.h): pools, backups, gathers and the walk
// over the power-of-two levels run inside the library with every particle resident on the device; this class forwards.
class AsyncMPM2D : public MPM2D {
 public:
  void initialize(const Config &config) override {  // AsyncMPM<dim>::initialize, src/async/async_mpm.cpp:13-55
    MPM2D::initialize(config);
    mpmhip_async_config a{};
    a.unit_delta_t = config.get("unit_delta_t", 1e-6f);  // :24-27
    a.max_units = (int64_t)config.get("max_units", 8192.0);
    a.cfl_dt_mul = config.get("cfl_dt_mul", 1.0f);
    a.strength_dt_mul = config.get("strength_dt_mul", 1.0f);
    a.left_boundary = config.get("left_boundary", false) ? 1 : 0;  // :43-53
    check(mpmhip2d_async_begin(ctx_, &a), ctx_);
  }
  std::string add_particles(const Config &config) override {  // :57-75: the new particles go to their blocks' pools
    const std::string r = MPM2D::add_particles(config);
    check(mpmhip2d_async_pool_particles(ctx_), ctx_);
    return r;
  }
  std::string add_particles(const Config &config, int64_t n, const float *x, const float *v, float maximum = 0) override {
    const std::string r = MPM2D::add_particles(config, n, x, v, maximum);
    check(mpmhip2d_async_pool_particles(ctx_), ctx_);
    return r;
  }
  std::string add_rigid_body(const Config &, int64_t, const float *, ScriptPosition = nullptr, ScriptRotation = nullptr) override {
    throw std::runtime_error("rigid bodies cannot be combined with asynchronous stepping");
  }
  void step(real dt) override { check(mpmhip2d_async_step(ctx_, dt), ctx_); frame++; }  // :380-421
  void substep() override { throw std::runtime_error("AsyncMPM steps with step(dt)"); }
  real get_current_time() const override { return (real)mpmhip2d_async_current_time(ctx_); }
  // the views of "the particles" list every container of every particle pool, as AsyncMPM<dim>::visualize does
  // (src/async/async_visualize.cpp:86-96)
  int64_t get_num_particles() const override {
    const int64_t n = mpmhip2d_async_load_pools(ctx_);
    check((int)std::min<int64_t>(n, 0), ctx_);
    return n;
  }
  std::vector<Particle2D> get_particles() const override {
    check((int)std::min<int64_t>(mpmhip2d_async_load_pools(ctx_), 0), ctx_);
    return particles_of_arrays();
  }
  std::string get_name() const override { return "async_mpm"; }  // src/async/async_mpm.h:251-253
  int64_t current_t_int() const { int64_t o[8]; check(mpmhip2d_async_state(ctx_, o), ctx_); return o[0]; }
  int64_t update_counter() const { int64_t o[8]; check(mpmhip2d_async_state(ctx_, o), ctx_); return o[1]; }
};

// the factory the reference reaches through `create_instance<Simulation2D>(name)`
inline std::unique_ptr<MPM2D> create_simulation2(const std::string &name) {
  if (name == "async_mpm") return std::make_unique<AsyncMPM2D>();
  if (name != "mpm") throw std::runtime_error("no Simulation2D implementation named '" + name + "' (registered: 'mpm', 'async_mpm')");
  return std::make_unique<MPM2D>();
}

}  // namespace mpm_amd
