// include/mpm_amd/mpm.h — `MPM<3>`: the reference's simulation class (src/mpm.h:56-489, src/mpm.cpp) as a thin
// C++ host layer over the C ABI of libmpmhip (include/mpmhip.h).  Same method names and semantics as the
// reference's `Simulation3D` implementation registered as "mpm" (src/mpm.cpp:983-988):
//   initialize(Config)            src/mpm.cpp:26-75     add_particles(Config) -> std::string   :77-270
//   step(real dt)                 :428-439              substep()                                :452-575
//   get_current_time()            src/mpm.h:99          general_action(Config) -> std::string   :920-978
//   sort_particles_and_populate_grid / rasterize_optimized / normalize_grid_and_apply_external_force +
//   apply_grid_boundary_conditions / resample_optimized: the phase functions, one C-ABI call each.
// All device work happens in the HIP kernels behind the ABI; this header holds no numerics.  Errors throw
// std::runtime_error with the library's message (reference: TC_ASSERT / TC_ERROR abort).
// CPIC rigid bodies: add_rigid_body(config, triangles) = add_particles(type='rigid', ...) with the mesh handed over as
// triangles; scripted motions are std::function<Vector(real)> like the reference's (src/mpm_rigid_body.cpp:79-92).
// Joints: add_articulation(config) = general_action(action='add_articulation', ...) (src/mpm.cpp:923-933, src/articulation.cpp).
// Out of scope, as in DESIGN.md §7: textures/meshes in add_particles (the mesh LOADER), rigid-rigid collisions, rendering.
#pragma once
#include <algorithm>
#include <cstdio>
#include <functional>
#include <memory>
#include <stdexcept>
#include <array>
#include <string>
#include <vector>

#include "../mpmhip.h"
#include "kernel.h"
#include "particles.h"

namespace mpm_amd {

struct RenderParticle {  // what get_render_particles() hands to a renderer (src/visualize.cpp:102-154)
  Vector3 position, velocity;
  int32_t id;
};

template <int dim>
class MPM;

template <>
class MPM<3> {
 public:
  static constexpr int D = 3;
  using Vector = Vector3;
  using VectorI = Vector3i;

  MPM() = default;
  MPM(const MPM &) = delete;
  MPM &operator=(const MPM &) = delete;
  virtual ~MPM() { if (ctx_) mpmhip_destroy(ctx_); }

  // --- MPM<dim>::initialize (src/mpm.cpp:26-75; config keys README.md:234-256)
  virtual void initialize(const Config &config) {
    if (config.has_key("delta_t")) throw std::runtime_error("Please use 'base_delta_t' instead of 'delta_t'");  // :41-42
    // physics-changing keys of the reference that this library does not implement: refused, not ignored
    for (const char *k : {"gravity_cutting", "sand_climb", "sand_crawler", "stork_nod", "energy_experiment",
                          "visualize_cdf", "visualize_particle_cdf", "benchmark_rasterize", "benchmark_resample"})
      if (config.get(k, false)) throw std::runtime_error(std::string("config key '") + k + "' is not implemented by this library");
    if (config.get("expr_leaky_levelset", 0) != 0 || config.get("remove_particles", 0) != 0 ||
        config.get("coupling_iterations", 1) != 1 || config.get("cdf_expand", 0) != 0)
      throw std::runtime_error("expr_leaky_levelset / remove_particles / coupling_iterations / cdf_expand are not implemented");
    dirichlet_ = config.get("dirichlet_boundary_radius", 0.0f) > 0;  // :541-544 -> apply_dirichlet_boundary_conditions (:401-412)
    res = config.get_vec("res", VectorI(0, 0, 0));
    if (res[0] <= 0) throw std::runtime_error("config key 'res' is required");
    delta_x = config.get("delta_x", 1.0f / res[0]);
    base_delta_t = config.get("base_delta_t", 1e-4f) * config.get("dt_multiplier", 1.0f);
    cfg_ = mpmhip_config{};
    for (int k = 0; k < 3; k++) cfg_.res[k] = res[k];
    cfg_.dx = delta_x;
    cfg_.dt = base_delta_t;
    const Vector g = config.get_vec("gravity", Vector(0.0f, -10.0f, 0.0f));
    for (int k = 0; k < 3; k++) cfg_.gravity[k] = g[k];
    cfg_.particle_gravity = config.get("particle_gravity", true);  // :47
    cfg_.apic_damping = config.get("apic_damping", 0.0f);
    cfg_.rpic_damping = config.get("rpic_damping", 0.0f);
    cfg_.clean_boundary = config.get("clean_boundary", true);
    cfg_.reorder_interval = config.get("reorder_interval", 1000);  // :45
    cfg_.max_particles = (int64_t)config.get("max_particles", (double)(1 << 25));  // :773-775
    cfg_.max_blocks = (int64_t)config.get("max_blocks", 0.0);
    cfg_.device = config.get("device", 0);
    cfg_.discard_apic_b = !config.get("keep_apic_b", keep_apic_b_default());
    cfg_.generic_path = !config.get("optimized", true);  // src/mpm.cpp:508-515,546-552
    cfg_.deterministic = config.get("deterministic", false);  // (no reference key: bitwise reproducible runs, include/mpmhip.h)
    cfg_.particle_collision = config.get("particle_collision", false);  // src/mpm.cpp:566-569
    verbose_bgeo = config.get("verbose_bgeo", false);   // src/visualize.cpp:22
    frame_directory = config.get("frame_directory", "");  // injected by the python driver, async_mpm.py:49
    check(mpmhip_create(&cfg_, &ctx_), nullptr);
    // CPIC coupling constants (src/mpm.cpp:35,40)
    check(mpmhip_set_rigid_coupling(ctx_, config.get("penalty", 0.0f), config.get("pushing_force", 20000.0f)), ctx_);
    check(mpmhip_set_articulation_iterations(ctx_, config.get("articulation_iterations", 100)), ctx_);  // src/mpm.h:279-280
    check(mpmhip_set_dirichlet(ctx_, dirichlet_ ? 1 : 0), ctx_);
    check(mpmhip_set_rigid_levelset_collision(ctx_, config.get("rigid_body_levelset_collision", false) ? 1 : 0), ctx_);  // src/mpm.cpp:535-538
    frame = 0;
    frame_count = 0;
  }

  // --- analytic level set (set_levelset(DynamicLevelSet) of the reference, for half-spaces): phi = n.x + d
  void set_levelset(const std::vector<Vector4> &planes, real friction) {
    std::vector<float> flat;
    for (auto &p : planes) flat.insert(flat.end(), p.begin(), p.end());
    check(mpmhip_set_levelset(ctx_, (int32_t)planes.size(), flat.data(), friction), ctx_);
  }

  // general form: planes, spheres, axis-aligned cuboids (taichi LevelSet::add_plane / add_sphere / add_cuboid)
  void set_levelset(const std::vector<mpmhip_shape> &shapes, real friction) {
    check(mpmhip_set_levelset_shapes(ctx_, (int32_t)shapes.size(), shapes.data(), friction), ctx_);
  }

  // DynamicLevelSet(t0, t1, levelset(t0), levelset(t1)) as the python driver installs before every frame
  // (scripts/async/async_mpm.py:119-127): two key frames blended linearly in time, boundary velocity included
  void set_levelset(real t0, real t1, const std::vector<mpmhip_shape> &shapes0, const std::vector<mpmhip_shape> &shapes1,
                    real friction) {
    check(mpmhip_set_levelset_keyframes(ctx_, t0, t1, (int32_t)shapes0.size(), shapes0.data(), (int32_t)shapes1.size(),
                                        shapes1.data(), friction), ctx_);
  }

  // --- MPM<dim>::add_particles (src/mpm.cpp:77-270).  Sampling: the built-in benchmark generator
  // ("benchmark" = 125 | 8000, :149-186), a lattice "cube_lo"/"cube_hi" in cells, or explicit arrays through
  // the overload below.  Returns "" (the reference returns a rigid-body id only for type "rigid").
  virtual std::string add_particles(const Config &config) {
    std::vector<float> x;
    float maximum = config.get("ppc", config.get("maximum", 8.0f));
    if (config.get("benchmark", 0)) {
      const int b = config.get("benchmark", 0);
      float s;
      if (b == 125) s = 0.1f;
      else if (b == 8000) s = 0.4f;
      else throw std::runtime_error("s must be 125 or 8000");  // :162
      const int lower = (int)std::lround(res[0] * (0.5f - s)), higher = lower + (int)std::lround(res[0] * 2 * s);
      lattice(lower, higher, x);
      maximum = 1.0f;  // create_particle(..., 1, config): vol = dx^3 (:178)
    } else if (config.has_key("cube_lo")) {
      lattice(config.get("cube_lo", 0), config.get("cube_hi", 0), x);
    } else {
      throw std::runtime_error("add_particles(Config) needs 'benchmark' or 'cube_lo'/'cube_hi'; pass sampled positions "
                               "to add_particles(config, n, x, v)");
    }
    return add_particles(config, (int64_t)x.size() / 3, x.data(), nullptr, maximum);
  }

  // --- add_particles(type='rigid') (src/mpm.cpp:80-83 -> MPM::add_rigid_particle, src/mpm_rigid_body.cpp:130-252): the
  // mesh is handed over as n_triangles x 9 floats; config keys as in the scene scripts (codimensional is mandatory;
  // density, friction | friction0 + friction1, restitution, scale, initial_position, initial_rotation (Euler, degrees),
  // initial_velocity, initial_angular_velocity, rotation_axis, linear_damping, angular_damping, recenter,
  // reverse_vertices).  Scripted motions: t -> position, t -> Euler angles in degrees.  Returns the body's index as the
  // reference does (a string, >= "1").
  using ScriptFunction = std::function<Vector(real)>;
  std::string add_rigid_body(const Config &config, int64_t n_triangles, const float *triangles, ScriptFunction scripted_position = nullptr,
                             ScriptFunction scripted_rotation = nullptr) {
    if (!config.has_key("codimensional")) throw std::runtime_error("rigid bodies need the key 'codimensional'");
    if (config.has_key("friction") && (config.has_key("friction0") || config.has_key("friction1")))
      throw std::runtime_error("friction and friction0 / friction1 cannot coexist!");  // src/mpm_rigid_body.cpp:41-55
    if (!scripted_position && !config.has_key("initial_position"))
      throw std::runtime_error("Please specify one (and only one) of 'scripted_position' and 'initial_position'.");
    mpmhip_rigid_config r{};
    r.codimensional = config.get("codimensional", true);
    r.recenter = config.get("recenter", true);
    r.reverse_vertices = config.get("reverse_vertices", false);
    r.density = config.get("density", 0.0f);
    r.friction[0] = config.get("friction0", config.get("friction", 0.0f));
    r.friction[1] = config.get("friction1", config.get("friction", 0.0f));
    r.restitution = config.get("restitution", 0.0f);
    auto vec = [&](const char *key, const Vector &d, float *out) { const Vector v = config.get_vec(key, d); for (int k = 0; k < 3; k++) out[k] = v[k]; };
    vec("scale", Vector(1.0f, 1.0f, 1.0f), r.scale);
    vec("initial_position", Vector(0.0f, 0.0f, 0.0f), r.initial_position);
    vec("initial_rotation", Vector(0.0f, 0.0f, 0.0f), r.initial_rotation);
    vec("initial_velocity", Vector(0.0f, 0.0f, 0.0f), r.initial_velocity);
    vec("initial_angular_velocity", Vector(0.0f, 0.0f, 0.0f), r.initial_angular_velocity);
    vec("rotation_axis", Vector(0.0f, 0.0f, 0.0f), r.rotation_axis);
    r.linear_damping = config.get("linear_damping", 0.0f);
    r.angular_damping = config.get("angular_damping", 0.0f);
    // the library calls back once per scripted body and substep; the std::function objects live as long as this MPM
    auto trampoline = [](void *user, float t, float out[3]) {
      const Vector v = (*static_cast<ScriptFunction *>(user))(t);
      for (int k = 0; k < 3; k++) out[k] = v[k];
    };
    if (scripted_position) { scripts_.push_back(std::make_unique<ScriptFunction>(std::move(scripted_position))); r.scripted_position = trampoline; r.position_user = scripts_.back().get(); }
    if (scripted_rotation) { scripts_.push_back(std::make_unique<ScriptFunction>(std::move(scripted_rotation))); r.scripted_rotation = trampoline; r.rotation_user = scripts_.back().get(); }
    const int id = mpmhip_add_rigid_body(ctx_, &r, n_triangles, triangles);
    check(id, ctx_);
    return std::to_string(id);
  }
  bool has_rigid_body() const { return mpmhip_num_rigid_bodies(ctx_) > 1; }  // src/mpm.h:240-242
  // general_action(action='add_articulation', type=..., obj0=..., obj1=..., ...) (src/mpm.cpp:923-933): a joint between two
  // rigid bodies with the keys of src/articulation.cpp; obj1 absent = the background body
  std::string add_articulation(const Config &config) {
    static const char *names[] = {"rotation", "frozen", "distance", "axial_rotation", "motor", "stepper"};
    const std::string type = config.get("type", "");
    mpmhip_joint_config j{};
    j.type = -1;
    for (int k = 0; k < 6; k++) if (type == names[k]) j.type = k;
    if (j.type < 0) throw std::runtime_error("unknown articulation type '" + type + "'");
    if (!config.has_key("obj0")) throw std::runtime_error("add_articulation needs 'obj0'");
    j.obj0 = config.get("obj0", 0);
    j.obj1 = config.get("obj1", 0);
    auto vec = [&](const char *key, float *out) { const Vector v = config.get_vec(key, Vector(0.0f, 0.0f, 0.0f)); for (int k = 0; k < 3; k++) out[k] = v[k]; };
    vec("offset0", j.offset0);
    vec("offset1", j.offset1);
    vec("axis", j.axis);
    j.has_offset1 = config.has_key("offset1");
    j.has_target_distance = config.has_key("target_distance");
    j.target_distance = config.get("target_distance", 0.0f);
    j.penalty = config.get("penalty", -1.0f);
    j.axis_length = config.get("axis_length", -1.0f);
    j.power = config.get("power", 0.0f);
    j.angular_velocity = config.get("angular_velocity", 0.0f);
    check(mpmhip_add_articulation(ctx_, &j), ctx_);
    return "";
  }
  // position 3, rotation quaternion (w,x,y,z) 4, velocity 3, angular velocity 3, mass, inv_mass, inertia 9, inv_inertia 9
  std::vector<float> get_rigid_state(int id) const {
    std::vector<float> o(33);
    check(mpmhip_rigid_get_state(ctx_, id, o.data()), ctx_);
    return o;
  }

  virtual std::string add_particles(const Config &config, int64_t n, const float *x, const float *v, float maximum = 0) {
    const std::string type = config.get("type", "");
    if (type == "rigid") throw std::runtime_error("type='rigid': hand the mesh over with add_rigid_body(config, n_triangles, triangles)");
    if (maximum <= 0) maximum = config.get("ppc", config.get("maximum", 8.0f));
    const float vol = delta_x * delta_x * delta_x / maximum;  // :134-135
    const float mass = vol * config.get("density", 400.0f);
    const ParticleType t = create_particle_type(type, config, mass, vol);
    const int gid = mpmhip_add_group(ctx_, t.material, t.params);
    check(gid, ctx_);
    // "particle out of box or near boundary. Ignored." (:129-132, src/mpm.h:269-276)
    std::vector<float> xs, vs, Fs, auxs;
    const Vector v0 = config.get_vec("initial_velocity", Vector(0.0f, 0.0f, 0.0f));
    for (int64_t i = 0; i < n; i++) {
      bool near = false;
      for (int k = 0; k < 3; k++) {
        const float X = x[3 * i + k] / delta_x;
        near = near || X < 7.0f || X - res[k] > -7.0f;
      }
      if (near) continue;
      for (int k = 0; k < 3; k++) {
        xs.push_back(x[3 * i + k]);
        vs.push_back(v ? v[3 * i + k] : v0[k]);
      }
      for (int k = 0; k < 9; k++) Fs.push_back(k % 4 == 0 ? t.initial_dg : 0.0f);
      auxs.push_back(t.initial_aux);
    }
    const int64_t m = (int64_t)auxs.size();
    if (m) check(mpmhip_add_particles(ctx_, gid, m, xs.data(), vs.data(), Fs.data(), nullptr, auxs.data()), ctx_);
    types_.push_back(t);
    return "";
  }

  // --- time stepping
  virtual void step(real dt) { check(mpmhip_step(ctx_, dt), ctx_); frame++; }  // src/mpm.cpp:428-439 (dt < 0: one substep)
  void substep() { check(mpmhip_substep(ctx_), ctx_); }              // :452-575
  virtual real get_current_time() const { return (real)mpmhip_current_time(ctx_); }
  void synchronize() { check(mpmhip_synchronize(ctx_), ctx_); }

  // --- the phases of substep(), under the reference's names
  void sort_particles_and_populate_grid() { check(mpmhip_sort(ctx_), ctx_); }        // src/mpm.cpp:770-918
  void rasterize_optimized() { check(mpmhip_p2g(ctx_), ctx_); }                      // src/transfer.cpp:361-581
  void normalize_grid_and_apply_external_force() { check(mpmhip_grid_update(ctx_), ctx_); }  // src/mpm.cpp:277-294 (+ :296-372)
  void apply_grid_boundary_conditions() {}  // fused into the grid kernel above (src/mpm.cpp:296-372)
  void resample_optimized() { check(mpmhip_g2p(ctx_), ctx_); }                       // src/transfer.cpp:702-970

  virtual int64_t get_num_particles() const { const int64_t n = mpmhip_num_particles(ctx_); check((int)std::min<int64_t>(n, 0), ctx_); return n; }

  virtual std::vector<RenderParticle> get_render_particles() const { return get_render_particles_of_records(); }
  std::vector<RenderParticle> get_render_particles_of_records() const {
    const int64_t n = mpmhip_num_particles(ctx_);
    check((int)std::min<int64_t>(n, 0), ctx_);
    std::vector<float> x(3 * n), v(3 * n);
    std::vector<int32_t> id(n);
    check(mpmhip_download(ctx_, MPMHIP_F_X, x.data(), n), ctx_);
    check(mpmhip_download(ctx_, MPMHIP_F_V, v.data(), n), ctx_);
    check(mpmhip_download(ctx_, MPMHIP_F_ID, id.data(), n), ctx_);
    std::vector<RenderParticle> out(n);
    for (int64_t i = 0; i < n; i++) {
      out[i].position = Vector(x[3 * i], x[3 * i + 1], x[3 * i + 2]);
      out[i].velocity = Vector(v[3 * i], v[3 * i + 1], v[3 * i + 2]);
      out[i].id = id[i];
    }
    // slots are the order of the last sort, not a stable handle: a particle is identified by its creation id
    std::sort(out.begin(), out.end(), [](const RenderParticle &a, const RenderParticle &b) { return a.id < b.id; });
    return out;
  }

  // --- frame output: visualize() -> write_bgeo() -> write_partio(file) (src/visualize.cpp:156-159, src/mpm.h:333-337,
  // src/visualize.cpp:17-100).  The Houdini .bgeo bytes equal the reference's (rows assembled on the device).
  virtual void write_partio(const std::string &file_name) const { check(mpmhip_write_bgeo(ctx_, file_name.c_str(), verbose_bgeo), ctx_); }
  std::string write_bgeo() {
    if (frame_directory.empty()) throw std::runtime_error("write_bgeo() needs the config key 'frame_directory'");
    char name[32];
    std::snprintf(name, sizeof name, "/%04d.bgeo", ++frame_count);  // frames start at 1 (src/mpm.h:334-336)
    const std::string file_name = frame_directory + name;
    write_partio(file_name);
    // "Start from 1. (0 is the background rigid body.)" — every body's mesh in world space next to the frame (src/mpm.h:338-343)
    for (int i = 1; i < mpmhip_num_rigid_bodies(ctx_); i++) {
      std::snprintf(name, sizeof name, "/rigid_%03d_%04d", i, frame_count);
      write_rigid_body(i, frame_directory + name);
    }
    return file_name;
  }
  // MPM<3>::write_rigid_body (src/visualize.cpp:131-153): file_name.obj, three `v` records and one `f` record per triangle
  void write_rigid_body(int id, const std::string &file_name) const {
    const int64_t n = mpmhip_rigid_get_mesh(ctx_, id, 0, nullptr);
    check((int)n, ctx_);
    std::vector<float> tri((size_t)n * 9);
    if (n) check((int)mpmhip_rigid_get_mesh(ctx_, id, n, tri.data()), ctx_);
    FILE *f = std::fopen((file_name + ".obj").c_str(), "w");
    if (!f) throw std::runtime_error("cannot write " + file_name + ".obj");
    for (int64_t k = 0; k < 3 * n; k++) std::fprintf(f, "v %.9g %.9g %.9g\n", tri[3 * k], tri[3 * k + 1], tri[3 * k + 2]);
    for (int64_t k = 0; k < n; k++) std::fprintf(f, "f %lld %lld %lld\n", (long long)(3 * k + 1), (long long)(3 * k + 2), (long long)(3 * k + 3));
    std::fclose(f);
  }
  void visualize() { write_bgeo(); }

  // --- MPM<dim>::general_action (src/mpm.cpp:920-978): the actions that only need the hot path's state
  std::string general_action(const Config &config) {
    const std::string action = config.get("action", "");
    if (action == "add_articulation") return add_articulation(config);  // :923-933
    if (action == "calculate_energy") {  // :936-938 -> calculate_energy(), :1078-1110
      double kinetic = 0, potential = 0;
      check(mpmhip_calculate_energy(ctx_, &kinetic, &potential), ctx_);
      return std::to_string(kinetic + potential);
    }
    if (action == "save" || action == "load") {  // :940-960: whole-state snapshot to / from "file_name"
      const std::string fn = config.get("file_name", "");
      if (action == "save") {
        std::vector<char> buf((size_t)mpmhip_snapshot_size(ctx_));
        check(mpmhip_snapshot_save(ctx_, buf.data(), buf.size()), ctx_);
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
        check(mpmhip_snapshot_load(ctx_, buf.data(), buf.size()), ctx_);
      }
      return "";
    }
    if (action == "delete_particles_inside_level_set") {  // :962-974
      int64_t deleted = 0;
      check(mpmhip_delete_particles_inside_level_set(ctx_, &deleted), ctx_);
      return "";
    }
    throw std::runtime_error("general_action(action='" + action + "') is outside the scope of this build");
  }

  bool test() const { return true; }                          // src/mpm.cpp:577-580
  std::string get_debug_information() const { return ""; }    // :635-639
  // what the reference's Python driver sizes its video frames with (scripts/async/async_mpm.py:79-81; the method itself is
  // the absent taichi core's): the grid resolution in x and y
  std::array<int, 2> get_vis_resolution() const { return {res[0], res[1]}; }
  virtual std::string get_name() const { return "mpm"; }      // src/mpm.h:486-488
  mpmhip_ctx *ctx() const { return ctx_; }

  VectorI res;
  real delta_x = 0, base_delta_t = 0;
  int frame = 0, frame_count = 0;
  bool verbose_bgeo = false;
  std::string frame_directory;

 protected:
  virtual bool keep_apic_b_default() const { return false; }
  void lattice(int lower, int higher, std::vector<float> &x) const {  // src/mpm.cpp:164-180: cell centre +- 0.25 dx
    for (int i = lower; i < higher; i++)
      for (int j = lower; j < higher; j++)
        for (int k = lower; k < higher; k++)
          for (int s = 0; s < 8; s++) {
            x.push_back((i + 0.5f + ((s & 1) ? 0.25f : -0.25f)) * delta_x);
            x.push_back((j + 0.5f + ((s & 2) ? 0.25f : -0.25f)) * delta_x);
            x.push_back((k + 0.5f + ((s & 4) ? 0.25f : -0.25f)) * delta_x);
          }
  }
  static void check(int rc, const mpmhip_ctx *c) {
    if (rc < 0) throw std::runtime_error(std::string("libmpmhip error ") + std::to_string(rc) + ": " + mpmhip_last_error(c));
  }
  mpmhip_ctx *ctx_ = nullptr;
  mpmhip_config cfg_{};
  bool dirichlet_ = false;
  std::vector<ParticleType> types_;
  std::vector<std::unique_ptr<ScriptFunction>> scripts_;  // scripted motions of rigid bodies (called back by the library)
};

using MPM3D = MPM<3>;

// `AsyncMPM<3>`: the reference's asynchronous stepper (src/async/async_mpm.{h,cpp}) over the C ABI's "AsyncMPM, second half"
// (include/mpmhip.h): the block-local time stepping itself — pools, backups, gathers, the walk over the power-of-two levels —
// runs inside the library with every particle resident on the device; this class forwards, as MPM<3> does.
class AsyncMPM3D : public MPM3D {
 public:
  void initialize(const Config &config) override {  // AsyncMPM<dim>::initialize, src/async/async_mpm.cpp:13-55
    MPM3D::initialize(config);
    mpmhip_async_config a{};
    a.unit_delta_t = config.get("unit_delta_t", 1e-6f);  // :24-27
    a.max_units = (int64_t)config.get("max_units", 8192.0);
    a.cfl_dt_mul = config.get("cfl_dt_mul", 1.0f);
    a.strength_dt_mul = config.get("strength_dt_mul", 1.0f);
    a.left_boundary = config.get("left_boundary", false) ? 1 : 0;  // :43-53
    check(mpmhip_async_begin(ctx_, &a), ctx_);
  }
  std::string add_particles(const Config &config) override {  // :57-75: the new particles go to their blocks' pools
    const std::string r = MPM3D::add_particles(config);
    check(mpmhip_async_pool_particles(ctx_), ctx_);
    return r;
  }
  std::string add_particles(const Config &config, int64_t n, const float *x, const float *v, float maximum = 0) override {
    const std::string r = MPM3D::add_particles(config, n, x, v, maximum);
    check(mpmhip_async_pool_particles(ctx_), ctx_);
    return r;
  }
  void step(real dt) override { check(mpmhip_async_step(ctx_, dt), ctx_); frame++; }  // :380-421
  real get_current_time() const override { return (real)mpmhip_async_current_time(ctx_); }
  // the views of "the particles" list every container of every particle pool, as AsyncMPM<dim>::visualize does
  // (src/async/async_visualize.cpp:86-96)
  int64_t get_num_particles() const override {
    const int64_t n = mpmhip_async_download_pools(ctx_, 0, nullptr, nullptr);
    check((int)std::min<int64_t>(n, 0), ctx_);
    return n;
  }
  std::vector<RenderParticle> get_render_particles() const override {
    check(mpmhip_async_load_pools(ctx_), ctx_);
    return MPM3D::get_render_particles_of_records();
  }
  void write_partio(const std::string &file_name) const override {
    check(mpmhip_async_load_pools(ctx_), ctx_);
    MPM3D::write_partio(file_name);
  }
  std::string get_name() const override { return "async_mpm"; }  // src/async/async_mpm.h:251-253
  int64_t current_t_int() const { int64_t o[8]; check(mpmhip_async_state(ctx_, o), ctx_); return o[0]; }
  int64_t update_counter() const { int64_t o[8]; check(mpmhip_async_state(ctx_, o), ctx_); return o[1]; }

 protected:
  bool keep_apic_b_default() const override { return true; }  // (the P2G matrix is rebuilt from apic_b for every advance's dt)
};

// the factory the reference reaches through `create_instance<Simulation3D>(name)`: TC_IMPLEMENTATION(Simulation3D, MPM3D, "mpm")
// (src/mpm.cpp:983-988) and TC_IMPLEMENTATION(Simulation3D, AsyncMPM3D, "async_mpm") (src/async/async_mpm.cpp:423-427)
inline std::unique_ptr<MPM3D> create_simulation3(const std::string &name) {
  if (name == "async_mpm") return std::make_unique<AsyncMPM3D>();
  if (name != "mpm") throw std::runtime_error("no Simulation3D implementation named '" + name + "' (registered: 'mpm', 'async_mpm')");
  return std::make_unique<MPM3D>();
}

}  // namespace mpm_amd
