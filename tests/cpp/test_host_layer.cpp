// tests/cpp/test_host_layer.cpp — the C++ host layer (include/mpm_amd/{kernel,particles,mpm}.h) over libmpmhip.
//   ./test_host_layer cpu   kernel known-answer tests of the reference (src/tests.cpp:13-51: sum w = 1, sum dw = 0,
//                           fast == slow kernel to 1e-6; src/transfer.cpp:975-989), Config, the particle registry,
//                           and "no GPU => initialize() throws" (no CPU fallback)
//   ./test_host_layer gpu   MPM<3> end to end on a device: benchmark particles, step(), phases, energy, device
//                           constitutive calls through MPMParticle
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "mpm_amd/mpm.h"
#include "mpm_amd/mpm2d.h"
#include "mpm_amd/mpm88.h"

using namespace mpm_amd;

static int failures = 0;
#define CHECK(cond)                                                             \
  do {                                                                          \
    if (!(cond)) { std::printf("CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } \
  } while (0)

template <int dim, int order>
static void test_kernel(std::mt19937 &rng) {  // src/tests.cpp:13-26
  std::uniform_real_distribution<float> U(0.0f, 10.0f);
  for (int l = 0; l < 100; l++) {
    VectorND<dim, real> pos;
    for (int a = 0; a < dim; a++) pos[a] = U(rng);
    MPMKernel<dim, order> kernel(pos, 1.0f);
    for (int j = 0; j < dim; j++) {
      CHECK(std::fabs(kernel.w_cache[j].sum() - 1.0f) < 1e-5f);
      CHECK(std::fabs(kernel.dw_cache[j].sum()) < 1e-5f);
    }
  }
}

static void cpu_tests() {
  std::mt19937 rng(88);
  test_kernel<2, 3>(rng); test_kernel<2, 2>(rng); test_kernel<3, 3>(rng); test_kernel<3, 2>(rng); test_kernel<3, 1>(rng);
  std::uniform_real_distribution<float> U(0.5f, 10.0f);  // x < 0.5 has no valid quadratic stencil (int() truncates)
  for (int l = 0; l < 100; l++) {  // src/tests.cpp:35-51, src/transfer.cpp:975-989
    Vector3 pos(U(rng), U(rng), U(rng));
    MPMKernel<3, 2> slow(pos, 3.0f);
    MPMFastKernel32 fast(pos, 3.0f);
    Vector3 rel;
    for (int a = 0; a < 3; a++) rel[a] = pos[a] - (float)MPMKernel<3, 2>::get_stencil_start(pos[a]);
    MLSMPMFastKernel32 mls(rel);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        for (int k = 0; k < 3; k++) {
          const Vector4 a = slow.get_dw_w(Vector3i(i, j, k)), b = fast.get_dw_w(Vector3i(i, j, k));
          float d = 0;
          for (int c = 0; c < 4; c++) d += (a[c] - b[c]) * (a[c] - b[c]);
          CHECK(std::sqrt(d) < 1e-6f);
          CHECK(std::fabs(mls.kernels[i][j][k] - slow.get_w(Vector3i(i, j, k))) < 1e-6f);
          CHECK(rel[0] >= 0.5f && rel[0] < 1.5f);
        }
  }
  {  // closed form at fx = 1 (node-centred particle): (1/8, 3/4, 1/8), src/kernel.h:126-130
    MPMKernel<3, 2> k(Vector3(4.0f, 4.0f, 4.0f), 1.0f);
    CHECK(std::fabs(k.w_cache[0][0] - 0.125f) < 1e-7f && std::fabs(k.w_cache[0][1] - 0.75f) < 1e-7f && std::fabs(k.w_cache[0][2] - 0.125f) < 1e-7f);
    CHECK((MPMKernel<3, 2>::get_stencil_start(4.0f) == 3));
    CHECK((MPMKernel<3, 2>::get_stencil_start(4.6f) == 4));
    CHECK((MPMKernel<3, 1>::get_stencil_start(4.6f) == 4));
    CHECK((MPMKernel<3, 3>::get_stencil_start(4.6f) == 3));
    CHECK((MPMKernel<3, 2>::inv_D() == 4.0f) && (MPMKernel<3, 3>::inv_D() == 3.0f));
  }
  {  // Config + particle registry defaults (src/particles.cpp initialize() bodies)
    Config c;
    c.set("res", Vector3i(64, 64, 64)).set("gravity", Vector3(0, -9.8f, 0)).set("E", 2e5).set("flag", true);
    CHECK((c.get_vec("res", Vector3i(0, 0, 0))[2] == 64));
    CHECK(std::fabs(c.get_vec("gravity", Vector3(0, 0, 0))[1] + 9.8f) < 1e-6f);
    CHECK(c.get("missing", 7) == 7 && c.get("flag", false));
    const ParticleType jelly = create_particle_type("jelly", c, 1.0f, 2.0f);
    CHECK(jelly.material == MPMHIP_JELLY && std::fabs(jelly.params[2] - 2e5f / 2.6f) < 1.0f);
    const ParticleType sand = create_particle_type("sand", Config(), 1.0f, 1.0f);
    CHECK(sand.material == MPMHIP_SAND && sand.params[2] == 136038.0f && sand.params[3] == 204057.0f);
    CHECK(std::fabs(sand.params[4] - std::sqrt(2.0 / 3.0) * 2.0 * 0.5 / 2.5) < 1e-6);  // friction angle 30 deg
    const ParticleType snow = create_particle_type("snow", Config(), 1.0f, 1.0f);
    CHECK(snow.initial_aux == 1.0f && snow.params[4] == 10.0f && std::fabs(snow.params[5] - 2.5e-2f) < 1e-9f);
    bool threw = false;
    try { create_particle_type("custard", Config(), 1, 1); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);
    threw = false;
    try { create_particle_type("jelly", Config().set("compressibility", 1.0), 1, 1); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);
  }
  {  // no device: initialize() must fail loudly (the library has no CPU path); "delta_t" is rejected as in :41-42
    MPM<3> sim;
    bool threw = false;
    try { sim.initialize(Config().set("res", Vector3i(32, 32, 32)).set("delta_t", 1e-3)); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);
    if (!std::getenv("MPMHIP_TEST_HAS_GPU")) {
      threw = false;
      std::string msg;
      try { sim.initialize(Config().set("res", Vector3i(32, 32, 32))); } catch (const std::runtime_error &e) { threw = true; msg = e.what(); }
      CHECK(threw);
      CHECK(msg.find("HIP") != std::string::npos);
    }
  }
  {  // the 2D factory: TC_IMPLEMENTATION(Simulation2D, MPM2D, "mpm") / (Simulation2D, AsyncMPM2D, "async_mpm")
    CHECK(create_simulation2("mpm")->get_name() == "mpm" && create_simulation2("async_mpm")->get_name() == "async_mpm");
    bool threw = false;
    try { create_simulation2("custard"); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);
    threw = false;
    try { create_simulation2("mpm")->initialize(Config().set("res", "64,64").set("delta_t", 1e-3)); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);
  }
}

static void gpu_tests_2d() {
  {  // MPM<2>: a jelly square in free fall above a floor line
    auto sim = create_simulation2("mpm");
    sim->initialize(Config().set("res", "64,64").set("base_delta_t", 1e-4).set("gravity", "0,-10"));
    mpmhip_shape floor{};
    floor.type = 0; floor.p[1] = 1.0f; floor.p[3] = -0.2f;  // phi = y - 0.2
    sim->set_levelset(std::vector<mpmhip_shape>{floor}, 0.4f);
    CHECK(sim->add_particles(Config().set("type", "jelly").set("square_lo", 24).set("square_hi", 40)) == "");
    CHECK(sim->get_num_particles() == 16 * 16 * 4);
    sim->step(-1.0f);
    CHECK(std::fabs(sim->get_current_time() - 1e-4f) < 1e-9f);
    sim->step(2e-3f);
    const auto ps = sim->get_particles();
    CHECK((int64_t)ps.size() == 16 * 16 * 4 && ps.front().id == 0 && ps.back().id == 16 * 16 * 4 - 1);
    double vy = 0;
    for (auto &q : ps) vy += q.velocity[1];
    vy /= ps.size();
    CHECK(vy < -0.015 && vy > -0.025);  // ~20 substeps of free fall
    {  // the frame file: "Bgeo" magic, version 5, one row per particle
      const std::string path = "/tmp/mpm_amd_host_layer_2d.bgeo";
      sim->write_partio(path);
      FILE *f = std::fopen(path.c_str(), "rb");
      unsigned char h[13] = {0};
      CHECK(f && std::fread(h, 1, 13, f) == 13);
      if (f) std::fclose(f);
      CHECK(h[0] == 'B' && h[1] == 'g' && h[2] == 'e' && h[3] == 'o' && h[4] == 'V' && h[8] == 5);
      CHECK(((h[9] << 24) | (h[10] << 16) | (h[11] << 8) | h[12]) == 16 * 16 * 4);
      std::remove(path.c_str());
    }
    bool threw = false;
    try { sim->add_articulation(Config().set("type", "motor").set("obj0", 1)); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);
  }
  {  // AsyncMPM<2>: a stiff and a soft square side by side step with their own powers of two of unit_delta_t
    auto as = create_simulation2("async_mpm");
    as->initialize(Config().set("res", "128,128").set("unit_delta_t", 2e-6).set("max_units", 1024).set("gravity", "0,-10"));
    as->add_particles(Config().set("type", "elastic").set("square_lo", 30).set("square_hi", 54));
    as->add_particles(Config().set("type", "sand").set("square_lo", 54).set("square_hi", 78));
    const int64_t n0 = 2 * 24 * 24 * 4;
    CHECK(as->get_num_particles() == n0);
    as->step(2e-3f);
    as->step(2e-3f);
    auto *a = dynamic_cast<AsyncMPM2D *>(as.get());
    CHECK(a && a->current_t_int() >= 2000 && a->update_counter() > n0);
    CHECK(as->get_current_time() >= 4e-3f);
    const auto ps = as->get_particles();
    CHECK((int64_t)ps.size() >= n0);  // an id can sit in more than one pool
    bool finite = true;
    int32_t max_id = -1;
    for (auto &q : ps) { finite = finite && std::isfinite(q.position[1]) && std::isfinite(q.F[0]); max_id = std::max(max_id, q.id); }
    CHECK(finite && max_id == n0 - 1);
    {  // save / load: the pools, the block table and the clocks travel; the restarted stepper is where the saved one was
      const std::string path = "/tmp/mpm_amd_host_layer_async2d.snap";
      as->general_action(Config().set("action", "save").set("file_name", path.c_str()));
      auto bs = create_simulation2("async_mpm");
      bs->initialize(Config().set("res", "128,128").set("unit_delta_t", 2e-6).set("max_units", 1024).set("gravity", "0,-10"));
      bs->general_action(Config().set("action", "load").set("file_name", path.c_str()));
      auto *b2 = dynamic_cast<AsyncMPM2D *>(bs.get());
      CHECK(b2 && b2->current_t_int() == a->current_t_int() && bs->get_num_particles() == as->get_num_particles());
      std::remove(path.c_str());
    }
    bool threw = false;
    try { as->substep(); } catch (const std::exception &) { threw = true; }
    CHECK(threw);
    threw = false;
    const float seg[4] = {-0.05f, 0.0f, 0.05f, 0.0f};
    try { as->add_rigid_body(Config().set("codimensional", true).set("initial_position", "0.5,0.8"), 1, seg); } catch (const std::exception &) { threw = true; }
    CHECK(threw);
  }
}

static void gpu_tests() {
  auto sim = create_simulation3("mpm");
  sim->initialize(Config().set("res", Vector3i(64, 64, 64)).set("base_delta_t", 1e-4).set("gravity", Vector3(0, -10, 0)));
  sim->set_levelset(std::vector<Vector4>{Vector4(0.0f, 1.0f, 0.0f, -0.15f)}, -1.0f);
  CHECK(sim->add_particles(Config().set("type", "jelly").set("benchmark", 125)) == "");
  const int64_t n = sim->get_num_particles();
  CHECK(n == 13 * 13 * 13 * 8);  // round(64*0.4)=26..39: 13^3 cells x 8 (src/mpm.cpp:149-186)
  CHECK(sim->get_vis_resolution()[0] == 64 && sim->get_vis_resolution()[1] == 64);  // scripts/async/async_mpm.py:79-81
  {  // all-jelly scene: mechanical energy is defined (potential_energy exists for jelly, src/particles.cpp:400-407)
    const double e0 = std::stod(sim->general_action(Config().set("action", "calculate_energy")));
    CHECK(e0 >= 0 && e0 < 1e-3 && std::isfinite(e0));  // at rest, undeformed: only the first substep's gravity impulse
  }
  sim->add_particles(Config().set("type", "sand").set("cube_lo", 40).set("cube_hi", 44).set("initial_velocity", Vector3(0, -1, 0)));
  CHECK(sim->get_num_particles() == n + 4 * 4 * 4 * 8);
  sim->step(-1.0f);  // exactly one substep
  CHECK(std::fabs(sim->get_current_time() - 1e-4f) < 1e-9f);
  sim->step(1e-3f);
  CHECK(sim->get_current_time() > 0.9e-3f && sim->get_current_time() <= 1.1e-3f + 1e-4f);
  // the phases, under the reference's names
  sim->sort_particles_and_populate_grid();
  sim->rasterize_optimized();
  sim->normalize_grid_and_apply_external_force();
  sim->apply_grid_boundary_conditions();
  sim->resample_optimized();
  sim->synchronize();
  const auto rp = sim->get_render_particles();
  CHECK((int64_t)rp.size() == sim->get_num_particles());
  double vy = 0;
  for (auto &p : rp) vy += p.velocity[1];
  vy /= rp.size();
  CHECK(vy < -5e-3 && vy > -0.2);  // ~12 substeps of free fall (+ the sand block's -1 m/s)
  {  // sand has no potential_energy() in the reference (TC_NOT_IMPLEMENTED): the action must fail loudly
    bool threw = false;
    try { sim->general_action(Config().set("action", "calculate_energy")); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);
  }
  {  // save / load (src/mpm.cpp:940-960): a second simulation continues from the blob
    const std::string fn = "/tmp/mpm_amd_host_layer_snapshot.bin";
    sim->general_action(Config().set("action", "save").set("file_name", fn));
    auto sim2 = create_simulation3("mpm");
    sim2->initialize(Config().set("res", Vector3i(64, 64, 64)).set("base_delta_t", 1e-4).set("gravity", Vector3(0, -10, 0)));
    sim2->set_levelset(std::vector<Vector4>{Vector4(0.0f, 1.0f, 0.0f, -0.15f)}, -1.0f);
    sim2->general_action(Config().set("action", "load").set("file_name", fn));
    CHECK(sim2->get_num_particles() == sim->get_num_particles());
    CHECK(std::fabs(sim2->get_current_time() - sim->get_current_time()) < 1e-9f);
    sim->step(-1.0f); sim2->step(-1.0f);
    const auto a = sim->get_render_particles(), b = sim2->get_render_particles();
    double dmax = 0;
    for (size_t i = 0; i < a.size(); i++)
      for (int k = 0; k < 3; k++) dmax = std::max(dmax, (double)std::fabs(a[i].position[k] - b[i].position[k]));
    CHECK(a.size() == b.size() && dmax < 1e-6);
    std::remove(fn.c_str());
  }
  {  // visualize() -> frame_directory/0001.bgeo, 0002.bgeo (src/mpm.h:333-337); header and size as Partio writes them
    auto sim3 = create_simulation3("mpm");
    sim3->initialize(Config().set("res", Vector3i(64, 64, 64)).set("frame_directory", "/tmp").set("verbose_bgeo", true));
    sim3->add_particles(Config().set("type", "water").set("cube_lo", 30).set("cube_hi", 33));
    sim3->step(-1.0f);
    sim3->visualize();
    const std::string fn = sim3->write_bgeo();
    CHECK(fn == "/tmp/0002.bgeo" && sim3->frame_count == 2);
    FILE *f = std::fopen(fn.c_str(), "rb");
    CHECK(f != nullptr);
    if (f) {
      unsigned char h[17] = {0};
      CHECK(std::fread(h, 1, 17, f) == 17);
      CHECK(h[0] == 'B' && h[1] == 'g' && h[2] == 'e' && h[3] == 'o' && h[4] == 'V' && h[8] == 5);
      const uint32_t n_points = (uint32_t)h[9] << 24 | (uint32_t)h[10] << 16 | (uint32_t)h[11] << 8 | h[12];
      CHECK(n_points == 3 * 3 * 3 * 8);
      std::fseek(f, 0, SEEK_END);
      size_t want = 0;
      CHECK(mpmhip_bgeo_size(sim3->ctx(), 1, &want) == 0 && (size_t)std::ftell(f) == want);
      std::fclose(f);
    }
    std::remove("/tmp/0001.bgeo");
    std::remove("/tmp/0002.bgeo");
  }
  {  // the 2D demo (mls-mpm88.cpp): three squares, first step from rest = free fall with v.y = -200 dt
    MLSMPM88 demo;
    demo.add_object(0.55f, 0.45f, 0xED553B);
    demo.add_object(0.45f, 0.65f, 0xF2B134);
    demo.add_object(0.55f, 0.85f, 0x068587);
    CHECK(demo.num_particles() == 3000);
    demo.advance(1);
    const auto ps = demo.particles();
    double vy = 0, vx = 0;
    for (auto &q : ps) { vy += q.v[1]; vx += std::fabs(q.v[0]); }
    CHECK(std::fabs(vy / ps.size() + 200.0 * 1e-4) < 1e-6 && vx / ps.size() < 1e-6);
    CHECK(ps[0].c == 0xED553B && ps[2999].c == 0x068587 && std::fabs(ps[5].F[0] - 1.0f) < 1e-5f);
    demo.advance(200);
    CHECK(demo.particles()[1500].x[1] < 0.65f + 0.08f);
  }
  {  // CPIC: add_particles(type='rigid') — a scripted plate pushed down through a block of jelly
    auto sim4 = create_simulation3("mpm");
    sim4->initialize(Config().set("res", Vector3i(32, 32, 32)).set("base_delta_t", 1e-4).set("gravity", Vector3(0, -10, 0))
                         .set("max_particles", 100000.0));
    const float h = 0.2f;
    const float tri[18] = {-h, 0, -h, h, 0, -h, h, 0, h, -h, 0, -h, h, 0, h, -h, 0, h};
    const std::string id = sim4->add_rigid_body(Config().set("codimensional", true).set("friction", 0.3), 2, tri,
                                                [](real t) { return Vector3(0.5f, 0.55f - 1.0f * t, 0.5f); });
    CHECK(id == "1" && sim4->has_rigid_body());
    sim4->add_particles(Config().set("type", "jelly").set("cube_lo", 10).set("cube_hi", 22));
    for (int i = 0; i < 20; i++) sim4->substep();
    sim4->synchronize();
    const auto st = sim4->get_rigid_state(1);
    CHECK(std::fabs(st[1] - (0.55f - 20e-4f)) < 1e-6f);  // on its script
    CHECK(std::fabs(st[8] + 1.0f) < 1e-3f);              // moving with the script's secant velocity
    CHECK(st[14] == 0.0f);                               // a scripted translation answers impulses with infinite mass
    double below = 0, above = 0;  // the plate separates the block: particles under it are pushed down faster than those above
    for (auto &q : sim4->get_render_particles()) (q.position[1] < st[1] ? below : above) += 1;
    CHECK(below > 1000 && above > 1000);
  }
  {  // joints: general_action(action='add_articulation') — a stepper to the background spins a free plate about z
    auto sim5 = create_simulation3("mpm");
    sim5->initialize(Config().set("res", Vector3i(32, 32, 32)).set("base_delta_t", 1e-4).set("gravity", Vector3(0, 0, 0))
                         .set("max_particles", 4096.0));
    const float h = 0.1f;
    const float tri[18] = {-h, 0, -h, h, 0, -h, h, 0, h, -h, 0, -h, h, 0, h, -h, 0, h};
    CHECK(sim5->add_rigid_body(Config().set("codimensional", true).set("friction", 0.3).set("initial_position", Vector3(0.5f, 0.5f, 0.5f)), 2, tri) == "1");
    CHECK(sim5->general_action(Config().set("action", "add_articulation").set("type", "stepper").set("obj0", 1)
                                   .set("axis", Vector3(0, 0, 1)).set("angular_velocity", 3.0)) == "");
    bool threw = false;
    try { sim5->general_action(Config().set("action", "add_articulation").set("type", "spring").set("obj0", 1)); } catch (const std::exception &) { threw = true; }
    CHECK(threw);
    sim5->add_particles(Config().set("type", "jelly").set("cube_lo", 4).set("cube_hi", 6));  // (far from the plate)
    for (int i = 0; i < 5; i++) sim5->substep();
    sim5->synchronize();
    const auto st = sim5->get_rigid_state(1);
    CHECK(std::fabs(st[12] - 3.0f) < 1e-4f && std::fabs(st[10]) < 1e-5f && std::fabs(st[11]) < 1e-5f);  // angular velocity = (0, 0, 3)
    CHECK(std::fabs(st[0] - 0.5f) < 1e-5f && std::fabs(st[1] - 0.5f) < 1e-5f);                          // the hinge holds the centre
  }
  {  // create_simulation3("async_mpm"): AsyncMPM<3> (src/async/async_mpm.cpp) — block-local time steps, pools resident on the device
    auto as = create_simulation3("async_mpm");
    CHECK(as->get_name() == "async_mpm");
    as->initialize(Config().set("res", Vector3i(32, 32, 32)).set("unit_delta_t", 2e-6).set("max_units", 1024.0).set("max_particles", 16384.0));
    as->set_levelset(std::vector<Vector4>{Vector4(0, 1, 0, -0.2f)}, 0.4f);
    as->add_particles(Config().set("type", "elastic").set("cube_lo", 9).set("cube_hi", 15).set("initial_velocity", Vector3(0.2f, 0, 0)));
    as->add_particles(Config().set("type", "sand").set("cube_lo", 15).set("cube_hi", 21));
    const int64_t n0 = as->get_num_particles();
    CHECK(n0 == 2 * 6 * 6 * 6 * 8);
    const int64_t bytes0 = mpmhip_host_particle_bytes(as->ctx());
    as->step(2.5e-3f);
    as->step(2.5e-3f);
    CHECK(mpmhip_host_particle_bytes(as->ctx()) == bytes0);  // stepping moved no particle data across the host boundary
    auto *a3 = dynamic_cast<AsyncMPM3D *>(as.get());
    CHECK(a3 != nullptr && a3->current_t_int() >= 2500 && a3->update_counter() > n0);
    CHECK(std::fabs(as->get_current_time() - 2e-6f * (float)a3->current_t_int()) < 1e-9f);
    const auto rp = as->get_render_particles();  // every pool container (ids can repeat, as in the reference)
    CHECK((int64_t)rp.size() == as->get_num_particles() && (int64_t)rp.size() >= n0);
    bool finite = true;
    int32_t max_id = -1;
    for (auto &q : rp) { finite = finite && std::isfinite(q.position[1]) && std::isfinite(q.velocity[1]); max_id = std::max(max_id, q.id); }
    CHECK(finite && max_id == n0 - 1);
    bool threw = false;
    try { as->step(-1.0f); } catch (const std::exception &) { threw = true; }
    CHECK(threw);  // the synchronous single substep belongs to "mpm"
  }
  // device constitutive code through the particle surface: F = I => zero force; plasticity(cdg) = F <- cdg F for jelly
  MPMParticle p;
  p.type = create_particle_type("jelly", Config(), 1.0f, 1e-6f);
  const Matrix3 f0 = p.calculate_force(sim->ctx());
  for (float v : f0) CHECK(std::fabs(v) < 1e-9f);
  Matrix3 cdg{{1.01f, 0, 0, 0, 1, 0, 0, 0, 0.99f}};
  p.plasticity(sim->ctx(), cdg);
  CHECK(std::fabs(p.dg_e[0] - 1.01f) < 1e-6f && std::fabs(p.dg_e[8] - 0.99f) < 1e-6f);
  const Matrix3 f1 = p.calculate_force(sim->ctx());
  CHECK(f1[0] < 0 && f1[8] > 0);  // -vol P F^T: stretched axis pulls back, compressed axis pushes
  CHECK(p.get_allowed_dt(1.0f / 64) == 0.0f);  // JellyParticle::get_allowed_dt returns 0 (src/particles.cpp:418-420)
  // per-material get_allowed_dt: the host expression equals what the device evaluates (mpmhip_debug_allowed_dt)
  for (const char *name : {"sand", "snow", "water", "elastic", "von_mises", "visco"}) {
    MPMParticle q;
    q.type = create_particle_type(name, Config(), 400.0f * 1e-6f, 1e-6f);
    q.aux = q.type.initial_aux;
    q.dg_e = Matrix3{{1.02f, 0.01f, 0, -0.01f, 0.98f, 0.02f, 0, 0.01f, 1.01f}};
    q.v = Vector3(0.3f, -0.2f, 0.1f);
    const float host = q.get_allowed_dt(1.0f / 64);
    float dev = 0;
    const int rc = mpmhip_debug_allowed_dt(sim->ctx(), q.type.material, q.type.params, 1, q.dg_e.data(), &q.aux, q.v.data(), 1.0f / 64, &dev);
    CHECK(rc == 0);
    CHECK(host > 0 && std::fabs(host - dev) <= 2e-5f * host);
  }
}

int main(int argc, char **argv) {
  const std::string mode = argc > 1 ? argv[1] : "cpu";
  try {
    if (mode == "cpu") cpu_tests();
    else { gpu_tests(); gpu_tests_2d(); }
  } catch (const std::exception &e) {
    std::printf("unexpected exception: %s\n", e.what());
    return 2;
  }
  std::printf("%s: %d failure(s)\n", mode.c_str(), failures);
  return failures ? 1 : 0;
}
