// taichi_mpm_amd/csrc/rigid_api.h — host side of the CPIC rigid coupling (included by mpmhip.hip inside extern "C")
// Part of libmpmhip.  Device side: k_rigid.h (bodies, colored distance field, particle colours), k_rigid_transfer.h.
//
// A rigid body = triangles (mesh space) + pose + mass properties.  Creation follows MPM::add_rigid_particle
// (src/mpm_rigid_body.cpp:130-252): scale the mesh, mass / centre of mass / inertia, recentre, sample boundary particles
// on every triangle with the reference's own loop, drop samples near the domain wall.  The rigid-body arithmetic itself
// (what an impulse does, how a script becomes a velocity) belongs to the absent taichi core; the conventions used here
// are those of the body the reference build is tested against (oracle/taichi_shim/taichi/dynamics/rigid_body_shim.h)
// and are restated in k_rigid.h.  Joints between bodies (MPM::articulate) are in k_joints.h; rigid-rigid collisions
// (MPM::rigidify, src/mpm.cpp:468, libccd) are NOT part of this library — the Python layer warns when a scene has more than
// one body that could collide.

static void quat_from_euler_deg(const float deg[3], float q[4]) {  // X * Y * Z, src/mpm_rigid_body.cpp:108-114
  const float r = (float)(M_PI / 180.0);
  float ax[3][4];
  for (int k = 0; k < 3; k++) {
    const float a = deg[k] * r;
    ax[k][0] = std::cos(a / 2); ax[k][1] = ax[k][2] = ax[k][3] = 0.0f;
    ax[k][1 + k] = std::sin(a / 2);
  }
  auto mul = [](const float a[4], const float b[4], float o[4]) {
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  };
  float xy[4];
  mul(ax[0], ax[1], xy);
  mul(xy, ax[2], q);
}
static void quat_to_matrix(const float q[4], float R[9]) {
  const float w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}

// mass, centre of mass, inertia about it (body frame) of a triangle mesh: a shell of surface density `density`
// (codimensional) or the enclosed solid of volume density `density` (signed tetrahedra against the origin)
static bool mesh_mass_properties(const std::vector<float> &tri, bool codimensional, double density, double &M, double com[3],
                                 double I[9]) {
  double S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  M = 0; com[0] = com[1] = com[2] = 0;
  auto add = [&](double w, const double a[3], const double b[3]) {
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) S[r][c] += w * 0.5 * (a[r] * b[c] + b[r] * a[c]);
  };
  for (size_t e = 0; e < tri.size() / 9; e++) {
    double v[3][3];
    for (int q = 0; q < 3; q++) for (int k = 0; k < 3; k++) v[q][k] = tri[9 * e + 3 * q + k];
    const double s[3] = {v[0][0] + v[1][0] + v[2][0], v[0][1] + v[1][1] + v[2][1], v[0][2] + v[1][2] + v[2][2]};
    double m, cw, sw;
    if (codimensional) {
      const double a[3] = {v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2]}, b[3] = {v[2][0] - v[0][0], v[2][1] - v[0][1], v[2][2] - v[0][2]};
      const double n[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
      m = 0.5 * std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) * density; cw = 1.0 / 3.0; sw = 1.0 / 12.0;
    } else {
      const double det = v[0][0] * (v[1][1] * v[2][2] - v[1][2] * v[2][1]) - v[0][1] * (v[1][0] * v[2][2] - v[1][2] * v[2][0]) +
                         v[0][2] * (v[1][0] * v[2][1] - v[1][1] * v[2][0]);
      m = det / 6.0 * density; cw = 0.25; sw = 1.0 / 20.0;
    }
    M += m;
    for (int k = 0; k < 3; k++) com[k] += m * cw * s[k];
    for (int q = 0; q < 3; q++) add(m * sw, v[q], v[q]);
    add(m * sw, s, s);
  }
  if (!(std::abs(M) > 0)) return false;
  for (int k = 0; k < 3; k++) com[k] /= M;
  const double flip = M < 0 ? -1.0 : 1.0;
  M *= flip;
  double Sc[3][3];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Sc[r][c] = flip * S[r][c] - M * com[r] * com[c];
  const double tr = Sc[0][0] + Sc[1][1] + Sc[2][2];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) I[3 * r + c] = (r == c ? tr : 0.0) - Sc[r][c];
  return true;
}
static bool invert3(const double a[9], double o[9]) {
  const double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
  if (det == 0) return false;
  const double id = 1.0 / det;
  o[0] = (a[4] * a[8] - a[5] * a[7]) * id; o[1] = (a[2] * a[7] - a[1] * a[8]) * id; o[2] = (a[1] * a[5] - a[2] * a[4]) * id;
  o[3] = (a[5] * a[6] - a[3] * a[8]) * id; o[4] = (a[0] * a[8] - a[2] * a[6]) * id; o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
  o[6] = (a[3] * a[7] - a[4] * a[6]) * id; o[7] = (a[1] * a[6] - a[0] * a[7]) * id; o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
  return true;
}

static int rigid_enable(mpmhip_ctx *c) {
  auto &R = c->rigid;
  if (R.enabled) return MPMHIP_OK;
  if (c->T.enabled) return fail(c, MPMHIP_EINVAL, "rigid bodies are not supported on a tiled (multi-GPU) ctx");
  if (c->async.enabled) return fail(c, MPMHIP_EINVAL, "rigid bodies are not supported with asynchronous stepping");
  hipError_t e = hipSuccess;
  auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  R.max_pages = (uint32_t)std::min<int64_t>((int64_t)c->NB, std::max<int64_t>(8192, (int64_t)c->P.max_blocks * 2));
  R.max_pages = (R.max_pages + CDF_POOLS - 1) / CDF_POOLS * CDF_POOLS;
  const int rpd[3] = {c->P.res[0] / 4 + 2, c->P.res[1] / 4 + 2, c->P.res[2] / 8 + 2};
  R.rpage_words = ((size_t)rpd[0] * rpd[1] * rpd[2] + 31) / 32;
  A(dmalloc(&R.d_rb, (size_t)MAX_RIGID));
  A(dmalloc(&R.d_joints, (size_t)MAX_JOINTS));
  A(dmalloc(&R.cdf.slot, (size_t)c->NB));
  A(dmalloc(&R.cdf.page_key, (size_t)R.max_pages));
  A(dmalloc(&R.cdf.mind, (size_t)R.max_pages * 64));
  A(dmalloc(&R.cdf.tags, (size_t)R.max_pages * 64));
  A(dmalloc(&R.d_counters, (size_t)CDF_POOLS + 4));  // [0, CDF_POOLS) pages handed out per sub-pool, [CDF_POOLS] cutting_counter,
                                                    // [CDF_POOLS + 1] length of d_rigid_list
  A(dmalloc(&R.cdf.rpage, R.rpage_words));
  A(dmalloc(&R.d_bnd, (size_t)c->cap));
  A(dmalloc(&R.d_blk_rigid, (size_t)c->P.max_blocks + 1));
  A(dmalloc(&R.d_rigid_list, (size_t)c->P.max_blocks + 1));
  if (e != hipSuccess) return fail(c, MPMHIP_ENOMEM, "rigid coupling: device allocation failed: %s", hipGetErrorString(e));
  R.cdf.n_pages = R.d_counters; R.cdf.error = &c->cnt->error;
  R.cdf.nb_axis = 1 << c->P.kbits;
  R.cdf.max_pages = R.max_pages;
  R.cdf.pool_cap = R.max_pages / CDF_POOLS;
  for (int k = 0; k < 3; k++) R.cdf.rpd[k] = rpd[k];
  HIPCHK(c, hipMemset(R.d_rb, 0, sizeof(RigidBodyDev) * MAX_RIGID));
  {  // body 0, the background (RigidBody::set_as_background): at the origin, unrotated, immovable — joints may link to it
    RigidBodyDev G;
    memset(&G, 0, sizeof G);
    G.R[0] = G.R[4] = G.R[8] = 1.0f;
    G.q[0] = 1.0f;
    G.mass = 1.0f;
    HIPCHK(c, hipMemcpy(R.d_rb, &G, sizeof G, hipMemcpyHostToDevice));
  }
  HIPCHK(c, hipMemset(R.cdf.slot, 0xFF, sizeof(uint32_t) * (size_t)c->NB));
  HIPCHK(c, hipMemset(R.cdf.mind, 0xFF, sizeof(unsigned long long) * (size_t)R.max_pages * 64));
  HIPCHK(c, hipMemset(R.cdf.tags, 0, sizeof(uint32_t) * (size_t)R.max_pages * 64));
  HIPCHK(c, hipMemset(R.cdf.page_key, 0, sizeof(uint32_t) * (size_t)R.max_pages));
  HIPCHK(c, hipMemset(R.d_counters, 0, sizeof(uint32_t) * (CDF_POOLS + 4)));
  HIPCHK(c, hipMemset(R.cdf.rpage, 0, sizeof(uint32_t) * R.rpage_words));
  HIPCHK(c, hipMemset(R.d_bnd, 0, sizeof(BndRec) * (size_t)c->cap));
  HIPCHK(c, hipMemset(R.d_blk_rigid, 0, (size_t)c->P.max_blocks + 1));
  R.bodies.clear();
  R.bodies.emplace_back();  // body 0: the background (MPM::initialize, src/mpm.cpp:72-74); it has no surface
  R.bodies[0].mass = 1.0f;
  R.bodies[0].inertia[0] = R.bodies[0].inertia[4] = R.bodies[0].inertia[8] = 1.0f;  // set_as_background: inertia 1, inverse 0
  R.joints.clear();
  R.enabled = true;
  return MPMHIP_OK;
}

static RigidXfer rigid_xfer(mpmhip_ctx *c) {
  RigidXfer X;
  X.C = c->rigid.cdf; X.rb = c->rigid.d_rb; X.bnd = c->rigid.d_bnd; X.blk_rigid = c->rigid.d_blk_rigid;
  X.rigid_list = c->rigid.d_rigid_list; X.n_rigid = c->rigid.d_counters + CDF_POOLS + 1;
  X.rp_in = (const float4 *)c->rp;
  X.penalty = c->rigid.penalty; X.pushing_force = c->rigid.pushing_force;
  return X;
}
static inline bool rigid_active(const mpmhip_ctx *c) { return c->rigid.enabled && c->rigid.bodies.size() > 1; }  // has_rigid_body()

// rasterize_rigid_boundary: clear last substep's pages, then the boundary particles write colours and distances
static int do_rigid_rasterize(mpmhip_ctx *c) {
  auto &R = c->rigid;
  hipLaunchKernelGGL(k_cdf_clear, dim3(16, CDF_POOLS), dim3(256), 0, c->stream, R.cdf);
  HIPCHK(c, hipMemsetAsync(R.cdf.n_pages, 0, sizeof(uint32_t) * CDF_POOLS, c->stream));
  HIPCHK(c, hipMemsetAsync(R.cdf.rpage, 0, sizeof(uint32_t) * R.rpage_words, c->stream));
  if (R.n_smp) {
    hipLaunchKernelGGL(k_cdf_alloc, dim3(particle_grid(R.n_smp)), dim3(256), 0, c->stream, c->P, R.cdf, (const RigidBodyDev *)R.d_rb,
                       (const RigidSample *)R.d_smp, R.n_smp);
    hipLaunchKernelGGL(k_cdf_rasterize, dim3(particle_grid((int64_t)R.n_smp * 27)), dim3(256), 0, c->stream, c->P, R.cdf,
                       (const RigidBodyDev *)R.d_rb, (const RigidSample *)R.d_smp, (const float *)R.d_elems, R.n_smp);
  }
  return launch_check(c, "rasterize_rigid_boundary");
}
static int do_rigid_gather(mpmhip_ctx *c) {
  auto &R = c->rigid;
  hipLaunchKernelGGL(k_gather_cdf, dim3(8192), dim3(256), 0, c->stream, c->P, R.cdf, c->rg, R.d_bnd, R.d_counters + CDF_POOLS,
                     (const Counters *)c->cnt, (const uint32_t *)c->act_blk, (const uint32_t *)c->act_start, (const uint32_t *)c->perm, ++R.gather_epoch);
  return launch_check(c, "gather_cdf");
}
static int do_rigid_block_flags(mpmhip_ctx *c) {
  auto &R = c->rigid;
  HIPCHK(c, hipMemsetAsync(R.d_counters + CDF_POOLS + 1, 0, sizeof(uint32_t), c->stream));
  hipLaunchKernelGGL(k_blk_rigid, dim3(256), dim3(256), 0, c->stream, c->P, (const Counters *)c->cnt, (const uint32_t *)c->act_blk, R.cdf,
                     R.d_blk_rigid, c->act_start, R.d_rigid_list, R.d_counters + CDF_POOLS + 1);
  return launch_check(c, "rigid block flags");
}
static int do_rigid_apply_tmp(mpmhip_ctx *c) {
  hipLaunchKernelGGL(k_rigid_apply_tmp, dim3(1), dim3(64), 0, c->stream, c->rigid.d_rb, (int)c->rigid.bodies.size());
  return launch_check(c, "rigid apply_tmp_velocity");
}
// MPM<dim>::rigid_body_levelset_collision (src/mpm_rigid_body.cpp:347-387): the boundary particles in the reference's order
// (k_rigid.h), then the sequential impulse chain
static int do_rigid_ls_collision(mpmhip_ctx *c) {
  auto &R = c->rigid;
  const uint32_t n = R.n_smp;
  if (n == 0 || c->LS.n == 0) return MPMHIP_OK;
  if (R.ls_cap < n) {
    for (int k = 0; k < 2; k++) { (void)hipFree(R.d_ls_keys[k]); (void)hipFree(R.d_ls_vals[k]); R.d_ls_keys[k] = nullptr; R.d_ls_vals[k] = nullptr; }
    (void)hipFree(R.d_ls_tmp); R.d_ls_tmp = nullptr; R.ls_tmp_bytes = 0;
    const size_t cap = (size_t)n + n / 4 + 1024;
    for (int k = 0; k < 2; k++) { HIPCHK(c, dmalloc(&R.d_ls_keys[k], cap)); HIPCHK(c, dmalloc(&R.d_ls_vals[k], cap)); }
    size_t bytes = 0;
    HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, R.d_ls_keys[0], R.d_ls_keys[1], R.d_ls_vals[0], R.d_ls_vals[1], (int)cap, 0, 64, c->stream));
    HIPCHK(c, hipMalloc(&R.d_ls_tmp, bytes));
    R.ls_tmp_bytes = bytes;
    R.ls_cap = (uint32_t)cap;
  }
  if (R.n_ranked < n) {  // boundary particles added since: behind everybody else, in creation order (appended to `particles`)
    std::vector<uint32_t> rk(n);
    // (the ctx stream does not synchronise with the blocking copies below: a collision kernel of an earlier substep may still
    // be writing the ranks)
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (R.n_ranked) HIPCHK(c, hipMemcpy(rk.data(), R.d_smp_rank, sizeof(uint32_t) * R.n_ranked, hipMemcpyDeviceToHost));
    for (uint32_t s = R.n_ranked; s < n; s++) rk[s] = s;
    (void)hipFree(R.d_smp_rank); R.d_smp_rank = nullptr;
    HIPCHK(c, dmalloc(&R.d_smp_rank, (size_t)n + n / 4 + 1024));
    HIPCHK(c, hipMemcpy(R.d_smp_rank, rk.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    R.n_ranked = n;
  }
  c->P.t = c->t;
  hipLaunchKernelGGL(k_rigid_ls_keys, dim3(particle_grid(n)), dim3(256), 0, c->stream, c->P, (const RigidBodyDev *)R.d_rb,
                     (const RigidSample *)R.d_smp, n, (const uint32_t *)R.d_smp_rank, R.d_ls_keys[0], R.d_ls_vals[0]);
  size_t bytes = R.ls_tmp_bytes;
  HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(R.d_ls_tmp, bytes, R.d_ls_keys[0], R.d_ls_keys[1], R.d_ls_vals[0], R.d_ls_vals[1], (int)n, 0, 64, c->stream));
  RigidRestitution rest;
  memset(&rest, 0, sizeof rest);
  for (size_t b = 1; b < R.bodies.size(); b++) rest.e[b] = R.bodies[b].cfg.restitution;
  hipLaunchKernelGGL(k_rigid_ls_collide, dim3(1), dim3(1024), 0, c->stream, c->P, (const LevelSetDev *)c->d_LS, R.d_rb, (int)R.bodies.size(),
                     (const RigidSample *)R.d_smp, (const uint32_t *)R.d_ls_vals[1], n, R.d_smp_rank, rest);
  return launch_check(c, "rigid_body_levelset_collision");
}
// advect_rigid_bodies(dt) at current_t = c->t: scripts are evaluated here, on the host, for this one substep
static int do_rigid_advect(mpmhip_ctx *c, float dt) {
  auto &R = c->rigid;
  RigidSteps st;
  memset(&st, 0, sizeof st);
  for (size_t b = 1; b < R.bodies.size(); b++) {
    const auto &B = R.bodies[b];
    RigidStep &s = st.s[b];
    if (B.cfg.scripted_position) {
      s.has_pos = 1;
      B.cfg.scripted_position(B.cfg.position_user, c->t, s.p0);
      B.cfg.scripted_position(B.cfg.position_user, c->t + dt, s.p1);
    }
    if (B.cfg.scripted_rotation) {
      s.has_rot = 1;
      float e0[3], e1[3];
      B.cfg.scripted_rotation(B.cfg.rotation_user, c->t, e0);
      B.cfg.scripted_rotation(B.cfg.rotation_user, c->t + dt, e1);
      quat_from_euler_deg(e0, s.q0);
      quat_from_euler_deg(e1, s.q1);
    }
  }
  hipLaunchKernelGGL(k_rigid_advect, dim3(1), dim3(64), 0, c->stream, R.d_rb, (int)R.bodies.size(), st, dt, c->P.g[0], c->P.g[1], c->P.g[2]);
  return launch_check(c, "advect_rigid_bodies");
}
// what substep() does with rigid bodies between the sort and P2G (src/mpm.cpp:466-472, 506-508)
// MPM::articulate (src/mpm.h:278-319): between the sort and rasterize_rigid_boundary (src/mpm.cpp:466-471)
static int do_rigid_articulate(mpmhip_ctx *c, float dt) {
  auto &R = c->rigid;
  if (R.joints.empty()) return MPMHIP_OK;
  hipLaunchKernelGGL(k_articulate, dim3(1), dim3(64), 0, c->stream, R.d_rb, (int)R.bodies.size(), (const JointDev *)R.d_joints,
                     (int)R.joints.size(), dt, R.joint_iterations);
  return launch_check(c, "articulate");
}
// In two halves: (a) the joints and the colored distance field depend on the bodies alone — substep_begin runs them on the
// side stream NEXT TO the particle sort; (b) the particles' colours and the block flags need both the field and the sort.
static int do_rigid_pre_a(mpmhip_ctx *c, hipStream_t on) {
  hipStream_t keep = c->stream;
  c->stream = on;  // (the launch helpers below enqueue on the ctx's stream)
  int rc;
  if (!(rc = do_rigid_articulate(c, c->P.dt))) rc = do_rigid_rasterize(c);
  c->stream = keep;
  return rc;
}
static int do_rigid_pre_b(mpmhip_ctx *c) {
  int rc;
  if ((rc = do_rigid_gather(c)) || (rc = do_rigid_block_flags(c))) return rc;
  return MPMHIP_OK;
}

int mpmhip_set_rigid_levelset_collision(mpmhip_ctx *c, int32_t enabled) {
  if (!c) return MPMHIP_EINVAL;
  c->rigid.ls_collision = enabled != 0;
  return MPMHIP_OK;
}

int mpmhip_set_rigid_coupling(mpmhip_ctx *c, float penalty, float pushing_force) {
  if (!c) return MPMHIP_EINVAL;
  c->rigid.penalty = penalty;
  c->rigid.pushing_force = pushing_force;
  return MPMHIP_OK;
}

int mpmhip_add_rigid_body(mpmhip_ctx *c, const mpmhip_rigid_config *cfg, int64_t n_triangles, const float *triangles) {
  if (!c || !cfg || !triangles || n_triangles <= 0) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->in_substep) return fail(c, MPMHIP_EINVAL, "add_rigid_body inside a substep");
  // check_scripting_parameters (src/mpm_rigid_body.cpp:15-56) as far as the struct can express it
  if (int rc = rigid_enable(c)) return rc;
  auto &R = c->rigid;
  if ((int)R.bodies.size() >= MAX_RIGID) return fail(c, MPMHIP_ECAPACITY, "at most %d rigid bodies (2 colour bits each in a 24-bit word)", MAX_RIGID - 1);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  mpmhip_ctx::HostRigid B;
  B.cfg = *cfg;
  const bool spos = cfg->scripted_position != nullptr, srot = cfg->scripted_rotation != nullptr;
  if (!cfg->recenter && !(spos && srot)) return fail(c, MPMHIP_EINVAL, "recenter = 0 needs a scripted position and rotation (src/mpm_rigid_body.cpp:186-190)");
  std::vector<float> tri(triangles, triangles + 9 * n_triangles);
  const float sc[3] = {cfg->scale[0] != 0 ? cfg->scale[0] : 1.0f, cfg->scale[1] != 0 ? cfg->scale[1] : 1.0f, cfg->scale[2] != 0 ? cfg->scale[2] : 1.0f};
  for (int64_t e = 0; e < n_triangles; e++) {
    if (cfg->reverse_vertices) for (int k = 0; k < 3; k++) std::swap(tri[9 * e + k], tri[9 * e + 3 + k]);
    for (int q = 0; q < 3; q++) for (int k = 0; k < 3; k++) tri[9 * e + 3 * q + k] *= sc[k];
  }
  const double density = cfg->density > 0 ? cfg->density : (cfg->codimensional ? 40.0 : 400.0);
  double M, com[3], I[9], invI[9];
  if (!mesh_mass_properties(tri, cfg->codimensional != 0, density, M, com, I) || !invert3(I, invI))
    return fail(c, MPMHIP_EINVAL, "rigid body: the mesh has no mass or a singular inertia tensor");
  if (!cfg->recenter) com[0] = com[1] = com[2] = 0.0;
  const float comf[3] = {(float)com[0], (float)com[1], (float)com[2]};
  for (int64_t e = 0; e < n_triangles; e++)
    for (int q = 0; q < 3; q++) for (int k = 0; k < 3; k++) tri[9 * e + 3 * q + k] -= comf[k];
  // the device record
  RigidBodyDev D;
  memset(&D, 0, sizeof D);
  float p0[3] = {cfg->initial_position[0], cfg->initial_position[1], cfg->initial_position[2]};
  if (spos) cfg->scripted_position(cfg->position_user, c->t, p0);
  float euler[3] = {cfg->initial_rotation[0], cfg->initial_rotation[1], cfg->initial_rotation[2]};
  if (srot) cfg->scripted_rotation(cfg->rotation_user, c->t, euler);
  quat_from_euler_deg(euler, D.q);
  quat_to_matrix(D.q, D.R);
  for (int k = 0; k < 3; k++) {
    D.pos[k] = p0[k];
    D.vel[k] = spos ? 0.0f : cfg->initial_velocity[k];
    D.omega[k] = srot ? 0.0f : cfg->initial_angular_velocity[k];
    D.axis[k] = cfg->rotation_axis[k];
  }
  D.mass = (float)M;
  D.inv_mass = spos ? 0.0f : (float)(1.0 / M);                    // set_infinity_mass
  for (int k = 0; k < 9; k++) D.inv_I[k] = srot ? 0.0f : (float)invI[k];  // set_infinity_inertia
  D.fric[0] = cfg->friction[0]; D.fric[1] = cfg->friction[1];
  D.lin_damp = cfg->linear_damping; D.ang_damp = cfg->angular_damping;
  D.scripted = (spos ? 1 : 0) | (srot ? 2 : 0);
  for (int k = 0; k < 9; k++) { B.inertia[k] = (float)I[k]; B.inv_inertia[k] = (float)invI[k]; }
  B.mass = (float)M;
  // boundary particles: the reference's sampling loop (src/mpm_rigid_body.cpp:214-235), in float like the reference
  const int body = (int)R.bodies.size();
  const float dx = c->P.dx;
  const size_t elem0 = R.h_elems.size() / 9;
  B.first_sample = (int)R.h_smp.size();
  int32_t allocated = 0;
  for (int64_t e = 0; e < n_triangles; e++) {
    const float *v0 = &tri[9 * e], *v1 = v0 + 3, *v2 = v0 + 6;
    float a[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]}, b[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
    const float xl = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), yl = std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    if (!(xl > 0.0f) || !(yl > 0.0f)) continue;  // a degenerate triangle carries no boundary particles
    for (int k = 0; k < 3; k++) { a[k] /= xl; b[k] /= yl; }
    for (float _x = std::min(xl / 3.0f, dx / 2.0f); _x < xl + dx; _x += dx)
      for (float _y = std::min(yl / 3.0f, dx / 2.0f); _y < yl + dx; _y += dx) {
        const float x = (_x < xl) ? _x : _x - dx / 2.0f, y = (_y < yl) ? _y : _y - dx / 2.0f;
        if (x / xl + y / yl > 1.0f - 1e-6f) continue;
        RigidSample s;
        for (int k = 0; k < 3; k++) s.off[k] = v0[k] + a[k] * x + b[k] * y;
        s.body = body; s.elem = (int)(elem0 + e);
        // align_with_rigid_body + near_boundary(*p) (:237-247): samples near the domain wall are not created
        float w[3];
        bool near_wall = false;
        for (int r = 0; r < 3; r++) {
          w[r] = (D.R[3 * r] * s.off[0] + D.R[3 * r + 1] * s.off[1] + D.R[3 * r + 2] * s.off[2] + D.pos[r]) * c->P.idx;
          near_wall = near_wall || w[r] < 7.0f || w[r] - (float)c->P.res[r] > -7.0f;
        }
        if (!near_wall) { R.h_smp.push_back(s); R.h_smp_id.push_back(c->next_pid + allocated); }
        allocated++;
      }
  }
  // every boundary particle took a creation id from the same counter as the material particles
  // (allocator.allocate_particle, src/particle_allocator.h:68-74): later material particles are numbered behind them
  c->next_pid += allocated;
  B.n_samples = (int)R.h_smp.size() - B.first_sample;
  B.first_elem = (int64_t)elem0; B.n_elems = n_triangles;
  R.h_elems.insert(R.h_elems.end(), tri.begin(), tri.end());
  // (re)upload samples and elements
  (void)hipFree(R.d_smp); (void)hipFree(R.d_elems);
  R.d_smp = nullptr; R.d_elems = nullptr;
  HIPCHK(c, dmalloc(&R.d_smp, std::max<size_t>(R.h_smp.size(), 1)));
  HIPCHK(c, dmalloc(&R.d_elems, std::max<size_t>(R.h_elems.size(), 9)));
  if (!R.h_smp.empty()) HIPCHK(c, hipMemcpy(R.d_smp, R.h_smp.data(), sizeof(RigidSample) * R.h_smp.size(), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(R.d_elems, R.h_elems.data(), sizeof(float) * R.h_elems.size(), hipMemcpyHostToDevice));
  R.n_smp = (uint32_t)R.h_smp.size();
  HIPCHK(c, hipMemcpy(R.d_rb + body, &D, sizeof D, hipMemcpyHostToDevice));
  R.bodies.push_back(B);
  return body;
}

int32_t mpmhip_num_rigid_bodies(const mpmhip_ctx *c) { return c && c->rigid.enabled ? (int32_t)c->rigid.bodies.size() : 1; }

// out[33]: position 3, quaternion (w, x, y, z) 4, velocity 3, angular velocity 3, mass, inv_mass, inertia 9 (body frame,
// row-major), inv_inertia 9 — the fields of RigidBody the coupling reads
int mpmhip_rigid_get_state(mpmhip_ctx *c, int32_t id, float *out) {
  if (!c || !out) return MPMHIP_EINVAL;
  if (!c->rigid.enabled || id < 1 || id >= (int)c->rigid.bodies.size()) return fail(c, MPMHIP_EINVAL, "no rigid body %d", id);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  RigidBodyDev D;
  HIPCHK(c, hipMemcpy(&D, c->rigid.d_rb + id, sizeof D, hipMemcpyDeviceToHost));
  int k = 0;
  for (int i = 0; i < 3; i++) out[k++] = D.pos[i];
  for (int i = 0; i < 4; i++) out[k++] = D.q[i];
  for (int i = 0; i < 3; i++) out[k++] = D.vel[i];
  for (int i = 0; i < 3; i++) out[k++] = D.omega[i];
  out[k++] = D.mass; out[k++] = D.inv_mass;
  for (int i = 0; i < 9; i++) out[k++] = c->rigid.bodies[id].inertia[i];
  for (int i = 0; i < 9; i++) out[k++] = D.inv_I[i];
  return MPMHIP_OK;
}
int mpmhip_rigid_set_velocity(mpmhip_ctx *c, int32_t id, const float *v, const float *w) {
  if (!c) return MPMHIP_EINVAL;
  if (!c->rigid.enabled || id < 1 || id >= (int)c->rigid.bodies.size()) return fail(c, MPMHIP_EINVAL, "no rigid body %d", id);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  RigidBodyDev D;
  HIPCHK(c, hipMemcpy(&D, c->rigid.d_rb + id, sizeof D, hipMemcpyDeviceToHost));
  for (int k = 0; k < 3; k++) { if (v) D.vel[k] = v[k]; if (w) D.omega[k] = w[k]; }
  HIPCHK(c, hipMemcpy(c->rigid.d_rb + id, &D, sizeof D, hipMemcpyHostToDevice));
  return MPMHIP_OK;
}
// boundary particles of body id (id < 0: all): world position, body-frame offset, body index.  Returns the count.
int64_t mpmhip_rigid_get_samples(mpmhip_ctx *c, int32_t id, int64_t cap, float *pos, float *offset, int32_t *body) {
  if (!c) return MPMHIP_EINVAL;
  if (!c->rigid.enabled) return 0;
  if (hipSetDevice(c->device) != hipSuccess) return MPMHIP_EHIP;
  auto &R = c->rigid;
  std::vector<float> w((size_t)R.n_smp * 3);
  if (R.n_smp && pos) {
    float *d = nullptr;
    HIPCHK(c, dmalloc(&d, (size_t)R.n_smp * 3));
    hipLaunchKernelGGL(k_rigid_sample_positions, dim3(particle_grid(R.n_smp)), dim3(256), 0, c->stream, (const RigidBodyDev *)R.d_rb,
                       (const RigidSample *)R.d_smp, R.n_smp, d);
    hipError_t e = hipMemcpyAsync(w.data(), d, sizeof(float) * w.size(), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    HIPCHK(c, e);
  }
  int64_t n = 0;
  for (size_t s = 0; s < R.h_smp.size(); s++) {
    if (id >= 0 && R.h_smp[s].body != id) continue;
    if (n < cap) {
      for (int k = 0; k < 3; k++) { if (pos) pos[3 * n + k] = w[3 * s + k]; if (offset) offset[3 * n + k] = R.h_smp[s].off[k]; }
      if (body) body[n] = R.h_smp[s].body;
    }
    n++;
  }
  return n;
}

// the body's triangles in world space (get_mesh_to_world applied to mesh->elements): what write_rigid_body puts into
// frame_directory/rigid_%03d_%04d.obj next to every .bgeo frame (src/visualize.cpp:102-154, src/mpm.h:333-343).
// out: 9 floats per triangle; returns the body's triangle count.
int64_t mpmhip_rigid_get_mesh(mpmhip_ctx *c, int32_t id, int64_t cap_triangles, float *out) {
  if (!c) return MPMHIP_EINVAL;
  if (!c->rigid.enabled || id < 1 || id >= (int)c->rigid.bodies.size()) return fail(c, MPMHIP_EINVAL, "no rigid body %d", id);
  if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return MPMHIP_EHIP;
  auto &R = c->rigid;
  RigidBodyDev D;
  HIPCHK(c, hipMemcpy(&D, R.d_rb + id, sizeof D, hipMemcpyDeviceToHost));
  const auto &B = R.bodies[id];
  if (out) {
    for (int64_t e = 0; e < std::min<int64_t>(B.n_elems, cap_triangles); e++)
      for (int q = 0; q < 3; q++) {
        const float *v = &R.h_elems[(size_t)(B.first_elem + e) * 9 + 3 * q];
        for (int r = 0; r < 3; r++) out[9 * e + 3 * q + r] = D.R[3 * r] * v[0] + D.R[3 * r + 1] * v[1] + D.R[3 * r + 2] * v[2] + D.pos[r];
      }
  }
  return B.n_elems;
}

// phase-level entry points (parity tests; substep() runs them itself)
int mpmhip_rasterize_rigid_boundary(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (!rigid_active(c)) return MPMHIP_OK;
  return do_rigid_rasterize(c);
}
int mpmhip_gather_cdf(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (!rigid_active(c)) return MPMHIP_OK;
  int rc = need_sorted(c, "gather_cdf");
  if (rc || (rc = do_rigid_gather(c))) return rc;
  return do_rigid_block_flags(c);
}
// general_action("add_articulation") (src/mpm.cpp:923-933): the joint's initialize() on the bodies' CURRENT poses
int mpmhip_add_articulation(mpmhip_ctx *c, const mpmhip_joint_config *cfg) {
  if (!c || !cfg) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  auto &R = c->rigid;
  const int nb = R.enabled ? (int)R.bodies.size() : 0;
  if (cfg->obj0 < 1 || cfg->obj0 >= nb) return fail(c, MPMHIP_EINVAL, "add_articulation: obj0 = %d is not a rigid body of this simulation", cfg->obj0);
  if (cfg->obj1 < 0 || cfg->obj1 >= nb) return fail(c, MPMHIP_EINVAL, "add_articulation: obj1 = %d is not a rigid body of this simulation", cfg->obj1);
  if (cfg->type < JOINT_ROTATION || cfg->type > JOINT_STEPPER) return fail(c, MPMHIP_EINVAL, "add_articulation: unknown type %d", cfg->type);
  if (c->in_substep) return fail(c, MPMHIP_EINVAL, "add_articulation inside a substep");
  if ((int)R.joints.size() >= MAX_JOINTS) return fail(c, MPMHIP_ECAPACITY, "at most %d articulations", MAX_JOINTS);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  JointBody jb[2];
  const int ids[2] = {cfg->obj0, cfg->obj1};
  for (int i = 0; i < 2; i++) {
    RigidBodyDev D;
    HIPCHK(c, hipMemcpy(&D, R.d_rb + ids[i], sizeof D, hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; k++) { jb[i].pos[k] = D.pos[k]; jb[i].vel[k] = D.vel[k]; jb[i].omega[k] = D.omega[k]; }
    for (int k = 0; k < 9; k++) { jb[i].R[k] = D.R[k]; jb[i].inv_I[k] = D.inv_I[k]; }
    jb[i].inv_mass = D.inv_mass;
  }
  JointConfig jc;
  jc.type = cfg->type; jc.obj0 = cfg->obj0; jc.obj1 = cfg->obj1; jc.has_offset1 = cfg->has_offset1; jc.has_target = cfg->has_target_distance;
  for (int k = 0; k < 3; k++) { jc.offset0[k] = cfg->offset0[k]; jc.offset1[k] = cfg->offset1[k]; jc.axis[k] = cfg->axis[k]; }
  jc.target_distance = cfg->target_distance; jc.penalty = cfg->penalty; jc.axis_length = cfg->axis_length;
  jc.power = cfg->power; jc.angular_velocity = cfg->angular_velocity;
  JointDev J;
  if (const char *why = joint_init(J, jc, jb[0], jb[1], R.bodies[cfg->obj0].inertia, R.bodies[cfg->obj1].inertia))
    return fail(c, MPMHIP_EINVAL, "add_articulation: %s", why);
  R.joints.push_back(J);
  HIPCHK(c, hipMemcpy(R.d_joints, R.joints.data(), sizeof(JointDev) * R.joints.size(), hipMemcpyHostToDevice));
  return MPMHIP_OK;
}
int32_t mpmhip_num_articulations(const mpmhip_ctx *c) { return c ? (int32_t)c->rigid.joints.size() : 0; }
int mpmhip_set_articulation_iterations(mpmhip_ctx *c, int32_t n) {
  if (!c || n < 0) return MPMHIP_EINVAL;
  c->rigid.joint_iterations = n;
  return MPMHIP_OK;
}
int mpmhip_articulate(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->rigid.enabled) return MPMHIP_OK;
  return do_rigid_articulate(c, c->P.dt);
}
int mpmhip_advect_rigid_bodies(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (!rigid_active(c)) return MPMHIP_OK;
  return do_rigid_advect(c, c->P.dt);
}
// dense (res+1)^3 views of the colored distance field: GridState::states (24 colour bits | body id + 1 << 24) and distance
int mpmhip_download_cdf(mpmhip_ctx *c, uint32_t *states, float *distance) {
  if (!c || !states || !distance) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t nodes = (size_t)(c->P.res[0] + 1) * (c->P.res[1] + 1) * (c->P.res[2] + 1);
  memset(states, 0, nodes * 4);
  memset(distance, 0, nodes * 4);
  if (!c->rigid.enabled) return MPMHIP_OK;
  uint32_t *ds = nullptr;
  float *dd = nullptr;
  HIPCHK(c, dmalloc(&ds, nodes));
  hipError_t e = dmalloc(&dd, nodes);
  if (e == hipSuccess) e = hipMemsetAsync(ds, 0, nodes * 4, c->stream);
  if (e == hipSuccess) e = hipMemsetAsync(dd, 0, nodes * 4, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_cdf_dense, dim3(1024), dim3(256), 0, c->stream, c->P, c->rigid.cdf, ds, dd);
    e = hipMemcpyAsync(states, ds, nodes * 4, hipMemcpyDeviceToHost, c->stream);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(distance, dd, nodes * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(ds); (void)hipFree(dd);
  HIPCHK(c, e);
  return MPMHIP_OK;
}
// what gather_cdf left for every live particle, in slot order (like mpmhip_download): boundary_normal 3, boundary_distance,
// near_boundary (0 / 1) — 5 floats per particle
int64_t mpmhip_download_boundary(mpmhip_ctx *c, float *out, int64_t n_capacity) {
  if (!c || !out) return MPMHIP_EINVAL;
  if (hipSetDevice(c->device) != hipSuccess) return MPMHIP_EHIP;
  std::vector<RecG> hg;
  int rc = fetch_records(c, hg, nullptr, nullptr);
  if (rc) return rc;
  std::vector<BndRec> hb(hg.size());
  if (c->rigid.enabled && !hg.empty()) HIPCHK(c, hipMemcpy(hb.data(), c->rigid.d_bnd, sizeof(BndRec) * hg.size(), hipMemcpyDeviceToHost));
  else memset(hb.data(), 0, sizeof(BndRec) * hb.size());
  int64_t m = 0;
  for (size_t i = 0; i < hg.size(); i++) {
    if (hg[i].pid < 0) continue;
    if (m >= n_capacity) return fail(c, MPMHIP_ECAPACITY, "download buffer too small");
    const bool fresh = hb[i].epoch == c->rigid.gather_epoch;  // (particles away from every body are not visited: zeros, as the reference leaves them)
    for (int k = 0; k < 3; k++) out[5 * m + k] = fresh ? hb[i].n[k] : 0.0f;
    out[5 * m + 3] = fresh ? hb[i].dist : 0.0f;
    out[5 * m + 4] = fresh ? (float)hb[i].near : 0.0f;
    m++;
  }
  return m;
}
