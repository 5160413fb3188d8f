// taichi_mpm_amd/csrc/frame2d_api.h — frame output of the 2D simulation object: MPM<2>::write_partio (src/visualize.cpp:17-100; the
// same Houdini .bgeo the 3D simulation writes, z = 0) — included by mpmhip.hip inside extern "C".  Part of libmpmhip; off the
// hot path: 2D scenes are small, so the rows are assembled on the host from one download (the 3D rows are built on the device,
// k_bgeo.h), with the header / footer writers of the 3D path.  With a resident asynchronous stepper the rows are the containers
// of every particle pool with their block's limits (AsyncMPM<2>::visualize, src/async/async_visualize.cpp:17-26,86-96).

static int64_t frame2d_rows(mpmhip2d_ctx *m) {
  int64_t n = m->async.resident ? mpmhip2d_async_load_pools(m) : mpmhip2d_num_particles(m);
  if (n < 0) return n;
  if (m->rigid_enabled) n += (int64_t)m->h_smp.size();  // the boundary particles of rigid bodies are rows too (type = 1)
  return n;
}
int mpmhip2d_bgeo_size(mpmhip2d_ctx *m, int32_t verbose, size_t *bytes) {
  if (!m || !bytes) return MPMHIP_EINVAL;
  const int64_t n = frame2d_rows(m);
  if (n < 0) return (int)n;
  *bytes = bgeo_bytes((uint32_t)n, verbose != 0);
  return MPMHIP_OK;
}
static int bgeo2d_encode(mpmhip2d_ctx *m, int32_t verbose, void *dst, size_t capacity, size_t *written);
int mpmhip2d_bgeo_encode(mpmhip2d_ctx *m, int32_t verbose, void *dst, size_t capacity, size_t *written) {
  if (!m || !dst || !written) return MPMHIP_EINVAL;
  try { return bgeo2d_encode(m, verbose, dst, capacity, written); }  // (host staging vectors: no exception leaves the C ABI)
  catch (const std::bad_alloc &) { return fail2d(m, MPMHIP_ENOMEM, "host allocation failed while assembling the frame"); }
}
static int bgeo2d_encode(mpmhip2d_ctx *m, int32_t verbose, void *dst, size_t capacity, size_t *written) {
  HIPCHK2D(m, hipSetDevice(m->device));
  auto &A = m->async;
  if (A.resident) {  // (a view of all pools; also pools particles added since)
    const int64_t r = mpmhip2d_async_load_pools(m);
    if (r < 0) return (int)r;
  }
  HIPCHK2D(m, hipStreamSynchronize(m->stream));
  const size_t ns = (size_t)m->n;
  std::vector<float> hx(2 * ns), hv(2 * ns), hB(4 * ns), ha(ns);
  std::vector<int32_t> hg(ns), hp(ns);
  std::vector<uint32_t> hstate, hblk;
  if (ns) {
    HIPCHK2D(m, hipMemcpy(hx.data(), m->x, 8 * ns, hipMemcpyDeviceToHost)); HIPCHK2D(m, hipMemcpy(hv.data(), m->v, 8 * ns, hipMemcpyDeviceToHost));
    HIPCHK2D(m, hipMemcpy(hB.data(), m->B, 16 * ns, hipMemcpyDeviceToHost)); HIPCHK2D(m, hipMemcpy(ha.data(), m->aux, 4 * ns, hipMemcpyDeviceToHost));
    HIPCHK2D(m, hipMemcpy(hg.data(), m->gid, 4 * ns, hipMemcpyDeviceToHost)); HIPCHK2D(m, hipMemcpy(hp.data(), m->pid, 4 * ns, hipMemcpyDeviceToHost));
    if (m->rigid_enabled) { hstate.resize(ns); HIPCHK2D(m, hipMemcpy(hstate.data(), m->d_states, 4 * ns, hipMemcpyDeviceToHost)); }
    if (A.resident && A.view) { hblk.resize(ns); HIPCHK2D(m, hipMemcpy(hblk.data(), A.d_blk_of, 4 * ns, hipMemcpyDeviceToHost)); }
  }
  // live material particles in ascending creation id (write_partio sorts by id, :39-43; stable: an id can sit in several pools)
  std::vector<uint32_t> order;
  order.reserve(ns);
  for (size_t i = 0; i < ns; i++) if (hp[i] >= 0) order.push_back((uint32_t)i);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hp[a] < hp[b]; });
  const bool with_rigid = m->rigid_enabled && !m->h_smp.empty();
  const uint32_t nm = (uint32_t)order.size(), n = nm + (with_rigid ? (uint32_t)m->h_smp.size() : 0u);
  const size_t total = bgeo_bytes(n, verbose != 0);
  if (total > capacity) return fail2d(m, MPMHIP_ECAPACITY, "bgeo image larger than the buffer");
  std::vector<mpm2d::Rigid2> hb;
  if (with_rigid) { hb.resize(mpm2d::MAX_RIGID2); HIPCHK2D(m, hipMemcpy(hb.data(), m->d_rb, sizeof(mpm2d::Rigid2) * mpm2d::MAX_RIGID2, hipMemcpyDeviceToHost)); }
  uint8_t *out = static_cast<uint8_t *>(dst);
  const std::vector<uint8_t> head = bgeo_header(n, verbose != 0);
  std::memcpy(out, head.data(), head.size());
  out += head.size();
  const size_t W = verbose ? BGEO_W_VERBOSE : BGEO_W_PLAIN;
  auto be32 = [](uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; };
  auto bef = [&](uint8_t *p, float f) { uint32_t u; std::memcpy(&u, &f, 4); be32(p, u); };
  size_t im = 0, ir = 0;
  for (uint32_t j = 0; j < n; j++, out += W * 4) {
    std::memset(out, 0, W * 4);
    const bool take_rigid = with_rigid && ir < m->h_smp.size() && (im >= nm || m->h_smp_id[ir] < hp[order[im]]);
    if (take_rigid) {  // position = anchor point, type = 1, v = the body's velocity there; the rest keeps the constructor values
      const mpm2d::Sample2 &S = m->h_smp[ir];
      const mpm2d::Rigid2 &B = hb[S.body];
      const float cs = std::cos(B.angle), sn = std::sin(B.angle);
      const float r0 = cs * S.off[0] - sn * S.off[1], r1 = sn * S.off[0] + cs * S.off[1];
      bef(out, r0 + B.pos[0]); bef(out + 4, r1 + B.pos[1]); bef(out + 12, 1.0f);
      be32(out + 16, 1u); be32(out + 20, (uint32_t)m->h_smp_id[ir]);
      be32(out + 24, 1u); be32(out + 28, 1u); be32(out + 32, 1u);
      bef(out + 36, B.vel[0] - B.omega * r1); bef(out + 40, B.vel[1] + B.omega * r0);
      ir++;
      continue;
    }
    const uint32_t s = order[im++];
    bef(out, hx[2 * s]); bef(out + 4, hx[2 * s + 1]); bef(out + 12, 1.0f);  // z = 0; homogeneous coordinate (BGEO.cpp:158-160)
    be32(out + 20, (uint32_t)hp[s]);
    uint32_t lim[3] = {1u, 1u, 1u};
    if (!hblk.empty()) { const uint32_t b = hblk[s]; lim[0] = (uint32_t)A.continuous[b]; lim[1] = (uint32_t)A.strength[b]; lim[2] = (uint32_t)A.cfl[b]; }
    be32(out + 24, lim[0]); be32(out + 28, lim[1]); be32(out + 32, lim[2]);
    bef(out + 36, hv[2 * s]); bef(out + 40, hv[2 * s + 1]);
    if (verbose) {
      const GroupParams &gp = m->groups[hg[s]];
      bef(out + 48, gp.p[0]);  // m
      const float d0 = gp.type == MPMHIP_WATER ? ha[s] : gp.type == MPMHIP_ELASTIC ? gp.p[4] : 0.0f;  // get_debug_info()
      bef(out + 64, d0); bef(out + 68, (float)gp.type);
      be32(out + 76, hstate.empty() ? 0u : hstate[s]);
      const float h = 0.5f * (hB[4 * s + 1] - hB[4 * s + 2]);  // || 0.5 (apic_b - apic_b^T) ||_F
      bef(out + 88, std::sqrt(h * h + h * h));
    }
  }
  const std::vector<uint8_t> prim = bgeo_prim_attr();
  std::memcpy(out, prim.data(), prim.size());
  out += prim.size();
  auto w32 = [&](uint32_t v) { for (int s = 24; s >= 0; s -= 8) *out++ = (uint8_t)(v >> s); };
  w32(n);
  if (n > (1u << 16)) { for (uint32_t i = 0; i < n; i++) w32(i); }  // BGEO.cpp:175-180
  else { for (uint32_t i = 0; i < n; i++) { *out++ = (uint8_t)(i >> 8); *out++ = (uint8_t)i; } }
  w32(0);
  *out++ = 0x00;
  *out++ = 0xff;
  *written = (size_t)(out - static_cast<uint8_t *>(dst));
  if (*written != total) return fail2d(m, MPMHIP_EINVAL, "internal: bgeo image size mismatch");
  return MPMHIP_OK;
}
int mpmhip2d_write_bgeo(mpmhip2d_ctx *m, const char *path, int32_t verbose) {
  if (!m || !path) return MPMHIP_EINVAL;
  size_t bytes = 0, written = 0;
  if (int rc = mpmhip2d_bgeo_size(m, verbose, &bytes)) return rc;
  std::vector<uint8_t> img;
  try { img.resize(bytes); } catch (const std::bad_alloc &) { return fail2d(m, MPMHIP_ENOMEM, "host allocation failed while assembling the frame"); }
  if (int rc = mpmhip2d_bgeo_encode(m, verbose, img.data(), img.size(), &written)) return rc;
  FILE *f = std::fopen(path, "wb");
  if (!f) return fail2d(m, MPMHIP_EINVAL, std::string("cannot open '") + path + "' for writing: " + std::strerror(errno));
  const size_t ok = std::fwrite(img.data(), 1, written, f);
  const int cl = std::fclose(f);
  if (ok != written || cl != 0) return fail2d(m, MPMHIP_EINVAL, std::string("short write to '") + path + "'");
  return MPMHIP_OK;
}

// ---- snapshots of the 2D simulation — MPM<2>::general_action(action = 'save' | 'load') (src/mpm.cpp:940-960; the reference
// serialises the particles, the rigid bodies and — AsyncMPM — every pool and the block table, src/async/async_mpm.h:120-172).
// Blob: header | group table | the particle arrays | [the bodies' records and the joints] | [the asynchronous stepper: block
// tables, clocks, the store's live containers].  Like the 3D snapshots it is loaded into an object of the same grid whose scene
// has been set up again: level set, configuration, the rigid bodies' outlines and scripts come from the scene, not the blob.
struct Snap2D {
  char magic[8];  // "MPM2DSNP"
  uint32_t abi, n_groups;
  int32_t res[2], next_pid, n_bodies, n_joints, has_async;
  int64_t n;
  float dx, base_dt, t, request_t;
  uint32_t n_dead, n_ranked;
  // asynchronous stepper (has_async)
  int32_t nb[2];
  float unit_delta_t, a_request_t, a_current_t, pad;
  int64_t nblk, containers, current_t_int, min_delta_t_int, max_delta_t_int, update_counter, step_counter;
};
static int snap2d_prepare(mpmhip2d_ctx *m) {  // a state the blob can describe: no view, counters settled, the store compacted
  HIPCHK2D(m, hipSetDevice(m->device));
  if (m->async.resident) {
    if (int rc = mpmhip2d_async_pool_particles(m)) return rc;  // (drops a view; pools what was added since)
    if (int rc = a2_settle(m)) return rc;
    if (int rc = a2_compact(m)) return rc;
  }
  HIPCHK2D(m, hipStreamSynchronize(m->stream));
  return MPMHIP_OK;
}
static size_t snap2d_bytes(const mpmhip2d_ctx *m) {
  size_t b = sizeof(Snap2D) + sizeof(GroupParams) * m->groups.size() + (size_t)m->n * (2 + 2 + 4 + 4 + 1 + 1 + 1) * 4;
  // (with rigid bodies also the particles' colour words — MPMParticle::states, the sticky history of which side of a body a particle is on)
  if (m->rigid_enabled) b += sizeof(mpm2d::Rigid2) * m->bodies.size() + sizeof(mpm2d::Joints2) + sizeof(uint32_t) * m->n_ranked + 4 * (size_t)m->n;
  if (m->async.resident) b += sizeof(int64_t) * 6 * m->async.nblk() + (size_t)m->async.size * (4 + 64);
  return b;
}
int64_t mpmhip2d_snapshot_size(mpmhip2d_ctx *m) {
  if (!m) return MPMHIP_EINVAL;
  if (int rc = snap2d_prepare(m)) return rc;
  return (int64_t)snap2d_bytes(m);
}
int mpmhip2d_snapshot_save(mpmhip2d_ctx *m, void *dst, size_t cap) {
  if (!m || !dst) return MPMHIP_EINVAL;
  if (int rc = snap2d_prepare(m)) return rc;
  if (cap < snap2d_bytes(m)) return fail2d(m, MPMHIP_ECAPACITY, "snapshot buffer too small");
  auto &A = m->async;
  Snap2D h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, "MPM2DSNP", 8);
  h.abi = MPMHIP_ABI_VERSION; h.n_groups = (uint32_t)m->groups.size();
  h.res[0] = m->P.res[0]; h.res[1] = m->P.res[1]; h.next_pid = m->next_pid;
  h.n_bodies = m->rigid_enabled ? (int32_t)m->bodies.size() : 0; h.n_joints = m->joints.n; h.has_async = A.resident ? 1 : 0;
  h.n = m->n; h.dx = m->P.dx; h.base_dt = m->base_dt; h.t = m->t; h.request_t = m->request_t; h.n_ranked = m->rigid_enabled ? m->n_ranked : 0;
  HIPCHK2D(m, hipMemcpy(&h.n_dead, m->n_dead, 4, hipMemcpyDeviceToHost));
  if (A.resident) {
    h.nb[0] = A.nb[0]; h.nb[1] = A.nb[1]; h.unit_delta_t = A.cfg.unit_delta_t; h.a_request_t = A.request_t; h.a_current_t = A.current_t;
    h.nblk = (int64_t)A.nblk(); h.containers = A.size; h.current_t_int = A.current_t_int; h.min_delta_t_int = A.min_delta_t_int;
    h.max_delta_t_int = A.max_delta_t_int; h.update_counter = A.update_counter; h.step_counter = A.step_counter;
  }
  char *p = (char *)dst;
  auto put = [&](const void *src, size_t bytes) { memcpy(p, src, bytes); p += bytes; };
  auto put_dev = [&](const void *src, size_t bytes) -> hipError_t {
    const hipError_t e = bytes ? hipMemcpy(p, src, bytes, hipMemcpyDeviceToHost) : hipSuccess;
    p += bytes;
    return e;
  };
  put(&h, sizeof h);
  put(m->groups.data(), sizeof(GroupParams) * m->groups.size());
  const size_t n = (size_t)m->n;
  HIPCHK2D(m, put_dev(m->x, 8 * n)); HIPCHK2D(m, put_dev(m->v, 8 * n)); HIPCHK2D(m, put_dev(m->F, 16 * n)); HIPCHK2D(m, put_dev(m->B, 16 * n));
  HIPCHK2D(m, put_dev(m->aux, 4 * n)); HIPCHK2D(m, put_dev(m->gid, 4 * n)); HIPCHK2D(m, put_dev(m->pid, 4 * n));
  if (m->rigid_enabled) {
    HIPCHK2D(m, put_dev(m->d_rb, sizeof(mpm2d::Rigid2) * m->bodies.size()));
    put(&m->joints, sizeof m->joints);
    HIPCHK2D(m, put_dev(m->d_smp_rank, sizeof(uint32_t) * m->n_ranked));
    HIPCHK2D(m, put_dev(m->d_states, 4 * n));
  }
  if (A.resident) {
    const size_t nb = sizeof(int64_t) * A.nblk();
    for (const std::vector<int64_t> *v : {&A.continuous, &A.strength, &A.cfl, &A.particle_t, &A.backup_t, &A.local_min}) put(v->data(), nb);
    HIPCHK2D(m, put_dev(A.tag, 4 * (size_t)A.size)); HIPCHK2D(m, put_dev(A.rec, 64 * (size_t)A.size));
  }
  return MPMHIP_OK;
}
static int snap2d_load(mpmhip2d_ctx *m, const void *src, size_t size);
int mpmhip2d_snapshot_load(mpmhip2d_ctx *m, const void *src, size_t size) {
  if (!m || !src || size < sizeof(Snap2D)) return MPMHIP_EINVAL;
  try { return snap2d_load(m, src, size); }
  catch (const std::bad_alloc &) { return fail2d(m, MPMHIP_ENOMEM, "host allocation failed while loading the snapshot"); }
}
static int snap2d_load(mpmhip2d_ctx *m, const void *src, size_t size) {
  HIPCHK2D(m, hipSetDevice(m->device));
  HIPCHK2D(m, hipStreamSynchronize(m->stream));
  auto &A = m->async;
  Snap2D h;
  memcpy(&h, src, sizeof h);
  if (memcmp(h.magic, "MPM2DSNP", 8) != 0 || h.abi != MPMHIP_ABI_VERSION) return fail2d(m, MPMHIP_EINVAL, "not a 2D snapshot of this ABI version");
  if (h.res[0] != m->P.res[0] || h.res[1] != m->P.res[1] || h.dx != m->P.dx) return fail2d(m, MPMHIP_EINVAL, "snapshot is of another grid");
  if (h.n < 0 || h.next_pid < 0 || h.n_groups > (uint32_t)MPMHIP_MAX_GROUPS || h.n_joints < 0 || h.n_joints > mpm2d::MAX_JOINTS2 || h.containers < 0 || h.nblk < 0)
    return fail2d(m, MPMHIP_EINVAL, "snapshot header inconsistent");
  const int bodies_here = m->rigid_enabled ? (int)m->bodies.size() : 0;
  if (h.n_bodies != bodies_here) return fail2d(m, MPMHIP_EINVAL, "snapshot has another number of rigid bodies than the scene: add the scene's bodies before loading");
  if ((h.has_async != 0) != A.resident) return fail2d(m, MPMHIP_EINVAL, "snapshot and simulation differ in being asynchronous steppers");
  if (h.has_async && (h.nb[0] != A.nb[0] || h.nb[1] != A.nb[1] || h.nblk != (int64_t)A.nblk() || h.unit_delta_t != A.cfg.unit_delta_t))
    return fail2d(m, MPMHIP_EINVAL, "snapshot is of another block table / unit_delta_t");
  if (h.n_bodies && h.n_ranked > (uint32_t)m->h_smp.size()) return fail2d(m, MPMHIP_EINVAL, "snapshot holds more boundary particles than the scene's bodies have");
  // (counts from an untrusted header: bounded by the blob before they are multiplied)
  if ((uint64_t)h.n > size / 60 || (uint64_t)h.containers > size / 68 || (uint64_t)h.nblk > size / 48 || h.n_ranked > size / 4)
    return fail2d(m, MPMHIP_EINVAL, "snapshot size mismatch");
  const size_t n = (size_t)h.n;
  size_t want = sizeof(Snap2D) + sizeof(GroupParams) * h.n_groups + n * 15 * 4;
  if (h.n_bodies) want += sizeof(mpm2d::Rigid2) * (size_t)h.n_bodies + sizeof(mpm2d::Joints2) + sizeof(uint32_t) * h.n_ranked + 4 * n;
  if (h.has_async) want += sizeof(int64_t) * 6 * (size_t)h.nblk + (size_t)h.containers * 68;
  if (size != want) return fail2d(m, MPMHIP_EINVAL, "snapshot size mismatch");
  const char *p = (const char *)src + sizeof h;
  // ---- validate everything that indexes device tables before anything of the object is touched
  const GroupParams *gs = reinterpret_cast<const GroupParams *>(p);
  for (uint32_t g = 0; g < h.n_groups; g++) {
    GroupParams gp;
    memcpy(&gp, gs + g, sizeof gp);
    if (gp.type < MPMHIP_VISCO || gp.type > MPMHIP_ELASTIC) return fail2d(m, MPMHIP_EINVAL, "snapshot holds an unknown material id");
  }
  const char *arr = p + sizeof(GroupParams) * h.n_groups;
  {
    const int32_t *gid = reinterpret_cast<const int32_t *>(arr + n * 13 * 4), *pid = gid + n;
    for (size_t i = 0; i < n; i++) {
      int32_t g, id;
      memcpy(&g, gid + i, 4); memcpy(&id, pid + i, 4);
      if (g < 0 || (uint32_t)g >= h.n_groups || id >= h.next_pid) return fail2d(m, MPMHIP_EINVAL, "snapshot particle names an unknown group or id");
    }
  }
  const char *q = arr + n * 15 * 4;
  if (h.n_bodies) {  // the joints index the body table on the device (k2_articulate)
    mpm2d::Joints2 J;
    memcpy(&J, q + sizeof(mpm2d::Rigid2) * (size_t)h.n_bodies, sizeof J);
    if (J.n < 0 || J.n > mpm2d::MAX_JOINTS2) return fail2d(m, MPMHIP_EINVAL, "snapshot joint table inconsistent");
    for (int i = 0; i < J.n; i++)
      if (J.j[i].obj0 < 0 || J.j[i].obj0 >= h.n_bodies || J.j[i].obj1 < 0 || J.j[i].obj1 >= h.n_bodies)
        return fail2d(m, MPMHIP_EINVAL, "snapshot joint names a body outside the scene's table");
    q += sizeof(mpm2d::Rigid2) * (size_t)h.n_bodies + sizeof(mpm2d::Joints2) + sizeof(uint32_t) * h.n_ranked + 4 * n;
  }
  if (h.has_async && !AsyncSched::limits_are_sane(q, (size_t)h.nblk))
    return fail2d(m, MPMHIP_EINVAL, "snapshot block table holds a time-step limit that is not a power of two in [1, 2^31]");
  if (h.has_async) {
    const char *tags = q + sizeof(int64_t) * 6 * (size_t)h.nblk, *recs = tags + 4 * (size_t)h.containers;
    for (size_t i = 0; i < (size_t)h.containers; i++) {
      uint32_t tg;
      memcpy(&tg, tags + 4 * i, 4);
      if (tg == AS_FREE) continue;
      if ((int64_t)(tg & ~AS_BACKUP) >= h.nblk) return fail2d(m, MPMHIP_EINVAL, "snapshot container names an unknown block");
      int32_t g, id;
      memcpy(&g, recs + 64 * i + 52, 4); memcpy(&id, recs + 64 * i + 56, 4);
      if (g < 0 || (uint32_t)g >= h.n_groups || id < 0 || id >= h.next_pid) return fail2d(m, MPMHIP_EINVAL, "snapshot container names an unknown group or id");
    }
  }
  // ---- load
  if (int rc = a2_grow_particles_any(m, (int64_t)n)) return rc;
  m->groups.assign(gs, gs + h.n_groups);
  if (h.n_groups) HIPCHK2D(m, hipMemcpy(m->d_groups, m->groups.data(), sizeof(GroupParams) * h.n_groups, hipMemcpyHostToDevice));
  auto get_dev = [&](void *dst, size_t bytes) -> hipError_t {
    const hipError_t e = bytes ? hipMemcpy(dst, arr, bytes, hipMemcpyHostToDevice) : hipSuccess;
    arr += bytes;
    return e;
  };
  HIPCHK2D(m, get_dev(m->x, 8 * n)); HIPCHK2D(m, get_dev(m->v, 8 * n)); HIPCHK2D(m, get_dev(m->F, 16 * n)); HIPCHK2D(m, get_dev(m->B, 16 * n));
  HIPCHK2D(m, get_dev(m->aux, 4 * n)); HIPCHK2D(m, get_dev(m->gid, 4 * n)); HIPCHK2D(m, get_dev(m->pid, 4 * n));
  m->n = h.n; m->next_pid = h.next_pid; m->t = h.t; m->request_t = h.request_t;
  HIPCHK2D(m, hipMemcpy(m->n_dead, &h.n_dead, 4, hipMemcpyHostToDevice));
  if (h.n_bodies) {
    HIPCHK2D(m, get_dev(m->d_rb, sizeof(mpm2d::Rigid2) * (size_t)h.n_bodies));
    memcpy(&m->joints, arr, sizeof m->joints); arr += sizeof m->joints;
    if (m->joints.n < 0 || m->joints.n > mpm2d::MAX_JOINTS2) { m->joints.n = 0; return fail2d(m, MPMHIP_EINVAL, "snapshot joint table inconsistent"); }
    (void)hipFree(m->d_smp_rank); m->d_smp_rank = nullptr; m->n_ranked = 0;
    if (h.n_ranked) {
      HIPCHK2D(m, dmalloc(&m->d_smp_rank, (size_t)m->h_smp.size() + m->h_smp.size() / 4 + 1024));
      HIPCHK2D(m, get_dev(m->d_smp_rank, sizeof(uint32_t) * h.n_ranked));
      m->n_ranked = h.n_ranked;
    }
    HIPCHK2D(m, get_dev(m->d_states, 4 * n));
  }
  if (h.has_async) {
    const size_t nb = sizeof(int64_t) * (size_t)h.nblk;
    for (std::vector<int64_t> *v : {&A.continuous, &A.strength, &A.cfl, &A.particle_t, &A.backup_t, &A.local_min}) { memcpy(v->data(), arr, nb); arr += nb; }
    A.size = A.size_ub = A.live = 0;
    if (int rc = a2_store_reserve(m, (uint32_t)h.containers + 1024)) return rc;
    HIPCHK2D(m, hipMemset(A.tag, 0xFF, sizeof(uint32_t) * (size_t)A.cap));
    HIPCHK2D(m, get_dev(A.tag, 4 * (size_t)h.containers)); HIPCHK2D(m, get_dev(A.rec, 64 * (size_t)h.containers));
    A.size = A.size_ub = A.live = (uint32_t)h.containers;
    AsyncCounters z;
    memset(&z, 0, sizeof z);
    z.size = (uint32_t)h.containers;
    HIPCHK2D(m, hipMemcpy(A.d_cnt, &z, sizeof z, hipMemcpyHostToDevice));
    A.pending_counters = false; A.view = false;
    A.current_t_int = h.current_t_int; A.min_delta_t_int = h.min_delta_t_int; A.max_delta_t_int = h.max_delta_t_int;
    A.update_counter = h.update_counter; A.step_counter = h.step_counter; A.request_t = h.a_request_t; A.current_t = h.a_current_t;
    A.limits_version++;  // (the neighbour lists are rebuilt from the loaded limits at the next update)
    if (m->next_pid > A.best_cap) { if (int rc = a2_best_reserve(m)) return rc; }
  }
  return MPMHIP_OK;
}
