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
int mpmhip2d_bgeo_encode(mpmhip2d_ctx *m, int32_t verbose, void *dst, size_t capacity, size_t *written) {
  if (!m || !dst || !written) return MPMHIP_EINVAL;
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
  std::vector<uint8_t> img(bytes);
  if (int rc = mpmhip2d_bgeo_encode(m, verbose, img.data(), img.size(), &written)) return rc;
  FILE *f = std::fopen(path, "wb");
  if (!f) return fail2d(m, MPMHIP_EINVAL, std::string("cannot open '") + path + "' for writing: " + std::strerror(errno));
  const size_t ok = std::fwrite(img.data(), 1, written, f);
  const int cl = std::fclose(f);
  if (ok != written || cl != 0) return fail2d(m, MPMHIP_EINVAL, std::string("short write to '") + path + "'");
  return MPMHIP_OK;
}
