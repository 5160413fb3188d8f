// taichi_mpm_amd/csrc/async2d_api.h — host side of AsyncMPM<2> (create_simulation2('async_mpm'): TC_IMPLEMENTATION(Simulation2D,
// AsyncMPM2D, "async_mpm"), src/async/async_mpm.cpp:423-427) over the 2D simulation object; included by mpmhip.hip inside
// extern "C".  Part of libmpmhip.  Device side: k_async2d.h; the block scheduler (limits, neighbour lists, action tables): the
// AsyncSched of async_sched.h, shared with the 3D stepper of async_api.h — read that file's header for the design: every
// container stays in HBM, an advance costs four small kernels around one MPM<2>::substep and one 32-byte read-back.

static int a2_grid(uint32_t n) { return (int)std::min<uint32_t>(std::max<uint32_t>((n + 255) / 256, 1), 2048); }

// the particle arrays of the 2D object hold at least `need` particles (its first m->n stay)
static int a2_grow_particles(mpmhip2d_ctx *m, int64_t need) {
  if (need <= m->cap) return MPMHIP_OK;
  HIPCHK2D(m, hipStreamSynchronize(m->stream));
  const size_t cap = (size_t)need + (size_t)need / 2 + 1024, keep = (size_t)m->n;
  hipError_t e = hipSuccess;
  auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  A(regrow(&m->x, 2 * keep, 2 * cap, false)); A(regrow(&m->v, 2 * keep, 2 * cap, false)); A(regrow(&m->F, 4 * keep, 4 * cap, false));
  A(regrow(&m->B, 4 * keep, 4 * cap, false)); A(regrow(&m->aux, keep, cap, false)); A(regrow(&m->gid, keep, cap, false));
  A(regrow(&m->pid, keep, cap, false));
  if (e != hipSuccess) return fail2d(m, MPMHIP_ENOMEM, std::string("growing the particle arrays failed: ") + hipGetErrorString(e));
  m->cap = (int64_t)cap;
  return MPMHIP_OK;
}
// the same for any 2D object (a snapshot larger than max_particles): the per-particle arrays of the CPIC coupling grow with them
static int a2_grow_particles_any(mpmhip2d_ctx *m, int64_t need) {
  if (need <= m->cap) return MPMHIP_OK;
  const size_t keep = (size_t)m->n;
  if (int rc = a2_grow_particles(m, need)) return rc;
  if (m->rigid_enabled) {
    hipError_t e = regrow(&m->d_states, keep, (size_t)m->cap, true);
    if (e == hipSuccess) e = regrow(&m->d_bnd, keep, (size_t)m->cap, true);
    if (e != hipSuccess) return fail2d(m, MPMHIP_ENOMEM, std::string("growing the particle arrays failed: ") + hipGetErrorString(e));
  }
  return MPMHIP_OK;
}
static int a2_store_reserve(mpmhip2d_ctx *m, uint32_t need) {  // room for `need` containers in total
  auto &A = m->async;
  if (need <= A.cap) return MPMHIP_OK;
  HIPCHK2D(m, hipStreamSynchronize(m->stream));
  const uint32_t cap = std::max<uint32_t>(need + need / 2, 4096), keep = std::min(A.size_ub, A.cap);
  hipError_t e = regrow(&A.rec, (size_t)keep * 4, (size_t)cap * 4, false);
  if (e == hipSuccess) e = regrow(&A.tag, (size_t)keep, (size_t)cap, false);
  // (kernels walk [0, upper bound of the size): every tag behind the containers in use says FREE)
  if (e == hipSuccess) e = hipMemset(A.tag + keep, 0xFF, sizeof(uint32_t) * (size_t)(cap - keep));
  (void)hipFree(A.rec2); (void)hipFree(A.tag2);
  A.rec2 = nullptr; A.tag2 = nullptr;  // (the compaction targets are re-allocated when a compaction runs)
  if (e != hipSuccess) return fail2d(m, MPMHIP_ENOMEM, std::string("async store: growing failed: ") + hipGetErrorString(e));
  A.cap = cap;
  return MPMHIP_OK;
}
// the action table -> the device, ordered on the object's (non-blocking) stream, from a ring of four pinned images (see
// async_upload_tbl of async_api.h)
static int a2_upload_tbl(mpmhip2d_ctx *m) {
  auto &A = m->async;
  const size_t nblk = A.tbl.size();
  uint8_t *img = A.h_tbl_pin + (size_t)(A.pin_next++ & 3) * nblk;
  memcpy(img, A.tbl.data(), nblk);
  HIPCHK2D(m, hipMemcpyAsync(A.d_tbl, img, nblk, hipMemcpyHostToDevice, m->stream));
  return MPMHIP_OK;
}
static int a2_best_reserve(mpmhip2d_ctx *m) {  // one dedup word per creation id
  auto &A = m->async;
  if ((int64_t)m->next_pid <= A.best_cap) return MPMHIP_OK;
  const size_t cap = (size_t)m->next_pid + (size_t)m->next_pid / 2 + 1024;
  (void)hipFree(A.best); A.best = nullptr;
  HIPCHK2D(m, dmalloc(&A.best, cap));
  HIPCHK2D(m, hipMemsetAsync(A.best, 0xFF, sizeof(unsigned long long) * cap, m->stream));
  A.best_cap = (int64_t)cap;
  return MPMHIP_OK;
}
// the ONE read-back of an advance: the transient counters (reset behind the copy) and the append cursor
static int a2_counters(mpmhip2d_ctx *m, AsyncCounters &h) {
  auto &A = m->async;
  HIPCHK2D(m, hipMemcpyAsync(A.h_cnt, A.d_cnt, sizeof h, hipMemcpyDeviceToHost, m->stream));
  HIPCHK2D(m, hipMemsetAsync(A.d_cnt, 0, 16, m->stream));
  HIPCHK2D(m, hipStreamSynchronize(m->stream));
  h = *A.h_cnt;
  if (h.pad[0] & mpm::SCAN_ERROR_BIT)
    return fail2d(m, MPMHIP_EHIP, "async store: a chained scan of the compaction waited in vain for a chunk that never published (k_sort.h)");
  A.live += h.n_append; A.live -= std::min(A.live, h.n_freed);
  A.size = A.size_ub = h.size;
  A.pending_counters = false;
  return MPMHIP_OK;
}
static int a2_settle(mpmhip2d_ctx *m) {
  if (!m->async.pending_counters) return MPMHIP_OK;
  AsyncCounters h;
  return a2_counters(m, h);
}
static int a2_compact(mpmhip2d_ctx *m) {
  auto &A = m->async;
  if (A.size == 0) return MPMHIP_OK;
  if (!A.rec2) {
    hipError_t e = dmalloc(&A.rec2, (size_t)A.cap * 4);
    if (e == hipSuccess) e = dmalloc(&A.tag2, (size_t)A.cap);
    if (e != hipSuccess) return fail2d(m, MPMHIP_ENOMEM, std::string("async store: compaction buffers: ") + hipGetErrorString(e));
  }
  HIPCHK2D(m, hipMemsetAsync(A.tag2, 0xFF, sizeof(uint32_t) * (size_t)A.cap, m->stream));
  const uint32_t nchunks = (A.size + 1023) / 1024;
  if (nchunks + 1 > A.scan_cap) {
    (void)hipFree(A.d_scan); A.d_scan = nullptr;
    HIPCHK2D(m, dmalloc(&A.d_scan, (size_t)nchunks + 1024));
    HIPCHK2D(m, hipMemsetAsync(A.d_scan, 0, sizeof(unsigned long long) * ((size_t)nchunks + 1024), m->stream));
    A.scan_cap = nchunks + 1024;
  }
  // (a chained scan: every launched workgroup must be resident — 64 of them always are)
  hipLaunchKernelGGL(mpm2d::k2a_compact, dim3(std::min<uint32_t>(nchunks, 64)), dim3(256), 0, m->stream, A.size, (const uint32_t *)A.tag,
                     (const float4 *)A.rec, A.tag2, A.rec2, A.d_scan, ++A.scan_epoch, A.d_cnt);
  HIPCHK2D(m, hipGetLastError());
  std::swap(A.rec, A.rec2); std::swap(A.tag, A.tag2);
  AsyncCounters h;
  if (int rc = a2_counters(m, h)) return rc;  // (size = live = what the scan counted)
  A.live = h.size;
  A.compactions++;
  return MPMHIP_OK;
}
// the object's arrays hold a view of the pools (mpmhip2d_async_load_pools): copies, dropped before anything else uses the arrays
static int a2_drop_view(mpmhip2d_ctx *m) {
  auto &A = m->async;
  if (!A.resident || !A.view) return MPMHIP_OK;
  A.view = false;
  m->n = 0;
  HIPCHK2D(m, hipMemsetAsync(m->n_dead, 0, sizeof(unsigned int), m->stream));
  return MPMHIP_OK;
}

// AsyncMPM<2>::initialize (src/async/async_mpm.cpp:13-55) on top of mpmhip2d_create
int mpmhip2d_async_begin(mpmhip2d_ctx *m, const mpmhip_async_config *cfg) {
  if (!m || !cfg) return MPMHIP_EINVAL;
  if (!(cfg->unit_delta_t > 0) || cfg->max_units < 1) return fail2d(m, MPMHIP_EINVAL, "unit_delta_t > 0 and max_units >= 1 required");
  if (m->rigid_enabled) return fail2d(m, MPMHIP_EINVAL, "asynchronous stepping cannot be combined with rigid bodies");
  HIPCHK2D(m, hipSetDevice(m->device));
  HIPCHK2D(m, hipStreamSynchronize(m->stream));
  auto &A = m->async;
  A.sched_enable(2, m->P.res, *cfg);
  A.sched_begin();
  const size_t nblk = A.nblk();
  a2_free(m);
  HIPCHK2D(m, dmalloc(&A.d_tab, 3 * nblk));
  HIPCHK2D(m, dmalloc(&A.d_tbl, nblk));
  HIPCHK2D(m, dmalloc(&A.d_rank, nblk));
  HIPCHK2D(m, dmalloc(&A.d_cnt, 1));
  HIPCHK2D(m, hipHostMalloc((void **)&A.h_tab, sizeof(uint32_t) * 3 * nblk, hipHostMallocDefault));
  HIPCHK2D(m, hipHostMalloc((void **)&A.h_tbl_pin, 4 * nblk, hipHostMallocDefault));
  HIPCHK2D(m, hipHostMalloc((void **)&A.h_cnt, sizeof(AsyncCounters), hipHostMallocDefault));
  HIPCHK2D(m, hipMemcpy(A.d_rank, A.rank_of.data(), sizeof(uint32_t) * nblk, hipMemcpyHostToDevice));
  HIPCHK2D(m, hipMemset(A.d_cnt, 0, sizeof(AsyncCounters)));
  A.cap = A.size = A.size_ub = A.live = 0;
  A.best_cap = 0; A.scan_cap = 0; A.scan_epoch = 0; A.pin_next = 0; A.compactions = 0;
  A.pending_counters = false; A.view = false;
  A.resident = true;
  return MPMHIP_OK;
}

// AsyncMPM<2>::add_particles (src/async/async_mpm.cpp:57-75): the particles currently in the object's arrays (just added by
// mpmhip2d_add_particles) move to the particle pools of their blocks; the arrays are empty afterwards.
int mpmhip2d_async_pool_particles(mpmhip2d_ctx *m) {
  if (!m) return MPMHIP_EINVAL;
  auto &A = m->async;
  if (!A.resident) return fail2d(m, MPMHIP_EINVAL, "mpmhip2d_async_begin first");
  HIPCHK2D(m, hipSetDevice(m->device));
  if (int rc = a2_drop_view(m)) return rc;
  if (m->n == 0) return MPMHIP_OK;
  if (int rc = a2_settle(m)) return rc;
  if (int rc = a2_store_reserve(m, A.size + (uint32_t)m->n)) return rc;
  hipLaunchKernelGGL(mpm2d::k2a_file, dim3(a2_grid((uint32_t)m->n)), dim3(256), 0, m->stream, m->P.idx, (uint32_t)m->n, (const float *)m->x,
                     (const float *)m->v, (const float *)m->F, (const float *)m->B, (const float *)m->aux, (const int32_t *)m->gid,
                     (const int32_t *)m->pid, (const uint8_t *)A.d_tbl, 1, A.nb[0], A.nb[1], A.cap, A.rec, A.tag, A.d_cnt);
  HIPCHK2D(m, hipGetLastError());
  AsyncCounters h;
  if (int rc = a2_counters(m, h)) return rc;
  m->n = 0;  // (creation ids keep counting: next_pid stays)
  HIPCHK2D(m, hipMemsetAsync(m->n_dead, 0, sizeof(unsigned int), m->stream));
  return MPMHIP_OK;
}

// AsyncMPM<2>::update_dt_limits (src/async/async_mpm.cpp:90-253) over the resident pools
static int a2_update_dt_limits(mpmhip2d_ctx *m) {
  auto &A = m->async;
  const size_t nblk = A.nblk();
  hipLaunchKernelGGL(mpm::k_async_table_reset, dim3(a2_grid((uint32_t)nblk)), dim3(256), 0, m->stream, (uint32_t)nblk, A.d_tab);
  hipLaunchKernelGGL(mpm2d::k2a_store_reduce, dim3(a2_grid(A.size_ub)), dim3(256), 0, m->stream, m->P.dx, A.size_ub, (const uint32_t *)A.tag,
                     (const float4 *)A.rec, (const GroupParams *)m->d_groups, A.d_tab);
  HIPCHK2D(m, hipGetLastError());
  HIPCHK2D(m, hipMemcpyAsync(A.h_tab, A.d_tab, sizeof(uint32_t) * 3 * nblk, hipMemcpyDeviceToHost, m->stream));
  HIPCHK2D(m, hipStreamSynchronize(m->stream));
  A.scratch = A.continuous;
  if (!A.limits_from_table(A.h_tab, m->P.dx)) return fail2d(m, MPMHIP_EINVAL, A.sched_err);
  if (A.scratch != A.continuous) A.limits_version++;
  A.rebuild_lists();
  return MPMHIP_OK;
}

static int substep2d(mpmhip2d_ctx *m);
// AsyncMPM<2>::advance (src/async/async_mpm.cpp:255-373)
static int a2_advance(mpmhip2d_ctx *m, int64_t limit) {
  auto &A = m->async;
  const int64_t t = A.current_t_int;
  if (!A.plan_gather(limit)) return fail2d(m, MPMHIP_EINVAL, A.sched_err);
  if (int rc = a2_best_reserve(m)) return rc;
  if (int rc = a2_upload_tbl(m)) return rc;
  // the working set holds at most one container per id: min(containers, ids handed out) particles
  if (int rc = a2_grow_particles(m, std::min<int64_t>((int64_t)A.size_ub, (int64_t)m->next_pid))) return rc;
  hipLaunchKernelGGL(mpm2d::k2a_mark, dim3(a2_grid(A.size_ub)), dim3(256), 0, m->stream, A.size_ub, (const uint32_t *)A.tag,
                     (const float4 *)A.rec, (const uint8_t *)A.d_tbl, (const uint32_t *)A.d_rank, A.best);
  hipLaunchKernelGGL(mpm2d::k2a_gather, dim3(a2_grid(A.size_ub)), dim3(256), 0, m->stream, A.size_ub, A.tag, (const float4 *)A.rec,
                     (const uint8_t *)A.d_tbl, (const uint32_t *)A.d_rank, A.best, m->x, m->v, m->F, m->B, m->aux, m->gid, m->pid, A.d_cnt);
  HIPCHK2D(m, hipGetLastError());
  AsyncCounters h;
  if (int rc = a2_counters(m, h)) return rc;  // the one read-back of an advance (also settles the previous one's appends)
  const uint32_t n_work = h.n_work;
  A.update_counter += n_work;
  // ONE ordinary substep of the working set with this level's dt (:327-329; step() sets base_delta_t / current_t, :405-408)
  m->n = n_work;
  HIPCHK2D(m, hipMemsetAsync(m->n_dead, 0, sizeof(unsigned int), m->stream));
  m->P.dt = A.cfg.unit_delta_t * (float)limit;
  m->t = A.cfg.unit_delta_t * (float)t;
  if (n_work) {
    if (int rc = substep2d(m)) return rc;
  }
  const bool any_clear = A.plan_file(limit);  // update backup_t and particle_t (:331-343), destinations of the results
  if (int rc = a2_upload_tbl(m)) return rc;
  if (any_clear)
    hipLaunchKernelGGL(mpm::k_async_clear, dim3(a2_grid(A.size)), dim3(256), 0, m->stream, A.size, A.tag, (const uint8_t *)A.d_tbl, A.d_cnt);
  if (n_work) {
    const uint32_t dead = A.size - std::min(A.size, A.live);
    if (dead > A.live + 65536 || (A.size + n_work > A.cap && dead > A.size / 4)) {  // (see async_compact_if_needed, async_api.h)
      if (int rc = a2_compact(m)) return rc;
    }
    if (int rc = a2_store_reserve(m, A.size + n_work)) return rc;
    hipLaunchKernelGGL(mpm2d::k2a_file, dim3(a2_grid(n_work)), dim3(256), 0, m->stream, m->P.idx, n_work, (const float *)m->x,
                       (const float *)m->v, (const float *)m->F, (const float *)m->B, (const float *)m->aux, (const int32_t *)m->gid,
                       (const int32_t *)m->pid, (const uint8_t *)A.d_tbl, 0, A.nb[0], A.nb[1], A.cap, A.rec, A.tag, A.d_cnt);
    A.size_ub = A.size + n_work;  // (the exact size comes with the next read-back)
  }
  HIPCHK2D(m, hipGetLastError());
  A.pending_counters = true;
  return MPMHIP_OK;
}

// AsyncMPM<2>::step (src/async/async_mpm.cpp:380-421)
int mpmhip2d_async_step(mpmhip2d_ctx *m, float dt) {
  if (!m) return MPMHIP_EINVAL;
  auto &A = m->async;
  if (!A.resident) return fail2d(m, MPMHIP_EINVAL, "mpmhip2d_async_begin first");
  if (dt < 0) return fail2d(m, MPMHIP_EINVAL, "AsyncMPM::step(dt < 0) is the synchronous substep of the base class");
  HIPCHK2D(m, hipSetDevice(m->device));
  if (int rc = a2_drop_view(m)) return rc;
  if (m->n) {  // (particles added since the last step and not yet pooled)
    if (int rc = mpmhip2d_async_pool_particles(m)) return rc;
  }
  A.request_t += dt;
  do {
    if (int rc = a2_update_dt_limits(m)) return rc;
    for (int64_t d = A.max_delta_t_int; d >= A.min_delta_t_int; d >>= 1)
      if (A.current_t_int % d == 0) {
        if (int rc = a2_advance(m, d)) return rc;
      }
    A.finish_round();
  } while (A.current_t < A.request_t);
  if (int rc = a2_settle(m)) return rc;
  m->n = 0;  // the arrays held the last working set: the state is in the pools
  m->t = A.current_t;
  m->P.dt = m->base_dt;
  A.step_counter++;
  return MPMHIP_OK;
}

// {current_t_int, update_counter, min_delta_t_int, max_delta_t_int, live containers, store size, compactions, steps}
int mpmhip2d_async_state(mpmhip2d_ctx *m, int64_t out[8]) {
  if (!m || !out) return MPMHIP_EINVAL;
  auto &A = m->async;
  if (!A.resident) return fail2d(m, MPMHIP_EINVAL, "mpmhip2d_async_begin first");
  HIPCHK2D(m, hipSetDevice(m->device));
  if (int rc = a2_settle(m)) return rc;
  out[0] = A.current_t_int; out[1] = A.update_counter; out[2] = A.min_delta_t_int; out[3] = A.max_delta_t_int;
  out[4] = A.live; out[5] = A.size; out[6] = A.compactions; out[7] = A.step_counter;
  return MPMHIP_OK;
}
double mpmhip2d_async_current_time(const mpmhip2d_ctx *m) { return m ? (double)m->async.current_t : 0.0; }

// dense view of the block table: nb[2] blocks per axis (block b = bx nb[1] + by, 8 x 16 nodes each); any output may be NULL.
// capacity < number of blocks: only the number is returned.
int64_t mpmhip2d_async_table(mpmhip2d_ctx *m, int32_t nb[2], int64_t capacity, int64_t *strength, int64_t *cfl, int64_t *continuous,
                             int64_t *count, int64_t *particle_t, int64_t *backup_t, int64_t *local_min) {
  if (!m || !m->async.resident || !nb) return MPMHIP_EINVAL;
  auto &A = m->async;
  nb[0] = A.nb[0]; nb[1] = A.nb[1];
  const int64_t n = (int64_t)A.nblk();
  if (capacity < n) return n;
  for (int64_t b = 0; b < n; b++) {
    if (strength) strength[b] = A.strength[b];
    if (cfl) cfl[b] = A.cfl[b];
    if (continuous) continuous[b] = A.continuous[b];
    if (count) count[b] = A.count[b];
    if (particle_t) particle_t[b] = A.particle_t[b];
    if (backup_t) backup_t[b] = A.backup_t[b];
    if (local_min) local_min[b] = A.local_min[b];
  }
  return n;
}

// AsyncMPM<2>::visualize's particle list (src/async/async_visualize.cpp:86-96): ALL containers of all particle pools become
// the object's particles (each at its block's particle_t; an id can occur more than once, as in the reference), so that
// mpmhip2d_download / _num_particles see the whole state, not the last working set.  The view is dropped by the next step
// or add.  Returns the number of containers.
int64_t mpmhip2d_async_load_pools(mpmhip2d_ctx *m) {
  if (!m) return MPMHIP_EINVAL;
  auto &A = m->async;
  if (!A.resident) return fail2d(m, MPMHIP_EINVAL, "mpmhip2d_async_begin first");
  HIPCHK2D(m, hipSetDevice(m->device));
  if (int rc = mpmhip2d_async_pool_particles(m)) return rc;  // (drops an earlier view; pools particles added since)
  if (int rc = a2_settle(m)) return rc;
  if (int rc = a2_grow_particles(m, (int64_t)A.live)) return rc;
  if (A.blk_of_cap < m->cap) {
    (void)hipFree(A.d_blk_of); A.d_blk_of = nullptr;
    HIPCHK2D(m, dmalloc(&A.d_blk_of, (size_t)m->cap));
    A.blk_of_cap = m->cap;
  }
  hipLaunchKernelGGL(mpm2d::k2a_load, dim3(a2_grid(A.size)), dim3(256), 0, m->stream, A.size, (const uint32_t *)A.tag, (const float4 *)A.rec,
                     m->x, m->v, m->F, m->B, m->aux, m->gid, m->pid, A.d_blk_of, A.d_cnt);
  HIPCHK2D(m, hipGetLastError());
  AsyncCounters h;
  if (int rc = a2_counters(m, h)) return rc;
  m->n = h.n_work;
  A.view = true;
  return m->n;
}
// the pool block of every particle of the view, in the order mpmhip2d_download lists them
int64_t mpmhip2d_async_view_blocks(mpmhip2d_ctx *m, int64_t capacity, int32_t *block) {
  if (!m || !block) return MPMHIP_EINVAL;
  auto &A = m->async;
  if (!A.resident || !A.view) return fail2d(m, MPMHIP_EINVAL, "mpmhip2d_async_load_pools first");
  if (capacity < m->n) return fail2d(m, MPMHIP_ECAPACITY, "block buffer too small");
  HIPCHK2D(m, hipSetDevice(m->device));
  HIPCHK2D(m, hipStreamSynchronize(m->stream));
  if (m->n) HIPCHK2D(m, hipMemcpy(block, A.d_blk_of, sizeof(uint32_t) * (size_t)m->n, hipMemcpyDeviceToHost));
  return m->n;
}
