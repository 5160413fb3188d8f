// taichi_mpm_amd/csrc/async_api.h — host side of the device-resident asynchronous stepper (included by mpmhip.hip inside
// extern "C").  Part of libmpmhip.  Device side: k_async.h.
//
// AsyncMPM<dim> (src/async/async_mpm.{h,cpp}): per scheduler block a continuous_dt_limit (a power of two of unit_delta_t),
// a particle pool at the block's own time and a backup pool at an earlier time; step() walks the power-of-two levels and
// advance(limit) runs ONE ordinary substep, dt = unit_delta_t * limit, on the blocks of that level plus frozen copies of
// their neighbours.  Here the block tables (a few integers per block) and the level walk are host code, as in the reference;
// every container stays in HBM (k_async.h: the store), and an advance costs four small kernels around the substep and
// two 16-byte read-backs (how many particles the working set has; how many containers were appended / freed).

static int async_store_reserve(mpmhip_ctx *c, uint32_t need) {  // room for `need` containers in total
  auto &S = c->async.store;
  if (need <= S.cap) return MPMHIP_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const uint32_t cap = std::max<uint32_t>(need + need / 2, 4096), keep = std::min(S.size_ub, S.cap);
  hipError_t e = hipSuccess;
  auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  A(regrow(&S.g, (size_t)keep * 4, (size_t)cap * 4, false)); A(regrow(&S.w, (size_t)keep * 4, (size_t)cap * 4, false));
  A(regrow(&S.tag, (size_t)keep, (size_t)cap, false)); A(regrow(&S.id, (size_t)keep, (size_t)cap, false));
  // (kernels walk [0, upper bound of the size): every tag behind the containers in use says FREE)
  if (e == hipSuccess) A(hipMemset(S.tag + keep, 0xFF, sizeof(uint32_t) * (size_t)(cap - keep)));
  (void)hipFree(S.g2); (void)hipFree(S.w2); (void)hipFree(S.tag2); (void)hipFree(S.id2);
  S.g2 = S.w2 = nullptr; S.tag2 = nullptr; S.id2 = nullptr;  // (the compaction targets are re-allocated when a compaction runs)
  if (e != hipSuccess) return fail(c, MPMHIP_ENOMEM, "async store: growing to %u containers failed: %s", cap, hipGetErrorString(e));
  S.cap = cap;
  return MPMHIP_OK;
}
// the per-block action table of an advance -> S.d_tbl, ORDERED ON THE CTX STREAM behind the kernels that still read the
// previous table (the ctx stream is non-blocking: a plain hipMemcpy would run beside them) and from pinned memory (the host
// refills A.tbl right away): a ring of four pinned images — every advance synchronises the stream between its two uploads, so
// an image is never rewritten while its copy is still pending
static int async_upload_tbl(mpmhip_ctx *c) {
  auto &A = c->async;
  auto &S = A.store;
  const size_t nblk = A.tbl.size();
  if (S.pin_cap < nblk) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (S.h_tbl_pin) (void)hipHostFree(S.h_tbl_pin);
    S.h_tbl_pin = nullptr;
    HIPCHK(c, hipHostMalloc((void **)&S.h_tbl_pin, 4 * nblk, hipHostMallocDefault));
    S.pin_cap = nblk;
  }
  uint8_t *img = S.h_tbl_pin + (size_t)(S.pin_next++ & 3) * S.pin_cap;
  memcpy(img, A.tbl.data(), nblk);
  HIPCHK(c, hipMemcpyAsync(S.d_tbl, img, nblk, hipMemcpyHostToDevice, c->stream));
  return MPMHIP_OK;
}
static int async_best_reserve(mpmhip_ctx *c) {  // one dedup word per creation id
  auto &S = c->async.store;
  if ((int64_t)c->next_pid <= S.best_cap) return MPMHIP_OK;
  const size_t cap = (size_t)c->next_pid + (size_t)c->next_pid / 2 + 1024;
  (void)hipFree(S.best); S.best = nullptr;
  HIPCHK(c, dmalloc(&S.best, cap));
  HIPCHK(c, hipMemsetAsync(S.best, 0xFF, sizeof(unsigned long long) * cap, c->stream));
  S.best_cap = (int64_t)cap;
  return MPMHIP_OK;
}
// the ONE read-back of an advance: the transient counters (reset behind the copy) and the append cursor.  Folds what earlier
// launches appended / freed into the host's view of the store.
static int async_counters(mpmhip_ctx *c, AsyncCounters &h, bool reset_after) {
  auto &S = c->async.store;
  AsyncCounters *pin = reinterpret_cast<AsyncCounters *>(c->h_pinned + 64);  // (the ctx's pinned page; the substep counters sit at its start)
  HIPCHK(c, hipMemcpyAsync(pin, S.d_cnt, sizeof h, hipMemcpyDeviceToHost, c->stream));
  if (reset_after) HIPCHK(c, hipMemsetAsync(S.d_cnt, 0, 16, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  h = *pin;
  if (h.pad[0] & SCAN_ERROR_BIT)
    return fail(c, MPMHIP_EHIP, "async store: a chained scan of the compaction waited %.0f s for a chunk that never published (k_sort.h: "
                "the launch did not fit the device's resident set?)", (double)SCAN_WAIT_TICKS / 1e8);
  if (reset_after) {
    S.live += h.n_append; S.live -= std::min(S.live, h.n_freed);
    S.size = S.size_ub = h.size;
    c->async.pending_counters = false;
  }
  return MPMHIP_OK;
}
struct AsTimer {  // wall time of a host-side section, accumulated into c->async.prof_ms[k] (synchronising sections included)
  double &acc; std::chrono::steady_clock::time_point t0;
  explicit AsTimer(double &a) : acc(a), t0(std::chrono::steady_clock::now()) {}
  ~AsTimer() { acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
static int as_grid(uint32_t n) { return (int)std::min<uint32_t>(std::max<uint32_t>((n + 255) / 256, 1), 4096); }

static int async_compact(mpmhip_ctx *c);
// (called right behind a read-back: S.size and S.live are exact)
static int async_compact_if_needed(mpmhip_ctx *c, uint32_t incoming) {
  auto &S = c->async.store;
  const uint32_t dead = S.size - std::min(S.size, S.live);
  // squeeze when the freed containers outnumber the live ones (the passes over the tags then cost twice what they must), or
  // when that avoids growing the store
  if (!(dead > S.live + 65536 || (S.size + incoming > S.cap && dead > S.size / 4))) return MPMHIP_OK;
  return async_compact(c);
}
static int async_compact(mpmhip_ctx *c) {
  auto &S = c->async.store;
  if (S.size == 0) return MPMHIP_OK;
  AsTimer whole(c->async.prof_ms[4]);
  if (!S.g2) {
    hipError_t e = hipSuccess;
    auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    A(dmalloc(&S.g2, (size_t)S.cap * 4)); A(dmalloc(&S.w2, (size_t)S.cap * 4)); A(dmalloc(&S.tag2, (size_t)S.cap)); A(dmalloc(&S.id2, (size_t)S.cap));
    if (e != hipSuccess) return fail(c, MPMHIP_ENOMEM, "async store: compaction buffers: %s", hipGetErrorString(e));
  }
  HIPCHK(c, hipMemsetAsync(S.tag2, 0xFF, sizeof(uint32_t) * (size_t)S.cap, c->stream));
  const uint32_t nchunks = (S.size + 1023) / 1024;
  if (nchunks + 1 > S.scan_cap) {
    (void)hipFree(S.d_scan); S.d_scan = nullptr;
    HIPCHK(c, dmalloc(&S.d_scan, (size_t)nchunks + 1024));
    HIPCHK(c, hipMemsetAsync(S.d_scan, 0, sizeof(unsigned long long) * ((size_t)nchunks + 1024), c->stream));
    S.scan_cap = nchunks + 1024;
  }
  hipLaunchKernelGGL(k_async_compact, dim3(std::min<uint32_t>(nchunks, scan_limit(c, (const void *)k_async_compact))), dim3(256), 0, c->stream, S.size,
                     (const uint32_t *)S.tag, (const int32_t *)S.id, (const float4 *)S.g, (const float4 *)S.w, S.tag2, S.id2, S.g2, S.w2,
                     S.d_scan, ++S.scan_epoch, S.d_cnt);
  if (int rc = launch_check(c, "async_compact")) return rc;
  std::swap(S.g, S.g2); std::swap(S.w, S.w2); std::swap(S.tag, S.tag2); std::swap(S.id, S.id2);
  AsyncCounters h;
  if (int rc = async_counters(c, h, true)) return rc;  // (size = live = what the scan counted)
  S.live = h.size;
  S.compactions++;
  return MPMHIP_OK;
}

// the ctx's records hold a view of the pools (load_pools): they are copies, dropped before anything else uses the records
static int async_drop_view(mpmhip_ctx *c) {
  auto &A = c->async;
  if (!A.resident || !A.records_are_view) return MPMHIP_OK;
  A.records_are_view = false;
  c->n_slots = 0; c->P.n_slots = 0;
  const uint32_t zero = 0;
  HIPCHK(c, hipMemcpy(&c->cnt->n_dead, &zero, sizeof zero, hipMemcpyHostToDevice));
  c->affine_valid = false; c->b_stale = false;
  c->keys_valid = true;
  return invalidate_keys(c);
}

// AsyncMPM<dim>::initialize (src/async/async_mpm.cpp:13-55) on top of mpmhip_async_enable: the pools become device-resident
int mpmhip_async_begin(mpmhip_ctx *c, const mpmhip_async_config *cfg) {
  if (!c || !cfg) return MPMHIP_EINVAL;
  if (rigid_active(c) || c->rigid.enabled) return fail(c, MPMHIP_EINVAL, "asynchronous stepping cannot be combined with rigid bodies");
  if (c->T.enabled) return fail(c, MPMHIP_EINVAL, "asynchronous stepping cannot be combined with the multi-GPU tiling");
  if (!c->P.store_b) return fail(c, MPMHIP_EINVAL, "asynchronous stepping needs a ctx that keeps apic_b (discard_apic_b = 0): the "
                                                   "P2G matrix depends on each advance's dt and is rebuilt from it");
  if (int rc = mpmhip_async_enable(c, cfg)) return rc;
  auto &A = c->async;
  auto &S = A.store;
  A.sched_begin();  // block times, the reference's block order, cached_neighbours
  const size_t nblk = A.nblk();
  HIPCHK(c, hipSetDevice(c->device));
  hipFree(S.d_tbl); hipFree(S.d_rank); hipFree(S.d_cnt);
  S.d_tbl = nullptr; S.d_rank = nullptr; S.d_cnt = nullptr;
  HIPCHK(c, dmalloc(&S.d_tbl, nblk));
  HIPCHK(c, dmalloc(&S.d_rank, nblk));
  HIPCHK(c, dmalloc(&S.d_cnt, 1));
  HIPCHK(c, hipMemcpy(S.d_rank, A.rank_of.data(), sizeof(uint32_t) * nblk, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemset(S.d_cnt, 0, sizeof(AsyncCounters)));
  S.size = S.size_ub = S.live = 0;
  A.resident = true;
  return MPMHIP_OK;
}

static int async_settle(mpmhip_ctx *c);
// AsyncMPM<dim>::add_particles (src/async/async_mpm.cpp:57-75): the particles currently in the ctx's records (just added by
// mpmhip_add_particles) move to the particle pools of their blocks; the ctx's record arrays are empty afterwards.
int mpmhip_async_pool_particles(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  auto &A = c->async;
  auto &S = A.store;
  if (!A.resident) return fail(c, MPMHIP_EINVAL, "mpmhip_async_begin first");
  HIPCHK(c, hipSetDevice(c->device));
  if (int rc = async_drop_view(c)) return rc;
  if (c->n_slots == 0) return MPMHIP_OK;
  if (int rc = async_settle(c)) return rc;
  if (int rc = ensure_b_current(c)) return rc;
  if (int rc = async_store_reserve(c, S.size + (uint32_t)c->n_slots)) return rc;
  hipLaunchKernelGGL(k_async_file, dim3(particle_grid(c->n_slots)), dim3(256), 0, c->stream, c->P, (const float4 *)c->rg,
                     (const float4 *)c->rp, (const float4 *)c->rb, (const uint8_t *)S.d_tbl, 1, A.nb[0], A.nb[1], A.nb[2], S.cap,
                     S.g, S.w, S.tag, S.id, S.d_cnt);
  if (int rc = launch_check(c, "async_file")) return rc;
  AsyncCounters h;
  if (int rc = async_counters(c, h, true)) return rc;
  c->n_slots = 0; c->P.n_slots = 0;  // (creation ids keep counting: next_pid stays)
  const uint32_t zero = 0;
  HIPCHK(c, hipMemcpy(&c->cnt->n_dead, &zero, sizeof zero, hipMemcpyHostToDevice));
  c->affine_valid = false; c->b_stale = false;
  c->keys_valid = true;
  return invalidate_keys(c);
}

// AsyncMPM<dim>::update_dt_limits (src/async/async_mpm.cpp:90-253) over the resident pools
static int async_update_dt_limits(mpmhip_ctx *c) {
  auto &A = c->async;
  auto &S = A.store;
  const size_t nblk = A.continuous.size();
  AsTimer whole(A.prof_ms[0]);
  hipLaunchKernelGGL(k_async_table_reset, dim3(as_grid((uint32_t)nblk)), dim3(256), 0, c->stream, (uint32_t)nblk, A.d_tab);
  hipLaunchKernelGGL(k_async_store_reduce, dim3(as_grid(S.size_ub)), dim3(256), 0, c->stream, c->P, S.size_ub, (const uint32_t *)S.tag,
                     (const float4 *)S.g, (const float4 *)S.w, (const GroupParams *)c->d_groups, A.d_tab);
  if (int rc = launch_check(c, "async_store_reduce")) return rc;
  A.scratch = A.continuous;
  if (int rc = async_limits_from_table(c)) return rc;  // the block state machine shared with mpmhip_async_update_dt_limits
  if (A.scratch != A.continuous) A.limits_version++;
  AsTimer lists(A.prof_ms[1]);
  A.rebuild_lists();  // larger / smaller neighbours per level (:183-247), local_min_dt_limit (:165-182)
  return MPMHIP_OK;
}

// AsyncMPM<dim>::advance (src/async/async_mpm.cpp:255-373)
static int async_advance(mpmhip_ctx *c, int64_t limit) {
  auto &A = c->async;
  auto &S = A.store;
  const int64_t t = A.current_t_int;
  AsTimer whole(A.prof_ms[2]);
  A.prof_ms[5] += 1.0;  // (advances)
  if (!A.plan_gather(limit)) return fail(c, MPMHIP_EINVAL, "%s", A.sched_err.c_str());
  if (int rc = async_best_reserve(c)) return rc;
  if (int rc = async_upload_tbl(c)) return rc;
  // the working set holds at most one container per id (duplicates are dropped by the gather), i.e. at most
  // min(containers, ids handed out) records: the ctx's record arrays get that room BEFORE the gather writes into them — a
  // C-ABI caller may have pooled several batches of up to `cap` particles each (mpmhip_async_pool_particles empties the slots)
  {
    const int64_t bound = std::min<int64_t>((int64_t)S.size_ub, (int64_t)c->next_pid);
    if (bound > c->cap) {
      if (int rc = mpmhip_reserve(c, bound + 1024)) return rc;
    }
  }
  hipLaunchKernelGGL(k_async_mark, dim3(as_grid(S.size_ub)), dim3(256), 0, c->stream, S.size_ub, (const uint32_t *)S.tag,
                     (const int32_t *)S.id, (const uint8_t *)S.d_tbl, (const uint32_t *)S.d_rank, S.best);
  hipLaunchKernelGGL(k_async_gather, dim3(as_grid(S.size_ub)), dim3(256), 0, c->stream, S.size_ub, S.tag, (const int32_t *)S.id,
                     (const uint8_t *)S.d_tbl, (const uint32_t *)S.d_rank, S.best, (const float4 *)S.g, (const float4 *)S.w,
                     (const GroupParams *)c->d_groups, (float4 *)c->rg, (float4 *)c->rp, (float4 *)c->rb, S.d_cnt);
  if (int rc = launch_check(c, "async_gather")) return rc;
  AsyncCounters h;
  if (int rc = async_counters(c, h, true)) return rc;  // the one read-back of an advance (also settles the previous one's appends)
  const uint32_t n_work = h.n_work;
  if ((int64_t)n_work > c->cap) return fail(c, MPMHIP_ECAPACITY, "async: a working set of %u particles exceeds the ctx capacity %lld", n_work, (long long)c->cap);
  A.update_counter += n_work;
  // ONE ordinary substep of the working set with this level's dt (:327-329; step() sets base_delta_t / current_t, :405-408)
  c->n_slots = n_work; c->P.n_slots = n_work;
  HIPCHK(c, hipMemsetAsync(&c->cnt->n_dead, 0, sizeof(uint32_t), c->stream));
  c->P.dt = A.cfg.unit_delta_t * (float)limit;
  c->t = A.cfg.unit_delta_t * (float)t;
  c->affine_valid = false; c->b_stale = false;
  c->keys_valid = true;
  if (int rc = invalidate_keys(c)) return rc;
  if (n_work) {
    AsTimer sub(A.prof_ms[3]);
    if (int rc = mpmhip_substep(c)) return rc;
    if (A.profile_sync) HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  const bool any_clear = A.plan_file(limit);  // update backup_t and particle_t (:331-343), destinations of the results
  if (int rc = async_upload_tbl(c)) return rc;
  if (any_clear)
    hipLaunchKernelGGL(k_async_clear, dim3(as_grid(S.size)), dim3(256), 0, c->stream, S.size, S.tag, (const uint8_t *)S.d_tbl, S.d_cnt);
  if (n_work) {
    if (int rc = async_compact_if_needed(c, n_work)) return rc;
    if (int rc = async_store_reserve(c, S.size + n_work)) return rc;
    hipLaunchKernelGGL(k_async_file, dim3(particle_grid(n_work)), dim3(256), 0, c->stream, c->P, (const float4 *)c->rg,
                       (const float4 *)c->rp, (const float4 *)c->rb, (const uint8_t *)S.d_tbl, 0, A.nb[0], A.nb[1], A.nb[2], S.cap,
                       S.g, S.w, S.tag, S.id, S.d_cnt);
    S.size_ub = S.size + n_work;  // (the exact size comes with the next read-back)
  }
  if (int rc = launch_check(c, "async_file")) return rc;
  A.pending_counters = true;
  return MPMHIP_OK;
}

// counters a previous advance left on the device (appended / freed containers): folded into the host's view of the store
static int async_settle(mpmhip_ctx *c) {
  if (!c->async.pending_counters) return MPMHIP_OK;
  AsyncCounters h;
  return async_counters(c, h, true);
}

// AsyncMPM<dim>::step (src/async/async_mpm.cpp:380-421)
int mpmhip_async_step(mpmhip_ctx *c, float dt) {
  if (!c) return MPMHIP_EINVAL;
  auto &A = c->async;
  if (!A.resident) return fail(c, MPMHIP_EINVAL, "mpmhip_async_begin first");
  if (dt < 0) return fail(c, MPMHIP_EINVAL, "AsyncMPM::step(dt < 0) is the synchronous substep of the base class");
  HIPCHK(c, hipSetDevice(c->device));
  if (int rc = async_drop_view(c)) return rc;
  if (c->n_slots) {  // (particles added since the last step and not yet pooled)
    if (int rc = mpmhip_async_pool_particles(c)) return rc;
  }
  A.request_t += dt;
  do {
    if (int rc = async_update_dt_limits(c)) return rc;
    for (int64_t d = A.max_delta_t_int; d >= A.min_delta_t_int; d >>= 1)
      if (A.current_t_int % d == 0) {
        if (int rc = async_advance(c, d)) return rc;
      }
    A.finish_round();
  } while (A.current_t < A.request_t);
  if (int rc = async_settle(c)) return rc;
  c->n_slots = 0; c->P.n_slots = 0;  // the records held the last working set: the state is in the pools
  c->t = A.current_t;
  A.step_counter++;
  return MPMHIP_OK;
}

// {current_t_int, update_counter, pool containers, backup containers + freed (store size - pool), compactions, store size}
int mpmhip_async_state(mpmhip_ctx *c, int64_t out[8]) {
  if (!c || !out) return MPMHIP_EINVAL;
  auto &A = c->async;
  if (!A.resident) return fail(c, MPMHIP_EINVAL, "mpmhip_async_begin first");
  HIPCHK(c, hipSetDevice(c->device));
  if (int rc = async_settle(c)) return rc;
  out[0] = A.current_t_int; out[1] = A.update_counter; out[2] = A.min_delta_t_int; out[3] = A.max_delta_t_int;
  out[4] = A.store.live; out[5] = A.store.size; out[6] = A.store.compactions; out[7] = A.step_counter;
  return MPMHIP_OK;
}
// host wall time so far, ms: {update_dt_limits, of which neighbour lists, advance, of which the substep (only meaningful with
// sync != 0: the stream is then synchronised behind every substep), compaction, number of advances}
int mpmhip_async_profile(mpmhip_ctx *c, int32_t sync, double out[6]) {
  if (!c || !out) return MPMHIP_EINVAL;
  for (int k = 0; k < 6; k++) out[k] = c->async.prof_ms[k];
  c->async.profile_sync = sync != 0;
  return MPMHIP_OK;
}
double mpmhip_async_current_time(const mpmhip_ctx *c) { return c ? (double)c->async.current_t : 0.0; }
int64_t mpmhip_host_particle_bytes(const mpmhip_ctx *c) { return c ? c->host_particle_bytes : MPMHIP_EINVAL; }

// per block of the dense table (mpmhip_async_table gives the limits): particle_t, backup_t, local_min_dt_limit
int64_t mpmhip_async_block_times(mpmhip_ctx *c, int64_t capacity, int64_t *particle_t, int64_t *backup_t, int64_t *local_min) {
  if (!c || !c->async.resident) return MPMHIP_EINVAL;
  auto &A = c->async;
  const int64_t n = (int64_t)A.particle_t.size();
  if (capacity < n) return n;
  for (int64_t b = 0; b < n; b++) {
    if (particle_t) particle_t[b] = A.particle_t[b];
    if (backup_t) backup_t[b] = A.backup_t[b];
    if (local_min) local_min[b] = A.local_min[b];
  }
  return n;
}

// Every container of every particle pool (each at its block's particle_t; an id can occur more than once, as in the
// reference) as rows of 27 floats {x3, v3, F9, apic_b9, aux, gid bits, id bits} + the container's block.  rows == NULL:
// only the count.  (Tests and host-side inspection: the stepping itself never calls this.)
int64_t mpmhip_async_download_pools(mpmhip_ctx *c, int64_t capacity, float *rows, int32_t *block) {
  if (!c) return MPMHIP_EINVAL;
  auto &A = c->async;
  auto &S = A.store;
  if (!A.resident) return fail(c, MPMHIP_EINVAL, "mpmhip_async_begin first");
  HIPCHK(c, hipSetDevice(c->device));
  if (int rc = async_settle(c)) return rc;
  float *d_rows = nullptr;
  uint32_t *d_blk = nullptr;
  const size_t m = std::max<size_t>(S.size, 1);
  HIPCHK(c, dmalloc(&d_rows, m * 27));
  hipError_t e = dmalloc(&d_blk, m);
  if (e != hipSuccess) { (void)hipFree(d_rows); return fail(c, MPMHIP_ENOMEM, "async download: %s", hipGetErrorString(e)); }
  hipLaunchKernelGGL(k_async_export, dim3(as_grid(S.size)), dim3(256), 0, c->stream, S.size, (const uint32_t *)S.tag, (const float4 *)S.g,
                     (const float4 *)S.w, d_rows, d_blk, S.d_cnt);
  AsyncCounters h;
  int rc = launch_check(c, "async_export");
  if (!rc) rc = async_counters(c, h, true);
  int64_t n = rc ? rc : (int64_t)h.n_work;
  if (!rc && rows && block) {
    if (capacity < n) { rc = fail(c, MPMHIP_ECAPACITY, "async download: %lld containers, room for %lld", (long long)n, (long long)capacity); n = rc; }
    else if (n) {
      e = hipMemcpy(rows, d_rows, sizeof(float) * 27 * (size_t)n, hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(block, d_blk, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost);
      if (e != hipSuccess) n = fail(c, MPMHIP_EHIP, "async download: %s", hipGetErrorString(e));
      c->host_particle_bytes += (int64_t)n * 27 * 4;
    }
  }
  (void)hipFree(d_rows); (void)hipFree(d_blk);
  return n;
}

// AsyncMPM<dim>::visualize's particle list (src/async/async_visualize.cpp:17-26,86-96): ALL containers of all particle pools
// become the ctx's records (so that frame output, downloads, energy and snapshots of the base class see the whole state, not
// the last working set), each with its pool block's (continuous, strength, cfl) limits for the `limit` attribute.
int mpmhip_async_load_pools(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  auto &A = c->async;
  auto &S = A.store;
  if (!A.resident) return fail(c, MPMHIP_EINVAL, "mpmhip_async_begin first");
  HIPCHK(c, hipSetDevice(c->device));
  if (int rc = async_settle(c)) return rc;
  if (int rc = mpmhip_async_pool_particles(c)) return rc;  // (drops an earlier view; pools particles added since)
  if ((int64_t)S.live > c->cap) {
    if (int rc = mpmhip_reserve(c, (int64_t)S.live + 1024)) return rc;
  }
  const size_t nblk = A.continuous.size();
  if (int rc = async_ensure_particle_arrays(c)) return rc;
  std::fill(A.tbl.begin(), A.tbl.end(), (uint8_t)AT_POOL1);
  if (int rc = async_upload_tbl(c)) return rc;
  hipLaunchKernelGGL(k_async_load, dim3(as_grid(S.size)), dim3(256), 0, c->stream, S.size, (const uint32_t *)S.tag, (const float4 *)S.g,
                     (const float4 *)S.w, (const GroupParams *)c->d_groups, (float4 *)c->rg, (float4 *)c->rp, (float4 *)c->rb,
                     A.d_blk_of, S.d_cnt);
  if (int rc = launch_check(c, "async_load")) return rc;
  AsyncCounters h;
  if (int rc = async_counters(c, h, true)) return rc;
  c->n_slots = h.n_work; c->P.n_slots = h.n_work;
  A.records_are_view = true;
  const uint32_t zero = 0;
  HIPCHK(c, hipMemcpy(&c->cnt->n_dead, &zero, sizeof zero, hipMemcpyHostToDevice));
  c->P.dt = c->cfg.dt;  // (the P2G matrices are rebuilt for the configured base step if anybody runs a synchronous substep)
  c->affine_valid = false; c->b_stale = false;
  c->keys_valid = true;
  if (int rc = invalidate_keys(c)) return rc;
  // the frame's `limit` attribute: the limits of the container's POOL block
  if (c->n_slots) {
    std::vector<int32_t> lim(3 * nblk);
    for (size_t b = 0; b < nblk; b++) { lim[3 * b] = (int32_t)A.continuous[b]; lim[3 * b + 1] = (int32_t)A.strength[b]; lim[3 * b + 2] = (int32_t)A.cfl[b]; }
    HIPCHK(c, hipMemcpy(A.d_blk_limits, lim.data(), sizeof(int32_t) * 3 * nblk, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_async_particle_limits, dim3(particle_grid(c->n_slots)), dim3(256), 0, c->stream, c->P,
                       (const uint32_t *)A.d_blk_of, (const int32_t *)A.d_blk_limits, A.d_particle_limits);
    if (int rc = launch_check(c, "async_particle_limits")) return rc;
    A.limits_valid = true;
  }
  return MPMHIP_OK;
}

// ---- snapshots of the asynchronous stepper (the reference serialises every pool and the block table: TC_IO of
// particle_pool_tmp / backup_pool_tmp / blocks, src/async/async_mpm.h:120-172).  Blob: header, group table, the per-block
// integers, the store's live containers (compacted).  Loaded into a ctx of the same grid on which mpmhip_async_begin has run
// with the same unit_delta_t; level set and configuration come from the scene, as for the synchronous snapshots.
struct SnapAsync {
  char magic[8];  // "MPMASYNC"
  uint32_t abi, n_groups;
  int32_t res[3], nb[3];
  float dx, unit_delta_t;
  int64_t nblk, containers, current_t_int, min_delta_t_int, max_delta_t_int, update_counter, step_counter;
  int32_t next_pid, pad;
  float request_t, current_t;
};
static size_t async_snapshot_bytes(const mpmhip_ctx *c, size_t containers) {
  return sizeof(SnapAsync) + sizeof(GroupParams) * c->groups.size() + sizeof(int64_t) * 6 * c->async.continuous.size() +
         containers * (sizeof(uint32_t) + sizeof(int32_t) + 2 * 4 * sizeof(float4));
}
static int async_force_compaction(mpmhip_ctx *c) {  // the store holds exactly its live containers afterwards
  if (int rc = async_settle(c)) return rc;
  return async_compact(c);
}
int64_t mpmhip_async_snapshot_size(mpmhip_ctx *c) {
  if (!c || !c->async.resident) return MPMHIP_EINVAL;
  if (hipSetDevice(c->device) != hipSuccess) return MPMHIP_EHIP;
  if (int rc = mpmhip_async_pool_particles(c)) return rc;
  if (int rc = async_force_compaction(c)) return rc;
  return (int64_t)async_snapshot_bytes(c, c->async.store.size);
}
int mpmhip_async_snapshot_save(mpmhip_ctx *c, void *dst, size_t cap) {
  if (!c || !dst) return MPMHIP_EINVAL;
  auto &A = c->async;
  auto &S = A.store;
  if (!A.resident) return fail(c, MPMHIP_EINVAL, "mpmhip_async_begin first");
  const int64_t need = mpmhip_async_snapshot_size(c);
  if (need < 0) return (int)need;
  if (cap < (size_t)need) return fail(c, MPMHIP_ECAPACITY, "snapshot buffer too small: %zu < %lld", cap, (long long)need);
  SnapAsync h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, "MPMASYNC", 8);
  h.abi = MPMHIP_ABI_VERSION; h.n_groups = (uint32_t)c->groups.size();
  for (int k = 0; k < 3; k++) { h.res[k] = c->P.res[k]; h.nb[k] = A.nb[k]; }
  h.dx = c->P.dx; h.unit_delta_t = A.cfg.unit_delta_t;
  h.nblk = (int64_t)A.continuous.size(); h.containers = S.size;
  h.current_t_int = A.current_t_int; h.min_delta_t_int = A.min_delta_t_int; h.max_delta_t_int = A.max_delta_t_int;
  h.update_counter = A.update_counter; h.step_counter = A.step_counter; h.next_pid = c->next_pid;
  h.request_t = A.request_t; h.current_t = A.current_t;
  char *p = (char *)dst;
  memcpy(p, &h, sizeof h); p += sizeof h;
  memcpy(p, c->groups.data(), sizeof(GroupParams) * c->groups.size()); p += sizeof(GroupParams) * c->groups.size();
  const size_t nb = sizeof(int64_t) * (size_t)h.nblk;
  for (const std::vector<int64_t> *v : {&A.continuous, &A.strength, &A.cfl, &A.particle_t, &A.backup_t, &A.local_min}) { memcpy(p, v->data(), nb); p += nb; }
  const size_t n = S.size;
  if (n) {
    HIPCHK(c, hipMemcpy(p, S.tag, sizeof(uint32_t) * n, hipMemcpyDeviceToHost)); p += sizeof(uint32_t) * n;
    HIPCHK(c, hipMemcpy(p, S.id, sizeof(int32_t) * n, hipMemcpyDeviceToHost)); p += sizeof(int32_t) * n;
    HIPCHK(c, hipMemcpy(p, S.g, sizeof(float4) * 4 * n, hipMemcpyDeviceToHost)); p += sizeof(float4) * 4 * n;
    HIPCHK(c, hipMemcpy(p, S.w, sizeof(float4) * 4 * n, hipMemcpyDeviceToHost));
    c->host_particle_bytes += (int64_t)n * 136;
  }
  return MPMHIP_OK;
}
int mpmhip_async_snapshot_load(mpmhip_ctx *c, const void *src, size_t size) {
  if (!c || !src || size < sizeof(SnapAsync)) return MPMHIP_EINVAL;
  auto &A = c->async;
  auto &S = A.store;
  if (!A.resident) return fail(c, MPMHIP_EINVAL, "mpmhip_async_begin first");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  SnapAsync h;
  memcpy(&h, src, sizeof h);
  if (memcmp(h.magic, "MPMASYNC", 8) != 0 || h.abi != MPMHIP_ABI_VERSION) return fail(c, MPMHIP_EINVAL, "not an asynchronous-stepper snapshot of this ABI version");
  for (int k = 0; k < 3; k++)
    if (h.res[k] != c->P.res[k] || h.nb[k] != A.nb[k]) return fail(c, MPMHIP_EINVAL, "snapshot is of a %dx%dx%d grid", h.res[0], h.res[1], h.res[2]);
  if (h.dx != c->P.dx || h.unit_delta_t != A.cfg.unit_delta_t) return fail(c, MPMHIP_EINVAL, "snapshot has delta_x = %g, unit_delta_t = %g", h.dx, h.unit_delta_t);
  if ((int)h.n_groups > c->groups_cap || h.nblk != (int64_t)A.continuous.size() || h.containers < 0) return fail(c, MPMHIP_EINVAL, "snapshot header inconsistent with this ctx");
  if (h.n_groups < 0) return fail(c, MPMHIP_EINVAL, "snapshot header inconsistent with this ctx");
  if ((uint64_t)h.containers > size / 136) return fail(c, MPMHIP_EINVAL, "snapshot size mismatch");  // (bounded before it is multiplied)
  if (size != sizeof(SnapAsync) + sizeof(GroupParams) * (size_t)h.n_groups + sizeof(int64_t) * 6 * (size_t)h.nblk +
                  (size_t)h.containers * (sizeof(uint32_t) + sizeof(int32_t) + 2 * 4 * sizeof(float4)))
    return fail(c, MPMHIP_EINVAL, "snapshot size mismatch");
  {  // the container arrays index the block table (tag), the block order and the per-id bids (id) on the device: a truncated
     // or foreign blob must not get that far
    const size_t n = (size_t)h.containers;
    const char *q = (const char *)src + sizeof h + sizeof(GroupParams) * h.n_groups + 6 * sizeof(int64_t) * (size_t)h.nblk;
    const uint32_t *tags = reinterpret_cast<const uint32_t *>(q);
    const int32_t *ids = reinterpret_cast<const int32_t *>(q + sizeof(uint32_t) * n);
    if (h.next_pid < 0) return fail(c, MPMHIP_EINVAL, "snapshot header inconsistent with this ctx (next id %d)", h.next_pid);
    // the scheduler's limits (first of the six block tables): powers of two in [1, 2^31], or update_dt_limits cannot use them
    if (!AsyncSched::limits_are_sane((const char *)src + sizeof h + sizeof(GroupParams) * h.n_groups, (size_t)h.nblk))
      return fail(c, MPMHIP_EINVAL, "snapshot block table holds a time-step limit that is not a power of two in [1, 2^31]");
    const char *recg = q + (sizeof(uint32_t) + sizeof(int32_t)) * n;  // RecG per container: the group id at byte 52
    for (size_t i = 0; i < n; i++) {
      uint32_t tg; int32_t id;
      memcpy(&tg, tags + i, 4); memcpy(&id, ids + i, 4);
      if (tg == AS_FREE) continue;
      uint32_t gid;
      memcpy(&gid, recg + 64 * i + 52, 4);
      if (gid >= (uint32_t)h.n_groups) return fail(c, MPMHIP_EINVAL, "snapshot container %zu names group %u of %d", i, gid, h.n_groups);
      if ((int64_t)(tg & ~AS_BACKUP) >= h.nblk) return fail(c, MPMHIP_EINVAL, "snapshot container %zu names block %u of %lld", i, tg & ~AS_BACKUP, (long long)h.nblk);
      if (id < 0 || id >= h.next_pid) return fail(c, MPMHIP_EINVAL, "snapshot container %zu has id %d outside [0, %d)", i, id, h.next_pid);
    }
  }
  const char *p = (const char *)src + sizeof h;
  {  // (nothing of the ctx is touched before the whole blob has been checked)
    std::vector<GroupParams> gs((size_t)h.n_groups);
    memcpy(gs.data(), p, sizeof(GroupParams) * h.n_groups); p += sizeof(GroupParams) * h.n_groups;
    for (const GroupParams &g : gs)
      if (g.type < MPMHIP_VISCO || g.type > MPMHIP_ELASTIC) return fail(c, MPMHIP_EINVAL, "snapshot holds an unknown material id %d", g.type);
    c->groups.swap(gs);
  }
  if (h.n_groups) HIPCHK(c, hipMemcpy(c->d_groups, c->groups.data(), sizeof(GroupParams) * h.n_groups, hipMemcpyHostToDevice));
  const size_t nb = sizeof(int64_t) * (size_t)h.nblk;
  for (std::vector<int64_t> *v : {&A.continuous, &A.strength, &A.cfl, &A.particle_t, &A.backup_t, &A.local_min}) { memcpy(v->data(), p, nb); p += nb; }
  const size_t n = (size_t)h.containers;
  S.size = S.size_ub = S.live = 0;
  if (int rc = async_store_reserve(c, (uint32_t)n + 1024)) return rc;
  HIPCHK(c, hipMemset(S.tag, 0xFF, sizeof(uint32_t) * (size_t)S.cap));
  if (n) {
    HIPCHK(c, hipMemcpy(S.tag, p, sizeof(uint32_t) * n, hipMemcpyHostToDevice)); p += sizeof(uint32_t) * n;
    HIPCHK(c, hipMemcpy(S.id, p, sizeof(int32_t) * n, hipMemcpyHostToDevice)); p += sizeof(int32_t) * n;
    HIPCHK(c, hipMemcpy(S.g, p, sizeof(float4) * 4 * n, hipMemcpyHostToDevice)); p += sizeof(float4) * 4 * n;
    HIPCHK(c, hipMemcpy(S.w, p, sizeof(float4) * 4 * n, hipMemcpyHostToDevice));
    c->host_particle_bytes += (int64_t)n * 136;
  }
  S.size = S.size_ub = S.live = (uint32_t)n;
  AsyncCounters z;
  memset(&z, 0, sizeof z);
  z.size = (uint32_t)n;
  HIPCHK(c, hipMemcpy(S.d_cnt, &z, sizeof z, hipMemcpyHostToDevice));
  A.pending_counters = false; A.records_are_view = false;
  A.current_t_int = h.current_t_int; A.min_delta_t_int = h.min_delta_t_int; A.max_delta_t_int = h.max_delta_t_int;
  A.update_counter = h.update_counter; A.step_counter = h.step_counter;
  A.request_t = h.request_t; A.current_t = h.current_t;
  A.limits_version++;  // (the neighbour lists are rebuilt from the loaded limits at the next update)
  c->next_pid = h.next_pid;
  c->n_slots = 0; c->P.n_slots = 0;
  c->t = A.current_t;
  c->affine_valid = false; c->b_stale = false;
  c->keys_valid = true;
  return invalidate_keys(c);
}
