// taichi_mpm_amd/csrc/tiled_api.h — the native data plane of a tiled (multi-GPU) run: plan, buffers, wires, migration
// and the per-substep loop (include/mpmhip.h: "Multi-GPU tiling, the native data plane").  Host code of libmpmhip,
// included by mpmhip.hip.  The reference has no multi-process code (dead `#ifdef TC_USE_MPI`, src/mpm.cpp:6-8): the contract
// is SURVEY section 8(e) — bricks, halo sum after P2G, particle migration after G2P, a few scalars per migration.
//
//   plan      box(R, S) = node_box(R) ∩ node_box(S), node_box(R) = [brick.lo - margin, brick.hi + margin + 2) clipped to the
//             occupied part of the grid AND to rank R's own occupancy box (the nodes R's particles can touch before the next
//             migration check: every rank's particle bounds travel in the migration table, so all ranks hold all boxes) — R has
//             no mass outside it to send and nothing outside it to gather, so clusters that do not meet exchange nothing
//             (configs[4]: 8 - 39 MB per rank and substep of empty slabs around the cuts before round 6); sorted by peer.
//             Every rank can compute every rank's plan, so a writer knows where a box lives in its reader's buffers.
//   arena     ONE device allocation per rank: flag words, two migration tables, two receive buffers (parity of the substep),
//             the migration inbox.  Sized for the worst case (clip = the whole grid), so it is allocated — and, for the IPC
//             wire, mapped by the peers — exactly once.
//   epochs    substep e: the writer stores into recv[e & 1] of the reader and then publishes e in the reader's word
//             flag[writer]; the reader waits for flag >= e before reading.  Two buffers suffice: a writer reaches substep
//             e + 2 only after it has seen the reader's epoch e + 1, which the reader publishes after its reads of substep e.
//             Migrations carry their own epoch m (tables alternate by m & 1; records wait for the whole table of m, which
//             every rank contributes to only after its import of m - 1).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: librccl is dlopen'ed, libmpmhip does not link it

namespace {

struct RcclApi {
  void *handle = nullptr;
  std::string error;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

RcclApi *rccl_api() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return &api;
  tried = true;
  const char *names[] = {getenv("MPMHIP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names) {
    if (!n || !*n) continue;
    api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (api.handle) break;
    api.error = dlerror();
  }
  if (!api.handle) return &api;
  bool ok = true;
  auto sym = [&](const char *name) { void *p = dlsym(api.handle, name); if (!p) { ok = false; api.error = std::string("missing symbol ") + name; } return p; };
  api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
  api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
  api.CommAbort = (decltype(api.CommAbort))sym("ncclCommAbort");
  api.Send = (decltype(api.Send))sym("ncclSend");
  api.Recv = (decltype(api.Recv))sym("ncclRecv");
  api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
  api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
  api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
  api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
  api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
  if (!ok) { dlclose(api.handle); api.handle = nullptr; }
  return &api;
}

#define NCCLCHK(c, call)                                                                                   \
  do {                                                                                                     \
    ncclResult_t r_ = (call);                                                                              \
    if (r_ != ncclSuccess) return fail((c), MPMHIP_EHIP, "%s failed: %s", #call, rccl_api()->GetErrorString(r_)); \
  } while (0)

constexpr int TN_MAX_WORLD = MPMHIP_MAX_HALO_BOXES;      // every other rank can be a halo peer
constexpr uint32_t TN_ROW = TN_MAX_WORLD + 8;            // words of a migration row: [counts(world) | lo3 | hi3 | speed | inbox capacity], padded
constexpr size_t TN_FLAG_BYTES = 4096;                   // halo epochs [world] | table epochs [world] | record epochs [world] | reduction epochs [world]
constexpr int TN_RED_N = MPMHIP_REDUCE_MAX_VALUES;       // doubles of one rank's row of a reduction (mpmhip_tiled_reduce)
constexpr size_t TN_RED_BYTES = 2 * sizeof(double) * TN_RED_N * TN_MAX_WORLD;  // two tables of [world] rows (parity of the reduction)
constexpr size_t TN_RED_PIN_WORD = 8192;                 // the staging of a reduction in the ctx's pinned page (behind the migration table)
// a rank's arena as a process addresses it (its own allocation, or a peer's mapped one): every rank lays its arena out alike
void tn_carve(mpmhip_ctx::TiledNative::Peer &P, char *base, size_t table_bytes, size_t recv_bytes) {
  P.flags = reinterpret_cast<uint32_t *>(base);
  char *p = base + TN_FLAG_BYTES;
  P.red[0] = reinterpret_cast<double *>(p);
  P.red[1] = P.red[0] + (size_t)TN_RED_N * TN_MAX_WORLD;
  p += TN_RED_BYTES;
  P.table[0] = reinterpret_cast<uint32_t *>(p);
  P.table[1] = reinterpret_cast<uint32_t *>(p + table_bytes);
  p += 2 * table_bytes;
  P.recv[0] = reinterpret_cast<float4 *>(p);
  P.recv[1] = reinterpret_cast<float4 *>(p + recv_bytes);
  P.inbox = reinterpret_cast<float4 *>(p + 2 * recv_bytes);
}

int tn_wait(mpmhip_ctx *c, const uint32_t *words, const std::vector<int> &who, int *d_idx, uint32_t epoch) {
  if (who.empty()) return MPMHIP_OK;
  hipLaunchKernelGGL(k_epoch_wait, dim3(1), dim3(64), 0, c->stream, words, (const int *)d_idx, (int)who.size(), epoch,
                     c->tn.timeout_ticks, c->cnt);
  return launch_check(c, "epoch_wait");
}

// node box of `rank` under the partition (cuts of c->T) and a clip box
// (occ: [world][6] = lo3, hi3 of every rank's occupancy box in nodes, or nullptr: only the global clip box)
void tn_node_box(const Tiling &T, const int clip_lo[3], const int clip_hi[3], const int *occ, int rank, int lo[3], int hi[3]) {
  const int pc[3] = {rank / (T.dims[1] * T.dims[2]), (rank / T.dims[2]) % T.dims[1], rank % T.dims[2]};
  for (int a = 0; a < 3; a++) {
    lo[a] = std::max(clip_lo[a], T.cuts[a][pc[a]] - T.margin);
    hi[a] = std::min(clip_hi[a], T.cuts[a][pc[a] + 1] + T.margin + 2);
    if (occ) { lo[a] = std::max(lo[a], occ[rank * 6 + a]); hi[a] = std::min(hi[a], occ[rank * 6 + 3 + a]); }
  }
}
// halo boxes of `rank`, sorted by peer, with their offsets (float4 nodes) in the rank's buffers
std::vector<mpmhip_ctx::TiledNative::Box> tn_plan(const Tiling &T, int world, const int clip_lo[3], const int clip_hi[3], const int *occ,
                                                   int rank, uint64_t *total) {
  std::vector<mpmhip_ctx::TiledNative::Box> out;
  int alo[3], ahi[3];
  tn_node_box(T, clip_lo, clip_hi, occ, rank, alo, ahi);
  uint64_t off = 0;
  for (int s = 0; s < world; s++) {
    if (s == rank) continue;
    int blo[3], bhi[3];
    tn_node_box(T, clip_lo, clip_hi, occ, s, blo, bhi);
    mpmhip_ctx::TiledNative::Box b;
    bool any = true;
    for (int a = 0; a < 3; a++) {
      b.lo[a] = std::max(alo[a], blo[a]);
      b.hi[a] = std::min(ahi[a], bhi[a]);
      any = any && b.lo[a] < b.hi[a];
    }
    if (!any) continue;
    b.peer = s;
    b.vol = (uint64_t)(b.hi[0] - b.lo[0]) * (b.hi[1] - b.lo[1]) * (b.hi[2] - b.lo[2]);
    b.off = off;
    b.peer_off = 0;
    off += b.vol;
    out.push_back(b);
  }
  if (total) *total = off;
  return out;
}

// slack of a freshly cut clip box around the particles: the room the look-ahead test of tn_mig_c asks for (2 margin + 3) and as
// much again, so that a re-plan is not followed by the next one a migration later
int tn_clip_slack(const Tiling &T) { return std::max(8, 4 * T.margin + 4); }

// (re)build this rank's plan for the current clip box and upload the device box tables; no allocation
int tn_apply_plan(mpmhip_ctx *c) {
  auto &N = c->tn;
  Tiling &T = c->T;
  uint64_t total = 0;
  const int *occ = N.occ_valid ? N.occ.data() : nullptr;
  N.boxes = tn_plan(T, N.world, N.clip_lo, N.clip_hi, occ, T.rank, &total);
  if (N.boxes.size() > MPMHIP_MAX_HALO_BOXES) return fail(c, MPMHIP_EINVAL, "too many halo boxes (%zu)", N.boxes.size());
  if (total > N.halo_cap) return fail(c, MPMHIP_ECAPACITY, "internal: halo plan of %llu nodes exceeds the arena (%llu)", (unsigned long long)total, (unsigned long long)N.halo_cap);
  N.total = total;
  for (auto &b : N.boxes) {  // where the box lives in the peer's buffers: the peer's own plan
    uint64_t ptotal = 0;
    auto pp = tn_plan(T, N.world, N.clip_lo, N.clip_hi, occ, b.peer, &ptotal);
    bool found = false;
    for (auto &q : pp)
      if (q.peer == T.rank) { b.peer_off = q.off; found = true; }
    if (!found) return fail(c, MPMHIP_EINVAL, "internal: halo box towards rank %d has no counterpart", b.peer);
  }
  // the overlap-free interior of this rank's node box and the device tables (as mpmhip_set_halo)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  T.n_boxes = 0; T.box_nodes = 0; T.box_blocks = 0;
  // (the node box as the boxes were cut from it — clip box and occupancy included: a box that spans the CUT node box along an axis
  // leaves the interior alone on that axis.  Until round 6 the uncut brick box stood here: a face box of a rank whose particles fill
  // only part of its brick then "ended inside" the two tangential axes and took the whole occupied part out of the interior —
  // at configs[3] over 8 bricks every block was boundary work and the overlap split had nothing to overlap the exchange with.
  // Work outside the cut node box does not exist: the rank has no particle there.)
  int nlo[3], nhi[3];
  tn_node_box(T, N.clip_lo, N.clip_hi, occ, T.rank, nlo, nhi);
  for (int a = 0; a < 3; a++) {
    if (nhi[a] < nlo[a]) nhi[a] = nlo[a];
    T.int_lo[a] = nlo[a]; T.int_hi[a] = nhi[a];
  }
  const size_t n = N.boxes.size();
  uint32_t boff = 0;
  bool empty_interior = false;
  // (halo boxes by peer writes: the IPC and the local wire — unless the local job carries them by RCCL self-sends, N.loop_rccl)
  const bool peer_wire = (N.wire == MPMHIP_WIRE_IPC || N.wire == MPMHIP_WIRE_LOCAL) && !N.loop_rccl;
  std::vector<DevBox> hb[2] = {std::vector<DevBox>(n), std::vector<DevBox>(n)};
  std::vector<int> idx(n);
  for (size_t i = 0; i < n; i++) {
    const auto &b = N.boxes[i];
    bool proper = false;
    for (int a = 0; a < 3; a++) {
      if (b.lo[a] <= nlo[a] && b.hi[a] >= nhi[a]) continue;
      proper = true;
      if (b.lo[a] <= nlo[a]) T.int_lo[a] = std::max(T.int_lo[a], b.hi[a]);
      else if (b.hi[a] >= nhi[a]) T.int_hi[a] = std::min(T.int_hi[a], b.lo[a]);
      else empty_interior = true;
    }
    if (!proper) empty_interior = true;
    for (int par = 0; par < 2; par++) {
      DevBox &d = hb[par][i];
      for (int a = 0; a < 3; a++) { d.lo[a] = b.lo[a]; d.dim[a] = b.hi[a] - b.lo[a]; }
      d.peer = b.peer; d.off = (uint32_t)b.off; d.boff = boff;
      if (peer_wire) {
        const auto &P = N.peers[b.peer];
        d.send = P.recv[par] ? P.recv[par] + b.peer_off : nullptr;  // (nullptr until the peers are connected)
        d.recv = N.recv[par] + b.off;
        d.flag = P.flags ? P.flags + T.rank : nullptr;
      } else {
        d.send = N.send + b.off;
        d.recv = N.recv[0] + b.off;
        d.flag = nullptr;
      }
    }
    idx[i] = b.peer;
    boff += box_blocks_of(hb[0][i].lo, hb[0][i].dim);
  }
  if (empty_interior) for (int a = 0; a < 3; a++) T.int_hi[a] = T.int_lo[a];
  if (n) {
    for (int par = 0; par < 2; par++) HIPCHK(c, hipMemcpy(N.d_boxes[par], hb[par].data(), sizeof(DevBox) * n, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(N.d_halo_idx, idx.data(), sizeof(int) * n, hipMemcpyHostToDevice));
  }
  N.halo_peers = idx;
  T.n_boxes = (int)n; T.box_nodes = (uint32_t)total; T.box_blocks = boff;
  c->d_boxes_cur = N.d_boxes[peer_wire ? (N.epoch & 1) : 0];
  return MPMHIP_OK;
}

bool tn_connected(const mpmhip_ctx *c) {
  const auto &N = c->tn;
  if (N.wire == MPMHIP_WIRE_RCCL) return N.comm != nullptr;
  return N.connected;
}

// ------------------------------------------------------------------------------------------------ the halo exchange
// begin: which box table the pack / grid kernels of this substep use (parity of the epoch) — called by mpmhip_substep_begin
void tn_begin_substep(mpmhip_ctx *c) {
  auto &N = c->tn;
  N.epoch++;
  const bool peer_wire = (N.wire == MPMHIP_WIRE_IPC || N.wire == MPMHIP_WIRE_LOCAL) && !N.loop_rccl;
  c->d_boxes_cur = N.d_boxes[peer_wire ? (N.epoch & 1) : 0];
}

int tn_exchange_start(mpmhip_ctx *c) {
  auto &N = c->tn;
  if (N.boxes.empty()) return MPMHIP_OK;
  if (N.wire != MPMHIP_WIRE_RCCL && !N.loop_rccl) {  // peer wires: k_halo_pack wrote the boxes into the peers' buffers; publish the epoch
    N.wait_merged = N.wire == MPMHIP_WIRE_IPC && !c->ov_active && N.merge_signal_wait;
    if (N.wait_merged) {  // (nothing runs between signal and wait: one launch; see k_epoch_signal_wait)
      hipLaunchKernelGGL(k_epoch_signal_wait, dim3(1), dim3(64), 0, c->stream, c->d_boxes_cur, (int)N.boxes.size(), (const uint32_t *)N.flags,
                         (const int *)N.d_halo_idx, (int)N.halo_peers.size(), N.epoch, N.timeout_ticks, c->cnt);
      return launch_check(c, "epoch_signal_wait");
    }
    if (N.defer_signal && !c->ov_active) {  // (mpmhip_tiled_advance_group publishes every rank's epoch with one launch)
      N.signal_deferred = true;
      return MPMHIP_OK;
    }
    hipLaunchKernelGGL(k_epoch_signal, dim3(1), dim3(64), 0, c->stream, c->d_boxes_cur, (int)N.boxes.size(), N.epoch);
    return launch_check(c, "epoch_signal");
  }
  RcclApi *R = rccl_api();
  hipStream_t st = c->stream;
  if (c->ov_active) {  // the exchange runs beside the interior kernels: side stream, fenced by events
    HIPCHK(c, hipEventRecord(N.ev_a, c->stream));
    HIPCHK(c, hipStreamWaitEvent(N.side, N.ev_a, 0));
    st = N.side;
  }
  NCCLCHK(c, R->GroupStart());
  for (const auto &b : N.boxes) {
    if (N.loop_rccl) {
      // one-GPU pre-flight of this very path (MPMHIP_WIRE_LOCAL_RCCL): the rank's communicator has ONE rank, every box is a send to
      // itself whose receive is aimed at the box's place in the peer ctx's receive buffer (same process: a plain pointer) — the same
      // group of RCCL kernels on the same stream behind the same fences, with the peer's recv[0] as the single receive buffer
      NCCLCHK(c, R->Send(N.send + b.off, (size_t)b.vol * 4, ncclFloat, 0, (ncclComm_t)N.comm, st));
      NCCLCHK(c, R->Recv(N.peers[b.peer].recv[0] + b.peer_off, (size_t)b.vol * 4, ncclFloat, 0, (ncclComm_t)N.comm, st));
      continue;
    }
    NCCLCHK(c, R->Send(N.send + b.off, (size_t)b.vol * 4, ncclFloat, b.peer, (ncclComm_t)N.comm, st));
    NCCLCHK(c, R->Recv(N.recv[0] + b.off, (size_t)b.vol * 4, ncclFloat, b.peer, (ncclComm_t)N.comm, st));
  }
  NCCLCHK(c, R->GroupEnd());
  if (st != c->stream) HIPCHK(c, hipEventRecord(N.ev_b, st));
  N.exch_on_side = st != c->stream;
  return MPMHIP_OK;
}

int tn_exchange_wait(mpmhip_ctx *c) {
  auto &N = c->tn;
  if (N.boxes.empty()) return MPMHIP_OK;
  if (N.loop_rccl) {
    // (this rank's boxes were written by its PEERS' groups: their events; its own group read its send buffer: its own event.  The
    // ranks of a local job share one stream and advance together — begin of every rank, then the ends —, so every event has been
    // recorded for this substep when the first end waits)
    if (N.exch_on_side) HIPCHK(c, hipStreamWaitEvent(c->stream, N.ev_b, 0));
    for (int p : N.halo_peers) {
      mpmhip_ctx *q = N.local_ctx[(size_t)p];
      if (q->tn.exch_on_side) HIPCHK(c, hipStreamWaitEvent(c->stream, q->tn.ev_b, 0));
    }
    return MPMHIP_OK;
  }
  if (N.wire == MPMHIP_WIRE_RCCL) {
    if (N.exch_on_side) HIPCHK(c, hipStreamWaitEvent(c->stream, N.ev_b, 0));
    return MPMHIP_OK;
  }
  if (N.wait_merged) return MPMHIP_OK;  // (tn_exchange_start's launch has waited already)
  return tn_wait(c, N.flags, N.halo_peers, N.d_halo_idx, N.epoch);
}

// ------------------------------------------------------------------------------------------------ migration
// phase A: scan (leavers per destination, bounds, top speed) and publish this rank's row of the table
int tn_mig_a(mpmhip_ctx *c) {
  auto &N = c->tn;
  const int world = N.world;
  int rc = ensure_counts(c, world);
  if (rc) return rc;
  N.mig_epoch++;
  hipLaunchKernelGGL(k_scan_init, dim3((world + 8 + 255) / 256), dim3(256), 0, c->stream, c->d_counts, world,
                     (uint32_t)std::min<uint64_t>(N.inbox_cap, 0xFFFFFFFFull));
  int grid = particle_grid(c->n_slots);
  if (grid > 128) grid = 128;
  hipLaunchKernelGGL(k_leaver_count, dim3(grid), dim3(256), 0, c->stream, c->P, c->T, (const float4 *)c->rg, (const float4 *)c->rp,
                     c->d_counts, reinterpret_cast<int *>(c->d_counts + world), c->cnt);
  if ((rc = launch_check(c, "leaver_count"))) return rc;
  uint32_t *table = N.table[N.mig_epoch & 1];
  if (N.wire == MPMHIP_WIRE_RCCL) {
    HIPCHK(c, hipMemcpyAsync(N.row, c->d_counts, sizeof(uint32_t) * (world + 8), hipMemcpyDeviceToDevice, c->stream));
    NCCLCHK(c, rccl_api()->AllGather(N.row, table, TN_ROW, ncclUint32, (ncclComm_t)N.comm, c->stream));
    return MPMHIP_OK;
  }
  PutList L;
  memset(&L, 0, sizeof L);
  for (int p = 0; p < world; p++) {
    L.src[p] = c->d_counts;
    L.dst[p] = (p == c->T.rank ? table : N.peers[p].table[N.mig_epoch & 1]) + (size_t)c->T.rank * TN_ROW;
    L.flag[p] = (p == c->T.rank ? N.flags : N.peers[p].flags) + TN_MAX_WORLD + c->T.rank;
    L.words[p] = (uint32_t)world + 8;
  }
  hipLaunchKernelGGL(k_put, dim3(world, 1), dim3(256), 0, c->stream, L, N.mig_epoch, N.d_done);
  return launch_check(c, "put (migration row)");
}

// phase B: read the table (the ONE synchronisation of a migration), pack the leavers and send them
int tn_mig_b(mpmhip_ctx *c) {
  auto &N = c->tn;
  const int world = N.world, me = c->T.rank;
  int rc;
  uint32_t *table = N.table[N.mig_epoch & 1];
  if (N.wire != MPMHIP_WIRE_RCCL && (rc = tn_wait(c, N.flags + TN_MAX_WORLD, N.all_ranks, N.d_all_idx, N.mig_epoch))) return rc;
  uint32_t *h = c->h_pinned + 2048;  // behind the counters and the cursors of mpmhip_export_leavers
  HIPCHK(c, hipMemcpyAsync(h, table, sizeof(uint32_t) * TN_ROW * world, hipMemcpyDeviceToHost, c->stream));
  Counters hc;
  if ((rc = read_counters(c, hc))) return rc;  // synchronises; reports a margin violation or a wait that timed out
  auto &M = N.mig;
  M.counts.assign((size_t)world * world, 0);
  M.rank_bounds.assign((size_t)world * 6, 0);
  int64_t total = 0;
  int lo[3] = {1 << 30, 1 << 30, 1 << 30}, hi[3] = {-(1 << 30), -(1 << 30), -(1 << 30)};
  float speed = 0.0f;
  for (int r = 0; r < world; r++) {
    const uint32_t *row = h + (size_t)r * TN_ROW;
    for (int s = 0; s < world; s++) { M.counts[(size_t)r * world + s] = row[s]; total += row[s]; }
    const int rlo[3] = {(int)row[world], (int)row[world + 1], (int)row[world + 2]};
    const int rhi[3] = {(int)row[world + 3], (int)row[world + 4], (int)row[world + 5]};
    for (int a = 0; a < 3; a++) { M.rank_bounds[(size_t)r * 6 + a] = rlo[a]; M.rank_bounds[(size_t)r * 6 + 3 + a] = rhi[a]; }
    if (rlo[0] <= rhi[0] && rlo[1] <= rhi[1] && rlo[2] <= rhi[2]) {  // (a rank without particles reports lo > hi)
      for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], rlo[a]); hi[a] = std::max(hi[a], rhi[a]); }
    }
    float s;
    memcpy(&s, &row[world + 6], 4);
    if (s > speed) speed = s;
  }
  for (int a = 0; a < 3; a++) { M.lo[a] = lo[a]; M.hi[a] = hi[a]; }
  M.speed = speed;
  M.total = total;
  M.n_out = 0; M.n_in = 0;
  for (int s = 0; s < world; s++) { M.n_out += M.counts[(size_t)me * world + s]; M.n_in += M.counts[(size_t)s * world + me]; }
  if (total == 0) return MPMHIP_OK;
  // records: grouped by destination in mig_send (k_leaver_pack), then one message per destination
  if ((size_t)M.n_out > N.mig_send_cap) {
    if (N.mig_send) (void)hipFree(N.mig_send);
    N.mig_send = nullptr;
    N.mig_send_cap = (size_t)M.n_out + (size_t)M.n_out / 2 + 4096;
    HIPCHK(c, dmalloc(&N.mig_send, N.mig_send_cap * 11));
  }
  for (int s = 0; s < world; s++) {  // every rank sees every inbox: all refuse together, BEFORE anybody writes beyond one
    int64_t arriving = 0;
    for (int r = 0; r < world; r++) arriving += M.counts[(size_t)r * world + s];
    const uint32_t room = h[(size_t)s * TN_ROW + world + 7];
    if (arriving > (int64_t)room)
      return fail(c, MPMHIP_ECAPACITY, "migration: %lld particles arriving at rank %d exceed its inbox of %u records (mpmhip_tiled_config.inbox_records)",
                  (long long)arriving, s, room);
  }
  std::vector<int64_t> mine((size_t)world);
  for (int s = 0; s < world; s++) mine[s] = M.counts[(size_t)me * world + s];
  if (M.n_out && (rc = mpmhip_export_leavers(c, world, mine.data(), N.mig_send))) return rc;
  N.migrated_out += M.n_out;
  if (N.wire == MPMHIP_WIRE_RCCL) {
    RcclApi *R = rccl_api();
    NCCLCHK(c, R->GroupStart());
    uint64_t soff = 0, roff = 0;
    for (int s = 0; s < world; s++) {
      const int64_t ns = M.counts[(size_t)me * world + s], nr = M.counts[(size_t)s * world + me];
      if (s != me && ns) NCCLCHK(c, R->Send(N.mig_send + soff * 11, (size_t)ns * MPMHIP_MIGRATE_FLOATS, ncclFloat, s, (ncclComm_t)N.comm, c->stream));
      if (s != me && nr) NCCLCHK(c, R->Recv(N.inbox + roff * 11, (size_t)nr * MPMHIP_MIGRATE_FLOATS, ncclFloat, s, (ncclComm_t)N.comm, c->stream));
      soff += ns; roff += nr;
    }
    NCCLCHK(c, R->GroupEnd());
    return MPMHIP_OK;
  }
  PutList L;
  memset(&L, 0, sizeof L);
  uint64_t soff = 0;
  uint32_t most = 0;
  for (int s = 0; s < world; s++) {
    const int64_t ns = M.counts[(size_t)me * world + s];
    uint64_t before = 0;  // records of lower ranks in s's inbox
    for (int r = 0; r < me; r++) before += M.counts[(size_t)r * world + s];
    L.src[s] = reinterpret_cast<const uint32_t *>(N.mig_send + soff * 11);
    L.dst[s] = s == me ? nullptr : reinterpret_cast<uint32_t *>(N.peers[s].inbox + before * 11);
    L.words[s] = s == me ? 0u : (uint32_t)(ns * MPMHIP_MIGRATE_FLOATS);
    L.flag[s] = (s == me ? N.flags : N.peers[s].flags) + 2 * TN_MAX_WORLD + me;
    if (s == me) { L.src[s] = nullptr; }
    most = std::max(most, L.words[s]);
    soff += ns;
  }
  const int chunks = (int)std::min<uint32_t>(64u, std::max<uint32_t>(1u, most / (256 * 16)));
  hipLaunchKernelGGL(k_put, dim3(world, chunks), dim3(256), 0, c->stream, L, N.mig_epoch, N.d_done);
  return launch_check(c, "put (migration records)");
}

// phase C: import the arrivals, keep the halo boxes wrapped around the particles, schedule the next migration
int tn_mig_c(mpmhip_ctx *c) {
  auto &N = c->tn;
  auto &M = N.mig;
  int rc;
  if (M.total > 0) {
    if (N.wire != MPMHIP_WIRE_RCCL && (rc = tn_wait(c, N.flags + 2 * TN_MAX_WORLD, N.all_ranks, N.d_all_idx, N.mig_epoch))) return rc;
    if (M.n_in && (rc = mpmhip_import_particles(c, M.n_in, N.inbox))) return rc;
    if (c->n_slots > (int64_t)(0.85 * (double)c->cap)) c->compact_requested = true;  // dead slots (leavers) pile up
  }
  if (!N.initial_scan) N.migrations++;
  // The occupancy boxes: do they still hold every node the particles can touch before the next check?
  // A particle with base cell b touches the nodes b .. b + 2; the table carries every rank's base-cell bounds (M.rank_bounds; M.lo / M.hi
  // bound ALL ranks' live particles), taken BEFORE this migration's records moved: a rank's particles after it lie inside its own
  // bounds joined with those of every rank that sent it records.
  // (a) Looking back: the halo boxes of the substeps since the last check were cut to these boxes, so mass splatted outside a rank's
  //     box was never exchanged.  The schedule below lets the particles travel margin cells between two checks if their top speed
  //     at most doubles, and (b) keeps 2 margin + 1 cells of room, but a particle deep inside a brick that speeds up further
  //     (it trips no margin test: error bit 2 only watches the brick's own faces) could have left the box unseen: that is an
  //     error here, not a silent loss.
  const int world = N.world;
  bool have = M.lo[0] <= M.hi[0] && M.lo[1] <= M.hi[1] && M.lo[2] <= M.hi[2];
  if (have) {
    auto nonempty = [](const int *b) { return b[0] <= b[3] && b[1] <= b[4] && b[2] <= b[5]; };
    for (int r = 0; r < world; r++) {
      const int *rb = &M.rank_bounds[(size_t)r * 6];
      if (!nonempty(rb)) continue;
      for (int a = 0; a < 3; a++) {
        const int olo = N.occ_valid ? N.occ[(size_t)r * 6 + a] : N.clip_lo[a], ohi = N.occ_valid ? N.occ[(size_t)r * 6 + 3 + a] : N.clip_hi[a];
        // (a side where the box ends at the grid itself holds everything there is: with clean_boundary off, or on the clamped generic
        // path, base cells reach res - 1, and no halo mass can lie beyond the last node)
        if (rb[a] < olo || (rb[3 + a] + 2 > ohi && ohi != c->P.res[a] + 1))
          return fail(c, MPMHIP_ECAPACITY, "tiled run: particles of rank %d left the clipped halo region on axis %d between two migrations (base cells "
                      "[%d, %d), halo boxes cut to nodes [%d, %d)): their top speed more than doubled since the last check — halo sums "
                      "of the substeps in between may be incomplete; lower mpmhip_tiled_config.migrate_cap or set migrate_interval",
                      r, a, rb[a], rb[3 + a], olo, ohi);
      }
    }
    // every rank's bounds once this migration's records have arrived
    std::vector<int> post(M.rank_bounds);
    for (int r = 0; r < world; r++)
      for (int s2 = 0; s2 < world; s2++) {
        if (s2 == r || M.counts[(size_t)s2 * world + r] == 0) continue;
        const int *sb = &M.rank_bounds[(size_t)s2 * 6];
        for (int a = 0; a < 3; a++) {
          post[(size_t)r * 6 + a] = std::min(post[(size_t)r * 6 + a], sb[a]);
          post[(size_t)r * 6 + 3 + a] = std::max(post[(size_t)r * 6 + 3 + a], sb[3 + a]);
        }
      }
    // (b) Looking ahead: room for a travel of 2 margin cells on every side (the schedule plans for margin)
    bool covers = N.occ_valid;
    const int room = 2 * c->T.margin;
    for (int r = 0; r < world && covers; r++) {
      const int *pb = &post[(size_t)r * 6];
      if (!nonempty(pb)) continue;
      for (int a = 0; a < 3; a++) {
        const int olo = N.occ[(size_t)r * 6 + a], ohi = N.occ[(size_t)r * 6 + 3 + a];
        covers = covers && (pb[a] - room - 1 >= olo || olo == 0);
        covers = covers && (pb[3 + a] + room + 3 <= ohi || ohi == c->P.res[a] + 1);
      }
    }
    // fresh boxes: the bounds with tn_clip_slack cells around them (a rank without particles: an empty box — it neither sends nor gathers)
    const int s = tn_clip_slack(c->T);
    std::vector<int> fresh((size_t)world * 6, 0);
    int flo[3], fhi[3];
    for (int a = 0; a < 3; a++) { flo[a] = std::max(0, M.lo[a] - s); fhi[a] = std::min(c->P.res[a] + 1, M.hi[a] + s + 2); }
    for (int r = 0; r < world; r++) {
      const int *pb = &post[(size_t)r * 6];
      if (!nonempty(pb)) continue;
      for (int a = 0; a < 3; a++) {
        fresh[(size_t)r * 6 + a] = std::max(0, pb[a] - s);
        fresh[(size_t)r * 6 + 3 + a] = std::min(c->P.res[a] + 1, pb[3 + a] + s + 2);
      }
    }
    // ... and a re-plan also when the boxes in use have become much too wide (clusters that have drifted apart; the first check of a
    // job, whose boxes are the global clip box of the set-up): all ranks' plans together, which every rank computes alike
    bool shrink = false;
    if (covers) {
      uint64_t now = 0, then = 0;
      for (int r = 0; r < world; r++) {
        uint64_t t0 = 0, t1 = 0;
        (void)tn_plan(c->T, world, N.clip_lo, N.clip_hi, N.occ.data(), r, &t0);
        (void)tn_plan(c->T, world, flo, fhi, fresh.data(), r, &t1);
        now += t0; then += t1;
      }
      shrink = 2 * then < now;
    }
    if (!covers || shrink) {
      for (int a = 0; a < 3; a++) { N.clip_lo[a] = flo[a]; N.clip_hi[a] = fhi[a]; }
      N.occ = fresh;
      N.occ_valid = true;
      if ((rc = tn_apply_plan(c))) return rc;
      N.replans++;
    }
  }
  if (N.initial_scan) return MPMHIP_OK;  // (the scan in front of a job's first substep plans, it does not move the schedule)
  // schedule: after a migration every particle is inside its brick and needs margin / speed substeps to cross the margin;
  // half of that leaves room for the speed to double in between; never sooner than the CFL schedule, never later than the cap
  int64_t n = N.migrate_interval;
  if (N.adaptive_cap > n && std::isfinite(M.speed))
    n = std::max<int64_t>(n, std::min<int64_t>(N.adaptive_cap, (int64_t)(0.5 * c->T.margin / std::max(M.speed, 1e-9f))));
  N.next_migration = N.k + n;
  return MPMHIP_OK;
}

// ------------------------------------------------------------------------------------------------ reductions
// SURVEY section 8(e) collective (3): a few scalars over all ranks (energy, live particles, the sticky error word).
//   RCCL       ncclAllReduce on a device row, read back
//   peer wires every rank writes its row into every rank's table (k_put: the writer of the last row piece publishes the
//              reduction's epoch), waits for the world's epochs, reads the table back and reduces it in RANK ORDER on the host —
//              every rank gets the bit-identical result.  Tables alternate by the parity of the reduction: a rank reaches
//              reduction r + 2 only after it has read the table of r + 1, which every rank fills after its read of r.
// put: this rank's share on its way (no host synchronisation); finish: wait, read back, reduce.
int tn_reduce_put(mpmhip_ctx *c, const double *vals, int n, int op) {
  auto &N = c->tn;
  if (n < 1 || n > TN_RED_N) return fail(c, MPMHIP_EINVAL, "reduce: 1 .. %d values", TN_RED_N);
  if (op != MPMHIP_REDUCE_SUM && op != MPMHIP_REDUCE_MAX && op != MPMHIP_REDUCE_MIN) return fail(c, MPMHIP_EINVAL, "reduce: unknown operation %d", op);
  N.red_epoch++;
  double *h = reinterpret_cast<double *>(c->h_pinned + TN_RED_PIN_WORD);
  for (int i = 0; i < TN_RED_N; i++) h[i] = i < n ? vals[i] : 0.0;
  HIPCHK(c, hipMemcpyAsync(N.d_red, h, sizeof(double) * TN_RED_N, hipMemcpyHostToDevice, c->stream));
  if (N.wire == MPMHIP_WIRE_RCCL && N.comm) {  // (also with one rank: the binding is exercised on every box)
    const ncclRedOp_t rop = op == MPMHIP_REDUCE_SUM ? ncclSum : (op == MPMHIP_REDUCE_MAX ? ncclMax : ncclMin);
    NCCLCHK(c, rccl_api()->AllReduce(N.d_red, N.d_red, (size_t)n, ncclDouble, rop, (ncclComm_t)N.comm, c->stream));
    return MPMHIP_OK;
  }
  if (N.world == 1) return MPMHIP_OK;
  PutList L;
  memset(&L, 0, sizeof L);
  const int par = (int)(N.red_epoch & 1u);
  for (int p = 0; p < N.world; p++) {
    const auto &P = N.peers[p];
    L.src[p] = reinterpret_cast<const uint32_t *>(N.d_red);
    L.dst[p] = reinterpret_cast<uint32_t *>(P.red[par] + (size_t)c->T.rank * TN_RED_N);
    L.flag[p] = P.flags + 3 * TN_MAX_WORLD + c->T.rank;
    L.words[p] = 2u * (uint32_t)TN_RED_N;
  }
  hipLaunchKernelGGL(k_put, dim3(N.world, 1), dim3(256), 0, c->stream, L, N.red_epoch, N.d_done);
  return launch_check(c, "put (reduction row)");
}
int tn_reduce_finish(mpmhip_ctx *c, double *vals, int n, int op) {
  auto &N = c->tn;
  double *h = reinterpret_cast<double *>(c->h_pinned + TN_RED_PIN_WORD);
  if (N.world == 1 || N.wire == MPMHIP_WIRE_RCCL) {
    HIPCHK(c, hipMemcpyAsync(h, N.d_red, sizeof(double) * TN_RED_N, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; i++) vals[i] = h[i];
    return MPMHIP_OK;
  }
  int rc = tn_wait(c, N.flags + 3 * TN_MAX_WORLD, N.all_ranks, N.d_all_idx, N.red_epoch);
  if (rc) return rc;
  const int par = (int)(N.red_epoch & 1u);
  HIPCHK(c, hipMemcpyAsync(h, N.peers[c->T.rank].red[par], sizeof(double) * TN_RED_N * N.world, hipMemcpyDeviceToHost, c->stream));
  Counters hc;
  if ((rc = read_counters(c, hc))) return rc;  // synchronises; reports a wait that timed out
  for (int i = 0; i < n; i++) {
    double v = h[i];
    for (int r = 1; r < N.world; r++) {
      const double w = h[(size_t)r * TN_RED_N + i];
      v = op == MPMHIP_REDUCE_SUM ? v + w : (op == MPMHIP_REDUCE_MAX ? std::max(v, w) : std::min(v, w));
    }
    vals[i] = v;
  }
  return MPMHIP_OK;
}

int tn_check_ready(mpmhip_ctx *c, const char *who) {
  if (!c->tn.on) return fail(c, MPMHIP_EINVAL, "%s needs mpmhip_tiled_setup first", who);
  if (!tn_connected(c)) return fail(c, MPMHIP_EINVAL, "%s: the wire is not connected (mpmhip_comm_init / mpmhip_tiled_ipc_connect / mpmhip_tiled_connect_local)", who);
  return MPMHIP_OK;
}

// the whole job's energy on one rank (mpmhip_calculate_energy of a tiled ctx): this rank's share with the peers' halo sums, then
// the sum over the ranks.  out = {kinetic, potential, particles without potential_energy()}
int tn_energy(mpmhip_ctx *c, double out[3]) {
  int rc = tn_check_ready(c, "calculate_energy");
  if (rc) return rc;
  if (c->tn.wire == MPMHIP_WIRE_LOCAL) return fail(c, MPMHIP_EINVAL, "ranks of a local job compute their energy together: mpmhip_calculate_energy_group");
  if ((rc = energy_begin(c)) || (rc = tn_exchange_start(c)) || (rc = tn_exchange_wait(c)) || (rc = energy_end(c, out))) return rc;
  if ((rc = tn_reduce_put(c, out, 3, MPMHIP_REDUCE_SUM))) return rc;
  return tn_reduce_finish(c, out, 3, MPMHIP_REDUCE_SUM);
}

void tn_free(mpmhip_ctx *c) {
  auto &N = c->tn;
  for (auto &p : N.peers)
    if (p.ipc_base) (void)hipIpcCloseMemHandle(p.ipc_base);
  N.peers.clear();
  if (N.arena) (void)hipFree(N.arena);
  if (N.send) (void)hipFree(N.send);
  if (N.mig_send) (void)hipFree(N.mig_send);
  if (N.row) (void)hipFree(N.row);
  for (int k = 0; k < 2; k++) if (N.d_boxes[k]) (void)hipFree(N.d_boxes[k]);
  if (N.d_halo_idx) (void)hipFree(N.d_halo_idx);
  if (N.d_all_idx) (void)hipFree(N.d_all_idx);
  if (N.d_done) (void)hipFree(N.d_done);
  if (N.d_red) (void)hipFree(N.d_red);
  if (N.side) { (void)hipStreamSynchronize(N.side); (void)hipStreamDestroy(N.side); }
  if (N.ev_a) (void)hipEventDestroy(N.ev_a);
  if (N.ev_b) (void)hipEventDestroy(N.ev_b);
  if (N.comm && rccl_api()->handle) (void)rccl_api()->CommDestroy((ncclComm_t)N.comm);
  N = mpmhip_ctx::TiledNative();
}

}  // namespace

extern "C" {

int mpmhip_comm_unique_id(uint8_t id[MPMHIP_COMM_ID_BYTES]) {
  if (!id) return MPMHIP_EINVAL;
  RcclApi *R = rccl_api();
  if (!R->handle) return fail(nullptr, MPMHIP_EHIP, "librccl could not be loaded: %s", R->error.c_str());
  ncclUniqueId u;
  static_assert(sizeof u == MPMHIP_COMM_ID_BYTES, "ncclUniqueId size");
  ncclResult_t r = R->GetUniqueId(&u);
  if (r != ncclSuccess) return fail(nullptr, MPMHIP_EHIP, "ncclGetUniqueId failed: %s", R->GetErrorString(r));
  memcpy(id, &u, sizeof u);
  return MPMHIP_OK;
}

int mpmhip_comm_init(mpmhip_ctx *c, const uint8_t id[MPMHIP_COMM_ID_BYTES], int32_t rank, int32_t world) {
  if (!c || !id) return MPMHIP_EINVAL;
  if (world < 1 || world > TN_MAX_WORLD || rank < 0 || rank >= world) return fail(c, MPMHIP_EINVAL, "rank %d of %d (at most %d ranks)", rank, world, TN_MAX_WORLD);
  if (c->tn.comm) return fail(c, MPMHIP_EINVAL, "this ctx already has a communicator");
  RcclApi *R = rccl_api();
  if (!R->handle) return fail(c, MPMHIP_EHIP, "librccl could not be loaded: %s", R->error.c_str());
  HIPCHK(c, hipSetDevice(c->device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  ncclComm_t comm = nullptr;
  NCCLCHK(c, R->CommInitRank(&comm, world, u, rank));
  c->tn.comm = comm;
  c->tn.comm_rank = rank; c->tn.comm_world = world;
  return MPMHIP_OK;
}

int mpmhip_comm_destroy(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  if (!c->tn.comm) return MPMHIP_OK;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  NCCLCHK(c, rccl_api()->CommDestroy((ncclComm_t)c->tn.comm));
  c->tn.comm = nullptr;
  return MPMHIP_OK;
}

int mpmhip_comm_selftest(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  auto &N = c->tn;
  if (!N.comm) return fail(c, MPMHIP_EINVAL, "mpmhip_comm_init first");
  RcclApi *R = rccl_api();
  HIPCHK(c, hipSetDevice(c->device));
  const int world = N.comm_world, me = N.comm_rank, n = 1024;
  uint32_t *d = nullptr;
  HIPCHK(c, dmalloc(&d, (size_t)n * (world + 3)));
  std::vector<uint32_t> h((size_t)n * (world + 3));
  for (int i = 0; i < n; i++) h[i] = (uint32_t)(me * 1000003 + i);
  hipError_t e = hipMemcpy(d, h.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice);
  auto bail = [&](int code) { (void)hipFree(d); return code; };
  if (e != hipSuccess) return bail(fail(c, MPMHIP_EHIP, "selftest upload: %s", hipGetErrorString(e)));
  uint32_t *gathered = d + n, *ring = d + (size_t)n * (world + 1), *ring_out = ring + n;
  ncclResult_t r = R->AllGather(d, gathered, n, ncclUint32, (ncclComm_t)N.comm, c->stream);
  const int next = (me + 1) % world, prev = (me + world - 1) % world;
  if (r == ncclSuccess) r = R->GroupStart();
  if (r == ncclSuccess) r = R->Send(d, n, ncclUint32, next, (ncclComm_t)N.comm, c->stream);
  if (r == ncclSuccess) r = R->Recv(ring, n, ncclUint32, prev, (ncclComm_t)N.comm, c->stream);
  if (r == ncclSuccess) r = R->GroupEnd();
  // all-reduce (sum) of the first 8 words as uint32: word i becomes sum over ranks of (rank * 1000003 + i), into ring_out
  if (r == ncclSuccess) r = R->AllReduce(d, ring_out, 8, ncclUint32, ncclSum, (ncclComm_t)N.comm, c->stream);
  if (r != ncclSuccess) return bail(fail(c, MPMHIP_EHIP, "selftest: %s", R->GetErrorString(r)));
  e = hipMemcpyAsync(h.data(), d, sizeof(uint32_t) * h.size(), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return bail(fail(c, MPMHIP_EHIP, "selftest read-back: %s", hipGetErrorString(e)));
  for (int s = 0; s < world; s++)
    for (int i = 0; i < n; i++)
      if (h[(size_t)n * (1 + s) + i] != (uint32_t)(s * 1000003 + i)) return bail(fail(c, MPMHIP_EHIP, "selftest: all-gather delivered wrong data (rank %d, word %d)", s, i));
  for (int i = 0; i < n; i++)
    if (h[(size_t)n * (world + 1) + i] != (uint32_t)(prev * 1000003 + i)) return bail(fail(c, MPMHIP_EHIP, "selftest: send / receive delivered wrong data (word %d)", i));
  for (int i = 0; i < 8; i++) {
    uint32_t want = 0;
    for (int s = 0; s < world; s++) want += (uint32_t)(s * 1000003 + i);
    if (h[(size_t)n * (world + 2) + i] != want) return bail(fail(c, MPMHIP_EHIP, "selftest: all-reduce delivered wrong data (word %d)", i));
  }
  return bail(MPMHIP_OK);
}

int mpmhip_tiled_setup(mpmhip_ctx *c, const mpmhip_tiled_config *cfg, const int32_t *cuts_x, const int32_t *cuts_y, const int32_t *cuts_z) {
  if (!c || !cfg || !cuts_x || !cuts_y || !cuts_z) return MPMHIP_EINVAL;
  if (c->in_substep) return fail(c, MPMHIP_EINVAL, "tiled_setup inside a substep");
  if (cfg->wire != MPMHIP_WIRE_RCCL && cfg->wire != MPMHIP_WIRE_IPC && cfg->wire != MPMHIP_WIRE_LOCAL && cfg->wire != MPMHIP_WIRE_LOCAL_RCCL)
    return fail(c, MPMHIP_EINVAL, "unknown wire %d", cfg->wire);
  const bool loop_rccl = cfg->wire == MPMHIP_WIRE_LOCAL_RCCL;
  if (loop_rccl && (!c->tn.comm || c->tn.comm_world != 1))
    return fail(c, MPMHIP_EINVAL, "MPMHIP_WIRE_LOCAL_RCCL: every ctx of the job needs its own ONE-rank communicator (mpmhip_comm_init(ctx, id, 0, 1))");
  if (cfg->world != cfg->dims[0] * cfg->dims[1] * cfg->dims[2] || cfg->world > TN_MAX_WORLD) return fail(c, MPMHIP_EINVAL, "world %d does not match dims (at most %d ranks)", cfg->world, TN_MAX_WORLD);
  if (cfg->migrate_interval < 0 || cfg->migrate_interval > cfg->margin) return fail(c, MPMHIP_EINVAL, "migrate_interval must be in [0, margin]");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int rc = mpmhip_set_partition(c, cfg->rank, cfg->dims, cuts_x, cuts_y, cuts_z, cfg->margin);
  if (rc) return rc;
  void *keep_comm = c->tn.comm;
  const int keep_rank = c->tn.comm_rank, keep_world = c->tn.comm_world;
  c->tn.comm = nullptr;
  tn_free(c);
  auto &N = c->tn;
  N.comm = keep_comm; N.comm_rank = keep_rank; N.comm_world = keep_world;
  N.loop_rccl = loop_rccl;
  if (N.comm && !loop_rccl && (N.comm_world != cfg->world || N.comm_rank != cfg->rank)) return fail(c, MPMHIP_EINVAL, "the communicator is rank %d of %d, the partition rank %d of %d", N.comm_rank, N.comm_world, cfg->rank, cfg->world);
  N.world = cfg->world;
  N.wire = loop_rccl ? MPMHIP_WIRE_LOCAL : cfg->wire;  // (everything but the halo boxes travels as on the local wire)
  for (int a = 0; a < 3; a++) {
    N.clip_lo[a] = std::max(0, cfg->clip_lo[a]);
    N.clip_hi[a] = std::min(c->P.res[a] + 1, cfg->clip_hi[a]);
    if (N.clip_lo[a] >= N.clip_hi[a]) return fail(c, MPMHIP_EINVAL, "empty clip box on axis %d", a);
  }
  N.migrate_interval = cfg->migrate_interval > 0 ? cfg->migrate_interval : cfg->margin;
  N.adaptive_cap = cfg->migrate_interval > 0 ? 0 : (cfg->migrate_cap > 0 ? cfg->migrate_cap : 64);
  N.k = 0; N.next_migration = N.migrate_interval;
  N.occ.assign((size_t)N.world * 6, 0);
  N.occ_valid = false; N.initial_scan = false;
  N.merge_signal_wait = !(getenv("MPMHIP_TILE_MERGE_WAIT") && atoi(getenv("MPMHIP_TILE_MERGE_WAIT")) == 0);  // (A/B knob)
  N.timeout_ticks = 100000000ull * (unsigned long long)std::max(1, getenv("MPMHIP_TILE_WAIT_S") ? atoi(getenv("MPMHIP_TILE_WAIT_S")) : 20);
  // worst-case halo volume over all ranks: the clip box = the whole grid (so that every rank lays its arena out alike)
  const int full_lo[3] = {0, 0, 0}, full_hi[3] = {c->P.res[0] + 1, c->P.res[1] + 1, c->P.res[2] + 1};
  uint64_t cap = 0;
  for (int r = 0; r < N.world; r++) {
    uint64_t t = 0;
    const auto plan = tn_plan(c->T, N.world, full_lo, full_hi, nullptr, r, &t);
    if (plan.size() > MPMHIP_MAX_HALO_BOXES) return fail(c, MPMHIP_EINVAL, "too many halo boxes");
    cap = std::max(cap, t);
  }
  if (cap >= (1ull << 31)) return fail(c, MPMHIP_EINVAL, "halo boxes too large");
  N.halo_cap = std::max<uint64_t>(cap, 1);
  N.inbox_cap = cfg->inbox_records > 0 ? (uint64_t)cfg->inbox_records : std::max<uint64_t>(65536, (uint64_t)c->cap / 8);
  // arena: [flags 4 KiB | reduction tables 16 KiB | table 0 | table 1 | recv 0 | recv 1 | inbox]
  const size_t table_bytes = ((sizeof(uint32_t) * TN_ROW * TN_MAX_WORLD + 255) / 256) * 256;
  const size_t recv_bytes = ((sizeof(float4) * N.halo_cap + 255) / 256) * 256;
  N.arena_bytes = TN_FLAG_BYTES + TN_RED_BYTES + 2 * table_bytes + 2 * recv_bytes + sizeof(float4) * 11 * N.inbox_cap;
  const bool peer_wire = N.wire == MPMHIP_WIRE_IPC || N.wire == MPMHIP_WIRE_LOCAL;
  hipError_t e;
  if (N.wire == MPMHIP_WIRE_IPC) {
    // written by kernels of other devices while kernels of this one poll and read it: fine-grained (coherent at system
    // scope while kernels run); MPMHIP_TILE_COARSE=1 allocates ordinary device memory instead (single-device tests)
    const bool coarse = getenv("MPMHIP_TILE_COARSE") && atoi(getenv("MPMHIP_TILE_COARSE")) != 0;
    e = coarse ? hipMalloc((void **)&N.arena, N.arena_bytes) : hipExtMallocWithFlags((void **)&N.arena, N.arena_bytes, hipDeviceMallocFinegrained);
  } else {
    e = hipMalloc((void **)&N.arena, N.arena_bytes);
  }
  if (e != hipSuccess) return fail(c, MPMHIP_ENOMEM, "halo arena of %zu bytes: %s", N.arena_bytes, hipGetErrorString(e));
  HIPCHK(c, hipMemset(N.arena, 0, N.arena_bytes));
  mpmhip_ctx::TiledNative::Peer self;
  tn_carve(self, N.arena, table_bytes, recv_bytes);
  N.flags = self.flags; N.table[0] = self.table[0]; N.table[1] = self.table[1];
  N.recv[0] = self.recv[0]; N.recv[1] = self.recv[1]; N.inbox = self.inbox;
  N.table_bytes = table_bytes; N.recv_bytes = recv_bytes;
  N.peers.assign((size_t)N.world, mpmhip_ctx::TiledNative::Peer());
  N.peers[cfg->rank] = self;
  if (!peer_wire || N.loop_rccl) HIPCHK(c, dmalloc(&N.send, (size_t)N.halo_cap));
  HIPCHK(c, dmalloc(&N.row, (size_t)TN_ROW));
  HIPCHK(c, hipMemset(N.row, 0, sizeof(uint32_t) * TN_ROW));
  for (int k = 0; k < 2; k++) HIPCHK(c, dmalloc(&N.d_boxes[k], (size_t)MPMHIP_MAX_HALO_BOXES));
  HIPCHK(c, dmalloc(&N.d_halo_idx, (size_t)MPMHIP_MAX_HALO_BOXES));
  HIPCHK(c, dmalloc(&N.d_all_idx, (size_t)TN_MAX_WORLD));
  HIPCHK(c, dmalloc(&N.d_red, (size_t)TN_RED_N * (TN_MAX_WORLD + 1)));
  HIPCHK(c, dmalloc(&N.d_done, (size_t)TN_MAX_WORLD + 1));
  HIPCHK(c, hipMemset(N.d_done, 0, sizeof(uint32_t) * (TN_MAX_WORLD + 1)));
  N.all_ranks.resize((size_t)N.world);
  for (int r = 0; r < N.world; r++) N.all_ranks[r] = r;
  HIPCHK(c, hipMemcpy(N.d_all_idx, N.all_ranks.data(), sizeof(int) * N.world, hipMemcpyHostToDevice));
  if (N.wire == MPMHIP_WIRE_RCCL || N.loop_rccl) {
    HIPCHK(c, hipStreamCreateWithFlags(&N.side, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&N.ev_a, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&N.ev_b, hipEventDisableTiming));
  }
  N.on = true;
  N.connected = N.world == 1;
  c->overlap = cfg->overlap != 0;
  return tn_apply_plan(c);
}

int mpmhip_tiled_ipc_handle(mpmhip_ctx *c, uint8_t handle[MPMHIP_IPC_HANDLE_BYTES]) {
  if (!c || !handle) return MPMHIP_EINVAL;
  if (!c->tn.on || c->tn.wire != MPMHIP_WIRE_IPC) return fail(c, MPMHIP_EINVAL, "ipc_handle needs mpmhip_tiled_setup with MPMHIP_WIRE_IPC");
  HIPCHK(c, hipSetDevice(c->device));
  hipIpcMemHandle_t h;
  static_assert(sizeof h == MPMHIP_IPC_HANDLE_BYTES, "hipIpcMemHandle_t size");
  HIPCHK(c, hipIpcGetMemHandle(&h, c->tn.arena));
  memcpy(handle, &h, sizeof h);
  return MPMHIP_OK;
}

int mpmhip_tiled_ipc_connect(mpmhip_ctx *c, const uint8_t *handles) {
  if (!c) return MPMHIP_EINVAL;
  auto &N = c->tn;
  if (!N.on || N.wire != MPMHIP_WIRE_IPC) return fail(c, MPMHIP_EINVAL, "ipc_connect needs mpmhip_tiled_setup with MPMHIP_WIRE_IPC");
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<uint8_t> gathered;
  if (!handles) {  // through the ctx's communicator
    if (!N.comm) return fail(c, MPMHIP_EINVAL, "ipc_connect(NULL) needs mpmhip_comm_init");
    uint8_t mine[MPMHIP_IPC_HANDLE_BYTES];
    int rc = mpmhip_tiled_ipc_handle(c, mine);
    if (rc) return rc;
    uint8_t *d = nullptr;
    HIPCHK(c, dmalloc(&d, (size_t)MPMHIP_IPC_HANDLE_BYTES * (N.world + 1)));
    hipError_t e = hipMemcpy(d, mine, sizeof mine, hipMemcpyHostToDevice);
    ncclResult_t r = ncclSuccess;
    if (e == hipSuccess) r = rccl_api()->AllGather(d, d + MPMHIP_IPC_HANDLE_BYTES, MPMHIP_IPC_HANDLE_BYTES, ncclUint8, (ncclComm_t)N.comm, c->stream);
    gathered.resize((size_t)MPMHIP_IPC_HANDLE_BYTES * N.world);
    if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(gathered.data(), d + MPMHIP_IPC_HANDLE_BYTES, gathered.size(), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (r != ncclSuccess) return fail(c, MPMHIP_EHIP, "all-gather of the IPC handles: %s", rccl_api()->GetErrorString(r));
    HIPCHK(c, e);
    handles = gathered.data();
  }
  const size_t table_bytes = N.table_bytes, recv_bytes = N.recv_bytes;
  for (int p = 0; p < N.world; p++) {
    if (p == c->T.rank) continue;
    auto &P = N.peers[p];
    if (P.ipc_base) { (void)hipIpcCloseMemHandle(P.ipc_base); P.ipc_base = nullptr; }
    hipIpcMemHandle_t h;
    memcpy(&h, handles + (size_t)p * MPMHIP_IPC_HANDLE_BYTES, sizeof h);
    void *base = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return fail(c, MPMHIP_EHIP, "hipIpcOpenMemHandle of rank %d's arena: %s", p, hipGetErrorString(e));
    P.ipc_base = base;
    tn_carve(P, static_cast<char *>(base), table_bytes, recv_bytes);
  }
  N.connected = true;
  return tn_apply_plan(c);
}

int mpmhip_tiled_connect_local(mpmhip_ctx *const *ctxs, int32_t n) {
  if (!ctxs || n < 1) return MPMHIP_EINVAL;
  for (int r = 0; r < n; r++) {
    mpmhip_ctx *c = ctxs[r];
    if (!c) return MPMHIP_EINVAL;
    if (!c->tn.on || c->tn.wire != MPMHIP_WIRE_LOCAL || c->tn.world != n || c->T.rank != r)
      return fail(c, MPMHIP_EINVAL, "connect_local: ctx %d must be set up as rank %d of %d with MPMHIP_WIRE_LOCAL", r, r, n);
    if (c->device != ctxs[0]->device) return fail(c, MPMHIP_EINVAL, "connect_local: all ctx must live on one device");
    if (c->tn.halo_cap != ctxs[0]->tn.halo_cap) return fail(c, MPMHIP_EINVAL, "connect_local: the ranks' arenas are laid out differently (not the same partition?)");
  }
  for (int r = 0; r < n; r++) {
    mpmhip_ctx *c = ctxs[r];
    if (c->tn.loop_rccl != ctxs[0]->tn.loop_rccl) return fail(c, MPMHIP_EINVAL, "connect_local: the ranks were set up with different wires");
    for (int p = 0; p < n; p++)
      if (p != r) c->tn.peers[p] = ctxs[p]->tn.peers[p];  // (a rank's own entry describes its arena)
    c->tn.local_ctx.assign(ctxs, ctxs + n);
    c->tn.connected = true;
    int rc = tn_apply_plan(c);
    if (rc) return rc;
  }
  return MPMHIP_OK;
}

static int tn_substep_parts(mpmhip_ctx *c, int part) {  // 0 begin (+ start of the exchange), 1 interior, 2 wait + end
  int rc;
  if (part == 0) {
    if ((rc = mpmhip_substep_begin(c))) return rc;
    return tn_exchange_start(c);
  }
  if (part == 1) return mpmhip_substep_interior(c);
  if ((rc = tn_exchange_wait(c))) return rc;
  return mpmhip_substep_end(c);
}

int64_t mpmhip_tiled_advance(mpmhip_ctx *c, int64_t n) {
  if (!c || n < 0) return MPMHIP_EINVAL;
  int rc = tn_check_ready(c, "tiled_advance");
  if (rc) return rc;
  auto &N = c->tn;
  if (N.wire == MPMHIP_WIRE_LOCAL && N.world > 1) return fail(c, MPMHIP_EINVAL, "ranks of a local job advance together: mpmhip_tiled_advance_group");
  HIPCHK(c, hipSetDevice(c->device));
  if (n > 0 && !N.occ_valid && N.world > 1) {
    // in front of a job's first substep: one scan + table exchange (no particle is outside its brick yet) — every rank learns every
    // rank's bounds and the halo boxes shrink from the set-up's global clip box to the ranks' own occupancy (tn_mig_c)
    N.initial_scan = true;
    rc = tn_mig_a(c);
    if (!rc) rc = tn_mig_b(c);
    if (!rc) rc = tn_mig_c(c);
    N.initial_scan = false;
    if (rc) return rc;
  }
  for (int64_t i = 0; i < n; i++) {
    for (int part = 0; part < 3; part++)
      if ((rc = tn_substep_parts(c, part))) { c->in_substep = false; c->cur_ev = nullptr; return rc; }
    N.k++;
    if (N.k >= N.next_migration && N.world > 1) {
      if ((rc = tn_mig_a(c)) || (rc = tn_mig_b(c)) || (rc = tn_mig_c(c))) return rc;
    }
  }
  return n;
}

int64_t mpmhip_tiled_advance_group(mpmhip_ctx *const *ctxs, int32_t n_ctx, int64_t n) {
  if (!ctxs || n_ctx < 1 || n < 0) return MPMHIP_EINVAL;
  for (int r = 0; r < n_ctx; r++) {
    if (!ctxs[r]) return MPMHIP_EINVAL;
    int rc = tn_check_ready(ctxs[r], "tiled_advance_group");
    if (rc) return rc;
    if (ctxs[r]->tn.wire != MPMHIP_WIRE_LOCAL || ctxs[r]->tn.world != n_ctx || ctxs[r]->T.rank != r)
      return fail(ctxs[r], MPMHIP_EINVAL, "advance_group: ctx %d is not rank %d of a local job of %d", r, r, n_ctx);
    if (ctxs[r]->tn.loop_rccl && ctxs[r]->stream != ctxs[0]->stream)
      return fail(ctxs[r], MPMHIP_EINVAL, "MPMHIP_WIRE_LOCAL_RCCL: the ranks of the job must share one stream (mpmhip_set_stream)");
  }
  // ranks that share ONE stream publish their halo epochs with one launch behind the last rank's pack (k_epoch_signal_group)
  bool one_stream = n_ctx > 1 && n_ctx <= MPMHIP_MAX_HALO_BOXES && !(getenv("MPMHIP_TILE_GROUP_SIGNAL") && atoi(getenv("MPMHIP_TILE_GROUP_SIGNAL")) == 0);
  for (int r = 1; r < n_ctx; r++) one_stream = one_stream && ctxs[r]->stream == ctxs[0]->stream && ctxs[r]->device == ctxs[0]->device;
  bool scan0 = false;
  for (int r = 0; r < n_ctx; r++) scan0 = scan0 || !ctxs[r]->tn.occ_valid;
  if (n > 0 && scan0 && n_ctx > 1) {  // (the planning scan in front of the job's first substep: see mpmhip_tiled_advance)
    int rc = MPMHIP_OK;
    for (int r = 0; r < n_ctx; r++) ctxs[r]->tn.initial_scan = true;
    for (int ph = 0; ph < 3 && !rc; ph++)
      for (int r = 0; r < n_ctx && !rc; r++) {
        mpmhip_ctx *c = ctxs[r];
        if (hipSetDevice(c->device) != hipSuccess) { rc = fail(c, MPMHIP_EHIP, "hipSetDevice failed"); break; }
        rc = ph == 0 ? tn_mig_a(c) : (ph == 1 ? tn_mig_b(c) : tn_mig_c(c));
      }
    for (int r = 0; r < n_ctx; r++) ctxs[r]->tn.initial_scan = false;
    if (rc) return rc;
  }
  for (int64_t i = 0; i < n; i++) {
    // begin of every rank (sort, [boundary] P2G, pack: the peers' boxes are written), then interior + end rank by rank
    for (int r = 0; r < n_ctx; r++) {
      ctxs[r]->tn.defer_signal = one_stream && !ctxs[r]->tn.loop_rccl;  // (only inside this loop: energy / reductions signal per rank)
      ctxs[r]->tn.signal_deferred = false;
      int rc = tn_substep_parts(ctxs[r], 0);
      ctxs[r]->tn.defer_signal = false;
      if (rc) { ctxs[r]->in_substep = false; ctxs[r]->cur_ev = nullptr; return rc; }
    }
    {
      SignalGroup G;
      memset(&G, 0, sizeof G);
      bool any = false;
      for (int r = 0; r < n_ctx; r++) {
        auto &N = ctxs[r]->tn;
        if (!N.signal_deferred) continue;
        N.signal_deferred = false;
        G.boxes[r] = ctxs[r]->d_boxes_cur; G.n[r] = (int)N.boxes.size(); G.epoch[r] = N.epoch;
        any = true;
      }
      if (any) {
        hipLaunchKernelGGL(k_epoch_signal_group, dim3(n_ctx), dim3(64), 0, ctxs[0]->stream, G);
        if (int rc = launch_check(ctxs[0], "epoch_signal_group")) return rc;
      }
    }
    for (int r = 0; r < n_ctx; r++)
      for (int part = 1; part < 3; part++) {
        int rc = tn_substep_parts(ctxs[r], part);
        if (rc) { ctxs[r]->in_substep = false; ctxs[r]->cur_ev = nullptr; return rc; }
      }
    bool due = false;
    for (int r = 0; r < n_ctx; r++) { ctxs[r]->tn.k++; due = due || ctxs[r]->tn.k >= ctxs[r]->tn.next_migration; }
    if (due && n_ctx > 1) {
      for (int ph = 0; ph < 3; ph++)
        for (int r = 0; r < n_ctx; r++) {
          mpmhip_ctx *c = ctxs[r];
          HIPCHK(c, hipSetDevice(c->device));
          int rc = ph == 0 ? tn_mig_a(c) : (ph == 1 ? tn_mig_b(c) : tn_mig_c(c));
          if (rc) return rc;
        }
    }
  }
  return n;
}

int mpmhip_tiled_reduce(mpmhip_ctx *c, double *values, int32_t n, int32_t op) {
  if (!c || !values) return MPMHIP_EINVAL;
  int rc = tn_check_ready(c, "tiled_reduce");
  if (rc) return rc;
  if (c->tn.wire == MPMHIP_WIRE_LOCAL && c->tn.world > 1) return fail(c, MPMHIP_EINVAL, "ranks of a local job reduce together: mpmhip_tiled_reduce_group");
  HIPCHK(c, hipSetDevice(c->device));
  if ((rc = tn_reduce_put(c, values, n, op))) return rc;
  return tn_reduce_finish(c, values, n, op);
}

static int tn_check_group(mpmhip_ctx *const *ctxs, int32_t n_ctx, const char *who) {
  if (!ctxs || n_ctx < 1) return MPMHIP_EINVAL;
  for (int r = 0; r < n_ctx; r++) {
    if (!ctxs[r]) return MPMHIP_EINVAL;
    int rc = tn_check_ready(ctxs[r], who);
    if (rc) return rc;
    if (ctxs[r]->tn.wire != MPMHIP_WIRE_LOCAL || ctxs[r]->tn.world != n_ctx || ctxs[r]->T.rank != r)
      return fail(ctxs[r], MPMHIP_EINVAL, "%s: ctx %d is not rank %d of a local job of %d", who, r, r, n_ctx);
  }
  return MPMHIP_OK;
}

int mpmhip_tiled_reduce_group(mpmhip_ctx *const *ctxs, int32_t n_ctx, double *values, int32_t n, int32_t op) {
  int rc = tn_check_group(ctxs, n_ctx, "tiled_reduce_group");
  if (rc || !values) return rc ? rc : MPMHIP_EINVAL;
  for (int r = 0; r < n_ctx; r++) {  // every rank's row on its way, then every rank reads the table
    HIPCHK(ctxs[r], hipSetDevice(ctxs[r]->device));
    if ((rc = tn_reduce_put(ctxs[r], values + (size_t)r * n, n, op))) return rc;
  }
  for (int r = 0; r < n_ctx; r++)
    if ((rc = tn_reduce_finish(ctxs[r], values + (size_t)r * n, n, op))) return rc;
  return MPMHIP_OK;
}

int mpmhip_calculate_energy_group(mpmhip_ctx *const *ctxs, int32_t n_ctx, double *kinetic, double *potential) {
  int rc = tn_check_group(ctxs, n_ctx, "calculate_energy_group");
  if (rc || !kinetic || !potential) return rc ? rc : MPMHIP_EINVAL;
  std::vector<double> e((size_t)n_ctx * 3);
  for (int r = 0; r < n_ctx; r++) {
    mpmhip_ctx *c = ctxs[r];
    HIPCHK(c, hipSetDevice(c->device));
    if (c->in_substep) return fail(c, MPMHIP_EINVAL, "calculate_energy inside a substep");
    if ((rc = energy_begin(c)) || (rc = tn_exchange_start(c))) return rc;
  }
  for (int r = 0; r < n_ctx; r++)
    if ((rc = tn_exchange_wait(ctxs[r])) || (rc = energy_end(ctxs[r], &e[(size_t)r * 3]))) return rc;
  if ((rc = mpmhip_tiled_reduce_group(ctxs, n_ctx, e.data(), 3, MPMHIP_REDUCE_SUM))) return rc;
  *kinetic = e[0];
  *potential = e[1];
  if (e[2] != 0.0)
    return fail(ctxs[0], MPMHIP_ENOTIMPL, "%.0f particles are of a type without potential_energy() (reference: TC_NOT_IMPLEMENTED); "
                "kinetic energy is valid", e[2]);
  return MPMHIP_OK;
}

// {live particles, active blocks, sticky error word (OR), particles migrated out so far} of the whole job
static int tn_totals_local(mpmhip_ctx *c, double v[8]) {
  Counters h;
  HIPCHK(c, hipSetDevice(c->device));
  // (the error word is read as it is: read_counters would turn a set bit into this rank's failure before the others have seen it)
  Counters *pin = reinterpret_cast<Counters *>(c->h_pinned);
  HIPCHK(c, hipMemcpyAsync(pin, c->cnt, sizeof h, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  h = *pin;
  v[0] = (double)(c->n_slots - (int64_t)h.n_dead);  // (= mpmhip_num_particles)
  v[1] = (double)h.n_active; v[2] = (double)c->tn.migrated_out;
  for (int b = 0; b < 5; b++) v[3 + b] = (double)((h.error >> b) & 1u);
  return MPMHIP_OK;
}
static void tn_totals_out(const double v[8], int64_t out[4]) {
  out[0] = (int64_t)v[0]; out[1] = (int64_t)v[1]; out[3] = (int64_t)v[2];
  int64_t err = 0;
  for (int b = 0; b < 5; b++) if (v[3 + b] > 0.0) err |= 1ll << b;
  out[2] = err;
}
int mpmhip_tiled_totals(mpmhip_ctx *c, int64_t out[4]) {
  if (!c || !out) return MPMHIP_EINVAL;
  int rc = tn_check_ready(c, "tiled_totals");
  if (rc) return rc;
  if (c->tn.wire == MPMHIP_WIRE_LOCAL && c->tn.world > 1) return fail(c, MPMHIP_EINVAL, "ranks of a local job: mpmhip_tiled_totals_group");
  double v[8];
  if ((rc = tn_totals_local(c, v)) || (rc = tn_reduce_put(c, v, 8, MPMHIP_REDUCE_SUM)) || (rc = tn_reduce_finish(c, v, 8, MPMHIP_REDUCE_SUM))) return rc;
  tn_totals_out(v, out);
  return MPMHIP_OK;
}
int mpmhip_tiled_totals_group(mpmhip_ctx *const *ctxs, int32_t n_ctx, int64_t out[4]) {
  int rc = tn_check_group(ctxs, n_ctx, "tiled_totals_group");
  if (rc || !out) return rc ? rc : MPMHIP_EINVAL;
  std::vector<double> v((size_t)n_ctx * 8);
  for (int r = 0; r < n_ctx; r++)
    if ((rc = tn_totals_local(ctxs[r], &v[(size_t)r * 8]))) return rc;
  if ((rc = mpmhip_tiled_reduce_group(ctxs, n_ctx, v.data(), 8, MPMHIP_REDUCE_SUM))) return rc;
  tn_totals_out(v.data(), out);
  return MPMHIP_OK;
}

int mpmhip_tiled_state(mpmhip_ctx *c, int64_t out[8]) {
  if (!c || !out) return MPMHIP_EINVAL;
  const auto &N = c->tn;
  out[0] = N.k; out[1] = N.next_migration; out[2] = N.migrated_out; out[3] = N.migrations; out[4] = N.replans;
  out[5] = (int64_t)N.boxes.size(); out[6] = (int64_t)N.total; out[7] = N.on ? (N.loop_rccl ? (int)MPMHIP_WIRE_LOCAL_RCCL : N.wire) : 0;
  return MPMHIP_OK;
}

int32_t mpmhip_tiled_plan(mpmhip_ctx *c, int32_t capacity, mpmhip_halo_box *out) {
  if (!c) return MPMHIP_EINVAL;
  const auto &N = c->tn;
  if (!N.on) return fail(c, MPMHIP_EINVAL, "tiled_plan needs mpmhip_tiled_setup first");
  const int32_t n = (int32_t)N.boxes.size();
  if (!out || capacity < n) return n;
  const bool peer_wire = (N.wire == MPMHIP_WIRE_IPC || N.wire == MPMHIP_WIRE_LOCAL) && !N.loop_rccl;
  for (int32_t i = 0; i < n; i++) {
    const auto &b = N.boxes[i];
    for (int a = 0; a < 3; a++) { out[i].lo[a] = b.lo[a]; out[i].hi[a] = b.hi[a]; }
    out[i].peer = b.peer; out[i].reserved = (int32_t)b.peer_off;
    out[i].send = peer_wire ? nullptr : (void *)(N.send + b.off);
    out[i].recv = N.recv[0] + b.off;
  }
  return n;
}

}  // extern "C"
