// taichi_mpm_amd/csrc/mpmhip.hip — MI355X (gfx950) MLS-MPM time-stepping core: the C ABI (include/mpmhip.h) and
// the host side of the ctx.  The kernels live next to this file, one header per phase:
//   mpm_common.h (records, parameter blocks, Morton keys)   mpm_math.h (3x3 math, constitutive models, level set)
//   k_sort.h  k_p2g.h  k_grid.h  k_g2p.h  k_tiling.h  k_particles.h  k_debug.h
//   k_bgeo.h (.bgeo frame rows)   k_mpm88.h (the 2D dense-grid demo, its own small object)
//
// One substep (reference: MPM<3>::substep, src/mpm.cpp:452-575):
//
//   sort   (index sort; replaces sort_particles_and_populate_grid src/mpm.cpp:770-918 and
//           clear_boundary_particles :582-633 — dead particles simply drop out of the index)
//          [k_build_keys]  key = Morton(block) << 6 | cell-in-block per particle (normally produced by the
//                          previous substep's k_g2p, which knows the new position)
//          k_sort_front   ONE launch, two roles (grids up to 508 nodes per axis; do_sort):
//            block table  byte flags -> active-block bitmap + popcount prefix (dense slot of every active block)
//                         + active-block list, one single-pass chained scan
//            ranks        rank of each particle in its cell on counters indexed by the KEY (one global atomic per run of equal
//                         keys, or — when the particle order has decayed — an LDS hash per 1024 slots and one atomic per distinct
//                         cell), four consecutive slots per thread, one packed (rank, key) word per slot
//                         (larger grids: k_block_table, then k_rank on counters indexed by the block's dense slot)
//          k_cell_table   per-cell counts -> start of every cell / block in the sorted index (single pass); with the owner list
//                         of the grid pass where that pass walks it (small problems, tiled contexts)
//          k_perm_keyed   sorted position -> particle slot (k_perm after k_rank)
//   P2G    k_p2g   one wavefront per active 4x4x4-cell block, ONE LANE PER CELL: register accumulation of the
//                  27x4 node contributions over the cell's particles (records prefetched two particles ahead),
//                  ordered non-atomic float4 merge into the block's 6^3-node LDS tile, tile written out whole
//                                                                                (src/transfer.cpp:467-569)
//   grid   k_grid  sums the <=8 overlapping block tiles of every touched grid block, normalises,
//                  gravity + level-set boundary                                  (src/mpm.cpp:277-372)
//   G2P    k_g2p   6^3 velocity tile in LDS, 27-tap gather, F update, plasticity AND the next substep's stress
//                  from one eigen-solve, advection, next key                     (src/transfer.cpp:837-954)
//
// Data layout (fp32, resident in HBM for the life of the ctx):
//   particles  two arrays of 64-byte records indexed by a stable particle slot (the reference's
//              ParticleAllocator pool index, src/particle_allocator.h:36):
//                RecG {x3, aux, F9, gid, pid, -}   what G2P reads and rewrites
//                RecP {x3, v3, A9, mass}           what P2G reads;  A = stress*(-4 inv_dx dt) + apic_b*(4 m)
//              (src/transfer.cpp:521-522) is produced by G2P, so P2G carries no constitutive work, and
//              apic_b itself goes to a side array (it is only ever consumed through A).
//              Particles never move in memory between substeps: `perm` (sorted position -> slot) is rebuilt
//              every substep and both transfer kernels gather whole 64-byte records through it, one record per
//              lane.  This mirrors the reference's own structure — sorted index array every substep
//              (mpm.cpp:785-807) + physical reorder only every `reorder_interval` substeps (:811-813).
//   blocks     an "active" block = 4x4x4 cells holding >=1 particle; bitmap over the Morton block space +
//              per-word popcount prefix -> dense slot = rank in Morton order, no pass over the whole grid.
//   tiles      float4[216] per active block: its private (4+2)^3-node P2G result.
//   gridv      float4[64] per touched grid block (v.xyz, m); slot = 8a+o of the owning (active block a,
//              corner offset o); `fat_slot` maps Morton block -> slot.
// No float atomics anywhere (LDS float atomics cost ~2 cycles per lane on gfx950; global ones leave the L2).

#include <hip/hip_runtime.h>

#include <immintrin.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include <hipcub/hipcub.hpp>  // one plain device sort (the reference's order of the rigid boundary particles, rigid_api.h)
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only (tiled_api.h): librccl is dlopen'ed, this library does not link it


#ifndef MPM_G2P_MINW
#define MPM_G2P_MINW 2  // __launch_bounds__ waves/SIMD of k_g2p.  (256, 3) states the 168-VGPR budget explicitly but was measured 3 % slower
                        // (profiles/r03_h_ab_refactor.txt: w2 against default); the budget is guarded by tests/test_kernel_budget_cpu.py instead
#endif
#include "mpm_common.h"
#include "k_sort.h"
#include "k_particles.h"
#include "k_async.h"
#include "k_p2g.h"
#include "k_grid.h"
#include "k_tiling.h"
#include "k_g2p.h"
#include "k_g2p_packed.h"
#include "k_rigid_transfer.h"
#include "k_debug.h"
#include "k_bgeo.h"
#include "k_mpm88.h"
#include "k_mpm2d.h"
#include "k_async2d.h"


// ================================================================================================ host side
using namespace mpm;

static thread_local std::string g_create_error;

enum { PH_SORT = 0, PH_P2G = 1, PH_EXCH = 2, PH_GRID = 3, PH_G2P = 4, PH_COUNT = 5 };

struct mpmhip_ctx {
  mpmhip_config cfg;
  Params P;
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  std::string err;
  // particles
  int64_t cap = 0;
  int64_t n_slots = 0;  // slots in use (live + deleted)
  int32_t next_pid = 0;
  RecG *rg = nullptr, *rg2 = nullptr;
  RecP *rp = nullptr, *rp2 = nullptr;
  float *rb = nullptr, *rb2 = nullptr;
  uint32_t *key = nullptr, *rank = nullptr, *perm = nullptr;
  // blocks
  uint32_t NB = 0;
  uint8_t *blk_flag = nullptr;
  uint32_t *bits = nullptr, *wprefix = nullptr, *act_blk = nullptr, *act_start = nullptr;
  uint32_t *cell_cnt = nullptr, *cell_start = nullptr, *fat_slot = nullptr;
  bool sort_keyed = false;
  bool deterministic = false;  // mpmhip_config.deterministic (env MPMHIP_DETERMINISTIC): in-cell order by creation id behind every sort (do_sort)
  uint32_t *cellcnt_key = nullptr;  // [64 NB] cell counters indexed by KEY (Morton block << 6 | cell): the two-launch front of the sort (do_sort); nullptr on grids beyond 2^21 blocks
  uint32_t *nbr = nullptr, *own_list = nullptr;  // k_cell_table -> k_grid: 32-word neighbour row per active block, list of owned (block, candidate) pairs
  FillStats *d_stats = nullptr;  // device address of the pinned page's statistics words (h_pinned + FILL_STATS_WORD): k_cell_table stores there
  int grid_walk = -1;            // walk of the substep's grid pass (k_grid.h): 2 owner list, 0 per block / per (block, candidate) as until round 4,
                                 // -1 by size and tiling (env MPMHIP_GRID_WALK: A/B)
  bool list_valid = false;       // the last sort built neighbour rows + owner list (do_sort -> do_grid)
  int grid_wgs = 0;              // workgroups of the grid pass; 0: from the last sort's owner count (env MPMHIP_GRID_WGS)
  unsigned long long *scan_slots = nullptr;  // [256] k_block_table + [ct_grid] k_cell_table: {epoch, chunk sum}
  uint32_t list_clear_epoch = 0;  // sort epoch at which the scan words of the list form of k_cell_table were last zeroed (do_sort)
  uint32_t sort_epoch = 0, bt_slots = 0, ct_slots = 0;  // scan_slots: [bt_slots] k_block_table | [ct_slots] k_cell_table_plain | [ct_slots] k_cell_table
  uint32_t scan_grid = 256;  // workgroups of the single-pass scan kernels: three eighths of what the device keeps resident (the lowest
                             // of the plain kernels; scan_limit() answers per kernel)
  std::map<const void *, uint32_t> scan_limits;  // kernel -> three eighths of its resident workgroups
  int scan_grid_env = 0;
  uint32_t rank_wgs_cap = 4096u;  // workgroups of the rank role (k_rank / k_sort_front): MPMHIP_RANK_WGS (tuning)
  float4 *tiles = nullptr, *gridv = nullptr, *dense = nullptr;
  Counters *cnt = nullptr;
  std::vector<GroupParams> groups;
  GroupParams *d_groups = nullptr;
  int groups_cap = G2P_LDS_GROUPS;  // k_g2p mirrors the whole table in LDS
  bool sorted = false;        // perm / cell_start describe the current positions
  bool keys_valid = false;    // key[] + block flags describe the current positions (set by k_g2p)
  uint32_t *pidc = nullptr;   // creation id per slot beside key[] (Params::pidc points here while the deterministic mode is on)
  int cell_order_wgs = 24;    // env MPMHIP_CELL_ORDER_WGS: workgroups per CU of k_cell_order_blocks' launch
  int cell_order_form = 1;    // env MPMHIP_CELL_ORDER: 1 k_cell_order_blocks (a wave per block through LDS), 0 k_cell_order (a lane per cell)
  bool pidc_valid = false;    // ... and was written together with the current key[] (every key writer does while Params::pidc is set)
  bool affine_valid = false;  // RecP.A matches (F, aux, apic_b)
  bool b_stale = false;       // discard_apic_b: the side array is behind RecP.A (k_g2p did not write it)
  bool ordered = false;       // the records lie in the order of the last sort (k_g2p wrote them at their sorted positions)
  bool compact = false;       // ... and the live ones occupy exactly [0, cnt->n_sorted): n_slots may shrink to that
  int p2g_wgs = 16384;        // workgroups of k_p2g (env MPMHIP_P2G_WGS)
  int p2g_split = 11;         // tuning knob (env MPMHIP_P2G_SPLIT): 10*NS + PS, see do_p2g
  int g2p_wgs = 0;            // workgroups of k_g2p; 0: by size (env MPMHIP_G2P_WGS pins it)
  int n_cus = 256;            // compute units of the device
  int g2p_packed = -1;        // k_g2p_packed instead of k_g2p: -1 by size (from 2 M slots on; no rigid bodies, no tiling), 0 never, 1 wherever it
                              // applies (env MPMHIP_G2P_PACKED)
  uint32_t *chunk_blk = nullptr;  // per 256 positions of the sorted index: the block holding the first (k_cell_table -> k_g2p_packed)
  int rigid_wgs = 2048;       // workgroups of k_p2g_rigid (one per wave slot of the device), twice those of k_g2p_rigid (env MPMHIP_RIGID_WGS: tuning)
  uint32_t rank_runs_mul = 3; // k_rank takes its LDS-hash path when runs * this > slots (env MPMHIP_RANK_RUNS_MUL: tuning)
  int ct_blocks = 0;          // blocks per chunk of k_cell_table: 0 by size, 16, 32, 64 (env MPMHIP_CT_BLOCKS: tuning)
  int g2p_minw = 12;          // tuning knob (env MPMHIP_G2P_MINW): 10 + __launch_bounds__ waves/SIMD of k_g2p
  int reorder_interval = 0;   // physical reorder every this many substeps (0 = never); env MPMHIP_REORDER_INTERVAL
  float t = 0.0f, request_t = 0.0f;  // `real` accumulators, as in the reference (src/mpm.h:99, mpm.cpp:573)
  int64_t substeps = 0;
  // profiling
  int profiling = 0;  // 0 off, 1 every phase, 2 only G2P, 3 only P2G (two events per substep instead of six),
                      // 4 the parts of a tiled substep: begin / interior / end, nothing recorded INSIDE a part
  struct Ev { hipEvent_t e[PH_COUNT + 1]; bool ov; };  // ov: the substep ran split (boundary / interior)
  std::vector<Ev> ev_pool;
  size_t ev_used = 0;
  double phase_ms[PH_COUNT] = {0, 0, 0, 0, 0};
  int64_t prof_substeps = 0;
  int ev_level = 0;  // level the pooled events were recorded with
  int prof_every = 1;  // levels 2 / 3: bracket the kernel in every prof_every-th substep only (mpmhip_set_profile_sampling)
  Ev *cur_ev = nullptr;  // events of the substep between substep_begin and substep_end
  // tiling
  LevelSetDev LS;
  LevelSetDev *d_LS = nullptr;  // device copy for k_g2p (k_grid takes it by value)
  Tiling T;
  DevBox *d_boxes = nullptr;
  uint32_t *d_counts = nullptr;
  int *d_bounds = nullptr;
  uint32_t *h_pinned = nullptr;  // 64 KiB of pinned host memory for small readbacks (counters, migration table)
  static constexpr int FILL_STATS_WORD = 16368;  // word offset of the block-fill statistics in that page (do_sort, g2p_is_packed)
  double *d_energy = nullptr;
  int counts_cap = 0;
  bool compact_requested = false;
  bool in_substep = false;
  struct AsyncState : AsyncSched {  // the block scheduler (async_sched.h) + what this ctx keeps on the device for it
    bool enabled = false, limits_valid = false;
    uint32_t *d_tab = nullptr, *d_blk_of = nullptr;
    int32_t *d_blk_limits = nullptr, *d_particle_limits = nullptr;
    int64_t blk_of_cap = 0;
    // the resident stepper (async_api.h): the device store of pool / backup containers
    bool resident = false, pending_counters = false;
    bool records_are_view = false;  // the ctx's records are copies of the pools (mpmhip_async_load_pools), not new particles
    double prof_ms[6] = {0, 0, 0, 0, 0, 0};
    uint32_t *h_tab = nullptr;
    size_t h_tab_cap = 0;
    bool profile_sync = false;
    struct Store {
      uint32_t cap = 0, size = 0, live = 0;   // containers allocated / in use incl. freed ones / not freed (as of the last read-back)
      uint32_t size_ub = 0;                   // upper bound of the size now (launch bound: tags behind the size say FREE)
      float4 *g = nullptr, *w = nullptr, *g2 = nullptr, *w2 = nullptr;
      uint32_t *tag = nullptr, *tag2 = nullptr;
      int32_t *id = nullptr, *id2 = nullptr;
      unsigned long long *best = nullptr, *d_scan = nullptr;
      int64_t best_cap = 0;
      uint32_t scan_cap = 0, scan_epoch = 0;
      uint8_t *d_tbl = nullptr;
      uint8_t *h_tbl_pin = nullptr;  // four pinned images of the action table (async_upload_tbl)
      size_t pin_cap = 0;
      uint32_t pin_next = 0;
      uint32_t *d_rank = nullptr;
      AsyncCounters *d_cnt = nullptr;
      int64_t compactions = 0;
    } store;
  } async;
  int64_t host_particle_bytes = 0;  // particle data copied between host and device so far (uploads, downloads, snapshots)
  // CPIC rigid coupling (rigid_api.h): bodies 1.. (0 = background), their boundary particles, the colored distance field
  struct HostRigid {
    mpmhip_rigid_config cfg{};
    float mass = 0.0f, inertia[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, inv_inertia[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int first_sample = 0, n_samples = 0;
    int64_t first_elem = 0, n_elems = 0;
  };
  struct RigidState {
    bool enabled = false;
    std::vector<HostRigid> bodies;
    RigidBodyDev *d_rb = nullptr;
    RigidSample *d_smp = nullptr;
    float *d_elems = nullptr;
    uint32_t n_smp = 0;
    std::vector<RigidSample> h_smp;
    std::vector<int32_t> h_smp_id;  // creation id of every boundary particle (they share the material particles' counter)
    std::vector<float> h_elems;
    CdfDev cdf{};
    BndRec *d_bnd = nullptr;
    uint8_t *d_blk_rigid = nullptr;
    hipStream_t side = nullptr;          // the colour-aware transfer kernels run here, next to the plain ones on the ctx stream
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int concurrent = 7;                  // which pairs run side by side: 1 P2G, 2 G2P, 4 rasterisation | sort (env MPMHIP_RIGID_CONCURRENT; 0: one stream)
    uint32_t *d_rigid_list = nullptr;  // [max_blocks + 1] the flagged blocks as a list; its length is d_counters[CDF_POOLS + 1]
    uint32_t *d_counters = nullptr;  // [0, CDF_POOLS) pages handed out per sub-pool, [CDF_POOLS] cutting_counter
    uint32_t max_pages = 0;
    uint32_t gather_epoch = 0;  // stamps the boundary records of the particles the last gather_cdf visited
    size_t rpage_words = 0;
    float penalty = 0.0f, pushing_force = 20000.0f;  // MPM::initialize defaults, src/mpm.cpp:35,40
    bool ls_collision = false;       // config key rigid_body_levelset_collision (src/mpm.cpp:535-538)
    uint32_t *d_smp_rank = nullptr;  // position of every boundary particle in the reference's sorted particle list (among the boundary particles)
    uint32_t n_ranked = 0, ls_cap = 0;
    unsigned long long *d_ls_keys[2] = {nullptr, nullptr};
    uint32_t *d_ls_vals[2] = {nullptr, nullptr};
    void *d_ls_tmp = nullptr;
    size_t ls_tmp_bytes = 0;
    std::vector<JointDev> joints;    // MPM::articulations, in the order they were added
    JointDev *d_joints = nullptr;
    int joint_iterations = 100;      // 'articulation_iterations' (src/mpm.h:279-280)
  } rigid;
  bool overlap = false;        // mpmhip_set_overlap: split tiled substeps into boundary / interior work
  bool ov_active = false, interior_done = false;  // state of the substep in flight
  const DevBox *d_boxes_cur = nullptr;  // the box table the pack / grid kernels of the substep in flight read
  bool dirichlet = false;      // mpmhip_set_dirichlet
#ifdef MPMHIP_TIMING_BUILD
  unsigned long long *p2g_tlog = nullptr;  // [3 max_blocks]: begin, end (100 MHz wall clock), fullest cell << 32 | particles of the block
#endif
  // the native data plane of a tiled run (tiled_api.h): plan, arena, wire, migration state
  struct TiledNative {
    struct Box { int peer; int lo[3], hi[3]; uint64_t vol, off, peer_off; };  // off / peer_off: float4 nodes into this rank's / the peer's buffers
    struct Peer {  // a rank's arena as THIS process addresses it
      uint32_t *flags = nullptr, *table[2] = {nullptr, nullptr};
      float4 *recv[2] = {nullptr, nullptr}, *inbox = nullptr;
      double *red[2] = {nullptr, nullptr};  // reduction tables (tiled_api.h: tn_reduce_*)
      void *ipc_base = nullptr;  // mapped with hipIpcOpenMemHandle (closed on destroy)
    };
    struct Mig { std::vector<int64_t> counts; std::vector<int> rank_bounds /* [world][6]: lo3, hi3 base cells, before the records moved */; int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}; float speed = 0.0f; int64_t total = 0, n_out = 0, n_in = 0; } mig;
    bool on = false, connected = false, exch_on_side = false;
    bool defer_signal = false, signal_deferred = false;  // local job: the ranks' epochs are published by ONE launch of the group (tiled_api.h)
    bool loop_rccl = false;               // MPMHIP_WIRE_LOCAL_RCCL: a local job whose halo boxes travel by RCCL self-sends (tiled_api.h)
    std::vector<mpmhip_ctx *> local_ctx;  // the ranks of a local job (mpmhip_tiled_connect_local)
    bool wait_merged = false, merge_signal_wait = true;  // IPC wire, no overlap split: signal + wait of a substep in one launch
    int world = 1, wire = 0;
    int clip_lo[3] = {0, 0, 0}, clip_hi[3] = {0, 0, 0};
    std::vector<int> occ;  // [world][6]: every rank's occupancy box (lo3, hi3, nodes): node_box(r) is cut to it (tiled_api.h: plan)
    bool occ_valid = false, initial_scan = false;  // (until the scan in front of the first substep: the global clip box only)
    std::vector<Box> boxes;
    std::vector<int> halo_peers, all_ranks;
    uint64_t total = 0, halo_cap = 0, inbox_cap = 0;
    size_t arena_bytes = 0, table_bytes = 0, recv_bytes = 0, mig_send_cap = 0;
    char *arena = nullptr;
    uint32_t *flags = nullptr, *table[2] = {nullptr, nullptr}, *row = nullptr, *d_done = nullptr;
    float4 *send = nullptr, *recv[2] = {nullptr, nullptr}, *inbox = nullptr, *mig_send = nullptr;
    DevBox *d_boxes[2] = {nullptr, nullptr};
    int *d_halo_idx = nullptr, *d_all_idx = nullptr;
    std::vector<Peer> peers;
    uint32_t epoch = 0, mig_epoch = 0, red_epoch = 0;
    double *d_red = nullptr;  // this rank's row of a reduction (+ room for the all-reduced row)
    unsigned long long timeout_ticks = 2000000000ull;  // of the 100 MHz wall clock
    void *comm = nullptr;  // ncclComm_t
    int comm_rank = 0, comm_world = 1;
    hipStream_t side = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    int64_t k = 0, next_migration = 0, migrated_out = 0, migrations = 0, replans = 0;
    int migrate_interval = 4, adaptive_cap = 64;
  } tn;
};
namespace { void tn_free(mpmhip_ctx *c); void tn_begin_substep(mpmhip_ctx *c); }

static int fail(mpmhip_ctx *c, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_create_error = buf;
  return code;
}

#define HIPCHK(c, call)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) return fail((c), MPMHIP_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

template <typename T>
static hipError_t dmalloc(T **p, size_t count) { return hipMalloc((void **)p, count * sizeof(T)); }

// new array of `count` elements holding the first `keep` elements of *p (the rest zero-filled when `zero`); *p is freed
template <typename T>
static hipError_t regrow(T **p, size_t keep, size_t count, bool zero) {
  T *q = nullptr;
  hipError_t e = hipMalloc((void **)&q, count * sizeof(T));
  if (e != hipSuccess) return e;
  if (zero) e = hipMemset(q, 0, count * sizeof(T));
  if (e == hipSuccess && keep && *p) e = hipMemcpy(q, *p, keep * sizeof(T), hipMemcpyDeviceToDevice);
  if (e != hipSuccess) { (void)hipFree(q); return e; }
  (void)hipFree(*p);
  *p = q;
  return hipSuccess;
}

static int particle_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b < 1) b = 1;
  if (b > 8192) b = 8192;
  return (int)b;
}

static int launch_check(mpmhip_ctx *c, const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(c, MPMHIP_EHIP, "launch of %s failed: %s", what, hipGetErrorString(e));
  return MPMHIP_OK;
}

template <typename K, typename... Args>
static int run_debug(mpmhip_ctx *c, K kernel, Args... args) {
  hipLaunchKernelGGL(kernel, dim3(256), dim3(256), 0, c->stream, args...);
  int rc = launch_check(c, "debug kernel");
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MPMHIP_OK;
}

// ---------------------------------------------------------------------------------------------- .bgeo frames
// Houdini .bgeo v5 as Partio writes it (external/partio/src/io/BGEO.cpp:57-194) with write_partio's attribute table
// (src/visualize.cpp:24-38).  Big-endian throughout.
namespace {
struct BgeoAttr { const char *name; uint16_t count; int32_t houdini_type; };  // 0 float, 1 int, 5 vector
const BgeoAttr BGEO_PLAIN[] = {{"type", 1, 1}, {"index", 1, 1}, {"limit", 3, 1}, {"v", 3, 5}};
const BgeoAttr BGEO_VERBOSE[] = {{"m", 1, 5}, {"boundary_normal", 3, 5}, {"debug", 3, 5}, {"states", 1, 1},
                                 {"boundary_distance", 1, 0}, {"near_boundary", 1, 1}, {"apic_frobenius_norm", 1, 0}};
void put32(std::vector<uint8_t> &o, uint32_t v) { for (int s = 24; s >= 0; s -= 8) o.push_back((uint8_t)(v >> s)); }
void put16(std::vector<uint8_t> &o, uint16_t v) { o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); }
void put_str(std::vector<uint8_t> &o, const char *s) {
  const size_t n = std::strlen(s);
  put16(o, (uint16_t)n);
  o.insert(o.end(), s, s + n);
}
std::vector<uint8_t> bgeo_header(uint32_t n, bool verbose) {
  std::vector<uint8_t> o;
  put32(o, 0x4267656fu);  // "Bgeo"
  o.push_back('V');
  put32(o, 5);
  put32(o, n); put32(o, 1); put32(o, 0);  // points, one primitive, point groups
  put32(o, 0); put32(o, verbose ? 11 : 4); put32(o, 0); put32(o, 1); put32(o, 0);  // prim groups, point / vertex / prim / detail attributes
  auto table = [&](const BgeoAttr *a, int m) {
    for (int i = 0; i < m; i++) {
      put_str(o, a[i].name);
      put16(o, a[i].count);
      put32(o, (uint32_t)a[i].houdini_type);
      for (int k = 0; k < a[i].count; k++) put32(o, 0);  // defaults
    }
  };
  table(BGEO_PLAIN, 4);
  if (verbose) table(BGEO_VERBOSE, 7);
  return o;
}
std::vector<uint8_t> bgeo_prim_attr() {  // the "generator" = "papi" primitive attribute and the primitive's head
  std::vector<uint8_t> o;
  put_str(o, "generator");
  put16(o, 1); put32(o, 4); put32(o, 1);
  put_str(o, "papi");
  put32(o, 0x8000);
  return o;
}
size_t bgeo_bytes(uint32_t n, bool verbose) {
  const size_t w = verbose ? BGEO_W_VERBOSE : BGEO_W_PLAIN;
  return bgeo_header(n, verbose).size() + (size_t)n * w * 4 + bgeo_prim_attr().size() + 4 +
         (size_t)n * (n > (1u << 16) ? 4 : 2) + 4 + 2;
}
}  // namespace

// live particles in ascending creation id (write_partio sorts by id, src/visualize.cpp:39-43) -> slot list
static int bgeo_order(mpmhip_ctx *c, std::vector<uint32_t> &order, std::vector<int32_t> *ids_out = nullptr) {
  order.clear();
  const size_t ns = (size_t)c->n_slots;
  if (!ns) return MPMHIP_OK;
  int32_t *d_ids = nullptr;
  HIPCHK(c, dmalloc(&d_ids, ns));
  hipLaunchKernelGGL(k_bgeo_ids, dim3(particle_grid(c->n_slots)), dim3(256), 0, c->stream, c->P, (const RecG *)c->rg, d_ids);
  std::vector<int32_t> ids(ns);
  hipError_t e = hipMemcpyAsync(ids.data(), d_ids, ns * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d_ids);
  HIPCHK(c, e);
  order.reserve(ns);
  bool ascending = true;
  int32_t last = -1, max_id = -1;
  for (size_t s = 0; s < ns; s++) {
    if (ids[s] < 0) continue;
    order.push_back((uint32_t)s);
    ascending = ascending && ids[s] > last;
    last = ids[s];
    max_id = std::max(max_id, ids[s]);
  }
  auto finish = [&]() { if (ids_out) { ids_out->resize(order.size()); for (size_t j = 0; j < order.size(); j++) (*ids_out)[j] = ids[order[j]]; } return MPMHIP_OK; };
  if (ascending) return finish();  // slots are handed out in creation order: true until the first physical reorder
  if ((size_t)max_id < 16 * order.size() + 1024) {  // ids are unique: a direct table, swept in id order
    std::vector<uint32_t> slot_of(max_id + 1, 0xFFFFFFFFu);
    for (uint32_t s : order) slot_of[ids[s]] = s;
    size_t m = 0;
    for (uint32_t s : slot_of)
      if (s != 0xFFFFFFFFu) order[m++] = s;
  } else {
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return ids[a] < ids[b]; });
  }
  return finish();
}

extern "C" {

static uint32_t scan_limit(mpmhip_ctx *c, const void *kernel);   // (defined with do_sort, below)
static uint32_t scan_resident_set(int n_cus, int per_cu);

uint32_t mpmhip_abi_version(void) { return MPMHIP_ABI_VERSION; }

const char *mpmhip_last_error(const mpmhip_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int mpmhip_create(const mpmhip_config *cfg, mpmhip_ctx **out) {
  if (!cfg || !out) return fail(nullptr, MPMHIP_EINVAL, "null argument");
  *out = nullptr;
  for (int k = 0; k < 3; k++)
    if (cfg->res[k] < 8 || cfg->res[k] > 1000) return fail(nullptr, MPMHIP_EINVAL, "res[%d]=%d outside [8,1000]", k, cfg->res[k]);
  if (!(cfg->dx > 0) || !(cfg->dt >= 0)) return fail(nullptr, MPMHIP_EINVAL, "dx must be > 0 and dt >= 0");
  if (cfg->max_particles <= 0 || cfg->max_particles >= (1ll << 31)) return fail(nullptr, MPMHIP_EINVAL, "max_particles out of range");
  if (cfg->n_planes < 0 || cfg->n_planes > 8) return fail(nullptr, MPMHIP_EINVAL, "n_planes out of range");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, MPMHIP_EHIP, "no HIP device available (libmpmhip has no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, MPMHIP_EINVAL, "device %d of %d", cfg->device, ndev);
  mpmhip_ctx *c = new (std::nothrow) mpmhip_ctx();
  if (!c) return fail(nullptr, MPMHIP_ENOMEM, "host allocation failed");
  c->cfg = *cfg;
  c->device = cfg->device;
  if (const char *e = getenv("MPMHIP_G2P_MINW")) c->g2p_minw = atoi(e);
  if (const char *e = getenv("MPMHIP_G2P_WGS")) c->g2p_wgs = atoi(e) > 0 ? atoi(e) : 0;
  if (const char *e = getenv("MPMHIP_RIGID_CONCURRENT")) c->rigid.concurrent = atoi(e);
  if (const char *e = getenv("MPMHIP_RIGID_WGS")) c->rigid_wgs = atoi(e) > 1 ? atoi(e) : 2048;
  if (const char *e = getenv("MPMHIP_RANK_RUNS_MUL")) c->rank_runs_mul = (uint32_t)atoi(e);
  if (const char *e = getenv("MPMHIP_CT_BLOCKS")) c->ct_blocks = atoi(e);
  if (const char *e = getenv("MPMHIP_P2G_SPLIT")) c->p2g_split = atoi(e);
  if (const char *e = getenv("MPMHIP_G2P_PACKED")) c->g2p_packed = atoi(e);
  if (const char *e = getenv("MPMHIP_P2G_WGS")) c->p2g_wgs = atoi(e) > 0 ? atoi(e) : 16384;
  if (const char *e = getenv("MPMHIP_GRID_WALK")) c->grid_walk = atoi(e);
  if (const char *e = getenv("MPMHIP_GRID_WGS")) c->grid_wgs = atoi(e) > 0 ? atoi(e) : 0;
  c->reorder_interval = cfg->reorder_interval;
  if (const char *e = getenv("MPMHIP_REORDER_INTERVAL")) c->reorder_interval = atoi(e);
  c->deterministic = cfg->deterministic != 0;
  if (const char *e = getenv("MPMHIP_DETERMINISTIC")) c->deterministic = atoi(e) != 0;
  if (const char *e = getenv("MPMHIP_CELL_ORDER")) c->cell_order_form = atoi(e) != 0;
  if (const char *e = getenv("MPMHIP_CELL_ORDER_WGS")) c->cell_order_wgs = std::max(1, atoi(e));
#ifdef MPMHIP_ABLATE_BUILD
  const int ablate = getenv("MPMHIP_ABLATE") ? atoi(getenv("MPMHIP_ABLATE")) : 0;
#else
  const int ablate = 0;  // (the default library has no ablation paths: MPM_ABLATE is the constant false)
#endif
  auto bail = [&](int code) { g_create_error = c->err; mpmhip_destroy(c); return code; };
  if (hipSetDevice(c->device) != hipSuccess) { fail(c, MPMHIP_EHIP, "hipSetDevice failed"); return bail(MPMHIP_EHIP); }
  Params &P = c->P;
  memset(&P, 0, sizeof P);
  memset(&c->T, 0, sizeof c->T);
  int maxnb = 0;
  for (int k = 0; k < 3; k++) {
    P.res[k] = cfg->res[k];
    P.g[k] = cfg->gravity[k];
    int nb = (cfg->res[k] + 1 + BS - 1) / BS + 1;
    if (nb > maxnb) maxnb = nb;
  }
  P.dx = cfg->dx; P.idx = 1.0f / cfg->dx; P.dt = cfg->dt;
  P.particle_gravity = cfg->particle_gravity; P.apic_damping = cfg->apic_damping; P.rpic_damping = cfg->rpic_damping;
  P.clean_boundary = cfg->clean_boundary;
  P.store_b = cfg->discard_apic_b ? 0 : 1;
  P.clamp_pos = cfg->generic_path ? 1 : 0;
  P.ablate = ablate;
  P.test_small_rank = getenv("MPMHIP_TEST_SMALL_RANK") ? atoi(getenv("MPMHIP_TEST_SMALL_RANK")) : 0;
  memset(&c->LS, 0, sizeof c->LS);
  c->LS.particle_collision = cfg->particle_collision;
  P.particle_collision = cfg->particle_collision;
  if (mpmhip_set_levelset(c, cfg->n_planes, &cfg->planes[0][0], cfg->friction) != MPMHIP_OK) return bail(MPMHIP_EINVAL);
  int kbits = 1;
  while ((1 << kbits) < maxnb) kbits++;
  if (kbits > 8) { fail(c, MPMHIP_EINVAL, "grid too large for 32-bit keys"); return bail(MPMHIP_EINVAL); }
  P.kbits = kbits;
  c->NB = 1u << (3 * kbits);
  P.nbw = c->NB / 32u;
  if (P.nbw == 0) P.nbw = 1;
  c->cap = cfg->max_particles;
  int64_t mb = cfg->max_blocks;
  if (mb <= 0) mb = c->cap / 48 + 4096;  // a block of 64 cells at >= ~1 particle/cell on average, plus slack
  if (mb > (int64_t)c->NB) mb = c->NB;
  P.max_blocks = (uint32_t)mb;

  hipError_t e = hipSuccess;
  auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  A(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  c->stream = c->own_stream;
  A(dmalloc(&c->rg, (size_t)c->cap));
  A(dmalloc(&c->rp, (size_t)c->cap));
  A(dmalloc(&c->rb, (size_t)c->cap * BW));
  A(dmalloc(&c->rg2, (size_t)c->cap));  // k_g2p writes the updated records here, at their sorted positions; the two sets swap
  A(dmalloc(&c->rp2, (size_t)c->cap));
  A(dmalloc(&c->rb2, (size_t)c->cap * BW));
  A(dmalloc(&c->key, (size_t)c->cap));
  A(dmalloc(&c->pidc, (size_t)c->cap));
  A(dmalloc(&c->rank, (size_t)c->cap));
  A(dmalloc(&c->perm, (size_t)c->cap));
  A(dmalloc(&c->chunk_blk, (size_t)c->cap / 256 + 2));
  A(dmalloc(&c->bits, (size_t)P.nbw));
  A(dmalloc(&c->blk_flag, (size_t)P.nbw * 32));
  A(dmalloc(&c->wprefix, (size_t)P.nbw));
  A(dmalloc(&c->fat_slot, (size_t)c->NB));
  A(dmalloc(&c->act_blk, (size_t)mb + 1));
  A(dmalloc(&c->act_start, (size_t)mb + 2));
  A(dmalloc(&c->cell_cnt, (size_t)mb * BC));
  A(dmalloc(&c->cell_start, (size_t)mb * BC + 1));
  // key-indexed cell counters (256 B per block of the WHOLE block space: 67 MB at 128^3, 537 MB at 256^3..508^3): with them the ranks
  // need no block table and share a launch with it (k_sort_front); grids of 2^24 blocks (res > 508) keep the four-launch sort
  const int sort_v1 = getenv("MPMHIP_SORT_V1") ? atoi(getenv("MPMHIP_SORT_V1")) : 0;  // (A/B: 1 four launches, no table; 2 four launches, table allocated)
  c->sort_keyed = c->NB <= (1u << 21) && sort_v1 == 0;  // (the table itself is allocated behind every other buffer, below)
  A(dmalloc(&c->nbr, (size_t)mb * 32));
  A(dmalloc(&c->own_list, (size_t)mb * 8));
  c->bt_slots = (P.nbw + 255) / 256;
  c->ct_slots = (uint32_t)(((size_t)mb + 15) / 16 + 1);  // k_cell_table chunks are >= 16 blocks
  const size_t n_slots64 = c->bt_slots + 2 * (size_t)c->ct_slots;
  A(dmalloc(&c->scan_slots, n_slots64));
  A(dmalloc(&c->tiles, (size_t)mb * TN));
  A(dmalloc(&c->gridv, (size_t)mb * 8 * BC));
  A(dmalloc(&c->cnt, 1));
  A(dmalloc(&c->d_LS, 1));
  A(hipHostMalloc((void **)&c->h_pinned, 65536, hipHostMallocDefault));
  if (e == hipSuccess && c->h_pinned) memset(c->h_pinned, 0, 65536);
  if (e == hipSuccess) {
    void *dp = nullptr;
    A(hipHostGetDevicePointer(&dp, c->h_pinned, 0));
    c->d_stats = reinterpret_cast<FillStats *>(reinterpret_cast<uint32_t *>(dp) + mpmhip_ctx::FILL_STATS_WORD);
    if (getenv("MPMHIP_NO_STATS_STORE") && atoi(getenv("MPMHIP_NO_STATS_STORE"))) c->d_stats = nullptr;  // (A/B: the round-4 copy instead)
  }
  A(dmalloc(&c->d_groups, (size_t)c->groups_cap));
  if (e == hipSuccess && c->NB <= (1u << 21) && sort_v1 != 1) {
    // (an optimisation's table, 256 B per block of the whole block space: 67 MB at 128^3, 537 MB from 256^3 to 508^3.  A device that
    // cannot spare it keeps the four-launch sort instead of failing the create — or a later mpmhip_reserve, a tiled arena, the next
    // rank sharing the device: the table is only taken while it is at most an eighth of what is free NOW, behind every other buffer)
    size_t free_b = 0, total_b = 0;
    const size_t table_b = (size_t)c->NB * BC * sizeof(uint32_t);
    const bool room = hipMemGetInfo(&free_b, &total_b) == hipSuccess && table_b <= free_b / 8;
    if (!room || dmalloc(&c->cellcnt_key, (size_t)c->NB * BC) != hipSuccess) {
      (void)hipGetLastError();
      c->cellcnt_key = nullptr;
      c->sort_keyed = false;
    }
  }
  if (e != hipSuccess) {
    fail(c, MPMHIP_ENOMEM, "device allocation failed: %s (max_particles=%lld, max_blocks=%lld)", hipGetErrorString(e),
         (long long)c->cap, (long long)mb);
    return bail(MPMHIP_ENOMEM);
  }
  P.pidc = c->deterministic ? c->pidc : nullptr;
  A(hipMemset(c->bits, 0, sizeof(uint32_t) * P.nbw));
  A(hipMemset(c->blk_flag, 0, (size_t)P.nbw * 32));
  A(hipMemset(c->cell_cnt, 0, sizeof(uint32_t) * (size_t)mb * BC));
  if (c->cellcnt_key) A(hipMemset(c->cellcnt_key, 0, sizeof(uint32_t) * (size_t)c->NB * BC));
  A(hipMemset(c->cell_start, 0, sizeof(uint32_t) * ((size_t)mb * BC + 1)));
  A(hipMemset(c->act_start, 0, sizeof(uint32_t) * ((size_t)mb + 2)));
  A(hipMemset(c->cnt, 0, sizeof(Counters)));
  A(hipMemcpy(c->d_LS, &c->LS, sizeof c->LS, hipMemcpyHostToDevice));
  A(hipMemset(c->scan_slots, 0, sizeof(unsigned long long) * n_slots64));  // epoch 0 is never used
  A(hipMemset(c->fat_slot, 0, sizeof(uint32_t) * (size_t)c->NB));
  A(hipMemset(c->rb, 0, sizeof(float) * (size_t)c->cap * BW));
  {  // the single-pass scans spin on their predecessors: their grids must fit on the device all at once (k_sort.h)
    int cus = 0;
    A(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
    if (cus > 0) c->n_cus = cus;
    if (const char *e = getenv("MPMHIP_SCAN_GRID")) c->scan_grid_env = atoi(e);  // (tuning)
    if (const char *e = getenv("MPMHIP_RANK_WGS")) c->rank_wgs_cap = (uint32_t)std::max(1, atoi(e));  // (tuning)
  }
  A(hipDeviceSynchronize());
  if (e != hipSuccess) { fail(c, MPMHIP_EHIP, "device init failed: %s", hipGetErrorString(e)); return bail(MPMHIP_EHIP); }
  // every chained scan this ctx can launch (k_sort.h; the block-table role of k_sort_front included: its workgroups are the first of
  // their launch): the grid the host will use must lie inside the set the device certainly keeps resident — asked of the occupancy
  // API once, here, instead of being assumed at the first sort.  (The waits are bounded on top of it: k_sort.h.)
  {
#define MPM_CT_ALL(K) (const void *)K<16, false>, (const void *)K<32, false>, (const void *)K<64, false>, (const void *)K<16, true>, (const void *)K<32, true>, (const void *)K<64, true>
    const void *scans[] = {(const void *)k_sort_front, (const void *)k_block_table, MPM_CT_ALL(k_cell_table), MPM_CT_ALL(k_cell_table_plain),
                           (const void *)k_async_compact};
#undef MPM_CT_ALL
    for (const void *k : scans) {
      int per_cu = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 256, 0) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        fail(c, MPMHIP_EHIP, "the occupancy query of a chained-scan kernel failed: its launch cannot be sized safely");
        return bail(MPMHIP_EHIP);
      }
      if (scan_limit(c, k) > scan_resident_set(c->n_cus, per_cu)) {
        fail(c, MPMHIP_EINVAL, "a chained scan would be launched with %u workgroups, the device keeps %u resident (MPMHIP_SCAN_GRID=%d?)",
             scan_limit(c, k), scan_resident_set(c->n_cus, per_cu), c->scan_grid_env);
        return bail(MPMHIP_EINVAL);
      }
    }
  }
  *out = c;
  return MPMHIP_OK;
}

void mpmhip_destroy(mpmhip_ctx *c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->own_stream) hipStreamSynchronize(c->own_stream);
  for (auto &ev : c->ev_pool)
    for (int k = 0; k <= PH_COUNT; k++) hipEventDestroy(ev.e[k]);
  hipFree(c->rg); hipFree(c->rp); hipFree(c->rb); hipFree(c->rg2); hipFree(c->rp2); hipFree(c->rb2);
  hipFree(c->key); hipFree(c->pidc); hipFree(c->rank); hipFree(c->perm); hipFree(c->chunk_blk); hipFree(c->blk_flag); hipFree(c->bits); hipFree(c->wprefix);
  hipFree(c->fat_slot); hipFree(c->act_blk); hipFree(c->act_start); hipFree(c->cell_cnt);
  hipFree(c->cell_start); hipFree(c->cellcnt_key); hipFree(c->nbr); hipFree(c->own_list); hipFree(c->scan_slots); hipFree(c->tiles); hipFree(c->gridv); hipFree(c->dense);
  hipFree(c->async.d_tab); hipFree(c->async.d_blk_of); hipFree(c->async.d_blk_limits); hipFree(c->async.d_particle_limits);
  if (c->async.h_tab) hipHostFree(c->async.h_tab);
  if (c->async.store.h_tbl_pin) hipHostFree(c->async.store.h_tbl_pin);
  { auto &S = c->async.store; hipFree(S.g); hipFree(S.w); hipFree(S.g2); hipFree(S.w2); hipFree(S.tag); hipFree(S.tag2); hipFree(S.id);
    hipFree(S.id2); hipFree(S.best); hipFree(S.d_scan); hipFree(S.d_tbl); hipFree(S.d_rank); hipFree(S.d_cnt); }
  hipFree(c->cnt); hipFree(c->d_groups); hipFree(c->d_boxes); hipFree(c->d_LS); hipFree(c->d_counts); hipFree(c->d_bounds); if (c->h_pinned) hipHostFree(c->h_pinned); hipFree(c->d_energy);
  { auto &R = c->rigid; hipFree(R.d_rb); hipFree(R.d_smp); hipFree(R.d_elems); hipFree(R.cdf.slot); hipFree(R.cdf.page_key); hipFree(R.cdf.mind);
    hipFree(R.cdf.tags); hipFree(R.cdf.rpage); hipFree(R.d_bnd); if (R.side) { hipStreamSynchronize(R.side); hipStreamDestroy(R.side); } if (R.ev_fork) hipEventDestroy(R.ev_fork); if (R.ev_join) hipEventDestroy(R.ev_join);
    hipFree(R.d_blk_rigid); hipFree(R.d_rigid_list); hipFree(R.d_counters); hipFree(R.d_joints);
    hipFree(R.d_smp_rank); hipFree(R.d_ls_keys[0]); hipFree(R.d_ls_keys[1]); hipFree(R.d_ls_vals[0]); hipFree(R.d_ls_vals[1]); hipFree(R.d_ls_tmp); }
  tn_free(c);
  if (c->own_stream) hipStreamDestroy(c->own_stream);
  delete c;
}

int mpmhip_set_stream(mpmhip_ctx *c, void *s) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return MPMHIP_OK;
}

int mpmhip_set_deterministic(mpmhip_ctx *c, int32_t enabled) {
  if (!c) return MPMHIP_EINVAL;
  if (c->in_substep) return fail(c, MPMHIP_EINVAL, "set_deterministic inside a substep");
  c->deterministic = enabled != 0;
  c->P.pidc = c->deterministic ? c->pidc : nullptr;  // (the key writers keep the ids beside the keys from now on; until a G2P has, k_cell_order reads the records)
  c->pidc_valid = false;
  return MPMHIP_OK;
}

int mpmhip_set_dirichlet(mpmhip_ctx *c, int32_t enabled) {
  if (!c) return MPMHIP_EINVAL;
  c->dirichlet = enabled != 0;
  return MPMHIP_OK;
}

int mpmhip_set_levelset_shapes(mpmhip_ctx *c, int32_t n, const mpmhip_shape *shapes, float friction) {
  if (!c || n < 0 || n > MPMHIP_MAX_SHAPES || (n > 0 && !shapes)) return MPMHIP_EINVAL;
  for (int i = 0; i < n; i++) {
    if (shapes[i].type < 0 || shapes[i].type > 2) return fail(c, MPMHIP_EINVAL, "shape %d: unknown type %d", i, shapes[i].type);
    if (shapes[i].type == 1 && !(shapes[i].p[3] > 0)) return fail(c, MPMHIP_EINVAL, "shape %d: sphere radius must be > 0", i);
    if (shapes[i].type == 2)
      for (int k = 0; k < 3; k++)
        if (!(shapes[i].p[k] < shapes[i].p[3 + k])) return fail(c, MPMHIP_EINVAL, "shape %d: cuboid needs lo < hi", i);
  }
  c->LS.n = n;
  c->LS.friction = friction;
  c->LS.dynamic = 0; c->LS.n1 = 0;
  for (int i = 0; i < n; i++) {
    c->LS.s[i].type = shapes[i].type;
    c->LS.s[i].inside_out = shapes[i].inside_out;
    for (int k = 0; k < 6; k++) c->LS.s[i].p[k] = shapes[i].p[k];
  }
  if (c->d_LS) {  // (not yet allocated while mpmhip_create installs the config's planes: create uploads it)
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(c->d_LS, &c->LS, sizeof c->LS, hipMemcpyHostToDevice));
  }
  return MPMHIP_OK;
}

static int check_shapes(mpmhip_ctx *c, int32_t n, const mpmhip_shape *shapes) {
  if (n < 0 || n > MPMHIP_MAX_SHAPES || (n > 0 && !shapes)) return fail(c, MPMHIP_EINVAL, "between 0 and %d level-set shapes", MPMHIP_MAX_SHAPES);
  for (int i = 0; i < n; i++) {
    if (shapes[i].type < 0 || shapes[i].type > 2) return fail(c, MPMHIP_EINVAL, "shape %d: unknown type %d", i, shapes[i].type);
    if (shapes[i].type == 1 && !(shapes[i].p[3] > 0)) return fail(c, MPMHIP_EINVAL, "shape %d: sphere radius must be > 0", i);
    if (shapes[i].type == 2)
      for (int k = 0; k < 3; k++)
        if (!(shapes[i].p[k] < shapes[i].p[3 + k])) return fail(c, MPMHIP_EINVAL, "shape %d: cuboid needs lo < hi", i);
  }
  return MPMHIP_OK;
}

int mpmhip_set_levelset_keyframes(mpmhip_ctx *c, float t0, float t1, int32_t n0, const mpmhip_shape *shapes0, int32_t n1,
                                  const mpmhip_shape *shapes1, float friction) {
  if (!c) return MPMHIP_EINVAL;
  if (!(t1 > t0)) return fail(c, MPMHIP_EINVAL, "key frame times must satisfy t0 < t1");
  if (int rc = check_shapes(c, n0, shapes0)) return rc;
  if (int rc = check_shapes(c, n1, shapes1)) return rc;
  if (int rc = mpmhip_set_levelset_shapes(c, n0, shapes0, friction)) return rc;
  c->LS.dynamic = 1; c->LS.n1 = n1; c->LS.t0 = t0; c->LS.t1 = t1;
  for (int i = 0; i < n1; i++) {
    c->LS.s1[i].type = shapes1[i].type;
    c->LS.s1[i].inside_out = shapes1[i].inside_out;
    for (int k = 0; k < 6; k++) c->LS.s1[i].p[k] = shapes1[i].p[k];
  }
  HIPCHK(c, hipMemcpy(c->d_LS, &c->LS, sizeof c->LS, hipMemcpyHostToDevice));
  return MPMHIP_OK;
}

int mpmhip_set_levelset(mpmhip_ctx *c, int32_t n_planes, const float *planes, float friction) {
  if (!c || n_planes < 0 || n_planes > 8 || (n_planes > 0 && !planes)) return MPMHIP_EINVAL;
  mpmhip_shape sh[8];
  memset(sh, 0, sizeof sh);
  for (int i = 0; i < n_planes; i++)
    for (int k = 0; k < 4; k++) sh[i].p[k] = planes[4 * i + k];
  return mpmhip_set_levelset_shapes(c, n_planes, sh, friction);
}

int mpmhip_add_group(mpmhip_ctx *c, int32_t material, const float params[MPMHIP_NPARAM]) {
  if (!c || !params) return MPMHIP_EINVAL;
  if (material < MPMHIP_VISCO || material > MPMHIP_ELASTIC) return fail(c, MPMHIP_EINVAL, "unknown material id %d", material);
  if ((int)c->groups.size() >= c->groups_cap) return fail(c, MPMHIP_ECAPACITY, "too many particle groups (max %d)", c->groups_cap);
  if (!(params[0] > 0) || !(params[1] > 0)) return fail(c, MPMHIP_EINVAL, "group mass and vol must be > 0");
  GroupParams g;
  memset(&g, 0, sizeof g);
  memcpy(g.p, params, sizeof g.p);
  g.type = material;
  c->groups.push_back(g);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(c->d_groups, c->groups.data(), sizeof(GroupParams) * c->groups.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return (int)c->groups.size() - 1;
}

// The particle positions changed behind k_g2p's back (uploads, new particles, deletions): key[] and the block flags
// G2P wrote for the OLD positions must not leak into the next sort — k_build_keys only ORs new flags on top, so stale
// ones would turn into phantom active blocks (empty tiles, inflated n_active, spurious capacity errors).
static int invalidate_keys(mpmhip_ctx *c) {
  c->sorted = false;
  c->ordered = c->compact = false;
  c->pidc_valid = false;
  if (c->keys_valid) {
    c->keys_valid = false;
    HIPCHK(c, hipMemsetAsync(c->blk_flag, 0, (size_t)c->P.nbw * 32, c->stream));
  }
  return MPMHIP_OK;
}

// synchronise and read the device counters; reports the sticky capacity error
static int read_counters(mpmhip_ctx *c, Counters &h) {
  // through the ctx's pinned page: an "async" copy into pageable memory is staged and costs ~50 us more
  Counters *pin = reinterpret_cast<Counters *>(c->h_pinned);
  HIPCHK(c, hipMemcpyAsync(pin, c->cnt, sizeof h, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  h = *pin;
  if (c->compact && !c->sorted && !c->in_substep && (int64_t)h.n_sorted < c->n_slots) {
    // k_g2p left the live records in [0, n_sorted): the slots behind are all dead (particles deleted in earlier substeps)
    const uint32_t tail = (uint32_t)(c->n_slots - (int64_t)h.n_sorted);
    h.n_dead = h.n_dead >= tail ? h.n_dead - tail : 0u;
    c->n_slots = h.n_sorted;
    c->P.n_slots = h.n_sorted;
    HIPCHK(c, hipMemcpy(&c->cnt->n_dead, &h.n_dead, sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  if (h.error & 1u)
    return fail(c, MPMHIP_ECAPACITY, "active blocks (%u) exceed max_blocks (%u): recreate the ctx with a larger max_blocks",
                h.n_active, c->P.max_blocks);
  if (h.error & 2u)
    return fail(c, MPMHIP_ECAPACITY, "a particle moved more than margin=%d cells outside this rank's brick between "
                "two migrations: migrate more often or raise the margin", c->T.margin);
  if (h.error & 16u)
    return fail(c, MPMHIP_EHIP, "tiled run: a peer rank's halo / migration epoch did not arrive within the wait limit "
                "(MPMHIP_TILE_WAIT_S): a rank is missing, stalled or out of step");
  if (h.error & SCAN_ERROR_BIT)
    return fail(c, MPMHIP_EHIP, "sort: a chained scan waited %.0f s for a chunk that never published its sum (k_sort.h): the launch did not "
                "fit the device's resident set (another runtime, a partitioned device, a debugger?) — lower MPMHIP_SCAN_GRID",
                (double)SCAN_WAIT_TICKS / 1e8);
  if (h.error & 4u)
    return fail(c, MPMHIP_ECAPACITY, "the colored distance field of the rigid bodies needs more than %u pages of 4^3 nodes: "
                "recreate the ctx with a larger max_blocks", c->rigid.max_pages);
  return MPMHIP_OK;
}

// discard_apic_b: bring the apic_b side array up to date from RecP.A before anybody reads it or invalidates A
static int ensure_b_current(mpmhip_ctx *c) {
  if (!c->b_stale) return MPMHIP_OK;
  if (!c->affine_valid) return fail(c, MPMHIP_EINVAL, "internal: apic_b is stale and the affine matrices are invalid");
  hipLaunchKernelGGL(k_recover_b, dim3(particle_grid(c->n_slots)), dim3(256), 0, c->stream, c->P, (const RecG *)c->rg,
                     (const RecP *)c->rp, c->rb, (const GroupParams *)c->d_groups);
  int rc = launch_check(c, "recover_b");
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->b_stale = false;
  return MPMHIP_OK;
}

static int async_drop_view(mpmhip_ctx *c);
int mpmhip_add_particles(mpmhip_ctx *c, int32_t group, int64_t n, const float *x, const float *v, const float *F,
                         const float *B, const float *aux) {
  if (!c || n < 0 || (n > 0 && !x)) return MPMHIP_EINVAL;
  if (group < 0 || group >= (int)c->groups.size()) return fail(c, MPMHIP_EINVAL, "unknown group %d", group);
  if (n == 0) return MPMHIP_OK;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (int rc = async_drop_view(c)) return rc;  // (resident async stepper: records that only mirror the pools go first)
  if (c->n_slots + n > c->cap)
    return fail(c, MPMHIP_ECAPACITY, "particle capacity exceeded: %lld + %lld > %lld", (long long)c->n_slots, (long long)n, (long long)c->cap);
  if (int rc = ensure_b_current(c)) return rc;  // A of every particle is recomputed from apic_b below
  const int mat = c->groups[group].type;
  // Jp = 1 (:204), j = 1 (:460), logJp = 0 (:595), visco_tau = 1000 (:65)
  const float aux0 = (mat == MPMHIP_SNOW || mat == MPMHIP_WATER) ? 1.0f : (mat == MPMHIP_VISCO ? 1000.0f : 0.0f);
  std::vector<RecG> hg((size_t)n);
  std::vector<RecP> hp((size_t)n);
  std::vector<float> hb((size_t)n * BW, 0.0f);
  for (int64_t i = 0; i < n; i++) {
    RecG &g = hg[i];
    RecP &p = hp[i];
    for (int k = 0; k < 3; k++) {
      g.x[k] = p.x[k] = x[3 * i + k];
      p.v[k] = v ? v[3 * i + k] : 0.0f;
    }
    for (int k = 0; k < 9; k++) {
      g.F[k] = F ? F[9 * i + k] : ((k % 4 == 0) ? 1.0f : 0.0f);
      p.A[k] = 0.0f;
      hb[(size_t)i * BW + k] = B ? B[9 * i + k] : 0.0f;
    }
    g.aux = aux ? aux[i] : aux0;
    g.gid = (uint32_t)group;
    p.mass = c->groups[group].p[0];
    g.pid = c->next_pid + (int32_t)i;
    g.pad = 0;
  }
  c->next_pid += (int32_t)n;
  c->host_particle_bytes += (int64_t)n * (sizeof(RecG) + sizeof(RecP) + sizeof(float) * BW);
  HIPCHK(c, hipMemcpy(c->rg + c->n_slots, hg.data(), sizeof(RecG) * n, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->rp + c->n_slots, hp.data(), sizeof(RecP) * n, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->rb + (size_t)c->n_slots * BW, hb.data(), sizeof(float) * n * BW, hipMemcpyHostToDevice));
  c->n_slots += n;
  c->P.n_slots = (uint32_t)c->n_slots;
  c->affine_valid = false;
  return invalidate_keys(c);
}

// drop every particle (groups, level set and configuration stay): the asynchronous stepper reloads the working set of
// every advance (AsyncMPM<dim>::advance gathers block pools into `particles`, src/async/async_mpm.cpp:255-318)
int mpmhip_clear_particles(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->in_substep) return fail(c, MPMHIP_EINVAL, "clear_particles inside a substep");
  c->n_slots = 0; c->P.n_slots = 0; c->next_pid = 0;
  c->affine_valid = false; c->b_stale = false;
  const uint32_t zero = 0;
  HIPCHK(c, hipMemcpy(&c->cnt->n_dead, &zero, sizeof zero, hipMemcpyHostToDevice));
  c->keys_valid = true;  // force the flag array to be cleared
  return invalidate_keys(c);
}
// base_delta_t / current_t of the next substeps (AsyncMPM<dim>::step sets both per advance, src/async/async_mpm.cpp:405-408)
int mpmhip_set_dt(mpmhip_ctx *c, float dt) {
  if (!c || !(dt >= 0)) return MPMHIP_EINVAL;
  if (c->in_substep) return fail(c, MPMHIP_EINVAL, "set_dt inside a substep");
  if (int rc = ensure_b_current(c)) return rc;  // the stored P2G matrices carry the old dt: they are rebuilt from apic_b
  c->P.dt = dt;
  c->affine_valid = false;
  return MPMHIP_OK;
}
int mpmhip_set_time(mpmhip_ctx *c, double t) {
  if (!c) return MPMHIP_EINVAL;
  c->t = (float)t; c->request_t = (float)t;
  return MPMHIP_OK;
}
// the three clocks of a ctx (current_t, the request_t accumulator of step(), the substep counter that phases the physical
// reorder): read / restored when a host layer re-creates a ctx (e.g. to grow it) so that the run continues unchanged
int mpmhip_get_clock(const mpmhip_ctx *c, double *t, double *request_t, int64_t *substeps) {
  if (!c) return MPMHIP_EINVAL;
  if (t) *t = c->t;
  if (request_t) *request_t = c->request_t;
  if (substeps) *substeps = c->substeps;
  return MPMHIP_OK;
}
int mpmhip_set_clock(mpmhip_ctx *c, double t, double request_t, int64_t substeps) {
  if (!c || substeps < 0) return MPMHIP_EINVAL;
  c->t = (float)t; c->request_t = (float)request_t; c->substeps = substeps;
  return MPMHIP_OK;
}

int64_t mpmhip_num_particles(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  if (hipSetDevice(c->device) != hipSuccess) return MPMHIP_EHIP;
  Counters h;
  int rc = read_counters(c, h);
  return rc ? rc : c->n_slots - (int64_t)h.n_dead;
}

// host mirrors of the record arrays (download / upload are not on the hot path)
static int fetch_records(mpmhip_ctx *c, std::vector<RecG> &hg, std::vector<RecP> *hp, std::vector<float> *hb) {
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t n = (size_t)c->n_slots;
  c->host_particle_bytes += (int64_t)n * (sizeof(RecG) + (hp ? sizeof(RecP) : 0) + (hb ? sizeof(float) * BW : 0));
  hg.resize(n);
  if (n) HIPCHK(c, hipMemcpy(hg.data(), c->rg, sizeof(RecG) * n, hipMemcpyDeviceToHost));
  if (hp) { hp->resize(n); if (n) HIPCHK(c, hipMemcpy(hp->data(), c->rp, sizeof(RecP) * n, hipMemcpyDeviceToHost)); }
  if (hb) { hb->resize(n * BW); if (n) HIPCHK(c, hipMemcpy(hb->data(), c->rb, sizeof(float) * n * BW, hipMemcpyDeviceToHost)); }
  return MPMHIP_OK;
}

int mpmhip_download(mpmhip_ctx *c, int32_t field, void *dst, int64_t n_capacity) {
  if (!c || !dst) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (field < MPMHIP_F_X || field > MPMHIP_F_STATES) return fail(c, MPMHIP_EINVAL, "unknown field %d", field);
  if (field == MPMHIP_F_B)
    if (int rc = ensure_b_current(c)) return rc;
  std::vector<RecG> hg;
  std::vector<RecP> hp;
  std::vector<float> hb;
  int rc = fetch_records(c, hg, field == MPMHIP_F_V ? &hp : nullptr, field == MPMHIP_F_B ? &hb : nullptr);
  if (rc) return rc;
  int64_t m = 0;
  for (size_t i = 0; i < hg.size(); i++) {  // live slots, in slot order
    if (hg[i].pid < 0) continue;
    if (m >= n_capacity) return fail(c, MPMHIP_ECAPACITY, "download buffer holds %lld particles, more are alive", (long long)n_capacity);
    float *f = (float *)dst;
    int32_t *q = (int32_t *)dst;
    switch (field) {
      case MPMHIP_F_X: for (int k = 0; k < 3; k++) f[3 * m + k] = hg[i].x[k]; break;
      case MPMHIP_F_V: for (int k = 0; k < 3; k++) f[3 * m + k] = hp[i].v[k]; break;
      case MPMHIP_F_B: for (int k = 0; k < 9; k++) f[9 * m + k] = hb[i * BW + k]; break;
      case MPMHIP_F_F: for (int k = 0; k < 9; k++) f[9 * m + k] = hg[i].F[k]; break;
      case MPMHIP_F_AUX: f[m] = hg[i].aux; break;
      case MPMHIP_F_GID: q[m] = (int32_t)hg[i].gid; break;
      case MPMHIP_F_ID: q[m] = hg[i].pid; break;
      case MPMHIP_F_STATES: q[m] = (int32_t)hg[i].pad; break;
    }
    m++;
  }
  return (int)m;
}

int mpmhip_upload(mpmhip_ctx *c, int32_t field, const void *src, int64_t n) {
  if (!c || !src) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (field < MPMHIP_F_X || field > MPMHIP_F_STATES || field == MPMHIP_F_GID)
    return fail(c, MPMHIP_EINVAL, "field %d cannot be uploaded", field);
  std::vector<RecG> hg;
  std::vector<RecP> hp;
  std::vector<float> hb;
  int rc = ensure_b_current(c);
  if (rc) return rc;
  if ((rc = fetch_records(c, hg, &hp, &hb))) return rc;
  int64_t live = 0;
  for (auto &g : hg) live += g.pid >= 0;
  if (n != live) return fail(c, MPMHIP_EINVAL, "upload of %lld records but the ctx holds %lld particles", (long long)n, (long long)live);
  const float *f = (const float *)src;
  const int32_t *q = (const int32_t *)src;
  int64_t m = 0;
  for (size_t i = 0; i < hg.size(); i++) {
    if (hg[i].pid < 0) continue;
    switch (field) {
      case MPMHIP_F_X: for (int k = 0; k < 3; k++) hg[i].x[k] = hp[i].x[k] = f[3 * m + k]; break;
      case MPMHIP_F_V: for (int k = 0; k < 3; k++) hp[i].v[k] = f[3 * m + k]; break;
      case MPMHIP_F_B: for (int k = 0; k < 9; k++) hb[i * BW + k] = f[9 * m + k]; break;
      case MPMHIP_F_F: for (int k = 0; k < 9; k++) hg[i].F[k] = f[9 * m + k]; break;
      case MPMHIP_F_AUX: hg[i].aux = f[m]; break;
      case MPMHIP_F_STATES: hg[i].pad = (uint32_t)q[m]; break;
      case MPMHIP_F_ID:
        hg[i].pid = q[m] < 0 ? 0 : q[m];
        if (hg[i].pid + 1 > c->next_pid) c->next_pid = hg[i].pid + 1;
        break;
    }
    m++;
  }
  const size_t ns = hg.size();
  c->host_particle_bytes += (int64_t)ns * (sizeof(RecG) + sizeof(RecP) + sizeof(float) * BW);
  HIPCHK(c, hipMemcpy(c->rg, hg.data(), sizeof(RecG) * ns, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->rp, hp.data(), sizeof(RecP) * ns, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->rb, hb.data(), sizeof(float) * ns * BW, hipMemcpyHostToDevice));
  if (field == MPMHIP_F_X || field == MPMHIP_F_V)
    if (int rc = invalidate_keys(c)) return rc;
  if (field == MPMHIP_F_B || field == MPMHIP_F_F || field == MPMHIP_F_AUX) c->affine_valid = false;
  return MPMHIP_OK;
}

// ------------------------------------------------------------------------------------------------ phases
static int do_reorder(mpmhip_ctx *c);

static inline bool rigid_active(const mpmhip_ctx *c);
// workgroups a single-pass scan kernel may be launched with: three eighths of what the device keeps resident of THAT kernel (a
// quarter until round 4: after impact C3 has 21 k active blocks = 335 chunks of k_cell_table, and with 256 workgroups 79 of them
// took a second chunk behind their first — sort 98 -> 88 us, profiles/r04_l_scan_grid.txt); the margin is for kernels of a second
// stream (CPIC) beside the scans.  Per kernel since round 5: the list forms of k_cell_table hold fewer workgroups per CU than the
// plain ones, and the lowest of all of them would cost the plain ones their grid.
// The arithmetic of that bound, host-only (tests/test_host_cpu.py drives it through mpmhip_debug_scan_grid).  `per_cu` = what the
// occupancy API answers for the kernel at 256 threads.  That answer can be one workgroup per CU HIGH (MI355X guide: 256-thread blocks
// are admitted up to min(API, 8, ...) per CU, one fewer than the API says at 81..112 SGPRs), so the set that is certainly resident is
// min(per_cu, 8) - 1 per CU (at least 1).  Three eighths of the API's number lies inside it for every per_cu (3/8 p <= p - 1 from
// p = 2 on; p = 1: a third of the CUs), a request from the environment is cut to half the API's number AND to that set.
static uint32_t scan_resident_set(int n_cus, int per_cu) { return (uint32_t)(std::max(1, n_cus) * std::max(1, std::min(per_cu, 8) - 1)); }
static uint32_t scan_grid_for(int n_cus, int per_cu, int env_request) {
  per_cu = std::max(1, per_cu);
  int lim = std::max(1, std::max(1, n_cus) * per_cu * 3 / 8);
  if (env_request > 0) lim = std::max(1, std::min(env_request, std::max(1, n_cus) * per_cu / 2));
  return std::min<uint32_t>((uint32_t)lim, scan_resident_set(n_cus, per_cu));
}
static uint32_t scan_limit(mpmhip_ctx *c, const void *kernel) {
  const void *key = kernel;
  auto it = c->scan_limits.find(key);
  if (it != c->scan_limits.end()) return it->second;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
  const uint32_t lim = scan_grid_for(c->n_cus, per_cu, c->scan_grid_env);
  c->scan_limits[key] = lim;
  return lim;
}
static int do_sort(mpmhip_ctx *c) {
  Params &P = c->P;
  hipStream_t st = c->stream;
  const int pg = particle_grid(c->n_slots);
  if (!c->keys_valid) {
    hipLaunchKernelGGL(k_build_keys, dim3(pg), dim3(256), 0, st, P, c->rg, c->rp, c->cnt, c->key, c->blk_flag);
    c->pidc_valid = P.pidc != nullptr;
  }
  // blocks per chunk of k_cell_table: few blocks -> finer chunks (shorter chains, more workgroups).  16 below 2 M slots, 64 from 6 M on
  // (16 costs 10 us at 8 M: its 1 100 chunks no longer fit the scans' resident grid), 32 in between — a rank of a 2-brick job
  // holds 4 M particles in 8 788 blocks: 17.4 us with 64 (as long as the whole 8 M problem takes: the kernel is a latency chain)
  // With the key-indexed counters (k_sort_front) 16 is the best or within 3 us of it at every size (profiles/r05_s_ct_blocks.txt: a rank
  // of 4 M 47.5 -> 44 us, C3 after impact 89.6 -> 86.5, the lattice 67.3 -> 64.2 where 32 gives 61.5): the plain table of that form keeps
  // 1 100..1 340 chunks resident in one round.
  const bool keyed_ct = c->sort_keyed && c->cellcnt_key != nullptr;
  const int ct = (c->ct_blocks == 16 || c->ct_blocks == 32 || c->ct_blocks == 64) ? c->ct_blocks
                 : (keyed_ct || c->n_slots < (2 << 20) ? 16 : (c->n_slots < (6 << 20) ? 32 : 64));
  const uint32_t bt_chunks = (P.nbw + 255) / 256, ct_chunks = (P.max_blocks + ct - 1) / ct;
  // Owner list of the grid pass (k_sort.h, k_grid.h): below 2 M slots it takes the pass from 17 to 7.5 us (1 M particles) for 2..3 us
  // in k_rank + k_cell_table; a tiled ctx always builds it (the per-block walk with the halo-box code in it thrashes the instruction
  // cache: 31 -> 19 us at 4 M particles per rank); an untiled ctx of 2 M slots and more does not — there the pass is bound by its
  // 180 MB of tile reads either way (30.4 against 30.5 us at 8 M) and the rows cost the sort 8 us (profiles/r05_e_*_census.txt).
  c->list_valid = c->grid_walk == 2 || (c->grid_walk < 0 && (c->T.enabled || c->n_slots < (2 << 20)));
  uint32_t *const nbr = c->list_valid ? c->nbr : nullptr;
  uint32_t epoch = ++c->sort_epoch;
  if ((epoch & 0x7FFFFFu) == 0u) epoch = ++c->sort_epoch;  // (k_cell_table's scan words keep 23 bits of it; 0 = never published)
  // (single-pass scans: never more workgroups than are resident at once, see k_sort.h)
  const uint32_t rank_wgs = std::max(1u, std::min<uint32_t>((P.n_slots + RANK_BATCH - 1) / RANK_BATCH, c->rank_wgs_cap));
  const bool keyed = c->sort_keyed && c->cellcnt_key != nullptr;
  const uint32_t bt_wgs = std::min(bt_chunks, keyed ? scan_limit(c, (const void *)k_sort_front) : scan_limit(c, (const void *)k_block_table));
  if (keyed) {
    // block table and in-cell ranks in ONE launch (k_sort_front): the ranks count into key-indexed counters and need no table
    hipLaunchKernelGGL(k_sort_front, dim3(bt_wgs + rank_wgs), dim3(256), 0, st, P, c->blk_flag, c->bits, c->wprefix, c->act_blk, c->cnt,
                       c->scan_slots, epoch, bt_wgs, c->key, c->rank, c->cellcnt_key);
  } else {
    hipLaunchKernelGGL(k_block_table, dim3(bt_wgs), dim3(256), 0, st, P, c->blk_flag, c->bits, c->wprefix, c->act_blk, c->cnt, c->scan_slots, epoch);
    hipLaunchKernelGGL(k_rank, dim3(rank_wgs), dim3(256), 0, st, P, c->key, c->rank, c->cell_cnt, c->bits, c->wprefix,
                       c->cnt, (const uint32_t *)c->act_blk, nbr);
  }
  uint32_t *const counters = keyed ? c->cellcnt_key : c->cell_cnt;
#define MPM_CT(K, CT) (keyed ? K<CT, true> : K<CT, false>)
  auto ct_list = ct == 16 ? MPM_CT(k_cell_table, 16) : (ct == 32 ? MPM_CT(k_cell_table, 32) : MPM_CT(k_cell_table, 64));
  auto ct_plain = ct == 16 ? MPM_CT(k_cell_table_plain, 16) : (ct == 32 ? MPM_CT(k_cell_table_plain, 32) : MPM_CT(k_cell_table_plain, 64));
  const uint32_t ct_wgs = std::min(ct_chunks, c->list_valid ? scan_limit(c, (const void *)ct_list) : scan_limit(c, (const void *)ct_plain));
  if (c->list_valid && epoch - c->list_clear_epoch >= (1u << 22)) {
    // the list form's scan words keep 23 bits of the epoch, and a word is only rewritten when that form runs with the chunk in use: one
    // left from epoch e would match again at e + 2^23 (8.4 M sorts: reachable in a long run).  Zeroing them every 2^22 sorts keeps every
    // surviving word younger than that; the epoch field 0 is never published.
    HIPCHK(c, hipMemsetAsync(c->scan_slots + c->bt_slots + c->ct_slots, 0, sizeof(unsigned long long) * c->ct_slots, st));
    c->list_clear_epoch = epoch;
  }
  if (c->list_valid)  // (the two forms publish different scan words: each has its own slots)
    hipLaunchKernelGGL(ct_list, dim3(ct_wgs), dim3(256), 0, st, P,
                       c->cnt, counters, c->act_start, c->cell_start, c->scan_slots + c->bt_slots + c->ct_slots, epoch, c->rank_runs_mul,
                       c->chunk_blk, nbr, c->own_list, c->d_stats, (const uint32_t *)c->act_blk, (const uint32_t *)c->bits, (const uint32_t *)c->wprefix);
  else
    hipLaunchKernelGGL(ct_plain, dim3(ct_wgs), dim3(256), 0, st, P,
                       c->cnt, counters, c->act_start, c->cell_start, c->scan_slots + c->bt_slots, epoch, c->rank_runs_mul, c->chunk_blk,
                       c->d_stats, (const uint32_t *)c->act_blk);
#undef MPM_CT
  if (keyed)
    hipLaunchKernelGGL(k_perm_keyed, dim3(pg), dim3(256), 0, st, P, (const Counters *)c->cnt, c->key, c->rank, c->cell_start, c->perm,
                       (const uint32_t *)c->bits, (const uint32_t *)c->wprefix);
  else
    hipLaunchKernelGGL(k_perm, dim3(pg), dim3(256), 0, st, P, (const Counters *)c->cnt, c->key, c->rank, c->cell_start, c->perm);
  if (c->deterministic) {
    // every cell's entries in ascending creation id (k_sort.h: k_cell_order); rank[] is idle until the next sort: the ordered index goes
    // there and then IS the index
    const bool compact = c->pidc_valid && P.pidc;
    if (c->cell_order_form == 0) {  // (A/B: one lane per cell over the whole table)
      const int cg = (int)std::min<uint64_t>(8192u, ((uint64_t)P.max_blocks * BC + 255) / 256);
      hipLaunchKernelGGL((compact ? k_cell_order<true> : k_cell_order<false>), dim3(cg), dim3(256), 0, st, P, (const Counters *)c->cnt,
                         (const uint32_t *)c->cell_start, (const uint32_t *)c->perm, (const float4 *)c->rg, c->rank);
    } else {  // one wave per active block through LDS: 7 workgroups per CU resident (22 KiB each), a few blocks per wave
      const int cg = (int)std::min<uint64_t>((uint64_t)c->n_cus * (uint64_t)c->cell_order_wgs, ((uint64_t)P.max_blocks + 3) / 4);
      hipLaunchKernelGGL((compact ? k_cell_order_blocks<true> : k_cell_order_blocks<false>), dim3(std::max(1, cg)), dim3(256), 0, st, P,
                         (const Counters *)c->cnt, (const uint32_t *)c->cell_start, (const uint32_t *)c->perm, (const float4 *)c->rg, c->rank);
    }
    std::swap(c->perm, c->rank);
  }
  // (k_cell_table's last chunk stores (live particles, active blocks, owner entries) of this sort straight into the pinned page,
  // never waited for: the host picks the G2P walk by how full the blocks are (g2p_is_packed) and sizes the grid pass's launch
  // from numbers that may be a few substeps old — until round 5 a hipMemcpyAsync every 16th sort, i.e. a blit kernel in the loop)
  if (!c->d_stats && (c->sort_epoch & 15u) == 1u) (void)hipMemcpyAsync(c->h_pinned + mpmhip_ctx::FILL_STATS_WORD, c->cnt, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
  c->sorted = true;
  c->keys_valid = false;  // key[] now holds k_rank's packed (rank, cell index) words
  c->pidc_valid = false;
  int rc = launch_check(c, "sort");
  if (rc) return rc;
  // sort_allocator (src/mpm.cpp:752-768, every reorder_interval substeps :811-813): needed here only for records that
  // did not come out of k_g2p (fresh uploads, arrivals of a migration) — k_g2p itself leaves the records in sorted order
  // every substep, so a running simulation never pays for a separate reorder (nor for its host synchronisation)
  // (not for the working sets of the resident asynchronous stepper: they live for ONE substep)
  if ((c->reorder_interval > 0 && !c->ordered && !c->async.resident) || c->compact_requested) {
    c->compact_requested = false;
    return do_reorder(c);
  }
  return MPMHIP_OK;
}

// physical reorder into sorted order + compaction of deleted slots (sort_allocator, src/mpm.cpp:752-768).
// Needs the live count on the host, hence one synchronisation: keep reorder_interval large.
static int do_reorder(mpmhip_ctx *c) {
  const int pg = particle_grid(c->n_slots * 4);
  hipLaunchKernelGGL(k_gather_records, dim3(pg), dim3(256), 0, c->stream, c->cnt, c->perm, (const float4 *)c->rg,
                     (const float4 *)c->rp, (const float4 *)c->rb, (float4 *)c->rg2, (float4 *)c->rp2, (float4 *)c->rb2);
  hipLaunchKernelGGL(k_identity_perm, dim3(particle_grid(c->n_slots)), dim3(256), 0, c->stream, c->cnt, c->perm);
  int rc = launch_check(c, "reorder");
  if (rc) return rc;
  Counters h;
  if ((rc = read_counters(c, h))) return rc;
  std::swap(c->rg, c->rg2); std::swap(c->rp, c->rp2); std::swap(c->rb, c->rb2);
  c->ordered = true;
  c->n_slots = h.n_sorted;
  c->P.n_slots = h.n_sorted;
  const uint32_t zero = 0;
  HIPCHK(c, hipMemcpy(&c->cnt->n_dead, &zero, sizeof zero, hipMemcpyHostToDevice));
  return MPMHIP_OK;
}

static RigidXfer rigid_xfer(mpmhip_ctx *c);
// bit t set = some particle group of the ctx is of material type t
// The plain transfer kernel (every block away from the bodies) and the colour-aware one (the flagged blocks) touch disjoint
// blocks and particles, and each is latency-bound at two waves per SIMD: they run side by side, the colour-aware kernel on a
// second stream that waits for what the ctx stream has enqueued so far (fork) and is waited for before anything else (join).
static int rigid_fork(mpmhip_ctx *c, hipStream_t *s, int which) {
  auto &R = c->rigid;
  *s = c->stream;
  if (!(R.concurrent & which)) return MPMHIP_OK;
  if (!R.side) {
    // the highest priority: the colour-aware kernels are the longer ones of a pair and get their wave slots first
    // (8 M scene: 1.163 -> 1.150 ms per substep against the default priority)
    int lo = 0, hi = 0;  // (numerically lower = higher priority)
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    HIPCHK(c, hipStreamCreateWithPriority(&R.side, hipStreamNonBlocking, hi));
    HIPCHK(c, hipEventCreateWithFlags(&R.ev_fork, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&R.ev_join, hipEventDisableTiming));
  }
  HIPCHK(c, hipEventRecord(R.ev_fork, c->stream));
  HIPCHK(c, hipStreamWaitEvent(R.side, R.ev_fork, 0));
  *s = R.side;
  return MPMHIP_OK;
}
static int rigid_join(mpmhip_ctx *c, hipStream_t s) {
  if (s == c->stream) return MPMHIP_OK;
  HIPCHK(c, hipEventRecord(c->rigid.ev_join, s));
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->rigid.ev_join, 0));
  return MPMHIP_OK;
}
static uint32_t material_mask(const mpmhip_ctx *c) {
  uint32_t mask = 0;
  for (const GroupParams &g : c->groups) mask |= 1u << (g.type & 31);
  return mask;
}
static int do_rigid_apply_tmp(mpmhip_ctx *c);

static int do_p2g(mpmhip_ctx *c, int phase = 0) {
  if (!c->affine_valid) {
    hipLaunchKernelGGL(k_affine, dim3(particle_grid(c->n_slots)), dim3(256), 0, c->stream, c->P, c->rg, c->rp, c->rb,
                       c->d_groups);
    c->affine_valid = true;
  }
  // one wavefront per block (all 27 nodes, all particles) measured fastest at 256^3 / 8 M: 0.187 ms against
  // 0.237 (two waves splitting the nodes) and 0.225 (two waves splitting the particles); knob: 10*NS + PS.  Splitting
  // the particles does not help small problems either (128^3 / 1 M, where only 2 448 waves exist: 34 us with one wave per
  // block, 37 with two, 48 with four)
  auto kern = k_p2g<1, 1, 2>;
  int nt = 64;
#ifdef MPMHIP_TUNING_VARIANTS  // (A/B libraries only, profiles/: the default library carries the one kernel that was kept)
  switch (c->p2g_split) {
    case 21: kern = k_p2g<2, 1, 3>; nt = 128; break;
    case 12: kern = k_p2g<1, 2, 2>; nt = 128; break;
    default: break;
  }
#endif
  const bool rigid = rigid_active(c);
  if (rigid) { kern = k_p2g<1, 1, 2, true>; nt = 64; }
  hipStream_t rs = c->stream;
  if (rigid) { if (int rc = rigid_fork(c, &rs, 1)) return rc; }
  hipLaunchKernelGGL(kern, dim3(c->p2g_wgs), dim3(nt), 0, c->stream, c->P,
                     (const float4 *)c->rp, c->cnt, c->act_blk, c->cell_start, c->perm, c->d_groups, c->tiles, c->T, phase,
                     rigid ? (const uint8_t *)c->rigid.d_blk_rigid : (const uint8_t *)nullptr
#ifdef MPMHIP_TIMING_BUILD
                     , c->p2g_tlog
#endif
                     );
  if (rigid) {  // blocks near a body (block_op_rigid), then RigidBody::apply_tmp_velocity (src/transfer.cpp:578-580)
    auto rk = k_p2g_rigid<MAT_ALL>;
    switch (material_mask(c)) {  // one material in the ctx: the kernel that carries only its calculate_force()
#define MPM_ONE_MATERIAL(t) case 1u << (t): rk = k_p2g_rigid<1u << (t)>; break;
      MPM_ONE_MATERIAL(MPMHIP_VISCO) MPM_ONE_MATERIAL(MPMHIP_SNOW) MPM_ONE_MATERIAL(MPMHIP_LINEAR) MPM_ONE_MATERIAL(MPMHIP_JELLY)
      MPM_ONE_MATERIAL(MPMHIP_WATER) MPM_ONE_MATERIAL(MPMHIP_SAND) MPM_ONE_MATERIAL(MPMHIP_VON_MISES) MPM_ONE_MATERIAL(MPMHIP_ELASTIC)
#undef MPM_ONE_MATERIAL
      default: break;
    }
    hipLaunchKernelGGL(rk, dim3(c->rigid_wgs), dim3(64), 0, rs, c->P, (const float4 *)c->rp, (const float4 *)c->rg, c->cnt,
                       c->act_blk, c->cell_start, c->perm, c->d_groups, c->tiles, rigid_xfer(c));
    if (int rc = rigid_join(c, rs)) return rc;
    if (int rc = do_rigid_apply_tmp(c)) return rc;
  }
  return launch_check(c, "p2g");
}
static int do_grid(mpmhip_ctx *c, int mode, int phase = 0) {
  c->P.t = c->t;  // this->current_t of the substep in flight (src/mpm.cpp:532-533)
  c->LS.dirichlet = c->dirichlet ? 1 : 0;
  // mode 0 (the substep's pass) and mode 4 (energy) walk the owner list when the last sort built one (do_sort: small problems and
  // every tiled ctx): one wave per touched grid block, launched at the size of the list as the last sort reported it (+ 12 %; the
  // walk is a grid-stride loop, so a stale number costs time, never correctness).  Otherwise, and for the dense views: the walks
  // of rounds 1-4 — per block at >= 2 M slots, per (block, candidate) below.
  if ((mode == 0 || mode == 4) && c->list_valid) {
    const volatile FillStats *fs = reinterpret_cast<const volatile FillStats *>(c->h_pinned + mpmhip_ctx::FILL_STATS_WORD);
    uint64_t n_own = fs->n_own;
    if (n_own == 0) n_own = std::min<uint64_t>((uint64_t)c->P.max_blocks * 8u, 32768u);  // (before the first sort has reported)
    int wgs = (int)std::min<uint64_t>(8192u, std::max<uint64_t>(64u, (n_own + n_own / 8 + 3) / 4 + 8));
    if (c->grid_wgs > 0) wgs = c->grid_wgs;
    hipLaunchKernelGGL(mode == 0 ? k_grid_list<0> : k_grid_list<4>, dim3(wgs), dim3(256), 0, c->stream, c->P, c->cnt, (const uint32_t *)c->nbr,
                       (const uint32_t *)c->own_list, c->tiles, c->gridv, c->fat_slot, reinterpret_cast<double *>(c->dense), c->T,
                       c->d_boxes_cur, c->LS, phase);
    return launch_check(c, "grid");
  }
  if (mode == 4 && c->T.n_boxes > 0)
    // (only the owner-list walk knows which rank counts a halo node's kinetic energy — the lowest that holds mass on it; the per-block
    // walk would add every halo node on every rank that holds it)
    return fail(c, MPMHIP_EINVAL, "calculate_energy of a tiled ctx needs the owner-list walk of the grid pass: do not set MPMHIP_GRID_WALK=0");
  const bool per_cand = mode == 0 && c->n_slots < (2 << 20);  // small per-GPU problem: latency-bound, see k_grid.h
  auto kern = mode == 0 ? (per_cand ? k_grid_blocks<0, true> : k_grid_blocks<0, false>)
                        : (mode == 1 ? k_grid_blocks<1, false>
                                     : (mode == 2 ? k_grid_blocks<2, false> : (mode == 3 ? k_grid_blocks<3, false> : k_grid_blocks<4, false>)));
  int wgs = per_cand ? 16384 : 4096;
  if (c->grid_wgs > 0 && mode == 0) wgs = c->grid_wgs;
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, c->stream, c->P, c->cnt, c->act_blk, c->bits, c->wprefix, c->tiles,
                     c->gridv, c->fat_slot, c->dense, c->T, c->d_boxes_cur, c->LS, phase);
  return launch_check(c, "grid");
}
// which G2P kernel the plain blocks of the next substep get (bench.py names the kernel of its roofline after it).  By size and by
// how full the blocks are: k_g2p's chunks stay inside a block, so with 512 particles per block (the lattice the reference's benchmark
// seeds) they are full and it is the faster walk by a few microseconds on most boxes; with 350 per block (the same scene after the
// impact) a third of its lanes idle and the packed walk wins by 30 us.  The numbers come from the sort, a few substeps late.
static bool g2p_is_packed(const mpmhip_ctx *c, int phase) {
  bool packed = c->g2p_packed != 0;
  if (c->g2p_packed < 0) {
    const volatile uint32_t *fill = c->h_pinned + mpmhip_ctx::FILL_STATS_WORD;  // {live particles, active blocks} of a recent sort (0, 0 before the first)
    const uint32_t n_live = fill[0], n_act = fill[1];
    packed = c->n_slots >= (2 << 20) && n_act > 0 && (uint64_t)n_live < (uint64_t)n_act * 448u;
  }
  const uint32_t mask = material_mask(c);
  const bool one_plain_material = mask && !(mask & (mask - 1)) && mask != (1u << MPMHIP_VISCO);
  return packed && one_plain_material && !rigid_active(c) && !c->P.store_b && phase == 0 && !c->T.enabled && c->chunk_blk;
}
static int do_g2p(mpmhip_ctx *c, int phase = 0) {
  c->P.t = c->t;
  const bool sb = c->P.store_b != 0;
  // __launch_bounds__(256, 2) only PERMITS 256 VGPRs; what decides the speed is whether the allocation stays <= 168, i.e.
  // whether THREE workgroups fit a CU's register file (512 per SIMD lane): 168 VGPRs 0.303 ms, 180 VGPRs 0.347 ms on the same
  // box (profiles/r03_b_ab_vgpr.txt).  Forcing the bound (256, 3) makes the ALL-material kernel spill (its scratch reloads
  // wait on the prefetched records: vmcnt is shared and in-order) and costs the others 3 % (profiles/r03_h_ab_refactor.txt),
  // so the budget is met by specialising instead and guarded by tests/test_kernel_budget_cpu.py:
  // the kernel is instantiated per material SET (k_g2p.h: MATS): one material in the whole ctx (the benchmark configurations,
  // most scene scripts) -> the kernel that carries only that material's constitutive code (sand: 2 833 instructions and 153
  // VGPRs against 5 552 / 180 for all eight); no visco group -> the set without it (161 VGPRs: visco, with two eigen-solves
  // and a matrix exponential, is what pushes the full kernel over the budget).
  constexpr uint32_t NO_VISCO = MAT_ALL & ~(1u << MPMHIP_VISCO);
  const uint32_t mask = material_mask(c);
  const bool rigid = rigid_active(c), no_visco = !(mask & (1u << MPMHIP_VISCO));
  auto kern = sb ? (no_visco ? k_g2p<256, MPM_G2P_MINW, true, true, false, NO_VISCO> : k_g2p<256, 2, true, true>)
                 : (no_visco ? k_g2p<256, MPM_G2P_MINW, true, false, false, NO_VISCO> : k_g2p<256, 2, true, false>);
  int nt = 256;
#ifdef MPMHIP_TUNING_VARIANTS  // (A/B libraries only: 3 / 4 waves per SIMD spill, 128-entry chunks measured slower — DESIGN.md §4)
  switch (c->g2p_minw) {  // tuning knob: 10 + waves/SIMD target; 23: 128-entry chunks
    case 13: kern = sb ? k_g2p<256, 3, true, true> : k_g2p<256, 3, true, false>; break;
    case 14: kern = sb ? k_g2p<256, 4, true, true> : k_g2p<256, 4, true, false>; break;
    case 23: kern = sb ? k_g2p<128, 3, true, true> : k_g2p<128, 3, true, false>; nt = 128; break;
    default: break;
  }
#endif
  if (rigid) {
    kern = sb ? (no_visco ? k_g2p<256, MPM_G2P_MINW, true, true, true, NO_VISCO> : k_g2p<256, 2, true, true, true>)
              : (no_visco ? k_g2p<256, MPM_G2P_MINW, true, false, true, NO_VISCO> : k_g2p<256, 2, true, false, true>);
    if (!sb) switch (mask) {
#define MPM_ONE_MATERIAL(t) case 1u << (t): kern = k_g2p<256, MPM_G2P_MINW, true, false, true, 1u << (t)>; break;
      MPM_ONE_MATERIAL(MPMHIP_VISCO) MPM_ONE_MATERIAL(MPMHIP_SNOW) MPM_ONE_MATERIAL(MPMHIP_LINEAR) MPM_ONE_MATERIAL(MPMHIP_JELLY)
      MPM_ONE_MATERIAL(MPMHIP_WATER) MPM_ONE_MATERIAL(MPMHIP_SAND) MPM_ONE_MATERIAL(MPMHIP_VON_MISES) MPM_ONE_MATERIAL(MPMHIP_ELASTIC)
#undef MPM_ONE_MATERIAL
      default: break;
    }
  } else if (!sb) {
    switch (mask) {
#define MPM_ONE_MATERIAL(t) case 1u << (t): kern = k_g2p<256, MPM_G2P_MINW, true, false, false, 1u << (t)>; break;
      MPM_ONE_MATERIAL(MPMHIP_VISCO) MPM_ONE_MATERIAL(MPMHIP_SNOW) MPM_ONE_MATERIAL(MPMHIP_LINEAR) MPM_ONE_MATERIAL(MPMHIP_JELLY)
      MPM_ONE_MATERIAL(MPMHIP_WATER) MPM_ONE_MATERIAL(MPMHIP_SAND) MPM_ONE_MATERIAL(MPMHIP_VON_MISES) MPM_ONE_MATERIAL(MPMHIP_ELASTIC)
#undef MPM_ONE_MATERIAL
      default: break;
    }
  }
  // packed chunks (k_g2p_packed.h): -3.5 us of 303 on the lattice of C3, -14 us of 373 after impact; at 1 M particles +7 us of 50
  // (768 workgroups with a handful of chunks each: the walk's set-up is not amortised) — hence by size
  // (one-material instantiations only: they stay below the 168 VGPRs of three workgroups per CU — 163 to 167; the kernel for
  // mixed materials would have 177, the visco one 183: those scenes keep k_g2p)
  decltype(&k_g2p_packed<256, MPM_G2P_MINW, false, 1u << MPMHIP_SAND>) pk = nullptr;
  if (g2p_is_packed(c, phase)) switch (mask) {
#define MPM_ONE_MATERIAL(t) case 1u << (t): pk = k_g2p_packed<256, MPM_G2P_MINW, false, 1u << (t)>; break;
      MPM_ONE_MATERIAL(MPMHIP_SNOW) MPM_ONE_MATERIAL(MPMHIP_LINEAR) MPM_ONE_MATERIAL(MPMHIP_JELLY) MPM_ONE_MATERIAL(MPMHIP_WATER)
      MPM_ONE_MATERIAL(MPMHIP_SAND) MPM_ONE_MATERIAL(MPMHIP_VON_MISES) MPM_ONE_MATERIAL(MPMHIP_ELASTIC)
#undef MPM_ONE_MATERIAL
      default: break;
    }
  if (pk) {
    // four times the device's resident set (three workgroups per CU): with equal work items what is left of the launch's tail is
    // the partly filled last round — 4 096 workgroups are 5.33 rounds of 768.  At C3, lattice / after impact: 3 072 -> 287 / 336 us,
    // 4 096 -> 287 / 352, 6 144 -> 291 / 343, 2 304 -> 296 / 346, 1 536 -> 292 / 350, 768 -> 304 / 350 (profiles/r04_u_g2p_wgs.txt)
    const int wgs = c->g2p_wgs > 0 ? c->g2p_wgs : (c->n_slots < (2 << 20) ? 768 : 12 * c->n_cus);
    hipLaunchKernelGGL(pk, dim3(wgs), dim3(256), 0, c->stream, c->P, (const float4 *)c->rg, (float4 *)c->rg2, (float4 *)c->rp2,
                       (float4 *)c->rb2, c->cnt, c->act_blk, c->act_start, c->perm, c->d_groups, c->gridv, c->fat_slot, c->cnt, c->key,
                       c->blk_flag, (const LevelSetDev *)c->d_LS, (const uint32_t *)c->chunk_blk
#ifdef MPMHIP_TIMING_BUILD
                       , c->p2g_tlog
#endif
                       );
    c->sorted = false; c->keys_valid = true; c->affine_valid = true;
    c->pidc_valid = c->P.pidc != nullptr;
    if (!c->P.store_b) c->b_stale = true;
    return launch_check(c, "g2p_packed");
  }
  hipStream_t rs = c->stream;
  if (rigid) { if (int rc = rigid_fork(c, &rs, 2)) return rc; }
  // 4 096 workgroups at 8 M particles (2 048 / 8 192 measured no better); below ~2 M slots the device's resident set (three
  // workgroups per CU = 768) walking ~6 chunks each WITH the record prefetch beats one chunk per workgroup: 51.9 -> 46.1 us at
  // 1 M particles (profiles/r04_b_knobs.txt; 512 and 1 024 are slower again)
  const int g2p_wgs = c->g2p_wgs > 0 ? c->g2p_wgs : (c->n_slots < (2 << 20) ? 768 : 4096);
  hipLaunchKernelGGL(kern, dim3(g2p_wgs), dim3(nt), 0, c->stream, c->P, (const float4 *)c->rg, (float4 *)c->rg2, (float4 *)c->rp2,
                     (float4 *)c->rb2, c->cnt, c->act_blk, c->act_start, c->perm, c->d_groups, c->gridv, c->fat_slot, c->cnt, c->key,
                     c->blk_flag, (const LevelSetDev *)c->d_LS, phase_box(c->T), phase);
  if (rigid) {
    auto rk = k_g2p_rigid<MAT_ALL>;
    switch (mask) {
#define MPM_ONE_MATERIAL(t) case 1u << (t): rk = k_g2p_rigid<1u << (t)>; break;
      MPM_ONE_MATERIAL(MPMHIP_VISCO) MPM_ONE_MATERIAL(MPMHIP_SNOW) MPM_ONE_MATERIAL(MPMHIP_LINEAR) MPM_ONE_MATERIAL(MPMHIP_JELLY)
      MPM_ONE_MATERIAL(MPMHIP_WATER) MPM_ONE_MATERIAL(MPMHIP_SAND) MPM_ONE_MATERIAL(MPMHIP_VON_MISES) MPM_ONE_MATERIAL(MPMHIP_ELASTIC)
#undef MPM_ONE_MATERIAL
      default: break;
    }
    hipLaunchKernelGGL(rk, dim3(c->rigid_wgs / 2), dim3(256), 0, rs, c->P, (const float4 *)c->rg, (float4 *)c->rg2, (float4 *)c->rp2,
                       (float4 *)c->rb2, c->cnt, c->act_blk, c->act_start, c->perm, c->d_groups, c->gridv, c->fat_slot, c->cnt, c->key,
                       c->blk_flag, (const LevelSetDev *)c->d_LS, rigid_xfer(c));
    if (int rc = rigid_join(c, rs)) return rc;
    if (int rc = do_rigid_apply_tmp(c)) return rc;
  }
  c->sorted = false;       // positions moved
  c->keys_valid = true;    // ... and their keys / block flags are ready for the next sort
  c->pidc_valid = c->P.pidc != nullptr;
  c->affine_valid = true;  // A was produced together with F
  if (!c->P.store_b) c->b_stale = true;
  return launch_check(c, "g2p");
}

// behind the LAST k_g2p launch of a substep: the buffers it wrote become the current records
static void swap_records(mpmhip_ctx *c) {
  std::swap(c->rg, c->rg2); std::swap(c->rp, c->rp2); std::swap(c->rb, c->rb2);
  c->ordered = c->compact = true;
}

static int need_sorted(mpmhip_ctx *c, const char *who) {
  if (!c->sorted) return fail(c, MPMHIP_EINVAL, "%s needs sorted particles: call mpmhip_sort first", who);
  return MPMHIP_OK;
}

#include "rigid_api.h"

int mpmhip_sort(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  return do_sort(c);
}
int mpmhip_p2g(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = need_sorted(c, "p2g");
  return rc ? rc : do_p2g(c);
}
int mpmhip_grid_update(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = need_sorted(c, "grid_update");
  return rc ? rc : do_grid(c, 0);
}
int mpmhip_g2p(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = need_sorted(c, "g2p");
  if (rc || (rc = do_g2p(c))) return rc;
  swap_records(c);
  return MPMHIP_OK;
}

static int get_events(mpmhip_ctx *c, mpmhip_ctx::Ev **out) {
  if (c->ev_used == c->ev_pool.size()) {
    mpmhip_ctx::Ev ev;
    for (int k = 0; k <= PH_COUNT; k++) HIPCHK(c, hipEventCreate(&ev.e[k]));
    c->ev_pool.push_back(ev);
  }
  *out = &c->ev_pool[c->ev_used++];
  return MPMHIP_OK;
}

static int collect_events(mpmhip_ctx *c) {
  if (c->ev_used == 0) return MPMHIP_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (size_t i = 0; i < c->ev_used; i++) {
    if (c->ev_level == 4) {  // begin part -> "p2g", interior part (split substeps) -> "grid", end part -> "g2p"
      float ms = 0;
      HIPCHK(c, hipEventElapsedTime(&ms, c->ev_pool[i].e[0], c->ev_pool[i].e[1]));
      c->phase_ms[PH_P2G] += ms;
      if (c->ev_pool[i].ov) {
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev_pool[i].e[2], c->ev_pool[i].e[3]));
        c->phase_ms[PH_GRID] += ms;
      }
      HIPCHK(c, hipEventElapsedTime(&ms, c->ev_pool[i].e[4], c->ev_pool[i].e[5]));
      c->phase_ms[PH_G2P] += ms;
      c->prof_substeps++;
      continue;
    }
    if (c->ev_pool[i].ov && c->ev_level >= 2) {  // split substep: the interior launch was bracketed separately
      float ms = 0;
      const int a = c->ev_level == 2 ? 0 : 3;
      HIPCHK(c, hipEventElapsedTime(&ms, c->ev_pool[i].e[a], c->ev_pool[i].e[a + 1]));
      c->phase_ms[c->ev_level == 2 ? PH_G2P : PH_P2G] += ms;
    }
    for (int k = 0; k < PH_COUNT; k++) {
      if ((c->ev_level == 2 && k != PH_G2P) || (c->ev_level == 3 && k != PH_P2G)) continue;
      float ms = 0;
      HIPCHK(c, hipEventElapsedTime(&ms, c->ev_pool[i].e[k], c->ev_pool[i].e[k + 1]));
      c->phase_ms[k] += ms;
    }
    c->prof_substeps++;
  }
  c->ev_used = 0;
  return MPMHIP_OK;
}

static int do_halo_pack(mpmhip_ctx *c) {
  if (c->T.n_boxes == 0) return MPMHIP_OK;
  int nb = (int)((c->T.box_blocks + 3) / 4);  // one wave per grid block of a box
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(k_halo_pack, dim3(nb), dim3(256), 0, c->stream, c->P, c->T, c->d_boxes_cur, c->bits, c->wprefix,
                     (const float4 *)c->tiles);
  return launch_check(c, "halo_pack");
}

// A tiled substep is begin [exchange] end, or — with mpmhip_set_overlap — begin [exchange starts] interior
// [exchange done] end: begin runs the sort, the P2G of the blocks that touch a halo box and the halo pack; interior
// runs everything that cannot touch a halo node (P2G, grid, G2P) while the boxes are on the wire; end finishes the
// boundary part with the peers' sums.  With level-1 profiling the split is off so the phase table stays clean.
int mpmhip_substep_begin(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->in_substep) return fail(c, MPMHIP_EINVAL, "substep_begin called twice without substep_end");
  int rc;
  mpmhip_ctx::Ev *ev = nullptr;
  int lvl = c->profiling;
  c->ov_active = c->overlap && c->T.n_boxes > 0 && lvl != 1;
  c->interior_done = false;
  if ((lvl == 2 || lvl == 3) && c->prof_every > 1 && (c->substeps % c->prof_every) != 0) lvl = 0;  // not a sampled substep
  if (lvl) {
    if (c->ev_used >= 4096 && (rc = collect_events(c))) return rc;
    if ((rc = get_events(c, &ev))) return rc;
    ev->ov = c->ov_active;
    if (lvl == 1 || lvl == 4) HIPCHK(c, hipEventRecord(ev->e[0], c->stream));
  }
  // articulate + rasterize_rigid_boundary (src/mpm.cpp:466-472) next to the sort, gather_cdf (:506-508) behind both
  const bool bodies = rigid_active(c);
  hipStream_t rs = c->stream;
  if (bodies && ((rc = rigid_fork(c, &rs, 4)) || (rc = do_rigid_pre_a(c, rs)))) return rc;
  if ((rc = do_sort(c))) return rc;
  if (bodies && ((rc = rigid_join(c, rs)) || (rc = do_rigid_pre_b(c)))) return rc;
  if (ev && (lvl == 1 || lvl == 3)) HIPCHK(c, hipEventRecord(ev->e[1], c->stream));
  if ((rc = do_p2g(c, c->ov_active ? 1 : 0))) return rc;
  if (ev && lvl == 3) HIPCHK(c, hipEventRecord(ev->e[2], c->stream));
  if (c->tn.on) tn_begin_substep(c);  // (native data plane: this substep's epoch and box table)
  if ((rc = do_halo_pack(c))) return rc;
  if (ev && lvl == 1) HIPCHK(c, hipEventRecord(ev->e[2], c->stream));
  if (ev && lvl == 4) HIPCHK(c, hipEventRecord(ev->e[1], c->stream));
  c->cur_ev = ev;
  c->in_substep = true;
  return MPMHIP_OK;
}

int mpmhip_substep_interior(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->in_substep) return fail(c, MPMHIP_EINVAL, "substep_interior without substep_begin");
  if (!c->ov_active || c->interior_done) return MPMHIP_OK;
  int rc;
  mpmhip_ctx::Ev *ev = c->cur_ev;
  const int lvl = c->profiling;
  if (ev && lvl == 3) HIPCHK(c, hipEventRecord(ev->e[3], c->stream));
  if (ev && lvl == 4) HIPCHK(c, hipEventRecord(ev->e[2], c->stream));
  if ((rc = do_p2g(c, 2))) return rc;
  if (ev && lvl == 3) HIPCHK(c, hipEventRecord(ev->e[4], c->stream));
  if ((rc = do_grid(c, 0, 2))) return rc;
  if (ev && lvl == 2) HIPCHK(c, hipEventRecord(ev->e[0], c->stream));
  if ((rc = do_g2p(c, 2))) return rc;
  if (ev && lvl == 2) HIPCHK(c, hipEventRecord(ev->e[1], c->stream));
  if (ev && lvl == 4) HIPCHK(c, hipEventRecord(ev->e[3], c->stream));
  c->interior_done = true;
  return MPMHIP_OK;
}

int mpmhip_substep_end(mpmhip_ctx *c) {  // grid (+ halo sum), G2P
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->in_substep) return fail(c, MPMHIP_EINVAL, "substep_end without substep_begin");
  int rc;
  if (c->ov_active && !c->interior_done && (rc = mpmhip_substep_interior(c))) return rc;  // caller skipped it
  mpmhip_ctx::Ev *ev = c->cur_ev;
  c->cur_ev = nullptr;
  c->in_substep = false;
  const int lvl = c->profiling, ph = c->ov_active ? 1 : 0;
  if (rigid_active(c) && c->rigid.ls_collision && (rc = do_rigid_ls_collision(c))) return rc;  // src/mpm.cpp:535-538
  if (ev && lvl == 1) HIPCHK(c, hipEventRecord(ev->e[3], c->stream));
  if (ev && lvl == 4) HIPCHK(c, hipEventRecord(ev->e[4], c->stream));
  if ((rc = do_grid(c, 0, ph))) return rc;
  if (ev && (lvl == 1 || lvl == 2)) HIPCHK(c, hipEventRecord(ev->e[4], c->stream));
  if ((rc = do_g2p(c, ph))) return rc;
  if (ev && (lvl == 1 || lvl == 2 || lvl == 4)) HIPCHK(c, hipEventRecord(ev->e[5], c->stream));
  swap_records(c);
  if (rigid_active(c) && (rc = do_rigid_advect(c, c->P.dt))) return rc;  // src/mpm.cpp:570-572
  c->t += c->P.dt;  // src/mpm.cpp:573
  c->substeps++;
  return MPMHIP_OK;
}

int mpmhip_set_overlap(mpmhip_ctx *c, int32_t enabled) {
  if (!c) return MPMHIP_EINVAL;
  if (c->in_substep) return fail(c, MPMHIP_EINVAL, "set_overlap inside a substep");
  c->overlap = enabled != 0;
  return MPMHIP_OK;
}

int64_t mpmhip_tiled_run(mpmhip_ctx *c, int64_t n, int64_t until_migration, mpmhip_exchange_fn exchange, void *user) {
  if (!c || n < 0) return MPMHIP_EINVAL;
  if (c->tn.on) return fail(c, MPMHIP_EINVAL, "this ctx has a native plan (mpmhip_tiled_setup): advance it with mpmhip_tiled_advance");
  if (c->T.n_boxes > 0 && !exchange) return fail(c, MPMHIP_EINVAL, "this ctx has halo boxes: tiled_run needs an exchange callback");
  if (until_migration > 0 && until_migration < n) n = until_migration;
  for (int64_t i = 0; i < n; i++) {
    int rc = mpmhip_substep_begin(c);
    if (rc) return rc;
    if (c->T.n_boxes > 0) {
      int32_t e = exchange(user, MPMHIP_EXCHANGE_START);
      if (!e && (rc = mpmhip_substep_interior(c))) e = rc;
      if (!e) e = exchange(user, MPMHIP_EXCHANGE_WAIT);
      if (e) {
        c->in_substep = false; c->cur_ev = nullptr;  // (the substep is abandoned: the ctx can be stepped or destroyed again)
        return e < 0 ? e : fail(c, MPMHIP_EINVAL, "tiled_run: the exchange callback returned %d", (int)e);
      }
    }
    if ((rc = mpmhip_substep_end(c))) return rc;
  }
  return n;
}

int mpmhip_substep(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  if (c->T.n_boxes > 0)
    return fail(c, MPMHIP_EINVAL, "this ctx has halo boxes: drive it with substep_begin / exchange / substep_end");
  int rc = mpmhip_substep_begin(c);
  return rc ? rc : mpmhip_substep_end(c);
}

int mpmhip_run_substeps(mpmhip_ctx *c, int32_t n) {
  for (int32_t i = 0; i < n; i++) {
    int rc = mpmhip_substep(c);
    if (rc) return rc;
  }
  return MPMHIP_OK;
}

int mpmhip_step(mpmhip_ctx *c, float dt) {  // MPM<dim>::step, src/mpm.cpp:428-439
  if (!c) return MPMHIP_EINVAL;
  if (dt < 0) {
    int rc = mpmhip_substep(c);
    c->request_t = c->t;
    return rc;
  }
  c->request_t += dt;
  while (c->t + c->P.dt < c->request_t) {
    int rc = mpmhip_substep(c);
    if (rc) return rc;
  }
  return MPMHIP_OK;
}

double mpmhip_current_time(const mpmhip_ctx *c) { return c ? (double)c->t : 0.0; }

int mpmhip_synchronize(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  Counters h;
  return read_counters(c, h);
}

static int ensure_dense(mpmhip_ctx *c, size_t &nodes) {
  nodes = (size_t)(c->P.res[0] + 1) * (c->P.res[1] + 1) * (c->P.res[2] + 1);
  if (!c->dense) {
    if (dmalloc(&c->dense, nodes) != hipSuccess) return fail(c, MPMHIP_ENOMEM, "dense grid staging allocation failed");
  }
  return MPMHIP_OK;
}

int mpmhip_download_grid(mpmhip_ctx *c, int32_t which, float *dst) {
  if (!c || !dst || (which != 0 && which != 1)) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  size_t nodes;
  int rc = ensure_dense(c, nodes);
  if (rc) return rc;
  HIPCHK(c, hipMemsetAsync(c->dense, 0, nodes * sizeof(float4), c->stream));
  if ((rc = do_grid(c, which == 0 ? 1 : 3))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(dst, c->dense, nodes * sizeof(float4), hipMemcpyDeviceToHost));
  return MPMHIP_OK;
}

int mpmhip_upload_grid(mpmhip_ctx *c, const float *src) {
  if (!c || !src) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = need_sorted(c, "upload_grid");
  if (rc) return rc;
  size_t nodes;
  if ((rc = ensure_dense(c, nodes))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(c->dense, src, nodes * sizeof(float4), hipMemcpyHostToDevice));
  return do_grid(c, 2);
}

// ------------------------------------------------------------------------------------------------ snapshots
// Whole-state save / load (reference: TC_IO serialization of MPM<dim> + ParticleAllocator, src/mpm.h:38-54,134-169,
// general_action "save"/"load" src/mpm.cpp:940-960).  The blob holds the raw records — including the P2G affine
// matrices — so a restart continues exactly where the run stopped (up to the in-cell summation order).
struct SnapHeader {
  char magic[8];  // "MPMHIP01"
  uint32_t abi, n_groups;
  int64_t n_slots, substeps;
  int32_t next_pid, b_stale, store_b, res[3];
  float t, request_t, dx, dt;
  uint32_t n_dead, pad;
};

// scenes with rigid bodies append the bodies' records (pose, velocities, mass properties) and the joints: meshes, boundary
// particles and scripts come from the scene again (the script adds the same bodies before it loads), like the level set
struct SnapRigid {
  char magic[8];  // "MPMRIGID"
  uint32_t n_bodies, n_joints, sizeof_body, sizeof_joint;
};
static size_t snapshot_rigid_bytes(const mpmhip_ctx *c) {
  return rigid_active(c) ? sizeof(SnapRigid) + sizeof(RigidBodyDev) * c->rigid.bodies.size() + sizeof(JointDev) * c->rigid.joints.size() : 0;
}
static size_t snapshot_bytes(const mpmhip_ctx *c) {
  return sizeof(SnapHeader) + sizeof(GroupParams) * c->groups.size() +
         (size_t)c->n_slots * (sizeof(RecG) + sizeof(RecP) + sizeof(float) * BW) + snapshot_rigid_bytes(c);
}

int64_t mpmhip_snapshot_size(mpmhip_ctx *c) { return c ? (int64_t)snapshot_bytes(c) : MPMHIP_EINVAL; }

int mpmhip_snapshot_save(mpmhip_ctx *c, void *dst, size_t cap) {
  if (!c || !dst) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->in_substep) return fail(c, MPMHIP_EINVAL, "snapshot inside a substep");
  if (cap < snapshot_bytes(c)) return fail(c, MPMHIP_ECAPACITY, "snapshot buffer too small: %zu < %zu", cap, snapshot_bytes(c));
  Counters hc;
  int rc = read_counters(c, hc);
  if (rc) return rc;
  if (!c->affine_valid) {  // make A current first (fresh uploads), so that the blob is self-consistent
    hipLaunchKernelGGL(k_affine, dim3(particle_grid(c->n_slots)), dim3(256), 0, c->stream, c->P, c->rg, c->rp, c->rb, c->d_groups);
    c->affine_valid = true;
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  SnapHeader h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, "MPMHIP01", 8);
  h.abi = MPMHIP_ABI_VERSION; h.n_groups = (uint32_t)c->groups.size();
  h.n_slots = c->n_slots; h.substeps = c->substeps; h.next_pid = c->next_pid;
  h.b_stale = c->b_stale; h.store_b = c->P.store_b;
  for (int k = 0; k < 3; k++) h.res[k] = c->P.res[k];
  h.t = c->t; h.request_t = c->request_t; h.dx = c->P.dx; h.dt = c->P.dt; h.n_dead = hc.n_dead;
  char *p = (char *)dst;
  memcpy(p, &h, sizeof h); p += sizeof h;
  memcpy(p, c->groups.data(), sizeof(GroupParams) * c->groups.size()); p += sizeof(GroupParams) * c->groups.size();
  const size_t n = (size_t)c->n_slots;
  if (n) {
    HIPCHK(c, hipMemcpy(p, c->rg, sizeof(RecG) * n, hipMemcpyDeviceToHost)); p += sizeof(RecG) * n;
    HIPCHK(c, hipMemcpy(p, c->rp, sizeof(RecP) * n, hipMemcpyDeviceToHost)); p += sizeof(RecP) * n;
    HIPCHK(c, hipMemcpy(p, c->rb, sizeof(float) * BW * n, hipMemcpyDeviceToHost)); p += sizeof(float) * BW * n;
  }
  if (rigid_active(c)) {
    SnapRigid r;
    memcpy(r.magic, "MPMRIGID", 8);
    r.n_bodies = (uint32_t)c->rigid.bodies.size(); r.n_joints = (uint32_t)c->rigid.joints.size();
    r.sizeof_body = (uint32_t)sizeof(RigidBodyDev); r.sizeof_joint = (uint32_t)sizeof(JointDev);
    memcpy(p, &r, sizeof r); p += sizeof r;
    HIPCHK(c, hipMemcpy(p, c->rigid.d_rb, sizeof(RigidBodyDev) * r.n_bodies, hipMemcpyDeviceToHost)); p += sizeof(RigidBodyDev) * r.n_bodies;
    if (r.n_joints) memcpy(p, c->rigid.joints.data(), sizeof(JointDev) * r.n_joints);
  }
  return MPMHIP_OK;
}

int mpmhip_snapshot_load(mpmhip_ctx *c, const void *src, size_t size) {
  if (!c || !src || size < sizeof(SnapHeader)) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  SnapHeader h;
  memcpy(&h, src, sizeof h);
  if (memcmp(h.magic, "MPMHIP01", 8) != 0 || h.abi != MPMHIP_ABI_VERSION) return fail(c, MPMHIP_EINVAL, "not a libmpmhip snapshot of this ABI version");
  for (int k = 0; k < 3; k++)
    if (h.res[k] != c->P.res[k]) return fail(c, MPMHIP_EINVAL, "snapshot is of a %dx%dx%d grid", h.res[0], h.res[1], h.res[2]);
  if (h.dx != c->P.dx) return fail(c, MPMHIP_EINVAL, "snapshot has delta_x = %g, the ctx %g", h.dx, c->P.dx);
  // the stored P2G affine matrices carry the factor -4 inv_dx dt of the saving ctx (and apic_b is recovered with it)
  if (h.dt != c->P.dt) return fail(c, MPMHIP_EINVAL, "snapshot has base_delta_t = %g, the ctx %g", h.dt, c->P.dt);
  if (h.n_slots < 0 || h.n_slots > c->cap) return fail(c, MPMHIP_ECAPACITY, "snapshot holds %lld particle slots, capacity is %lld", (long long)h.n_slots, (long long)c->cap);
  if ((int)h.n_groups > c->groups_cap) return fail(c, MPMHIP_ECAPACITY, "snapshot holds %u groups", h.n_groups);
  const size_t n = (size_t)h.n_slots;
  const size_t base = sizeof h + sizeof(GroupParams) * h.n_groups + n * (sizeof(RecG) + sizeof(RecP) + sizeof(float) * BW);
  if (size < base) return fail(c, MPMHIP_EINVAL, "snapshot is truncated");
  // the rigid section: the scene must have added the same bodies (meshes, scripts) before it loads
  SnapRigid sr;
  memset(&sr, 0, sizeof sr);
  const bool has_rigid = size >= base + sizeof sr && memcmp((const char *)src + base, "MPMRIGID", 8) == 0;
  if (has_rigid) {
    memcpy(&sr, (const char *)src + base, sizeof sr);
    if (sr.sizeof_body != sizeof(RigidBodyDev) || sr.sizeof_joint != sizeof(JointDev) || sr.n_joints > (uint32_t)MAX_JOINTS ||
        size < base + sizeof sr + (size_t)sr.sizeof_body * sr.n_bodies + (size_t)sr.sizeof_joint * sr.n_joints)
      return fail(c, MPMHIP_EINVAL, "snapshot: the rigid-body section is damaged or of another build");
    if (!rigid_active(c) || c->rigid.bodies.size() != sr.n_bodies)
      return fail(c, MPMHIP_EINVAL, "snapshot holds %u rigid bodies: add the scene's bodies (same meshes, same order) before loading",
                  sr.n_bodies - 1);
  } else if (rigid_active(c)) {
    return fail(c, MPMHIP_EINVAL, "snapshot holds no rigid bodies, this simulation has %d", (int)c->rigid.bodies.size() - 1);
  }
  {
    const RecG *srg = reinterpret_cast<const RecG *>((const char *)src + sizeof h + sizeof(GroupParams) * h.n_groups);
    for (size_t i = 0; i < n; i++)
      if (srg[i].pid >= 0 && srg[i].gid >= h.n_groups)
        return fail(c, MPMHIP_EINVAL, "snapshot record %zu refers to group %u of %u", i, srg[i].gid, h.n_groups);
  }
  const char *p = (const char *)src + sizeof h;
  c->groups.assign((const GroupParams *)p, (const GroupParams *)p + h.n_groups); p += sizeof(GroupParams) * h.n_groups;
  if (h.n_groups) HIPCHK(c, hipMemcpy(c->d_groups, c->groups.data(), sizeof(GroupParams) * h.n_groups, hipMemcpyHostToDevice));
  if (n) {
    HIPCHK(c, hipMemcpy(c->rg, p, sizeof(RecG) * n, hipMemcpyHostToDevice)); p += sizeof(RecG) * n;
    HIPCHK(c, hipMemcpy(c->rp, p, sizeof(RecP) * n, hipMemcpyHostToDevice)); p += sizeof(RecP) * n;
    HIPCHK(c, hipMemcpy(c->rb, p, sizeof(float) * BW * n, hipMemcpyHostToDevice));
  }
  c->n_slots = h.n_slots; c->P.n_slots = (uint32_t)h.n_slots;
  c->substeps = h.substeps; c->next_pid = h.next_pid;
  c->t = h.t; c->request_t = h.request_t;
  // A travels in the records; apic_b is current only if the saving ctx kept it up to date
  c->affine_valid = true;
  c->b_stale = h.b_stale != 0 || (h.store_b == 0);
  if (c->P.store_b && c->b_stale) {  // this ctx keeps apic_b: rebuild it from A once
    int rc = ensure_b_current(c);
    if (rc) return rc;
  }
  c->sorted = c->keys_valid = false;  // keys and block flags are rebuilt by the next sort
  c->ordered = c->compact = false;
  HIPCHK(c, hipMemset(c->blk_flag, 0, (size_t)c->P.nbw * 32));
  Counters hc;
  memset(&hc, 0, sizeof hc);
  hc.n_dead = h.n_dead;
  HIPCHK(c, hipMemcpy(c->cnt, &hc, sizeof hc, hipMemcpyHostToDevice));
  if (has_rigid) {  // poses, velocities and joints of the saved run; colours travel in the records, the CDF is rebuilt every substep
    const char *q = (const char *)src + base + sizeof sr;
    HIPCHK(c, hipMemcpy(c->rigid.d_rb, q, sizeof(RigidBodyDev) * sr.n_bodies, hipMemcpyHostToDevice)); q += sizeof(RigidBodyDev) * sr.n_bodies;
    c->rigid.joints.assign((const JointDev *)q, (const JointDev *)q + sr.n_joints);
    if (sr.n_joints) HIPCHK(c, hipMemcpy(c->rigid.d_joints, c->rigid.joints.data(), sizeof(JointDev) * sr.n_joints, hipMemcpyHostToDevice));
  }
  return MPMHIP_OK;
}

int mpmhip_delete_particles_inside_level_set(mpmhip_ctx *c, int64_t *deleted) {
  if (!c || !deleted) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  *deleted = 0;
  if (c->LS.n <= 0 || c->n_slots == 0) return MPMHIP_OK;
  if (int rc = ensure_b_current(c)) return rc;  // the recovery reads every live record: do it before some die
  Counters before, after;
  if (int rc = read_counters(c, before)) return rc;
  c->P.t = c->t;
  hipLaunchKernelGGL(k_delete_inside_levelset, dim3(particle_grid(c->n_slots)), dim3(256), 0, c->stream, c->P, (RecG *)c->rg,
                     c->LS, c->cnt);
  if (int rc = launch_check(c, "delete_inside_levelset")) return rc;
  if (int rc = read_counters(c, after)) return rc;
  *deleted = (int64_t)after.n_dead - (int64_t)before.n_dead;
  if (*deleted) return invalidate_keys(c);  // key[] still lists the deleted ones: rebuild
  return MPMHIP_OK;
}

// ---------------------------------------------------------------------------------------------- .bgeo frames
int mpmhip_bgeo_size(mpmhip_ctx *c, int32_t verbose, size_t *bytes) {
  if (!c || !bytes) return MPMHIP_EINVAL;
  int64_t n = mpmhip_num_particles(c);
  if (n < 0) return (int)n;
  if (c->rigid.enabled) n += (int64_t)c->rigid.h_smp.size();  // the boundary particles of rigid bodies are rows too (type = 1)
  *bytes = bgeo_bytes((uint32_t)n, verbose != 0);
  return MPMHIP_OK;
}

int mpmhip_bgeo_encode(mpmhip_ctx *c, int32_t verbose, void *dst, size_t capacity, size_t *written) {
  if (!c || !dst || !written) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (verbose)
    if (int rc = ensure_b_current(c)) return rc;
  std::vector<uint32_t> order;
  std::vector<int32_t> ids;
  const bool with_rigid = c->rigid.enabled && !c->rigid.h_smp.empty();
  if (int rc = bgeo_order(c, order, with_rigid ? &ids : nullptr)) return rc;
  const uint32_t nm = (uint32_t)order.size();  // material particles
  const uint32_t n = nm + (with_rigid ? (uint32_t)c->rigid.h_smp.size() : 0u);
  const size_t total = bgeo_bytes(n, verbose != 0);
  int32_t *limits = nullptr;  // async stepping: per-slot (dt_limit, stiffness_limit, cfl_limit) of the particle's block
  if (c->async.enabled && c->async.limits_valid) limits = c->async.d_particle_limits;
  if (total > capacity)
    return fail(c, MPMHIP_ECAPACITY, "bgeo image needs %zu bytes, the buffer holds %zu", total, capacity);
  uint8_t *out = static_cast<uint8_t *>(dst);
  const std::vector<uint8_t> head = bgeo_header(n, verbose != 0);
  std::memcpy(out, head.data(), head.size());
  out += head.size();
  const size_t W = verbose ? BGEO_W_VERBOSE : BGEO_W_PLAIN;
  const size_t row_bytes = (size_t)n * W * 4, mat_bytes = (size_t)nm * W * 4;
  std::vector<uint8_t> mat_rows;  // with rigid bodies the material rows are merged with the boundary particles' by creation id
  uint8_t *mat_dst = out;
  if (with_rigid) { mat_rows.resize(mat_bytes); mat_dst = mat_rows.data(); }
  if (nm) {
    uint32_t *d_order = nullptr, *d_rows = nullptr;
    hipError_t e = dmalloc(&d_order, (size_t)nm);
    if (e == hipSuccess) e = hipMalloc((void **)&d_rows, mat_bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(d_order, order.data(), (size_t)nm * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
      const dim3 grid(particle_grid(nm)), wg(256);
      if (verbose)
        hipLaunchKernelGGL(k_bgeo_rows<true>, grid, wg, 0, c->stream, nm, (const uint32_t *)d_order, (const RecG *)c->rg,
                           (const RecP *)c->rp, (const float *)c->rb, (const GroupParams *)c->d_groups, (const int32_t *)limits, d_rows);
      else
        hipLaunchKernelGGL(k_bgeo_rows<false>, grid, wg, 0, c->stream, nm, (const uint32_t *)d_order, (const RecG *)c->rg,
                           (const RecP *)c->rp, (const float *)c->rb, (const GroupParams *)c->d_groups, (const int32_t *)limits, d_rows);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(mat_dst, d_rows, mat_bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_order);
    (void)hipFree(d_rows);
    HIPCHK(c, e);
  }
  if (with_rigid) {
    // rows of the boundary particles (RigidBoundaryParticle, src/boundary_particle.h): position = anchor point, type = 1,
    // v = body velocity at the anchor (align_with_rigid_body); every other field keeps MPMParticle's constructor value
    auto &R = c->rigid;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<RigidBodyDev> hb(MAX_RIGID);
    HIPCHK(c, hipMemcpy(hb.data(), R.d_rb, sizeof(RigidBodyDev) * MAX_RIGID, hipMemcpyDeviceToHost));
    auto be32 = [](uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; };
    auto bef = [&](uint8_t *p, float f) { uint32_t u; std::memcpy(&u, &f, 4); be32(p, u); };
    size_t im = 0, ir = 0;
    uint8_t *row = out;
    for (uint32_t j = 0; j < n; j++, row += W * 4) {
      const bool take_rigid = ir < R.h_smp.size() && (im >= nm || R.h_smp_id[ir] < ids[im]);
      if (!take_rigid) { std::memcpy(row, mat_rows.data() + (im++) * W * 4, W * 4); continue; }
      const RigidSample &S = R.h_smp[ir];
      const RigidBodyDev &B = hb[S.body];
      float ro[3], x[3], v[3];
      for (int r = 0; r < 3; r++) ro[r] = B.R[3 * r] * S.off[0] + B.R[3 * r + 1] * S.off[1] + B.R[3 * r + 2] * S.off[2];
      for (int r = 0; r < 3; r++) x[r] = ro[r] + B.pos[r];
      v[0] = B.vel[0] + (B.omega[1] * ro[2] - B.omega[2] * ro[1]);
      v[1] = B.vel[1] + (B.omega[2] * ro[0] - B.omega[0] * ro[2]);
      v[2] = B.vel[2] + (B.omega[0] * ro[1] - B.omega[1] * ro[0]);
      std::memset(row, 0, W * 4);
      bef(row, x[0]); bef(row + 4, x[1]); bef(row + 8, x[2]); bef(row + 12, 1.0f);
      be32(row + 16, 1u); be32(row + 20, (uint32_t)R.h_smp_id[ir]);
      be32(row + 24, 1u); be32(row + 28, 1u); be32(row + 32, 1u);
      bef(row + 36, v[0]); bef(row + 40, v[1]); bef(row + 44, v[2]);
      ir++;
    }
  }
  out += row_bytes;
  const std::vector<uint8_t> prim = bgeo_prim_attr();
  std::memcpy(out, prim.data(), prim.size());
  out += prim.size();
  auto w32 = [&](uint32_t v) { for (int s = 24; s >= 0; s -= 8) *out++ = (uint8_t)(v >> s); };
  w32(n);
  if (n > (1u << 16)) {  // BGEO.cpp:175-180: point numbers as int32 above 65536 points, uint16 otherwise
    for (uint32_t i = 0; i < n; i++) w32(i);
  } else {
    for (uint32_t i = 0; i < n; i++) { *out++ = (uint8_t)(i >> 8); *out++ = (uint8_t)i; }
  }
  w32(0);
  *out++ = 0x00;
  *out++ = 0xff;
  *written = (size_t)(out - static_cast<uint8_t *>(dst));
  if (*written != total) return fail(c, MPMHIP_EINVAL, "internal: bgeo image is %zu bytes, expected %zu", *written, total);
  return MPMHIP_OK;
}

int mpmhip_write_bgeo(mpmhip_ctx *c, const char *path, int32_t verbose) {
  if (!c || !path) return MPMHIP_EINVAL;
  const size_t len = std::strlen(path);
  if (len >= 3 && std::strcmp(path + len - 3, ".gz") == 0)
    return fail(c, MPMHIP_ENOTIMPL, "gzip-compressed .bgeo (the reference writes plain %%04d.bgeo, src/mpm.h:336)");
  size_t bytes = 0, written = 0;
  if (int rc = mpmhip_bgeo_size(c, verbose, &bytes)) return rc;
  std::vector<uint8_t> img;
  try { img.resize(bytes); } catch (const std::bad_alloc &) { return fail(c, MPMHIP_ENOMEM, "host allocation of %zu bytes failed", bytes); }
  if (int rc = mpmhip_bgeo_encode(c, verbose, img.data(), img.size(), &written)) return rc;
  FILE *f = std::fopen(path, "wb");
  if (!f) return fail(c, MPMHIP_EINVAL, "cannot open '%s' for writing: %s", path, std::strerror(errno));
  const size_t ok = std::fwrite(img.data(), 1, written, f);
  const int cl = std::fclose(f);
  if (ok != written || cl != 0) return fail(c, MPMHIP_EINVAL, "short write to '%s': %s", path, std::strerror(errno));
  return MPMHIP_OK;
}

// MPM<dim>::calculate_energy (src/mpm.cpp:1078-1110): sort + P2G, kinetic energy of the grid, potential energy
// of the particles.  Leaves the ctx sorted with fresh P2G tiles (like the reference, which leaves its grid rasterized).
// calculate_energy (src/mpm.cpp:1078-1110) in two halves, so that a tiled job can put its halo exchange between them:
//   begin  sort, rasterize (P2G) [+ halo pack and the start of the exchange]
//   end    [the peers' sums have arrived] grid kinetic energy, particles' potential energy -> out = {kinetic, potential, particles of
//          a type without potential_energy()}: this ctx's SHARE — on a tiled ctx a node's kinetic energy is counted by the lowest
//          rank that holds mass on it (k_grid mode 4), so the shares add up to the one-ctx energy.
static int energy_begin(mpmhip_ctx *c) {
  int rc;
  c->ov_active = false;
  if ((rc = do_sort(c))) return rc;
  if ((rc = do_p2g(c))) return rc;
  if (c->tn.on && c->T.n_boxes > 0) {
    tn_begin_substep(c);
    if ((rc = do_halo_pack(c))) return rc;
  }
  return MPMHIP_OK;
}
static int energy_end(mpmhip_ctx *c, double out[3]) {
  int rc;
  if (!c->d_energy) HIPCHK(c, dmalloc(&c->d_energy, 4));
  HIPCHK(c, hipMemsetAsync(c->d_energy, 0, 4 * sizeof(double), c->stream));
  float4 *const dense_saved = c->dense;
  c->dense = reinterpret_cast<float4 *>(c->d_energy);  // k_grid<4> accumulates into its `dense` argument
  rc = do_grid(c, 4);
  c->dense = dense_saved;
  if (rc) return rc;
  double *acc = c->d_energy;
  hipLaunchKernelGGL(k_potential_energy, dim3(particle_grid(c->n_slots)), dim3(256), 0, c->stream, c->P, (const RecG *)c->rg,
                     (const GroupParams *)c->d_groups, acc + 1);
  if ((rc = launch_check(c, "potential_energy"))) return rc;
  double *h = reinterpret_cast<double *>(c->h_pinned + 12288);  // (pinned: behind the reductions' staging)
  HIPCHK(c, hipMemcpyAsync(h, acc, 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < 3; i++) out[i] = h[i];
  return MPMHIP_OK;
}
namespace { int tn_energy(mpmhip_ctx *c, double out[3]); }  // (tiled_api.h: exchange + reduction over the ranks)

int mpmhip_calculate_energy(mpmhip_ctx *c, double *kinetic, double *potential) {
  if (!c || !kinetic || !potential) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->in_substep) return fail(c, MPMHIP_EINVAL, "calculate_energy inside a substep");
  double e[3];
  int rc;
  if (c->tn.on && c->tn.world > 1) {  // the whole job's energy: collective over the ranks
    if ((rc = tn_energy(c, e))) return rc;
  } else {
    if (c->T.n_boxes > 0) return fail(c, MPMHIP_ENOTIMPL, "calculate_energy on a ctx tiled through the callback path (mpmhip_set_halo): "
                                      "use the native data plane (mpmhip_tiled_setup), whose calculate_energy covers the whole job");
    if ((rc = energy_begin(c)) || (rc = energy_end(c, e))) return rc;
  }
  *kinetic = e[0];
  *potential = e[1];
  if (e[2] != 0.0)
    return fail(c, MPMHIP_ENOTIMPL, "%.0f particles are of a type without potential_energy() (reference: TC_NOT_IMPLEMENTED); "
                "kinetic energy is valid", e[2]);
  return MPMHIP_OK;
}

int mpmhip_set_profiling(mpmhip_ctx *c, int32_t level) {
  if (!c || level < 0 || level > 4) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->in_substep) return fail(c, MPMHIP_EINVAL, "set_profiling inside a substep");
  int rc = collect_events(c);  // pending events belong to the old level
  c->profiling = level;
  c->ev_level = level;
  return rc;
}
int mpmhip_set_profile_sampling(mpmhip_ctx *c, int32_t every) {
  if (!c || every < 1) return MPMHIP_EINVAL;
  c->prof_every = every;
  return MPMHIP_OK;
}
int mpmhip_profile_reset(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = collect_events(c);
  for (int k = 0; k < PH_COUNT; k++) c->phase_ms[k] = 0;
  c->prof_substeps = 0;
  return rc;
}
int mpmhip_profile(mpmhip_ctx *c, char *json, size_t cap) {
  if (!c || !json || cap == 0) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = collect_events(c);
  if (rc) return rc;
  Counters h;
  if ((rc = read_counters(c, h))) return rc;
  int w = snprintf(json, cap,
                   "{\"substeps\":%lld,\"particles\":%lld,\"active_blocks\":%u,\"rank_mode\":%u,\"phases\":{\"sort\":%.6f,\"p2g\":%.6f,"
                   "\"exchange\":%.6f,\"grid\":%.6f,\"g2p\":%.6f}}",
                   (long long)c->prof_substeps, (long long)(c->n_slots - h.n_dead), h.n_active, h.rank_mode, c->phase_ms[PH_SORT],
                   c->phase_ms[PH_P2G], c->phase_ms[PH_EXCH], c->phase_ms[PH_GRID], c->phase_ms[PH_G2P]);
  return (w < 0 || (size_t)w >= cap) ? fail(c, MPMHIP_EINVAL, "profile buffer too small") : MPMHIP_OK;
}

// ------------------------------------------------------------------------------------------------ tiling (host)
int mpmhip_set_partition(mpmhip_ctx *c, int32_t rank, const int32_t dims[3], const int32_t *cuts_x,
                         const int32_t *cuts_y, const int32_t *cuts_z, int32_t margin) {
  if (!c || !dims || !cuts_x || !cuts_y || !cuts_z) return MPMHIP_EINVAL;
  if (c->rigid.enabled) return fail(c, MPMHIP_EINVAL, "a ctx with rigid bodies cannot be tiled");
  const int32_t *cuts[3] = {cuts_x, cuts_y, cuts_z};
  Tiling T;
  memset(&T, 0, sizeof T);
  int world = 1;
  for (int a = 0; a < 3; a++) {
    if (dims[a] < 1 || dims[a] > MPMHIP_MAX_PARTS) return fail(c, MPMHIP_EINVAL, "dims[%d]=%d outside [1,%d]", a, dims[a], MPMHIP_MAX_PARTS);
    T.dims[a] = dims[a];
    world *= dims[a];
    for (int k = 0; k <= dims[a]; k++) {
      T.cuts[a][k] = cuts[a][k];
      if (k > 0 && cuts[a][k] <= cuts[a][k - 1]) return fail(c, MPMHIP_EINVAL, "cuts of axis %d are not increasing", a);
    }
    if (cuts[a][0] != 0 || cuts[a][dims[a]] < c->P.res[a]) return fail(c, MPMHIP_EINVAL, "cuts of axis %d do not cover [0,res)", a);
  }
  if (rank < 0 || rank >= world) return fail(c, MPMHIP_EINVAL, "rank %d of %d", rank, world);
  if (margin < 1) return fail(c, MPMHIP_EINVAL, "margin must be >= 1 cell");
  const int pc[3] = {rank / (dims[1] * dims[2]), (rank / dims[2]) % dims[1], rank % dims[2]};
  for (int a = 0; a < 3; a++) { T.lo[a] = T.cuts[a][pc[a]]; T.hi[a] = T.cuts[a][pc[a] + 1]; }
  T.enabled = 1; T.rank = rank; T.margin = margin;
  c->T = T;  // halo boxes (if any) must be set again
  return MPMHIP_OK;
}

int mpmhip_set_halo(mpmhip_ctx *c, int32_t n, const mpmhip_halo_box *boxes) {
  if (!c || n < 0 || n > MPMHIP_MAX_HALO_BOXES || (n > 0 && !boxes)) return MPMHIP_EINVAL;
  if (n > 0 && !c->T.enabled) return fail(c, MPMHIP_EINVAL, "set_halo needs set_partition first");
  if (c->tn.on) return fail(c, MPMHIP_EINVAL, "this ctx has a native plan (mpmhip_tiled_setup): its halo boxes are the library's");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  Tiling &T = c->T;
  T.n_boxes = 0; T.box_nodes = 0; T.box_blocks = 0;
  if (n == 0) return MPMHIP_OK;
  // node box this rank's particles can touch: base cells in [lo-margin, hi+margin), stencil base..base+2
  int nlo[3], nhi[3];
  for (int a = 0; a < 3; a++) {
    nlo[a] = std::max(0, T.lo[a] - T.margin);
    nhi[a] = std::min(c->P.res[a] + 1, T.hi[a] + T.margin + 2);
    T.int_lo[a] = nlo[a]; T.int_hi[a] = nhi[a];
  }
  std::vector<DevBox> hb((size_t)n);
  uint64_t off = 0;
  uint32_t boff = 0;
  bool empty_interior = false;
  for (int i = 0; i < n; i++) {
    const mpmhip_halo_box &b = boxes[i];
    if (i > 0 && b.peer < boxes[i - 1].peer) return fail(c, MPMHIP_EINVAL, "halo boxes must be sorted by peer rank");
    if (b.peer == T.rank || !b.send || !b.recv) return fail(c, MPMHIP_EINVAL, "halo box %d: bad peer or null buffer", i);
    bool proper = false;
    for (int a = 0; a < 3; a++) {
      if (b.lo[a] < 0 || b.hi[a] <= b.lo[a] || b.hi[a] > c->P.res[a] + 1) return fail(c, MPMHIP_EINVAL, "halo box %d: bad extent on axis %d", i, a);
      hb[i].lo[a] = b.lo[a]; hb[i].dim[a] = b.hi[a] - b.lo[a];
      // shrink the overlap-free interior along every axis where the box is a proper sub-range of the node box
      if (b.lo[a] <= nlo[a] && b.hi[a] >= nhi[a]) continue;
      proper = true;
      if (b.lo[a] <= nlo[a]) T.int_lo[a] = std::max(T.int_lo[a], b.hi[a]);
      else if (b.hi[a] >= nhi[a]) T.int_hi[a] = std::min(T.int_hi[a], b.lo[a]);
      else empty_interior = true;
    }
    if (!proper) empty_interior = true;
    hb[i].peer = b.peer; hb[i].off = (uint32_t)off; hb[i].boff = boff;
    boff += box_blocks_of(hb[i].lo, hb[i].dim);
    hb[i].send = (float4 *)b.send; hb[i].recv = (const float4 *)b.recv; hb[i].flag = nullptr;
    off += (uint64_t)hb[i].dim[0] * hb[i].dim[1] * hb[i].dim[2];
    if (off >= (1ull << 31)) return fail(c, MPMHIP_EINVAL, "halo boxes too large");
  }
  if (empty_interior) for (int a = 0; a < 3; a++) T.int_hi[a] = T.int_lo[a];
  if (!c->d_boxes) HIPCHK(c, dmalloc(&c->d_boxes, (size_t)MPMHIP_MAX_HALO_BOXES));
  HIPCHK(c, hipMemcpy(c->d_boxes, hb.data(), sizeof(DevBox) * n, hipMemcpyHostToDevice));
  c->d_boxes_cur = c->d_boxes;
  T.n_boxes = n; T.box_nodes = (uint32_t)off; T.box_blocks = boff;
  return MPMHIP_OK;
}

int mpmhip_halo_pack(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = need_sorted(c, "halo_pack");
  return rc ? rc : do_halo_pack(c);
}

static int ensure_counts(mpmhip_ctx *c, int world) {
  if (!c->T.enabled) return fail(c, MPMHIP_EINVAL, "no partition set");
  if (world != c->T.dims[0] * c->T.dims[1] * c->T.dims[2]) return fail(c, MPMHIP_EINVAL, "world=%d does not match the partition", world);
  if (c->counts_cap < world) {
    hipFree(c->d_counts);
    c->d_counts = nullptr;
    HIPCHK(c, dmalloc(&c->d_counts, (size_t)world + 8));  // + the 6 bounds and the speed of mpmhip_migration_scan + one word of the native data plane
    c->counts_cap = world;
  }
  return MPMHIP_OK;
}

// one pass over the particles, one synchronisation: leaver counts per destination + bounding box of the base cells
// + the fastest particle (cells per substep)
int mpmhip_migration_scan(mpmhip_ctx *c, int32_t world, int64_t *counts, int32_t lo[3], int32_t hi[3],
                          float *max_cells_per_substep) {
  if (!c || !counts) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = ensure_counts(c, world);
  if (rc) return rc;
  if (c->in_substep) return fail(c, MPMHIP_EINVAL, "migration inside a substep");
  if ((size_t)world + 7 + sizeof(Counters) / 4 > 65536 / 4) return fail(c, MPMHIP_EINVAL, "world too large");
  hipLaunchKernelGGL(k_scan_init, dim3((world + 8 + 255) / 256), dim3(256), 0, c->stream, c->d_counts, world, 0u);
  int grid = particle_grid(c->n_slots);
  if (grid > 128) grid = 128;  // few workgroups: 6 same-address atomics each for the bounds
  hipLaunchKernelGGL(k_leaver_count, dim3(grid), dim3(256), 0, c->stream, c->P, c->T, (const float4 *)c->rg,
                     (const float4 *)c->rp, c->d_counts, reinterpret_cast<int *>(c->d_counts + world), c->cnt);
  if ((rc = launch_check(c, "leaver_count"))) return rc;
  uint32_t *h = c->h_pinned + sizeof(Counters) / 4;  // behind the counters read_counters() fetches
  HIPCHK(c, hipMemcpyAsync(h, c->d_counts, sizeof(uint32_t) * ((size_t)world + 7), hipMemcpyDeviceToHost, c->stream));
  Counters hc;
  if ((rc = read_counters(c, hc))) return rc;  // the ONE synchronisation; reports the margin violation
  for (int i = 0; i < world; i++) counts[i] = h[i];
  for (int k = 0; k < 3; k++) {
    if (lo) lo[k] = (int32_t)h[world + k];
    if (hi) hi[k] = (int32_t)h[world + 3 + k];
  }
  if (max_cells_per_substep) std::memcpy(max_cells_per_substep, &h[world + 6], sizeof(float));
  return MPMHIP_OK;
}

int mpmhip_leaver_counts(mpmhip_ctx *c, int32_t world, int64_t *counts) {
  return mpmhip_migration_scan(c, world, counts, nullptr, nullptr, nullptr);
}

int mpmhip_export_leavers(mpmhip_ctx *c, int32_t world, const int64_t *counts, void *dev_records) {
  if (!c || !counts) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = ensure_counts(c, world);
  if (rc) return rc;
  if (c->in_substep) return fail(c, MPMHIP_EINVAL, "migration inside a substep");
  std::vector<uint32_t> cur((size_t)world);
  uint64_t off = 0;
  for (int i = 0; i < world; i++) { cur[i] = (uint32_t)off; off += (uint64_t)counts[i]; }
  if (off == 0) return MPMHIP_OK;
  if (!dev_records) return MPMHIP_EINVAL;
  uint32_t *pin = c->h_pinned + 1024;  // stays untouched until the next migration: no synchronisation needed
  memcpy(pin, cur.data(), sizeof(uint32_t) * world);
  HIPCHK(c, hipMemcpyAsync(c->d_counts, pin, sizeof(uint32_t) * world, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_leaver_pack, dim3(particle_grid(c->n_slots)), dim3(256), 0, c->stream, c->P, c->T, (float4 *)c->rg,
                     (const float4 *)c->rp, (const float4 *)c->rb, c->key, c->d_counts, (float4 *)dev_records, c->cnt);
  return launch_check(c, "leaver_pack");
}

int mpmhip_import_particles(mpmhip_ctx *c, int64_t n, const void *dev_records) {
  if (!c || n < 0 || (n > 0 && !dev_records)) return MPMHIP_EINVAL;
  if (n == 0) return MPMHIP_OK;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->in_substep || c->sorted) return fail(c, MPMHIP_EINVAL, "import_particles between sort and G2P");
  if (c->n_slots + n > c->cap)
    return fail(c, MPMHIP_ECAPACITY, "particle capacity exceeded on import: %lld + %lld > %lld (request_compaction or a larger max_particles)",
                (long long)c->n_slots, (long long)n, (long long)c->cap);
  hipLaunchKernelGGL(k_import, dim3(particle_grid(n)), dim3(256), 0, c->stream, c->P, (uint32_t)n, (uint32_t)c->n_slots,
                     (const float4 *)dev_records, (float4 *)c->rg, (float4 *)c->rp, (float4 *)c->rb, c->key, c->blk_flag,
                     c->cnt);
  c->n_slots += n;
  c->P.n_slots = (uint32_t)c->n_slots;
  c->compact = false;  // live records now also sit behind the range k_g2p compacted
  return launch_check(c, "import");
}

int mpmhip_active_bounds(mpmhip_ctx *c, int32_t lo[3], int32_t hi[3]) {
  if (!c || !lo || !hi) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->d_bounds) HIPCHK(c, dmalloc(&c->d_bounds, 6));
  const int init[6] = {1 << 30, 1 << 30, 1 << 30, -1, -1, -1};
  int h[6];
  HIPCHK(c, hipMemcpyAsync(c->d_bounds, init, sizeof init, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // `init` lives on this stack frame
  hipLaunchKernelGGL(k_active_bounds, dim3(64), dim3(256), 0, c->stream, c->P, (const Counters *)c->cnt,
                     (const uint32_t *)c->act_blk, c->d_bounds);
  int rc = launch_check(c, "active_bounds");
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(h, c->d_bounds, sizeof h, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < 3; k++) { lo[k] = h[k]; hi[k] = h[3 + k]; }
  return MPMHIP_OK;
}

int64_t mpmhip_num_slots(mpmhip_ctx *c) { return c ? c->n_slots : MPMHIP_EINVAL; }

int mpmhip_request_compaction(mpmhip_ctx *c) {
  if (!c) return MPMHIP_EINVAL;
  c->compact_requested = true;
  return MPMHIP_OK;
}

int64_t mpmhip_capacity(mpmhip_ctx *c) { return c ? c->cap : MPMHIP_EINVAL; }

int mpmhip_reserve(mpmhip_ctx *c, int64_t max_particles) {
  if (!c) return MPMHIP_EINVAL;
  if (max_particles <= c->cap) return MPMHIP_OK;
  if (max_particles >= (1ll << 31)) return fail(c, MPMHIP_EINVAL, "max_particles out of range");
  if (c->in_substep) return fail(c, MPMHIP_EINVAL, "reserve inside a substep");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t n = (size_t)c->n_slots, cap = (size_t)max_particles;
  hipError_t e = hipSuccess;
  auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  A(regrow(&c->rg, n, cap, false)); A(regrow(&c->rp, n, cap, false)); A(regrow(&c->rb, n * BW, cap * BW, true));
  A(regrow(&c->rg2, 0, cap, false)); A(regrow(&c->rp2, 0, cap, false)); A(regrow(&c->rb2, 0, cap * BW, false));
  A(regrow(&c->key, 0, cap, false)); A(regrow(&c->pidc, 0, cap, false)); A(regrow(&c->rank, 0, cap, false)); A(regrow(&c->perm, 0, cap, false)); A(regrow(&c->chunk_blk, 0, cap / 256 + 2, false));
  if (c->rigid.d_bnd) A(regrow(&c->rigid.d_bnd, n, cap, true));
  if (c->async.d_blk_of) {  // (re-allocated at the size of the ctx by the next update_dt_limits)
    (void)hipFree(c->async.d_blk_of); (void)hipFree(c->async.d_particle_limits);
    c->async.d_blk_of = nullptr; c->async.d_particle_limits = nullptr; c->async.blk_of_cap = 0; c->async.limits_valid = false;
  }
  if (e != hipSuccess) return fail(c, MPMHIP_ENOMEM, "growing the particle arrays to %lld failed: %s", (long long)max_particles, hipGetErrorString(e));
  c->cap = max_particles;
  c->cfg.max_particles = max_particles;
  c->P.pidc = c->deterministic ? c->pidc : nullptr;
  c->pidc_valid = false;
  if (c->cfg.max_blocks <= 0) {  // auto-sized block table: same rule as mpmhip_create
    int64_t mb = c->cap / 48 + 4096;
    if (mb > (int64_t)c->NB) mb = c->NB;
    if ((uint32_t)mb > c->P.max_blocks) {
      const size_t m = (size_t)mb;
      A(regrow(&c->act_blk, 0, m + 1, false)); A(regrow(&c->act_start, 0, m + 2, true));
      A(regrow(&c->cell_cnt, 0, m * BC, true)); A(regrow(&c->cell_start, 0, m * BC + 1, true));
      A(regrow(&c->nbr, 0, m * 32, false)); A(regrow(&c->own_list, 0, m * 8, false));
      c->ct_slots = (uint32_t)((m + 15) / 16 + 1);
      A(regrow(&c->scan_slots, 0, c->bt_slots + 2 * (size_t)c->ct_slots, true));  // (epoch 0 is never used)
      c->list_clear_epoch = c->sort_epoch;
      A(regrow(&c->tiles, 0, m * TN, false)); A(regrow(&c->gridv, 0, m * 8 * BC, false));
      if (c->rigid.d_blk_rigid) { A(regrow(&c->rigid.d_blk_rigid, 0, m + 1, true)); A(regrow(&c->rigid.d_rigid_list, 0, m + 1, false)); }
      if (e != hipSuccess) return fail(c, MPMHIP_ENOMEM, "growing the block table to %lld failed: %s", (long long)mb, hipGetErrorString(e));
      c->P.max_blocks = (uint32_t)mb;
      HIPCHK(c, hipMemset(c->fat_slot, 0, sizeof(uint32_t) * (size_t)c->NB));  // slots of the old grid array
    }
  }
  c->keys_valid = true;  // (forces invalidate_keys to clear the block flags: the next sort rebuilds keys from the records)
  return invalidate_keys(c);
}

// the launch bound of the chained scans as a function of what the occupancy API answered (no device needed: scan_grid_for)
int mpmhip_debug_scan_grid(int32_t n_cus, int32_t per_cu, int32_t env_request, uint32_t *limit, uint32_t *resident) {
  if (!limit || !resident || n_cus < 1 || per_cu < 0) return MPMHIP_EINVAL;
  *limit = scan_grid_for(n_cus, per_cu, env_request);
  *resident = scan_resident_set(n_cus, per_cu);
  return MPMHIP_OK;
}
int mpmhip_debug_g2p_is_packed(const mpmhip_ctx *c) { return c ? (g2p_is_packed(c, 0) ? 1 : 0) : MPMHIP_EINVAL; }
int mpmhip_debug_copy_bandwidth(mpmhip_ctx *c, size_t bytes, int32_t iters, double *gb_per_s) {
  if (!c || !gb_per_s || iters <= 0 || bytes < 16) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = bytes / 16;
  float4 *a = nullptr, *b = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = dmalloc(&a, n);
  if (e == hipSuccess) e = dmalloc(&b, n);
  if (e == hipSuccess) e = hipMemsetAsync(a, 0, n * 16, c->stream);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  double best = 0.0;
  // best over a few shapes of the same copy (loads in flight per lane, workgroups per CU, store hint): the
  // yardstick should be the best a plain kernel does, not one arbitrary launch shape
  using Kern = void (*)(float4 *, const float4 *, size_t);
  const Kern kerns[] = {k_stream_copy<1, false>, k_stream_copy<4, false>, k_stream_copy<4, true>, k_stream_copy<8, true>};
  const int grids[] = {256 * 4, 256 * 16, 256 * 64};
  for (const Kern &k : kerns)
    for (int g : grids)
      for (int it = 0; e == hipSuccess && it < iters + 1; it++) {  // the first pass of a shape is a warm-up
        hipEventRecord(e0, c->stream);
        hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, c->stream, b, (const float4 *)a, n);
        hipEventRecord(e1, c->stream);
        e = hipEventSynchronize(e1);
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && it > 0 && ms > 0.0f) best = std::max(best, 2.0 * (double)n * 16.0 / (ms * 1e-3) / 1e9);
        if (it == iters && getenv("MPMHIP_BW_VERBOSE"))
          std::fprintf(stderr, "copy variant %d grid %d: %.0f GB/s\n", (int)(&k - kerns), g, 2.0 * (double)n * 16.0 / (ms * 1e-3) / 1e9);
      }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(a);
  (void)hipFree(b);
  HIPCHK(c, e);
  *gb_per_s = best;
  return MPMHIP_OK;
}

// 64-byte record gather of known size (calibration of the FETCH_SIZE counter, profiles/calibrate_fetch.py): n (a power
// of two) records read through an index of the given pattern, `iters` launches; *gb_per_s = (64 + 4) n / best time
int mpmhip_debug_gather_bandwidth(mpmhip_ctx *c, int64_t n, int32_t mode, int32_t iters, double *gb_per_s) {
  if (!c || !gb_per_s || iters <= 0 || n < 1024 || (n & (n - 1)) || n > (1ll << 30) || mode < 0 || mode > 2) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  float4 *rec = nullptr, *out = nullptr;
  uint32_t *idx = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const int grid = 256 * 16;
  hipError_t e = dmalloc(&rec, (size_t)n * 4);
  if (e == hipSuccess) e = dmalloc(&idx, (size_t)n);
  if (e == hipSuccess) e = dmalloc(&out, (size_t)grid * 4);
  if (e == hipSuccess) e = hipMemsetAsync(rec, 0, (size_t)n * 64, c->stream);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  double best = 0.0;
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_probe_indices, dim3(4096), dim3(256), 0, c->stream, idx, (uint32_t)n, (int)mode);
    for (int it = 0; e == hipSuccess && it < iters + 1; it++) {
      hipEventRecord(e0, c->stream);
      hipLaunchKernelGGL(k_gather_records_probe, dim3(grid), dim3(256), 0, c->stream, (const float4 *)rec, (const uint32_t *)idx,
                         (uint32_t)n, out);
      hipEventRecord(e1, c->stream);
      e = hipEventSynchronize(e1);
      float ms = 0.0f;
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
      if (e == hipSuccess && it > 0 && ms > 0.0f) best = std::max(best, 68.0 * (double)n / (ms * 1e-3) / 1e9);
    }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(rec); (void)hipFree(idx); (void)hipFree(out);
  HIPCHK(c, e);
  *gb_per_s = best;
  return MPMHIP_OK;
}

// ------------------------------------------------------------------------------------------------ 2D demo (configs[0])
struct mpmhip_mpm88 {
  mpm88::Params P{};
  int device = 0;
  int64_t n = 0, cap = 0;
  float *x = nullptr, *v = nullptr, *F = nullptr, *C = nullptr, *Jp = nullptr, *grid = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
};
static thread_local std::string g_mpm88_create_error;
static int fail88(mpmhip_mpm88 *m, int code, const std::string &msg) {
  (m ? m->err : g_mpm88_create_error) = msg;
  return code;
}
#define HIPCHK88(m, call)                                                                                      \
  do {                                                                                                         \
    hipError_t e_ = (call);                                                                                    \
    if (e_ != hipSuccess) return fail88((m), MPMHIP_EHIP, std::string(#call " failed: ") + hipGetErrorString(e_)); \
  } while (0)

const char *mpmhip_mpm88_last_error(const mpmhip_mpm88 *m) { return m ? m->err.c_str() : g_mpm88_create_error.c_str(); }

int mpmhip_mpm88_create(int32_t n_grid, float dt, int32_t plastic, int32_t device, mpmhip_mpm88 **out) {
  if (!out || n_grid < 4 || n_grid > 8192 || !(dt > 0.0f)) return fail88(nullptr, MPMHIP_EINVAL, "mpm88: bad n_grid / dt");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail88(nullptr, MPMHIP_EHIP, "no HIP device available: libmpmhip has no CPU fallback");
  if (device < 0 || device >= ndev) return fail88(nullptr, MPMHIP_EINVAL, "mpm88: no such device");
  mpmhip_mpm88 *m = new (std::nothrow) mpmhip_mpm88;
  if (!m) return fail88(nullptr, MPMHIP_ENOMEM, "host allocation failed");
  m->device = device;
  const float E = 1e4f, nu = 0.2f;  // mls-mpm88.cpp:8-9
  m->P.n = n_grid; m->P.dt = dt; m->P.dx = 1.0f / n_grid; m->P.inv_dx = 1.0f / m->P.dx;
  m->P.mu_0 = E / (2 * (1 + nu)); m->P.lambda_0 = E * nu / ((1 + nu) * (1 - 2 * nu)); m->P.hardening = 10.0f;
  m->P.mass = 1.0f; m->P.vol = 1.0f; m->P.plastic = plastic ? 1 : 0;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = dmalloc(&m->grid, (size_t)3 * (n_grid + 1) * (n_grid + 1));
  if (e != hipSuccess) {
    const int rc = fail88(nullptr, MPMHIP_EHIP, std::string("mpm88 create: ") + hipGetErrorString(e));
    mpmhip_mpm88_destroy(m);
    return rc;
  }
  *out = m;
  return MPMHIP_OK;
}

void mpmhip_mpm88_destroy(mpmhip_mpm88 *m) {
  if (!m) return;
  hipSetDevice(m->device);
  if (m->stream) { hipStreamSynchronize(m->stream); hipStreamDestroy(m->stream); }
  hipFree(m->x); hipFree(m->v); hipFree(m->F); hipFree(m->C); hipFree(m->Jp); hipFree(m->grid);
  delete m;
}

int64_t mpmhip_mpm88_num_particles(const mpmhip_mpm88 *m) { return m ? m->n : MPMHIP_EINVAL; }

int mpmhip_mpm88_add(mpmhip_mpm88 *m, int64_t n, const float *x, const float *v, const float *F, const float *C, const float *Jp) {
  if (!m || n < 0 || (n > 0 && !x)) return MPMHIP_EINVAL;
  if (n == 0) return MPMHIP_OK;
  HIPCHK88(m, hipSetDevice(m->device));
  HIPCHK88(m, hipStreamSynchronize(m->stream));
  const int64_t total = m->n + n;
  if (total > m->cap) {  // grow: new arrays, old contents copied over
    const int64_t cap = std::max<int64_t>(total, 2 * m->cap);
    float **arrs[5] = {&m->x, &m->v, &m->F, &m->C, &m->Jp};
    const int width[5] = {2, 2, 4, 4, 1};
    for (int a = 0; a < 5; a++) {
      float *fresh = nullptr;
      HIPCHK88(m, dmalloc(&fresh, (size_t)cap * width[a]));
      if (m->n) HIPCHK88(m, hipMemcpy(fresh, *arrs[a], sizeof(float) * m->n * width[a], hipMemcpyDeviceToDevice));
      (void)hipFree(*arrs[a]);
      *arrs[a] = fresh;
    }
    m->cap = cap;
  }
  std::vector<float> h;
  auto put = [&](float *dst, const float *src, int width, const float *dflt) -> hipError_t {
    if (!src) {
      h.resize((size_t)n * width);
      for (int64_t i = 0; i < n; i++)
        for (int k = 0; k < width; k++) h[(size_t)i * width + k] = dflt[k];
      src = h.data();
    }
    return hipMemcpy(dst + m->n * width, src, sizeof(float) * n * width, hipMemcpyHostToDevice);
  };
  const float zero4[4] = {0, 0, 0, 0}, eye[4] = {1, 0, 0, 1}, one[1] = {1};
  HIPCHK88(m, put(m->x, x, 2, zero4));
  HIPCHK88(m, put(m->v, v, 2, zero4));   // Particle(x, c, v = Vec(0)): F(1), C(0), Jp(1)  (mls-mpm88.cpp:12-13)
  HIPCHK88(m, put(m->F, F, 4, eye));
  HIPCHK88(m, put(m->C, C, 4, zero4));
  HIPCHK88(m, put(m->Jp, Jp, 1, one));
  m->n = total;
  return MPMHIP_OK;
}

int mpmhip_mpm88_advance(mpmhip_mpm88 *m, int32_t steps) {
  if (!m || steps < 0) return MPMHIP_EINVAL;
  HIPCHK88(m, hipSetDevice(m->device));
  const int nn = m->P.n + 1;
  const dim3 pg((unsigned)std::max<int64_t>((m->n + 255) / 256, 1)), gg((unsigned)((nn * nn + 255) / 256)), wg(256);
  for (int s = 0; s < steps; s++) {
    HIPCHK88(m, hipMemsetAsync(m->grid, 0, sizeof(float) * 3 * nn * nn, m->stream));  // :17
    if (m->n)
      hipLaunchKernelGGL(mpm88::k_p2g, pg, wg, 0, m->stream, m->P, m->n, (const float *)m->x, (const float *)m->v,
                         (const float *)m->F, (const float *)m->C, (const float *)m->Jp, m->grid);
    hipLaunchKernelGGL(mpm88::k_grid, gg, wg, 0, m->stream, m->P, m->grid);
    if (m->n)
      hipLaunchKernelGGL(mpm88::k_g2p, pg, wg, 0, m->stream, m->P, m->n, m->x, m->v, m->F, m->C, m->Jp, (const float *)m->grid);
    HIPCHK88(m, hipGetLastError());
  }
  return MPMHIP_OK;
}

int mpmhip_mpm88_download(mpmhip_mpm88 *m, float *x, float *v, float *F, float *C, float *Jp) {
  if (!m) return MPMHIP_EINVAL;
  HIPCHK88(m, hipSetDevice(m->device));
  HIPCHK88(m, hipStreamSynchronize(m->stream));
  float *dst[5] = {x, v, F, C, Jp};
  const float *src[5] = {m->x, m->v, m->F, m->C, m->Jp};
  const int width[5] = {2, 2, 4, 4, 1};
  for (int a = 0; a < 5; a++)
    if (dst[a] && m->n) HIPCHK88(m, hipMemcpy(dst[a], src[a], sizeof(float) * m->n * width[a], hipMemcpyDeviceToHost));
  return MPMHIP_OK;
}

int mpmhip_mpm88_download_grid(mpmhip_mpm88 *m, float *grid) {
  if (!m || !grid) return MPMHIP_EINVAL;
  HIPCHK88(m, hipSetDevice(m->device));
  HIPCHK88(m, hipStreamSynchronize(m->stream));
  const int nn = m->P.n + 1;
  HIPCHK88(m, hipMemcpy(grid, m->grid, sizeof(float) * 3 * nn * nn, hipMemcpyDeviceToHost));
  return MPMHIP_OK;
}

// ------------------------------------------------------------------------------------------------ AsyncMPM (first half)
// Block-local time-step limits of the reference's asynchronous stepper (src/async/async_mpm.{h,cpp}): the per-block
// reduction over the particles runs on the device, the block state machine (power-of-two limits, :112-164) on the host.
int mpmhip_async_enable(mpmhip_ctx *c, const mpmhip_async_config *cfg) {
  if (!c || !cfg) return MPMHIP_EINVAL;
  if (!(cfg->unit_delta_t > 0) || cfg->max_units < 1) return fail(c, MPMHIP_EINVAL, "unit_delta_t > 0 and max_units >= 1 required");
  if (rigid_active(c) || c->rigid.enabled) return fail(c, MPMHIP_EINVAL, "asynchronous stepping cannot be combined with rigid bodies");
  if (c->T.enabled) return fail(c, MPMHIP_EINVAL, "asynchronous stepping cannot be combined with the multi-GPU tiling");
  HIPCHK(c, hipSetDevice(c->device));
  auto &A = c->async;
  A.sched_enable(3, c->P.res, *cfg);
  const size_t nblk = A.nblk();
  hipFree(A.d_tab); hipFree(A.d_blk_limits);
  A.d_tab = nullptr; A.d_blk_limits = nullptr;
  HIPCHK(c, dmalloc(&A.d_tab, 3 * nblk));
  HIPCHK(c, dmalloc(&A.d_blk_limits, 3 * nblk));
  A.enabled = true; A.limits_valid = false;
  return MPMHIP_OK;
}

static int async_ensure_particle_arrays(mpmhip_ctx *c) {
  auto &A = c->async;
  if (A.blk_of_cap < c->cap) {
    hipFree(A.d_blk_of); hipFree(A.d_particle_limits);
    A.d_blk_of = nullptr; A.d_particle_limits = nullptr;
    HIPCHK(c, dmalloc(&A.d_blk_of, (size_t)c->cap));
    HIPCHK(c, dmalloc(&A.d_particle_limits, 3 * (size_t)c->cap));
    A.blk_of_cap = c->cap;
  }
  return MPMHIP_OK;
}
static int async_reset_table(mpmhip_ctx *c) {
  // (min allowed dt, max |v|^2, count) per block: min starts at the bits of 0.1f (":104 min_allowed_dt = 0.1"), max at 1e-16f
  auto &A = c->async;
  const size_t nblk = A.strength.size();
  std::vector<uint32_t> init(3 * nblk);
  const float f01 = 0.1f, fv = 1e-16f;
  uint32_t b01, bv;
  memcpy(&b01, &f01, 4); memcpy(&bv, &fv, 4);
  for (size_t b = 0; b < nblk; b++) { init[3 * b] = b01; init[3 * b + 1] = bv; init[3 * b + 2] = 0; }
  HIPCHK(c, hipMemcpyAsync(A.d_tab, init.data(), sizeof(uint32_t) * 3 * nblk, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // (`init` lives on this stack frame)
  return MPMHIP_OK;
}
// the reduced table -> the block state machine of AsyncMPM<dim>::update_dt_limits (AsyncSched::limits_from_table)
static int async_limits_from_table(mpmhip_ctx *c) {
  auto &A = c->async;
  const size_t nblk = A.strength.size();
  if (A.h_tab_cap < 3 * nblk) {  // pinned staging (a copy into pageable memory is staged by the runtime: ~50 us more)
    if (A.h_tab) (void)hipHostFree(A.h_tab);
    A.h_tab = nullptr;
    HIPCHK(c, hipHostMalloc((void **)&A.h_tab, sizeof(uint32_t) * 3 * nblk, hipHostMallocDefault));
    A.h_tab_cap = 3 * nblk;
  }
  HIPCHK(c, hipMemcpyAsync(A.h_tab, A.d_tab, sizeof(uint32_t) * 3 * nblk, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (!A.limits_from_table(A.h_tab, c->P.dx)) return fail(c, MPMHIP_EINVAL, "%s", A.sched_err.c_str());
  return MPMHIP_OK;
}

int mpmhip_async_update_dt_limits(mpmhip_ctx *c) {  // AsyncMPM<dim>::update_dt_limits, src/async/async_mpm.cpp:90-164
  if (!c) return MPMHIP_EINVAL;
  auto &A = c->async;
  if (!A.enabled) return fail(c, MPMHIP_EINVAL, "mpmhip_async_enable first");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t nblk = A.strength.size();
  if (int rc = async_ensure_particle_arrays(c)) return rc;
  if (int rc = async_reset_table(c)) return rc;
  hipLaunchKernelGGL(k_async_block_reduce, dim3(particle_grid(c->n_slots)), dim3(256), 0, c->stream, c->P, (const RecG *)c->rg,
                     (const RecP *)c->rp, (const GroupParams *)c->d_groups, A.nb[0], A.nb[1], A.nb[2], A.d_tab, A.d_blk_of);
  if (int rc = launch_check(c, "async_block_reduce")) return rc;
  if (int rc = async_limits_from_table(c)) return rc;
  // what the frame output shows per particle (src/async/async_visualize.cpp:17-26)
  std::vector<int32_t> lim(3 * nblk);
  for (size_t b = 0; b < nblk; b++) { lim[3 * b] = (int32_t)A.continuous[b]; lim[3 * b + 1] = (int32_t)A.strength[b]; lim[3 * b + 2] = (int32_t)A.cfl[b]; }
  HIPCHK(c, hipMemcpy(A.d_blk_limits, lim.data(), sizeof(int32_t) * 3 * nblk, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_async_particle_limits, dim3(particle_grid(c->n_slots)), dim3(256), 0, c->stream, c->P,
                     (const uint32_t *)A.d_blk_of, (const int32_t *)A.d_blk_limits, A.d_particle_limits);
  if (int rc = launch_check(c, "async_particle_limits")) return rc;
  A.limits_valid = true;
  return MPMHIP_OK;
}

// non-empty scheduler blocks: corner node (3 ints), strength / cfl / continuous limits, particle count.  Returns the number.
int64_t mpmhip_async_blocks(mpmhip_ctx *c, int64_t capacity, int32_t *coord, int64_t *strength, int64_t *cfl, int64_t *continuous,
                            int64_t *count, int64_t min_max[2]) {
  if (!c || !c->async.enabled) return MPMHIP_EINVAL;
  auto &A = c->async;
  int64_t n = 0;
  for (int bx = 0; bx < A.nb[0]; bx++)
    for (int by = 0; by < A.nb[1]; by++)
      for (int bz = 0; bz < A.nb[2]; bz++) {
        const size_t b = ((size_t)bx * A.nb[1] + by) * A.nb[2] + bz;
        if (!A.count[b]) continue;
        if (n >= capacity) return fail(c, MPMHIP_ECAPACITY, "block buffer too small");
        if (coord) { coord[3 * n] = bx * 4; coord[3 * n + 1] = by * 4; coord[3 * n + 2] = bz * 8; }
        if (strength) strength[n] = A.strength[b];
        if (cfl) cfl[n] = A.cfl[b];
        if (continuous) continuous[n] = A.continuous[b];
        if (count) count[n] = A.count[b];
        n++;
      }
  if (min_max) { min_max[0] = A.min_delta_t_int; min_max[1] = A.max_delta_t_int; }
  return n;
}

// dense view of the block table: nb[3] blocks per axis (block b = (bx nb[1] + by) nb[2] + bz), limits of EVERY block
int64_t mpmhip_async_table(mpmhip_ctx *c, int32_t nb[3], int64_t capacity, int64_t *strength, int64_t *cfl, int64_t *continuous,
                           int64_t *count) {
  if (!c || !c->async.enabled || !nb) return MPMHIP_EINVAL;
  auto &A = c->async;
  for (int k = 0; k < 3; k++) nb[k] = A.nb[k];
  const int64_t n = (int64_t)A.continuous.size();
  if (capacity < n) return n;  // (query of the size)
  for (int64_t b = 0; b < n; b++) {
    if (strength) strength[b] = A.strength[b];
    if (cfl) cfl[b] = A.cfl[b];
    if (continuous) continuous[b] = A.continuous[b];
    if (count) count[b] = A.count[b];
  }
  return n;
}

int mpmhip_async_set_time_int(mpmhip_ctx *c, int64_t t_int) {
  if (!c || !c->async.enabled || t_int < 0) return MPMHIP_EINVAL;
  c->async.current_t_int = t_int;
  return MPMHIP_OK;
}

#include "async_api.h"

int mpmhip_debug_allowed_dt(mpmhip_ctx *c, int32_t material, const float params[MPMHIP_NPARAM], int64_t n, const float *F,
                            const float *aux, const float *v, float dx, float *out) {
  if (!c || n <= 0) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  GroupParams g;
  if (material < MPMHIP_VISCO || material > MPMHIP_ELASTIC) return fail(c, MPMHIP_EINVAL, "unknown material id %d", material);
  memset(&g, 0, sizeof g);
  memcpy(g.p, params, sizeof g.p);
  g.type = material;
  float *dF, *dA, *dV, *dO;
  HIPCHK(c, dmalloc(&dF, 9 * n)); HIPCHK(c, dmalloc(&dA, n)); HIPCHK(c, dmalloc(&dV, 3 * n)); HIPCHK(c, dmalloc(&dO, n));
  HIPCHK(c, hipMemcpy(dF, F, sizeof(float) * 9 * n, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(dA, aux, sizeof(float) * n, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(dV, v, sizeof(float) * 3 * n, hipMemcpyHostToDevice));
  int rc = run_debug(c, k_debug_allowed_dt, g, n, (const float *)dF, (const float *)dA, (const float *)dV, dx, dO);
  if (!rc) HIPCHK(c, hipMemcpy(out, dO, sizeof(float) * n, hipMemcpyDeviceToHost));
  hipFree(dF); hipFree(dA); hipFree(dV); hipFree(dO);
  return rc;
}

// ------------------------------------------------------------------------------------------------ MPM<2>
// The reference's 2D simulation (create_simulation2('mpm') -> MPM<2>, which runs the generic transfer path): its own small
// object — SoA particle arrays, dense (res+1)^2 grid — see k_mpm2d.h.
struct mpmhip2d_ctx {
  mpm2d::Params P{};
  LevelSetDev LS{};
  int device = 0;
  int64_t n = 0, cap = 0;
  float *x = nullptr, *v = nullptr, *F = nullptr, *B = nullptr, *aux = nullptr, *grid = nullptr;
  int32_t *gid = nullptr, *pid = nullptr;
  unsigned int *n_dead = nullptr;
  GroupParams *d_groups = nullptr;
  std::vector<GroupParams> groups;
  int32_t next_pid = 0;
  float t = 0.0f, request_t = 0.0f;
  hipStream_t stream = nullptr;
  std::string err;
  // CPIC rigid bodies (k_rigid2d.h): segments, boundary particles, dense colored distance field
  struct HostRigid2 { mpmhip2d_rigid_config cfg{}; float mass = 0, inertia = 0; int64_t first_elem = 0, n_elems = 0; };
  bool rigid_enabled = false;
  std::vector<HostRigid2> bodies;
  std::vector<mpm2d::Sample2> h_smp;
  std::vector<int32_t> h_smp_id;  // creation id of every boundary particle (they share the material particles' counter)
  std::vector<float> h_elems;
  mpm2d::Rigid2 *d_rb = nullptr;
  mpm2d::Sample2 *d_smp = nullptr;
  float *d_elems = nullptr;
  unsigned long long *d_mind = nullptr;
  uint32_t *d_tags = nullptr, *d_states = nullptr;
  mpm2d::Bnd2 *d_bnd = nullptr;
  float penalty = 0.0f, pushing_force = 20000.0f;
  // rigid_body_levelset_collision: the boundary particles' order (see k2_ls_keys)
  bool ls_collision = false;
  uint32_t *d_smp_rank = nullptr, *d_ls_vals[2] = {nullptr, nullptr}, n_ranked = 0, ls_cap = 0;
  unsigned long long *d_ls_keys[2] = {nullptr, nullptr};
  void *d_ls_tmp = nullptr;
  size_t ls_tmp_bytes = 0;
  mpm2d::Joints2 joints{};     // MPM<2>::articulations ('rotation' joints)
  int joint_iterations = 100;  // 'articulation_iterations'
  float base_dt = 0.0f;        // the configured "base_delta_t" (P.dt is what the next substep uses: the async stepper sets it per advance)
  // AsyncMPM<2> (async2d_api.h): the block scheduler + the device store of pool / backup containers (k_async2d.h)
  struct Async2 : AsyncSched {
    bool resident = false, pending_counters = false;
    bool view = false;  // the particle arrays are copies of the pools (mpmhip2d_async_load_pools), not new particles
    uint32_t cap = 0, size = 0, live = 0, size_ub = 0;  // containers allocated / in use incl. freed ones / not freed / upper bound now
    float4 *rec = nullptr, *rec2 = nullptr;
    uint32_t *tag = nullptr, *tag2 = nullptr, *d_tab = nullptr, *d_rank = nullptr, *d_blk_of = nullptr, *h_tab = nullptr;
    unsigned long long *best = nullptr, *d_scan = nullptr;
    int64_t best_cap = 0, blk_of_cap = 0, compactions = 0;
    uint32_t scan_cap = 0, scan_epoch = 0, pin_next = 0;
    uint8_t *d_tbl = nullptr, *h_tbl_pin = nullptr;
    AsyncCounters *d_cnt = nullptr, *h_cnt = nullptr;
  } async;
};
static void a2_free(mpmhip2d_ctx *m) {
  auto &A = m->async;
  hipFree(A.rec); hipFree(A.rec2); hipFree(A.tag); hipFree(A.tag2); hipFree(A.d_tab); hipFree(A.d_rank); hipFree(A.d_blk_of);
  hipFree(A.best); hipFree(A.d_scan); hipFree(A.d_tbl); hipFree(A.d_cnt);
  if (A.h_tab) hipHostFree(A.h_tab);
  if (A.h_tbl_pin) hipHostFree(A.h_tbl_pin);
  if (A.h_cnt) hipHostFree(A.h_cnt);
  A.rec = A.rec2 = nullptr; A.tag = A.tag2 = A.d_tab = A.d_rank = A.d_blk_of = A.h_tab = nullptr;
  A.best = A.d_scan = nullptr; A.d_tbl = A.h_tbl_pin = nullptr; A.d_cnt = A.h_cnt = nullptr;
  A.blk_of_cap = 0;  // (d_blk_of is gone: the next load_pools must allocate it again)
  A.resident = false;
}
static int a2_drop_view(mpmhip2d_ctx *m);
static int a2_grow_particles(mpmhip2d_ctx *m, int64_t need);
int mpmhip2d_async_step(mpmhip2d_ctx *m, float dt);
static thread_local std::string g_2d_create_error;
static int fail2d(mpmhip2d_ctx *m, int code, const std::string &msg) {
  (m ? m->err : g_2d_create_error) = msg;
  return code;
}
#define HIPCHK2D(m, call)                                                                                      \
  do {                                                                                                         \
    hipError_t e_ = (call);                                                                                    \
    if (e_ != hipSuccess) return fail2d((m), MPMHIP_EHIP, std::string(#call " failed: ") + hipGetErrorString(e_)); \
  } while (0)

const char *mpmhip2d_last_error(const mpmhip2d_ctx *m) { return m ? m->err.c_str() : g_2d_create_error.c_str(); }

int mpmhip2d_create(const mpmhip2d_config *cfg, mpmhip2d_ctx **out) {
  if (!cfg || !out) return fail2d(nullptr, MPMHIP_EINVAL, "null argument");
  *out = nullptr;
  for (int k = 0; k < 2; k++)
    if (cfg->res[k] < 8 || cfg->res[k] > 16384) return fail2d(nullptr, MPMHIP_EINVAL, "res outside [8,16384]");
  if (!(cfg->dx > 0) || !(cfg->dt >= 0) || cfg->max_particles <= 0) return fail2d(nullptr, MPMHIP_EINVAL, "dx > 0, dt >= 0, max_particles > 0 required");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail2d(nullptr, MPMHIP_EHIP, "no HIP device available (libmpmhip has no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail2d(nullptr, MPMHIP_EINVAL, "no such device");
  mpmhip2d_ctx *m = new (std::nothrow) mpmhip2d_ctx;
  if (!m) return fail2d(nullptr, MPMHIP_ENOMEM, "host allocation failed");
  m->device = cfg->device;
  mpm2d::Params &P = m->P;
  P.res[0] = cfg->res[0]; P.res[1] = cfg->res[1];
  P.dx = cfg->dx; P.idx = 1.0f / cfg->dx; P.dt = cfg->dt; P.t = 0.0f;
  m->base_dt = cfg->dt;
  P.g[0] = cfg->gravity[0]; P.g[1] = cfg->gravity[1];
  P.particle_gravity = cfg->particle_gravity; P.apic_damping = cfg->apic_damping; P.rpic_damping = cfg->rpic_damping;
  P.clean_boundary = cfg->clean_boundary; P.particle_collision = cfg->particle_collision; P.clamp_pos = 1;
  memset(&m->LS, 0, sizeof m->LS);
  m->cap = cfg->max_particles;
  const size_t c = (size_t)m->cap, nodes = (size_t)(P.res[0] + 1) * (P.res[1] + 1);
  hipError_t e = hipSetDevice(m->device);
  auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  A(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
  A(dmalloc(&m->x, 2 * c)); A(dmalloc(&m->v, 2 * c)); A(dmalloc(&m->F, 4 * c)); A(dmalloc(&m->B, 4 * c)); A(dmalloc(&m->aux, c));
  A(dmalloc(&m->gid, c)); A(dmalloc(&m->pid, c)); A(dmalloc(&m->grid, 3 * nodes)); A(dmalloc(&m->n_dead, 1));
  A(dmalloc(&m->d_groups, (size_t)MPMHIP_MAX_GROUPS));
  if (e == hipSuccess) e = hipMemset(m->n_dead, 0, sizeof(unsigned int));
  if (e != hipSuccess) {
    const int rc = fail2d(nullptr, MPMHIP_ENOMEM, std::string("mpmhip2d_create: ") + hipGetErrorString(e));
    mpmhip2d_destroy(m);
    return rc;
  }
  *out = m;
  return MPMHIP_OK;
}

void mpmhip2d_destroy(mpmhip2d_ctx *m) {
  if (!m) return;
  hipSetDevice(m->device);
  if (m->stream) { hipStreamSynchronize(m->stream); hipStreamDestroy(m->stream); }
  hipFree(m->x); hipFree(m->v); hipFree(m->F); hipFree(m->B); hipFree(m->aux); hipFree(m->gid); hipFree(m->pid); hipFree(m->grid);
  hipFree(m->n_dead); hipFree(m->d_groups);
  hipFree(m->d_rb); hipFree(m->d_smp); hipFree(m->d_elems); hipFree(m->d_mind); hipFree(m->d_tags); hipFree(m->d_states); hipFree(m->d_bnd);
  a2_free(m);
  hipFree(m->d_smp_rank); hipFree(m->d_ls_tmp);
  for (int k = 0; k < 2; k++) { hipFree(m->d_ls_keys[k]); hipFree(m->d_ls_vals[k]); }
  delete m;
}

// level set in the plane: the shapes of mpmhip_shape with z ignored (plane = line n.x + d, sphere = disc, cuboid = box with
// p[2], p[5] spanning z = 0); two key frames like mpmhip_set_levelset_keyframes when n1 >= 0
int mpmhip2d_set_levelset(mpmhip2d_ctx *m, int32_t n0, const mpmhip_shape *shapes0, int32_t n1, const mpmhip_shape *shapes1,
                          float t0, float t1, float friction) {
  if (!m || n0 < 0 || n0 > MPMHIP_MAX_SHAPES || n1 > MPMHIP_MAX_SHAPES || (n0 > 0 && !shapes0) || (n1 > 0 && !shapes1)) return MPMHIP_EINVAL;
  if (n1 >= 0 && !(t1 > t0)) return fail2d(m, MPMHIP_EINVAL, "key frame times must satisfy t0 < t1");
  HIPCHK2D(m, hipSetDevice(m->device));
  HIPCHK2D(m, hipStreamSynchronize(m->stream));
  LevelSetDev &L = m->LS;
  memset(&L, 0, sizeof L);
  L.n = n0; L.friction = friction; L.dynamic = n1 >= 0; L.n1 = n1 >= 0 ? n1 : 0; L.t0 = t0; L.t1 = t1;
  auto put = [](ShapeDev &d, const mpmhip_shape &s) {
    d.type = s.type; d.inside_out = s.inside_out;
    for (int k = 0; k < 6; k++) d.p[k] = s.p[k];
    if (s.type == 2) { d.p[2] = -1e30f; d.p[5] = 1e30f; }  // a box in the plane: unbounded along z
    if (s.type == 1) d.p[2] = 0.0f;
    if (s.type == 0) d.p[2] = 0.0f;
  };
  for (int i = 0; i < n0; i++) put(L.s[i], shapes0[i]);
  for (int i = 0; i < L.n1; i++) put(L.s1[i], shapes1[i]);
  return MPMHIP_OK;
}

int mpmhip2d_set_dirichlet(mpmhip2d_ctx *m, int32_t enabled, float distance_left, float distance_right, float velocity_left,
                           float velocity_right) {
  if (!m) return MPMHIP_EINVAL;
  m->P.dirichlet = enabled != 0;
  m->P.dl = distance_left; m->P.dr = distance_right; m->P.vl = velocity_left; m->P.vr = velocity_right;
  return MPMHIP_OK;
}

int mpmhip2d_add_group(mpmhip2d_ctx *m, int32_t material, const float params[MPMHIP_NPARAM]) {
  if (!m || !params) return MPMHIP_EINVAL;
  if (material < MPMHIP_VISCO || material > MPMHIP_ELASTIC) return fail2d(m, MPMHIP_EINVAL, "unknown material id");
  if ((int)m->groups.size() >= MPMHIP_MAX_GROUPS) return fail2d(m, MPMHIP_ECAPACITY, "too many particle groups");
  if (!(params[0] > 0) || !(params[1] > 0)) return fail2d(m, MPMHIP_EINVAL, "group mass and vol must be > 0");
  GroupParams g;
  memset(&g, 0, sizeof g);
  memcpy(g.p, params, sizeof g.p);
  g.type = material;
  m->groups.push_back(g);
  HIPCHK2D(m, hipSetDevice(m->device));
  HIPCHK2D(m, hipStreamSynchronize(m->stream));
  HIPCHK2D(m, hipMemcpy(m->d_groups, m->groups.data(), sizeof(GroupParams) * m->groups.size(), hipMemcpyHostToDevice));
  return (int)m->groups.size() - 1;
}

int mpmhip2d_add_particles(mpmhip2d_ctx *m, int32_t group, int64_t n, const float *x, const float *v, const float *F, const float *B,
                           const float *aux) {
  if (!m || n < 0 || (n > 0 && !x)) return MPMHIP_EINVAL;
  if (group < 0 || group >= (int)m->groups.size()) return fail2d(m, MPMHIP_EINVAL, "unknown group");
  if (n == 0) return MPMHIP_OK;
  HIPCHK2D(m, hipSetDevice(m->device));
  if (int rc = a2_drop_view(m)) return rc;  // (resident async stepper: arrays that only mirror the pools go first)
  if (m->n + n > m->cap) {
    if (!m->async.resident) return fail2d(m, MPMHIP_ECAPACITY, "particle capacity exceeded");
    if (int rc = a2_grow_particles(m, m->n + n)) return rc;  // (a resident stepper's arrays hold one batch or one working set)
  }
  HIPCHK2D(m, hipStreamSynchronize(m->stream));
  const int mat = m->groups[group].type;
  const float aux0 = (mat == MPMHIP_SNOW || mat == MPMHIP_WATER) ? 1.0f : (mat == MPMHIP_VISCO ? 1000.0f : 0.0f);
  std::vector<float> h;
  auto put = [&](float *dst, const float *src, int width, const float *dflt) -> hipError_t {
    if (!src) {
      h.resize((size_t)n * width);
      for (int64_t i = 0; i < n; i++)
        for (int k = 0; k < width; k++) h[(size_t)i * width + k] = dflt[k];
      src = h.data();
    }
    return hipMemcpy(dst + m->n * width, src, sizeof(float) * n * width, hipMemcpyHostToDevice);
  };
  const float zero4[4] = {0, 0, 0, 0}, eye[4] = {1, 0, 0, 1}, a0[1] = {aux0};
  HIPCHK2D(m, put(m->x, x, 2, zero4));
  HIPCHK2D(m, put(m->v, v, 2, zero4));
  HIPCHK2D(m, put(m->F, F, 4, eye));
  HIPCHK2D(m, put(m->B, B, 4, zero4));
  HIPCHK2D(m, put(m->aux, aux, 1, a0));
  std::vector<int32_t> ids((size_t)n), gs((size_t)n, group);
  for (int64_t i = 0; i < n; i++) ids[i] = m->next_pid + (int32_t)i;
  m->next_pid += (int32_t)n;
  HIPCHK2D(m, hipMemcpy(m->pid + m->n, ids.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
  HIPCHK2D(m, hipMemcpy(m->gid + m->n, gs.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
  m->n += n;
  return MPMHIP_OK;
}

static mpm2d::RigidArgs2 rigid_args2(mpmhip2d_ctx *m) {
  mpm2d::RigidArgs2 R;
  memset(&R, 0, sizeof R);
  R.enabled = m->rigid_enabled && m->bodies.size() > 1;  // has_rigid_body()
  R.mind = m->d_mind; R.tags = m->d_tags; R.rb = m->d_rb; R.states = m->d_states; R.bnd = m->d_bnd;
  R.penalty = m->penalty; R.pushing_force = m->pushing_force;
  return R;
}
// rasterize_rigid_boundary + gather_cdf (src/mpm.cpp:466-472, 506-508)
static int rigid2_pre(mpmhip2d_ctx *m) {
  const size_t nodes = (size_t)(m->P.res[0] + 1) * (m->P.res[1] + 1);
  HIPCHK2D(m, hipMemsetAsync(m->d_mind, 0xFF, sizeof(unsigned long long) * nodes, m->stream));
  HIPCHK2D(m, hipMemsetAsync(m->d_tags, 0, sizeof(uint32_t) * nodes, m->stream));
  const mpm2d::RigidArgs2 R = rigid_args2(m);
  const uint32_t ns = (uint32_t)m->h_smp.size();
  if (ns)
    hipLaunchKernelGGL(mpm2d::k2_cdf_rasterize, dim3((ns + 255) / 256), dim3(256), 0, m->stream, m->P.res[0], m->P.res[1], m->P.dx, m->P.idx, R,
                       (const mpm2d::Sample2 *)m->d_smp, (const float *)m->d_elems, ns);
  if (m->n)
    hipLaunchKernelGGL(mpm2d::k2_gather_cdf, dim3((unsigned)((m->n + 255) / 256)), dim3(256), 0, m->stream, m->P.res[0], m->P.res[1], m->P.dx,
                       m->P.idx, R, m->n, (const float *)m->x, (const int32_t *)m->pid);
  HIPCHK2D(m, hipGetLastError());
  return MPMHIP_OK;
}
// rigid_body_levelset_collision(current_t, dt) (src/mpm_rigid_body.cpp:347-387), between normalize_grid and the grid boundary
// condition (src/mpm.cpp:535-538); see do_rigid_ls_collision of rigid_api.h
static int rigid2_ls_collision(mpmhip2d_ctx *m) {
  const uint32_t n = (uint32_t)m->h_smp.size();
  if (n == 0 || m->LS.n == 0) return MPMHIP_OK;
  if (m->ls_cap < n) {
    for (int k = 0; k < 2; k++) { (void)hipFree(m->d_ls_keys[k]); (void)hipFree(m->d_ls_vals[k]); m->d_ls_keys[k] = nullptr; m->d_ls_vals[k] = nullptr; }
    (void)hipFree(m->d_ls_tmp); m->d_ls_tmp = nullptr; m->ls_tmp_bytes = 0;
    const size_t cap = (size_t)n + n / 4 + 1024;
    for (int k = 0; k < 2; k++) { HIPCHK2D(m, dmalloc(&m->d_ls_keys[k], cap)); HIPCHK2D(m, dmalloc(&m->d_ls_vals[k], cap)); }
    size_t bytes = 0;
    HIPCHK2D(m, hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, m->d_ls_keys[0], m->d_ls_keys[1], m->d_ls_vals[0], m->d_ls_vals[1], (int)cap, 0, 64, m->stream));
    HIPCHK2D(m, hipMalloc(&m->d_ls_tmp, bytes));
    m->ls_tmp_bytes = bytes;
    m->ls_cap = (uint32_t)cap;
  }
  if (m->n_ranked < n) {  // boundary particles added since: behind everybody else, in creation order (appended to `particles`)
    std::vector<uint32_t> rk(n);
    HIPCHK2D(m, hipStreamSynchronize(m->stream));
    if (m->n_ranked) HIPCHK2D(m, hipMemcpy(rk.data(), m->d_smp_rank, sizeof(uint32_t) * m->n_ranked, hipMemcpyDeviceToHost));
    for (uint32_t s = m->n_ranked; s < n; s++) rk[s] = s;
    (void)hipFree(m->d_smp_rank); m->d_smp_rank = nullptr;
    HIPCHK2D(m, dmalloc(&m->d_smp_rank, (size_t)n + n / 4 + 1024));
    HIPCHK2D(m, hipMemcpy(m->d_smp_rank, rk.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    m->n_ranked = n;
  }
  hipLaunchKernelGGL(mpm2d::k2_ls_keys, dim3((n + 255) / 256), dim3(256), 0, m->stream, m->P.idx, (const mpm2d::Rigid2 *)m->d_rb,
                     (const mpm2d::Sample2 *)m->d_smp, n, (const uint32_t *)m->d_smp_rank, m->d_ls_keys[0], m->d_ls_vals[0]);
  size_t bytes = m->ls_tmp_bytes;
  HIPCHK2D(m, hipcub::DeviceRadixSort::SortPairs(m->d_ls_tmp, bytes, m->d_ls_keys[0], m->d_ls_keys[1], m->d_ls_vals[0], m->d_ls_vals[1], (int)n, 0, 64, m->stream));
  mpm2d::Restitution2 rest;
  memset(&rest, 0, sizeof rest);
  for (size_t b = 1; b < m->bodies.size(); b++) rest.e[b] = m->bodies[b].cfg.restitution;
  hipLaunchKernelGGL(mpm2d::k2_ls_collide, dim3(1), dim3(1024), 0, m->stream, m->P, m->LS, m->d_rb, (int)m->bodies.size(),
                     (const mpm2d::Sample2 *)m->d_smp, (const uint32_t *)m->d_ls_vals[1], n, m->d_smp_rank, rest);
  HIPCHK2D(m, hipGetLastError());
  return MPMHIP_OK;
}
static int rigid2_advect(mpmhip2d_ctx *m) {
  mpm2d::Steps2 st;
  memset(&st, 0, sizeof st);
  const float dt = m->P.dt, rad = (float)(M_PI / 180.0);
  for (size_t b = 1; b < m->bodies.size(); b++) {
    const auto &cfg = m->bodies[b].cfg;
    float o[3];
    if (cfg.scripted_position) {
      st.s[b].has_pos = 1;
      cfg.scripted_position(cfg.position_user, m->t, o); st.s[b].p0[0] = o[0]; st.s[b].p0[1] = o[1];
      cfg.scripted_position(cfg.position_user, m->t + dt, o); st.s[b].p1[0] = o[0]; st.s[b].p1[1] = o[1];
    }
    if (cfg.scripted_rotation) {
      st.s[b].has_rot = 1;
      cfg.scripted_rotation(cfg.rotation_user, m->t, o); st.s[b].a0 = o[0] * rad;
      cfg.scripted_rotation(cfg.rotation_user, m->t + dt, o); st.s[b].a1 = o[0] * rad;
    }
  }
  hipLaunchKernelGGL(mpm2d::k2_rigid_advect, dim3(1), dim3(64), 0, m->stream, m->d_rb, (int)m->bodies.size(), st, dt, m->P.g[0], m->P.g[1]);
  HIPCHK2D(m, hipGetLastError());
  return MPMHIP_OK;
}

static int substep2d(mpmhip2d_ctx *m);
int mpmhip2d_substep(mpmhip2d_ctx *m) {  // MPM<2>::substep, src/mpm.cpp:452-575
  if (!m) return MPMHIP_EINVAL;
  if (m->async.resident) return fail2d(m, MPMHIP_EINVAL, "an asynchronous stepper steps with mpmhip2d_async_step (mpmhip2d_step forwards to it)");
  return substep2d(m);
}
static int substep2d(mpmhip2d_ctx *m) {
  HIPCHK2D(m, hipSetDevice(m->device));
  const size_t nodes = (size_t)(m->P.res[0] + 1) * (m->P.res[1] + 1);
  m->P.t = m->t;
  const dim3 pg((unsigned)std::max<int64_t>((m->n + 255) / 256, 1)), gg((unsigned)((nodes + 255) / 256)), wg(256);
  HIPCHK2D(m, hipMemsetAsync(m->grid, 0, sizeof(float) * 3 * nodes, m->stream));
  const mpm2d::RigidArgs2 R = rigid_args2(m);
  if (R.enabled) {
    if (m->joints.n)  // articulate, between the sort and rasterize_rigid_boundary (src/mpm.cpp:466-471)
      hipLaunchKernelGGL(mpm2d::k2_articulate, dim3(1), dim3(64), 0, m->stream, m->d_rb, m->joints, m->joint_iterations);
    if (int rc = rigid2_pre(m)) return rc;
  }
  if (m->n)
    hipLaunchKernelGGL(mpm2d::k_p2g, pg, wg, 0, m->stream, m->P, m->n, (const float *)m->x, m->v, (const float *)m->F,
                       (const float *)m->B, (const float *)m->aux, (const int32_t *)m->gid, (const int32_t *)m->pid,
                       (const GroupParams *)m->d_groups, m->grid, R);
  if (R.enabled) hipLaunchKernelGGL(mpm2d::k2_rigid_apply_tmp, dim3(1), dim3(64), 0, m->stream, m->d_rb, (int)m->bodies.size());
  if (R.enabled && m->ls_collision) {
    if (int rc = rigid2_ls_collision(m)) return rc;
  }
  hipLaunchKernelGGL(mpm2d::k_grid, gg, wg, 0, m->stream, m->P, m->LS, m->grid);
  if (m->n)
    hipLaunchKernelGGL(mpm2d::k_g2p, pg, wg, 0, m->stream, m->P, m->LS, m->n, m->x, m->v, m->F, m->B, m->aux, (const int32_t *)m->gid,
                       m->pid, (const GroupParams *)m->d_groups, (const float *)m->grid, m->n_dead, R);
  if (R.enabled) {
    hipLaunchKernelGGL(mpm2d::k2_rigid_apply_tmp, dim3(1), dim3(64), 0, m->stream, m->d_rb, (int)m->bodies.size());
    if (int rc = rigid2_advect(m)) return rc;
  }
  HIPCHK2D(m, hipGetLastError());
  m->t += m->P.dt;
  return MPMHIP_OK;
}

// ---- CPIC rigid bodies in 2D: add_particles(type='rigid') of MPM<2> (src/mpm_rigid_body.cpp:130-252, dim = 2 branches)
int mpmhip2d_set_rigid_levelset_collision(mpmhip2d_ctx *m, int32_t enabled) {
  if (!m) return MPMHIP_EINVAL;
  m->ls_collision = enabled != 0;
  return MPMHIP_OK;
}
int mpmhip2d_set_rigid_coupling(mpmhip2d_ctx *m, float penalty, float pushing_force) {
  if (!m) return MPMHIP_EINVAL;
  m->penalty = penalty; m->pushing_force = pushing_force;
  return MPMHIP_OK;
}
int mpmhip2d_add_rigid_body(mpmhip2d_ctx *m, const mpmhip2d_rigid_config *cfg, int64_t n_segments, const float *segments) {
  if (!m || !cfg || !segments || n_segments <= 0) return MPMHIP_EINVAL;
  if (m->async.resident) return fail2d(m, MPMHIP_EINVAL, "rigid bodies cannot be combined with asynchronous stepping");
  HIPCHK2D(m, hipSetDevice(m->device));
  HIPCHK2D(m, hipStreamSynchronize(m->stream));
  const size_t nodes = (size_t)(m->P.res[0] + 1) * (m->P.res[1] + 1);
  if (!m->rigid_enabled) {
    hipError_t e = hipSuccess;
    auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    A(dmalloc(&m->d_rb, (size_t)mpm2d::MAX_RIGID2)); A(dmalloc(&m->d_mind, nodes)); A(dmalloc(&m->d_tags, nodes));
    A(dmalloc(&m->d_states, (size_t)m->cap)); A(dmalloc(&m->d_bnd, (size_t)m->cap));
    if (e != hipSuccess) return fail2d(m, MPMHIP_ENOMEM, std::string("rigid coupling: ") + hipGetErrorString(e));
    HIPCHK2D(m, hipMemset(m->d_rb, 0, sizeof(mpm2d::Rigid2) * mpm2d::MAX_RIGID2));
    HIPCHK2D(m, hipMemset(m->d_states, 0, sizeof(uint32_t) * (size_t)m->cap));
    HIPCHK2D(m, hipMemset(m->d_bnd, 0, sizeof(mpm2d::Bnd2) * (size_t)m->cap));
    m->bodies.clear();
    m->bodies.emplace_back();  // the background body
    m->rigid_enabled = true;
  }
  if ((int)m->bodies.size() >= mpm2d::MAX_RIGID2) return fail2d(m, MPMHIP_ECAPACITY, "too many rigid bodies");
  const bool spos = cfg->scripted_position != nullptr, srot = cfg->scripted_rotation != nullptr;
  if (!cfg->recenter && !(spos && srot)) return fail2d(m, MPMHIP_EINVAL, "recenter = 0 needs a scripted position and rotation");
  std::vector<float> seg(segments, segments + 4 * n_segments);
  const float sc[2] = {cfg->scale[0] != 0 ? cfg->scale[0] : 1.0f, cfg->scale[1] != 0 ? cfg->scale[1] : 1.0f};
  for (int64_t e = 0; e < n_segments; e++) {
    if (cfg->reverse_vertices) { std::swap(seg[4 * e], seg[4 * e + 2]); std::swap(seg[4 * e + 1], seg[4 * e + 3]); }
    for (int q = 0; q < 2; q++) for (int k = 0; k < 2; k++) seg[4 * e + 2 * q + k] *= sc[k];
  }
  // mass, centre of mass, inertia: the segments as a shell of line density `density` (the rigid body the reference build
  // is tested against, oracle/taichi_shim/.../rigid_body_shim.h, treats 2D bodies this way whether codimensional or not)
  const double density = cfg->density > 0 ? cfg->density : (cfg->codimensional ? 40.0 : 400.0);
  double M = 0, com[2] = {0, 0}, S00 = 0, S11 = 0;
  for (int64_t e = 0; e < n_segments; e++) {
    const double a[2] = {seg[4 * e], seg[4 * e + 1]}, b[2] = {seg[4 * e + 2], seg[4 * e + 3]};
    const double mm = std::sqrt((b[0] - a[0]) * (b[0] - a[0]) + (b[1] - a[1]) * (b[1] - a[1])) * density;
    M += mm;
    for (int k = 0; k < 2; k++) com[k] += mm * 0.5 * (a[k] + b[k]);
    S00 += mm / 6.0 * (a[0] * a[0] + b[0] * b[0] + (a[0] + b[0]) * (a[0] + b[0]));
    S11 += mm / 6.0 * (a[1] * a[1] + b[1] * b[1] + (a[1] + b[1]) * (a[1] + b[1]));
  }
  if (!(M > 0)) return fail2d(m, MPMHIP_EINVAL, "rigid body without mass");
  com[0] /= M; com[1] /= M;
  const double I = (S00 - M * com[0] * com[0]) + (S11 - M * com[1] * com[1]);
  if (!cfg->recenter) com[0] = com[1] = 0.0;
  for (int64_t e = 0; e < n_segments; e++)
    for (int q = 0; q < 2; q++) for (int k = 0; k < 2; k++) seg[4 * e + 2 * q + k] -= (float)com[k];
  mpm2d::Rigid2 D;
  memset(&D, 0, sizeof D);
  float o[3] = {cfg->initial_position[0], cfg->initial_position[1], 0};
  if (spos) cfg->scripted_position(cfg->position_user, m->t, o);
  D.pos[0] = o[0]; D.pos[1] = o[1];
  float ang[3] = {cfg->initial_rotation, 0, 0};
  if (srot) cfg->scripted_rotation(cfg->rotation_user, m->t, ang);
  D.angle = ang[0] * (float)(M_PI / 180.0);
  D.vel[0] = spos ? 0.0f : cfg->initial_velocity[0]; D.vel[1] = spos ? 0.0f : cfg->initial_velocity[1];
  D.omega = srot ? 0.0f : cfg->initial_angular_velocity;
  D.mass = (float)M; D.inv_mass = spos ? 0.0f : (float)(1.0 / M); D.inv_I = srot ? 0.0f : (float)(1.0 / I);
  D.fric[0] = cfg->friction[0]; D.fric[1] = cfg->friction[1];
  D.lin_damp = cfg->linear_damping; D.ang_damp = cfg->angular_damping;
  D.scripted = (spos ? 1 : 0) | (srot ? 2 : 0);
  // boundary particles (src/mpm_rigid_body.cpp:196-207): max(ceil(length / dx), 2) per segment, at the mid points of equal parts
  const int body = (int)m->bodies.size();
  const size_t elem0 = m->h_elems.size() / 4;
  const float cs = std::cos(D.angle), sn = std::sin(D.angle);
  int32_t allocated = 0;
  for (int64_t e = 0; e < n_segments; e++) {
    const float a[2] = {seg[4 * e], seg[4 * e + 1]}, b[2] = {seg[4 * e + 2], seg[4 * e + 3]};
    const float len = std::sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]));
    const int ns = std::max((int)std::ceil(len * m->P.idx), 2);
    for (int j = 0; j < ns; j++) {
      const float t = (0.5f + j) / ns;
      mpm2d::Sample2 s;
      s.off[0] = a[0] * (1.0f - t) + b[0] * t; s.off[1] = a[1] * (1.0f - t) + b[1] * t;
      s.body = body; s.elem = (int)(elem0 + e);
      const float w[2] = {(cs * s.off[0] - sn * s.off[1] + D.pos[0]) * m->P.idx, (sn * s.off[0] + cs * s.off[1] + D.pos[1]) * m->P.idx};
      const bool near_wall = w[0] < 7.0f || w[1] < 7.0f || w[0] - m->P.res[0] > -7.0f || w[1] - m->P.res[1] > -7.0f;
      if (!near_wall) { m->h_smp.push_back(s); m->h_smp_id.push_back(m->next_pid + allocated); }
      allocated++;
    }
  }
  m->next_pid += allocated;  // boundary particles take creation ids from the same counter (src/particle_allocator.h:68-74)
  m->h_elems.insert(m->h_elems.end(), seg.begin(), seg.end());
  (void)hipFree(m->d_smp); (void)hipFree(m->d_elems);
  m->d_smp = nullptr; m->d_elems = nullptr;
  HIPCHK2D(m, dmalloc(&m->d_smp, std::max<size_t>(m->h_smp.size(), 1)));
  HIPCHK2D(m, dmalloc(&m->d_elems, std::max<size_t>(m->h_elems.size(), 4)));
  if (!m->h_smp.empty()) HIPCHK2D(m, hipMemcpy(m->d_smp, m->h_smp.data(), sizeof(mpm2d::Sample2) * m->h_smp.size(), hipMemcpyHostToDevice));
  HIPCHK2D(m, hipMemcpy(m->d_elems, m->h_elems.data(), sizeof(float) * m->h_elems.size(), hipMemcpyHostToDevice));
  HIPCHK2D(m, hipMemcpy(m->d_rb + body, &D, sizeof D, hipMemcpyHostToDevice));
  mpmhip2d_ctx::HostRigid2 H;
  H.cfg = *cfg; H.mass = (float)M; H.inertia = (float)I; H.first_elem = (int64_t)elem0; H.n_elems = n_segments;
  m->bodies.push_back(H);
  return body;
}
// out[10]: position 2, angle (radians), velocity 2, angular velocity, mass, inv_mass, inertia, inv_inertia
// general_action("add_articulation") of MPM<2>: the 'rotation' joint (the only one the reference's 2D scenes use; its 'frozen' and
// 'stepper' are TC_NOT_IMPLEMENTED for dim = 2, src/articulation.cpp:65-67,317)
int mpmhip2d_add_articulation(mpmhip2d_ctx *m, const mpmhip_joint_config *cfg) {
  if (!m || !cfg) return MPMHIP_EINVAL;
  const int nb = m->rigid_enabled ? (int)m->bodies.size() : 0;
  if (cfg->type != MPMHIP_JOINT_ROTATION) return fail2d(m, MPMHIP_EINVAL, "add_articulation: only type='rotation' is built for 2D simulations");
  if (cfg->obj0 < 1 || cfg->obj0 >= nb) return fail2d(m, MPMHIP_EINVAL, "add_articulation: obj0 = " + std::to_string(cfg->obj0) + " is not a rigid body of this simulation");
  if (cfg->obj1 < 0 || cfg->obj1 >= nb) return fail2d(m, MPMHIP_EINVAL, "add_articulation: obj1 = " + std::to_string(cfg->obj1) + " is not a rigid body of this simulation");
  if (m->joints.n >= mpm2d::MAX_JOINTS2) return fail2d(m, MPMHIP_ECAPACITY, "at most " + std::to_string(mpm2d::MAX_JOINTS2) + " articulations");
  mpm2d::Joint2 &J = m->joints.j[m->joints.n++];
  J.obj0 = cfg->obj0; J.obj1 = cfg->obj1;
  J.I0 = m->bodies[cfg->obj0].inertia;
  J.I1 = cfg->obj1 == 0 ? 1.0f : m->bodies[cfg->obj1].inertia;  // (the background body: inertia 1, set_as_background)
  return MPMHIP_OK;
}
int mpmhip2d_set_articulation_iterations(mpmhip2d_ctx *m, int32_t n) {
  if (!m || n < 0) return MPMHIP_EINVAL;
  m->joint_iterations = n;
  return MPMHIP_OK;
}
int mpmhip2d_rigid_get_state(mpmhip2d_ctx *m, int32_t id, float *out) {
  if (!m || !out) return MPMHIP_EINVAL;
  if (!m->rigid_enabled || id < 1 || id >= (int)m->bodies.size()) return fail2d(m, MPMHIP_EINVAL, "no such rigid body");
  HIPCHK2D(m, hipSetDevice(m->device));
  HIPCHK2D(m, hipStreamSynchronize(m->stream));
  mpm2d::Rigid2 D;
  HIPCHK2D(m, hipMemcpy(&D, m->d_rb + id, sizeof D, hipMemcpyDeviceToHost));
  out[0] = D.pos[0]; out[1] = D.pos[1]; out[2] = D.angle; out[3] = D.vel[0]; out[4] = D.vel[1]; out[5] = D.omega;
  out[6] = D.mass; out[7] = D.inv_mass; out[8] = m->bodies[id].inertia; out[9] = D.inv_I;
  return MPMHIP_OK;
}
// the body's segments in world space (write_rigid_body's .poly file, src/visualize.cpp:105-130): 4 floats per segment
int64_t mpmhip2d_rigid_get_mesh(mpmhip2d_ctx *m, int32_t id, int64_t cap_segments, float *out) {
  if (!m) return MPMHIP_EINVAL;
  if (!m->rigid_enabled || id < 1 || id >= (int)m->bodies.size()) return fail2d(m, MPMHIP_EINVAL, "no such rigid body");
  if (hipSetDevice(m->device) != hipSuccess || hipStreamSynchronize(m->stream) != hipSuccess) return MPMHIP_EHIP;
  mpm2d::Rigid2 D;
  HIPCHK2D(m, hipMemcpy(&D, m->d_rb + id, sizeof D, hipMemcpyDeviceToHost));
  const auto &B = m->bodies[id];
  const float cs = std::cos(D.angle), sn = std::sin(D.angle);
  if (out)
    for (int64_t e = 0; e < std::min<int64_t>(B.n_elems, cap_segments); e++)
      for (int q = 0; q < 2; q++) {
        const float *v = &m->h_elems[(size_t)(B.first_elem + e) * 4 + 2 * q];
        out[4 * e + 2 * q] = cs * v[0] - sn * v[1] + D.pos[0];
        out[4 * e + 2 * q + 1] = sn * v[0] + cs * v[1] + D.pos[1];
      }
  return B.n_elems;
}
// world positions of the boundary particles of body id (id < 0: all); returns the count
int64_t mpmhip2d_rigid_get_samples(mpmhip2d_ctx *m, int32_t id, int64_t cap, float *pos) {
  if (!m) return MPMHIP_EINVAL;
  if (!m->rigid_enabled) return 0;
  if (hipSetDevice(m->device) != hipSuccess) return MPMHIP_EHIP;
  const uint32_t ns = (uint32_t)m->h_smp.size();
  std::vector<float> w((size_t)ns * 2);
  if (ns && pos) {
    float *d = nullptr;
    HIPCHK2D(m, dmalloc(&d, (size_t)ns * 2));
    hipLaunchKernelGGL(mpm2d::k2_sample_positions, dim3((ns + 255) / 256), dim3(256), 0, m->stream, (const mpm2d::Rigid2 *)m->d_rb,
                       (const mpm2d::Sample2 *)m->d_smp, ns, d);
    hipError_t e = hipMemcpyAsync(w.data(), d, sizeof(float) * w.size(), hipMemcpyDeviceToHost, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    (void)hipFree(d);
    HIPCHK2D(m, e);
  }
  int64_t n = 0;
  for (size_t s = 0; s < m->h_smp.size(); s++) {
    if (id >= 0 && m->h_smp[s].body != id) continue;
    if (n < cap && pos) { pos[2 * n] = w[2 * s]; pos[2 * n + 1] = w[2 * s + 1]; }
    n++;
  }
  return n;
}
// rasterize_rigid_boundary + gather_cdf as a phase (parity tests), then: dense (res+1)^2 grid states / distances, and per
// live particle in slot order states, boundary distance, normal (2), near flag
int mpmhip2d_cdf_phase(mpmhip2d_ctx *m) {
  if (!m) return MPMHIP_EINVAL;
  HIPCHK2D(m, hipSetDevice(m->device));
  if (!(m->rigid_enabled && m->bodies.size() > 1)) return MPMHIP_OK;
  return rigid2_pre(m);
}
int mpmhip2d_download_cdf(mpmhip2d_ctx *m, uint32_t *states, float *distance) {
  if (!m || !states || !distance) return MPMHIP_EINVAL;
  const size_t nodes = (size_t)(m->P.res[0] + 1) * (m->P.res[1] + 1);
  memset(states, 0, nodes * 4); memset(distance, 0, nodes * 4);
  if (!m->rigid_enabled) return MPMHIP_OK;
  HIPCHK2D(m, hipSetDevice(m->device));
  HIPCHK2D(m, hipStreamSynchronize(m->stream));
  std::vector<unsigned long long> hm(nodes);
  std::vector<uint32_t> ht(nodes);
  HIPCHK2D(m, hipMemcpy(hm.data(), m->d_mind, nodes * 8, hipMemcpyDeviceToHost));
  HIPCHK2D(m, hipMemcpy(ht.data(), m->d_tags, nodes * 4, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < nodes; i++) {
    const bool has = hm[i] != ~0ull;
    states[i] = (ht[i] & 0xFFFFFFu) | (has ? ((uint32_t)(hm[i] & 0xFFu) << 24) : 0u);
    if (has) { const uint32_t bits = (uint32_t)(hm[i] >> 32); float d; memcpy(&d, &bits, 4); distance[i] = d * m->P.dx; }
  }
  return MPMHIP_OK;
}
int64_t mpmhip2d_download_colours(mpmhip2d_ctx *m, int64_t capacity, uint32_t *states, float *distance, float *normal, int32_t *near) {
  if (!m) return MPMHIP_EINVAL;
  if (hipSetDevice(m->device) != hipSuccess || hipStreamSynchronize(m->stream) != hipSuccess) return MPMHIP_EHIP;
  const size_t n = (size_t)m->n;
  std::vector<int32_t> hp(n);
  std::vector<uint32_t> hs(n, 0u);
  std::vector<mpm2d::Bnd2> hb(n);
  memset(hb.data(), 0, sizeof(mpm2d::Bnd2) * n);
  if (n) {
    HIPCHK2D(m, hipMemcpy(hp.data(), m->pid, n * 4, hipMemcpyDeviceToHost));
    if (m->rigid_enabled) {
      HIPCHK2D(m, hipMemcpy(hs.data(), m->d_states, n * 4, hipMemcpyDeviceToHost));
      HIPCHK2D(m, hipMemcpy(hb.data(), m->d_bnd, sizeof(mpm2d::Bnd2) * n, hipMemcpyDeviceToHost));
    }
  }
  int64_t k = 0;
  for (size_t i = 0; i < n; i++) {
    if (hp[i] < 0) continue;
    if (k >= capacity) return fail2d(m, MPMHIP_ECAPACITY, "download buffer too small");
    if (states) states[k] = hs[i];
    if (distance) distance[k] = hb[i].dist;
    if (normal) { normal[2 * k] = hb[i].n[0]; normal[2 * k + 1] = hb[i].n[1]; }
    if (near) near[k] = (int32_t)hb[i].near;
    k++;
  }
  return k;
}

int mpmhip2d_step(mpmhip2d_ctx *m, float dt) {  // MPM<dim>::step, src/mpm.cpp:428-439 (virtual: AsyncMPM<2>::step when the stepper is resident)
  if (!m) return MPMHIP_EINVAL;
  if (m->async.resident) return mpmhip2d_async_step(m, dt);
  if (dt < 0) {
    const int rc = mpmhip2d_substep(m);
    m->request_t = m->t;
    return rc;
  }
  m->request_t += dt;
  while (m->t + m->P.dt < m->request_t)
    if (int rc = mpmhip2d_substep(m)) return rc;
  return MPMHIP_OK;
}

double mpmhip2d_current_time(const mpmhip2d_ctx *m) { return m ? (double)m->t : 0.0; }

int64_t mpmhip2d_num_particles(mpmhip2d_ctx *m) {
  if (!m) return MPMHIP_EINVAL;
  if (hipSetDevice(m->device) != hipSuccess || hipStreamSynchronize(m->stream) != hipSuccess) return MPMHIP_EHIP;
  unsigned int dead = 0;
  if (hipMemcpy(&dead, m->n_dead, sizeof dead, hipMemcpyDeviceToHost) != hipSuccess) return MPMHIP_EHIP;
  return m->n - (int64_t)dead;
}

// live particles in slot order; any output may be NULL.  Returns the number written or a negative error.
int64_t mpmhip2d_download(mpmhip2d_ctx *m, int64_t capacity, float *x, float *v, float *F, float *B, float *aux, int32_t *gid,
                          int32_t *id) {
  if (!m) return MPMHIP_EINVAL;
  if (hipSetDevice(m->device) != hipSuccess || hipStreamSynchronize(m->stream) != hipSuccess) return fail2d(m, MPMHIP_EHIP, "synchronise failed");
  const size_t n = (size_t)m->n;
  std::vector<float> hx(2 * n), hv(2 * n), hF(4 * n), hB(4 * n), ha(n);
  std::vector<int32_t> hg(n), hp(n);
  if (n) {
    HIPCHK2D(m, hipMemcpy(hx.data(), m->x, 8 * n, hipMemcpyDeviceToHost)); HIPCHK2D(m, hipMemcpy(hv.data(), m->v, 8 * n, hipMemcpyDeviceToHost));
    HIPCHK2D(m, hipMemcpy(hF.data(), m->F, 16 * n, hipMemcpyDeviceToHost)); HIPCHK2D(m, hipMemcpy(hB.data(), m->B, 16 * n, hipMemcpyDeviceToHost));
    HIPCHK2D(m, hipMemcpy(ha.data(), m->aux, 4 * n, hipMemcpyDeviceToHost)); HIPCHK2D(m, hipMemcpy(hg.data(), m->gid, 4 * n, hipMemcpyDeviceToHost));
    HIPCHK2D(m, hipMemcpy(hp.data(), m->pid, 4 * n, hipMemcpyDeviceToHost));
  }
  int64_t k = 0;
  for (size_t i = 0; i < n; i++) {
    if (hp[i] < 0) continue;
    if (k >= capacity) return fail2d(m, MPMHIP_ECAPACITY, "download buffer too small");
    if (x) { x[2 * k] = hx[2 * i]; x[2 * k + 1] = hx[2 * i + 1]; }
    if (v) { v[2 * k] = hv[2 * i]; v[2 * k + 1] = hv[2 * i + 1]; }
    if (F) for (int q = 0; q < 4; q++) F[4 * k + q] = hF[4 * i + q];
    if (B) for (int q = 0; q < 4; q++) B[4 * k + q] = hB[4 * i + q];
    if (aux) aux[k] = ha[i];
    if (gid) gid[k] = hg[i];
    if (id) id[k] = hp[i];
    k++;
  }
  return k;
}

int mpmhip2d_download_grid(mpmhip2d_ctx *m, float *grid) {  // (v.x, v.y, m) per node of the (res+1)^2 grid after the last substep
  if (!m || !grid) return MPMHIP_EINVAL;
  HIPCHK2D(m, hipSetDevice(m->device));
  HIPCHK2D(m, hipStreamSynchronize(m->stream));
  HIPCHK2D(m, hipMemcpy(grid, m->grid, sizeof(float) * 3 * (size_t)(m->P.res[0] + 1) * (m->P.res[1] + 1), hipMemcpyDeviceToHost));
  return MPMHIP_OK;
}

#include "async2d_api.h"
#include "frame2d_api.h"

// ------------------------------------------------------------------------------------------------ debug math
int mpmhip_debug_cond_census(mpmhip_ctx *c, double out[MPMHIP_COND_CENSUS_WORDS]) {
  if (!c || !out) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->in_substep) return fail(c, MPMHIP_EINVAL, "cond_census inside a substep");
  constexpr int W = 8 + COND_BINS;
  static_assert(W == MPMHIP_COND_CENSUS_WORDS, "census layout");
  unsigned long long *d = nullptr;
  HIPCHK(c, dmalloc(&d, (size_t)W));
  hipError_t e = hipMemsetAsync(d, 0, sizeof(unsigned long long) * W, c->stream);
  std::vector<unsigned long long> h((size_t)W);
  if (e == hipSuccess && c->n_slots > 0) {
    hipLaunchKernelGGL(k_cond_census, dim3(particle_grid(c->n_slots)), dim3(256), 0, c->stream, c->P, (const float4 *)c->rg, d);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d, sizeof(unsigned long long) * W, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  HIPCHK(c, e);
  for (int i = 0; i < W; i++) out[i] = (double)h[i];
  const uint32_t bits = (uint32_t)h[4];
  float mx;
  memcpy(&mx, &bits, 4);
  out[4] = mx;
  return MPMHIP_OK;
}

int mpmhip_debug_svd3(mpmhip_ctx *c, int64_t n, const float *F, float *U, float *S, float *V) {
  if (!c || n <= 0) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  float *dF, *dU, *dS, *dV;
  HIPCHK(c, dmalloc(&dF, 9 * n)); HIPCHK(c, dmalloc(&dU, 9 * n)); HIPCHK(c, dmalloc(&dS, 3 * n)); HIPCHK(c, dmalloc(&dV, 9 * n));
  HIPCHK(c, hipMemcpy(dF, F, sizeof(float) * 9 * n, hipMemcpyHostToDevice));
  int rc = run_debug(c, k_debug_svd, n, (const float *)dF, dU, dS, dV);
  if (!rc) {
    HIPCHK(c, hipMemcpy(U, dU, sizeof(float) * 9 * n, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(S, dS, sizeof(float) * 3 * n, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(V, dV, sizeof(float) * 9 * n, hipMemcpyDeviceToHost));
  }
  hipFree(dF); hipFree(dU); hipFree(dS); hipFree(dV);
  return rc;
}

static int make_group(mpmhip_ctx *c, int32_t material, const float *params, GroupParams &g) {
  if (material < MPMHIP_VISCO || material > MPMHIP_ELASTIC) return fail(c, MPMHIP_EINVAL, "unknown material id %d", material);
  memset(&g, 0, sizeof g);
  memcpy(g.p, params, sizeof g.p);
  g.type = material;
  return MPMHIP_OK;
}

int mpmhip_debug_force(mpmhip_ctx *c, int32_t material, const float params[MPMHIP_NPARAM], int64_t n, const float *F,
                       const float *aux, float *out) {
  if (!c || n <= 0) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  GroupParams g;
  int rc = make_group(c, material, params, g);
  if (rc) return rc;
  float *dF, *dA, *dO;
  HIPCHK(c, dmalloc(&dF, 9 * n)); HIPCHK(c, dmalloc(&dA, n)); HIPCHK(c, dmalloc(&dO, 9 * n));
  HIPCHK(c, hipMemcpy(dF, F, sizeof(float) * 9 * n, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(dA, aux, sizeof(float) * n, hipMemcpyHostToDevice));
  rc = run_debug(c, k_debug_force, g, n, (const float *)dF, (const float *)dA, dO);
  if (!rc) HIPCHK(c, hipMemcpy(out, dO, sizeof(float) * 9 * n, hipMemcpyDeviceToHost));
  hipFree(dF); hipFree(dA); hipFree(dO);
  return rc;
}

int mpmhip_debug_plasticity(mpmhip_ctx *c, int32_t material, const float params[MPMHIP_NPARAM], int64_t n,
                            const float *cdg, float *F, float *aux, float *next_force) {
  if (!c || n <= 0) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  GroupParams g;
  int rc = make_group(c, material, params, g);
  if (rc) return rc;
  float *dC, *dF, *dA, *dO = nullptr;
  HIPCHK(c, dmalloc(&dC, 9 * n)); HIPCHK(c, dmalloc(&dF, 9 * n)); HIPCHK(c, dmalloc(&dA, n));
  if (next_force) HIPCHK(c, dmalloc(&dO, 9 * n));
  HIPCHK(c, hipMemcpy(dC, cdg, sizeof(float) * 9 * n, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(dF, F, sizeof(float) * 9 * n, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(dA, aux, sizeof(float) * n, hipMemcpyHostToDevice));
  rc = run_debug(c, k_debug_plasticity, g, n, (const float *)dC, dF, dA, dO);
  if (!rc) {
    HIPCHK(c, hipMemcpy(F, dF, sizeof(float) * 9 * n, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(aux, dA, sizeof(float) * n, hipMemcpyDeviceToHost));
    if (next_force) HIPCHK(c, hipMemcpy(next_force, dO, sizeof(float) * 9 * n, hipMemcpyDeviceToHost));
  }
  hipFree(dC); hipFree(dF); hipFree(dA); hipFree(dO);
  return rc;
}

}  // extern "C"

#include "tiled_api.h"

#ifdef MPMHIP_TIMING_BUILD
// (variant library only; not part of include/mpmhip.h) per-block stamps of the NEXT k_p2g launches (enable), or their read-back
extern "C" int mpmhip_timing_p2g_blocks(mpmhip_ctx *c, int32_t enable, unsigned long long *out, int64_t capacity_blocks) {
  if (!c) return MPMHIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (out && c->p2g_tlog) {
    const int64_t n = std::min<int64_t>(capacity_blocks, c->P.max_blocks);
    HIPCHK(c, hipMemcpy(out, c->p2g_tlog, sizeof(unsigned long long) * 3 * (size_t)n, hipMemcpyDeviceToHost));
  }
  if (enable && !c->p2g_tlog) HIPCHK(c, dmalloc(&c->p2g_tlog, (size_t)c->P.max_blocks * 3));
  if (enable) HIPCHK(c, hipMemset(c->p2g_tlog, 0, sizeof(unsigned long long) * 3 * (size_t)c->P.max_blocks));
  if (!enable && c->p2g_tlog) { (void)hipFree(c->p2g_tlog); c->p2g_tlog = nullptr; }
  return MPMHIP_OK;
}
#endif
