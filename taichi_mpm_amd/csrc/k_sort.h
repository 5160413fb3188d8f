// taichi_mpm_amd/csrc/k_sort.h — the per-substep index sort: active-block table, in-cell ranks, cell table, sorted index, physical reorder
// Part of libmpmhip (see mpmhip.hip for the substep overview and the data layout).
#pragma once
#include "mpm_common.h"

namespace mpm {

// ------------------------------------------------------------------------------------------------ sort
// standalone key builder (first substep, after uploads, phase-level API); afterwards k_g2p produces the keys
__global__ __launch_bounds__(256) void k_build_keys(Params P, RecG *__restrict__ rg, const RecP *__restrict__ rp,
                                                    Counters *cnt, uint32_t *__restrict__ key,
                                                    uint8_t *__restrict__ blk_flag) {
  const uint32_t n = P.n_slots;
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t nloop = (n + stride - 1) / stride;
  for (uint32_t it = 0; it < nloop; it++) {  // uniform trip count: all lanes take part in the shuffle
    const uint32_t i = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t kk = INVALID, bkey = INVALID;
    if (i < n) {
      const float4 g0 = reinterpret_cast<const float4 *>(rg + i)[0];
      const int32_t pid = rg[i].pid;
      if (pid >= 0) {
        const float4 p0 = reinterpret_cast<const float4 *>(rp + i)[0];
        const float4 p1 = reinterpret_cast<const float4 *>(rp + i)[1];
        const float x[3] = {g0.x, g0.y, g0.z}, v[3] = {p0.w, p1.x, p1.y};
        kk = particle_key(P, x, v, bkey);
        if (kk == INVALID) {  // delete for good (clear_boundary_particles)
          rg[i].pid = -1;
          atomicAdd(&cnt->n_dead, 1u);
        }
      }
      key[i] = kk;
      if (P.pidc) P.pidc[i] = (uint32_t)pid;  // (read AFTER a deletion above would be -1 as well: a dead slot is in no cell)
    }
    flag_block(blk_flag, bkey);
  }
}

// ---- single-pass chained scans.  Both tables below are prefix sums over data produced by the previous kernel.
// Instead of the classic three launches (partials, scan of partials, apply) a workgroup publishes the sum of its
// chunk as ONE 64-bit word {epoch, value} (agent-scope atomic: the 8 XCDs' L2s are not coherent for plain
// accesses; the word is self-contained, so relaxed ordering suffices), sums the words of the chunks before it
// (spinning until their epoch matches) and finishes its chunk.  A chunk publishes BEFORE it waits, and the host
// launches no more workgroups than the device keeps resident at once (scan_grid in mpmhip.hip, three eighths of the
// occupancy limit), so every chunk a workgroup waits for has been published or is being computed: no deadlock.
// (Round 1 handed the chunks out through a ticket counter instead, which needs no co-residency: ~800 returning
// atomics on ONE address per launch, 6 us of the sort at 8 M particles and 5 us at 1 M.)  The epoch changes with
// every sort: nothing is cleared by memsets.
__device__ __forceinline__ uint32_t wg_exclusive_scan_256(uint32_t v, uint32_t *lds /*>=4*/, uint32_t &total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t u = __shfl_up(inc, off);
    if ((int)lane >= off) inc += u;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (uint32_t w = 0; w < wave; w++) base += lds[w];
  total = lds[0] + lds[1] + lds[2] + lds[3];
  __syncthreads();
  return base + inc - v;
}

__device__ __forceinline__ void publish(unsigned long long *slot, uint32_t epoch, uint32_t value) {
  __hip_atomic_store(slot, ((unsigned long long)epoch << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// A wait on another workgroup's word is BOUNDED: the host sizes every chained scan's launch from the occupancy API so that the chunks a
// workgroup waits for are resident (scan_limit / scan_residency_check in mpmhip.hip), but dispatch order and residency are not
// promised by HIP (MI355X guide: "bound every spin") — another runtime, a partitioned device or a debugger must cost a sticky error
// (Counters::error bit 8, reported by the next synchronising call), not the GPU.  A chunk that gives up counts the missing sums as
// zero: every index derived from them is then too SMALL, i.e. still inside its array.
constexpr unsigned long long SCAN_WAIT_TICKS = 400000000ull;  // of the 100 MHz wall clock: 4 s
constexpr uint32_t SCAN_ERROR_BIT = 8u;
__device__ __forceinline__ bool scan_wait_expired(unsigned long long &t0, uint32_t &polls, uint32_t *err) {
  if ((++polls & 1023u) != 0u) return false;  // (the clock is read once per 1 024 polls)
  const unsigned long long now = wall_clock64();
  if (t0 == 0ull) { t0 = now; return false; }
  if (now - t0 <= SCAN_WAIT_TICKS) return false;
  if (err) atomicOr(err, SCAN_ERROR_BIT);
  return true;
}
// sum of the published values of chunks [0, chunk): every thread of the 256-thread workgroup gets the result
__device__ __forceinline__ uint32_t sum_predecessors(const unsigned long long *slots, uint32_t chunk, uint32_t epoch,
                                                     uint32_t *lds, uint32_t *err) {
  uint32_t pre = 0;
  unsigned long long t0 = 0ull;
  uint32_t polls = 0u;
  for (uint32_t j = threadIdx.x; j < chunk; j += 256) {
    unsigned long long w;
    bool gave_up = false;
    while ((uint32_t)((w = __hip_atomic_load(slots + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != epoch) {
      __builtin_amdgcn_s_sleep(1);
      if (scan_wait_expired(t0, polls, err)) { gave_up = true; break; }
    }
    if (!gave_up) pre += (uint32_t)w;
  }
  uint32_t total;
  wg_exclusive_scan_256(pre, lds, total);
  return total;
}
// The cell table's scan carries TWO sums in one word: particles (31 bits: n_slots < 2^31) and the chunk's owner entries of the grid
// pass (<= 8 per block, <= 512 per chunk: 10 bits); the epoch keeps the remaining 23 bits (consecutive sorts always differ).
__device__ __forceinline__ unsigned long long wg_exclusive_scan_256_u64(unsigned long long v, unsigned long long *lds /*>=4*/,
                                                                        unsigned long long &total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long u = __shfl_up(inc, off);
    if ((int)lane >= off) inc += u;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  unsigned long long base = 0;
  for (uint32_t w = 0; w < wave; w++) base += lds[w];
  total = lds[0] + lds[1] + lds[2] + lds[3];
  __syncthreads();
  return base + inc - v;
}
__device__ __forceinline__ void publish2(unsigned long long *slot, uint32_t epoch, uint32_t particles, uint32_t owners) {
  __hip_atomic_store(slot, ((unsigned long long)(epoch & 0x7FFFFFu) << 41) | ((unsigned long long)owners << 31) | particles,
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// sums over the chunks [0, chunk): (owners << 32) | particles, for every thread of the 256-thread workgroup
__device__ __forceinline__ unsigned long long sum_predecessors2(const unsigned long long *slots, uint32_t chunk, uint32_t epoch,
                                                                unsigned long long *lds, uint32_t *err) {
  unsigned long long pre = 0;
  unsigned long long t0 = 0ull;
  uint32_t polls = 0u;
  for (uint32_t j = threadIdx.x; j < chunk; j += 256) {
    unsigned long long w;
    bool gave_up = false;
    while ((uint32_t)((w = __hip_atomic_load(slots + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 41) != (epoch & 0x7FFFFFu)) {
      __builtin_amdgcn_s_sleep(1);
      if (scan_wait_expired(t0, polls, err)) { gave_up = true; break; }  // (see sum_predecessors)
    }
    if (!gave_up) pre += (((w >> 31) & 0x3FFull) << 32) | (w & 0x7FFFFFFFull);
  }
  unsigned long long total;
  wg_exclusive_scan_256_u64(pre, lds, total);
  return total;
}
// chunks are dealt round-robin by workgroup id: chunk = blockIdx.x + round * gridDim.x (bid / nwg: the workgroup's number among
// those of its role and their count, when a launch carries two roles — k_sort_front)
__device__ __forceinline__ uint32_t next_chunk(uint32_t &round, const uint32_t bid = blockIdx.x, const uint32_t nwg = gridDim.x) {
  __syncthreads();  // the previous chunk's LDS readers are done
  return bid + (round++) * nwg;
}

// Active-block table, one launch: byte flags -> bitmap `bits` (bit b of word w = block with Morton key 32w+b;
// the flags are cleared behind), per-word prefix `wprefix` (active blocks with key < 32w) = dense slot of every
// active block, the list act_blk[slot] = key, and cnt->n_active.  Chunk = 256 bitmap words, one per thread.
__device__ __forceinline__ void block_table_body(const Params &P, uint8_t *__restrict__ blk_flag, uint32_t *__restrict__ bits,
                                                 uint32_t *__restrict__ wprefix, uint32_t *__restrict__ act_blk, Counters *cnt,
                                                 unsigned long long *__restrict__ slots, const uint32_t epoch, const uint32_t bid,
                                                 const uint32_t nwg, uint32_t *lds /*>= 8*/) {
  const uint32_t nchunks = (P.nbw + 255) / 256;
  uint32_t round = 0;
  while (true) {
    const uint32_t chunk = next_chunk(round, bid, nwg);
    if (chunk >= nchunks) return;
    const uint32_t w = chunk * 256 + threadIdx.x;
    uint32_t m = 0;
    if (w < P.nbw) {
      uint4 *src = reinterpret_cast<uint4 *>(blk_flag + (size_t)w * 32);
      const uint4 lo = src[0], hi = src[1];
      const uint32_t q[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const uint32_t v = q[i];  // four 0/1 bytes -> four bits
        m |= ((v & 1u) | ((v >> 7) & 2u) | ((v >> 14) & 4u) | ((v >> 21) & 8u)) << (4 * i);
      }
      if (m) { src[0] = make_uint4(0, 0, 0, 0); src[1] = make_uint4(0, 0, 0, 0); }
    }
    uint32_t total;
    const uint32_t excl = wg_exclusive_scan_256(__popc(m), lds, total);
    if (threadIdx.x == 0) publish(slots + chunk, epoch, total);
    const uint32_t chunk_base = sum_predecessors(slots, chunk, epoch, lds, &cnt->error);
    uint32_t run = chunk_base + excl;
    if (w < P.nbw) {
      bits[w] = m;
      wprefix[w] = run;
      uint32_t mm = m;
      while (mm) {
        const int b = __ffs(mm) - 1;
        mm &= mm - 1;
        if (run < P.max_blocks) act_blk[run] = (w << 5) | (uint32_t)b;
        run++;
      }
    }
    if (chunk == nchunks - 1 && threadIdx.x == 255) {
      const uint32_t grand = chunk_base + total;
      if (grand > P.max_blocks) atomicOr(&cnt->error, 1u);  // (k_cdf_rasterize may set its own bit from the side stream)
      cnt->n_active = grand;
    }
  }
}
__global__ __launch_bounds__(256) void k_block_table(Params P, uint8_t *__restrict__ blk_flag,
                                                     uint32_t *__restrict__ bits, uint32_t *__restrict__ wprefix,
                                                     uint32_t *__restrict__ act_blk, Counters *cnt,
                                                     unsigned long long *__restrict__ slots, uint32_t epoch) {
  __shared__ uint32_t lds[8];
  block_table_body(P, blk_flag, bits, wprefix, act_blk, cnt, slots, epoch, blockIdx.x, gridDim.x, lds);
}

// rank of each particle inside its cell.  Overwrites key[i] with ONE word per slot, (rank << cb) | cidx, where
// cidx = slot(block)*64 + cell and cb = the bits cidx needs for THIS sort's number of active blocks (packed_cell_bits):
// k_perm reads 4 bytes per slot instead of a cell index and a rank.  A rank that does not fit the remaining bits
// (at C3: 11 bits = 2 047 particles in one cell) is stored as all-ones in the word and in full in rank[i] — a side
// array nothing touches in an ordinary sort.
// Two paths, chosen per sort from the statistics of the previous one (cnt->rank_mode, set by k_cell_table):
//  mode 0  runs of equal cells in adjacent slots (the records lie in the order of the previous sort, i.e. cell by
//          cell) are aggregated into ONE returning global atomic per run.  Cheapest while the runs are long (the
//          freshly seeded lattice, 8 per cell), 4x slower in the impact phase of C3, when the order of the slots has
//          decayed and the cells are hit from many waves at once.
//  mode 1  a workgroup takes 1024 consecutive slots, counts them per cell in an LDS hash table (open addressing on
//          cidx; nothing is assumed about which cells a batch holds), reserves each cell's range with ONE global
//          atomic per distinct cell of the batch and hands the local ranks out from LDS: the same cost whatever the order.
// Both count the runs they see (cnt->run_heads); more than one run per 3 slots selects mode 1 for the next sort.
// FOUR CONSECUTIVE slots per thread: the keys come in and the packed words go out as 16-byte vectors, the block-table
// lookup is shared by a thread's slots when they lie in one block (they nearly always do), and a run of equal cells is
// followed through a thread's registers before it crosses to the next lane — a quarter of the memory instructions of
// one slot per lane (measured at C3, 8 M particles, one box: sort 0.081 -> 0.072 ms; wider batches per thread were
// slower).  A wave owns a tile of 256 consecutive slots; element e = 4 lane + j.
constexpr int RANK_BATCH = 1024, RANK_TAB_BITS = 11, RANK_TAB = 1 << RANK_TAB_BITS;  // 2 table entries per slot of a batch

// bits of the cell index in the packed word: n_active*64 < 2^cb strictly, so a valid word never equals INVALID
__device__ __forceinline__ uint32_t packed_cell_bits(const Params &P, uint32_t n_active) {
  if (P.test_small_rank) return 29u;  // TEST KNOB: a 3-bit rank field, so that small scenes exercise the side array
  const uint32_t cb = 32u - (uint32_t)__clz((int)(n_active * (uint32_t)BC));
  return cb < 6u ? 6u : cb;
}
// bits of the KEY in the packed word of the keyed sort (rank_body<true>, k_perm_keyed): Morton block (3 kbits) << 6 | cell
__device__ __forceinline__ uint32_t keyed_bits(const Params &P) {
  if (P.test_small_rank) return 29u;  // TEST KNOB: a 3-bit rank field, so that small scenes exercise the side array
  return 3u * (uint32_t)P.kbits + 6u;
}
struct RunInfo {
  bool head[4];   // element j starts a run of equal cidx inside the wave's tile
  int jh[4];      // local index of the head of element j's run, -1: the run came in from an earlier lane
  int len[4];     // run length (heads only)
  int s_in;       // tile position of the head of the incoming run (valid when some jh < 0)
  uint32_t nheads;
};
__device__ __forceinline__ RunInfo tile_runs(const uint32_t c[4], int lane) {
  RunInfo R;
  const uint32_t prevc = __shfl_up(c[3], 1);
  R.head[0] = (lane == 0) || (c[0] != prevc);
#pragma unroll
  for (int j = 1; j < 4; j++) R.head[j] = c[j] != c[j - 1];
  int hf = -1, hl = -1;
#pragma unroll
  for (int j = 3; j >= 0; j--) if (R.head[j]) hf = j;
#pragma unroll
  for (int j = 0; j < 4; j++) if (R.head[j]) hl = j;
  // last head at or before each lane (inclusive prefix max), first head after it (exclusive suffix min)
  int pmax = hl >= 0 ? 4 * lane + hl : -1;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int u = __shfl_up(pmax, off);
    if (lane >= off) pmax = max(pmax, u);
  }
  int s_in = __shfl_up(pmax, 1);
  if (lane == 0) s_in = 0;  // (unused: lane 0 starts with a head)
  int smin = hf >= 0 ? 4 * lane + hf : 256;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int u = __shfl_down(smin, off);
    if (lane + off < 64) smin = min(smin, u);
  }
  int n_out = __shfl_down(smin, 1);
  if (lane == 63) n_out = 256;
  R.s_in = s_in;
  int cur = -1;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (R.head[j]) cur = j;
    R.jh[j] = cur;
  }
  int nxt = n_out - 4 * lane;  // local index (possibly >= 4) of the next head after element j
#pragma unroll
  for (int j = 3; j >= 0; j--) {
    R.len[j] = nxt - j;
    if (R.head[j]) nxt = j;
  }
  R.nheads = (uint32_t)R.head[0] + (uint32_t)R.head[1] + (uint32_t)R.head[2] + (uint32_t)R.head[3];
  return R;
}
// value of each element's run head: v[j] for runs that start in this thread, the last head's value of the lane the
// incoming run started in otherwise
__device__ __forceinline__ void run_broadcast(const RunInfo &R, const uint32_t v[4], uint32_t out[4]) {
  uint32_t lastv = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) if (R.head[j]) lastv = v[j];
  const uint32_t vin = __shfl(lastv, R.s_in >> 2);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    uint32_t x = vin;
#pragma unroll
    for (int h = 0; h < 4; h++) if (R.jh[j] == h) x = v[h];
    out[j] = x;
  }
}
__device__ __forceinline__ uint32_t run_offset(const RunInfo &R, int lane, int j) {
  return (uint32_t)(R.jh[j] >= 0 ? j - R.jh[j] : 4 * lane + j - R.s_in);
}

// ---- preparation of the grid pass (k_grid.h), whose work items are the touched grid blocks: the tile of active block b overlaps
// the 8 grid blocks c = b + o, o in {0,1}^3; c is processed by its OWNER, the source block c - q with the smallest q.
//   nbr[32 a + n]  n < 27: dense slot of b + (n/9-1, n/3%3-1, n%3-1) or INVALID; [27] Morton key of b; [28] owner mask of a
//                  (k_rank: one active block per wave, beside its own work — bitmap and prefix are complete since k_block_table)
//   own_list[]     one entry 8 a + o per owned candidate, compacted with the second sum of k_cell_table's scan (cnt->n_own entries)
// so k_grid starts from a list without holes: one wave per touched grid block, one coalesced 128-byte row instead of the chain
// block list -> bitmap -> prefix, no wave that finds out it owns nothing (at 1 M particles 85 % of the (block, candidate) pairs).
__device__ __forceinline__ constexpr uint32_t nb27_bit(int dx, int dy, int dz) { return 1u << (((dx + 1) * 3 + (dy + 1)) * 3 + (dz + 1)); }
__device__ __forceinline__ constexpr uint32_t lower_sources(int o) {  // neighbours that own candidate o before this block does
  uint32_t m = 0;
  for (int q = 0; q < o; q++) m |= nb27_bit((o >> 2) - (q >> 2), ((o >> 1) & 1) - ((q >> 1) & 1), (o & 1) - (q & 1));
  return m;
}
struct FillStats { uint32_t n_live, n_active, n_own, epoch; };  // what the last sort saw, stored to a pinned host page (never waited for)

// the row of active block a; all 64 lanes of a wave call it
__device__ __forceinline__ void write_neighbour_row(const Params &P, const uint32_t a, const uint32_t key, const uint32_t lane,
                                                    const uint32_t *__restrict__ bits, const uint32_t *__restrict__ wprefix,
                                                    uint32_t *__restrict__ nbr) {
  uint32_t nslot = INVALID;
  if (lane < 27) {
    int bx, by, bz;
    demorton3(key, bx, by, bz);
    const int sx = bx + (int)lane / 9 - 1, sy = by + ((int)lane / 3) % 3 - 1, sz = bz + (int)lane % 3 - 1;
    if (sx >= 0 && sy >= 0 && sz >= 0) {
      const uint32_t bk = morton3(sx, sy, sz);
      if (bk < P.nbw * 32u && block_active(bits, bk)) {
        // (a slot beyond max_blocks has no tile: the block table overflowed, the sticky capacity error is set)
        const uint32_t ns = block_slot(bits, wprefix, bk);
        if (ns < P.max_blocks) nslot = ns;
      }
    }
  }
  const uint32_t amask = (uint32_t)__ballot(nslot != INVALID);
  uint32_t m = 0;
#pragma unroll
  for (int o = 0; o < 8; o++)
    if (!(amask & lower_sources(o))) m |= 1u << o;
  if (lane < 32) nbr[(size_t)a * 32 + lane] = lane < 27 ? nslot : (lane == 27 ? key : (lane == 28 ? m : 0u));
}

// KEYED (the two-launch front of the sort, k_sort_front): the cell counters are indexed by the KEY itself (Morton block << 6 | cell:
// cell_cnt = cellcnt_key[64 NB]), so the ranks need no block table — they are handed out WHILE the block table is built, by the other
// workgroups of the same launch; the packed word is (rank << kb) | key, kb = 3 kbits + 6, and k_perm_keyed looks the block's slot up.
template <bool KEYED>
__device__ __forceinline__ void rank_body(const Params &P, uint32_t *__restrict__ key, uint32_t *__restrict__ rank,
                                          uint32_t *__restrict__ cell_cnt, const uint32_t *__restrict__ bits,
                                          const uint32_t *__restrict__ wprefix, Counters *cnt,
                                          const uint32_t *__restrict__ act_blk, uint32_t *__restrict__ nbr, const uint32_t bid,
                                          const uint32_t nwg, uint32_t *tkey, uint32_t *tcnt, uint32_t &s_heads) {
  const uint32_t n = P.n_slots;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t mode = cnt->rank_mode;  // uniform over the grid
  if (threadIdx.x == 0) s_heads = 0u;
  uint32_t my_heads = 0u;
  const uint32_t na = KEYED ? 0u : min(cnt->n_active, P.max_blocks);  // (KEYED: the block table does not exist yet)
  const uint32_t cb = KEYED ? keyed_bits(P) : packed_cell_bits(P, na), rmax = (1u << (32u - cb)) - 1u;
  // the grid pass's neighbour rows (see above): one active block per wave; the block's key is requested here, the lookups follow
  // behind the batches (one more round trip at the end of a wave instead of two in front of its own loads)
  const uint32_t row_a = bid * 4u + (uint32_t)wave;
  uint32_t row_key = 0;
  if (!KEYED && nbr && row_a < na) row_key = act_blk[row_a];  // (nbr == nullptr: this ctx's grid pass does not use the list)
  const uint32_t nbatch = (n + RANK_BATCH - 1) / RANK_BATCH;
  // the keys of a workgroup's NEXT batch are requested before the atomics of this one are waited for (a launch with fewer workgroups
  // than batches: do_sort)
  auto load_keys = [&](uint32_t b, uint32_t (&k)[4]) {
    const uint32_t i0 = b * RANK_BATCH + (uint32_t)wave * 256u + 4u * (uint32_t)lane;
    k[0] = INVALID; k[1] = INVALID; k[2] = INVALID; k[3] = INVALID;
    if (i0 + 4u <= n) {
      const uint4 kk = *reinterpret_cast<const uint4 *>(key + i0);
      k[0] = kk.x; k[1] = kk.y; k[2] = kk.z; k[3] = kk.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) if (i0 + j < n) k[j] = key[i0 + j];
    }
  };
  uint32_t kn[4] = {INVALID, INVALID, INVALID, INVALID};
  if (bid < nbatch) load_keys(bid, kn);
  for (uint32_t b = bid; b < nbatch; b += nwg) {
    const uint32_t i0 = b * RANK_BATCH + (uint32_t)wave * 256u + 4u * (uint32_t)lane;
    uint32_t k[4] = {kn[0], kn[1], kn[2], kn[3]};
    if (b + nwg < nbatch) load_keys(b + nwg, kn);
    uint32_t c[4];
    uint32_t last_blk = INVALID, last_slot = INVALID;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      c[j] = INVALID;
      if (KEYED) {
        c[j] = k[j];
      } else if (k[j] != INVALID) {
        const uint32_t blk = k[j] >> 6;
        if (blk != last_blk) { last_slot = block_slot(bits, wprefix, blk); last_blk = blk; }
        if (last_slot < P.max_blocks) c[j] = last_slot * BC + (k[j] & 63u);
      }
    }
    const RunInfo R = tile_runs(c, lane);
    my_heads += R.nheads;
    uint32_t r[4];
    if (mode == 0u) {
      uint32_t base[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (R.head[j] && c[j] != INVALID) base[j] = atomicAdd(&cell_cnt[c[j]], (uint32_t)R.len[j]);
      uint32_t bb[4];
      run_broadcast(R, base, bb);
#pragma unroll
      for (int j = 0; j < 4; j++) r[j] = bb[j] + run_offset(R, lane, j);
    } else {
      for (int t = threadIdx.x; t < RANK_TAB; t += 256) { tkey[t] = INVALID; tcnt[t] = 0u; }
      __syncthreads();
      uint32_t hh[4] = {0u, 0u, 0u, 0u}, ff[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (R.head[j] && c[j] != INVALID) {
          uint32_t h = (c[j] * 2654435761u) >> (32 - RANK_TAB_BITS);
          for (;;) {
            const uint32_t seen = atomicCAS(&tkey[h], INVALID, c[j]);
            if (seen == INVALID || seen == c[j]) break;
            h = (h + 1u) & (RANK_TAB - 1);
          }
          hh[j] = h;
          ff[j] = atomicAdd(&tcnt[h], (uint32_t)R.len[j]);
        }
      }
      uint32_t ent[4], loc[4];
      run_broadcast(R, hh, ent);
      run_broadcast(R, ff, loc);
      __syncthreads();
      for (int t = threadIdx.x; t < RANK_TAB; t += 256)
        if (tkey[t] != INVALID) tcnt[t] = atomicAdd(&cell_cnt[tkey[t]], tcnt[t]);
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 4; j++) r[j] = (c[j] != INVALID) ? tcnt[ent[j]] + loc[j] + run_offset(R, lane, j) : 0u;
      __syncthreads();
    }
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      w[j] = INVALID;
      if (c[j] != INVALID) {
        if (r[j] >= rmax) { rank[i0 + j] = r[j]; r[j] = rmax; }  // (crowded cell: the full rank goes to the side array)
        w[j] = (r[j] << cb) | c[j];
      }
    }
    if (i0 + 4u <= n) {
      *reinterpret_cast<uint4 *>(key + i0) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (i0 + j < n) key[i0 + j] = w[j];
    }
  }
  if (!KEYED && nbr) {
    if (row_a < na) write_neighbour_row(P, row_a, row_key, (uint32_t)lane, bits, wprefix, nbr);
    for (uint32_t a = row_a + nwg * 4u; a < na; a += nwg * 4u)  // (more active blocks than waves: tiny particle counts)
      write_neighbour_row(P, a, act_blk[a], (uint32_t)lane, bits, wprefix, nbr);
  }
  __syncthreads();
  if ((bid & 15u) == 0u) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) my_heads += __shfl_xor(my_heads, off);
    if (lane == 0 && my_heads) atomicAdd(&s_heads, my_heads);
    __syncthreads();
    if (threadIdx.x == 0 && s_heads) atomicAdd(&cnt->run_heads, s_heads * 16u);
  }
}
__global__ __launch_bounds__(256) void k_rank(Params P, uint32_t *__restrict__ key, uint32_t *__restrict__ rank,
                                               uint32_t *__restrict__ cell_cnt, const uint32_t *__restrict__ bits,
                                               const uint32_t *__restrict__ wprefix, Counters *cnt,
                                               const uint32_t *__restrict__ act_blk, uint32_t *__restrict__ nbr) {
  __shared__ uint32_t tkey[RANK_TAB], tcnt[RANK_TAB];
  __shared__ uint32_t s_heads;
  rank_body<false>(P, key, rank, cell_cnt, bits, wprefix, cnt, act_blk, nbr, blockIdx.x, gridDim.x, tkey, tcnt, s_heads);
}

// The front of the sort as ONE launch (a ctx with key-indexed cell counters: do_sort): the first bt_wgs workgroups build the block
// table (they are dispatched first, so the chunks a workgroup of the chained scan waits for are resident), all others hand out the
// in-cell ranks — two jobs that only meet in k_cell_table.  One launch (~6.5 us of latency chain at 1 M particles, 7.3 at 8 M) less,
// and k_rank without its block-table lookups.
__global__ __launch_bounds__(256) void k_sort_front(Params P, uint8_t *__restrict__ blk_flag, uint32_t *__restrict__ bits,
                                                    uint32_t *__restrict__ wprefix, uint32_t *__restrict__ act_blk, Counters *cnt,
                                                    unsigned long long *__restrict__ slots, uint32_t epoch, uint32_t bt_wgs,
                                                    uint32_t *__restrict__ key, uint32_t *__restrict__ rank,
                                                    uint32_t *__restrict__ cellcnt_key) {
  __shared__ uint32_t tkey[RANK_TAB], tcnt[RANK_TAB];
  __shared__ uint32_t s_heads;
  if (blockIdx.x < bt_wgs) {
    block_table_body(P, blk_flag, bits, wprefix, act_blk, cnt, slots, epoch, blockIdx.x, bt_wgs, tkey);
    return;
  }
  rank_body<true>(P, key, rank, cellcnt_key, nullptr, nullptr, cnt, nullptr, nullptr, blockIdx.x - bt_wgs, gridDim.x - bt_wgs, tkey, tcnt,
                  s_heads);
}

// Cell table, one launch: per-cell counts (k_rank) -> act_start[a] (first sorted position of active block a,
// sentinel at [n_active]) and cell_start[a*64 + c] (sentinel at [n_active*64]): the particles of cell i are
// perm[cell_start[i] .. cell_start[i+1]).  Zeroes the counters behind itself.  Chunk = the 64 blocks
// [CT_BLOCKS t, CT_BLOCKS (t + 1)): each of the 4 waves takes CT_BLOCKS / 4 of them, one lane per cell.
//
// The same scan compacts the owner list of the grid pass (see above k_rank) from the rows' owner masks: its second sum.
// KEYED: the counters are indexed by the blocks' KEYS (k_sort_front handed the ranks out before the block table existed), and the
// neighbour rows are written HERE (k_rank, which writes them otherwise, has no block table in that form): the 27 lookups of a wave's
// blocks are dealt to its 64 lanes as (block, neighbour) pairs — one round trip, not one per block.
template <int CT_BLOCKS, bool KEYED>  // blocks per chunk: 64 for large problems, 16 when there are few blocks (shorter chains,
                                       // more workgroups: 35 -> 31 us of sort at 1 M particles, but 90 -> 100 us at 8 M)
__global__ __launch_bounds__(256) void k_cell_table(Params P, Counters *cnt, uint32_t *__restrict__ cell_cnt,
                                                    uint32_t *__restrict__ act_start,
                                                    uint32_t *__restrict__ cell_start,
                                                    unsigned long long *__restrict__ slots, uint32_t epoch,
                                                    uint32_t rank_runs_mul, uint32_t *__restrict__ chunk_blk,
                                                    uint32_t *__restrict__ nbr, uint32_t *__restrict__ own_list,
                                                    FillStats *__restrict__ stats, const uint32_t *__restrict__ act_blk,
                                                    const uint32_t *__restrict__ bits, const uint32_t *__restrict__ wprefix) {
  __shared__ unsigned long long lds[8];
  __shared__ uint32_t amask_s[CT_BLOCKS];
  constexpr int CT_BPW = CT_BLOCKS / 4;  // blocks per wave
  __shared__ uint32_t blk_tot[CT_BLOCKS], blk_cnt[CT_BLOCKS], own_tot[CT_BLOCKS], own_mask[CT_BLOCKS];
  // exclusive in-block prefix of every cell of the chunk's blocks, parked in LDS between the two halves of a chunk (in registers
  // it costs the kernel its fifth workgroup per CU, and the host sizes the scans' grids from what stays resident)
  __shared__ uint32_t excl_s[CT_BLOCKS][64];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // k_rank of this sort is complete: its run statistics choose the path of the next one (see k_rank)
    cnt->rank_mode = (cnt->run_heads * rank_runs_mul > P.n_slots) ? 1u : 0u;
    cnt->run_heads = 0u;
  }
  const uint32_t na = min(cnt->n_active, P.max_blocks);
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t round = 0;
  while (true) {
    const uint32_t chunk = next_chunk(round);
    const uint32_t a0 = chunk * CT_BLOCKS;
    uint32_t mykey = 0;  // lane i < CT_BPW: the key of the wave's i-th block (KEYED; requested before the block count is looked at)
    if constexpr (KEYED) {
      if (lane < CT_BPW && a0 + wave * CT_BPW + lane < P.max_blocks) mykey = act_blk[a0 + wave * CT_BPW + lane];
    }
    if (a0 >= na && !(na == 0 && chunk == 0)) return;
    // the counters of the wave's blocks, requested in front of the neighbour rows' lookups (KEYED): the two chains run side by side
    uint32_t cc[CT_BPW];
#pragma unroll
    for (int i = 0; i < CT_BPW; i++) {
      const uint32_t a = a0 + wave * CT_BPW + i;
      cc[i] = 0u;
      if (a < na) {
        const size_t row = KEYED ? (size_t)__shfl(mykey, i) * BC : (size_t)a * BC;
        cc[i] = cell_cnt[row + lane];
        cell_cnt[row + lane] = 0;
      }
    }
    if constexpr (KEYED) {
      if (nbr) {  // neighbour rows + owner masks of the wave's blocks: (block, neighbour) pairs dealt to the lanes
        if (lane < CT_BPW) amask_s[wave * CT_BPW + lane] = 0u;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < (27 * CT_BPW + 63) / 64; q++) {
          const uint32_t pair = (uint32_t)q * 64u + lane, i = pair / 27u, nb = pair - 27u * i;
          const uint32_t a = a0 + wave * CT_BPW + i;
          const uint32_t bkey = __shfl(mykey, (int)(i < (uint32_t)CT_BPW ? i : 0u));
          if (i < (uint32_t)CT_BPW && a < na) {
            int bx, by, bz;
            demorton3(bkey, bx, by, bz);
            const int sx = bx + (int)nb / 9 - 1, sy = by + ((int)nb / 3) % 3 - 1, sz = bz + (int)nb % 3 - 1;
            uint32_t nslot = INVALID;
            if (sx >= 0 && sy >= 0 && sz >= 0) {
              const uint32_t bk = morton3(sx, sy, sz);
              if (bk < P.nbw * 32u && block_active(bits, bk)) {
                const uint32_t ns = block_slot(bits, wprefix, bk);
                if (ns < P.max_blocks) nslot = ns;  // (a slot beyond max_blocks has no tile: the sticky capacity error is set)
              }
            }
            nbr[(size_t)a * 32 + nb] = nslot;
            if (nslot != INVALID) atomicOr(&amask_s[wave * CT_BPW + i], 1u << nb);
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (lane < CT_BPW) {  // owner masks of the wave's blocks: one lane per block
      const uint32_t a = a0 + wave * CT_BPW + lane;
      uint32_t m = 0u;
      if (nbr && a < na) {
        if constexpr (KEYED) {
          const uint32_t amask = amask_s[wave * CT_BPW + lane];
#pragma unroll
          for (int o = 0; o < 8; o++)
            if (!(amask & lower_sources(o))) m |= 1u << o;
          nbr[(size_t)a * 32 + 27] = mykey;
          nbr[(size_t)a * 32 + 28] = m;
        } else {
          m = nbr[(size_t)a * 32 + 28];  // (k_rank wrote the row)
        }
      }
      own_mask[wave * CT_BPW + lane] = m;
      own_tot[wave * CT_BPW + lane] = (uint32_t)__popc(m);
    }
#pragma unroll
    for (int i = 0; i < CT_BPW; i++) {
      const uint32_t c = cc[i];
      uint32_t v = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(v, off);
        if ((int)lane >= off) v += u;
      }
      excl_s[wave * CT_BPW + i][lane] = v - c;
      if (lane == 63) { blk_tot[wave * CT_BPW + i] = v; blk_cnt[wave * CT_BPW + i] = v; }
    }
    __syncthreads();
    // exclusive scan of the block totals (threads 0..CT_BLOCKS-1 hold one block each; other threads contribute 0):
    // (owner entries << 32) | particles
    const unsigned long long mine =
        threadIdx.x < CT_BLOCKS ? (((unsigned long long)own_tot[threadIdx.x] << 32) | blk_tot[threadIdx.x]) : 0ull;
    unsigned long long total;
    const unsigned long long boff = wg_exclusive_scan_256_u64(mine, lds, total);
    if (threadIdx.x == 0) publish2(slots + chunk, epoch, (uint32_t)total, (uint32_t)(total >> 32));
    if (threadIdx.x < CT_BLOCKS) { blk_tot[threadIdx.x] = (uint32_t)boff; own_tot[threadIdx.x] = (uint32_t)(boff >> 32); }
    const unsigned long long base2 = sum_predecessors2(slots, chunk, epoch, lds, &cnt->error);  // its barriers also cover blk_tot / own_tot
    const uint32_t chunk_base = (uint32_t)base2, own_base = (uint32_t)(base2 >> 32);
#pragma unroll
    for (int i = 0; i < CT_BPW; i++) {
      const uint32_t a = a0 + wave * CT_BPW + i;
      if (a < na) {
        const uint32_t start = chunk_base + blk_tot[wave * CT_BPW + i];
        cell_start[(size_t)a * BC + lane] = start + excl_s[wave * CT_BPW + i][lane];
        if (lane == 0) act_start[a] = start;
        const uint32_t om = own_mask[wave * CT_BPW + i];
        if (lane < 8 && ((om >> lane) & 1u))
          own_list[own_base + own_tot[wave * CT_BPW + i] + (uint32_t)__popc(om & ((1u << lane) - 1u))] = a * 8u + lane;
        // k_g2p_packed walks the sorted index in chunks of 256 positions: the block holding position 256 k, for every k inside this block
        if (chunk_blk && lane == 63)
          for (uint32_t k = (start + 255u) >> 8, end = start + blk_cnt[wave * CT_BPW + i]; (k << 8) < end; k++) chunk_blk[k] = a;
      }
    }
    if (a0 + CT_BLOCKS >= na && threadIdx.x == 0) {  // last chunk: sentinels + live count
      const uint32_t grand = chunk_base + (uint32_t)total, n_own = own_base + (uint32_t)(total >> 32);
      act_start[na] = grand;
      cell_start[(size_t)na * BC] = grand;
      cnt->n_sorted = grand;
      cnt->n_own = n_own;
      if (stats) { stats->n_live = grand; stats->n_active = na; stats->n_own = n_own; stats->epoch = epoch; }
    }
    if (na == 0) return;
  }
}

// The cell table WITHOUT the owner list (an untiled ctx of 2 M slots and more: mpmhip.hip: do_sort), as it was until round 4: the
// second sum of the scan, the masks and the LDS-parked prefixes cost 2 us at 8 M particles that this configuration does not get back.
template <int CT_BLOCKS, bool KEYED>
__global__ __launch_bounds__(256) void k_cell_table_plain(Params P, Counters *cnt, uint32_t *__restrict__ cell_cnt,
                                                    uint32_t *__restrict__ act_start,
                                                    uint32_t *__restrict__ cell_start,
                                                    unsigned long long *__restrict__ slots, uint32_t epoch,
                                                    uint32_t rank_runs_mul, uint32_t *__restrict__ chunk_blk,
                                                          FillStats *__restrict__ stats, const uint32_t *__restrict__ act_blk) {
  __shared__ uint32_t lds[8];
  constexpr int CT_BPW = CT_BLOCKS / 4;  // blocks per wave
  __shared__ uint32_t blk_tot[CT_BLOCKS];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // k_rank of this sort is complete: its run statistics choose the path of the next one (see k_rank)
    cnt->rank_mode = (cnt->run_heads * rank_runs_mul > P.n_slots) ? 1u : 0u;
    cnt->run_heads = 0u;
  }
  const uint32_t na = min(cnt->n_active, P.max_blocks);
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t round = 0;
  while (true) {
    const uint32_t chunk = next_chunk(round);
    const uint32_t a0 = chunk * CT_BLOCKS;
    uint32_t mykey = 0;     // KEYED (counters indexed by the blocks' keys, see k_cell_table): lane i < CT_BPW holds the i-th block's
    if constexpr (KEYED) {  // (requested before the block count is looked at: one round trip less in front of the counters; entries
                            // beyond the count are stale and never used)
      if (lane < CT_BPW && a0 + wave * CT_BPW + lane < P.max_blocks) mykey = act_blk[a0 + wave * CT_BPW + lane];
    }
    if (a0 >= na && !(na == 0 && chunk == 0)) return;
    uint32_t excl[CT_BPW];  // exclusive in-block prefix of this lane's cell, for the wave's blocks
    uint32_t tot[CT_BPW];   // (lane 63: the block's particle count)
#pragma unroll
    for (int i = 0; i < CT_BPW; i++) {
      const uint32_t a = a0 + wave * CT_BPW + i;
      uint32_t c = 0;
      if (a < na) {
        const size_t row = KEYED ? (size_t)__shfl(mykey, i) * BC : (size_t)a * BC;
        c = cell_cnt[row + lane];
        cell_cnt[row + lane] = 0;
      }
      uint32_t v = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(v, off);
        if ((int)lane >= off) v += u;
      }
      excl[i] = v - c;
      tot[i] = v;
      if (lane == 63) blk_tot[wave * CT_BPW + i] = v;
    }
    __syncthreads();
    // exclusive scan of the 64 block totals (threads 0..63 hold one block each; other threads contribute 0)
    const uint32_t mine = threadIdx.x < CT_BLOCKS ? blk_tot[threadIdx.x] : 0u;
    uint32_t total;
    const uint32_t boff = wg_exclusive_scan_256(mine, lds, total);
    if (threadIdx.x == 0) publish(slots + chunk, epoch, total);
    if (threadIdx.x < CT_BLOCKS) blk_tot[threadIdx.x] = boff;
    const uint32_t chunk_base = sum_predecessors(slots, chunk, epoch, lds, &cnt->error);  // its barriers also cover blk_tot
#pragma unroll
    for (int i = 0; i < CT_BPW; i++) {
      const uint32_t a = a0 + wave * CT_BPW + i;
      if (a < na) {
        const uint32_t start = chunk_base + blk_tot[wave * CT_BPW + i];
        cell_start[(size_t)a * BC + lane] = start + excl[i];
        if (lane == 0) act_start[a] = start;
        // k_g2p_packed walks the sorted index in chunks of 256 positions: the block holding position 256 k, for every k inside this block
        if (chunk_blk && lane == 63)
          for (uint32_t k = (start + 255u) >> 8; (k << 8) < start + tot[i]; k++) chunk_blk[k] = a;
      }
    }
    if (a0 + CT_BLOCKS >= na && threadIdx.x == 0) {  // last chunk: sentinels + live count
      const uint32_t grand = chunk_base + total;
      act_start[na] = grand;
      cell_start[(size_t)na * BC] = grand;
      cnt->n_sorted = grand;
      cnt->n_own = 0u;
      if (stats) { stats->n_live = grand; stats->n_active = na; stats->n_own = 0u; stats->epoch = epoch; }
    }
    if (na == 0) return;
  }
}

// sorted position -> particle slot (the reference's sorted `particles` index vector, src/mpm.cpp:800-807) from the
// packed words of k_rank.  (Four CONSECUTIVE slots per thread as in k_rank were measured slower here: the scattered 4-byte stores
// dominate, and one slot per lane and instruction keeps them coalesced.)
__global__ __launch_bounds__(256) void k_perm(Params P, const Counters *__restrict__ cnt,
                                              const uint32_t *__restrict__ key, const uint32_t *__restrict__ rank,
                                              const uint32_t *__restrict__ cell_start, uint32_t *__restrict__ perm) {
  const uint32_t n = P.n_slots;
  const uint32_t cb = packed_cell_bits(P, min(cnt->n_active, P.max_blocks)), rmax = (1u << (32u - cb)) - 1u;
  const uint32_t cmask = (1u << cb) - 1u;
  // (four slots of a thread one grid stride apart walk their chains side by side, as in k_perm_keyed below)
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4u * stride) {
    uint32_t w[4], r[4], cs[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t i = i0 + (uint32_t)j * stride;
      w[j] = i < n ? key[i] : INVALID;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      r[j] = 0u; cs[j] = 0u;
      if (w[j] == INVALID) continue;
      r[j] = w[j] >> cb;
      if (r[j] == rmax) r[j] = rank[i0 + (uint32_t)j * stride];
      cs[j] = cell_start[w[j] & cmask];
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (w[j] != INVALID) perm[cs[j] + r[j]] = i0 + (uint32_t)j * stride;
  }
}

// the same from the packed words of the keyed sort (k_sort_front): (rank << kb) | key, the block's slot looked up here (consecutive
// slots nearly always share the block: the two table words come from the cache)
__global__ __launch_bounds__(256) void k_perm_keyed(Params P, const Counters *__restrict__ cnt,
                                                    const uint32_t *__restrict__ key, const uint32_t *__restrict__ rank,
                                                    const uint32_t *__restrict__ cell_start, uint32_t *__restrict__ perm,
                                                    const uint32_t *__restrict__ bits, const uint32_t *__restrict__ wprefix) {
  const uint32_t n = P.n_slots;
  const uint32_t kb = keyed_bits(P), rmax = (1u << (32u - kb)) - 1u, kmask = (1u << kb) - 1u;
  // A slot is a chain of three dependent loads (packed word -> bitmap + prefix -> cell start) in front of one store, and the kernel has
  // nothing else to do: four slots of a thread, one grid stride apart (every instruction stays coalesced), walk the chain side by side.
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4u * stride) {
    uint32_t w[4], bw[4], pf[4], r[4], cs[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t i = i0 + (uint32_t)j * stride;
      w[j] = i < n ? key[i] : INVALID;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      bw[j] = 0u; pf[j] = 0u; r[j] = 0u;
      if (w[j] == INVALID) continue;
      const uint32_t bk = (w[j] & kmask) >> 6;
      bw[j] = bits[bk >> 5];
      pf[j] = wprefix[bk >> 5];
      r[j] = w[j] >> kb;
      if (r[j] == rmax) r[j] = rank[i0 + (uint32_t)j * stride];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      cs[j] = INVALID;
      if (w[j] == INVALID) continue;
      const uint32_t k = w[j] & kmask, bk = k >> 6;
      const uint32_t slot = pf[j] + __popc(bw[j] & ((1u << (bk & 31u)) - 1u));  // (= block_slot)
      if (slot < P.max_blocks) cs[j] = cell_start[(size_t)slot * BC + (k & 63u)];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (cs[j] == INVALID) continue;
      // (pos < n always holds on a healthy ctx.  After a block-table overflow — sticky error bit 1, the ctx is to be recreated — the
      // key-indexed counters of the blocks that found no slot are never zeroed, so a caller that keeps stepping would get inflated ranks
      // there: the bound keeps such a ctx inside its arrays)
      const uint32_t pos = cs[j] + r[j];
      if (pos < n) perm[pos] = i0 + (uint32_t)j * stride;
    }
  }
}

// ---- deterministic mode (mpmhip_config.deterministic).  The in-cell ranks above come from atomics (k_rank / k_sort_front), so the
// order of the particles INSIDE a cell of the sorted index — and with it the summation order of P2G's per-cell register sums and
// the last bits of everything downstream — differs from run to run.  Behind k_perm this launch puts every cell's entries in
// ascending CREATION ID (RecG.pid, unique): the one order that does not depend on where a particle happens to lie in memory
// (slots change with every substep, with a migration, with a snapshot).  The reference's sort key is unique for the same reason:
// (offset >> 5) << 25 | i, src/mpm.cpp:785-795.  One lane per cell: the ids of its (typically 8) particles are gathered from the
// records — up to 16 independent loads, one round trip —, every entry's place is the number of smaller ids, the ordered entries
// go to `out` (rank[], idle between k_perm and the next sort: do_sort swaps the two arrays).  Fuller cells count through memory,
// a cell of more than 64 particles with its whole wave.
// COMPACT = the ids come from Params::pidc (4 bytes per slot, written by whoever wrote the slot's key: the records' writers in the
// deterministic mode) — 32 MB gathered at 8 M particles instead of one 64-byte record line per particle (512 MB);
// otherwise from the records (a ctx whose keys were written before the mode was switched on): 147 us of gathered record lines at C3.
template <bool COMPACT>
__device__ __forceinline__ uint32_t pid_of(const Params &P, const float4 *__restrict__ rg, uint32_t slot) {
  if constexpr (COMPACT) return P.pidc[slot];
  else return __float_as_uint(rg[(size_t)slot * 4 + 3].z);  // RecG.pid (>= 0 for every entry of the sorted index)
}
// the 64 cells of one block, one lane per cell, straight from global memory (cells of any size; the whole wave calls it)
template <bool COMPACT>
__device__ __forceinline__ void cell_order_lanes(const Params &P, const uint32_t *__restrict__ perm, const float4 *__restrict__ rg,
                                                 uint32_t *__restrict__ out, const uint32_t s, const uint32_t n, const uint32_t lane) {
  if (n == 1u) {
    out[s] = perm[s];
  } else if (n > 1u && n <= 16u) {
    uint32_t pv[16], id[16];
#pragma unroll
    for (int i = 0; i < 16; i++) pv[i] = (uint32_t)i < n ? perm[s + i] : 0u;
#pragma unroll
    for (int i = 0; i < 16; i++) id[i] = (uint32_t)i < n ? pid_of<COMPACT>(P, rg, pv[i]) : INVALID;  // (the padding is smaller than nothing)
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if ((uint32_t)i < n) {
        uint32_t r = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) r += id[j] < id[i] ? 1u : 0u;
        out[s + r] = pv[i];
      }
    }
  } else if (n > 16u && n <= 64u) {
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t slot = perm[s + i], pi = pid_of<COMPACT>(P, rg, slot);
      uint32_t r = 0;
      for (uint32_t j = 0; j < n; j++) r += pid_of<COMPACT>(P, rg, perm[s + j]) < pi ? 1u : 0u;
      out[s + r] = slot;
    }
  }
  unsigned long long big = __ballot(n > 64u);
  while (big) {
    const int l = __ffsll((long long)big) - 1;
    big &= big - 1;
    const uint32_t bs = __shfl(s, l), bn = __shfl(n, l);
    for (uint32_t i = lane; i < bn; i += 64u) {
      const uint32_t slot = perm[bs + i], pi = pid_of<COMPACT>(P, rg, slot);
      uint32_t r = 0;
      for (uint32_t j = 0; j < bn; j++) r += pid_of<COMPACT>(P, rg, perm[bs + j]) < pi ? 1u : 0u;
      out[bs + r] = slot;
    }
  }
}
// One lane per cell over the whole cell table (the round-6 first form; kept for A/B: MPMHIP_CELL_ORDER=0).  Adjacent lanes read
// perm[] / write out[] at a stride of their cells' sizes: every instruction touches 64 different 32-byte sectors.
template <bool COMPACT>
__global__ __launch_bounds__(256) void k_cell_order(Params P, const Counters *__restrict__ cnt, const uint32_t *__restrict__ cell_start,
                                                    const uint32_t *__restrict__ perm, const float4 *__restrict__ rg,
                                                    uint32_t *__restrict__ out) {
  const uint32_t ncell = min(cnt->n_active, P.max_blocks) * (uint32_t)BC;
  const uint32_t stride = gridDim.x * blockDim.x, lane = threadIdx.x & 63;
  const uint32_t nloop = (ncell + stride - 1) / stride;
  for (uint32_t it = 0; it < nloop; it++) {  // uniform trip count: all lanes take part in the ballot
    const uint32_t c = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = 0, n = 0;
    if (c < ncell) { s = cell_start[c]; n = cell_start[c + 1] - s; }
    cell_order_lanes<COMPACT>(P, perm, rg, out, s, n, lane);
  }
}
// One WAVE per active block (the default): the block's entries of perm[] and their ids are staged in LDS by coalesced loads
// (entry e of the block by lane e mod 64), a lane orders its cell there, the ordered entries leave by coalesced stores — the global
// traffic is perm[] once, the ids once, out[] once, all in full lines.  The ordered entries overwrite the staged ones (a lane holds
// its cell's entries in registers by then); a cell of more than 16 particles stores straight to out[] and leaves INVALID behind (no
// slot number: the copy-out skips it).  A block of more than CO_CAP entries takes the per-lane walk.
// 22 KiB of LDS and 71 VGPRs per workgroup: 7 workgroups per CU — the launch is a chain of three dependent round trips per block
// (cell row, index, ids), so waves in flight are what it runs on.  Measured on one box (profiles/r06_w_co_ab.txt; the deterministic
// mode's extra time per C3 substep): this form 45 us; entries padded to word e + e / 8 so that cell-major accesses fall on different
// banks: 111 VGPRs 52 us, held to 96 / 80 by the launch bounds (spills) 50 / 57 us; the next block's row and index requested ahead
// of time: 92 VGPRs, 52 us; a separate output slab (48 KiB, 3 workgroups per CU): 53 us; one lane per cell (k_cell_order): 51 us.
constexpr uint32_t CO_CAP = 704;
template <bool COMPACT>
__global__ __launch_bounds__(256, 6) void k_cell_order_blocks(Params P, const Counters *__restrict__ cnt, const uint32_t *__restrict__ cell_start,
                                                           const uint32_t *__restrict__ perm, const float4 *__restrict__ rg,
                                                           uint32_t *__restrict__ out) {
  constexpr uint32_t K = CO_CAP / 64, SLAB = CO_CAP;
  __shared__ uint32_t s_id[4][SLAB], s_pv[4][SLAB];
  const uint32_t na = min(cnt->n_active, P.max_blocks);
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t *const ids = s_id[wave], *const pvs = s_pv[wave];
  for (uint32_t b = blockIdx.x * 4u + wave; b < na; b += gridDim.x * 4u) {  // (wave-uniform)
    const uint32_t cs = cell_start[(size_t)b * BC + lane], ce = cell_start[(size_t)b * BC + lane + 1];
    const uint32_t s0 = __shfl(cs, 0), nb = __shfl(ce, 63) - s0, n = ce - cs, o = cs - s0;
    if (nb == 0u) continue;
    if (nb > CO_CAP) {
      cell_order_lanes<COMPACT>(P, perm, rg, out, cs, n, lane);
      continue;
    }
    {  // stage: all index loads first, then all id gathers (two round trips for the block)
      uint32_t pv[K], id[K];
#pragma unroll
      for (uint32_t k = 0; k < K; k++) pv[k] = k * 64u + lane < nb ? perm[s0 + k * 64u + lane] : 0u;
#pragma unroll
      for (uint32_t k = 0; k < K; k++) id[k] = k * 64u + lane < nb ? pid_of<COMPACT>(P, rg, pv[k]) : INVALID;
#pragma unroll
      for (uint32_t k = 0; k < K; k++)
        if (k * 64u + lane < nb) { pvs[k * 64u + lane] = pv[k]; ids[k * 64u + lane] = id[k]; }
    }
    __builtin_amdgcn_wave_barrier();  // (DS operations of one wave execute in program order)
    if (n > 1u && n <= 16u) {
      uint32_t cp[16], ci[16];
#pragma unroll
      for (int i = 0; i < 16; i++) { cp[i] = (uint32_t)i < n ? pvs[o + i] : 0u; ci[i] = (uint32_t)i < n ? ids[o + i] : INVALID; }
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if ((uint32_t)i < n) {
          uint32_t r = 0;
#pragma unroll
          for (int j = 0; j < 16; j++) r += ci[j] < ci[i] ? 1u : 0u;
          pvs[o + r] = cp[i];
        }
      }
    } else if (n > 16u && n <= 64u) {
      for (uint32_t i = 0; i < n; i++) {
        const uint32_t pi = ids[o + i];
        uint32_t r = 0;
        for (uint32_t j = 0; j < n; j++) r += ids[o + j] < pi ? 1u : 0u;
        out[cs + r] = pvs[o + i];
      }
      for (uint32_t i = 0; i < n; i++) pvs[o + i] = INVALID;
    }
    unsigned long long big = __ballot(n > 64u);
    while (big) {  // a cell of more than 64 particles: its whole wave
      const int l = __ffsll((long long)big) - 1;
      big &= big - 1;
      const uint32_t bo = __shfl(o, l), bn = __shfl(n, l);
      for (uint32_t i = lane; i < bn; i += 64u) {
        const uint32_t pi = ids[bo + i];
        uint32_t r = 0;
        for (uint32_t j = 0; j < bn; j++) r += ids[bo + j] < pi ? 1u : 0u;
        out[s0 + bo + r] = pvs[bo + i];
      }
      __builtin_amdgcn_wave_barrier();
      for (uint32_t i = lane; i < bn; i += 64u) pvs[bo + i] = INVALID;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (uint32_t k = 0; k < K; k++)
      if (k * 64u + lane < nb) {
        const uint32_t v = pvs[k * 64u + lane];
        if (v != INVALID) out[s0 + k * 64u + lane] = v;
      }
    __builtin_amdgcn_wave_barrier();  // (the next block's staging overwrites the slabs)
  }
}

// physical reorder + compaction (sort_allocator, src/mpm.cpp:752-768): records gathered into sorted order
__global__ __launch_bounds__(256) void k_gather_records(const Counters *__restrict__ cnt,
                                                        const uint32_t *__restrict__ perm, const float4 *__restrict__ rg,
                                                        const float4 *__restrict__ rp, const float4 *__restrict__ rb,
                                                        float4 *__restrict__ rg2, float4 *__restrict__ rp2,
                                                        float4 *__restrict__ rb2) {
  const uint32_t n = cnt->n_sorted;
  // one float4 per thread: 4 threads per 64-byte record
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n * 4u; t += gridDim.x * blockDim.x) {
    const uint32_t j = t >> 2, q = t & 3;
    const uint32_t i = perm[j];
    rg2[(size_t)j * 4 + q] = rg[(size_t)i * 4 + q];
    rp2[(size_t)j * 4 + q] = rp[(size_t)i * 4 + q];
    if (q < 3) rb2[(size_t)j * 3 + q] = rb[(size_t)i * 3 + q];
  }
}
__global__ __launch_bounds__(256) void k_identity_perm(const Counters *__restrict__ cnt, uint32_t *__restrict__ perm) {
  const uint32_t n = cnt->n_sorted;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) perm[i] = i;
}


}  // namespace mpm
