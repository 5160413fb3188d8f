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
    }
    flag_block(blk_flag, bkey);
  }
}

// ---- single-pass chained scans.  Both tables below are prefix sums over data produced by the previous kernel.
// Instead of the classic three launches (partials, scan of partials, apply) a workgroup publishes the sum of its
// chunk as ONE 64-bit word {epoch, value} (agent-scope atomic: the 8 XCDs' L2s are not coherent for plain
// accesses; the word is self-contained, so relaxed ordering suffices), sums the words of the chunks before it
// (spinning until their epoch matches) and finishes its chunk.  Chunks are handed out by a ticket counter, so a
// chunk's predecessors have always started and never wait on it: no deadlock, no co-residency assumption.  The
// epoch changes with every sort and each kernel zeroes the OTHER kernel's ticket: nothing is cleared by memsets.
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
// sum of the published values of chunks [0, chunk): every thread of the 256-thread workgroup gets the result
__device__ __forceinline__ uint32_t sum_predecessors(const unsigned long long *slots, uint32_t chunk, uint32_t epoch,
                                                     uint32_t *lds) {
  uint32_t pre = 0;
  for (uint32_t j = threadIdx.x; j < chunk; j += 256) {
    unsigned long long w;
    while ((uint32_t)((w = __hip_atomic_load(slots + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != epoch)
      __builtin_amdgcn_s_sleep(1);
    pre += (uint32_t)w;
  }
  uint32_t total;
  wg_exclusive_scan_256(pre, lds, total);
  return total;
}
__device__ __forceinline__ uint32_t take_ticket(uint32_t *ticket, uint32_t *s_chunk) {
  __syncthreads();  // the previous chunk's readers of *s_chunk are done
  if (threadIdx.x == 0) *s_chunk = atomicAdd(ticket, 1u);
  __syncthreads();
  return *s_chunk;
}

// Active-block table, one launch: byte flags -> bitmap `bits` (bit b of word w = block with Morton key 32w+b;
// the flags are cleared behind), per-word prefix `wprefix` (active blocks with key < 32w) = dense slot of every
// active block, the list act_blk[slot] = key, and cnt->n_active.  Chunk = 256 bitmap words, one per thread.
__global__ __launch_bounds__(256) void k_block_table(Params P, uint8_t *__restrict__ blk_flag,
                                                     uint32_t *__restrict__ bits, uint32_t *__restrict__ wprefix,
                                                     uint32_t *__restrict__ act_blk, Counters *cnt,
                                                     unsigned long long *__restrict__ slots, uint32_t *ticket,
                                                     uint32_t epoch) {
  __shared__ uint32_t lds[8];
  __shared__ uint32_t s_chunk;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticket[1] = 0;  // k_cell_table's counter (it is not running now)
  const uint32_t nchunks = (P.nbw + 255) / 256;
  while (true) {
    const uint32_t chunk = take_ticket(ticket, &s_chunk);
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
    const uint32_t chunk_base = sum_predecessors(slots, chunk, epoch, lds);
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
      if (grand > P.max_blocks) cnt->error |= 1u;
      cnt->n_active = grand;
    }
  }
}

// rank of each particle inside its cell; overwrites key[i] with cidx = slot(block)*64 + cell.  Two paths, chosen
// per sort from the statistics of the previous one (cnt->rank_mode, set by k_cell_table):
//  mode 0  runs of equal keys in adjacent lanes (the slots are in the order of the last physical reorder, i.e. cell by
//          cell) are aggregated into ONE returning global atomic per run.  Cheapest while the runs are long: 39 us
//          at 8 M particles on the freshly seeded lattice (8 per cell), but 162 us in the impact phase of C3, when the
//          particle order has decayed and the cells are hit from many waves at once.
//  mode 1  a workgroup takes 1024 consecutive slots, counts them per cell in an LDS hash table (open addressing on
//          cidx; nothing is assumed about which cells a batch holds), reserves each cell's range with ONE global
//          atomic per distinct cell of the batch and hands the local ranks out from LDS: ~70 us whatever the order.
// Both count the runs they see (cnt->run_heads); more than one run per 3 slots selects mode 1 for the next sort.
constexpr int RANK_PER_THREAD = 4, RANK_BATCH = 256 * RANK_PER_THREAD, RANK_TAB = 2048;

// cidx of slot i (INVALID: dead / out of range / block table overflow)
__device__ __forceinline__ uint32_t rank_cidx(const Params &P, const uint32_t *__restrict__ key,
                                              const uint32_t *__restrict__ bits, const uint32_t *__restrict__ wprefix,
                                              uint32_t i, uint32_t n) {
  if (i >= n) return INVALID;
  const uint32_t k = key[i];
  if (k == INVALID) return INVALID;
  const uint32_t slot = block_slot(bits, wprefix, k >> 6);
  return (slot < P.max_blocks) ? slot * BC + (k & 63u) : INVALID;
}
// the run of equal values around this lane: [start, end) in lane numbers; heads = mask of the first lanes of all runs
__device__ __forceinline__ void lane_run(uint32_t c, uint32_t lane, int &start, int &end, unsigned long long &heads) {
  const uint32_t prev = __shfl_up(c, 1);
  heads = __ballot((lane == 0) || (c != prev));
  const unsigned long long le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
  start = 63 - __clzll(heads & le);
  const unsigned long long above = heads & ~le;
  end = above ? (__ffsll((long long)above) - 1) : 64;
}

__global__ __launch_bounds__(256) void k_rank(Params P, uint32_t *__restrict__ key, uint32_t *__restrict__ rank,
                                              uint32_t *__restrict__ cell_cnt, const uint32_t *__restrict__ bits,
                                              const uint32_t *__restrict__ wprefix, Counters *cnt) {
  __shared__ uint32_t tkey[RANK_TAB], tcnt[RANK_TAB];  // mode 1.  tcnt: particles of the batch in that cell, then their global base
  __shared__ uint32_t s_heads;
  const uint32_t n = P.n_slots;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t mode = cnt->rank_mode;  // uniform over the grid
  if (threadIdx.x == 0) s_heads = 0u;
  uint32_t my_heads = 0u;  // lane 0 of each wave counts its wave's runs
  if (mode == 0u) {
    // a workgroup takes batches of RANK_BATCH consecutive slots, RANK_PER_THREAD per thread: the keys, the block-table
    // lookups and the returning atomics of a thread's slots are issued back to back, so their round trips overlap
    // (C3, 8 M particles: sort 0.092 -> 0.087 ms on the lattice against one dependent chain per slot; A/B on one box)
    const uint32_t nbatch = (n + RANK_BATCH - 1) / RANK_BATCH;
    for (uint32_t b = blockIdx.x; b < nbatch; b += gridDim.x) {
      uint32_t cidx[RANK_PER_THREAD], base[RANK_PER_THREAD];
      int start[RANK_PER_THREAD];
#pragma unroll
      for (int u = 0; u < RANK_PER_THREAD; u++) cidx[u] = rank_cidx(P, key, bits, wprefix, b * RANK_BATCH + u * 256 + threadIdx.x, n);
#pragma unroll
      for (int u = 0; u < RANK_PER_THREAD; u++) {
        int end;
        unsigned long long H;
        lane_run(cidx[u], lane, start[u], end, H);
        my_heads += (uint32_t)__popcll(H);
        base[u] = 0;
        if ((int)lane == start[u] && cidx[u] != INVALID) base[u] = atomicAdd(&cell_cnt[cidx[u]], (uint32_t)(end - start[u]));
      }
#pragma unroll
      for (int u = 0; u < RANK_PER_THREAD; u++) {
        const uint32_t i = b * RANK_BATCH + u * 256 + threadIdx.x;
        const uint32_t r = __shfl(base[u], start[u]) + (lane - (uint32_t)start[u]);
        if (i < n) {
          key[i] = cidx[u];
          rank[i] = r;
        }
      }
    }
    __syncthreads();
  } else {
    const uint32_t nbatch = (n + RANK_BATCH - 1) / RANK_BATCH;
    for (uint32_t b = blockIdx.x; b < nbatch; b += gridDim.x) {
      for (int t = threadIdx.x; t < RANK_TAB; t += 256) { tkey[t] = INVALID; tcnt[t] = 0u; }
      __syncthreads();
      uint32_t cidx[RANK_PER_THREAD], ent[RANK_PER_THREAD], loc[RANK_PER_THREAD];
#pragma unroll
      for (int u = 0; u < RANK_PER_THREAD; u++) {
        const uint32_t i = b * RANK_BATCH + u * 256 + threadIdx.x;
        const uint32_t c = rank_cidx(P, key, bits, wprefix, i, n);
        cidx[u] = c;
        int start, end;
        unsigned long long H;
        lane_run(c, lane, start, end, H);  // a run enters the table once, with its length
        my_heads += (uint32_t)__popcll(H);
        uint32_t h = 0u, first = 0u;
        if ((int)lane == start && c != INVALID) {
          h = (c * 2654435761u) >> 21;  // 11 bits: RANK_TAB entries, at most half of them used
          for (;;) {
            const uint32_t seen = atomicCAS(&tkey[h], INVALID, c);
            if (seen == INVALID || seen == c) break;
            h = (h + 1u) & (RANK_TAB - 1);
          }
          first = atomicAdd(&tcnt[h], (uint32_t)(end - start));
        }
        ent[u] = __shfl(h, start);
        loc[u] = __shfl(first, start) + (lane - (uint32_t)start);
      }
      __syncthreads();
      for (int t = threadIdx.x; t < RANK_TAB; t += 256)
        if (tkey[t] != INVALID) tcnt[t] = atomicAdd(&cell_cnt[tkey[t]], tcnt[t]);
      __syncthreads();
#pragma unroll
      for (int u = 0; u < RANK_PER_THREAD; u++) {
        const uint32_t i = b * RANK_BATCH + u * 256 + threadIdx.x;
        if (i < n) {
          key[i] = cidx[u];
          rank[i] = (cidx[u] != INVALID) ? tcnt[ent[u]] + loc[u] : 0u;
        }
      }
      __syncthreads();
    }
  }
  // statistics from every 16th workgroup only (scaled): thousands of atomics on ONE address serialise at ~13 ns each
  if ((blockIdx.x & 15u) == 0u) {
    if (lane == 0 && my_heads) atomicAdd(&s_heads, my_heads);
    __syncthreads();
    if (threadIdx.x == 0 && s_heads) atomicAdd(&cnt->run_heads, s_heads * 16u);
  }
}

// Cell table, one launch: per-cell counts (k_rank) -> act_start[a] (first sorted position of active block a,
// sentinel at [n_active]) and cell_start[a*64 + c] (sentinel at [n_active*64]): the particles of cell i are
// perm[cell_start[i] .. cell_start[i+1]).  Zeroes the counters behind itself.  Chunk = the 64 blocks
// [CT_BLOCKS t, CT_BLOCKS (t + 1)): each of the 4 waves takes CT_BLOCKS / 4 of them, one lane per cell.
template <int CT_BLOCKS>  // blocks per chunk: 64 for large problems, 16 when there are few blocks (shorter chains,
                           // more workgroups: 35 -> 31 us of sort at 1 M particles, but 90 -> 100 us at 8 M)
__global__ __launch_bounds__(256) void k_cell_table(Params P, Counters *cnt, uint32_t *__restrict__ cell_cnt,
                                                    uint32_t *__restrict__ act_start,
                                                    uint32_t *__restrict__ cell_start,
                                                    unsigned long long *__restrict__ slots, uint32_t *ticket,
                                                    uint32_t epoch) {
  __shared__ uint32_t lds[8];
  __shared__ uint32_t s_chunk;
  constexpr int CT_BPW = CT_BLOCKS / 4;  // blocks per wave
  __shared__ uint32_t blk_tot[CT_BLOCKS];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ticket[0] = 0;  // k_block_table's counter (it is not running now)
    // k_rank of this sort is complete: its run statistics choose the path of the next one (see k_rank)
    cnt->rank_mode = (cnt->run_heads * 3u > P.n_slots) ? 1u : 0u;
    cnt->run_heads = 0u;
  }
  const uint32_t na = min(cnt->n_active, P.max_blocks);
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  while (true) {
    const uint32_t chunk = take_ticket(ticket + 1, &s_chunk);
    const uint32_t a0 = chunk * CT_BLOCKS;
    if (a0 >= na && !(na == 0 && chunk == 0)) return;
    uint32_t excl[CT_BPW];  // exclusive in-block prefix of this lane's cell, for the wave's blocks
#pragma unroll
    for (int i = 0; i < CT_BPW; i++) {
      const uint32_t a = a0 + wave * CT_BPW + i;
      uint32_t c = 0;
      if (a < na) {
        c = cell_cnt[(size_t)a * BC + lane];
        cell_cnt[(size_t)a * BC + lane] = 0;
      }
      uint32_t v = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(v, off);
        if ((int)lane >= off) v += u;
      }
      excl[i] = v - c;
      if (lane == 63) blk_tot[wave * CT_BPW + i] = v;
    }
    __syncthreads();
    // exclusive scan of the 64 block totals (threads 0..63 hold one block each; other threads contribute 0)
    const uint32_t mine = threadIdx.x < CT_BLOCKS ? blk_tot[threadIdx.x] : 0u;
    uint32_t total;
    const uint32_t boff = wg_exclusive_scan_256(mine, lds, total);
    if (threadIdx.x == 0) publish(slots + chunk, epoch, total);
    if (threadIdx.x < CT_BLOCKS) blk_tot[threadIdx.x] = boff;
    const uint32_t chunk_base = sum_predecessors(slots, chunk, epoch, lds);  // its barriers also cover blk_tot
#pragma unroll
    for (int i = 0; i < CT_BPW; i++) {
      const uint32_t a = a0 + wave * CT_BPW + i;
      if (a < na) {
        const uint32_t start = chunk_base + blk_tot[wave * CT_BPW + i];
        cell_start[(size_t)a * BC + lane] = start + excl[i];
        if (lane == 0) act_start[a] = start;
      }
    }
    if (a0 + CT_BLOCKS >= na && threadIdx.x == 0) {  // last chunk: sentinels + live count
      const uint32_t grand = chunk_base + total;
      act_start[na] = grand;
      cell_start[(size_t)na * BC] = grand;
      cnt->n_sorted = grand;
    }
    if (na == 0) return;
  }
}

// sorted position -> particle slot (the reference's sorted `particles` index vector, src/mpm.cpp:800-807)
__global__ __launch_bounds__(256) void k_perm(Params P, const uint32_t *__restrict__ key,
                                              const uint32_t *__restrict__ rank,
                                              const uint32_t *__restrict__ cell_start, uint32_t *__restrict__ perm) {
  const uint32_t n = P.n_slots;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t c = key[i];
    if (c != INVALID) perm[cell_start[c] + rank[i]] = i;
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
