// taichi_mpm_amd/csrc/k_async.h — AsyncMPM on the device: the particle pools of the asynchronous stepper and the kernels that
// move particles between them and the working set of an advance (AsyncMPM<dim>::advance, src/async/async_mpm.cpp:255-373;
// gather_from_pool, src/async/async_mpm.h:198-234).  Part of libmpmhip; host side: async_api.h.
//
// The reference keeps, per scheduler block (SPGrid block of 4 x 4 x 8 nodes), a `particle_pool` (containers at the block's
// own time) and a `backup_pool` (copies at an earlier time).  Here both live in ONE device-resident STORE of containers
// (two 64-byte records per container + a tag word + the particle id); a pool is the set of containers carrying its tag:
//     tag = block | AS_BACKUP bit,  AS_FREE for a dropped container.
// An advance never moves a container inside the store: it re-tags (pool -> backup, clear) in place, copies the winners of
// the gather into the ctx's record arrays (the working set of ONE ordinary substep), and appends the results behind the
// store's end.  Freed containers are squeezed out by an order-preserving compaction when they outnumber the live ones.
// No particle data crosses the PCIe bus during stepping: the host sees block tables and four counters.
#pragma once
#include "mpm_common.h"
#include "k_sort.h"
#include "async_sched.h"  // the action bits AT_* of the per-block table

namespace mpm {

constexpr uint32_t AS_BACKUP = 0x80000000u, AS_FREE = INVALID;
struct AsyncCounters {
  uint32_t n_work, n_append, n_freed, n_live;  // transient: reset by the host behind every read-back
  uint32_t size, pad[3];                       // persistent: containers in use incl. freed ones (the append cursor); pad[0]: sticky
                                               // error word (bit 8: a chained scan of the compaction waited in vain, k_sort.h)
};

// One 64-byte record pair per container: g = the ctx's RecG image (x3, aux, F9, gid, id, -), w = {v3, -}, {apic_b[0..3]},
// {apic_b[4..7]}, {apic_b[8], -, -, -}.

// gather_from_pool's "the first copy of an id wins" (particles_cnt[id] != global_cnt): copies are ordered by gather phase
// (smaller-step pools, this step's pools, larger-step backups), then by the reference's block order, then by position in the
// store (= insertion order).  Smallest key wins.
__device__ __forceinline__ unsigned long long async_key(uint32_t cat, uint32_t rank, uint32_t e) {
  return ((unsigned long long)cat << 60) | ((unsigned long long)rank << 32) | e;
}
__device__ __forceinline__ int async_category(uint32_t tag, uint8_t act) {
  if (tag == AS_FREE) return -1;
  if (!(tag & AS_BACKUP)) return (act & AT_POOL0) ? 0 : ((act & AT_POOL1) ? 1 : -1);
  return (act & AT_BACKUP) ? 2 : -1;
}

__global__ __launch_bounds__(256) void k_async_mark(uint32_t size, const uint32_t *__restrict__ tag, const int32_t *__restrict__ id,
                                                    const uint8_t *__restrict__ tbl, const uint32_t *__restrict__ rank_of,
                                                    unsigned long long *__restrict__ best) {
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < size; e += gridDim.x * blockDim.x) {
    const uint32_t t = tag[e];
    if (t == AS_FREE) continue;
    const uint32_t b = t & ~AS_BACKUP;
    const int cat = async_category(t, tbl[b]);
    if (cat >= 0) atomicMin(&best[id[e]], async_key((uint32_t)cat, rank_of[b], e));
  }
}

// winners -> the ctx's records (slot = order of arrival), then the in-place re-tagging of backup_current_dt_limit
__global__ __launch_bounds__(256) void k_async_gather(uint32_t size, uint32_t *__restrict__ tag, const int32_t *__restrict__ id,
                                                      const uint8_t *__restrict__ tbl, const uint32_t *__restrict__ rank_of,
                                                      unsigned long long *__restrict__ best, const float4 *__restrict__ sg,
                                                      const float4 *__restrict__ sw, const GroupParams *__restrict__ groups,
                                                      float4 *__restrict__ rg, float4 *__restrict__ rp, float4 *__restrict__ rb,
                                                      AsyncCounters *cnt) {
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < size; e += gridDim.x * blockDim.x) {
    const uint32_t t = tag[e];
    if (t == AS_FREE) continue;
    const uint32_t b = t & ~AS_BACKUP;
    const uint8_t act = tbl[b];
    const int cat = async_category(t, act);
    if (cat >= 0 && best[id[e]] == async_key((uint32_t)cat, rank_of[b], e)) {
      best[id[e]] = ~0ull;  // (one winner per id: the table is clean again after the pass)
      const uint32_t s = atomicAdd(&cnt->n_work, 1u);
      const float4 g0 = sg[(size_t)e * 4], g1 = sg[(size_t)e * 4 + 1], g2 = sg[(size_t)e * 4 + 2], g3 = sg[(size_t)e * 4 + 3];
      const float4 w0 = sw[(size_t)e * 4], w1 = sw[(size_t)e * 4 + 1], w2 = sw[(size_t)e * 4 + 2], w3 = sw[(size_t)e * 4 + 3];
      rg[(size_t)s * 4] = g0; rg[(size_t)s * 4 + 1] = g1; rg[(size_t)s * 4 + 2] = g2; rg[(size_t)s * 4 + 3] = g3;
      const float mass = groups[__float_as_uint(g3.y)].p[0];
      rp[(size_t)s * 4] = make_float4(g0.x, g0.y, g0.z, w0.x);  // (the P2G matrix A is rebuilt by k_affine: it depends on this advance's dt)
      rp[(size_t)s * 4 + 1] = make_float4(w0.y, w0.z, 0.0f, 0.0f);
      rp[(size_t)s * 4 + 2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      rp[(size_t)s * 4 + 3] = make_float4(0.0f, 0.0f, 0.0f, mass);
      rb[(size_t)s * 3] = w1; rb[(size_t)s * 3 + 1] = w2; rb[(size_t)s * 3 + 2] = w3;
    }
    if (act & AT_SWAP) {
      if (t & AS_BACKUP) { tag[e] = AS_FREE; atomicAdd(&cnt->n_freed, 1u); }
      else tag[e] = t | AS_BACKUP;
    }
  }
}

// after the substep: backups that have served (":337-341 backup_pool[offset].clear()")
__global__ __launch_bounds__(256) void k_async_clear(uint32_t size, uint32_t *__restrict__ tag, const uint8_t *__restrict__ tbl,
                                                     AsyncCounters *cnt) {
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < size; e += gridDim.x * blockDim.x) {
    const uint32_t t = tag[e];
    if (t != AS_FREE && (t & AS_BACKUP) && (tbl[t & ~AS_BACKUP] & AT_CLEAR)) { tag[e] = AS_FREE; atomicAdd(&cnt->n_freed, 1u); }
  }
}

// scheduler block of a position: the reference's SPGrid block (4 x 4 x 8 nodes) holding the particle's base node
// (get_grid_base_pos, src/mpm.h:252-255; src/async/async_mpm.cpp:350-355); INVALID outside the block table
__device__ __forceinline__ uint32_t async_block_of(const Params &P, float x0, float x1, float x2, int nbx, int nby, int nbz) {
  const float X[3] = {x0 * P.idx, x1 * P.idx, x2 * P.idx};
  if (!(X[0] >= 0.5f && X[1] >= 0.5f && X[2] >= 0.5f)) return INVALID;
  const int bx = (int)(X[0] - 0.5f) >> 2, by = (int)(X[1] - 0.5f) >> 2, bz = (int)(X[2] - 0.5f) >> 3;
  if (bx >= nbx || by >= nby || bz >= nbz) return INVALID;
  return ((uint32_t)bx * nby + by) * nbz + bz;
}

// the working set after its substep -> containers behind the store's end (":345-375 update backup_pool and particle_pool").
// all_to_pool: AsyncMPM::add_particles (src/async/async_mpm.cpp:57-75) — every particle goes to its block's pool.
__global__ __launch_bounds__(256) void k_async_file(Params P, const float4 *__restrict__ rg, const float4 *__restrict__ rp,
                                                    const float4 *__restrict__ rb, const uint8_t *__restrict__ tbl, int all_to_pool,
                                                    int nbx, int nby, int nbz, uint32_t cap, float4 *__restrict__ sg,
                                                    float4 *__restrict__ sw, uint32_t *__restrict__ tag, int32_t *__restrict__ id,
                                                    AsyncCounters *cnt) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < P.n_slots; i += gridDim.x * blockDim.x) {
    const float4 g3 = rg[(size_t)i * 4 + 3];
    const int32_t pid = __float_as_int(g3.z);
    if (pid < 0) continue;  // deleted by the substep (clear_boundary_particles): not filed back, as in the reference
    const float4 g0 = rg[(size_t)i * 4];
    const uint32_t b = async_block_of(P, g0.x, g0.y, g0.z, nbx, nby, nbz);
    if (b == INVALID) continue;
    const uint8_t act = all_to_pool ? (uint8_t)AT_DEST_POOL : tbl[b];
    if (!(act & (AT_DEST_POOL | AT_DEST_BACKUP))) continue;
    const uint32_t e = atomicAdd(&cnt->size, 1u);  // (the append cursor lives on the device: the host knows an upper bound)
    if (e >= cap) { atomicSub(&cnt->size, 1u); continue; }  // (the host sized the store for the whole working set: cannot happen)
    atomicAdd(&cnt->n_append, 1u);
    const float4 p0 = rp[(size_t)i * 4], p1 = rp[(size_t)i * 4 + 1];
    sg[(size_t)e * 4] = g0; sg[(size_t)e * 4 + 1] = rg[(size_t)i * 4 + 1]; sg[(size_t)e * 4 + 2] = rg[(size_t)i * 4 + 2];
    sg[(size_t)e * 4 + 3] = g3;
    sw[(size_t)e * 4] = make_float4(p0.w, p1.x, p1.y, 0.0f);
    sw[(size_t)e * 4 + 1] = rb[(size_t)i * 3]; sw[(size_t)e * 4 + 2] = rb[(size_t)i * 3 + 1]; sw[(size_t)e * 4 + 3] = rb[(size_t)i * 3 + 2];
    tag[e] = (act & AT_DEST_POOL) ? b : (b | AS_BACKUP);
    id[e] = pid;
  }
}

// order-preserving compaction of the store (chained single-pass scan of k_sort.h: chunk = 1024 containers)
__global__ __launch_bounds__(256) void k_async_compact(uint32_t size, const uint32_t *__restrict__ tag, const int32_t *__restrict__ id,
                                                       const float4 *__restrict__ sg, const float4 *__restrict__ sw,
                                                       uint32_t *__restrict__ tag2, int32_t *__restrict__ id2,
                                                       float4 *__restrict__ sg2, float4 *__restrict__ sw2,
                                                       unsigned long long *__restrict__ slots, uint32_t epoch, AsyncCounters *cnt) {
  __shared__ uint32_t lds[8];
  const uint32_t nchunks = (size + 1023u) / 1024u;
  uint32_t round = 0;
  while (true) {
    const uint32_t chunk = next_chunk(round);
    if (chunk >= nchunks) return;
    const uint32_t e0 = chunk * 1024u + threadIdx.x * 4u;
    uint32_t t[4], live = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      t[j] = (e0 + j < size) ? tag[e0 + j] : AS_FREE;
      live += t[j] != AS_FREE;
    }
    uint32_t total;
    const uint32_t excl = wg_exclusive_scan_256(live, lds, total);
    if (threadIdx.x == 0) publish(slots + chunk, epoch, total);
    uint32_t o = sum_predecessors(slots, chunk, epoch, lds, &cnt->pad[0]) + excl;  // (a wait that expires: sticky bit in pad[0], read back by the host)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (t[j] == AS_FREE) continue;
      const uint32_t e = e0 + j;
      tag2[o] = t[j]; id2[o] = id[e];
#pragma unroll
      for (int q = 0; q < 4; q++) { sg2[(size_t)o * 4 + q] = sg[(size_t)e * 4 + q]; sw2[(size_t)o * 4 + q] = sw[(size_t)e * 4 + q]; }
      o++;
    }
    if (chunk == nchunks - 1 && threadIdx.x == 255) { cnt->n_live = o; cnt->size = o; }
  }
}

// (min allowed dt = 0.1, max |v|^2 = 1e-16, count = 0) per block: the start values of the reduction (":104-105")
__global__ __launch_bounds__(256) void k_async_table_reset(uint32_t nblk, uint32_t *__restrict__ tab) {
  for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += gridDim.x * blockDim.x) {
    tab[3 * (size_t)b] = __float_as_uint(0.1f); tab[3 * (size_t)b + 1] = __float_as_uint(1e-16f); tab[3 * (size_t)b + 2] = 0u;
  }
}

// update_dt_limits, device half (src/async/async_mpm.cpp:91-111) over the POOL containers of the store: per block the
// smallest get_allowed_dt(dx), the largest |v|^2 and the number of containers (same table as k_async_block_reduce)
__global__ __launch_bounds__(256) void k_async_store_reduce(Params P, uint32_t size, const uint32_t *__restrict__ tag,
                                                            const float4 *__restrict__ sg, const float4 *__restrict__ sw,
                                                            const GroupParams *__restrict__ groups, uint32_t *__restrict__ tab) {
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < size; e += gridDim.x * blockDim.x) {
    const uint32_t t = tag[e];
    if (t == AS_FREE || (t & AS_BACKUP)) continue;
    const float4 g0 = sg[(size_t)e * 4], g1 = sg[(size_t)e * 4 + 1], g2 = sg[(size_t)e * 4 + 2], g3 = sg[(size_t)e * 4 + 3];
    const float4 w0 = sw[(size_t)e * 4];
    mat3 F;
    F.m[0] = g1.x; F.m[1] = g1.y; F.m[2] = g1.z; F.m[3] = g1.w; F.m[4] = g2.x; F.m[5] = g2.y; F.m[6] = g2.z; F.m[7] = g2.w; F.m[8] = g3.x;
    const float v[3] = {w0.x, w0.y, w0.z};
    const float adt = allowed_dt(groups[__float_as_uint(g3.y)], F, g0.w, v, P.dx);
    atomicMin(&tab[3 * (size_t)t + 0], __float_as_uint(fmaxf(adt, 0.0f)));
    atomicMax(&tab[3 * (size_t)t + 1], __float_as_uint(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]));
    atomicAdd(&tab[3 * (size_t)t + 2], 1u);
  }
}

// every pool container -> the ctx's records (AsyncMPM::visualize's particle list: duplicates of an id included, each with the
// block of the pool it sits in, src/async/async_visualize.cpp:17-26,86-96)
__global__ __launch_bounds__(256) void k_async_load(uint32_t size, const uint32_t *__restrict__ tag, const float4 *__restrict__ sg,
                                                    const float4 *__restrict__ sw, const GroupParams *__restrict__ groups,
                                                    float4 *__restrict__ rg, float4 *__restrict__ rp, float4 *__restrict__ rb,
                                                    uint32_t *__restrict__ blk_of, AsyncCounters *cnt) {
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < size; e += gridDim.x * blockDim.x) {
    const uint32_t t = tag[e];
    if (t == AS_FREE || (t & AS_BACKUP)) continue;
    const uint32_t s = atomicAdd(&cnt->n_work, 1u);
    const float4 g0 = sg[(size_t)e * 4], g3 = sg[(size_t)e * 4 + 3], w0 = sw[(size_t)e * 4];
    rg[(size_t)s * 4] = g0; rg[(size_t)s * 4 + 1] = sg[(size_t)e * 4 + 1]; rg[(size_t)s * 4 + 2] = sg[(size_t)e * 4 + 2]; rg[(size_t)s * 4 + 3] = g3;
    rp[(size_t)s * 4] = make_float4(g0.x, g0.y, g0.z, w0.x);
    rp[(size_t)s * 4 + 1] = make_float4(w0.y, w0.z, 0.0f, 0.0f);
    rp[(size_t)s * 4 + 2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    rp[(size_t)s * 4 + 3] = make_float4(0.0f, 0.0f, 0.0f, groups[__float_as_uint(g3.y)].p[0]);
    rb[(size_t)s * 3] = sw[(size_t)e * 4 + 1]; rb[(size_t)s * 3 + 1] = sw[(size_t)e * 4 + 2]; rb[(size_t)s * 3 + 2] = sw[(size_t)e * 4 + 3];
    blk_of[s] = t;
  }
}

// pool containers -> flat arrays (tests, frame output): one row per container, in store order
__global__ __launch_bounds__(256) void k_async_export(uint32_t size, const uint32_t *__restrict__ tag, const float4 *__restrict__ sg,
                                                      const float4 *__restrict__ sw, float *__restrict__ out27, uint32_t *__restrict__ blk,
                                                      AsyncCounters *cnt) {
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < size; e += gridDim.x * blockDim.x) {
    const uint32_t t = tag[e];
    if (t == AS_FREE || (t & AS_BACKUP)) continue;
    const uint32_t o = atomicAdd(&cnt->n_work, 1u);
    const float *g = reinterpret_cast<const float *>(sg + (size_t)e * 4), *w = reinterpret_cast<const float *>(sw + (size_t)e * 4);
    float *r = out27 + (size_t)o * 27;
    r[0] = g[0]; r[1] = g[1]; r[2] = g[2];                      // x
    r[3] = w[0]; r[4] = w[1]; r[5] = w[2];                      // v
#pragma unroll
    for (int k = 0; k < 9; k++) { r[6 + k] = g[4 + k]; r[15 + k] = w[4 + k]; }  // F, apic_b
    r[24] = g[3]; r[25] = g[13]; r[26] = g[14];                 // aux, gid (bits), id (bits)
    blk[o] = t;
  }
}

}  // namespace mpm
