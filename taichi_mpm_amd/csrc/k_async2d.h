// taichi_mpm_amd/csrc/k_async2d.h — AsyncMPM<2> on the device (create_simulation2('async_mpm'): TC_IMPLEMENTATION(Simulation2D,
// AsyncMPM2D, "async_mpm"), src/async/async_mpm.cpp:423-427): the particle pools of the asynchronous stepper for the 2D
// simulation object of k_mpm2d.h.  Part of libmpmhip; host side: async2d_api.h; the block scheduler: async_sched.h.
//
// Same design as k_async.h (read its header first): ONE device-resident store of containers, a pool = the containers carrying
// its tag (block | AS_BACKUP bit, AS_FREE for a dropped one); an advance re-tags in place, copies the winners of the gather
// into the 2D object's particle arrays (the working set of one MPM<2>::substep) and appends the results behind the store's
// end.  A 2D container is ONE 64-byte record: {x2, v2}, {F4}, {apic_b4}, {aux, gid bits, id bits, -}.  A scheduler block is
// the reference's 2D SPGrid block: 8 x 16 nodes (SPGrid_Mask<5, 5, 2>: block_xbits 3, block_ybits 4).
#pragma once
#include "k_async.h"
#include "k_mpm2d.h"

namespace mpm2d {

using mpm::AS_BACKUP;
using mpm::AS_FREE;
using mpm::AsyncCounters;
using mpm::INVALID;

__device__ __forceinline__ int32_t a2_id(const float4 *rec, uint32_t e) { return __float_as_int(rec[(size_t)e * 4 + 3].z); }

__global__ __launch_bounds__(256) void k2a_mark(uint32_t size, const uint32_t *__restrict__ tag, const float4 *__restrict__ rec,
                                                const uint8_t *__restrict__ tbl, const uint32_t *__restrict__ rank_of,
                                                unsigned long long *__restrict__ best) {
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < size; e += gridDim.x * blockDim.x) {
    const uint32_t t = tag[e];
    if (t == AS_FREE) continue;
    const uint32_t b = t & ~AS_BACKUP;
    const int cat = mpm::async_category(t, tbl[b]);
    if (cat >= 0) atomicMin(&best[a2_id(rec, e)], mpm::async_key((uint32_t)cat, rank_of[b], e));
  }
}

// winners -> the 2D object's particle arrays (slot = order of arrival), then the in-place re-tagging of backup_current_dt_limit
__global__ __launch_bounds__(256) void k2a_gather(uint32_t size, uint32_t *__restrict__ tag, const float4 *__restrict__ rec,
                                                  const uint8_t *__restrict__ tbl, const uint32_t *__restrict__ rank_of,
                                                  unsigned long long *__restrict__ best, float *__restrict__ x, float *__restrict__ v,
                                                  float *__restrict__ F, float *__restrict__ B, float *__restrict__ aux,
                                                  int32_t *__restrict__ gid, int32_t *__restrict__ pid, AsyncCounters *cnt) {
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < size; e += gridDim.x * blockDim.x) {
    const uint32_t t = tag[e];
    if (t == AS_FREE) continue;
    const uint32_t b = t & ~AS_BACKUP;
    const uint8_t act = tbl[b];
    const int cat = mpm::async_category(t, act);
    if (cat >= 0) {
      const float4 r3 = rec[(size_t)e * 4 + 3];
      const int32_t id = __float_as_int(r3.z);
      if (best[id] == mpm::async_key((uint32_t)cat, rank_of[b], e)) {
        best[id] = ~0ull;  // (one winner per id: the table is clean again after the pass)
        const uint32_t s = atomicAdd(&cnt->n_work, 1u);
        const float4 r0 = rec[(size_t)e * 4], r1 = rec[(size_t)e * 4 + 1], r2 = rec[(size_t)e * 4 + 2];
        x[2 * s] = r0.x; x[2 * s + 1] = r0.y; v[2 * s] = r0.z; v[2 * s + 1] = r0.w;
        F[4 * s] = r1.x; F[4 * s + 1] = r1.y; F[4 * s + 2] = r1.z; F[4 * s + 3] = r1.w;
        B[4 * s] = r2.x; B[4 * s + 1] = r2.y; B[4 * s + 2] = r2.z; B[4 * s + 3] = r2.w;
        aux[s] = r3.x; gid[s] = __float_as_int(r3.y); pid[s] = id;
      }
    }
    if (act & mpm::AT_SWAP) {
      if (t & AS_BACKUP) { tag[e] = AS_FREE; atomicAdd(&cnt->n_freed, 1u); }
      else tag[e] = t | AS_BACKUP;
    }
  }
}

// scheduler block of a position: the 2D SPGrid block (8 x 16 nodes) holding the particle's base node (get_grid_base_pos,
// src/mpm.h:252-255; src/async/async_mpm.cpp:350-355); INVALID outside the block table
__device__ __forceinline__ uint32_t a2_block_of(float idx, float x0, float x1, int nbx, int nby) {
  const float X0 = x0 * idx, X1 = x1 * idx;
  if (!(X0 >= 0.5f && X1 >= 0.5f)) return INVALID;
  const int bx = (int)(X0 - 0.5f) >> 3, by = (int)(X1 - 0.5f) >> 4;
  if (bx >= nbx || by >= nby) return INVALID;
  return (uint32_t)bx * nby + by;
}

// the working set after its substep -> containers behind the store's end (":345-375 update backup_pool and particle_pool").
// all_to_pool: AsyncMPM::add_particles (src/async/async_mpm.cpp:57-75) — every particle goes to its block's pool.
__global__ __launch_bounds__(256) void k2a_file(float idx, uint32_t n, const float *__restrict__ x, const float *__restrict__ v,
                                                const float *__restrict__ F, const float *__restrict__ B, const float *__restrict__ aux,
                                                const int32_t *__restrict__ gid, const int32_t *__restrict__ pid,
                                                const uint8_t *__restrict__ tbl, int all_to_pool, int nbx, int nby, uint32_t cap,
                                                float4 *__restrict__ rec, uint32_t *__restrict__ tag, AsyncCounters *cnt) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int32_t id = pid[i];
    if (id < 0) continue;  // deleted by the substep (clear_boundary_particles): not filed back, as in the reference
    const float x0 = x[2 * i], x1 = x[2 * i + 1];
    const uint32_t b = a2_block_of(idx, x0, x1, nbx, nby);
    if (b == INVALID) continue;
    const uint8_t act = all_to_pool ? (uint8_t)mpm::AT_DEST_POOL : tbl[b];
    if (!(act & (mpm::AT_DEST_POOL | mpm::AT_DEST_BACKUP))) continue;
    const uint32_t e = atomicAdd(&cnt->size, 1u);  // (the append cursor lives on the device: the host knows an upper bound)
    if (e >= cap) { atomicSub(&cnt->size, 1u); continue; }  // (the host sized the store for the whole working set: cannot happen)
    atomicAdd(&cnt->n_append, 1u);
    rec[(size_t)e * 4] = make_float4(x0, x1, v[2 * i], v[2 * i + 1]);
    rec[(size_t)e * 4 + 1] = make_float4(F[4 * i], F[4 * i + 1], F[4 * i + 2], F[4 * i + 3]);
    rec[(size_t)e * 4 + 2] = make_float4(B[4 * i], B[4 * i + 1], B[4 * i + 2], B[4 * i + 3]);
    rec[(size_t)e * 4 + 3] = make_float4(aux[i], __int_as_float(gid[i]), __int_as_float(id), 0.0f);
    tag[e] = (act & mpm::AT_DEST_POOL) ? b : (b | AS_BACKUP);
  }
}

// order-preserving compaction of the store (chained single-pass scan of k_sort.h: chunk = 1024 containers)
__global__ __launch_bounds__(256) void k2a_compact(uint32_t size, const uint32_t *__restrict__ tag, const float4 *__restrict__ rec,
                                                   uint32_t *__restrict__ tag2, float4 *__restrict__ rec2,
                                                   unsigned long long *__restrict__ slots, uint32_t epoch, AsyncCounters *cnt) {
  __shared__ uint32_t lds[8];
  const uint32_t nchunks = (size + 1023u) / 1024u;
  uint32_t round = 0;
  while (true) {
    const uint32_t chunk = mpm::next_chunk(round);
    if (chunk >= nchunks) return;
    const uint32_t e0 = chunk * 1024u + threadIdx.x * 4u;
    uint32_t t[4], live = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      t[j] = (e0 + j < size) ? tag[e0 + j] : AS_FREE;
      live += t[j] != AS_FREE;
    }
    uint32_t total;
    const uint32_t excl = mpm::wg_exclusive_scan_256(live, lds, total);
    if (threadIdx.x == 0) mpm::publish(slots + chunk, epoch, total);
    uint32_t o = mpm::sum_predecessors(slots, chunk, epoch, lds, &cnt->pad[0]) + excl;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (t[j] == AS_FREE) continue;
      const uint32_t e = e0 + j;
      tag2[o] = t[j];
#pragma unroll
      for (int q = 0; q < 4; q++) rec2[(size_t)o * 4 + q] = rec[(size_t)e * 4 + q];
      o++;
    }
    if (chunk == nchunks - 1 && threadIdx.x == 255) { cnt->n_live = o; cnt->size = o; }
  }
}

// update_dt_limits, device half (src/async/async_mpm.cpp:91-111) over the POOL containers of the store: per block the
// smallest get_allowed_dt(dx), the largest |v|^2 and the number of containers.  get_allowed_dt has no dim-dependent term
// (src/particles.cpp:136-155 ...): the 2 x 2 F is embedded in a 3 x 3 one.
__global__ __launch_bounds__(256) void k2a_store_reduce(float dx, uint32_t size, const uint32_t *__restrict__ tag,
                                                        const float4 *__restrict__ rec, const GroupParams *__restrict__ groups,
                                                        uint32_t *__restrict__ tab) {
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < size; e += gridDim.x * blockDim.x) {
    const uint32_t t = tag[e];
    if (t == AS_FREE || (t & AS_BACKUP)) continue;
    const float4 r0 = rec[(size_t)e * 4], r1 = rec[(size_t)e * 4 + 1], r3 = rec[(size_t)e * 4 + 3];
    mpm::mat3 F;
    F.m[0] = r1.x; F.m[1] = r1.y; F.m[2] = 0.0f; F.m[3] = r1.z; F.m[4] = r1.w; F.m[5] = 0.0f; F.m[6] = 0.0f; F.m[7] = 0.0f; F.m[8] = 1.0f;
    const float v[3] = {r0.z, r0.w, 0.0f};
    const float adt = mpm::allowed_dt(groups[__float_as_uint(r3.y)], F, r3.x, v, dx);
    atomicMin(&tab[3 * (size_t)t + 0], __float_as_uint(fmaxf(adt, 0.0f)));
    atomicMax(&tab[3 * (size_t)t + 1], __float_as_uint(v[0] * v[0] + v[1] * v[1]));
    atomicAdd(&tab[3 * (size_t)t + 2], 1u);
  }
}

// every pool container -> the 2D object's particle arrays (AsyncMPM::visualize's particle list: duplicates of an id included,
// src/async/async_visualize.cpp:86-96) — and, for tests and host-side inspection, its block (blk may be null)
__global__ __launch_bounds__(256) void k2a_load(uint32_t size, const uint32_t *__restrict__ tag, const float4 *__restrict__ rec,
                                                float *__restrict__ x, float *__restrict__ v, float *__restrict__ F, float *__restrict__ B,
                                                float *__restrict__ aux, int32_t *__restrict__ gid, int32_t *__restrict__ pid,
                                                uint32_t *__restrict__ blk, AsyncCounters *cnt) {
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < size; e += gridDim.x * blockDim.x) {
    const uint32_t t = tag[e];
    if (t == AS_FREE || (t & AS_BACKUP)) continue;
    const uint32_t s = atomicAdd(&cnt->n_work, 1u);
    const float4 r0 = rec[(size_t)e * 4], r1 = rec[(size_t)e * 4 + 1], r2 = rec[(size_t)e * 4 + 2], r3 = rec[(size_t)e * 4 + 3];
    x[2 * s] = r0.x; x[2 * s + 1] = r0.y; v[2 * s] = r0.z; v[2 * s + 1] = r0.w;
    F[4 * s] = r1.x; F[4 * s + 1] = r1.y; F[4 * s + 2] = r1.z; F[4 * s + 3] = r1.w;
    B[4 * s] = r2.x; B[4 * s + 1] = r2.y; B[4 * s + 2] = r2.z; B[4 * s + 3] = r2.w;
    aux[s] = r3.x; gid[s] = __float_as_int(r3.y); pid[s] = __float_as_int(r3.z);
    if (blk) blk[s] = t;
  }
}

}  // namespace mpm2d
