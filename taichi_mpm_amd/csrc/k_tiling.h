// taichi_mpm_amd/csrc/k_tiling.h — multi-GPU tiling kernels: halo pack, migration scan / pack / import
// Part of libmpmhip (see mpmhip.hip for the substep overview and the data layout).
#pragma once
#include "mpm_common.h"

namespace mpm {

// ------------------------------------------------------------------------------------------------ tiling
// This rank's partial (m v, m) sums on every halo box -> the box's send buffer.  One WAVE per grid block (4^3 nodes) a box touches,
// one lane per node: lanes 0..7 look the <= 8 active source blocks c - q up whose 6^3 tiles overlap the block (once per wave — until
// round 6 every NODE did its own eight bitmap lookups: 9.9 us for the 200 k box nodes of a rank of eight bricks, as long as the grid
// pass itself), every lane sums its node's tile values in the order q = 0..7 (the same sum as k_grid).  Lanes whose node lies outside
// the box (blocks cut by its faces) store nothing.
// Peer-write wires: B.send is the box's place in the PEER's receive buffer, so the pack IS the exchange; k_epoch_signal,
// launched behind it, publishes the substep's epoch in every peer's flag word.
__global__ __launch_bounds__(256) void k_halo_pack(Params P, Tiling T, const DevBox *__restrict__ boxes,
                                                   const uint32_t *__restrict__ bits,
                                                   const uint32_t *__restrict__ wprefix,
                                                   const float4 *__restrict__ tiles) {
  const int l = threadIdx.x & 63, lx = l >> 4, ly = (l >> 2) & 3, lz = l & 3;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t w = wave; w < T.box_blocks; w += nwaves) {
    // the block's box: the offsets ascend, so its index is the number of boxes that start at or before w (independent
    // wave-uniform loads: one round trip instead of a chain of them)
    int b = 0;
    for (int i = 1; i < T.n_boxes; i++) b += w >= boxes[i].boff ? 1 : 0;
    const DevBox &B = boxes[b];
    const uint32_t r = w - B.boff;
    const int nb1 = box_blocks_axis(B.lo[1], B.dim[1]), nb2 = box_blocks_axis(B.lo[2], B.dim[2]);
    const int cz = (B.lo[2] >> 2) + (int)(r % (uint32_t)nb2), cy = (B.lo[1] >> 2) + (int)((r / (uint32_t)nb2) % (uint32_t)nb1),
              cx = (B.lo[0] >> 2) + (int)(r / (uint32_t)(nb2 * nb1));
    uint32_t s = INVALID;
    if (l < 8) {
      const int sx = cx - (l >> 2), sy = cy - ((l >> 1) & 1), sz = cz - (l & 1);
      if (sx >= 0 && sy >= 0 && sz >= 0) {
        const uint32_t bk = morton3(sx, sy, sz);
        if (bk < P.nbw * 32u && block_active(bits, bk)) {
          const uint32_t t = block_slot(bits, wprefix, bk);
          if (t < P.max_blocks) s = t;
        }
      }
    }
    const int x = cx * BS + lx - B.lo[0], y = cy * BS + ly - B.lo[1], z = cz * BS + lz - B.lo[2];
    const bool inbox = (unsigned)x < (unsigned)B.dim[0] && (unsigned)y < (unsigned)B.dim[1] && (unsigned)z < (unsigned)B.dim[2];
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const uint32_t sslot = __shfl(s, q);  // wave-uniform
      const int tx = lx + 4 * (q >> 2), ty = ly + 4 * ((q >> 1) & 1), tz = lz + 4 * (q & 1);
      if (sslot != INVALID && inbox && tx < TS && ty < TS && tz < TS) {
        const float4 v = tiles[(size_t)sslot * TN + (tx * TS + ty) * TS + tz];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    if (inbox) B.send[((size_t)x * B.dim[1] + y) * B.dim[2] + z] = acc;
  }
}

// ------------------------------------------------------------------------------------------------ peer-write wires
// lane i < n waits until word[idx[i]] has reached `epoch` (the peers publish monotonically increasing epochs).  A bounded
// wait: after `timeout_ticks` of the 100 MHz wall clock the sticky error bit 16 is set and the kernel returns — a peer
// that never arrives must cost an error message, not the GPU.
// lane b < n publishes `epoch` in boxes[b].flag (this rank's word in the peer's array of halo epochs).  Launched BEHIND
// k_halo_pack on the same stream: a kernel boundary orders that kernel's stores before these (a release fence inside
// k_halo_pack — one per workgroup — writes the L2 back every time: 72 us instead of 6 at 160 k box nodes, measured).
__global__ __launch_bounds__(64) void k_epoch_signal(const DevBox *__restrict__ boxes, int n, uint32_t epoch) {
  const int b = threadIdx.x;
  if (b == 0) __atomic_thread_fence(__ATOMIC_RELEASE);  // (system scope)
  if (b < n) __hip_atomic_store(boxes[b].flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the same for ALL ranks of a local job in one launch (mpmhip_tiled_advance_group: the ranks share a stream, so one kernel boundary
// behind the last rank's k_halo_pack orders every rank's stores): workgroup r publishes rank r's epoch in its peers' flag words.  A rank
// of a real job pays ONE launch for signal + wait (k_epoch_signal_wait); K virtual ranks paid 2 K until round 6, now K + 1.
struct SignalGroup {
  const DevBox *boxes[MPMHIP_MAX_HALO_BOXES];
  int n[MPMHIP_MAX_HALO_BOXES];
  uint32_t epoch[MPMHIP_MAX_HALO_BOXES];
};
__global__ __launch_bounds__(64) void k_epoch_signal_group(SignalGroup G) {
  const int r = blockIdx.x, b = threadIdx.x;
  if (b == 0) __atomic_thread_fence(__ATOMIC_RELEASE);  // (system scope)
  if (b < G.n[r]) __hip_atomic_store(G.boxes[r][b].flag, G.epoch[r], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// signal + wait in ONE launch (the IPC wire without the overlap split: nothing runs between the two, and a launch costs ~3.7 us of a
// ~130 us substep at 8 bricks): lanes b < n_sig publish, then lanes i < n_wait poll.  Every rank publishes before it polls, so two
// ranks never wait for each other's store.  NOT for ranks that share one process's hardware queues (MPMHIP_WIRE_LOCAL: the polling
// kernel of one virtual rank could sit in front of the publishing kernel of another).
__global__ __launch_bounds__(64) void k_epoch_signal_wait(const DevBox *__restrict__ boxes, int n_sig, const uint32_t *words,
                                                          const int *__restrict__ idx, int n_wait, uint32_t epoch,
                                                          unsigned long long timeout_ticks, Counters *cnt) {
  const int i = threadIdx.x;
  if (i == 0) __atomic_thread_fence(__ATOMIC_RELEASE);  // (system scope)
  if (i < n_sig) __hip_atomic_store(boxes[i].flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  if (i >= n_wait) return;
  const uint32_t *w = words + idx[i];
  const unsigned long long t0 = wall_clock64();
  while ((int32_t)(__hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
    __builtin_amdgcn_s_sleep(16);
    if (wall_clock64() - t0 > timeout_ticks) {
      atomicOr(&cnt->error, 16u);
      return;
    }
  }
}

__global__ __launch_bounds__(64) void k_epoch_wait(const uint32_t *words, const int *__restrict__ idx, int n, uint32_t epoch,
                                                   unsigned long long timeout_ticks, Counters *cnt) {
  const int i = threadIdx.x;
  if (i >= n) return;
  const uint32_t *w = words + idx[i];
  const unsigned long long t0 = wall_clock64();
  while ((int32_t)(__hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
    __builtin_amdgcn_s_sleep(16);
    if (wall_clock64() - t0 > timeout_ticks) {
      atomicOr(&cnt->error, 16u);
      return;
    }
  }
}

// gridDim.x destinations x gridDim.y chunks: `words[d]` 4-byte words from src[d] to dst[d]; the workgroup of a destination
// that finishes last stores epoch -> flag[d] (release, system scope).  done[d] counts the finished chunks (left at 0).
struct PutList {
  const uint32_t *src[MPMHIP_MAX_HALO_BOXES];
  uint32_t *dst[MPMHIP_MAX_HALO_BOXES];
  uint32_t *flag[MPMHIP_MAX_HALO_BOXES];
  uint32_t words[MPMHIP_MAX_HALO_BOXES];
};
__global__ __launch_bounds__(256) void k_put(PutList L, uint32_t epoch, uint32_t *done) {
  const int d = blockIdx.x;
  const uint32_t n = L.words[d];
  const uint32_t *__restrict__ s = L.src[d];
  uint32_t *__restrict__ o = L.dst[d];
  const uint32_t t = blockIdx.y * blockDim.x + threadIdx.x, stride = gridDim.y * blockDim.x;
  if (((uintptr_t)s & 15) == 0 && ((uintptr_t)o & 15) == 0) {
    const uint32_t n4 = n >> 2;
    for (uint32_t i = t; i < n4; i += stride) reinterpret_cast<uint4 *>(o)[i] = reinterpret_cast<const uint4 *>(s)[i];
    for (uint32_t i = (n4 << 2) + t; i < n; i += stride) o[i] = s[i];
  } else {
    for (uint32_t i = t; i < n; i += stride) o[i] = s[i];
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t prev = __hip_atomic_fetch_add(&done[d], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.y - 1) {
      __hip_atomic_store(&done[d], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (L.flag[d]) __hip_atomic_store(L.flag[d], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// bounding box (cells) of the active blocks of the last sort: out[0..2] = min, out[3..5] = max (exclusive)
__global__ __launch_bounds__(256) void k_active_bounds(Params P, const Counters *__restrict__ cnt,
                                                       const uint32_t *__restrict__ act_blk, int *__restrict__ out) {
  const uint32_t na = min(cnt->n_active, P.max_blocks);
  int lo[3] = {1 << 30, 1 << 30, 1 << 30}, hi[3] = {-1, -1, -1};
  for (uint32_t a = blockIdx.x * blockDim.x + threadIdx.x; a < na; a += gridDim.x * blockDim.x) {
    int b[3];
    demorton3(act_blk[a], b[0], b[1], b[2]);
#pragma unroll
    for (int k = 0; k < 3; k++) { lo[k] = min(lo[k], b[k] * BS); hi[k] = max(hi[k], b[k] * BS + BS); }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[k] = min(lo[k], __shfl_xor(lo[k], off));
      hi[k] = max(hi[k], __shfl_xor(hi[k], off));
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(&out[k], lo[k]); atomicMax(&out[3 + k], hi[k]); }
  }
}

__device__ __forceinline__ int part_index(const int *cuts, int n, int c) {
  int p = 0;
  while (p + 1 < n && c >= cuts[p + 1]) p++;
  return p;
}
// destination rank of a live particle at x (brick containing its base cell); -1 if it is not representable
__device__ __forceinline__ int dest_rank(const Params &P, const Tiling &T, float4 g0, bool &beyond_margin) {
  int b[3];
  const float X[3] = {g0.x * P.idx, g0.y * P.idx, g0.z * P.idx};
  beyond_margin = false;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (!isfinite(X[k]) || X[k] < 0.5f) return -1;
    b[k] = (int)(X[k] - 0.5f);
    if (b[k] < T.lo[k] - T.margin || b[k] >= T.hi[k] + T.margin) beyond_margin = true;
  }
  return (part_index(T.cuts[0], T.dims[0], b[0]) * T.dims[1] + part_index(T.cuts[1], T.dims[1], b[1])) * T.dims[2] +
         part_index(T.cuts[2], T.dims[2], b[2]);
}

__global__ void k_scan_init(uint32_t *__restrict__ counts, int world, uint32_t tail) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < world) counts[t] = 0;
  else if (t < world + 3) counts[t] = (uint32_t)(1 << 30);
  else if (t < world + 6) counts[t] = (uint32_t)-1;
  else if (t < world + 7) counts[t] = 0;  // max speed (bits of a non-negative float)
  else if (t < world + 8) counts[t] = tail;  // (native data plane: the capacity of this rank's migration inbox)
}
// counts[d] = live particles whose base cell belongs to rank d != this rank; bounds[0..2] / [3..5] = min / max+1 of
// the base cells of all live particles; bounds[6] = bits of the largest |v|_inf dt / dx (cells per substep) among them
// (one pass, one wave-reduced atomic set per wave)
__global__ __launch_bounds__(256) void k_leaver_count(Params P, Tiling T, const float4 *__restrict__ rg,
                                                      const float4 *__restrict__ rp, uint32_t *__restrict__ counts,
                                                      int *__restrict__ bounds, Counters *cnt) {
  int lo[3] = {1 << 30, 1 << 30, 1 << 30}, hi[3] = {-1, -1, -1};
  float speed = 0.0f;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < P.n_slots; i += gridDim.x * blockDim.x) {
    if (__float_as_int(rg[(size_t)i * 4 + 3].z) < 0) continue;
    const float4 g0 = rg[(size_t)i * 4];
    bool beyond;
    const int d = dest_rank(P, T, g0, beyond);
    if (d < 0) continue;
    if (beyond) atomicOr(&cnt->error, 2u);
    if (d != T.rank) atomicAdd(&counts[d], 1u);
    const int b[3] = {(int)(g0.x * P.idx - 0.5f), (int)(g0.y * P.idx - 0.5f), (int)(g0.z * P.idx - 0.5f)};
#pragma unroll
    for (int k = 0; k < 3; k++) { lo[k] = min(lo[k], b[k]); hi[k] = max(hi[k], b[k] + 1); }
    const float4 p0 = rp[(size_t)i * 4], p1 = rp[(size_t)i * 4 + 1];  // v = (p0.w, p1.x, p1.y)
    speed = fmaxf(speed, fmaxf(fabsf(p0.w), fmaxf(fabsf(p1.x), fabsf(p1.y))));
  }
  // wave reduce -> workgroup reduce -> 6 atomics per WORKGROUP (same-address atomics serialise at ~13 ns each)
  __shared__ int red[4][6];
  __shared__ float sred[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) speed = fmaxf(speed, __shfl_xor(speed, off));
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = speed;
#pragma unroll
  for (int k = 0; k < 3; k++) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[k] = min(lo[k], __shfl_xor(lo[k], off));
      hi[k] = max(hi[k], __shfl_xor(hi[k], off));
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][k] = lo[k]; red[threadIdx.x >> 6][3 + k] = hi[k]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int k = threadIdx.x;
    const int l = min(min(red[0][k], red[1][k]), min(red[2][k], red[3][k]));
    const int h = max(max(red[0][3 + k], red[1][3 + k]), max(red[2][3 + k], red[3][3 + k]));
    if (h >= 0) { atomicMin(&bounds[k], l); atomicMax(&bounds[3 + k], h); }
  }
  if (threadIdx.x == 3) {
    const float s = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3])) * P.dt * P.idx;
    if (s > 0.0f) atomicMax(reinterpret_cast<uint32_t *>(&bounds[6]), __float_as_uint(s));  // NaN speeds are deleted by G2P
  }
}

// cursor[d] starts at the first record index of destination d; leavers are removed from this rank
__global__ __launch_bounds__(256) void k_leaver_pack(Params P, Tiling T, float4 *__restrict__ rg,
                                                     const float4 *__restrict__ rp, const float4 *__restrict__ rb,
                                                     uint32_t *__restrict__ key, uint32_t *__restrict__ cursor,
                                                     float4 *__restrict__ out, Counters *cnt) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < P.n_slots; i += gridDim.x * blockDim.x) {
    const float4 g3 = rg[(size_t)i * 4 + 3];
    if (__float_as_int(g3.z) < 0) continue;
    bool beyond;
    const int d = dest_rank(P, T, rg[(size_t)i * 4], beyond);
    if (d < 0 || d == T.rank) continue;
    const size_t o = (size_t)atomicAdd(&cursor[d], 1u) * 11;
#pragma unroll
    for (int q = 0; q < 4; q++) out[o + q] = rg[(size_t)i * 4 + q];
#pragma unroll
    for (int q = 0; q < 4; q++) out[o + 4 + q] = rp[(size_t)i * 4 + q];
#pragma unroll
    for (int q = 0; q < 3; q++) out[o + 8 + q] = rb[(size_t)i * 3 + q];
    rg[(size_t)i * 4 + 3] = make_float4(g3.x, g3.y, __int_as_float(-1), 0.0f);
    key[i] = INVALID;
    atomicAdd(&cnt->n_dead, 1u);
  }
}

// arrivals appended at slots base .. base+n; their keys and block flags join the ones k_g2p produced
__global__ __launch_bounds__(256) void k_import(Params P, uint32_t n, uint32_t base, const float4 *__restrict__ in,
                                                float4 *__restrict__ rg, float4 *__restrict__ rp,
                                                float4 *__restrict__ rb, uint32_t *__restrict__ key,
                                                uint8_t *__restrict__ blk_flag, Counters *cnt) {
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t nloop = (n + stride - 1) / stride;
  for (uint32_t it = 0; it < nloop; it++) {  // uniform trip count (flag_block shuffles)
    const uint32_t j = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t bkey = INVALID;
    if (j < n) {
      const size_t i = (size_t)base + j, o = (size_t)j * 11;
      const float4 g0 = in[o], g3 = in[o + 3], p0 = in[o + 4], p1 = in[o + 5];
      const float x[3] = {g0.x, g0.y, g0.z}, v[3] = {p0.w, p1.x, p1.y};
      const uint32_t kk = particle_key(P, x, v, bkey);
      int32_t pid = __float_as_int(g3.z);
      if (kk == INVALID) {
        pid = -1;
        atomicAdd(&cnt->n_dead, 1u);
      }
      key[i] = kk;
      if (P.pidc) P.pidc[i] = (uint32_t)pid;
      rg[i * 4 + 0] = g0; rg[i * 4 + 1] = in[o + 1]; rg[i * 4 + 2] = in[o + 2];
      rg[i * 4 + 3] = make_float4(g3.x, g3.y, __int_as_float(pid), 0.0f);
      rp[i * 4 + 0] = p0; rp[i * 4 + 1] = p1; rp[i * 4 + 2] = in[o + 6]; rp[i * 4 + 3] = in[o + 7];
      rb[i * 3 + 0] = in[o + 8]; rb[i * 3 + 1] = in[o + 9]; rb[i * 3 + 2] = in[o + 10];
    }
    flag_block(blk_flag, bkey);
  }
}


}  // namespace mpm
