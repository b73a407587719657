// K6, second generation: relation-query cross-attention with FULL-LINE Q / context traffic.
//
// Same arithmetic, mask semantics, tile order and per-unit matrix-core code as psg_xattn_mfma.hip (HF-IB:464-466,
// 487-496 driven by V4:168-170, 179-185; see that file's header).  What changes is how a unit's 32 x 64 Q tile
// gets in and its context tile gets out.  The first-generation kernel loaded the Q fragments straight into the
// MFMA B-operand layout: 4 instructions per unit, each touching 32 cache lines for 32 bytes apiece, and stored
// the context with 8 instructions of 16 bytes per line; a copy-only build of it ran as slowly as the real one
//  - the kernel was bound by the number of line touches, not by bytes or flops.  Here
//   * Q arrives by LDS-DMA (global_load_lds_dwordx4): one instruction = 8 rows x 128 B, i.e. 8 whole lines, into
//     a per-wave ring of two 4 KiB slots; the bank-conflict swizzle is applied to the SOURCE address (the LDS
//     image of a DMA is lane-linear), and the fragment reads undo it; the pair ids of the unit ride along as a
//     fifth, 4-byte-per-lane DMA, so no ordinary load sits next to the DMAs (hipcc would drain the whole queue
//     at its first use);
//   * the DMA of unit u+2 is issued when unit u has left its slot: a full unit time (several microseconds) of
//     lead, counted vmcnt waits only;
//   * the context tile goes back through the same slot: 8-byte swizzled LDS writes from the accumulator layout,
//     16-byte reads in row order, four 16-byte stores per lane = 8 whole lines per instruction;
//   * one 8-wave workgroup per CU (one copy of K_h / V_h^T in LDS instead of two).
// LDS: K/V image + 8 waves x 2 slots x (4096 + 256) B; geometries whose image leaves no room (L > 320) take the
// first-generation kernel.
#include <type_traits>

#include "psg_common.h"

typedef float xd_f32x16 __attribute__((ext_vector_type(16)));
typedef float xd_f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t xd_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t xd_u32x2 __attribute__((ext_vector_type(2)));

#define XD_KSTRIDE 144   // bytes per K row in LDS: 64 bf16 + 16 B pad
#define XD_SLOT 4352     // 4096 B tile + 256 B pair ids
#define XD_CLS_COST 3     // a cls tile costs about this many pair tiles (static schedule)

__device__ __forceinline__ float xd_xchg_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xd_xchg_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// LDS traffic of the per-wave slots is inline asm: a compiler-visible ds access next to a pending LDS-DMA makes
// hipcc wait vmcnt(0) (it treats the DMA as an LDS write that may alias), which would serialise DMA and compute.
__device__ __forceinline__ xd_u32x4 xd_lds_read128(uint32_t a) {
  xd_u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory");
  return v;
}
__device__ __forceinline__ uint32_t xd_lds_read32(uint32_t a) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a) : "memory");
  return v;
}
__device__ __forceinline__ void xd_lds_write64(uint32_t a, xd_u32x2 v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v) : "memory");
}
__device__ __forceinline__ void xd_lds_write32(uint32_t a, uint32_t v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory");
}
template <int N_>
__device__ __forceinline__ void xd_vmwait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

template <typename E, int NC, int XD_WAVES>   // NC = key chunks of 128 (L <= 128 NC); XD_WAVES waves per workgroup (1 per CU)
__global__ void __launch_bounds__(XD_WAVES * 64, (XD_WAVES + 3) / 4)
cross_attn_dma_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
                      const uint64_t* __restrict__ bits, int words, const int32_t* __restrict__ pair_index, int N,
                      int64_t R, int L, int nq, int heads, int policy, uint16_t* __restrict__ out,
                      long long* __restrict__ trace, int poll_wt, int dyn, const int32_t* __restrict__ q_index,
                      const uint16_t* __restrict__ q_cls) {
  // bit 0: poll mode; bit 1: output rows stored write-through (option xattn_wt: no dirty L2 lines to flush when the launch
  // ends: 70.0 -> 67.6 us in situ at C2; `nt` stores measured no better, non-temporal Q tiles 1-4 us WORSE - round 6)
  const int poll = poll_wt & 1, wt = poll_wt & 2;
  // q_index != nullptr (nq == 33 only): q holds the 33 projected query rows per PROMPT, pair p reads block q_index[p] - the
  // pair tiles look the block up with one scalar load (the pair is wave-uniform), the cls tiles (row 0 of 32 different
  // pairs) read q_cls [P][hidden], which the caller gathered (P rows instead of 33 P: psg_qformer_cross_attn_indexed)
  // trace != nullptr (psg_set_trace_buffer(PSG_TRACE_CROSS_ATTN), debugging only): 32 stamps per wave:
  // [0] start, [1] K/V staged, [2] = [1], [3 + i] end of unit i, [31] unit count
  long long* tr = trace ? trace + ((int64_t)blockIdx.x * XD_WAVES + (threadIdx.x >> 6)) * 32 : nullptr;
  if (tr && (threadIdx.x & 63) == 0) tr[0] = __builtin_readcyclecounter();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Lpad = (L + 31) & ~31;
  const int NT = Lpad >> 5;
  const int VS = Lpad * 2 + 16;  // bytes per V^T row in LDS
  unsigned char* k_lds = smem;
  unsigned char* vt_lds = smem + (size_t)Lpad * XD_KSTRIDE;
  unsigned char* mean_lds = vt_lds + (size_t)64 * VS;   // 64 floats: mean of V_h over the L keys
  uint64_t* bits_lds = reinterpret_cast<uint64_t*>(mean_lds + 256);
  const int bits_bytes = (N * words * 8 + 15) & ~15;
  unsigned char* slots = reinterpret_cast<unsigned char*>(bits_lds) + bits_bytes;
  const int h = blockIdx.x % heads;
  const int g = blockIdx.x / heads;
  const int G = gridDim.x / heads;
  const int hidden = heads * 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int r8 = lane >> 3, pc = lane & 7;               // row-order view of a wave: 8 rows x 8 16-byte pieces

  const bool aligned = nq == 33;
  const int64_t P = R / nq;
  const int64_t NCLS = aligned ? (P + 31) >> 5 : 0;
  const int64_t ntile = aligned ? P + NCLS : (R + 31) >> 5;
  const unsigned char* kfrag_base = k_lds + l31 * XD_KSTRIDE + hi * 16;
  const unsigned char* vfrag_base = vt_lds + l31 * VS + hi * 16;
  unsigned char* my_slots = slots + wid * (2 * XD_SLOT);
  const uint32_t slot_lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)my_slots;
  const uint32_t mean_lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)mean_lds;

  // row rr (0..31) of a tile: global row, validity, pair
  auto tile_row = [&](int64_t tile, int rr, int64_t& row, bool& valid, int64_t& pair) {
    if (aligned) {
      if (tile >= NCLS) {
        pair = tile - NCLS;
        row = pair * 33 + 1 + rr;
        valid = true;
      } else {
        const int64_t pr = tile * 32 + rr;
        valid = pr < P;
        pair = valid ? pr : P - 1;
        row = pair * 33;
      }
    } else {
      row = tile * 32 + rr;
      valid = row < R;
      if (!valid) row = R - 1;
      pair = row / nq;
    }
  };
  // DMA of one unit into slot s: 4 x (8 rows x 128 B), piece (pc ^ r8) of row 8 i + r8 lands in slot position pc
  // of that row; + the pair ids of rows 0..31 (lanes 32..63 repeat them)
  // poll mode: the unit's arrival is read off its LDS slot - the pair-id words (the LAST of the five DMAs; loads land in
  // order) are set to a sentinel no pair id can take before the DMAs are issued, and the consumer spins on them.  A
  // vmcnt count cannot tell the Q tile of unit u from the context stores of unit u - 1 that sit behind it in the same
  // queue (gfx9 counts loads and stores together, and stores retire only when L2 has them): waiting "until at most
  // the DMAs of unit u + 1 are outstanding" made every unit wait for the previous unit's stores to retire.
  auto issue = [&](int64_t tile, int s) {
    unsigned char* dst = my_slots + s * XD_SLOT;
    if (poll) {
      xd_lds_write32(slot_lds0 + (uint32_t)(s * XD_SLOT) + 4096u + (uint32_t)(lane * 4), 0xffffffffu);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the sentinel is in LDS before a DMA can overwrite it
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int64_t row, pair;
      bool valid;
      tile_row(tile, 8 * i + r8, row, valid, pair);
      const uint16_t* src = q + row * hidden + h * 64 + ((pc ^ r8) * 8);
      if (q_index) {
        if (tile >= NCLS) {
          const int qi = q_index[__builtin_amdgcn_readfirstlane((int)pair)];
          src = q + ((int64_t)qi * 33 + 1 + 8 * i + r8) * hidden + h * 64 + ((pc ^ r8) * 8);
        } else {
          src = q_cls + pair * hidden + h * 64 + ((pc ^ r8) * 8);
        }
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    }
    {
      int64_t row, pair;
      bool valid;
      tile_row(tile, l31, row, valid, pair);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pair_index + pair),
                                       (__attribute__((address_space(3))) void*)(dst + 4096), 4, 0, 0);
    }
  };

  const float rcpN = 1.0f / (float)N;
  // Static, cost-aware schedule over the W waves that own head h.  The cls tiles (per-row masks, nearly every key
  // tile) cost about XD_CLS_COST pair tiles.  They are dealt round-robin first; the waves that got one more than
  // the others (w < rem) sit out the first XD_CLS_COST rounds of the pair tiles.  unit n of wave w -> tile id.
  const int64_t W = (int64_t)G * XD_WAVES;
  const int64_t w = (int64_t)g * XD_WAVES + wid;
  const int64_t cls_rem = NCLS % W;
  const int64_t my_cls = NCLS / W + (w < cls_rem ? 1 : 0);
  const int64_t A = W - cls_rem;                                   // waves taking part in the skipped rounds
  // Dynamic deal inside the workgroup (dyn): the unit cost varies with the pair's mask (active key tiles), so with the
  // static deal the slowest of the 2016 waves ends ~20 % after the mean.  Queues in global memory do not pay here (a
  // returning atomic on 12 hot words: 20 us per pull with 2016 pullers, measured), but most of the spread is BETWEEN THE
  // WAVES OF ONE WORKGROUP: the workgroup owns the tiles g, g + G, g + 2G, ... of its head (cls tiles first) and its
  // eight waves draw the next one from an LDS counter (ds_add_rtn: ~100 ns, no memory traffic).
  const int64_t ncls_g = dyn ? (NCLS > g ? (NCLS - g + G - 1) / G : 0) : 0;
  uint32_t* cursor = reinterpret_cast<uint32_t*>(slots + (size_t)XD_WAVES * 2 * XD_SLOT);   // 16 bytes behind the slots
  const uint32_t cursor_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)cursor;
  auto tile_of = [&](int64_t n) -> int64_t {                       // n-th tile of this workgroup; -1: none
    if (n < ncls_g) return g + n * G;
    const int64_t idx = g + (n - ncls_g) * G;
    return idx < ntile - NCLS ? NCLS + idx : -1;
  };
  auto draw = [&]() -> int64_t {
    uint32_t r = 0;
    if (lane == 0) {
      const uint32_t one = 1u;
      asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(cursor_addr), "v"(one) : "memory");
    }
    return tile_of((int64_t)__builtin_amdgcn_readfirstlane(r));
  };
  auto tile_at = [&](int64_t n) -> int64_t {                       // -1: no such unit
    if (n < my_cls) return w + n * W;                               // cls tiles come first (ids [0, NCLS))
    const int64_t kk = n - my_cls;
    int64_t idx;
    if (w < cls_rem) idx = XD_CLS_COST * A + kk * W + w;
    else idx = kk < XD_CLS_COST ? kk * A + (w - cls_rem) : XD_CLS_COST * A + (kk - XD_CLS_COST) * W + w;
    const int64_t npair = ntile - NCLS;
    return idx < npair ? NCLS + idx : -1;
  };
  union {
    uint32_t u[4];
    typename E::v8 v;
  } b_one;
  b_one.u[0] = hi ? 0u : E::ONE;
  b_one.u[1] = b_one.u[2] = b_one.u[3] = 0u;
  const float C8 = 0.125f * 1.4426950408889634f;
  const float bias_raw = policy == PSG_EMPTY_UNIFORM ? -3.4028234663852886e38f : -80000.0f;

  // ---- prologue: every global request of the workgroup is issued before the first wait - K_h / V_h (registers),
  // the object bit table, then the Q tiles of the first two units by DMA - so the staging costs one memory round
  // trip (a load -> LDS-write loop, or the DMAs ahead of plain loads, paid one per step: 8 us of 54)
  constexpr int IT = NC;                                 // (Lpad/2 key pairs * 8 chunks) / (64 XD_WAVES) threads <= NC
  uint4 kv[IT][2], vv[IT][2];
  {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = tid + it * (XD_WAVES * 64);
      const int m = e >> 3, c = e & 7;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int key = 2 * m + u;
        kv[it][u] = make_uint4(0, 0, 0, 0);
        vv[it][u] = make_uint4(0, 0, 0, 0);
        if (key < L) {
          kv[it][u] = *reinterpret_cast<const uint4*>(k + (int64_t)key * hidden + h * 64 + c * 8);
          vv[it][u] = *reinterpret_cast<const uint4*>(v + (int64_t)key * hidden + h * 64 + c * 8);
        }
      }
    }
  }
  uint64_t bitreg[2] = {0, 0};
  const int nbw = N * words;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + i * (XD_WAVES * 64);
    if (e < nbw) bitreg[i] = bits[e];
  }
  // (dyn: the first two units of a wave are fixed - they are requested before the first barrier; the cursor starts behind them)
  const int64_t t_u0 = dyn ? tile_of(wid) : tile_at(0), t_u1 = dyn ? tile_of(XD_WAVES + wid) : tile_at(1);
  if (dyn && tid == 0) *cursor = 2 * XD_WAVES;
  if (t_u0 >= 0) issue(t_u0, 0);
  if (t_u1 >= 0) issue(t_u1, 1);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + i * (XD_WAVES * 64);
    if (e < nbw) bits_lds[e] = bitreg[i];
  }
  for (int e = tid + 2 * (XD_WAVES * 64); e < nbw; e += XD_WAVES * 64) bits_lds[e] = bits[e];   // N * words > 1024
  {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = tid + it * (XD_WAVES * 64);
      const int m = e >> 3, c = e & 7;
      if (2 * m < Lpad) {
        *reinterpret_cast<uint4*>(k_lds + (2 * m) * XD_KSTRIDE + c * 16) = kv[it][0];
        *reinterpret_cast<uint4*>(k_lds + (2 * m + 1) * XD_KSTRIDE + c * 16) = kv[it][1];
        const int o = (2 * m) & 15;
        const int pos = (o & 3) | ((o & 8) >> 1) | ((o & 4) << 1);  // swap key bits 2 <-> 3 inside a 16-key group
        const int kcol = (((2 * m) & ~15) | pos) * 2;
        const uint32_t a[4] = {vv[it][0].x, vv[it][0].y, vv[it][0].z, vv[it][0].w};
        const uint32_t bq[4] = {vv[it][1].x, vv[it][1].y, vv[it][1].z, vv[it][1].w};
#pragma unroll
        for (int d2 = 0; d2 < 4; ++d2) {
          const uint32_t lo = (a[d2] & 0xffffu) | (bq[d2] << 16);
          const uint32_t hi2 = (a[d2] >> 16) | (bq[d2] & 0xffff0000u);
          *reinterpret_cast<uint32_t*>(vt_lds + (c * 8 + 2 * d2) * VS + kcol) = lo;
          *reinterpret_cast<uint32_t*>(vt_lds + (c * 8 + 2 * d2 + 1) * VS + kcol) = hi2;
        }
      }
    }
  }
  __syncthreads();
  {                                                     // mean of V_h over the keys (pad keys hold zeros): 8 lanes per
    const int d = tid >> 3, part = tid & 7;             // dim, each summing every 8th 16-byte piece of the V^T row
    if (d < 64) {
      float sum = 0.f;
      const uint16_t* vr = reinterpret_cast<const uint16_t*>(vt_lds + d * VS);
      for (int kk = part * 8; kk < Lpad; kk += 64) {
        const uint4 x = *reinterpret_cast<const uint4*>(vr + kk);
        const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) sum += E::to_f32((uint16_t)(xs[e] & 0xffffu)) + E::to_f32((uint16_t)(xs[e] >> 16));
      }
      sum += __shfl_xor(sum, 1);
      sum += __shfl_xor(sum, 2);
      sum += __shfl_xor(sum, 4);
      if (part == 0) reinterpret_cast<float*>(mean_lds)[d] = sum / (float)L;
    }
  }
  __syncthreads();

  if (tr && lane == 0) tr[1] = tr[2] = __builtin_readcyclecounter();

  // ---- one unit: fragments and pair id from the slot, attention, context tile back through the slot ----
  auto run_unit = [&](int64_t tile, int s, auto al_tag) {
    constexpr bool AL = decltype(al_tag)::value;
    const uint32_t sl = slot_lds0 + (uint32_t)(s * XD_SLOT);
    // fragment (row l31, piece c = 2 s4 + hi) sits at slot position c ^ (l31 & 7) of its row
    union {
      xd_u32x4 u;
      typename E::v8 b;
    } qf[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
      qf[s4].u = xd_lds_read128(sl + (uint32_t)(l31 * 128 + (((2 * s4 + hi) ^ (l31 & 7)) * 16)));
    const int pidx = (int)xd_lds_read32(sl + 4096u + (uint32_t)(l31 * 4));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    int oi, oj;
    if (N <= 1024) {
      oi = (int)(((float)pidx + 0.5f) * rcpN);
      oj = pidx - oi * N;
    } else {
      oi = pidx / N;
      oj = pidx % N;
    }
    const uint32_t* wi = reinterpret_cast<const uint32_t*>(bits_lds + (int64_t)oi * words);
    const uint32_t* wj = reinterpret_cast<const uint32_t*>(bits_lds + (int64_t)oj * words);
    uint32_t needmask = 0;      // wave-uniform
    bool force_all = false;
    bool use_mean = false;
    if constexpr (AL) {
      for (int t = 0; t < NT; ++t)
        needmask |= ((__builtin_amdgcn_readfirstlane(wi[t] | wj[t]) != 0u) ? 1u : 0u) << t;
      if (needmask == 0u) {
        if (policy == PSG_EMPTY_UNIFORM) {
          use_mean = true;                                // uniform softmax over the L real keys: mean_k V[k]
        } else {
          force_all = true;
          needmask = NT >= 32 ? 0xffffffffu : (1u << NT) - 1u;
        }
      }
    } else {
      bool row_empty = true;
      for (int t = 0; t < NT; ++t) {
        const uint32_t w = wi[t] | wj[t];
        row_empty = row_empty && (w == 0u);
        needmask |= (__any(w != 0u) ? 1u : 0u) << t;
      }
      if (__any(row_empty)) needmask = NT >= 32 ? 0xffffffffu : (1u << NT) - 1u;
    }

    xd_f32x16 o0 = {0}, o1 = {0};
    float m_run = -INFINITY, l_run = 0.f;
    if (use_mean) {                                       // (asm reads: see xd_lds_read128)
      xd_u32x4 m0[4], m1[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        m0[rr] = xd_lds_read128(mean_lds_addr + (uint32_t)(8 * rr + 4 * hi) * 4u);
        m1[rr] = xd_lds_read128(mean_lds_addr + (uint32_t)(32 + 8 * rr + 4 * hi) * 4u);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o0[4 * rr + e] = __uint_as_float(m0[rr][e]);
          o1[4 * rr + e] = __uint_as_float(m1[rr][e]);
        }
      l_run = 1.0f;
    }
    while (needmask != 0u) {
      const int t = __builtin_ctz(needmask);
      needmask &= needmask - 1u;
      uint32_t word = wi[t] | wj[t];
      const int left = L - 32 * t;                       // real keys in this tile (>= 1)
      xd_f32x16 acc;
      const unsigned char* kp = kfrag_base + t * 32 * XD_KSTRIDE;
      if constexpr (AL) {
        if (force_all) word = left >= 32 ? 0xffffffffu : (1u << left) - 1u;
        union {
          uint32_t u[4];
          typename E::v8 v;
        } a_bias;
        a_bias.u[0] = (((word >> l31) & 1u) | (uint32_t)hi) ? 0u : E::NEG_2_15;   // 0 / -2^15 in bf16
        a_bias.u[1] = a_bias.u[2] = a_bias.u[3] = 0u;
        acc = E::mfma32(a_bias.v, b_one.v, (xd_f32x16){0});
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const typename E::v8 a = *reinterpret_cast<const typename E::v8*>(kp + s4 * 32);
          acc = E::mfma32(a, qf[s4].b, acc);
        }
      } else {
        {
          const typename E::v8 a = *reinterpret_cast<const typename E::v8*>(kp);
          acc = E::mfma32(a, qf[0].b, (xd_f32x16){0});
        }
#pragma unroll
        for (int s4 = 1; s4 < 4; ++s4) {
          const typename E::v8 a = *reinterpret_cast<const typename E::v8*>(kp + s4 * 32);
          acc = E::mfma32(a, qf[s4].b, acc);
        }
        const uint32_t inv = ~word >> (4 * hi);
        const bool has_pad = left < 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int koff = (r & 3) + 8 * (r >> 2);
          const int mb = __builtin_amdgcn_sbfe((int)inv, koff, 1);  // -1 if masked
          float y = acc[r] + __uint_as_float((uint32_t)mb & __float_as_uint(bias_raw));
          if (has_pad && (koff + 4 * hi >= left)) y = -INFINITY;
          acc[r] = y;
        }
      }
      float cmax = acc[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) cmax = fmaxf(cmax, acc[r]);
      cmax = xd_xchg_max(cmax);
      const float m_new = fmaxf(m_run, cmax);
      float alpha, csum = 0.f;
      if constexpr (AL) {
        const float mc = m_new * C8;
        alpha = __builtin_amdgcn_exp2f(fmaf(m_run, C8, -mc));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(fmaf(acc[r], C8, -mc));
          acc[r] = pv;
          csum += pv;
        }
      } else {
        alpha = __builtin_amdgcn_exp2f((m_run - m_new) * C8);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f((acc[r] - m_new) * C8);
          acc[r] = pv;
          csum += pv;
        }
      }
      csum = xd_xchg_sum(csum);
      l_run = l_run * alpha + csum;
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o0[r] *= alpha;
        o1[r] *= alpha;
      }
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        union {
          uint32_t u[4];
          typename E::v8 v;
        } pf;
#pragma unroll
        for (int e = 0; e < 4; ++e) pf.u[e] = E::pack(acc[8 * gg + 2 * e], acc[8 * gg + 2 * e + 1]);
        const unsigned char* vp = vfrag_base + (t * 32 + 16 * gg) * 2;
        const typename E::v8 a0 = *reinterpret_cast<const typename E::v8*>(vp);
        const typename E::v8 a1 = *reinterpret_cast<const typename E::v8*>(vp + 32 * VS);
        o0 = E::mfma32(a0, pf.v, o0);
        o1 = E::mfma32(a1, pf.v, o1);
      }
    }
    // ---- context tile: lane (q = l31, hi) holds O[q][32 dt + (r&3) + 8 (r>>2) + 4 hi]; 16-byte piece index of
    // dims 8 rr + 4 hi .. + 3 is rr (first half) / 4 + rr (second half); swizzled by the row like the Q image
    const float inv_l = 1.0f / l_run;
    // the MFMA results feed an asm consumer: give the matrix pipe its wait states explicitly
    asm volatile("s_nop 15" ::: "memory");
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      xd_u32x2 w0, w1;
      w0[0] = E::pack(o0[4 * rr] * inv_l, o0[4 * rr + 1] * inv_l);
      w0[1] = E::pack(o0[4 * rr + 2] * inv_l, o0[4 * rr + 3] * inv_l);
      w1[0] = E::pack(o1[4 * rr] * inv_l, o1[4 * rr + 1] * inv_l);
      w1[1] = E::pack(o1[4 * rr + 2] * inv_l, o1[4 * rr + 3] * inv_l);
      xd_lds_write64(sl + (uint32_t)(l31 * 128 + ((rr ^ (l31 & 7)) * 16) + hi * 8), w0);
      xd_lds_write64(sl + (uint32_t)(l31 * 128 + (((4 + rr) ^ (l31 & 7)) * 16) + hi * 8), w1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // wave-private slot: the wave's own writes have landed
    xd_u32x4 orow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) orow[i] = xd_lds_read128(sl + (uint32_t)((8 * i + r8) * 128 + ((pc ^ r8) * 16)));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int64_t row, pair;
      bool valid;
      tile_row(tile, 8 * i + r8, row, valid, pair);
      if (AL || valid) {
        uint16_t* dst = out + row * hidden + h * 64 + pc * 8;
        if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(orow[i]) : "memory");
        else *reinterpret_cast<uint4*>(dst) = make_uint4(orow[i][0], orow[i][1], orow[i][2], orow[i][3]);
      }
    }
  };

  // ---- unit loop: unit u sits in slot u & 1; its DMA (5 instructions) was issued one unit earlier.  Loads
  // complete in order, stores may not: waiting until at most the 5 DMAs of unit u + 1 are outstanding also
  // retires the previous unit's 4 stores, which is conservative and correct.
  int64_t t = t_u0, t1 = t_u1;
  for (int u = 0; t >= 0; ++u) {
    if (poll) {
      const uint32_t pa = slot_lds0 + (uint32_t)((u & 1) * XD_SLOT) + 4096u + (uint32_t)(lane * 4);
      for (int spin = 0; spin < (1 << 22); ++spin) {          // bounded: a lost DMA must not hang the GPU
        const uint32_t got = xd_lds_read32(pa);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (__all(got != 0xffffffffu)) break;
        __builtin_amdgcn_s_sleep(1);
      }
    } else if (t1 >= 0) {
      xd_vmwait<5>();
    } else {
      xd_vmwait<0>();
    }
    if (aligned && t >= NCLS) run_unit(t, u & 1, std::true_type{});
    else run_unit(t, u & 1, std::false_type{});
    const int64_t t2 = dyn ? (t1 >= 0 ? draw() : -1) : tile_at(u + 2);
    if (t2 >= 0) issue(t2, u & 1);
    if (tr && lane == 0) {
      if (u < 28) tr[3 + u] = __builtin_readcyclecounter();
      tr[31] = u + 1;
    }
    t = t1;
    t1 = t2;
  }
}

static size_t xd_lds_bytes(int N, int words, int L, int waves) {
  const int Lpad = (L + 31) & ~31;
  return (size_t)Lpad * XD_KSTRIDE + (size_t)64 * (Lpad * 2 + 16) + 256 + (((size_t)N * words * 8 + 15) & ~(size_t)15) +
         (size_t)waves * 2 * XD_SLOT + 16;
}

// smallest LDS footprint of the kernel family (8 waves); the dispatcher compares it with the 160 KiB of a CU
extern "C" int psg_cross_attn_dma_lds_bytes(int N, int words, int L) { return (int)xd_lds_bytes(N, words, L, 8); }

template <typename E>
static int xd_launch(psg_ctx* ctx, const void* q, const void* k, const void* v, const uint64_t* bits, int words,
                     const int32_t* pair_index, int N, int P, int L, int nq, int heads, int policy, void* out,
                     hipStream_t st, const int32_t* q_index, const void* q_cls) {
  const int Lpad = (L + 31) & ~31;
  // 8 waves per CU, ten for launches with few tiles per wave when the K/V image leaves room for ten slot pairs
  // (L <= 256): with the tiles drawn from the LDS counter, BASELINE C2 (2579 tiles per head, 15 per wave) runs 6 % faster
  // with ten (more Q streams in flight through the staging phase and the tail), C4 (61 per wave) 3 % slower.
  // Option xattn_waves: 8 = this rule, 10 = always ten, 0 = always eight.
  const int64_t ntile0 = nq == 33 ? (int64_t)P + (P + 31) / 32 : ((int64_t)P * nq + 31) / 32;
  const int64_t per_wave8 = ntile0 / (((int64_t)ctx->num_cu / heads > 0 ? (int64_t)ctx->num_cu / heads : 1) * 8);
  const bool want10 = ctx->opt.xattn_waves == 10 || (ctx->opt.xattn_waves == 8 && per_wave8 < 32 && per_wave8 >= 4);
  const int waves = (want10 && xd_lds_bytes(N, words, L, 10) <= 160 * 1024) ? 10 : 8;
  const size_t lds = xd_lds_bytes(N, words, L, waves);
  const int NC = (Lpad / 32 + 3) / 4;
  PSG_REQUIRE(lds <= 160 * 1024 && NC >= 1 && NC <= 3, PSG_ERR_UNSUPPORTED,
              "psg_qformer_cross_attn(dma): L=%d needs %zu B of LDS", L, lds);
  const int64_t R = (int64_t)P * nq;
  const int64_t ntile = nq == 33 ? (int64_t)P + (P + 31) / 32 : (R + 31) / 32;
  int64_t G = ctx->num_cu / heads;                       // one workgroup per CU
  const int64_t maxG = (ntile + waves - 1) / waves;
  if (G > maxG) G = maxG;
  if (G < 1) G = 1;
  const int dyn = (nq == 33 && ctx->opt.xattn_dynamic && ntile >= 4 * G * waves) ? 1 : 0;   // short launches stay static
  const int64_t trace_n = G * heads * waves * 32;
  long long* trace = (ctx->trace_kind == PSG_TRACE_CROSS_ATTN && ctx->trace_words >= trace_n) ? ctx->trace : nullptr;
#define XDLAUNCH(NC_, W_)                                                                                          \
  do {                                                                                                             \
    hipError_t e = hipFuncSetAttribute((const void*)cross_attn_dma_kernel<E, NC_, W_>,                                \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
    if (e != hipSuccess) {                                                                                         \
      psg_set_error("psg_qformer_cross_attn(dma): hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));       \
      return PSG_ERR_HIP;                                                                                          \
    }                                                                                                              \
    cross_attn_dma_kernel<E, NC_, W_><<<(unsigned)(G * heads), W_ * 64, lds, st>>>(                                   \
        (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v, bits, words, pair_index, N, R, L, nq, heads,   \
        policy, (uint16_t*)out, trace, (ctx->opt.xattn_poll & 1) | (ctx->opt.xattn_wt ? 2 : 0), dyn, q_index,       \
        (const uint16_t*)q_cls);                                                                                    \
  } while (0)
  if (waves == 10) {
    if (NC == 1) XDLAUNCH(1, 10);
    else XDLAUNCH(2, 10);
  } else {
    if (NC == 1) XDLAUNCH(1, 8);
    else if (NC == 2) XDLAUNCH(2, 8);
    else XDLAUNCH(3, 8);
  }
#undef XDLAUNCH
  PSG_CHECK_LAUNCH("psg_qformer_cross_attn(dma)");
  return PSG_OK;
}

int psg_cross_attn_dma_launch(psg_ctx* ctx, const void* q, const void* k, const void* v, const uint64_t* bits,
                              int words, const int32_t* pair_index, int N, int P, int L, int nq, int heads, int policy,
                              void* out, int dtype, hipStream_t st, const int32_t* q_index, const void* q_cls) {
  PSG_DISPATCH_E16(dtype, "psg_qformer_cross_attn(dma)",
                   return xd_launch<E>(ctx, q, k, v, bits, words, pair_index, N, P, L, nq, heads, policy, out, st, q_index,
                                       q_cls));
}
