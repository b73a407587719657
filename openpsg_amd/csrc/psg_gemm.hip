// K15: weight-streaming "skinny" GEMM for the batched LMM decode step.
//
//   part[s][M][N] = x[M][Ks] . w[N][Ks]^T      M <= 32 rows (the selected pairs), bf16 in, fp32 out
//   y = sum_s part[s]                           (summed, in split order, by the CONSUMER kernel)
//
// Replaces the HF Linear layers of the Llama decoder (q/k/v/o_proj, gate/up/down_proj, lm_head;
// HF-LL:163-177, 243-281) for the decode steps of V4:305-312.  The reference runs them at batch 1,
// once per selected pair; here all selected pairs share one pass, so each weight byte leaves HBM
// once per step.  The kernel is HBM-bound by construction (2*M flops per weight element); its
// roofline is the HBM stream of w.
//
// What measurements on MI355X dictated (tools/membench, tools/bench_kernels.py):
//   * the MFMA-A-layout weight read (16 rows x 64 B per instruction) streams at 5.1-5.4 TB/s by
//     itself, but a first version that fetched the x operand from L2 with per-lane loads ran at
//     2.7 TB/s: x costs 2.25x the weight bytes through L2->L1 and 2 of every 3 vector-memory
//     issue slots.  With x hoisted out of the loop the same kernel reached 4.8-5.25 TB/s.
//   => x must come from LDS, which needs waves that share a K range: so K is split ACROSS
//      workgroups (grid.y) and a workgroup = 4 waves x 16-row slabs that walk the same K range.
//   * the split-K partials are not reduced here: every consumer of a decode projection is one of
//     our own row kernels (rmsnorm, rotary+KV write, SwiGLU gate, greedy step), which sum the S
//     fp32 slices in a fixed order while loading - deterministic, no atomics, no extra launch.
//
// Work layout (gfx950):
//   * grid = (G, S); workgroup (gx, by) owns the K blocks (64 elements each)
//     [KB by / S, KB (by+1) / S) and walks the 64-row slabs gx, gx + G, ... persistently; wave w owns
//     rows 16 w .. 16 w + 15 of each slab;
//   * x[0..M)[K range] is staged ONCE per workgroup into LDS (row stride padded by 16 B) and reused
//     for every slab; the first weight batch is already in flight while that happens;
//   * lane (n = lane&15, kq = lane>>4) reads bytes [32 kq, 32 kq + 32) of row n's 128-byte block
//     with two non-temporal 16-byte loads straight into the A operands of two
//     v_mfma_f32_16x16x32_bf16 (a weight byte is used once: no LDS round trip for w).  K inside a
//     block is thereby permuted; x fragments are read from LDS with the same permutation;
//   * two register sets of SG_U blocks each are software-pipelined over the flattened
//     (slab, batch) stream, so a wave always has one batch of loads in flight while it multiplies
//     the other, across slab boundaries too.
#include <stdlib.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "psg_common.h"

typedef float gf32x4_t __attribute__((ext_vector_type(4)));

#define SG_ROWS 64      // weight rows per workgroup (4 waves x 16)
#define SG_U 4          // K blocks per register set

template <typename E>
struct SgFrag {
  typename E::v8 v[SG_U][2];
};

template <typename E>
__device__ __forceinline__ void sg_load(SgFrag<E>& f, const uint16_t* __restrict__ wp, int kb) {
  using gbf16x8_t = typename E::v8;
#pragma unroll
  for (int u = 0; u < SG_U; ++u) {
    f.v[u][0] = __builtin_nontemporal_load(reinterpret_cast<const gbf16x8_t*>(wp + (int64_t)(kb + u) * 64));
    f.v[u][1] = __builtin_nontemporal_load(reinterpret_cast<const gbf16x8_t*>(wp + (int64_t)(kb + u) * 64 + 8));
  }
}

template <typename E>
__device__ __forceinline__ void sg_mma(const SgFrag<E>& f, const unsigned char* xs0, const unsigned char* xs1, int kbl,
                                       gf32x4_t& acc0, gf32x4_t& acc1) {
  using gbf16x8_t = typename E::v8;
#pragma unroll
  for (int u = 0; u < SG_U; ++u) {
    const int o = (kbl + u) * 128;
    const gbf16x8_t b00 = *reinterpret_cast<const gbf16x8_t*>(xs0 + o);
    const gbf16x8_t b01 = *reinterpret_cast<const gbf16x8_t*>(xs0 + o + 16);
    const gbf16x8_t b10 = *reinterpret_cast<const gbf16x8_t*>(xs1 + o);
    const gbf16x8_t b11 = *reinterpret_cast<const gbf16x8_t*>(xs1 + o + 16);
    acc0 = E::mfma16(f.v[u][0], b00, acc0);
    acc1 = E::mfma16(f.v[u][0], b10, acc1);
    acc0 = E::mfma16(f.v[u][1], b01, acc0);
    acc1 = E::mfma16(f.v[u][1], b11, acc1);
  }
}

template <typename E>
__global__ void __launch_bounds__(256) skinny_gemm_kernel(const uint16_t* __restrict__ x,
                                                          const uint16_t* __restrict__ w, float* __restrict__ part,
                                                          int M, int N, int K, int xstride) {
  using gbf16x8_t = typename E::v8;
  extern __shared__ __attribute__((aligned(16))) unsigned char xs[];   // [M][xstride bytes]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n = lane & 15, kq = lane >> 4;
  const int S = gridDim.y, by = blockIdx.y, G = gridDim.x, gx = blockIdx.x;
  const int KB = K >> 6;
  const int kbA = (int)(((int64_t)KB * by) / S), kbB = (int)(((int64_t)KB * (by + 1)) / S);
  const int nkb = kbB - kbA;
  const int nb = nkb / SG_U;                                        // full batches per slab
  const int nslab_all = (N + SG_ROWS - 1) / SG_ROWS;
  const int nslab = gx < nslab_all ? (nslab_all - gx + G - 1) / G : 0;   // slabs gx, gx+G, ... (persistent)
  // weight pointer of this lane for slab t (rows beyond N are clamped; their results are dropped)
  auto wrow = [&](int t) -> const uint16_t* {
    int r = (gx + t * G) * SG_ROWS + wid * 16 + n;
    r = r < N ? r : N - 1;
    return w + (int64_t)r * K + (int64_t)kbA * 64 + kq * 16;
  };
  const int total = nslab * nb;                                     // flattened (slab, batch) stream
  SgFrag<E> fa, fb;
  int lt = 0, lb = 0;                                               // load cursor
  auto load_next = [&](SgFrag<E>& f) {
    sg_load(f, wrow(lt), lb * SG_U);
    if (++lb == nb) { lb = 0; ++lt; }
  };
  if (total > 0) load_next(fa);                                     // in flight during the x staging

  // stage x[0..M)[kbA*64 .. kbB*64) once per workgroup: 16-byte pieces, coalesced along K
  const int pieces = nkb * 8;
  for (int e = tid; e < M * pieces; e += 256) {
    const int r = e / pieces, c = e - r * pieces;
    const uint4 v = *reinterpret_cast<const uint4*>(x + (int64_t)r * K + (int64_t)kbA * 64 + c * 8);
    *reinterpret_cast<uint4*>(xs + r * xstride + c * 16) = v;
  }
  __syncthreads();

  // x rows >= M are clamped: they only feed output columns >= M, which are never stored
  const unsigned char* xs0 = xs + min(n, M - 1) * xstride + kq * 32;
  const unsigned char* xs1 = xs + min(16 + n, M - 1) * xstride + kq * 32;
  gf32x4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  int ct = 0, cb = 0;                                               // compute cursor
  auto finish_slab = [&]() {
    const uint16_t* wp = wrow(ct);
    for (int kb = nb * SG_U; kb < nkb; ++kb) {                      // remainder K blocks of this slab
      const gbf16x8_t a0 = __builtin_nontemporal_load(reinterpret_cast<const gbf16x8_t*>(wp + (int64_t)kb * 64));
      const gbf16x8_t a1 = __builtin_nontemporal_load(reinterpret_cast<const gbf16x8_t*>(wp + (int64_t)kb * 64 + 8));
      const int o = kb * 128;
      acc0 = E::mfma16(a0, *reinterpret_cast<const gbf16x8_t*>(xs0 + o), acc0);
      acc1 = E::mfma16(a0, *reinterpret_cast<const gbf16x8_t*>(xs1 + o), acc1);
      acc0 = E::mfma16(a1, *reinterpret_cast<const gbf16x8_t*>(xs0 + o + 16), acc0);
      acc1 = E::mfma16(a1, *reinterpret_cast<const gbf16x8_t*>(xs1 + o + 16), acc1);
    }
    // D[row = weight row 4 kq + r][col = x row lane&15] -> part[by][m][n0 + 4 kq + r], 16-byte stores
    const int n0 = (gx + ct * G) * SG_ROWS + wid * 16;
    if (n0 + 16 <= N) {
      float* pp = part + ((int64_t)by * M) * N + n0 + 4 * kq;
      if (n < M) *reinterpret_cast<float4*>(pp + (int64_t)n * N) = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
      if (16 + n < M)
        *reinterpret_cast<float4*>(pp + (int64_t)(16 + n) * N) = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
    }
    acc0 = (gf32x4_t){0, 0, 0, 0};
    acc1 = (gf32x4_t){0, 0, 0, 0};
    cb = 0;
    ++ct;
  };
  if (nb == 0) {                                                    // K range shorter than one batch
    for (int t = 0; t < nslab; ++t) finish_slab();
    return;
  }
  for (int j = 0; j < total; j += 2) {
    if (j + 1 < total) load_next(fb);
    sg_mma(fa, xs0, xs1, cb * SG_U, acc0, acc1);
    if (++cb == nb) finish_slab();
    if (j + 1 >= total) break;
    if (j + 2 < total) load_next(fa);
    sg_mma(fb, xs0, xs1, cb * SG_U, acc0, acc1);
    if (++cb == nb) finish_slab();
  }
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA variant: the weight stream goes HBM -> LDS with global_load_lds_dwordx4 in FULL 128-byte
// lines (one instruction = 8 rows x 128 B; tools/membench measured 6.0-6.6 TB/s for this pattern vs
// 5.1-5.4 TB/s for the 16-rows-x-64-B pattern the MFMA A layout forces on direct-to-register loads).
// The LDS image of a DMA is lane-linear, so the bank-conflict swizzle is applied to the SOURCE:
// lane (r = lane>>3, p = lane&7) fetches piece p ^ r of row r, and the MFMA fragment read of
// (row n, piece c) looks at slot c ^ (n & 7).  Each wave owns a private 3-slot ring of
// UD-K-block batches; the only synchronisation is the issuing wave's own counted vmcnt.
// ---------------------------------------------------------------------------------------------
// Workgroup barrier for LDS traffic only.  __syncthreads() would also drain the vector-memory counter whenever an
// LDS-DMA is in flight (the DMA is a pending LDS write to the compiler): at every slab end the whole prefetch ring
// would be waited for and the weight stream would stall.  Ordering of the partial tile needs lgkmcnt only.
__device__ __forceinline__ void sgd_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void sgd_ds_write128(uint32_t lds_addr, gf32x4_t v) {
  // v holds MFMA results: the matrix-core -> LDS-store wait states are not inserted for an asm consumer
  asm volatile("s_nop 15\n\tds_write_b128 %0, %1" ::"v"(lds_addr), "v"(v) : "memory");
}
__device__ __forceinline__ gf32x4_t sgd_ds_read128(uint32_t lds_addr) {
  gf32x4_t v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr) : "memory");
  return v;
}

template <int N_>
__device__ __forceinline__ void sgd_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}
// wait until at most `newer` batches (2*UD DMA instructions each) issued after the wanted one are outstanding
template <int UD, int MAXN>
struct SgdWait {
  static __device__ __forceinline__ void go(int newer) {
    if (newer >= MAXN) sgd_wait<(MAXN * 2 * UD < 63 ? MAXN * 2 * UD : 63)>();
    else SgdWait<UD, MAXN - 1>::go(newer);
  }
};
template <int UD>
struct SgdWait<UD, 0> {
  static __device__ __forceinline__ void go(int) { sgd_wait<0>(); }
};

// ---- row-operation prologues of the fused decode projections (psg_skinny_gemm_fused) ------------------------------
// Outputs that other workgroups of the same launch read (the x operand) are stored WRITE-THROUGH: 8-byte relaxed
// agent-scope atomic stores lower to `global_store_dwordx2 ... sc1`, so no release fence is needed before the counter.
template <typename T>
__device__ __forceinline__ void pro_store4(T* p, int64_t i, const float (&v)[4]) {
  uint16_t h[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    T t;
    Act<T>::st(&t, 0, v[e]);
    h[e] = t.v;
  }
  const unsigned long long w = (unsigned long long)h[0] | ((unsigned long long)h[1] << 16) |
                               ((unsigned long long)h[2] << 32) | ((unsigned long long)h[3] << 48);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p + i), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// RMSNorm (+ residual add) of ONE row by a 512-thread workgroup, in the arithmetic ORDER of rmsnorm_kernel<T, 1>
// launched with VT threads (psg_rmsnorm: VT = 1024 for hidden >= 4096, else 256): thread t also plays virtual thread
// t + 512, the per-wave sums land in the virtual wave's slot and are added in slot order -> bit-identical results.
template <typename E, int VT>
__device__ __forceinline__ void pro_rmsnorm_row(const psg_prologue& p, int row, int M, int hidden,
                                                typename E::act* __restrict__ xout, float* s_part, int tid) {
  using T = typename E::act;
  constexpr int NV = VT > 512 ? VT / 512 : 1;
  const int lane = tid & 63, wid = tid >> 6;
  T* resid = reinterpret_cast<T*>(p.resid);
  float v[NV][4];
  float4 g[NV], d4[NV];
  typename Act<T>::raw4 vr[NV];
  bool ok[NV];
  int64_t idx[NV];
  int col[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int t = tid + 512 * k;
    ok[k] = t < VT && t * 4 < hidden;
    col[k] = ok[k] ? t * 4 : 0;
    idx[k] = (int64_t)row * hidden + col[k];
    vr[k] = Act<T>::ldr4(resid, idx[k]);
    g[k] = *reinterpret_cast<const float4*>(p.norm_w + col[k]);
  }
  const bool has_delta = p.in != nullptr;
  if (has_delta) {
    if (p.in_splits > 0) {
      ldn_splits<float4, NV>(p.in, p.in_splits, (int64_t)M * hidden, idx, d4);
    } else {
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        float t4[4];
        Act<T>::ld4(reinterpret_cast<const T*>(p.in), idx[k], t4);
        d4[k] = make_float4(t4[0], t4[1], t4[2], t4[3]);
      }
    }
  }
  float ss[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    Act<T>::cv4(vr[k], v[k]);
    if (has_delta) {
      const float d[4] = {Act<T>::rnd(d4[k].x), Act<T>::rnd(d4[k].y), Act<T>::rnd(d4[k].z), Act<T>::rnd(d4[k].w)};
#pragma unroll
      for (int e = 0; e < 4; ++e) v[k][e] = Act<T>::rnd(v[k][e] + d[e]);
    }
    ss[k] = 0.f;
    if (ok[k]) {
#pragma unroll
      for (int e = 0; e < 4; ++e) ss[k] += v[k][e] * v[k][e];
    }
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const float r = wave_sum(ss[k]);
    if (lane == 0) s_part[wid + 8 * k] = r;
  }
  __syncthreads();
  float tot = 0.f;
  for (int i = 0; i < VT / 64; ++i) tot += s_part[i];
  const float inv = 1.0f / sqrtf(tot / (float)hidden + p.eps);
#pragma unroll
  for (int k = 0; k < NV; ++k)
    if (ok[k]) {
      if (has_delta) Act<T>::st4(resid, idx[k], v[k]);          // read back by the next launch only: plain store
      const float o[4] = {g[k].x * (v[k][0] * inv), g[k].y * (v[k][1] * inv), g[k].z * (v[k][2] * inv),
                          g[k].w * (v[k][3] * inv)};
      pro_store4<T>(xout, idx[k], o);
    }
}

template <typename E, int WAVES, int UD, int SGD_SLOTS, int AUX, int XDMA, int PRO = 0>
__global__ void __launch_bounds__(WAVES * 64) skinny_gemm_dma_kernel(const uint16_t* __restrict__ x,
                                                                     const uint16_t* __restrict__ w,
                                                                     float* __restrict__ part, int M, int N, int K,
                                                                     int xstride, long long* __restrict__ trace,
                                                                     const psg_prologue pro) {
  using gbf16x8_t = typename E::v8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // trace != nullptr (psg_set_trace_buffer(PSG_TRACE_SKINNY_GEMM), debugging only): 8 cycle-counter stamps per wave
  long long* tr = trace ? trace + (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * WAVES + (threadIdx.x >> 6)) * 8
                        : nullptr;
  if (tr && (threadIdx.x & 63) == 0) tr[0] = __builtin_readcyclecounter();
  constexpr int ROWS = WAVES * 16;
  constexpr int BATCH_BYTES = UD * 2048;
  constexpr int RING_BYTES = SGD_SLOTS * BATCH_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n = lane & 15, kq = lane >> 4;
  const int S = gridDim.y, by = blockIdx.y, G = gridDim.x, gx = blockIdx.x;
  const int KB = K >> 6;
  const int kbA = (int)((unsigned)(KB * by) / (unsigned)S), kbB = (int)((unsigned)(KB * (by + 1)) / (unsigned)S);   // K <= 2^20
  const int nkb = kbB - kbA;
  const int nb = nkb / UD;
  const int nslab_all = (N + ROWS - 1) / ROWS;
  const int nslab = gx < nslab_all ? (nslab_all - gx + G - 1) / G : 0;
  constexpr int OT_PITCH = ROWS + 4;                                // output tile pitch (floats)
  unsigned char* ring = smem + wid * RING_BYTES;                    // this wave's private ring
  // [32][OT_PITCH] fp32 partial tile behind the rings, addressed by its LDS byte address (see finish_slab)
  const uint32_t otile_lds =
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(smem + WAVES * RING_BYTES);
  unsigned char* xs = smem + WAVES * RING_BYTES + 32 * OT_PITCH * 4;            // shared x slice [M][xstride]
  const int total = nslab * nb;

  // DMA source of this lane: row (lane>>3) of an 8-row group, piece (lane&7) ^ (lane>>3)
  const int dr = lane >> 3, dp = (lane & 7) ^ (lane >> 3);
  auto dma_src = [&](int t) -> const uint16_t* {
    int r = (gx + t * G) * ROWS + wid * 16 + dr;
    r = r + 8 < N ? r : (N - 9 > 0 ? N - 9 : 0);                    // keep rows r and r+8 in range (results dropped)
    return w + (int64_t)r * K + (int64_t)kbA * 64 + dp * 8;
  };
  int lt = 0, lb = 0, lj = 0;
  auto issue = [&]() {
    const uint16_t* src = dma_src(lt) + (int64_t)lb * UD * 64;
    unsigned char* dst = ring + (lj % SGD_SLOTS) * BATCH_BYTES;
#pragma unroll
    for (int u = 0; u < UD; ++u) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + u * 64),
                                       (__attribute__((address_space(3))) void*)(dst + u * 2048), 16, 0, AUX);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + u * 64 + (int64_t)8 * K),
                                       (__attribute__((address_space(3))) void*)(dst + u * 2048 + 1024), 16, 0, AUX);
    }
    ++lj;
    if (++lb == nb) { lb = 0; ++lt; }
  };
  const int pieces = nkb * 8;
  auto stage_x_dma = [&](auto aux_tag) {
    constexpr int XAUX = decltype(aux_tag)::value;
    const int cpr = (pieces + 63) >> 6;                             // 1 KiB chunks per row
    const int items = M * cpr;
    for (int it = wid; it < items; it += WAVES) {
      const int r = it / cpr, j = it - r * cpr;
      const int c = j * 64 + lane;
      if (c < pieces)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(x + (int64_t)r * K + (int64_t)kbA * 64 + c * 8),
            (__attribute__((address_space(3))) void*)(xs + r * xstride + j * 1024), 16, 0, XAUX);
    }
  };
  if constexpr (XDMA && PRO == 0) {
    // x slice by LDS-DMA as well: one instruction = up to 64 16-byte pieces of ONE row (lanes past the row's
    // slice are masked off), issued ahead of the weight stream, no VGPR round trip and no ordinary load next to
    // the DMAs (hipcc drains the whole DMA queue at every use of a plain load while a DMA is in flight)
    stage_x_dma(std::integral_constant<int, 0>{});
  }
  for (int i = 0; i < SGD_SLOTS - 1; ++i)
    if (i < total) issue();
  if (tr && lane == 0) tr[1] = __builtin_readcyclecounter();

  if constexpr (PRO != 0) {
    // ---- fused row operation: the first workgroups produce x while every ring fills ----
    static_assert(XDMA == 1 && WAVES == 8, "fused prologues are built for the 8-wave LDS-DMA variant");
    using T = typename E::act;
    const int wg = by * G + gx;
    float* scratch = reinterpret_cast<float*>(smem + WAVES * RING_BYTES);     // the partial-tile area, free until slab 0 ends
    unsigned units = 0;
    if constexpr (PRO == PSG_PRO_RMSNORM) {
      if (wg < M) {
        if (K >= 4096) pro_rmsnorm_row<E, 1024>(pro, wg, M, K, (T*)const_cast<uint16_t*>(x), scratch, tid);
        else pro_rmsnorm_row<E, 256>(pro, wg, M, K, (T*)const_cast<uint16_t*>(x), scratch, tid);
        units = 1;
      }
    }
    if (tr && lane == 0) tr[6] = __builtin_readcyclecounter();      // row operation computed (or nothing to do)
    // publish: every storing wave drains its write-through stores, then ONE lane counts the workgroup's units in
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && units) __hip_atomic_fetch_add(pro.sync, units, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // consume: one wave polls the ONE counter (relaxed, agent scope), bounded; x is then read with sc1 loads
    // (write-through producer + sc1 consumer: no acquire fence, and nothing of x was cached by this launch before)
    if (wid == 0) {
      const unsigned want = PRO == PSG_PRO_RMSNORM ? (unsigned)M : 0u;
      unsigned spins = 0;
      while (__hip_atomic_load(pro.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 22)) {                                  // ~seconds: a producer died; do not hang the GPU
          if (lane == 0) __hip_atomic_store(pro.sync + 1, 0x5047u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __syncthreads();
    if (tr && lane == 0) tr[7] = __builtin_readcyclecounter();      // counter complete
    stage_x_dma(std::integral_constant<int, 16>{});                 // aux 16 = sc1
  }

  // stage x once per workgroup (plain loads -> ds_write); the DMAs above are already in flight
  // (issuing all of a thread's x loads at once, with or without a division-free row/piece mapping, was
  // measured SLOWER, 3.42 vs 2.97 ms per decode step: 256 workgroups then hit the same 160 KB of x in L2 in one
  // burst; this loop spreads them)
  if constexpr (!XDMA) {
    constexpr int XU = 1;                                           // x loads in flight per thread: 2 measured slower (3.11 vs 2.97 ms/step), 8 slower still (3.42): they queue in front of the weight stream
    const int total_x = M * pieces;
    for (int e0 = tid; e0 < total_x; e0 += XU * WAVES * 64) {
      uint4 v[XU];
      int off[XU];
#pragma unroll
      for (int u = 0; u < XU; ++u) {
        const int e = e0 + u * WAVES * 64;
        off[u] = -1;
        if (e < total_x) {
          const int r = e / pieces, c = e - r * pieces;
          v[u] = *reinterpret_cast<const uint4*>(x + (int64_t)r * K + (int64_t)kbA * 64 + c * 8);
          off[u] = r * xstride + c * 16;
        }
      }
#pragma unroll
      for (int u = 0; u < XU; ++u)
        if (off[u] >= 0) *reinterpret_cast<uint4*>(xs + off[u]) = v[u];
    }
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this wave's x rows (and first weight batches) landed
  }
  __syncthreads();
  if (tr && lane == 0) tr[2] = __builtin_readcyclecounter();

  const unsigned char* xs0 = xs + min(n, M - 1) * xstride + kq * 32;
  const unsigned char* xs1 = xs + min(16 + n, M - 1) * xstride + kq * 32;
  // fragment (row n, piece c = 2 kq + j) sits at slot c ^ (n & 7) of row n
  const int arow = (n >> 3) * 1024 + (n & 7) * 128;
  const int a0off = arow + (((2 * kq) ^ (n & 7)) * 16), a1off = arow + (((2 * kq + 1) ^ (n & 7)) * 16);
  gf32x4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  int ct = 0, cb = 0;
  auto finish_slab = [&]() {
    int r = (gx + ct * G) * ROWS + wid * 16 + n;
    r = r < N ? r : N - 1;
    const uint16_t* wp = w + (int64_t)r * K + (int64_t)kbA * 64 + kq * 16;
    for (int kb = nb * UD; kb < nkb; ++kb) {                        // remainder K blocks: direct loads
      const gbf16x8_t a0 = __builtin_nontemporal_load(reinterpret_cast<const gbf16x8_t*>(wp + (int64_t)kb * 64));
      const gbf16x8_t a1 = __builtin_nontemporal_load(reinterpret_cast<const gbf16x8_t*>(wp + (int64_t)kb * 64 + 8));
      const int o = kb * 128;
      acc0 = E::mfma16(a0, *reinterpret_cast<const gbf16x8_t*>(xs0 + o), acc0);
      acc1 = E::mfma16(a0, *reinterpret_cast<const gbf16x8_t*>(xs1 + o), acc1);
      acc0 = E::mfma16(a1, *reinterpret_cast<const gbf16x8_t*>(xs0 + o + 16), acc0);
      acc1 = E::mfma16(a1, *reinterpret_cast<const gbf16x8_t*>(xs1 + o + 16), acc1);
    }
    // partial tile of the workgroup: [M][ROWS] fp32 through LDS, then full 512-byte rows to HBM
    // (per-wave 64-byte segments cost ~8 % of the kernel: half-line writes at a 4*N-byte stride)
    // The tile's LDS traffic is written as inline asm: to hipcc a pending LDS-DMA is a pending LDS write that may
    // alias ANY ds access it generates itself, so a compiler-visible ds_write / ds_read here waits vmcnt(0) first
    // and the prefetched weight batches of the next slab drain at every slab end.
    sgd_lds_barrier();                                              // previous slab's tile fully read
    {
      const uint32_t tp = otile_lds + (uint32_t)(wid * 16 + 4 * kq) * 4u;
      if (n < M) sgd_ds_write128(tp + (uint32_t)(n * OT_PITCH) * 4u, acc0);
      if (16 + n < M) sgd_ds_write128(tp + (uint32_t)((16 + n) * OT_PITCH) * 4u, acc1);
    }
    sgd_lds_barrier();
    {
      const int nblk = (gx + ct * G) * ROWS;
      constexpr int C4 = ROWS / 4;                                  // float4 columns per row
      for (int e = tid; e < M * C4; e += WAVES * 64) {
        const int m = e / C4, c4 = e - m * C4;
        if (nblk + c4 * 4 + 4 <= N) {
          const gf32x4_t v = sgd_ds_read128(otile_lds + (uint32_t)(m * OT_PITCH + c4 * 4) * 4u);
          *reinterpret_cast<float4*>(part + ((int64_t)by * M + m) * N + nblk + c4 * 4) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
    acc0 = (gf32x4_t){0, 0, 0, 0};
    acc1 = (gf32x4_t){0, 0, 0, 0};
    cb = 0;
    ++ct;
  };
  if (nb == 0) {
    for (int t = 0; t < nslab; ++t) finish_slab();
    return;
  }
  for (int j = 0; j < total; ++j) {
    if (j + SGD_SLOTS - 1 < total) issue();                         // refills the slot consumed at j - 1
    SgdWait<UD, SGD_SLOTS - 1>::go(total - 1 - j);                  // batches issued after batch j may stay in flight
    if (tr && lane == 0 && j == 0) tr[3] = __builtin_readcyclecounter();
    const unsigned char* slot = ring + (j % SGD_SLOTS) * BATCH_BYTES;
#pragma unroll
    for (int u = 0; u < UD; ++u) {
      const gbf16x8_t a0 = *reinterpret_cast<const gbf16x8_t*>(slot + u * 2048 + a0off);
      const gbf16x8_t a1 = *reinterpret_cast<const gbf16x8_t*>(slot + u * 2048 + a1off);
      const int o = (cb * UD + u) * 128;
      const gbf16x8_t b00 = *reinterpret_cast<const gbf16x8_t*>(xs0 + o);
      const gbf16x8_t b01 = *reinterpret_cast<const gbf16x8_t*>(xs0 + o + 16);
      const gbf16x8_t b10 = *reinterpret_cast<const gbf16x8_t*>(xs1 + o);
      const gbf16x8_t b11 = *reinterpret_cast<const gbf16x8_t*>(xs1 + o + 16);
      acc0 = E::mfma16(a0, b00, acc0);
      acc1 = E::mfma16(a0, b10, acc1);
      acc0 = E::mfma16(a1, b01, acc0);
      acc1 = E::mfma16(a1, b11, acc1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // ring reads retired before the slot is refilled
    if (++cb == nb) {
      if (tr && lane == 0 && ct == nslab - 1) tr[4] = __builtin_readcyclecounter();
      finish_slab();
    }
  }
  if (tr && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tr[5] = __builtin_readcyclecounter();
    if constexpr (PRO == 0) {
      tr[6] = nslab;
      tr[7] = nkb;
    }
  }
}

// Cost of one launch in slab-rounds x slab height (x K range, the same for all candidates of one split count): a
// workgroup walks ceil(slabs / column groups) slabs of 16 w rows.
static int sg_round_cost(const psg_ctx* ctx, int N, int splits, int w) {
  const int g = ctx->num_cu / splits > 0 ? ctx->num_cu / splits : 1;
  const int ns = (N + 16 * w - 1) / (16 * w);
  return ((ns + g - 1) / g) * w;
}
// LDS of the LDS-DMA variant: rings + partial tile + x slice
static size_t sg_dma_lds(int M, int K, int splits, int w, int ud, int sl) {
  const int KB = K >> 6, nkb_max = (KB + splits - 1) / splits;
  return (size_t)w * sl * ud * 2048 + (size_t)32 * (16 * w + 4) * 4 + (size_t)M * (nkb_max * 128 + 16);
}
static bool sg_wide_ok(const psg_ctx* ctx) {
  return ctx->opt.skinny_wide && ctx->opt.skinny_dma == 813 && ctx->opt.skinny_nt && ctx->opt.skinny_xdma;
}

// split count: K range per workgroup ~1024 elements (x slice <= 64 KiB of LDS at M = 32), and
// enough workgroups to give every CU several waves.
static int sg_plan(const psg_ctx* ctx, int M, int N, int K) {
  const int forced = ctx->opt.skinny_splits;
  const int KB = K >> 6;
  int S = forced > 0 ? forced : (K + 512) / 1024;
  if (S < 1) S = 1;
  if (forced <= 0 && S > 8) S = 8;       // K = 11008: 8 slices measured faster than 11 (fewer partial bytes)
  const int blocks_n = (N + SG_ROWS - 1) / SG_ROWS;
  if (forced <= 0)
    while (S < KB / 4 && (int64_t)blocks_n * S < 2 * ctx->num_cu) S *= 2;   // small N: split deeper
  const int balance = ctx->opt.skinny_balance;
  if (balance && forced <= 0 && 2 * S <= 8 && 2 * S <= KB / 4) {
    // balance: one workgroup per CU works through ceil(units / CUs) rounds of (slab, slice) units; if the last
    // round is mostly empty (q/k/v projection: 96 slabs x 4 slices = 1.5 rounds), twice the slices fill it
    const double r1 = (double)((N + 127) / 128) * S / ctx->num_cu, r2 = 2.0 * r1;   // 128-row slabs (8-wave DMA variant)
    const double e1 = r1 / (double)(int64_t)(r1 + 0.999999), e2 = r2 / (double)(int64_t)(r2 + 0.999999);
    if (e1 < 0.8 && e2 > e1 + 0.15) S *= 2;
  }
  if (S > KB) S = KB;
  if (S > PSG_MAX_SPLITS && forced <= 0) S = PSG_MAX_SPLITS;
  if (S < 1) S = 1;
  // LDS bound: M * (ceil(KB/S)*128 + 16) <= 96 KiB
  while ((int64_t)M * (((KB + S - 1) / S) * 128 + 16) > 96 * 1024 && S < KB) ++S;
  // Half the slices with 12-wave (192-row) slabs where the walk is as balanced (q/k/v: 96 slabs of 128 rows x 8
  // slices = 3 rounds -> 64 slabs of 192 rows x 4 slices = 1 round of the same bytes): half the fp32 partials to
  // write here and to read in the consumer, one slab epilogue instead of three.
  if (forced <= 0 && sg_wide_ok(ctx) && N >= 1024 && K >= 1024 && S >= 8 && S % 2 == 0) {
    const int S2 = S / 2;
    const int cur = sg_round_cost(ctx, N, S, 8) < sg_round_cost(ctx, N, S, 11) ? sg_round_cost(ctx, N, S, 8)
                                                                                : sg_round_cost(ctx, N, S, 11);
    // per-workgroup bytes ~ rounds x rows x K / S: compare cost / S
    if ((double)sg_round_cost(ctx, N, S2, 12) / S2 <= 1.001 * (double)cur / S && sg_dma_lds(M, K, S2, 12, 1, 3) <= 160 * 1024)
      S = S2;
  }
  return S;
}

// fp32 instantiation (psg_gemm_f32.hip)
int psg_sgf_plan(const psg_ctx* ctx, int M, int N, int K);
int psg_sgf_launch(psg_ctx* ctx, const void* x, const void* w, float* part, int M, int N, int K, int splits,
                   void* stream, bool w16);
bool psg_sgf_fits(const psg_ctx* ctx, int M, int K, int splits);

extern "C" int psg_skinny_gemm_plan(psg_ctx* ctx, int M, int N, int K, int dtype, int* splits) {
  PSG_REQUIRE(ctx && splits, PSG_ERR_INVALID, "psg_skinny_gemm_plan: NULL argument");
  PSG_REQUIRE(M >= 1 && M <= 32, PSG_ERR_UNSUPPORTED, "psg_skinny_gemm: M=%d (1..32 rows)", M);
  if (dtype == PSG_F32) {
    PSG_REQUIRE(N >= 16 && N % 16 == 0 && K >= 32 && K % 32 == 0, PSG_ERR_UNSUPPORTED,
                "psg_skinny_gemm(f32): N=%d must be a multiple of 16, K=%d a multiple of 32", N, K);
    *splits = psg_sgf_plan(ctx, M, N, K);
    PSG_REQUIRE(psg_sgf_fits(ctx, M, K, *splits), PSG_ERR_UNSUPPORTED,
                "psg_skinny_gemm(f32): %d rows of K=%d do not fit the LDS beside the weight rings at %d slices "
                "(K <= 11776 at 32 rows, <= 20480 at 20); use the library GEMM for this shape", M, K, *splits);
    return PSG_OK;
  }
  PSG_REQUIRE(N > 0 && N % 16 == 0 && K >= 64 && K % 64 == 0, PSG_ERR_UNSUPPORTED,
              "psg_skinny_gemm: N=%d must be a multiple of 16, K=%d a multiple of 64", N, K);
  *splits = sg_plan(ctx, M, N, K);
  return PSG_OK;
}

template <typename E>
static int sg_launch(psg_ctx* ctx, const void* x, const void* w, float* part, int M, int N, int K, int splits,
                     void* stream, const psg_prologue* pro = nullptr) {
  PSG_REQUIRE(ctx && x && w && part, PSG_ERR_INVALID, "psg_skinny_gemm: NULL argument");
  PSG_REQUIRE(M >= 1 && M <= 32, PSG_ERR_UNSUPPORTED, "psg_skinny_gemm: M=%d (1..32 rows)", M);
  PSG_REQUIRE(N > 0 && N % 16 == 0 && K >= 64 && K % 64 == 0, PSG_ERR_UNSUPPORTED,
              "psg_skinny_gemm: N=%d must be a multiple of 16, K=%d a multiple of 64", N, K);
  const int KB = K >> 6;
  PSG_REQUIRE(splits >= 1 && splits <= KB, PSG_ERR_INVALID, "psg_skinny_gemm: splits=%d (1..%d)", splits, KB);
  const int nkb_max = (KB + splits - 1) / splits;
  const int xstride = nkb_max * 128 + 16;
  const size_t lds = (size_t)M * xstride;
  PSG_REQUIRE(lds <= 160 * 1024, PSG_ERR_UNSUPPORTED, "psg_skinny_gemm: x slice needs %zu B of LDS; use more splits",
              lds);
  if (lds > ctx->skinny_lds_configured && lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)skinny_gemm_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) {
      psg_set_error("psg_skinny_gemm: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
      return PSG_ERR_HIP;
    }
    ctx->skinny_lds_configured = lds;
  }
  // persistent over row slabs: ~3 workgroups per CU in total, every workgroup keeps its x slice in LDS
  const int nslab = (N + SG_ROWS - 1) / SG_ROWS;
  const int wg_per_cu = ctx->opt.skinny_wg_per_cu < 1 ? 1 : ctx->opt.skinny_wg_per_cu;
  int G = (wg_per_cu * ctx->num_cu + splits - 1) / splits;
  if (G > nslab) G = nslab;
  if (G < 1) G = 1;
  const int dma_cfg = ctx->opt.skinny_dma;     // <waves><K blocks per batch><ring slots>; 0 = register variant
  if (dma_cfg > 0 && N >= 1024 && K >= 1024) {
    int wv = dma_cfg / 100;
    const int ud = (dma_cfg / 10) % 10, sl = dma_cfg % 10;
    // Slab height against round quantisation (option skinny_wide): a workgroup walks ceil(slabs / G) slabs and the
    // launch lasts as long as the workgroups with one slab more (one CU's DMA ring is latency-bound at about its
    // share of the HBM rate, so the last, partly empty round takes as long as a full one).  gate/up (N = 22016,
    // 4 slices): 172 slabs of 128 rows over 64 column groups = 2.69 -> 3 rounds; 126 slabs of 176 rows (11 waves)
    // = 1.97 -> 2 rounds of 1.375x the rows: 8 % less.  Same per-row arithmetic: bit-identical partials.
    if (sg_wide_ok(ctx) && !pro) {
      if ((double)sg_round_cost(ctx, N, splits, 11) < 0.95 * sg_round_cost(ctx, N, splits, wv) &&
          sg_dma_lds(M, K, splits, 11, 1, 3) <= 160 * 1024)
        wv = 11;
      if ((double)sg_round_cost(ctx, N, splits, 12) < 0.95 * sg_round_cost(ctx, N, splits, wv) &&
          sg_dma_lds(M, K, splits, 12, 1, 3) <= 160 * 1024)
        wv = 12;
    }
    const int rows = wv * 16;
    const int nslab_d = (N + rows - 1) / rows;
    const size_t ldsd = (size_t)wv * sl * ud * 2048 + (size_t)32 * (rows + 4) * 4 + lds;
    PSG_REQUIRE(ldsd <= 160 * 1024, PSG_ERR_UNSUPPORTED, "psg_skinny_gemm(dma): %zu B of LDS", ldsd);
    int per_cu = (int)((160 * 1024) / ldsd);              // workgroups per CU that LDS admits
    if (per_cu * wv > 16) per_cu = 16 / wv;               // and at most 16 waves per CU
    if (per_cu < 1) per_cu = 1;
    int Gd = (per_cu * ctx->num_cu + splits - 1) / splits;
    if (pro) Gd = (per_cu * ctx->num_cu) / splits;        // fused hand-off: the whole grid must be resident at once
    if (Gd > nslab_d) Gd = nslab_d;
    if (Gd < 1) Gd = 1;
    dim3 gridd(Gd, splits);
    hipStream_t st = (hipStream_t)stream;
    if (pro) {
      PSG_REQUIRE(wv == 8 && ud == 1 && sl == 3 && ctx->opt.skinny_nt && ctx->opt.skinny_xdma, PSG_ERR_UNSUPPORTED,
                  "psg_skinny_gemm_fused: needs the default skinny_dma=813 / nt / xdma kernel");
      PSG_REQUIRE(Gd * splits <= per_cu * ctx->num_cu && Gd * splits >= M, PSG_ERR_UNSUPPORTED,
                  "psg_skinny_gemm_fused: grid %d x %d cannot host the row operation (M=%d, %d CUs)", Gd, splits, M,
                  ctx->num_cu);
      const psg_prologue pv = *pro;
      const int64_t trace_nf = (int64_t)Gd * splits * 8 * 8;
      long long* trace_f = (ctx->trace_kind == PSG_TRACE_SKINNY_GEMM && ctx->trace_words >= trace_nf) ? ctx->trace : nullptr;
#define SGD_F(PRO)                                                                                                 \
  do {                                                                                                             \
    (void)hipFuncSetAttribute((const void*)skinny_gemm_dma_kernel<E, 8, 1, 3, 2, 1, PRO>,                             \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                             \
    skinny_gemm_dma_kernel<E, 8, 1, 3, 2, 1, PRO><<<gridd, 512, ldsd, st>>>((const uint16_t*)x, (const uint16_t*)w,   \
                                                                           part, M, N, K, xstride, trace_f, pv);   \
  } while (0)
      switch (pv.kind) {
        case PSG_PRO_RMSNORM: SGD_F(PSG_PRO_RMSNORM); break;
        default:
          psg_set_error("psg_skinny_gemm_fused: prologue kind %d not built", pv.kind);
          return PSG_ERR_UNSUPPORTED;
      }
#undef SGD_F
      PSG_CHECK_LAUNCH("psg_skinny_gemm_fused");
      return PSG_OK;
    }
    // per-wave stamps go to the caller's buffer (psg_set_trace_buffer) when it is large enough
    const int64_t trace_n = (int64_t)Gd * splits * wv * 8;
    long long* trace = (ctx->trace_kind == PSG_TRACE_SKINNY_GEMM && ctx->trace_words >= trace_n) ? ctx->trace : nullptr;
    // weights are read once by one CU: non-temporal (aux = 2) DMA loads
    const int nt = ctx->opt.skinny_nt, xdma = ctx->opt.skinny_xdma;
#define SGD_L(WV, UD, SL, AUX, XD)                                                                                 \
  do {                                                                                                             \
    (void)hipFuncSetAttribute((const void*)skinny_gemm_dma_kernel<E, WV, UD, SL, AUX, XD>,                            \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                             \
    skinny_gemm_dma_kernel<E, WV, UD, SL, AUX, XD><<<gridd, WV * 64, ldsd, st>>>(                                     \
        (const uint16_t*)x, (const uint16_t*)w, part, M, N, K, xstride, trace, psg_prologue{});                    \
  } while (0)
#define SGD(WV, UD, SL)                                                                                            \
  do {                                                                                                             \
    if (nt && xdma) SGD_L(WV, UD, SL, 2, 1);                                                                       \
    else if (nt) SGD_L(WV, UD, SL, 2, 0);                                                                          \
    else SGD_L(WV, UD, SL, 0, 0);                                                                                  \
  } while (0)
    if (wv == 8 && ud == 1 && sl == 3) SGD(8, 1, 3);
    else if (wv == 11 && ud == 1 && sl == 3) SGD_L(11, 1, 3, 2, 1);
    else if (wv == 12 && ud == 1 && sl == 3) SGD_L(12, 1, 3, 2, 1);
    else if (wv == 8 && ud == 1 && sl == 5) SGD(8, 1, 5);
    else if (wv == 8 && ud == 2 && sl == 3) SGD(8, 2, 3);
    else if (wv == 4 && ud == 1 && sl == 4) SGD(4, 1, 4);
    else if (wv == 4 && ud == 1 && sl == 6) SGD(4, 1, 6);
    else if (wv == 4 && ud == 2 && sl == 3) SGD(4, 2, 3);
    else if (wv == 4 && ud == 2 && sl == 4) SGD(4, 2, 4);
    else {
      psg_set_error("psg_skinny_gemm: unknown skinny_dma option %d", dma_cfg);
      return PSG_ERR_INVALID;
    }
#undef SGD
#undef SGD_L
    PSG_CHECK_LAUNCH("psg_skinny_gemm(dma)");
    return PSG_OK;
  }
  PSG_REQUIRE(!pro, PSG_ERR_UNSUPPORTED, "psg_skinny_gemm_fused: N=%d K=%d is below the LDS-DMA kernel's range", N, K);
  dim3 grid(G, splits);
  skinny_gemm_kernel<E><<<grid, 256, lds, (hipStream_t)stream>>>((const uint16_t*)x, (const uint16_t*)w, part, M, N, K,
                                                              xstride);
  PSG_CHECK_LAUNCH("psg_skinny_gemm");
  return PSG_OK;
}

extern "C" int psg_skinny_gemm(psg_ctx* ctx, const void* x, const void* w, float* part, int M, int N, int K,
                               int splits, int dtype, void* stream) {
  if (dtype == PSG_F32) return psg_sgf_launch(ctx, x, w, part, M, N, K, splits, stream, false);
  PSG_DISPATCH_E16(dtype, "psg_skinny_gemm", return sg_launch<E>(ctx, x, w, part, M, N, K, splits, stream));
}

// fp32 activations x weights STORED as fp16 (a frozen fp16 checkpoint the reference upcasts on load, V4:99-100): the
// arithmetic of psg_skinny_gemm(PSG_F32) on the widened weights, bit for bit, at half the weight bytes (psg_gemm_f32.hip)
extern "C" int psg_skinny_gemm_w16(psg_ctx* ctx, const float* x, const void* w_f16, float* part, int M, int N, int K,
                                   int splits, void* stream) {
  return psg_sgf_launch(ctx, x, w_f16, part, M, N, K, splits, stream, true);
}

extern "C" int psg_skinny_gemm_fused(psg_ctx* ctx, const psg_prologue* pro, void* x, const void* w, float* part, int M,
                                     int N, int K, int splits, int dtype, void* stream) {
  PSG_REQUIRE(ctx && pro && pro->sync, PSG_ERR_INVALID, "psg_skinny_gemm_fused: NULL argument");
  if (pro->kind == PSG_PRO_RMSNORM) {
    PSG_REQUIRE(pro->resid && pro->norm_w, PSG_ERR_INVALID, "psg_skinny_gemm_fused(rmsnorm): NULL resid / norm_w");
    PSG_REQUIRE(K == 4096 || (K <= 1024 && K % 4 == 0), PSG_ERR_UNSUPPORTED,
                "psg_skinny_gemm_fused(rmsnorm): hidden=%d (built for 4096 and <= 1024)", K);
    PSG_REQUIRE(pro->in_splits >= 0 && pro->in_splits <= PSG_MAX_SPLITS, PSG_ERR_INVALID,
                "psg_skinny_gemm_fused: in_splits=%d", pro->in_splits);
  }
  PSG_DISPATCH_E16(dtype, "psg_skinny_gemm_fused", return sg_launch<E>(ctx, x, w, part, M, N, K, splits, stream, pro));
}

// y[i] = sum_s part[s][i] in split order, converted to the activation dtype (tests, generic consumers)
template <typename T>
__global__ void reduce_partials_kernel(const float* __restrict__ part, int S, int64_t n4, T* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 a = *reinterpret_cast<const float4*>(part + i * 4);
    for (int s = 1; s < S; ++s) {
      const float4 b = *reinterpret_cast<const float4*>(part + ((int64_t)s * n4 + i) * 4);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float v[4] = {a.x, a.y, a.z, a.w};
    Act<T>::st4(y, i * 4, v);
  }
}

extern "C" int psg_reduce_partials(psg_ctx* ctx, const float* part, int splits, int64_t n, void* y, int dtype,
                                   void* stream) {
  PSG_REQUIRE(ctx && part && y, PSG_ERR_INVALID, "psg_reduce_partials: NULL argument");
  PSG_REQUIRE(splits >= 1 && n > 0 && n % 4 == 0, PSG_ERR_INVALID, "psg_reduce_partials: splits=%d n=%lld", splits,
              (long long)n);
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  PSG_DISPATCH_DTYPE(dtype, "psg_reduce_partials",
                     (reduce_partials_kernel<T><<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(part, splits, n / 4,
                                                                                                 (T*)y)));
  PSG_CHECK_LAUNCH("psg_reduce_partials");
  return PSG_OK;
}
