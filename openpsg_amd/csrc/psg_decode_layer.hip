// One Llama decoder layer of the decode step (M <= 32 rows, fp32 weights = the reference's precision, V4:99-100) as ONE
// persistent launch: RMSNorm -> q|k|v projection -> rotary + KV append + attention -> o projection -> RMSNorm ->
// gate|up projection -> SwiGLU -> down projection (HF-LL:53-67, 130-214, 243-281 via V4:293-312), bit-identical to the
// chain of eight launches it replaces (psg_rmsnorm / psg_skinny_gemm / psg_decode_attn / psg_silu_mul).
//
// Why: the chain's four weight-streaming launches each spend ~6 us filling and draining their DMA rings around the
// 5-57 us their bytes need, and the four row kernels between them are ~5-15 us of dependent round trips during which
// HBM idles - 192 us per layer for 809 MB = 4.2 TB/s, against 6.3 TB/s achievable.  Here 256 workgroups (one per CU)
// stay resident for the whole layer; a wave's weight ring is refilled across every phase boundary (the next
// projection's weights depend on nothing), so the row operations and hand-offs run under the stream.
//
// Structure (MI355X_MICROARCH.md "Persistent kernels"; cdna_hip_programming.md Guideline 16):
//   * workgroup b = (gx, by): K slice `by` of S, column group `gx` of 256 / S, exactly the (slab, slice) arithmetic of
//     skinny_gemm_f32_kernel (psg_gemm_f32.hip): per-wave LDS-DMA rings of non-temporal 128-byte lines, the x slice in
//     LDS, v_mfma_f32_16x16x1_4b + v_mfma_f32_4x4x1_16b, fp32 split-K partials summed in slice order by their reader;
//   * every value another workgroup reads is stored WRITE-THROUGH (8-byte relaxed agent-scope atomic stores =
//     global_store_dwordx2 sc1) and read with sc1 loads / sc1 LDS-DMA; a producer drains its stores (vmcnt(0)), the
//     workgroup meets at a barrier, ONE lane adds to an arrival counter; a consumer's wave 0 polls the counters it needs
//     (relaxed, bounded, s_sleep) and releases the workgroup through a barrier.  No fences, no grid-wide barrier: each
//     edge waits for exactly the workgroups that produce its bytes:
//       RMSNorm  : column owner b (16 columns) sums the split-K slices + residual, publishes the residual and the
//                  quad-level partial sums of squares; EVERY workgroup then rebuilds the row statistics in the reduction
//                  tree of rmsnorm_kernel<float, 1, float> (1024 threads) and normalises its own x slice into LDS;
//       attention: head h = column group gx of the q|k|v projection (slabs h, 32 + h, 64 + h), so a head's 8 producers
//                  count into one word; the units (row, head) are dealt to all workgroups, four waves each, in the
//                  arithmetic of decode_attn4_kernel<float>;
//       o proj   : K slice by = heads 4 by .. 4 by + 3: waits for the units of those heads;
//       SwiGLU   : (row, 128-column block) items dealt to waves, each waits for the two column groups that produced its
//                  gate and up blocks; the down projection waits for the blocks of its K slice.
//   * all counters live in a caller-provided, zeroed int32 array (one per launch: a memset node replayed first);
//     every poll is bounded and reports through cnt[PSG_DL_TIMEOUT].
// Shapes: hidden = 4096 = 32 heads x 128, inter % 128 == 0, 256 compute units (Llama-2-7B on MI355X); anything else
// keeps the launch chain (the host wrapper checks).  One persistent launch at a time per device: two of them could each
// hold CUs the other's missing workgroups need (the engine uses it for `forward`, not for the in-flight slots).
#include <stdlib.h>

#include "psg_common.h"
#include "psg_decode_math.h"

#define PSG_DL_WG 256
#define PSG_DL_WAVES 12      // waves per workgroup; a projection streams with W <= 12 of them (its slab = 16 W rows)
// Arrival counters: every counter has a 256-byte slot of its own (PSG_DL_SLOT words apart) - counters sharing a cache
// line share one memory channel, which serialises its atomics and polls at ~90 per us (1720 SwiGLU arrivals on three
// lines cost 20 us).  Slot numbers:
#define PSG_DL_CNT_X1 0        // 8 (b % 8: 32 arrivals each) + 1 top (8 arrivals) + 8 go flags: layer-input columns reduced
#define PSG_DL_CNT_HEAD 17     // 32: q|k|v column group gx done for K slice by, 8 arrivals each
#define PSG_DL_CNT_ATT 49      // 8: attention units of heads 4 g .. 4 g + 3 stored, 4 * M arrivals each
#define PSG_DL_CNT_OSLAB 57    // 32: o-projection slab done for K slice by, 8 arrivals each
#define PSG_DL_CNT_X2 89       // 8 + 1 + 8 as X1: post-attention columns reduced
#define PSG_DL_CNT_GU 106      // 32: gate|up column group gx done for K slice by, 8 arrivals each
#define PSG_DL_CNT_H 138       // inter / 128 slots: SwiGLU block j stored, M arrivals each
#define PSG_DL_CNT_DGRP 236    // 16: down-projection column group gx done for K slice by, 16 arrivals each (layer chaining)
#define PSG_DL_TIMEOUT 255
#define PSG_DL_SLOT 64         // words per slot
#define PSG_DL_NCNT (256 * PSG_DL_SLOT)

typedef float df32x4_t __attribute__((ext_vector_type(4)));
typedef float df32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned long long du64;

struct psg_dl_layer {          // one decoder layer's tensors (psg_decode_layers: an array of these in device memory)
  const float* ln1;
  const float* ln2;
  const float* wqkv;
  const float* wo;
  const float* wgu;
  const float* wdown;
  float* kc;
  float* vc;
};

struct psg_dl_args {
  const psg_dl_layer* table;  // n_layers entries, device memory
  int n_layers;
  float* down_part2;          // the other half of the double-buffered down partials (layer l writes buffer l & 1)
  float* resid;               // [M][D] residual stream, updated in place
  const float* delta;         // previous layer's down partials [dsplits][M][D] (NULL: none)
  int dsplits;
  const int32_t* tok_pair;
  const int32_t* tok_pos;
  const float* cos_tab;
  const float* sin_tab;
  float* qkv_part;            // [8][M][3 D]
  float* att;                 // [M][D]
  float* o_part;              // [8][M][D]
  float* ssq;                 // [2][256][32] quad sums of squares of the two RMSNorms
  float* gu_part;             // [8][M][2 I]
  float* h;                   // [M][I]
  float* down_part;           // [16][M][D]   (output: the next layer's delta)
  unsigned* cnt;              // [PSG_DL_NCNT], zeroed
  long long* trace;           // psg_set_trace_buffer(PSG_TRACE_DECODE_LAYER): 24 wall-clock stamps per workgroup, or NULL
  int M, D, I, heads, ctx;
  float eps;
};

// Thread / workgroup index through an opaque (empty) asm statement: everything derived from them is recomputed where it is
// used instead of being hoisted out of the layer loop - hoisted, the per-phase address arithmetic of four projections and
// five row phases stays live across the whole loop body and spills INSIDE the weight streams (measured: 188 -> 294 us).
__device__ __forceinline__ int dl_tid() {
  int t = (int)threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}
__device__ __forceinline__ int dl_bid() {
  int t = (int)blockIdx.x;
  asm volatile("" : "+s"(t));
  return t;
}

// ---- write-through stores / sc1 loads --------------------------------------------------------------------------------
__device__ __forceinline__ void dl_st2(float* p, float a, float b) {              // 8-byte aligned
  const du64 w = (du64)__float_as_uint(a) | ((du64)__float_as_uint(b) << 32);
  __hip_atomic_store(reinterpret_cast<du64*>(p), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void dl_st1(float* p, float a) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 dl_ld2(const float* p) {
  const du64 w = __hip_atomic_load(reinterpret_cast<const du64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float2(__uint_as_float((unsigned)w), __uint_as_float((unsigned)(w >> 32)));
}
__device__ __forceinline__ float dl_ld1(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void dl_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void dl_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// publish: every wave has drained its stores (dl_drain) BEFORE it comes here; ONE lane counts the workgroup in
__device__ __forceinline__ void dl_publish(unsigned* cnt, int slot, unsigned n = 1u) {
  dl_barrier();
  if (dl_tid() == 0) __hip_atomic_fetch_add(cnt + slot * PSG_DL_SLOT, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool dl_poll(unsigned* cnt, int slot, unsigned want) {
  return __hip_atomic_load(cnt + slot * PSG_DL_SLOT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want;
}
__device__ __forceinline__ void dl_timeout(unsigned* cnt, unsigned code) {
  __hip_atomic_store(cnt + PSG_DL_TIMEOUT * PSG_DL_SLOT, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wave 0 polls `nw` consecutive slots (nw <= 64) until each is >= want; the workgroup is released through a barrier
__device__ __forceinline__ void dl_wait(unsigned* cnt, int first, int nw, unsigned want) {
  if (dl_tid() < 64) {
    const int lane = dl_tid();
    unsigned spins = 0;
    for (;;) {
      bool ok = true;
      if (lane < nw) ok = dl_poll(cnt, first + lane, want);
      if (__all(ok)) break;
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 21)) {                                    // a producer died: report, do not hang the GPU
        if (lane == 0) dl_timeout(cnt, 0x5047u + (unsigned)first);
        break;
      }
    }
  }
  dl_barrier();
}
// All-to-all edge (every workgroup waits for every workgroup): XCD-hierarchical (MI355X_MICROARCH.md barrier-xcd) - the
// workgroups of a b % 8 class count into their own slot, the last of a class counts into the top slot, the last of those
// raises the eight go flags, and a workgroup polls the flag of ITS class only (32 pollers per line instead of 256).
__device__ __forceinline__ void dl_arrive_all(unsigned* cnt, int base) {       // stores drained by the caller
  dl_barrier();
  if (dl_tid() == 0) {
    const int cls = dl_bid() & 7;
    if (__hip_atomic_fetch_add(cnt + (base + cls) * PSG_DL_SLOT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == PSG_DL_WG / 8 - 1)
      if (__hip_atomic_fetch_add(cnt + (base + 8) * PSG_DL_SLOT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 7u)
        for (int x = 0; x < 8; ++x)
          __hip_atomic_store(cnt + (base + 9 + x) * PSG_DL_SLOT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ void dl_wait_all(unsigned* cnt, int base) { dl_wait(cnt, base + 9 + (dl_bid() & 7), 1, 1u); }

// sum over the four 16-lane rows (psg_gemm_f32.hip: sgf_sum_kq)
__device__ __forceinline__ float dl_sum_kq(float v) {
  float a = v, b = v;
  asm volatile("s_nop 7\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  const float s = a + b;
  float c = s, d = s;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(d));
  return c + d;
}
__device__ __forceinline__ void dl_ds_write128(uint32_t lds_addr, df32x4_t v) {
  asm volatile("s_nop 15\n\tds_write_b128 %0, %1" ::"v"(lds_addr), "v"(v) : "memory");
}
__device__ __forceinline__ df32x4_t dl_ds_read128(uint32_t lds_addr) {
  df32x4_t v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr) : "memory");
  return v;
}
template <int N_>
__device__ __forceinline__ void dl_vmwait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}
template <int MAXN>
struct DlWait {
  static __device__ __forceinline__ void go(int newer) {
    if (newer >= MAXN) dl_vmwait<(MAXN * 2 < 63 ? MAXN * 2 : 63)>();
    else DlWait<MAXN - 1>::go(newer);
  }
};
template <>
struct DlWait<0> {
  static __device__ __forceinline__ void go(int) { dl_vmwait<0>(); }
};

#define DL_XPAD 16
#define DL_BLOCK 2048                                                   // 16 rows x 128 B of one wave's ring slot

// One weight-streaming projection of this workgroup: the (slab, K slice) walk of skinny_gemm_f32_kernel with W-wave slabs
// (the chain's own choices: 12 waves for q|k|v, 11 for gate|up - an even number of slab rounds per column group -, 8 else).
// Waves >= W take no part in the stream; they still meet the two barriers of every slab end and help store its tile.
template <int W, int SLOTS, int G16, int G4>
struct DlGemm {
  static constexpr int ROWS = W * 16;
  static constexpr int OT_PITCH = ROWS + 4;
  static constexpr int MP = G16 * 16 + G4 * 4;
  static constexpr int NG4 = G4 > 0 ? G4 : 1, NG16 = G16 > 0 ? G16 : 1;
  const unsigned char* wb;
  int64_t row_bytes;
  int N, M, G, gx, by, kbA, nkb, nslab, total, xstride;
  unsigned char* ring;
  int lt, lb, ls;
  const unsigned char* src;

  __device__ __forceinline__ const unsigned char* dma_src(int t) const {
    const int lane = dl_tid() & 63, wid = dl_tid() >> 6;
    const int dr = lane >> 3, dp = (lane & 7) ^ (lane >> 3);
    int r = (gx + t * G) * ROWS + (wid < W ? wid : 0) * 16 + dr;
    r = r + 8 < N ? r : (N - 9 > 0 ? N - 9 : 0);
    return wb + (int64_t)r * row_bytes + (int64_t)kbA * 128 + dp * 16;
  }
  __device__ __forceinline__ void setup(const float* w, int N_, int K, int S, int by_, int gx_, int M_, unsigned char* smem) {
    const int wid = dl_tid() >> 6;
    wb = reinterpret_cast<const unsigned char*>(w);
    row_bytes = (int64_t)K * 4;
    N = N_; M = M_; G = PSG_DL_WG / S; gx = gx_; by = by_;
    const int KB = K >> 5;
    kbA = (int)((unsigned)(KB * by) / (unsigned)S);
    const int kbB = (int)((unsigned)(KB * (by + 1)) / (unsigned)S);
    nkb = kbB - kbA;
    const int nslab_all = (N + ROWS - 1) / ROWS;
    nslab = gx < nslab_all ? (nslab_all - gx + G - 1) / G : 0;
    total = nslab * nkb;
    xstride = ((KB + S - 1) / S) * 128 + DL_XPAD;
    ring = smem + wid * (SLOTS * DL_BLOCK);
    lt = lb = ls = 0;
    src = dma_src(0);
  }
  __device__ __forceinline__ void issue() {
    unsigned char* dst = ring + ls * DL_BLOCK;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 2);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 8 * row_bytes),
                                     (__attribute__((address_space(3))) void*)(dst + 1024), 16, 0, 2);
    src += 128;
    if (++ls == SLOTS) ls = 0;
    if (++lb == nkb) { lb = 0; ++lt; src = dma_src(lt); }
  }
  // the first blocks of the stream: issued as soon as the rings are free, long before x exists
  __device__ __forceinline__ void prefetch() {
    if ((int)(dl_tid() >> 6) >= W) return;
    for (int i = 0; i < SLOTS - 1; ++i)
      if (i < total) issue();
  }
  // the load cursor behind `prefetch`, without issuing anything: lets the caller drop this object across a row phase
  // (its registers are needed there) and rebuild it with setup() + skip_prefetched()
  __device__ __forceinline__ void skip_prefetched() {
    for (int i = 0; i < SLOTS - 1; ++i)
      if (i < total) {
        src += 128;
        if (++ls == SLOTS) ls = 0;
        if (++lb == nkb) { lb = 0; ++lt; src = dma_src(lt); }
      }
  }
  // x slice [M][K slice] from global (written through by other workgroups) by sc1 LDS-DMA; caller waits + barriers
  __device__ __forceinline__ void stage_x_dma(const float* x, int64_t x_row_floats, unsigned char* xs) const {
    const int lane = dl_tid() & 63, wid = dl_tid() >> 6;
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(x);
    const int pieces = nkb * 8;
    const int cpr = (pieces + 63) >> 6;
    const int items = M * cpr;
    for (int it = wid; it < items; it += PSG_DL_WAVES) {
      const int r = it / cpr, j = it - r * cpr;
      const int c = j * 64 + lane;
      if (c < pieces)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(xb + (int64_t)r * x_row_floats * 4 + (int64_t)kbA * 128 + c * 16),
            (__attribute__((address_space(3))) void*)(xs + r * xstride + j * 1024), 16, 0, 16);
    }
  }
  // the stream; x is in LDS (all waves past a barrier).  part[by][M][N] written through.
  __device__ __forceinline__ void run(float* __restrict__ part, unsigned char* smem, const unsigned char* xs) {
    const int tid = dl_tid(), lane = tid & 63, wid = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const uint32_t otile_lds =
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(smem + PSG_DL_WAVES * SLOTS * DL_BLOCK);
    const unsigned char* x16[NG16];
    const unsigned char* x4[NG4];
#pragma unroll
    for (int g = 0; g < G16; ++g) x16[g] = xs + min(16 * g + n, M - 1) * xstride + kq * 32;
#pragma unroll
    for (int q = 0; q < G4; ++q) x4[q] = xs + min(16 * G16 + 4 * q + (lane & 3), M - 1) * xstride + kq * 32;
    const int arow = (n >> 3) * 1024 + (n & 7) * 128;
    const int a0off = arow + (((2 * kq) ^ (n & 7)) * 16), a1off = arow + (((2 * kq + 1) ^ (n & 7)) * 16);
    const df32x4_t zero4 = {0, 0, 0, 0};
    const df32x16_t zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    df32x16_t acc16[NG16];
    df32x4_t acc4[NG4];
#pragma unroll
    for (int g = 0; g < G16; ++g) acc16[g] = zero16;
#pragma unroll
    for (int q = 0; q < G4; ++q) acc4[q] = zero4;
    int ct = 0, cb = 0, cs = 0;
    const bool streams = wid < W;
    auto finish_slab = [&]() {
      if (streams) {
#pragma unroll
        for (int q = 0; q < G4; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc4[q][r] = dl_sum_kq(acc4[q][r]);
      }
      dl_barrier();                                                   // previous slab's tile fully read
      if (streams) {
        const uint32_t tp16 = otile_lds + (uint32_t)(n * OT_PITCH + wid * 16 + 4 * kq) * 4u;
#pragma unroll
        for (int g = 0; g < G16; ++g) {
          df32x4_t v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (acc16[g][r] + acc16[g][4 + r]) + (acc16[g][8 + r] + acc16[g][12 + r]);
          dl_ds_write128(tp16 + (uint32_t)(16 * g * OT_PITCH) * 4u, v);
        }
        const int g4 = (lane >> 2) & 3, j = lane & 3;
        const uint32_t tp4 = otile_lds + (uint32_t)((16 * G16 + j) * OT_PITCH + wid * 16 + 4 * g4) * 4u;
#pragma unroll
        for (int q = 0; q < G4; ++q)
          if (kq == (q & 3)) dl_ds_write128(tp4 + (uint32_t)(4 * q * OT_PITCH) * 4u, acc4[q]);
      }
      dl_barrier();
      {
        const int nblk = (gx + ct * G) * ROWS;
        constexpr int C4 = ROWS / 4;
        for (int e = tid; e < M * C4; e += PSG_DL_WAVES * 64) {
          const int m = e / C4, c4 = e - m * C4;
          if (nblk + c4 * 4 + 4 <= N) {
            const df32x4_t v = dl_ds_read128(otile_lds + (uint32_t)(m * OT_PITCH + c4 * 4) * 4u);
            float* dst = part + ((int64_t)by * M + m) * N + nblk + c4 * 4;
            dl_st2(dst, v[0], v[1]);
            dl_st2(dst + 2, v[2], v[3]);
          }
        }
      }
#pragma unroll
      for (int g = 0; g < G16; ++g) acc16[g] = zero16;
#pragma unroll
      for (int q = 0; q < G4; ++q) acc4[q] = zero4;
      cb = 0;
      ++ct;
    };
    if (!streams) {                                                   // idle in this projection: the slab ends only
      for (int t = 0; t < nslab; ++t) finish_slab();
      return;
    }
    if (nkb == 0) {
      for (int t = 0; t < nslab; ++t) finish_slab();
      return;
    }
    for (int j = 0; j < total; ++j) {
      if (j + SLOTS - 1 < total) issue();
      DlWait<SLOTS - 1>::go(total - 1 - j);
      const unsigned char* slot = ring + cs * DL_BLOCK;
      if (++cs == SLOTS) cs = 0;
      const int o = cb * 128;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const df32x4_t a = *reinterpret_cast<const df32x4_t*>(slot + (h ? a1off : a0off));
        df32x4_t b16[NG16], b4[NG4];
#pragma unroll
        for (int g = 0; g < G16; ++g) b16[g] = *reinterpret_cast<const df32x4_t*>(x16[g] + o + 16 * h);
#pragma unroll
        for (int q = 0; q < G4; ++q) b4[q] = *reinterpret_cast<const df32x4_t*>(x4[q] + o + 16 * h);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int g = 0; g < G16; ++g) acc16[g] = __builtin_amdgcn_mfma_f32_16x16x1f32(a[i], b16[g][i], acc16[g], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < G4; ++q) acc4[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[i], b4[q][i], acc4[q], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (++cb == nkb) finish_slab();
    }
  }
};

// ---- RMSNorm, producer side: workgroup b owns columns [16 b, 16 b + 16) -----------------------------------------------
// thread (m, q) = chain thread 4 b + q of row m (rmsnorm_kernel<float, 1, float> with 1024 threads: 4 columns each)
__device__ __forceinline__ void dl_norm_owner(const psg_dl_args& a, const float* delta, int dsplits, float* ssq_out, int b) {
  const int tid = dl_tid();
  const int m = tid >> 2, q = tid & 3;
  const int D = a.D;
  float ss = 0.f;
  if (m < a.M) {
    const int col = 16 * b + 4 * q;
    const int64_t i = (int64_t)m * D + col;
    const float2 r0 = dl_ld2(a.resid + i), r1 = dl_ld2(a.resid + i + 2);
    float v[4] = {r0.x, r0.y, r1.x, r1.y};
    if (delta) {
      float d[4];
      const int64_t slice = (int64_t)a.M * D;
      float2 t0[PSG_MAX_SPLITS], t1[PSG_MAX_SPLITS];
#pragma unroll
      for (int s = 0; s < PSG_MAX_SPLITS; ++s)
        if (s < dsplits) {
          t0[s] = dl_ld2(delta + (int64_t)s * slice + i);
          t1[s] = dl_ld2(delta + (int64_t)s * slice + i + 2);
        }
      d[0] = t0[0].x; d[1] = t0[0].y; d[2] = t1[0].x; d[3] = t1[0].y;
#pragma unroll
      for (int s = 1; s < PSG_MAX_SPLITS; ++s)
        if (s < dsplits) { d[0] += t0[s].x; d[1] += t0[s].y; d[2] += t1[s].x; d[3] += t1[s].y; }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += d[e];
      dl_st2(a.resid + i, v[0], v[1]);
      dl_st2(a.resid + i + 2, v[2], v[3]);
    }
    ss = psg_sumsq4(v, ss);
  }
  // the first two levels of wave_sum's tree (row_shr:1, row_shr:2): lane 4 k + 3 = (t3 + t2) + (t1 + t0)
  ss += psg_dpp<0x111, 0xf>(0.f, ss);
  ss += psg_dpp<0x112, 0xf>(0.f, ss);
  if (m < a.M && q == 3) dl_st1(ssq_out + m * PSG_DL_WG + b, ss);       // [row][owner]
}

// ---- RMSNorm, consumer side: row statistics from the 256 quad sums, then this workgroup's x slice into LDS -----------
// The slice's residual values are requested first (they do not depend on the statistics): one round trip for both.
__device__ __forceinline__ void dl_norm_stage(const psg_dl_args& a, const float* ssq, const float* gamma, int kbA, int nkb,
                                              int xstride, unsigned char* xs, float* s_inv) {
  const int tid = dl_tid(), lane = tid & 63, wid = tid >> 6;
  constexpr int NT = PSG_DL_WAVES * 64;
  constexpr int MAXE = 8;                                              // float2 pairs per thread: 24 rows x 256 pairs / 768
  const int c0 = kbA * 32;
  const int pairs = nkb * 16;                                         // float2 columns of the slice
  const int total = a.M * pairs;
  float2 v[MAXE], g[MAXE];
  int off[MAXE], row[MAXE];
#pragma unroll
  for (int k = 0; k < MAXE; ++k) {
    const int e = tid + k * NT;
    off[k] = -1;
    if (e < total) {
      const int m = e / pairs, c = (e - m * pairs) * 2;
      v[k] = dl_ld2(a.resid + (int64_t)m * a.D + c0 + c);
      g[k] = *reinterpret_cast<const float2*>(gamma + c0 + c);
      off[k] = m * xstride + c * 4;
      row[k] = m;
    }
  }
  for (int m = wid; m < a.M; m += PSG_DL_WAVES) {
    // lane L: quads 4 L .. 4 L + 3 = one 16-lane row of the chain's wave L / 4 (rmsnorm_kernel, 1024 threads)
    const float2 qa = dl_ld2(ssq + m * PSG_DL_WG + 4 * lane), qb = dl_ld2(ssq + m * PSG_DL_WG + 4 * lane + 2);
    float r = (qb.y + qb.x) + (qa.y + qa.x);                          // row_shr:4, row_shr:8
    r += psg_dpp<0x111, 0xf>(0.f, r);                                 // row_bcast:15 / row_bcast:31 have the quad shape too:
    r += psg_dpp<0x112, 0xf>(0.f, r);                                 // lane 4 w + 3 = (R3 + R2) + (R1 + R0) = chain wave w
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 4 * w + 3));
    if (lane == 0) s_inv[m] = 1.0f / sqrtf(tot / (float)a.D + a.eps);
  }
  dl_barrier();
#pragma unroll
  for (int k = 0; k < MAXE; ++k)
    if (off[k] >= 0) {
      const float inv = s_inv[row[k]];
      *reinterpret_cast<float2*>(xs + off[k]) = make_float2(g[k].x * (v[k].x * inv), g[k].y * (v[k].y * inv));
    }
}

// ---- attention unit (row, head) by four waves: psg_decode_attn4_unit<float>, the arithmetic of decode_attn4_kernel ------
// The unit's first wave waits for the column groups that produced its q, k and v columns INSIDE the loader, i.e. after
// the cached keys and values were requested (their addresses need `pos` only): the cache round trip runs under the wait.
struct DlAttnArgs {
  const float* qkv_part;
  float* att;
  float* kc;
  float* vc;
  const int32_t* tok_pair;
  const int32_t* tok_pos;
  const float* cos_tab;
  const float* sin_tab;
  unsigned* cnt;
  int M, D, heads, ctx;
};
__device__ __forceinline__ void dl_attn_round(const DlAttnArgs a, int unit, PsgDecodeAttnScratch* sc) {
  const int heads = a.heads, hidden = a.D;
  const bool valid = unit >= 0;                                     // false: this group of four waves only meets the barriers
  const int row = valid ? unit / heads : 0, h = valid ? unit % heads : 0;
  const int pos = valid ? a.tok_pos[row] : -1;
  const int64_t sl = (int64_t)a.M * 3 * hidden;
  const float* qp = a.qkv_part;
  unsigned* cnt = a.cnt;
  auto ld = [&](const int64_t (&idx)[6], float (&x)[6]) {           // ldn_splits<float, 6> with sc1 loads: slices in order
    {
      // q, k, v columns (h 128 .., D + h 128 .., 2 D + h 128 ..) lie in the 192-row slabs col / 192 (two where 128 columns
      // straddle a slab end), produced by the column groups slab % 32: lanes 0..5 poll one group each
      const int lane = dl_tid() & 63;
      const int col = (lane >> 1) * hidden + h * 128 + ((lane & 1) ? 127 : 0);
      const int grp = lane < 6 ? (col / 192) & 31 : 0;
      unsigned spins = 0;
      for (;;) {
        const bool ok = lane >= 6 || dl_poll(cnt, PSG_DL_CNT_HEAD + grp, 8u);
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 21)) {
          if (lane == 0) dl_timeout(cnt, 0x5247u);
          break;
        }
      }
    }
    float tt[4][6];                                                  // two batches of four slices (register budget)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int e = 0; e < 6; ++e) tt[s][e] = dl_ld1(qp + (int64_t)s * sl + idx[e]);
#pragma unroll
    for (int e = 0; e < 6; ++e) x[e] = tt[0][e];
#pragma unroll
    for (int s = 1; s < 4; ++s)
#pragma unroll
      for (int e = 0; e < 6; ++e) x[e] += tt[s][e];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int e = 0; e < 6; ++e) tt[s][e] = dl_ld1(qp + (int64_t)(s + 4) * sl + idx[e]);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int e = 0; e < 6; ++e) x[e] += tt[s][e];
  };
  float* att = a.att;
  auto st = [&](int64_t i, float v) { dl_st1(att + i, v); };
  psg_decode_attn4_unit<float>(pos >= 0, (int)(dl_tid() & 255), row, h, pos, pos >= 0 ? a.tok_pair[row] : 0, heads, a.ctx,
                               a.cos_tab, a.sin_tab, a.kc, a.vc, ld, st, sc);
  __syncthreads();                                                  // scratch free for the next round
}

template <int SLOTS, int G16, int G4>
__global__ void __launch_bounds__(PSG_DL_WAVES * 64) decode_layer_f32_kernel(const psg_dl_args a0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using GemmQ = DlGemm<12, SLOTS, G16, G4>;                           // q|k|v: 192-row slabs, 64 of them = 2 rounds of 32 groups
  using GemmO = DlGemm<8, SLOTS, G16, G4>;                            // o, down: 128-row slabs
  using GemmG = DlGemm<11, SLOTS, G16, G4>;                           // gate|up: 176-row slabs, 126 of them = 4 rounds (3.94)
  constexpr int MP = G16 * 16 + G4 * 4;
  const int b = dl_bid(), tid = dl_tid(), lane = tid & 63, wid = tid >> 6;
  const int M = a0.M, D = a0.D, I = a0.I;
  // LDS: rings | partial tile (sized for 12-wave slabs) | x slice
  unsigned char* const xs = smem + PSG_DL_WAVES * SLOTS * DL_BLOCK + MP * (PSG_DL_WAVES * 16 + 4) * 4;
  float* const s_inv = reinterpret_cast<float*>(smem + PSG_DL_WAVES * SLOTS * DL_BLOCK);   // the tile area, free between phases
  const int nl = a0.n_layers;
  // Layers are chained inside the launch (psg_decode_layers): layer l's down projection leaves its partials in buffer
  // l & 1; the owners of layer l + 1 wait for the column group that produced their slab (16 K slices), and every
  // workgroup requests the next layer's first q|k|v blocks as soon as its own down stream has ended.  The workspace
  // buffers are reused layer after layer: each is rewritten only behind an all-to-all edge of the NEXT layer, which
  // every reader of the previous layer has passed.
  for (int l = 0; l < nl; ++l) {
  // per-layer tensors are read from the table where they are used (a modified copy of the argument block would keep
  // ~40 scalar registers alive across every phase: spills inside the streams)
  const psg_dl_args& a = a0;
  const psg_dl_layer* const T = a0.table + l;
  float* const down_part = (l & 1) ? a0.down_part2 : a0.down_part;
  const float* const delta = l > 0 ? ((l & 1) ? a0.down_part : a0.down_part2) : a0.delta;   // the previous layer's buffer
  const int dsplits = l > 0 ? 16 : a0.dsplits;
  unsigned* const cnt = a0.cnt + (int64_t)l * PSG_DL_NCNT;
  long long* const tr = (a0.trace && l == nl - 1) ? a0.trace + (int64_t)b * 24 : nullptr;
#define DL_STAMP(i)                                             \
  do {                                                          \
    if (tr && tid == 0) tr[i] = (long long)wall_clock64();      \
  } while (0)
  DL_STAMP(0);

  // ---- q|k|v: S = 8, G = 32: 192-row slabs; a head's q, k and v rows lie in up to six slabs (dl_attn_round) ------------
  if (l == 0) {
    GemmQ g;
    g.setup(T->wqkv, 3 * D, D, 8, b & 7, b >> 3, M, smem);
    g.prefetch();
  } else {
    // owner b's 16 columns lie in down slab b >> 3, produced (for its 16 K slices) by column group (b >> 3) & 15
    dl_wait(cnt - PSG_DL_NCNT, PSG_DL_CNT_DGRP + ((b >> 3) & 15), 1, 16u);
  }
  dl_norm_owner(a, delta, dsplits, a.ssq, b);
  dl_drain();
  dl_arrive_all(cnt, PSG_DL_CNT_X1);
  DL_STAMP(1);
  dl_wait_all(cnt, PSG_DL_CNT_X1);
  DL_STAMP(2);
  {
    GemmQ g;
    g.setup(T->wqkv, 3 * D, D, 8, b & 7, b >> 3, M, smem);
    g.skip_prefetched();
    dl_norm_stage(a, a.ssq, T->ln1, g.kbA, g.nkb, g.xstride, xs, s_inv);
    dl_barrier();
    DL_STAMP(3);
    g.run(a.qkv_part, smem, xs);
  }
  DL_STAMP(4);
  // the o projection's first blocks (slab b >> 3, K slice b & 7) while the attention runs
  {
    GemmO go;
    go.setup(T->wo, D, D, 8, b & 7, b >> 3, M, smem);
    dl_drain();                                                     // this wave's partial-tile stores are out
    go.prefetch();                                                  // its own ring is free: the next stream starts now
  }
  dl_publish(cnt, PSG_DL_CNT_HEAD + (b >> 3));
  DL_STAMP(5);

  // ---- attention: units (row, head) dealt to workgroups, three at a time (four waves each): one round for 24 x 32 units --
  {
    PsgDecodeAttnScratch* sc = reinterpret_cast<PsgDecodeAttnScratch*>(xs) + (wid >> 2);
    const int nunit = M * a.heads;                                    // unit u = row * heads + head
    for (int u0 = 3 * b; u0 < nunit; u0 += 3 * PSG_DL_WG) {
      const int u = u0 + (wid >> 2);
      const DlAttnArgs aa = {a.qkv_part, a.att, T->kc, T->vc, a.tok_pair, a.tok_pos, a.cos_tab, a.sin_tab, cnt, a.M, a.D, a.heads, a.ctx};
      dl_attn_round(aa, u < nunit ? u : -1, sc);
      dl_drain();
      dl_barrier();
      if ((tid & 255) == 0 && u < nunit)
        __hip_atomic_fetch_add(cnt + (PSG_DL_CNT_ATT + ((u % a.heads) >> 2)) * PSG_DL_SLOT, 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  // ---- o projection: K slice by = heads 4 by .. 4 by + 3 -----------------------------------------------------------------
  DL_STAMP(6);
  dl_wait(cnt, PSG_DL_CNT_ATT + (b & 7), 1, (unsigned)(4 * M));
  DL_STAMP(7);
  {
    GemmO go;
    go.setup(T->wo, D, D, 8, b & 7, b >> 3, M, smem);
    go.skip_prefetched();
    go.stage_x_dma(a.att, D, xs);
    dl_drain();
    dl_barrier();
    DL_STAMP(8);
    go.run(a.o_part, smem, xs);
  }
  DL_STAMP(9);
  {
    GemmG gg;
    gg.setup(T->wgu, 2 * I, D, 8, b & 7, b >> 3, M, smem);
    dl_drain();
    gg.prefetch();
  }
  dl_publish(cnt, PSG_DL_CNT_OSLAB + (b >> 3));

  // ---- post-attention RMSNorm: owner b's 16 columns lie in o slab b >> 3 ---------------------------------------------------
  dl_wait(cnt, PSG_DL_CNT_OSLAB + (b >> 3), 1, 8u);
  DL_STAMP(10);
  dl_norm_owner(a, a.o_part, 8, a.ssq + PSG_DL_WG * 32, b);
  dl_drain();
  dl_arrive_all(cnt, PSG_DL_CNT_X2);
  DL_STAMP(11);
  dl_wait_all(cnt, PSG_DL_CNT_X2);
  DL_STAMP(12);
  {
    GemmG gg;
    gg.setup(T->wgu, 2 * I, D, 8, b & 7, b >> 3, M, smem);
    gg.skip_prefetched();
    dl_norm_stage(a, a.ssq + PSG_DL_WG * 32, T->ln2, gg.kbA, gg.nkb, gg.xstride, xs, s_inv);
    dl_barrier();
    DL_STAMP(13);
    gg.run(a.gu_part, smem, xs);
  }
  DL_STAMP(14);
  {
    GemmO gd;
    gd.setup(T->wdown, D, I, 16, b & 15, b >> 4, M, smem);
    dl_drain();
    gd.prefetch();
  }
  dl_publish(cnt, PSG_DL_CNT_GU + (b >> 3));

  // ---- SwiGLU: items (row, 128-column block) dealt to waves; silu_mul_kernel<float>'s arithmetic ---------------------------
  {
    const int nblk = I >> 7;
    const int nitem = M * nblk;
    const int64_t sl = (int64_t)M * 2 * I;
    for (int it0 = b * PSG_DL_WAVES; it0 < nitem; it0 += PSG_DL_WG * PSG_DL_WAVES) {
      const int it = it0 + wid;
      const bool liv = it < nitem;
      const int m = liv ? it / nblk : 0, j = liv ? it - m * nblk : 0;
      // gate columns [128 j, +128) and up columns [I + 128 j, +128) lie in 176-row slabs (first and last column may
      // straddle a slab end): their column groups (slab % 32), for the round's 12 items - lanes 0..47 poll one each
      if (tid < 64) {
        const int ii = lane >> 2, part = lane & 3;                   // item of the round, (gate | up) x (first | last column)
        const int itl = min(it0 + ii, nitem - 1);
        const int jl = itl % nblk;
        const int col = (part >> 1) * I + jl * 128 + ((part & 1) ? 127 : 0);
        const int grp = (col / 176) & 31;
        unsigned spins = 0;
        for (;;) {
          const bool ok = lane >= 4 * PSG_DL_WAVES || dl_poll(cnt, PSG_DL_CNT_GU + grp, 8u);
          if (__all(ok)) break;
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1u << 21)) {
            if (lane == 0) dl_timeout(cnt, 0x5147u);
            break;
          }
        }
      }
      dl_barrier();
      if (liv) {
        const int c = j * 128 + 2 * lane;
        const int64_t ig = (int64_t)m * 2 * I + c, iu = ig + I;
        float2 tg[8], tu[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          tg[s] = dl_ld2(a.gu_part + (int64_t)s * sl + ig);
          tu[s] = dl_ld2(a.gu_part + (int64_t)s * sl + iu);
        }
        float2 gsum = tg[0], usum = tu[0];
#pragma unroll
        for (int s = 1; s < 8; ++s) { gsum.x += tg[s].x; gsum.y += tg[s].y; usum.x += tu[s].x; usum.y += tu[s].y; }
        const float s0 = gsum.x / (1.0f + expf(-gsum.x)), s1 = gsum.y / (1.0f + expf(-gsum.y));
        dl_st2(a.h + (int64_t)m * I + c, s0 * usum.x, s1 * usum.y);
        dl_drain();
        if (lane == 0)
          __hip_atomic_fetch_add(cnt + (PSG_DL_CNT_H + j) * PSG_DL_SLOT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }

  // ---- down projection: S = 16, G = 16; K slice by: blocks of 128 columns [kbA / 4, (kbB + 3) / 4) --------------------------
  DL_STAMP(15);
  {
    GemmO gd;
    gd.setup(T->wdown, D, I, 16, b & 15, b >> 4, M, smem);
    gd.skip_prefetched();
    const int j0 = gd.kbA >> 2, j1 = (gd.kbA + gd.nkb + 3) >> 2;
    dl_wait(cnt, PSG_DL_CNT_H + j0, j1 - j0, (unsigned)M);
    DL_STAMP(16);
    gd.stage_x_dma(a.h, I, xs);
    dl_drain();
    dl_barrier();
    DL_STAMP(17);
    gd.run(down_part, smem, xs);
  }
  DL_STAMP(18);
  if (l + 1 < nl) {                                                 // chain: next layer's first q|k|v blocks, then count in
    GemmQ g;
    g.setup(T[1].wqkv, 3 * D, D, 8, b & 7, b >> 3, M, smem);
    dl_drain();
    g.prefetch();
    dl_publish(cnt, PSG_DL_CNT_DGRP + (b >> 4));
  }
#undef DL_STAMP
  }
}

__global__ void dl_write_table_kernel(psg_dl_layer* tab, const psg_dl_layer one) { *tab = one; }
static int64_t dl_ws_floats(int M, int hidden, int inter) {
  return (int64_t)8 * M * 3 * hidden + (int64_t)M * hidden + (int64_t)8 * M * hidden + 2 * 256 * 32 + (int64_t)8 * M * 2 * inter +
         (int64_t)M * inter;
}
static size_t dl_lds(int M, int slots, int mp) {
  const size_t xmax = (size_t)M * (22 * 128 + DL_XPAD);               // the down projection's slice (K = 11008, S = 16)
  const size_t attn = 3 * sizeof(PsgDecodeAttnScratch);
  return (size_t)PSG_DL_WAVES * slots * DL_BLOCK + (size_t)mp * (PSG_DL_WAVES * 16 + 4) * 4 + (xmax > attn ? xmax : attn);
}

extern "C" int psg_decode_layer_workspace(psg_ctx* ctx, int M, int hidden, int inter, int64_t* floats, int64_t* counters) {
  PSG_REQUIRE(ctx && floats && counters, PSG_ERR_INVALID, "psg_decode_layer_workspace: NULL argument");
  PSG_REQUIRE(M >= 1 && M <= 32, PSG_ERR_UNSUPPORTED, "psg_decode_layer: M=%d (1..32 rows)", M);
  // qkv_part 8 M 3D | att M D | o_part 8 M D | ssq 2*256*32 | gu_part 8 M 2I | h M I   (down_part is the caller's output)
  *floats = dl_ws_floats(M, hidden, inter) + 16;                   // + a one-entry layer table (psg_decode_layer)
  *counters = PSG_DL_NCNT;
  return PSG_OK;
}

extern "C" int psg_decode_layer_supported(psg_ctx* ctx, int M, int hidden, int inter, int heads, int dtype) {
  if (!ctx) return 0;
  const int KBd = inter >> 5;
  return dtype == PSG_F32 && ctx->num_cu == PSG_DL_WG && M >= 13 && M <= 24 && hidden == 4096 && heads == 32 &&
         inter % 128 == 0 && inter / 128 <= PSG_DL_CNT_DGRP - PSG_DL_CNT_H && (KBd + 15) / 16 <= 22 && inter >= 2048;
}

static int dl_launch(psg_ctx* ctx, psg_dl_args& a, float* workspace, uint32_t* counters, hipStream_t st, const char* who) {
  const int M = a.M, hidden = a.D, inter = a.I;
  float* w = workspace;
  a.qkv_part = w; w += (size_t)8 * M * 3 * hidden;
  a.att = w; w += (size_t)M * hidden;
  a.o_part = w; w += (size_t)8 * M * hidden;
  a.ssq = w; w += 2 * 256 * 32;
  a.gu_part = w; w += (size_t)8 * M * 2 * inter;
  a.h = w;
  a.cnt = counters;
  a.trace = (ctx->trace_kind == PSG_TRACE_DECODE_LAYER && ctx->trace && ctx->trace_words >= (int64_t)PSG_DL_WG * 24) ? ctx->trace : nullptr;
  const int g16 = 1, g4 = M <= 16 ? 0 : (M - 16 + 3) / 4;
  const int mp = g16 * 16 + g4 * 4;
  const size_t lds = dl_lds(M, 3, mp);
  PSG_REQUIRE(lds <= 160 * 1024, PSG_ERR_UNSUPPORTED, "%s: %zu B of LDS", who, lds);
#define DL_K(A, B)                                                                                          \
  do {                                                                                                      \
    (void)hipFuncSetAttribute((const void*)decode_layer_f32_kernel<3, A, B>,                                \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                      \
    decode_layer_f32_kernel<3, A, B><<<PSG_DL_WG, PSG_DL_WAVES * 64, lds, st>>>(a);                         \
  } while (0)
  switch (g4) {
    case 0: DL_K(1, 0); break;
    case 1: DL_K(1, 1); break;
    default: DL_K(1, 2); break;
  }
#undef DL_K
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    psg_set_error("%s: launch failed: %s", who, hipGetErrorString(e));
    return PSG_ERR_HIP;
  }
  return PSG_OK;
}

// resid [M][hidden] fp32 in/out; delta = the previous layer's down partials (delta_splits slices, or NULL);
// down_part [16][M][hidden] out; workspace / counters as psg_decode_layer_workspace (counters zeroed by the caller).
extern "C" int psg_decode_layer(psg_ctx* ctx, void* resid, const void* delta, int delta_splits, const float* ln1,
                                const float* ln2, const void* wqkv, const void* wo, const void* wgu, const void* wdown,
                                const int32_t* tok_pair, const int32_t* tok_pos, const float* rope_cos,
                                const float* rope_sin, int M, int hidden, int inter, int heads, int ctx_len, float eps,
                                void* k_cache, void* v_cache, float* workspace, uint32_t* counters, float* down_part,
                                int dtype, void* stream) {
  PSG_REQUIRE(ctx && resid && ln1 && ln2 && wqkv && wo && wgu && wdown && tok_pair && tok_pos && rope_cos && rope_sin &&
                  k_cache && v_cache && workspace && counters && down_part,
              PSG_ERR_INVALID, "psg_decode_layer: NULL argument");
  PSG_REQUIRE(psg_decode_layer_supported(ctx, M, hidden, inter, heads, dtype), PSG_ERR_UNSUPPORTED,
              "psg_decode_layer: M=%d hidden=%d inter=%d heads=%d dtype=%d on %d CUs (fp32, 13..24 rows, 4096 = 32 x 128, "
              "256 CUs)", M, hidden, inter, heads, dtype, ctx->num_cu);
  PSG_REQUIRE(delta_splits >= 0 && delta_splits <= PSG_MAX_SPLITS && (delta || delta_splits == 0), PSG_ERR_INVALID,
              "psg_decode_layer: delta_splits=%d", delta_splits);
  psg_dl_args a;
  // the kernel reads its layers from a device table: a one-entry table behind the workspace's last buffer (psg_decode_layer_workspace counts it)
  psg_dl_layer* tab = reinterpret_cast<psg_dl_layer*>(workspace + dl_ws_floats(M, hidden, inter));
  const psg_dl_layer one = {ln1, ln2, (const float*)wqkv, (const float*)wo, (const float*)wgu, (const float*)wdown,
                            (float*)k_cache, (float*)v_cache};
  dl_write_table_kernel<<<1, 1, 0, (hipStream_t)stream>>>(tab, one);
  a.table = tab; a.n_layers = 1; a.down_part2 = nullptr;
  a.resid = (float*)resid;
  a.delta = delta_splits > 0 ? (const float*)delta : nullptr;
  a.dsplits = delta_splits;
  a.tok_pair = tok_pair; a.tok_pos = tok_pos; a.cos_tab = rope_cos; a.sin_tab = rope_sin;
  a.down_part = down_part;
  a.M = M; a.D = hidden; a.I = inter; a.heads = heads; a.ctx = ctx_len; a.eps = eps;
  return dl_launch(ctx, a, workspace, counters, (hipStream_t)stream, "psg_decode_layer");
}

// n_layers decoder layers chained inside ONE launch.  layer_table: device array of n_layers x 8 pointers
// {ln1, ln2, wqkv, wo, wgu, wdown, k_cache, v_cache}; delta feeds layer 0 (NULL: none); layer l leaves its down partials in
// down_parts + (l & 1) * 16 M hidden - the caller's final RMSNorm reads buffer (n_layers - 1) & 1; counters: n_layers blocks
// of psg_decode_layer_workspace's size, zeroed.
extern "C" int psg_decode_layers(psg_ctx* ctx, void* resid, const void* delta, int delta_splits, const void* layer_table,
                                 int n_layers, const int32_t* tok_pair, const int32_t* tok_pos, const float* rope_cos,
                                 const float* rope_sin, int M, int hidden, int inter, int heads, int ctx_len, float eps,
                                 float* workspace, uint32_t* counters, float* down_parts, int dtype, void* stream) {
  PSG_REQUIRE(ctx && resid && layer_table && tok_pair && tok_pos && rope_cos && rope_sin && workspace && counters && down_parts,
              PSG_ERR_INVALID, "psg_decode_layers: NULL argument");
  PSG_REQUIRE(n_layers >= 1 && n_layers <= 1024, PSG_ERR_INVALID, "psg_decode_layers: n_layers=%d", n_layers);
  PSG_REQUIRE(psg_decode_layer_supported(ctx, M, hidden, inter, heads, dtype), PSG_ERR_UNSUPPORTED,
              "psg_decode_layers: M=%d hidden=%d inter=%d heads=%d dtype=%d on %d CUs (fp32, 13..24 rows, 4096 = 32 x 128, "
              "256 CUs)", M, hidden, inter, heads, dtype, ctx->num_cu);
  PSG_REQUIRE(delta_splits >= 0 && delta_splits <= PSG_MAX_SPLITS && (delta || delta_splits == 0), PSG_ERR_INVALID,
              "psg_decode_layers: delta_splits=%d", delta_splits);
  psg_dl_args a;
  a.table = (const psg_dl_layer*)layer_table; a.n_layers = n_layers;
  a.resid = (float*)resid;
  a.delta = delta_splits > 0 ? (const float*)delta : nullptr;
  a.dsplits = delta_splits;
  a.tok_pair = tok_pair; a.tok_pos = tok_pos; a.cos_tab = rope_cos; a.sin_tab = rope_sin;
  a.down_part = down_parts;
  a.down_part2 = down_parts + (size_t)16 * M * hidden;
  a.M = M; a.D = hidden; a.I = inter; a.heads = heads; a.ctx = ctx_len; a.eps = eps;
  return dl_launch(ctx, a, workspace, counters, (hipStream_t)stream, "psg_decode_layers");
}
