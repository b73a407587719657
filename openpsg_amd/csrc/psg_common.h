// Shared device/host helpers for libpsg_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <type_traits>

#include "../../include/psg_hip.h"

// Tunables of a context (psg_set_option).  Defaults are the measured-best settings; psg_create reads the PSG_*
// environment variables of the same names ONCE per context, so an experiment needs no rebuild.  There is no
// process-global mutable state: everything a launch consults lives here.
struct psg_opts {
  int skinny_splits = 0;        // > 0: forced split-K count of psg_skinny_gemm_plan
  int skinny_balance = 1;       // last-round balance rule of the split planner
  int skinny_wg_per_cu = 3;     // register variant: persistent workgroups per CU
  int skinny_dma = 813;         // LDS-DMA variant <waves><K blocks per batch><ring slots>; 0 = register variant
  int skinny_nt = 1;            // non-temporal weight DMAs
  int skinny_xdma = 1;          // x slice staged by LDS-DMA as well
  int skinny_f32_slots = 3;     // fp32 weight-streaming kernel (psg_gemm_f32.hip): 2 KB blocks per wave's DMA ring (3 or 5)
  int skinny_f32_waves = 0;     // fp32 kernel: forced slab height in wavefronts (8, 11, 12, 16); 0 = the planner's choice
  int skinny_wide = 1;          // 11-wave (176-row) slabs where they save a round of slabs (gate/up projection)
  int selfattn_scalar = 0;      // Q-Former self-attention: scalar checker kernel even in bf16
  int decode_attn_1wave = 0;    // decode attention: one wave per (pair, head) instead of a workgroup
  int dense_gemm_var = 0;       // psg_dense_gemm ablation builds (1: no MFMA, 2: no staging); 0 = the real kernel
  int qformer_own_gemm = 1;     // 1: Q-Former FFN1 on psg_dense_gemm (fused bias + GELU); 2: EVERY Q-Former projection on it (row-count invariant: bit-exact pair sharding); 0: library GEMMs
  int xattn_waves = 8;          // LDS-DMA cross-attention: 8 = ten waves for launches with < 32 tiles per wave, else eight; 10 = always ten; 0 = always eight
  int xattn_dma = 1;            // cross-attention: LDS-DMA kernel (psg_xattn_dma.hip) when its LDS image fits
  int ln_half_wave = 1;         // add + LayerNorm on 16-bit rows: half a wave per row, 16-byte accesses
  // host-side variants of the engines (read by openpsg_amd/qformer.py / llm.py through psg_get_option: one mechanism,
  // per context, for every A/B switch - none is a process-wide environment toggle of the Python layer)
  int qformer_share_qkv = 1;    // layer 0: ONE Q/K/V projection of the 33 learned query rows for all pairs
  int qformer_cls_input_space = 1;  // selection phase of the last layer in the input space (psg_qformer_cls_attn_input)
  int qformer_dedup_prompts = 1;    // prompt-only work of the two-layer Q-Former once per distinct prompt
  int llm_fuse_rmsnorm = 0;     // decode step: RMSNorm as the prologue of the projection it feeds (psg_skinny_gemm_fused)
  int prefill_attn_scalar = 0;  // prompt pass: scalar cache-attention kernel instead of the matrix-core one
  int llm_fuse_split = 1;       // fp32s prompt pass: operand splits / result scalings inside the row kernels (psg_split.hip)
  int wt_stores = 1;            // decode-step kernels of the fp32 modes store their outputs write-through (sc1): nothing dirty
                                // in the eight L2s when a launch ends - 0.5-0.8 us per kernel boundary.  Bits:
                                // 1 = fp32 weight-streaming GEMM + rmsnorm / silu / attention (16-byte stores: -3 ms per
                                // image), 2 = the PAIR GEMM of fp16-valued weights (4-byte stores: measured +4 ms, off),
                                // 4 = rmsnorm_split2 / split_f16x2 (8-byte stores)
  int qformer_split_cls_input_space = 1;   // fp32s mode: selection phase in the input space + prompt de-duplication (own products)
  int split_i2 = 1;             // fp32s own-GEMM products (Q-Former, row-invariant Llama prompt pass): interleaved hi / lo
                                // operands through psg_dense_gemm_split (3 products from one staging; 0: the K' = 3K form)
  int decode_persistent = 0;    // fp32 decode steps: one persistent launch per decoder layer (psg_decode_layer) instead of the
                                // chain of eight launches (bit-identical; Llama-2-7B width, 13..24 rows, 256 CUs)
  int llm_w16 = 1;              // fp32 engines: stream projection weights that are fp16 VALUES (verified per tensor) as fp16
  int batch_gemm_bn = 0;        // psg_batch_gemm: forced slab height (256 or 128 weight rows); 0 = the planner's estimate
  int batch_gemm_mode = 0;      // psg_batch_gemm: 1 = slab-aligned slices, 2 = stream-K ranges; 0 = the planner's estimate
  int batch_gemm_grid = 0;      // psg_batch_gemm: workgroups (0 = one per CU): fewer workgroups = fewer fp32 slices
  int batch_gemm_var = 0;       // psg_batch_gemm ablation runs (1: no slice stores, 2: no MFMA, 3: x staged for the first K step only)
  int decode_batch_gemm = 1;    // decode steps of 33..160 rows (forward_batch): psg_batch_gemm instead of the library GEMM
  int xattn_dynamic = 1;        // LDS-DMA cross-attention: a workgroup's waves draw their tiles from an LDS counter
  int xattn_wt = 1;             // LDS-DMA cross-attention: output rows stored write-through (sc1): no dirty L2 lines to flush
                                // when the launch ends (70.0 -> 67.6 us in situ at C2)
  int xattn_poll = 0;           // LDS-DMA cross-attention: a unit's Q tile is awaited by polling a sentinel in its LDS slot
                                // instead of a vmcnt count (which also waits for the previous unit's stores to retire)
};

struct psg_ctx {
  int device;
  int num_cu;
  char arch[64];
  psg_opts opt;
  // caller-provided device buffer for per-wave cycle-counter stamps (psg_set_trace_buffer); the library never
  // allocates, copies or synchronises for it
  long long* trace = nullptr;
  int64_t trace_words = 0;
  int trace_kind = 0;           // PSG_TRACE_*
  size_t skinny_lds_configured = 0;
};

void psg_set_error(const char* fmt, ...);

#define PSG_REQUIRE(cond, code, ...)      \
  do {                                    \
    if (!(cond)) {                        \
      psg_set_error(__VA_ARGS__);         \
      return (code);                      \
    }                                     \
  } while (0)

#define PSG_CHECK_LAUNCH(name)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      psg_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return PSG_ERR_HIP;                                                   \
    }                                                                       \
  } while (0)

// ---- activation storage types -------------------------------------------------------------
struct bf16_t {
  uint16_t v;
};

__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {  // round to nearest even
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

struct f16_t {
  uint16_t v;
};
__device__ __forceinline__ float f16_to_f32(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ uint16_t f32_to_f16(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }  // RNE

template <typename T>
struct Act;
template <>
struct Act<float> {
  static __device__ __forceinline__ float rnd(float v) { return v; }
  static __device__ __forceinline__ float ld(const float* p, int64_t i) { return p[i]; }
  static __device__ __forceinline__ void st(float* p, int64_t i, float v) { p[i] = v; }
  static __device__ __forceinline__ void ld4(const float* p, int64_t i, float (&o)[4]) {
    float4 t = *reinterpret_cast<const float4*>(p + i);
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
  }
  static __device__ __forceinline__ void st4(float* p, int64_t i, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p + i) = make_float4(v[0], v[1], v[2], v[3]);
  }
  // raw loads: the value stays as stored until cv / cv4, so nothing forces a wait at the load site
  typedef float raw1;
  typedef float4 raw4;
  static __device__ __forceinline__ raw1 ldr(const float* p, int64_t i) { return p[i]; }
  static __device__ __forceinline__ raw4 ldr4(const float* p, int64_t i) { return *reinterpret_cast<const float4*>(p + i); }
  static __device__ __forceinline__ float cv(raw1 r) { return r; }
  static __device__ __forceinline__ void cv4(raw4 t, float (&o)[4]) { o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w; }
};
template <>
struct Act<bf16_t> {
  static __device__ __forceinline__ float rnd(float v) { return bf16_to_f32(f32_to_bf16(v)); }  // as stored
  static __device__ __forceinline__ float ld(const bf16_t* p, int64_t i) { return bf16_to_f32(p[i].v); }
  static __device__ __forceinline__ void st(bf16_t* p, int64_t i, float v) { p[i].v = f32_to_bf16(v); }
  static __device__ __forceinline__ void ld4(const bf16_t* p, int64_t i, float (&o)[4]) {
    ushort4 t = *reinterpret_cast<const ushort4*>(p + i);
    o[0] = bf16_to_f32(t.x); o[1] = bf16_to_f32(t.y); o[2] = bf16_to_f32(t.z); o[3] = bf16_to_f32(t.w);
  }
  static __device__ __forceinline__ void st4(bf16_t* p, int64_t i, const float (&v)[4]) {
    ushort4 t;
    t.x = f32_to_bf16(v[0]); t.y = f32_to_bf16(v[1]); t.z = f32_to_bf16(v[2]); t.w = f32_to_bf16(v[3]);
    *reinterpret_cast<ushort4*>(p + i) = t;
  }
  typedef uint16_t raw1;
  typedef ushort4 raw4;
  static __device__ __forceinline__ raw1 ldr(const bf16_t* p, int64_t i) { return p[i].v; }
  static __device__ __forceinline__ raw4 ldr4(const bf16_t* p, int64_t i) { return *reinterpret_cast<const ushort4*>(p + i); }
  static __device__ __forceinline__ float cv(raw1 r) { return bf16_to_f32(r); }
  static __device__ __forceinline__ void cv4(raw4 t, float (&o)[4]) {
    o[0] = bf16_to_f32(t.x); o[1] = bf16_to_f32(t.y); o[2] = bf16_to_f32(t.z); o[3] = bf16_to_f32(t.w);
  }
};

template <>
struct Act<f16_t> {
  static __device__ __forceinline__ float rnd(float v) { return f16_to_f32(f32_to_f16(v)); }
  static __device__ __forceinline__ float ld(const f16_t* p, int64_t i) { return f16_to_f32(p[i].v); }
  static __device__ __forceinline__ void st(f16_t* p, int64_t i, float v) { p[i].v = f32_to_f16(v); }
  static __device__ __forceinline__ void ld4(const f16_t* p, int64_t i, float (&o)[4]) {
    ushort4 t = *reinterpret_cast<const ushort4*>(p + i);
    o[0] = f16_to_f32(t.x); o[1] = f16_to_f32(t.y); o[2] = f16_to_f32(t.z); o[3] = f16_to_f32(t.w);
  }
  static __device__ __forceinline__ void st4(f16_t* p, int64_t i, const float (&v)[4]) {
    ushort4 t;
    t.x = f32_to_f16(v[0]); t.y = f32_to_f16(v[1]); t.z = f32_to_f16(v[2]); t.w = f32_to_f16(v[3]);
    *reinterpret_cast<ushort4*>(p + i) = t;
  }
  typedef uint16_t raw1;
  typedef ushort4 raw4;
  static __device__ __forceinline__ raw1 ldr(const f16_t* p, int64_t i) { return p[i].v; }
  static __device__ __forceinline__ raw4 ldr4(const f16_t* p, int64_t i) { return *reinterpret_cast<const ushort4*>(p + i); }
  static __device__ __forceinline__ float cv(raw1 r) { return f16_to_f32(r); }
  static __device__ __forceinline__ void cv4(raw4 t, float (&o)[4]) {
    o[0] = f16_to_f32(t.x); o[1] = f16_to_f32(t.y); o[2] = f16_to_f32(t.z); o[3] = f16_to_f32(t.w);
  }
};

// ---- 16-bit element traits of the matrix-core kernels (bf16 / fp16: same kernels, other MFMA opcodes) ------------
typedef float psg_f32x16 __attribute__((ext_vector_type(16)));
typedef float psg_f32x4 __attribute__((ext_vector_type(4)));
typedef float psg_f32x2 __attribute__((ext_vector_type(2)));
struct EBf16 {
  using act = bf16_t;
  typedef __bf16 v8 __attribute__((ext_vector_type(8)));
  typedef __bf16 v2 __attribute__((ext_vector_type(2)));
  static constexpr uint32_t ONE = 0x3f80u, NEG_2_15 = 0xc700u;                  // 1.0, -32768.0
  static __device__ __forceinline__ psg_f32x16 mfma32(v8 a, v8 b, psg_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ psg_f32x4 mfma16(v8 a, v8 b, psg_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack(float lo, float hi) {         // v_cvt_pk_bf16_f32 (RNE)
    psg_f32x2 f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, v2));
  }
  static __device__ __forceinline__ float to_f32(uint16_t x) { return bf16_to_f32(x); }
  static __device__ __forceinline__ uint16_t from_f32(float x) { return f32_to_bf16(x); }
};
struct EF16 {
  using act = f16_t;
  typedef _Float16 v8 __attribute__((ext_vector_type(8)));
  typedef _Float16 v2 __attribute__((ext_vector_type(2)));
  static constexpr uint32_t ONE = 0x3c00u, NEG_2_15 = 0xf800u;
  static __device__ __forceinline__ psg_f32x16 mfma32(v8 a, v8 b, psg_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ psg_f32x4 mfma16(v8 a, v8 b, psg_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack(float lo, float hi) {         // RNE, not v_cvt_pkrtz
    psg_f32x2 f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, v2));
  }
  static __device__ __forceinline__ float to_f32(uint16_t x) { return f16_to_f32(x); }
  static __device__ __forceinline__ uint16_t from_f32(float x) { return f32_to_f16(x); }
};
// run BODY with `E` = the element traits of a 16-bit activation dtype
#define PSG_DISPATCH_E16(dtype, NAME, ...)                                            \
  do {                                                                                \
    if ((dtype) == PSG_BF16) {                                                        \
      using E = EBf16;                                                                \
      __VA_ARGS__;                                                                    \
    } else if ((dtype) == PSG_F16) {                                                  \
      using E = EF16;                                                                 \
      __VA_ARGS__;                                                                    \
    } else {                                                                          \
      psg_set_error("%s: needs a 16-bit activation dtype, got %d", NAME, (int)(dtype)); \
      return PSG_ERR_UNSUPPORTED;                                                     \
    }                                                                                 \
  } while (0)

// ---- wave (64 lanes) reductions ------------------------------------------------------------
// On the VALU's data-parallel-primitive path (row shifts + row broadcasts; the result is read from lane 63 and is
// wave-uniform): ~8 instructions of a few cycles each.  The __shfl_xor butterfly these replaced goes through the LDS
// crossbar (ds_bpermute, ~100 cycles per step, 6 dependent steps) - in the latency-bound row kernels of the decode
// step and in the one-wave-per-row LayerNorm that was a visible share of the kernel.
template <int CTRL, int ROWS>
__device__ __forceinline__ float psg_dpp(float old, float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, x), CTRL,
                                                                ROWS, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += psg_dpp<0x111, 0xf>(0.f, v);   // row_shr:1
  v += psg_dpp<0x112, 0xf>(0.f, v);   // row_shr:2
  v += psg_dpp<0x114, 0xf>(0.f, v);   // row_shr:4
  v += psg_dpp<0x118, 0xf>(0.f, v);   // row_shr:8  -> lane 15 of every row of 16 = the row's sum
  v += psg_dpp<0x142, 0xa>(0.f, v);   // row_bcast:15 into rows 1 and 3
  v += psg_dpp<0x143, 0xc>(0.f, v);   // row_bcast:31 into rows 2 and 3 -> lane 63 = the wave's sum
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_sum_dpp(float v) { return wave_sum(v); }
__device__ __forceinline__ float wave_max(float v) {               // lanes without a source keep their own value
  v = fmaxf(v, psg_dpp<0x111, 0xf>(v, v));
  v = fmaxf(v, psg_dpp<0x112, 0xf>(v, v));
  v = fmaxf(v, psg_dpp<0x114, 0xf>(v, v));
  v = fmaxf(v, psg_dpp<0x118, 0xf>(v, v));
  v = fmaxf(v, psg_dpp<0x142, 0xa>(v, v));
  v = fmaxf(v, psg_dpp<0x143, 0xc>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// sum over the 4 lanes of a quad, in every lane of the quad (lane ^ 1, then lane ^ 2: same association as the
// __shfl_xor pair it replaces)
__device__ __forceinline__ float quad_sum(float v) {
  v += psg_dpp<0xB1, 0xf>(0.f, v);    // quad_perm:[1,0,3,2]
  v += psg_dpp<0x4E, 0xf>(0.f, v);    // quad_perm:[2,3,0,1]
  return v;
}

#define PSG_MAX_SPLITS 16  // split-K slices a consumer kernel can sum (psg_skinny_gemm_plan stays below)

#define PSG_FMIN (-3.402823466e+38f)  // torch.finfo(float32).min, the legacy additive mask value

// ---- split-K partial inputs --------------------------------------------------------------------
// The decode projections (psg_skinny_gemm) leave fp32 partials part[S][rows][cols]; their consumers
// sum the S slices in split order while loading, then round once to the activation dtype (exactly
// what a GEMM with an activation-dtype output would have stored).
template <typename T>
__device__ __forceinline__ void ld4_in(const void* __restrict__ in, int S, int64_t slice, int64_t i, float (&o)[4]) {
  if (S > 0) {
    const float* p = reinterpret_cast<const float*>(in);
    float4 t[PSG_MAX_SPLITS];
#pragma unroll
    for (int s = 0; s < PSG_MAX_SPLITS; ++s)
      if (s < S) t[s] = *reinterpret_cast<const float4*>(p + (int64_t)s * slice + i);
    float4 a = t[0];
#pragma unroll
    for (int s = 1; s < PSG_MAX_SPLITS; ++s)
      if (s < S) { a.x += t[s].x; a.y += t[s].y; a.z += t[s].z; a.w += t[s].w; }
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = Act<T>::rnd(o[e]);
  } else {
    Act<T>::ld4(reinterpret_cast<const T*>(in), i, o);
  }
}
template <typename T>
__device__ __forceinline__ float ld1_in(const void* __restrict__ in, int S, int64_t slice, int64_t i) {
  if (S > 0) {
    const float* p = reinterpret_cast<const float*>(in);
    float t[PSG_MAX_SPLITS];
#pragma unroll
    for (int s = 0; s < PSG_MAX_SPLITS; ++s)
      if (s < S) t[s] = p[(int64_t)s * slice + i];
    float a = t[0];
#pragma unroll
    for (int s = 1; s < PSG_MAX_SPLITS; ++s)
      if (s < S) a += t[s];
    return Act<T>::rnd(a);
  }
  return Act<T>::ld(reinterpret_cast<const T*>(in), i);
}

// NV values per slice in ONE pass: every load of every slice is issued before the first sum.  Separate ld4_in /
// ld1_in calls serialise (each call's slice-count branches end in its own s_waitcnt: one HBM round trip per call —
// six of them for q/k/v in the decode attention), and the decode row kernels are nothing but round trips.
// Sums run in slice order per value, exactly as ld4_in / ld1_in.  V = float or float4; S >= 1.
__device__ __forceinline__ void psg_acc(float& a, const float& b) { a += b; }
__device__ __forceinline__ void psg_acc(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
template <typename V, int NV>
__device__ __forceinline__ void ldn_splits(const void* __restrict__ in, int S, int64_t slice, const int64_t (&idx)[NV],
                                           V (&o)[NV]) {
  const float* p = reinterpret_cast<const float*>(in);
  V t[PSG_MAX_SPLITS][NV];
#pragma unroll
  for (int s = 0; s < PSG_MAX_SPLITS; ++s)
    if (s < S) {
#pragma unroll
      for (int v = 0; v < NV; ++v) t[s][v] = *reinterpret_cast<const V*>(p + (int64_t)s * slice + idx[v]);
    }
#pragma unroll
  for (int v = 0; v < NV; ++v) o[v] = t[0][v];
#pragma unroll
  for (int s = 1; s < PSG_MAX_SPLITS; ++s)
    if (s < S) {
#pragma unroll
      for (int v = 0; v < NV; ++v) psg_acc(o[v], t[s][v]);
    }
}

// 16-byte store written through the L2 (agent scope: sc1) - a kernel whose outputs are all stored this way leaves no dirty
// line behind for the release at its end
__device__ __forceinline__ void psg_st4_wt(float* p, float a, float b, float c, float d) {
  typedef float psg_wt_f4 __attribute__((ext_vector_type(4)));
  const psg_wt_f4 v = {a, b, c, d};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// dispatch on the activation dtype enum
#define PSG_DISPATCH_DTYPE(dtype, NAME, ...)                       \
  do {                                                             \
    if ((dtype) == PSG_F32) {                                      \
      using T = float;                                             \
      __VA_ARGS__;                                                 \
    } else if ((dtype) == PSG_BF16) {                              \
      using T = bf16_t;                                            \
      __VA_ARGS__;                                                 \
    } else if ((dtype) == PSG_F16) {                               \
      using T = f16_t;                                             \
      __VA_ARGS__;                                                 \
    } else {                                                       \
      psg_set_error("%s: unknown dtype %d", NAME, (int)(dtype));   \
      return PSG_ERR_INVALID;                                      \
    }                                                              \
  } while (0)
