// Dense 16-bit GEMM with fused epilogues for the relation Q-Former (HF-IB:519-596, 664-672):
//
//   out[M][N] = epilogue( x[M][K] . w[N][K]^T + bias[N] )        epilogue: none | exact-erf GELU
//
// x, w, out row-major with K / N contiguous (w is an nn.Linear weight as stored), bf16 or fp16, fp32 accumulate.
// Replaces `F.linear` + `psg_bias_gelu` (two passes over the 82.5 k x 3072 FFN intermediate) for the Q-Former's
// intermediate projections; the library GEMM stays the default wherever this kernel is not faster.
//
// Structure (gfx950): 256 x 256 x 64 tiles, 8 waves as 2 (M) x 4 (N) -> 128 x 64 per wave = 4 x 2 accumulator tiles of
// v_mfma_f32_32x32x16 (32 MFMAs per K step and wave); persistent workgroups (one per CU) walk the output tiles.
//   * both operand tiles go L2 -> LDS by global_load_lds_dwordx4 into a double buffer (128 KiB), 8 rows x 128 B per
//     instruction, XOR-swizzled on the SOURCE address (the LDS image of a DMA is lane-linear) with (row >> 1) & 7,
//     which makes the ds_read_b128 fragment reads of the 32-row MFMA layout conflict-free;
//   * the 8 DMAs of the next K tile are issued beside the first two matrix sub-steps (not as a burst), waited for at
//     the top of the next K step; the first K tile of the NEXT output tile is requested during the last K step, so
//     its latency sits behind the epilogue;
//   * fragment reads are inline asm (hipcc puts s_waitcnt vmcnt(0) before every ds_read it emits itself while an
//     LDS-DMA is pending) and run one sub-step ahead of the MFMAs; one barrier per K step;
//   * the MFMA operands are swapped (D^T = W . X^T): a lane holds 4-column chunks of one output ROW, the two
//     half-waves alternate chunks, and one v_permlane32_swap per chunk pair gives 16-byte stores;
//   * workgroups are numbered so that the column tiles of one row block run on the same XCD (its L2 serves the x
//     tile to all of them).
// Measured at the Q-Former's shapes (tools/dense_gemm_bench.py): 0.8-1.0 PFLOP/s against 0.85-1.08 for the library
// kernel; with the GELU epilogue it replaces library GEMM + psg_bias_gelu (550 vs 620 us at 82.5 k x 3072 x 768).
// Ablation builds (option dense_gemm_var): staging alone sustains 10.5 TB/s L2 -> LDS (= 1.34 PFLOP/s at this
// tile's 128 FLOP/B), the matrix phase alone 1.24 PFLOP/s; walking K from a tile-dependent offset (to de-phase
// workgroups that share a panel) was measured slower: the shared bursts are L2 hits.
#include "psg_common.h"

#define DG_BM 256
#define DG_BN 256
#define DG_BK 64

// exact-erf GELU, Abramowitz-Stegun 7.1.26 (same arithmetic as bias_gelu_rows_bf16_kernel in psg_rowops.hip; the
// cheaper 7.1.28 form - one transcendental instead of two - measured the same epilogue time)
__device__ __forceinline__ float dg_gelu(float v) {
  const float x = v * 0.70710678118654752440f;
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  const float erf_abs = fmaf(-poly * t, e, 1.0f);
  return 0.5f * v * (1.0f + copysignf(erf_abs, x));
}

template <int N_>
__device__ __forceinline__ void dg_vmwait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}
__device__ __forceinline__ void dg_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Fragment reads are inline asm: next to a pending LDS-DMA hipcc makes every ds_read it generates itself wait
// vmcnt(0) first (the DMA is an LDS write that may alias), which would serialise the prefetch behind the reads.
typedef uint32_t dg_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dg_u32x4 dg_lds_read128(uint32_t a) {
  dg_u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory");
  return v;
}

// VAR (ablation builds): 0 normal, 1 no MFMA, 2 no staging after the first K tile.  OUT32: fp32 output
// out32[m][n] = epilogue(acc * row_scale[m] * col_scale[n] + bias[n]) - the split-fp16 products of the fp32s mode
// (psg_split.hip: the scales are the powers of two that undo the operands' row scaling).
// Every output element is ONE k-ordered accumulation over the whole K (no split-K, tiles walk K front to back), so a
// row's result does not depend on M or on the tile it falls into: a pair shard reproduces the rows of the
// full pass bit for bit.
template <typename E, int GELU, int VAR, int OUT32 = 0>
__global__ void __launch_bounds__(512, 2)
dense_gemm_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, const float* __restrict__ bias,
                  uint16_t* __restrict__ out, int M, int N, int K, const float* __restrict__ row_scale,
                  const float* __restrict__ col_scale) {
  using v8 = typename E::v8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2 buffers][A 32 KiB | B 32 KiB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int NB = N / DG_BN, MB = (M + DG_BM - 1) / DG_BM;
  const int MB8 = MB >= 8 ? (MB + 7) / 8 * 8 : MB;         // fewer than 8 row blocks: plain numbering (no idle XCD slots)
  const int ntile = MB8 * NB;
  const int wm = wid >> 2, wn = wid & 3;                   // wave tile: rows [128 wm, +128), cols [64 wn, +64)
  // persistent workgroups (one per CU) walk tiles b, b + grid, ...  XCD-aware numbering: tile t lives on XCD
  // t % 8 (= the XCD of its workgroup as long as the grid is a multiple of 8); the NB column tiles of a row block
  // share an XCD, whose L2 then serves the x tile to all of them
  auto tile_mn = [&](int t, int& mb, int& nb) {
    if (MB < 8) {                                          // row blocks innermost: the MB tiles of a column block
      mb = t % MB;                                         // are neighbours (they share the w tile through L2)
      nb = t / MB;
      return;
    }
    const int xcd = t & 7, idx = t >> 3;
    mb = (idx / NB) * 8 + xcd;
    nb = idx % NB;
  };

  // staging: one instruction = 8 rows x 128 B; lane -> row 8 g + (lane >> 3), 16-byte slot lane & 7, which holds
  // source piece slot ^ swz(row), swz(row) = (row >> 1) & 7 (conflict-free for the 32 x 32 x 16 fragment reads: a
  // ds_read_b128 lane group sees 16 rows whose even / odd members get 8 distinct slots each).  A 256 x 64 tile =
  // 32 instructions, 4 per wave; rows past M are clamped (never stored)
  const int srow = lane >> 3, sslot = lane & 7;
  auto stage_x = [&](int m0, int kt, int buf) {
    unsigned char* ab = smem + buf * 65536;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (wid * 4 + i) * 8 + srow;               // 0..255
      const int piece = sslot ^ ((r >> 1) & 7);
      int gr = m0 + r;
      gr = gr < M ? gr : M - 1;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(x + (int64_t)gr * K + kt * DG_BK + piece * 8),
          (__attribute__((address_space(3))) void*)(ab + (wid * 4 + i) * 1024), 16, 0, 0);
    }
  };
  auto stage_w = [&](int n0, int kt, int buf) {
    unsigned char* ab = smem + buf * 65536;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (wid * 4 + i) * 8 + srow;
      const int piece = sslot ^ ((r >> 1) & 7);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(w + (int64_t)(n0 + r) * K + kt * DG_BK + piece * 8),
          (__attribute__((address_space(3))) void*)(ab + 32768 + (wid * 4 + i) * 1024), 16, 0, 0);
    }
  };

  const int l31 = lane & 31, hi = lane >> 5;
  const int nk = K / DG_BK;
  // first real tile of this workgroup
  int t = blockIdx.x, mb, nb;
  for (;; t += gridDim.x) {
    if (t >= ntile) return;
    tile_mn(t, mb, nb);
    if (mb < MB) break;
  }
  int par = 0;                                              // LDS buffer of the K tile about to be consumed
  stage_x(mb * DG_BM, 0, 0);
  stage_w(nb * DG_BN, 0, 0);
  // per-lane fragment row offsets (bytes) and swizzles: A rows 128 wm + 32 i + l31, B rows 64 wn + 32 j + l31
  uint32_t arow[4], brow[2], aswz[4], bswz[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wm * 128 + i * 32 + l31;
    arow[i] = (uint32_t)(r * 128);
    aswz[i] = (uint32_t)((r >> 1) & 7);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = wn * 64 + j * 32 + l31;
    brow[j] = (uint32_t)(32768 + r * 128);
    bswz[j] = (uint32_t)((r >> 1) & 7);
  }
  union Frag {
    dg_u32x4 u;
    v8 v;
  };
  for (;;) {
    const int m0 = mb * DG_BM, n0 = nb * DG_BN;
    // next real tile (its first K tile is requested during this tile's last K step: in flight during the epilogue)
    int tn = t + gridDim.x, mbn = 0, nbn = 0;
    for (; tn < ntile; tn += gridDim.x) {
      tile_mn(tn, mbn, nbn);
      if (mbn < MB) break;
    }
    const bool has_next = tn < ntile;

    psg_f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = (psg_f32x16){0};

    for (int kt = 0; kt < nk; ++kt) {
      const int buf = par;
      par ^= 1;
      const bool more_k = kt + 1 < nk;
      const bool pf = VAR != 2 && (more_k || has_next);
      const int pm0 = more_k ? m0 : mbn * DG_BM, pn0 = more_k ? n0 : nbn * DG_BN, pkt = more_k ? kt + 1 : 0;
      dg_vmwait<0>();                                       // tile kt landed (requested during the previous K step)
      dg_lds_barrier();                                     // every wave's part of tile kt is in LDS; nobody reads buf ^ 1 any more
      const uint32_t base = smem_lds + (uint32_t)(buf * 65536);
      // four sub-steps of 16 in k, 8 MFMAs (32 x 32 x 16) each; the 6 fragment reads of sub-step s+1 are issued
      // before the MFMAs of sub-step s; the DMAs of the next K tile are issued beside sub-steps 0 and 1
      Frag af[2][4], bf[2][2];
      auto read_frags = [&](int sub, Frag (&a_)[4], Frag (&b_)[2]) {
        const uint32_t piece = (uint32_t)(2 * sub + hi);    // 16-byte piece (8 elements) of the 128-byte row
#pragma unroll
        for (int j = 0; j < 2; ++j) b_[j].u = dg_lds_read128(base + brow[j] + ((piece ^ bswz[j]) << 4));
#pragma unroll
        for (int i = 0; i < 4; ++i) a_[i].u = dg_lds_read128(base + arow[i] + ((piece ^ aswz[i]) << 4));
      };
#define DG_MMA(AF, BF)                                                                            \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) {    \
    if (VAR == 1) {                                                                               \
      asm volatile("" ::"v"(BF[j].u), "v"(AF[i].u));                                              \
    } else {                                                                                      \
      acc[i][j] = E::mfma32(BF[j].v, AF[i].v, acc[i][j]);   /* D[n][m]: swapped operands */       \
    }                                                                                             \
  }
      read_frags(0, af[0], bf[0]);
      read_frags(1, af[1], bf[1]);
      if (pf) stage_x(pm0, pkt, buf ^ 1);
      asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      DG_MMA(af[0], bf[0])
      __builtin_amdgcn_sched_barrier(0);
      read_frags(2, af[0], bf[0]);
      if (pf) stage_w(pn0, pkt, buf ^ 1);
      asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      DG_MMA(af[1], bf[1])
      __builtin_amdgcn_sched_barrier(0);
      read_frags(3, af[1], bf[1]);
      asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      DG_MMA(af[0], bf[0])
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      DG_MMA(af[1], bf[1])
      __builtin_amdgcn_s_setprio(0);
#undef DG_MMA
      // no barrier here: this buffer is refilled by DMAs that are issued after the NEXT K step's barrier, which
      // every wave reaches only after its own lgkmcnt(0) above, i.e. after its last read of this buffer
    }

    // epilogue.  acc[i][j][reg] = C[m][n] with m = m0 + 128 wm + 32 i + l31 and
    // n = n0 + 64 wn + 32 j + 8 (reg >> 2) + 4 hi + (reg & 3): the two half-waves hold alternating 4-column chunks of
    // a row, so one v_permlane32_swap per pair of chunks gives every lane 8 consecutive columns = one 16-byte store
    // (lanes 0-31: columns 16 q .. +7, lanes 32-63: columns 16 q + 8 .. +15)
    float4 bv[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        bv[j][q] = bias ? *reinterpret_cast<const float4*>(bias + n0 + wn * 64 + j * 32 + 8 * q + 4 * hi)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (OUT32) {
      float* out32 = reinterpret_cast<float*>(out);
      float4 cs[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          cs[j][q] = col_scale ? *reinterpret_cast<const float4*>(col_scale + n0 + wn * 64 + j * 32 + 8 * q + 4 * hi)
                               : make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 128 + i * 32 + l31;
        const float rs = (row_scale && m < M) ? row_scale[m] : 1.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float v[4] = {acc[i][j][4 * q] * (rs * cs[j][q].x) + bv[j][q].x, acc[i][j][4 * q + 1] * (rs * cs[j][q].y) + bv[j][q].y,
                          acc[i][j][4 * q + 2] * (rs * cs[j][q].z) + bv[j][q].z, acc[i][j][4 * q + 3] * (rs * cs[j][q].w) + bv[j][q].w};
            if (GELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = dg_gelu(v[e]);
            }
            if (m < M)
              *reinterpret_cast<float4*>(out32 + (int64_t)m * N + n0 + wn * 64 + j * 32 + 8 * q + 4 * hi) =
                  make_float4(v[0], v[1], v[2], v[3]);
          }
      }
    } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 128 + i * 32 + l31;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        uint32_t pk[4][2];                                   // [chunk q][2 words] = 4 columns at 8 q + 4 hi
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4] = {acc[i][j][4 * q] + bv[j][q].x, acc[i][j][4 * q + 1] + bv[j][q].y,
                        acc[i][j][4 * q + 2] + bv[j][q].z, acc[i][j][4 * q + 3] + bv[j][q].w};
          if (GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = dg_gelu(v[e]);
          }
          pk[q][0] = (uint32_t)E::from_f32(v[0]) | ((uint32_t)E::from_f32(v[1]) << 16);
          pk[q][1] = (uint32_t)E::from_f32(v[2]) | ((uint32_t)E::from_f32(v[3]) << 16);
        }
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {                     // chunk pair (2 q2, 2 q2 + 1) = columns 16 q2 .. + 15
          uint32_t a0 = pk[2 * q2][0], a1 = pk[2 * q2][1], b0 = pk[2 * q2 + 1][0], b1 = pk[2 * q2 + 1][1];
          // vdst = chunk 2 q2, src = chunk 2 q2 + 1: upper half of vdst <-> lower half of src
          auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
          auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
          // lanes 0-31 now hold [own chunk 2q2 | upper's chunk 2q2] = columns 16 q2 + 0..7;
          // lanes 32-63 hold [lower's chunk 2q2+1 | own chunk 2q2+1] = columns 16 q2 + 8..15
          if (m < M) {
            const int n = n0 + wn * 64 + j * 32 + 16 * q2 + 8 * hi;
            *reinterpret_cast<uint4*>(out + (int64_t)m * N + n) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
          }
        }
      }
    }
    }
    if (!has_next) return;
    t = tn;
    mb = mbn;
    nb = nbn;
  }
}

extern "C" int psg_dense_gemm_ex(psg_ctx* ctx, const void* x, const void* w, const float* bias, int epilogue, void* out,
                                 int64_t M, int N, int K, int dtype, int out_dtype, const float* row_scale,
                                 const float* col_scale, void* stream);
extern "C" int psg_dense_gemm(psg_ctx* ctx, const void* x, const void* w, const float* bias, int epilogue, void* out,
                              int64_t M, int N, int K, int dtype, void* stream) {
  return psg_dense_gemm_ex(ctx, x, w, bias, epilogue, out, M, N, K, dtype, dtype, nullptr, nullptr, stream);
}

extern "C" int psg_dense_gemm_ex(psg_ctx* ctx, const void* x, const void* w, const float* bias, int epilogue, void* out,
                                 int64_t M, int N, int K, int dtype, int out_dtype, const float* row_scale,
                                 const float* col_scale, void* stream) {
  PSG_REQUIRE(ctx && x && w && out, PSG_ERR_INVALID, "psg_dense_gemm: NULL argument");
  PSG_REQUIRE(out_dtype == dtype || out_dtype == PSG_F32, PSG_ERR_UNSUPPORTED,
              "psg_dense_gemm: output dtype %d (the operand dtype %d or fp32)", out_dtype, dtype);
  PSG_REQUIRE(out_dtype == PSG_F32 || (!row_scale && !col_scale), PSG_ERR_UNSUPPORTED,
              "psg_dense_gemm: row / column scales need the fp32 output");
  PSG_REQUIRE(M >= 0 && N > 0 && K > 0 && M < (1ll << 31), PSG_ERR_INVALID, "psg_dense_gemm: M=%lld N=%d K=%d",
              (long long)M, N, K);
  PSG_REQUIRE(N % DG_BN == 0 && K % DG_BK == 0, PSG_ERR_UNSUPPORTED,
              "psg_dense_gemm: N=%d must be a multiple of %d and K=%d of %d", N, DG_BN, K, DG_BK);
  PSG_REQUIRE(epilogue == PSG_EPI_NONE || epilogue == PSG_EPI_GELU, PSG_ERR_INVALID, "psg_dense_gemm: epilogue=%d",
              epilogue);
  if (M == 0) return PSG_OK;
  const int NB = N / DG_BN, MB = (int)((M + DG_BM - 1) / DG_BM);
  const int MB8 = MB >= 8 ? (MB + 7) / 8 * 8 : MB;
  int grid_i = ctx->num_cu / 8 * 8;                          // persistent: one workgroup per CU, a multiple of 8 (XCD map)
  if (grid_i > MB8 * NB) grid_i = MB8 * NB;
  const unsigned grid = (unsigned)grid_i;
  const size_t lds = 2 * 65536;
#define DGL(G, V, O)                                                                                                 \
  do {                                                                                                              \
    hipError_t e = hipFuncSetAttribute((const void*)dense_gemm_kernel<E, G, V, O>,                                     \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
    if (e != hipSuccess) {                                                                                          \
      psg_set_error("psg_dense_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));                               \
      return PSG_ERR_HIP;                                                                                           \
    }                                                                                                               \
    dense_gemm_kernel<E, G, V, O><<<grid, 512, lds, (hipStream_t)stream>>>(                                            \
        (const uint16_t*)x, (const uint16_t*)w, bias, (uint16_t*)out, (int)M, N, K, row_scale, col_scale);          \
  } while (0)
  const int var = ctx->opt.dense_gemm_var;
  if (out_dtype == PSG_F32 && dtype != PSG_F32) {
    PSG_DISPATCH_E16(dtype, "psg_dense_gemm", if (epilogue == PSG_EPI_GELU) DGL(1, 0, 1); else DGL(0, 0, 1));
  } else {
    PSG_DISPATCH_E16(dtype, "psg_dense_gemm",
                     if (var == 1) DGL(0, 1, 0); else if (var == 2) DGL(0, 2, 0); else if (epilogue == PSG_EPI_GELU) DGL(1, 0, 0);
                     else DGL(0, 0, 0));
  }
#undef DGL
  PSG_CHECK_LAUNCH("psg_dense_gemm");
  return PSG_OK;
}
