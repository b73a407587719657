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
template <int N_>
__device__ __forceinline__ void dg_lgkmwait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N_) : "memory");
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
// full pass bit for bit - whatever the tile geometry.
//
// Tile geometry <WM, WN, TI, TJ>: WM x WN waves, each TI x TJ accumulator tiles of 32 x 32 ->
// BM = 32 WM TI rows of x by BN = 32 WN TJ rows of w.  256 x 256 (2, 4, 4, 2) is the Q-Former's tile; the Llama prompt
// pass (M ~ 980 rows = 4 row blocks, HF-LL:163-177) picks the width per shape so that the tiles fill the 256 CUs in
// whole rounds (psg_dense_gemm_tiled).  N need not be a multiple of BN: w rows past N are clamped in the staging and
// never stored.
//
// EPI = PSG_EPI_SWIGLU (Llama MLP, HF-LL:163-177): w holds gate and up rows interleaved in groups of 8
// (w[16 p + r] = gate[8 p + r], w[16 p + 8 + r] = up[8 p + r], r < 8: psg_interleave_gate_up), so a lane's accumulator
// chunks alternate gate / up of the SAME 4 columns; out[m][8 p + r] = silu(gate) * up, N / 2 columns, with the
// roundings of the separate kernels (GEMM output, act_fn(gate), product: psg_silu_mul).
//
// SPL (psg_dense_gemm_split, fp32s mode): BOTH operands are psg_split_f16x3(order 2) images of fp32 matrices - per 32 k
// [hi(32) | lo(32)], 2K elements per row - so a 64-element K tile is one 32-wide block of the original K with its high
// parts in 16-byte pieces 0..3 and its low parts in pieces 4..7.  The tile is staged ONCE and feeds three products,
//     acc += xh.wh + xh.wl + xl.wh        (6 MFMA groups per K tile: A pieces (0,1) x B (0,1), A (0,1) x B (2,3), A (2,3) x B (0,1))
// against the K' = 3K form's [xh | xh | xl] . [wh | wl | wh]^T, which stages xh and wh twice: 3 products from 64 KB of
// operand tiles instead of 2 - this kernel waits for its operand tiles, not for the matrix pipe (see above), so that is
// its speed.  Still ONE k-ordered accumulation per output element (per block: hh, hl, lh), independent of M and tile.
template <typename E, int EPI, int VAR, int OUT32, int WM, int WN, int TI, int TJ, int SPL = 0>
__global__ void __launch_bounds__(WM * WN * 64, (WM * WN >= 8 ? 2 : 1))
dense_gemm_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, const float* __restrict__ bias,
                  uint16_t* __restrict__ out, int M, int N, int K, const float* __restrict__ row_scale,
                  const float* __restrict__ col_scale) {
  using v8 = typename E::v8;
  constexpr int NW = WM * WN, BM = WM * TI * 32, BN = WN * TJ * 32;
  constexpr int A_BYTES = BM * 128, BUF_BYTES = (BM + BN) * 128;        // one K tile: [A: BM rows | B: BN rows] x 128 B
  constexpr int AI = BM / 8 / NW, BI = BN / 8 / NW;                      // staging instructions per wave
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split into 8-row DMA instructions per wave");
  constexpr bool GELU = EPI == PSG_EPI_GELU, SWIGLU = EPI == PSG_EPI_SWIGLU;
  static_assert(!(SWIGLU && OUT32), "the SwiGLU epilogue writes the 16-bit activation");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2 buffers][A | B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int NB = (N + BN - 1) / BN, MB = (M + BM - 1) / BM;
  const int MB8 = (MB + 7) / 8 * 8, NB8 = (NB + 7) / 8 * 8;
  const int ntile = MB >= 8 ? MB8 * NB : MB * NB8;
  const int wm = wid / WN, wn = wid % WN;                  // wave tile: rows [32 TI wm, +32 TI), cols [32 TJ wn, +32 TJ)
  // persistent workgroups (one per CU) walk tiles b, b + grid, ...  XCD-aware numbering: tile t lives on XCD
  // t % 8 (= the XCD of its workgroup as long as the grid is a multiple of 8).  8 or more row blocks: the NB column
  // tiles of a row block share an XCD, whose L2 then serves the x tile to all of them.  Fewer (the prompt pass): the
  // MB row blocks of a column block share an XCD and run side by side - the w tile comes from HBM once.
  auto tile_mn = [&](int t, int& mb, int& nb) {
    const int xcd = t & 7, idx = t >> 3;
    if (MB < 8) {
      mb = idx % MB;
      nb = (idx / MB) * 8 + xcd;
      return;
    }
    mb = (idx / NB) * 8 + xcd;
    nb = idx % NB;
  };

  // staging: one instruction = 8 rows x 128 B; lane -> row 8 g + (lane >> 3), 16-byte slot lane & 7, which holds
  // source piece slot ^ swz(row), swz(row) = (row >> 1) & 7 (conflict-free for the 32 x 32 x 16 fragment reads: a
  // ds_read_b128 lane group sees 16 rows whose even / odd members get 8 distinct slots each).  Rows past M / N are
  // clamped (never stored)
  const int srow = lane >> 3, sslot = lane & 7;
  auto stage_x = [&](int m0, int kt, int buf) {
    unsigned char* ab = smem + buf * BUF_BYTES;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int r = (wid * AI + i) * 8 + srow;
      const int piece = sslot ^ ((r >> 1) & 7);
      int gr = m0 + r;
      gr = gr < M ? gr : M - 1;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(x + (int64_t)gr * K + kt * DG_BK + piece * 8),
          (__attribute__((address_space(3))) void*)(ab + (wid * AI + i) * 1024), 16, 0, 0);
    }
  };
  auto stage_w = [&](int n0, int kt, int buf) {
    unsigned char* ab = smem + buf * BUF_BYTES + A_BYTES;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int r = (wid * BI + i) * 8 + srow;
      const int piece = sslot ^ ((r >> 1) & 7);
      int gr = n0 + r;
      gr = gr < N ? gr : N - 1;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(w + (int64_t)gr * K + kt * DG_BK + piece * 8),
          (__attribute__((address_space(3))) void*)(ab + (wid * BI + i) * 1024), 16, 0, 0);
    }
  };

  const int l31 = lane & 31, hi = lane >> 5;
  const int nk = K / DG_BK;
  // first real tile of this workgroup
  int t = blockIdx.x, mb, nb;
  for (;; t += gridDim.x) {
    if (t >= ntile) return;
    tile_mn(t, mb, nb);
    if (mb < MB && nb < NB) break;
  }
  int par = 0;                                              // LDS buffer of the K tile about to be consumed
  stage_x(mb * BM, 0, 0);
  stage_w(nb * BN, 0, 0);
  // per-lane fragment row offsets (bytes) and swizzles: A rows 32 TI wm + 32 i + l31, B rows 32 TJ wn + 32 j + l31
  uint32_t arow[TI], brow[TJ], aswz[TI], bswz[TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int r = wm * (TI * 32) + i * 32 + l31;
    arow[i] = (uint32_t)(r * 128);
    aswz[i] = (uint32_t)((r >> 1) & 7);
  }
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int r = wn * (TJ * 32) + j * 32 + l31;
    brow[j] = (uint32_t)(A_BYTES + r * 128);
    bswz[j] = (uint32_t)((r >> 1) & 7);
  }
  union Frag {
    dg_u32x4 u;
    v8 v;
  };
  for (;;) {
    const int m0 = mb * BM, n0 = nb * BN;
    // next real tile (its first K tile is requested during this tile's last K step: in flight during the epilogue)
    int tn = t + gridDim.x, mbn = 0, nbn = 0;
    for (; tn < ntile; tn += gridDim.x) {
      tile_mn(tn, mbn, nbn);
      if (mbn < MB && nbn < NB) break;
    }
    const bool has_next = tn < ntile;

    psg_f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) acc[i][j] = (psg_f32x16){0};

    for (int kt = 0; kt < nk; ++kt) {
      const int buf = par;
      par ^= 1;
      const bool more_k = kt + 1 < nk;
      const bool pf = VAR != 2 && (more_k || has_next);
      const int pm0 = more_k ? m0 : mbn * BM, pn0 = more_k ? n0 : nbn * BN, pkt = more_k ? kt + 1 : 0;
      dg_vmwait<0>();                                       // tile kt landed (requested during the previous K step)
      dg_lds_barrier();                                     // every wave's part of tile kt is in LDS; nobody reads buf ^ 1 any more
      const uint32_t base = smem_lds + (uint32_t)(buf * BUF_BYTES);
      // four sub-steps of 16 in k, TI x TJ MFMAs (32 x 32 x 16) each; the TI + TJ fragment reads of sub-step s+1 are
      // issued before the MFMAs of sub-step s; the DMAs of the next K tile are issued beside sub-steps 0 and 1
      Frag af[2][TI], bf[2][TJ];
      auto read_frags = [&](int sub, Frag (&a_)[TI], Frag (&b_)[TJ]) {
        const uint32_t piece = (uint32_t)(2 * sub + hi);    // 16-byte piece (8 elements) of the 128-byte row
#pragma unroll
        for (int j = 0; j < TJ; ++j) b_[j].u = dg_lds_read128(base + brow[j] + ((piece ^ bswz[j]) << 4));
#pragma unroll
        for (int i = 0; i < TI; ++i) a_[i].u = dg_lds_read128(base + arow[i] + ((piece ^ aswz[i]) << 4));
      };
#define DG_MMA(AF, BF)                                                                            \
  _Pragma("unroll") for (int i = 0; i < TI; ++i) _Pragma("unroll") for (int j = 0; j < TJ; ++j) {  \
    if (VAR == 1) {                                                                               \
      asm volatile("" ::"v"(BF[j].u), "v"(AF[i].u));                                              \
    } else {                                                                                      \
      acc[i][j] = E::mfma32(BF[j].v, AF[i].v, acc[i][j]);   /* D[n][m]: swapped operands */       \
    }                                                                                             \
  }
      if constexpr (SPL) {
        // pieces 2 s + hi: s = 0, 1 high parts (k 0..15, 16..31 of the block), s = 2, 3 low parts.  Reads return in order
        Frag bl[2][TJ];
        auto read_a = [&](int sub, Frag (&a_)[TI]) {
          const uint32_t piece = (uint32_t)(2 * sub + hi);
#pragma unroll
          for (int i = 0; i < TI; ++i) a_[i].u = dg_lds_read128(base + arow[i] + ((piece ^ aswz[i]) << 4));
        };
        auto read_b = [&](int sub, Frag (&b_)[TJ]) {
          const uint32_t piece = (uint32_t)(2 * sub + hi);
#pragma unroll
          for (int j = 0; j < TJ; ++j) b_[j].u = dg_lds_read128(base + brow[j] + ((piece ^ bswz[j]) << 4));
        };
        read_frags(0, af[0], bf[0]);
        read_frags(1, af[1], bf[1]);
        if (pf) stage_x(pm0, pkt, buf ^ 1);
        dg_lgkmwait<TI + TJ>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        DG_MMA(af[0], bf[0])                                  // xh . wh, k 0..15
        __builtin_amdgcn_sched_barrier(0);
        read_b(2, bl[0]);
        if (pf) stage_w(pn0, pkt, buf ^ 1);
        dg_lgkmwait<TJ>();
        __builtin_amdgcn_sched_barrier(0);
        DG_MMA(af[1], bf[1])                                  // xh . wh, k 16..31
        __builtin_amdgcn_sched_barrier(0);
        read_b(3, bl[1]);
        dg_lgkmwait<TJ>();
        __builtin_amdgcn_sched_barrier(0);
        DG_MMA(af[0], bl[0])                                  // xh . wl, k 0..15
        __builtin_amdgcn_sched_barrier(0);
        read_a(2, af[0]);
        dg_lgkmwait<TI>();
        __builtin_amdgcn_sched_barrier(0);
        DG_MMA(af[1], bl[1])                                  // xh . wl, k 16..31
        __builtin_amdgcn_sched_barrier(0);
        read_a(3, af[1]);
        dg_lgkmwait<TI>();
        __builtin_amdgcn_sched_barrier(0);
        DG_MMA(af[0], bf[0])                                  // xl . wh, k 0..15
        dg_lgkmwait<0>();
        __builtin_amdgcn_sched_barrier(0);
        DG_MMA(af[1], bf[1])                                  // xl . wh, k 16..31
        __builtin_amdgcn_s_setprio(0);
      } else {
      read_frags(0, af[0], bf[0]);
      read_frags(1, af[1], bf[1]);
      if (pf) stage_x(pm0, pkt, buf ^ 1);
      dg_lgkmwait<TI + TJ>();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      DG_MMA(af[0], bf[0])
      __builtin_amdgcn_sched_barrier(0);
      read_frags(2, af[0], bf[0]);
      if (pf) stage_w(pn0, pkt, buf ^ 1);
      dg_lgkmwait<TI + TJ>();
      __builtin_amdgcn_sched_barrier(0);
      DG_MMA(af[1], bf[1])
      __builtin_amdgcn_sched_barrier(0);
      read_frags(3, af[1], bf[1]);
      dg_lgkmwait<TI + TJ>();
      __builtin_amdgcn_sched_barrier(0);
      DG_MMA(af[0], bf[0])
      dg_lgkmwait<0>();
      __builtin_amdgcn_sched_barrier(0);
      DG_MMA(af[1], bf[1])
      __builtin_amdgcn_s_setprio(0);
      }
#undef DG_MMA
      // no barrier here: this buffer is refilled by DMAs that are issued after the NEXT K step's barrier, which
      // every wave reaches only after its own lgkmcnt(0) above, i.e. after its last read of this buffer
    }

    // epilogue.  acc[i][j][reg] = C[m][n] with m = m0 + 32 TI wm + 32 i + l31 and
    // n = n0 + 32 TJ wn + 32 j + 8 (reg >> 2) + 4 hi + (reg & 3): the two half-waves hold alternating 4-column chunks of
    // a row, so one v_permlane32_swap per pair of chunks gives every lane 8 consecutive columns = one 16-byte store
    // (lanes 0-31: columns 16 q .. +7, lanes 32-63: columns 16 q + 8 .. +15)
    const int cw = n0 + wn * (TJ * 32);                      // first column of the wave tile
    if constexpr (SWIGLU) {
      const int No = N >> 1;
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const int m = m0 + wm * (TI * 32) + i * 32 + l31;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          uint32_t pk[2][2];                                 // [16-group p][2 words] = 4 output columns at 8 p + 4 hi
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            uint16_t h[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float g = E::to_f32(E::from_f32(acc[i][j][8 * p + e]));         // the GEMM's 16-bit output
              const float u = E::to_f32(E::from_f32(acc[i][j][8 * p + 4 + e]));
              const float sg = E::to_f32(E::from_f32(g / (1.0f + expf(-g))));       // HF rounds act_fn(gate)
              h[e] = E::from_f32(sg * u);
            }
            pk[p][0] = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
            pk[p][1] = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
          }
          auto r0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
          auto r1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
          const int nin = cw + j * 32 + 16 * hi;             // the 16 interleaved columns this lane's 8 outputs come from
          if (m < M && nin + 16 <= N)
            *reinterpret_cast<uint4*>(out + (int64_t)m * No + (nin >> 1)) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
        }
      }
    } else {
    float4 bv[TJ][4];
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = cw + j * 32 + 8 * q + 4 * hi;
        bv[j][q] = (bias && n + 4 <= N) ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    if constexpr (OUT32) {
      float* out32 = reinterpret_cast<float*>(out);
      float4 cs[TJ][4];
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = cw + j * 32 + 8 * q + 4 * hi;
          cs[j][q] = (col_scale && n + 4 <= N) ? *reinterpret_cast<const float4*>(col_scale + n)
                                               : make_float4(1.f, 1.f, 1.f, 1.f);
        }
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const int m = m0 + wm * (TI * 32) + i * 32 + l31;
        const float rs = (row_scale && m < M) ? row_scale[m] : 1.f;
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float v[4] = {acc[i][j][4 * q] * (rs * cs[j][q].x) + bv[j][q].x, acc[i][j][4 * q + 1] * (rs * cs[j][q].y) + bv[j][q].y,
                          acc[i][j][4 * q + 2] * (rs * cs[j][q].z) + bv[j][q].z, acc[i][j][4 * q + 3] * (rs * cs[j][q].w) + bv[j][q].w};
            if (GELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = dg_gelu(v[e]);
            }
            const int n = cw + j * 32 + 8 * q + 4 * hi;
            if (m < M && n + 4 <= N)
              *reinterpret_cast<float4*>(out32 + (int64_t)m * N + n) = make_float4(v[0], v[1], v[2], v[3]);
          }
      }
    } else {
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const int m = m0 + wm * (TI * 32) + i * 32 + l31;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
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
          const int n = cw + j * 32 + 16 * q2 + 8 * hi;
          if (m < M && n + 8 <= N)
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

// The built tile geometries (enum psg_tile -> waves and 32 x 32 accumulator tiles per wave); a short list on purpose:
// every entry is instantiated for two element types and two epilogues
struct dg_geom {
  int id, wm, wn, ti, tj;
};
static const dg_geom DG_GEOMS[] = {
    {PSG_TILE_256x256, 2, 4, 4, 2}, {PSG_TILE_256x192, 4, 2, 2, 3}, {PSG_TILE_256x128, 4, 2, 2, 2},
    {PSG_TILE_256x64, 8, 1, 1, 2},  {PSG_TILE_128x128, 2, 2, 2, 2},
};

template <typename E, int EPI, int VAR, int OUT32, int WM, int WN, int TI, int TJ, int SPL = 0>
static int dg_launch(psg_ctx* ctx, const void* x, const void* w, const float* bias, void* out, int64_t M, int N, int K,
                     const float* row_scale, const float* col_scale, void* stream) {
  constexpr int BM = WM * TI * 32, BN = WN * TJ * 32;
  const int NB = (N + BN - 1) / BN, MB = (int)((M + BM - 1) / BM);
  const int ntile = MB >= 8 ? (MB + 7) / 8 * 8 * NB : MB * ((NB + 7) / 8 * 8);
  int grid_i = ctx->num_cu / 8 * 8;                          // persistent: one workgroup per CU, a multiple of 8 (XCD map)
  if (grid_i > ntile) grid_i = ntile;
  const size_t lds = 2 * (size_t)(BM + BN) * 128;
  auto k = dense_gemm_kernel<E, EPI, VAR, OUT32, WM, WN, TI, TJ, SPL>;
  hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    psg_set_error("psg_dense_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
    return PSG_ERR_HIP;
  }
  k<<<(unsigned)grid_i, WM * WN * 64, lds, (hipStream_t)stream>>>((const uint16_t*)x, (const uint16_t*)w, bias,
                                                                   (uint16_t*)out, (int)M, N, K, row_scale, col_scale);
  PSG_CHECK_LAUNCH("psg_dense_gemm");
  return PSG_OK;
}

// tile that fills the CUs best for [M, N]: cost = rounds x tile area, ties to the larger tile (less operand traffic)
static int dg_auto_tile(const psg_ctx* ctx, int64_t M, int N, bool o32 = false) {
  int best = PSG_TILE_256x256;
  double best_cost = 1e300;
  const int cus = ctx->num_cu / 8 * 8;
  for (const dg_geom& g : DG_GEOMS) {
    if (g.id == PSG_TILE_128x128) continue;                  // 4-wave tile: explicit requests only
    if (o32 && g.id == PSG_TILE_256x192) continue;           // (not built with the fp32 output)
    const int BM = g.wm * g.ti * 32, BN = g.wn * g.tj * 32;
    const int64_t tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const int64_t rounds = (tiles + cus - 1) / cus;
    // narrow tiles read more LDS per MFMA: measured per-round time ~ area x {1, 1.04, 1.1, 1.5}
    const double pen = BN >= 256 ? 1.0 : BN >= 192 ? 1.04 : BN >= 128 ? 1.1 : 1.5;
    const double cost = (double)rounds * BM * BN * pen;
    if (cost < best_cost * 0.999) {
      best_cost = cost;
      best = g.id;
    }
  }
  return best;
}

extern "C" int psg_dense_gemm_tiled(psg_ctx* ctx, const void* x, const void* w, const float* bias, int epilogue,
                                    void* out, int64_t M, int N, int K, int dtype, int out_dtype, const float* row_scale,
                                    const float* col_scale, int tile, void* stream);
extern "C" int psg_dense_gemm_ex(psg_ctx* ctx, const void* x, const void* w, const float* bias, int epilogue, void* out,
                                 int64_t M, int N, int K, int dtype, int out_dtype, const float* row_scale,
                                 const float* col_scale, void* stream) {
  return psg_dense_gemm_tiled(ctx, x, w, bias, epilogue, out, M, N, K, dtype, out_dtype, row_scale, col_scale,
                              PSG_TILE_256x256, stream);
}
extern "C" int psg_dense_gemm(psg_ctx* ctx, const void* x, const void* w, const float* bias, int epilogue, void* out,
                              int64_t M, int N, int K, int dtype, void* stream) {
  return psg_dense_gemm_tiled(ctx, x, w, bias, epilogue, out, M, N, K, dtype, dtype, nullptr, nullptr,
                              PSG_TILE_256x256, stream);
}

extern "C" int psg_dense_gemm_tiled(psg_ctx* ctx, const void* x, const void* w, const float* bias, int epilogue,
                                    void* out, int64_t M, int N, int K, int dtype, int out_dtype, const float* row_scale,
                                    const float* col_scale, int tile, void* stream) {
  PSG_REQUIRE(ctx && x && w && out, PSG_ERR_INVALID, "psg_dense_gemm: NULL argument");
  PSG_REQUIRE(out_dtype == dtype || out_dtype == PSG_F32, PSG_ERR_UNSUPPORTED,
              "psg_dense_gemm: output dtype %d (the operand dtype %d or fp32)", out_dtype, dtype);
  PSG_REQUIRE(out_dtype == PSG_F32 || (!row_scale && !col_scale), PSG_ERR_UNSUPPORTED,
              "psg_dense_gemm: row / column scales need the fp32 output");
  PSG_REQUIRE(M >= 0 && N > 0 && K > 0 && M < (1ll << 31), PSG_ERR_INVALID, "psg_dense_gemm: M=%lld N=%d K=%d",
              (long long)M, N, K);
  PSG_REQUIRE(N % 16 == 0 && K % DG_BK == 0, PSG_ERR_UNSUPPORTED,
              "psg_dense_gemm: N=%d must be a multiple of 16 and K=%d of %d", N, K, DG_BK);
  PSG_REQUIRE(epilogue == PSG_EPI_NONE || epilogue == PSG_EPI_GELU || epilogue == PSG_EPI_SWIGLU, PSG_ERR_INVALID,
              "psg_dense_gemm: epilogue=%d", epilogue);
  PSG_REQUIRE(epilogue != PSG_EPI_SWIGLU || (out_dtype == dtype && !bias), PSG_ERR_UNSUPPORTED,
              "psg_dense_gemm: the SwiGLU epilogue takes no bias and writes the operand dtype");
  if (M == 0) return PSG_OK;
  if (tile == PSG_TILE_AUTO) tile = dg_auto_tile(ctx, M, N, out_dtype == PSG_F32 && dtype != PSG_F32);
  const int var = ctx->opt.dense_gemm_var;
  const bool o32 = out_dtype == PSG_F32 && dtype != PSG_F32;
  PSG_REQUIRE(tile == PSG_TILE_256x256 || var == 0, PSG_ERR_UNSUPPORTED, "psg_dense_gemm: ablation builds exist for the 256 x 256 tile");
  PSG_REQUIRE(tile == PSG_TILE_256x256 || epilogue != PSG_EPI_GELU || o32, PSG_ERR_UNSUPPORTED,
              "psg_dense_gemm: tile %d with the GELU epilogue is built for the fp32 output (the split products)", tile);
  PSG_REQUIRE(!(o32 && tile == PSG_TILE_256x192), PSG_ERR_UNSUPPORTED, "psg_dense_gemm: the 256 x 192 tile has no fp32 output");
#define DGL(EPI, V, O, WM, WN, TI, TJ) \
  return dg_launch<E, EPI, V, O, WM, WN, TI, TJ>(ctx, x, w, bias, out, M, N, K, row_scale, col_scale, stream)
#define DG_TILE(WM, WN, TI, TJ)                                                        \
  PSG_DISPATCH_E16(dtype, "psg_dense_gemm", if (epilogue == PSG_EPI_SWIGLU) DGL(PSG_EPI_SWIGLU, 0, 0, WM, WN, TI, TJ); \
                   else DGL(PSG_EPI_NONE, 0, 0, WM, WN, TI, TJ))
  // fp32 output (split products of the fp32s mode): the small-M projections of the Q-Former fill the CUs with smaller tiles
#define DG_TILE32(WM, WN, TI, TJ)                                                      \
  PSG_DISPATCH_E16(dtype, "psg_dense_gemm", if (epilogue == PSG_EPI_GELU) DGL(PSG_EPI_GELU, 0, 1, WM, WN, TI, TJ); \
                   else DGL(PSG_EPI_NONE, 0, 1, WM, WN, TI, TJ))
  switch (tile) {
    case PSG_TILE_256x256:
      if (o32) {
        PSG_REQUIRE(epilogue != PSG_EPI_SWIGLU, PSG_ERR_UNSUPPORTED, "psg_dense_gemm: SwiGLU with fp32 output");
        PSG_DISPATCH_E16(dtype, "psg_dense_gemm",
                         if (epilogue == PSG_EPI_GELU) DGL(PSG_EPI_GELU, 0, 1, 2, 4, 4, 2); else DGL(PSG_EPI_NONE, 0, 1, 2, 4, 4, 2));
      } else {
        PSG_DISPATCH_E16(dtype, "psg_dense_gemm",
                         if (var == 1) DGL(PSG_EPI_NONE, 1, 0, 2, 4, 4, 2); else if (var == 2) DGL(PSG_EPI_NONE, 2, 0, 2, 4, 4, 2);
                         else if (epilogue == PSG_EPI_GELU) DGL(PSG_EPI_GELU, 0, 0, 2, 4, 4, 2);
                         else if (epilogue == PSG_EPI_SWIGLU) DGL(PSG_EPI_SWIGLU, 0, 0, 2, 4, 4, 2);
                         else DGL(PSG_EPI_NONE, 0, 0, 2, 4, 4, 2));
      }
      break;
    case PSG_TILE_256x192: DG_TILE(4, 2, 2, 3); break;
    case PSG_TILE_256x128:
      if (o32) { PSG_REQUIRE(epilogue != PSG_EPI_SWIGLU, PSG_ERR_UNSUPPORTED, "psg_dense_gemm: SwiGLU with fp32 output"); DG_TILE32(4, 2, 2, 2); }
      else DG_TILE(4, 2, 2, 2);
      break;
    case PSG_TILE_256x64:
      if (o32) { PSG_REQUIRE(epilogue != PSG_EPI_SWIGLU, PSG_ERR_UNSUPPORTED, "psg_dense_gemm: SwiGLU with fp32 output"); DG_TILE32(8, 1, 1, 2); }
      else DG_TILE(8, 1, 1, 2);
      break;
    case PSG_TILE_128x128:
      if (o32) { PSG_REQUIRE(epilogue != PSG_EPI_SWIGLU, PSG_ERR_UNSUPPORTED, "psg_dense_gemm: SwiGLU with fp32 output"); DG_TILE32(2, 2, 2, 2); }
      else DG_TILE(2, 2, 2, 2);
      break;
    default: break;
  }
#undef DG_TILE
#undef DG_TILE32
#undef DGL
  psg_set_error("psg_dense_gemm: unknown tile %d", tile);
  return PSG_ERR_INVALID;
}

// ---- fp32-grade product of two fp32 matrices given as interleaved hi / lo fp16 images (psg_split_f16x3 order 2) ---------
extern "C" int psg_dense_gemm_split(psg_ctx* ctx, const void* x2, const void* w2, const float* bias, int epilogue, float* out,
                                    int64_t M, int N, int K2, const float* row_scale, const float* col_scale, int tile,
                                    void* stream) {
  PSG_REQUIRE(ctx && x2 && w2 && out && row_scale && col_scale, PSG_ERR_INVALID, "psg_dense_gemm_split: NULL argument");
  PSG_REQUIRE(M >= 0 && N > 0 && K2 > 0 && M < (1ll << 31), PSG_ERR_INVALID, "psg_dense_gemm_split: M=%lld N=%d K2=%d",
              (long long)M, N, K2);
  PSG_REQUIRE(N % 16 == 0 && K2 % DG_BK == 0, PSG_ERR_UNSUPPORTED,
              "psg_dense_gemm_split: N=%d must be a multiple of 16 and K2=%d (= 2 K) of %d", N, K2, DG_BK);
  PSG_REQUIRE(epilogue == PSG_EPI_NONE || epilogue == PSG_EPI_GELU, PSG_ERR_INVALID, "psg_dense_gemm_split: epilogue=%d",
              epilogue);
  if (M == 0) return PSG_OK;
  if (tile == PSG_TILE_AUTO) tile = dg_auto_tile(ctx, M, N, true);
  const void* x = x2;
  const void* w = w2;
  const int K = K2;
#define DGS(WM, WN, TI, TJ)                                                                                              \
  do {                                                                                                                   \
    if (epilogue == PSG_EPI_GELU)                                                                                        \
      return dg_launch<EF16, PSG_EPI_GELU, 0, 1, WM, WN, TI, TJ, 1>(ctx, x, w, bias, out, M, N, K, row_scale, col_scale, stream); \
    return dg_launch<EF16, PSG_EPI_NONE, 0, 1, WM, WN, TI, TJ, 1>(ctx, x, w, bias, out, M, N, K, row_scale, col_scale, stream);   \
  } while (0)
  switch (tile) {
    case PSG_TILE_256x256: DGS(2, 4, 4, 2);
    case PSG_TILE_256x128: DGS(4, 2, 2, 2);
    case PSG_TILE_256x64: DGS(8, 1, 1, 2);
    case PSG_TILE_128x128: DGS(2, 2, 2, 2);
    default: break;
  }
#undef DGS
  psg_set_error("psg_dense_gemm_split: tile %d (256x256, 256x128, 256x64, 128x128 or auto)", tile);
  return PSG_ERR_INVALID;
}

// ---- gate / up interleave for the SwiGLU epilogue ---------------------------------------------------------------------
// out[16 p + r] = gate_up[8 p + r], out[16 p + 8 + r] = gate_up[inter + 8 p + r] (r < 8): rows of K 16-bit elements
__global__ void __launch_bounds__(256) interleave_gate_up_kernel(const uint16_t* __restrict__ gu, uint16_t* __restrict__ out,
                                                                 int inter, int K) {
  const int orow = blockIdx.x;                               // 0 .. 2 inter
  const int p = orow >> 4, r = orow & 15;
  const int srow = r < 8 ? 8 * p + r : inter + 8 * p + (r - 8);
  const uint4* s = reinterpret_cast<const uint4*>(gu + (int64_t)srow * K);
  uint4* d = reinterpret_cast<uint4*>(out + (int64_t)orow * K);
  for (int c = threadIdx.x; c < K / 8; c += 256) d[c] = s[c];
}

extern "C" int psg_interleave_gate_up(psg_ctx* ctx, const void* gate_up, void* out, int inter, int K, void* stream) {
  PSG_REQUIRE(ctx && gate_up && out, PSG_ERR_INVALID, "psg_interleave_gate_up: NULL argument");
  PSG_REQUIRE(inter > 0 && inter % 8 == 0 && K > 0 && K % 8 == 0, PSG_ERR_UNSUPPORTED,
              "psg_interleave_gate_up: inter=%d and K=%d must be multiples of 8", inter, K);
  interleave_gate_up_kernel<<<2 * inter, 256, 0, (hipStream_t)stream>>>((const uint16_t*)gate_up, (uint16_t*)out, inter, K);
  PSG_CHECK_LAUNCH("psg_interleave_gate_up");
  return PSG_OK;
}
