// Decode-step projections for SEVERAL images' selected pairs decoded together (`head.forward_batch`, V4:112 lifted: 2 / 4 /
// 8 images = 40 / 80 / 160 decode rows): part[s][m][n] = split-K slices of x[M][K] . w[N][K]^T, 33 <= M <= 160, 16-bit
// operands, fp32 slices summed by the consumer row kernels exactly like psg_skinny_gemm's (psg_rmsnorm, psg_decode_attn,
// psg_silu_mul, psg_greedy_step).
//
// Why a kernel of its own.  psg_skinny_gemm keeps its x slice resident in LDS (<= 32 rows); at 160 rows the slice
// that fits needs 16 K slices (126 MB of fp32 slices for the 100 MB q|k|v weight).  The library GEMM the batch path
// used instead streams the COLD weights at 2.0 TB/s at 160 rows (o: 1.07, down: 1.25 TB/s; 3.0 TB/s at 40 rows;
// tools/batched_decode_gemm_bench.py) - its tiles are chosen for arithmetic, and N = 4096 gives it 16-64 of them.
// Here the weight is the streamed operand and x rides along:
//   * work unit = (slab of BN weight rows, K step of 64); the NB x K/64 units are dealt to the 256 workgroups as
//     CONTIGUOUS ranges (stream-K): every workgroup streams the same number of weight bytes whatever N is (22016 =
//     86 x 256 has no even split-K), a range crosses at most one slab boundary per K/64 units;
//   * per unit both operand tiles go L2/HBM -> LDS by global_load_lds_dwordx4 (8 rows x 128 B per instruction, XOR
//     swizzle on the source address as in psg_dense_gemm.hip) into a 3-unit ring, two units in flight ahead of the MFMAs, with
//     counted vmcnt; x (M x 128 B per unit) comes from L2 - it is 0.4-1.3 MB and every workgroup walks it;
//   * 8 waves = 2 (x rows) x 4 (weight rows): a wave owns TJ 32-row weight tiles and ceil(MT / 2) of the MT 32-row x
//     tiles (v_mfma_f32_32x32x16, D[m][n]: a lane holds one output COLUMN, so a slice store is 128 contiguous bytes per
//     half-wave); fragment reads run one k sub-step ahead of the MFMAs, the next unit's DMAs are issued between them;
//   * a segment (the part of a range inside one slab) ends in ONE fp32 slice: slot = rank of the segment inside its
//     slab; the workgroup that ends a slab zero-fills the slots the slab did not use.  Where slabs x slices fill the
//     grid (q|k|v: 48 x 5, o / down: 32 x 8, lm_head: 125 x 2) the ranges are ALIGNED to the slabs instead: one segment
//     and one flush per workgroup, no unused slot.  The host plans BN, the mode and the slot count by walking the same
//     integer arithmetic (psg_batch_gemm_plan).
// Results: every output element is a sum of <= S k-ordered partial sums; which k ranges they cover depends on (M, N, K)
// only - not on the other rows - so a pair's tokens do not depend on its batch neighbours as long as the row count's
// plan is the same (tests: against the fp32 reference and against psg_skinny_gemm on <= 32 of the rows).
#include "psg_common.h"

#define BG_BK 64
#define BG_WAVES 8
#define BG_GRID_MAX 1024
// ring depth.  Deeper rings where the LDS would hold them (4-6 units for the 128-row slabs / <= 96 rows of x) were measured
// SLOWER (down projection at 160 rows: 35.0 -> 38.1 us, q|k|v at 40 rows: 26.3 -> 30.1): three units everywhere
constexpr int bg_nst(int) { return 3; }

template <int N_>
__device__ __forceinline__ void bg_vmwait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}
template <int N_>
__device__ __forceinline__ void bg_lgkmwait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N_) : "memory");
}
__device__ __forceinline__ void bg_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
typedef uint32_t bg_u32x4 __attribute__((ext_vector_type(4)));
// (inline asm: beside a pending LDS-DMA hipcc puts s_waitcnt vmcnt(0) before every ds_read it emits itself)
__device__ __forceinline__ bg_u32x4 bg_lds_read128(uint32_t a) {
  bg_u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory");
  return v;
}

// first workgroup whose unit range [i T / G, (i + 1) T / G) holds unit u
__host__ __device__ static inline int bg_owner(int64_t u, int64_t T, int G) { return (int)(((u + 1) * G - 1) / T); }

// S_al > 0: aligned ranges (workgroup i = slab i / S_al, slice i % S_al of its K walk); 0: stream-K
// PAIR (psg_split_gemm_w16): x holds TWO planes of the same M <= 32 rows - the high and the low fp16 part of fp32
// activations, [2][M][K] - staged as x tiles 0 and 1; every wave owns both tiles and ONE weight tile (8 waves along the
// weight rows), the two accumulators are added at the flush and multiplied by the row's inverse power-of-two scale:
// part[slot][m][n] = (xh . w + xl . w) 2^-t, M rows.
template <typename E, int MT, int TJ, int WM = 2, bool PAIR = false>
__global__ void __launch_bounds__(BG_WAVES * 64, 1)
batch_gemm_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, float* __restrict__ part, int M, int N,
                  int K, int S, int S_al, int var, const float* __restrict__ row_scale, int wt) {
  using v8 = typename E::v8;
  static_assert(!PAIR || (MT == 2 && WM == 1 && TJ == 1), "the pair form is two x tiles, one weight tile per wave");
  constexpr int WN = BG_WAVES / WM, TI = (MT + WM - 1) / WM, BM = MT * 32, BN = WN * TJ * 32;
  constexpr int STAGE = (BM + BN) * 128;                     // one unit: [x: BM rows | w: BN rows] x 128 B
  constexpr int NST = bg_nst(STAGE);
  static_assert(NST >= 3, "the ring needs three units");
  constexpr int NI = (BM + BN) / 8, PER = NI / BG_WAVES, REM = NI % BG_WAVES;   // DMA instructions per unit / per wave
  constexpr int XI = BM / 8;                                 // the first XI instructions of a unit fetch x rows
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int nk = K / BG_BK, NB = (N + BN - 1) / BN;
  const int64_t T = (int64_t)NB * nk;
  const int G = gridDim.x;
  int64_t g0, g1;
  if (S_al > 0) {
    const int sb = blockIdx.x / S_al, sl = blockIdx.x - sb * S_al;
    g0 = (int64_t)sb * nk + (int64_t)sl * nk / S_al;
    g1 = (int64_t)sb * nk + (int64_t)(sl + 1) * nk / S_al;
  } else {
    g0 = (int64_t)blockIdx.x * T / G;
    g1 = ((int64_t)blockIdx.x + 1) * T / G;
  }
  if (g0 >= g1) return;
  const int wm = wid / WN, wn = wid % WN;
  const int ti0 = wm * TI;
  const bool full = ti0 + TI <= MT;                          // this wave owns TI x tiles (else TI - 1)

  const int srow = lane >> 3, sslot = lane & 7;
  const int my_cnt = PER + (wid < REM ? 1 : 0);
  // instruction idx of a unit: rows 8 idx .. + 7 of the combined [x | w] tile; a wave issues idx = wid, wid + 8, ...
  auto stage_one = [&](int slab, int kt, unsigned char* sb, int idx) {
    const int r = idx * 8 + srow;
    const int piece = sslot ^ ((r >> 1) & 7);                // (BM is a multiple of 16: the operand-local row's swizzle)
    if (idx < XI) {
      if (var == 3 && kt > 0) return;
      int gr = r < M ? r : M - 1;
      if (PAIR) gr = (r >> 5) * M + ((r & 31) < M ? (r & 31) : M - 1);      // plane r >> 5, row r & 31 of [2][M][K]
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(x + (int64_t)gr * K + kt * BG_BK + piece * 8),
          (__attribute__((address_space(3))) void*)(sb + idx * 1024), 16, 0, 0);
    } else {
      int gr = slab * BN + (r - BM);
      gr = gr < N ? gr : N - 1;
      __builtin_amdgcn_global_load_lds(                      // a weight byte is read once by one CU: non-temporal
          (const __attribute__((address_space(1))) void*)(w + (int64_t)gr * K + kt * BG_BK + piece * 8),
          (__attribute__((address_space(3))) void*)(sb + idx * 1024), 16, 0, 2);
    }
  };
  auto stage = [&](int slab, int kt, int buf, int q0, int q1) {   // this wave's instructions q0 .. q1 - 1 of the unit
    unsigned char* sb = smem + buf * STAGE;
    for (int q = q0; q < q1; ++q)
      if (q < my_cnt) stage_one(slab, kt, sb, wid + BG_WAVES * q);
  };

  const int l31 = lane & 31, hi = lane >> 5;
  uint32_t arow[TI], aswz[TI], brow[TJ], bswz[TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    int t_ = ti0 + i;
    t_ = t_ < MT ? t_ : MT - 1;                              // (a wave with TI - 1 tiles re-reads its last one: uniform counts)
    const int r = t_ * 32 + l31;
    arow[i] = (uint32_t)(r * 128);
    aswz[i] = (uint32_t)((r >> 1) & 7);
  }
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int r = (wn * TJ + j) * 32 + l31;
    brow[j] = (uint32_t)(BM * 128 + r * 128);
    bswz[j] = (uint32_t)((r >> 1) & 7);
  }

  union Frag {
    bg_u32x4 u;
    v8 v;
  };
  psg_f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = (psg_f32x16){0};

  int slab = (int)(g0 / nk), kt = (int)(g0 - (int64_t)slab * nk);
  int pslab = slab, pkt = kt;                                // the unit the next DMA stage fetches
  auto advance = [&](int& s_, int& k_) {
    if (++k_ == nk) {
      k_ = 0;
      ++s_;
    }
  };
  int issued = 0;                                            // units in flight (prologue: NST - 1)
#pragma unroll
  for (int p = 0; p < NST - 1; ++p)
    if (g0 + p < g1) {
      stage(pslab, pkt, p, 0, PER + 1);
      advance(pslab, pkt);
      ++issued;
    }
  bool after_flush = false;
  int buf = 0;
  constexpr int QH = (PER + 2) / 2;                          // DMA instructions issued beside the first sub-step
  for (int64_t u = g0; u < g1; ++u) {
    // unit u landed?  loads complete in order: the younger unit's loads may stay outstanding
    const int ahead = issued - 1;                            // younger units in flight (<= NST - 2)
    if (after_flush || ahead <= 0) bg_vmwait<0>();           // (stores of a flush may complete out of order with loads)
    else if (wid < REM) {
      if (ahead == 1) bg_vmwait<PER + 1>();
      else if (ahead == 2) bg_vmwait<2 * (PER + 1)>();
      else if (ahead == 3) bg_vmwait<3 * (PER + 1)>();
      else bg_vmwait<4 * (PER + 1)>();
    } else {
      if (ahead == 1) bg_vmwait<PER>();
      else if (ahead == 2) bg_vmwait<2 * PER>();
      else if (ahead == 3) bg_vmwait<3 * PER>();
      else bg_vmwait<4 * PER>();
    }
    after_flush = false;
    bg_lds_barrier();                                        // unit u is in LDS; nobody reads the buffer of unit u - 1 any more
    --issued;
    const bool pf = u + (NST - 1) < g1;
    int nb_ = buf + (NST - 1);
    nb_ = nb_ >= NST ? nb_ - NST : nb_;
    const uint32_t base = smem_lds + (uint32_t)(buf * STAGE);
    Frag af[2][TI], bf[2][TJ];
    auto read_frags = [&](int sub, Frag (&a_)[TI], Frag (&b_)[TJ]) {
      const uint32_t piece = (uint32_t)(2 * sub + hi);
#pragma unroll
      for (int j = 0; j < TJ; ++j) b_[j].u = bg_lds_read128(base + brow[j] + ((piece ^ bswz[j]) << 4));
#pragma unroll
      for (int i = 0; i < TI; ++i) a_[i].u = bg_lds_read128(base + arow[i] + ((piece ^ aswz[i]) << 4));
    };
    auto mma = [&](Frag (&a_)[TI], Frag (&b_)[TJ]) {
      if (var == 2) return;
#pragma unroll
      for (int i = 0; i < TI; ++i)
        if (i + 1 < TI || full) {
#pragma unroll
          for (int j = 0; j < TJ; ++j) acc[i][j] = E::mfma32(a_[i].v, b_[j].v, acc[i][j]);     // D[m][n]
        }
    };
    read_frags(0, af[0], bf[0]);
    read_frags(1, af[1], bf[1]);
    if (pf) stage(pslab, pkt, nb_, 0, QH);
    bg_lgkmwait<TI + TJ>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mma(af[0], bf[0]);
    __builtin_amdgcn_sched_barrier(0);
    read_frags(2, af[0], bf[0]);
    if (pf) stage(pslab, pkt, nb_, QH, PER + 1);
    bg_lgkmwait<TI + TJ>();
    __builtin_amdgcn_sched_barrier(0);
    mma(af[1], bf[1]);
    __builtin_amdgcn_sched_barrier(0);
    read_frags(3, af[1], bf[1]);
    bg_lgkmwait<TI + TJ>();
    __builtin_amdgcn_sched_barrier(0);
    mma(af[0], bf[0]);
    bg_lgkmwait<0>();
    __builtin_amdgcn_sched_barrier(0);
    mma(af[1], bf[1]);
    __builtin_amdgcn_s_setprio(0);
    if (pf) {
      advance(pslab, pkt);
      ++issued;
    }

    const bool slab_end = kt == nk - 1;
    if ((slab_end || u == g1 - 1) && var != 1) {             // the segment ends: one fp32 slice
      int slot;
      if (S_al > 0) slot = (int)blockIdx.x % S_al;
      else slot = (int)blockIdx.x - bg_owner((int64_t)slab * nk, T, G);
      const int nslots = (slab_end && S_al == 0) ? S : slot + 1;   // ending a slab (stream-K): zero the slots it did not use
      // D[m][n]: register r of a lane = row 8 (r >> 2) + 4 hi + (r & 3) of the tile, column lane & 31.  Row pointers are
      // wave-uniform (scalar), the lane adds ONE 32-bit offset: no per-store address registers
      const int lane_off = 4 * hi * N + l31;
      if constexpr (PAIR) {
        for (int s_ = slot; s_ < nslots; ++s_) {
          float* ps = part + (int64_t)s_ * M * N;
          const bool real = s_ == slot;
          const int n0 = slab * BN + wn * 32;
          if (n0 + l31 < N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int mrow = 8 * (r >> 2) + (r & 3);
              float* rowp = ps + (int64_t)mrow * N + n0;
              if (mrow + 4 * hi < M) {
                const float val = real ? (acc[0][0][r] + acc[1][0][r]) * row_scale[mrow + 4 * hi] : 0.f;
                if (wt)                                      // option wt_stores: agent-scope store = written through the L2
                  __hip_atomic_store(reinterpret_cast<unsigned*>(rowp + lane_off), __float_as_uint(val), __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
                else rowp[lane_off] = val;
              }
            }
          }
        }
      } else
      for (int s_ = slot; s_ < nslots; ++s_) {
        float* ps = part + (int64_t)s_ * M * N;
        const bool real = s_ == slot;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          if (!(i + 1 < TI || full)) continue;
          const int mt0 = (ti0 + i) * 32;
#pragma unroll
          for (int j = 0; j < TJ; ++j) {
            const int n0 = slab * BN + (wn * TJ + j) * 32;
            if (n0 + l31 >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int mrow = mt0 + 8 * (r >> 2) + (r & 3);            // + 4 hi: in lane_off
              float* rowp = ps + (int64_t)mrow * N + n0;                // uniform
              if (mrow + 4 * hi < M) rowp[lane_off] = real ? acc[i][j][r] : 0.f;
            }
            asm volatile("" ::: "memory");
          }
        }
      }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (psg_f32x16){0};
      after_flush = true;
    }
    advance(slab, kt);
    buf = buf + 1 == NST ? 0 : buf + 1;
  }
}

// ---- host side: plan and launch --------------------------------------------------------------------------------------
struct bg_plan {
  int mt, tj, grid, slots, s_al;
};

static int bg_streamk_slots(int N, int K, int BN, int G) {
  const int nk = K / BG_BK, NB = (N + BN - 1) / BN;
  const int64_t T = (int64_t)NB * nk;
  int smax = 1;
  for (int b = 0; b < NB; ++b) {
    const int i0 = bg_owner((int64_t)b * nk, T, G), i1 = bg_owner((int64_t)(b + 1) * nk - 1, T, G);
    if (i1 - i0 + 1 > smax) smax = i1 - i0 + 1;
  }
  return smax;
}

// Candidates: BN = 256 / 128 weight rows per slab x {aligned slices, stream-K}.  Estimated time = the longest range's
// units x the unit's cost (L2 -> LDS staging of both tiles at ~40 B/clk/CU, or the unit's share of the HBM stream) + the
// fp32 slices it leaves (written here, read by the consumer).  Aligned ranges need slabs x slices ~ the grid; stream-K
// balances any shape at the price of a second flush per workgroup and the zero-filled slots.
static bg_plan bg_make_plan(const psg_ctx* ctx, int64_t M, int N, int K, int bn, int mode_, bool pair = false) {
  bg_plan best{0, 0, 0, 0, 0};
  double best_t = 1e300;
  const int mt = (pair || M <= 64) ? 2 : M <= 96 ? 3 : 5;
  int G = ctx->num_cu;
  if (G > BG_GRID_MAX) G = BG_GRID_MAX;
  if (ctx->opt.batch_gemm_grid > 0 && ctx->opt.batch_gemm_grid < G) G = ctx->opt.batch_gemm_grid;
  const int nk = K / BG_BK;
  const int forced_bn = bn ? bn : ctx->opt.batch_gemm_bn, forced_mode = mode_ ? mode_ : ctx->opt.batch_gemm_mode;
  for (int tj = 2; tj >= 1; --tj) {
    const int BN = pair ? 256 : 4 * tj * 32;                 // (the pair form: 8 waves x one 32-row weight tile)
    if (pair && tj != 1) continue;
    if (!pair && forced_bn && forced_bn != BN) continue;
    const int NB = (N + BN - 1) / BN;
    const int64_t T = (int64_t)NB * nk;
    const double unit_us = fmax((double)(mt * 32 + BN) * 128 / (40.0 * 2.4e3), (double)BN * 128 * G / 5.8e6);
    for (int mode = 1; mode <= 2; ++mode) {
      if (forced_mode && forced_mode != mode) continue;
      bg_plan p{mt, tj, 0, 0, 0};
      double units;
      if (mode == 1) {
        int s_al = G / NB;
        if (s_al < 1) continue;
        if (s_al > nk) s_al = nk;
        if (s_al > PSG_MAX_SPLITS) s_al = PSG_MAX_SPLITS;    // what a consumer kernel sums
        p.s_al = p.slots = s_al;
        p.grid = NB * s_al;
        units = (double)((nk + s_al - 1) / s_al);
      } else {
        p.grid = T < G ? (int)T : G;                         // every workgroup gets at least one unit: slot ranks are contiguous
        p.slots = bg_streamk_slots(N, K, BN, p.grid);
        while (p.slots > PSG_MAX_SPLITS && p.grid > NB) {    // few slabs, long K walks: fewer workgroups = longer segments
          p.grid = p.grid * 7 / 8 > NB ? p.grid * 7 / 8 : NB;
          p.slots = bg_streamk_slots(N, K, BN, p.grid);
        }
        if (p.slots > PSG_MAX_SPLITS) continue;
        units = (double)((T + p.grid - 1) / p.grid);
      }
      const double slice_bytes = (double)p.slots * (pair ? 32 : M) * N * 4;   // (pair form: the plan must not follow the row count)
      const double t = units * unit_us + slice_bytes / 3.5e6 + slice_bytes / 8e6 + (mode == 2 ? 3.0 : 0.0);
      if (t < best_t) {
        best_t = t;
        best = p;
      }
    }
  }
  return best;
}

extern "C" int psg_batch_gemm_plan(psg_ctx* ctx, int64_t M, int N, int K, int dtype, int slab_rows, int mode, int* slots) {
  PSG_REQUIRE(ctx && slots, PSG_ERR_INVALID, "psg_batch_gemm_plan: NULL argument");
  PSG_REQUIRE(dtype == PSG_BF16 || dtype == PSG_F16, PSG_ERR_UNSUPPORTED, "psg_batch_gemm_plan: dtype %d (bf16 / fp16)", dtype);
  PSG_REQUIRE(M >= 1 && M <= 160 && N >= 16 && N % 16 == 0 && K >= BG_BK && K % BG_BK == 0, PSG_ERR_UNSUPPORTED,
              "psg_batch_gemm: M=%lld (1..160), N=%d (multiple of 16), K=%d (multiple of %d)", (long long)M, N, K, BG_BK);
  PSG_REQUIRE((slab_rows == 0 || slab_rows == 128 || slab_rows == 256) && mode >= 0 && mode <= 2, PSG_ERR_INVALID,
              "psg_batch_gemm_plan: slab_rows=%d (0, 128, 256), mode=%d (0, 1, 2)", slab_rows, mode);
  const bg_plan p = bg_make_plan(ctx, M, N, K, slab_rows, mode);
  PSG_REQUIRE(p.grid > 0, PSG_ERR_UNSUPPORTED, "psg_batch_gemm_plan: no plan for M=%lld N=%d K=%d under the forced options",
              (long long)M, N, K);
  *slots = p.slots;
  return PSG_OK;
}

template <typename E, int MT, int TJ, int WM = 2, bool PAIR = false>
static int bg_launch(const bg_plan& p, const void* x, const void* w, float* part, int M, int N, int K, int var,
                     void* stream, const float* row_scale = nullptr, int wt = 0) {
  constexpr int STAGE = (MT * 32 + (BG_WAVES / WM) * TJ * 32) * 128;
  constexpr int NST_ = bg_nst(STAGE);
  static_assert(4 * ((MT * 32 + (BG_WAVES / WM) * TJ * 32) / 8 / BG_WAVES + 1) <= 63, "vmcnt field");
  auto k = batch_gemm_kernel<E, MT, TJ, WM, PAIR>;
  hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, NST_ * STAGE);
  if (e != hipSuccess) {
    psg_set_error("psg_batch_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
    return PSG_ERR_HIP;
  }
  k<<<(unsigned)p.grid, BG_WAVES * 64, NST_ * STAGE, (hipStream_t)stream>>>((const uint16_t*)x, (const uint16_t*)w, part, M,
                                                                             N, K, p.slots, p.s_al, var, row_scale, wt);
  PSG_CHECK_LAUNCH("psg_batch_gemm");
  return PSG_OK;
}

extern "C" int psg_batch_gemm(psg_ctx* ctx, const void* x, const void* w, float* part, int64_t M, int N, int K, int slots,
                              int dtype, int slab_rows, int mode, void* stream) {
  PSG_REQUIRE(ctx && x && w && part, PSG_ERR_INVALID, "psg_batch_gemm: NULL argument");
  int want = 0;
  const int rc = psg_batch_gemm_plan(ctx, M, N, K, dtype, slab_rows, mode, &want);
  if (rc != PSG_OK) return rc;
  PSG_REQUIRE(slots == want, PSG_ERR_INVALID, "psg_batch_gemm: slots=%d, the plan for M=%lld N=%d K=%d writes %d", slots,
              (long long)M, N, K, want);
  const bg_plan p = bg_make_plan(ctx, M, N, K, slab_rows, mode);
#define BG_GO(MT_, TJ_) return bg_launch<E, MT_, TJ_>(p, x, w, part, (int)M, N, K, ctx->opt.batch_gemm_var, stream)
  PSG_DISPATCH_E16(dtype, "psg_batch_gemm",
                   if (p.tj == 2) {
                     if (p.mt == 2) BG_GO(2, 2); else if (p.mt == 3) BG_GO(3, 2); else BG_GO(5, 2);
                   } else {
                     if (p.mt == 2) BG_GO(2, 1); else if (p.mt == 3) BG_GO(3, 1); else BG_GO(5, 1);
                   });
#undef BG_GO
  return PSG_ERR_INVALID;
}

// ---- fp32 activations x weights that are fp16 values, as TWO fp16 products on the 16-bit matrix cores -----------------
// x 2^t = xh + xl (psg_split_f16x2: planes [2][M][K], inv_scale = 2^-t), w exactly fp16 (a frozen fp16 checkpoint that
// the reference upcasts on load, V4:99-100 + configs/psg/baseline_v4_ov.py:61-65): x . w = (xh . w + xl . w) 2^-t with
// every product exact in fp32 - what is lost against the fp32 product is x's split residual, 2^-22 relative, the class of
// the fp32s prompt pass (psg_split.hip) with ONE weight segment instead of three because wl = 0.  M <= 32 rows.
extern "C" int psg_split_gemm_w16_plan(psg_ctx* ctx, int M, int N, int K, int mode, int* slots) {
  PSG_REQUIRE(ctx && slots, PSG_ERR_INVALID, "psg_split_gemm_w16_plan: NULL argument");
  PSG_REQUIRE(M >= 1 && M <= 32 && N >= 16 && N % 16 == 0 && K >= BG_BK && K % BG_BK == 0 && mode >= 0 && mode <= 2,
              PSG_ERR_UNSUPPORTED, "psg_split_gemm_w16: M=%d (1..32), N=%d (multiple of 16), K=%d (multiple of %d), mode=%d", M,
              N, K, BG_BK, mode);
  const bg_plan p = bg_make_plan(ctx, M, N, K, 0, mode, true);
  PSG_REQUIRE(p.grid > 0 && p.slots <= PSG_MAX_SPLITS, PSG_ERR_UNSUPPORTED,
              "psg_split_gemm_w16_plan: no plan with <= %d slices for M=%d N=%d K=%d", PSG_MAX_SPLITS, M, N, K);
  *slots = p.slots;
  return PSG_OK;
}

extern "C" int psg_split_gemm_w16(psg_ctx* ctx, const void* x2, const float* inv_scale, const void* w16, float* part, int M,
                                  int N, int K, int slots, int mode, void* stream) {
  PSG_REQUIRE(ctx && x2 && inv_scale && w16 && part, PSG_ERR_INVALID, "psg_split_gemm_w16: NULL argument");
  int want = 0;
  const int rc = psg_split_gemm_w16_plan(ctx, M, N, K, mode, &want);
  if (rc != PSG_OK) return rc;
  PSG_REQUIRE(slots == want, PSG_ERR_INVALID, "psg_split_gemm_w16: slots=%d, the plan for M=%d N=%d K=%d writes %d", slots, M,
              N, K, want);
  const bg_plan p = bg_make_plan(ctx, M, N, K, 0, mode, true);
  return bg_launch<EF16, 2, 1, 1, true>(p, x2, w16, part, M, N, K, ctx->opt.batch_gemm_var, stream, inv_scale,
                                        (ctx->opt.wt_stores >> 1) & 1);
}
