// K15 at the reference's own precision: weight-streaming "skinny" GEMM with fp32 weights and activations.
//
//   part[s][M][N] = x[M][Ks] . w[N][Ks]^T      M <= 32 rows (the selected pairs), fp32 in, fp32 out
//   y = sum_s part[s]                           (summed, in split order, by the CONSUMER kernel)
//
// The reference loads the LLM without a dtype (V4:99-100): its q/k/v/o/gate/up/down projections and lm_head
// (HF-LL:163-177, 243-281) run in fp32, one pair at a time (V4:293-312).  This is the decode-step projection of the
// fp32 verification mode - the only mode inside the north star's 1e-3 / argmax-exact tolerance - as the same
// HBM-bound stream the 16-bit modes use (psg_gemm.hip): every weight byte leaves HBM once per step for ALL selected
// pairs.  4 bytes per weight: 27 GB per decode step of Llama-2-7B, 2 * M flops per 4 bytes = 10 flop/B at M = 20.
//
// Arithmetic: exact fp32 on the matrix cores - f32 in / f32 accumulate, bit-for-bit a k-ordered fmaf chain, at the
// f32 VECTOR rate (MI355X_MICROARCH.md: 32 fma/clk/SIMD).  That rate is the design constraint the 16-bit kernel does
// not have: 64 weights (256 B) against 16 x rows occupy a SIMD's matrix pipe for 32 cycles, so M = 20 padded to 32 rows
// would need 16 B/clk/CU of matrix pipe under an HBM stream of 11-13 B/clk/CU.  The kernel therefore works in the
// MULTI-BLOCK forms, which take the same A register (lane (n, kq) holds w[n][k(kq)]) and differ in how many x rows
// they serve:
//   v_mfma_f32_16x16x1_4b_f32  four 16x16 outer products, one per kq: 16 x rows, 32 cycles, 16 accumulator registers
//   v_mfma_f32_4x4x1_16b_f32   sixteen 4x4 outer products, block 4 kq + (n >> 2): 4 x rows, 8 cycles, 4 registers
// M = 20 is one of each: 40 cycles per weight fragment instead of 64.  Both leave, per output, FOUR partial sums - one
// per kq, each the stream-ordered fmaf chain of that kq's k's - which are added as (kq0 + kq1) + (kq2 + kq3) when the
// slab ends (register-wise for the 16-row form, across the wave's four 16-lane rows for the 4-row form).  Every x row
// therefore goes through identical arithmetic whatever its position in the batch and whatever M is: a pair's decode is
// bit-for-bit independent of which other pairs share the step and of the rank it was dealt to.  (Measured on MI355X: a
// version built from 4x4x1 alone - 5 per fragment at M = 20 - stalls ~20 cycles of issue per instruction,
// SQ_WAIT_INST_ANY 37 % of the wave cycles at 33 % matrix-pipe occupancy: 4.5 TB/s against 5.6 TB/s at M = 4.)
//
// Everything else is the byte layout of skinny_gemm_dma_kernel (psg_gemm.hip), with a K block of 32 floats = 128 B:
//   * grid = (G, S): workgroup (gx, by) owns K blocks [KB by / S, KB (by + 1) / S) and walks the 16*WAVES-row slabs
//     gx, gx + G, ... persistently; wave w owns rows 16 w .. 16 w + 15 of each slab;
//   * weights go HBM -> LDS by global_load_lds_dwordx4 in full 128-byte lines (one instruction = 8 rows x 128 B,
//     non-temporal), source-swizzled (lane (r, p) fetches piece p ^ r of row r) so that the fragment reads are
//     conflict-free; each wave owns a private ring of SLOTS blocks, synchronised by its own counted vmcnt only;
//   * x[0..M)[K range] is staged once per workgroup by LDS-DMA and reused for every slab;
//   * K inside a block is permuted the same way for w and x: lane (n, kq) reads the 16-byte pieces 2 kq and 2 kq + 1
//     of row n; float i of a piece is the operand of the i-th MFMA on that piece.
//
// W16 (psg_skinny_gemm_w16): the same kernel over weights STORED as fp16.  The reference's LLM is a frozen fp16 checkpoint
// (configs/psg/baseline_v4_ov.py:61-65: Llama-2-7b-hf, `freeze_layers=[..., 'relation_head.language_model']`) that
// `from_pretrained` upcasts to fp32 (V4:99-100): every weight IS an fp16 value, and reading it as 2 bytes and widening it
// in the register (v_cvt_f32_f16: exact) feeds the same f32 MFMAs the same operands in the same order - results
// bit-identical to the fp32-weight stream at half the HBM bytes.  The engine keeps fp16 storage only for tensors it has
// verified to round-trip (w.half().float() == w, every element); anything trained stays fp32.  Layout differences: a K
// block of 32 weights is 64 B per row, one DMA instruction per 16-row block (lane -> row lane >> 2, piece lane & 3,
// source-swizzled by (row >> 1) & 3), twice the ring slots for the same bytes in flight.
#include <stdlib.h>

#include <type_traits>

#include "psg_common.h"

typedef float sf32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void sgf_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// The partial tile's LDS traffic is inline asm for the reason given in psg_gemm.hip: to hipcc a pending LDS-DMA may
// alias any ds access it generates itself, and it would drain the prefetch ring at every slab end.
__device__ __forceinline__ void sgf_ds_write128(uint32_t lds_addr, sf32x4_t v) {
  asm volatile("s_nop 15\n\tds_write_b128 %0, %1" ::"v"(lds_addr), "v"(v) : "memory");
}
__device__ __forceinline__ sf32x4_t sgf_ds_read128(uint32_t lds_addr) {
  sf32x4_t v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr) : "memory");
  return v;
}
template <int N_>
__device__ __forceinline__ void sgf_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}
// wait until at most `newer` blocks (2 DMA instructions each) issued after the wanted one are outstanding
template <int MAXN, int PER = 2>
struct SgfWait {
  static __device__ __forceinline__ void go(int newer) {
    if (newer >= MAXN) sgf_wait<(MAXN * PER < 63 ? MAXN * PER : 63)>();
    else SgfWait<MAXN - 1, PER>::go(newer);
  }
};
template <int PER>
struct SgfWait<0, PER> {
  static __device__ __forceinline__ void go(int) { sgf_wait<0>(); }
};

// sum over the four 16-lane rows of a wave (= the four kq partials of an output), in every lane: (kq0 + kq1) + (kq2 + kq3).
// v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second, v_permlane32_swap
// the upper half of the first with the lower half of the second; fed two copies of a value they leave "this row pair's
// first" / "second" in the two registers.  Inline asm: ROCm 7.2's __builtin_amdgcn_permlane16_swap returns its FIRST
// result in both vector elements (v_add_f32 v1, v1, v1 in the ISA), and an asm operand gets no hazard padding from
// hipcc - the s_nop covers matrix-core result -> VALU read (2-pass MFMA: 5 wait states) and VALU write -> permlane read.
__device__ __forceinline__ float sgf_sum_kq(float v) {
  float a = v, b = v;
  asm volatile("s_nop 7\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));      // [r0 r0 r2 r2], [r1 r1 r3 r3]
  const float s = a + b;
  float c = s, d = s;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(d));      // [lo lo], [hi hi]
  return c + d;
}

typedef float sf32x16_t __attribute__((ext_vector_type(16)));

// G16 sixteen-row groups (x rows 16 G .. 16 G + 15) followed by G4 four-row groups (x rows 16 G16 + 4 q ..)
template <int WAVES, int SLOTS, int G16, int G4, bool W16 = false>
__global__ void __launch_bounds__(WAVES * 64) skinny_gemm_f32_kernel(const float* __restrict__ x,
                                                                     const void* __restrict__ w,
                                                                     float* __restrict__ part, int M, int N, int K,
                                                                     int xstride, int wt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int ROWS = WAVES * 16;
  constexpr int WROW = W16 ? 64 : 128;                              // bytes of one row of a 32-weight K block
  constexpr int BLOCK_BYTES = 16 * WROW;                            // 16 rows
  constexpr int RING_BYTES = SLOTS * BLOCK_BYTES;
  constexpr int OT_PITCH = ROWS + 4;                                // output tile pitch (floats)
  constexpr int MP = G16 * 16 + G4 * 4;                             // x rows of the tile
  constexpr int NG4 = G4 > 0 ? G4 : 1, NG16 = G16 > 0 ? G16 : 1;    // array extents (zero-length arrays are not C++)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n = lane & 15, kq = lane >> 4;
  const int S = gridDim.y, by = blockIdx.y, G = gridDim.x, gx = blockIdx.x;
  const int KB = K >> 5;                                            // 32-float blocks
  const int kbA = (int)((unsigned)(KB * by) / (unsigned)S), kbB = (int)((unsigned)(KB * (by + 1)) / (unsigned)S);
  const int nkb = kbB - kbA;
  const int nslab_all = (N + ROWS - 1) / ROWS;
  const int nslab = gx < nslab_all ? (nslab_all - gx + G - 1) / G : 0;
  unsigned char* ring = smem + wid * RING_BYTES;                    // this wave's private ring
  // partial tile [MP x rows][OT_PITCH] fp32 behind the rings, addressed by its LDS byte address
  const uint32_t otile_lds =
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(smem + WAVES * RING_BYTES);
  unsigned char* xs = smem + WAVES * RING_BYTES + MP * OT_PITCH * 4;            // shared x slice [M][xstride bytes]
  const int total = nslab * nkb;                                    // flattened (slab, block) stream
  const unsigned char* wb = reinterpret_cast<const unsigned char*>(w);
  const int64_t row_bytes = (int64_t)K * (W16 ? 2 : 4);
  const int64_t xrow_bytes = (int64_t)K * 4;

  // DMA source of this lane.  fp32 weights: row (lane >> 3) of an 8-row group, piece (lane & 7) ^ (lane >> 3), two
  // instructions per block; fp16 weights: row (lane >> 2) of the 16-row block, piece (lane & 3) ^ ((row >> 1) & 3), one
  const int dr = W16 ? lane >> 2 : lane >> 3;
  const int dp = W16 ? (lane & 3) ^ ((dr >> 1) & 3) : (lane & 7) ^ (lane >> 3);
  auto dma_src = [&](int t) -> const unsigned char* {
    int r = (gx + t * G) * ROWS + wid * 16 + dr;
    if (W16) r = r < N ? r : N - 1;                                 // rows past N: clamped (results dropped)
    else r = r + 8 < N ? r : (N - 9 > 0 ? N - 9 : 0);               // keep rows r and r + 8 in range
    return wb + (int64_t)r * row_bytes + (int64_t)kbA * WROW + dp * 16;
  };
  int lt = 0, lb = 0, ls = 0;                                       // load cursor: slab, block, ring slot
  const unsigned char* src = dma_src(0);
  auto issue = [&]() {
    unsigned char* dst = ring + ls * BLOCK_BYTES;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 2);
    if (!W16)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 8 * row_bytes),
                                       (__attribute__((address_space(3))) void*)(dst + 1024), 16, 0, 2);
    src += WROW;
    if (++ls == SLOTS) ls = 0;
    if (++lb == nkb) { lb = 0; ++lt; src = dma_src(lt); }
  };
  // x slice by LDS-DMA: one instruction = up to 64 16-byte pieces of ONE row (lanes past the slice are masked off),
  // issued ahead of the weight stream; no ordinary load next to the DMAs
  {
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(x);
    const int pieces = nkb * 8;
    const int cpr = (pieces + 63) >> 6;                             // 1 KiB chunks per row
    const int items = M * cpr;
    for (int it = wid; it < items; it += WAVES) {
      const int r = it / cpr, j = it - r * cpr;
      const int c = j * 64 + lane;
      if (c < pieces)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(xb + (int64_t)r * xrow_bytes + (int64_t)kbA * 128 + c * 16),
            (__attribute__((address_space(3))) void*)(xs + r * xstride + j * 1024), 16, 0, 0);
    }
  }
  for (int i = 0; i < SLOTS - 1; ++i)
    if (i < total) issue();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this wave's x rows (and first weight blocks) landed
  __syncthreads();

  // B operands, the k's of this lane's kq: 16-row group -> x row 16 G + (lane & 15); 4-row group -> x row
  // 16 G16 + 4 q + (lane & 3).  Rows >= M are clamped (they only feed output rows >= M, which are never stored)
  const unsigned char* x16[NG16];
  const unsigned char* x4[NG4];
#pragma unroll
  for (int g = 0; g < G16; ++g) x16[g] = xs + min(16 * g + n, M - 1) * xstride + kq * 32;
#pragma unroll
  for (int q = 0; q < G4; ++q) x4[q] = xs + min(16 * G16 + 4 * q + (lane & 3), M - 1) * xstride + kq * 32;
  // fp32 weights: fragment (row n, piece c = 2 kq + j) sits at slot c ^ (n & 7) of row n; fp16 weights: the 8 halfs of
  // this lane's kq are ONE 16-byte piece, at slot kq ^ ((n >> 1) & 3) of the 64-byte row n
  const int arow = W16 ? n * 64 : (n >> 3) * 1024 + (n & 7) * 128;
  const int a0off = W16 ? arow + ((kq ^ ((n >> 1) & 3)) * 16) : arow + (((2 * kq) ^ (n & 7)) * 16);
  const int a1off = W16 ? a0off : arow + (((2 * kq + 1) ^ (n & 7)) * 16);
  const sf32x4_t zero4 = {0, 0, 0, 0};
  const sf32x16_t zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  sf32x16_t acc16[NG16];
  sf32x4_t acc4[NG4];
#pragma unroll
  for (int g = 0; g < G16; ++g) acc16[g] = zero16;
#pragma unroll
  for (int q = 0; q < G4; ++q) acc4[q] = zero4;
  int ct = 0, cb = 0, cs = 0;                                       // compute cursor: slab, block, ring slot
  auto finish_slab = [&]() {
    // 4-row groups: block b = lane >> 2 = 4 kq + g holds D_b[r][j] (register r, lane 4 b + j) = sum over this kq's k of
    // w[4 g + r][k] x[row 4 q + j][k]; the four kq partials are summed across the wave's four 16-lane rows
#pragma unroll
    for (int q = 0; q < G4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc4[q][r] = sgf_sum_kq(acc4[q][r]);
    sgf_lds_barrier();                                              // previous slab's tile fully read
    {
      // 16-row groups: register 4 b + r of lane (j = lane & 15, ig = lane >> 4) = block b = kq's partial of
      // w[4 ig + r][.] x[row 16 G + j][.]; the partials are summed register-wise in the same (0 + 1) + (2 + 3) tree
      const uint32_t tp16 = otile_lds + (uint32_t)(n * OT_PITCH + wid * 16 + 4 * kq) * 4u;
#pragma unroll
      for (int g = 0; g < G16; ++g) {
        sf32x4_t v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (acc16[g][r] + acc16[g][4 + r]) + (acc16[g][8 + r] + acc16[g][12 + r]);
        sgf_ds_write128(tp16 + (uint32_t)(16 * g * OT_PITCH) * 4u, v);
      }
      // 4-row groups: every lane holds the total; row kq = q & 3 stores group q: tile[row 4 q + j][4 g + r]
      const int g4 = (lane >> 2) & 3, j = lane & 3;
      const uint32_t tp4 = otile_lds + (uint32_t)((16 * G16 + j) * OT_PITCH + wid * 16 + 4 * g4) * 4u;
#pragma unroll
      for (int q = 0; q < G4; ++q)
        if (kq == (q & 3)) sgf_ds_write128(tp4 + (uint32_t)(4 * q * OT_PITCH) * 4u, acc4[q]);
    }
    sgf_lds_barrier();
    {
      const int nblk = (gx + ct * G) * ROWS;
      constexpr int C4 = ROWS / 4;                                  // float4 columns per row
      for (int e = tid; e < M * C4; e += WAVES * 64) {
        const int m = e / C4, c4 = e - m * C4;
        if (nblk + c4 * 4 + 4 <= N) {
          const sf32x4_t v = sgf_ds_read128(otile_lds + (uint32_t)(m * OT_PITCH + c4 * 4) * 4u);
          float* dst = part + ((int64_t)by * M + m) * N + nblk + c4 * 4;
          if (wt) psg_st4_wt(dst, v[0], v[1], v[2], v[3]);         // option wt_stores: the slices leave the L2 as they are written
          else *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
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
  if (nkb == 0) {
    for (int t = 0; t < nslab; ++t) finish_slab();
    return;
  }
  for (int j = 0; j < total; ++j) {
    if (j + SLOTS - 1 < total) issue();                             // refills the slot consumed at j - 1
    SgfWait<SLOTS - 1, W16 ? 1 : 2>::go(total - 1 - j);             // blocks issued after block j may stay in flight
    const unsigned char* slot = ring + cs * BLOCK_BYTES;
    if (++cs == SLOTS) cs = 0;
    const int o = cb * 128;
    // k order of every output's kq partial: (block, piece 2 kq: floats 0..3, piece 2 kq + 1: floats 0..3)
    typedef _Float16 sf16x8_t __attribute__((ext_vector_type(8)));
    sf16x8_t a16;
    if (W16) a16 = *reinterpret_cast<const sf16x8_t*>(slot + a0off);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      sf32x4_t a;
      if (W16) {                                                    // exact widening: the operands of the fp32-weight stream
        a[0] = (float)a16[4 * h]; a[1] = (float)a16[4 * h + 1]; a[2] = (float)a16[4 * h + 2]; a[3] = (float)a16[4 * h + 3];
      } else {
        a = *reinterpret_cast<const sf32x4_t*>(slot + (h ? a1off : a0off));
      }
      sf32x4_t b16[NG16], b4[NG4];
#pragma unroll
      for (int g = 0; g < G16; ++g) b16[g] = *reinterpret_cast<const sf32x4_t*>(x16[g] + o + 16 * h);
#pragma unroll
      for (int q = 0; q < G4; ++q) b4[q] = *reinterpret_cast<const sf32x4_t*>(x4[q] + o + 16 * h);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int g = 0; g < G16; ++g) acc16[g] = __builtin_amdgcn_mfma_f32_16x16x1f32(a[i], b16[g][i], acc16[g], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < G4; ++q) acc4[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[i], b4[q][i], acc4[q], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // ring reads retired before the slot is refilled
    if (++cb == nkb) finish_slab();
  }
}

#define SGF_XPAD 16   // bytes between x rows beyond the slice: bank shift of 4 dwords per row
// LDS of a launch: rings + partial tile + x slice
static size_t sgf_lds(int M, int K, int splits, int wv, int slots) {
  const int KB = K >> 5, nkb_max = (KB + splits - 1) / splits, mp = M > 12 ? (M <= 16 ? 16 : (M > 28 ? 32 : (M + 3) / 4 * 4)) : (M + 3) / 4 * 4;
  return (size_t)wv * slots * 2048 + (size_t)mp * (16 * wv + 4) * 4 + (size_t)M * (nkb_max * 128 + SGF_XPAD);
}
static int sgf_slots(const psg_ctx* ctx) { return ctx->opt.skinny_f32_slots == 5 ? 5 : 3; }
// rounds of slabs a workgroup walks x slab height: what the launch lasts, up to the K range (the same for every
// candidate of one split count)
static int sgf_round_cost(const psg_ctx* ctx, int N, int splits, int wv) {
  const int g = ctx->num_cu / splits > 0 ? ctx->num_cu / splits : 1;
  const int ns = (N + 16 * wv - 1) / (16 * wv);
  return ((ns + g - 1) / g) * wv;
}

// Split count of the fp32 kernel: one workgroup per CU (its LDS holds the rings, the tile and the x slice), 8 slices of
// K where K allows it (column groups = CUs / slices), deeper only while the x slice does not fit next to the rings.
// The count does NOT depend on M (the LDS bound is taken at 32 rows): the slices are summed in order by the consumer, so
// a plan that followed the row count would make a pair's result depend on how many pairs share the step.
int psg_sgf_plan(const psg_ctx* ctx, int M, int N, int K) {
  (void)M;
  const int forced = ctx->opt.skinny_splits;
  const int KB = K >> 5;
  int S = forced > 0 ? forced : 8;
  while (S > 1 && KB / S < 4) S >>= 1;                               // at least 4 blocks (128 floats) per slice
  if (forced <= 0)
    while (S < PSG_MAX_SPLITS && S * 2 <= KB && sgf_lds(32, K, S, 8, 3) > 156 * 1024) S *= 2;
  if (S > KB) S = KB;
  if (S > PSG_MAX_SPLITS) S = PSG_MAX_SPLITS;
  if (S < 1) S = 1;
  return S;
}

// Does a launch of M rows at this split count fit?  (The plan bounds the LDS at 32 rows with at most PSG_MAX_SPLITS
// slices - what a consumer can sum: M = 32 fits up to K = 11776, M = 20 up to K = 20480.  Beyond that the kernel refuses
// and the caller takes the library GEMM: psg_skinny_gemm_plan reports it before anything is launched.)
bool psg_sgf_fits(const psg_ctx* ctx, int M, int K, int splits) {
  (void)ctx;
  return sgf_lds(M, K, splits, 8, 3) <= 160 * 1024;
}

int psg_sgf_launch(psg_ctx* ctx, const void* x, const void* w, float* part, int M, int N, int K, int splits,
                   void* stream, bool w16) {
  PSG_REQUIRE(ctx && x && w && part, PSG_ERR_INVALID, "psg_skinny_gemm(f32): NULL argument");
  PSG_REQUIRE(M >= 1 && M <= 32, PSG_ERR_UNSUPPORTED, "psg_skinny_gemm(f32): M=%d (1..32 rows)", M);
  PSG_REQUIRE(N >= 16 && N % 16 == 0 && K >= 32 && K % 32 == 0 && K <= (1 << 20), PSG_ERR_UNSUPPORTED,
              "psg_skinny_gemm(f32): N=%d must be a multiple of 16, K=%d a multiple of 32", N, K);
  const int KB = K >> 5;
  PSG_REQUIRE(splits >= 1 && splits <= KB && splits <= PSG_MAX_SPLITS, PSG_ERR_INVALID,
              "psg_skinny_gemm(f32): splits=%d (1..%d)", splits, KB < PSG_MAX_SPLITS ? KB : PSG_MAX_SPLITS);
  const int nkb_max = (KB + splits - 1) / splits;
  const int xstride = nkb_max * 128 + SGF_XPAD;
  int slots = sgf_slots(ctx);
  if (slots > 3 && sgf_lds(M, K, splits, 8, slots) > 160 * 1024) slots = 3;       // large M x K slice: shallower rings
  // slab height against round quantisation, as in the 16-bit kernel (gate/up: 172 slabs of 128 rows over 32 column
  // groups = 6 rounds, 126 slabs of 176 rows = 4 rounds of 1.375x the rows: 8 % less)
  int wv = 8;
  const int forced_wv = ctx->opt.skinny_f32_waves;
  if (forced_wv == 8 || forced_wv == 11 || forced_wv == 12 || forced_wv == 16) {
    wv = forced_wv;
  } else if (ctx->opt.skinny_wide) {
    const int g = ctx->num_cu / splits > 0 ? ctx->num_cu / splits : 1;
    for (int cand : {11, 12}) {
      if (sgf_lds(M, K, splits, cand, slots) > 160 * 1024) continue;
      const int cc = sgf_round_cost(ctx, N, splits, cand), cw = sgf_round_cost(ctx, N, splits, wv);
      // cheaper walk - or the same walk in fewer, evenly filled rounds (q/k/v: 3 rounds of 128 rows -> 2 of 192: one slab
      // epilogue less and 12 waves' worth of DMAs in flight; measured 46.9 -> 43.1 us)
      const int ns = (N + 16 * cand - 1) / (16 * cand);
      if ((double)cc < 0.95 * cw || (cc == cw && ns % g == 0)) wv = cand;
    }
  }
  const size_t lds = sgf_lds(M, K, splits, wv, slots);
  PSG_REQUIRE(lds <= 160 * 1024, PSG_ERR_UNSUPPORTED, "psg_skinny_gemm(f32): %zu B of LDS; use more splits", lds);
  const int rows = wv * 16;
  const int nslab = (N + rows - 1) / rows;
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu * wv > 16) per_cu = 16 / wv;
  if (per_cu < 1) per_cu = 1;
  int G = (per_cu * ctx->num_cu + splits - 1) / splits;
  if (G > nslab) G = nslab;
  if (G < 1) G = 1;
  const dim3 grid(G, splits);
  hipStream_t st = (hipStream_t)stream;
  const int wt = ctx->opt.wt_stores & 1;
  // x rows: 16-row groups (16x16x1_4b) then 4-row groups (4x4x1_16b); up to 12 rows go through 4-row groups only
  const int g16 = M <= 12 ? 0 : (M <= 28 ? 1 : 2), g4 = M <= 12 ? (M + 3) / 4 : (M <= 16 || M > 28 ? 0 : (M - 16 + 3) / 4);
#define SGF_K(WV, SL, A, B)                                                                              \
  do {                                                                                                   \
    if (w16) {                                      /* fp16-stored weights: 1 KB blocks, twice the slots */ \
      (void)hipFuncSetAttribute((const void*)skinny_gemm_f32_kernel<WV, 2 * SL, A, B, true>,             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                 \
      skinny_gemm_f32_kernel<WV, 2 * SL, A, B, true><<<grid, WV * 64, lds, st>>>((const float*)x, w, part, M, N, K, \
                                                                                 xstride, wt);           \
    } else {                                                                                             \
      (void)hipFuncSetAttribute((const void*)skinny_gemm_f32_kernel<WV, SL, A, B>,                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                 \
      skinny_gemm_f32_kernel<WV, SL, A, B><<<grid, WV * 64, lds, st>>>((const float*)x, w, part, M, N, K, xstride, wt); \
    }                                                                                                    \
  } while (0)
#define SGF_L(WV, A, B)                   \
  do {                                    \
    if (slots == 5) SGF_K(WV, 5, A, B);   \
    else SGF_K(WV, 3, A, B);              \
  } while (0)
#define SGF_W(WV)                                    \
  do {                                               \
    switch (g16 * 4 + g4) {                          \
      case 1: SGF_L(WV, 0, 1); break;                \
      case 2: SGF_L(WV, 0, 2); break;                \
      case 3: SGF_L(WV, 0, 3); break;                \
      case 4: SGF_L(WV, 1, 0); break;                \
      case 5: SGF_L(WV, 1, 1); break;                \
      case 6: SGF_L(WV, 1, 2); break;                \
      case 7: SGF_L(WV, 1, 3); break;                \
      default: SGF_L(WV, 2, 0); break;               \
    }                                                \
  } while (0)
  if (wv == 8) SGF_W(8);
  else if (wv == 11) SGF_W(11);
  else if (wv == 16) SGF_W(16);
  else SGF_W(12);
#undef SGF_W
#undef SGF_L
#undef SGF_K
  PSG_CHECK_LAUNCH("psg_skinny_gemm(f32)");
  return PSG_OK;
}
