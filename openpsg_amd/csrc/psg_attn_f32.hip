// fp32 attention on the matrix cores: the reference-precision (fp32 / fp32s) versions of
//   K6  relation-query cross-attention   HF-IB:464-496 via V4:168-170, 179-185 (pair masks V4:430-433)
//   K5  Q-Former self-attention           HF-IB:471-515
//   K14 Llama prompt-pass attention       HF-LL:191-214 (causal, compact sequences)
// Until round 5 these modes ran on the scalar checker kernels (cross_attn_simple_kernel<float> 4.8 ms,
// qformer_self_attn_kernel<float> 1.6 ms, llm_attn_kernel<float> 2.4 ms per image); the 16-bit matrix-core kernels
// (psg_xattn_dma.hip, psg_selfattn_mfma.hip, psg_prefill_attn_mfma.hip) cannot carry an fp32 statement.
//
// Arithmetic: v_mfma_f32_16x16x4_f32 - f32 in, f32 accumulate, bit-for-bit an fmaf chain, at the f32 vector rate
// (64 FLOP/clk/SIMD, MI355X_MICROARCH.md) - for BOTH contractions; softmax statistics in fp32 registers.
//
// One wave per unit, no LDS staging, no barriers.  Everything is laid out so that no value ever changes lanes except
// the two 4-way reductions of the softmax:
//   * scores are computed TRANSPOSED, S^T[key][row] = K . Q^T: A = a key tile (lane (m, kq) holds K[key m][16 j + 4 kq + i]
//     for MFMA (j, i)), B = the query tile (lane (n, kq) holds Q[row n][16 j + 4 kq + i]); both are 16-byte loads along
//     the head dimension.  D: lane (n, g) register r = S^T[key 4 g + r][row n];
//   * those registers ARE the B operand of the second contraction, out^T[dim][row] = V^T . P^T, when step r of a key
//     tile contracts the keys {4 kq + r}: lane (n, kq) register r = P[key 4 kq + r][row n].  Its A operand, lane (m, kq) =
//     V[key 4 kq + r][dim], comes from ONE 16-byte load per 64 dims: lane m reads V[key][64 h + 4 m .. 4 m + 3] and
//     element e feeds the output tile 4 h + e whose row m is dim 64 h + 4 m + e (the 16-dim output tiles interleave);
//   * D of that contraction: lane (n, g) register r of tile 4 h + e = out[row n][64 h + 16 g + 4 r + e]: a row's softmax
//     statistics (max, sum: per row n = lane & 15, identical in the four lane groups) rescale it lane-locally, and the
//     four tiles of an h give 16-byte stores of 4 consecutive dims.
// Keys are COMPACTED: a unit builds the list of the key rows its mask admits (a masked key contributes exactly 0: HF's
// additive finfo.min absorbs the score, exp underflows) and contracts over ceil(count / 16) tiles of gathered rows -
// the union of two object rectangles covers ~45 of 256 patches, so this is 3-5x less matrix work than skipping
// 16-key tiles and 5x less than the dense contraction.  K / V rows are gathered straight from L2 (1.5 MB per layer).
#include <stdlib.h>

#include "psg_common.h"

typedef float af4 __attribute__((ext_vector_type(4)));

// max / sum over the four 16-lane rows of a wave, result in every lane (v_permlane16_swap / v_permlane32_swap; inline
// asm as in psg_gemm_f32.hip: the ROCm 7.2 builtin returns its first result twice, and an asm operand gets no hazard
// padding from hipcc - s_nop covers VALU write -> permlane read)
__device__ __forceinline__ float af_rows4_sum(float v) {
  float a = v, b = v;
  asm volatile("s_nop 3\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  const float s = a + b;
  float c = s, d = s;
  asm volatile("s_nop 3\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(d));
  return c + d;
}
__device__ __forceinline__ float af_rows4_max(float v) {
  float a = v, b = v;
  asm volatile("s_nop 3\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  const float s = fmaxf(a, b);
  float c = s, d = s;
  asm volatile("s_nop 3\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(d));
  return fmaxf(c, d);
}

// The shared tile loop.  HD: head dim (64 / 128); NQT: query tiles (16 rows each) that share the key loop.
//   qrow[t]   this lane's query row of tile t (row n = lane & 15), pointing at the head's first dim
//   keys      per-wave LDS list of key row indices (int32), padded to a multiple of 16 with a valid index
//   nk        number of listed keys (>= 1)
//   kbase / vbase + index * kstride (floats): the head's K / V rows
//   mask      functor: float operator()(float s, int slot, int t) -> masked score (-INFINITY = no contribution)
//   out[t]    this lane's output row (row n of tile t), or nullptr: nothing stored for that row
template <int HD, int NQT, class Mask>
__device__ __forceinline__ void af_tiles(const float* const (&qrow)[NQT], const int32_t* keys, int nk,
                                         const float* __restrict__ kbase, const float* __restrict__ vbase, int64_t kstride,
                                         float scale, Mask mask, float* const (&out)[NQT]) {
  constexpr int NJ = HD / 16, NH = HD / 64, NDT = HD / 16;
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  af4 qf[NQT][NJ], acc[NQT][NDT];
  float mrun[NQT], lrun[NQT];
  const af4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NQT; ++t) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) qf[t][j] = *reinterpret_cast<const af4*>(qrow[t] + 16 * j + 4 * g);
#pragma unroll
    for (int d = 0; d < NDT; ++d) acc[t][d] = zero;
    mrun[t] = -INFINITY;
    lrun[t] = 0.f;
  }
  const int ntile = (nk + 15) >> 4;
  for (int kt = 0; kt < ntile; ++kt) {
    // gathered rows: K of key slot 16 kt + n (A of the scores), V of key slots 16 kt + 4 g + r (A of the output)
    const int32_t ki = keys[16 * kt + n];
    const int32_t* vk = keys + 16 * kt + 4 * g;
    const float* kp = kbase + (int64_t)ki * kstride + 4 * g;
    af4 kf[NJ], vf[4][NH];
#pragma unroll
    for (int j = 0; j < NJ; ++j) kf[j] = *reinterpret_cast<const af4*>(kp + 16 * j);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool live = 16 * kt + 4 * g + r < nk;                    // a listed key: its V row is defined
      const float* vp = vbase + (int64_t)vk[r] * kstride + 4 * n;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        af4 v = *reinterpret_cast<const af4*>(vp + 64 * h);
        vf[r][h] = live ? v : zero;                                  // 0 * NaN of an unwritten cache row would poison the row
      }
    }
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
      af4 s = zero;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j][i], qf[t][j][i], s, 0, 0, 0);
      float tmax = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int slot = 16 * kt + 4 * g + r;
        float v = slot < nk ? s[r] * scale : -INFINITY;
        v = mask(v, slot, t);
        s[r] = v;
        tmax = fmaxf(tmax, v);
      }
      tmax = af_rows4_max(tmax);
      const float mnew = fmaxf(mrun[t], tmax);
      const bool none = mnew == -INFINITY;                           // nothing admitted so far for this row
      const float alpha = none ? 1.f : expf(mrun[t] - mnew);
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[r] = none ? 0.f : expf(s[r] - mnew);
        psum += s[r];
      }
      lrun[t] = lrun[t] * alpha + af_rows4_sum(psum);
      mrun[t] = mnew;
#pragma unroll
      for (int d = 0; d < NDT; ++d) acc[t][d] *= alpha;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[t][4 * h + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r][h][e], s[r], acc[t][4 * h + e], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < NQT; ++t) {
    if (out[t] == nullptr) continue;
    const float inv = 1.0f / lrun[t];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        af4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = acc[t][4 * h + e][r] * inv;
        *reinterpret_cast<af4*>(out[t] + 64 * h + 16 * g + 4 * r) = o;
      }
  }
}

struct AfNoMask {
  __device__ __forceinline__ float operator()(float s, int, int) const { return s; }
};

// appends the set bits of `word` (bit b = key base + b) to a wave's key list; returns the new count (wave-uniform)
__device__ __forceinline__ int af_append_bits(int32_t* keys, int cnt, uint64_t word, int base, int limit) {
  const int lane = threadIdx.x & 63;
  const bool on = ((word >> lane) & 1ull) && base + lane < limit;
  const uint64_t b = __ballot(on);
  if (on) keys[cnt + __popcll(b & ((1ull << lane) - 1ull))] = base + lane;
  return cnt + __popcll(b);
}
__device__ __forceinline__ int af_pad_list(int32_t* keys, int cnt) {   // pad to a multiple of 16 with a listed key
  const int lane = threadIdx.x & 63;
  const int up = (cnt + 15) & ~15;
  if (cnt + lane < up) keys[cnt + lane] = keys[0];
  return up;
}

// -------------------------------------------------------------------------------------------------------------
// K6.  Units: [0, PU): (pair p, head h) - rows 1 .. nq-1 of the pair in chunks of 32 (they share the pair's mask);
// [PU, PU + CU): (group of 16 pairs, head) - row 0 (the cls row) of 16 consecutive pairs as ONE query tile, the key
// list = the union of the 16 masks, each column masked by its own pair's bits.  nq = 1 (the selection phase of the
// last layer) has cls units only.  A pair whose mask union is empty attends uniformly (HF: every score is absorbed by
// finfo.min) or unmasked (legacy -10000 policy): its unit lists all L keys.
// -------------------------------------------------------------------------------------------------------------
#define AFX_MAXW 8   // bit words per object (L <= 512 patches)
struct AfClsMask {
  const uint64_t* w;         // LDS: this lane's pair (column n): union of its two objects' bits, AFX_MAXW words
  const int32_t* keys;
  bool empty, uniform;
  __device__ __forceinline__ float operator()(float s, int slot, int) const {
    const int k = keys[slot];
    const bool on = (w[k >> 6] >> (k & 63)) & 1ull;
    const float masked = on ? s : -INFINITY;
    const float e = uniform ? 0.f : s;
    return s == -INFINITY ? s : (empty ? e : masked);
  }
};
struct AfUniform {           // an empty pair under the 'uniform' policy: every listed key scores 0
  bool uniform;
  __device__ __forceinline__ float operator()(float s, int, int) const { return (uniform && s != -INFINITY) ? 0.f : s; }
};

__global__ void __launch_bounds__(256) cross_attn_f32_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ v, const uint64_t* __restrict__ bits,
                                                             int words, const int32_t* __restrict__ pair_index, int N,
                                                             int P, int L, int nq, int heads, int policy,
                                                             float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) int32_t af_smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, n = lane & 15;
  const int lcap = (L + 15) & ~15;
  int32_t* keys = af_smem + wid * lcap;
  const int64_t unit = (int64_t)blockIdx.x * 4 + wid;
  const int hidden = heads * 64;
  const int64_t PU = nq > 1 ? (int64_t)P * heads : 0;
  const int groups = (P + 15) >> 4;
  if (unit >= PU + (int64_t)groups * heads) return;
  const float scale = 0.125f;
  if (unit < PU) {
    const int p = (int)(unit / heads), h = (int)(unit % heads);
    const int pidx = pair_index[p];
    const uint64_t* bi = bits + (int64_t)(pidx / N) * words;
    const uint64_t* bj = bits + (int64_t)(pidx % N) * words;
    int cnt = 0;
    for (int wd = 0; wd < words; ++wd) cnt = af_append_bits(keys, cnt, bi[wd] | bj[wd], wd * 64, L);
    const bool empty = cnt == 0;
    if (empty) {
      for (int b = 0; b < L; b += 64)
        if (b + lane < L) keys[b + lane] = b + lane;
      cnt = L;
    }
    af_pad_list(keys, cnt);
    __builtin_amdgcn_wave_barrier();
    const AfUniform mk{empty && policy == PSG_EMPTY_UNIFORM};
    for (int r0 = 1; r0 < nq; r0 += 32) {
      const float* qrow[2];
      float* orow[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int row = r0 + 16 * t + n;
        const int rc = row < nq ? row : nq - 1;
        qrow[t] = q + ((int64_t)p * nq + rc) * hidden + h * 64;
        orow[t] = row < nq ? out + ((int64_t)p * nq + row) * hidden + h * 64 : nullptr;
      }
      if (r0 + 16 < nq)
        af_tiles<64, 2>(qrow, keys, cnt, k + h * 64, v + h * 64, hidden, scale, mk, orow);
      else {
        const float* q1[1] = {qrow[0]};
        float* o1[1] = {orow[0]};
        af_tiles<64, 1>(q1, keys, cnt, k + h * 64, v + h * 64, hidden, scale, mk, o1);
      }
    }
    return;
  }
  // cls unit
  const int64_t cu = unit - PU;
  const int gi = (int)(cu / heads), h = (int)(cu % heads);
  const int p = gi * 16 + n;
  const int pc = p < P ? p : P - 1;
  const int pidx = pair_index[pc];
  uint64_t* mw = reinterpret_cast<uint64_t*>(af_smem + 4 * lcap) + (wid * 16 + n) * AFX_MAXW;   // [16 pairs][AFX_MAXW]
  bool any = false;
  const int g = lane >> 4;
  for (int i = g; i < AFX_MAXW; i += 4) {                            // the four lane groups share a pair's words
    uint64_t w = i < words ? (bits[(int64_t)(pidx / N) * words + i] | bits[(int64_t)(pidx % N) * words + i]) : 0ull;
    if (i == (L >> 6) && (L & 63)) w &= (1ull << (L & 63)) - 1ull;
    if (i > (L >> 6) || (i == (L >> 6) && !(L & 63))) w = 0ull;
    mw[i] = w;
    any |= w != 0ull;
  }
  // a pair is empty when none of its words (held by the lanes n, n + 16, n + 32, n + 48) has a bit
  const uint64_t anyb = __ballot(any);
  const bool pair_any = ((anyb | (anyb >> 16) | (anyb >> 32) | (anyb >> 48)) >> n) & 1ull;
  const bool some_empty = ((anyb | (anyb >> 16) | (anyb >> 32) | (anyb >> 48)) & 0xffffull) != 0xffffull;
  __builtin_amdgcn_wave_barrier();
  AfClsMask mk;
  mk.w = mw;
  mk.keys = keys;
  mk.empty = !pair_any;
  mk.uniform = policy == PSG_EMPTY_UNIFORM;
  // Key list of a cls unit: ALIGNED tiles of 16 keys in natural order, those in which any pair of the group has a key.
  // A tile without a key of pair n leaves that column's state untouched bit for bit (max, sum and accumulators are
  // multiplied by exp(0) and get + 0), so a pair's result does not depend on which pairs share its group - a pair shard
  // reproduces the full pass exactly.
  int cnt = 0;
  if (some_empty) {
    for (int b = 0; b < L; b += 64)
      if (b + lane < L) keys[b + lane] = b + lane;
    cnt = L;
  } else {
    const uint64_t* gw = reinterpret_cast<const uint64_t*>(af_smem + 4 * lcap) + (size_t)wid * 16 * AFX_MAXW;
    for (int wd = 0; wd < words; ++wd) {
      uint64_t u = 0ull;                                              // union over the 16 pairs of the group
#pragma unroll
      for (int i = 0; i < 16; ++i) u |= gw[i * AFX_MAXW + wd];
      // spread every 16-bit field that has a bit over the whole field
      uint64_t t = u | (u >> 1);
      t |= t >> 2;
      t |= t >> 4;
      t |= t >> 8;                                                    // bit 16 f = OR of field f
      t &= 0x0001000100010001ull;
      t = (t << 16) - t;                                              // 0xffff per flagged field
      const bool on = ((t >> lane) & 1ull) && wd * 64 + lane < L;     // keys past L only shorten the very last tile
      const uint64_t bal = __ballot(on);
      if (on) keys[cnt + __popcll(bal & ((1ull << lane) - 1ull))] = wd * 64 + lane;
      cnt += __popcll(bal);
    }
  }
  af_pad_list(keys, cnt);
  __builtin_amdgcn_wave_barrier();
  const float* q1[1] = {q + (int64_t)pc * nq * hidden + h * 64};
  float* o1[1] = {p < P ? out + (int64_t)p * nq * hidden + h * 64 : nullptr};
  af_tiles<64, 1>(q1, keys, cnt, k + h * 64, v + h * 64, hidden, scale, mk, o1);
}

int psg_cross_attn_f32_launch(const void* q, const void* k, const void* v, const uint64_t* bits, int words,
                              const int32_t* pair_index, int N, int P, int L, int nq, int heads, int policy, void* out,
                              hipStream_t st) {
  PSG_REQUIRE(words <= AFX_MAXW && L <= 64 * AFX_MAXW, PSG_ERR_UNSUPPORTED, "cross_attn(f32): L=%d (up to %d patches)", L,
              64 * AFX_MAXW);
  const int lcap = (L + 15) & ~15;
  const int64_t units = (nq > 1 ? (int64_t)P * heads : 0) + (int64_t)((P + 15) / 16) * heads;
  const size_t lds = (size_t)4 * lcap * sizeof(int32_t) + (size_t)4 * 16 * AFX_MAXW * sizeof(uint64_t);
  cross_attn_f32_kernel<<<(unsigned)((units + 3) / 4), 256, lds, st>>>(
      (const float*)q, (const float*)k, (const float*)v, bits, words, pair_index, N, P, L, nq, heads, policy, (float*)out);
  PSG_CHECK_LAUNCH("psg_qformer_cross_attn(f32)");
  return PSG_OK;
}

// -------------------------------------------------------------------------------------------------------------
// K5.  Unit = (pair, head): keys = the nq query rows + the text rows the mask admits (<= 64), queries = every row
// (q_only 0), the nq query rows (1) or the cls row alone, written compactly to out[p] (2).  q_shared: the query rows'
// Q | K | V are ONE [nq][3 hidden] block for all pairs (layer 0, where they are the embedded learned queries) and `qkv`
// holds the text rows only.  The rows of a pair live in two row ranges (or two tensors), so the key list holds sequence
// POSITIONS and a per-wave LDS table maps a position to its row's element offset from `qkv`.
// -------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) qformer_self_attn_f32_kernel(const float* __restrict__ qkv,
                                                                    const float* __restrict__ q_shared,
                                                                    const uint8_t* __restrict__ text_mask, int B, int Tt,
                                                                    int nq, int heads, int q_only,
                                                                    float* __restrict__ out) {
  __shared__ int32_t s_keys[4][64];
  __shared__ int64_t s_off[4][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
  const int64_t unit = (int64_t)blockIdx.x * 4 + wid;
  if (unit >= (int64_t)B * heads) return;
  const int p = (int)(unit / heads), h = (int)(unit % heads);
  const int hidden = heads * 64, rs = 3 * hidden;
  const int S = nq + Tt;
  int32_t* keys = s_keys[wid];
  int64_t* off = s_off[wid];
  const float* qbase = q_shared ? q_shared : qkv;
  const int64_t qrow0 = q_shared ? 0 : (int64_t)p * nq;
  const int64_t trow0 = q_shared ? (int64_t)p * Tt : (int64_t)B * nq + (int64_t)p * Tt;
  bool valid = lane < S;
  if (valid && lane >= nq) valid = text_mask[(int64_t)p * Tt + (lane - nq)] != 0;
  const uint64_t vb = __ballot(valid);
  const int cnt = __popcll(vb);                                        // >= nq: the query rows are always keys
  if (valid) keys[__popcll(vb & ((1ull << lane) - 1ull))] = lane;
  if (lane < S)
    off[lane] = lane < nq ? (qbase + (qrow0 + lane) * rs) - qkv : (trow0 + (lane - nq)) * (int64_t)rs;
  const int up = af_pad_list(keys, cnt);
  __builtin_amdgcn_wave_barrier();
  const int nrows = q_only == 2 ? 1 : (q_only ? nq : S);
  const float scale = 0.125f;
  const af4 zero = {0.f, 0.f, 0.f, 0.f};
  for (int r0 = 0; r0 < nrows; r0 += 32) {
    const bool two = r0 + 16 < nrows;                                  // a second query tile shares the key loop
    af4 qf[2][4], acc[2][4];
    float mrun[2], lrun[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = r0 + 16 * t + n;
      const float* qp = qkv + off[row < nrows ? row : nrows - 1] + h * 64 + 4 * g;
#pragma unroll
      for (int j = 0; j < 4; ++j) qf[t][j] = *reinterpret_cast<const af4*>(qp + 16 * j);
#pragma unroll
      for (int d = 0; d < 4; ++d) acc[t][d] = zero;
      mrun[t] = -INFINITY;
      lrun[t] = 0.f;
    }
    for (int kt = 0; kt < (up >> 4); ++kt) {
      const float* kp = qkv + off[keys[16 * kt + n]] + hidden + h * 64 + 4 * g;
      af4 kf[4], vf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) kf[j] = *reinterpret_cast<const af4*>(kp + 16 * j);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        vf[r] = *reinterpret_cast<const af4*>(qkv + off[keys[16 * kt + 4 * g + r]] + 2 * hidden + h * 64 + 4 * n);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t == 1 && !two) continue;
        af4 s = zero;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j][i], qf[t][j][i], s, 0, 0, 0);
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[r] = 16 * kt + 4 * g + r < cnt ? s[r] * scale : -INFINITY;
          tmax = fmaxf(tmax, s[r]);
        }
        tmax = af_rows4_max(tmax);
        const float mnew = fmaxf(mrun[t], tmax);                       // finite: every tile holds a listed key
        const float alpha = expf(mrun[t] - mnew);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[r] = expf(s[r] - mnew);
          psum += s[r];
        }
        lrun[t] = lrun[t] * alpha + af_rows4_sum(psum);
        mrun[t] = mnew;
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[t][d] *= alpha;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[t][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r][e], s[r], acc[t][e], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = r0 + 16 * t + n;
      if (row >= nrows || (t == 1 && !two)) continue;
      // output rows follow the layout of the layer's activations (query rows of all pairs, then text rows)
      const int64_t orow = q_only == 2 ? (int64_t)p
                                       : (row < nq ? (int64_t)p * nq + row : (int64_t)B * nq + (int64_t)p * Tt + (row - nq));
      float* op = out + orow * hidden + h * 64;
      const float inv = 1.0f / lrun[t];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        af4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = acc[t][e][r] * inv;
        *reinterpret_cast<af4*>(op + 16 * g + 4 * r) = o;
      }
    }
  }
}

int psg_self_attn_f32_launch(const void* qkv, const void* q_shared, const uint8_t* text_mask, int B, int T_, int nq,
                             int heads, int query_rows_only, void* out, hipStream_t st) {
  const int64_t units = (int64_t)B * heads;
  qformer_self_attn_f32_kernel<<<(unsigned)((units + 3) / 4), 256, 0, st>>>(
      (const float*)qkv, (const float*)q_shared, text_mask, B, T_, nq, heads, query_rows_only, (float*)out);
  PSG_CHECK_LAUNCH("psg_qformer_self_attn(f32)");
  return PSG_OK;
}

// -------------------------------------------------------------------------------------------------------------
// K14, prompt pass.  Unit = (pair, head, query tile of 16 rows): causal attention of a pair-major prompt batch over the
// pair's cache slots; sequences are compact (slot = position; rows past a pair's length carry pos = -1 and give
// zeros).  q [pairs * rpp][heads * 128] rotated queries; caches [pairs][heads][ctx][128], written by psg_rope_kvwrite.
// -------------------------------------------------------------------------------------------------------------
struct AfCausal {
  int qpos;                  // position of this lane's query row (row n of the tile)
  __device__ __forceinline__ float operator()(float s, int slot, int) const { return slot <= qpos ? s : -INFINITY; }
};

__global__ void __launch_bounds__(256) prefill_attn_f32_kernel(const float* __restrict__ q, const float* __restrict__ kc,
                                                               const float* __restrict__ vc,
                                                               const int32_t* __restrict__ tok_pos, int pairs, int rpp,
                                                               int heads, int ctx, float* __restrict__ out) {
  __shared__ int32_t s_keys[4][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, n = lane & 15;
  const int qtiles = (rpp + 15) >> 4;
  const int64_t unit = (int64_t)blockIdx.x * 4 + wid;
  if (unit >= (int64_t)pairs * heads * qtiles) return;
  // the LAST query tiles (most keys) are dealt first
  const int qt = qtiles - 1 - (int)(unit / ((int64_t)pairs * heads));
  const int64_t ph = unit % ((int64_t)pairs * heads);
  const int p = (int)(ph / heads), h = (int)(ph % heads);
  const int hidden = heads * 128;
  int32_t* keys = s_keys[wid];
  keys[lane] = min(lane, ctx - 1);                                     // identity: key slot = cache slot
  __builtin_amdgcn_wave_barrier();
  const int row = qt * 16 + n;
  const int pos = row < rpp ? tok_pos[(int64_t)p * rpp + row] : -1;
  // rows of a pair are compact: valid rows first.  Keys this tile can see: slots 0 .. max position of its rows
  const int seen = (int)__popcll(__ballot(pos >= 0 && lane < 16));
  float* orow = row < rpp ? out + ((int64_t)p * rpp + row) * hidden + h * 128 : nullptr;
  if (seen == 0) {                                                     // padding rows only: defined output, never consumed
    if (orow) {
      const af4 zero = {0.f, 0.f, 0.f, 0.f};
      const int g = lane >> 4;
      *reinterpret_cast<af4*>(orow + 16 * g) = zero;
      *reinterpret_cast<af4*>(orow + 16 * g + 4) = zero;
      *reinterpret_cast<af4*>(orow + 16 * g + 8) = zero;
      *reinterpret_cast<af4*>(orow + 16 * g + 12) = zero;
      *reinterpret_cast<af4*>(orow + 64 + 16 * g) = zero;
      *reinterpret_cast<af4*>(orow + 64 + 16 * g + 4) = zero;
      *reinterpret_cast<af4*>(orow + 64 + 16 * g + 8) = zero;
      *reinterpret_cast<af4*>(orow + 64 + 16 * g + 12) = zero;
    }
    return;
  }
  const int nk = qt * 16 + seen;                                       // slots [0, nk) are written cache rows
  const int rc = min(row, qt * 16 + seen - 1);
  const float* q1[1] = {q + ((int64_t)p * rpp + rc) * hidden + h * 128};
  float* o1[1] = {orow};
  const int64_t cbase = ((int64_t)p * heads + h) * ctx * 128;
  const AfCausal mk{pos >= 0 ? pos : 1 << 30};                         // a padding row of a mixed tile: any finite result
  af_tiles<128, 1>(q1, keys, nk, kc + cbase, vc + cbase, 128, 0.08838834764831845f, mk, o1);
  if (pos < 0 && orow) {                                               // padding rows of a mixed tile: zeros, as the scalar kernel
    const af4 zero = {0.f, 0.f, 0.f, 0.f};
    const int g = lane >> 4;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int r = 0; r < 4; ++r) *reinterpret_cast<af4*>(orow + 64 * hh + 16 * g + 4 * r) = zero;
  }
}

int psg_prefill_attn_f32_launch(const void* q, const void* kc, const void* vc, const int32_t* tok_pos, int pairs, int rpp,
                                int heads, int ctx, void* out, hipStream_t st) {
  PSG_REQUIRE(rpp >= 1 && rpp <= 64, PSG_ERR_UNSUPPORTED, "psg_prefill_attn(f32): %d rows per pair (1..64)", rpp);
  const int64_t units = (int64_t)pairs * heads * ((rpp + 15) / 16);
  prefill_attn_f32_kernel<<<(unsigned)((units + 3) / 4), 256, 0, st>>>((const float*)q, (const float*)kc, (const float*)vc,
                                                                       tok_pos, pairs, rpp, heads, ctx, (float*)out);
  PSG_CHECK_LAUNCH("psg_prefill_attn(f32)");
  return PSG_OK;
}
