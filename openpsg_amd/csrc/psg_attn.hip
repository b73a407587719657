// Attention kernels that are not the MFMA cross-attention:
//   K5  Q-Former self-attention (S = 33 + T <= 64 keys, 12 heads x 64)      HF-IB:471-515
//   K6' scalar fp32 cross-attention (checker / fp32 verification variant)     HF-IB:487-496
//   K14 Llama attention over the KV cache, prefill + decode (head_dim 128)    HF-LL:191-214
#include <stdlib.h>

#include <type_traits>

#include "psg_common.h"
#include "psg_decode_math.h"

int psg_self_attn_mfma_launch(const void* qkv, const void* q_shared, const uint8_t* text_mask, int B, int T_, int nq, int heads,
                              int query_rows_only, void* out, int dtype, hipStream_t st);
int psg_self_attn_f32_launch(const void* qkv, const void* q_shared, const uint8_t* text_mask, int B, int T_, int nq,
                             int heads, int query_rows_only, void* out, hipStream_t st);

__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// ---------------------------------------------------------------------------------------------
// K5.  One wave per (pair, head).  Lane j owns key j: K[j][0..63] lives in 64 registers of lane j,
// V[j][d] for all j lives in lane d.  Per query row: 64 readlane+fma for q.K^T, a wave softmax,
// 64 readlane+fma for P.V.  No LDS, no inter-wave traffic.
// ---------------------------------------------------------------------------------------------
// q_cls != nullptr (cls-only mode): queries come from the compact [B][hidden] tensor q_cls, and `qkv` holds K | V only
// ([rows][2*hidden]: the last layer's selection phase never projects the queries of rows 1..32).
template <typename T>
__global__ void __launch_bounds__(256) qformer_self_attn_kernel(const T* __restrict__ qkv,
                                                                const uint8_t* __restrict__ text_mask, int B, int Tt,
                                                                int nq, int heads, int q_only, T* __restrict__ out,
                                                                const T* __restrict__ q_cls = nullptr) {
  const int unit = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  if (unit >= B * heads) return;
  const int p = unit / heads, h = unit % heads;
  const int hidden = heads * 64;
  const int S = nq + Tt;
  const int64_t qrow0 = (int64_t)p * nq;                      // query rows of this pair
  const int64_t trow0 = (int64_t)B * nq + (int64_t)p * Tt;    // text rows of this pair
  auto row_of = [&](int j) -> int64_t { return j < nq ? qrow0 + j : trow0 + (j - nq); };
  const int rs = q_cls ? 2 * hidden : 3 * hidden;             // row stride of `qkv`
  const int koff = q_cls ? 0 : hidden, voff = q_cls ? hidden : 2 * hidden;

  // K^T: lane j reads its own key row (64 contiguous elements)
  float kreg[64];
  bool valid = lane < S;
  if (valid && lane >= nq) valid = text_mask[(int64_t)p * Tt + (lane - nq)] != 0;
  {
    const int64_t r = row_of(lane < S ? lane : 0);
    const T* kp = qkv + r * rs + koff + h * 64;
#pragma unroll
    for (int d = 0; d < 64; d += 4) {
      float t[4];
      Act<T>::ld4(kp, d, t);
      kreg[d] = t[0]; kreg[d + 1] = t[1]; kreg[d + 2] = t[2]; kreg[d + 3] = t[3];
    }
  }
  // V: lane d reads column d of every key row (coalesced)
  float vreg[64];
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    vreg[j] = 0.f;
    if (j < S) vreg[j] = Act<T>::ld(qkv, row_of(j) * rs + voff + h * 64 + lane);
  }
  // q_only: 0 = every row, 1 = the nq query rows, 2 = the cls row (row 0) only, written COMPACT to out[p]
  const int nrows = q_only == 2 ? 1 : (q_only ? nq : S);
  for (int i = 0; i < nrows; ++i) {
    const int64_t r = row_of(i);
    const float qv = q_cls ? Act<T>::ld(q_cls, (int64_t)p * hidden + h * 64 + lane) : Act<T>::ld(qkv, r * 3 * hidden + h * 64 + lane);
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < 64; ++d) s = fmaf(readlane_f(qv, d), kreg[d], s);
    s *= 0.125f;                                  // 1/sqrt(64)
    s = valid ? s : PSG_FMIN;                     // additive finfo.min absorbs the score
    if (lane >= S) s = -INFINITY;                 // not a key at all
    const float m = wave_max(s);
    float pr = expf(s - m);
    const float denom = wave_sum(pr);
    pr = pr / denom;
    pr = Act<T>::rnd(pr);
    float o = 0.f;
#pragma unroll
    for (int j = 0; j < 64; ++j)
      if (j < S) o = fmaf(readlane_f(pr, j), vreg[j], o);
    Act<T>::st(out, (q_only == 2 ? (int64_t)p : r) * hidden + h * 64 + lane, o);
  }
}

extern "C" int psg_qformer_self_attn(psg_ctx* ctx, const void* qkv, const uint8_t* text_mask, int B, int T_, int nq,
                                     int heads, int query_rows_only, void* out, int dtype, void* stream) {
  PSG_REQUIRE(ctx && qkv && out && (text_mask || T_ == 0), PSG_ERR_INVALID, "psg_qformer_self_attn: NULL argument");
  PSG_REQUIRE(B > 0 && T_ >= 0 && nq > 0 && heads > 0, PSG_ERR_INVALID, "psg_qformer_self_attn: B=%d T=%d nq=%d", B,
              T_, nq);
  PSG_REQUIRE(nq + T_ <= 64, PSG_ERR_UNSUPPORTED,
              "psg_qformer_self_attn: %d query rows + %d prompt tokens > 64 keys (one wavefront)", nq, T_);
  // bf16 activations with the standard geometry run on the matrix cores (psg_selfattn_mfma.hip);
  // option selfattn_scalar forces the scalar kernel (on-device cross-check)
  const int force_scalar = ctx->opt.selfattn_scalar;
  if ((dtype == PSG_BF16 || dtype == PSG_F16) && nq >= 32 && nq + T_ <= 64 && !force_scalar)
    return psg_self_attn_mfma_launch(qkv, nullptr, text_mask, B, T_, nq, heads, query_rows_only, out, dtype, (hipStream_t)stream);
  if (dtype == PSG_F32 && !force_scalar)                    // exact f32 matrix instructions (psg_attn_f32.hip)
    return psg_self_attn_f32_launch(qkv, nullptr, text_mask, B, T_, nq, heads, query_rows_only, out, (hipStream_t)stream);
  int64_t units = (int64_t)B * heads;
  PSG_DISPATCH_DTYPE(dtype, "psg_qformer_self_attn",
                     (qformer_self_attn_kernel<T><<<(unsigned)((units + 3) / 4), 256, 0, (hipStream_t)stream>>>(
                         (const T*)qkv, text_mask, B, T_, nq, heads, query_rows_only, (T*)out)));
  PSG_CHECK_LAUNCH("psg_qformer_self_attn");
  return PSG_OK;
}

// cls-row attention for the last layer's selection phase, streaming version: ONE wave per pair, lane l < hidden/16 owns
// 16 consecutive features (4 lanes = one head), so a key row is one 2 x 16-byte (bf16) request per lane - the scalar
// kernel above, run per (pair, head) with 2-byte value gathers, took 285 us for 2500 pairs; this one is bound by the
// 361 MB of K | V it reads.  Scores of a head live in the registers of its 4 lanes (quad reduce, no wave reductions),
// the output features accumulate in the lane that stores them; 8 key rows are requested per step before the first
// one is used; the probabilities pass through a per-wave LDS table.
template <typename T>
__global__ void __launch_bounds__(256) qformer_self_attn_cls_kernel(const T* __restrict__ q_cls, const T* __restrict__ kv,
                                                                    const uint8_t* __restrict__ text_mask, int B, int Tt,
                                                                    int nq, int heads, T* __restrict__ out) {
  __shared__ float s_sc[4][16][64];                             // [wave][quad = head][key]
  const int p = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (p >= B) return;                                           // whole wave
  const int hidden = heads * 64, S = nq + Tt;
  const bool on = lane < hidden / 16;
  const int e0 = (on ? lane : 0) * 16;
  const int64_t qrow0 = (int64_t)p * nq, trow0 = (int64_t)B * nq + (int64_t)p * Tt;
  auto row_of = [&](int j) -> int64_t {
    j = j < S ? j : S - 1;
    return j < nq ? qrow0 + j : trow0 + (j - nq);
  };
  bool kvalid = lane < S;
  if (kvalid && lane >= nq) kvalid = text_mask[(int64_t)p * Tt + (lane - nq)] != 0;
  const unsigned long long valid64 = __ballot(kvalid);
  float qv[16];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float t[4];
    Act<T>::ld4(q_cls, (int64_t)p * hidden + e0 + 4 * c, t);
#pragma unroll
    for (int e = 0; e < 4; ++e) qv[4 * c + e] = t[e];
  }
  float* my = &s_sc[wv][lane >> 2][0];
  const int nch = (S + 7) >> 3;
#pragma unroll 1
  for (int ch = 0; ch < nch; ++ch) {
    typename Act<T>::raw4 kr[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const T* kp = kv + row_of(ch * 8 + u) * 2 * hidden + e0;
#pragma unroll
      for (int c = 0; c < 4; ++c) kr[u][c] = Act<T>::ldr4(kp, 4 * c);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float t[4];
        Act<T>::cv4(kr[u][c], t);
#pragma unroll
        for (int e = 0; e < 4; ++e) a = fmaf(qv[4 * c + e], t[e], a);
      }
      a = quad_sum(a);
      const int j = ch * 8 + u;
      a *= 0.125f;                                              // 1/sqrt(64)
      a = ((valid64 >> j) & 1ull) ? a : PSG_FMIN;               // additive finfo.min absorbs the score
      if ((lane & 3) == 0) my[j] = j < S ? a : -INFINITY;       // j >= S: not a key at all
    }
  }
  __builtin_amdgcn_wave_barrier();
  const int nk = nch * 8;
  float m = -INFINITY;
  for (int j = 0; j < nk; ++j) m = fmaxf(m, my[j]);
  float denom = 0.f;
  for (int j = 0; j < nk; ++j) denom += expf(my[j] - m);
  __builtin_amdgcn_wave_barrier();
  if ((lane & 3) == 0)
    for (int j = 0; j < nk; ++j) my[j] = Act<T>::rnd(expf(my[j] - m) / denom);
  __builtin_amdgcn_wave_barrier();
  float o[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) o[e] = 0.f;
#pragma unroll 1
  for (int ch = 0; ch < nch; ++ch) {
    typename Act<T>::raw4 vr[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const T* vp = kv + row_of(ch * 8 + u) * 2 * hidden + hidden + e0;
#pragma unroll
      for (int c = 0; c < 4; ++c) vr[u][c] = Act<T>::ldr4(vp, 4 * c);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float pj = my[ch * 8 + u];                          // 0 for j >= S (exp(-inf))
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float t[4];
        Act<T>::cv4(vr[u][c], t);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[4 * c + e] = fmaf(pj, t[e], o[4 * c + e]);
      }
    }
  }
  if (on) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float t[4] = {o[4 * c], o[4 * c + 1], o[4 * c + 2], o[4 * c + 3]};
      Act<T>::st4(out, (int64_t)p * hidden + e0 + 4 * c, t);
    }
  }
}

// Last-layer, first phase: only the cls row (row 0) of every pair feeds the existence head (V4:206-209), so its
// attention over the pair's nq + T keys is all the selection needs; out [B][hidden] is compact (one row per pair).
// The rows 1..32 of the SELECTED pairs are computed afterwards by the ordinary kernel on the gathered pairs.
extern "C" int psg_qformer_self_attn_cls(psg_ctx* ctx, const void* q_cls, const void* kv, const uint8_t* text_mask, int B,
                                         int T_, int nq, int heads, void* out, int dtype, void* stream) {
  PSG_REQUIRE(ctx && q_cls && kv && out && (text_mask || T_ == 0), PSG_ERR_INVALID,
              "psg_qformer_self_attn_cls: NULL argument");
  PSG_REQUIRE(B > 0 && T_ >= 0 && nq > 0 && heads > 0 && nq + T_ <= 64, PSG_ERR_INVALID,
              "psg_qformer_self_attn_cls: B=%d T=%d nq=%d", B, T_, nq);
  if (heads * 64 <= 1024) {                                     // one wave per pair, 16 features per lane
    PSG_DISPATCH_DTYPE(dtype, "psg_qformer_self_attn_cls",
                       (qformer_self_attn_cls_kernel<T><<<(unsigned)((B + 3) / 4), 256, 0, (hipStream_t)stream>>>(
                           (const T*)q_cls, (const T*)kv, text_mask, B, T_, nq, heads, (T*)out)));
    PSG_CHECK_LAUNCH("psg_qformer_self_attn_cls");
    return PSG_OK;
  }
  int64_t units = (int64_t)B * heads;
  PSG_DISPATCH_DTYPE(dtype, "psg_qformer_self_attn_cls",
                     (qformer_self_attn_kernel<T><<<(unsigned)((units + 3) / 4), 256, 0, (hipStream_t)stream>>>(
                         (const T*)kv, text_mask, B, T_, nq, heads, 2, (T*)out, (const T*)q_cls)));
  PSG_CHECK_LAUNCH("psg_qformer_self_attn_cls");
  return PSG_OK;
}

// The same attention with the key / value projections folded into the INPUT space (no K | V tensor at all).  For the
// cls query q of a pair and head h, score_j = q_h . (W_k,h x_j + b_k,h) / 8 = (W_k,h^T q_h) . x_j / 8 + const_h, and the
// constant drops out of the softmax; context_h = sum_j p_j (W_v,h x_j + b_v,h) = W_v,h (sum_j p_j x_j) + b_v,h.  So the
// caller projects the queries back through W_k (g = W_k,h^T q_h: [heads][B][hidden] in fp32 - a 768-term dot product
// of rounded factors would lose what the 64-term one keeps -, two tiny batched GEMMs around this kernel) and the kernel needs the layer's INPUT rows only: per pair it reads (nq + T) x hidden values once instead of
// twice that of K | V, and the K | V projection of every row of every pair (277 GFLOP at 2500 pairs: the largest GEMM
// of the selection phase) disappears.  One workgroup per pair: the rows are staged in LDS once (73 KB in bf16); the
// scores g_h . x_j and the weighted row means xbar[h] = sum_j p_j x_j both run on the matrix cores (16-bit rows; per-lane
// arithmetic with wave reductions in the fp32 verification mode), the probabilities pass through an LDS table.  Built for the Q-Former geometry (hidden 768 = 12 x 64).
template <typename T> struct ClsE { using type = void; };
template <> struct ClsE<bf16_t> { using type = EBf16; };
template <> struct ClsE<f16_t> { using type = EF16; };

template <typename T>
__global__ void __launch_bounds__(256) qformer_cls_attn_input_kernel(const T* __restrict__ x, const T* __restrict__ xt,
                                                                     const int32_t* __restrict__ text_index,
                                                                     const float* __restrict__ g,
                                                                     const uint8_t* __restrict__ text_mask, int B, int Tt,
                                                                     int nq, float* __restrict__ xbar) {
  constexpr int H = 768, HPW = 3, NK = 3;
  constexpr bool MM = !std::is_same<T, float>::value;             // 16-bit rows: scores on the matrix cores
  constexpr int RSE = H + (MM ? 8 : 0);                           // LDS row stride in elements (+16 B: conflict-free fragment reads)
  extern __shared__ __attribute__((aligned(16))) unsigned char cls_smem[];
  const int S = nq + Tt;
  T* xs = reinterpret_cast<T*>(cls_smem);                                       // [S][RSE]
  float* sc = reinterpret_cast<float*>(cls_smem + (size_t)S * RSE * sizeof(T));  // [12][64]
  const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // text rows: block `ti` of xt (the pair's own block, or the block of its prompt when the text rows are shared by all
  // pairs with the same prompt: text_index); query rows: block p of x
  const int ti = text_index ? text_index[p] : p;
  const int64_t qrow0 = (int64_t)p * nq, trow0 = (int64_t)ti * Tt;
  constexpr int EPC = 16 / (int)sizeof(T), CPR = H / EPC;
  const int nchunk = S * CPR;
  // 16-bit rows: the first GD steps of this lane's g fragments (A operand of the score product, see below) are
  // requested before the rows are staged, the rest GD steps ahead of their use
  constexpr int GD = 12;
  const float* gp = g + ((int64_t)((lane & 15) < 12 ? (lane & 15) : 0) * B + p) * H + (lane >> 4) * 8;
  float4 gbuf[MM ? GD : 1][2];
  if constexpr (MM) {
#pragma unroll
    for (int ks = 0; ks < GD; ++ks) {
      gbuf[ks][0] = *reinterpret_cast<const float4*>(gp + ks * 32);
      gbuf[ks][1] = *reinterpret_cast<const float4*>(gp + ks * 32 + 4);
    }
  }
  for (int i0 = tid; i0 < nchunk; i0 += 256 * 6) {               // six 16-byte requests per thread and round trip
    uint4 v0, v1, v2, v3, v4, v5;                                 // (all 18 at once measured the same: 103 us)
    auto ldc = [&](int i) {                                       // chunk i (clamped: the tail re-reads the last chunk)
      i = i < nchunk ? i : nchunk - 1;
      const int j = i / CPR, c = i - j * CPR;
      const T* src = j < nq ? x + (qrow0 + j) * H : xt + (trow0 + (j - nq)) * H;
      return *reinterpret_cast<const uint4*>(src + c * EPC);
    };
    v0 = ldc(i0); v1 = ldc(i0 + 256); v2 = ldc(i0 + 512); v3 = ldc(i0 + 768); v4 = ldc(i0 + 1024); v5 = ldc(i0 + 1280);
    auto stc = [&](int i, const uint4& v) {
      if (i < nchunk) {
        const int j = i / CPR, c = i - j * CPR;
        *reinterpret_cast<uint4*>(xs + j * RSE + c * EPC) = v;
      }
    };
    stc(i0, v0); stc(i0 + 256, v1); stc(i0 + 512, v2); stc(i0 + 768, v3); stc(i0 + 1024, v4); stc(i0 + 1280, v5);
  }
  bool kvalid = lane < S;
  if (kvalid && lane >= nq) kvalid = text_mask[(int64_t)ti * Tt + (lane - nq)] != 0;
  const unsigned long long valid64 = __ballot(kvalid);
  if constexpr (MM) {
    // Scores on the matrix cores: wave w owns keys 16 w .. 16 w + 15 (S <= 64), D[head][key] = sum_c G[head][c] X[key][c] over
    // 24 steps of 32 features.  A = G (rows 12..15 zero) is read from global as fp32 and split into a 16-bit head and
    // a 16-bit remainder (two matrix instructions per step: the product keeps ~16 bits of g), B = the staged rows.
    using E = typename ClsE<T>::type;
    using v8 = typename E::v8;
    __syncthreads();
    if (wv * 16 < S) {
      const int n = lane & 15, kq = lane >> 4;
      const int jrow = wv * 16 + n < S ? wv * 16 + n : S - 1;     // rows past S: duplicates, their columns are never read
      const unsigned char* xrow = reinterpret_cast<const unsigned char*>(xs + jrow * RSE) + kq * 16;
      const bool hv = n < 12;
      psg_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < H / 32; ++ks) {
        const float4 ga = gbuf[ks % GD][0], gb = gbuf[ks % GD][1];
        if (ks + GD < H / 32) {                                    // refill the slot GD steps ahead (an L2 round trip)
          gbuf[ks % GD][0] = *reinterpret_cast<const float4*>(gp + (ks + GD) * 32);
          gbuf[ks % GD][1] = *reinterpret_cast<const float4*>(gp + (ks + GD) * 32 + 4);
        }
        const float gv[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float a0 = hv ? gv[2 * i] : 0.f, a1 = hv ? gv[2 * i + 1] : 0.f;
          hi[i] = E::pack(a0, a1);
          lo[i] = E::pack(a0 - E::to_f32((uint16_t)(hi[i] & 0xffffu)), a1 - E::to_f32((uint16_t)(hi[i] >> 16)));
        }
        const uint4 hq = make_uint4(hi[0], hi[1], hi[2], hi[3]), lq = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        const v8 bx = *reinterpret_cast<const v8*>(xrow + ks * 64);
        acc = E::mfma16(__builtin_bit_cast(v8, hq), bx, acc);
        acc = E::mfma16(__builtin_bit_cast(v8, lq), bx, acc);
      }
      const int j = wv * 16 + n;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int h = 4 * kq + r;
        const float v = ((valid64 >> j) & 1ull) ? acc[r] * 0.125f : PSG_FMIN;   // 1/sqrt(64); finfo.min absorbs the score
        if (h < 12) sc[h * 64 + j] = v;
      }
    }
    __syncthreads();
  } else {
    float gq[HPW][NK][4];
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const float4 t = *reinterpret_cast<const float4*>(g + ((int64_t)(wv * HPW + hh) * B + p) * H + lane * 4 + 256 * k);
        gq[hh][k][0] = t.x; gq[hh][k][1] = t.y; gq[hh][k][2] = t.z; gq[hh][k][3] = t.w;
      }
    __syncthreads();
    // fp32 verification mode: wave w owns heads 3w..3w+2, a lane owns features {4 lane + 256 k}, one wave reduction per
    // (head, key)
#pragma unroll 1
    for (int j = 0; j < S; ++j) {
      float xv[NK][4];
#pragma unroll
      for (int k = 0; k < NK; ++k) Act<T>::ld4(xs, j * RSE + lane * 4 + 256 * k, xv[k]);
#pragma unroll
      for (int hh = 0; hh < HPW; ++hh) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
          for (int e = 0; e < 4; ++e) a = fmaf(gq[hh][k][e], xv[k][e], a);
        a = wave_sum_dpp(a) * 0.125f;                               // 1/sqrt(64)
        a = ((valid64 >> j) & 1ull) ? a : PSG_FMIN;                 // additive finfo.min absorbs the score
        if (lane == 0) sc[(wv * HPW + hh) * 64 + j] = a;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int hh = 0; hh < HPW; ++hh) {
    float* my = sc + (wv * HPW + hh) * 64;
    const float v = lane < S ? my[lane] : -INFINITY;
    const float m = wave_max(v);
    const float e = expf(v - m);                                  // 0 for lane >= S
    const float denom = wave_sum(e);
    my[lane] = Act<T>::rnd(e / denom);
  }
  if constexpr (MM) {
    // Weighted row means on the matrix cores: D[head][feature] = sum_j P[head][j] X[j][feature], contraction over the
    // keys in one or two steps of 32; wave w owns features 192 w .. 192 w + 191 (12 tiles of 16) for all heads.  The B
    // fragment wants 8 consecutive KEYS of one feature, i.e. a column of the row-major LDS image: eight 2-byte reads.
    using E = typename ClsE<T>::type;
    using v8 = typename E::v8;
    __syncthreads();                                              // every head's probabilities are in the table
    const int n = lane & 15, kq = lane >> 4;
    const int nks = S > 32 ? 2 : 1;
    v8 pa[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint32_t w[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = ks * 32 + 8 * kq + 2 * i;
        const float p0 = n < 12 ? sc[n * 64 + j] : 0.f, p1 = n < 12 ? sc[n * 64 + j + 1] : 0.f;   // 0 for keys >= S
        w[i] = E::pack(p0, p1);
      }
      pa[ks] = __builtin_bit_cast(v8, make_uint4(w[0], w[1], w[2], w[3]));
    }
    const uint16_t* xu = reinterpret_cast<const uint16_t*>(xs);
#pragma unroll 2
    for (int t = 0; t < 12; ++t) {
      const int c = (wv * 12 + t) * 16 + n;
      psg_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int ks = 0; ks < nks; ++ks) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int r0 = ks * 32 + 8 * kq + 2 * i, r1 = r0 + 1;
          r0 = r0 < S ? r0 : S - 1;                               // keys past S: probability 0, any finite row
          r1 = r1 < S ? r1 : S - 1;
          w[i] = (uint32_t)xu[r0 * RSE + c] | ((uint32_t)xu[r1 * RSE + c] << 16);
        }
        acc = E::mfma16(pa[ks], __builtin_bit_cast(v8, make_uint4(w[0], w[1], w[2], w[3])), acc);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int h = 4 * kq + r;
        if (h < 12) xbar[((int64_t)h * B + p) * H + c] = acc[r];
      }
    }
  } else {
    __builtin_amdgcn_wave_barrier();
    // fp32 verification mode: wave w owns heads 3w..3w+2, a lane owns features {4 lane + 256 k}
    float o[HPW][NK][4];
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
      for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[hh][k][e] = 0.f;
#pragma unroll 1
    for (int j = 0; j < S; ++j) {
      float xv[NK][4];
#pragma unroll
      for (int k = 0; k < NK; ++k) Act<T>::ld4(xs, j * RSE + lane * 4 + 256 * k, xv[k]);
#pragma unroll
      for (int hh = 0; hh < HPW; ++hh) {
        const float pj = sc[(wv * HPW + hh) * 64 + j];
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
          for (int e = 0; e < 4; ++e) o[hh][k][e] = fmaf(pj, xv[k][e], o[hh][k][e]);
      }
    }
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
      for (int k = 0; k < NK; ++k)
        *reinterpret_cast<float4*>(xbar + ((int64_t)(wv * HPW + hh) * B + p) * H + lane * 4 + 256 * k) =
            make_float4(o[hh][k][0], o[hh][k][1], o[hh][k][2], o[hh][k][3]);
  }
}

// g [heads][B][hidden] = W_k,h^T q_h (fp32); x_query [B*nq][hidden]: the layer's input query rows; x_text: its text rows,
// block text_index[p] (or p when text_index is NULL) of T rows per pair - pairs with the same prompt may share one block
// (and one row of text_mask); xbar [heads][B][hidden] (fp32) = sum_j softmax_j(g_h . x_j / 8 + mask_j) x_j.
// PSG_ERR_UNSUPPORTED outside hidden 768 / 12 heads or when the rows of one pair do not fit the LDS (fp32 with more
// than 48 rows): use the K | V form then.
extern "C" int psg_qformer_cls_attn_input(psg_ctx* ctx, const void* x_query, const void* x_text, const int32_t* text_index,
                                          const void* g, const uint8_t* text_mask, int B, int T_, int nq, int heads,
                                          int hidden, void* xbar, int dtype, void* stream) {
  PSG_REQUIRE(ctx && x_query && g && xbar && ((text_mask && x_text) || T_ == 0), PSG_ERR_INVALID,
              "psg_qformer_cls_attn_input: NULL argument");
  PSG_REQUIRE(B > 0 && T_ >= 0 && nq > 0 && nq + T_ <= 64, PSG_ERR_INVALID, "psg_qformer_cls_attn_input: B=%d T=%d nq=%d", B,
              T_, nq);
  PSG_REQUIRE(hidden == 768 && heads == 12, PSG_ERR_UNSUPPORTED,
              "psg_qformer_cls_attn_input: built for hidden 768 = 12 heads x 64, got %d / %d", hidden, heads);
  const size_t esz = dtype == PSG_F32 ? 4 : 2;
  const size_t lds = (size_t)(nq + T_) * (hidden + (esz == 2 ? 8 : 0)) * esz + (size_t)heads * 64 * sizeof(float);
  PSG_REQUIRE(lds <= 160 * 1024, PSG_ERR_UNSUPPORTED, "psg_qformer_cls_attn_input: %zu B of LDS for %d rows", lds, nq + T_);
#define CLS_IN(TT)                                                                                                     \
  do {                                                                                                                 \
    (void)hipFuncSetAttribute((const void*)qformer_cls_attn_input_kernel<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              160 * 1024);                                                                             \
    qformer_cls_attn_input_kernel<TT><<<(unsigned)B, 256, lds, (hipStream_t)stream>>>(                                  \
        (const TT*)x_query, (const TT*)x_text, text_index, (const float*)g, text_mask, B, T_, nq, (float*)xbar);       \
  } while (0)
  PSG_DISPATCH_DTYPE(dtype, "psg_qformer_cls_attn_input", CLS_IN(T));
#undef CLS_IN
  PSG_CHECK_LAUNCH("psg_qformer_cls_attn_input");
  return PSG_OK;
}

// First-layer variant: the nq query rows entering layer 0 are identical for every pair, so their fused Q/K/V
// projection is one [nq][3*hidden] block shared by all pairs; qkv_text holds the text rows [B*T][3*hidden].
extern "C" int psg_qformer_self_attn_shared(psg_ctx* ctx, const void* qkv_query, const void* qkv_text,
                                            const uint8_t* text_mask, int B, int T_, int nq, int heads, void* out,
                                            int dtype, void* stream) {
  PSG_REQUIRE(ctx && qkv_query && out && ((qkv_text && text_mask) || T_ == 0), PSG_ERR_INVALID,
              "psg_qformer_self_attn_shared: NULL argument");
  PSG_REQUIRE(B > 0 && T_ >= 0 && nq > 0 && heads > 0 && nq + T_ <= 64, PSG_ERR_INVALID,
              "psg_qformer_self_attn_shared: B=%d T=%d nq=%d", B, T_, nq);
  if (dtype == PSG_F32)
    return psg_self_attn_f32_launch(qkv_text ? qkv_text : qkv_query, qkv_query, text_mask, B, T_, nq, heads, 0, out,
                                    (hipStream_t)stream);
  PSG_REQUIRE(dtype == PSG_BF16 || dtype == PSG_F16, PSG_ERR_UNSUPPORTED, "psg_qformer_self_attn_shared: dtype %d", dtype);
  return psg_self_attn_mfma_launch(qkv_text ? qkv_text : qkv_query, qkv_query, text_mask, B, T_, nq, heads, 0, out,
                                   dtype, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// K6' scalar cross-attention: one 256-thread block per (pair, head); fp32 math regardless of the
// storage dtype.  Scores [nq][Lpad] live in LDS.  Used as the fp32 verification variant and as the
// on-device checker of the MFMA kernel; never the benchmarked path.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) cross_attn_simple_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                                const T* __restrict__ v,
                                                                const uint64_t* __restrict__ bits, int words,
                                                                const int32_t* __restrict__ pair_index, int N, int L,
                                                                int nq, int heads, int policy, T* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int p = blockIdx.x / heads, h = blockIdx.x % heads;
  const int hidden = heads * 64;
  const int Lp = (L + 3) & ~3;
  float* qs = smem;                 // [nq][64]
  float* sc = smem + nq * 64;       // [nq][Lp]
  const int tid = threadIdx.x;
  const int pidx = pair_index[p];
  const int oi = pidx / N, oj = pidx % N;
  for (int e = tid; e < nq * 64; e += 256)
    qs[e] = Act<T>::ld(q, ((int64_t)p * nq + e / 64) * hidden + h * 64 + (e % 64));
  __syncthreads();
  const float masked_val = policy == PSG_EMPTY_UNIFORM ? PSG_FMIN : -10000.0f;
  for (int l0 = 0; l0 < L; l0 += 256) {
    const int l = l0 + tid;
    if (l < L) {
      float kr[64];
#pragma unroll
      for (int d = 0; d < 64; d += 4) {
        float t[4];
        Act<T>::ld4(k, (int64_t)l * hidden + h * 64 + d, t);
        kr[d] = t[0]; kr[d + 1] = t[1]; kr[d + 2] = t[2]; kr[d + 3] = t[3];
      }
      const uint64_t w = bits[(int64_t)oi * words + (l >> 6)] | bits[(int64_t)oj * words + (l >> 6)];
      const bool on = (w >> (l & 63)) & 1ull;
      for (int i = 0; i < nq; ++i) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 64; ++d) s = fmaf(qs[i * 64 + d], kr[d], s);
        s *= 0.125f;
        // HF adds the mask: s + finfo.min == finfo.min in fp32; s - 10000 for the legacy variant
        sc[i * Lp + l] = on ? s : (policy == PSG_EMPTY_UNIFORM ? masked_val : s + masked_val);
      }
    }
  }
  __syncthreads();
  const int lane = tid & 63, wid = tid >> 6;
  for (int i = wid; i < nq; i += 4) {
    float m = -INFINITY;
    for (int l = lane; l < L; l += 64) m = fmaxf(m, sc[i * Lp + l]);
    m = wave_max(m);
    float sum = 0.f;
    for (int l = lane; l < L; l += 64) {
      const float e = expf(sc[i * Lp + l] - m);
      sc[i * Lp + l] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int l = lane; l < L; l += 64) {
      float pv = sc[i * Lp + l] * inv;
      pv = Act<T>::rnd(pv);
      sc[i * Lp + l] = pv;
    }
  }
  __syncthreads();
  for (int i = wid; i < nq; i += 4) {
    float o = 0.f;
    for (int l = 0; l < L; ++l) o = fmaf(sc[i * Lp + l], Act<T>::ld(v, (int64_t)l * hidden + h * 64 + lane), o);
    Act<T>::st(out, ((int64_t)p * nq + i) * hidden + h * 64 + lane, o);
  }
}

int psg_cross_attn_simple_launch(const void* q, const void* k, const void* v, const uint64_t* bits, int words,
                                 const int32_t* pair_index, int N, int P, int L, int nq, int heads, int policy,
                                 void* out, int dtype, hipStream_t st) {
  const int Lp = (L + 3) & ~3;
  const size_t lds = (size_t)(nq * 64 + nq * Lp) * sizeof(float);
  PSG_REQUIRE(lds <= 160 * 1024, PSG_ERR_UNSUPPORTED, "cross_attn(simple): L=%d needs %zu B of LDS", L, lds);
  PSG_DISPATCH_DTYPE(dtype, "psg_qformer_cross_attn(simple)", {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)cross_attn_simple_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds);
    cross_attn_simple_kernel<T><<<P * heads, 256, lds, st>>>((const T*)q, (const T*)k, (const T*)v, bits, words,
                                                            pair_index, N, L, nq, heads, policy, (T*)out);
  });
  PSG_CHECK_LAUNCH("psg_qformer_cross_attn(simple)");
  return PSG_OK;
}

// ---------------------------------------------------------------------------------------------
// K14.  One wave per (token row, head); head_dim = 128.  Keys are the cache slots [0, pos] of the
// row's pair (compacted layout: no pad slots, so the causal bound is the only mask).  Lane j owns
// key j of each 64-key chunk (reads its 128-element row with 16-byte loads); q is broadcast from
// LDS; softmax is online across chunks in fp32; P.V has lane d own dims d and d+64.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) llm_attn_kernel(const T* __restrict__ q, const T* __restrict__ kc,
                                                       const T* __restrict__ vc, const int32_t* __restrict__ tok_pair,
                                                       const int32_t* __restrict__ tok_pos, int64_t rows, int heads,
                                                       int ctx, T* __restrict__ out) {
  __shared__ float s_q[4][128];
  __shared__ float s_p[4][64];
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (wave >= rows * heads) return;          // whole wave exits together (wave-uniform)
  const int64_t row = wave / heads;
  const int h = (int)(wave % heads);
  const int pos = tok_pos[row];
  const int hidden = heads * 128;
  if (pos < 0) {                             // padding row: defined output, never consumed
    Act<T>::st(out, row * hidden + h * 128 + lane, 0.f);
    Act<T>::st(out, row * hidden + h * 128 + lane + 64, 0.f);
    return;
  }
  s_q[wid][lane] = Act<T>::ld(q, row * hidden + h * 128 + lane);
  s_q[wid][lane + 64] = Act<T>::ld(q, row * hidden + h * 128 + lane + 64);
  __builtin_amdgcn_wave_barrier();
  const int64_t cbase = ((int64_t)tok_pair[row] * heads + h) * ctx * 128;
  const float scale = 0.08838834764831845f;  // 1/sqrt(128)
  float m_run = -INFINITY, l_run = 0.f, o1 = 0.f, o2 = 0.f;
  for (int base = 0; base <= pos; base += 64) {
    const int j = base + lane;
    float s = -INFINITY;
    if (j <= pos) {
      const T* kp = kc + cbase + (int64_t)j * 128;
      float acc = 0.f;
#pragma unroll 8
      for (int d = 0; d < 128; d += 4) {
        float t[4];
        Act<T>::ld4(kp, d, t);
        acc = fmaf(s_q[wid][d], t[0], acc);
        acc = fmaf(s_q[wid][d + 1], t[1], acc);
        acc = fmaf(s_q[wid][d + 2], t[2], acc);
        acc = fmaf(s_q[wid][d + 3], t[3], acc);
      }
      s = acc * scale;
    }
    const float m_new = fmaxf(m_run, wave_max(s));
    const float alpha = expf(m_run - m_new);
    const float pj = expf(s - m_new);
    l_run = l_run * alpha + wave_sum(pj);
    o1 *= alpha;
    o2 *= alpha;
    s_p[wid][lane] = pj;
    __builtin_amdgcn_wave_barrier();
    const int nk = min(64, pos + 1 - base);
    for (int jj = 0; jj < nk; ++jj) {
      const float pv = s_p[wid][jj];
      const T* vp = vc + cbase + (int64_t)(base + jj) * 128;
      o1 = fmaf(pv, Act<T>::ld(vp, lane), o1);
      o2 = fmaf(pv, Act<T>::ld(vp, lane + 64), o2);
    }
    __builtin_amdgcn_wave_barrier();
    m_run = m_new;
  }
  const float inv = 1.0f / l_run;
  Act<T>::st(out, row * hidden + h * 128 + lane, o1 * inv);
  Act<T>::st(out, row * hidden + h * 128 + lane + 64, o2 * inv);
}

extern "C" int psg_llm_attn(psg_ctx* ctx_, const void* q, const void* k_cache, const void* v_cache,
                            const int32_t* tok_pair, const int32_t* tok_pos, int64_t rows, int heads, int head_dim,
                            int ctx, void* out, int dtype, void* stream) {
  PSG_REQUIRE(ctx_ && q && k_cache && v_cache && tok_pair && tok_pos && out, PSG_ERR_INVALID,
              "psg_llm_attn: NULL argument");
  PSG_REQUIRE(head_dim == 128, PSG_ERR_UNSUPPORTED, "psg_llm_attn: head_dim=%d (kernel is built for 128)", head_dim);
  if (rows == 0) return PSG_OK;
  int64_t waves = rows * heads;
  PSG_DISPATCH_DTYPE(dtype, "psg_llm_attn",
                     (llm_attn_kernel<T><<<(unsigned)((waves + 3) / 4), 256, 0, (hipStream_t)stream>>>(
                         (const T*)q, (const T*)k_cache, (const T*)v_cache, tok_pair, tok_pos, rows, heads, ctx,
                         (T*)out)));
  PSG_CHECK_LAUNCH("psg_llm_attn");
  return PSG_OK;
}

// ---------------------------------------------------------------------------------------------
// K13 + K14 fused for the decode step: rotary (HF-LL:130-160) + KV-cache append + attention over
// the cache (HF-LL:191-214) in ONE launch.  One wave per (pair, head); the new token's q/k/v come
// from the qkv projection (activation tensor or split-K partials), the new key/value are written
// to slot `pos` and used from registers (no read-after-write through memory); the cached keys
// [0, pos) are handled lane-per-key, P.V runs 4 keys per iteration with independent loads.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) decode_attn_kernel(const void* __restrict__ qkv, int qs,
                                                          const int32_t* __restrict__ tok_pair,
                                                          const int32_t* __restrict__ tok_pos,
                                                          const float* __restrict__ cos_tab,
                                                          const float* __restrict__ sin_tab, int rows, int heads,
                                                          int ctx, T* __restrict__ kc, T* __restrict__ vc,
                                                          T* __restrict__ out) {
  __shared__ float s_q[4][128];
  __shared__ float s_p[4][64];
  const int wave = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (wave >= rows * heads) return;
  const int row = wave / heads, h = wave % heads;
  const int pos = tok_pos[row];
  const int hidden = heads * 128;
  if (pos < 0) return;
  const int64_t base = (int64_t)row * 3 * hidden + h * 128;
  const int64_t sl = (int64_t)rows * 3 * hidden;
  const float q1 = ld1_in<T>(qkv, qs, sl, base + lane), q2 = ld1_in<T>(qkv, qs, sl, base + lane + 64);
  const float k1 = ld1_in<T>(qkv, qs, sl, base + hidden + lane), k2 = ld1_in<T>(qkv, qs, sl, base + hidden + lane + 64);
  const float v1 = ld1_in<T>(qkv, qs, sl, base + 2 * hidden + lane);
  const float v2 = ld1_in<T>(qkv, qs, sl, base + 2 * hidden + lane + 64);
  const float cs = cos_tab[pos * 64 + lane], sn = sin_tab[pos * 64 + lane];   // cos/sin(pos * inv_freq), HF-LL:115-128
  auto rnd = [](float f) { return Act<T>::rnd(f); };
  const float qa = rnd(q1 * cs - q2 * sn), qb = rnd(q2 * cs + q1 * sn);      // stored dtype, as the unfused path
  const float ka = rnd(k1 * cs - k2 * sn), kb = rnd(k2 * cs + k1 * sn);
  const int64_t cbase = ((int64_t)tok_pair[row] * heads + h) * ctx * 128;
  Act<T>::st(kc, cbase + (int64_t)pos * 128 + lane, ka);
  Act<T>::st(kc, cbase + (int64_t)pos * 128 + lane + 64, kb);
  Act<T>::st(vc, cbase + (int64_t)pos * 128 + lane, v1);
  Act<T>::st(vc, cbase + (int64_t)pos * 128 + lane + 64, v2);
  s_q[wid][lane] = qa;
  s_q[wid][lane + 64] = qb;
  __builtin_amdgcn_wave_barrier();
  const float scale = 0.08838834764831845f;  // 1/sqrt(128)
  const float s_new = wave_sum(qa * ka + qb * kb) * scale;     // the new key, from registers
  float m_run = s_new, l_run = 1.0f;                           // exp(s_new - m) = 1
  float o1 = rnd(v1), o2 = rnd(v2);
  for (int b0 = 0; b0 < pos; b0 += 64) {
    const int j = b0 + lane;
    float s = -INFINITY;
    if (j < pos) {
      // the lane's whole key row (128 elements) is requested before the first use: one round trip
      const T* kp = kc + cbase + (int64_t)j * 128;
      float t[32][4];
#pragma unroll
      for (int d = 0; d < 32; ++d) Act<T>::ld4(kp, d * 4, t[d]);
      float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
      for (int d = 0; d < 32; d += 2) {
        acc0 = fmaf(s_q[wid][4 * d], t[d][0], acc0);
        acc0 = fmaf(s_q[wid][4 * d + 1], t[d][1], acc0);
        acc0 = fmaf(s_q[wid][4 * d + 2], t[d][2], acc0);
        acc0 = fmaf(s_q[wid][4 * d + 3], t[d][3], acc0);
        acc1 = fmaf(s_q[wid][4 * d + 4], t[d + 1][0], acc1);
        acc1 = fmaf(s_q[wid][4 * d + 5], t[d + 1][1], acc1);
        acc1 = fmaf(s_q[wid][4 * d + 6], t[d + 1][2], acc1);
        acc1 = fmaf(s_q[wid][4 * d + 7], t[d + 1][3], acc1);
      }
      s = (acc0 + acc1) * scale;
    }
    const float m_new = fmaxf(m_run, wave_max(s));
    const float alpha = expf(m_run - m_new);
    const float pj = expf(s - m_new);
    l_run = l_run * alpha + wave_sum(pj);
    o1 *= alpha;
    o2 *= alpha;
    s_p[wid][lane] = pj;
    __builtin_amdgcn_wave_barrier();
    const int nk = min(64, pos - b0);
    const T* vp = vc + cbase + (int64_t)b0 * 128;
    int jj = 0;
    for (; jj + 32 <= nk; jj += 32) {                  // 64 loads in flight per lane
      float a[32], c[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        a[u] = Act<T>::ld(vp, (int64_t)(jj + u) * 128 + lane);
        c[u] = Act<T>::ld(vp, (int64_t)(jj + u) * 128 + lane + 64);
      }
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        const float pv = s_p[wid][jj + u];
        o1 = fmaf(pv, a[u], o1);
        o2 = fmaf(pv, c[u], o2);
      }
    }
    for (; jj + 8 <= nk; jj += 8) {
      float a[8], c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a[u] = Act<T>::ld(vp, (int64_t)(jj + u) * 128 + lane);
        c[u] = Act<T>::ld(vp, (int64_t)(jj + u) * 128 + lane + 64);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float pv = s_p[wid][jj + u];
        o1 = fmaf(pv, a[u], o1);
        o2 = fmaf(pv, c[u], o2);
      }
    }
    for (; jj < nk; ++jj) {
      const float pv = s_p[wid][jj];
      o1 = fmaf(pv, Act<T>::ld(vp, (int64_t)jj * 128 + lane), o1);
      o2 = fmaf(pv, Act<T>::ld(vp, (int64_t)jj * 128 + lane + 64), o2);
    }
    __builtin_amdgcn_wave_barrier();
    m_run = m_new;
  }
  const float inv = 1.0f / l_run;
  Act<T>::st(out, (int64_t)row * hidden + h * 128 + lane, o1 * inv);
  Act<T>::st(out, (int64_t)row * hidden + h * 128 + lane + 64, o2 * inv);
}

// Latency-oriented variant used for the decode step: one WORKGROUP (4 waves) per (pair, head).
// The single-wave kernel above spends its time in dependent round trips (keys, then values); here
// wave w owns keys {64 b + 16 w .. + 15}, four lanes share a key (32 dims each, quad shuffle
// reduce), so the key pass and the value pass of a <= 64-token context are one round trip each;
// the four partial (m, l, o) states and the new token's own term are merged through LDS.
template <typename T>
__global__ void __launch_bounds__(256) decode_attn4_kernel(const void* __restrict__ qkv, int qs,
                                                           const int32_t* __restrict__ tok_pair,
                                                           const int32_t* __restrict__ tok_pos,
                                                           const float* __restrict__ cos_tab,
                                                           const float* __restrict__ sin_tab, int rows, int heads,
                                                           int ctx, T* __restrict__ kc, T* __restrict__ vc,
                                                           T* __restrict__ out, int wt = 0) {
  __shared__ PsgDecodeAttnScratch sc;
  const int unit = blockIdx.x;
  const int row = unit / heads, h = unit % heads;
  const int pos = tok_pos[row];
  const int hidden = heads * 128;
  if (pos < 0) return;                                        // whole workgroup (uniform)
  const int64_t sl = (int64_t)rows * 3 * hidden;
  auto ld = [&](const int64_t (&idx)[6], float (&x)[6]) {
    if (qs > 0) {                                             // split-K partials: all slices of q, k, v in one pass
      ldn_splits<float, 6>(qkv, qs, sl, idx, x);
#pragma unroll
      for (int e = 0; e < 6; ++e) x[e] = Act<T>::rnd(x[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 6; ++e) x[e] = Act<T>::ld(reinterpret_cast<const T*>(qkv), idx[e]);
    }
  };
  auto st = [&](int64_t i, float v) {
    if constexpr (std::is_same<T, float>::value) {
      if (wt) {                                                 // option wt_stores: 4-byte agent-scope store = write-through
        __hip_atomic_store(reinterpret_cast<unsigned*>(out) + i, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
    }
    Act<T>::st(out, i, v);
  };
  // the arithmetic lives in psg_decode_math.h: the persistent decoder layer (psg_decode_layer.hip) runs the same unit
  psg_decode_attn4_unit<T>(true, (int)threadIdx.x, row, h, pos, tok_pair[row], heads, ctx, cos_tab, sin_tab, kc, vc, ld,
                           st, &sc, wt != 0);
}

extern "C" int psg_decode_attn(psg_ctx* ctx_, const void* qkv, int qkv_splits, const int32_t* tok_pair,
                               const int32_t* tok_pos, const float* rope_cos, const float* rope_sin, int rows,
                               int heads, int head_dim, int ctx, void* k_cache, void* v_cache, void* out, int dtype,
                               void* stream) {
  PSG_REQUIRE(ctx_ && qkv && tok_pair && tok_pos && rope_cos && rope_sin && k_cache && v_cache && out, PSG_ERR_INVALID,
              "psg_decode_attn: NULL argument");
  PSG_REQUIRE(head_dim == 128, PSG_ERR_UNSUPPORTED, "psg_decode_attn: head_dim=%d (kernel is built for 128)", head_dim);
  PSG_REQUIRE(qkv_splits >= 0 && qkv_splits <= PSG_MAX_SPLITS, PSG_ERR_INVALID, "psg_decode_attn: qkv_splits=%d",
              qkv_splits);
  if (rows == 0) return PSG_OK;
  const int waves = rows * heads;
  const int single = ctx_->opt.decode_attn_1wave;    // option decode_attn_1wave: the one-wave-per-head kernel
  if (!single) {
    PSG_DISPATCH_DTYPE(dtype, "psg_decode_attn",
                       (decode_attn4_kernel<T><<<waves, 256, 0, (hipStream_t)stream>>>(
                           qkv, qkv_splits, tok_pair, tok_pos, rope_cos, rope_sin, rows, heads, ctx, (T*)k_cache,
                           (T*)v_cache, (T*)out, ctx_->opt.wt_stores & 1)));
    PSG_CHECK_LAUNCH("psg_decode_attn");
    return PSG_OK;
  }
  PSG_DISPATCH_DTYPE(dtype, "psg_decode_attn",
                     (decode_attn_kernel<T><<<(waves + 3) / 4, 256, 0, (hipStream_t)stream>>>(
                         qkv, qkv_splits, tok_pair, tok_pos, rope_cos, rope_sin, rows, heads, ctx, (T*)k_cache,
                         (T*)v_cache, (T*)out)));
  PSG_CHECK_LAUNCH("psg_decode_attn");
  return PSG_OK;
}
