// K5 on the matrix cores: Q-Former self-attention (HF-IB:471-515, eager 176-196) for bf16 activations.
//
// One wave per (pair, head): S = 33 + T <= 64 tokens, head_dim 64.  Everything is read straight
// from the fused QKV matrix (rows [0, B*33) = query rows, pair-major; rows [B*33, ...) = text rows):
//   * S^T = K . Q^T with v_mfma_f32_32x32x16_bf16: 2 key tiles x 2 query tiles x 4 k-steps; the K and Q
//     fragments are 16-byte loads of a row's 8 consecutive head dims - no LDS;
//   * a lane owns one query row per query tile (column lane&31 of D) and 16 keys per key tile, so
//     the padding mask (V4:158-159: additive finfo.min on masked prompt tokens) is a per-wave 64-bit
//     ballot tested in registers and the softmax needs one lane^32 exchange;
//   * O^T = V^T . P^T: the V^T fragment of a lane (one head dim, 8 keys) is gathered with 2-byte loads
//     (64 B coalesced per half-wave) - 64 small loads per (pair, head) instead of an LDS transpose.
// The scalar kernel in psg_attn.hip (1.4 ms per layer at N = 50) stays as the fp32 verification path.
#include "psg_common.h"

typedef float sa_f32x16 __attribute__((ext_vector_type(16)));
typedef float sa_f32x2 __attribute__((ext_vector_type(2)));



template <typename E>
__global__ void __launch_bounds__(256, 2)
self_attn_mfma_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ q_shared,
                      const uint8_t* __restrict__ text_mask, int B, int Tt, int nq, int heads, int q_only,
                      uint16_t* __restrict__ out) {
  const int unit = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (unit >= B * heads) return;
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int p = unit / heads, h = unit % heads;
  const int hidden = heads * 64, ld = 3 * hidden;
  const int S = nq + Tt;
  const int64_t qrow0 = (int64_t)p * nq, trow0 = (int64_t)B * nq + (int64_t)p * Tt;
  auto row_of = [&](int j) -> int64_t {                         // output row of token j
    j = j < S ? j : S - 1;                                      // clamp: rows beyond S are masked / dropped
    return j < nq ? qrow0 + j : trow0 + (j - nq);
  };
  // input row of token j.  q_shared != nullptr (first layer): the query rows entering the layer are the same
  // for every pair (learned query tokens through the embedding LayerNorm), so their Q/K/V projection
  // [nq][3*hidden] is computed once and read from L2, and `qkv` holds the text rows only ([B*Tt][3*hidden]).
  auto in_row = [&](int j) -> const uint16_t* {
    j = j < S ? j : S - 1;
    if (q_shared) return j < nq ? q_shared + (int64_t)j * ld : qkv + ((int64_t)p * Tt + (j - nq)) * ld;
    return qkv + (j < nq ? qrow0 + j : trow0 + (j - nq)) * ld;
  };
  // validity of key `lane` (query rows are always valid; prompt tokens follow the attention mask)
  bool kvalid = lane < S;
  if (kvalid && lane >= nq) kvalid = text_mask[(int64_t)p * Tt + (lane - nq)] != 0;
  const unsigned long long valid64 = __ballot(kvalid);
  const unsigned long long exist64 = S >= 64 ? ~0ull : ((1ull << S) - 1ull);

  // fragments: lane (idx = lane&31, hi) holds row (32 tile + idx), head dims 16 s + 8 hi .. +7
  typename E::v8 kf[2][4], qf[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const uint16_t* rp = in_row(32 * t + l31) + h * 64 + hi * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[t][s] = *reinterpret_cast<const typename E::v8*>(rp + s * 16);
      kf[t][s] = *reinterpret_cast<const typename E::v8*>(rp + hidden + s * 16);
    }
  }
  sa_f32x16 sc[2][2];                                           // [key tile][query tile]
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      sc[kt][qt] = (sa_f32x16){0};
#pragma unroll
      for (int s = 0; s < 4; ++s)
        sc[kt][qt] = E::mfma32(kf[kt][s], qf[qt][s], sc[kt][qt]);
    }
  // masks for key tile 1 (keys 32..63), pre-shifted by 4*hi; tile 0 = first 32 query rows: always valid
  const uint32_t inv1 = (~(uint32_t)(valid64 >> 32)) >> (4 * hi);         // 1 = masked
  const uint32_t nex1 = (~(uint32_t)(exist64 >> 32)) >> (4 * hi);         // 1 = not a key at all
  const float C = 0.125f * 1.4426950408889634f;                           // 1/sqrt(64) * log2(e)
  float inv_l[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int koff = (r & 3) + 8 * (r >> 2);
      float y0 = sc[0][qt][r] * C;
      const int mb = __builtin_amdgcn_sbfe((int)inv1, koff, 1);
      float y1 = fmaf(sc[1][qt][r], C, __uint_as_float((uint32_t)mb & 0xff7fffffu));   // + finfo.min if masked
      if ((nex1 >> koff) & 1u) y1 = -INFINITY;
      sc[0][qt][r] = y0;
      sc[1][qt][r] = y1;
      m = fmaxf(m, fmaxf(y0, y1));
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e0 = __builtin_amdgcn_exp2f(sc[0][qt][r] - m), e1 = __builtin_amdgcn_exp2f(sc[1][qt][r] - m);
      sc[0][qt][r] = e0;
      sc[1][qt][r] = e1;
      sum += e0 + e1;
    }
    sum += __shfl_xor(sum, 32, 64);
    inv_l[qt] = 1.0f / sum;
  }
  // O^T[d][q] += V^T[d][keys] . P^T[keys][q]; key slice (kt, g): slot (hi, m) <-> key 32 kt + 16 g + (m&3) + 8 (m>>2) + 4 hi
  sa_f32x16 o[2][2];                                            // [d tile][query tile]
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) o[dt][qt] = (sa_f32x16){0};
  const int voff = 2 * hidden + h * 64 + l31;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      union {
        uint32_t u[4];
        typename E::v8 v;
      } pf[2], vf[2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          pf[qt].u[e] = E::pack(sc[kt][qt][8 * g + 2 * e], sc[kt][qt][8 * g + 2 * e + 1]);
      uint16_t ve[2][8];
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int key = 32 * kt + 16 * g + (m & 3) + 8 * (m >> 2) + 4 * hi;
        const uint16_t* vp = in_row(key) + voff;
        ve[0][m] = vp[0];
        ve[1][m] = vp[32];
      }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 4; ++e) vf[dt].u[e] = (uint32_t)ve[dt][2 * e] | ((uint32_t)ve[dt][2 * e + 1] << 16);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
          o[dt][qt] = E::mfma32(vf[dt].v, pf[qt].v, o[dt][qt]);
    }
  // lane (q = lane&31, hi) holds O[q][32 dt + (r&3) + 8 (r>>2) + 4 hi] for its row of each query tile
  const int nrows = q_only ? nq : S;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qi = 32 * qt + l31;
    if (qi < nrows) {
      uint16_t* op = out + row_of(qi) * hidden + h * 64 + 4 * hi;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          uint2 wv;
          wv.x = E::pack(o[dt][qt][4 * rr] * inv_l[qt], o[dt][qt][4 * rr + 1] * inv_l[qt]);
          wv.y = E::pack(o[dt][qt][4 * rr + 2] * inv_l[qt], o[dt][qt][4 * rr + 3] * inv_l[qt]);
          *reinterpret_cast<uint2*>(op + 32 * dt + 8 * rr) = wv;
        }
    }
  }
}

int psg_self_attn_mfma_launch(const void* qkv, const void* q_shared, const uint8_t* text_mask, int B, int T_, int nq,
                              int heads, int query_rows_only, void* out, int dtype, hipStream_t st) {
  const int64_t units = (int64_t)B * heads;
  PSG_DISPATCH_E16(dtype, "psg_qformer_self_attn(mfma)",
                   (self_attn_mfma_kernel<E><<<(unsigned)((units + 3) / 4), 256, 0, st>>>(
                       (const uint16_t*)qkv, (const uint16_t*)q_shared, text_mask, B, T_, nq, heads, query_rows_only,
                       (uint16_t*)out)));
  PSG_CHECK_LAUNCH("psg_qformer_self_attn(mfma)");
  return PSG_OK;
}
