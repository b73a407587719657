// K14 (prefill) on the matrix cores: causal self-attention of the LLM prompt pass (HF-LL:191-214, eager
// attention: scores = q.k / sqrt(128), causal + padding mask, fp32 softmax, probabilities cast to the value
// dtype, p.v) for bf16 activations.
//
// The prefill batch is pair-major: pair p owns rows [p*S, (p+1)*S), S = 32 visual tokens + the longest
// prompt (<= 64 here); tok_pos[row] = position of the token in its compacted sequence (== row index inside
// the pair for real tokens) or -1 for the padding rows at the end.  One wave per (pair, head):
//   * S^T = K . Q^T with v_mfma_f32_32x32x16_bf16, 2 key tiles x 2 query tiles x 8 k-steps (head_dim 128);
//     Q comes from the rotated query matrix, K from the cache rows the rotary kernel has just written,
//     both as 16-byte loads of a row's 8 consecutive head dims - no LDS;
//   * a lane owns one query row per query tile and 16 keys per key tile: the causal / padding mask is
//     index arithmetic in registers, the softmax needs one lane^32 exchange;
//   * O^T = V^T . P^T over 4 tiles of 32 head dims; the V^T fragment of a lane (one head dim, 8 keys) is
//     gathered with 2-byte loads from the cache (64 B coalesced per half-wave).
// The scalar kernel (psg_llm_attn, one wave per (row, head): 51 us per layer at 920 rows) stays for fp32 and
// for prompts longer than 64 rows.
#include "psg_common.h"

typedef float pa_f32x16 __attribute__((ext_vector_type(16)));
typedef float pa_f32x2 __attribute__((ext_vector_type(2)));



template <typename E>
__global__ void __launch_bounds__(64)
prefill_attn_mfma_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ kc,
                         const uint16_t* __restrict__ vc, const int32_t* __restrict__ tok_pos, int pairs, int S,
                         int heads, int ctx, uint16_t* __restrict__ out) {
  const int unit = blockIdx.x;
  const int p = unit / heads, h = unit % heads;
  const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
  const int hidden = heads * 128;
  const int64_t row0 = (int64_t)p * S;
  const int64_t cbase = ((int64_t)p * heads + h) * ctx * 128;
  // position of token `lane` of this pair (-1: padding row, also for lane >= S)
  const int mypos = lane < S ? tok_pos[row0 + lane] : -1;
  const unsigned long long valid64 = __ballot(mypos >= 0);
  auto rclamp = [&](int j) { return j < S ? j : S - 1; };
  // cache rows are only defined for the real tokens (a prefix of the pair's rows): never read past them
  const int nreal = __popcll(valid64);
  auto kclamp = [&](int j) { return j < nreal ? j : (nreal > 0 ? nreal - 1 : 0); };

  pa_f32x16 sc[2][2];                                           // [key tile][query tile]
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) sc[kt][qt] = (pa_f32x16){0};
  // fragments: lane (idx = lane&31, hi) holds row (32 tile + idx), head dims 16 s + 8 hi .. +7
  const uint16_t* qp[2];
  const uint16_t* kp[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int r = rclamp(32 * t + l31);
    qp[t] = q + (row0 + r) * hidden + h * 128 + hi * 8;
    kp[t] = kc + cbase + (int64_t)kclamp(r) * 128 + hi * 8;   // key j of the pair lives in cache row j (positions are compact)
  }
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    typename E::v8 qf[2], kf[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      qf[t] = *reinterpret_cast<const typename E::v8*>(qp[t] + s * 16);
      kf[t] = *reinterpret_cast<const typename E::v8*>(kp[t] + s * 16);
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
        sc[kt][qt] = E::mfma32(kf[kt], qf[qt], sc[kt][qt]);
  }
  // mask: key j is visible to query row i iff j <= i and key j is a real token (HF-LL causal + padding mask)
  const float C = 0.08838834764831845f * 1.4426950408889634f;  // 1/sqrt(128) * log2(e)
  float inv_l[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qi = 32 * qt + l31;
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const bool vis = key <= qi && ((valid64 >> key) & 1ull);
        const float y = vis ? sc[kt][qt][r] * C : -INFINITY;
        sc[kt][qt][r] = y;
        m = fmaxf(m, y);
      }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    if (m == -INFINITY) m = 0.f;                                // padding query row: every exp below is 0
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(sc[kt][qt][r] - m);
        sc[kt][qt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32, 64);
    const bool qvalid = qi < 64 && ((valid64 >> qi) & 1ull);   // padding rows: zeros (defined, never consumed)
    inv_l[qt] = (qvalid && sum > 0.f) ? 1.0f / sum : 0.f;
  }
  // O^T[d][q] += V^T[d][keys] . P^T[keys][q]; key slice (kt, g): slot (hi, m) <-> key 32 kt + 16 g + (m&3) + 8 (m>>2) + 4 hi
  pa_f32x16 o[4][2];                                            // [d tile][query tile]
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) o[dt][qt] = (pa_f32x16){0};
  const uint16_t* vbase = vc + cbase + l31;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      union {
        uint32_t u[4];
        typename E::v8 v;
      } pf[2], vf[4];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          pf[qt].u[e] = E::pack(sc[kt][qt][8 * g + 2 * e], sc[kt][qt][8 * g + 2 * e + 1]);
      uint16_t ve[4][8];
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int key = kclamp(32 * kt + 16 * g + (m & 3) + 8 * (m >> 2) + 4 * hi);
        const uint16_t* vp = vbase + (int64_t)key * 128;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) ve[dt][m] = vp[32 * dt];
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int e = 0; e < 4; ++e) vf[dt].u[e] = (uint32_t)ve[dt][2 * e] | ((uint32_t)ve[dt][2 * e + 1] << 16);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
          o[dt][qt] = E::mfma32(vf[dt].v, pf[qt].v, o[dt][qt]);
    }
  // lane (q = lane&31, hi) holds O[q][32 dt + (r&3) + 8 (r>>2) + 4 hi] for its row of each query tile;
  // padding rows get zeros (defined output, never consumed: same as the scalar kernel)
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qi = 32 * qt + l31;
    if (qi < S) {
      uint16_t* op = out + (row0 + qi) * hidden + h * 128 + 4 * hi;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
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

// The same attention with the rotary embedding and the KV-cache write folded in (HF-LL:130-160 + 191-214):
// reads the fused projection output qkv [rows][3*hidden] directly, rotates Q and K in registers (a lane's
// fragment for k-step s holds dims 16 s + 8 hi .. +7; the partner dims + 64 are k-step s + 4 of the same lane),
// writes the rotated K and the V rows to the cache for the decode steps, and never materialises Q.
template <typename E>
__global__ void __launch_bounds__(64)
prefill_attn_rope_mfma_kernel(const uint16_t* __restrict__ qkv, const int32_t* __restrict__ tok_pos,
                              const float* __restrict__ cos_tab, const float* __restrict__ sin_tab, int pairs, int S,
                              int heads, int ctx, uint16_t* __restrict__ kc, uint16_t* __restrict__ vc,
                              uint16_t* __restrict__ out) {
  const int unit = blockIdx.x;
  const int p = unit / heads, h = unit % heads;
  const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
  const int hidden = heads * 128;
  const int64_t ld = 3 * (int64_t)hidden;
  const int64_t row0 = (int64_t)p * S;
  const int64_t cbase = ((int64_t)p * heads + h) * ctx * 128;
  const int mypos = lane < S ? tok_pos[row0 + lane] : -1;
  const unsigned long long valid64 = __ballot(mypos >= 0);
  auto rclamp = [&](int j) { return j < S ? j : S - 1; };

  // V rows of the real tokens -> cache (16-byte pieces, 16 per row).  All 16 requests of a lane are issued before the
  // first store: as a load -> store loop this was up to 16 dependent round trips at the head of a latency-bound kernel.
  // The stores are unconditional (a conditional one pulls its load down next to it): a piece of a padding row or past
  // the last row is redirected to row 0, whose own data it then rewrites - cache rows of padding tokens stay untouched.
  if (valid64 & 1ull) {                                        // compact sequences: a pair with any token has row 0
    uint4 vx[16];
    int rv[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int i = lane + 64 * it, r = i >> 4, c = i & 15;
      rv[it] = (i < S * 16 && ((valid64 >> (r & 63)) & 1ull)) ? r : 0;
      vx[it] = *reinterpret_cast<const uint4*>(qkv + (row0 + rv[it]) * ld + 2 * hidden + h * 128 + c * 8);
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int c = (lane + 64 * it) & 15;
      *reinterpret_cast<uint4*>(vc + cbase + (int64_t)rv[it] * 128 + c * 8) = vx[it];
    }
  }

  pa_f32x16 sc[2][2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) sc[kt][qt] = (pa_f32x16){0};
  int rr[2];
  bool rreal[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    rr[t] = rclamp(32 * t + l31);
    rreal[t] = (32 * t + l31 < S) && ((valid64 >> rr[t]) & 1ull);
  }
  // Every request of the wave is issued before the first dependent instruction: Q / K fragments of all four k-steps,
  // the rotary table rows, and (below) the V gathers.  One wave runs per SIMD here (pairs x heads = 640 waves on 1024
  // SIMDs), so the kernel's time is its chain of memory round trips; per-k-step loads made it five of them.
  union F8 {
    uint4 u;
    uint16_t h[8];
    typename E::v8 v;
  };
  F8 qa[4][2], qb[4][2], ka[4][2], kb[4][2];
  float4 csr[4][2][2], snr[4][2][2];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const uint16_t* rp = qkv + (row0 + rr[t]) * ld + h * 128 + 16 * s + 8 * hi;
      qa[s][t].u = *reinterpret_cast<const uint4*>(rp);
      qb[s][t].u = *reinterpret_cast<const uint4*>(rp + 64);
      ka[s][t].u = *reinterpret_cast<const uint4*>(rp + hidden);
      kb[s][t].u = *reinterpret_cast<const uint4*>(rp + hidden + 64);
      // position of a real token == its row index inside the pair (compact sequences)
      const float* cp = cos_tab + rr[t] * 64 + 16 * s + 8 * hi;
      const float* sp = sin_tab + rr[t] * 64 + 16 * s + 8 * hi;
      csr[s][t][0] = *reinterpret_cast<const float4*>(cp);
      csr[s][t][1] = *reinterpret_cast<const float4*>(cp + 4);
      snr[s][t][0] = *reinterpret_cast<const float4*>(sp);
      snr[s][t][1] = *reinterpret_cast<const float4*>(sp + 4);
    }
  // V^T fragments: lane (l31, hi) gathers head dim l31 + 32 dt of 8 keys per (key tile, half): 2-byte loads
  const uint16_t* vbase = qkv + row0 * ld + 2 * hidden + h * 128 + l31;
  uint16_t ve[2][2][4][8];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int key = rclamp(32 * kt + 16 * g + (m & 3) + 8 * (m >> 2) + 4 * hi);
        const uint16_t* vp = vbase + (int64_t)key * ld;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) ve[kt][g][dt][m] = vp[32 * dt];
      }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float cs[8] = {csr[s][t][0].x, csr[s][t][0].y, csr[s][t][0].z, csr[s][t][0].w,
                           csr[s][t][1].x, csr[s][t][1].y, csr[s][t][1].z, csr[s][t][1].w};
      const float sn[8] = {snr[s][t][0].x, snr[s][t][0].y, snr[s][t][0].z, snr[s][t][0].w,
                           snr[s][t][1].x, snr[s][t][1].y, snr[s][t][1].z, snr[s][t][1].w};
      F8 qa2, qb2, ka2, kb2;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float q1 = E::to_f32(qa[s][t].h[e]), q2 = E::to_f32(qb[s][t].h[e]);
        const float k1 = E::to_f32(ka[s][t].h[e]), k2 = E::to_f32(kb[s][t].h[e]);
        // q*cos + rotate_half(q)*sin, rotate_half(x) = cat(-x2, x1)
        qa2.h[e] = E::from_f32(q1 * cs[e] - q2 * sn[e]);
        qb2.h[e] = E::from_f32(q2 * cs[e] + q1 * sn[e]);
        ka2.h[e] = E::from_f32(k1 * cs[e] - k2 * sn[e]);
        kb2.h[e] = E::from_f32(k2 * cs[e] + k1 * sn[e]);
      }
      qa[s][t] = qa2;
      qb[s][t] = qb2;
      ka[s][t] = ka2;
      kb[s][t] = kb2;
      if (rreal[t]) {
        uint16_t* kp = kc + cbase + (int64_t)rr[t] * 128 + 16 * s + 8 * hi;
        *reinterpret_cast<uint4*>(kp) = ka2.u;
        *reinterpret_cast<uint4*>(kp + 64) = kb2.u;
      }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        sc[kt][qt] = E::mfma32(ka[s][kt].v, qa[s][qt].v, sc[kt][qt]);
        sc[kt][qt] = E::mfma32(kb[s][kt].v, qb[s][qt].v, sc[kt][qt]);
      }
  }
  const float C = 0.08838834764831845f * 1.4426950408889634f;  // 1/sqrt(128) * log2(e)
  float inv_l[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qi = 32 * qt + l31;
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const bool vis = key <= qi && ((valid64 >> key) & 1ull);
        const float y = vis ? sc[kt][qt][r] * C : -INFINITY;
        sc[kt][qt][r] = y;
        m = fmaxf(m, y);
      }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    if (m == -INFINITY) m = 0.f;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(sc[kt][qt][r] - m);
        sc[kt][qt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32, 64);
    const bool qvalid = qi < 64 && ((valid64 >> qi) & 1ull);
    inv_l[qt] = (qvalid && sum > 0.f) ? 1.0f / sum : 0.f;
  }
  pa_f32x16 o[4][2];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) o[dt][qt] = (pa_f32x16){0};
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      union {
        uint32_t u[4];
        typename E::v8 v;
      } pf[2], vf[4];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          pf[qt].u[e] = E::pack(sc[kt][qt][8 * g + 2 * e], sc[kt][qt][8 * g + 2 * e + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          vf[dt].u[e] = (uint32_t)ve[kt][g][dt][2 * e] | ((uint32_t)ve[kt][g][dt][2 * e + 1] << 16);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
          o[dt][qt] = E::mfma32(vf[dt].v, pf[qt].v, o[dt][qt]);
    }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qi = 32 * qt + l31;
    if (qi < S) {
      uint16_t* op = out + (row0 + qi) * hidden + h * 128 + 4 * hi;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          uint2 wv;
          wv.x = E::pack(o[dt][qt][4 * r4] * inv_l[qt], o[dt][qt][4 * r4 + 1] * inv_l[qt]);
          wv.y = E::pack(o[dt][qt][4 * r4 + 2] * inv_l[qt], o[dt][qt][4 * r4 + 3] * inv_l[qt]);
          *reinterpret_cast<uint2*>(op + 32 * dt + 8 * r4) = wv;
        }
    }
  }
}

extern "C" int psg_prefill_attn_rope(psg_ctx* ctx_, const void* qkv, const int32_t* tok_pos, const float* rope_cos,
                                     const float* rope_sin, int pairs, int rows_per_pair, int heads, int head_dim,
                                     int ctx, void* k_cache, void* v_cache, void* out, int dtype, void* stream) {
  PSG_REQUIRE(ctx_ && qkv && tok_pos && rope_cos && rope_sin && k_cache && v_cache && out, PSG_ERR_INVALID,
              "psg_prefill_attn_rope: NULL argument");
  PSG_REQUIRE(dtype == PSG_BF16 || dtype == PSG_F16, PSG_ERR_UNSUPPORTED, "psg_prefill_attn_rope: 16-bit activations only");
  PSG_REQUIRE(head_dim == 128, PSG_ERR_UNSUPPORTED, "psg_prefill_attn_rope: head_dim=%d (kernel is built for 128)",
              head_dim);
  PSG_REQUIRE(rows_per_pair >= 1 && rows_per_pair <= 64 && rows_per_pair <= ctx, PSG_ERR_UNSUPPORTED,
              "psg_prefill_attn_rope: rows_per_pair=%d (1..64, <= ctx=%d)", rows_per_pair, ctx);
  PSG_REQUIRE(pairs >= 0 && heads > 0, PSG_ERR_INVALID, "psg_prefill_attn_rope: pairs=%d heads=%d", pairs, heads);
  if (pairs == 0) return PSG_OK;
  PSG_DISPATCH_E16(dtype, "psg_prefill_attn_rope",
                   (prefill_attn_rope_mfma_kernel<E><<<(unsigned)(pairs * heads), 64, 0, (hipStream_t)stream>>>(
                       (const uint16_t*)qkv, tok_pos, rope_cos, rope_sin, pairs, rows_per_pair, heads, ctx,
                       (uint16_t*)k_cache, (uint16_t*)v_cache, (uint16_t*)out)));
  PSG_CHECK_LAUNCH("psg_prefill_attn_rope");
  return PSG_OK;
}

int psg_prefill_attn_f32_launch(const void* q, const void* kc, const void* vc, const int32_t* tok_pos, int pairs, int rpp,
                                int heads, int ctx, void* out, hipStream_t st);

extern "C" int psg_prefill_attn(psg_ctx* ctx_, const void* q, const void* k_cache, const void* v_cache,
                                const int32_t* tok_pos, int pairs, int rows_per_pair, int heads, int head_dim, int ctx,
                                void* out, int dtype, void* stream) {
  PSG_REQUIRE(ctx_ && q && k_cache && v_cache && tok_pos && out, PSG_ERR_INVALID, "psg_prefill_attn: NULL argument");
  PSG_REQUIRE(dtype == PSG_BF16 || dtype == PSG_F16 || dtype == PSG_F32, PSG_ERR_UNSUPPORTED,
              "psg_prefill_attn: dtype %d", dtype);
  PSG_REQUIRE(head_dim == 128, PSG_ERR_UNSUPPORTED, "psg_prefill_attn: head_dim=%d (kernel is built for 128)", head_dim);
  PSG_REQUIRE(rows_per_pair >= 1 && rows_per_pair <= 64 && rows_per_pair <= ctx, PSG_ERR_UNSUPPORTED,
              "psg_prefill_attn: rows_per_pair=%d (1..64, <= ctx=%d); longer prompts: psg_llm_attn", rows_per_pair, ctx);
  PSG_REQUIRE(pairs >= 0 && heads > 0, PSG_ERR_INVALID, "psg_prefill_attn: pairs=%d heads=%d", pairs, heads);
  if (pairs == 0) return PSG_OK;
  if (dtype == PSG_F32)                                      // exact f32 matrix instructions (psg_attn_f32.hip)
    return psg_prefill_attn_f32_launch(q, k_cache, v_cache, tok_pos, pairs, rows_per_pair, heads, ctx, out,
                                       (hipStream_t)stream);
  PSG_DISPATCH_E16(dtype, "psg_prefill_attn",
                   (prefill_attn_mfma_kernel<E><<<(unsigned)(pairs * heads), 64, 0, (hipStream_t)stream>>>(
                       (const uint16_t*)q, (const uint16_t*)k_cache, (const uint16_t*)v_cache, tok_pos, pairs,
                       rows_per_pair, heads, ctx, (uint16_t*)out)));
  PSG_CHECK_LAUNCH("psg_prefill_attn");
  return PSG_OK;
}
