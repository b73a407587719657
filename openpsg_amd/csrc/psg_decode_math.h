// Arithmetic shared by the decode step's row kernels (psg_rowops.hip, psg_attn.hip) and the persistent decoder layer
// (psg_decode_layer.hip), which must reproduce them BIT FOR BIT: everything where the compiler would otherwise be free
// to fuse a multiply into an add is written out here (explicit fmaf / separate mul + add under fp contract(off)), in the
// forms hipcc chose for the fp32 row kernels at ROCm 7.2, so that both users compile to the same operations whatever
// their surrounding code looks like.
#pragma once
#include "psg_common.h"

// sum of squares of one thread's four residual values (rmsnorm_kernel): ((0 + v0^2) + v1^2) + ..., products rounded
__device__ __forceinline__ float psg_sumsq4(const float (&v)[4], float ss) {
#pragma clang fp contract(off)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float p = v[e] * v[e];
    ss = ss + p;
  }
  return ss;
}

// half-split rotary (HF-LL:130-160): (x1, x2) = dims (d, d + 64) -> x1 cos - x2 sin, x2 cos + x1 sin
__device__ __forceinline__ void psg_rope_pair(float x1, float x2, float cs, float sn, float& a, float& b) {
#pragma clang fp contract(off)
  const float t2 = sn * x2, t1 = sn * x1;
  a = __builtin_fmaf(cs, x1, -t2);
  b = __builtin_fmaf(cs, x2, t1);
}

struct PsgDecodeAttnScratch {
  float q[128];
  float p[4][16];
  float o[4][128];
  float ml[4][2];
  float snew[4];
};

// K13 + K14 for ONE unit (row, head) by four waves (256 threads, tid = 0..255): rotary (HF-LL:130-160), KV-cache append,
// attention over the cache (HF-LL:191-214).  Wave w owns keys {64 b + 16 w .. + 15}, four lanes share a key (32 dims
// each, quad reduce); the four partial (m, l, o) states and the new token's own term are merged through LDS.
//   live      false: nothing to do (the four waves still meet the two workgroup barriers)
//   ld(idx,x) the new token's q1 q2 k1 k2 v1 v2 (element indices into [rows][3 hidden]) summed over the split-K slices
//   st(i, v)  stores the output element i of [rows][hidden]
template <typename T, class LoadQKV, class StoreOut>
__device__ __forceinline__ void psg_decode_attn4_unit(bool live, int tid, int row, int h, int pos, int pair, int heads,
                                                      int ctx, const float* __restrict__ cos_tab,
                                                      const float* __restrict__ sin_tab, T* __restrict__ kc,
                                                      T* __restrict__ vc, const LoadQKV& ld, const StoreOut& st,
                                                      PsgDecodeAttnScratch* sc) {
#pragma clang fp contract(off)
  const int lane = tid & 63, wid = tid >> 6;
  const int hidden = heads * 128;
  const int64_t cbase = ((int64_t)pair * heads + h) * ctx * 128;
  const float scale = 0.08838834764831845f;                   // 1/sqrt(128)
  auto rnd = [](float f) { return Act<T>::rnd(f); };
  const int kl = lane >> 2, part = lane & 3;
  // Keys and values of the first 64 cached positions are requested BEFORE the new token's projections are summed:
  // their addresses depend on `pos` only, so the cache read, the split-K partials and the rotary tables share one
  // round trip instead of three dependent ones (projections -> barrier -> keys -> values).
  typename Act<T>::raw4 t[8];
  typename Act<T>::raw1 a[16], c[16];
  auto load_kv = [&](int b0) {
    const int j = b0 + 16 * wid + kl;
    const T* kp = kc + cbase + (int64_t)(j < pos ? j : 0) * 128 + part * 32;
#pragma unroll
    for (int d = 0; d < 8; ++d) t[d] = Act<T>::ldr4(kp, d * 4);
    const int kbase = b0 + 16 * wid;
    const int nk = min(16, pos - kbase);
    const T* vp = vc + cbase + (int64_t)(nk > 0 ? kbase : 0) * 128;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int uu = u < nk ? u : 0;
      a[u] = Act<T>::ldr(vp, (int64_t)uu * 128 + lane);
      c[u] = Act<T>::ldr(vp, (int64_t)uu * 128 + lane + 64);
    }
  };
  if (live) load_kv(0);                                       // pos == 0: clamped to row 0, never used
  float vn1 = 0.f, vn2 = 0.f;
  if (live && wid == 0) {                                     // new token: rotary, cache append, own score
    const int64_t base = (int64_t)row * 3 * hidden + h * 128;
    const float cs = cos_tab[pos * 64 + lane], sn = sin_tab[pos * 64 + lane];
    const int64_t idx[6] = {base + lane, base + lane + 64, base + hidden + lane, base + hidden + lane + 64,
                            base + 2 * hidden + lane, base + 2 * hidden + lane + 64};
    float x[6];
    ld(idx, x);
    const float q1 = x[0], q2 = x[1], k1 = x[2], k2 = x[3], v1 = x[4], v2 = x[5];
    float qa, qb, ka, kb;
    psg_rope_pair(q1, q2, cs, sn, qa, qb);
    psg_rope_pair(k1, k2, cs, sn, ka, kb);
    qa = rnd(qa); qb = rnd(qb); ka = rnd(ka); kb = rnd(kb);
    Act<T>::st(kc, cbase + (int64_t)pos * 128 + lane, ka);
    Act<T>::st(kc, cbase + (int64_t)pos * 128 + lane + 64, kb);
    Act<T>::st(vc, cbase + (int64_t)pos * 128 + lane, v1);
    Act<T>::st(vc, cbase + (int64_t)pos * 128 + lane + 64, v2);
    sc->q[lane] = qa;
    sc->q[lane + 64] = qb;
    const float dot = __builtin_fmaf(qa, ka, qb * kb);
    const float sn_ = wave_sum(dot) * scale;
    if (lane == 0) sc->snew[0] = sn_;
    vn1 = rnd(v1);
    vn2 = rnd(v2);
  }
  __syncthreads();
  float m_run = -INFINITY, l_run = 0.f, o1 = 0.f, o2 = 0.f;
  if (live) {
    for (int b0 = 0; b0 < pos; b0 += 64) {
      if (b0 > 0) load_kv(b0);
      const int j = b0 + 16 * wid + kl;
      float s = -INFINITY;
      {
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
          const float* qq = sc->q + part * 32 + d * 4;
          float kf[4];
          Act<T>::cv4(t[d], kf);
          acc = __builtin_fmaf(qq[0], kf[0], acc);
          acc = __builtin_fmaf(qq[1], kf[1], acc);
          acc = __builtin_fmaf(qq[2], kf[2], acc);
          acc = __builtin_fmaf(qq[3], kf[3], acc);
        }
        acc = quad_sum(acc);
        if (j < pos) s = acc * scale;
      }
      const float m_new = fmaxf(m_run, wave_max(s));
      if (m_new == -INFINITY) continue;                          // this wave has no key in this pass (uniform)
      const float alpha = expf(m_run - m_new);
      const float pj = expf(s - m_new);                          // replicated over the 4 lanes of a key
      l_run = __builtin_fmaf(wave_sum(pj), 0.25f, l_run * alpha);
      o1 *= alpha;
      o2 *= alpha;
      if (part == 0) sc->p[wid][kl] = pj;
      __builtin_amdgcn_wave_barrier();
      const int nk = min(16, pos - (b0 + 16 * wid));             // > 0 here
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float pv = u < nk ? sc->p[wid][u] : 0.f;
        o1 = __builtin_fmaf(pv, Act<T>::cv(a[u]), o1);
        o2 = __builtin_fmaf(pv, Act<T>::cv(c[u]), o2);
      }
      __builtin_amdgcn_wave_barrier();
      m_run = m_new;
    }
  }
  sc->o[wid][lane] = o1;
  sc->o[wid][lane + 64] = o2;
  if (lane == 0) {
    sc->ml[wid][0] = m_run;
    sc->ml[wid][1] = l_run;
  }
  __syncthreads();
  if (live && wid == 0) {
    const float sn_ = sc->snew[0];
    float m = sn_;
#pragma unroll
    for (int w = 0; w < 4; ++w) m = fmaxf(m, sc->ml[w][0]);
    float e_new = expf(sn_ - m);
    float l = e_new, r1 = e_new * vn1, r2 = e_new * vn2;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = expf(sc->ml[w][0] - m);                    // exp(-inf) = 0 for a wave without keys
      l = __builtin_fmaf(f, sc->ml[w][1], l);
      r1 = __builtin_fmaf(f, sc->o[w][lane], r1);
      r2 = __builtin_fmaf(f, sc->o[w][lane + 64], r2);
    }
    const float inv = 1.0f / l;
    st((int64_t)row * hidden + h * 128 + lane, r1 * inv);
    st((int64_t)row * hidden + h * 128 + lane + 64, r2 * inv);
  }
}
