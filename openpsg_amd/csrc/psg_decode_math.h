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
  float o[4][2][128];        // [wave][half-wave][dim]: the two half-waves of a wave take alternate keys
  float ml[4][2];
  float snew[4];
};

// value of lane (lane ^ X) within the lane's 32-lane half (ds_swizzle, bit-mask mode: and 0x1f, or 0, xor X)
template <int X>
__device__ __forceinline__ float psg_swz_xor(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (X << 10) | 0x1f));
}

// K13 + K14 for ONE unit (row, head) by four waves (256 threads, tid = 0..255): rotary (HF-LL:130-160), KV-cache append,
// attention over the cache (HF-LL:191-214).  Wave w owns keys {64 b + 16 w .. + 15}: 8 KB (fp32) of K and of V, each
// CONTIGUOUS in the cache [pair][head][ctx][128].  Round 6 layout: a load instruction covers two whole rows - half-wave hf,
// chunk c = lane & 31 reads the 4 dims 4 c .. 4 c + 3 of key 2 d + hf - so every instruction is one coalesced 1 KB (fp32)
// access of fully used 128-byte lines (round 5: 4 lanes per key at a 128-byte stride, 64 different lines per instruction,
// 16 bytes of each: the L1 re-fetched every line up to 8 times - 16.7 us for 37 MB at fp32).  A key's score is 4 FMAs per
// lane and a 32-lane transposing tree (9 ds_swizzle for the 8 keys of a half-wave: the value count halves while the lane
// distance does), which leaves it on 4 lanes; V rows accumulate 4 dims per lane, the half-waves' partial outputs and
// the four waves' (m, l, o) states and the new token's own term are merged through LDS.
//   live      false: nothing to do (the four waves still meet the two workgroup barriers)
//   ld(idx,x) the new token's q1 q2 k1 k2 v1 v2 (element indices into [rows][3 hidden]) summed over the split-K slices
//   st(i, v)  stores the output element i of [rows][hidden]
template <typename T, class LoadQKV, class StoreOut>
__device__ __forceinline__ void psg_decode_attn4_unit(bool live, int tid, int row, int h, int pos, int pair, int heads,
                                                      int ctx, const float* __restrict__ cos_tab,
                                                      const float* __restrict__ sin_tab, T* __restrict__ kc,
                                                      T* __restrict__ vc, const LoadQKV& ld, const StoreOut& st,
                                                      PsgDecodeAttnScratch* sc, bool wt = false) {
#pragma clang fp contract(off)
  const int lane = tid & 63, wid = tid >> 6;
  const int hidden = heads * 128;
  const int64_t cbase = ((int64_t)pair * heads + h) * ctx * 128;
  const float scale = 0.08838834764831845f;                   // 1/sqrt(128)
  auto rnd = [](float f) { return Act<T>::rnd(f); };
  const int hf = lane >> 5, c = lane & 31;
  // after the transposing tree lane c of half hf holds the score of key 2 d + hf, d = bits 4, 3, 2 of c (on 4 lanes)
  const int kloc = 2 * (4 * ((c >> 4) & 1) + 2 * ((c >> 3) & 1) + ((c >> 2) & 1)) + hf;
  // Keys and values of the first 64 cached positions are requested BEFORE the new token's projections are summed:
  // their addresses depend on `pos` only, so the cache read, the split-K partials and the rotary tables share one
  // round trip instead of three dependent ones (projections -> barrier -> keys -> values).
  typename Act<T>::raw4 kr[8], vr[8];
  auto load_kv = [&](int b0) {
    const int kbase = b0 + 16 * wid;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const int j = kbase + 2 * d + hf;                       // rows >= pos: clamped to row 0 (written whenever pos > 0)
      const int64_t off = cbase + (int64_t)(j < pos ? j : 0) * 128 + c * 4;
      kr[d] = Act<T>::ldr4(kc, off);
      vr[d] = Act<T>::ldr4(vc, off);
    }
  };
  if (live) load_kv(0);                                       // pos == 0: row 0, never used
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
    // wt (fp32 caches, option wt_stores): the appended rows written through the L2 like the kernel's output
    auto app = [&](T* base, int64_t i, float v) {
      if constexpr (sizeof(T) == 4) {
        if (wt) {
          __hip_atomic_store(reinterpret_cast<unsigned*>(base) + i, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          return;
        }
      }
      Act<T>::st(base, i, v);
    };
    app(kc, cbase + (int64_t)pos * 128 + lane, ka);
    app(kc, cbase + (int64_t)pos * 128 + lane + 64, kb);
    app(vc, cbase + (int64_t)pos * 128 + lane, v1);
    app(vc, cbase + (int64_t)pos * 128 + lane + 64, v2);
    sc->q[lane] = qa;
    sc->q[lane + 64] = qb;
    const float dot = __builtin_fmaf(qa, ka, qb * kb);
    const float sn_ = wave_sum(dot) * scale;
    if (lane == 0) sc->snew[0] = sn_;
    vn1 = rnd(v1);
    vn2 = rnd(v2);
  }
  __syncthreads();
  float m_run = -INFINITY, l_run = 0.f;
  float o[4] = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    const float4 q4 = *reinterpret_cast<const float4*>(sc->q + c * 4);
    for (int b0 = 0; b0 < pos; b0 += 64) {
      if (b0 > 0) load_kv(b0);
      const int kbase = b0 + 16 * wid;
      float pt[8];
#pragma unroll
      for (int d = 0; d < 8; ++d) {                           // this lane's 4 dims of key 2 d + hf
        float kf[4];
        Act<T>::cv4(kr[d], kf);
        float acc = q4.x * kf[0];
        acc = __builtin_fmaf(q4.y, kf[1], acc);
        acc = __builtin_fmaf(q4.z, kf[2], acc);
        acc = __builtin_fmaf(q4.w, kf[3], acc);
        pt[d] = acc;
      }
      // 32-lane sums of 8 values: lanes 16 apart split the values (bit 4 of c: d < 4 | d >= 4), then 8 apart, then 4
      float p4[4], p2[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float keep = (c & 16) ? pt[i + 4] : pt[i], send = (c & 16) ? pt[i] : pt[i + 4];
        p4[i] = keep + psg_swz_xor<16>(send);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float keep = (c & 8) ? p4[i + 2] : p4[i], send = (c & 8) ? p4[i] : p4[i + 2];
        p2[i] = keep + psg_swz_xor<8>(send);
      }
      float acc;
      {
        const float keep = (c & 4) ? p2[1] : p2[0], send = (c & 4) ? p2[0] : p2[1];
        acc = keep + psg_swz_xor<4>(send);
      }
      acc = acc + psg_swz_xor<2>(acc);
      acc = acc + psg_swz_xor<1>(acc);
      const float s = (kbase + kloc < pos) ? acc * scale : -INFINITY;
      const float m_new = fmaxf(m_run, wave_max(s));
      if (m_new == -INFINITY) continue;                          // this wave has no key in this pass (uniform)
      const float alpha = expf(m_run - m_new);
      const float pj = expf(s - m_new);                          // replicated over the 4 lanes of a key
      l_run = __builtin_fmaf(wave_sum(pj), 0.25f, l_run * alpha);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] *= alpha;
      if ((c & 3) == 0) sc->p[wid][kloc] = pj;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const int kl2 = 2 * d + hf;
        const float pv = kbase + kl2 < pos ? sc->p[wid][kl2] : 0.f;
        float vf[4];
        Act<T>::cv4(vr[d], vf);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __builtin_fmaf(pv, vf[e], o[e]);
      }
      __builtin_amdgcn_wave_barrier();
      m_run = m_new;
    }
  }
  *reinterpret_cast<float4*>(&sc->o[wid][hf][c * 4]) = make_float4(o[0], o[1], o[2], o[3]);
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
      r1 = __builtin_fmaf(f, sc->o[w][0][lane] + sc->o[w][1][lane], r1);
      r2 = __builtin_fmaf(f, sc->o[w][0][lane + 64] + sc->o[w][1][lane + 64], r2);
    }
    const float inv = 1.0f / l;
    st((int64_t)row * hidden + h * 128 + lane, r1 * inv);
    st((int64_t)row * hidden + h * 128 + lane + 64, r2 * inv);
  }
}
