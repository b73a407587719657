// K6: relation-query cross-attention on the CDNA4 matrix cores (primary kernel of the path).
//
// Replaces HF-IB:464-466, 487-496 as driven by V4:168-170, 179-185: the reference expands the SAME
// [L,256] patch tensor to all B = N^2 pairs and re-projects K/V per pair; here K/V [L,768] are
// projected once per image and every pair reads them from LDS.  The pair mask
// (mask_i | mask_j, V4:430-433) is never materialised: each lane ORs two rows of the per-object
// bitmask table and tests bits in registers.
//
// Work layout (gfx950, wave = 64):
//   * query rows of all pairs are one flat list of R = P*33 rows, cut into 32-row tiles
//     (33 = 1 cls row + 32 relation rows, so there is no padding waste beyond the last tile);
//   * a workgroup (4 waves) owns one head h: it stages K_h [Lpad][64] and V_h^T [64][Lpad] (bf16)
//     into LDS once (row strides padded by 16 B => conflict-free ds_read_b128) and then walks
//     row tiles persistently, one tile per wave per iteration;
//   * S^T = K . Q^T with v_mfma_f32_32x32x16_bf16 (A = K fragment from LDS, B = Q fragment held
//     in registers), so a lane owns ONE query row (column lane&31 of D) and 16 keys per 32-key
//     tile: the mask is per lane, the softmax reduction is in-register plus one lane^32 exchange;
//   * O^T = V^T . P^T the same way (A = V^T fragment from LDS, B = P fragment = this lane's
//     exponentiated scores packed to bf16), so the online-softmax rescale and the final 1/l are
//     lane-local.  V^T is stored with key bits 2<->3 swapped inside every 16-key group, which makes
//     the accumulator registers of the S^T tile line up with the B-operand slots of the P.V MFMA
//     without any cross-lane shuffle;
//   * keys are processed in chunks of 128 (4 S^T tiles, 64 accumulator registers) with an fp32
//     online softmax, so any L works with one instantiation.
//
// Mask semantics: masked key => score + finfo.min (== finfo.min in
// fp32) so an all-masked row is a UNIFORM softmax over the L real keys; keys in [L, Lpad) are
// padding and get -inf (weight exactly 0 in every case).
#include "psg_common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

#define XA_KSTRIDE 144  // bytes per K row in LDS: 64 bf16 + 16 B pad
#define XA_CT 4         // S^T tiles (of 32 keys) per online-softmax chunk
#define XA_GRAB 4       // row tiles a wave takes from the work counter per atomic

// value of the partner lane (lane ^ 32) via v_permlane32_swap (VALU, no LDS round trip)
__device__ __forceinline__ float xchg32_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xchg32_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  f32x2_t f = {lo, hi};
  bf16x2_t b = __builtin_convertvector(f, bf16x2_t);  // v_cvt_pk_bf16_f32 on gfx950
  return __builtin_bit_cast(uint32_t, b);
}

template <int NC>   // NC = key chunks of 128 (L <= 128 NC): unrolled so the prefetched mask words index statically
__global__ void __launch_bounds__(256, 2)
cross_attn_mfma_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
                       const uint64_t* __restrict__ bits, int words, const int32_t* __restrict__ pair_index, int N,
                       int64_t R, int L, int nq, int heads, int policy, int* __restrict__ counters,
                       uint16_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Lpad = (L + 31) & ~31;
  const int NT = Lpad >> 5;
  const int VS = Lpad * 2 + 16;  // bytes per V^T row in LDS
  unsigned char* k_lds = smem;
  unsigned char* vt_lds = smem + (size_t)Lpad * XA_KSTRIDE;
  const int h = blockIdx.x % heads;
  const int g = blockIdx.x / heads;
  const int G = gridDim.x / heads;
  const int hidden = heads * 64;
  const int tid = threadIdx.x;

  // ---- stage K_h and V_h^T (once per workgroup) ----
  for (int e = tid; e < Lpad * 8; e += 256) {
    const int key = e >> 3, c = e & 7;
    uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
    if (key < L) {
      kv = *reinterpret_cast<const uint4*>(k + (int64_t)key * hidden + h * 64 + c * 8);
      vv = *reinterpret_cast<const uint4*>(v + (int64_t)key * hidden + h * 64 + c * 8);
    }
    *reinterpret_cast<uint4*>(k_lds + key * XA_KSTRIDE + c * 16) = kv;
    const int o = key & 15;
    const int pos = (o & 3) | ((o & 8) >> 1) | ((o & 4) << 1);  // swap bits 2 <-> 3
    const int kcol = ((key & ~15) | pos) * 2;
    const uint16_t* ve = reinterpret_cast<const uint16_t*>(&vv);
#pragma unroll
    for (int d = 0; d < 8; ++d) *reinterpret_cast<uint16_t*>(vt_lds + (c * 8 + d) * VS + kcol) = ve[d];
  }
  __syncthreads();

  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  // Row tiles.  nq == 33: tile t < P = rows 1..32 of pair t (the 32 relation queries share ONE pair
  // mask, so most 32-key tiles are masked for the whole tile and are skipped below); tiles >= P batch
  // the cls rows (row 0) of 32 consecutive pairs.  Other nq: flat 32-row tiles.
  const bool aligned = nq == 33;
  const int64_t P = R / nq;
  const int64_t ntile = aligned ? P + ((P + 31) >> 5) : (R + 31) >> 5;
  const float C = 0.125f * 1.4426950408889634f;  // 1/sqrt(64) * log2(e)
  const uint32_t bias_bits = policy == PSG_EMPTY_UNIFORM ? 0xff7fffffu /* finfo.min */
                                                         : __float_as_uint(-10000.0f * 1.4426950408889634f);
  const unsigned char* kfrag_base = k_lds + l31 * XA_KSTRIDE + hi * 16;
  const unsigned char* vfrag_base = vt_lds + l31 * VS + hi * 16;

  // One unit of work = (row tile, head h).  Its operands (Q fragments, the rows' pair-mask words) sit
  // behind a chain of dependent global loads (pair_index -> bits -> words); with most key tiles skipped
  // a unit is short, so the NEXT unit's operands are fetched while the current one is computed.
  struct XUnit {
    bf16x8_t qf[4];
    uint64_t mw[2 * NC];
    int64_t row;
    bool rvalid;
  };
  auto fetch = [&](int64_t tile, XUnit& u) {
    if (aligned) {
      if (tile < P) {
        u.row = tile * 33 + 1 + l31;
        u.rvalid = true;
      } else {
        const int64_t pr = (tile - P) * 32 + l31;
        u.rvalid = pr < P;
        u.row = (u.rvalid ? pr : P - 1) * 33;
      }
    } else {
      u.row = tile * 32 + l31;
      u.rvalid = u.row < R;
      if (!u.rvalid) u.row = R - 1;
    }
    // Q fragments: B operand of S^T = K.Q^T; lane (q = lane&31, hi) holds Q[q][16 s + 8 hi .. +7]
    const uint16_t* qp = q + u.row * hidden + h * 64 + hi * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) u.qf[s] = *reinterpret_cast<const bf16x8_t*>(qp + s * 16);
    const int pidx = pair_index[u.row / nq];
    const uint64_t* bi = bits + (int64_t)(pidx / N) * words;
    const uint64_t* bj = bits + (int64_t)(pidx % N) * words;
#pragma unroll
    for (int w = 0; w < 2 * NC; ++w) u.mw[w] = w < words ? (bi[w] | bj[w]) : 0ull;
  };
  // Dynamic distribution: unit costs differ by 5x (a pair with an empty mask union needs every key
  // tile, most pairs need one or two), and a static split left the slowest wave at 2x the mean.
  // Waves of the workgroups that own head h pull batches of XA_GRAB tiles from counters[h]
  // (zeroed by the launcher on the same stream); one returning atomic per batch.
  int64_t qbase = 0;
  int qk = XA_GRAB;
  auto next_tile = [&]() -> int64_t {
    if (qk == XA_GRAB) {
      int b = 0;
      if (lane == 0) b = atomicAdd(counters + h, XA_GRAB);
      qbase = __builtin_amdgcn_readfirstlane(b);
      qk = 0;
    }
    return qbase + qk++;
  };
  XUnit cur, nxt;
  int64_t tile = next_tile();
  if (tile < ntile) fetch(tile, cur);
  while (tile < ntile) {
    const int64_t tile_next = next_tile();
    if (tile_next < ntile) fetch(tile_next, nxt);
    const int64_t row = cur.row;
    const bool rvalid = cur.rvalid;
    bf16x8_t qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = cur.qf[s];

    // a row whose pair mask is empty attends to every key (uniform softmax): it needs all tiles
    uint64_t anybits = 0;
#pragma unroll
    for (int w = 0; w < 2 * NC; ++w) anybits |= cur.mw[w];
    const bool any_empty_row = __any(anybits == 0ull);

    f32x16_t o0 = {0}, o1 = {0};
    float m_run = -INFINITY, l_run = 0.f;

#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      const int c0 = cc * XA_CT;
      if (c0 >= NT) break;
      // inverted mask words (1 = masked), pre-shifted by 4*hi so the bit index is a constant per register
      uint32_t inv[XA_CT];
      bool need[XA_CT];
      {
        const uint64_t m0 = cur.mw[2 * cc], m1 = cur.mw[2 * cc + 1];
        inv[0] = ~(uint32_t)m0 >> (4 * hi);
        inv[1] = ~(uint32_t)(m0 >> 32) >> (4 * hi);
        inv[2] = ~(uint32_t)m1 >> (4 * hi);
        inv[3] = ~(uint32_t)(m1 >> 32) >> (4 * hi);
        // a 32-key tile no row of this row tile attends to contributes exactly 0: skip it (wave-uniform)
        need[0] = any_empty_row || __any((uint32_t)m0 != 0u);
        need[1] = any_empty_row || __any((uint32_t)(m0 >> 32) != 0u);
        need[2] = any_empty_row || __any((uint32_t)m1 != 0u);
        need[3] = any_empty_row || __any((uint32_t)(m1 >> 32) != 0u);
      }
#pragma unroll
      for (int t = 0; t < XA_CT; ++t) need[t] = need[t] && (c0 + t < NT);
      if (!(need[0] || need[1] || need[2] || need[3])) continue;
      f32x16_t acc[XA_CT];
#pragma unroll
      for (int t = 0; t < XA_CT; ++t) {
        acc[t] = (f32x16_t){0};
        if (need[t]) {
          const unsigned char* kp = kfrag_base + (c0 + t) * 32 * XA_KSTRIDE;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(kp + s * 32);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[s], acc[t], 0, 0, 0);
          }
        }
      }
      // scaled + masked scores in the log2 domain, chunk max
      float cmax = -INFINITY;
#pragma unroll
      for (int t = 0; t < XA_CT; ++t) {
        if (need[t]) {
          const bool has_pad = (c0 + t + 1) * 32 > L;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int koff = (r & 3) + 8 * (r >> 2);
            const int mb = __builtin_amdgcn_sbfe((int)inv[t], koff, 1);  // -1 if masked
            float y = fmaf(acc[t][r], C, __uint_as_float((uint32_t)mb & bias_bits));
            if (has_pad && ((c0 + t) * 32 + koff + 4 * hi >= L)) y = -INFINITY;
            acc[t][r] = y;
            cmax = fmaxf(cmax, y);
          }
        }
      }
      cmax = xchg32_max(cmax);
      const float m_new = fmaxf(m_run, cmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float csum = 0.f;
#pragma unroll
      for (int t = 0; t < XA_CT; ++t)
        if (need[t]) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float pv = __builtin_amdgcn_exp2f(acc[t][r] - m_new);
            acc[t][r] = pv;
            csum += pv;
          }
        }
      csum = xchg32_sum(csum);
      l_run = l_run * alpha + csum;
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o0[r] *= alpha;
        o1[r] *= alpha;
      }
      // O^T += V^T . P^T : A = V^T fragment (LDS), B = this lane's P values packed to bf16
#pragma unroll
      for (int t = 0; t < XA_CT; ++t)
        if (need[t]) {
#pragma unroll
          for (int gg = 0; gg < 2; ++gg) {
            union {
              uint32_t u[4];
              bf16x8_t v;
            } pf;
#pragma unroll
            for (int e = 0; e < 4; ++e) pf.u[e] = pack_bf16x2(acc[t][8 * gg + 2 * e], acc[t][8 * gg + 2 * e + 1]);
            const unsigned char* vp = vfrag_base + ((c0 + t) * 32 + 16 * gg) * 2;
            const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(vp);
            const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(vp + 32 * VS);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, pf.v, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, pf.v, o1, 0, 0, 0);
          }
        }
    }
    // epilogue: lane (q, hi) holds O[q][32 dt + (r&3) + 8 (r>>2) + 4 hi]
    if (rvalid) {
      const float inv_l = 1.0f / l_run;
      uint16_t* op = out + row * hidden + h * 64 + 4 * hi;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        uint2 w0, w1;
        w0.x = pack_bf16x2(o0[4 * rr] * inv_l, o0[4 * rr + 1] * inv_l);
        w0.y = pack_bf16x2(o0[4 * rr + 2] * inv_l, o0[4 * rr + 3] * inv_l);
        w1.x = pack_bf16x2(o1[4 * rr] * inv_l, o1[4 * rr + 1] * inv_l);
        w1.y = pack_bf16x2(o1[4 * rr + 2] * inv_l, o1[4 * rr + 3] * inv_l);
        *reinterpret_cast<uint2*>(op + 8 * rr) = w0;
        *reinterpret_cast<uint2*>(op + 32 + 8 * rr) = w1;
      }
    }
    cur = nxt;
    tile = tile_next;
  }
}

int psg_cross_attn_simple_launch(const void* q, const void* k, const void* v, const uint64_t* bits, int words,
                                 const int32_t* pair_index, int N, int P, int L, int nq, int heads, int policy,
                                 void* out, int dtype, hipStream_t st);

extern "C" int psg_qformer_cross_attn(psg_ctx* ctx, const void* q, const void* k, const void* v, const uint64_t* bits,
                                      int words, const int32_t* pair_index, int N, int P, int L, int nq, int heads,
                                      int empty_policy, int variant, void* out, int32_t* work_counters, int dtype,
                                      void* stream) {
  PSG_REQUIRE(ctx && q && k && v && bits && pair_index && out, PSG_ERR_INVALID, "psg_qformer_cross_attn: NULL argument");
  PSG_REQUIRE(N > 0 && P >= 0 && L > 0 && nq > 0 && heads > 0 && words * 64 >= L, PSG_ERR_INVALID,
              "psg_qformer_cross_attn: N=%d P=%d L=%d nq=%d heads=%d words=%d", N, P, L, nq, heads, words);
  PSG_REQUIRE(empty_policy == PSG_EMPTY_UNIFORM || empty_policy == PSG_EMPTY_UNMASKED, PSG_ERR_INVALID,
              "psg_qformer_cross_attn: empty_policy=%d", empty_policy);
  if (P == 0) return PSG_OK;
  hipStream_t st = (hipStream_t)stream;
  if (variant == PSG_XATTN_SIMPLE)
    return psg_cross_attn_simple_launch(q, k, v, bits, words, pair_index, N, P, L, nq, heads, empty_policy, out, dtype,
                                        st);
  PSG_REQUIRE(variant == PSG_XATTN_MFMA, PSG_ERR_INVALID, "psg_qformer_cross_attn: variant=%d", variant);
  PSG_REQUIRE(dtype == PSG_BF16, PSG_ERR_UNSUPPORTED,
              "psg_qformer_cross_attn: the MFMA variant computes in bf16; use PSG_XATTN_SIMPLE for fp32");
  PSG_REQUIRE(work_counters != nullptr, PSG_ERR_INVALID,
              "psg_qformer_cross_attn: the MFMA variant needs work_counters (int32[heads] of caller memory)");
  const int Lpad = (L + 31) & ~31;
  const size_t lds = (size_t)Lpad * XA_KSTRIDE + (size_t)64 * (Lpad * 2 + 16);
  PSG_REQUIRE(lds <= 160 * 1024, PSG_ERR_UNSUPPORTED, "psg_qformer_cross_attn: L=%d needs %zu B of LDS (> 160 KiB)", L,
              lds);
  const int NC = (Lpad / 32 + XA_CT - 1) / XA_CT;
  PSG_REQUIRE(NC >= 1 && NC <= 4, PSG_ERR_UNSUPPORTED, "psg_qformer_cross_attn: L=%d (MFMA variant handles L <= 512)", L);
  const void* kfn = NC == 1 ? (const void*)cross_attn_mfma_kernel<1>
                  : NC == 2 ? (const void*)cross_attn_mfma_kernel<2>
                  : NC == 3 ? (const void*)cross_attn_mfma_kernel<3> : (const void*)cross_attn_mfma_kernel<4>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      psg_set_error("psg_qformer_cross_attn: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
      return PSG_ERR_HIP;
    }
  }
  const int64_t R = (int64_t)P * nq;
  const int64_t ntile = nq == 33 ? (int64_t)P + (P + 31) / 32 : (R + 31) / 32;
  // persistent grid: ~2 workgroups per CU, at least one tile per wave
  int blocks_per_cu = lds * 2 <= 160 * 1024 ? 2 : 1;
  int64_t G = ((int64_t)ctx->num_cu * blocks_per_cu + heads - 1) / heads;
  const int64_t maxG = (ntile + 3) / 4;
  if (G > maxG) G = maxG;
  if (G < 1) G = 1;
  {
    hipError_t e = hipMemsetAsync(work_counters, 0, sizeof(int32_t) * heads, st);   // a memset node under capture
    if (e != hipSuccess) {
      psg_set_error("psg_qformer_cross_attn: hipMemsetAsync: %s", hipGetErrorString(e));
      return PSG_ERR_HIP;
    }
  }
#define XLAUNCH(NC_)                                                                                           \
  cross_attn_mfma_kernel<NC_><<<(unsigned)(G * heads), 256, lds, st>>>(                                        \
      (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v, bits, words, pair_index, N, R, L, nq, heads, \
      empty_policy, work_counters, (uint16_t*)out)
  if (NC == 1) XLAUNCH(1);
  else if (NC == 2) XLAUNCH(2);
  else if (NC == 3) XLAUNCH(3);
  else XLAUNCH(4);
#undef XLAUNCH
  PSG_CHECK_LAUNCH("psg_qformer_cross_attn");
  return PSG_OK;
}
