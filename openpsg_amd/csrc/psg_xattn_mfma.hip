// K6: relation-query cross-attention on the CDNA4 matrix cores (primary kernel of the path).
//
// Replaces HF-IB:464-466, 487-496 as driven by V4:168-170, 179-185: the reference expands the SAME
// [L,256] patch tensor to all B = N^2 pairs and re-projects K/V per pair; here K/V [L,768] are
// projected once per image and every pair reads them from LDS.  The pair mask
// (mask_i | mask_j, V4:430-433) is never materialised: each lane ORs two rows of the per-object
// bitmask table and tests bits in registers.
//
// Work layout (gfx950, wave = 64):
//   * a workgroup (4 waves, 2 per CU) owns one head h: it stages K_h [Lpad][64], V_h^T [64][Lpad] (bf16, row
//     strides padded by 16 B => conflict-free ds_read_b128), the mean of V_h and the whole object bit table
//     into LDS once, then its waves walk row tiles in a static round-robin order;
//   * row tiles (nq == 33): first the tiles that batch the cls rows (row 0) of 32 consecutive pairs - they
//     carry per-row masks - then one tile per pair = its rows 1..32, which share ONE mask, so the mask words
//     are wave-uniform, 32-key tiles nobody attends to are skipped (they contribute exactly 0), and an empty
//     union under the "uniform" policy is just the mean of V;
//   * S^T = K . Q^T with v_mfma_f32_32x32x16_bf16 (A = K fragment from LDS, B = Q fragment held in
//     registers), so a lane owns ONE query row (column lane&31 of D) and 16 keys per 32-key tile; for pair
//     tiles the additive mask is one more MFMA (A'[key][0] = 0 / -2^15, B'[0][row] = 1) instead of three
//     VALU instructions per score; the softmax reduction is in-register plus one lane^32 exchange;
//   * O^T = V^T . P^T the same way (A = V^T fragment from LDS, B = this lane's exponentiated scores packed to
//     bf16), so the online-softmax rescale and the final 1/l are lane-local.  V^T is stored with key bits
//     2<->3 swapped inside every 16-key group, which makes the accumulator registers of the S^T tile line up
//     with the B-operand slots of the P.V MFMA without any cross-lane shuffle;
//   * the active key tiles of a unit are a dynamic loop (per-tile online softmax, 22 KB of code instead of
//     55 KB unrolled); the Q fragments and pair ids of the next two units are in flight while one is computed.
// The kernel moves Q in and the context out once each (PMC: 126 MB + 124 MB at N = 50) and is bound by that
// traffic, 128 bytes per row per head; see DESIGN.md for the timeline that led here.
//
// Mask semantics: masked key => score + finfo.min (== finfo.min in fp32) so an
// all-masked row is a UNIFORM softmax over the L real keys; keys in [L, Lpad) are padding (weight 0).
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "psg_common.h"

typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

#define XA_KSTRIDE 144  // bytes per K row in LDS: 64 bf16 + 16 B pad

// value of the partner lane (lane ^ 32) via v_permlane32_swap (VALU, no LDS round trip)
__device__ __forceinline__ float xchg32_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xchg32_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}



template <typename E, int NC>   // NC = key chunks of 128 (L <= 128 NC): unrolled so the prefetched mask words index statically
__global__ void __launch_bounds__(256, 2)
cross_attn_mfma_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
                       const uint64_t* __restrict__ bits, int words, const int32_t* __restrict__ pair_index, int N,
                       int64_t R, int L, int nq, int heads, int policy,
                       uint16_t* __restrict__ out, long long* __restrict__ trace) {
  // trace != nullptr (psg_set_trace_buffer(PSG_TRACE_CROSS_ATTN), debugging only): 32 timestamps per wave
  long long* tr = trace ? trace + ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 : nullptr;
  if (tr && (threadIdx.x & 63) == 0) tr[0] = __builtin_readcyclecounter();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Lpad = (L + 31) & ~31;
  const int NT = Lpad >> 5;
  const int VS = Lpad * 2 + 16;  // bytes per V^T row in LDS
  unsigned char* k_lds = smem;
  unsigned char* vt_lds = smem + (size_t)Lpad * XA_KSTRIDE;
  unsigned char* mean_lds = vt_lds + (size_t)64 * VS;   // 64 floats: mean of V_h over the L keys
  uint64_t* bits_lds = reinterpret_cast<uint64_t*>(mean_lds + 256);   // object bit rows [N][words]
  const int h = blockIdx.x % heads;
  const int g = blockIdx.x / heads;
  const int G = gridDim.x / heads;
  const int hidden = heads * 64;
  const int tid = threadIdx.x;


  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  // Row tiles.  nq == 33: tile t < P = rows 1..32 of pair t (the 32 relation queries share ONE pair
  // mask, so most 32-key tiles are masked for the whole tile and are skipped below); tiles >= P batch
  // the cls rows (row 0) of 32 consecutive pairs.  Other nq: flat 32-row tiles.
  const bool aligned = nq == 33;
  const int64_t P = R / nq;
  const int64_t NCLS = aligned ? (P + 31) >> 5 : 0;
  const int64_t ntile = aligned ? P + NCLS : (R + 31) >> 5;
  const unsigned char* kfrag_base = k_lds + l31 * XA_KSTRIDE + hi * 16;
  const unsigned char* vfrag_base = vt_lds + l31 * VS + hi * 16;

  // One unit of work = (row tile, head h).  Its operands sit behind a chain of dependent global loads
  // (pair_index -> object bit rows -> mask words); most key tiles are skipped, so a unit is short (~1 us)
  // and that chain must be off the critical path.  Three units are in flight: unit A is computed while
  // the Q fragments and mask words of unit B are loading (its pair id arrived during the previous unit)
  // and the pair id of unit C is loading.  Nothing in a fetch waits on a load issued in the same step.
  struct XUnit {
    typename E::v8 qf[4];
    int pidx;                          // pair id p = i * N + j of this lane's row
    int64_t row;
    bool rvalid;
  };
  // Queue order (nq == 33): the NCLS tiles that batch the cls rows (row 0) of 32 consecutive pairs come
  // FIRST - they carry per-row masks and need most key tiles, ~5x the cost of a pair tile; then tile
  // NCLS + p = rows 1..32 of pair p.
  auto unit_rows = [&](int64_t tile, int64_t& row, bool& rvalid, int64_t& pair) {
    if (aligned) {
      if (tile >= NCLS) {
        pair = tile - NCLS;
        row = pair * 33 + 1 + l31;
        rvalid = true;
      } else {
        const int64_t pr = tile * 32 + l31;
        rvalid = pr < P;
        pair = rvalid ? pr : P - 1;
        row = pair * 33;
      }
    } else {
      row = tile * 32 + l31;
      rvalid = row < R;
      if (!rvalid) row = R - 1;
      pair = row / nq;
    }
  };
  // A fetch only ISSUES loads (Q fragments and the pair id); the object bit rows live in LDS, so nothing
  // in the unit's operand chain depends on another global load.
  auto fetch = [&](int64_t tile, XUnit& u) {
    int64_t pair;
    unit_rows(tile, u.row, u.rvalid, pair);
    // Q fragments: B operand of S^T = K.Q^T; lane (q = lane&31, hi) holds Q[q][16 s + 8 hi .. +7]
    const uint16_t* qp = q + u.row * hidden + h * 64 + hi * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) u.qf[s] = *reinterpret_cast<const typename E::v8*>(qp + s * 16);
    u.pidx = pair_index[pair];
  };
  const float rcpN = 1.0f / (float)N;
  // Static round-robin distribution over the waves that own head h (wave w of G*4 takes tiles w, w + 4G, ...).
  // The expensive cls tiles come first in the order, so every wave gets at most one or two of them, and a
  // pair with an empty mask union costs less than an average pair (mean-of-V shortcut), so the remaining
  // cost spread averages out over the ~15 units of a wave.  (A global atomic work queue was measured here:
  // 2048 waves x 2 returning atomics on 12 words cost 44 us of start-up plus a round trip every 4 units.)
  const int64_t wstride = (int64_t)G * 4;
  int64_t wnext = (int64_t)g * 4 + wid;
  auto next_tile = [&]() -> int64_t {
    const int64_t t = wnext;
    wnext += wstride;
    return t;
  };
  // B operand of the mask-bias MFMA: B'[k = 0][row] = 1 for every row, all other k-slots 0
  union {
    uint32_t u[4];
    typename E::v8 v;
  } b_one;
  b_one.u[0] = hi ? 0u : E::ONE;
  b_one.u[1] = b_one.u[2] = b_one.u[3] = 0u;
  const float C8 = 0.125f * 1.4426950408889634f;
  const float bias_raw = policy == PSG_EMPTY_UNIFORM ? -3.4028234663852886e38f : -80000.0f;  // generic path, pre-scale

  // One unit = (row tile, head).  Two code paths:
  //  AL (nq == 33, tile < P): the 32 rows are rows 1..32 of ONE pair and share its mask, so the mask words
  //     are wave-uniform (scalar tile skipping) and the additive mask is applied by the matrix core: one
  //     extra MFMA per key tile adds A'[key][0] * B'[0][row] = bias(key) * 1 to S^T, which replaces three
  //     VALU instructions per score.  A pair with an empty union under the "uniform" policy is the mean of
  //     V over the L keys (precomputed per workgroup): no bias has to absorb the scores, so a moderate
  //     bias (-2^15, exact in bf16) and the fused exp2(fma(s, C, -m C)) are safe.
  //  generic (cls-row tiles, other nq): per-row masks in VALU, absorbing finfo.min bias as in the reference
  //     (HF additive mask), exp2((s - m) * C) so that equal scores give exactly 2^0.
  auto run_unit = [&](const XUnit& cur, auto al_tag) {
    constexpr bool AL = decltype(al_tag)::value;
    const int64_t row = cur.row;
    const bool rvalid = cur.rvalid;
    typename E::v8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = cur.qf[s];

    // i = pidx / N, j = pidx % N; exact in fp32 for pidx < 2^24 and N <= 1024
    int oi, oj;
    if (N <= 1024) {
      oi = (int)(((float)cur.pidx + 0.5f) * rcpN);
      oj = cur.pidx - oi * N;
    } else {
      oi = cur.pidx / N;
      oj = cur.pidx % N;
    }
    // object bit rows as 32-bit words: word t = keys [32 t, 32 t + 32), 1 = attend
    const uint32_t* wi = reinterpret_cast<const uint32_t*>(bits_lds + (int64_t)oi * words);
    const uint32_t* wj = reinterpret_cast<const uint32_t*>(bits_lds + (int64_t)oj * words);
    // which 32-key tiles does this row tile need?  A tile no row attends to contributes exactly 0: skipped.
    uint32_t needmask = 0;      // wave-uniform
    bool force_all = false;     // AL + "unmasked" policy + empty union: plain attention over the L real keys
    if constexpr (AL) {
      for (int t = 0; t < NT; ++t)
        needmask |= ((__builtin_amdgcn_readfirstlane(wi[t] | wj[t]) != 0u) ? 1u : 0u) << t;
      if (needmask == 0u) {
        if (policy == PSG_EMPTY_UNIFORM) {
          // uniform softmax over the L real keys: out = mean_k V[k]
          const float* mean = reinterpret_cast<const float*>(mean_lds);
          uint16_t* op = out + row * hidden + h * 64 + 4 * hi;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const float4 m0 = *reinterpret_cast<const float4*>(mean + 8 * rr + 4 * hi);
            const float4 m1 = *reinterpret_cast<const float4*>(mean + 32 + 8 * rr + 4 * hi);
            uint2 w0, w1;
            w0.x = E::pack(m0.x, m0.y);
            w0.y = E::pack(m0.z, m0.w);
            w1.x = E::pack(m1.x, m1.y);
            w1.y = E::pack(m1.z, m1.w);
            *reinterpret_cast<uint2*>(op + 8 * rr) = w0;
            *reinterpret_cast<uint2*>(op + 32 + 8 * rr) = w1;
          }
          return;
        }
        force_all = true;
        needmask = NT >= 32 ? 0xffffffffu : (1u << NT) - 1u;
      }
    } else {
      bool any_empty = false, row_empty = true;
      for (int t = 0; t < NT; ++t) {
        const uint32_t w = wi[t] | wj[t];
        row_empty = row_empty && (w == 0u);
        needmask |= (__any(w != 0u) ? 1u : 0u) << t;
      }
      // a row whose pair mask is empty attends to every key (uniform softmax): it needs all tiles
      any_empty = __any(row_empty);
      if (any_empty) needmask = NT >= 32 ? 0xffffffffu : (1u << NT) - 1u;
    }

    f32x16_t o0 = {0}, o1 = {0};
    float m_run = -INFINITY, l_run = 0.f;

    while (needmask != 0u) {
      const int t = __builtin_ctz(needmask);
      needmask &= needmask - 1u;
      uint32_t word = wi[t] | wj[t];
      const int left = L - 32 * t;                       // real keys in this tile (>= 1)
      f32x16_t acc;
      const unsigned char* kp = kfrag_base + t * 32 * XA_KSTRIDE;
      if constexpr (AL) {
        if (force_all) word = left >= 32 ? 0xffffffffu : (1u << left) - 1u;
        // A'[key = lane&31][k = 0] = 0 if the pair attends to this key, else -2^15 (bf16 0xc700)
        union {
          uint32_t u[4];
          typename E::v8 v;
        } a_bias;
        a_bias.u[0] = (((word >> l31) & 1u) | (uint32_t)hi) ? 0u : E::NEG_2_15;
        a_bias.u[1] = a_bias.u[2] = a_bias.u[3] = 0u;
        acc = E::mfma32(a_bias.v, b_one.v, (f32x16_t){0});
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const typename E::v8 a = *reinterpret_cast<const typename E::v8*>(kp + s * 32);
          acc = E::mfma32(a, qf[s], acc);
        }
      } else {
        {
          const typename E::v8 a = *reinterpret_cast<const typename E::v8*>(kp);
          acc = E::mfma32(a, qf[0], (f32x16_t){0});
        }
#pragma unroll
        for (int s = 1; s < 4; ++s) {
          const typename E::v8 a = *reinterpret_cast<const typename E::v8*>(kp + s * 32);
          acc = E::mfma32(a, qf[s], acc);
        }
        // additive mask per (row, key): register r of a lane is key (r&3) + 8 (r>>2) + 4 hi of the tile
        const uint32_t inv = ~word >> (4 * hi);
        const bool has_pad = left < 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int koff = (r & 3) + 8 * (r >> 2);
          const int mb = __builtin_amdgcn_sbfe((int)inv, koff, 1);  // -1 if masked
          float y = acc[r] + __uint_as_float((uint32_t)mb & __float_as_uint(bias_raw));
          if (has_pad && (koff + 4 * hi >= left)) y = -INFINITY;
          acc[r] = y;
        }
      }
      // online softmax over the raw scores (the scale is positive)
      float cmax = acc[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) cmax = fmaxf(cmax, acc[r]);
      cmax = xchg32_max(cmax);
      const float m_new = fmaxf(m_run, cmax);
      float alpha, csum = 0.f;
      if constexpr (AL) {
        const float mc = m_new * C8;
        alpha = __builtin_amdgcn_exp2f(fmaf(m_run, C8, -mc));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(fmaf(acc[r], C8, -mc));
          acc[r] = pv;
          csum += pv;
        }
      } else {
        alpha = __builtin_amdgcn_exp2f((m_run - m_new) * C8);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f((acc[r] - m_new) * C8);
          acc[r] = pv;
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
      for (int gg = 0; gg < 2; ++gg) {
        union {
          uint32_t u[4];
          typename E::v8 v;
        } pf;
#pragma unroll
        for (int e = 0; e < 4; ++e) pf.u[e] = E::pack(acc[8 * gg + 2 * e], acc[8 * gg + 2 * e + 1]);
        const unsigned char* vp = vfrag_base + (t * 32 + 16 * gg) * 2;
        const typename E::v8 a0 = *reinterpret_cast<const typename E::v8*>(vp);
        const typename E::v8 a1 = *reinterpret_cast<const typename E::v8*>(vp + 32 * VS);
        o0 = E::mfma32(a0, pf.v, o0);
        o1 = E::mfma32(a1, pf.v, o1);
      }
    }
    // epilogue: lane (q, hi) holds O[q][32 dt + (r&3) + 8 (r>>2) + 4 hi]
    if (AL || rvalid) {
      const float inv_l = 1.0f / l_run;
      uint16_t* op = out + row * hidden + h * 64 + 4 * hi;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        uint2 w0, w1;
        w0.x = E::pack(o0[4 * rr] * inv_l, o0[4 * rr + 1] * inv_l);
        w0.y = E::pack(o0[4 * rr + 2] * inv_l, o0[4 * rr + 3] * inv_l);
        w1.x = E::pack(o1[4 * rr] * inv_l, o1[4 * rr + 1] * inv_l);
        w1.y = E::pack(o1[4 * rr + 2] * inv_l, o1[4 * rr + 3] * inv_l);
        *reinterpret_cast<uint2*>(op + 8 * rr) = w0;
        *reinterpret_cast<uint2*>(op + 32 + 8 * rr) = w1;
      }
    }
  };

  int nun = 0;
  auto stamp = [&]() {
    if (tr && lane == 0) {
      if (nun < 28) tr[3 + nun] = __builtin_readcyclecounter();
      ++nun;
      tr[31] = nun;
    }
  };
  // The first units' operands are requested BEFORE the K/V staging: the first touch of Q by 2048 waves at
  // once takes 15-25 us, which now overlaps the staging and the cls unit instead of following them.
  const int64_t tlast = ntile - 1;
  int64_t t = next_tile();
  int64_t tal = t;                     // first pair tile of this wave
  if (aligned) {
    while (tal < NCLS) tal += wstride;
  } else {
    tal = ntile;
  }
  XUnit uf, u0, u1, u2;
  fetch(t < tlast ? t : tlast, uf);
  fetch(tal < tlast ? tal : tlast, u0);
  fetch(tal + wstride < tlast ? tal + wstride : tlast, u1);

  for (int e = threadIdx.x; e < N * words; e += 256) bits_lds[e] = bits[e];
  // ---- stage K_h and V_h^T (once per workgroup) ----
  // All global loads of a thread are issued before the first LDS write (a load -> write loop paid one
  // L2 round trip per iteration: 20 us of prologue).  A thread owns the key PAIR (2m, 2m+1) of one 8-dim
  // chunk c: the two keys are neighbours in the V^T row, so the transposed writes are 32-bit.
  {
    constexpr int IT = 2 * NC;                           // (Lpad/2 key pairs * 8 chunks) / 256 threads <= 2 NC
    uint4 kv[IT][2], vv[IT][2];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = tid + it * 256;
      const int m = e >> 3, c = e & 7;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int key = 2 * m + u;
        kv[it][u] = make_uint4(0, 0, 0, 0);
        vv[it][u] = make_uint4(0, 0, 0, 0);
        if (key < L) {
          kv[it][u] = *reinterpret_cast<const uint4*>(k + (int64_t)key * hidden + h * 64 + c * 8);
          vv[it][u] = *reinterpret_cast<const uint4*>(v + (int64_t)key * hidden + h * 64 + c * 8);
        }
      }
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = tid + it * 256;
      const int m = e >> 3, c = e & 7;
      if (2 * m < Lpad) {
        *reinterpret_cast<uint4*>(k_lds + (2 * m) * XA_KSTRIDE + c * 16) = kv[it][0];
        *reinterpret_cast<uint4*>(k_lds + (2 * m + 1) * XA_KSTRIDE + c * 16) = kv[it][1];
        const int o = (2 * m) & 15;
        const int pos = (o & 3) | ((o & 8) >> 1) | ((o & 4) << 1);  // swap bits 2 <-> 3 (bit 0 stays: pair adjacent)
        const int kcol = (((2 * m) & ~15) | pos) * 2;
        const uint32_t a[4] = {vv[it][0].x, vv[it][0].y, vv[it][0].z, vv[it][0].w};
        const uint32_t bq[4] = {vv[it][1].x, vv[it][1].y, vv[it][1].z, vv[it][1].w};
#pragma unroll
        for (int d2 = 0; d2 < 4; ++d2) {
          const uint32_t lo = (a[d2] & 0xffffu) | (bq[d2] << 16);            // dim 2 d2    of keys 2m, 2m+1
          const uint32_t hi2 = (a[d2] >> 16) | (bq[d2] & 0xffff0000u);      // dim 2 d2 + 1
          *reinterpret_cast<uint32_t*>(vt_lds + (c * 8 + 2 * d2) * VS + kcol) = lo;
          *reinterpret_cast<uint32_t*>(vt_lds + (c * 8 + 2 * d2 + 1) * VS + kcol) = hi2;
        }
      }
    }
  }
  __syncthreads();
  if (tid < 64) {                                       // pad keys hold zeros: sum over all Lpad slots
    float sum = 0.f;
    const uint16_t* vr = reinterpret_cast<const uint16_t*>(vt_lds + tid * VS);
    for (int kk = 0; kk < Lpad; kk += 8) {
      const uint4 x = *reinterpret_cast<const uint4*>(vr + kk);
      const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) sum += E::to_f32((uint16_t)(xs[e] & 0xffffu)) + E::to_f32((uint16_t)(xs[e] >> 16));
    }
    reinterpret_cast<float*>(mean_lds)[tid] = sum / (float)L;
  }
  __syncthreads();
  if (tr && lane == 0) tr[1] = __builtin_readcyclecounter();
  if (tr && lane == 0) tr[2] = __builtin_readcyclecounter();
  // Phase 1: tiles with per-row masks (the cls tiles at the head of the order: at most one or two per wave;
  // every tile when nq != 33).
  {
    bool first = true;
    while (t < ntile && t < tal) {
      if (!first) fetch(t, uf);
      first = false;
      run_unit(uf, std::false_type{});
      stamp();
      t = next_tile();
    }
  }
  if (t >= ntile) return;
  // Phase 2: pair tiles.  Three units in flight - one is computed, the operands of the next have been
  // loading for one unit time, the loads of the third are issued now.  The three register sets rotate by
  // unrolling (a copy `a = b` would have to wait for b's loads); fetches are unconditional (tile clamped to
  // the last one) because a branch around them makes the compiler wait for the fresh loads at the join.
  // The compiler's s_waitcnt counts at the loop header are the minimum over the entry edge and the back edge
  // of "memory operations issued after the load".  In steady state 8 output stores sit between two fetches;
  // with fewer operations on the entry edge every unit would wait for the previous unit's stores to be
  // acknowledged.  Volatile (harmless) loads stand in for them.
#pragma unroll
  for (int e = 0; e < 16; ++e) (void)*reinterpret_cast<const volatile int*>(pair_index);
#define XA_STEP(COMPUTE, FILL)                                                  \
  {                                                                             \
    const int64_t t2 = t + 2 * wstride;                                         \
    fetch(t2 < tlast ? t2 : tlast, FILL);                                       \
    run_unit(COMPUTE, std::true_type{});                                        \
    stamp();                                                                    \
    t += wstride;                                                               \
    if (t >= ntile) break;                                                      \
  }
  for (;;) {
    XA_STEP(u0, u2)
    XA_STEP(u1, u0)
    XA_STEP(u2, u1)
  }
#undef XA_STEP
}

int psg_cross_attn_dma_launch(psg_ctx* ctx, const void* q, const void* k, const void* v, const uint64_t* bits,
                              int words, const int32_t* pair_index, int N, int P, int L, int nq, int heads, int policy,
                              void* out, int dtype, hipStream_t st, const int32_t* q_index = nullptr,
                              const void* q_cls = nullptr);
extern "C" int psg_cross_attn_dma_lds_bytes(int N, int words, int L);

int psg_cross_attn_simple_launch(const void* q, const void* k, const void* v, const uint64_t* bits, int words,
                                 const int32_t* pair_index, int N, int P, int L, int nq, int heads, int policy,
                                 void* out, int dtype, hipStream_t st);

template <typename E>
static int xa_v1_launch(psg_ctx* ctx, const void* q, const void* k, const void* v, const uint64_t* bits, int words,
                        const int32_t* pair_index, int N, int P, int L, int nq, int heads, int empty_policy, void* out,
                        hipStream_t st) {
  const int Lpad = (L + 31) & ~31;
  const size_t lds = (size_t)Lpad * XA_KSTRIDE + (size_t)64 * (Lpad * 2 + 16) + 256 + (((size_t)N * words * 8 + 15) & ~(size_t)15);
  PSG_REQUIRE(lds <= 160 * 1024, PSG_ERR_UNSUPPORTED, "psg_qformer_cross_attn: L=%d needs %zu B of LDS (> 160 KiB)", L,
              lds);
  const int NC = (Lpad / 32 + 3) / 4;                     // template parameter: staging passes / 128 keys
  PSG_REQUIRE(NC >= 1 && NC <= 4, PSG_ERR_UNSUPPORTED, "psg_qformer_cross_attn: L=%d (MFMA variant handles L <= 512)", L);
  const void* kfn = NC == 1 ? (const void*)cross_attn_mfma_kernel<E, 1>
                  : NC == 2 ? (const void*)cross_attn_mfma_kernel<E, 2>
                  : NC == 3 ? (const void*)cross_attn_mfma_kernel<E, 3> : (const void*)cross_attn_mfma_kernel<E, 4>;
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
  const int64_t trace_n = G * heads * 4 * 32;
  long long* trace = (ctx->trace_kind == PSG_TRACE_CROSS_ATTN && ctx->trace_words >= trace_n) ? ctx->trace : nullptr;
#define XLAUNCH(NC_)                                                                                           \
  cross_attn_mfma_kernel<E, NC_><<<(unsigned)(G * heads), 256, lds, st>>>(                                        \
      (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v, bits, words, pair_index, N, R, L, nq, heads, \
      empty_policy, (uint16_t*)out, trace)
  if (NC == 1) XLAUNCH(1);
  else if (NC == 2) XLAUNCH(2);
  else if (NC == 3) XLAUNCH(3);
  else XLAUNCH(4);
#undef XLAUNCH
  PSG_CHECK_LAUNCH("psg_qformer_cross_attn");
  return PSG_OK;
}

int psg_cross_attn_f32_launch(const void* q, const void* k, const void* v, const uint64_t* bits, int words,
                              const int32_t* pair_index, int N, int P, int L, int nq, int heads, int policy, void* out,
                              hipStream_t st);

extern "C" int psg_qformer_cross_attn(psg_ctx* ctx, const void* q, const void* k, const void* v, const uint64_t* bits,
                                      int words, const int32_t* pair_index, int N, int P, int L, int nq, int heads,
                                      int empty_policy, int variant, void* out, int dtype, void* stream) {
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
  PSG_REQUIRE(variant == PSG_XATTN_MFMA || variant == PSG_XATTN_MFMA_V1, PSG_ERR_INVALID,
              "psg_qformer_cross_attn: variant=%d", variant);
  // fp32 activations: exact f32 matrix instructions over the compacted key list (psg_attn_f32.hip)
  if (dtype == PSG_F32 && variant == PSG_XATTN_MFMA)
    return psg_cross_attn_f32_launch(q, k, v, bits, words, pair_index, N, P, L, nq, heads, empty_policy, out, st);
  PSG_REQUIRE(dtype == PSG_BF16 || dtype == PSG_F16, PSG_ERR_UNSUPPORTED,
              "psg_qformer_cross_attn: PSG_XATTN_MFMA_V1 computes in bf16 / fp16");
  // second-generation kernel (full-line Q / context traffic through LDS-DMA) whenever its LDS image fits
  if (variant == PSG_XATTN_MFMA && ctx->opt.xattn_dma && psg_cross_attn_dma_lds_bytes(N, words, L) <= 160 * 1024 &&
      L <= 384)
    return psg_cross_attn_dma_launch(ctx, q, k, v, bits, words, pair_index, N, P, L, nq, heads, empty_policy, out, dtype,
                                     st);
  PSG_DISPATCH_E16(dtype, "psg_qformer_cross_attn", return xa_v1_launch<E>(ctx, q, k, v, bits, words, pair_index, N, P, L, nq,
                                                                            heads, empty_policy, out, st));
}


// psg_qformer_cross_attn with the queries stored ONCE PER PROMPT (HF-IB:464-496 over the prompt-deduplicated layer 0):
// q_u [U][33][hidden] = the projected query rows of the U distinct prompts, q_index [P] = the prompt of each pair, q_cls
// [P][hidden] = row 0 of every pair (gathered by the caller: P rows).  Saves the [P x 33][hidden] expansion of q (127 MB
// written and read again at BASELINE C2).  Only the LDS-DMA kernel takes the index: PSG_ERR_UNSUPPORTED when it cannot run
// (fp32, L > 384, LDS image too large, option xattn_dma = 0) - the caller then expands q and calls psg_qformer_cross_attn.
extern "C" int psg_qformer_cross_attn_indexed(psg_ctx* ctx, const void* q_u, const int32_t* q_index, const void* q_cls,
                                              const void* k, const void* v, const uint64_t* bits, int words,
                                              const int32_t* pair_index, int N, int P, int L, int heads, int empty_policy,
                                              void* out, int dtype, void* stream) {
  PSG_REQUIRE(ctx && q_u && q_index && q_cls && k && v && bits && pair_index && out, PSG_ERR_INVALID,
              "psg_qformer_cross_attn_indexed: NULL argument");
  PSG_REQUIRE(N > 0 && P >= 0 && L > 0 && heads > 0 && words * 64 >= L, PSG_ERR_INVALID,
              "psg_qformer_cross_attn_indexed: N=%d P=%d L=%d heads=%d words=%d", N, P, L, heads, words);
  PSG_REQUIRE(empty_policy == PSG_EMPTY_UNIFORM || empty_policy == PSG_EMPTY_UNMASKED, PSG_ERR_INVALID,
              "psg_qformer_cross_attn_indexed: empty_policy=%d", empty_policy);
  PSG_REQUIRE((dtype == PSG_BF16 || dtype == PSG_F16) && ctx->opt.xattn_dma && L <= 384 &&
                  psg_cross_attn_dma_lds_bytes(N, words, L) <= 160 * 1024,
              PSG_ERR_UNSUPPORTED, "psg_qformer_cross_attn_indexed: needs the LDS-DMA kernel (16-bit, L <= 384, image in LDS)");
  if (P == 0) return PSG_OK;
  return psg_cross_attn_dma_launch(ctx, q_u, k, v, bits, words, pair_index, N, P, L, 33, heads, empty_policy, out, dtype,
                                   (hipStream_t)stream, q_index, q_cls);
}
