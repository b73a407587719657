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

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  f32x2_t f = {lo, hi};
  bf16x2_t b = __builtin_convertvector(f, bf16x2_t);  // v_cvt_pk_bf16_f32 on gfx950
  return __builtin_bit_cast(uint32_t, b);
}

__global__ void __launch_bounds__(256, 2)
cross_attn_mfma_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
                       const uint64_t* __restrict__ bits, int words, const int32_t* __restrict__ pair_index, int N,
                       int64_t R, int L, int nq, int heads, int policy, int tiles_per_head_block,
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
  const int64_t ntile = (R + 31) >> 5;
  const float C = 0.125f * 1.4426950408889634f;  // 1/sqrt(64) * log2(e)
  const uint32_t bias_bits = policy == PSG_EMPTY_UNIFORM ? 0xff7fffffu /* finfo.min */
                                                         : __float_as_uint(-10000.0f * 1.4426950408889634f);
  const unsigned char* kfrag_base = k_lds + l31 * XA_KSTRIDE + hi * 16;
  const unsigned char* vfrag_base = vt_lds + l31 * VS + hi * 16;

  for (int64_t tile = (int64_t)g * 4 + wid; tile < ntile; tile += (int64_t)G * 4) {
    const int64_t row = tile * 32 + l31;
    const bool rvalid = row < R;
    const int64_t rowc = rvalid ? row : R - 1;
    // Q fragments: B operand of S^T = K.Q^T; lane (q = lane&31, hi) holds Q[q][16 s + 8 hi .. +7]
    bf16x8_t qf[4];
    {
      const uint16_t* qp = q + rowc * hidden + h * 64 + hi * 8;
#pragma unroll
      for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8_t*>(qp + s * 16);
    }
    const int pidx = pair_index[rowc / nq];
    const uint64_t* bi = bits + (int64_t)(pidx / N) * words;
    const uint64_t* bj = bits + (int64_t)(pidx % N) * words;

    f32x16_t o0 = {0}, o1 = {0};
    float m_run = -INFINITY, l_run = 0.f;

    for (int c0 = 0; c0 < NT; c0 += XA_CT) {
      // inverted mask words (1 = masked), pre-shifted by 4*hi so the bit index is a constant per register
      uint32_t inv[XA_CT];
      {
        const int w0 = c0 >> 1;
        const uint64_t m0 = w0 < words ? (bi[w0] | bj[w0]) : 0ull;
        const uint64_t m1 = (w0 + 1) < words ? (bi[w0 + 1] | bj[w0 + 1]) : 0ull;
        inv[0] = ~(uint32_t)m0 >> (4 * hi);
        inv[1] = ~(uint32_t)(m0 >> 32) >> (4 * hi);
        inv[2] = ~(uint32_t)m1 >> (4 * hi);
        inv[3] = ~(uint32_t)(m1 >> 32) >> (4 * hi);
      }
      f32x16_t acc[XA_CT];
#pragma unroll
      for (int t = 0; t < XA_CT; ++t) {
        acc[t] = (f32x16_t){0};
        if (c0 + t < NT) {
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
        if (c0 + t < NT) {
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
      cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
      const float m_new = fmaxf(m_run, cmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float csum = 0.f;
#pragma unroll
      for (int t = 0; t < XA_CT; ++t)
        if (c0 + t < NT) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float pv = __builtin_amdgcn_exp2f(acc[t][r] - m_new);
            acc[t][r] = pv;
            csum += pv;
          }
        }
      csum += __shfl_xor(csum, 32, 64);
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
        if (c0 + t < NT) {
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
  }
}

int psg_cross_attn_simple_launch(const void* q, const void* k, const void* v, const uint64_t* bits, int words,
                                 const int32_t* pair_index, int N, int P, int L, int nq, int heads, int policy,
                                 void* out, int dtype, hipStream_t st);

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
  PSG_REQUIRE(variant == PSG_XATTN_MFMA, PSG_ERR_INVALID, "psg_qformer_cross_attn: variant=%d", variant);
  PSG_REQUIRE(dtype == PSG_BF16, PSG_ERR_UNSUPPORTED,
              "psg_qformer_cross_attn: the MFMA variant computes in bf16; use PSG_XATTN_SIMPLE for fp32");
  const int Lpad = (L + 31) & ~31;
  const size_t lds = (size_t)Lpad * XA_KSTRIDE + (size_t)64 * (Lpad * 2 + 16);
  PSG_REQUIRE(lds <= 160 * 1024, PSG_ERR_UNSUPPORTED, "psg_qformer_cross_attn: L=%d needs %zu B of LDS (> 160 KiB)", L,
              lds);
  static size_t configured = 0;
  if (lds > configured) {
    hipError_t e = hipFuncSetAttribute((const void*)cross_attn_mfma_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      psg_set_error("psg_qformer_cross_attn: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
      return PSG_ERR_HIP;
    }
    configured = lds;
  }
  const int64_t R = (int64_t)P * nq;
  const int64_t ntile = (R + 31) / 32;
  // persistent grid: ~2 workgroups per CU, at least one tile per wave
  int blocks_per_cu = lds * 2 <= 160 * 1024 ? 2 : 1;
  int64_t G = ((int64_t)ctx->num_cu * blocks_per_cu + heads - 1) / heads;
  const int64_t maxG = (ntile + 3) / 4;
  if (G > maxG) G = maxG;
  if (G < 1) G = 1;
  cross_attn_mfma_kernel<<<(unsigned)(G * heads), 256, lds, st>>>(
      (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v, bits, words, pair_index, N, R, L, nq, heads,
      empty_policy, 0, (uint16_t*)out);
  PSG_CHECK_LAUNCH("psg_qformer_cross_attn");
  return PSG_OK;
}
