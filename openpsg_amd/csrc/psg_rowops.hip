// Row-wise HBM-bound kernels: one wave (64 lanes) per row, 4 contiguous elements per lane per
// 256-element chunk, statistics by wavefront shuffle reductions in fp32.
//   K4  Q-Former embeddings + LayerNorm        (HF-IB:728-757)
//       add + LayerNorm                        (HF-IB:519-530, 585-596)
//       bias + GELU(erf)                       (HF-IB:563-577)
//   K8  pair-existence head                    (V4:206-209)
//   K12 RMSNorm (+ residual add)               (HF-LL:53-67)
//   K13 rotary + KV-cache write                (HF-LL:130-160)
//       SwiGLU gate                            (HF-LL:163-177)
//   K16 greedy argmax step                     (V4:305-312)
#include <type_traits>

#include "psg_common.h"
#include "psg_decode_math.h"

// ---- LayerNorm over a row held in registers ---------------------------------------------------
template <int NCH>
__device__ __forceinline__ void ln_row(float (&v)[NCH][4], int hidden, float eps, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, int lane) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
  const float mean = wave_sum_dpp(s) / (float)hidden;             // row-shift reductions: no LDS crossbar round trips
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float d = v[c][e] - mean;
      q += d * d;
    }
  const float rstd = 1.0f / sqrtf(wave_sum_dpp(q) / (float)hidden + eps);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 256 + lane * 4;
    const float4 g = *reinterpret_cast<const float4*>(gamma + col);
    const float4 b = *reinterpret_cast<const float4*>(beta + col);
    v[c][0] = (v[c][0] - mean) * rstd * g.x + b.x;
    v[c][1] = (v[c][1] - mean) * rstd * g.y + b.y;
    v[c][2] = (v[c][2] - mean) * rstd * g.z + b.z;
    v[c][3] = (v[c][3] - mean) * rstd * g.w + b.w;
  }
}

// ---- K4 ---------------------------------------------------------------------------------------
template <typename T, int NCH>
__global__ void qformer_embed_kernel(const int32_t* __restrict__ ids, int B, int Tt, const float* __restrict__ word,
                                     const float* __restrict__ pos, const float* __restrict__ query, int nq,
                                     const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                     int hidden, T* __restrict__ out) {
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int64_t nqrows = (int64_t)B * nq;
  if (row >= nqrows + (int64_t)B * Tt) return;
  float v[NCH][4];
  if (row < nqrows) {                                 // (nq == 0 never gets here)
    const int r = (int)(row % nq);
#pragma unroll
    for (int c = 0; c < NCH; ++c) Act<float>::ld4(query, (int64_t)r * hidden + c * 256 + lane * 4, v[c]);
  } else {
    const int64_t tr = row - nqrows;
    const int t = (int)(tr % Tt);
    const int id = ids[tr];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float a[4], b[4];
      Act<float>::ld4(word, (int64_t)id * hidden + c * 256 + lane * 4, a);
      Act<float>::ld4(pos, (int64_t)t * hidden + c * 256 + lane * 4, b);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[c][e] = a[e] + b[e];
    }
  }
  ln_row<NCH>(v, hidden, eps, gamma, beta, lane);
#pragma unroll
  for (int c = 0; c < NCH; ++c) Act<T>::st4(out, row * hidden + c * 256 + lane * 4, v[c]);
}

extern "C" int psg_qformer_embed(psg_ctx* ctx, const int32_t* ids, int B, int T_, const float* word_emb,
                                 const float* pos_emb, const float* query_rows, int nq, const float* ln_w,
                                 const float* ln_b, float eps, int hidden, void* out, int dtype, void* stream) {
  PSG_REQUIRE(ctx && word_emb && pos_emb && query_rows && ln_w && ln_b && out && (ids || T_ == 0), PSG_ERR_INVALID,
              "psg_qformer_embed: NULL argument");
  PSG_REQUIRE(hidden == 768, PSG_ERR_UNSUPPORTED, "psg_qformer_embed: hidden=%d (kernel is built for 768)", hidden);
  PSG_REQUIRE(B > 0 && T_ >= 0 && nq >= 0 && nq + T_ > 0, PSG_ERR_INVALID, "psg_qformer_embed: B=%d T=%d nq=%d", B, T_,
              nq);                                    // nq == 0: text rows only; T == 0: query rows only
  int64_t rows = (int64_t)B * (nq + T_);
  dim3 grid((unsigned)((rows + 3) / 4));
  PSG_DISPATCH_DTYPE(dtype, "psg_qformer_embed",
                     (qformer_embed_kernel<T, 3><<<grid, 256, 0, (hipStream_t)stream>>>(
                         ids, B, T_, word_emb, pos_emb, query_rows, nq, ln_w, ln_b, eps, hidden, (T*)out)));
  PSG_CHECK_LAUNCH("psg_qformer_embed");
  return PSG_OK;
}

// ---- add + LayerNorm --------------------------------------------------------------------------
template <typename T, int NCH>
__global__ void add_layernorm_kernel(const T* __restrict__ x, const T* __restrict__ res, const float* __restrict__ bias,
                                     const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                     int64_t rows, int hidden, T* __restrict__ out, int res_period,
                                     const int32_t* __restrict__ res_index) {
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  // periodic residual: a [res_period][hidden] table; indexed: rows come in groups of res_period, group g takes block
  // res_index[g] of the table
  int64_t rrow = res_period > 0 ? row % res_period : row;
  if (res_index) rrow += (int64_t)res_index[row / res_period] * res_period;
  float v[NCH][4];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 256 + lane * 4;
    Act<T>::ld4(x, row * hidden + col, v[c]);
    if (bias) {
      const float4 b = *reinterpret_cast<const float4*>(bias + col);
      v[c][0] += b.x; v[c][1] += b.y; v[c][2] += b.z; v[c][3] += b.w;
    }
    if (res) {
      float r[4];
      Act<T>::ld4(res, rrow * hidden + col, r);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[c][e] += r[e];
    }
  }
  ln_row<NCH>(v, hidden, eps, gamma, beta, lane);
#pragma unroll
  for (int c = 0; c < NCH; ++c) Act<T>::st4(out, row * hidden + c * 256 + lane * 4, v[c]);
}

// 16-bit rows of 768: HALF a wave per row, 16-byte accesses (a lane owns 3 x 8 consecutive features), the two sums by
// row shifts + one row broadcast inside each half (no LDS crossbar, no block barrier).  Same arithmetic as
// add_layernorm_kernel (fp32 sums in another association): x + bias + residual -> LayerNorm -> 16-bit.
template <typename E>
__global__ void __launch_bounds__(256) add_layernorm16_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ res,
                                                              const float* __restrict__ bias, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps, int64_t rows,
                                                              uint16_t* __restrict__ out, int res_period,
                                                              const int32_t* __restrict__ res_index) {
  constexpr int H = 768;
  const int lane = threadIdx.x & 63, hl = lane & 31;
  int64_t row = ((((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) << 1) + (lane >> 5);
  const bool live = row < rows;
  if (!live) row = rows - 1;                                      // keeps the half-wave in step; nothing is stored
  int64_t rrow = res_period > 0 ? row % res_period : row;
  if (res_index) rrow += (int64_t)res_index[row / res_period] * res_period;
  float v[3][8];
  uint4 xw[3], rw[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int col = (c * 32 + hl) * 8;
    xw[c] = *reinterpret_cast<const uint4*>(x + row * H + col);
    if (res) rw[c] = *reinterpret_cast<const uint4*>(res + rrow * H + col);
  }
  auto unpack = [](const uint4& w, float (&f)[8]) {
    const uint32_t u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = E::to_f32((uint16_t)(u[i] & 0xffffu));
      f[2 * i + 1] = E::to_f32((uint16_t)(u[i] >> 16));
    }
  };
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int col = (c * 32 + hl) * 8;
    unpack(xw[c], v[c]);
    if (bias) {
      const float4 b0 = *reinterpret_cast<const float4*>(bias + col), b1 = *reinterpret_cast<const float4*>(bias + col + 4);
      v[c][0] += b0.x; v[c][1] += b0.y; v[c][2] += b0.z; v[c][3] += b0.w;
      v[c][4] += b1.x; v[c][5] += b1.y; v[c][6] += b1.z; v[c][7] += b1.w;
    }
    if (res) {
      float r[8];
      unpack(rw[c], r);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[c][e] += r[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[c][e];
  }
  auto half_sum = [&](float t) {                                  // sum over the 32 lanes of this lane's half
    t += psg_dpp<0x111, 0xf>(0.f, t);
    t += psg_dpp<0x112, 0xf>(0.f, t);
    t += psg_dpp<0x114, 0xf>(0.f, t);
    t += psg_dpp<0x118, 0xf>(0.f, t);
    t += psg_dpp<0x142, 0xa>(0.f, t);                             // row 0 -> row 1, row 2 -> row 3: lanes 31 / 63 hold the halves
    const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), 31));
    const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), 63));
    return lane < 32 ? lo : hi;
  };
  const float mean = half_sum(s) * (1.0f / H);
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float d = v[c][e] - mean;
      q += d * d;
    }
  const float rstd = 1.0f / sqrtf(half_sum(q) * (1.0f / H) + eps);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int col = (c * 32 + hl) * 8;
    const float4 g0 = *reinterpret_cast<const float4*>(gamma + col), g1 = *reinterpret_cast<const float4*>(gamma + col + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(beta + col), b1 = *reinterpret_cast<const float4*>(beta + col + 4);
    const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (v[c][e] - mean) * rstd * g[e] + b[e];
    if (live)
      *reinterpret_cast<uint4*>(out + row * H + col) =
          make_uint4(E::pack(o[0], o[1]), E::pack(o[2], o[3]), E::pack(o[4], o[5]), E::pack(o[6], o[7]));
  }
}

// Mixed mode (16-bit GEMM operands, fp32 residual stream): x is the 16-bit output of a projection, the residual is the
// fp32 copy of the previous LayerNorm's output, and the result is written twice - fp32 (the next residual, never
// rounded to 16 bits) and 16-bit (the next projection's operand).  A wave per row of 768; the residual row follows the
// same plain / periodic / indexed rules as add_layernorm_kernel.
template <typename E>
__global__ void __launch_bounds__(256) add_layernorm_res32_kernel(const uint16_t* __restrict__ x, const float* __restrict__ res,
                                                                  const float* __restrict__ bias, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float eps, int64_t rows,
                                                                  uint16_t* __restrict__ out16, float* __restrict__ out32,
                                                                  int res_period, const int32_t* __restrict__ res_index) {
  constexpr int H = 768;
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  int64_t rrow = res_period > 0 ? row % res_period : row;
  if (res_index) rrow += (int64_t)res_index[row / res_period] * res_period;
  float v[3][4];
  uint2 xw[3];
  float4 rw[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int col = c * 256 + lane * 4;
    xw[c] = *reinterpret_cast<const uint2*>(x + row * H + col);
    if (res) rw[c] = *reinterpret_cast<const float4*>(res + rrow * H + col);
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int col = c * 256 + lane * 4;
    v[c][0] = E::to_f32((uint16_t)(xw[c].x & 0xffffu));
    v[c][1] = E::to_f32((uint16_t)(xw[c].x >> 16));
    v[c][2] = E::to_f32((uint16_t)(xw[c].y & 0xffffu));
    v[c][3] = E::to_f32((uint16_t)(xw[c].y >> 16));
    if (bias) {
      const float4 b = *reinterpret_cast<const float4*>(bias + col);
      v[c][0] += b.x; v[c][1] += b.y; v[c][2] += b.z; v[c][3] += b.w;
    }
    if (res) {
      v[c][0] += rw[c].x; v[c][1] += rw[c].y; v[c][2] += rw[c].z; v[c][3] += rw[c].w;
    }
  }
  ln_row<3>(v, H, eps, gamma, beta, lane);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int col = c * 256 + lane * 4;
    if (out32) *reinterpret_cast<float4*>(out32 + row * H + col) = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
    if (out16)
      *reinterpret_cast<uint2*>(out16 + row * H + col) = make_uint2(E::pack(v[c][0], v[c][1]), E::pack(v[c][2], v[c][3]));
  }
}

extern "C" int psg_add_layernorm_res32(psg_ctx* ctx, const void* x, const float* residual, int res_period,
                                       const int32_t* res_index, const float* bias, const float* gamma, const float* beta,
                                       float eps, int64_t rows, int hidden, void* out16, float* out32, int dtype,
                                       void* stream) {
  PSG_REQUIRE(ctx && x && gamma && beta && (out16 || out32), PSG_ERR_INVALID, "psg_add_layernorm_res32: NULL argument");
  PSG_REQUIRE(hidden == 768, PSG_ERR_UNSUPPORTED, "psg_add_layernorm_res32: hidden=%d (kernel is built for 768)", hidden);
  PSG_REQUIRE(res_period >= 0 && (!res_index || (res_period > 0 && rows % res_period == 0)), PSG_ERR_INVALID,
              "psg_add_layernorm_res32: res_period=%d rows=%lld", res_period, (long long)rows);
  if (rows == 0) return PSG_OK;
  dim3 grid((unsigned)((rows + 3) / 4));
  PSG_DISPATCH_E16(dtype, "psg_add_layernorm_res32",
                   (add_layernorm_res32_kernel<E><<<grid, 256, 0, (hipStream_t)stream>>>(
                       (const uint16_t*)x, residual, bias, gamma, beta, eps, rows, (uint16_t*)out16, out32, res_period,
                       res_index)));
  PSG_CHECK_LAUNCH("psg_add_layernorm_res32");
  return PSG_OK;
}

static int add_layernorm_launch(const char* who, psg_ctx* ctx, const void* x, const void* residual, int res_period,
                                const int32_t* res_index, const float* bias, const float* gamma, const float* beta, float eps, int64_t rows,
                                int hidden, void* out, int dtype, void* stream) {
  PSG_REQUIRE(ctx && x && gamma && beta && out, PSG_ERR_INVALID, "%s: NULL argument", who);
  PSG_REQUIRE(hidden == 768, PSG_ERR_UNSUPPORTED, "%s: hidden=%d (kernel is built for 768)", who, hidden);
  if (rows == 0) return PSG_OK;
  if ((dtype == PSG_BF16 || dtype == PSG_F16) && ctx->opt.ln_half_wave) {     // 8 rows per 256-thread workgroup
    dim3 grid16((unsigned)((rows + 7) / 8));
    PSG_DISPATCH_E16(dtype, who,
                     (add_layernorm16_kernel<E><<<grid16, 256, 0, (hipStream_t)stream>>>(
                         (const uint16_t*)x, (const uint16_t*)residual, bias, gamma, beta, eps, rows, (uint16_t*)out,
                         res_period, res_index)));
    PSG_CHECK_LAUNCH(who);
    return PSG_OK;
  }
  dim3 grid((unsigned)((rows + 3) / 4));
  PSG_DISPATCH_DTYPE(dtype, who,
                     (add_layernorm_kernel<T, 3><<<grid, 256, 0, (hipStream_t)stream>>>(
                         (const T*)x, (const T*)residual, bias, gamma, beta, eps, rows, hidden, (T*)out, res_period,
                         res_index)));
  PSG_CHECK_LAUNCH(who);
  return PSG_OK;
}

extern "C" int psg_add_layernorm(psg_ctx* ctx, const void* x, const void* residual, const float* bias,
                                 const float* gamma, const float* beta, float eps, int64_t rows, int hidden, void* out,
                                 int dtype, void* stream) {
  return add_layernorm_launch("psg_add_layernorm", ctx, x, residual, 0, nullptr, bias, gamma, beta, eps, rows, hidden, out,
                              dtype, stream);
}

// residual row of output row r is residual_table[r % table_rows]: the layer-0 query rows of every pair share ONE
// [33][768] block of embeddings, which therefore never has to be written out per pair
extern "C" int psg_add_layernorm_periodic(psg_ctx* ctx, const void* x, const void* residual_table, int table_rows,
                                          const float* bias, const float* gamma, const float* beta, float eps,
                                          int64_t rows, int hidden, void* out, int dtype, void* stream) {
  PSG_REQUIRE(residual_table && table_rows > 0, PSG_ERR_INVALID, "psg_add_layernorm_periodic: table_rows=%d", table_rows);
  return add_layernorm_launch("psg_add_layernorm_periodic", ctx, x, residual_table, table_rows, nullptr, bias, gamma, beta,
                              eps, rows, hidden, out, dtype, stream);
}

// residual row of output row r is residual_table[block_index[r / group] * group + r % group]: the rows come in groups
// (the 33 query rows of a pair) and several groups share one block of the table (the pair's prompt)
extern "C" int psg_add_layernorm_indexed(psg_ctx* ctx, const void* x, const void* residual_table, const int32_t* block_index,
                                         int group, const float* bias, const float* gamma, const float* beta, float eps,
                                         int64_t rows, int hidden, void* out, int dtype, void* stream) {
  PSG_REQUIRE(residual_table && block_index && group > 0 && rows % group == 0, PSG_ERR_INVALID,
              "psg_add_layernorm_indexed: group=%d rows=%lld", group, (long long)rows);
  return add_layernorm_launch("psg_add_layernorm_indexed", ctx, x, residual_table, group, block_index, bias, gamma, beta,
                              eps, rows, hidden, out, dtype, stream);
}

// ---- bias + GELU(erf) -------------------------------------------------------------------------
template <typename T>
__global__ void bias_gelu_kernel(const T* __restrict__ x, const float* __restrict__ bias, int64_t n4, int cols,
                                 T* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float v[4];
    Act<T>::ld4(x, i * 4, v);
    if (bias) {
      const int col = (int)((i * 4) % cols);
      const float4 b = *reinterpret_cast<const float4*>(bias + col);
      v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
    Act<T>::st4(out, i * 4, v);
  }
}

// bf16 activations, many rows (the Q-Former FFN: 82.5 k x 3072): the generic kernel above is VALU-bound there
// (64-bit index modulo per 4 elements, libm erff).  One row per blockIdx.y, 8 columns per thread (16-byte
// accesses), erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, three orders below bf16 resolution; the
// fp32 verification mode keeps erff).
__device__ __forceinline__ float gelu_erf_as(float v) {
  // the reference computes 0.5 v (1 + erf(v / sqrt 2)) in fp32; erf here is Abramowitz-Stegun 7.1.26,
  // |error| <= 1.5e-7 (about two fp32 ulps of a value near 1).  It differs from erff only where 1 + erf cancels
  // (v < -5, |gelu| < 1e-5), where the reference's own result is rounding noise of the same size.
  const float x = v * 0.70710678118654752440f;
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  const float erf_abs = fmaf(-poly * t, e, 1.0f);
  return 0.5f * v * (1.0f + copysignf(erf_abs, x));
}

template <typename E>
__global__ void __launch_bounds__(256) bias_gelu_rows_bf16_kernel(const uint16_t* __restrict__ x,
                                                                   const float* __restrict__ bias, int64_t rows,
                                                                   int cols, uint16_t* __restrict__ out) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (c >= cols) return;
  float b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (bias) {
    *reinterpret_cast<float4*>(b) = *reinterpret_cast<const float4*>(bias + c);
    *reinterpret_cast<float4*>(b + 4) = *reinterpret_cast<const float4*>(bias + c + 4);
  }
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    const uint4 xv = *reinterpret_cast<const uint4*>(x + r * cols + c);
    const uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w};
    uint32_t ow[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v0 = E::to_f32((uint16_t)(xw[e] & 0xffffu)) + b[2 * e];
      const float v1 = E::to_f32((uint16_t)(xw[e] >> 16)) + b[2 * e + 1];
      ow[e] = (uint32_t)E::from_f32(gelu_erf_as(v0)) | ((uint32_t)E::from_f32(gelu_erf_as(v1)) << 16);
    }
    *reinterpret_cast<uint4*>(out + r * cols + c) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
  }
}

extern "C" int psg_bias_gelu(psg_ctx* ctx, const void* x, const float* bias, int64_t rows, int cols, void* out,
                             int dtype, void* stream) {
  PSG_REQUIRE(ctx && x && out, PSG_ERR_INVALID, "psg_bias_gelu: NULL argument");
  PSG_REQUIRE(cols > 0 && cols % 4 == 0, PSG_ERR_INVALID, "psg_bias_gelu: cols=%d must be a multiple of 4", cols);
  if (rows == 0) return PSG_OK;
  if ((dtype == PSG_BF16 || dtype == PSG_F16) && rows >= 256 && cols % 8 == 0) {
    const dim3 grid((unsigned)((cols / 8 + 255) / 256), (unsigned)(rows < 16384 ? rows : 16384));
    PSG_DISPATCH_E16(dtype, "psg_bias_gelu",
                     (bias_gelu_rows_bf16_kernel<E><<<grid, 256, 0, (hipStream_t)stream>>>(
                         (const uint16_t*)x, bias, rows, cols, (uint16_t*)out)));
    PSG_CHECK_LAUNCH("psg_bias_gelu");
    return PSG_OK;
  }
  int64_t n4 = rows * cols / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  PSG_DISPATCH_DTYPE(dtype, "psg_bias_gelu",
                     (bias_gelu_kernel<T><<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>((const T*)x, bias, n4,
                                                                                           cols, (T*)out)));
  PSG_CHECK_LAUNCH("psg_bias_gelu");
  return PSG_OK;
}

// ---- K8 existence head ------------------------------------------------------------------------
template <typename T>
__global__ void exist_head_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                  int P, int nq, int hidden, float* __restrict__ logit, float* __restrict__ prob) {
  const int p = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  if (p >= P) return;
  const int64_t base = (int64_t)p * nq * hidden;  // row 0 of the pair = rel_cls_query row (V4:206)
  float acc = 0.f;
  for (int c = lane * 4; c < hidden; c += 256) {
    float v[4];
    Act<T>::ld4(x, base + c, v);
    const float4 ww = *reinterpret_cast<const float4*>(w + c);
    acc += v[0] * ww.x + v[1] * ww.y + v[2] * ww.z + v[3] * ww.w;
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    const float l = acc + b[0];
    logit[p] = l;
    if (prob) prob[p] = 1.0f / (1.0f + expf(-l));
  }
}

extern "C" int psg_exist_head(psg_ctx* ctx, const void* x, const float* w, const float* b, int P, int nq, int hidden,
                              float* logit, float* prob, int dtype, void* stream) {
  PSG_REQUIRE(ctx && x && w && b && logit, PSG_ERR_INVALID, "psg_exist_head: NULL argument");
  PSG_REQUIRE(hidden % 4 == 0 && P >= 0 && nq > 0, PSG_ERR_INVALID, "psg_exist_head: P=%d nq=%d hidden=%d", P, nq,
              hidden);
  if (P == 0) return PSG_OK;
  PSG_DISPATCH_DTYPE(dtype, "psg_exist_head",
                     (exist_head_kernel<T><<<(P + 3) / 4, 256, 0, (hipStream_t)stream>>>((const T*)x, w, b, P, nq,
                                                                                       hidden, logit, prob)));
  PSG_CHECK_LAUNCH("psg_exist_head");
  return PSG_OK;
}

// ---- K12 RMSNorm (+ residual add) -------------------------------------------------------------
// One 256-thread workgroup per row (decode has only K ~ 20 rows of 4096: a single wave walking a
// row serialises ~16 dependent HBM round trips).  Thread t owns the 4-element chunks t, t+256, ...;
// every load is issued before the first store; the sum of squares is reduced wave -> LDS -> block.
// R = storage type of the residual stream: T (HF: a model cast to 16 bits keeps x = residual + attn in 16 bits) or float
// (mixed mode: 16-bit GEMM operands, fp32 residual stream - the sum of 2 x layers updates is never rounded to 16 bits).
template <typename T, int NCH, typename R = T>
__global__ void __launch_bounds__(1024) rmsnorm_kernel(R* __restrict__ resid, const void* __restrict__ delta,
                                                      int dsplits, int64_t dslice, const float* __restrict__ w,
                                                      float eps, int hidden, T* __restrict__ out, int wt = 0) {
  __shared__ float s_part[16];
  const int64_t row = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nthr = blockDim.x;
  float v[NCH][4], d[NCH][4];
  float4 g[NCH];
  typename Act<R>::raw4 vr[NCH];                              // converted after the partials are requested: no early wait
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * nthr + tid) * 4;
    if (col < hidden) {
      vr[c] = Act<R>::ldr4(resid, row * hidden + col);
      g[c] = *reinterpret_cast<const float4*>(w + col);
    }
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * nthr + tid) * 4;
    if (col < hidden && delta) ld4_in<T>(delta, dsplits, dslice, row * hidden + col, d[c]);
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * nthr + tid) * 4;
    if (col < hidden) Act<R>::cv4(vr[c], v[c]);
  }
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * nthr + tid) * 4;
    if (col < hidden) {
      if (delta) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[c][e] += d[c][e];
          // the residual stream is rounded to its storage type (16 bits as HF does, x = residual + attn; or fp32)
          v[c][e] = Act<R>::rnd(v[c][e]);
        }
      }
      ss = psg_sumsq4(v[c], ss);                              // pinned form: psg_decode_layer.hip reproduces it
    }
  }
  ss = wave_sum(ss);
  if (lane == 0) s_part[wid] = ss;
  __syncthreads();
  ss = 0.f;
  for (int i = 0; i < (nthr >> 6); ++i) ss += s_part[i];
  const float inv = 1.0f / sqrtf(ss / (float)hidden + eps);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * nthr + tid) * 4;
    if (col < hidden) {
      float o[4] = {g[c].x * (v[c][0] * inv), g[c].y * (v[c][1] * inv), g[c].z * (v[c][2] * inv),
                    g[c].w * (v[c][3] * inv)};
      if constexpr (std::is_same<T, float>::value && std::is_same<R, float>::value) {
        if (wt) {                                             // option wt_stores (fp32 decode chain)
          if (delta) psg_st4_wt(reinterpret_cast<float*>(resid) + row * hidden + col, v[c][0], v[c][1], v[c][2], v[c][3]);
          psg_st4_wt(reinterpret_cast<float*>(out) + row * hidden + col, o[0], o[1], o[2], o[3]);
          continue;
        }
      }
      if (delta) Act<R>::st4(resid, row * hidden + col, v[c]);
      Act<T>::st4(out, row * hidden + col, o);
    }
  }
}

// Prompt pass (hundreds of 16-bit rows): one WAVE per row, 16-byte accesses (a lane owns NCH x 8 features), the sum of
// squares by DPP - no LDS, no workgroup barrier.  Per-element arithmetic as in rmsnorm_kernel.
// R32: the residual stream is fp32 (two 16-byte accesses per 8 features), delta and out stay 16-bit.
template <typename E, int NCH, bool R32 = false>
__global__ void __launch_bounds__(256) rmsnorm_rows16_kernel(void* __restrict__ resid_, const uint16_t* __restrict__ delta,
                                                             const float* __restrict__ w, float eps, int64_t rows, int hidden,
                                                             uint16_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= rows) return;                                        // whole wave
  uint16_t* resid = reinterpret_cast<uint16_t*>(resid_);
  float* resid32 = reinterpret_cast<float*>(resid_);
  uint4 rw[NCH], dw[NCH];
  float4 rf[NCH][2];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * 8;
    if (R32) {
      rf[c][0] = *reinterpret_cast<const float4*>(resid32 + row * hidden + col);
      rf[c][1] = *reinterpret_cast<const float4*>(resid32 + row * hidden + col + 4);
    } else {
      rw[c] = *reinterpret_cast<const uint4*>(resid + row * hidden + col);
    }
    if (delta) dw[c] = *reinterpret_cast<const uint4*>(delta + row * hidden + col);
  }
  auto unpack = [](const uint4& q, float (&f)[8]) {
    const uint32_t u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = E::to_f32((uint16_t)(u[i] & 0xffffu));
      f[2 * i + 1] = E::to_f32((uint16_t)(u[i] >> 16));
    }
  };
  float v[NCH][8];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (R32) {
      v[c][0] = rf[c][0].x; v[c][1] = rf[c][0].y; v[c][2] = rf[c][0].z; v[c][3] = rf[c][0].w;
      v[c][4] = rf[c][1].x; v[c][5] = rf[c][1].y; v[c][6] = rf[c][1].z; v[c][7] = rf[c][1].w;
    } else {
      unpack(rw[c], v[c]);
    }
    if (delta) {
      float d[8];
      unpack(dw[c], d);
#pragma unroll
      for (int e = 0; e < 8; ++e)                                  // the residual stream is 16-bit (HF) or fp32 (mixed mode)
        v[c][e] = R32 ? v[c][e] + d[e] : E::to_f32(E::from_f32(v[c][e] + d[e]));
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += v[c][e] * v[c][e];
  }
  ss = wave_sum(ss);
  const float inv = 1.0f / sqrtf(ss / (float)hidden + eps);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * 8;
    if (delta) {
      if (R32) {
        *reinterpret_cast<float4*>(resid32 + row * hidden + col) = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
        *reinterpret_cast<float4*>(resid32 + row * hidden + col + 4) = make_float4(v[c][4], v[c][5], v[c][6], v[c][7]);
      } else {
        *reinterpret_cast<uint4*>(resid + row * hidden + col) =
            make_uint4(E::pack(v[c][0], v[c][1]), E::pack(v[c][2], v[c][3]), E::pack(v[c][4], v[c][5]), E::pack(v[c][6], v[c][7]));
      }
    }
    const float4 g0 = *reinterpret_cast<const float4*>(w + col), g1 = *reinterpret_cast<const float4*>(w + col + 4);
    const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = g[e] * (v[c][e] * inv);
    *reinterpret_cast<uint4*>(out + row * hidden + col) =
        make_uint4(E::pack(o[0], o[1]), E::pack(o[2], o[3]), E::pack(o[4], o[5]), E::pack(o[6], o[7]));
  }
}

extern "C" int psg_rmsnorm(psg_ctx* ctx, void* resid, const void* delta, int delta_splits, const float* w, float eps,
                           int64_t rows, int hidden, void* out, int dtype, int resid_dtype, void* stream) {
  PSG_REQUIRE(ctx && resid && w && out, PSG_ERR_INVALID, "psg_rmsnorm: NULL argument");
  PSG_REQUIRE(resid_dtype == dtype || resid_dtype == PSG_F32, PSG_ERR_INVALID,
              "psg_rmsnorm: resid_dtype=%d must be the activation dtype (%d) or PSG_F32", resid_dtype, dtype);
  const bool r32 = resid_dtype == PSG_F32 && dtype != PSG_F32;
  PSG_REQUIRE(delta_splits >= 0 && (delta || delta_splits == 0), PSG_ERR_INVALID, "psg_rmsnorm: delta_splits=%d",
              delta_splits);
  PSG_REQUIRE(hidden % 4 == 0 && hidden > 0 && hidden <= 8192, PSG_ERR_UNSUPPORTED,
              "psg_rmsnorm: hidden=%d must be a multiple of 4 and <= 8192", hidden);
  if (rows == 0) return PSG_OK;
  PSG_REQUIRE(delta_splits <= PSG_MAX_SPLITS, PSG_ERR_UNSUPPORTED, "psg_rmsnorm: delta_splits=%d > %d", delta_splits,
              PSG_MAX_SPLITS);
  dim3 grid((unsigned)rows);
  hipStream_t st = (hipStream_t)stream;
  if (rows > 64 && delta_splits == 0 && (dtype == PSG_BF16 || dtype == PSG_F16) && (hidden == 4096 || hidden == 1024 ||
      hidden == 512) && ctx->opt.ln_half_wave) {                  // prompt pass: a wave per row, 16-byte accesses
    const unsigned blocks = (unsigned)((rows + 3) / 4);
#define RNR(N)                                                                                                       \
  if (r32) {                                                                                                         \
    PSG_DISPATCH_E16(dtype, "psg_rmsnorm",                                                                           \
                     (rmsnorm_rows16_kernel<E, N, true><<<blocks, 256, 0, st>>>(resid, (const uint16_t*)delta, w, eps, \
                                                                               rows, hidden, (uint16_t*)out)));      \
  } else                                                                                                             \
    PSG_DISPATCH_E16(dtype, "psg_rmsnorm",                                                                           \
                     (rmsnorm_rows16_kernel<E, N><<<blocks, 256, 0, st>>>(resid, (const uint16_t*)delta, w, eps,      \
                                                                         rows, hidden, (uint16_t*)out)))
    if (hidden == 4096) RNR(8);
    else if (hidden == 1024) RNR(2);
    else RNR(1);
#undef RNR
    PSG_CHECK_LAUNCH("psg_rmsnorm");
    return PSG_OK;
  }
  // decode-sized launches (a handful of rows) are latency bound: spread each row over 16 waves
  const int wt = ((ctx->opt.wt_stores & 1) && rows <= 64) ? 1 : 0;
  const int nthr = (rows <= 64 && hidden >= 4096) ? 1024 : 256;
  const int nch = (hidden + 4 * nthr - 1) / (4 * nthr);
#define RN(N)                                                                                                        \
  if (r32) {                                                                                                         \
    PSG_DISPATCH_DTYPE(dtype, "psg_rmsnorm",                                                                         \
                       (rmsnorm_kernel<T, N, float><<<grid, nthr, 0, st>>>((float*)resid, delta, delta_splits,       \
                                                                          rows * hidden, w, eps, hidden, (T*)out))); \
  } else                                                                                                             \
    PSG_DISPATCH_DTYPE(dtype, "psg_rmsnorm",                                                                         \
                       (rmsnorm_kernel<T, N><<<grid, nthr, 0, st>>>((T*)resid, delta, delta_splits, rows * hidden, w, \
                                                                   eps, hidden, (T*)out, wt)))
  switch (nch) {
    case 1: RN(1); break;
    case 2: RN(2); break;
    case 3: RN(3); break;
    case 4: RN(4); break;
    case 5: RN(5); break;
    case 6: RN(6); break;
    case 7: RN(7); break;
    default: RN(8); break;
  }
#undef RN
  PSG_CHECK_LAUNCH("psg_rmsnorm");
  return PSG_OK;
}

// ---- K13 rotary (half-split) + KV-cache write -------------------------------------------------
// One wave per (row, head); head_dim = 128: lane l holds dims l and l + 64 (the rotate_half pair).
template <typename T>
__global__ void rope_kvwrite_kernel(const void* __restrict__ qkv, int qs, const int32_t* __restrict__ tok_pair,
                                    const int32_t* __restrict__ tok_pos, const int32_t* __restrict__ rope_pos,
                                    const float* __restrict__ cos_tab,
                                    const float* __restrict__ sin_tab, int64_t rows, int heads, int ctx,
                                    T* __restrict__ q_out, T* __restrict__ kc, T* __restrict__ vc) {
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= rows * heads) return;
  const int64_t row = wave / heads;
  const int h = (int)(wave % heads);
  const int pos = tok_pos[row];
  if (pos < 0) return;  // padding row
  const int hidden = heads * 128;
  const int64_t base = row * 3 * hidden + h * 128;
  // cos/sin(pos * inv_freq) from the caller's table: a precise cosf/sinf per wave cost ~9 us of
  // large-argument range reduction, more than the rest of the kernel
  // rope_pos (optional): rotary position when it differs from the cache slot (training forward: HF numbers the
  // positions over the PADDED sequence, V4:327-330, while the cache holds the compacted tokens)
  const int rp = rope_pos ? rope_pos[row] : pos;
  const float cs = cos_tab[rp * 64 + lane], sn = sin_tab[rp * 64 + lane];
  const int64_t sl = rows * 3 * hidden;  // split-K slice stride
  const float q1 = ld1_in<T>(qkv, qs, sl, base + lane), q2 = ld1_in<T>(qkv, qs, sl, base + lane + 64);
  const float k1 = ld1_in<T>(qkv, qs, sl, base + hidden + lane), k2 = ld1_in<T>(qkv, qs, sl, base + hidden + lane + 64);
  const float v1 = ld1_in<T>(qkv, qs, sl, base + 2 * hidden + lane);
  const float v2 = ld1_in<T>(qkv, qs, sl, base + 2 * hidden + lane + 64);
  // q*cos + rotate_half(q)*sin, rotate_half(x) = cat(-x2, x1)   (HF-LL:130-160)
  float qa, qb, ka, kb;
  psg_rope_pair(q1, q2, cs, sn, qa, qb);                    // pinned form (psg_decode_math.h), shared with the decode step
  psg_rope_pair(k1, k2, cs, sn, ka, kb);
  Act<T>::st(q_out, row * hidden + h * 128 + lane, qa);
  Act<T>::st(q_out, row * hidden + h * 128 + lane + 64, qb);
  const int64_t cbase = (((int64_t)tok_pair[row] * heads + h) * ctx + pos) * 128;
  Act<T>::st(kc, cbase + lane, ka);
  Act<T>::st(kc, cbase + lane + 64, kb);
  Act<T>::st(vc, cbase + lane, v1);
  Act<T>::st(vc, cbase + lane + 64, v2);
}

extern "C" int psg_rope_kvwrite(psg_ctx* ctx_, const void* qkv, int qkv_splits, const int32_t* tok_pair,
                                const int32_t* tok_pos, const int32_t* rope_pos, const float* rope_cos,
                                const float* rope_sin, int64_t rows, int heads, int head_dim, int ctx, void* q_out,
                                void* k_cache, void* v_cache, int dtype, void* stream) {
  PSG_REQUIRE(ctx_ && qkv && tok_pair && tok_pos && rope_cos && rope_sin && q_out && k_cache && v_cache, PSG_ERR_INVALID,
              "psg_rope_kvwrite: NULL argument");
  PSG_REQUIRE(head_dim == 128, PSG_ERR_UNSUPPORTED, "psg_rope_kvwrite: head_dim=%d (kernel is built for 128)",
              head_dim);
  if (rows == 0) return PSG_OK;
  int64_t waves = rows * heads;
  PSG_DISPATCH_DTYPE(dtype, "psg_rope_kvwrite",
                     (rope_kvwrite_kernel<T><<<(unsigned)((waves + 3) / 4), 256, 0, (hipStream_t)stream>>>(
                         qkv, qkv_splits, tok_pair, tok_pos, rope_pos, rope_cos, rope_sin, rows, heads, ctx, (T*)q_out,
                         (T*)k_cache, (T*)v_cache)));
  PSG_CHECK_LAUNCH("psg_rope_kvwrite");
  return PSG_OK;
}

// ---- SwiGLU gate ------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) silu_mul_kernel(const void* __restrict__ gu, int S, int64_t rows, int inter, T* __restrict__ out,
                                                       int wt = 0) {
  const int64_t n4 = rows * inter / 4;
  const int i4 = inter / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / i4;
    const int c = (int)(i % i4) * 4;
    float g[4], u[4], o[4];
    if (S > 0) {                                              // gate and up partials of all slices in one pass
      const int64_t idx[2] = {r * 2 * inter + c, r * 2 * inter + inter + c};
      float4 gu4[2];
      ldn_splits<float4, 2>(gu, S, rows * 2 * inter, idx, gu4);
      g[0] = Act<T>::rnd(gu4[0].x); g[1] = Act<T>::rnd(gu4[0].y); g[2] = Act<T>::rnd(gu4[0].z); g[3] = Act<T>::rnd(gu4[0].w);
      u[0] = Act<T>::rnd(gu4[1].x); u[1] = Act<T>::rnd(gu4[1].y); u[2] = Act<T>::rnd(gu4[1].z); u[3] = Act<T>::rnd(gu4[1].w);
    } else {
      Act<T>::ld4(reinterpret_cast<const T*>(gu), r * 2 * inter + c, g);
      Act<T>::ld4(reinterpret_cast<const T*>(gu), r * 2 * inter + inter + c, u);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float s = g[e] / (1.0f + expf(-g[e]));
      s = Act<T>::rnd(s);  // HF rounds act_fn(gate) before the product
      o[e] = s * u[e];
    }
    if constexpr (std::is_same<T, float>::value) {
      if (wt) {
        psg_st4_wt(reinterpret_cast<float*>(out) + r * inter + c, o[0], o[1], o[2], o[3]);
        continue;
      }
    }
    Act<T>::st4(out, r * inter + c, o);
  }
}

// prompt pass (hundreds of dense bf16 rows): 8 columns per thread (16-byte loads), one row per blockIdx.y,
// no index division; the decode step (<= 64 rows, split-K partials) keeps the kernel above
template <typename E>
__global__ void __launch_bounds__(256) silu_mul_rows_bf16_kernel(const uint16_t* __restrict__ gu, int64_t rows,
                                                                  int inter, uint16_t* __restrict__ out) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (c >= inter) return;
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    const uint4 gv = *reinterpret_cast<const uint4*>(gu + r * 2 * inter + c);
    const uint4 uv = *reinterpret_cast<const uint4*>(gu + r * 2 * inter + inter + c);
    const uint32_t gw[4] = {gv.x, gv.y, gv.z, gv.w}, uw[4] = {uv.x, uv.y, uv.z, uv.w};
    uint32_t ow[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float g0 = E::to_f32((uint16_t)(gw[e] & 0xffffu)), g1 = E::to_f32((uint16_t)(gw[e] >> 16));
      const float u0 = E::to_f32((uint16_t)(uw[e] & 0xffffu)), u1 = E::to_f32((uint16_t)(uw[e] >> 16));
      // same arithmetic as silu_mul_kernel<T>: silu in fp32, rounded to the activation type (HF rounds act_fn(gate)), product
      const float s0 = E::to_f32(E::from_f32(g0 / (1.0f + expf(-g0))));
      const float s1 = E::to_f32(E::from_f32(g1 / (1.0f + expf(-g1))));
      ow[e] = (uint32_t)E::from_f32(s0 * u0) | ((uint32_t)E::from_f32(s1 * u1) << 16);
    }
    *reinterpret_cast<uint4*>(out + r * inter + c) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
  }
}

extern "C" int psg_silu_mul(psg_ctx* ctx, const void* gate_up, int splits, int64_t rows, int inter, void* out,
                            int dtype, void* stream) {
  PSG_REQUIRE(ctx && gate_up && out, PSG_ERR_INVALID, "psg_silu_mul: NULL argument");
  PSG_REQUIRE(inter > 0 && inter % 4 == 0, PSG_ERR_INVALID, "psg_silu_mul: inter=%d must be a multiple of 4", inter);
  if (rows == 0) return PSG_OK;
  if ((dtype == PSG_BF16 || dtype == PSG_F16) && splits == 0 && rows > 64 && inter % 8 == 0) {
    const dim3 grid((unsigned)((inter / 8 + 255) / 256), (unsigned)(rows < 32768 ? rows : 32768));
    PSG_DISPATCH_E16(dtype, "psg_silu_mul",
                     (silu_mul_rows_bf16_kernel<E><<<grid, 256, 0, (hipStream_t)stream>>>(
                         (const uint16_t*)gate_up, rows, inter, (uint16_t*)out)));
    PSG_CHECK_LAUNCH("psg_silu_mul");
    return PSG_OK;
  }
  int64_t blocks = (rows * inter / 4 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  PSG_DISPATCH_DTYPE(dtype, "psg_silu_mul",
                     (silu_mul_kernel<T><<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(
                         gate_up, splits, rows, inter, (T*)out, ((ctx->opt.wt_stores & 1) && rows <= 64) ? 1 : 0)));
  PSG_CHECK_LAUNCH("psg_silu_mul");
  return PSG_OK;
}

// ---- K16 greedy step: argmax over the vocabulary + per-pair bookkeeping -------------------------
// TE / TX: element types of the embedding table and of the next step's residual rows (x_out, may be nullptr): the row of
// the chosen token is copied there by the same workgroup - the embedding gather of the next decode step needs no launch.
template <typename T, typename TE, typename TX>
__global__ void __launch_bounds__(1024) greedy_step_kernel(const void* __restrict__ logits, int S, int vocab, int step,
                                                           int max_new, int eos, int suppress,
                                                           int32_t* __restrict__ tokens, int32_t* __restrict__ done,
                                                           int32_t* __restrict__ next_ids,
                                                           int32_t* __restrict__ tok_pos, const TE* __restrict__ embed,
                                                           int hidden, TX* __restrict__ x_out) {
  __shared__ float s_val[16];
  __shared__ int s_idx[16];
  __shared__ int s_tok;
  const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t slice = (int64_t)gridDim.x * vocab;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  // 4 vocabulary entries per thread and pass (16-byte loads per split-K slice): one workgroup per pair has to
  // pull 4 slices x 128 KB through a single CU, and scalar 4-byte loads made that 38 us of a decode step
  const int v4 = (vocab % 4 == 0 && ((int64_t)k * vocab) % 4 == 0) ? vocab / 4 : 0;
  int i4_begin = tid;
  if (S >= 1 && S <= 4 && v4 > 0) {
    // split-K logits of the lm_head (<= 4 slices): four positions per thread and round trip - every slice of all four
    // requested before the first sum (one position per trip made the 8 passes over the vocabulary 8 dependent trips:
    // 16 us).  Sums in slice order, rounded to the activation type: the same values as ld4_in.
    constexpr int NB = 4;
    const float* p = reinterpret_cast<const float*>(logits) + (int64_t)k * vocab;
    const int bd = (int)blockDim.x;
    for (int i0 = tid; i0 < v4; i0 += NB * bd) {
      float4 t[4][NB];
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx)
        if (sidx < S) {
#pragma unroll
          for (int u = 0; u < NB; ++u) {
            const int j4 = i0 + u * bd < v4 ? i0 + u * bd : v4 - 1;
            t[sidx][u] = *reinterpret_cast<const float4*>(p + (int64_t)sidx * slice + 4 * j4);
          }
        }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        float4 a = t[0][u];
#pragma unroll
        for (int sidx = 1; sidx < 4; ++sidx)
          if (sidx < S) { a.x += t[sidx][u].x; a.y += t[sidx][u].y; a.z += t[sidx][u].z; a.w += t[sidx][u].w; }
        const float v[4] = {Act<T>::rnd(a.x), Act<T>::rnd(a.y), Act<T>::rnd(a.z), Act<T>::rnd(a.w)};
        const int j4 = i0 + u * bd;
        if (j4 < v4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int i = 4 * j4 + e;
            const float x = i == suppress ? -INFINITY : v[e];
            if (x > best || (x == best && i < bi)) {
              best = x;
              bi = i;
            }
          }
        }
      }
    }
    i4_begin = v4;                                               // the generic loop below has nothing left
  }
  for (int i4 = i4_begin; i4 < v4; i4 += blockDim.x) {
    float v[4];
    ld4_in<T>(logits, S, slice, (int64_t)k * vocab + 4 * i4, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = 4 * i4 + e;
      const float x = i == suppress ? -INFINITY : v[e];
      if (x > best || (x == best && i < bi)) {  // first maximal index, like torch.argmax
        best = x;
        bi = i;
      }
    }
  }
  for (int i = 4 * v4 + tid; i < vocab; i += blockDim.x) {
    float v = ld1_in<T>(logits, S, slice, (int64_t)k * vocab + i);
    if (i == suppress) v = -INFINITY;
    if (v > best || (v == best && i < bi)) {
      best = v;
      bi = i;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(best, o, 64);
    int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  }
  if (lane == 0) {
    s_val[wid] = best;
    s_idx[wid] = bi;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
      if (s_val[w] > best || (s_val[w] == best && s_idx[w] < bi)) {
        best = s_val[w];
        bi = s_idx[w];
      }
    const int was_done = done[k];
    // every logit NaN (an fp16 overflow upstream): bi keeps its initial 0x7fffffff - emit token 0 rather than read
    // 2^31 * hidden elements past the embedding table
    if (bi < 0 || bi >= vocab) bi = 0;
    tokens[(int64_t)k * max_new + step] = was_done ? -1 : bi;
    if (!was_done && bi == eos) done[k] = 1;
    next_ids[k] = bi;
    tok_pos[k] += 1;
    s_tok = bi;
  }
  if (x_out) {                                                   // kernel-uniform
    __syncthreads();
    const TE* src = embed + (int64_t)s_tok * hidden;
    for (int c = tid * 4; c < hidden; c += (int)blockDim.x * 4) {
      float v[4];
      Act<TE>::ld4(src, c, v);
      Act<TX>::st4(x_out, (int64_t)k * hidden + c, v);
    }
  }
}

extern "C" int psg_greedy_step(psg_ctx* ctx, const void* logits, int splits, int K, int vocab, int step, int max_new, int eos,
                               int suppress_token, int32_t* tokens, int32_t* done, int32_t* next_ids, int32_t* tok_pos,
                               const void* embed, int embed_dtype, int hidden, void* x_out, int x_dtype, int dtype,
                               void* stream) {
  PSG_REQUIRE(ctx && logits && tokens && done && next_ids && tok_pos, PSG_ERR_INVALID,
              "psg_greedy_step: NULL argument");
  PSG_REQUIRE(K > 0 && vocab > 0 && step >= 0 && step < max_new, PSG_ERR_INVALID,
              "psg_greedy_step: K=%d vocab=%d step=%d max_new=%d", K, vocab, step, max_new);
  PSG_REQUIRE(!x_out || (embed && hidden > 0 && hidden % 4 == 0), PSG_ERR_INVALID,
              "psg_greedy_step: x_out needs the embedding table and hidden %% 4 == 0 (hidden=%d)", hidden);
  hipStream_t st = (hipStream_t)stream;
#define GS(TE_, TX_)                                                                                                  \
  PSG_DISPATCH_DTYPE(dtype, "psg_greedy_step",                                                                        \
                     (greedy_step_kernel<T, TE_, TX_><<<K, 1024, 0, st>>>(logits, splits, vocab, step, max_new, eos,   \
                                                                         suppress_token, tokens, done, next_ids,      \
                                                                         tok_pos, (const TE_*)embed, hidden,          \
                                                                         (TX_*)x_out)))
  if (!x_out || (embed_dtype == PSG_F32 && x_dtype == PSG_F32)) GS(float, float);
  else if (embed_dtype == PSG_BF16 && x_dtype == PSG_BF16) GS(bf16_t, bf16_t);
  else if (embed_dtype == PSG_BF16 && x_dtype == PSG_F32) GS(bf16_t, float);
  else if (embed_dtype == PSG_F16 && x_dtype == PSG_F16) GS(f16_t, f16_t);
  else if (embed_dtype == PSG_F16 && x_dtype == PSG_F32) GS(f16_t, float);
  else {
    psg_set_error("psg_greedy_step: embedding dtype %d -> residual dtype %d unsupported", embed_dtype, x_dtype);
    return PSG_ERR_UNSUPPORTED;
  }
#undef GS
  PSG_CHECK_LAUNCH("psg_greedy_step");
  return PSG_OK;
}
