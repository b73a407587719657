// fp32-grade products on the 16-bit matrix cores: the prompt pass of the reference-precision mode.
//
// The reference runs the LLM in fp32 (V4:99-100).  Its prompt pass - ~1000 rows x 6.5 G parameters, HF-LL:163-177,
// 243-281 - is 12.7 TFLOP per image; exact fp32 on gfx950 runs at the f32 VECTOR rate (157 TFLOP/s peak, there is no
// xf32 / TF32 class, MI355X_MICROARCH.md), i.e. >= 81 ms, measured 104 ms in the library SGEMM.  The 16-bit matrix
// cores are 16x faster, and an fp32 value is the sum of two fp16 values to 2^-22:
//
//     x 2^t = xh + xl (+ 2^-22),  w 2^s = wh + wl (+ 2^-22)          xh = fp16(x 2^t), xl = fp16(x 2^t - xh)
//     x . w = 2^-(s+t) (xh.wh + xh.wl + xl.wh)  (+ 3 * 2^-22 |x||w|)
//
// Every fp16 x fp16 product is exact in fp32 (11 + 11 mantissa bits) and the sums are fp32: what is lost against an
// fp32 GEMM is the xl.wl term and the two split residuals, ~7e-7 relative per product against fp32's 6e-8 rounding -
// 700x closer to fp32 than any single 16-bit operand format (2^-11 = 4.9e-4).  The three partial GEMMs are ONE fp16
// GEMM over a K axis three times as long:  [xh | xh | xl] . [wh | wl | wh]^T.
//
// psg_split_f16x3: one fp32 row -> its three fp16 K segments.  The row is scaled by a power of two (exact) so that its
// largest magnitude lands in [2^13, 2^14): xl then sits ~2^-11 below and stays in fp16's normal range for every
// element down to 2^-13 of the row's maximum (below that the absolute error is bounded by fp16's subnormal spacing,
// 2^-24 x the row scale).  inv_scale[row] = 2^-t undoes it after the GEMM (psg_scale_rows_cols), exactly.
#include "psg_common.h"

#define SP_ORDER_HHL 0   // activations: [xh | xh | xl]
#define SP_ORDER_HLH 1   // weights:     [wh | wl | wh]
#define SP_ORDER_I2 2    // either operand of psg_dense_gemm_split: per 32 k [hi(32) | lo(32)], 2K elements per row - every
                         // value ONCE (the three products are formed in the GEMM from one staging of each part)

__global__ void __launch_bounds__(256) split_f16x3_kernel(const float* __restrict__ x, int64_t row_stride, int K, int order,
                                                          uint16_t* __restrict__ out, float* __restrict__ inv_scale) {
  __shared__ float s_max[4];
  const int64_t row = blockIdx.x;
  const float* xr = x + row * row_stride;
  const int tid = threadIdx.x;
  float mx = 0.f;
  for (int c = tid * 4; c < K; c += 256 * 4) {
    const float4 v = *reinterpret_cast<const float4*>(xr + c);
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  mx = wave_max(mx);
  if ((tid & 63) == 0) s_max[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
  // scale = 2^(13 - floor(log2(mx))), clamped to fp32's normal range; a zero / non-finite row keeps scale 1
  int e = (int)((__float_as_uint(mx) >> 23) & 255u) - 127;
  if (!(mx > 0.f) || e == 128) e = 13;
  int se = 127 + 13 - e;
  se = se < 1 ? 1 : (se > 253 ? 253 : se);                                         // 2^-(se - 127) stays a normal number too
  const float scale = __uint_as_float((uint32_t)se << 23);
  if (tid == 0) inv_scale[row] = __uint_as_float((uint32_t)(254 - se) << 23);      // 2^-(se - 127)
  uint16_t* o0 = out + row * (order == SP_ORDER_I2 ? 2 : 3) * (int64_t)K;
  uint16_t* oh2 = o0 + (order == SP_ORDER_HHL ? K : 2 * (int64_t)K);               // second copy of the high part
  uint16_t* ol = o0 + (order == SP_ORDER_HHL ? 2 * (int64_t)K : K);                // low part
  for (int c = tid * 4; c < K; c += 256 * 4) {
    const float4 v = *reinterpret_cast<const float4*>(xr + c);
    const float s[4] = {v.x * scale, v.y * scale, v.z * scale, v.w * scale};
    ushort4 h, l;
    uint16_t* hp = &h.x;
    uint16_t* lp = &l.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint16_t hb = f32_to_f16(s[i]);                                          // RNE
      hp[i] = hb;
      lp[i] = f32_to_f16(s[i] - f16_to_f32(hb));                                     // the difference is exact in fp32
    }
    if (order == SP_ORDER_I2) {                                                      // block c / 32: [hi 32 | lo 32]
      uint16_t* ob = o0 + (c >> 5) * 64 + (c & 31);
      *reinterpret_cast<ushort4*>(ob) = h;
      *reinterpret_cast<ushort4*>(ob + 32) = l;
      continue;
    }
    *reinterpret_cast<ushort4*>(o0 + c) = h;
    *reinterpret_cast<ushort4*>(oh2 + c) = h;
    *reinterpret_cast<ushort4*>(ol + c) = l;
  }
}

extern "C" int psg_split_f16x3(psg_ctx* ctx, const float* x, int64_t rows, int K, int64_t row_stride, int order,
                               void* out, float* inv_scale, void* stream) {
  PSG_REQUIRE(ctx && x && out && inv_scale, PSG_ERR_INVALID, "psg_split_f16x3: NULL argument");
  PSG_REQUIRE(rows >= 0 && K > 0 && K % 4 == 0 && row_stride >= K && row_stride % 4 == 0 &&
                  (order == SP_ORDER_HHL || order == SP_ORDER_HLH || (order == SP_ORDER_I2 && K % 32 == 0)) &&
                  rows < (1ll << 31),
              PSG_ERR_INVALID, "psg_split_f16x3: rows=%lld K=%d stride=%lld order=%d", (long long)rows, K,
              (long long)row_stride, order);
  if (rows == 0) return PSG_OK;
  split_f16x3_kernel<<<(unsigned)rows, 256, 0, (hipStream_t)stream>>>(x, row_stride, K, order, (uint16_t*)out, inv_scale);
  PSG_CHECK_LAUNCH("psg_split_f16x3");
  return PSG_OK;
}

// psg_split_f16x2: fp32 rows -> two PLANES of fp16, out[0][row][K] = xh, out[1][row][K] = xl (same row scale as above):
// the operand of psg_split_gemm_w16, whose weight is an fp16 value and needs no low part
__global__ void __launch_bounds__(256) split_f16x2_kernel(const float* __restrict__ x, int64_t row_stride, int K, int64_t rows,
                                                          uint16_t* __restrict__ out, float* __restrict__ inv_scale) {
  __shared__ float s_max[4];
  const int64_t row = blockIdx.x;
  const float* xr = x + row * row_stride;
  const int tid = threadIdx.x;
  float mx = 0.f;
  for (int c = tid * 4; c < K; c += 256 * 4) {
    const float4 v = *reinterpret_cast<const float4*>(xr + c);
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  mx = wave_max(mx);
  if ((tid & 63) == 0) s_max[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
  int e = (int)((__float_as_uint(mx) >> 23) & 255u) - 127;
  if (!(mx > 0.f) || e == 128) e = 13;
  int se = 127 + 13 - e;
  se = se < 1 ? 1 : (se > 253 ? 253 : se);
  const float scale = __uint_as_float((uint32_t)se << 23);
  if (tid == 0) inv_scale[row] = __uint_as_float((uint32_t)(254 - se) << 23);
  uint16_t* oh = out + row * (int64_t)K;
  uint16_t* ol = out + (rows + row) * (int64_t)K;
  for (int c = tid * 4; c < K; c += 256 * 4) {
    const float4 v = *reinterpret_cast<const float4*>(xr + c);
    const float s[4] = {v.x * scale, v.y * scale, v.z * scale, v.w * scale};
    ushort4 h, l;
    uint16_t* hp = &h.x;
    uint16_t* lp = &l.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint16_t hb = f32_to_f16(s[i]);
      hp[i] = hb;
      lp[i] = f32_to_f16(s[i] - f16_to_f32(hb));
    }
    *reinterpret_cast<ushort4*>(oh + c) = h;
    *reinterpret_cast<ushort4*>(ol + c) = l;
  }
}

__device__ __forceinline__ float sp_row_scale(float mx, float* inv_scale_out);
__device__ __forceinline__ void sp_store2(uint16_t* oh, uint16_t* ol, int c, const float (&x)[4], float scale, int wt = 0);
// decode steps (<= 64 rows): 1024 threads per row, the row held in registers between the maximum and the stores - one
// read of x and one barrier instead of two passes (the launch is latency, not bytes)
template <int NCH>
__global__ void __launch_bounds__(1024) split_f16x2_small_kernel(const float* __restrict__ x, int64_t row_stride, int K,
                                                                 int64_t rows, uint16_t* __restrict__ out,
                                                                 float* __restrict__ inv_scale, int wt) {
  __shared__ float s_max[16];
  const int64_t row = blockIdx.x;
  const float* xr = x + row * row_stride;
  const int tid = threadIdx.x;
  float v[NCH][4];
  float mx = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 1024 + tid) * 4;
    if (col < K) {
      const float4 t = *reinterpret_cast<const float4*>(xr + col);
      v[c][0] = t.x; v[c][1] = t.y; v[c][2] = t.z; v[c][3] = t.w;
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(t.x), fabsf(t.y))), fmaxf(fabsf(t.z), fabsf(t.w)));
    }
  }
  mx = wave_max(mx);
  if ((tid & 63) == 0) s_max[tid >> 6] = mx;
  __syncthreads();
  mx = 0.f;
  for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s_max[i]);
  const float scale = sp_row_scale(mx, tid == 0 ? inv_scale + row : nullptr);     // the scale of split_f16x2_kernel
  uint16_t* oh = out + row * (int64_t)K;
  uint16_t* ol = out + (rows + row) * (int64_t)K;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 1024 + tid) * 4;
    if (col < K) sp_store2(oh, ol, col, v[c], scale, wt);
  }
}

extern "C" int psg_split_f16x2(psg_ctx* ctx, const float* x, int64_t rows, int K, int64_t row_stride, void* out,
                               float* inv_scale, void* stream) {
  PSG_REQUIRE(ctx && x && out && inv_scale, PSG_ERR_INVALID, "psg_split_f16x2: NULL argument");
  PSG_REQUIRE(rows >= 0 && K > 0 && K % 4 == 0 && row_stride >= K && row_stride % 4 == 0 && rows < (1ll << 31),
              PSG_ERR_INVALID, "psg_split_f16x2: rows=%lld K=%d stride=%lld", (long long)rows, K, (long long)row_stride);
  if (rows == 0) return PSG_OK;
  if (rows <= 64 && K <= 16384) {
    hipStream_t st = (hipStream_t)stream;
    if (K <= 4096) split_f16x2_small_kernel<1><<<(unsigned)rows, 1024, 0, st>>>(x, row_stride, K, rows, (uint16_t*)out, inv_scale, (ctx->opt.wt_stores >> 2) & 1);
    else if (K <= 12288) split_f16x2_small_kernel<3><<<(unsigned)rows, 1024, 0, st>>>(x, row_stride, K, rows, (uint16_t*)out, inv_scale, (ctx->opt.wt_stores >> 2) & 1);
    else split_f16x2_small_kernel<4><<<(unsigned)rows, 1024, 0, st>>>(x, row_stride, K, rows, (uint16_t*)out, inv_scale, (ctx->opt.wt_stores >> 2) & 1);
    PSG_CHECK_LAUNCH("psg_split_f16x2");
    return PSG_OK;
  }
  split_f16x2_kernel<<<(unsigned)rows, 256, 0, (hipStream_t)stream>>>(x, row_stride, K, rows, (uint16_t*)out, inv_scale);
  PSG_CHECK_LAUNCH("psg_split_f16x2");
  return PSG_OK;
}

// y[m][n] *= row_scale[m] * col_scale[n]  (powers of two: exact), in place
__global__ void __launch_bounds__(256) scale_rows_cols_kernel(float* __restrict__ y, int64_t rows, int N,
                                                              const float* __restrict__ row_scale,
                                                              const float* __restrict__ col_scale) {
  const int64_t n4 = N / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows * n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / n4;
    const int c = (int)(i - m * n4) * 4;
    const float rs = row_scale[m];
    const float4 cs = *reinterpret_cast<const float4*>(col_scale + c);
    float4 v = *reinterpret_cast<float4*>(y + m * N + c);
    v.x *= rs * cs.x; v.y *= rs * cs.y; v.z *= rs * cs.z; v.w *= rs * cs.w;
    *reinterpret_cast<float4*>(y + m * N + c) = v;
  }
}

extern "C" int psg_scale_rows_cols(psg_ctx* ctx, float* y, int64_t rows, int N, const float* row_scale,
                                   const float* col_scale, void* stream) {
  PSG_REQUIRE(ctx && y && row_scale && col_scale, PSG_ERR_INVALID, "psg_scale_rows_cols: NULL argument");
  PSG_REQUIRE(rows >= 0 && N > 0 && N % 4 == 0, PSG_ERR_INVALID, "psg_scale_rows_cols: rows=%lld N=%d", (long long)rows, N);
  if (rows == 0) return PSG_OK;
  int64_t blocks = (rows * (N / 4) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  scale_rows_cols_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(y, rows, N, row_scale, col_scale);
  PSG_CHECK_LAUNCH("psg_scale_rows_cols");
  return PSG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the split and the un-scaling live in the row kernels that produce / consume the operands (prompt pass of the
// fp32s mode, HF-LL:53-67, 130-177, 243-281).  A projection's raw fp16-GEMM result y is handed on WITH its two scale
// vectors; its reader applies y * (row_scale * col_scale) while loading (exactly what psg_scale_rows_cols stores), and the
// RMSNorm / SwiGLU kernels write their result straight as [hi | hi | lo] fp16 segments + the row's inverse scale (exactly
// what psg_split_f16x3 computes from their fp32 output): bit-identical to the separate kernels, 7 launches per layer less.
// ---------------------------------------------------------------------------------------------------------------------
#include "psg_decode_math.h"

__device__ __forceinline__ float sp_row_scale(float mx, float* inv_scale_out) {   // psg_split_f16x3's scale of a row maximum
  int e = (int)((__float_as_uint(mx) >> 23) & 255u) - 127;
  if (!(mx > 0.f) || e == 128) e = 13;
  int se = 127 + 13 - e;
  se = se < 1 ? 1 : (se > 253 ? 253 : se);
  if (inv_scale_out) *inv_scale_out = __uint_as_float((uint32_t)(254 - se) << 23);
  return __uint_as_float((uint32_t)se << 23);
}
__device__ __forceinline__ void sp_store2(uint16_t* oh, uint16_t* ol, int c, const float (&x)[4], float scale, int wt);   // two planes
__device__ __forceinline__ void sp_store3(uint16_t* o0, int K, int c, const float (&x)[4], float scale) {   // [hi | hi | lo]
  ushort4 h, l;
  uint16_t* hp = &h.x;
  uint16_t* lp = &l.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float s = x[i] * scale;
    const uint16_t hb = f32_to_f16(s);
    hp[i] = hb;
    lp[i] = f32_to_f16(s - f16_to_f32(hb));
  }
  *reinterpret_cast<ushort4*>(o0 + c) = h;
  *reinterpret_cast<ushort4*>(o0 + K + c) = h;
  *reinterpret_cast<ushort4*>(o0 + 2 * (int64_t)K + c) = l;
}

// resid += delta * (d_rs[row] * d_cs[col]);  x = RMSNorm(resid) * w  ->  out3 [rows][3 hidden] fp16, inv_scale [rows].
// One NTHR-thread workgroup per row in the arithmetic order of rmsnorm_kernel<float, NCH, float> launched with NTHR threads
// (psg_rmsnorm's rule: 1024 threads for <= 64 rows of >= 4096 columns, else 256 - the same sum-of-squares tree either way).
template <int NCH, int NTHR>
__global__ void __launch_bounds__(NTHR) rmsnorm_split_kernel(float* __restrict__ resid, const float* __restrict__ delta,
                                                            const float* __restrict__ d_rs, const float* __restrict__ d_cs,
                                                            const float* __restrict__ w, float eps, int hidden,
                                                            uint16_t* __restrict__ out3, float* __restrict__ inv_scale,
                                                            int dslices, int64_t dstride, int planes) {
  __shared__ float s_part[NTHR / 64], s_max[NTHR / 64];
  const int64_t row = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float v[NCH][4];
  float4 g[NCH];
  const float rs = delta ? d_rs[row] : 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * NTHR + tid) * 4;
    if (col < hidden) {
      const float4 r = *reinterpret_cast<const float4*>(resid + row * hidden + col);
      g[c] = *reinterpret_cast<const float4*>(w + col);
      v[c][0] = r.x; v[c][1] = r.y; v[c][2] = r.z; v[c][3] = r.w;
      if (delta) {
        float4 d = *reinterpret_cast<const float4*>(delta + row * hidden + col);
        for (int sl = 1; sl < dslices; ++sl) {                  // K segments of the product as separate slices: summed in order
          const float4 e = *reinterpret_cast<const float4*>(delta + sl * dstride + row * hidden + col);
          d.x += e.x; d.y += e.y; d.z += e.z; d.w += e.w;
        }
        const float4 cs = *reinterpret_cast<const float4*>(d_cs + col);
        v[c][0] += d.x * (rs * cs.x); v[c][1] += d.y * (rs * cs.y); v[c][2] += d.z * (rs * cs.z); v[c][3] += d.w * (rs * cs.w);
      }
    }
  }
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if ((c * NTHR + tid) * 4 < hidden) ss = psg_sumsq4(v[c], ss);
  ss = wave_sum(ss);
  if (lane == 0) s_part[wid] = ss;
  __syncthreads();
  ss = 0.f;
  for (int i = 0; i < NTHR / 64; ++i) ss += s_part[i];
  const float inv = 1.0f / sqrtf(ss / (float)hidden + eps);
  float x[NCH][4];
  float mx = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * NTHR + tid) * 4;
    if (col < hidden) {
      if (delta) *reinterpret_cast<float4*>(resid + row * hidden + col) = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
      x[c][0] = g[c].x * (v[c][0] * inv); x[c][1] = g[c].y * (v[c][1] * inv);
      x[c][2] = g[c].z * (v[c][2] * inv); x[c][3] = g[c].w * (v[c][3] * inv);
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(x[c][0]), fabsf(x[c][1]))), fmaxf(fabsf(x[c][2]), fabsf(x[c][3])));
    }
  }
  mx = wave_max(mx);
  if (lane == 0) s_max[wid] = mx;
  __syncthreads();
  mx = s_max[0];
#pragma unroll
  for (int i = 1; i < NTHR / 64; ++i) mx = fmaxf(mx, s_max[i]);
  const float scale = sp_row_scale(mx, tid == 0 ? inv_scale + row : nullptr);
  uint16_t* o0 = out3 + row * 3 * (int64_t)hidden;
  uint16_t* oh = out3 + row * (int64_t)hidden;                   // planes == 2: [2][rows][hidden] (dstride = rows x hidden)
  uint16_t* ol = oh + dstride;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * NTHR + tid) * 4;
    if (col < hidden) {
      if (planes == 2) sp_store2(oh, ol, col, x[c], scale);
      else sp_store3(o0, hidden, col, x[c], scale);
    }
  }
}

extern "C" int psg_rmsnorm_split(psg_ctx* ctx, float* resid, const float* delta, const float* delta_row_scale,
                                 const float* delta_col_scale, int delta_slices, const float* w, float eps, int64_t rows,
                                 int hidden, void* out3, float* inv_scale, int planes, void* stream) {
  PSG_REQUIRE(ctx && resid && w && out3 && inv_scale && (!delta || (delta_row_scale && delta_col_scale)), PSG_ERR_INVALID,
              "psg_rmsnorm_split: NULL argument");
  PSG_REQUIRE(hidden % 4 == 0 && hidden > 0 && hidden <= 8192 && rows >= 0 && rows < (1ll << 31), PSG_ERR_UNSUPPORTED,
              "psg_rmsnorm_split: hidden=%d rows=%lld", hidden, (long long)rows);
  if (rows == 0) return PSG_OK;
  const int nthr = (rows <= 64 && hidden >= 4096) ? 1024 : 256;       // psg_rmsnorm's choice: the same summation tree
  const int nch = (hidden + 4 * nthr - 1) / (4 * nthr);
  hipStream_t st = (hipStream_t)stream;
#define RNS(N, T)                                                                                                   \
  rmsnorm_split_kernel<N, T><<<(unsigned)rows, T, 0, st>>>(resid, delta, delta_row_scale, delta_col_scale, w, eps, hidden, \
                                                          (uint16_t*)out3, inv_scale, delta_slices, rows * (int64_t)hidden, planes)
  if (nthr == 1024) {
    if (nch <= 1) RNS(1, 1024);
    else RNS(2, 1024);
  } else switch (nch) {
    case 1: RNS(1, 256); break;
    case 2: RNS(2, 256); break;
    case 3: RNS(3, 256); break;
    case 4: RNS(4, 256); break;
    case 5: RNS(5, 256); break;
    case 6: RNS(6, 256); break;
    case 7: RNS(7, 256); break;
    default: RNS(8, 256); break;
  }
#undef RNS
  PSG_CHECK_LAUNCH("psg_rmsnorm_split");
  return PSG_OK;
}

// act = silu(g) * u of gate_up * (row_scale x col_scale)  ->  out3 [rows][3 inter] fp16, inv_scale [rows]; a workgroup per row
#define SP_MAXCH 16
__global__ void __launch_bounds__(256) silu_mul_split_kernel(const float* __restrict__ gu, const float* __restrict__ rsv,
                                                             const float* __restrict__ csv, int inter,
                                                             uint16_t* __restrict__ out3, float* __restrict__ inv_scale,
                                                             int slices, int64_t rows, int planes) {
  __shared__ float s_max[4];
  const int64_t row = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float rs = rsv[row];
  const float* gr = gu + row * 2 * (int64_t)inter;
  float o[SP_MAXCH][4];
  float mx = 0.f;
#pragma unroll
  for (int c = 0; c < SP_MAXCH; ++c) {
    const int col = (c * 256 + tid) * 4;
    if (col < inter) {
      float4 g4 = *reinterpret_cast<const float4*>(gr + col), u4 = *reinterpret_cast<const float4*>(gr + inter + col);
      for (int sl = 1; sl < slices; ++sl) {                           // the product as slices (two-plane operand): summed in order
        const float* gs = gr + sl * rows * 2 * (int64_t)inter;
        const float4 g5 = *reinterpret_cast<const float4*>(gs + col), u5 = *reinterpret_cast<const float4*>(gs + inter + col);
        g4.x += g5.x; g4.y += g5.y; g4.z += g5.z; g4.w += g5.w;
        u4.x += u5.x; u4.y += u5.y; u4.z += u5.z; u4.w += u5.w;
      }
      const float4 cg = *reinterpret_cast<const float4*>(csv + col), cu = *reinterpret_cast<const float4*>(csv + inter + col);
      const float g[4] = {g4.x * (rs * cg.x), g4.y * (rs * cg.y), g4.z * (rs * cg.z), g4.w * (rs * cg.w)};
      const float u[4] = {u4.x * (rs * cu.x), u4.y * (rs * cu.y), u4.z * (rs * cu.z), u4.w * (rs * cu.w)};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float s = g[e] / (1.0f + expf(-g[e]));                  // silu_mul_kernel<float>
        o[c][e] = s * u[e];
        mx = fmaxf(mx, fabsf(o[c][e]));
      }
    }
  }
  mx = wave_max(mx);
  if (lane == 0) s_max[wid] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
  const float scale = sp_row_scale(mx, tid == 0 ? inv_scale + row : nullptr);
  uint16_t* o0 = out3 + row * 3 * (int64_t)inter;
  uint16_t* oh = out3 + row * (int64_t)inter;
  uint16_t* ol = oh + rows * (int64_t)inter;
#pragma unroll
  for (int c = 0; c < SP_MAXCH; ++c) {
    const int col = (c * 256 + tid) * 4;
    if (col < inter) {
      if (planes == 2) sp_store2(oh, ol, col, o[c], scale);
      else sp_store3(o0, inter, col, o[c], scale);
    }
  }
}

extern "C" int psg_silu_mul_split(psg_ctx* ctx, const float* gate_up, const float* row_scale, const float* col_scale,
                                  int slices, int64_t rows, int inter, void* out3, float* inv_scale, int planes, void* stream) {
  PSG_REQUIRE(ctx && gate_up && row_scale && col_scale && out3 && inv_scale, PSG_ERR_INVALID,
              "psg_silu_mul_split: NULL argument");
  PSG_REQUIRE(inter > 0 && inter % 4 == 0 && inter <= SP_MAXCH * 1024 && rows >= 0 && rows < (1ll << 31),
              PSG_ERR_UNSUPPORTED, "psg_silu_mul_split: inter=%d (multiple of 4, <= %d)", inter, SP_MAXCH * 1024);
  if (rows == 0) return PSG_OK;
  silu_mul_split_kernel<<<(unsigned)rows, 256, 0, (hipStream_t)stream>>>(gate_up, row_scale, col_scale, inter,
                                                                         (uint16_t*)out3, inv_scale, slices, rows, planes);
  PSG_CHECK_LAUNCH("psg_silu_mul_split");
  return PSG_OK;
}

// psg_rope_kvwrite on a raw q|k|v projection result with its scale vectors (fp32; rotary position = cache slot)
__global__ void rope_kvwrite_scaled_kernel(const float* __restrict__ qkv, const float* __restrict__ rsv,
                                           const float* __restrict__ csv, const int32_t* __restrict__ tok_pair,
                                           const int32_t* __restrict__ tok_pos, const float* __restrict__ cos_tab,
                                           const float* __restrict__ sin_tab, int64_t rows, int heads, int ctx,
                                           float* __restrict__ q_out, float* __restrict__ kc, float* __restrict__ vc,
                                           int slices) {
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= rows * heads) return;
  const int64_t row = wave / heads;
  const int h = (int)(wave % heads);
  const int pos = tok_pos[row];
  if (pos < 0) return;
  const int hidden = heads * 128;
  const int c0 = h * 128 + lane;
  const int64_t base = row * 3 * hidden;
  const float rs = rsv[row];
  const float cs = cos_tab[pos * 64 + lane], sn = sin_tab[pos * 64 + lane];
  const int64_t sstride = rows * 3 * (int64_t)hidden;
  auto ld = [&](int col) {
    float v = qkv[base + col];
    for (int sl = 1; sl < slices; ++sl) v += qkv[sl * sstride + base + col];
    return v * (rs * csv[col]);
  };
  const float q1 = ld(c0), q2 = ld(c0 + 64), k1 = ld(hidden + c0), k2 = ld(hidden + c0 + 64);
  const float v1 = ld(2 * hidden + c0), v2 = ld(2 * hidden + c0 + 64);
  float qa, qb, ka, kb;
  psg_rope_pair(q1, q2, cs, sn, qa, qb);
  psg_rope_pair(k1, k2, cs, sn, ka, kb);
  q_out[row * hidden + c0] = qa;
  q_out[row * hidden + c0 + 64] = qb;
  const int64_t cbase = (((int64_t)tok_pair[row] * heads + h) * ctx + pos) * 128;
  kc[cbase + lane] = ka;
  kc[cbase + lane + 64] = kb;
  vc[cbase + lane] = v1;
  vc[cbase + lane + 64] = v2;
}

extern "C" int psg_rope_kvwrite_scaled(psg_ctx* ctx_, const float* qkv, const float* row_scale, const float* col_scale,
                                       const int32_t* tok_pair, const int32_t* tok_pos, const float* rope_cos,
                                       const float* rope_sin, int slices, int64_t rows, int heads, int head_dim, int ctx,
                                       float* q_out, float* k_cache, float* v_cache, void* stream) {
  PSG_REQUIRE(ctx_ && qkv && row_scale && col_scale && tok_pair && tok_pos && rope_cos && rope_sin && q_out && k_cache &&
                  v_cache, PSG_ERR_INVALID, "psg_rope_kvwrite_scaled: NULL argument");
  PSG_REQUIRE(head_dim == 128, PSG_ERR_UNSUPPORTED, "psg_rope_kvwrite_scaled: head_dim=%d (kernel is built for 128)", head_dim);
  if (rows == 0) return PSG_OK;
  const int64_t waves = rows * heads;
  rope_kvwrite_scaled_kernel<<<(unsigned)((waves + 3) / 4), 256, 0, (hipStream_t)stream>>>(
      qkv, row_scale, col_scale, tok_pair, tok_pos, rope_cos, rope_sin, rows, heads, ctx, q_out, k_cache, v_cache, slices);
  PSG_CHECK_LAUNCH("psg_rope_kvwrite_scaled");
  return PSG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Decode steps of the fp32s mode over fp16-valued weights (psg_split_gemm_w16): the row kernels in front of a projection
// write its operand straight as the two fp16 planes [2][rows][K] + the rows' inverse scales - the arithmetic of
// psg_rmsnorm (fp32 rows, fp32 split-K slices summed in slice order) followed by psg_split_f16x2, bit for bit, in one
// launch instead of two (HF-LL:53-67).  (The SwiGLU gate stays two launches: its row maximum spans 11008 columns, and one
// workgroup per row took 24 us against 5.2 + 4.9 for psg_silu_mul over all CUs + psg_split_f16x2.)
// ---------------------------------------------------------------------------------------------------------------------
// wt (option wt_stores, decode steps): the two 8-byte stores written through the L2 (agent-scope stores = sc1)
__device__ __forceinline__ void sp_store2(uint16_t* oh, uint16_t* ol, int c, const float (&x)[4], float scale, int wt) {
  ushort4 h, l;
  uint16_t* hp = &h.x;
  uint16_t* lp = &l.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float s = x[i] * scale;
    const uint16_t hb = f32_to_f16(s);
    hp[i] = hb;
    lp[i] = f32_to_f16(s - f16_to_f32(hb));
  }
  if (wt) {
    const uint64_t hw = (uint64_t)h.x | ((uint64_t)h.y << 16) | ((uint64_t)h.z << 32) | ((uint64_t)h.w << 48);
    const uint64_t lw = (uint64_t)l.x | ((uint64_t)l.y << 16) | ((uint64_t)l.z << 32) | ((uint64_t)l.w << 48);
    __hip_atomic_store(reinterpret_cast<uint64_t*>(oh + c), hw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<uint64_t*>(ol + c), lw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  *reinterpret_cast<ushort4*>(oh + c) = h;
  *reinterpret_cast<ushort4*>(ol + c) = l;
}

// rmsnorm_kernel<float, NCH, float> (psg_rowops.hip: nthr threads per row, the same loads, sums and roundings) + the split
template <int NCH>
__global__ void __launch_bounds__(1024) rmsnorm_split2_kernel(float* __restrict__ resid, const void* __restrict__ delta,
                                                              int dsplits, int64_t dslice, const float* __restrict__ w,
                                                              float eps, int hidden, int64_t rows,
                                                              uint16_t* __restrict__ out2, float* __restrict__ inv_scale,
                                                              int wt) {
  __shared__ float s_part[16], s_max[16];
  const int64_t row = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nthr = blockDim.x;
  float v[NCH][4], d[NCH][4];
  float4 g[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * nthr + tid) * 4;
    if (col < hidden) {
      const float4 r = *reinterpret_cast<const float4*>(resid + row * hidden + col);
      v[c][0] = r.x; v[c][1] = r.y; v[c][2] = r.z; v[c][3] = r.w;
      g[c] = *reinterpret_cast<const float4*>(w + col);
    }
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * nthr + tid) * 4;
    if (col < hidden && delta) ld4_in<float>(delta, dsplits, dslice, row * hidden + col, d[c]);
  }
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * nthr + tid) * 4;
    if (col < hidden) {
      if (delta) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[c][e] += d[c][e];
      }
      ss = psg_sumsq4(v[c], ss);
    }
  }
  ss = wave_sum(ss);
  if (lane == 0) s_part[wid] = ss;
  __syncthreads();
  ss = 0.f;
  for (int i = 0; i < (nthr >> 6); ++i) ss += s_part[i];
  const float inv = 1.0f / sqrtf(ss / (float)hidden + eps);
  float o[NCH][4];
  float mx = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * nthr + tid) * 4;
    if (col < hidden) {
      if (delta) {
        if (wt) psg_st4_wt(resid + row * hidden + col, v[c][0], v[c][1], v[c][2], v[c][3]);
        else *reinterpret_cast<float4*>(resid + row * hidden + col) = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
      }
      o[c][0] = g[c].x * (v[c][0] * inv); o[c][1] = g[c].y * (v[c][1] * inv);
      o[c][2] = g[c].z * (v[c][2] * inv); o[c][3] = g[c].w * (v[c][3] * inv);
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o[c][0]), fabsf(o[c][1]))), fmaxf(fabsf(o[c][2]), fabsf(o[c][3])));
    }
  }
  mx = wave_max(mx);
  if (lane == 0) s_max[wid] = mx;
  __syncthreads();
  mx = 0.f;
  for (int i = 0; i < (nthr >> 6); ++i) mx = fmaxf(mx, s_max[i]);
  const float scale = sp_row_scale(mx, tid == 0 ? inv_scale + row : nullptr);
  uint16_t* oh = out2 + row * (int64_t)hidden;
  uint16_t* ol = out2 + (rows + row) * (int64_t)hidden;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * nthr + tid) * 4;
    if (col < hidden) sp_store2(oh, ol, col, o[c], scale, wt);
  }
}

extern "C" int psg_rmsnorm_split2(psg_ctx* ctx, float* resid, const float* delta, int delta_splits, const float* w, float eps,
                                  int64_t rows, int hidden, void* out2, float* inv_scale, void* stream) {
  PSG_REQUIRE(ctx && resid && w && out2 && inv_scale, PSG_ERR_INVALID, "psg_rmsnorm_split2: NULL argument");
  PSG_REQUIRE(hidden % 4 == 0 && hidden > 0 && hidden <= 8192 && rows >= 0 && rows < (1ll << 31), PSG_ERR_UNSUPPORTED,
              "psg_rmsnorm_split2: hidden=%d (multiple of 4, <= 8192)", hidden);
  PSG_REQUIRE(delta_splits >= 0 && delta_splits <= PSG_MAX_SPLITS && (!delta || delta_splits >= 1), PSG_ERR_INVALID,
              "psg_rmsnorm_split2: delta_splits=%d (fp32 split-K slices)", delta_splits);
  if (rows == 0) return PSG_OK;
  const int nthr = (rows <= 64 && hidden >= 4096) ? 1024 : 256;       // psg_rmsnorm's choice: the same summation tree
  const int nch = (hidden + 4 * nthr - 1) / (4 * nthr);
  hipStream_t st = (hipStream_t)stream;
#define RS2(N)                                                                                                          \
  rmsnorm_split2_kernel<N><<<(unsigned)rows, nthr, 0, st>>>(resid, delta, delta_splits, rows * (int64_t)hidden, w, eps, \
                                                            hidden, rows, (uint16_t*)out2, inv_scale, (ctx->opt.wt_stores >> 2) & 1)
  switch (nch) {
    case 1: RS2(1); break;
    case 2: RS2(2); break;
    case 3: RS2(3); break;
    case 4: RS2(4); break;
    case 5: RS2(5); break;
    case 6: RS2(6); break;
    case 7: RS2(7); break;
    default: RS2(8); break;
  }
#undef RS2
  PSG_CHECK_LAUNCH("psg_rmsnorm_split2");
  return PSG_OK;
}
