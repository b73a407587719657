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
  uint16_t* o0 = out + row * 3 * (int64_t)K;
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
    *reinterpret_cast<ushort4*>(o0 + c) = h;
    *reinterpret_cast<ushort4*>(oh2 + c) = h;
    *reinterpret_cast<ushort4*>(ol + c) = l;
  }
}

extern "C" int psg_split_f16x3(psg_ctx* ctx, const float* x, int64_t rows, int K, int64_t row_stride, int order,
                               void* out, float* inv_scale, void* stream) {
  PSG_REQUIRE(ctx && x && out && inv_scale, PSG_ERR_INVALID, "psg_split_f16x3: NULL argument");
  PSG_REQUIRE(rows >= 0 && K > 0 && K % 4 == 0 && row_stride >= K && row_stride % 4 == 0 &&
                  (order == SP_ORDER_HHL || order == SP_ORDER_HLH) && rows < (1ll << 31),
              PSG_ERR_INVALID, "psg_split_f16x3: rows=%lld K=%d stride=%lld order=%d", (long long)rows, K,
              (long long)row_stride, order);
  if (rows == 0) return PSG_OK;
  split_f16x3_kernel<<<(unsigned)rows, 256, 0, (hipStream_t)stream>>>(x, row_stride, K, order, (uint16_t*)out, inv_scale);
  PSG_CHECK_LAUNCH("psg_split_f16x3");
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
