// the integer / row kernels of the TRAINING branch of RelationTransformerHeadV4
// (forward arithmetic of the losses; the matrix work reuses the inference kernels).
//
//   psg_train_object_bitmasks  V4:371-399 (prepare_train): ground-truth thing masks resampled to the patch grid
//                              by bilinear interpolation (align_corners=False) and thresholded at 0.5, stuff masks
//                              from the nearest-resampled semantic map == category; packed like psg_object_bitmasks
//   psg_bce_with_logits        V4:463-482, binary case: mean BCE-with-logits x rel_cls_loss_weight
//   psg_cross_entropy_rows     V4:327-341: per-row -log softmax(logits)[label] (label < 0 = ignored), fp32
#include "psg_common.h"

// ATen upsample_bilinear2d, align_corners = false: src = scale * (dst + 0.5) - 0.5 clamped at 0, scale = in / out
// in float; i1 = i0 + (i0 < in - 1); value = h0 (w0 p00 + w1 p01) + h1 (w0 p10 + w1 p11)
__device__ __forceinline__ void bilinear_src(int dst, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.0f - l1;
}

__global__ void __launch_bounds__(64) train_object_bitmasks_kernel(const uint8_t* __restrict__ thing_masks,
                                                                   const int32_t* __restrict__ sem, int H, int W,
                                                                   const int32_t* __restrict__ is_thing,
                                                                   const int32_t* __restrict__ category,
                                                                   const int32_t* __restrict__ thing_index, int N,
                                                                   int gh, int gw, uint64_t* __restrict__ bits,
                                                                   int words) {
  const int n = blockIdx.x / words, wd = blockIdx.x % words;
  const int l = wd * 64 + threadIdx.x;
  bool on = false;
  if (n < N && l < gh * gw) {
    const int r = l / gw, c = l % gw;
    if (is_thing[n]) {
      const uint8_t* m = thing_masks + (int64_t)thing_index[n] * H * W;
      int y0, y1, x0, x1;
      float hy0, hy1, wx0, wx1;
      bilinear_src(r, (float)H / (float)gh, H, y0, y1, hy0, hy1);
      bilinear_src(c, (float)W / (float)gw, W, x0, x1, wx0, wx1);
      const float p00 = m[(int64_t)y0 * W + x0], p01 = m[(int64_t)y0 * W + x1];
      const float p10 = m[(int64_t)y1 * W + x0], p11 = m[(int64_t)y1 * W + x1];
      const float v = hy0 * (wx0 * p00 + wx1 * p01) + hy1 * (wx0 * p10 + wx1 * p11);
      on = v > 0.5f;
    } else {                                               // legacy nearest: src = min(floor(dst * scale), in - 1)
      const int y = min((int)floorf((float)r * ((float)H / (float)gh)), H - 1);
      const int x = min((int)floorf((float)c * ((float)W / (float)gw)), W - 1);
      on = (float)sem[(int64_t)y * W + x] == (float)category[n];
    }
  }
  const unsigned long long b = __ballot(on);
  if (threadIdx.x == 0 && n < N) bits[(int64_t)n * words + wd] = b;
}

extern "C" int psg_train_object_bitmasks(psg_ctx* ctx, const uint8_t* thing_masks, int n_thing, const int32_t* sem,
                                         int H, int W, const int32_t* is_thing, const int32_t* category,
                                         const int32_t* thing_index, int N, int gh, int gw, uint64_t* bits, int words,
                                         void* stream) {
  PSG_REQUIRE(ctx && sem && is_thing && category && thing_index && bits && (thing_masks || n_thing == 0),
              PSG_ERR_INVALID, "psg_train_object_bitmasks: NULL argument");
  PSG_REQUIRE(N > 0 && H > 0 && W > 0 && gh > 0 && gw > 0 && words * 64 >= gh * gw && n_thing >= 0, PSG_ERR_INVALID,
              "psg_train_object_bitmasks: N=%d H=%d W=%d grid=%dx%d words=%d", N, H, W, gh, gw, words);
  train_object_bitmasks_kernel<<<(unsigned)(N * words), 64, 0, (hipStream_t)stream>>>(
      thing_masks, sem, H, W, is_thing, category, thing_index, N, gh, gw, bits, words);
  PSG_CHECK_LAUNCH("psg_train_object_bitmasks");
  return PSG_OK;
}

// mean over n of max(x,0) - x y + log(1 + exp(-|x|)), times `weight`; one workgroup, fixed summation order
__global__ void __launch_bounds__(256) bce_with_logits_kernel(const float* __restrict__ logit,
                                                              const float* __restrict__ label, int n, float weight,
                                                              float* __restrict__ out) {
  __shared__ float s_part[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float x = logit[i], y = label[i];
    acc += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (s_part[0] + s_part[1] + s_part[2] + s_part[3]) / (float)n * weight;
}

extern "C" int psg_bce_with_logits(psg_ctx* ctx, const float* logit, const float* label, int n, float weight,
                                   float* out, void* stream) {
  PSG_REQUIRE(ctx && logit && label && out && n > 0, PSG_ERR_INVALID, "psg_bce_with_logits: bad argument (n=%d)", n);
  bce_with_logits_kernel<<<1, 256, 0, (hipStream_t)stream>>>(logit, label, n, weight, out);
  PSG_CHECK_LAUNCH("psg_bce_with_logits");
  return PSG_OK;
}

// one workgroup per row: loss[row] = logsumexp(x) - x[label]; label < 0 -> loss 0 (ignore_index)
template <typename T>
__global__ void __launch_bounds__(256) cross_entropy_rows_kernel(const T* __restrict__ logits, int vocab,
                                                                 const int32_t* __restrict__ labels,
                                                                 float* __restrict__ loss) {
  __shared__ float s_red[4];
  const int64_t row = blockIdx.x;
  const int lab = labels[row];
  if (lab < 0 || lab >= vocab) {                          // uniform per workgroup
    if (threadIdx.x == 0) loss[row] = 0.f;
    return;
  }
  const T* x = logits + row * vocab;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < vocab; i += 256) m = fmaxf(m, Act<T>::ld(x, i));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < vocab; i += 256) s += expf(Act<T>::ld(x, i) - m);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss[row] = logf(s_red[0] + s_red[1] + s_red[2] + s_red[3]) + m - Act<T>::ld(x, lab);
}

extern "C" int psg_cross_entropy_rows(psg_ctx* ctx, const void* logits, int64_t rows, int vocab, const int32_t* labels,
                                      float* loss, int dtype, void* stream) {
  PSG_REQUIRE(ctx && logits && labels && loss && vocab > 0 && rows >= 0, PSG_ERR_INVALID,
              "psg_cross_entropy_rows: bad argument");
  if (rows == 0) return PSG_OK;
  PSG_DISPATCH_DTYPE(dtype, "psg_cross_entropy_rows",
                     (cross_entropy_rows_kernel<T><<<(unsigned)rows, 256, 0, (hipStream_t)stream>>>(
                         (const T*)logits, vocab, labels, loss)));
  PSG_CHECK_LAUNCH("psg_cross_entropy_rows");
  return PSG_OK;
}
