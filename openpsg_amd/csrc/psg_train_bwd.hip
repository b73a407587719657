// SURVEY 8f rank 3, gradient path: the row / attention kernels of the TRAINING branch of RelationTransformerHeadV4 with
// their backward counterparts (the reference back-propagates binary_rel_cls_loss and rel_llm_loss, V4:327-351, 463-482,
// driven by tools/train.py:239-246; the LLM is frozen, CFG:65, so its weights need no gradient but its activations do:
// the loss reaches language_projection and the Q-Former THROUGH the 32 Llama layers).
//
// Training batches are tiny (<= 32 sampled pairs through the Q-Former, V4:29-30; <= 4 pairs through the LLM, V4:38),
// so these are plain fp32 kernels - one wave per row or per (sequence, head, query row) - written for exactness against
// autograd on the CPU oracle, not for speed; the dense projections and their weight gradients go through the library
// GEMM.  Every kernel is the exact adjoint of the forward kernel next to it:
//
//   psg_train_layernorm_fwd / _bwd     HF-IB LayerNorm (eps 1e-12): y = (x - mean) * rstd * gamma + beta
//   psg_train_rmsnorm_fwd / _bwd       HF-LL:53-67 (weight frozen: no weight gradient)
//   psg_train_attn_fwd / _bwd          softmax(q.k * scale + additive mask) v for Q-Former self- / cross-attention
//                                      (HF-IB:176-196, keys / values shared by all sequences when Bk == 1) and the Llama
//                                      attention (HF-LL:191-214); an all-masked row is a uniform softmax, and its score
//                                      gradient is p (dP - sum p dP) like any other row - what autograd computes for the
//                                      reference's additive finfo.min masks
//   psg_train_gelu_fwd / _bwd          exact-erf GELU (HF-IB:563-577)
//   psg_train_silu_mul_fwd / _bwd      SwiGLU gate (HF-LL:163-177)
//   psg_train_rope                     half-split rotary (HF-LL:130-160); sign = -1 is its adjoint
//   psg_train_ce_bwd / psg_train_bce_bwd   gradients of psg_cross_entropy_rows / psg_bce_with_logits
#include "psg_common.h"

#define TR_FMIN (-3.4028234663852886e38f)

// ---- LayerNorm ----------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) tr_layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, int64_t rows,
                                                               int hidden, float* __restrict__ y, float* __restrict__ mean,
                                                               float* __restrict__ rstd) {
  const int lane = threadIdx.x & 63;
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= rows) return;
  const float* xr = x + row * hidden;
  float s = 0.f;
  for (int c = lane; c < hidden; c += 64) s += xr[c];
  const float mu = wave_sum(s) / (float)hidden;
  float v = 0.f;
  for (int c = lane; c < hidden; c += 64) {
    const float d = xr[c] - mu;
    v += d * d;
  }
  const float rs = 1.0f / sqrtf(wave_sum(v) / (float)hidden + eps);
  for (int c = lane; c < hidden; c += 64) y[row * hidden + c] = (xr[c] - mu) * rs * gamma[c] + beta[c];
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma; dgamma += dy * xhat, dbeta += dy (atomics)
__global__ void __launch_bounds__(256) tr_layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               const float* __restrict__ gamma, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, int64_t rows, int hidden,
                                                               float* __restrict__ dx, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta) {
  const int lane = threadIdx.x & 63;
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= rows) return;
  const float mu = mean[row], rs = rstd[row];
  const float* xr = x + row * hidden;
  const float* dr = dy + row * hidden;
  float a = 0.f, b = 0.f;
  for (int c = lane; c < hidden; c += 64) {
    const float g = dr[c] * gamma[c], xh = (xr[c] - mu) * rs;
    a += g;
    b += g * xh;
  }
  a = wave_sum(a) / (float)hidden;
  b = wave_sum(b) / (float)hidden;
  for (int c = lane; c < hidden; c += 64) {
    const float xh = (xr[c] - mu) * rs;
    dx[row * hidden + c] = rs * (dr[c] * gamma[c] - a - xh * b);
    if (dgamma) atomicAdd(dgamma + c, dr[c] * xh);
    if (dbeta) atomicAdd(dbeta + c, dr[c]);
  }
}

extern "C" int psg_train_layernorm_fwd(psg_ctx* ctx, const float* x, const float* gamma, const float* beta, float eps,
                                       int64_t rows, int hidden, float* y, float* mean, float* rstd, void* stream) {
  PSG_REQUIRE(ctx && x && gamma && beta && y && mean && rstd && hidden > 0, PSG_ERR_INVALID,
              "psg_train_layernorm_fwd: bad argument");
  if (rows == 0) return PSG_OK;
  tr_layernorm_fwd_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(x, gamma, beta, eps, rows, hidden, y,
                                                                                      mean, rstd);
  PSG_CHECK_LAUNCH("psg_train_layernorm_fwd");
  return PSG_OK;
}

extern "C" int psg_train_layernorm_bwd(psg_ctx* ctx, const float* x, const float* dy, const float* gamma, const float* mean,
                                       const float* rstd, int64_t rows, int hidden, float* dx, float* dgamma, float* dbeta,
                                       void* stream) {
  PSG_REQUIRE(ctx && x && dy && gamma && mean && rstd && dx && hidden > 0, PSG_ERR_INVALID,
              "psg_train_layernorm_bwd: bad argument");
  if (rows == 0) return PSG_OK;
  tr_layernorm_bwd_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(x, dy, gamma, mean, rstd, rows, hidden,
                                                                                      dx, dgamma, dbeta);
  PSG_CHECK_LAUNCH("psg_train_layernorm_bwd");
  return PSG_OK;
}

// ---- RMSNorm (weight frozen) ----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) tr_rmsnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             float eps, int64_t rows, int hidden, float* __restrict__ y,
                                                             float* __restrict__ rstd) {
  const int lane = threadIdx.x & 63;
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= rows) return;
  const float* xr = x + row * hidden;
  float s = 0.f;
  for (int c = lane; c < hidden; c += 64) s += xr[c] * xr[c];
  const float rs = 1.0f / sqrtf(wave_sum(s) / (float)hidden + eps);
  for (int c = lane; c < hidden; c += 64) y[row * hidden + c] = w[c] * (xr[c] * rs);
  if (lane == 0) rstd[row] = rs;
}

// y = w x r, r = (mean x^2 + eps)^-1/2:  dx = r (g - x r^2 mean(g x)), g = dy w
__global__ void __launch_bounds__(256) tr_rmsnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ w, const float* __restrict__ rstd,
                                                             int64_t rows, int hidden, float* __restrict__ dx) {
  const int lane = threadIdx.x & 63;
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= rows) return;
  const float rs = rstd[row];
  const float* xr = x + row * hidden;
  const float* dr = dy + row * hidden;
  float a = 0.f;
  for (int c = lane; c < hidden; c += 64) a += dr[c] * w[c] * xr[c];
  a = wave_sum(a) / (float)hidden;
  for (int c = lane; c < hidden; c += 64) dx[row * hidden + c] = rs * (dr[c] * w[c] - xr[c] * rs * rs * a);
}

extern "C" int psg_train_rmsnorm_fwd(psg_ctx* ctx, const float* x, const float* w, float eps, int64_t rows, int hidden,
                                     float* y, float* rstd, void* stream) {
  PSG_REQUIRE(ctx && x && w && y && rstd && hidden > 0, PSG_ERR_INVALID, "psg_train_rmsnorm_fwd: bad argument");
  if (rows == 0) return PSG_OK;
  tr_rmsnorm_fwd_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(x, w, eps, rows, hidden, y, rstd);
  PSG_CHECK_LAUNCH("psg_train_rmsnorm_fwd");
  return PSG_OK;
}

extern "C" int psg_train_rmsnorm_bwd(psg_ctx* ctx, const float* x, const float* dy, const float* w, const float* rstd,
                                     int64_t rows, int hidden, float* dx, void* stream) {
  PSG_REQUIRE(ctx && x && dy && w && rstd && dx && hidden > 0, PSG_ERR_INVALID, "psg_train_rmsnorm_bwd: bad argument");
  if (rows == 0) return PSG_OK;
  tr_rmsnorm_bwd_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(x, dy, w, rstd, rows, hidden, dx);
  PSG_CHECK_LAUNCH("psg_train_rmsnorm_bwd");
  return PSG_OK;
}

// ---- attention -------------------------------------------------------------------------------------------------------
// q [B][Sq][H*D], k / v [Bk][Sk][H*D] (Bk = B, or 1 = shared by every sequence), keep uint8 [B][Mq][Sk] (Mq = Sq, or
// 1 = one key mask for all query rows; 1 = attend), p [B][H][Sq][Sk] (saved for the backward), out [B][Sq][H*D].
// drop (may be NULL) uint8 [B][H][Sq][Sk]: attention-probability dropout (HF-IB:176-196 `self.dropout(attention_probs)`,
// active when the reference trains): out = sum_j p_j drop_j drop_scale v_j with drop_scale = 1 / (1 - p_drop); p is saved
// BEFORE the dropout.  One wave per (b, h, query row); Sk <= 1024; D <= 128.
#define TR_MAXK 16   // keys per lane

__global__ void __launch_bounds__(64) tr_attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, const uint8_t* __restrict__ keep,
                                                         int B, int Bk, int H, int Sq, int Sk, int D, int Mq, float scale,
                                                         const uint8_t* __restrict__ drop, float drop_scale,
                                                         float* __restrict__ p, float* __restrict__ out) {
  __shared__ float s_p[TR_MAXK * 64];
  const int lane = threadIdx.x;
  const int i = blockIdx.x % Sq, h = (blockIdx.x / Sq) % H, b = blockIdx.x / (Sq * H);
  const int hid = H * D;
  const float* qr = q + ((int64_t)b * Sq + i) * hid + h * D;
  const float* kb = k + (int64_t)(Bk == 1 ? 0 : b) * Sk * hid + h * D;
  const float* vb = v + (int64_t)(Bk == 1 ? 0 : b) * Sk * hid + h * D;
  const uint8_t* mk = keep + ((int64_t)b * Mq + (Mq == 1 ? 0 : i)) * Sk;
  float s[TR_MAXK];
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < TR_MAXK; ++t) {
    const int j = t * 64 + lane;
    s[t] = -INFINITY;
    if (j < Sk) {
      float acc = 0.f;
      for (int d = 0; d < D; ++d) acc += qr[d] * kb[(int64_t)j * hid + d];
      acc *= scale;
      if (!mk[j]) acc = acc + TR_FMIN;                     // additive finfo.min (absorbs the score), as the reference
      s[t] = acc;
      mx = fmaxf(mx, acc);
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < TR_MAXK; ++t) {
    const int j = t * 64 + lane;
    if (j < Sk) {
      s[t] = expf(s[t] - mx);
      sum += s[t];
    }
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  float* pr = p + (((int64_t)b * H + h) * Sq + i) * Sk;
  const uint8_t* dr = drop ? drop + (((int64_t)b * H + h) * Sq + i) * Sk : nullptr;
#pragma unroll
  for (int t = 0; t < TR_MAXK; ++t) {
    const int j = t * 64 + lane;
    if (j < Sk) {
      const float pv = s[t] * inv;
      s_p[j] = dr ? (dr[j] ? pv * drop_scale : 0.f) : pv;
      pr[j] = pv;
    }
  }
  __syncthreads();
  for (int d = lane; d < D; d += 64) {
    float acc = 0.f;
    for (int j = 0; j < Sk; ++j) acc += s_p[j] * vb[(int64_t)j * hid + d];
    out[((int64_t)b * Sq + i) * hid + h * D + d] = acc;
  }
}

// dP_j = dO . v_j (x drop_j drop_scale under dropout); c = sum_j p_j dP_j; dS_j = p_j (dP_j - c); dq = scale sum_j dS_j k_j;
// dk_j += scale dS_j q; dv_j += p_j (drop_j drop_scale) dO  (dk / dv by atomics: rows of many queries - and, shared keys, of
// many sequences - add up)
__global__ void __launch_bounds__(64) tr_attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, const float* __restrict__ p,
                                                         const float* __restrict__ dout, int B, int Bk, int H, int Sq, int Sk,
                                                         int D, float scale, const uint8_t* __restrict__ drop,
                                                         float drop_scale, float* __restrict__ dq, float* __restrict__ dk,
                                                         float* __restrict__ dv) {
  __shared__ float s_ds[TR_MAXK * 64];
  __shared__ float s_p[TR_MAXK * 64];
  const int lane = threadIdx.x;
  const int i = blockIdx.x % Sq, h = (blockIdx.x / Sq) % H, b = blockIdx.x / (Sq * H);
  const int hid = H * D;
  const int64_t qoff = ((int64_t)b * Sq + i) * hid + h * D;
  const int64_t kvoff = (int64_t)(Bk == 1 ? 0 : b) * Sk * hid + h * D;
  const float* pr = p + (((int64_t)b * H + h) * Sq + i) * Sk;
  const uint8_t* dr = drop ? drop + (((int64_t)b * H + h) * Sq + i) * Sk : nullptr;
  float dp[TR_MAXK];
  float c = 0.f;
#pragma unroll
  for (int t = 0; t < TR_MAXK; ++t) {
    const int j = t * 64 + lane;
    dp[t] = 0.f;
    if (j < Sk) {
      float acc = 0.f;
      for (int d = 0; d < D; ++d) acc += dout[qoff + d] * v[kvoff + (int64_t)j * hid + d];
      if (dr) acc = dr[j] ? acc * drop_scale : 0.f;
      dp[t] = acc;
      c += pr[j] * acc;
    }
  }
  c = wave_sum(c);
#pragma unroll
  for (int t = 0; t < TR_MAXK; ++t) {
    const int j = t * 64 + lane;
    if (j < Sk) {
      s_p[j] = dr ? (dr[j] ? pr[j] * drop_scale : 0.f) : pr[j];
      s_ds[j] = pr[j] * (dp[t] - c) * scale;
    }
  }
  __syncthreads();
  for (int d = lane; d < D; d += 64) {
    float acc = 0.f;
    const float qd = q[qoff + d], dod = dout[qoff + d];
    for (int j = 0; j < Sk; ++j) {
      const float ds = s_ds[j];
      acc += ds * k[kvoff + (int64_t)j * hid + d];
      if (ds != 0.f) atomicAdd(dk + kvoff + (int64_t)j * hid + d, ds * qd);
      const float pj = s_p[j];
      if (pj != 0.f) atomicAdd(dv + kvoff + (int64_t)j * hid + d, pj * dod);
    }
    dq[qoff + d] = acc;
  }
}

extern "C" int psg_train_attn_fwd(psg_ctx* ctx, const float* q, const float* k, const float* v, const uint8_t* keep, int B,
                                  int Bk, int H, int Sq, int Sk, int D, int Mq, float scale, const uint8_t* drop,
                                  float drop_scale, float* p, float* out, void* stream) {
  PSG_REQUIRE(ctx && q && k && v && keep && p && out, PSG_ERR_INVALID, "psg_train_attn_fwd: NULL argument");
  PSG_REQUIRE(B >= 0 && (Bk == B || Bk == 1) && H > 0 && Sq > 0 && Sk > 0 && Sk <= TR_MAXK * 64 && D > 0 && D <= 128 &&
                  (Mq == 1 || Mq == Sq),
              PSG_ERR_UNSUPPORTED, "psg_train_attn_fwd: B=%d Bk=%d H=%d Sq=%d Sk=%d D=%d Mq=%d", B, Bk, H, Sq, Sk, D, Mq);
  if (B == 0) return PSG_OK;
  tr_attn_fwd_kernel<<<(unsigned)(B * H * Sq), 64, 0, (hipStream_t)stream>>>(q, k, v, keep, B, Bk, H, Sq, Sk, D, Mq, scale,
                                                                           drop, drop_scale, p, out);
  PSG_CHECK_LAUNCH("psg_train_attn_fwd");
  return PSG_OK;
}

extern "C" int psg_train_attn_bwd(psg_ctx* ctx, const float* q, const float* k, const float* v, const float* p,
                                  const float* dout, int B, int Bk, int H, int Sq, int Sk, int D, float scale,
                                  const uint8_t* drop, float drop_scale, float* dq, float* dk, float* dv, void* stream) {
  PSG_REQUIRE(ctx && q && k && v && p && dout && dq && dk && dv, PSG_ERR_INVALID, "psg_train_attn_bwd: NULL argument");
  PSG_REQUIRE(B >= 0 && (Bk == B || Bk == 1) && H > 0 && Sq > 0 && Sk > 0 && Sk <= TR_MAXK * 64 && D > 0 && D <= 128,
              PSG_ERR_UNSUPPORTED, "psg_train_attn_bwd: B=%d Bk=%d H=%d Sq=%d Sk=%d D=%d", B, Bk, H, Sq, Sk, D);
  if (B == 0) return PSG_OK;
  // dk / dv are accumulated: the caller hands them in zeroed
  tr_attn_bwd_kernel<<<(unsigned)(B * H * Sq), 64, 0, (hipStream_t)stream>>>(q, k, v, p, dout, B, Bk, H, Sq, Sk, D, scale, drop,
                                                                           drop_scale, dq, dk, dv);
  PSG_CHECK_LAUNCH("psg_train_attn_bwd");
  return PSG_OK;
}

// ---- element-wise: GELU, SwiGLU gate, rotary ---------------------------------------------------------------------------
__global__ void tr_gelu_kernel(const float* __restrict__ x, const float* __restrict__ dy, int64_t n, float* __restrict__ o) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752f));
  if (dy) o[i] = dy[i] * (cdf + v * 0.3989422804014327f * expf(-0.5f * v * v));   // d/dx [x Phi(x)] = Phi + x phi
  else o[i] = v * cdf;
}

extern "C" int psg_train_gelu_fwd(psg_ctx* ctx, const float* x, int64_t n, float* y, void* stream) {
  PSG_REQUIRE(ctx && x && y && n >= 0, PSG_ERR_INVALID, "psg_train_gelu_fwd: bad argument");
  if (n == 0) return PSG_OK;
  tr_gelu_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, nullptr, n, y);
  PSG_CHECK_LAUNCH("psg_train_gelu_fwd");
  return PSG_OK;
}

extern "C" int psg_train_gelu_bwd(psg_ctx* ctx, const float* x, const float* dy, int64_t n, float* dx, void* stream) {
  PSG_REQUIRE(ctx && x && dy && dx && n >= 0, PSG_ERR_INVALID, "psg_train_gelu_bwd: bad argument");
  if (n == 0) return PSG_OK;
  tr_gelu_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, dy, n, dx);
  PSG_CHECK_LAUNCH("psg_train_gelu_bwd");
  return PSG_OK;
}

// gu [rows][2 * inter] = gate | up; y = silu(gate) * up
__global__ void tr_silu_mul_kernel(const float* __restrict__ gu, const float* __restrict__ dy, int64_t rows, int inter,
                                   float* __restrict__ y, float* __restrict__ dgu) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * inter) return;
  const int64_t r = i / inter;
  const int c = (int)(i % inter);
  const float g = gu[r * 2 * inter + c], u = gu[r * 2 * inter + inter + c];
  const float sg = 1.0f / (1.0f + expf(-g));
  if (!dy) {
    y[i] = g * sg * u;
  } else {
    const float d = dy[i];
    dgu[r * 2 * inter + c] = d * u * sg * (1.0f + g * (1.0f - sg));
    dgu[r * 2 * inter + inter + c] = d * g * sg;
  }
}

extern "C" int psg_train_silu_mul_fwd(psg_ctx* ctx, const float* gu, int64_t rows, int inter, float* y, void* stream) {
  PSG_REQUIRE(ctx && gu && y && inter > 0, PSG_ERR_INVALID, "psg_train_silu_mul_fwd: bad argument");
  if (rows == 0) return PSG_OK;
  tr_silu_mul_kernel<<<(unsigned)((rows * inter + 255) / 256), 256, 0, (hipStream_t)stream>>>(gu, nullptr, rows, inter, y,
                                                                                           nullptr);
  PSG_CHECK_LAUNCH("psg_train_silu_mul_fwd");
  return PSG_OK;
}

extern "C" int psg_train_silu_mul_bwd(psg_ctx* ctx, const float* gu, const float* dy, int64_t rows, int inter, float* dgu,
                                      void* stream) {
  PSG_REQUIRE(ctx && gu && dy && dgu && inter > 0, PSG_ERR_INVALID, "psg_train_silu_mul_bwd: bad argument");
  if (rows == 0) return PSG_OK;
  tr_silu_mul_kernel<<<(unsigned)((rows * inter + 255) / 256), 256, 0, (hipStream_t)stream>>>(gu, dy, rows, inter, nullptr,
                                                                                           dgu);
  PSG_CHECK_LAUNCH("psg_train_silu_mul_bwd");
  return PSG_OK;
}

// x [rows][heads * head_dim], pos int32 [rows] (row of the cos / sin tables [table_rows][head_dim / 2]):
// y = x cos + rotate_half(x) sin * sign.  sign = +1: HF-LL:130-160; sign = -1: its adjoint (the rotation by -angle).
__global__ void tr_rope_kernel(const float* __restrict__ x, const int32_t* __restrict__ pos, const float* __restrict__ cs,
                               const float* __restrict__ sn, int table_rows, int64_t rows, int heads, int head_dim,
                               float sign, float* __restrict__ y) {
  const int half = head_dim / 2;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * heads * half) return;
  const int d = (int)(i % half);
  const int h = (int)((i / half) % heads);
  const int64_t r = i / ((int64_t)half * heads);
  const int64_t base = (r * heads + h) * head_dim;
  int pr = pos[r];
  pr = pr < 0 ? 0 : (pr >= table_rows ? table_rows - 1 : pr);  // never read outside the tables (RopeFn checks the range)
  const float c = cs[(int64_t)pr * half + d], s = sn[(int64_t)pr * half + d] * sign;
  const float a = x[base + d], b = x[base + d + half];
  y[base + d] = a * c - b * s;                               // rotate_half(x) = [-x2, x1]
  y[base + d + half] = b * c + a * s;
}

extern "C" int psg_train_rope(psg_ctx* ctx, const float* x, const int32_t* pos, const float* rope_cos, const float* rope_sin,
                              int table_rows, int64_t rows, int heads, int head_dim, float sign, float* y, void* stream) {
  PSG_REQUIRE(ctx && x && pos && rope_cos && rope_sin && y && heads > 0 && head_dim > 0 && head_dim % 2 == 0 &&
                  table_rows > 0,
              PSG_ERR_INVALID, "psg_train_rope: bad argument");
  if (rows == 0) return PSG_OK;
  const int64_t n = rows * heads * (head_dim / 2);
  tr_rope_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, pos, rope_cos, rope_sin, table_rows,
                                                                             rows, heads, head_dim, sign, y);
  PSG_CHECK_LAUNCH("psg_train_rope");
  return PSG_OK;
}

// ---- loss gradients ---------------------------------------------------------------------------------------------------
// dlogits[row] = dloss[row] * (softmax(logits[row]) - onehot(label)); rows with label < 0 (ignore_index) get zeros
__global__ void __launch_bounds__(256) tr_ce_bwd_kernel(const float* __restrict__ logits, int vocab,
                                                        const int32_t* __restrict__ labels, const float* __restrict__ dloss,
                                                        float* __restrict__ dlogits) {
  __shared__ float s_red[4];
  const int64_t row = blockIdx.x;
  const int lab = labels[row];
  const float* x = logits + row * vocab;
  float* dx = dlogits + row * vocab;
  if (lab < 0 || lab >= vocab) {
    for (int i = threadIdx.x; i < vocab; i += 256) dx[i] = 0.f;
    return;
  }
  float m = -INFINITY;
  for (int i = threadIdx.x; i < vocab; i += 256) m = fmaxf(m, x[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < vocab; i += 256) s += expf(x[i] - m);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float inv = 1.0f / (s_red[0] + s_red[1] + s_red[2] + s_red[3]);
  const float g = dloss[row];
  for (int i = threadIdx.x; i < vocab; i += 256) dx[i] = g * (expf(x[i] - m) * inv - (i == lab ? 1.0f : 0.0f));
}

extern "C" int psg_train_ce_bwd(psg_ctx* ctx, const float* logits, int64_t rows, int vocab, const int32_t* labels,
                                const float* dloss, float* dlogits, void* stream) {
  PSG_REQUIRE(ctx && logits && labels && dloss && dlogits && vocab > 0, PSG_ERR_INVALID, "psg_train_ce_bwd: bad argument");
  if (rows == 0) return PSG_OK;
  tr_ce_bwd_kernel<<<(unsigned)rows, 256, 0, (hipStream_t)stream>>>(logits, vocab, labels, dloss, dlogits);
  PSG_CHECK_LAUNCH("psg_train_ce_bwd");
  return PSG_OK;
}

// loss = weight / n * sum_i bce(x_i, y_i):  dx_i = dloss * weight / n * (sigmoid(x_i) - y_i)
__global__ void tr_bce_bwd_kernel(const float* __restrict__ logit, const float* __restrict__ label, int n, float weight,
                                  const float* __restrict__ dloss, float* __restrict__ dlogit) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  dlogit[i] = dloss[0] * weight / (float)n * (1.0f / (1.0f + expf(-logit[i])) - label[i]);
}

extern "C" int psg_train_bce_bwd(psg_ctx* ctx, const float* logit, const float* label, int n, float weight,
                                 const float* dloss, float* dlogit, void* stream) {
  PSG_REQUIRE(ctx && logit && label && dloss && dlogit && n > 0, PSG_ERR_INVALID, "psg_train_bce_bwd: bad argument");
  tr_bce_bwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(logit, label, n, weight, dloss, dlogit);
  PSG_CHECK_LAUNCH("psg_train_bce_bwd");
  return PSG_OK;
}
