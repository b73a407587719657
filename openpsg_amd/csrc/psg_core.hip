// Context, error handling and the integer / byte kernels of the path:
//   K2 mask -> patch grid, K3 object bitmasks, K9 top-k selector, row gather.
#include <ctype.h>
#include <stdarg.h>
#include <stdlib.h>

#include "psg_common.h"

static thread_local char g_err[512] = "";

void psg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* psg_last_error(void) { return g_err; }
extern "C" int psg_version(void) { return PSG_ABI_VERSION; }

struct OptName {
  const char* name;
  int psg_opts::*field;
};
static const OptName kOpts[] = {
    {"skinny_splits", &psg_opts::skinny_splits},       {"skinny_balance", &psg_opts::skinny_balance},
    {"skinny_wg_per_cu", &psg_opts::skinny_wg_per_cu}, {"skinny_dma", &psg_opts::skinny_dma},
    {"skinny_nt", &psg_opts::skinny_nt},               {"skinny_xdma", &psg_opts::skinny_xdma},
    {"skinny_wide", &psg_opts::skinny_wide},           {"skinny_f32_slots", &psg_opts::skinny_f32_slots},
    {"skinny_f32_waves", &psg_opts::skinny_f32_waves}, {"decode_persistent", &psg_opts::decode_persistent},
    {"llm_fuse_split", &psg_opts::llm_fuse_split},
    {"split_i2", &psg_opts::split_i2},
    {"qformer_split_cls_input_space", &psg_opts::qformer_split_cls_input_space},
    {"wt_stores", &psg_opts::wt_stores},
    {"selfattn_scalar", &psg_opts::selfattn_scalar},   {"decode_attn_1wave", &psg_opts::decode_attn_1wave},
    {"xattn_dma", &psg_opts::xattn_dma},               {"xattn_waves", &psg_opts::xattn_waves},
    {"dense_gemm_var", &psg_opts::dense_gemm_var},     {"qformer_own_gemm", &psg_opts::qformer_own_gemm},
    {"ln_half_wave", &psg_opts::ln_half_wave},         {"xattn_poll", &psg_opts::xattn_poll},
    {"xattn_wt", &psg_opts::xattn_wt},
    {"xattn_dynamic", &psg_opts::xattn_dynamic},
    {"llm_w16", &psg_opts::llm_w16},
    {"batch_gemm_bn", &psg_opts::batch_gemm_bn},       {"decode_batch_gemm", &psg_opts::decode_batch_gemm},
    {"batch_gemm_var", &psg_opts::batch_gemm_var},     {"batch_gemm_mode", &psg_opts::batch_gemm_mode},
    {"batch_gemm_grid", &psg_opts::batch_gemm_grid},
    {"qformer_share_qkv", &psg_opts::qformer_share_qkv},
    {"qformer_cls_input_space", &psg_opts::qformer_cls_input_space},
    {"qformer_dedup_prompts", &psg_opts::qformer_dedup_prompts},
    {"llm_fuse_rmsnorm", &psg_opts::llm_fuse_rmsnorm}, {"prefill_attn_scalar", &psg_opts::prefill_attn_scalar},
};

// "8x1x3" (waves x K blocks x ring slots) is accepted for skinny_dma next to a plain integer
static int parse_opt(const char* name, const char* text) {
  int a = 0, b = 0, c = 3;
  if (!strcmp(name, "skinny_dma") && sscanf(text, "%dx%dx%d", &a, &b, &c) >= 2) return a * 100 + b * 10 + c;
  return atoi(text);
}

extern "C" int psg_set_option(psg_ctx* ctx, const char* name, int value) {
  PSG_REQUIRE(ctx && name, PSG_ERR_INVALID, "psg_set_option: NULL argument");
  for (const OptName& o : kOpts)
    if (!strcmp(o.name, name)) {
      ctx->opt.*(o.field) = value;
      return PSG_OK;
    }
  psg_set_error("psg_set_option: unknown option '%s'", name);
  return PSG_ERR_INVALID;
}

extern "C" int psg_get_option(psg_ctx* ctx, const char* name, int* value) {
  PSG_REQUIRE(ctx && name && value, PSG_ERR_INVALID, "psg_get_option: NULL argument");
  for (const OptName& o : kOpts)
    if (!strcmp(o.name, name)) {
      *value = ctx->opt.*(o.field);
      return PSG_OK;
    }
  psg_set_error("psg_get_option: unknown option '%s'", name);
  return PSG_ERR_INVALID;
}

extern "C" int psg_set_trace_buffer(psg_ctx* ctx, int kind, void* device_buffer, int64_t bytes) {
  PSG_REQUIRE(ctx, PSG_ERR_INVALID, "psg_set_trace_buffer: ctx is NULL");
  PSG_REQUIRE(kind == PSG_TRACE_NONE || (device_buffer && bytes >= 8), PSG_ERR_INVALID,
              "psg_set_trace_buffer: kind=%d needs a device buffer", kind);
  ctx->trace_kind = kind;
  ctx->trace = kind == PSG_TRACE_NONE ? nullptr : (long long*)device_buffer;
  ctx->trace_words = kind == PSG_TRACE_NONE ? 0 : bytes / 8;
  return PSG_OK;
}

extern "C" int psg_create(int device, psg_ctx** out) {
  PSG_REQUIRE(out != nullptr, PSG_ERR_INVALID, "psg_create: out is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    psg_set_error("psg_create: no HIP device visible (%s)", e == hipSuccess ? "count=0" : hipGetErrorString(e));
    return PSG_ERR_NO_DEVICE;
  }
  PSG_REQUIRE(device >= 0 && device < n, PSG_ERR_INVALID, "psg_create: device %d out of range [0,%d)", device, n);
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    psg_set_error("psg_create: hipGetDeviceProperties: %s", hipGetErrorString(e));
    return PSG_ERR_HIP;
  }
  psg_ctx* c = new psg_ctx();
  for (const OptName& o : kOpts) {                          // environment overrides, read once per context
    std::string env = "PSG_";
    for (const char* q = o.name; *q; ++q) env += (char)toupper(*q);
    if (const char* e = getenv(env.c_str())) c->opt.*(o.field) = parse_opt(o.name, e);
  }
  c->device = device;
  c->num_cu = prop.multiProcessorCount;
  strncpy(c->arch, prop.gcnArchName, sizeof(c->arch) - 1);
  c->arch[sizeof(c->arch) - 1] = 0;
  *out = c;
  return PSG_OK;
}

extern "C" int psg_destroy(psg_ctx* ctx) {
  delete ctx;
  return PSG_OK;
}

extern "C" int psg_device_info(psg_ctx* ctx, int* num_cu, char* arch, int arch_len) {
  PSG_REQUIRE(ctx != nullptr, PSG_ERR_INVALID, "psg_device_info: ctx is NULL");
  if (num_cu) *num_cu = ctx->num_cu;
  if (arch && arch_len > 0) {
    strncpy(arch, ctx->arch, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return PSG_OK;
}

// ---------------------------------------------------------------------------------------------
// K2: nearest(ori->img) o zero-pad(img->pad) o nearest(pad->grid)   (V4:416-423)
// ATen legacy 'nearest': src = min(floor(dst * float(in/out)), in - 1), scale in float32.
// ---------------------------------------------------------------------------------------------
__global__ void mask_grid_kernel(const int32_t* __restrict__ pan, int H0, int W0, int img_h, int img_w,
                                 int pad_h, int pad_w, int gh, int gw, float* __restrict__ grid) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= gh * gw) return;
  int r = idx / gw, c = idx % gw;
  const float sy2 = (float)pad_h / (float)gh, sx2 = (float)pad_w / (float)gw;
  const float sy1 = (float)H0 / (float)img_h, sx1 = (float)W0 / (float)img_w;
  int y1 = min((int)floorf((float)r * sy2), pad_h - 1);
  int x1 = min((int)floorf((float)c * sx2), pad_w - 1);
  float v = 0.0f;  // F.pad value=0 (aliases void and category-0 instance-0)
  if (y1 < img_h && x1 < img_w) {
    int y0 = min((int)floorf((float)y1 * sy1), H0 - 1);
    int x0 = min((int)floorf((float)x1 * sx1), W0 - 1);
    v = (float)pan[(int64_t)y0 * W0 + x0];  // .float() of the id map (exact below 2^24)
  }
  grid[idx] = v;
}

extern "C" int psg_mask_grid(psg_ctx* ctx, const int32_t* pan, int H0, int W0, int img_h, int img_w, int pad_h,
                             int pad_w, int gh, int gw, float* grid, void* stream) {
  PSG_REQUIRE(ctx && pan && grid, PSG_ERR_INVALID, "psg_mask_grid: NULL argument");
  PSG_REQUIRE(H0 > 0 && W0 > 0 && img_h > 0 && img_w > 0 && pad_h >= img_h && pad_w >= img_w && gh > 0 && gw > 0,
              PSG_ERR_INVALID, "psg_mask_grid: bad shapes ori %dx%d img %dx%d pad %dx%d grid %dx%d", H0, W0, img_h,
              img_w, pad_h, pad_w, gh, gw);
  int n = gh * gw;
  mask_grid_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(pan, H0, W0, img_h, img_w, pad_h, pad_w, gh, gw,
                                                                     grid);
  PSG_CHECK_LAUNCH("psg_mask_grid");
  return PSG_OK;
}

// ---------------------------------------------------------------------------------------------
// K3: bits[n][w] bit b = (grid[64 w + b] == float(id_n))   (V4:425-429); one wave per word
// ---------------------------------------------------------------------------------------------
__global__ void object_bitmasks_kernel(const float* __restrict__ grid, int L, const int32_t* __restrict__ ids, int N,
                                       uint64_t* __restrict__ bits, int words) {
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (wave >= N * words) return;
  int n = wave / words, w = wave % words;
  int l = w * 64 + lane;
  bool hit = (l < L) && (grid[l] == (float)ids[n]);
  unsigned long long m = __ballot(hit);
  if (lane == 0) bits[(int64_t)n * words + w] = (uint64_t)m;
}

extern "C" int psg_object_bitmasks(psg_ctx* ctx, const float* grid, int L, const int32_t* object_ids, int N,
                                   uint64_t* bits, int words, void* stream) {
  PSG_REQUIRE(ctx && grid && object_ids && bits, PSG_ERR_INVALID, "psg_object_bitmasks: NULL argument");
  PSG_REQUIRE(L > 0 && N > 0 && words * 64 >= L, PSG_ERR_INVALID, "psg_object_bitmasks: L=%d N=%d words=%d", L, N,
              words);
  int waves = N * words;
  int blocks = (waves + 3) / 4;
  object_bitmasks_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(grid, L, object_ids, N, bits, words);
  PSG_CHECK_LAUNCH("psg_object_bitmasks");
  return PSG_OK;
}

// ---------------------------------------------------------------------------------------------
// K9: top-k by k rounds of block arg-max (k = 20, n <= ~10^4: latency-trivial, deterministic).
// Order: larger score first, ties -> lower index (the reference's tie order is unspecified).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool topk_better(float a, int ia, float b, int ib) {
  return (a > b) || (a == b && ia < ib);
}

__global__ void __launch_bounds__(1024) topk_kernel(const float* __restrict__ score, int n, int k,
                                                    int32_t* __restrict__ out_idx, float* __restrict__ out_val) {
  __shared__ float s_val[16];
  __shared__ int s_idx[16];
  __shared__ float prev_val;
  __shared__ int prev_idx;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) {
    prev_val = INFINITY;
    prev_idx = -1;
  }
  __syncthreads();
  for (int round = 0; round < k; ++round) {
    const float pv = prev_val;
    const int pi = prev_idx;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < n; i += blockDim.x) {
      float v = score[i];
      if (v != v) v = -INFINITY;  // NaN sorts last
      // strictly after the previous pick in the (value desc, index asc) order
      bool after = (v < pv) || (v == pv && i > pi);
      if (after && topk_better(v, i, best, bi)) {
        best = v;
        bi = i;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float ov = __shfl_xor(best, o, 64);
      int oi = __shfl_xor(bi, o, 64);
      if (topk_better(ov, oi, best, bi)) {
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
        if (topk_better(s_val[w], s_idx[w], best, bi)) {
          best = s_val[w];
          bi = s_idx[w];
        }
      if (bi == 0x7fffffff) {  // fewer than k candidates
        out_idx[round] = -1;
        if (out_val) out_val[round] = -INFINITY;
      } else {
        out_idx[round] = bi;
        if (out_val) out_val[round] = best;
      }
      prev_val = best;
      prev_idx = bi;
    }
    __syncthreads();
  }
}

// Rank-based variant for one image's worth of pairs (n <= 3072, k <= 64): no k dependent rounds.  The scores
// sit in LDS; wave w ranks the elements of segment w among themselves (a lane compares its own <= n/1024 elements with
// every element of the segment: LDS broadcast reads); an element can only be among the k best overall if fewer than k
// elements of its own segment beat it, so at most 16 k candidates remain, which are ranked among themselves the same
// way.  The order (larger score first, ties -> lower index, NaN last) is total, so ranks are unique: same output as
// topk_kernel.
constexpr int TOPK_NMAX = 3072, TOPK_KMAX = 64;
// (score, index) as ONE unsigned 64-bit key whose order is the selector's order: the float's bits made monotone in the
// high word, the complemented index in the low word (larger key = better).  One 64-bit compare per pair of elements:
// the 16 waves of the workgroup share 4 SIMDs, so the n^2 / 16 comparisons are what the kernel costs.
__device__ __forceinline__ unsigned long long topk_key(float v, int i) {
  if (v != v) v = -INFINITY;                                      // NaN sorts last
  if (v == 0.f) v = 0.f;                                          // -0 == +0 for the comparison
  uint32_t u = __float_as_uint(v);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)u << 32) | (uint32_t)(0xffffffffu - (uint32_t)i);
}
template <int OWN>
__global__ void __launch_bounds__(1024) topk_rank_kernel(const float* __restrict__ score, int n, int k,
                                                         int32_t* __restrict__ out_idx, float* __restrict__ out_val) {
  __shared__ unsigned long long s_k[TOPK_NMAX];
  __shared__ unsigned long long c_k[16 * TOPK_KMAX];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int i = tid; i < n; i += 1024) s_k[i] = topk_key(score[i], i);
  for (int i = tid; i < 16 * k; i += 1024) c_k[i] = 0ull;          // empty candidate slot: below every real key
  if (tid < k) {                                                  // fewer than k elements: the tail stays "none"
    out_idx[tid] = -1;
    if (out_val) out_val[tid] = -INFINITY;
  }
  __syncthreads();
  const int seg = (n + 15) / 16, s0 = wid * seg, s1 = min(n, s0 + seg);
  unsigned long long own[OWN];
  int rank[OWN];
#pragma unroll
  for (int u = 0; u < OWN; ++u) {
    const int i = s0 + lane + 64 * u;
    own[u] = i < s1 ? s_k[i] : ~0ull;                             // no element: never a candidate
    rank[u] = 0;
  }
  for (int j = s0; j < s1; ++j) {
    const unsigned long long o = s_k[j];
#pragma unroll
    for (int u = 0; u < OWN; ++u) rank[u] += o > own[u] ? 1 : 0;
  }
#pragma unroll
  for (int u = 0; u < OWN; ++u)
    if (s0 + lane + 64 * u < s1 && rank[u] < k) c_k[wid * k + rank[u]] = own[u];
  __syncthreads();
  const int nc = 16 * k;
  if (tid < nc && c_k[tid] != 0ull) {
    const unsigned long long mine = c_k[tid];
    int r = 0;
    for (int j = 0; j < nc; ++j) r += c_k[j] > mine ? 1 : 0;
    if (r < k) {
      const int i = (int)(0xffffffffu - (uint32_t)mine);
      out_idx[r] = i;
      if (out_val) {
        const float v = score[i];
        out_val[r] = v != v ? -INFINITY : v;
      }
    }
  }
}

extern "C" int psg_topk(psg_ctx* ctx, const float* score, int n, int k, int32_t* out_idx, float* out_val,
                        void* stream) {
  PSG_REQUIRE(ctx && score && out_idx, PSG_ERR_INVALID, "psg_topk: NULL argument");
  PSG_REQUIRE(n > 0 && k > 0, PSG_ERR_INVALID, "psg_topk: n=%d k=%d", n, k);
  const int own = ((n + 15) / 16 + 63) / 64;                       // elements per lane of a segment's wave
  if (own <= 3 && k <= TOPK_KMAX) {                               // n <= 3072: the n^2 / 16 comparisons of ONE workgroup beat k rounds
    hipStream_t st = (hipStream_t)stream;                         // (2500 pairs: 23 vs 52 us; 10 000 pairs: 141 vs 80 us, so not there)
    if (own <= 1) topk_rank_kernel<1><<<1, 1024, 0, st>>>(score, n, k, out_idx, out_val);
    else if (own <= 2) topk_rank_kernel<2><<<1, 1024, 0, st>>>(score, n, k, out_idx, out_val);
    else topk_rank_kernel<3><<<1, 1024, 0, st>>>(score, n, k, out_idx, out_val);
  } else
    topk_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(score, n, k, out_idx, out_val);
  PSG_CHECK_LAUNCH("psg_topk");
  return PSG_OK;
}

// ---------------------------------------------------------------------------------------------
// Rows of the SELECTED pairs for the second phase of the last Q-Former layer (V4:215, 235-237: pair_feature of the chosen
// pairs only): one launch instead of the index arithmetic (arange / mul / add / where / clamp / cat / index_select: ~20
// library launches of 5 us) that depended on the selection.  sel[s] is a global pair id; a pair this chunk owns
// (first <= id < first + count) sits at position id - first + slot_off of the pass, any other slot is computed as the
// chunk's own first pair (a valid pair of the same image) and flagged in mine[s] = 0.
//   out rows [0, K*nq): query rows q of slot s = xq[pos * nq + q];  rows [K*nq, K*(nq+T)): text rows, block
//   text_index[pos] (or pos) of xt;  mask_out[s][t] = text_mask[pos][t];  pair_out[s] = pair_index[pos].
// ---------------------------------------------------------------------------------------------
template <typename T_>
__global__ void gather_pair_rows_kernel(const T_* __restrict__ xq, const T_* __restrict__ xt,
                                        const int32_t* __restrict__ text_index, const uint8_t* __restrict__ text_mask,
                                        const int32_t* __restrict__ pair_index, const int32_t* __restrict__ sel, int K,
                                        int first, int count, int slot_off, int nq, int T, int cols,
                                        T_* __restrict__ out, uint8_t* __restrict__ mask_out,
                                        int32_t* __restrict__ pair_out, uint8_t* __restrict__ mine_out) {
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int64_t nrow = (int64_t)K * (nq + T);
  if (row >= nrow) return;
  const bool is_q = row < (int64_t)K * nq;
  const int s = is_q ? (int)(row / nq) : (int)((row - (int64_t)K * nq) / T);
  const int r = is_q ? (int)(row % nq) : (int)((row - (int64_t)K * nq) % T);
  const int id = sel[s];
  const bool mine = id >= first && id < first + count;
  const int pos = (mine ? id - first : 0) + slot_off;
  const T_* src;
  if (is_q) {
    src = xq + ((int64_t)pos * nq + r) * cols;
    if (r == 0 && lane == 0) {
      if (pair_out) pair_out[s] = pair_index[pos];
      if (mine_out) mine_out[s] = mine ? 1 : 0;
    }
  } else {
    const int blk = text_index ? text_index[pos] : pos;
    src = xt + ((int64_t)blk * T + r) * cols;
    if (lane == 0 && mask_out) mask_out[(int64_t)s * T + r] = text_mask[(int64_t)pos * T + r];
  }
  for (int c = lane * 4; c < cols; c += 256) {
    float v[4];
    Act<T_>::ld4(src, c, v);
    Act<T_>::st4(out, row * cols + c, v);
  }
}

extern "C" int psg_gather_pair_rows(psg_ctx* ctx, const void* xq, const void* xt, const int32_t* text_index,
                                    const uint8_t* text_mask, const int32_t* pair_index, const int32_t* sel, int K,
                                    int first, int count, int slot_off, int nq, int Tt, int cols, void* out,
                                    uint8_t* mask_out, int32_t* pair_out, uint8_t* mine_out, int dtype, void* stream) {
  PSG_REQUIRE(ctx && xq && sel && out && (xt || Tt == 0) && (text_mask || !mask_out) && (pair_index || !pair_out),
              PSG_ERR_INVALID, "psg_gather_pair_rows: NULL argument");
  PSG_REQUIRE(K >= 0 && nq > 0 && Tt >= 0 && cols > 0 && cols % 4 == 0 && count > 0 && slot_off >= 0, PSG_ERR_INVALID,
              "psg_gather_pair_rows: K=%d nq=%d T=%d cols=%d count=%d", K, nq, Tt, cols, count);
  if (K == 0) return PSG_OK;
  const int64_t rows = (int64_t)K * (nq + Tt);
  PSG_DISPATCH_DTYPE(dtype, "psg_gather_pair_rows",
                     (gather_pair_rows_kernel<T><<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(
                         (const T*)xq, (const T*)xt, text_index, text_mask, pair_index, sel, K, first, count, slot_off, nq,
                         Tt, cols, (T*)out, mask_out, pair_out, mine_out)));
  PSG_CHECK_LAUNCH("psg_gather_pair_rows");
  return PSG_OK;
}

// ---------------------------------------------------------------------------------------------
// row gather: dst[r][:] = src[idx[r]][:]  (idx < 0 -> zeros); one wave per row, 4 elements / lane
// ---------------------------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ void gather_rows_kernel(const TS* __restrict__ src, const int32_t* __restrict__ idx, int64_t n, int cols,
                                   int64_t sstride, TD* __restrict__ dst, int64_t dstride) {
  int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (row >= n) return;
  int s = idx[row];
  for (int c = lane * 4; c < cols; c += 256) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (s >= 0) Act<TS>::ld4(src, (int64_t)s * sstride + c, v);
    Act<TD>::st4(dst, row * dstride + c, v);
  }
}

extern "C" int psg_gather_rows(psg_ctx* ctx, const void* src, int src_dtype, const int32_t* idx, int64_t n, int cols,
                               int64_t src_row_stride, void* dst, int dst_dtype, int64_t dst_row_stride,
                               void* stream) {
  PSG_REQUIRE(ctx && src && idx && dst, PSG_ERR_INVALID, "psg_gather_rows: NULL argument");
  PSG_REQUIRE(n >= 0 && cols > 0 && cols % 4 == 0 && src_row_stride % 4 == 0 && dst_row_stride % 4 == 0,
              PSG_ERR_INVALID, "psg_gather_rows: cols/strides must be multiples of 4 (cols=%d)", cols);
  if (n == 0) return PSG_OK;
  dim3 grid((unsigned)((n + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
#define GR(TS, TD) \
  gather_rows_kernel<TS, TD><<<grid, 256, 0, st>>>((const TS*)src, idx, n, cols, src_row_stride, (TD*)dst, dst_row_stride)
  if (src_dtype == PSG_F32 && dst_dtype == PSG_F32) GR(float, float);
  else if (src_dtype == PSG_F32 && dst_dtype == PSG_BF16) GR(float, bf16_t);
  else if (src_dtype == PSG_BF16 && dst_dtype == PSG_BF16) GR(bf16_t, bf16_t);
  else if (src_dtype == PSG_BF16 && dst_dtype == PSG_F32) GR(bf16_t, float);
  else if (src_dtype == PSG_F32 && dst_dtype == PSG_F16) GR(float, f16_t);
  else if (src_dtype == PSG_F16 && dst_dtype == PSG_F16) GR(f16_t, f16_t);
  else if (src_dtype == PSG_F16 && dst_dtype == PSG_F32) GR(f16_t, float);
  else {
    psg_set_error("psg_gather_rows: bad dtypes %d -> %d", src_dtype, dst_dtype);
    return PSG_ERR_INVALID;
  }
#undef GR
  PSG_CHECK_LAUNCH("psg_gather_rows");
  return PSG_OK;
}
