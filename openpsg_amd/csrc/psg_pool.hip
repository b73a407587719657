// masked-mean object pooling of the v1-v3 detectors
// (kings_sgg/models/detectors/openseed_relation.py:453-468):
//
//   emb[n][c] = sum_{y,x} feat[c][y][x] * m_n[y][x] / (sum_{y,x} m_n[y][x] + 1e-8)
//
// where m_n is object n's mask resampled to the feature resolution by nearest -> zero-pad -> nearest
// (the index chain of V4:416-423, but the zero padding applies to the MASK: padding is no object).
// The reference multiplies the whole [C,Hf,Wf] map by each of the N masks (N x 67 MB of traffic).
// Panoptic masks are disjoint, so every feature element belongs to at most one object: here each is
// read exactly once.
//   pass 0 (pool_slot):    resampled object slot per feature pixel;
//   pass 1 (pool_index):   one workgroup per object compacts its pixel indices in ascending order
//                          into its own list region and counts them: deterministic, no atomics;
//   pass 2a (pool_partial): one wave per (object, 2048-pixel segment, 4 channels): 16 independent gathers
//                          in flight per lane (rectangular objects give contiguous runs);
//   pass 2b (pool_final):  segment partials summed in segment order, divided by (count + 1e-8).
#include "psg_common.h"

#define PL_SEG 2048

__global__ void pool_slot_kernel(const int32_t* __restrict__ pan, int H0, int W0, int img_h, int img_w, int pad_h,
                                 int pad_w, int Hf, int Wf, const int32_t* __restrict__ ids, int N,
                                 int32_t* __restrict__ slot) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Hf * Wf) return;
  const int r = idx / Wf, c = idx % Wf;
  const float sy2 = (float)pad_h / (float)Hf, sx2 = (float)pad_w / (float)Wf;
  const float sy1 = (float)H0 / (float)img_h, sx1 = (float)W0 / (float)img_w;
  const int y1 = min((int)floorf((float)r * sy2), pad_h - 1);
  const int x1 = min((int)floorf((float)c * sx2), pad_w - 1);
  int sl = -1;
  if (y1 < img_h && x1 < img_w) {
    const int y0 = min((int)floorf((float)y1 * sy1), H0 - 1);
    const int x0 = min((int)floorf((float)x1 * sx1), W0 - 1);
    const int id = pan[(int64_t)y0 * W0 + x0];
    for (int n = 0; n < N; ++n)
      if (ids[n] == id) {                                      // ids are distinct in a panoptic result
        sl = n;
        break;
      }
  }
  slot[idx] = sl;
}

__global__ void __launch_bounds__(1024) pool_index_kernel(const int32_t* __restrict__ slot, int HW,
                                                          int32_t* __restrict__ cnt, int32_t* __restrict__ list) {
  __shared__ int s_wave[16];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int32_t* mylist = list + (int64_t)n * HW;
  int base = 0;
  for (int i0 = 0; i0 < HW; i0 += 1024) {
    const int i = i0 + tid;
    const bool hit = i < HW && slot[i] == n;
    const unsigned long long m = __ballot(hit);
    if (lane == 0) s_wave[wid] = __popcll(m);
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int v = s_wave[w];
      woff += w < wid ? v : 0;
      total += v;
    }
    if (hit) mylist[base + woff + __popcll(m & ((1ull << lane) - 1ull))] = i;
    base += total;
    __syncthreads();
  }
  if (tid == 0) cnt[n] = base;
}

__global__ void __launch_bounds__(256) pool_partial_kernel(const float* __restrict__ feat, int C, int HW, int maxseg,
                                                           const int32_t* __restrict__ cnt,
                                                           const int32_t* __restrict__ list, int N,
                                                           float* __restrict__ partial) {
  const int64_t unit = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int cq_n = C / 4;
  if (unit >= (int64_t)N * maxseg * cq_n) return;
  const int cq = (int)(unit % cq_n);
  const int seg = (int)((unit / cq_n) % maxseg);
  const int n = (int)(unit / ((int64_t)cq_n * maxseg));
  const int m = cnt[n];
  const int p0 = seg * PL_SEG;
  if (p0 >= m) return;                                          // wave-uniform
  const int p1 = min(m, p0 + PL_SEG);
  const int32_t* lp = list + (int64_t)n * HW;
  const float* f0 = feat + (int64_t)(4 * cq) * HW;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  int i = p0 + lane;
  for (; i + 192 < p1; i += 256) {
    int px[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) px[u] = lp[i + 64 * u];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) a[ch] += f0[(int64_t)ch * HW + px[u]];
  }
  for (; i < p1; i += 64) {
    const int px = lp[i];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) a[ch] += f0[(int64_t)ch * HW + px];
  }
#pragma unroll
  for (int ch = 0; ch < 4; ++ch) {
    float sv = a[ch];
    for (int o = 32; o > 0; o >>= 1) sv += __shfl_xor(sv, o, 64);
    if (lane == 0) partial[((int64_t)n * maxseg + seg) * C + 4 * cq + ch] = sv;
  }
}

__global__ void pool_final_kernel(const float* __restrict__ partial, int C, int maxseg, const int32_t* __restrict__ cnt,
                                  int N, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * C) return;
  const int n = idx / C, c = idx % C;
  const int m = cnt[n];
  const int nseg = (m + PL_SEG - 1) / PL_SEG;
  float sv = 0.f;
  for (int k = 0; k < nseg; ++k) sv += partial[((int64_t)n * maxseg + k) * C + c];
  out[idx] = sv / ((float)m + 1e-8f);
}

static int64_t pool_ws_ints(int HW, int N, int C) {
  const int maxseg = (HW + PL_SEG - 1) / PL_SEG;
  return (int64_t)N + HW + (int64_t)N * HW + (int64_t)N * maxseg * C;   // counts | slot map | lists | partials (fp32)
}

extern "C" int psg_masked_mean_pool_workspace(psg_ctx* ctx, int C, int Hf, int Wf, int N, int64_t* bytes) {
  PSG_REQUIRE(ctx && bytes && C > 0 && Hf > 0 && Wf > 0 && N > 0, PSG_ERR_INVALID,
              "psg_masked_mean_pool_workspace: bad argument");
  *bytes = pool_ws_ints(Hf * Wf, N, C) * (int64_t)sizeof(int32_t);
  return PSG_OK;
}

extern "C" int psg_masked_mean_pool(psg_ctx* ctx, const float* feat, int C, int Hf, int Wf, const int32_t* pan, int H0,
                                    int W0, int img_h, int img_w, int pad_h, int pad_w, const int32_t* object_ids,
                                    int N, float* out, int32_t* workspace, int64_t workspace_bytes, void* stream) {
  PSG_REQUIRE(ctx && feat && pan && object_ids && out && workspace, PSG_ERR_INVALID,
              "psg_masked_mean_pool: NULL argument");
  PSG_REQUIRE(C > 0 && C % 4 == 0 && Hf > 0 && Wf > 0 && N > 0 && H0 > 0 && W0 > 0 && img_h > 0 && img_w > 0 &&
                  pad_h >= img_h && pad_w >= img_w,
              PSG_ERR_INVALID, "psg_masked_mean_pool: C=%d (multiple of 4) Hf=%d Wf=%d N=%d", C, Hf, Wf, N);
  const int HW = Hf * Wf;
  PSG_REQUIRE(workspace_bytes >= pool_ws_ints(HW, N, C) * (int64_t)sizeof(int32_t), PSG_ERR_INVALID,
              "psg_masked_mean_pool: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int maxseg = (HW + PL_SEG - 1) / PL_SEG;
  int32_t* cnt = workspace;
  int32_t* slot = workspace + N;
  int32_t* list = slot + HW;
  float* partial = reinterpret_cast<float*>(list + (int64_t)N * HW);
  pool_slot_kernel<<<(HW + 255) / 256, 256, 0, st>>>(pan, H0, W0, img_h, img_w, pad_h, pad_w, Hf, Wf, object_ids, N,
                                                    slot);
  PSG_CHECK_LAUNCH("psg_masked_mean_pool(slot)");
  pool_index_kernel<<<N, 1024, 0, st>>>(slot, HW, cnt, list);
  PSG_CHECK_LAUNCH("psg_masked_mean_pool(index)");
  const int64_t waves = (int64_t)N * maxseg * (C / 4);
  pool_partial_kernel<<<(unsigned)((waves + 3) / 4), 256, 0, st>>>(feat, C, HW, maxseg, cnt, list, N, partial);
  PSG_CHECK_LAUNCH("psg_masked_mean_pool(partial)");
  pool_final_kernel<<<(N * C + 255) / 256, 256, 0, st>>>(partial, C, maxseg, cnt, N, out);
  PSG_CHECK_LAUNCH("psg_masked_mean_pool(final)");
  return PSG_OK;
}

// Split-mean variant (`_mask_pooling(output_size > 1)`, openseed_relation.py:175-200): the object's masked pixels, in
// row-major order, are cut into `k` contiguous chunks (the first m mod k one pixel longer) and each chunk is averaged:
// out[n][c] = mean of chunk c.  An object with fewer pixels than chunks repeats its pixel list (chunk c = pixel c mod m),
// an object without pixels gives zeros.  Same slot / index passes; one wave per (object, chunk, 4 channels) sums its
// chunk in list order - deterministic.
__global__ void __launch_bounds__(256) pool_split_kernel(const float* __restrict__ feat, int C, int HW,
                                                         const int32_t* __restrict__ cnt, const int32_t* __restrict__ list,
                                                         int N, int k, float* __restrict__ out) {
  const int64_t unit = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int cq_n = C / 4;
  if (unit >= (int64_t)N * k * cq_n) return;
  const int cq = (int)(unit % cq_n);
  const int ch_i = (int)((unit / cq_n) % k);
  const int n = (int)(unit / ((int64_t)cq_n * k));
  const int m = cnt[n];
  float* o = out + ((int64_t)n * k + ch_i) * C + 4 * cq;
  if (m <= 0) {                                                 // wave-uniform
    if (lane < 4) o[lane] = 0.f;
    return;
  }
  int start, size;
  if (m < k) {                                                  // the pixel list repeated up to k entries, one per chunk
    start = ch_i % m;
    size = 1;
  } else {
    const int base = m / k, rem = m % k;
    start = ch_i * base + min(ch_i, rem);
    size = base + (ch_i < rem ? 1 : 0);
  }
  const int32_t* lp = list + (int64_t)n * HW + start;
  const float* f0 = feat + (int64_t)(4 * cq) * HW;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = lane; i < size; i += 64) {
    const int px = lp[i];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) a[ch] += f0[(int64_t)ch * HW + px];
  }
#pragma unroll
  for (int ch = 0; ch < 4; ++ch) {
    const float sv = wave_sum(a[ch]);
    if (lane == 0) o[ch] = sv / (float)size;
  }
}

extern "C" int psg_masked_split_mean_pool(psg_ctx* ctx, const float* feat, int C, int Hf, int Wf, const int32_t* pan,
                                          int H0, int W0, int img_h, int img_w, int pad_h, int pad_w,
                                          const int32_t* object_ids, int N, int output_size, float* out,
                                          int32_t* workspace, int64_t workspace_bytes, void* stream) {
  PSG_REQUIRE(ctx && feat && pan && object_ids && out && workspace, PSG_ERR_INVALID,
              "psg_masked_split_mean_pool: NULL argument");
  PSG_REQUIRE(C > 0 && C % 4 == 0 && Hf > 0 && Wf > 0 && N > 0 && H0 > 0 && W0 > 0 && img_h > 0 && img_w > 0 &&
                  pad_h >= img_h && pad_w >= img_w && output_size >= 1,
              PSG_ERR_INVALID, "psg_masked_split_mean_pool: C=%d (multiple of 4) Hf=%d Wf=%d N=%d output_size=%d", C, Hf,
              Wf, N, output_size);
  const int HW = Hf * Wf;
  PSG_REQUIRE(workspace_bytes >= pool_ws_ints(HW, N, C) * (int64_t)sizeof(int32_t), PSG_ERR_INVALID,
              "psg_masked_split_mean_pool: workspace too small (psg_masked_mean_pool_workspace)");
  hipStream_t st = (hipStream_t)stream;
  int32_t* cnt = workspace;
  int32_t* slot = workspace + N;
  int32_t* list = slot + HW;
  pool_slot_kernel<<<(HW + 255) / 256, 256, 0, st>>>(pan, H0, W0, img_h, img_w, pad_h, pad_w, Hf, Wf, object_ids, N,
                                                    slot);
  PSG_CHECK_LAUNCH("psg_masked_split_mean_pool(slot)");
  pool_index_kernel<<<N, 1024, 0, st>>>(slot, HW, cnt, list);
  PSG_CHECK_LAUNCH("psg_masked_split_mean_pool(index)");
  const int64_t waves = (int64_t)N * output_size * (C / 4);
  pool_split_kernel<<<(unsigned)((waves + 3) / 4), 256, 0, st>>>(feat, C, HW, cnt, list, N, output_size, out);
  PSG_CHECK_LAUNCH("psg_masked_split_mean_pool(split)");
  return PSG_OK;
}

// ---------------------------------------------------------------------------------------------
// SURVEY 8f rank 4, second half: the bilinear relation scorer of the closed-set heads
// (kings_sgg/models/relation_heads/relation_transformer_head_v2.py:204-209, same form in v1 / v3):
//
//   sub = sub_pred(obj_emb).reshape(B, N, R, C).permute(0, 2, 1, 3)       [B, R, N, C]
//   obj = obj_pred(obj_emb).reshape(B, N, R, C).permute(0, 2, 1, 3)
//   pred = einsum('nrsc,nroc->nrso', sub, obj)                            [B, R, N, N]
//
// i.e. R independent N x N x C products per image.  The kernel reads the Linear outputs in their natural
// [B][N][R*C] layout (the permute is index arithmetic) and uses the exact-f32 matrix cores
// (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain over c), one wave per 32 x 32 output tile, operands staged
// through LDS in 32-column chunks so that global reads are 128-byte row pieces.
// ---------------------------------------------------------------------------------------------
typedef float bl_f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(64) bilinear_scores_kernel(const float* __restrict__ sub,
                                                             const float* __restrict__ obj, int B, int N, int R,
                                                             int C, float* __restrict__ pred) {
  __shared__ float s_a[32][33], s_b[32][33];
  const int lane = threadIdx.x;
  const int nt = (N + 31) >> 5;
  int u = blockIdx.x;
  const int ot = u % nt; u /= nt;
  const int st = u % nt; u /= nt;
  const int r = u % R;
  const int b = u / R;
  const int64_t ld = (int64_t)R * C;                                // row stride of the [B][N][R*C] inputs
  const float* sp = sub + ((int64_t)b * N) * ld + (int64_t)r * C;
  const float* op = obj + ((int64_t)b * N) * ld + (int64_t)r * C;
  bl_f32x16 acc = {0};
  const int i = lane & 31, kh = lane >> 5;
  for (int c0 = 0; c0 < C; c0 += 32) {
    // stage [32 rows][32 c] of both operands: lane -> (row = lane>>3 + 8 j, 4 floats at c0 + 4 (lane&7))
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (lane >> 3) + 8 * j, cc = 4 * (lane & 7);
      const int s = st * 32 + row, o = ot * 32 + row;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = c0 + cc + e;
        s_a[row][cc + e] = (s < N && c < C) ? sp[(int64_t)s * ld + c] : 0.f;
        s_b[row][cc + e] = (o < N && c < C) ? op[(int64_t)o * ld + c] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; k += 2)                                  // A[i][k + kh], B[k + kh][j = i]
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s_a[i][k + kh], s_b[i][k + kh], acc, 0, 0, 0);
    __syncthreads();
  }
  // D: col = lane & 31 (o), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (s)
  float* pp = pred + (((int64_t)b * R + r) * N) * N;
  const int o = ot * 32 + i;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int s = st * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
    if (s < N && o < N) pp[(int64_t)s * N + o] = acc[reg];
  }
}

extern "C" int psg_bilinear_scores(psg_ctx* ctx, const float* sub, const float* obj, int B, int N, int R, int C,
                                   float* pred, void* stream) {
  PSG_REQUIRE(ctx && sub && obj && pred, PSG_ERR_INVALID, "psg_bilinear_scores: NULL argument");
  PSG_REQUIRE(B > 0 && N > 0 && R > 0 && C > 0, PSG_ERR_INVALID, "psg_bilinear_scores: B=%d N=%d R=%d C=%d", B, N, R, C);
  const int nt = (N + 31) / 32;
  const int64_t units = (int64_t)B * R * nt * nt;
  PSG_REQUIRE(units < (1ll << 31), PSG_ERR_UNSUPPORTED, "psg_bilinear_scores: %lld tiles", (long long)units);
  bilinear_scores_kernel<<<(unsigned)units, 64, 0, (hipStream_t)stream>>>(sub, obj, B, N, R, C, pred);
  PSG_CHECK_LAUNCH("psg_bilinear_scores");
  return PSG_OK;
}
