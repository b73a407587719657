// K1: patch embedding (timm PatchEmbed = Conv2d(C, C_out, k=16, s=16) + flatten + transpose, V4:410)
// as an exact-fp32 split-K GEMM on the f32 matrix cores.
//
//   out[l][o] = bias[o] + sum_k A[l][k] * W[o][k],   k = c*256 + dy*16 + dx,
//   A[l][k] = feat[c][16 py + dy][16 px + dx],  l = py * gw + px           (K = C*256 = 65536)
//
// The library route (MIOpen -> im2col + a 128x128-tiled fp32 GEMM) runs this M = N = 256, K = 65536
// problem on FOUR workgroups (3.5 ms on MI355X).  It is 8.6 GFLOP over 134 MB of inputs: HBM-side
// it is worth ~25 us, on the f32 MFMA pipe (157 TF peak) ~60 us.  So: split K over the whole chip.
//
//   * grid = (ceil(L/128), C_out/128, S): a workgroup owns a 128 x 128 output tile and a K slice
//     of C/S channels; 4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 tiles of v_mfma_f32_32x32x2_f32
//     (exact fp32: bitwise an fmaf chain in k order - this kernel is used in both dtype modes);
//   * K is walked in chunks of 32 (= 2 dy rows of one channel): the A chunk is read straight from
//     the NCHW feature map (16 contiguous floats per patch, adjacent patches adjacent in memory),
//     the B chunk is 128 full 128-byte lines of W; both are prefetched into registers while the
//     previous chunk is multiplied out of LDS (pitch 33 floats: conflict-free ds_read_b32);
//   * partial tiles go to part[S][Lpad][C_out] and a second tiny kernel sums them in split order
//     and adds the bias (deterministic; the 2 x 16 MB of partial traffic stays in the MALL).
#include "psg_common.h"

typedef float pe_f32x16 __attribute__((ext_vector_type(16)));

#define PE_KC 32
#define PE_PITCH 33

__global__ void __launch_bounds__(256, 2)
patch_embed_kernel(const float* __restrict__ feat, const float* __restrict__ w, float* __restrict__ part, int C, int Hf,
                   int Wf, int gw, int L, int Cout, int ch_per_split) {
  __shared__ float a_lds[128 * PE_PITCH];
  __shared__ float b_lds[128 * PE_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 128, sp = blockIdx.z;
  const int c_begin = sp * ch_per_split, c_end = min(C, c_begin + ch_per_split);
  const int nchunk = (c_end - c_begin) * 8;                     // 8 chunks (2 dy rows each) per channel
  const int64_t Kfull = (int64_t)C * 256;
  const int wr = wid >> 1, wc = wid & 1;                        // wave -> 64 x 64 sub-tile

  // loader geometry: element e = tid + 256 r (r = 0..3) -> row i = e / 8 (0..127), quad q = e % 8
  int a_off[4];                                                 // feature offset of (patch i, quad q) within a channel plane
  bool a_ok[4];
  int64_t b_off[4];
  int lds_off[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int e = tid + 256 * r;
    const int i = e >> 3, q = e & 7;
    const int l = m0 + i;
    a_ok[r] = l < L;
    const int py = a_ok[r] ? l / gw : 0, px = a_ok[r] ? l % gw : 0;
    a_off[r] = (16 * py + (q >> 2)) * Wf + 16 * px + (q & 3) * 4;
    b_off[r] = (int64_t)(n0 + i) * Kfull + q * 4;
    lds_off[r] = i * PE_PITCH + q * 4;
  }
  float4 ra[4], rb[4];
  auto load_chunk = [&](int ch) {                               // ch -> channel c = c_begin + ch/8, dy0 = 2 (ch % 8)
    const int c = c_begin + (ch >> 3), dy0 = (ch & 7) * 2;
    const float* fa = feat + (int64_t)c * Hf * Wf + dy0 * Wf;
    const float* fb = w + (int64_t)c * 256 + dy0 * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ra[r] = a_ok[r] ? *reinterpret_cast<const float4*>(fa + a_off[r]) : make_float4(0.f, 0.f, 0.f, 0.f);
      rb[r] = *reinterpret_cast<const float4*>(fb + b_off[r]);
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* pa = a_lds + lds_off[r];
      float* pb = b_lds + lds_off[r];
      pa[0] = ra[r].x; pa[1] = ra[r].y; pa[2] = ra[r].z; pa[3] = ra[r].w;
      pb[0] = rb[r].x; pb[1] = rb[r].y; pb[2] = rb[r].z; pb[3] = rb[r].w;
    }
  };
  pe_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (pe_f32x16){0};

  if (nchunk > 0) load_chunk(0);
  const int l31 = lane & 31, hi = lane >> 5;
  const float* ap = a_lds + (wr * 64 + l31) * PE_PITCH + hi;    // A[i = lane&31][k = 2 s + (lane>>5)]
  const float* bp = b_lds + (wc * 64 + l31) * PE_PITCH + hi;    // B[k][j = lane&31] = W[n0 + j][k]
  for (int ch = 0; ch < nchunk; ++ch) {
    __syncthreads();                                            // previous chunk fully consumed
    store_chunk();
    __syncthreads();
    if (ch + 1 < nchunk) load_chunk(ch + 1);                    // in flight during the MFMAs below
#pragma unroll
    for (int s = 0; s < PE_KC / 2; ++s) {
      const float a0 = ap[2 * s], a1 = ap[32 * PE_PITCH + 2 * s];
      const float b0 = bp[2 * s], b1 = bp[32 * PE_PITCH + 2 * s];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
  // D[row i = (r&3) + 8 (r>>2) + 4 hi][col j = lane&31]
  const int Lpad = gridDim.x * 128;
  float* pp = part + ((int64_t)sp * Lpad + m0 + wr * 64) * Cout + n0 + wc * 64 + l31;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        pp[(int64_t)row * Cout + j * 32] = acc[i][j][r];
      }
}

__global__ void patch_embed_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias, int S,
                                          int Lpad, int L, int Cout, float* __restrict__ out) {
  const int64_t n4 = (int64_t)L * Cout / 4;
  const int64_t slice = (int64_t)Lpad * Cout;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 a = *reinterpret_cast<const float4*>(part + i * 4);
    for (int s = 1; s < S; ++s) {
      const float4 b = *reinterpret_cast<const float4*>(part + s * slice + i * 4);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const int col = (int)((i * 4) % Cout);
    const float4 bb = *reinterpret_cast<const float4*>(bias + col);
    *reinterpret_cast<float4*>(out + i * 4) = make_float4(a.x + bb.x, a.y + bb.y, a.z + bb.z, a.w + bb.w);
  }
}

static int pe_splits(const psg_ctx* ctx, int C, int mt, int nt) {
  // ~2 workgroups per CU; a split owns whole channels
  int S = (2 * ctx->num_cu + mt * nt - 1) / (mt * nt);
  if (S > C) S = C;
  if (S < 1) S = 1;
  const int cps = (C + S - 1) / S;
  return (C + cps - 1) / cps;
}

extern "C" int psg_patch_embed_workspace(psg_ctx* ctx, int C, int Hf, int Wf, int Cout, int patch, int64_t* bytes) {
  PSG_REQUIRE(ctx && bytes, PSG_ERR_INVALID, "psg_patch_embed_workspace: NULL argument");
  PSG_REQUIRE(patch == 16 && Cout % 128 == 0 && C > 0 && Hf >= 16 && Wf >= 16, PSG_ERR_UNSUPPORTED,
              "psg_patch_embed: patch=%d Cout=%d (built for 16x16 patches, Cout %% 128 == 0)", patch, Cout);
  const int L = (Hf / 16) * (Wf / 16);
  const int mt = (L + 127) / 128, nt = Cout / 128;
  *bytes = (int64_t)pe_splits(ctx, C, mt, nt) * mt * 128 * Cout * sizeof(float);
  return PSG_OK;
}

extern "C" int psg_patch_embed(psg_ctx* ctx, const float* feat, int C, int Hf, int Wf, const float* weight,
                               const float* bias, int Cout, int patch, float* out, float* workspace,
                               int64_t workspace_bytes, void* stream) {
  PSG_REQUIRE(ctx && feat && weight && bias && out && workspace, PSG_ERR_INVALID, "psg_patch_embed: NULL argument");
  PSG_REQUIRE(patch == 16 && Cout % 128 == 0 && C > 0 && Hf >= 16 && Wf >= 16 && Wf % 4 == 0, PSG_ERR_UNSUPPORTED,
              "psg_patch_embed: patch=%d Cout=%d Wf=%d (built for 16x16 patches, Cout %% 128 == 0, Wf %% 4 == 0)", patch,
              Cout, Wf);
  const int gh = Hf / 16, gw = Wf / 16, L = gh * gw;
  const int mt = (L + 127) / 128, nt = Cout / 128;
  const int S = pe_splits(ctx, C, mt, nt);
  const int cps = (C + S - 1) / S;
  PSG_REQUIRE(workspace_bytes >= (int64_t)S * mt * 128 * Cout * (int64_t)sizeof(float), PSG_ERR_INVALID,
              "psg_patch_embed: workspace too small (%lld B)", (long long)workspace_bytes);
  hipStream_t st = (hipStream_t)stream;
  patch_embed_kernel<<<dim3(mt, nt, S), 256, 0, st>>>(feat, weight, workspace, C, Hf, Wf, gw, L, Cout, cps);
  PSG_CHECK_LAUNCH("psg_patch_embed");
  const int64_t n4 = (int64_t)L * Cout / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  patch_embed_reduce_kernel<<<blocks, 256, 0, st>>>(workspace, bias, S, mt * 128, L, Cout, out);
  PSG_CHECK_LAUNCH("psg_patch_embed(reduce)");
  return PSG_OK;
}
