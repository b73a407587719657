// Standalone probe (no torch in the process): for the fp32s prompt-pass products (fp16 operands, fp32 result, row-major
// x [M][K'] . w [N][K']^T) ask hipBLASLt for up to 128 candidate algorithms, time each, print the heuristic's first pick
// against the fastest.   hipcc -O2 probe.cpp -o probe -lhipblaslt ;  ./probe [M]
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_rand(unsigned short* p, long long n, unsigned seed, float scale) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned h = (unsigned)(i * 2654435761u) ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    const float u = ((h & 0xffffff) / 16777216.0f - 0.5f) * 2.0f * scale;      // uniform in (-scale, scale)
    p[i] = __builtin_bit_cast(unsigned short, (_Float16)u);
  }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 960;
  const int kmul = argc > 2 ? atoi(argv[2]) : 3;
  struct Sh { const char* name; int N, K; } shapes[] = {{"qkv", 12288, 4096}, {"o", 4096, 4096}, {"gate_up", 22016, 4096},
                                                        {"gate", 11008, 4096}, {"down", 4096, 11008}};
  hipblasLtHandle_t h;
  CK(hipblasLtCreate(&h));
  size_t ws_bytes = 256u << 20;
  void* ws;
  CK(hipMalloc(&ws, ws_bytes));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  for (auto& s : shapes) {
    const int64_t N = s.N, K = (int64_t)s.K * kmul;
    void *x, *w[2], *y;
    CK(hipMalloc(&x, M * K * 2));
    CK(hipMalloc(&w[0], N * K * 2));
    CK(hipMalloc(&w[1], N * K * 2));
    CK(hipMalloc(&y, (int64_t)M * N * 4));
    fill_rand<<<2048, 256>>>((unsigned short*)x, (long long)M * K, 1u, 1.0f);
    fill_rand<<<2048, 256>>>((unsigned short*)w[0], N * K, 2u, 0.02f);
    fill_rand<<<2048, 256>>>((unsigned short*)w[1], N * K, 3u, 0.02f);
    CK(hipDeviceSynchronize());
    // row-major y[M][N] = x[M][K] w[N][K]^T  ==  column-major y^T[N][M] = w^T-stored... : A = w (K x N col-major, op T), B = x (K x M, op N)
    hipblasLtMatmulDesc_t desc;
    CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
    CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
    CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
    hipblasLtMatrixLayout_t la, lb, lc;
    CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16F, K, N, K));       // w: K x N column-major (ld K) -> op T gives N x K
    CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16F, K, M, K));       // x: K x M column-major
    CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_32F, N, M, N));       // y^T: N x M column-major = y row-major
    hipblasLtMatmulPreference_t pref;
    CK(hipblasLtMatmulPreferenceCreate(&pref));
    CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof(ws_bytes)));
    std::vector<hipblasLtMatmulHeuristicResult_t> res(128);
    int got = 0;
    CK(hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, lc, lc, pref, 128, res.data(), &got));
    const float alpha = 1.f, beta = 0.f;
    std::vector<std::pair<float, int>> times;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int a = 0; a < got; ++a) {
      bool ok = true;
      for (int i = 0; i < 2 && ok; ++i)
        ok = hipblasLtMatmul(h, desc, &alpha, w[i & 1], la, x, lb, &beta, y, lc, y, lc, &res[a].algo, ws, ws_bytes, st) == 0;
      if (!ok) continue;
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < 6; ++i)
        hipblasLtMatmul(h, desc, &alpha, w[i & 1], la, x, lb, &beta, y, lc, y, lc, &res[a].algo, ws, ws_bytes, st);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      times.push_back({ms / 6 * 1e3f, a});
    }
    if (times.empty()) { printf("%s: no algorithm\n", s.name); continue; }
    const float first = times[0].first;
    auto srt = times;
    std::sort(srt.begin(), srt.end());
    const double fl = 2.0 * M * N * K;
    printf("%-8s M=%d N=%lld K'=%lld: %d algos; heuristic first %.1f us (%.0f TF/s); best #%d %.1f us (%.0f TF/s); 2nd %.1f 3rd %.1f\n",
           s.name, M, (long long)N, (long long)K, got, first, fl / first / 1e6, srt[0].second, srt[0].first, fl / srt[0].first / 1e6,
           srt.size() > 1 ? srt[1].first : 0.f, srt.size() > 2 ? srt[2].first : 0.f);
    fflush(stdout);
    hipFree(x); hipFree(w[0]); hipFree(w[1]); hipFree(y);
  }
  return 0;
}
