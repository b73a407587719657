// Infinity Cache (256 MiB MALL) probe - development aid, not part of the product library.
//   (1) steady-state streaming-read rate of a buffer of S MB read again and again (S below / above the cache size),
//       with default-policy and with non-temporal 16-byte loads;
//   (2) the prefetch question behind DESIGN 4.12: a COLD 64 MB slice read by kernel G (nt loads, as the decode GEMMs
//       stream their weights) with and without a preceding plain-load pass P over the same slice - is a slice that
//       another kernel just pulled through the memory side served faster than from HBM?
// Build: hipcc --offload-arch=gfx950 -O3 -o mall_probe mall_probe.hip ; run: ./mall_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool NT>
__global__ void __launch_bounds__(512) rd(const u32x4* __restrict__ p, size_t n16, unsigned* __restrict__ out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  for (; i + 7 * stride < n16; i += 8 * stride) {
    u32x4 a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= a[u];
  }
  for (; i < n16; i += stride) acc ^= p[i];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

// one 64-byte line per `step` bytes of the slice: warms the address translation (and nothing else worth mentioning)
__global__ void __launch_bounds__(256) touch(const unsigned char* __restrict__ p, size_t bytes, size_t step, unsigned* __restrict__ out) {
  unsigned acc = 0;
  for (size_t o = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * step; o < bytes; o += (size_t)gridDim.x * blockDim.x * step)
    acc ^= *reinterpret_cast<const unsigned*>(p + o);
  if (acc == 0x12345678u) out[0] = 1;
}

int main() {
  const size_t POOL = (size_t)6 << 30;
  unsigned char* buf;
  unsigned* out;
  CK(hipMalloc(&buf, POOL));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 1, POOL));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const int grid = 256 * 2, thr = 512;
  auto pass = [&](bool nt, size_t off, size_t bytes) {
    if (nt) rd<true><<<grid, thr, 0, st>>>((const u32x4*)(buf + off), bytes / 16, out);
    else rd<false><<<grid, thr, 0, st>>>((const u32x4*)(buf + off), bytes / 16, out);
  };
  printf("# (1) the same S MB read 12 times back to back; GB/s of passes 3..12\n");
  for (int nt = 0; nt < 2; ++nt)
    for (size_t mb : {16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 512, 1024, 4096}) {
      const size_t bytes = mb << 20;
      pass(nt, 0, bytes);
      pass(nt, 0, bytes);
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < 10; ++r) pass(nt, 0, bytes);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%s S=%5zu MB: %8.1f us per pass = %7.1f GB/s\n", nt ? "nt     " : "default", mb, ms * 100.f, bytes * 10 / (ms * 1e-3) / 1e9);
    }
  printf("# (2) cold slice (rotating through a 6 GB pool), pass G (nt) alone vs plain pass P then G; us of G\n");
  for (size_t mb : {32, 64, 128, 200}) {
    const size_t bytes = mb << 20;
    for (int mode = 0; mode < 3; ++mode) {             // 0: G alone (cold)  1: P default then G nt  2: P nt then G nt
      float tot = 0, totp = 0;
      const int reps = 12;
      size_t off = 0;
      for (int r = 0; r < reps; ++r) {
        off = (off + ((size_t)512 << 20)) % (POOL - bytes);
        hipEvent_t a, b, c;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&c));
        CK(hipEventRecord(a, st));
        if (mode) pass(mode == 2, off, bytes);
        CK(hipEventRecord(b, st));
        pass(true, off, bytes);
        CK(hipEventRecord(c, st));
        CK(hipEventSynchronize(c));
        float mp, mg;
        CK(hipEventElapsedTime(&mp, a, b));
        CK(hipEventElapsedTime(&mg, b, c));
        if (r >= 2) { tot += mg; totp += mp; }
        CK(hipEventDestroy(a)); CK(hipEventDestroy(b)); CK(hipEventDestroy(c));
      }
      const float g = tot / (reps - 2) * 1e3f, p = totp / (reps - 2) * 1e3f;
      printf("S=%4zu MB %s: P %7.1f us, G %7.1f us = %7.1f GB/s\n", mb,
             mode == 0 ? "G alone (cold)     " : mode == 1 ? "P(default) then G  " : "P(nt) then G       ", p, g, bytes / (g * 1e-6) / 1e9);
    }
  }
  printf("# (3) cold slice: what a preceding TOUCH of one line per step bytes (address translation only) buys pass G; us of G\n");
  for (size_t mb : {64, 200}) {
    const size_t bytes = mb << 20;
    for (size_t step : {(size_t)0, (size_t)4096, (size_t)65536, (size_t)(2 << 20), (size_t)1}) {   // 0: none; 1: P over the first 16 MB only
      float tot = 0, totp = 0;
      const int reps = 12;
      size_t off = 0;
      for (int r = 0; r < reps; ++r) {
        off = (off + ((size_t)512 << 20)) % (POOL - bytes);
        hipEvent_t a, b, c;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&c));
        CK(hipEventRecord(a, st));
        if (step == 1) pass(false, off, (size_t)16 << 20);
        else if (step) touch<<<64, 256, 0, st>>>(buf + off, bytes, step, out);
        CK(hipEventRecord(b, st));
        pass(true, off, bytes);
        CK(hipEventRecord(c, st));
        CK(hipEventSynchronize(c));
        float mp, mg;
        CK(hipEventElapsedTime(&mp, a, b));
        CK(hipEventElapsedTime(&mg, b, c));
        if (r >= 2) { tot += mg; totp += mp; }
        CK(hipEventDestroy(a)); CK(hipEventDestroy(b)); CK(hipEventDestroy(c));
      }
      const float g = tot / (reps - 2) * 1e3f, pp = totp / (reps - 2) * 1e3f;
      printf("S=%4zu MB step %8zu: touch %7.1f us, G %7.1f us = %7.1f GB/s\n", mb, step, pp, g, bytes / (g * 1e-6) / 1e9);
    }
  }
  printf("# (4) G twice over the same cold slice (second pass: translation warm, data > L2): us\n");
  for (size_t mb : {64, 200, 360}) {
    const size_t bytes = mb << 20;
    float t1 = 0, t2 = 0;
    size_t off = 0;
    for (int r = 0; r < 12; ++r) {
      off = (off + ((size_t)512 << 20)) % (POOL - bytes);
      hipEvent_t a, b, c;
      CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&c));
      CK(hipEventRecord(a, st));
      pass(true, off, bytes);
      CK(hipEventRecord(b, st));
      pass(true, off, bytes);
      CK(hipEventRecord(c, st));
      CK(hipEventSynchronize(c));
      float m1, m2;
      CK(hipEventElapsedTime(&m1, a, b));
      CK(hipEventElapsedTime(&m2, b, c));
      if (r >= 2) { t1 += m1; t2 += m2; }
      CK(hipEventDestroy(a)); CK(hipEventDestroy(b)); CK(hipEventDestroy(c));
    }
    printf("S=%4zu MB: first %7.1f us, second %7.1f us\n", mb, t1 * 100.f, t2 * 100.f);
  }
  return 0;
}
