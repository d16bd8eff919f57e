// Probe for the split-fp16 projection scheme: operand lane maps of v_mfma_f32_32x32x16_f16,
// whether fp16 subnormal operands survive, and the error of the 3-product hi/lo split against fp64 next to the error of
// the exact-fp32 MFMA chain.  Diagnostic only; build: hipcc --offload-arch=gfx950 -O2 tools/mfma_f16_probe.hip -o /tmp/probe
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// D[32][32] = A[32][K] . B[K][32], K = 16 * steps, operands given as fp16 planes; one wave.
__global__ void mfma_f16(const _Float16* A, const _Float16* B, float* Dout, int K) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int s = 0; s < K / 16; ++s) {
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) {
      a[j] = A[r * K + 16 * s + 8 * h + j];
      b[j] = B[(16 * s + 8 * h + j) * 32 + r];
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) Dout[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

// the 3-product split: A = Ah + Al, B = Bh + Bl (fp16 each); D = Ah.Bh + Ah.Bl + Al.Bh in ONE fp32 accumulator
__global__ void mfma_split(const float* A, const float* B, float* Dout, int K, float bscale) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int s = 0; s < K / 16; ++s) {
    f16x8 ah, al, bh, bl;
    for (int j = 0; j < 8; ++j) {
      const float x = A[r * K + 16 * s + 8 * h + j];
      const _Float16 xh = (_Float16)x;
      ah[j] = xh;
      al[j] = (_Float16)(x - (float)xh);
      const float w = B[(16 * s + 8 * h + j) * 32 + r] * bscale;
      const _Float16 wh = (_Float16)w;
      bh[j] = wh;
      bl[j] = (_Float16)(w - (float)wh);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
  }
  const float inv = 1.0f / bscale;
  for (int i = 0; i < 16; ++i) Dout[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i] * inv;
}

__global__ void mfma_f32(const float* A, const float* B, float* Dout, int K) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], B[(k + h) * 32 + r], acc, 0, 0, 0);
  for (int i = 0; i < 16; ++i) Dout[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); return 1; } } while (0)

int main() {
  const int K = 128;
  std::vector<_Float16> hA(32 * K), hB(K * 32);
  std::vector<float> fA(32 * K), fB(K * 32), D(32 * 32), D2(32 * 32);
  void *dA, *dB, *dD, *dfA, *dfB;
  CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dD, D.size() * 4));
  CK(hipMalloc(&dfA, fA.size() * 4)); CK(hipMalloc(&dfB, fB.size() * 4));
  // 1. lane map with small integers (exact), asymmetric operands
  for (int i = 0; i < 32; ++i) for (int k = 0; k < K; ++k) hA[i * K + k] = (_Float16)(float)((i * 7 + k * 3) % 11 - 5);
  for (int k = 0; k < K; ++k) for (int j = 0; j < 32; ++j) hB[k * 32 + j] = (_Float16)(float)((k * 5 + j * 13) % 7 - 3);
  CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(mfma_f16, dim3(1), dim3(64), 0, 0, (const _Float16*)dA, (const _Float16*)dB, (float*)dD, K);
  CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double s = 0;
    for (int k = 0; k < K; ++k) s += (double)(float)hA[i * K + k] * (double)(float)hB[k * 32 + j];
    if (s != (double)D[i * 32 + j]) ++bad;
  }
  printf("lane-map check (integers, K=%d): %d mismatches of 1024\n", K, bad);
  // 2. fp16 subnormal operands: a = 2^-20 (subnormal), b = 2^10 -> 16 * 2^-10 per 16-step
  for (auto& v : hA) v = (_Float16)9.5367431640625e-07f;
  for (auto& v : hB) v = (_Float16)1024.0f;
  CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(mfma_f16, dim3(1), dim3(64), 0, 0, (const _Float16*)dA, (const _Float16*)dB, (float*)dD, K);
  CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
  printf("subnormal fp16 operand: D[0][0] = %.9g (kept: %.9g, flushed: 0)\n", D[0], K * 9.5367431640625e-07 * 1024.0);
  // 3. split accuracy: LayerNorm-like rows (N(0,1)) against Glorot-like weights (U(-0.15,0.15)); weights pre-scaled by 2^12
  srand(7);
  auto rnd = []() { return (rand() + 0.5) / (RAND_MAX + 1.0); };
  for (auto& v : fA) v = (float)(std::sqrt(-2.0 * std::log(rnd())) * std::cos(6.283185307179586 * rnd()));
  for (auto& v : fB) v = (float)((rnd() * 2 - 1) * 0.153);
  CK(hipMemcpy(dfA, fA.data(), fA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dfB, fB.data(), fB.size() * 4, hipMemcpyHostToDevice));
  for (float sc : {1.0f, 4096.0f}) {
    hipLaunchKernelGGL(mfma_split, dim3(1), dim3(64), 0, 0, (const float*)dfA, (const float*)dfB, (float*)dD, K, sc);
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(mfma_f32, dim3(1), dim3(64), 0, 0, (const float*)dfA, (const float*)dfB, (float*)dD, K);
    CK(hipMemcpy(D2.data(), dD, D2.size() * 4, hipMemcpyDeviceToHost));
    double e_split = 0, e_f32 = 0, nrm = 0, rms_s = 0, rms_f = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double s = 0, sa = 0;
      for (int k = 0; k < K; ++k) { const double p = (double)fA[i * K + k] * (double)fB[k * 32 + j]; s += p; sa += std::fabs(p); }
      e_split = std::fmax(e_split, std::fabs(D[i * 32 + j] - s) / sa);
      e_f32 = std::fmax(e_f32, std::fabs(D2[i * 32 + j] - s) / sa);
      rms_s += (D[i * 32 + j] - s) * (D[i * 32 + j] - s); rms_f += (D2[i * 32 + j] - s) * (D2[i * 32 + j] - s); nrm += s * s;
    }
    printf("weight scale %g: split-fp16 (3 products) max err / sum|ab| = %.3g (rms rel %.3g)   exact-fp32 MFMA chain: %.3g (rms rel %.3g)\n",
           sc, e_split, std::sqrt(rms_s / nrm), e_f32, std::sqrt(rms_f / nrm));
  }
  return 0;
}
