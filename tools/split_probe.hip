// Is the 6-instruction hi / lo fp16 split (v_fma_mixlo_f16 / v_fma_mixhi_f16 write the residual's fp16 straight into a register half) the
// same bits as the 8-instruction form (v_fma_mix_f32 residual, then v_cvt_pk_f16_f32)?  And is xor32 through v_permlane32_swap the same
// as through ds_bpermute?  hipcc --offload-arch=gfx950 -O2 tools/split_probe.hip -o /tmp/split_probe && /tmp/split_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8(const float4 v, f16x4& h, f16x4& l) {
  unsigned h01, h23, l01, l23;
  float r0, r1, r2, r3;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h01) : "v"(v.x), "v"(v.y));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h23) : "v"(v.z), "v"(v.w));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h01), "v"(v.x));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h01), "v"(v.y));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(h23), "v"(v.z));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(h23), "v"(v.w));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l01) : "v"(r0), "v"(r1));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l23) : "v"(r2), "v"(r3));
  h = __builtin_bit_cast(f16x4, u32x2{h01, h23});
  l = __builtin_bit_cast(f16x4, u32x2{l01, l23});
}
__device__ __forceinline__ void split6(const float4 v, f16x4& h, f16x4& l) {
  unsigned h01, h23, l01, l23;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h01) : "v"(v.x), "v"(v.y));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h23) : "v"(v.z), "v"(v.w));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l01) : "v"(h01), "v"(v.x));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l01) : "v"(h01), "v"(v.y));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l23) : "v"(h23), "v"(v.z));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l23) : "v"(h23), "v"(v.w));
  h = __builtin_bit_cast(f16x4, u32x2{h01, h23});
  l = __builtin_bit_cast(f16x4, u32x2{l01, l23});
}
__global__ void probe(const float4* x, int n, unsigned long long* bad, float* xs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  f16x4 h8, l8, h6, l6;
  split8(x[i], h8, l8);
  split6(x[i], h6, l6);
  const u32x2 a8 = __builtin_bit_cast(u32x2, h8), b8 = __builtin_bit_cast(u32x2, l8), a6 = __builtin_bit_cast(u32x2, h6), b6 = __builtin_bit_cast(u32x2, l6);
  if (a8[0] != a6[0] || a8[1] != a6[1] || b8[0] != b6[0] || b8[1] != b6[1]) atomicAdd(bad, 1ull);
  const float v = x[i].x;
  const float a = v + __shfl_xor(v, 32);
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float b = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  if (__float_as_uint(a) != __float_as_uint(b)) atomicAdd(bad + 1, 1ull);
  const float a16 = v + __shfl_xor(v, 16);
  auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float b16 = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
  if (__float_as_uint(a16) != __float_as_uint(b16)) atomicAdd(bad + 1, 1ull);
  const float am = fmaxf(v, __shfl_xor(v, 32)), bm = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  if (__float_as_uint(am) != __float_as_uint(bm)) atomicAdd(bad + 1, 1ull);
  if (i == 0) xs[0] = a, xs[1] = b;
}
int main() {
  const int n = 1 << 22;
  std::vector<float> h(4 * (size_t)n);
  srand(1);
  for (size_t i = 0; i < h.size(); ++i) {
    // every binade from 2^-30 to 2^17, both signs, plus exact fp16 values, zeros and tiny numbers (lo parts that are fp16 subnormals)
    const int e = rand() % 48 - 30;
    const float m = 1.0f + (float)(rand() & 0xffffff) / 16777216.0f;
    float v = ldexpf(m, e) * ((rand() & 1) ? 1.f : -1.f);
    if (i % 97 == 0) v = 0.f;
    if (i % 89 == 0) v = (float)(_Float16)v;
    h[i] = v;
  }
  float4* d; unsigned long long* bad; float* xs;
  hipMalloc(&d, h.size() * 4); hipMalloc(&bad, 16); hipMalloc(&xs, 8);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemset(bad, 0, 16);
  probe<<<n / 256, 256>>>(d, n, bad, xs);
  unsigned long long hb[2];
  hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost);
  printf("split6 vs split8: %llu of %d float4 differ; xor32 / xor16 / max-xor32 by lane swaps vs bpermute: %llu differ\n", hb[0], n, hb[1]);
  return hb[0] || hb[1];
}
