// Split-fp16 MFMA helpers shared by the forward kernels (scann_kernels.hip) and the fused backward kernels
// (scann_train_fused.hip).  Device code only; the scheme and its evidence are described in scann_kernels.hip (edge-tile section).
#pragma once
#include "scann_internal.h"

namespace scann {

// Diagnostic build only (-DSCANN_STAMPS): per-workgroup phase timestamps, written to a buffer nothing else reads.
#ifdef SCANN_STAMPS
#define STAMP(buf, slot)                                                                      \
  do {                                                                                        \
    if ((buf) && threadIdx.x == 0) {                                                          \
      unsigned long long t_;                                                                  \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
      (buf)[(size_t)blockIdx.x * 16 + (slot)] = t_;                                            \
    }                                                                                         \
  } while (0)
#define STAMP_IF(buf, slot, cond)                                                             \
  do {                                                                                        \
    if ((buf) && (cond)) {                                                                    \
      unsigned long long t_;                                                                  \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
      (buf)[(size_t)blockIdx.x * 16 + (slot)] = t_;                                            \
    }                                                                                         \
  } while (0)
#define STAMP_REAL(buf, slot)                                                                 \
  do {                                                                                        \
    if ((buf) && threadIdx.x == 0) {                                                          \
      unsigned long long t_;                                                                  \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
      (buf)[(size_t)blockIdx.x * 16 + (slot)] = t_;                                            \
    }                                                                                         \
  } while (0)
#define STAMP_HWID(buf, slot) /* where the workgroup runs: HW_ID (wave / SIMD / CU / SE) in the low word, XCC_ID in the high */ \
  do {                                                                                        \
    if ((buf) && threadIdx.x == 0) {                                                          \
      unsigned hw_, xcc_;                                                                     \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                       \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                     \
      (buf)[(size_t)blockIdx.x * 16 + (slot)] = ((unsigned long long)xcc_ << 32) | hw_;        \
    }                                                                                         \
  } while (0)
#else
#define STAMP_HWID(buf, slot) do {} while (0)
#define STAMP(buf, slot) do {} while (0)
#define STAMP_IF(buf, slot, cond) do {} while (0)
#define STAMP_REAL(buf, slot) do {} while (0)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

// C/D map of the 32x32 MFMA: register i of lane l holds row (i&3) + 8*(i>>2) + 4*(l>>5), column l&31.
__device__ __forceinline__ int acc_row(int i, int lane) { return (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5); }

// ---- split-fp16 projection helpers (shared by the atom and edge kernels; scheme: see the edge-tile section) ------------

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int PLANE_STRIDE = 136;  // halfs per staged row: 128 + 8 pad = 272 B (conflict-free b128 fragment reads)

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2).  Consecutive tiles hold neighbouring
// atoms of the same structures and gather the same centre rows, so give every XCD a CONTIGUOUS run of tiles
// (bijective for any grid size).  Speed only: any placement is correct.
__device__ __forceinline__ int xcd_tile(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7, i = b >> 3;
  return x * q + min(x, r) + i;
}

// hi / lo fp16 parts of four consecutive fp32 values: hi = fp16(x) (round to nearest), lo = fp16(x - hi).  Written out as the
// SIX instructions it takes -- two packed conversions, then the four residuals x - hi formed by v_fma_mixlo_f16 / v_fma_mixhi_f16, which
// read the fp16 half of a register as an fp32 operand (no conversion back), compute hi * -1 + x exactly in fp32 and write its fp16
// rounding straight into the low / high half of the destination (no second pair of packed conversions: the 8-instruction form of rounds
// 2-4) -- because hipcc makes 15 of the plain C form.  Same bits as both (tools/split_probe.hip: every binade, fp16 subnormal residuals).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4(const float4 v, f16x4& h, f16x4& l) {
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

// One wave's slab of a split weight (pack_weight_f16): [wave][k-step][plane hi|lo][lane][8 halfs] -- per k-step and plane
// one coalesced 1-KiB read.  KS = K / 16 k-steps.
template <int KS, int KT = KS>
__device__ __forceinline__ void load_wsplit(const _Float16* __restrict__ Wp, int wave, int lane, f16x8 (&wh)[KS], f16x8 (&wl)[KS], int s0 = 0) {
  // (uniform base + 32-bit per-lane byte offset: the saddr form of global_load -- ONE offset VGPR for the whole slab instead of a
  // 64-bit address pair per 4 KiB of it; the slab of a wave is KT x 2 KiB, far below 4 GiB)
  const char* __restrict__ base = reinterpret_cast<const char*>(Wp);
  const unsigned off = ((unsigned)wave * (KT * 2 * 64) + (unsigned)lane) * 16u;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    // (the k-step's 2 KiB go into the UNIFORM base -- an s_add -- not into the per-lane offset: added to the offset they are one address
    //  VGPR per KiB beyond the instruction's 4 KiB immediate range, which hipcc then keeps alive, or spills, for the next slab's loads)
    wh[s] = *reinterpret_cast<const f16x8*>((base + (size_t)((2 * (s0 + s)) * 64 * 16)) + off);
    wl[s] = *reinterpret_cast<const f16x8*>((base + (size_t)((2 * (s0 + s) + 1) * 64 * 16)) + off);
  }
}

// acc[rt] (transposed: lane = row 32 rt + (lane & 31), registers = 16 of the wave's 32 columns, see mma128T) +=
// X[rows][0 .. 16 KS) . W[0 .. 16 KS)[32 wave .. +32), X given as hi / lo planes in LDS.  Operand map of
// v_mfma_f32_32x32x16_f16 (checked with exact integers by tools/mfma_f16_probe.hip): lane l supplies A[i = l & 31][k = 8 (l >> 5) + j]
// and B[k = 8 (l >> 5) + j][j' = l & 31], j = 0..7; weights are the A operand, so the product comes out transposed.
template <int KS, bool FIRST = false, int STRIDE = PLANE_STRIDE, int RT = 2>
__device__ __forceinline__ void mma_split(const _Float16* __restrict__ sH, const _Float16* __restrict__ sL, const f16x8 (&wh)[KS],
                                          const f16x8 (&wl)[KS], int lane, f32x16 (&acc)[RT]) {
  const int off = (lane & 31) * STRIDE + 8 * (lane >> 5);
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < KS; ++s) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const f16x8 xh = *reinterpret_cast<const f16x8*>(sH + off + rt * 32 * STRIDE + 16 * s);
      const f16x8 xl = *reinterpret_cast<const f16x8*>(sL + off + rt * 32 * STRIDE + 16 * s);
      // FIRST: the chain starts from the inline constant 0 (no zero-fill of the 16 accumulator registers)
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[s], xh, FIRST && s == 0 ? zero : acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], xl, acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], xh, acc[rt], 0, 0, 0);
    }
  }
}

// v[lane] + v[lane ^ 32] through v_permlane32_swap (gfx950: the upper half of one register changes places with the lower half of another
// -- a VALU instruction, no LDS round trip as the ds_bpermute of __shfl_xor): after swapping two copies of v, one holds the lower lanes'
// values in both halves and the other the upper lanes', and their sum is the same addition in both halves (fp32 + commutes: same bits).
__device__ __forceinline__ float xor32(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Global access as (uniform base pointer) + (32-bit per-lane BYTE offset): compiles to the saddr form of global_load / global_store
// -- one offset VGPR per row instead of a 64-bit address pair per tensor (the tensors here are < 4 GiB each: scann_batch_upload
// checks n_edge * 512 < 2^32).
__device__ __forceinline__ float4 ld4(const float* __restrict__ base, unsigned byte_off) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void st4(float* __restrict__ base, unsigned byte_off, const float4 v) {
  *reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
// the same for max, and for lane ^ 16 (v_permlane16_swap: odd rows of 16 lanes of one register <-> even rows of the other)
__device__ __forceinline__ float xor32_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor16(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// max over the whole wave, every lane getting it: three DPP steps inside a row of 16 (lane ^ 1, lane ^ 2, the 8-lane mirror, the 16-lane
// mirror), then the two lane swaps -- six VALU instructions in place of six ds_bpermute round trips
__device__ __forceinline__ float wave_max64(float v) {
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4E, 0xF, 0xF, true)));
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x141, 0xF, 0xF, true)));  // row_half_mirror
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x140, 0xF, 0xF, true)));  // row_mirror
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  return xor32_max(v);
}
// sum over the whole wave / over each half of 32 lanes, every lane getting it: the additions of the lane ^ 1, ^ 2, ^ 4, ^ 8, ^ 16 (, ^ 32)
// butterfly -- after a step every group holds ONE value, so the mirror a DPP step pairs a lane with carries what its xor partner does
__device__ __forceinline__ float sum32(float v) {
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1, 0xF, 0xF, true));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4E, 0xF, 0xF, true));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x140, 0xF, 0xF, true));  // row_mirror
  return xor16(v);
}
__device__ __forceinline__ float wave_sum64(float v) { return xor32(sum32(v)); }
// Sum over groups of 8 consecutive lanes, every lane getting the total, through DPP operands (VALU; hipcc's __shfl_xor is a ds_bpermute:
// an LDS round trip per step): lane ^ 1 and lane ^ 2 as quad permutations, then the 8-lane mirror (lane i <-> 7 - i), which pairs each quad
// -- uniform by then -- with the other one.  Every step adds the same two numbers as the __shfl_xor(1 / 2 / 4) ladder: same bits.
__device__ __forceinline__ float sum8(float v) {
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1, 0xF, 0xF, true));   // quad_perm:[1,0,3,2]
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4E, 0xF, 0xF, true));   // quad_perm:[2,3,0,1]
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  return v;
}
// LayerNormalization(epsilon = 1e-6) in the forward kernels (attention.py:35,111,113; Keras' non-fused form is x inv + (beta - mean inv),
// inv = rsqrt(var + eps) gamma: four roundings per element).  Here 1 / sqrt(var + eps) is ONE v_rsq_f32 (1 ulp; the correctly rounded
// sqrt + division it replaces are ~28 instructions per row) and an element is two fused multiply-adds, ((x rstd - mean rstd) gamma +
// beta): two roundings, two instructions instead of four.  Measured against the fp64 restatement the outputs sit no further away than
// with the Keras form (tests/test_gpu_parity.py keeps its bounds; profiles/r05_notes.md has the numbers).
__device__ __forceinline__ float ln_rstd(float var) { return __builtin_amdgcn_rsqf(var + 1e-6f); }
__device__ __forceinline__ float ln_apply(float x, float rstd, float nmr /* -mean * rstd */, float g, float b) {
  return fmaf(fmaf(x, rstd, nmr), g, b);
}
// one 32-bit word the same way (an `int` index into a pointer costs a 64-bit shift-and-add per access)
__device__ __forceinline__ int ld1i(const int32_t* __restrict__ base, unsigned byte_off) {
  return *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ float ld1f(const float* __restrict__ base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

// ---- elementwise helpers shared by the forward kernels (scann_kernels.hip) and the fused backward kernels ----

// swish(x) = x * sigmoid(x) on the hardware transcendental units: v_exp_f32 (2^x) and v_rcp_f32, 1 ulp each.
// Measured effect on the end-to-end parity error: DESIGN.md "numerics".
__device__ __forceinline__ float swishf(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}

// swish(x) + g with ONE rounding of the product-sum: x * sigmoid(x) + g as an fma.  This is what hipcc made of
// `f4add(f4swish(v), g)` in edge_kernel all along (the helpers' multiply and add carry the default contraction flags, and the
// kernel's `fp contract(off)` pragma does not reach into them), silently and only while both sat in one basic block; written out so
// that edge_kernel and the structure-resident kernel (where they do not) round alike.
__device__ __forceinline__ float swish_plus(float x, float g) {
  return fmaf(x, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f)), g);
}
__device__ __forceinline__ float4 f4swish_plus(float4 v, float4 g) {
  return make_float4(swish_plus(v.x, g.x), swish_plus(v.y, g.y), swish_plus(v.z, g.z), swish_plus(v.w, g.w));
}

__device__ __forceinline__ float swish_exact(float x) { return x * (1.0f / (1.0f + expf(-x))); }

// e^x through v_exp_f32 (2^x, 1 ulp); used where the argument is <= 0 (softmax numerators).
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4swish(float4 a) { return make_float4(swishf(a.x), swishf(a.y), swishf(a.z), swishf(a.w)); }
__device__ __forceinline__ float f4sum(float4 a) { return (a.x + a.y) + (a.z + a.w); }

constexpr int BASIS_STRIDE = 72;  // halfs per staged basis row: 64 + 8 pad = 144 B (conflict-free b128 fragment reads)
// exp(-(x - c)^2 / 0.25)  (custom_layers.py:63-65, width 0.5 squared at :51)
__device__ __forceinline__ float gauss(float x, float c) {
  const float d = x - c;
  return expf(-(d * d) / 0.25f);
}
// v_exp_f32 form for the fused basis MLP: |abs error| <= ~2e-8 (value * |arg| * 6e-8 peaks at arg = -1)
__device__ __forceinline__ float gauss_fast(float x, float c) {
  const float d = x - c;
  return fast_exp(-(d * d) * 4.0f);
}

// acc = X . W for the tile staged in (sH, sL), W's halves already in (whA, wlA) / (whB, wlB); when NEXT, the halves of the
// following weight are requested into the same registers as soon as the MFMAs that read them have been issued.
template <bool NEXT, int RT = 2>
__device__ __forceinline__ void gemm_tile(const _Float16* __restrict__ sH, const _Float16* __restrict__ sL, f16x8 (&whA)[4], f16x8 (&wlA)[4],
                                          f16x8 (&whB)[4], f16x8 (&wlB)[4], const _Float16* __restrict__ next, int wave, int lane,
                                          f32x16 (&acc)[RT]) {
  mma_split<4, true, PLANE_STRIDE, RT>(sH, sL, whA, wlA, lane, acc);
  __builtin_amdgcn_sched_barrier(0);
  if (NEXT) load_wsplit<4, 8>(next, wave, lane, whA, wlA, 0);
  __builtin_amdgcn_sched_barrier(0);
  mma_split<4, false, PLANE_STRIDE, RT>(sH + 64, sL + 64, whB, wlB, lane, acc);
  __builtin_amdgcn_sched_barrier(0);
  if (NEXT) load_wsplit<4, 8>(next, wave, lane, whB, wlB, 4);
  __builtin_amdgcn_sched_barrier(0);
}

// ---- exact-fp32 projections: the fallback for operands outside the split-fp16 range (EX instantiations of atom_kernel / edge_kernel,
// run by run_forward when a forward's range guard fired) -- v_mfma_f32_32x32x2_f32 on fp32 rows staged in the SAME tile buffer
// ([rows][LDS_STRIDE] fp32 = the bytes of the two fp16 planes) against the fp32 fragment-order image of the kernel (pack_weight:
// [wave 4][t 16][lane 64] float4 = W[8 t + 4 (lane >> 5) + 0..3][32 wave + (lane & 31)]), unscaled.  1/16 of the f16 pipe's rate.
template <int NTT>
__device__ __forceinline__ void load_wexact(const float* __restrict__ Wp, int wave, int lane, float4 (&w)[NTT], int t0) {
  const char* __restrict__ base = reinterpret_cast<const char*>(Wp);
  const unsigned off = ((unsigned)wave * (16 * 64) + (unsigned)lane) * 16u;
#pragma unroll
  for (int t = 0; t < NTT; ++t) w[t] = *reinterpret_cast<const float4*>((base + (size_t)((t0 + t) * 64 * 16)) + off);
}
// acc[rt] (+)= X[rows][8 t0 .. 8 (t0 + NTT)) . W[same k][32 wave .. +32); sX points at the tile's first row, column 8 t0
template <int NTT, bool FIRST, int RT>
__device__ __forceinline__ void mma_exact(const float* __restrict__ sX, const float4 (&w)[NTT], int lane, f32x16 (&acc)[RT]) {
  const float* xrow = sX + (lane & 31) * LDS_STRIDE + 4 * (lane >> 5);
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NTT; ++t)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const float4 x = *reinterpret_cast<const float4*>(xrow + rt * 32 * LDS_STRIDE + 8 * t);
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].x, x.x, FIRST && t == 0 ? zero : acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].y, x.y, acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].z, x.z, acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].w, x.w, acc[rt], 0, 0, 0);
    }
}

// One 128x128 kernel's operand registers of a wave, in two halves (A: k < 64, B: k >= 64): a split-fp16 slab or, EX, fp32 fragments.
// Same 64 VGPRs either way.
template <bool EX> struct WRegs;
template <> struct WRegs<false> { f16x8 hA[4], lA[4], hB[4], lB[4]; };
template <> struct WRegs<true> { float4 A[8], B[8]; };
template <bool EX>
__device__ __forceinline__ void load_whalf(WRegs<EX>& w, const _Float16* __restrict__ W, int wave, int lane, int half) {
  if constexpr (EX) {
    if (half == 0) load_wexact<8>(reinterpret_cast<const float*>(W), wave, lane, w.A, 0);
    else load_wexact<8>(reinterpret_cast<const float*>(W), wave, lane, w.B, 8);
  } else {
#ifdef SCANN_DIAG_HALFW  // diagnostic (wrong results): half the weight bytes -- how much of the kernels' time is weight streaming?
    if (half == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { w.hB[i] = w.hA[i]; w.lB[i] = w.lA[i]; }
      return;
    }
#endif
    if (half == 0) load_wsplit<4, 8>(W, wave, lane, w.hA, w.lA, 0);
    else load_wsplit<4, 8>(W, wave, lane, w.hB, w.lB, 4);
  }
}
// half 0 starts the accumulators (FIRST), half 1 adds to them; tile = the staged rows (TR of them: planes at TR * PLANE_STRIDE halfs)
template <bool EX, int TR, int RT>
__device__ __forceinline__ void mma_half(const unsigned char* tile, const WRegs<EX>& w, int lane, f32x16 (&acc)[RT], int half) {
  if constexpr (EX) {
    const float* sX = reinterpret_cast<const float*>(tile);
    if (half == 0) mma_exact<8, true, RT>(sX, w.A, lane, acc);
    else mma_exact<8, false, RT>(sX + 64, w.B, lane, acc);
  } else {
    const _Float16* sH = reinterpret_cast<const _Float16*>(tile);
    const _Float16* sL = sH + TR * PLANE_STRIDE;
    if (half == 0) mma_split<4, true, PLANE_STRIDE, RT>(sH, sL, w.hA, w.lA, lane, acc);
    else mma_split<4, false, PLANE_STRIDE, RT>(sH + 64, sL + 64, w.hB, w.lB, lane, acc);
  }
}
// four consecutive columns of one staged row: hi / lo fp16 parts into the planes, or (EX) the fp32 values themselves
template <bool EX, int TR>
__device__ __forceinline__ void tile_store(unsigned char* tile, int row, int col, const float4 v) {
  if constexpr (EX) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(tile) + row * LDS_STRIDE + col) = v;
  } else {
    _Float16* sH = reinterpret_cast<_Float16*>(tile);
    _Float16* sL = sH + TR * PLANE_STRIDE;
    f16x4 h, l;
    split4(v, h, l);
    *reinterpret_cast<f16x4*>(sH + row * PLANE_STRIDE + col) = h;
    *reinterpret_cast<f16x4*>(sL + row * PLANE_STRIDE + col) = l;
  }
}
// Two neighbouring pieces of one staged row -- columns col .. col + 3 (v0) and col + 8 .. col + 11 (v1), col = 32 wave + 4 (lane >> 5) + 8 j
// with j even: the accumulator layout's pieces j and j + 1 -- as ONE 16-byte store per plane instead of two 8-byte stores per piece.
// tile_store's ds_write_b64 is served in groups of 16 consecutive lanes = 16 consecutive rows at one column, and rows r and r + 8 of the
// 272-byte planes share their banks ((68 r) mod 32 = 4 (r mod 8)): every one of them a two-way conflict -- 4/5 of the conflict cycles
// of edge_kernel and atom_kernel (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 19.5 % / 18 %, profiles/r05g16_counters.txt).  A lane and its
// partner (lane ^ 32: same row, the other four columns of each piece) trade halves through v_permlane32_swap (VALU): the lower lane
// ends up with all eight columns of piece j, the upper lane with those of piece j + 1, and ds_write_b128 is served in groups of 8
// consecutive lanes = rows r .. r + 7 x 4 banks each = the 32 banks once.  The bytes in LDS are the same as tile_store's.
#ifdef SCANN_DIAG_TILE_B64  // A/B build: round 5's 8-byte stores everywhere
constexpr bool TILE_B64_ALL = true;
#else
constexpr bool TILE_B64_ALL = false;
#endif
#ifdef SCANN_DIAG_EPI_B64  // A/B build: 8-byte stores in edge_kernel's gated-row epilogue only
constexpr bool TILE_B64_EPI = true;
#else
constexpr bool TILE_B64_EPI = false;
#endif
template <bool EX, int TR, bool B64 = false>
__device__ __forceinline__ void tile_store2(unsigned char* tile, int row, int col, const float4 v0, const float4 v1) {
  if constexpr (B64 || TILE_B64_ALL) {
    tile_store<EX, TR>(tile, row, col, v0);
    tile_store<EX, TR>(tile, row, col + 8, v1);
    return;
  }
  if constexpr (EX) {
    float* p = reinterpret_cast<float*>(tile) + row * LDS_STRIDE + col;
    *reinterpret_cast<float4*>(p) = v0;
    *reinterpret_cast<float4*>(p + 8) = v1;
  } else {
    _Float16* sH = reinterpret_cast<_Float16*>(tile);
    _Float16* sL = sH + TR * PLANE_STRIDE;
    f16x4 h0, l0, h1, l1;
    split4(v0, h0, l0);
    split4(v1, h1, l1);
    const u32x2 a = __builtin_bit_cast(u32x2, h0), b = __builtin_bit_cast(u32x2, h1);
    const u32x2 c = __builtin_bit_cast(u32x2, l0), d = __builtin_bit_cast(u32x2, l1);
    // swap(x, y): [0] = {x's lower lanes, y's lower lanes}, [1] = {x's upper lanes, y's upper lanes}
    const auto h01 = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false), h23 = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
    const auto l01 = __builtin_amdgcn_permlane32_swap(c.x, d.x, false, false), l23 = __builtin_amdgcn_permlane32_swap(c.y, d.y, false, false);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int o = row * PLANE_STRIDE + col + (col & 4);  // lower lanes: piece j's first column; upper lanes: piece j + 1's
    *reinterpret_cast<u32x4*>(sH + o) = u32x4{h01[0], h23[0], h01[1], h23[1]};
    *reinterpret_cast<u32x4*>(sL + o) = u32x4{l01[0], l23[0], l01[1], l23[1]};
  }
}
// acc = X . W for the staged tile with W in `w`; when NEXT, the halves of the following kernel are requested into the same registers
// as soon as the products that read them have been issued (gemm_tile's schedule)
template <bool EX, bool NEXT, int TR, int RT>
__device__ __forceinline__ void gemm_tile_x(const unsigned char* tile, WRegs<EX>& w, const _Float16* __restrict__ next, int wave, int lane,
                                            f32x16 (&acc)[RT]) {
  mma_half<EX, TR, RT>(tile, w, lane, acc, 0);
  __builtin_amdgcn_sched_barrier(0);
  if (NEXT) load_whalf<EX>(w, next, wave, lane, 0);
  __builtin_amdgcn_sched_barrier(0);
  mma_half<EX, TR, RT>(tile, w, lane, acc, 1);
  __builtin_amdgcn_sched_barrier(0);
  if (NEXT) load_whalf<EX>(w, next, wave, lane, 1);
  __builtin_amdgcn_sched_barrier(0);
}

}  // namespace scann
