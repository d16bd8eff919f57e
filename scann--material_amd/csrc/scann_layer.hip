// Layer launch (EXPERIMENT, off by default: env SCANN_FUSE_LAYERS=1 / scann_set_layer_fusion): the edge tiles of LocalAttention
// iteration l AND the atom tiles that turn its contexts into layer l + 1's centres and projections (or, after the last layer, into
// the readout's query / key rows) in ONE launch.
//
// The idea: an atom tile of layer l + 1 needs nothing but the context rows of ITS atoms, i.e. the handful of edge tiles centred on
// them -- not the whole layer -- so the atom tiles ride in the edge launch, each one a workgroup placed `delay` workgroups behind the
// edge tiles it depends on, waiting on a counter those tiles bump when their context rows are stored; the atom launch with its own
// fill and drain is gone.
//
// What was measured (profiles/r04_notes.md, "layer launches"): the same bytes as separate launches, no faults -- and no gain.  With
// the waits compiled out (SCANN_DIAG_NOWAIT, wrong results) a 10-batch layer launch takes 99.4 us wherever the atom tiles sit in the
// grid: exactly edge_kernel's 79 us + atom_kernel's 20 us.  The two kernels' times ADD when they share the chip -- atom_kernel's time
// is not fill / drain latency that idle slots could absorb, it is the same per-CU issue / operand-fetch resources the edge tiles
// use -- and the real waits cost on top of that (109 us at delay 96, 100 us at delay 400, i.e. parity at best; small launches, where
// slots are free, lose more: the agent-scope hand-off itself takes microseconds).  Kept, with its tests, as the measurement and for
// the machinery (work lists, XCD-local dependency counters, the fault fallback).
//
// Hand-off (one XCD only, by construction): workgroups are dealt round-robin over the 8 XCDs -- XCD = (start + b) % 8 with a start
// that differs from dispatch to dispatch (tools/xcc_probe.hip) -- and the work list gives every residue class b % 8 a contiguous run
// of edge tiles plus the atom tiles that cover exactly those tiles' atoms, so producer and consumer of a context row share one L2.
// Producer: context rows stored, `s_waitcnt vmcnt(0)` + workgroup barrier (the stores have reached L2: the L1 is write-through; this
// costs the edge tile nothing, SCANN_DIAG_TAILWAIT), then ONE lane adds to the counters of the (<= 2) atom tiles its atoms fall into
// (agent-scope atomic; workgroup scope was tried and never became visible to the poller).  The addend carries the workgroup's
// (XCC_ID - b) % 8 in a 7-bit field of its own, so the consumer sees both that all `need` producers have arrived and that every one of
// them ran on its XCD.  Consumer: one lane polls (agent-scope load), workgroup barrier, acquire fence, then the ordinary atom_kernel
// body.  Edge tiles never wait, and an atom tile's workgroup is only dispatched after the edge tiles ahead of it in the list, so the
// waits cannot deadlock; they are bounded anyway (a wait that runs out, or a producer from another XCD, sets `fault`: the host re-runs
// the batch through separate launches and keeps the handle there).  What the layer writes for the NEXT launch (centres, P1, P3, q)
// goes to the other half of a double buffer: edge tiles of this launch still gather the current ones.  The arithmetic is
// atom_kernel's and edge_kernel's own text (scann_*_body.inc): same bytes.
#include "scann_internal.h"
#include "scann_mma.h"

namespace scann {

template <bool FB, bool AFFN, int AMODE>
__global__ __launch_bounds__(256, 3) void layer_kernel(EdgeArgs ea, AtomArgs aa, LayerFuse f) {
  constexpr int LDS_EDGE = 2 * 64 * PLANE_STRIDE * 2 + TQ * LDS_STRIDE * 4 + 64 * NHEAD * 4 + 5 * D * 4 + 112;
  constexpr int LDS_ATOM = 2 * 64 * PLANE_STRIDE * 2 + 64 * 8 * 4 + 7 * D * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_EDGE > LDS_ATOM ? LDS_EDGE : LDS_ATOM];
  const int2 w = f.work[blockIdx.x];  // {kind, index}: 0 edge tile, 1 atom tile, -1 nothing (the XCDs' lists differ in length)
  if (w.x < 0) return;
  // The placement the hand-off relies on: workgroups are dealt round-robin over the 8 XCDs, from a start that differs from dispatch
  // to dispatch (measured: tools/xcc_probe.hip and the handle's own launches) -- XCD = (start + b) % 8.  Every arrival carries its
  // workgroup's `start` in the counter word, so the consumer sees that all its producers ran where it runs.
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned start = (xcc - blockIdx.x) & 7u;
  if (w.x == 0) {
    constexpr bool GUPD = true, EX = false;
    constexpr int RT = 2, TEK = 64;
    const EdgeArgs& a = ea;
    unsigned char* const sTile = smem;
    float* const sQ = reinterpret_cast<float*>(smem + 2 * TEK * PLANE_STRIDE * 2);
    float* const sE = sQ + TQ * LDS_STRIDE;
    float* const sPar = sE + TEK * NHEAD;
    int* const sOff = reinterpret_cast<int*>(sPar + 5 * D);
    {
#pragma clang fp contract(off)
#define SCANN_EDGE_TIX w.y
#include "scann_edge_body.inc"
#undef SCANN_EDGE_TIX
    }
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): this wave's context rows have been taken by the L2 ...
    __syncthreads();                // ... and so have every other wave's
    if (threadIdx.x == 0) {
      const int first = f.e_atile[2 * w.y], n = f.e_atile[2 * w.y + 1];
      const unsigned long long one = 1ull | (1ull << (8 + 7 * start));  // bits 0-7: arrivals; 7 bits per `start` value: arrivals from there
      for (int k = 0; k < n; ++k) __hip_atomic_fetch_add(f.a_count + first + k, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    if (threadIdx.x == 0) {
      const unsigned long long need = (unsigned long long)f.a_need[w.y];
      int spins = 0;
      unsigned long long v;
#ifdef SCANN_DIAG_NOWAIT  // diagnostic (wrong results): what the waits themselves cost
      v = need | (need << (8 + 7 * start));
      while (false) {
#else
      while (((v = __hip_atomic_load(f.a_count + w.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 255ull) < need) {
#endif
        __builtin_amdgcn_s_sleep(16);
        if (++spins > (1 << 20)) {  // ~a second: something is broken; say so instead of hanging the device
          *reinterpret_cast<volatile int32_t*>(f.fault) = 1;  // (host-pinned word: a plain store)
          break;
        }
      }
      if ((v & 255ull) >= need && v != (need | (need << (8 + 7 * start))))  // a producer ran on another XCD: its rows may not be in this L2
        *reinterpret_cast<volatile int32_t*>(f.fault) = 2 | (int)(start << 8) | (int)((xcc & 15u) << 12);
    }
    __syncthreads();
    // drop this CU's (possibly stale) L1 lines of the context rows.  (Measured: `buffer_inv sc0` or no invalidate at all -- nothing
    // on this CU can have read those rows since the launch began -- are 1.5 % faster; the formal acquire stays.)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    constexpr bool FFN = AFFN, EX = false;
    constexpr int MODE = AMODE, RT = 2, TAR = 64;
    const AtomArgs& a = aa;
    unsigned char* const sTile = smem;
    float* const sRed = reinterpret_cast<float*>(smem + 2 * TAR * PLANE_STRIDE * 2);
    float* const sPar = sRed + TAR * 8;
    {
#pragma clang fp contract(off)
#define SCANN_ATOM_BIX w.y
#include "scann_atom_body.inc"
#undef SCANN_ATOM_BIX
    }
  }
}

void launch_layer(const EdgeArgs& ea, const AtomArgs& aa, const LayerFuse& f, hipStream_t s) {
  if (f.n_block <= 0) return;
  const dim3 grid(f.n_block), block(256);
  const bool fb = ea.fuse_basis != 0;
#define SCANN_LAYER_CASE(FB_, FFN_, MODE_) hipLaunchKernelGGL((layer_kernel<FB_, FFN_, MODE_>), grid, block, 0, s, ea, aa, f)
  if (aa.mode == 0) {
    if (fb) { if (aa.ffn) SCANN_LAYER_CASE(true, true, 0); else SCANN_LAYER_CASE(true, false, 0); }
    else { if (aa.ffn) SCANN_LAYER_CASE(false, true, 0); else SCANN_LAYER_CASE(false, false, 0); }
  } else {
    if (fb) { if (aa.ffn) SCANN_LAYER_CASE(true, true, 2); else SCANN_LAYER_CASE(true, false, 2); }
    else { if (aa.ffn) SCANN_LAYER_CASE(false, true, 2); else SCANN_LAYER_CASE(false, false, 2); }
  }
#undef SCANN_LAYER_CASE
}

}  // namespace scann
