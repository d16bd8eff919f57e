// Row-owner edge kernel for the SCANN+ (g_update) inference forward: LocalAttention.call for the edges of a batch
// (attention.py:136-216), one WAVE per tile of <= 32 edges of whole atoms.
//
// edge_kernel (scann_kernels.hip) gives every wave 32 OUTPUT COLUMNS of a 64-edge tile: the activations cross LDS between
// the two GEMMs, each wave re-streams its weight slab from L2 per tile (2 KB per edge) and six workgroup barriers couple the
// four waves.  Here a wave owns 32 edge ROWS and all 128 columns of them:
//
//   * both split-fp16 weights of the layer (W2 = geometry third of filter_geo, Wk = key; 2 x 64 KB) are copied into LDS ONCE per
//     workgroup; one persistent workgroup of 8 waves per CU pulls tiles from per-XCD work queues until they are empty;
//   * the weights are the A operand, so a product comes out transposed: lane (r, h) holds edge row r and, per 32-column
//     block cb, the 16 columns 32 cb + 8 j + 4 h + i (j, i = 0..3) -- lanes (r, 0) and (r, 1) hold the whole row.  Geometry
//     rows are loaded from HBM in exactly that layout, and an accumulator tile in that layout IS the B operand of the next
//     MFMA (8 registers = one k-step) once the weight image uses the same k order inside a k-step (pack_weight_f16 perm):
//     no LDS staging of activations, no barrier after the weight copy, LayerNorm statistics are in-lane sums plus one
//     lane-pair exchange;
//   * softmax and context are sums over the edges of an atom = over a run of lanes: segmented Hillis-Steele scans with
//     ds_bpermute.  The combination tree of a segment depends only on offsets from its first lane, so a row's bits do not
//     depend on where in a tile (or batch) its atom lands (batch-composition invariance, DESIGN.md "Determinism").
//
// Used when the batch has no atom above 32 neighbours and nothing has to be kept for a backward pass; everything else
// (training forward, base branch, chunked atoms) stays on edge_kernel.  Atoms without edges are handled by iso_ctx_kernel.
#include "scann_internal.h"
#include "scann_mma.h"

namespace scann {

namespace {

__device__ __forceinline__ float swish_fast(float x, float& sig) {
  sig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
  return x * sig;
}
__device__ __forceinline__ float exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// value of `v` in lane `src` (any lane of the wave)
__device__ __forceinline__ float lane_get(int src, float v) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v)));
}
__device__ __forceinline__ float pair_sum(float v) { return v + __shfl_xor(v, 32); }
// DPP move: lanes without a source (outside the 16-lane row / masked rows) keep `old`
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_get(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// Work-queue ticket by lane 0 alone, WITHOUT the wait hipcc puts behind a returning atomic under `if (lane == 0)`: the ticket
// travels while the tile's row loads are issued.  ticket_wait<N>: N = vector-memory operations issued after the ticket.
__device__ __forceinline__ int ticket_issue(int* q) {
  int tk;
  unsigned long long keep;
  asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\ts_nop 4\n\tglobal_atomic_add %0, %2, %3, %4 sc0\n\ts_mov_b64 exec, %1"
               : "=&v"(tk), "=&s"(keep)
               : "v"(0), "v"(1), "s"(q)
               : "memory");
  return tk;
}
template <int N>
__device__ __forceinline__ int ticket_wait(int tk) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(tk) : "n"(N) : "memory");
  return __builtin_amdgcn_readfirstlane(tk);
}

// hi / lo fp16 parts of eight fp32 values (two float4 = one k-step of this lane's row)
__device__ __forceinline__ void split8(const float (&x)[8], f16x8& h, f16x8& l) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    h[i] = (_Float16)x[i];
    l[i] = (_Float16)(x[i] - (float)h[i]);
  }
}

// Diagnostic build only (-DSCANN_STAMPS): per-wave sums of the cycles spent in each phase, over all its tiles
#ifdef SCANN_STAMPS
#define RSTAMP(slot)                                                                            \
  do {                                                                                          \
    unsigned long long t_;                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                 \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    st_sum[slot] += t_ - st_prev;                                                               \
    st_prev = t_;                                                                               \
  } while (0)
#else
#define RSTAMP(slot) do {} while (0)
#endif

#ifndef STAGGER_SLEEP
#define STAGGER_SLEEP 24  // x 64 cycles per wave slot
#endif
constexpr int QSTRIDE = 32;  // ints between two queue heads (one 128-byte line each)

}  // namespace

// acc[cbo] (lane = edge row, 16 columns of block cbo) += X . W for the 128-wide rows whose lane-resident layout is x[cb][16]
// (the accumulator layout); W = split-fp16 image in LDS, [cbo 4][k-step 8][plane hi|lo][lane 64][8 halfs], k order permuted.
__device__ __forceinline__ void gemm_rows(const f16x8* __restrict__ sW, int lane, const f32x16 (&x)[4], f32x16 (&acc)[4]) {
  f16x8 xh, xl;
  {
    float xv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) xv[i] = x[0][i];
    split8(xv, xh, xl);
  }
  f16x8 wh = sW[lane], wl = sW[64 + lane];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    f16x8 nh = xh, nl = xl;
#pragma unroll
    for (int cbo = 0; cbo < 4; ++cbo) {
      // fragments of the next (k-step, column block) are requested before this block's MFMAs
      const int nc = cbo == 3 ? 0 : cbo + 1, ns = cbo == 3 ? s + 1 : s;
      f16x8 wh2 = wh, wl2 = wl;
      if (ns < 8) {
        wh2 = sW[((nc * 8 + ns) * 2) * 64 + lane];
        wl2 = sW[((nc * 8 + ns) * 2 + 1) * 64 + lane];
      }
      if (cbo == 1 && s < 7) {  // hi / lo parts of the next k-step's operand, under this step's MFMAs
        float xv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) xv[i] = x[(s + 1) >> 1][8 * ((s + 1) & 1) + i];
        split8(xv, nh, nl);
      }
      acc[cbo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc[cbo], 0, 0, 0);
      acc[cbo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc[cbo], 0, 0, 0);
      acc[cbo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc[cbo], 0, 0, 0);
      wh = wh2; wl = wl2;
      __builtin_amdgcn_sched_barrier(0);
    }
    xh = nh; xl = nl;
  }
}

// next tile of the launch for this wave: queue qx first, then the other seven (wave-uniform result, -1 when all are empty)
__device__ __forceinline__ int queue_pop(int* __restrict__ qcnt, int n_tile, int lane, int& qx, int& tried) {
  while (tried < 8) {
    int t = 0;
    if (lane == 0) t = atomicAdd(&qcnt[qx * QSTRIDE], 1);
    t = __builtin_amdgcn_readfirstlane(t);
    const int lo = (int)(((long long)n_tile * qx) >> 3), hi = (int)(((long long)n_tile * (qx + 1)) >> 3);
    if (lo + t < hi) return lo + t;
    qx = (qx + 1) & 7;
    ++tried;
  }
  return -1;
}

__global__ __launch_bounds__(512, 2) void edge_rows_kernel(EdgeRowsArgs a) {
#pragma clang fp contract(off)  // fusions are written out (fmaf): a row's bits must not depend on the code path around it
  __shared__ __attribute__((aligned(16))) f16x8 sW2[4 * 8 * 2 * 64];  // 64 KB: geometry third of filter_geo
  __shared__ __attribute__((aligned(16))) f16x8 sWk[4 * 8 * 2 * 64];  // 64 KB: key
  __shared__ __attribute__((aligned(16))) float sPar[5 * D];          // layer_norm_g gamma | beta | key bias | layer_norm gamma | beta
  const int tid = threadIdx.x, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  {
    const f16x8* __restrict__ g2 = reinterpret_cast<const f16x8*>(a.W2p);
    const f16x8* __restrict__ gk = reinterpret_cast<const f16x8*>(a.Wkp);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sW2[tid + 512 * i] = g2[tid + 512 * i];
      sWk[tid + 512 * i] = gk[tid + 512 * i];
    }
    if (tid < D) {
      sPar[tid] = a.lng_g[tid];
      sPar[D + tid] = a.lng_b[tid];
      sPar[2 * D + tid] = a.bk[tid];
      sPar[3 * D + tid] = a.ln_g[tid];
      sPar[4 * D + tid] = a.ln_b[tid];
    }
  }
  __syncthreads();  // the only workgroup barrier: from here on the waves run independently

  constexpr float WINV = 1.0f / WSCALE;
  int qx = blockIdx.x & 7, tried = 0;  // blocks b and b + 8 share an XCD: each XCD group drains its own run of tiles first
#ifdef SCANN_STAMPS
  unsigned long long st_sum[16] = {}, st_prev, st_begin;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
  st_begin = st_prev;
#endif
  // The eight waves run the same program: started together they would issue their load bursts together (and share the SIMD's
  // matrix pipe with their partner wave 4 slots away in every GEMM).  Start them one load burst apart.
  for (int w = __builtin_amdgcn_readfirstlane(tid >> 6); w > 0; --w) __builtin_amdgcn_s_sleep(STAGGER_SLEEP);
  int cur = queue_pop(a.qcnt, a.n_tile, lane, qx, tried);
  // indices of a tile: requested one tile ahead, so that a tile starts with its row loads
  int2 td = make_int2(0, 1);
  int icol = 0, ictr = 0, iseg = 0;
  if (cur >= 0) {
    td = a.tiles[cur];
    const int e = td.x + min(r, (td.y & 0xff) - 1);
    icol = a.edge_col[e]; ictr = a.edge_row[e]; iseg = a.edge_seg[e];
  }
  while (cur >= 0) {
    // ticket for the next tile: requested first, read once the row loads of this tile are on their way
    const int nqx = qx;
    const int tk = ticket_issue(a.qcnt + nqx * QSTRIDE);

    const int eb = __builtin_amdgcn_readfirstlane(td.x);
    const int ne = __builtin_amdgcn_readfirstlane(td.y & 0xff);
    const int nstep = __builtin_amdgcn_readfirstlane(td.y >> 8);  // scan steps that the longest atom of the tile needs
    const bool valid = r < ne;
    const int e = eb + min(r, ne - 1);  // lanes past the end repeat the last edge (their results are never stored)
    const int col = icol, ctr = ictr;
    const int s0 = min(r, ne - 1) - (iseg & 0xff), s1 = s0 + (iseg >> 8);  // this lane's atom = lanes [s0, s1) of its half (edge_seg = position | degree << 8)
    const unsigned eoff = ((unsigned)e * D + 4 * h) * 4, nboff = ((unsigned)col * D + 4 * h) * 4, ctoff = ((unsigned)ctr * D + 4 * h) * 4;

    // diagnostic switches (EdgeRowsArgs::diag, env SCANN_ROWS_DIAG; wrong results, timing only): which traffic costs what
    const unsigned nb3 = (a.diag & 1) ? ctoff : nboff, nbc = (a.diag & 2) ? ctoff : nboff;
    RSTAMP(0);  // index wait
    // ---- geometry rows (kept for the residual, exact fp32) and the accumulators' initial value (P1[i] + P3[j]) * 2^8 ----
    f32x16 g[4], acc[4], p3[4];
    __builtin_amdgcn_sched_barrier(0);
    // geometry in the TILED layout (written by the previous layer's launch of this kernel, same tile plan): the tile's ne x 512
    // bytes hold its 32 sixteen-byte pieces plane by plane, piece p of row r at (p ne + r) x 16 -- one load instruction reads two
    // contiguous runs instead of 32 rows x 32 bytes (a quarter of the cache-line lookups).  Layer 0 / debug: row-major.
    const unsigned tbase = (unsigned)eb * (D * 4) + (unsigned)(h * ne + min(r, ne - 1)) * 16, tstep = (unsigned)ne * 32;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 v = ld4(a.geom, a.geom_in_tiled ? tbase + (4 * cb + j) * tstep : eoff + 128 * cb + 32 * j);
        g[cb][4 * j] = v.x; g[cb][4 * j + 1] = v.y; g[cb][4 * j + 2] = v.z; g[cb][4 * j + 3] = v.w;
      }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 p1 = ld4(a.P1, ctoff + 128 * cb + 32 * j), v = ld4(a.P3, nb3 + 128 * cb + 32 * j);
        acc[cb][4 * j] = p1.x; acc[cb][4 * j + 1] = p1.y; acc[cb][4 * j + 2] = p1.z; acc[cb][4 * j + 3] = p1.w;
        p3[cb][4 * j] = v.x; p3[cb][4 * j + 1] = v.y; p3[cb][4 * j + 2] = v.z; p3[cb][4 * j + 3] = v.w;
      }
    __builtin_amdgcn_sched_barrier(0);  // 48 loads = 192 registers in flight, nothing else hoisted above them
    RSTAMP(11);  // issue of the 48 row loads
    // ---- next tile: the ticket is back (it is older than the 48 loads); its indices travel under this tile ----
    int nxt;
    {
      const int t = ticket_wait<48>(tk);
      RSTAMP(12);  // ticket wait
      const int lo = (int)(((long long)a.n_tile * nqx) >> 3), hi = (int)(((long long)a.n_tile * (nqx + 1)) >> 3);
      if (lo + t < hi) {
        nxt = lo + t;
      } else {
        qx = (qx + 1) & 7;
        ++tried;
        nxt = queue_pop(a.qcnt, a.n_tile, lane, qx, tried);
      }
      td = a.tiles[max(nxt, 0)];  // unconditional (a load under a branch is waited for at the branch's end)
      const int e2 = td.x + min(r, (td.y & 0xff) - 1);
      icol = a.edge_col[e2]; ictr = a.edge_row[e2]; iseg = a.edge_seg[e2];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[cb][i] = (acc[cb][i] + p3[cb][i]) * WSCALE;
    __builtin_amdgcn_sched_barrier(0);

    RSTAMP(1);  // row loads + ticket + accumulator init
    // ---- GEMM 1: V = G . W2 + P1[i] + P3[j]   (attention.py:142-151) ----
    gemm_rows(sW2, lane, g, acc);
    __builtin_amdgcn_sched_barrier(0);
    RSTAMP(2);  // GEMM 1
    // neighbour centre rows c[j] (attention.py:136): requested now (into the registers P3 left), used after the LayerNorm
    f32x16 cn[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 v = ld4(a.c, nbc + 128 * cb + 32 * j);
        cn[cb][4 * j] = v.x; cn[cb][4 * j + 1] = v.y; cn[cb][4 * j + 2] = v.z; cn[cb][4 * j + 3] = v.w;
      }
    __builtin_amdgcn_sched_barrier(0);

    RSTAMP(3);  // centre-row requests
    // ---- T = swish(V) + G ; LayerNorm_g (attention.py:152-153), all in this lane pair ----
    float part[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float v = acc[cb][i] * WINV;
        float sig;
        (void)swish_fast(v, sig);
        const float t = fmaf(v, sig, g[cb][i]);
        acc[cb][i] = t;
        s += t;
      }
      part[cb] = s;
    }
    const float mean = pair_sum((part[0] + part[1]) + (part[2] + part[3])) * (1.0f / D);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float d = acc[cb][i] - mean;
        acc[cb][i] = d;
        s = fmaf(d, d, s);
      }
      part[cb] = s;
    }
    {
      const float var = pair_sum((part[0] + part[1]) + (part[2] + part[3])) * (1.0f / D);
      const float rstd = 1.0f / sqrtf(var + 1e-6f);
      float* const gout = a.geom_out ? a.geom_out : const_cast<float*>(a.geom);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 ga = *reinterpret_cast<const float4*>(&sPar[32 * cb + 8 * j + 4 * h]);
          const float4 be = *reinterpret_cast<const float4*>(&sPar[D + 32 * cb + 8 * j + 4 * h]);
          float4 y;
          y.x = fmaf(acc[cb][4 * j] * rstd, ga.x, be.x);
          y.y = fmaf(acc[cb][4 * j + 1] * rstd, ga.y, be.y);
          y.z = fmaf(acc[cb][4 * j + 2] * rstd, ga.z, be.z);
          y.w = fmaf(acc[cb][4 * j + 3] * rstd, ga.w, be.w);
          if (valid && !a.geom_dead && !(a.diag & 8))  // threaded to the next layer (scann_model.py:415)
            st4(gout, a.geom_out_tiled ? tbase + (4 * cb + j) * tstep : eoff + 128 * cb + 32 * j, y);
          // ang = c[j] * geom'   (attention.py:157)
          acc[cb][4 * j] = cn[cb][4 * j] * y.x; acc[cb][4 * j + 1] = cn[cb][4 * j + 1] * y.y;
          acc[cb][4 * j + 2] = cn[cb][4 * j + 2] * y.z; acc[cb][4 * j + 3] = cn[cb][4 * j + 3] * y.w;
        }
    }

    __builtin_amdgcn_sched_barrier(0);
    RSTAMP(4);  // swish, LayerNorm_g, geom' store, ang
    // ---- GEMM 2: K = ang . Wk + bk   (attention.py:163); query rows of the centre atoms requested under it ----
    f32x16 qv[4], kk[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 v = ld4(a.q, ctoff + 128 * cb + 32 * j);
        qv[cb][4 * j] = v.x; qv[cb][4 * j + 1] = v.y; qv[cb][4 * j + 2] = v.z; qv[cb][4 * j + 3] = v.w;
      }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) kk[cb][i] = 0.f;
    __builtin_amdgcn_sched_barrier(0);
    gemm_rows(sWk, lane, acc, kk);
    __builtin_amdgcn_sched_barrier(0);
    RSTAMP(5);  // GEMM 2 (+ query row requests)

    // ---- logits e[n, hd] = (q[i, hd, :] * 16^-0.5) . K[n, hd, :]   (attention.py:180-183): head hd = 2 cb + jp ----
    float lg[NHEAD];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        float ee = 0.f;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = 2 * jp + jj;
          const float4 bk = *reinterpret_cast<const float4*>(&sPar[2 * D + 32 * cb + 8 * j + 4 * h]);
          const float k0 = fmaf(kk[cb][4 * j], WINV, bk.x), k1 = fmaf(kk[cb][4 * j + 1], WINV, bk.y);
          const float k2 = fmaf(kk[cb][4 * j + 2], WINV, bk.z), k3 = fmaf(kk[cb][4 * j + 3], WINV, bk.w);
          kk[cb][4 * j] = k0; kk[cb][4 * j + 1] = k1; kk[cb][4 * j + 2] = k2; kk[cb][4 * j + 3] = k3;
          ee = fmaf(qv[cb][4 * j], k0, ee); ee = fmaf(qv[cb][4 * j + 1], k1, ee);
          ee = fmaf(qv[cb][4 * j + 2], k2, ee); ee = fmaf(qv[cb][4 * j + 3], k3, ee);
        }
        lg[2 * cb + jp] = pair_sum(ee) * 0.25f;
      }

    RSTAMP(6);  // logits
    // ---- softmax over each atom's edges (attention.py:186-192): packed edges are all unmasked ----
    // segmented inclusive scans over the lanes [s0, s1) of the atom; lane s1 - 1 ends up with the atom's total
    const int last = 32 * h + s1 - 1;
    float mx[NHEAD];
#pragma unroll
    for (int hd = 0; hd < NHEAD; ++hd) mx[hd] = lg[hd];
    // maximum: any combination order gives the same bits, so the scan runs on DPP row shifts (no LDS round trips): four steps
    // inside each 16-lane row, then lane 15 of the row below for atoms that straddle lanes 15 | 16
#pragma unroll
    for (int hd = 0; hd < NHEAD; ++hd) {
      float v = mx[hd], t;
      t = dpp_get<0x111>(v, v); v = r - 1 >= s0 ? fmaxf(v, t) : v;
      t = dpp_get<0x112>(v, v); v = r - 2 >= s0 ? fmaxf(v, t) : v;
      t = dpp_get<0x114>(v, v); v = r - 4 >= s0 ? fmaxf(v, t) : v;
      t = dpp_get<0x118>(v, v); v = r - 8 >= s0 ? fmaxf(v, t) : v;
      t = dpp_get<0x142, 0xa>(v, v); v = (r >= 16 && s0 < 16) ? fmaxf(v, t) : v;
      mx[hd] = v;
    }
    float pw[NHEAD], ps[NHEAD];
#pragma unroll
    for (int hd = 0; hd < NHEAD; ++hd) {
      pw[hd] = exp_fast(lg[hd] - lane_get(last, mx[hd]));
      ps[hd] = pw[hd];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      if (k < nstep) {
        const int d = 1 << k;
        const bool ok = r - d >= s0;
        const int src = ok ? lane - d : lane;
#pragma unroll
        for (int hd = 0; hd < NHEAD; ++hd) {
          const float t = lane_get(src, ps[hd]);
          ps[hd] += ok ? t : 0.f;
        }
      }
    }
    RSTAMP(7);  // softmax scans
    // ---- context: sum over the atom's edges of attn * K (attention.py:194-206), same scan on the 64 columns of this lane ----
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) kk[cb][i] *= pw[2 * cb + (i >> 3)];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      if (k < nstep && !(a.diag & 4)) {
        const int d = 1 << k;
        const bool ok = r - d >= s0;
        const int src = ok ? lane - d : lane;
        f32x16 t[4];  // all 64 exchanges of the step in flight before the first add
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
          for (int i = 0; i < 16; ++i) t[cb][i] = lane_get(src, kk[cb][i]);
        if (ok) {
#pragma unroll
          for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int i = 0; i < 16; ++i) kk[cb][i] += t[cb][i];
        }
      }
    }
    RSTAMP(8);  // context scan
    // ---- lane s1 - 1 of every atom: context / sum + unscaled query (attention.py:198-212), LayerNorm (:214) -> HBM ----
    {
      float rs[NHEAD];
#pragma unroll
      for (int hd = 0; hd < NHEAD; ++hd) rs[hd] = __builtin_amdgcn_rcpf(ps[hd]);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float x = fmaf(kk[cb][i], rs[2 * cb + (i >> 3)], qv[cb][i]);
          kk[cb][i] = x;
          s += x;
        }
        part[cb] = s;
      }
      const float cmean = pair_sum((part[0] + part[1]) + (part[2] + part[3])) * (1.0f / D);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float d = kk[cb][i] - cmean;
          kk[cb][i] = d;
          s = fmaf(d, d, s);
        }
        part[cb] = s;
      }
      const float cvar = pair_sum((part[0] + part[1]) + (part[2] + part[3])) * (1.0f / D);
      const float crstd = 1.0f / sqrtf(cvar + 1e-6f);
      if (valid && r == s1 - 1) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float4 ga = *reinterpret_cast<const float4*>(&sPar[3 * D + 32 * cb + 8 * j + 4 * h]);
            const float4 be = *reinterpret_cast<const float4*>(&sPar[4 * D + 32 * cb + 8 * j + 4 * h]);
            float4 y;
            y.x = fmaf(kk[cb][4 * j] * crstd, ga.x, be.x);
            y.y = fmaf(kk[cb][4 * j + 1] * crstd, ga.y, be.y);
            y.z = fmaf(kk[cb][4 * j + 2] * crstd, ga.z, be.z);
            y.w = fmaf(kk[cb][4 * j + 3] * crstd, ga.w, be.w);
            st4(a.ctx, ctoff + 128 * cb + 32 * j, y);
          }
      }
    }

    RSTAMP(9);  // residual, LayerNorm, ctx store
#ifdef SCANN_STAMPS
    st_sum[10] += 1;
#endif
    cur = nxt;
  }
#ifdef SCANN_STAMPS
  if (a.stamps && lane == 0) {
    unsigned long long* o = a.stamps + (size_t)(blockIdx.x * 8 + (tid >> 6)) * 16;
    for (int i = 0; i < 11; ++i) o[i] = st_sum[i];
    o[11] = st_prev - st_begin;
    o[12] = st_sum[11];
    o[13] = st_sum[12];
  }
#endif
}

// Atoms without neighbours: every logit of the reference's row is masked, the softmax is uniform and the multiplicative mask
// removes it again (attention.py:186-206): context = query, then the LayerNorm (:214).  One wave per such atom.
__global__ __launch_bounds__(64) void iso_ctx_kernel(const int32_t* __restrict__ iso, const float* __restrict__ q,
                                                     const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                     float* __restrict__ ctx) {
  const int atom = iso[blockIdx.x], c = 2 * threadIdx.x;
  const float2 v = *reinterpret_cast<const float2*>(q + (size_t)atom * D + c);
  float s = v.x + v.y;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
  const float mean = s * (1.0f / D);
  const float dx = v.x - mean, dy = v.y - mean;
  float m2 = dx * dx + dy * dy;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) m2 += __shfl_xor(m2, o);
  const float rstd = 1.0f / sqrtf(m2 * (1.0f / D) + 1e-6f);
  float2 y;
  y.x = (dx * rstd) * ln_g[c] + ln_b[c];
  y.y = (dy * rstd) * ln_g[c + 1] + ln_b[c + 1];
  *reinterpret_cast<float2*>(ctx + (size_t)atom * D + c) = y;
}

void launch_edge_rows(const EdgeRowsArgs& a, int n_cu, hipStream_t s) {
  if (a.n_tile > 0) {
    const int wgs = (a.n_tile + 7) / 8;
    hipLaunchKernelGGL(edge_rows_kernel, dim3(wgs < n_cu ? wgs : n_cu), dim3(512), 0, s, a);
  }
  if (a.n_iso > 0) hipLaunchKernelGGL(iso_ctx_kernel, dim3(a.n_iso), dim3(64), 0, s, a.iso, a.q, a.ln_g, a.ln_b, a.ctx);
}

}  // namespace scann
