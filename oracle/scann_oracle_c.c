/* C/OpenMP port of oracle/scann_oracle.py (g_update=True path of the QM9 / MP2018 configs).
 * TEST INFRASTRUCTURE ONLY -- the CPU baseline timed by bench.py and a second checker; never on the product path.
 * PARITY UNPINNED like the NumPy oracle (the reference has no tests; TensorFlow is absent).
 *
 * It follows the reference graph literally in its padded-dense [B,M,N,d] layout:
 *   scann_model.py:362-389 (embedding, dense_embed, Gaussian expansions, neighbor_d/w),
 *   attention.py:136-216 (gather, concat[3d] GEMM, LN_g, gate, q/k projections, masked softmax, context, LN),
 *   attention.py:37-40 (ResidualNorm), attention.py:267-318 (GlobalAttention), scann_model.py:424-447 (head).
 * Dense = x @ W[in,out] + b; swish = x*sigmoid(x); LayerNorm eps 1e-6 non-fused form; softmax subtracts the max.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define D 128
#define H 8
#define HD 16
#define G 20

static inline float swishf_(float x) { return x * (1.0f / (1.0f + expf(-x))); }

/* y[rows,dout] = act(x[rows,din] @ W[din,dout] + b) */
static void dense(const float* x, const float* W, const float* b, float* y, long rows, int din, int dout, int act) {
#pragma omp parallel for schedule(static)
  for (long r = 0; r < rows; ++r) {
    float acc[D];
    for (int j = 0; j < dout; ++j) acc[j] = 0.f;
    const float* xr = x + r * din;
    for (int k = 0; k < din; ++k) {
      const float xv = xr[k];
      const float* wr = W + (long)k * dout;
      for (int j = 0; j < dout; ++j) acc[j] += xv * wr[j];
    }
    float* yr = y + r * dout;
    for (int j = 0; j < dout; ++j) {
      const float v = acc[j] + b[j];
      yr[j] = act ? swishf_(v) : v;
    }
  }
}

static void layer_norm_rows(float* x, const float* gamma, const float* beta, long rows) {
#pragma omp parallel for schedule(static)
  for (long r = 0; r < rows; ++r) {
    float* xr = x + r * D;
    float mean = 0.f;
    for (int j = 0; j < D; ++j) mean += xr[j];
    mean /= D;
    float var = 0.f;
    for (int j = 0; j < D; ++j) var += (xr[j] - mean) * (xr[j] - mean);
    var /= D;
    const float rstd = 1.0f / sqrtf(var + 1e-6f);
    for (int j = 0; j < D; ++j) {
      const float inv = rstd * gamma[j];
      xr[j] = xr[j] * inv + (beta[j] - mean * inv);
    }
  }
}

typedef struct {
  const float *q_w, *q_b, *k_w, *k_b, *fg_w, *fg_b, *ln_g, *ln_b, *lng_g, *lng_b;
  const float *f1_w, *f1_b, *f2_w, *f2_b, *lnr_g, *lnr_b;
} layer_w;

typedef struct {
  int n_atoms, emb, n_attention, use_attn_norm, use_ga_norm, relu_out;
  float gaussian_d;
  const float *embed, *de_w, *de_b, *nd_w, *nd_b, *nw_w, *nw_b;
  const layer_w* layers;
  const float *al_w, *al_b, *gq_w, *gq_b, *gk_w, *gk_b, *bf_w, *bf_b, *pp_w, *pp_b;
} model_w;

/* inputs: atomic[B,M] i32, atom_mask[B,M] f32, nbr[B,M,N] i32, nmask[B,M,N] f32, nweight, ndist [B,M,N] f32.
 * outputs: y[B], ga[B,M].  Returns 0, or -1 on allocation failure. */
int scann_oracle_forward(const model_w* mw, int B, int M, int N, const int32_t* atomic, const float* atom_mask,
                         const int32_t* nbr, const float* nmask, const float* nweight, const float* ndist, float* y,
                         float* ga) {
  const long A = (long)B * M, E = A * N;
  float* centers = malloc(sizeof(float) * A * D);
  float* embx = malloc(sizeof(float) * A * mw->emb);
  float* geom = malloc(sizeof(float) * E * D);
  float* tmpE = malloc(sizeof(float) * E * D);
  float* cat = malloc(sizeof(float) * E * 3 * D);
  float* key = malloc(sizeof(float) * E * D);
  float* query = malloc(sizeof(float) * A * D);
  float* ctx = malloc(sizeof(float) * A * D);
  float* t1 = malloc(sizeof(float) * A * D);
  float* gbas = malloc(sizeof(float) * E * G);
  if (!centers || !embx || !geom || !tmpE || !cat || !key || !query || !ctx || !t1 || !gbas) return -1;
  /* Embedding + dense_embed (scann_model.py:362,373) */
  for (long a = 0; a < A; ++a) memcpy(embx + a * mw->emb, mw->embed + (long)atomic[a] * mw->emb, sizeof(float) * mw->emb);
  dense(embx, mw->de_w, mw->de_b, centers, A, mw->emb, D, 1);
  /* Gaussian expansions + neighbor_d / neighbor_w (scann_model.py:378-389) */
  float cd[G], cw[G];
  for (int k = 0; k < G; ++k) {
    cd[k] = (float)(k * ((double)mw->gaussian_d / (G - 1)));
    cw[k] = (float)(k * (2.0 * M_PI / (G - 1)));
  }
  cd[G - 1] = mw->gaussian_d;
  cw[G - 1] = (float)(2.0 * M_PI);
#pragma omp parallel for schedule(static)
  for (long e = 0; e < E; ++e)
    for (int k = 0; k < G; ++k) {
      const float dd = ndist[e] - cd[k];
      gbas[e * G + k] = expf(-(dd * dd) / 0.25f);
    }
  dense(gbas, mw->nd_w, mw->nd_b, geom, E, G, D, 1);
#pragma omp parallel for schedule(static)
  for (long e = 0; e < E; ++e)
    for (int k = 0; k < G; ++k) {
      const float dd = nweight[e] - cw[k];
      gbas[e * G + k] = expf(-(dd * dd) / 0.25f);
    }
  dense(gbas, mw->nw_w, mw->nw_b, tmpE, E, G, D, 1);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < E * D; ++i) geom[i] *= tmpE[i];

  for (int l = 0; l < mw->n_attention; ++l) {
    const layer_w* lw = &mw->layers[l];
    /* gather + concat [centre, geometry, neighbour] (attention.py:136-150) */
#pragma omp parallel for schedule(static)
    for (long e = 0; e < E; ++e) {
      const long a = e / N, b = a / M;
      const float* cn = centers + ((long)b * M + nbr[e]) * D;
      float* c = cat + e * 3 * D;
      memcpy(c, centers + a * D, sizeof(float) * D);
      memcpy(c + D, geom + e * D, sizeof(float) * D);
      memcpy(c + 2 * D, cn, sizeof(float) * D);
    }
    dense(cat, lw->fg_w, lw->fg_b, tmpE, E, 3 * D, D, 1);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < E * D; ++i) geom[i] = tmpE[i] + geom[i];
    layer_norm_rows(geom, lw->lng_g, lw->lng_b, E); /* :153 */
    /* atom_neighbor * neighbor_geometry (:157) */
#pragma omp parallel for schedule(static)
    for (long e = 0; e < E; ++e) {
      const long a = e / N, b = a / M;
      const float* cn = centers + ((long)b * M + nbr[e]) * D;
      for (int j = 0; j < D; ++j) tmpE[e * D + j] = cn[j] * geom[e * D + j];
    }
    dense(centers, lw->q_w, lw->q_b, query, A, D, D, 0);
    dense(tmpE, lw->k_w, lw->k_b, key, E, D, D, 0);
    /* energy, masked softmax, context (:180-212) */
#pragma omp parallel for schedule(static)
    for (long a = 0; a < A; ++a) {
      float* c = ctx + a * D;
      for (int j = 0; j < D; ++j) c[j] = 0.f;
      for (int h = 0; h < H; ++h) {
        float en[256];
        float mx = -INFINITY;
        for (int n = 0; n < N; ++n) {
          const float* k = key + (a * N + n) * D + h * HD;
          const float* q = query + a * D + h * HD;
          float e = 0.f;
          for (int j = 0; j < HD; ++j) e += (q[j] * 0.25f) * k[j];
          e += (1.0f - nmask[a * N + n]) * -1e9f;
          en[n] = e;
          if (e > mx) mx = e;
        }
        float s = 0.f;
        for (int n = 0; n < N; ++n) { en[n] = expf(en[n] - mx); s += en[n]; }
        for (int n = 0; n < N; ++n) {
          const float at = nmask[a * N + n] * (en[n] / s);
          const float* k = key + (a * N + n) * D + h * HD;
          for (int j = 0; j < HD; ++j) c[h * HD + j] += at * k[j];
        }
      }
      for (int j = 0; j < D; ++j) c[j] += query[a * D + j];
    }
    layer_norm_rows(ctx, lw->ln_g, lw->ln_b, A);
    if (mw->use_attn_norm) { /* ResidualNorm (:37-40) */
      dense(ctx, lw->f1_w, lw->f1_b, t1, A, D, D, 1);
      dense(t1, lw->f2_w, lw->f2_b, centers, A, D, D, 0);
#pragma omp parallel for schedule(static)
      for (long i = 0; i < A * D; ++i) centers[i] += ctx[i];
      layer_norm_rows(centers, lw->lnr_g, lw->lnr_b, A);
    } else {
      memcpy(centers, ctx, sizeof(float) * A * D);
    }
  }
  /* after_Lc, GlobalAttention, head (scann_model.py:424-447, attention.py:267-318) */
  dense(centers, mw->al_w, mw->al_b, t1, A, D, D, 1);
  dense(t1, mw->gq_w, mw->gq_b, query, A, D, D, 0);
  dense(t1, mw->gk_w, mw->gk_b, ctx, A, D, D, 0);
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < B; ++b) {
    float agg[1024];
    float nrm = 0.f;
    for (int i = 0; i < M; ++i) {
      float s = 0.f;
      const float mi = atom_mask[b * M + i];
      for (int j = 0; j < M; ++j) {
        if (j == i) continue;
        const float mj = atom_mask[b * M + j];
        float e = 0.f;
        for (int k = 0; k < D; ++k) e += (mi * ctx[((long)b * M + i) * D + k]) * (mj * query[((long)b * M + j) * D + k]);
        s += e;
      }
      agg[i] = mi * s;
      nrm += agg[i] * agg[i];
    }
    nrm = sqrtf(nrm);
    float mx = -INFINITY;
    for (int i = 0; i < M; ++i) {
      if (mw->use_ga_norm) agg[i] = agg[i] / nrm;
      agg[i] += (1.0f - atom_mask[b * M + i]) * -1e9f;
      if (agg[i] > mx) mx = agg[i];
    }
    float s = 0.f;
    for (int i = 0; i < M; ++i) { agg[i] = expf(agg[i] - mx); s += agg[i]; }
    float rep[D];
    for (int k = 0; k < D; ++k) rep[k] = 0.f;
    for (int i = 0; i < M; ++i) {
      const float at = agg[i] / s;
      ga[b * M + i] = at;
      for (int k = 0; k < D; ++k) rep[k] += atom_mask[b * M + i] * (at * ctx[((long)b * M + i) * D + k]);
    }
    float out = 0.f;
    for (int j = 0; j < D; ++j) {
      float hsum = 0.f;
      for (int k = 0; k < D; ++k) hsum += rep[k] * mw->bf_w[k * D + j];
      out += swishf_(hsum + mw->bf_b[j]) * mw->pp_w[j];
    }
    out += mw->pp_b[0];
    y[b] = (mw->relu_out && out < 0.f) ? 0.f : out;
  }
  free(centers); free(embx); free(geom); free(tmpE); free(cat); free(key); free(query); free(ctx); free(t1); free(gbas);
  return 0;
}
