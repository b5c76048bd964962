/* TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle", also bench.py's cpu_baseline "port") of the
 * per-sample loop of fatchord/WaveRNN `WaveRNN.generate()`.
 *
 * Follows, stage by stage:
 *   models/fatchord_version.py:192-241   (state init, I -> rnn1 -> +res -> rnn2 -> +res -> fc1 -> fc2 -> fc3)
 *   ATen CPU gru_cell (what nn.GRUCell runs at :210,:214): r=sig(gh_r+gi_r) z=sig(gh_z+gi_z)
 *                                                       n=tanh(gi_n+gh_n*r) h'=(h-n)*z+n
 *   utils/distribution.py:87-123        (sample_from_discretized_mix_logistic)
 *   models/fatchord_version.py:231-237   (softmax -> Categorical.sample == argmax(p/q), q~Exp(1))
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load this library.  It is never
 * linked into, imported by or called from the product (wavernn_amd/).
 *
 * Parity pinning: checked against outputs of the reference itself (the .npz fixtures under tests/golden, produced by
 * scripts/make_golden.py) in tests/test_oracle_golden.py.
 *
 * Arithmetic: float32; every dot product is ONE sequential fmaf chain in ascending k (bias added after
 * the chain).  Rows of a layer are split over OpenMP threads; all B segments advance together so each
 * weight is read once per step (same blocking idea as the reference's batched addmm).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

typedef struct {
    int rnn_dims, fc_dims, feat_dims, aux_dims, n_classes;
    const float *I_w, *I_b;                       /* (H, 1+feat+aux), (H) */
    const float *w_ih1, *w_hh1, *b_ih1, *b_hh1;   /* (3H,H) (3H,H) (3H) (3H) */
    const float *w_ih2, *w_hh2, *b_ih2, *b_hh2;   /* (3H,H+aux) (3H,H) */
    const float *fc1_w, *fc1_b;                   /* (F, H+aux) */
    const float *fc2_w, *fc2_b;                   /* (F, F+aux) */
    const float *fc3_w, *fc3_b;                   /* (C, F) */
} wrnn_oracle_weights;

#define BP 16            /* segments are processed in blocks of BP (vector lanes) */

/* out[r][b] = bias[r] + sum_k W[r][k] * x[k][b]   (x stored k-major, BP wide).
 * RB rows are advanced together only to hide FMA latency; each (r,b) is still one ascending-k fmaf chain. */
#define RB 4
static void matvec_block(const float *W, const float *bias, int rows, int K, const float *xT, float *outT)
{
#pragma omp for schedule(static)
    for (int r0 = 0; r0 < rows; r0 += RB) {
        float acc[RB][BP];
        const int nr = rows - r0 < RB ? rows - r0 : RB;
        for (int i = 0; i < RB; ++i) for (int b = 0; b < BP; ++b) acc[i][b] = 0.f;
        if (nr == RB) {
            const float *w0 = W + (size_t)r0 * K, *w1 = w0 + K, *w2 = w1 + K, *w3 = w2 + K;
            for (int k = 0; k < K; ++k) {
                const float *x = xT + (size_t)k * BP;
                const float a0 = w0[k], a1 = w1[k], a2 = w2[k], a3 = w3[k];
                for (int b = 0; b < BP; ++b) {
                    acc[0][b] = fmaf(a0, x[b], acc[0][b]);
                    acc[1][b] = fmaf(a1, x[b], acc[1][b]);
                    acc[2][b] = fmaf(a2, x[b], acc[2][b]);
                    acc[3][b] = fmaf(a3, x[b], acc[3][b]);
                }
            }
        } else {
            for (int i = 0; i < nr; ++i) {
                const float *w = W + (size_t)(r0 + i) * K;
                for (int k = 0; k < K; ++k) {
                    const float wk = w[k];
                    const float *x = xT + (size_t)k * BP;
                    for (int b = 0; b < BP; ++b) acc[i][b] = fmaf(wk, x[b], acc[i][b]);
                }
            }
        }
        for (int i = 0; i < nr; ++i)
            for (int b = 0; b < BP; ++b) outT[(size_t)(r0 + i) * BP + b] = acc[i][b] + bias[r0 + i];
    }
}

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* mode: 0 = RAW, 1 = MOL.
 * mels [B,T,feat], aux [B,T,4*aux_dims], out [B,T].
 * MOL noise: u1 [T,B,10], u2 [T,B].  RAW noise: q [T,B,C] (u2 unused).
 * dbg_logits (optional, may be NULL): [T,B,C].  Returns 0 on success. */
int wrnn_oracle_loop(const wrnn_oracle_weights *w, int mode, int B, int T,
                     const float *mels, const float *aux, const float *n1, const float *n2,
                     float *out, float *dbg_logits, int nthreads)
{
    const int H = w->rnn_dims, F = w->fc_dims, M = w->feat_dims, A = w->aux_dims, C = w->n_classes;
    const int KI = 1 + M + A, K2 = H + A, K3 = H + A, K4 = F + A;
    if (mode == 1 && C != 30) return -1;
    if (nthreads > 0) omp_set_num_threads(nthreads);
    const float log_scale_min = (float)log(1e-14);

    for (int b0 = 0; b0 < B; b0 += BP) {
        const int nb = (B - b0 < BP) ? (B - b0) : BP;
        float *in0 = calloc((size_t)KI * BP, 4), *xi = calloc((size_t)H * BP, 4);
        float *h1 = calloc((size_t)H * BP, 4), *h2 = calloc((size_t)H * BP, 4);
        float *gi = calloc((size_t)3 * H * BP, 4), *gh = calloc((size_t)3 * H * BP, 4);
        float *in2 = calloc((size_t)K2 * BP, 4), *in3 = calloc((size_t)K3 * BP, 4), *in4 = calloc((size_t)K4 * BP, 4);
        float *y2 = calloc((size_t)F * BP, 4), *lg = calloc((size_t)C * BP, 4);
        float *tmp = calloc((size_t)(H > F ? H : F) * BP, 4);
        float xprev[BP];
        for (int b = 0; b < BP; ++b) xprev[b] = 0.f;

#pragma omp parallel
        for (int t = 0; t < T; ++t) {
            /* :203-209  x = I(cat[x, m_t, a1_t]) */
#pragma omp for schedule(static)
            for (int k = 0; k < KI; ++k)
                for (int b = 0; b < nb; ++b) {
                    const size_t bt = (size_t)(b0 + b) * T + t;
                    float v;
                    if (k == 0) v = xprev[b];
                    else if (k <= M) v = mels[bt * M + (k - 1)];
                    else v = aux[bt * 4 * A + (k - 1 - M)];
                    in0[(size_t)k * BP + b] = v;
                }
            matvec_block(w->I_w, w->I_b, H, KI, in0, xi);
            /* :210  h1 = rnn1(x, h1) */
            matvec_block(w->w_ih1, w->b_ih1, 3 * H, H, xi, gi);
            matvec_block(w->w_hh1, w->b_hh1, 3 * H, H, h1, gh);
#pragma omp for schedule(static)
            for (int j = 0; j < H; ++j)
                for (int b = 0; b < BP; ++b) {
                    const float r = sigmoidf_(gh[(size_t)j * BP + b] + gi[(size_t)j * BP + b]);
                    const float z = sigmoidf_(gh[(size_t)(H + j) * BP + b] + gi[(size_t)(H + j) * BP + b]);
                    const float n = tanhf(gi[(size_t)(2 * H + j) * BP + b] + gh[(size_t)(2 * H + j) * BP + b] * r);
                    const float h = h1[(size_t)j * BP + b];
                    tmp[(size_t)j * BP + b] = (h - n) * z + n;
                }
#pragma omp for schedule(static)
            for (int j = 0; j < K2; ++j)
                for (int b = 0; b < BP; ++b) {
                    if (j < H) {
                        const float hn = tmp[(size_t)j * BP + b];
                        h1[(size_t)j * BP + b] = hn;
                        in2[(size_t)j * BP + b] = xi[(size_t)j * BP + b] + hn;          /* :212 x = x + h1 */
                    } else
                        in2[(size_t)j * BP + b] = (b < nb) ? aux[((size_t)(b0 + b) * T + t) * 4 * A + A + (j - H)] : 0.f;
                }
            /* :213-214  h2 = rnn2(cat[x, a2_t], h2) */
            matvec_block(w->w_ih2, w->b_ih2, 3 * H, K2, in2, gi);
            matvec_block(w->w_hh2, w->b_hh2, 3 * H, H, h2, gh);
#pragma omp for schedule(static)
            for (int j = 0; j < H; ++j)
                for (int b = 0; b < BP; ++b) {
                    const float r = sigmoidf_(gh[(size_t)j * BP + b] + gi[(size_t)j * BP + b]);
                    const float z = sigmoidf_(gh[(size_t)(H + j) * BP + b] + gi[(size_t)(H + j) * BP + b]);
                    const float n = tanhf(gi[(size_t)(2 * H + j) * BP + b] + gh[(size_t)(2 * H + j) * BP + b] * r);
                    const float h = h2[(size_t)j * BP + b];
                    tmp[(size_t)j * BP + b] = (h - n) * z + n;
                }
#pragma omp for schedule(static)
            for (int j = 0; j < K3; ++j)
                for (int b = 0; b < BP; ++b) {
                    if (j < H) {
                        const float hn = tmp[(size_t)j * BP + b];
                        h2[(size_t)j * BP + b] = hn;
                        in3[(size_t)j * BP + b] = in2[(size_t)j * BP + b] + hn;         /* :216 x = x + h2 */
                    } else
                        in3[(size_t)j * BP + b] = (b < nb) ? aux[((size_t)(b0 + b) * T + t) * 4 * A + 2 * A + (j - H)] : 0.f;
                }
            /* :217-218  x = relu(fc1(cat[x, a3_t])) */
            matvec_block(w->fc1_w, w->fc1_b, F, K3, in3, tmp);
#pragma omp for schedule(static)
            for (int j = 0; j < K4; ++j)
                for (int b = 0; b < BP; ++b) {
                    if (j < F) { const float v = tmp[(size_t)j * BP + b]; in4[(size_t)j * BP + b] = v > 0.f ? v : 0.f; }
                    else in4[(size_t)j * BP + b] = (b < nb) ? aux[((size_t)(b0 + b) * T + t) * 4 * A + 3 * A + (j - F)] : 0.f;
                }
            /* :220-221  x = relu(fc2(cat[x, a4_t])) */
            matvec_block(w->fc2_w, w->fc2_b, F, K4, in4, y2);
#pragma omp for schedule(static)
            for (int j = 0; j < F; ++j)
                for (int b = 0; b < BP; ++b) { const float v = y2[(size_t)j * BP + b]; y2[(size_t)j * BP + b] = v > 0.f ? v : 0.f; }
            /* :223  logits = fc3(x) */
            matvec_block(w->fc3_w, w->fc3_b, C, F, y2, lg);
            /* sampling */
#pragma omp for schedule(static)
            for (int b = 0; b < nb; ++b) {
                const size_t tb = (size_t)t * B + (b0 + b);
                float s;
                if (dbg_logits) for (int c = 0; c < C; ++c) dbg_logits[tb * C + c] = lg[(size_t)c * BP + b];
                if (mode == 1) {
                    /* distribution.py:103-121 */
                    int km = 0; float best = 0.f;
                    for (int j = 0; j < 10; ++j) {
                        const float u = n1[tb * 10 + j];
                        const float v = lg[(size_t)j * BP + b] - logf(-logf(u));
                        if (j == 0 || v > best) { best = v; km = j; }
                    }
                    const float mean = lg[(size_t)(10 + km) * BP + b];
                    float ls = lg[(size_t)(20 + km) * BP + b];
                    if (ls < log_scale_min) ls = log_scale_min;
                    const float u = n2[tb];
                    s = mean + expf(ls) * (logf(u) - logf(1.f - u));
                    if (s < -1.f) s = -1.f;
                    if (s > 1.f) s = 1.f;
                } else {
                    /* :232-237  softmax, Categorical renormalisation, argmax(p/q) first-max wins */
                    float mx = lg[b];
                    for (int c = 1; c < C; ++c) { const float v = lg[(size_t)c * BP + b]; if (v > mx) mx = v; }
                    float sum = 0.f;
                    float *e = malloc((size_t)C * 4);
                    for (int c = 0; c < C; ++c) { e[c] = expf(lg[(size_t)c * BP + b] - mx); sum += e[c]; }
                    float sum2 = 0.f;
                    for (int c = 0; c < C; ++c) { e[c] = e[c] / sum; sum2 += e[c]; }
                    int km = 0; float best = 0.f;
                    for (int c = 0; c < C; ++c) {
                        const float v = (e[c] / sum2) / n1[tb * C + c];
                        if (c == 0 || v > best) { best = v; km = c; }
                    }
                    free(e);
                    s = 2.f * (float)km / ((float)C - 1.f) - 1.f;
                }
                out[(size_t)(b0 + b) * T + t] = s;
                xprev[b] = s;
            }
        }
        free(in0); free(xi); free(h1); free(h2); free(gi); free(gh); free(in2); free(in3); free(in4);
        free(y2); free(lg); free(tmp);
    }
    return 0;
}
