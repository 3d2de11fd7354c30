/*
 * birda_oracle.c -- CPU restatement of birda's segments->detections hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP product in
 * birda_amd/csrc/.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it; the product never links or calls it.
 *
 * PARITY STATUS: "parity unpinned" at the model boundary.  The reference delegates the
 * spectrogram + conv stack to birdnet-onnx 2.0.0-rc.16 -> ort 2.0.0-rc.12 -> ONNX Runtime
 * 1.24.2 running a downloaded birdnet.onnx (Cargo.lock:481-484,1889-1892; registry.json:20-22);
 * none of that is in /root/reference and the reference's tests pin no classifier output
 * (SURVEY.md 8c).  The model arithmetic below restates the published BirdNET v2.4 front-end
 *  and standard NHWC conv semantics, and is pinned against
 * numpy/scipy/torch-CPU vectors under tests/golden/ (tools/gen_golden.py).  The host-side
 * pieces (segmenter, PCM scaling, source sizing, resampler block loop, batching, threshold,
 * sort, CSV) restate reference code line by line and ARE pinned against the reference's own
 * unit-test expectations (tests/golden/reference_unit_cases.json).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  Plain C99 + libm (+ OpenMP across segments for the cpu_baseline leg).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define BO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------ */
/* BHM1 container (birda_amd/modelfile.py)                                               */
/* ------------------------------------------------------------------------------------ */
enum { OP_CONV = 1, OP_DWCONV = 2, OP_PWCONV = 3, OP_GAP = 4, OP_DENSE = 5, OP_SCALE = 6 };
enum { ACT_NONE, ACT_RELU, ACT_RELU6, ACT_SWISH, ACT_GELU_ERF, ACT_GELU_TANH, ACT_SIGMOID };
enum { OUT_NONE, OUT_SIGMOID, OUT_SOFTMAX };
#define NO_TENSOR 0xFFFFFFFFu

#pragma pack(push, 1)
typedef struct {
    char magic[4];
    uint32_t version, family, sample_rate, sample_count;
    float segment_duration;
    uint32_t n_classes, embedding_dim, n_branches, n_layers, output_activation, embedding_tensor;
    uint64_t blob_offset, blob_floats;
    uint32_t spec_h, spec_w;
    float norm_eps;
} bhm_header;
typedef struct {
    uint32_t frame_length, frame_step, fft_length, n_bins, n_mels, n_frames;
    float fmin, fmax, mag_scale, out_scale, out_shift;
    uint32_t flags;
    uint64_t mel_w_off;
} bhm_branch;
typedef struct {
    uint32_t op, act, in_tensor, res_tensor, cin, cout, kh, kw, sh, sw, pad_t, pad_l;
    uint32_t in_h, in_w, out_h, out_w, in_layout, reserved;
    uint64_t w_off, b_off;
} bhm_layer;
#pragma pack(pop)

typedef struct {
    bhm_header h;
    bhm_branch *br;
    bhm_layer *ly;
    float *blob;
    size_t *tensor_floats; /* per-segment size of tensor i (0..n_layers) */
} bo_model;

BO_API void bo_model_free(bo_model *m) {
    if (!m) return;
    free(m->br); free(m->ly); free(m->blob); free(m->tensor_floats); free(m);
}

BO_API bo_model *bo_model_load(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    bo_model *m = calloc(1, sizeof *m);
    unsigned char hdr[256];
    if (fread(hdr, 1, 256, f) != 256) goto fail;
    memcpy(&m->h, hdr, sizeof m->h);
    if (memcmp(m->h.magic, "BHM1", 4) || m->h.version != 1) goto fail;
    m->br = calloc(m->h.n_branches, sizeof *m->br);
    m->ly = calloc(m->h.n_layers, sizeof *m->ly);
    for (uint32_t i = 0; i < m->h.n_branches; i++) {
        unsigned char rec[64];
        if (fread(rec, 1, 64, f) != 64) goto fail;
        memcpy(&m->br[i], rec, sizeof m->br[i]);
    }
    for (uint32_t i = 0; i < m->h.n_layers; i++) {
        unsigned char rec[128];
        if (fread(rec, 1, 128, f) != 128) goto fail;
        memcpy(&m->ly[i], rec, sizeof m->ly[i]);
    }
    m->blob = malloc(m->h.blob_floats * sizeof(float));
    if (fseek(f, (long)m->h.blob_offset, SEEK_SET)) goto fail;
    if (fread(m->blob, sizeof(float), m->h.blob_floats, f) != m->h.blob_floats) goto fail;
    fclose(f);
    m->tensor_floats = calloc(m->h.n_layers + 1, sizeof(size_t));
    m->tensor_floats[0] = (size_t)m->h.n_branches * m->h.spec_h * m->h.spec_w;
    for (uint32_t i = 0; i < m->h.n_layers; i++)
        m->tensor_floats[i + 1] = (size_t)m->ly[i].out_h * m->ly[i].out_w * m->ly[i].cout;
    return m;
fail:
    fclose(f);
    bo_model_free(m);
    return NULL;
}

/* ModelConfig accessors -- reference src/inference/classifier.rs:360-377 */
BO_API uint32_t bo_sample_rate(const bo_model *m) { return m->h.sample_rate; }
BO_API uint32_t bo_sample_count(const bo_model *m) { return m->h.sample_count; }
BO_API float bo_segment_duration(const bo_model *m) { return m->h.segment_duration; }
BO_API uint32_t bo_n_classes(const bo_model *m) { return m->h.n_classes; }
BO_API uint32_t bo_embedding_dim(const bo_model *m) { return m->h.embedding_dim; }
BO_API uint32_t bo_n_layers(const bo_model *m) { return m->h.n_layers; }
BO_API uint64_t bo_tensor_floats(const bo_model *m, uint32_t t) { return m->tensor_floats[t]; }

/* ------------------------------------------------------------------------------------ */
/* generic mixed-radix complex FFT, double precision (any length)                        */
/* ------------------------------------------------------------------------------------ */
typedef struct { double re, im; } cpx;

/* twiddle table exp(sign*2*pi*i*j/N), cached per thread for the few sizes in use */
typedef struct { int n, sign; cpx *w; } tw_entry;
static __thread tw_entry tw_cache[16];
static __thread int tw_count = 0, tw_evict = 0;

static const cpx *twiddles(int n, int sign) {
    for (int i = 0; i < tw_count; i++)
        if (tw_cache[i].n == n && tw_cache[i].sign == sign) return tw_cache[i].w;
    cpx *w = malloc(sizeof(cpx) * (size_t)n);
    for (int j = 0; j < n; j++) {
        double ang = sign * 2.0 * M_PI * (double)j / (double)n;
        w[j].re = cos(ang); w[j].im = sin(ang);
    }
    int slot;
    if (tw_count < 16) slot = tw_count++;
    else { slot = tw_evict; tw_evict = (tw_evict + 1) % 16; free(tw_cache[slot].w); }
    tw_cache[slot].n = n; tw_cache[slot].sign = sign; tw_cache[slot].w = w;
    return w;
}

/* decimation in time over the smallest prime factor p of n: X[k + q*m] =
 * sum_r W_n^{r(k+qm)} Y_r[k], Y_r = FFT_m(x[r::p]); prime factors use the naive p x p DFT */
static void fft_rec(const cpx *in, cpx *out, int n, int stride, const cpx *tw, int tw_step, cpx *tmp) {
    if (n == 1) { out[0] = in[0]; return; }
    int p = 2;
    while (n % p) p = (p == 2) ? 3 : p + 2;
    const int m = n / p;
    for (int r = 0; r < p; r++)
        fft_rec(in + (size_t)r * stride, out + (size_t)r * m, m, stride * p, tw, tw_step * p, tmp);
    for (int k = 0; k < m; k++) {
        for (int q = 0; q < p; q++) {
            const int kk = k + q * m;
            double sr = 0, si = 0;
            for (int r = 0; r < p; r++) {
                const cpx w = tw[(size_t)(((long long)r * kk) % n) * tw_step];
                const cpx v = out[(size_t)r * m + k];
                sr += v.re * w.re - v.im * w.im;
                si += v.re * w.im + v.im * w.re;
            }
            tmp[q].re = sr; tmp[q].im = si;
        }
        for (int q = 0; q < p; q++) out[(size_t)q * m + k] = tmp[q];
    }
}

/* The same transform for n = 2^s without the recursion: bit-reversed load, then the s butterfly levels bottom-up.  Every output
 * is computed by exactly the operations fft_rec performs for p = 2 -- out[k] = a + b * W_n^{k step}, out[k + m] = a + b *
 * W_n^{(k + m) step}, both twiddles READ FROM THE TABLE (not negated), products and sums in the same order -- so the results are
 * bit-identical to the recursive form (the factor W^0 = (1, 0) of the first operand reproduces it exactly); it only drops the
 * call overhead and the per-element modulo, which were most of the front-end's time (0.24 s of 0.41 s per segment). */
static void fft_pow2(const cpx *in, cpx *out, int n, const cpx *tw) {
    int s = 0;
    while ((1 << s) < n) s++;
    for (int i = 0; i < n; i++) {
        int r = 0;
        for (int b = 0; b < s; b++) r |= ((i >> b) & 1) << (s - 1 - b);
        out[r] = in[i];
    }
    for (int len = 2; len <= n; len <<= 1) {            /* sub-transform length at this level */
        const int m = len >> 1, step = n / len;         /* fft_rec: n = len, tw_step = step */
        for (int base = 0; base < n; base += len)
            for (int k = 0; k < m; k++) {
                const cpx a = out[base + k], b = out[base + k + m];
                const cpx w0 = tw[(size_t)k * step], w1 = tw[(size_t)(k + m) * step];
                cpx lo, hi;
                lo.re = a.re + (b.re * w0.re - b.im * w0.im);
                lo.im = a.im + (b.re * w0.im + b.im * w0.re);
                hi.re = a.re + (b.re * w1.re - b.im * w1.im);
                hi.im = a.im + (b.re * w1.im + b.im * w1.re);
                out[base + k] = lo; out[base + k + m] = hi;
            }
    }
}

/* sign = -1 forward, +1 inverse (unscaled) */
static void fft_any(const cpx *in, cpx *out, int n, int sign) {
    if (n >= 2 && (n & (n - 1)) == 0) { fft_pow2(in, out, n, twiddles(n, sign)); return; }
    cpx *tmp = malloc(sizeof(cpx) * (size_t)(n > 2 ? n : 2));
    fft_rec(in, out, n, 1, twiddles(n, sign), 1, tmp);
    free(tmp);
}

BO_API void bo_fft(const double *in_re, const double *in_im, double *out_re, double *out_im, int n, int sign) {
    cpx *a = malloc(sizeof(cpx) * n), *b = malloc(sizeof(cpx) * n);
    for (int i = 0; i < n; i++) { a[i].re = in_re[i]; a[i].im = in_im ? in_im[i] : 0.0; }
    fft_any(a, b, n, sign);
    for (int i = 0; i < n; i++) { out_re[i] = b[i].re; out_im[i] = b[i].im; }
    free(a); free(b);
}

/* ------------------------------------------------------------------------------------ */
/* Front-end: BirdNET v2.4 spectrogram layer, SURVEY.md Appendix B [EXT]                 */
/*   x <- (x - min)/(max - min + eps); x <- 2(x - 0.5)                                   */
/*   stft(frame_length=L, frame_step=H, fft_length=L, periodic hann, pad_end=False)      */
/*   complex -> float keeps Re(); spec = Re(stft) . mel_W ; spec^2 ; ^(1/(1+exp(mag)))   */
/*   reverse mel axis; transpose to [mel, time]; optional folded-BN affine                */
/* output layout: spec[branch][mel][frame]                                               */
/* ------------------------------------------------------------------------------------ */
BO_API void bo_frontend(const bo_model *m, const float *seg, float *spec) {
    const uint32_t n = m->h.sample_count;
    float mn = seg[0], mx = seg[0];
    for (uint32_t i = 1; i < n; i++) { if (seg[i] < mn) mn = seg[i]; if (seg[i] > mx) mx = seg[i]; }
    float *x = malloc(sizeof(float) * n);
    const float denom = (mx - mn) + m->h.norm_eps;
    for (uint32_t i = 0; i < n; i++) {
        float v = (seg[i] - mn) / denom;
        v = v - 0.5f;
        x[i] = v * 2.0f;
    }
    for (uint32_t b = 0; b < m->h.n_branches; b++) {
        const bhm_branch *br = &m->br[b];
        const int L = (int)br->frame_length, H = (int)br->frame_step, nb = (int)br->n_bins;
        const int nm = (int)br->n_mels, nf = (int)br->n_frames;
        const float *W = m->blob + br->mel_w_off; /* [n_bins][n_mels] */
        const float expo = 1.0f / (1.0f + expf(br->mag_scale));
        float *win = malloc(sizeof(float) * L);
        for (int i = 0; i < L; i++) win[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * i / L));
        cpx *fin = malloc(sizeof(cpx) * L), *fout = malloc(sizeof(cpx) * L);
        float *re = malloc(sizeof(float) * nb), *macc = malloc(sizeof(float) * nm);
        float *out = spec + (size_t)b * nm * nf;
        for (int t = 0; t < nf; t++) {
            for (int i = 0; i < L; i++) { fin[i].re = (double)(x[(size_t)t * H + i] * win[i]); fin[i].im = 0; }
            fft_any(fin, fout, L, -1);
            for (int k = 0; k < nb; k++) re[k] = (float)fout[k].re;
            /* spec = Re(stft) . mel_W: every mel bin's sum runs over k ascending, as a dense MatMul does; the loops are ordered k
             * outer / j inner so that the rows of W are read contiguously (the sums themselves are unchanged) */
            for (int j = 0; j < nm; j++) macc[j] = 0.0f;
            for (int k = 0; k < nb; k++) {
                const float rk = re[k];
                const float *wr = W + (size_t)k * nm;
                for (int j = 0; j < nm; j++) macc[j] += rk * wr[j];
            }
            for (int j = 0; j < nm; j++) {
                float acc = macc[j];
                float p = acc * acc;
                float v = powf(p, expo);
                v = v * br->out_scale + br->out_shift;
                int row = (br->flags & 1u) ? (nm - 1 - j) : j;
                out[(size_t)row * nf + t] = v;
            }
        }
        free(win); free(fin); free(fout); free(re); free(macc);
    }
    free(x);
}

/* ------------------------------------------------------------------------------------ */
/* conv stack: NHWC, BN folded, bias + activation (+ residual after activation)          */
/* ------------------------------------------------------------------------------------ */
static inline float act_apply(float v, uint32_t act) {
    switch (act) {
    case ACT_RELU: return v > 0 ? v : 0;
    case ACT_RELU6: return v < 0 ? 0 : (v > 6 ? 6 : v);
    case ACT_SWISH: return v / (1.0f + expf(-v));
    case ACT_GELU_ERF: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    case ACT_GELU_TANH: return 0.5f * v * (1.0f + tanhf(0.7978845608028654f * (v + 0.044715f * v * v * v)));
    case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
    }
}

__attribute__((target_clones("avx2", "default")))
static void layer_conv(const bhm_layer *L, const float *W, const float *B, const float *in, float *out) {
    const int ih = L->in_h, iw = L->in_w, oh = L->out_h, ow = L->out_w, ci = L->cin, co = L->cout;
    const int kh = L->kh, kw = L->kw, sh = L->sh, sw = L->sw, pt = L->pad_t, pl = L->pad_l;
    for (int y = 0; y < oh; y++)
        for (int x = 0; x < ow; x++) {
            float *o = out + ((size_t)y * ow + x) * co;
            for (int n = 0; n < co; n++) o[n] = B[n];
            for (int dy = 0; dy < kh; dy++) {
                int iy = y * sh - pt + dy;
                if (iy < 0 || iy >= ih) continue;
                for (int dx = 0; dx < kw; dx++) {
                    int ix = x * sw - pl + dx;
                    if (ix < 0 || ix >= iw) continue;
                    for (int c = 0; c < ci; c++) {
                        float a = L->in_layout == 1 ? in[((size_t)c * ih + iy) * iw + ix]
                                                    : in[((size_t)iy * iw + ix) * ci + c];
                        const float *w = W + (((size_t)dy * kw + dx) * ci + c) * co;
                        for (int n = 0; n < co; n++) o[n] += a * w[n];
                    }
                }
            }
        }
}

__attribute__((target_clones("avx2", "default")))
static void layer_dwconv(const bhm_layer *L, const float *W, const float *B, const float *in, float *out) {
    const int ih = L->in_h, iw = L->in_w, oh = L->out_h, ow = L->out_w, c = L->cout;
    const int kh = L->kh, kw = L->kw, sh = L->sh, sw = L->sw, pt = L->pad_t, pl = L->pad_l;
    for (int y = 0; y < oh; y++)
        for (int x = 0; x < ow; x++) {
            float *o = out + ((size_t)y * ow + x) * c;
            for (int n = 0; n < c; n++) o[n] = B[n];
            for (int dy = 0; dy < kh; dy++) {
                int iy = y * sh - pt + dy;
                if (iy < 0 || iy >= ih) continue;
                for (int dx = 0; dx < kw; dx++) {
                    int ix = x * sw - pl + dx;
                    if (ix < 0 || ix >= iw) continue;
                    const float *a = in + ((size_t)iy * iw + ix) * c;
                    const float *w = W + ((size_t)dy * kw + dx) * c;
                    for (int n = 0; n < c; n++) o[n] += a[n] * w[n];
                }
            }
        }
}

/* 1x1 convolution / dense layer, out[r][n] = B[n] + sum_k in[r][k] W[k][n] with the sum over k ASCENDING and a separate
 * multiply and add per term (-ffp-contract=off) -- the order the first, row-at-a-time version of this function used and the
 * parity tests were pinned on.  This version computes 4 rows x 32 columns at a time in registers: W is read once per four rows
 * instead of once per row and the output row is no longer re-loaded for every k.  Per output element the operations and their
 * order are unchanged, so the results are bit-identical; it is what lets the `cpu_baseline` leg of bench.py run within a small
 * factor of an optimised CPU runtime instead of 27x below it. */
#define PW_RB 4
#define PW_NB 32
__attribute__((target_clones("avx512f", "avx2", "default")))
static void layer_pw(int rows, int ci, int co, const float *W, const float *B, const float *in, float *out) {
    for (int n0 = 0; n0 < co; n0 += PW_NB) {
        const int nn = co - n0 < PW_NB ? co - n0 : PW_NB;
        int r0 = 0;
        if (nn == PW_NB) {
            for (; r0 + PW_RB <= rows; r0 += PW_RB) {
                float acc[PW_RB][PW_NB];
                for (int r = 0; r < PW_RB; r++)
                    for (int j = 0; j < PW_NB; j++) acc[r][j] = B[n0 + j];
                const float *a0 = in + (size_t)r0 * ci;
                for (int k = 0; k < ci; k++) {
                    const float *w = W + (size_t)k * co + n0;
                    for (int r = 0; r < PW_RB; r++) {
                        const float av = a0[(size_t)r * ci + k];
                        for (int j = 0; j < PW_NB; j++) acc[r][j] += av * w[j];
                    }
                }
                for (int r = 0; r < PW_RB; r++)
                    for (int j = 0; j < PW_NB; j++) out[(size_t)(r0 + r) * co + n0 + j] = acc[r][j];
            }
        }
        for (; r0 < rows; r0++) {      /* remaining rows, and every row of a ragged last column block */
            float acc[PW_NB];
            for (int j = 0; j < nn; j++) acc[j] = B[n0 + j];
            const float *a = in + (size_t)r0 * ci;
            for (int k = 0; k < ci; k++) {
                const float av = a[k];
                const float *w = W + (size_t)k * co + n0;
                for (int j = 0; j < nn; j++) acc[j] += av * w[j];
            }
            for (int j = 0; j < nn; j++) out[(size_t)r0 * co + n0 + j] = acc[j];
        }
    }
}

static void layer_gap(const bhm_layer *L, const float *in, float *out) {
    const int px = L->in_h * L->in_w, c = L->cout;
    for (int n = 0; n < c; n++) out[n] = 0.0f;
    for (int p = 0; p < px; p++)
        for (int n = 0; n < c; n++) out[n] += in[(size_t)p * c + n];
    const float inv = 1.0f / (float)px;
    for (int n = 0; n < c; n++) out[n] *= inv;
}

/* Runs one segment through front-end + stack.  tensors[i] must hold tensor_floats[i]. */
static void forward_one(const bo_model *m, const float *seg, float **tensors) {
    bo_frontend(m, seg, tensors[0]);
    for (uint32_t i = 0; i < m->h.n_layers; i++) {
        const bhm_layer *L = &m->ly[i];
        const float *in = tensors[L->in_tensor];
        float *out = tensors[i + 1];
        const float *W = m->blob + L->w_off, *B = m->blob + L->b_off;
        const size_t nout = m->tensor_floats[i + 1];
        switch (L->op) {
        case OP_CONV: layer_conv(L, W, B, in, out); break;
        case OP_DWCONV: layer_dwconv(L, W, B, in, out); break;
        case OP_PWCONV: layer_pw((int)(L->out_h * L->out_w), (int)L->cin, (int)L->cout, W, B, in, out); break;
        case OP_DENSE: layer_pw(1, (int)L->cin, (int)L->cout, W, B, in, out); break;
        case OP_GAP: layer_gap(L, in, out); break;
        case OP_SCALE: {   /* squeeze-excite: every pixel's channels times the gate [C] (tensor res_tensor); ONNX Mul(x, gate) */
            const float *gate = tensors[L->res_tensor];
            const int px = (int)(L->out_h * L->out_w), c = (int)L->cout;
            for (int p = 0; p < px; p++)
                for (int n = 0; n < c; n++) out[(size_t)p * c + n] = in[(size_t)p * c + n] * gate[n];
            continue;      /* no bias, activation or residual on this op: res_tensor is the gate */
        }
        default: break;
        }
        if (L->act != ACT_NONE)
            for (size_t j = 0; j < nout; j++) out[j] = act_apply(out[j], L->act);
        if (L->res_tensor != NO_TENSOR) {
            const float *r = tensors[L->res_tensor];
            for (size_t j = 0; j < nout; j++) out[j] += r[j];
        }
    }
}

/*
 * The classifier forward -- stands in for birdnet_onnx::Classifier::predict_batch as called
 * from reference src/inference/classifier.rs:478-488: independent segments in, one row of
 * logits out per segment, order preserved.  `dump_tensor` >= 0 additionally copies that
 * tensor (per segment) to `dump` for layer-level parity; embeddings = tensor
 * header.embedding_tensor (1024-d for v2.4, README.md:574).
 */
BO_API int bo_forward(const bo_model *m, const float *segs, int n, float *logits, float *embeddings,
                      int dump_tensor, float *dump) {
    const uint32_t nt = m->h.n_layers + 1;
    int rc = 0;
    /* every thread keeps ONE set of tensor buffers for all the segments it takes: allocating the ~50 tensors (21 MB) afresh per
     * segment meant an mmap / page-fault / munmap round per tensor and segment, which 256 host threads serialised on */
#pragma omp parallel
    {
        float **t = malloc(sizeof(float *) * nt);
        for (uint32_t i = 0; i < nt; i++) t[i] = malloc(sizeof(float) * (m->tensor_floats[i] ? m->tensor_floats[i] : 1));
#pragma omp for schedule(dynamic, 1)
        for (int s = 0; s < n; s++) {
            forward_one(m, segs + (size_t)s * m->h.sample_count, t);
            memcpy(logits + (size_t)s * m->h.n_classes, t[nt - 1], sizeof(float) * m->h.n_classes);
            if (embeddings)
                memcpy(embeddings + (size_t)s * m->h.embedding_dim, t[m->h.embedding_tensor], sizeof(float) * m->h.embedding_dim);
            if (dump && dump_tensor >= 0 && (uint32_t)dump_tensor < nt)
                memcpy(dump + (size_t)s * m->tensor_floats[dump_tensor], t[dump_tensor], sizeof(float) * m->tensor_floats[dump_tensor]);
        }
        for (uint32_t i = 0; i < nt; i++) free(t[i]);
        free(t);
    }
    return rc;
}

/*
 * Activation + top-k + min-confidence -- birdnet_onnx PredictionResult as consumed at
 * reference src/pipeline/processor.rs:363-385 and configured at classifier.rs:269-283
 * (top_k = 5, constants.rs:178; min_confidence 0.1, constants.rs:25) [EXT for internals]:
 * v2.4 applies sigmoid to logits, Perch softmax.  Ties break to the lower
 * class index.  Returns the number of predictions kept (<= top_k), confidence descending.
 */
BO_API int bo_topk(const float *logits, int n_classes, int out_act, int top_k, float min_conf,
                   int *idx, float *conf) {
    float *p = malloc(sizeof(float) * n_classes);
    if (out_act == OUT_SIGMOID) {
        for (int i = 0; i < n_classes; i++) p[i] = 1.0f / (1.0f + expf(-logits[i]));
    } else if (out_act == OUT_SOFTMAX) {
        float mx = logits[0];
        for (int i = 1; i < n_classes; i++) if (logits[i] > mx) mx = logits[i];
        float sum = 0.0f;
        for (int i = 0; i < n_classes; i++) { p[i] = expf(logits[i] - mx); sum += p[i]; }
        for (int i = 0; i < n_classes; i++) p[i] /= sum;
    } else {
        memcpy(p, logits, sizeof(float) * n_classes);
    }
    int kept = 0;
    for (int k = 0; k < top_k && k < n_classes; k++) {
        int best = -1;
        for (int i = 0; i < n_classes; i++) {
            /* rank on the logit (monotone in p, avoids saturated-sigmoid ties) then index */
            if (isnan(p[i])) continue;
            int taken = 0;
            for (int j = 0; j < kept; j++) if (idx[j] == i) { taken = 1; break; }
            if (taken) continue;
            if (best < 0 || logits[i] > logits[best]) best = i;
        }
        if (best < 0) break;
        if (!(p[best] >= min_conf)) break;
        idx[kept] = best; conf[kept] = p[best]; kept++;
    }
    free(p);
    return kept;
}

/* ------------------------------------------------------------------------------------ */
/* Host path restatements                                                                */
/* ------------------------------------------------------------------------------------ */

/* append_samples -- reference src/audio/decode.rs:353-411.  Interleaved PCM in, mono f32 out.
 * S16: s / 32768.0 (:372-374); S32: s as f32 / 2147483648.0 (:388-391); F32 passthrough.
 * Multi-channel: sum of per-channel scaled values, then / channels (:360-368).           */
BO_API void bo_pcm16_to_mono(const int16_t *pcm, size_t frames, int channels, float *out) {
    for (size_t i = 0; i < frames; i++) {
        if (channels == 1) { out[i] = (float)pcm[i] / 32768.0f; continue; }
        float sum = 0.0f;
        for (int c = 0; c < channels; c++) sum += (float)pcm[i * channels + c] / 32768.0f;
        out[i] = sum / (float)channels;
    }
}
BO_API void bo_pcm32_to_mono(const int32_t *pcm, size_t frames, int channels, float *out) {
    for (size_t i = 0; i < frames; i++) {
        if (channels == 1) { out[i] = (float)pcm[i] / 2147483648.0f; continue; }
        float sum = 0.0f;
        for (int c = 0; c < channels; c++) sum += (float)pcm[i * channels + c] / 2147483648.0f;
        out[i] = sum / (float)channels;
    }
}
BO_API void bo_f32_to_mono(const float *pcm, size_t frames, int channels, float *out) {
    for (size_t i = 0; i < frames; i++) {
        if (channels == 1) { out[i] = pcm[i]; continue; }
        float sum = 0.0f;
        for (int c = 0; c < channels; c++) sum += pcm[i * channels + c];
        out[i] = sum / (float)channels;
    }
}

/*
 * StreamingDecoder::next_segment -- reference src/audio/decode.rs:150-202, over an in-memory
 * mono sample stream delivered in `packet` sized pieces (decode_next_packet, :205-245).
 */
typedef struct {
    const float *src; size_t src_len, src_pos, packet;
    float *buf; size_t buf_len, buf_cap;
    size_t samples_emitted; int eof;
} bo_segmenter;

BO_API bo_segmenter *bo_segmenter_new(const float *samples, size_t len, size_t packet) {
    bo_segmenter *s = calloc(1, sizeof *s);
    s->src = samples; s->src_len = len; s->packet = packet ? packet : 1152;
    return s;
}
BO_API void bo_segmenter_free(bo_segmenter *s) { if (s) { free(s->buf); free(s); } }

/* returns 1 and fills out[segment_samples], *start_sample; 0 when exhausted; -1 on
 * overlap >= segment (Error::Internal, decode.rs:156-162). */
BO_API int bo_segmenter_next(bo_segmenter *s, size_t segment_samples, size_t overlap_samples,
                             float *out, size_t *start_sample) {
    if (overlap_samples >= segment_samples) return -1;
    while (s->buf_len < segment_samples && !s->eof) { /* :165-167 */
        if (s->src_pos >= s->src_len) { s->eof = 1; break; }
        size_t take = s->src_len - s->src_pos < s->packet ? s->src_len - s->src_pos : s->packet;
        if (s->buf_len + take > s->buf_cap) {
            s->buf_cap = (s->buf_len + take) * 2;
            s->buf = realloc(s->buf, s->buf_cap * sizeof(float));
        }
        memcpy(s->buf + s->buf_len, s->src + s->src_pos, take * sizeof(float));
        s->buf_len += take; s->src_pos += take;
    }
    if (s->buf_len == 0) return 0; /* :170-172 */
    size_t take = segment_samples < s->buf_len ? segment_samples : s->buf_len; /* :175 */
    memcpy(out, s->buf, take * sizeof(float));
    for (size_t i = take; i < segment_samples; i++) out[i] = 0.0f; /* :178-181 */
    *start_sample = s->samples_emitted; /* :183 */
    size_t advance = take > overlap_samples ? take - overlap_samples : 0; /* saturating_sub :186 */
    if (advance > 0) {
        memmove(s->buf, s->buf + advance, (s->buf_len - advance) * sizeof(float));
        s->buf_len -= advance; s->samples_emitted += advance;
    } else { /* :191-196 */
        s->buf_len = 0; s->samples_emitted += take;
    }
    return 1;
}

/* decode_and_stream source sizing -- reference src/pipeline/processor.rs:67-82 */
BO_API size_t bo_source_samples(size_t target_samples, uint32_t source_rate, uint32_t target_rate) {
    if (source_rate == target_rate) return target_samples;
    return (size_t)ceil((double)target_samples * (double)source_rate / (double)target_rate);
}

/* segment_samples / overlap_samples -- reference src/pipeline/processor.rs:514,520:
 * (segment_duration * target_rate as f32) as usize, all in f32 */
BO_API size_t bo_duration_to_samples(float seconds, uint32_t rate) {
    float v = seconds * (float)rate;
    return v <= 0.0f ? 0 : (size_t)v;
}

/* estimate_segment_count -- reference src/output/progress.rs:80-92; -1 encodes None */
BO_API int64_t bo_estimate_segment_count(int has_duration, double duration, float seg, float overlap) {
    if (!has_duration) return -1;
    float step = seg - overlap;
    if (step <= 0.0f) return -1;
    return (int64_t)ceil(duration / (double)step);
}

/* effective batch size -- reference src/pipeline/processor.rs:531-545 */
BO_API size_t bo_effective_batch_size(size_t batch_size, int64_t estimated_segments) {
    if (estimated_segments < 0) return batch_size;
    size_t est = (size_t)estimated_segments;
    if (est == 0) return batch_size;
    return batch_size > est ? est : batch_size;
}

/* chunk times -- reference src/pipeline/processor.rs:90-94 (all f32) */
BO_API void bo_chunk_times(size_t start_sample, uint32_t source_rate, size_t segment_samples,
                           uint32_t target_rate, float *start_time, float *end_time) {
    float st = (float)start_sample / (float)source_rate;
    float dur = (float)segment_samples / (float)target_rate;
    *start_time = st; *end_time = st + dur;
}

/*
 * chunk_audio -- reference src/audio/chunker.rs:26-75 (exported, unused by the pipeline;
 * restated because the reference's only exact segment-count vectors live in its tests
 * :77-125).  Returns chunk count; start times written to `starts` (cap entries).
 */
BO_API size_t bo_chunk_audio_count(size_t n_samples, uint32_t rate, float chunk_duration, float overlap,
                                   float *starts, size_t cap) {
    size_t chunk = bo_duration_to_samples(chunk_duration, rate);
    size_t ovl = bo_duration_to_samples(overlap, rate);
    size_t step = chunk > ovl ? chunk - ovl : 0;
    if (step == 0) return 0;
    size_t count = 0;
    for (size_t pos = 0; pos < n_samples; pos += step) {
        if (starts && count < cap) starts[count] = (float)pos / (float)rate;
        count++;
    }
    return count;
}

/* ------------------------------------------------------------------------------------ */
/* Resampler: rubato 4.0.0 Fft<f32>, FixedSync::Both, chunk 1024, as driven by reference   */
/* src/audio/resample.rs:10-91.  rubato is not vendored (Cargo.lock:2300-2303) [EXT]; the  */
/* synchronous FFT resampler algorithm restated here is rubato's published one:            */
/*   gcd = gcd(fs_in, fs_out); fft_chunks = ceil(chunk / (fs_in/gcd))                      */
/*   fft_in = fft_chunks*fs_in/gcd, fft_out = fft_chunks*fs_out/gcd                        */
/*     (48k->32k: 342 chunks; 44.1k->32k: 3 -- matches resample.rs:311-316)                */
/*   filter: BlackmanHarris^2-windowed sinc of length fft_in, cutoff                       */
/*     0.4^(16/fft_in) (x fft_out/fft_in when downsampling), unit sum, scaled 1/(2 fft_in)           */
/*   per block: zero-pad to 2 fft_in, rFFT, multiply by filter spectrum, keep the first    */
/*     min(fft_in, fft_out)+... bins, irFFT at 2 fft_out, overlap-add halves.              */
/* Computed in double and rounded to f32 at block output (rubato computes in f32).        */
/* ------------------------------------------------------------------------------------ */
static unsigned gcd_u(unsigned a, unsigned b) { while (b) { unsigned t = a % b; a = b; b = t; } return a; }

typedef struct {
    int fft_in, fft_out;
    cpx *filter_f;     /* fft_in + 1 bins of the 2*fft_in real FFT of the filter */
    double *overlap;   /* fft_out */
} bo_resampler;

static double blackman_harris2(int i, int n) {
    /* rubato windows.rs: BlackmanHarris (4-term, periodic over n) squared */
    const double a0 = 0.35875, a1 = 0.48829, a2 = 0.14128, a3 = 0.01168;
    double x = 2.0 * M_PI * (double)i / (double)n;
    double w = a0 - a1 * cos(x) + a2 * cos(2 * x) - a3 * cos(3 * x);
    return w * w;
}

BO_API void bo_resampler_sizes(uint32_t from, uint32_t to, int chunk, int *fft_in, int *fft_out) {
    unsigned g = gcd_u(from, to);
    unsigned min_in = from / g;
    unsigned chunks = (unsigned)ceil((double)chunk / (double)min_in);
    *fft_in = (int)(chunks * (from / g));
    *fft_out = (int)(chunks * (to / g));
}

static bo_resampler *resampler_new(uint32_t from, uint32_t to, int chunk) {
    bo_resampler *r = calloc(1, sizeof *r);
    bo_resampler_sizes(from, to, chunk, &r->fft_in, &r->fft_out);
    const int ni = r->fft_in, no = r->fft_out;
    double cutoff = ni > no ? (double)powf(0.4f, 16.0f / (float)ni) * (double)no / (double)ni : (double)powf(0.4f, 16.0f / (float)ni);
    /* make_sincs(npoints = fft_in, factor = 1, cutoff, BlackmanHarris2): one sinc centred at
       npoints/2, normalised to unit sum, then scaled by 1/(2 fft_in) for the FFT round trip */
    double *sinc = calloc((size_t)2 * ni, sizeof(double));
    double sum = 0.0;
    for (int i = 0; i < ni; i++) {
        double x = (double)i - (double)(ni / 2); /* integer centre, as totpoints / 2 */
        double arg = M_PI * cutoff * x;
        double s = fabs(arg) < 1e-12 ? 1.0 : sin(arg) / arg;
        sinc[i] = s * blackman_harris2(i, ni);
        sum += sinc[i];
    }
    for (int i = 0; i < ni; i++) sinc[i] = sinc[i] / sum / (2.0 * ni);
    cpx *tin = calloc((size_t)2 * ni, sizeof(cpx)), *tout = calloc((size_t)2 * ni, sizeof(cpx));
    for (int i = 0; i < ni; i++) tin[i].re = sinc[i];
    fft_any(tin, tout, 2 * ni, -1);
    r->filter_f = malloc(sizeof(cpx) * (ni + 1));
    for (int k = 0; k <= ni; k++) r->filter_f[k] = tout[k];
    r->overlap = calloc(no, sizeof(double));
    free(sinc); free(tin); free(tout);
    return r;
}

static void resampler_free(bo_resampler *r) { if (r) { free(r->filter_f); free(r->overlap); free(r); } }

/* one block: fft_in samples in -> fft_out samples out */
static void resampler_block(bo_resampler *r, const float *in, float *out) {
    const int ni = r->fft_in, no = r->fft_out;
    cpx *a = calloc((size_t)2 * ni, sizeof(cpx)), *A = calloc((size_t)2 * ni, sizeof(cpx));
    for (int i = 0; i < ni; i++) a[i].re = in[i];
    fft_any(a, A, 2 * ni, -1);
    /* spectrum of the 2*fft_out real output: bins 0..fft_out; copy the first new_len bins */
    const int new_len = ni < no ? ni + 1 : no;
    cpx *B = calloc((size_t)2 * no, sizeof(cpx)), *b = calloc((size_t)2 * no, sizeof(cpx));
    for (int k = 0; k < new_len; k++) {
        cpx f = r->filter_f[k], v = A[k];
        B[k].re = v.re * f.re - v.im * f.im;
        B[k].im = v.re * f.im + v.im * f.re;
    }
    /* a real inverse FFT ignores Im of DC / Nyquist and mirrors the rest */
    B[0].im = 0.0;
    if (new_len > no) B[no].im = 0.0;
    for (int k = 1; k < no; k++) { B[2 * no - k].re = B[k].re; B[2 * no - k].im = -B[k].im; }
    fft_any(B, b, 2 * no, +1);
    for (int i = 0; i < no; i++) {
        out[i] = (float)(b[i].re + r->overlap[i]);
        r->overlap[i] = b[no + i].re;
    }
    free(a); free(A); free(B); free(b);
}

/*
 * resample -- reference src/audio/resample.rs:10-91.  Identity when rates are equal (:11-13);
 * a NEW resampler per call (:19-28); whole blocks (:34-55); the last partial block is
 * zero-padded and only ceil(remaining*to/from) outputs are kept (:58-88).
 * Returns the output length; `out` must hold bo_resample_max_len().
 */
BO_API size_t bo_resample_max_len(size_t n, uint32_t from, uint32_t to) {
    if (from == to) return n;
    int fi, fo;
    bo_resampler_sizes(from, to, 1024, &fi, &fo);
    return (n / (size_t)fi + 1) * (size_t)fo;
}

BO_API size_t bo_resample(const float *in, size_t n, uint32_t from, uint32_t to, float *out) {
    if (from == to) { memcpy(out, in, n * sizeof(float)); return n; }
    bo_resampler *r = resampler_new(from, to, 1024);
    const size_t need = (size_t)r->fft_in;
    size_t pos = 0, olen = 0;
    while (pos + need <= n) {
        resampler_block(r, in + pos, out + olen);
        olen += (size_t)r->fft_out; pos += need;
    }
    if (pos < n) {
        size_t remaining = n - pos;
        float *padded = calloc(need, sizeof(float));
        memcpy(padded, in + pos, remaining * sizeof(float));
        float *tmp = malloc(sizeof(float) * r->fft_out);
        resampler_block(r, padded, tmp);
        size_t frames = (size_t)ceil((double)remaining * (double)to / (double)from);
        if (frames > (size_t)r->fft_out) frames = (size_t)r->fft_out;
        memcpy(out + olen, tmp, frames * sizeof(float));
        olen += frames;
        free(padded); free(tmp);
    }
    resampler_free(r);
    return olen;
}

/* ------------------------------------------------------------------------------------ */
/* Detections, sort, CSV                                                                 */
/* ------------------------------------------------------------------------------------ */
typedef struct { float start, end, conf; int label; } bo_detection;

/* run_streaming_inference final sort -- reference src/pipeline/processor.rs:178-187:
 * start_time ascending, then confidence descending (ties: input order, i.e. stable) */
static int det_less(const bo_detection *a, const bo_detection *b) {
    if (a->start < b->start) return 1;
    if (a->start > b->start) return 0;
    return a->conf > b->conf;
}
BO_API void bo_sort_detections(bo_detection *d, size_t n) {
    for (size_t i = 1; i < n; i++) { /* insertion sort: stable, n is small */
        bo_detection k = d[i]; size_t j = i;
        while (j > 0 && det_less(&k, &d[j - 1])) { d[j] = d[j - 1]; j--; }
        d[j] = k;
    }
}

/* escape_csv -- reference src/output/csv.rs:126-132 */
static size_t csv_escape(const char *v, char *out) {
    if (strchr(v, ',') || strchr(v, '"') || strchr(v, '\n')) {
        size_t o = 0; out[o++] = '"';
        for (const char *p = v; *p; p++) { if (*p == '"') out[o++] = '"'; out[o++] = *p; }
        out[o++] = '"'; out[o] = 0; return o;
    }
    strcpy(out, v); return strlen(v);
}

/* Detection::from_label split at the first '_' (reference src/output/types.rs:58-79) and
 * CsvWriter::write_detection row "{:.1},{:.1},sci,common,{:.4},path" (csv.rs:55-66).
 * Returns bytes written (without NUL). */
BO_API size_t bo_csv_row(const char *label, float start, float end, float conf, const char *path, char *out) {
    char sci[1024], com[1024], esc[3][2100];
    const char *us = strchr(label, '_');
    if (us) { size_t k = (size_t)(us - label); memcpy(sci, label, k); sci[k] = 0; strcpy(com, us + 1); }
    else { strcpy(sci, label); strcpy(com, label); }
    csv_escape(sci, esc[0]); csv_escape(com, esc[1]); csv_escape(path, esc[2]);
    return (size_t)sprintf(out, "%.1f,%.1f,%s,%s,%.4f,%s\n", (double)start, (double)end, esc[0], esc[1], (double)conf, esc[2]);
}

/* header -- reference src/output/csv.rs:41-52; BOM constants.rs:437 */
BO_API size_t bo_csv_header(int bom, char *out) {
    size_t o = 0;
    if (bom) { out[o++] = (char)0xEF; out[o++] = (char)0xBB; out[o++] = (char)0xBF; }
    o += (size_t)sprintf(out + o, "Start (s),End (s),Scientific name,Common name,Confidence,File\n");
    return o;
}

/* ------------------------------------------------------------------------------------ */
/* process_file restated end to end for a mono/stereo in-memory PCM stream                */
/* (reference src/pipeline/processor.rs:418-796 + :114-190 + :220-410)                     */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    size_t n_segments, n_detections;
    size_t effective_batch, n_batches, n_padded_rows;
} bo_process_stats;

/*
 * samples: mono f32 at source_rate (already through append_samples).
 * labels: n_classes C strings.  csv_out: caller buffer (cap bytes).  Returns CSV length or
 * (size_t)-1 on error.  logits_out (optional) receives [n_segments][n_classes].
 */
BO_API size_t bo_process_stream(const bo_model *m, const char *const *labels, const float *samples,
                                size_t n_samples, uint32_t source_rate, float overlap, float min_conf,
                                int top_k, size_t batch_size, int bom, const char *file_path,
                                char *csv_out, size_t cap, float *logits_out, size_t logits_cap_rows,
                                bo_process_stats *stats) {
    const uint32_t target_rate = m->h.sample_rate;
    const float seg_dur = m->h.segment_duration;
    const size_t seg = bo_duration_to_samples(seg_dur, target_rate);      /* :514 */
    const size_t ovl = bo_duration_to_samples(overlap, target_rate);      /* :520 */
    const double duration = (double)n_samples / (double)source_rate;       /* decode.rs:102-105 n_frames/rate */
    const int64_t est = bo_estimate_segment_count(1, duration, seg_dur, overlap);
    const size_t eff = bo_effective_batch_size(batch_size, est);
    const size_t src_seg = bo_source_samples(seg, source_rate, target_rate);
    const size_t src_ovl = bo_source_samples(ovl, source_rate, target_rate);
    bo_segmenter *sg = bo_segmenter_new(samples, n_samples, 1152);
    float *raw = malloc(sizeof(float) * src_seg);
    size_t rs_cap = bo_resample_max_len(src_seg, source_rate, target_rate);
    if (rs_cap < seg) rs_cap = seg;
    float *rs = malloc(sizeof(float) * rs_cap);
    size_t cap_seg = 16, nseg = 0;
    float *segs = malloc(sizeof(float) * cap_seg * seg);
    float *st = malloc(sizeof(float) * cap_seg), *en = malloc(sizeof(float) * cap_seg);
    size_t start_sample; int rc;
    while ((rc = bo_segmenter_next(sg, src_seg, src_ovl, raw, &start_sample)) == 1) {
        size_t got = bo_resample(raw, src_seg, source_rate, target_rate, rs);
        if (nseg == cap_seg) {
            cap_seg *= 2;
            segs = realloc(segs, sizeof(float) * cap_seg * seg);
            st = realloc(st, sizeof(float) * cap_seg); en = realloc(en, sizeof(float) * cap_seg);
        }
        float *dst = segs + nseg * seg;                                   /* resize(seg, 0.0) :87 */
        size_t cp = got < seg ? got : seg;
        memcpy(dst, rs, cp * sizeof(float));
        for (size_t i = cp; i < seg; i++) dst[i] = 0.0f;
        bo_chunk_times(start_sample, source_rate, seg, target_rate, &st[nseg], &en[nseg]);
        nseg++;
    }
    bo_segmenter_free(sg); free(raw); free(rs);
    if (rc < 0) { free(segs); free(st); free(en); return (size_t)-1; }
    /* batches: full batches of `eff`, tail padded with zero rows up to `eff` (:240-258) */
    const uint32_t nc = m->h.n_classes;
    size_t ndet = 0, capdet = 64, n_batches = 0, n_padded = 0;
    bo_detection *det = malloc(sizeof(bo_detection) * capdet);
    float *batch = malloc(sizeof(float) * eff * seg);
    float *logits = malloc(sizeof(float) * eff * nc);
    int *idx = malloc(sizeof(int) * (size_t)top_k); float *conf = malloc(sizeof(float) * (size_t)top_k);
    for (size_t b0 = 0; b0 < nseg; b0 += eff) {
        size_t valid = nseg - b0 < eff ? nseg - b0 : eff;
        memcpy(batch, segs + b0 * seg, valid * seg * sizeof(float));
        memset(batch + valid * seg, 0, (eff - valid) * seg * sizeof(float));
        n_padded += eff - valid; n_batches++;
        bo_forward(m, batch, (int)eff, logits, NULL, -1, NULL);
        for (size_t i = 0; i < valid; i++) {                               /* :363-385 */
            if (logits_out && b0 + i < logits_cap_rows)
                memcpy(logits_out + (b0 + i) * nc, logits + i * nc, nc * sizeof(float));
            int k = bo_topk(logits + i * nc, (int)nc, (int)m->h.output_activation, top_k, min_conf, idx, conf);
            for (int j = 0; j < k; j++) {
                if (!(conf[j] >= min_conf)) continue;                      /* :375 */
                if (ndet == capdet) { capdet *= 2; det = realloc(det, sizeof(bo_detection) * capdet); }
                det[ndet].start = st[b0 + i]; det[ndet].end = en[b0 + i];
                det[ndet].conf = conf[j]; det[ndet].label = idx[j]; ndet++;
            }
        }
    }
    bo_sort_detections(det, ndet);                                         /* :178-187 */
    size_t o = bo_csv_header(bom, csv_out);
    char row[8192];
    for (size_t i = 0; i < ndet; i++) {
        size_t k = bo_csv_row(labels[det[i].label], det[i].start, det[i].end, det[i].conf, file_path, row);
        if (o + k + 1 > cap) { o = (size_t)-1; break; }
        memcpy(csv_out + o, row, k); o += k;
    }
    if (o != (size_t)-1) csv_out[o] = 0;
    if (stats) {
        stats->n_segments = nseg; stats->n_detections = ndet; stats->effective_batch = eff;
        stats->n_batches = n_batches; stats->n_padded_rows = n_padded;
    }
    free(det); free(batch); free(logits); free(idx); free(conf); free(segs); free(st); free(en);
    return o;
}

/* ------------------------------------------------------------------------------------ */
/* Range filter / species list                                     */
/* ------------------------------------------------------------------------------------ */

/* scientific_name -- reference src/inference/geomodel.rs:28-33: the part before the first '_'
 * when that part contains a space, else the whole label.  Returns its length (a prefix of label). */
BO_API size_t bo_scientific_name_len(const char *label) {
    const char *us = strchr(label, '_');
    if (!us) return strlen(label);
    for (const char *p = label; p < us; p++)
        if (*p == ' ') return (size_t)(us - label);
    return strlen(label);
}

/* species_key -- geomodel.rs:36-38: scientific name, lower-cased (ASCII here: binomials are Latin). */
static char *bo_species_key(const char *label) {
    size_t n = bo_scientific_name_len(label);
    char *k = malloc(n + 1);
    for (size_t i = 0; i < n; i++) {
        unsigned char ch = (unsigned char)label[i];
        k[i] = (char)((ch >= 'A' && ch <= 'Z') ? ch + 32 : ch);
    }
    k[n] = 0;
    return k;
}

/* SpeciesMapping::build + GeomodelScores::project -- geomodel.rs:58-93 and :140-162, flattened onto
 * class indices: out[c] = NaN when classifier label c has no geomodel entry (score_of -> None), else the
 * geomodel's score for that species, 0 when the score list omits it.  Of two classifier labels with one
 * scientific name the first is mapped (:62-74), the second stays unmatched.  geo_scores / score_species
 * list the reported LocationScores (species label, score) in order; later entries overwrite earlier ones
 * (HashMap::insert, :154-157).  Returns mapped_count (:103-105).  O(n^2): an oracle. */
BO_API size_t bo_project_scores(const char *const *geo_labels, size_t n_geo, const char *const *score_species,
                                const float *score_values, size_t n_scores, const char *const *cls_labels,
                                size_t n_cls, float *out) {
    char **ck = malloc(sizeof(char *) * (n_cls ? n_cls : 1));
    int *first = malloc(sizeof(int) * (n_cls ? n_cls : 1));
    for (size_t c = 0; c < n_cls; c++) {
        ck[c] = bo_species_key(cls_labels[c]);
        first[c] = 1;
        for (size_t d = 0; d < c; d++)
            if (!strcmp(ck[d], ck[c])) { first[c] = 0; break; }
        out[c] = NAN;
    }
    size_t mapped = 0;
    for (size_t g = 0; g < n_geo; g++) {
        char *gk = bo_species_key(geo_labels[g]);
        for (size_t c = 0; c < n_cls; c++)
            if (first[c] && !strcmp(ck[c], gk)) {
                if (isnan(out[c])) { out[c] = 0.0f; mapped++; }
                break;
            }
        free(gk);
    }
    for (size_t s = 0; s < n_scores; s++) {
        char *sk = bo_species_key(score_species[s]);
        for (size_t c = 0; c < n_cls; c++)
            if (first[c] && !strcmp(ck[c], sk)) {
                if (!isnan(out[c])) out[c] = score_values[s];   /* classifier_label_for: only mapped species */
                break;
            }
        free(sk);
    }
    for (size_t c = 0; c < n_cls; c++) free(ck[c]);
    free(ck); free(first);
    return mapped;
}

/* filter_predictions -- reference src/inference/geomodel_filter.rs:46-82 on one PredictionResult:
 *   score >= threshold: keep (confidence * score when rerank); score < threshold: drop;
 *   no entry (NaN): keep only under the keep policy without rerank (:33-35).
 * With rerank the survivors are re-sorted by confidence descending under f32 total order (:77-79; the
 * reference's sort is unstable: equal confidences keep their input order here).  Returns the count. */
BO_API int bo_filter_predictions(const int *idx, const float *conf, int n, const float *scores, float threshold,
                                 int keep_unmatched, int rerank, int *out_idx, float *out_conf) {
    const int keeps = keep_unmatched && !rerank;
    int k = 0;
    for (int i = 0; i < n; i++) {
        const float s = scores[idx[i]];
        if (isnan(s)) {
            if (keeps) { out_idx[k] = idx[i]; out_conf[k] = conf[i]; k++; }
        } else if (s >= threshold) {
            out_idx[k] = idx[i]; out_conf[k] = rerank ? conf[i] * s : conf[i]; k++;
        }
    }
    if (rerank)
        for (int i = 1; i < k; i++) {   /* insertion sort, descending, stable */
            const int ti = out_idx[i]; const float tc = out_conf[i];
            int j = i;
            while (j > 0 && out_conf[j - 1] < tc) { out_idx[j] = out_idx[j - 1]; out_conf[j] = out_conf[j - 1]; j--; }
            out_idx[j] = ti; out_conf[j] = tc;
        }
    return k;
}

/* species-list filter -- reference src/inference/classifier.rs:617-640: retain predictions whose species is
 * in the list (keep[class] != 0); applies only when no range filter is configured (:587, :617). */
BO_API int bo_species_retain(const int *idx, const float *conf, int n, const unsigned char *keep, int *out_idx,
                             float *out_conf) {
    int k = 0;
    for (int i = 0; i < n; i++)
        if (keep[idx[i]]) { out_idx[k] = idx[i]; out_conf[k] = conf[i]; k++; }
    return k;
}
