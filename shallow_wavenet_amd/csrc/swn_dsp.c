/* Host-side signal processing of run.sh stages 3 / 6 / 9: the MLSA synthesis filter that noise_shaping.py applies
 * through pysptk (noise_shaping.py:51-86: ps.mc2b + pysptk.synthesis.MLSADF / Synthesizer).  pysptk / SPTK are not vendored by the
 * reference and absent here, so this restates the PUBLISHED algorithm (Imai, Sumita, Furuichi 1983; SPTK's mlsadf: a Pade
 * approximation of exp() around a cascade of first-order all-pass sections), pinned by its definition: the filter must realise
 *     H(z) = exp( sum_m b(m) Phi_m(z) ),  Phi_0 = 1,  Phi_m = (1 - a^2) z^-1 / (1 - a z^-1) * ((z^-1 - a) / (1 - a z^-1))^(m-1)
 * (tests/test_dsp.py checks the impulse response against that transfer function, and shaping followed by inverse shaping against
 * the identity).  Parity with pysptk itself is unpinned.  CPU code, not on the MI355X path; built as libswn_dsp.so by the Makefile. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* mel-cepstrum -> MLSA filter coefficients: b(M) = c(M), b(m) = c(m) - a b(m+1) */
void swn_dsp_mc2b(const double* mc, int m, double alpha, double* b) {
    b[m] = mc[m];
    for (int i = m - 1; i >= 0; --i) b[i] = mc[i] - alpha * b[i + 1];
}
/* and back: c(M) = b(M), c(m) = b(m) + a b(m+1) */
void swn_dsp_b2mc(const double* b, int m, double alpha, double* mc) {
    double prev = b[m];
    mc[m] = prev;
    for (int i = m - 1; i >= 0; --i) { const double cur = b[i]; mc[i] = cur + alpha * prev; prev = cur; }
}

/* Pade coefficients of exp(w) ~ P(w) / P(-w), orders 4 and 5 (modified for the MLSA filter's stability range) */
static const double k_pade4[5] = {1.0, 4.999273e-1, 1.067005e-1, 1.170221e-2, 5.656279e-4};
static const double k_pade5[6] = {1.0, 4.999391e-1, 1.107098e-1, 1.369984e-2, 9.564853e-4, 3.041721e-5};

/* the basic filter F(z) - b(1) Phi_1(z): all-pass chain, state d[0 .. m+1] */
static double basic_fir(double x, const double* b, int m, double a, double* d) {
    double y = 0.0;
    d[0] = x;
    d[1] = (1.0 - a * a) * d[0] + a * d[1];
    for (int i = 2; i <= m; ++i) {
        d[i] += a * (d[i + 1] - d[i - 1]);
        y += d[i] * b[i];
    }
    for (int i = m + 1; i > 1; --i) d[i] = d[i - 1];
    return y;
}
/* first stage: exp(b(1) Phi_1) ; state d[0 .. 2 pd + 1] */
static double stage1(double x, const double* b, double a, int pd, const double* pp, double* d) {
    double out = 0.0, *pt = d + pd + 1;
    for (int i = pd; i >= 1; --i) {
        d[i] = (1.0 - a * a) * pt[i - 1] + a * d[i];
        pt[i] = d[i] * b[1];
        const double v = pt[i] * pp[i];
        x += (i & 1) ? v : -v;
        out += v;
    }
    pt[0] = x;
    return out + x;
}
/* second stage: exp(sum_{m >= 2} b(m) Phi_m) ; state d[0 .. pd (m + 2) + pd] */
static double stage2(double x, const double* b, int m, double a, int pd, const double* pp, double* d) {
    double out = 0.0, *pt = d + pd * (m + 2);
    for (int i = pd; i >= 1; --i) {
        pt[i] = basic_fir(pt[i - 1], b, m, a, d + (i - 1) * (m + 2));
        const double v = pt[i] * pp[i];
        x += (i & 1) ? v : -v;
        out += v;
    }
    pt[0] = x;
    return out + x;
}

/* y[n] = MLSA filter of x with coefficient frames b[n_frames][order + 1], linearly interpolated inside every hop of `hop`
 * samples (frame i holds at sample i * hop; the last frame is held), gain exp(b(0)) applied to the input sample.
 * returns 0, -1 on a bad argument, -2 when out of memory. */
int swn_dsp_mlsa_synthesis(const double* x, long n, const double* b, int n_frames, int order, double alpha, int pd, int hop,
                           double* y) {
    if (!x || !b || !y || n < 0 || n_frames < 1 || order < 1 || hop < 1 || (pd != 4 && pd != 5) || fabs(alpha) >= 1.0) return -1;
    const double* pp = pd == 4 ? k_pade4 : k_pade5;
    const size_t n1 = 2 * (size_t)(pd + 1), n2 = (size_t)pd * (order + 2) + pd + 1;
    double* d = (double*)calloc(n1 + n2, sizeof(double));
    double* cur = (double*)malloc(2 * (size_t)(order + 1) * sizeof(double));
    if (!d || !cur) { free(d); free(cur); return -2; }
    double* slope = cur + order + 1;
    for (long s = 0; s < n; ++s) {
        const long f = s / hop, r = s - f * hop;
        if (r == 0) {
            const long f0 = f < n_frames ? f : n_frames - 1, f1 = f + 1 < n_frames ? f + 1 : n_frames - 1;
            for (int k = 0; k <= order; ++k) {
                cur[k] = b[(size_t)f0 * (order + 1) + k];
                slope[k] = (b[(size_t)f1 * (order + 1) + k] - cur[k]) / hop;
            }
        }
        double v = x[s] * exp(cur[0]);
        v = stage1(v, cur, alpha, pd, pp, d);
        v = stage2(v, cur, order, alpha, pd, pp, d + n1);
        y[s] = v;
        for (int k = 0; k <= order; ++k) cur[k] += slope[k];
    }
    free(d); free(cur);
    return 0;
}
