/* ft8_oracle.c -- CPU restatement of PyFT8's receive hot path (receiver.py + decoders.py).
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product (pyft8_amd/) never does.  It is a from-scratch,
 * single-threaded, plain-C statement of WHAT the reference computes, pinned against golden
 * vectors captured from the real reference (tests/golden/, oracle/gen_golden.py):
 *   - integer / bit work (CRC-14, GF(2) elimination, message validity, unpack, hashes,
 *     candidate ordering, the ipass ladder) is exact;
 *   - floating-point stages follow the reference's fp32 data flow; numpy primitives that
 *     cannot be bit-matched (pocketfft, BLAS sdot, SIMD tanh/log10) are replaced by an
 *     explicitly ordered IEEE-754 "arithmetic contract" that the HIP kernels repeat
 *     operation for operation, so GPU-vs-oracle comparisons can be bit-exact while
 *     oracle-vs-reference agrees to ~1e-6 relative (tests assert <= 1e-4).
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (no FMA contraction, no reassociation).
 *
 * Reference line citations are to /root/reference/PyFT8/.
 */
#include "ft8_oracle.h"
#include "ft8_tables.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

typedef struct { float re, im; } cpx;

static const int COSTAS[7] = {3, 1, 4, 0, 6, 5, 2};             /* receiver.py:13 */

void ft8o_default_config(ft8o_config* c) {
    memset(c, 0, sizeof(*c));
    c->sync_score_min = 85.0f; c->max_cands = 200;               /* receiver.py:311 */
    c->f0_lo = 32; c->f0_hi = 960;                               /* receiver.py:234-235 */
    c->h0_lo = -37; c->h0_hi = 87;                               /* receiver.py:319 */
    c->bp_nc0_a = 35; c->bp_iters_a = 5;                         /* receiver.py:78,91 */
    c->bp_nc0_b = 90; c->bp_iters_b = 20;                        /* receiver.py:95 */
    c->osd_single = 30; c->osd_double = 2;                       /* decoders.py:223 */
    c->llr_sd_min = 5.0f;                                        /* receiver.py:30 */
    int p1920[] = {8, 4, 4, 5, 3, 0}, p3200[] = {8, 4, 4, 5, 5, 0}, p300[] = {5, 5, 4, 3, 0}, p320[] = {8, 8, 5, 0};
    memcpy(c->plan1920, p1920, sizeof(p1920)); memcpy(c->plan3200, p3200, sizeof(p3200));
    memcpy(c->plan300, p300, sizeof(p300));    memcpy(c->plan320, p320, sizeof(p320));
}

/* ------------------------------------------------------------------ arithmetic contract */
static inline float f_from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t bits_from_f(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* log10 for x > 0: exponent split + atanh series, ~1e-7 absolute on the dB-scale inputs.
 * Replaces np.log10 (receiver.py:170,292). */
float ft8o_log10f(float x) {
    if (!(x > 0.0f)) return (x == 0.0f) ? -INFINITY : NAN;
    if (x > 3.0e38f) return INFINITY;
    uint32_t ix = bits_from_f(x);
    int e = 0;
    if (ix < 0x00800000u) { x = x * 8388608.0f; ix = bits_from_f(x); e = -23; }
    e += (int)(ix >> 23) - 127;
    float m = f_from_bits((ix & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float s = (m - 1.0f) / (m + 1.0f);
    float s2 = s * s;
    /* contract (round 3): the Horner steps and the final combination are fused multiply-adds (C99 fmaf: one rounding each) */
    float p = 0.11111111f;
    p = fmaf(p, s2, 0.14285715f);
    p = fmaf(p, s2, 0.2f);
    p = fmaf(p, s2, 0.33333334f);
    p = fmaf(p, s2, 1.0f);
    float lnm = (2.0f * s) * p;
    float fe = (float)e;
    return fmaf(fe, 0.301025390625f, fmaf(fe, 4.6050390e-6f, lnm * 0.4342945f));
}

/* tanh: single-branch clamped rational x P(x^2) / Q(x^2) (odd degree 13 over even degree 6, the classic fast-tanh
 * coefficient set) evaluated as x (P / Q), <= 3.5e-7 relative error on the whole line, exactly +-1 beyond |x| ~ 7.9.
 * Replaces np.tanh (decoders.py:142).  Plain mul/add in the written order, one IEEE division. */
float ft8o_tanhf(float x) {
    if (x != x) return x;
    float xc = x;
    if (xc > 7.90531111f) xc = 7.90531111f;
    if (xc < -7.90531111f) xc = -7.90531111f;
    const float x2 = xc * xc;
    /* contract (round 3): Horner steps as fused multiply-adds (C99 fmaf) */
    float p = -2.76076847742355e-16f;
    p = fmaf(p, x2, 2.00018790482477e-13f);
    p = fmaf(p, x2, -8.60467152213735e-11f);
    p = fmaf(p, x2, 5.12229709037114e-08f);
    p = fmaf(p, x2, 1.48572235717979e-05f);
    p = fmaf(p, x2, 6.37261928875436e-04f);
    p = fmaf(p, x2, 4.89352455891786e-03f);
    float q = 1.19825839466702e-06f;
    q = fmaf(q, x2, 1.18534705686654e-04f);
    q = fmaf(q, x2, 2.26843463243900e-03f);
    q = fmaf(q, x2, 4.89352518554385e-03f);
    /* contract (round 5): the quotient first, then the multiplication by x -- the division's operands are then in
     * [4.9e-3, 0.91] for every x, which lets the kernel drop the range repairs of its division sequence */
    return xc * (p / q);
}

static inline cpx cmul(cpx a, cpx w) { cpx r; r.re = fmaf(a.re, w.re, -(a.im * w.im)); r.im = fmaf(a.re, w.im, a.im * w.re); return r; }
static inline cpx cadd(cpx a, cpx b) { cpx r = {a.re + b.re, a.im + b.im}; return r; }
static inline cpx csub(cpx a, cpx b) { cpx r = {a.re - b.re, a.im - b.im}; return r; }
static inline cpx mulnegi(cpx a) { cpx r = {a.im, -a.re}; return r; }       /* a * (-i) */

/* forward DFT primitives (sign -1) -- canonical operation order */
static inline void dft2(cpx* a) { cpx t = a[0]; a[0] = cadd(t, a[1]); a[1] = csub(t, a[1]); }
static inline void dft3(cpx* a) {
    cpx t1 = cadd(a[1], a[2]), t2 = csub(a[1], a[2]);
    cpx m = {fmaf(-0.5f, t1.re, a[0].re), fmaf(-0.5f, t1.im, a[0].im)};           /* contract (round 3): named fmas */
    cpx n = {0.86602540f * t2.re, 0.86602540f * t2.im};
    a[0] = cadd(a[0], t1);
    a[1].re = m.re + n.im; a[1].im = m.im - n.re;
    a[2].re = m.re - n.im; a[2].im = m.im + n.re;
}
static inline void dft4(cpx* a) {
    cpx t0 = cadd(a[0], a[2]), t1 = csub(a[0], a[2]), t2 = cadd(a[1], a[3]), t3 = csub(a[1], a[3]);
    a[0] = cadd(t0, t2); a[2] = csub(t0, t2);
    a[1].re = t1.re + t3.im; a[1].im = t1.im - t3.re;
    a[3].re = t1.re - t3.im; a[3].im = t1.im + t3.re;
}
static inline void dft5(cpx* a) {
    const float c1 = 0.30901699f, c2 = -0.80901699f, s1 = 0.95105652f, s2 = 0.58778525f;
    cpx t1 = cadd(a[1], a[4]), t2 = cadd(a[2], a[3]), t3 = csub(a[1], a[4]), t4 = csub(a[2], a[3]);
    /* contract (round 3): the constant multiplies of the radix-5 butterfly as named fmas (12 of its 48 operations go away) */
    cpx m1 = {fmaf(c2, t2.re, fmaf(c1, t1.re, a[0].re)), fmaf(c2, t2.im, fmaf(c1, t1.im, a[0].im))};
    cpx m2 = {fmaf(c1, t2.re, fmaf(c2, t1.re, a[0].re)), fmaf(c1, t2.im, fmaf(c2, t1.im, a[0].im))};
    cpx n1 = {fmaf(s1, t3.re, s2 * t4.re), fmaf(s1, t3.im, s2 * t4.im)};
    cpx n2 = {fmaf(s2, t3.re, -(s1 * t4.re)), fmaf(s2, t3.im, -(s1 * t4.im))};
    cpx t5 = cadd(t1, t2);
    a[0] = cadd(a[0], t5);
    a[1].re = m1.re + n1.im; a[1].im = m1.im - n1.re;
    a[4].re = m1.re - n1.im; a[4].im = m1.im + n1.re;
    a[2].re = m2.re + n2.im; a[2].im = m2.im - n2.re;
    a[3].re = m2.re - n2.im; a[3].im = m2.im + n2.re;
}
static inline void dft8(cpx* a) {
    const float h = 0.70710678f;
    cpx e[4] = {a[0], a[2], a[4], a[6]}, o[4] = {a[1], a[3], a[5], a[7]};
    dft4(e); dft4(o);
    cpx o1 = {h * (o[1].re + o[1].im), h * (o[1].im - o[1].re)};
    cpx o2 = mulnegi(o[2]);
    cpx o3 = {h * (o[3].im - o[3].re), -(h * (o[3].re + o[3].im))};
    a[0] = cadd(e[0], o[0]); a[4] = csub(e[0], o[0]);
    a[1] = cadd(e[1], o1);   a[5] = csub(e[1], o1);
    a[2] = cadd(e[2], o2);   a[6] = csub(e[2], o2);
    a[3] = cadd(e[3], o3);   a[7] = csub(e[3], o3);
}

/* twiddle table W_N^t = (cos 2pi t/N, -sin 2pi t/N), double -> float */
static void make_twiddle(int n, cpx* w) {
    for (int t = 0; t < n; t++) {
        double ang = (2.0 * M_PI * (double)t) / (double)n;
        w[t].re = (float)cos(ang); w[t].im = (float)(-sin(ang));
    }
}

typedef struct { int n; cpx* w; } twtab;
static twtab g_tw[8];
static const cpx* get_twiddle(int n) {
    for (int i = 0; i < 8; i++) if (g_tw[i].n == n) return g_tw[i].w;
    for (int i = 0; i < 8; i++) if (g_tw[i].n == 0) {
        g_tw[i].w = (cpx*)malloc(sizeof(cpx) * (size_t)n); make_twiddle(n, g_tw[i].w); g_tw[i].n = n; return g_tw[i].w;
    }
    return NULL;
}

/* Stockham autosort, decimation in frequency, radix list `plan` (product == n).
 * pass(r): m = n_cur/r; for p<m, q<s: a_j = x[q + s(p + j m)]; b = DFT_r(a);
 *          y[q + s(r p + j)] = b_j * W_N^{j p s}   (multiply skipped when j p == 0). */
static void fft_core(cpx* x, cpx* y, int N, const int32_t* plan, const cpx* W) {
    int n = N, s = 1;
    cpx* src = x; cpx* dst = y;
    for (int ip = 0; plan[ip]; ip++) {
        int r = plan[ip], m = n / r;
        for (int p = 0; p < m; p++) for (int q = 0; q < s; q++) {
            cpx a[8];
            for (int j = 0; j < r; j++) a[j] = src[q + s * (p + j * m)];
            switch (r) { case 2: dft2(a); break; case 3: dft3(a); break; case 4: dft4(a); break;
                         case 5: dft5(a); break; case 8: dft8(a); break; default: abort(); }
            for (int j = 0; j < r; j++) {
                cpx v = a[j];
                if (j * p != 0) v = cmul(v, W[(size_t)j * p * s]);
                dst[q + s * (r * p + j)] = v;
            }
        }
        cpx* t = src; src = dst; dst = t;
        n = m; s *= r;
    }
    if (src != x) memcpy(x, src, sizeof(cpx) * (size_t)N);
}

void ft8o_fft(float* data, int n, const int32_t* plan, float* scratch) {
    fft_core((cpx*)data, (cpx*)scratch, n, plan, get_twiddle(n));
}

/* ------------------------------------------------------------------ spectrogram (receiver.py:288-306) */
static float g_win[3840];
static int g_win_ok = 0;
static void make_window(void) {
    /* np.hanning(3840).astype(float32): 0.5 + 0.5 cos(pi (2i+1-M)/(M-1)) (receiver.py:236) */
    for (int i = 0; i < 3840; i++)
        g_win[i] = (float)(0.5 + 0.5 * cos(M_PI * (double)(2 * i + 1 - 3840) / 3839.0));
    g_win_ok = 1;
}

void ft8o_spectrogram(const int16_t* audio, const ft8o_config* c, float* grid) {
    if (!g_win_ok) make_window();
    const cpx* W = get_twiddle(1920);
    const cpx* WR = get_twiddle(3840);
    cpx* z = (cpx*)malloc(sizeof(cpx) * 1920 * 2);
    cpx* y = z + 1920;
    for (int k = 0; k < FT8O_GRID_COLS; k++) grid[k] = 1.0f;       /* row 0 never written (receiver.py:240,300) */
    for (int r = 1; r <= 375; r++) {
        int base = 480 * r - 3840;                                   /* window = last 3840 samples after hop r */
        for (int m = 0; m < 1920; m++) {
            int i0 = base + 2 * m, i1 = i0 + 1;
            float x0 = (i0 >= 0) ? (float)audio[i0] * g_win[2 * m] : 0.0f * g_win[2 * m];
            float x1 = (i1 >= 0) ? (float)audio[i1] * g_win[2 * m + 1] : 0.0f * g_win[2 * m + 1];
            z[m].re = x0; z[m].im = x1;
        }
        fft_core(z, y, 1920, c->plan1920, W);
        float* out = grid + (size_t)r * FT8O_GRID_COLS;
        for (int k = 0; k < FT8O_GRID_COLS; k++) {
            cpx a = z[k], b = z[(1920 - k) % 1920];
            float er = 0.5f * (a.re + b.re), ei = 0.5f * (a.im - b.im);
            float orr = 0.5f * (a.im + b.im), oi = 0.5f * (b.re - a.re);
            cpx w = WR[k];
            float xr = er + (w.re * orr - w.im * oi);
            float xi = ei + (w.re * oi + w.im * orr);
            /* 20 log10(|X| + 1e-12) (receiver.py:292) as 10 log10(max(|X|^2, 1e-24)): the same value to well below a float ulp wherever
             * |X| > 1e-6 (any frame that is not digital silence), exactly -240 dB for |X| = 0 as in the reference, and no square
             * root in the kernel's epilogue */
            float pw = xr * xr + xi * xi;
            if (!(pw > 1e-24f)) pw = 1e-24f;
            out[k] = 10.0f * ft8o_log10f(pw);
        }
    }
    free(z);
}

/* grid row accessor with the reference's modulo-750 wrap (receiver.py:240,347,360): rows that were
 * never written in an isolated frame (0, 376..749) hold 1.0 */
static inline float grid_at(const float* grid, int row, int col) {
    row %= 750; if (row < 0) row += 750;
    if (row >= 1 && row <= 375) return grid[(size_t)row * FT8O_GRID_COLS + col];
    return 1.0f;
}

/* ------------------------------------------------------------------ sync search (receiver.py:338-367) */
static const double W6 = (double)(-0.16666667163372040f);          /* np.float32(-1/6), receiver.py:323 */

/* Local re-search of the subtraction experiment (tests/pipeline/receiver_sub.py:434-445: search(range(f0 - 2, f0 + 2),
 * ignore_sync_score_min = True) around every subtracted signal): while a mask is set, only the columns with a non-zero byte are
 * searched and every score above 0 makes a candidate.  mask[f0 - f0_lo]; NULL = the configured search.  Process-global (test
 * infrastructure: one frame at a time per process). */
static uint8_t g_search_mask[4096]; static int g_search_mask_n = 0;
void ft8o_set_search_mask(const uint8_t* mask, int32_t n) {
    if (!mask || n <= 0) { g_search_mask_n = 0; return; }
    if (n > (int32_t)sizeof(g_search_mask)) n = (int32_t)sizeof(g_search_mask);
    memcpy(g_search_mask, mask, (size_t)n); g_search_mask_n = n;
}

int ft8o_sync_search(const float* grid, const ft8o_config* c, ft8o_cand* out) {
    int n = 0;
    for (int f0 = c->f0_lo; f0 < c->f0_hi; f0++) {
        if (g_search_mask_n && !(f0 - c->f0_lo < g_search_mask_n && g_search_mask[f0 - c->f0_lo])) continue;
        float best = 0.0f; int best_h0 = 0;
        for (int h0 = c->h0_lo; h0 < c->h0_hi; h0++) {
            double s1 = 0.0, tsum = 0.0;
            for (int s = 0; s < 7; s++) {
                int row = h0 + 148 + 4 * s;                         /* h0 + base_search_hops + search_hps */
                double t = 0.0;
                for (int b = 0; b < 14; b++) t += (double)grid_at(grid, row, f0 + b);
                tsum += t;
                s1 += (double)grid_at(grid, row, f0 + 2 * COSTAS[s]) + (double)grid_at(grid, row, f0 + 2 * COSTAS[s] + 1);
            }
            float score = (float)(s1 + W6 * (tsum - s1));
            if (score > best) { best = score; best_h0 = h0; }       /* first strict maximum */
        }
        if (best > (g_search_mask_n ? 0.0f : c->sync_score_min)) {
            memset(&out[n], 0, sizeof(out[n]));
            out[n].f0_idx = f0; out[n].h0_idx = best_h0; out[n].score = best; out[n].ipass = -1;
            n++;
        }
    }
    /* stable sort, score descending (receiver.py:366): insertion sort keeps f0 order on ties */
    for (int i = 1; i < n; i++) {
        ft8o_cand t = out[i]; int j = i - 1;
        while (j >= 0 && out[j].score < t.score) { out[j + 1] = out[j]; j--; }
        out[j + 1] = t;
    }
    return n < c->max_cands ? n : c->max_cands;
}

static const int PAYLOAD_SYM[58] = {7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,
    43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63,64,65,66,67,68,69,70,71};   /* receiver.py:14 */

void ft8o_payload(const float* grid, int f0_idx, int h0_idx, float* p) {
    for (int i = 0; i < 58; i++)
        for (int t = 0; t < 8; t++)
            p[i * 8 + t] = grid_at(grid, h0_idx + 4 + 4 * PAYLOAD_SYM[i], f0_idx + 1 + 2 * t);   /* receiver.py:358-362 */
}

/* numpy's float32 pairwise summation (np.mean on a contiguous f32 vector) */
static float pairwise_sum(const float* a, int n) {
    if (n < 8) { float r = 0.0f; for (int i = 0; i < n; i++) r += a[i]; return r; }
    if (n <= 128) {
        float r[8]; int i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; j++) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    int n2 = n / 2; n2 -= n2 % 8;
    return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}

static inline float max4(float a, float b, float c, float d) {
    float m = a; if (b > m) m = b; if (c > m) m = c; if (d > m) m = d; return m;
}

/* receiver.py:208-222.  Returns 1 if the sd gate passes (sd > 5), 0 => 'stop'. */
int ft8o_db_to_llr(const float* p, float* llr, float* sd_out, int32_t* snr_out) {
    float pmax = p[0], pmin = p[0];
    for (int i = 1; i < 464; i++) { if (p[i] > pmax) pmax = p[i]; if (p[i] < pmin) pmin = p[i]; }
    float d = (pmax - pmin) - 58.0f;
    int snr = (int)d; if (snr < -24) snr = -24; if (snr > 24) snr = 24;
    float sq[174];
    for (int i = 0; i < 58; i++) {
        const float* q = p + 8 * i;
        llr[3 * i + 0] = max4(q[4], q[5], q[6], q[7]) - max4(q[0], q[1], q[2], q[3]);
        llr[3 * i + 1] = max4(q[2], q[3], q[4], q[7]) - max4(q[0], q[1], q[5], q[6]);
        llr[3 * i + 2] = max4(q[1], q[2], q[6], q[7]) - max4(q[0], q[3], q[4], q[5]);
    }
    for (int i = 0; i < 174; i++) sq[i] = llr[i] * llr[i];
    float mean = pairwise_sum(llr, 174) / 174.0f;
    float var = pairwise_sum(sq, 174) / 174.0f - mean * mean;
    float sd = sqrtf(var);
    for (int i = 0; i < 174; i++) llr[i] = (2.83f * llr[i]) / sd;
    *sd_out = sd; *snr_out = snr;
    return !(sd <= 5.0f);
}

/* ------------------------------------------------------------------ cycle spectrum (receiver.py:280-286) */
static void cycle_spectrum_f32(const float* audio, const ft8o_config* c, float* spec_out);
void ft8o_cycle_spectrum(const int16_t* audio, const ft8o_config* c, float* spec_out) {
    float* x = (float*)malloc(sizeof(float) * FT8O_NSAMP);
    for (int i = 0; i < FT8O_NSAMP; i++) x[i] = (float)audio[i];          /* exact */
    cycle_spectrum_f32(x, c, spec_out);
    free(x);
}
/* the same transform of a float32 buffer (the subtraction experiment's ring holds float32 residuals, receiver_sub.py:266,273-276) */
void ft8o_cycle_spectrum_f32(const float* audio, const ft8o_config* c, float* spec_out) { cycle_spectrum_f32(audio, c, spec_out); }
static void cycle_spectrum_f32(const float* audio, const ft8o_config* c, float* spec_out) {
    const int N = 96000, N1 = 300, N2 = 320;
    const cpx* W = get_twiddle(N);
    const cpx* W1 = get_twiddle(N1);
    const cpx* W2 = get_twiddle(N2);
    cpx* A = (cpx*)malloc(sizeof(cpx) * (size_t)N);       /* A[k1][n2] */
    cpx* Z = (cpx*)malloc(sizeof(cpx) * (size_t)N);
    cpx col[320], scr[320];
    for (int n2 = 0; n2 < N2; n2++) {
        for (int n1 = 0; n1 < N1; n1++) {
            int m = N2 * n1 + n2;
            col[n1].re = (2 * m < FT8O_NSAMP) ? audio[2 * m] : 0.0f;
            col[n1].im = (2 * m + 1 < FT8O_NSAMP) ? audio[2 * m + 1] : 0.0f;
        }
        fft_core(col, scr, N1, c->plan300, W1);
        for (int k1 = 0; k1 < N1; k1++) {
            cpx v = col[k1];
            if (n2 * k1 != 0) v = cmul(v, W[(size_t)n2 * k1]);
            A[(size_t)k1 * N2 + n2] = v;
        }
    }
    for (int k1 = 0; k1 < N1; k1++) {
        memcpy(col, A + (size_t)k1 * N2, sizeof(cpx) * N2);
        fft_core(col, scr, N2, c->plan320, W2);
        for (int k2 = 0; k2 < N2; k2++) Z[k1 + N1 * k2] = col[k2];
    }
    cpx* X = (cpx*)spec_out;
    for (int k = 0; k < FT8O_SPEC_BINS; k++) {
        cpx a = Z[k], b = Z[(N - k) % N];
        float er = 0.5f * (a.re + b.re), ei = 0.5f * (a.im - b.im);
        float orr = 0.5f * (a.im + b.im), oi = 0.5f * (b.re - a.re);
        double ang = (2.0 * M_PI * (double)k) / 192000.0;
        float wr = (float)cos(ang), wi = (float)(-sin(ang));
        X[k].re = er + (wr * orr - wi * oi);
        X[k].im = ei + (wr * oi + wi * orr);
    }
    free(A); free(Z);
}

/* ------------------------------------------------------------------ fine sync (receiver.py:140-206) */
static double g_taper[100];
static int g_taper_ok = 0;
static void make_taper(void) {
    /* 0.5*(1+cos(linspace(pi,0,100))) == 0.5*(1+cos(linspace(-pi,0,100))) (receiver.py:183-184): rises 0 -> 1 */
    double step = (0.0 - M_PI) / 99.0;
    for (int i = 0; i < 100; i++) {
        double y = (i == 99) ? 0.0 : (double)i * step + M_PI;
        g_taper[i] = 0.5 * (1.0 + cos(y));
    }
    g_taper_ok = 1;
}

/* baseband series z[3200] @ 200 S/s for spectrum origin fb (receiver.py:180-186) */
static void fine_zsig_t(const float* spec, const ft8o_config* c, int fb, cpx* z, int tapered);
static void fine_zsig(const float* spec, const ft8o_config* c, int fb, cpx* z) { fine_zsig_t(spec, c, fb, z, 1); }
/* tapered = 0: the slice as the subtraction experiment takes it (receiver_sub.py:188-190: no edge tapers) */
static void fine_zsig_t(const float* spec, const ft8o_config* c, int fb, cpx* z, int tapered) {
    if (!g_taper_ok) make_taper();
    const cpx* S = (const cpx*)spec;
    const cpx* W = get_twiddle(3200);
    cpx* scr = (cpx*)malloc(sizeof(cpx) * 3200);
    for (int k = 0; k < 3200; k++) { z[k].re = 0.0f; z[k].im = 0.0f; }
    for (int k = 0; k < 850; k++) {
        cpx v = S[fb + k];
        if (tapered && k >= 750) { double t = g_taper[k - 750]; v.re = (float)((double)v.re * t); v.im = (float)((double)v.im * t); }
        z[k].re = v.re; z[k].im = -v.im;                           /* conj: inverse FFT = conj(FFT(conj)) */
    }
    for (int k = 0; k < 150; k++) {
        cpx v = S[fb - 150 + k];
        if (tapered && k < 100) { double t = g_taper[k]; v.re = (float)((double)v.re * t); v.im = (float)((double)v.im * t); }
        z[3050 + k].re = v.re; z[3050 + k].im = -v.im;
    }
    fft_core(z, scr, 3200, c->plan3200, W);
    const float inv = 0.0003125f;
    for (int k = 0; k < 3200; k++) { z[k].re = z[k].re * inv; z[k].im = -(z[k].im * inv); }
    free(scr);
}

/* |32-point DFT| tones 0..7 of symbol s starting at tb (receiver.py:189-195).
 * Contract: 32 = 4 x 8 decimation in time, outputs 0..7 only:
 *   u[n2][k] = DFT8_k( x[4 n1 + n2], n1 = 0..7 ),  X[k] = ((u0 + u1 W^k) + u2 W^2k) + u3 W^3k,  W = e^{-2 pi i/32} */
static void fine_symbol(const cpx* z, int tb, int s, float* g8) {
    static cpx W32[32]; static int ok = 0;
    if (!ok) { make_twiddle(32, W32); ok = 1; }
    int i0 = tb + 32 * s; if (i0 < 0) i0 = 0; if (i0 > 3168) i0 = 3168;
    cpx u[4][8];
    for (int n2 = 0; n2 < 4; n2++) {
        for (int n1 = 0; n1 < 8; n1++) u[n2][n1] = z[i0 + 4 * n1 + n2];
        dft8(u[n2]);
    }
    for (int k = 0; k < 8; k++) {
        cpx acc = u[0][k];
        for (int n2 = 1; n2 < 4; n2++) {
            cpx t = u[n2][k];
            if (k != 0) t = cmul(t, W32[(n2 * k) & 31]);
            acc = cadd(acc, t);
        }
        g8[k] = sqrtf(acc.re * acc.re + acc.im * acc.im);
    }
}

static float fine_score_block(const cpx* z, int tb, int sym0);
static float fine_score(const cpx* z, int tb) { return fine_score_block(z, tb, 36); }   /* receiver.py:197-206: middle Costas only */
static float fine_score_block(const cpx* z, int tb, int sym0) {
    /* contract: per symbol a, on = g[a][costas[a]] and off_a = sum over the other six tones b < 7 (b ascending, fp64);
     * S1 = sum_a on_a, S2 = sum_a off_a (a ascending); score = (float)(S1 + w6 * S2) */
    double s1 = 0.0, s2 = 0.0;
    for (int a = 0; a < 7; a++) {
        float g[8]; fine_symbol(z, tb, sym0 + a, g);
        double off = 0.0;
        for (int b = 0; b < 7; b++) if (b != COSTAS[a]) off += (double)g[b];
        s1 += (double)g[COSTAS[a]];
        s2 += off;
    }
    return (float)(s1 + W6 * s2);
}

/* Score of a NON-ZERO frequency tweak (receiver.py:147-161, 197-206) straight from the spectrum slice, without its time series.
 * The reference forms z = ifft(S) and scores |fft(z[i0 : i0 + 32])[t]| on the 7 symbols of the middle Costas block.  Substituting one
 * transform into the other (an exact identity, every one of the 1000 non-zero bins included):
 *     T[s][t] = 1/3200 sum_k X[k] e^{2 pi i k nb0 / 3200} D(k - 100 t) e^{2 pi i k s / 100},     D(m) = sum_{n < 32} e^{2 pi i n m / 3200},
 * k = -150 .. 849 the rolled slice's bins (X = tapered spectrum), nb0 = tb + 32 * 36 the block's first sample, s = 0 .. 6.
 * Two facts make it cheap.  (1) With m = r + 100 d, 0 <= r < 100, the Dirichlet kernel factors into a phase and a REAL number:
 *     D(m) = e^{i pi 31 r / 3200} g^d K(m),   g = e^{-i pi / 32},   K(m) = sin(pi r / 100) / sin(pi m / 3200)   (K(0) = 32, K(100 d) = 0),
 * so with k = r + 100 j the inner sum is a correlation of the phased bins with a real kernel,
 *     H[t][r] = sum_j b[k] K(k - 100 t),     b[k] = X[k] Phi[k],     Phi[k] = e^{2 pi i k nb0 / 3200} e^{i pi (31 r - 100 j) / 3200}
 * -- the factor g^{-t} left over has modulus 1 and only |T| is used, so it is dropped.  (2) e^{2 pi i k s / 100} has period 100 in k, and
 * the four residues p, 100 - p, 50 - p, 50 + p share one (cos, sin) = (cos, sin)(2 pi p s / 100) up to signs: with P_x = H[t][x] + H[t][100 - x],
 * M_x = H[t][x] - H[t][100 - x],
 *     T[s][t] = 1/3200 sum_{p = 0 .. 25} cos_ps (P_p + (-1)^s P_{50-p}) + i sin_ps (M_p - (-1)^s M_{50-p})
 * (p = 0 stands for the residues 0 and 50, p = 25 for the pair 25, 75 alone): four real multiply-adds per item, symbol and tone.
 * Contract (kernels/fine_sync.hpp: fine_fscore does exactly this, on 100 + 112 lanes):
 *   tables in double, rounded once: K[m] (m = -900 .. 899), (cos, sin)[s][p] (p = 0: (1, 0)), G[k] = e^{i pi (31 r - 100 j) / 3200};
 *   Phi[k] = cmul(conj W3200[(k nb0) mod 3200], G[k]), W3200 the FFT's twiddle table;
 *   b[q] = cmul(X[k_q], Phi[k_q]), q = 0 .. 9 ascending in k (the first and the last are the tapered bins: fp64 product, rounded once);
 *   H: per component  fma(b, K, acc)  in ascending q;
 *   T: 16 partial sums per (s, t) over the items p = c, c + 16: h1 = H[p], h2 = H[100 - p], h3 = H[50 - p], h4 = H[50 + p] (p = 0: h2 = h4 = 0;
 *      p = 25: h3 = h4 = 0); P = h1 + h2, M = h1 - h2, P' = h3 + h4, M' = h3 - h4; A = P + P', B = P - P', C = M - M', D = M + M';
 *      s = 0 adds A; even s:  re = fma(A.re, cos, fma(-C.im, sin, re)), im = fma(A.im, cos, fma(C.re, sin, im));  odd s: the same with B, D;
 *      the partials combined as the binary tree ((p0+p1)+(p2+p3)) + ((p4+p5)+(p6+p7)) ... ; |T| from the components scaled by 1/3200;
 *      fp64 on / off sums as in fine_score_block. */
static float g_K32[1800];                       /* K(m), m = -900 .. 899 */
static cpx g_TW100[100];                        /* e^{-2 pi i m / 100} */
static cpx g_CS100[6][26], g_G1000[1000];
static int g_fs_ok = 0;
static void make_fscore_tables(void) {
    for (int m = -900; m < 900; m++) {
        const int r = ((m % 100) + 100) % 100;
        g_K32[m + 900] = (r == 0) ? (m == 0 ? 32.0f : 0.0f) : (float)(sin(M_PI * (double)r / 100.0) / sin(M_PI * (double)m / 3200.0));
    }
    for (int m = 0; m < 100; m++) { const double a = -2.0 * M_PI * (double)m / 100.0; g_TW100[m].re = (float)cos(a); g_TW100[m].im = (float)sin(a); }
    for (int s = 1; s < 7; s++) for (int q = 0; q <= 25; q++) {
        const double a = 2.0 * M_PI * (double)((q * s) % 100) / 100.0;
        cpx v; v.re = (float)cos(a); v.im = (float)sin(a);
        if (q == 0) { v.re = 1.0f; v.im = 0.0f; }
        g_CS100[s - 1][q] = v;
    }
    for (int k = -150; k < 850; k++) {
        const int r = ((k % 100) + 100) % 100, j = (k - r) / 100;
        const double a = M_PI * (31.0 * (double)r - 100.0 * (double)j) / 3200.0;
        g_G1000[k + 150].re = (float)cos(a); g_G1000[k + 150].im = (float)sin(a);
    }
    g_fs_ok = 1;
}
static float fine_fscore(const float* spec, int fb, int nb0) {
    if (!g_taper_ok) make_taper();
    if (!g_fs_ok) make_fscore_tables();
    const cpx* S = (const cpx*)spec;
    const cpx* W = get_twiddle(3200);
    cpx Phi[1000];
    cpx H[7][100];
    for (int k = -150; k < 850; k++) {
        const cpx w = W[(((k * nb0) % 3200) + 3200) % 3200];
        cpx wc; wc.re = w.re; wc.im = -w.im;
        Phi[k + 150] = cmul(wc, g_G1000[k + 150]);
    }
    for (int r = 0; r < 100; r++) {
        const int jlo = r < 50 ? -1 : -2;
        cpx b[10];
        for (int q = 0; q < 10; q++) {
            const int k = r + 100 * (jlo + q);
            cpx x = S[fb + k];
            if (q == 0) { const double t = g_taper[k + 150]; x.re = (float)((double)x.re * t); x.im = (float)((double)x.im * t); }
            if (q == 9) { const double t = g_taper[k - 750]; x.re = (float)((double)x.re * t); x.im = (float)((double)x.im * t); }
            b[q] = cmul(x, Phi[k + 150]);
        }
        for (int t = 0; t < 7; t++) {
            float hx = 0.0f, hy = 0.0f;
            for (int q = 0; q < 10; q++) {
                const float kv = g_K32[r + 100 * (jlo + q - t) + 900];
                hx = fmaf(b[q].re, kv, hx);
                hy = fmaf(b[q].im, kv, hy);
            }
            H[t][r].re = hx; H[t][r].im = hy;
        }
    }
    float mag[7][7];
    for (int t = 0; t < 7; t++) {
        cpx part[7][16];
        for (int c = 0; c < 16; c++) {
            cpx acc[7];
            for (int s = 0; s < 7; s++) { acc[s].re = 0.0f; acc[s].im = 0.0f; }
            for (int i = 0; i < 2; i++) {
                const int q = c + 16 * i;
                if (q > 25) continue;
                const cpx zero = {0.0f, 0.0f};
                const cpx h1 = H[t][q];
                const cpx h2 = (q == 0) ? zero : H[t][100 - q];
                const cpx h3 = (q == 25) ? zero : H[t][50 - q];
                const cpx h4 = (q == 0 || q == 25) ? zero : H[t][50 + q];
                cpx P, M, P2, M2, A, B, C, D;
                P.re = h1.re + h2.re; P.im = h1.im + h2.im; M.re = h1.re - h2.re; M.im = h1.im - h2.im;
                P2.re = h3.re + h4.re; P2.im = h3.im + h4.im; M2.re = h3.re - h4.re; M2.im = h3.im - h4.im;
                A.re = P.re + P2.re; A.im = P.im + P2.im; B.re = P.re - P2.re; B.im = P.im - P2.im;
                C.re = M.re - M2.re; C.im = M.im - M2.im; D.re = M.re + M2.re; D.im = M.im + M2.im;
                acc[0].re = acc[0].re + A.re; acc[0].im = acc[0].im + A.im;
                for (int s = 1; s < 7; s++) {
                    const cpx e = g_CS100[s - 1][q];
                    const cpx X = (s & 1) ? B : A, Y = (s & 1) ? D : C;
                    acc[s].re = fmaf(X.re, e.re, fmaf(-Y.im, e.im, acc[s].re));
                    acc[s].im = fmaf(X.im, e.re, fmaf(Y.re, e.im, acc[s].im));
                }
            }
            for (int s = 0; s < 7; s++) part[s][c] = acc[s];
        }
        for (int s = 0; s < 7; s++) {
            float q4[4][2];
            for (int g = 0; g < 4; g++) {
                const cpx* p = &part[s][4 * g];
                q4[g][0] = (p[0].re + p[1].re) + (p[2].re + p[3].re);
                q4[g][1] = (p[0].im + p[1].im) + (p[2].im + p[3].im);
            }
            const float tr = (q4[0][0] + q4[1][0]) + (q4[2][0] + q4[3][0]), ti = (q4[0][1] + q4[1][1]) + (q4[2][1] + q4[3][1]);
            const float re = tr * 0.0003125f, im = ti * 0.0003125f;
            mag[s][t] = sqrtf(re * re + im * im);
        }
    }
    double s1 = 0.0, s2 = 0.0;
    for (int a = 0; a < 7; a++) {
        double off = 0.0;
        for (int b = 0; b < 7; b++) if (b != COSTAS[a]) off += (double)mag[a][b];
        s1 += (double)mag[a][COSTAS[a]];
        s2 += off;
    }
    return (float)(s1 + W6 * s2);
}

/* The final 79 x 8 grid of magnitudes, also straight from the spectrum slice (round 4, third step).  Same identity as fine_fscore with
 * nb0 = tb, the first sample of symbol 0, and eight tones:  H[t][r] = sum_j b[k] K(k - 100 t)  (b = X Phi, ascending k, fma(b, K, acc)),
 *     |T[s][t]| = 1/3200 | sum_{r < 100} H[t][r] e^{2 pi i r s / 100} |,   s = 0 .. 78,
 * the sum over r as the forward 100-point DFT of conj H (its magnitude is the same), 100 = 10 x 10:
 *     r = 10 r1 + r2, s = s1 + 10 s2:   A[r2][s1] = DFT10_{r1}(conj H[10 r1 + r2])[s1] * TW[r2 s1]   (TW[m] = e^{-2 pi i m / 100}; s1 = 0: no multiply)
 *                                       T[s1 + 10 s2] = DFT10_{r2}(A[r2][s1])[s2]
 * DFT10 (forward) by the prime-factor map: a0[n] = x[2n mod 10] + x[(5 + 2n) mod 10], a1[n] = their difference, n = 0 .. 4; dft5 of both;
 * y[6 k mod 10] = a0[k], y[(5 + 6 k) mod 10] = a1[k].  Magnitude: re = T.re * (1/3200), im = T.im * (1/3200), sqrtf(re re + im im).
 * The reference clamps a symbol's first sample to [0, 3168] (receiver.py:189-195: fine_symbol above): every symbol with tb + 32 s <= 0
 * reads the samples 0 .. 31, every one with tb + 32 s >= 3168 the samples 3168 .. 3199.  Those get the grid row of position p = 0 / 3168:
 * H with nb0 = p, |T[t]| = 1/3200 |sum_r H[t][r]| as 16 partial sums over r = c, c + 16, ... combined as the binary tree of fine_fscore. */
static void fscore_H(const cpx* S, int fb, int nb0, int ntone, cpx H[8][100]) {
    const cpx* W = get_twiddle(3200);
    for (int r = 0; r < 100; r++) {
        const int jlo = r < 50 ? -1 : -2;
        cpx b[10];
        for (int q = 0; q < 10; q++) {
            const int k = r + 100 * (jlo + q);
            cpx x = S[fb + k];
            if (q == 0) { const double t = g_taper[k + 150]; x.re = (float)((double)x.re * t); x.im = (float)((double)x.im * t); }
            if (q == 9) { const double t = g_taper[k - 750]; x.re = (float)((double)x.re * t); x.im = (float)((double)x.im * t); }
            const cpx w = W[(int)((((long long)k * nb0) % 3200 + 3200) % 3200)];
            cpx wc; wc.re = w.re; wc.im = -w.im;
            b[q] = cmul(x, cmul(wc, g_G1000[k + 150]));
        }
        for (int t = 0; t < ntone; t++) {
            float hx = 0.0f, hy = 0.0f;
            for (int q = 0; q < 10; q++) {
                const float kv = g_K32[r + 100 * (jlo + q - t) + 900];
                hx = fmaf(b[q].re, kv, hx);
                hy = fmaf(b[q].im, kv, hy);
            }
            H[t][r].re = hx; H[t][r].im = hy;
        }
    }
}
static void dft10_fwd(const cpx* x, cpx* y) {
    cpx a0[5], a1[5];
    for (int n = 0; n < 5; n++) { const cpx u = x[(2 * n) % 10], v = x[(5 + 2 * n) % 10]; a0[n] = cadd(u, v); a1[n] = csub(u, v); }
    dft5(a0); dft5(a1);
    for (int k = 0; k < 5; k++) { y[(6 * k) % 10] = a0[k]; y[(5 + 6 * k) % 10] = a1[k]; }
}
static void fine_grid_freq(const float* spec, int fb, int tb, float* g /*[79][8]*/) {
    if (!g_taper_ok) make_taper();
    if (!g_fs_ok) make_fscore_tables();
    const cpx* S = (const cpx*)spec;
    static __thread cpx H[8][100], A[8][10][10];
    fscore_H(S, fb, tb, 8, H);
    for (int t = 0; t < 8; t++) {
        for (int r2 = 0; r2 < 10; r2++) {
            cpx x[10], y[10];
            for (int r1 = 0; r1 < 10; r1++) { x[r1].re = H[t][10 * r1 + r2].re; x[r1].im = -H[t][10 * r1 + r2].im; }
            dft10_fwd(x, y);
            for (int s1 = 0; s1 < 10; s1++) A[t][s1][r2] = (s1 == 0) ? y[0] : cmul(y[s1], g_TW100[r2 * s1]);
        }
        for (int s1 = 0; s1 < 10; s1++) {
            cpx y[10];
            dft10_fwd(A[t][s1], y);
            for (int s2 = 0; s2 < 8; s2++) {
                const int s = s1 + 10 * s2;
                if (s > 78) continue;
                const float re = y[s2].re * 0.0003125f, im = y[s2].im * 0.0003125f;
                g[8 * s + t] = sqrtf(re * re + im * im);
            }
        }
    }
    /* clamped symbols */
    int s_lo = 0, s_hi = 78;
    while (s_lo <= 78 && tb + 32 * s_lo < 0) s_lo++;                 /* symbols 0 .. s_lo - 1 read position 0 */
    while (s_hi >= 0 && tb + 32 * s_hi > 3168) s_hi--;               /* symbols s_hi + 1 .. 78 read position 3168 */
    /* A symbol that starts exactly AT the clamp position reads the very samples the clamped ones read (clip() returns the same index):
     * in the reference its row is bit-identical to theirs, and osd_012's argsort sees exact |LLR| ties between them.  So whenever some
     * symbol lies strictly beyond, the boundary symbol takes the clamp row too; alone, it is an ordinary symbol (nothing to tie with). */
    if (s_lo > 0 && s_lo <= 78 && tb + 32 * s_lo == 0) s_lo++;
    if (s_hi < 78 && s_hi >= 0 && tb + 32 * s_hi == 3168) s_hi--;
    for (int side = 0; side < 2; side++) {
        if (side == 0 ? (s_lo == 0) : (s_hi == 78)) continue;
        fscore_H(S, fb, side == 0 ? 0 : 3168, 8, H);
        for (int t = 0; t < 8; t++) {
            cpx part[16];
            for (int c = 0; c < 16; c++) {
                float ax = 0.0f, ay = 0.0f;
                for (int i = 0; i < 7; i++) { const int r = c + 16 * i; if (r < 100) { ax = ax + H[t][r].re; ay = ay + H[t][r].im; } }
                part[c].re = ax; part[c].im = ay;
            }
            float q4[4][2];
            for (int q = 0; q < 4; q++) {
                const cpx* p = &part[4 * q];
                q4[q][0] = (p[0].re + p[1].re) + (p[2].re + p[3].re);
                q4[q][1] = (p[0].im + p[1].im) + (p[2].im + p[3].im);
            }
            const float tr = (q4[0][0] + q4[1][0]) + (q4[2][0] + q4[3][0]), ti = (q4[0][1] + q4[1][1]) + (q4[2][1] + q4[3][1]);
            const float re = tr * 0.0003125f, im = ti * 0.0003125f;
            const float m = sqrtf(re * re + im * im);
            if (side == 0) for (int s = 0; s < s_lo; s++) g[8 * s + t] = m;
            else for (int s = s_hi + 1; s <= 78; s++) g[8 * s + t] = m;
        }
    }
}

void ft8o_fine_grid(const float* spec, const ft8o_config* c, int fb, int tb, float* grid, float* score) {
    cpx* z = (cpx*)malloc(sizeof(cpx) * 3200);
    fine_zsig(spec, c, fb, z);
    for (int s = 0; s < 79; s++) fine_symbol(z, tb, s, grid + 8 * s);
    if (score) *score = fine_score(z, tb);
    free(z);
}

/* returns 1 continue, 0 stopped by Costas gate (llr untouched), -1 stopped by sd gate */
int ft8o_fine(const float* spec, const ft8o_config* c, int f0_idx, int h0_idx, int32_t* ttweak, int32_t* ftweak,
              int32_t* nsync, float* llr, float* sd, int32_t* snr, float* sgrid) {
    int fb0 = 50 * f0_idx;                                           /* int(0.5 + fHz*16), fHz = 3.125 f0 */
    int tb0 = 8 * h0_idx + (h0_idx < 0 ? 1 : 0);                     /* int(0.5 + tsec/0.005) truncates toward 0 */
    cpx* z = (cpx*)malloc(sizeof(cpx) * 3200);
    fine_zsig(spec, c, fb0, z);
    int tt = 0; float best = 0.0f;
    for (int i = 0; i < 8; i++) {                                    /* range(-8,8,2) */
        float sc = fine_score(z, tb0 + (-8 + 2 * i));
        if (i == 0 || sc > best) { best = sc; tt = -8 + 2 * i; }
    }
    int ft = 0;
    const float score_f0 = best;                                     /* f = 0: the time scan's own series and offset */
    /* The frequency-domain identities (fine_fscore, fine_grid_freq) hold while the middle Costas block of every tweak lies inside the
     * series, i.e. for h0 in [FT8O_MIN_H0_FD, FT8O_MAX_H0_FD] = [-140, 220] (search_time_range -6.1 .. +8.3 s).  A candidate further
     * out -- the reference takes any search_time_range (receiver.py:312, 319) and clamps every read (:189-195) -- is scored as the
     * reference scores it, in the time domain: one series per frequency tweak, the seven symbols of the block read at their clamped
     * positions, and the final grid symbol by symbol from the series of the chosen tweak (round 6). */
    const int far_out = h0_idx < FT8O_MIN_H0_FD || h0_idx > FT8O_MAX_H0_FD;
    for (int i = 0; i < 9; i++) {                                    /* range(-32,33,8) */
        int f = -32 + 8 * i;
        float sc;
        if (f == 0) sc = score_f0;
        else if (far_out) { fine_zsig(spec, c, fb0 + f, z); sc = fine_score(z, tb0 + tt); }
        else sc = fine_fscore(spec, fb0 + f, tb0 + tt + 32 * 36);
        if (i == 0 || sc > best) { best = sc; ft = f; }
    }
    float g[79 * 8];
    if (far_out) {
        fine_zsig(spec, c, fb0 + ft, z);
        for (int s = 0; s < 79; s++) fine_symbol(z, tb0 + tt, s, g + 8 * s);
    } else fine_grid_freq(spec, fb0 + ft, tb0 + tt, g);
    free(z);
    if (sgrid) memcpy(sgrid, g, sizeof(g));
    *ttweak = tt; *ftweak = ft;
    int nm = 0;
    for (int blk = 0; blk < 3; blk++) for (int a = 0; a < 7; a++) {  /* receiver.py:164-166 */
        const float* q = g + 8 * (36 * blk + a);
        int am = 0; for (int t = 1; t < 8; t++) if (q[t] > q[am]) am = t;
        if (am == COSTAS[a]) nm++;
    }
    *nsync = nm;
    if (nm <= 6) return 0;
    float p[464];
    for (int i = 0; i < 58; i++) for (int t = 0; t < 8; t++) p[8 * i + t] = 20.0f * ft8o_log10f(g[8 * PAYLOAD_SYM[i] + t]);
    return ft8o_db_to_llr(p, llr, sd, snr) ? 1 : -1;
}

/* ------------------------------------------------------------------ AP masks (receiver.py:21-27,109-117) */
static const int8_t AP_CQ[29]   = {0,0,0,0,0, 0,0,0,0,0, 0,0,0,0,0, 0,0,0,0,0, 0,0,0,0,0, 0,1,0,0};
static const int8_t AP_RR73[19] = {0,1, 1,1,1,1,1, 0,0,1,1,1, 0,1,0,1,0, 0,1};
static const int8_t AP_73[19]   = {0,1, 1,1,1,1,1, 0,1,0,0,1, 0,1,0,0,0, 0,1};
static const int8_t AP_RRR[19]  = {0,1, 1,1,1,1,1, 0,1,0,0,1, 0,0,1,0,0, 0,1};

void ft8o_set_ap(const float* llr0, int ap, float* llr) {
    memcpy(llr, llr0, sizeof(float) * 174);
    const int8_t* pat = NULL; int b0 = 0, n = 0;
    switch (ap) { case 1: pat = AP_CQ; b0 = 0; n = 29; break; case 2: pat = AP_RR73; b0 = 58; n = 19; break;
                  case 3: pat = AP_73; b0 = 58; n = 19; break; case 4: pat = AP_RRR; b0 = 58; n = 19; break; default: break; }
    for (int i = 0; i < n; i++) llr[b0 + i] = pat[i] ? 5.0f : -5.0f;
    if (ap == 1) { llr[74] = -5.0f; llr[75] = -5.0f; llr[76] = 5.0f; llr[57] = -5.0f; llr[58] = -5.0f; }
}

/* ------------------------------------------------------------------ CRC + message validity / rendering */
static unsigned crc14_of77(uint64_t lo, uint64_t hi) {               /* decoders.py:123-129 */
    unsigned r = 0;
    for (int i = 0; i < 96; i++) {
        unsigned b = 0;
        if (i < 77) { int pos = 76 - i; b = (unsigned)((pos >= 64 ? (hi >> (pos - 64)) : (lo >> pos)) & 1u); }
        unsigned top = (r >> 13) & 1u;
        r = ((r << 1) & 0x3FFFu) | b;
        if (top) r ^= 0x2757u;
    }
    return r;
}

typedef struct { uint32_t h; int m; char call[16]; } hent;
typedef struct { hent* e; int n, cap; } hashtab;
void* ft8o_hash_new(void) { return calloc(1, sizeof(hashtab)); }
void ft8o_hash_free(void* h) { if (h) { free(((hashtab*)h)->e); free(h); } }

static void hash_add(hashtab* ht, const char* call) {               /* databases.py:10-26 */
    if (!ht) return;
    static const char* A = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ/";
    char pad[12]; int L = (int)strlen(call);
    for (int i = 0; i < 11; i++) pad[i] = (i < L) ? call[i] : ' ';
    uint64_t x = 0;
    for (int i = 0; i < 11; i++) { const char* q = strchr(A, pad[i]); int64_t idx = (q && pad[i]) ? (q - A) : -1; x = 38 * x + (uint64_t)idx; }
    x = x * 47055833459ULL;
    static const int ms[3] = {10, 12, 22};
    for (int k = 0; k < 3; k++) {
        if (ht->n == ht->cap) { ht->cap = ht->cap ? 2 * ht->cap : 64; ht->e = (hent*)realloc(ht->e, sizeof(hent) * (size_t)ht->cap); }
        hent* e = &ht->e[ht->n++];
        e->h = (uint32_t)(x >> (64 - ms[k])); e->m = ms[k];
        strncpy(e->call, call, 15); e->call[15] = 0;
    }
}
static const char* hash_get(hashtab* ht, uint32_t h, int m) {
    if (ht) for (int i = ht->n - 1; i >= 0; i--) if (ht->e[i].m == m && ht->e[i].h == h) return ht->e[i].call;
    return "...";
}

static int alnum36(char ch) { return (ch >= '0' && ch <= '9') ? ch - '0' : (ch >= 'A' && ch <= 'Z') ? ch - 'A' + 10 : -1; }

/* decoders.py:95-115: 28-bit standard call -> text, NULL-equivalent (0) when implausible */
static int std_call28(uint32_t c28, char* out) {
    static const char* A0 = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ";
    static const char* A1 = "0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ";
    static const char* A3 = " ABCDEFGHIJKLMNOPQRSTUVWXYZ";
    char ch[7];
    int64_t nn = (int64_t)c28 - (2063592 + 4194304);
    if (nn < 0) { strcpy(ch, "ZZ9ZZZ"); }                             /* python divmod/negative-index quirk at c28 = 6257895 */
    else {
        int i0 = (int)(nn / 7085880); nn %= 7085880;                   /* 36*10*27^3 */
        int i1 = (int)(nn / 196830);  nn %= 196830;                    /* 10*27^3 */
        int i2 = (int)(nn / 19683);   nn %= 19683;
        int i3 = (int)(nn / 729);     nn %= 729;
        int i4 = (int)(nn / 27);      int i5 = (int)(nn % 27);
        ch[0] = A0[i0]; ch[1] = A1[i1]; ch[2] = (i2 < 10) ? (char)('0' + i2) : ' ';
        ch[3] = A3[i3]; ch[4] = A3[i4]; ch[5] = A3[i5]; ch[6] = 0;
    }
    int a = 0, b = 6;
    while (a < b && ch[a] == ' ') a++;
    while (b > a && ch[b - 1] == ' ') b--;
    int L = b - a;
    memcpy(out, ch + a, (size_t)L); out[L] = 0;
    if (L < 3) return 0;
    for (int i = 0; i < L; i++) if (out[i] == ' ') return 0;
    int d1 = out[1] >= '0' && out[1] <= '9', d2 = out[2] >= '0' && out[2] <= '9';
    if (out[0] >= 'A' && out[0] <= 'Z' && ((FT8_PFX1_MASK >> (out[0] - 'A')) & 1u) && d1)
        if (!(((FT8_PFX1_TRAP >> (out[0] - 'A')) & 1u) && d2)) return 1;
    int x0 = alnum36(out[0]), x1 = alnum36(out[1]);
    if (x0 >= 0 && x1 >= 0 && ((FT8_PFX2[x0] >> x1) & 1ULL) && d2) return 1;
    return 0;
}

/* decoders.py:70-93.  returns 0 for None */
static int call_29(hashtab* ht, uint32_t c29, int i3, char* out) {
    uint32_t pr = c29 & 1u, c28 = c29 >> 1;
    if (c28 < 3) { strcpy(out, c28 == 0 ? "DE" : c28 == 1 ? "QRZ" : "CQ"); return 1; }
    if (c28 < 1004) { sprintf(out, "CQ %03u", c28 - 3); return 1; }
    if (c28 < 21443) {
        uint32_t x = c28 - 1003; char t[5]; t[4] = 0;
        for (int i = 3; i >= 0; i--) { t[i] = " ABCDEFGHIJKLMNOPQRSTUVWXYZ"[x % 27]; x /= 27; }
        int a = 0, b = 4; while (a < b && t[a] == ' ') a++; while (b > a && t[b - 1] == ' ') b--;
        t[b] = 0; sprintf(out, "CQ %s", t + a); return 1;
    }
    if (c28 < 2063592u + 4194303u) { snprintf(out, 16, "<%.13s>", hash_get(ht, c28 - 2063592u, 22)); return 1; }
    char call[16];
    if (!std_call28(c28, call)) return 0;
    if (pr) strcat(call, i3 == 2 ? "/P" : "/R");
    size_t L = strlen(call);
    if (L >= 2 && call[L - 2] == '/' && call[L - 1] == 'R' && !(call[0] == 'A' || call[0] == 'K' || call[0] == 'N' || call[0] == 'W')) return 0;
    hash_add(ht, call);
    strcpy(out, call);
    return 1;
}

/* decoders.py:16-68.  Returns 1 and fills out[3] when unpack() yields a tuple, 0 for None.
 * With hash == NULL it is the pure validity predicate (no side effects). */
int ft8o_unpack77(void* hash, uint64_t lo, uint64_t hi, char out[3][16]) {
    hashtab* ht = (hashtab*)hash;
    char tmp[3][16];
    if (!out) out = tmp;
    out[0][0] = out[1][0] = out[2][0] = 0;
    if (lo == 0 && hi == 0) return 0;
    unsigned i3 = (unsigned)(lo & 7u);
    if (i3 == 1 || i3 == 2) {
        uint32_t g16 = (uint32_t)((lo >> 3) & 0xFFFFu);
        uint32_t cb29 = (uint32_t)((lo >> 19) & 0x1FFFFFFFu);
        uint32_t ca29 = (uint32_t)(((lo >> 48) | (hi << 16)) & 0x1FFFFFFFu);
        uint32_t g15 = g16 & 0x7FFFu;
        if (g15 == 0) return 0;
        char g[16];
        if (g15 < 32400) {
            unsigned a = g15 / 1800, nn = g15 % 1800, b = nn / 100; nn %= 100;
            g[0] = (char)('A' + a); g[1] = (char)('A' + b); g[2] = (char)('0' + nn / 10); g[3] = (char)('0' + nn % 10); g[4] = 0;
        } else if (g15 - 32400 <= 4) {
            static const char* T[5] = {"", "", "RRR", "RR73", "73"};
            strcpy(g, T[g15 - 32400]);
        } else {
            int v = (int)g15 - 32435;
            sprintf(g, "%s%+03d", (g16 >> 15) ? "R" : "", v);
        }
        int oka = call_29(ht, ca29, (int)i3, out[0]);
        int okb = call_29(ht, cb29, (int)i3, out[1]);
        strcpy(out[2], g);
        if (!oka || !okb || g[0] == 0) return 0;
        return 1;
    }
    if (i3 == 4) {
        unsigned cq = (unsigned)((lo >> 3) & 1u), rrr = (unsigned)((lo >> 4) & 3u), swp = (unsigned)((lo >> 6) & 1u);
        uint64_t c58 = ((lo >> 7) | (hi << 57)) & ((1ULL << 58) - 1);
        uint32_t hsh = (uint32_t)((hi >> 1) & 0xFFFu);
        if ((cq && rrr) || (!cq && !rrr)) return 0;
        char ca[16], cb[16], t[13];
        if (cq) strcpy(ca, "CQ"); else snprintf(ca, 16, "<%.13s>", hash_get(ht, hsh, 12));
        for (int i = 11; i >= 0; i--) { t[i] = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ/"[c58 % 38]; c58 /= 38; }
        t[12] = 0;
        int a = 0, b = 12; while (a < b && t[a] == ' ') a++; while (b > a && t[b - 1] == ' ') b--;
        t[b] = 0; strcpy(cb, t + a);
        hash_add(ht, cb);
        static const char* R[4] = {"", "RRR", "RR73", "73"};
        strcpy(out[0], swp ? cb : ca); strcpy(out[1], swp ? ca : cb); strcpy(out[2], R[rrr]);
        return 1;
    }
    return 0;
}

int ft8o_valid77(uint64_t lo, uint64_t hi) { return ft8o_unpack77(NULL, lo, hi, NULL); }

/* acceptance callback: called for every CRC-passing, non-zero 77-bit word in reference call order */
typedef int (*accept_fn)(void* ctx, uint64_t lo, uint64_t hi, int seq);
static int accept_pure(void* ctx, uint64_t lo, uint64_t hi, int seq) { (void)ctx; (void)seq; return ft8o_valid77(lo, hi); }

/* hard bits of 91 values -> (msg, crc); decoders.py:117-131.  0: fail, 1: crc ok but unpack None, 2: accepted */
static int crc_check_bits(const uint8_t* b91, accept_fn acc, void* ctx, int seq, uint64_t* lo_out, uint64_t* hi_out) {
    uint64_t lo = 0, hi = 0; unsigned crc = 0;
    for (int k = 0; k < 77; k++) if (b91[k]) { int pos = 76 - k; if (pos >= 64) hi |= 1ULL << (pos - 64); else lo |= 1ULL << pos; }
    for (int k = 77; k < 91; k++) crc = (crc << 1) | (b91[k] ? 1u : 0u);
    if (lo == 0 && hi == 0) return 0;
    if (crc14_of77(lo, hi) != crc) return 0;
    if (lo_out) { *lo_out = lo; *hi_out = hi; }
    return acc(ctx, lo, hi, seq) ? 2 : 1;
}

int ft8o_crc_valid91(const float* llr91, uint64_t* lo, uint64_t* hi) {
    uint8_t b[91]; for (int k = 0; k < 91; k++) b[k] = llr91[k] > 0.0f;
    return crc_check_bits(b, accept_pure, NULL, 0, lo, hi);
}

/* ------------------------------------------------------------------ LDPC BP (decoders.py:140-171) */
static int ldpc_core(float* llr, int max_nc0, int max_iters, accept_fn acc, void* ctx,
                     uint64_t* lo, uint64_t* hi, int32_t* n_its, int32_t* has_out) {
    float mc2v[FT8_NEDGE], newm[FT8_NEDGE], delta[FT8_NEDGE];
    memset(mc2v, 0, sizeof(mc2v));
    *n_its = -1; *has_out = 1;
    for (int it = 0; it < max_iters; it++) {
        int ncheck = 0;
        for (int c = 0; c < 83; c++) {
            int par = 0;
            for (int j = 0; j < FT8_CHK_N[c]; j++) par ^= (llr[FT8_CHK_V[c][j]] > 0.0f);
            ncheck += par;
        }
        if (it == 0 && ncheck > max_nc0) { *has_out = 0; return 0; }
        if (ncheck == 0) {
            uint8_t b[91]; for (int k = 0; k < 91; k++) b[k] = llr[k] > 0.0f;
            int r = crc_check_bits(b, acc, ctx, it + 1, lo, hi);
            if (r == 2) { *n_its = it; *has_out = 0; return 1; }
            /* reference does nothing this iteration => state is frozen; if the CRC passed (but unpack gave
             * None) every remaining iteration repeats the same failing unpack call.  The repeat count is
             * handed to the caller through n_its so the event log mirrors the reference's call sequence. */
            *n_its = (r == 1) ? -(2 + (max_iters - 1 - it)) : -1;
            return 0;
        }
        for (int c = 0; c < 83; c++) {
            int e0 = FT8_CHK_E0[c], n = FT8_CHK_N[c];
            float t[7], P = 0.0f;
            for (int j = 0; j < n; j++) {
                float v2c = llr[FT8_CHK_V[c][j]] - mc2v[e0 + j];
                t[j] = ft8o_tanhf(-v2c);
                P = (j == 0) ? t[0] : P * t[j];
            }
            for (int j = 0; j < n; j++) {
                /* reference: e = P/t; m = e/((e-1.18)(1.18+e)) (decoders.py:146-149).  Contract: the same quantity with
                 * numerator and denominator multiplied by t^2 -- one division; t == 0 (=> P == 0) still gives 0/0 = NaN */
                /* (round 3: P -+ 1.18 t as fused multiply-adds) */
                float nm = (P * t[j]) / (fmaf(-1.18f, t[j], P) * fmaf(1.18f, t[j], P));
                newm[e0 + j] = nm;
                delta[e0 + j] = nm - mc2v[e0 + j];
            }
        }
        for (int v = 0; v < 174; v++) {
            float col = 0.0f;
            col += delta[FT8_VAR_E[v][0]]; col += delta[FT8_VAR_E[v][1]]; col += delta[FT8_VAR_E[v][2]];
            llr[v] += col;
        }
        memcpy(mc2v, newm, sizeof(mc2v));
    }
    return 0;
}

int ft8o_ldpc(float* llr, int max_nc0, int max_iters, uint64_t* lo, uint64_t* hi, int32_t* n_its, int32_t* has_out) {
    return ldpc_core(llr, max_nc0, max_iters, accept_pure, NULL, lo, hi, n_its, has_out);
}

/* ------------------------------------------------------------------ np.argsort of a float32 vector (decoders.py:226)
 * osd_012 orders the 174 columns with np.argsort(-np.abs(llr)).  numpy is a third-party dependency of the reference (not under
 * /root/reference; pyproject.toml pins only >= 1.24) and its default argsort is UNSTABLE, so the order of equal keys -- the AP bits all
 * sit at |llr| = 5 -- is whatever the algorithm of the installed build does.  The reference outputs this oracle is pinned to were
 * produced by numpy 2.2.6 on an AVX-512 host, where np.argsort(float32) is x86-simd-sort's avx512_argsort (vendored under
 * numpy/_core/src/npysort/x86-simd-sort; dispatched from x86_simd_argsort.dispatch.cpp with hasnan = true).  Restated here from that
 * library's published algorithm:
 *   - a vector containing a NaN: std_argsort_withnan (src/xss-common-argsort.h) = std::sort of the index array with the comparator
 *     "both not NaN: a < b; a NaN: false; else true" -- libstdc++'s introsort (bits/stl_algo.h: __introsort_loop with depth limit
 *     2 lg n, median-of-three to first, unguarded partition, threshold 16, heapsort fallback, final insertion sort);
 *   - otherwise, n <= 256: argsort_n<..., 256> (xss-common-argsort.h), a DATA-OBLIVIOUS bitonic network over maxN = the smallest of
 *     8, 16, ..., 256 with 2 n > maxN, 8 keys per register (32-bit keys travel in ymm registers next to 64-bit indices in zmm), padded
 *     with +inf: each register sorted by sort_zmm_64bit (avx512-64bit-argsort.hpp), then bitonic_fullmerge_n_vec
 *     (xss-network-keyvaluesort.hpp): for 2, 4, ... registers per group, COEX of register i with the REVERSED register n-1-i,
 *     bitonic_clean_n_vec, bitonic_merge_zmm_64bit inside every register.  Every compare-exchange keeps a lane's (key, index) when
 *     min/max returns its own key (cmp_merge / COEX: mask = eq(result, own key)), i.e. it swaps iff key[lower wire] > key[upper wire]
 *     STRICTLY -- equal keys never move.  With wire = 8 * register + lane, every stage pairs wire w with w ^ m, the lower wire taking
 *     the minimum; the masks m are listed by argsort_masks() below.
 * (n > 256 partitions around a data-dependent pivot first; osd_012 never gets there.)
 * Pinned: tests/test_oracle_golden.py::test_argsort_equals_numpy runs this against np.argsort of the installed numpy on tie-laden,
 * NaN-laden and random vectors of every length up to 256 (where the installed numpy is the AVX-512 build); tools/argsort_check.py is
 * the 10^6-vector version of it (profiles/r05_argsort_check.txt). */
static int argsort_masks(int nvec, int* m) {
    static const int first[6] = {1, 3, 1, 7, 2, 1};                    /* sort_zmm_64bit: lanes ^1, reverse of 4, ^1, reverse of 8, ^2, ^1 */
    int n = 0;
    for (int i = 0; i < 6; i++) m[n++] = first[i];
    for (int per = 2; per <= nvec; per *= 2) {
        m[n++] = 8 * per - 1;                                           /* register i against the reversed register per-1-i */
        for (int num = per / 2; num >= 2; num /= 2) m[n++] = 8 * (num / 2);      /* bitonic_clean_n_vec: registers num/2 apart */
        m[n++] = 4; m[n++] = 2; m[n++] = 1;                             /* bitonic_merge_zmm_64bit */
    }
    return n;
}

static int nan_less(const float* x, int32_t a, int32_t b) {             /* std_argsort_withnan's comparator */
    if (!isnan(x[a]) && !isnan(x[b])) return x[a] < x[b];
    return isnan(x[a]) ? 0 : 1;
}
/* libstdc++ std::sort(first, last, comp) on an index array (bits/stl_algo.h, bits/stl_heap.h), function for function */
static void ss_unguarded_linear_insert(const float* x, int32_t* a, int last) {
    int32_t val = a[last]; int next = last - 1;
    while (nan_less(x, val, a[next])) { a[last] = a[next]; last = next; next--; }
    a[last] = val;
}
static void ss_insertion_sort(const float* x, int32_t* a, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; i++) {
        if (nan_less(x, a[i], a[first])) { int32_t val = a[i]; memmove(a + first + 1, a + first, sizeof(int32_t) * (size_t)(i - first)); a[first] = val; }
        else ss_unguarded_linear_insert(x, a, i);
    }
}
static void ss_push_heap(const float* x, int32_t* a, int first, int hole, int top, int32_t value) {
    int parent = (hole - 1) / 2;
    while (hole > top && nan_less(x, a[first + parent], value)) { a[first + hole] = a[first + parent]; hole = parent; parent = (hole - 1) / 2; }
    a[first + hole] = value;
}
static void ss_adjust_heap(const float* x, int32_t* a, int first, int hole, int len, int32_t value) {
    const int top = hole; int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (nan_less(x, a[first + child], a[first + child - 1])) child--;
        a[first + hole] = a[first + child]; hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); a[first + hole] = a[first + child - 1]; hole = child - 1; }
    ss_push_heap(x, a, first, hole, top, value);
}
static void ss_heapsort(const float* x, int32_t* a, int first, int last) {        /* __partial_sort(first, last, last) */
    const int len = last - first;
    if (len >= 2) for (int parent = (len - 2) / 2; ; parent--) { ss_adjust_heap(x, a, first, parent, len, a[first + parent]); if (parent == 0) break; }
    while (last - first > 1) { last--; int32_t v = a[last]; a[last] = a[first]; ss_adjust_heap(x, a, first, 0, last - first, v); }
}
static void ss_median_to_first(const float* x, int32_t* a, int result, int p, int q, int r) {
    int pick;
    if (nan_less(x, a[p], a[q])) pick = nan_less(x, a[q], a[r]) ? q : (nan_less(x, a[p], a[r]) ? r : p);
    else pick = nan_less(x, a[p], a[r]) ? p : (nan_less(x, a[q], a[r]) ? r : q);
    int32_t t = a[result]; a[result] = a[pick]; a[pick] = t;
}
static void ss_introsort_loop(const float* x, int32_t* a, int first, int last, int depth) {
    while (last - first > 16) {
        if (depth == 0) { ss_heapsort(x, a, first, last); return; }
        depth--;
        ss_median_to_first(x, a, first, first + 1, first + (last - first) / 2, last - 1);
        int lo = first + 1, hi = last;                                  /* __unguarded_partition(first + 1, last, pivot = first) */
        for (;;) {
            while (nan_less(x, a[lo], a[first])) lo++;
            hi--;
            while (nan_less(x, a[first], a[hi])) hi--;
            if (!(lo < hi)) break;
            int32_t t = a[lo]; a[lo] = a[hi]; a[hi] = t;
            lo++;
        }
        ss_introsort_loop(x, a, lo, last, depth);
        last = lo;
    }
}
static void std_sort_withnan(const float* x, int32_t* a, int n) {
    if (n < 2) return;
    int lg = 0; while ((n >> (lg + 1)) != 0) lg++;
    ss_introsort_loop(x, a, 0, n, 2 * lg);
    if (n > 16) { ss_insertion_sort(x, a, 0, 16); for (int i = 16; i < n; i++) ss_unguarded_linear_insert(x, a, i); }
    else ss_insertion_sort(x, a, 0, n);
}

int ft8o_argsort_f32(const float* x, int n, int32_t* out) {
    if (n < 0 || n > 256) return -1;
    for (int i = 0; i < n; i++) out[i] = i;
    if (n < 2) return 0;
    int has_nan = 0;
    for (int i = 0; i < n; i++) if (isnan(x[i])) has_nan = 1;
    if (has_nan) { std_sort_withnan(x, out, n); return 0; }
    int maxn = 256;
    while (maxn > 8 && 2 * n <= maxn) maxn /= 2;
    float key[256]; int32_t idx[256];
    for (int w = 0; w < maxn; w++) { key[w] = (w < n) ? x[w] : INFINITY; idx[w] = (w < n) ? w : 0; }
    int masks[40];
    const int nm = argsort_masks(maxn / 8, masks);
    for (int s = 0; s < nm; s++) {
        const int m = masks[s];
        for (int w = 0; w < maxn; w++) {
            const int u = w ^ m;
            if (u > w && key[w] > key[u]) { float tk = key[w]; key[w] = key[u]; key[u] = tk; int32_t ti = idx[w]; idx[w] = idx[u]; idx[u] = ti; }
        }
    }
    memcpy(out, idx, sizeof(int32_t) * (size_t)n);
    return 0;
}

/* ------------------------------------------------------------------ OSD (decoders.py:223-272) */
static void cw91_to_bits(const uint64_t* w, uint8_t* b91) { for (int k = 0; k < 91; k++) b91[k] = (uint8_t)((w[k >> 6] >> (k & 63)) & 1ULL); }

/* One OSD trial: codeword words w[3] (174 bits).  The extension gate (osd_max_hd > 0) drops the trial -- no unpack() call -- when the
 * codeword is further than max_hd bit positions from the hard decisions hard[3]. */
static int osd_trial(const uint64_t* w, const uint64_t* hard, int max_hd, accept_fn acc, void* ctx, int trial,
                     uint64_t* lo, uint64_t* hi, int32_t* hd_out) {
    const uint64_t M1 = (1ULL << 27) - 1, M2 = (1ULL << 46) - 1;
    int hd = __builtin_popcountll(w[0] ^ hard[0]) + __builtin_popcountll(w[1] ^ hard[1]) + __builtin_popcountll((w[2] ^ hard[2]) & M2);
    if (max_hd > 0 && hd > max_hd) return 0;
    uint64_t m[2] = {w[0], w[1] & M1};
    uint8_t b[91]; cw91_to_bits(m, b);
    int r = crc_check_bits(b, acc, ctx, trial, lo, hi);
    if (r == 2 && hd_out) *hd_out = hd;
    return r;
}

static int osd_core(const float* llr, int singles, int doubles, int triples, int max_hd, accept_fn acc, void* ctx,
                    uint64_t* lo, uint64_t* hi, int32_t* trial_out, int32_t* info_cols, int32_t* hd_out) {
    /* reliability order: colperm = np.argsort(-np.abs(llr)) (decoders.py:226) exactly as the reference's numpy orders it, equal keys
     * (the AP bits at |llr| = 5) and NaNs included: ft8o_argsort_f32 above */
    int32_t order[174];
    float key[174];
    for (int i = 0; i < 174; i++) key[i] = -fabsf(llr[i]);
    ft8o_argsort_f32(key, 174, order);
    uint64_t G[91][3];
    memcpy(G, FT8_G0, sizeof(G));
    uint8_t used[91]; memset(used, 0, sizeof(used));
    int prow[91], pcol[91], k = 0;
    for (int ic = 0; ic < 174 && k < 91; ic++) {
        int col = order[ic], w = col >> 6; uint64_t bit = 1ULL << (col & 63);
        int piv = -1;
        for (int r = 0; r < 91; r++) if (!used[r] && (G[r][w] & bit)) { piv = r; break; }
        if (piv < 0) continue;
        used[piv] = 1;
        for (int r = 0; r < 91; r++) if (r != piv && (G[r][w] & bit)) { G[r][0] ^= G[piv][0]; G[r][1] ^= G[piv][1]; G[r][2] ^= G[piv][2]; }
        prow[k] = piv; pcol[k] = col; k++;
    }
    if (info_cols) for (int i = 0; i < 91; i++) info_cols[i] = (i < k) ? pcol[i] : -1;
    uint64_t hard[3] = {0, 0, 0};
    for (int v = 0; v < 174; v++) if (llr[v] > 0.0f) hard[v >> 6] |= 1ULL << (v & 63);
    uint64_t cw0[3] = {0, 0, 0};
    for (int i = 0; i < k; i++) if (llr[pcol[i]] > 0.0f) { cw0[0] ^= G[prow[i]][0]; cw0[1] ^= G[prow[i]][1]; cw0[2] ^= G[prow[i]][2]; }
    int trial = 0;
#define FLIP(i) G[prow[90 - (i)]]
    /* order-0 */
    if (osd_trial(cw0, hard, max_hd, acc, ctx, trial, lo, hi, hd_out) == 2) { *trial_out = trial; return 1; }
    trial++;
    for (int i = 0; i < singles; i++, trial++) {
        const uint64_t* f = FLIP(i);
        uint64_t w[3] = {cw0[0] ^ f[0], cw0[1] ^ f[1], cw0[2] ^ f[2]};
        if (osd_trial(w, hard, max_hd, acc, ctx, trial, lo, hi, hd_out) == 2) { *trial_out = trial; return 1; }
    }
    for (int i = 0; i < singles; i++) for (int j = 0; j < doubles; j++) if (j < i) {
        const uint64_t* f = FLIP(i); const uint64_t* g = FLIP(j);
        uint64_t w[3] = {cw0[0] ^ f[0] ^ g[0], cw0[1] ^ f[1] ^ g[1], cw0[2] ^ f[2] ^ g[2]};
        if (osd_trial(w, hard, max_hd, acc, ctx, trial, lo, hi, hd_out) == 2) { *trial_out = trial; return 1; }
        trial++;
    }
    /* extension: order-3 reprocessing over the `triples` least reliable basis positions (no reference counterpart) */
    for (int i = 0; i < triples; i++) for (int j = 0; j < i; j++) for (int q = 0; q < j; q++) {
        const uint64_t* f = FLIP(i); const uint64_t* g = FLIP(j); const uint64_t* e = FLIP(q);
        uint64_t w[3] = {cw0[0] ^ f[0] ^ g[0] ^ e[0], cw0[1] ^ f[1] ^ g[1] ^ e[1], cw0[2] ^ f[2] ^ g[2] ^ e[2]};
        if (osd_trial(w, hard, max_hd, acc, ctx, trial, lo, hi, hd_out) == 2) { *trial_out = trial; return 1; }
        trial++;
    }
#undef FLIP
    *trial_out = -1;
    return 0;
}

int ft8o_osd(const float* llr, int singles, int doubles, uint64_t* lo, uint64_t* hi, int32_t* trial, int32_t* info_cols) {
    return osd_core(llr, singles, doubles, 0, 0, accept_pure, NULL, lo, hi, trial, info_cols, NULL);
}
int ft8o_osd_ext(const float* llr, int singles, int doubles, int triples, int max_hd, uint64_t* lo, uint64_t* hi, int32_t* trial,
                 int32_t* info_cols, int32_t* hd_out) {
    return osd_core(llr, singles, doubles, triples, max_hd, accept_pure, NULL, lo, hi, trial, info_cols, hd_out);
}

/* ------------------------------------------------------------------ whole frame (receiver.py:68-107, 338-367, 389-398) */
typedef struct {
    hashtab* ht; ft8o_event* log; int32_t cap, n; int cand, ipass, slot;
    char last[3][16]; int repeat;
} fctx;

static int accept_frame(void* vctx, uint64_t lo, uint64_t hi, int seq) {
    fctx* f = (fctx*)vctx;
    int v = ft8o_unpack77(f->ht, lo, hi, f->last);
    if (f->n < f->cap) { ft8o_event* e = &f->log[f->n]; e->msg_lo = lo; e->msg_hi = hi; e->cand = f->cand; e->ipass = f->ipass; e->valid = v; e->pad = (f->slot << 16) | (seq & 0xffff); }
    f->n++;
    return v;
}

typedef struct { float llr0[174]; float llr[174]; float saved[5][174]; int saved_ap[5]; int n_saved; float sd_key; int ipass; } cstate;

static int run_ldpc_frame(fctx* fc, float* llr, int nc0, int iters, ft8o_cand* c, int32_t* has_out) {
    uint64_t lo, hi; int32_t nits;
    int ok = ldpc_core(llr, nc0, iters, accept_frame, fc, &lo, &hi, &nits, has_out);
    if (!ok && nits <= -2) {
        /* frozen all-checks-satisfied state: the reference repeats the failing unpack each remaining iteration */
        int rep = -nits - 2;
        if (fc->n > 0 && fc->n <= fc->cap) { ft8o_event last = fc->log[fc->n - 1]; for (int i = 0; i < rep; i++) accept_frame(fc, last.msg_lo, last.msg_hi, (last.pad & 0xffff) + 1 + i); }
        else for (int i = 0; i < rep; i++) fc->n++;
    }
    if (ok) { c->msg_lo = lo; c->msg_hi = hi; c->n_its = nits; }
    return ok;
}

int ft8o_decode_frame(const int16_t* audio, const ft8o_config* cfg, ft8o_cand* cands, int32_t* n_cands,
                      ft8o_event* log, int32_t log_cap, int32_t* n_log, ft8o_msg* msgs, int32_t msg_cap, int32_t* n_msgs) {
    float* grid = (float*)malloc(sizeof(float) * FT8O_GRID_ROWS * FT8O_GRID_COLS);
    float* spec = NULL;
    ft8o_cand* all = (ft8o_cand*)malloc(sizeof(ft8o_cand) * 2048);     /* one candidate per f0 bin at most: < 960 (1888 in the wide build) */
    ft8o_spectrogram(audio, cfg, grid);
    int n = ft8o_sync_search(grid, cfg, all);
    memcpy(cands, all, sizeof(ft8o_cand) * (size_t)n);
    free(all);
    *n_cands = n;
    cstate* st = (cstate*)calloc((size_t)(n > 0 ? n : 1), sizeof(cstate));
    fctx fc; memset(&fc, 0, sizeof(fc));
    fc.ht = (hashtab*)ft8o_hash_new(); fc.log = log; fc.cap = log_cap;
    int nm = 0;
    char (*seen)[48] = (char (*)[48])malloc(48 * (size_t)(n > 0 ? n : 1)); int nseen = 0;
    int* order = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int rnd = 0; rnd < 8; rnd++) {
        int m = 0;
        for (int i = 0; i < n; i++) if (cands[i].status == FT8O_ST_NONE) order[m++] = i;
        for (int i = 1; i < m; i++) {                                  /* stable, llr_sd descending (receiver.py:392) */
            int t = order[i]; int j = i - 1;
            while (j >= 0 && st[order[j]].sd_key < st[t].sd_key) { order[j + 1] = order[j]; j--; }
            order[j + 1] = t;
        }
        for (int oi = 0; oi < m; oi++) {
            int ci = order[oi]; ft8o_cand* c = &cands[ci]; cstate* s = &st[ci];
            int ip = s->ipass; fc.cand = ci; fc.ipass = ip;
            int done = 0; int32_t has_out;
            uint64_t lo, hi;
            if (ip == 0) {
                float p[464];
                ft8o_payload(grid, c->f0_idx, c->h0_idx, p);
                int ok = ft8o_db_to_llr(p, s->llr, &c->grid_sd, &c->snr_grid);
                s->sd_key = c->grid_sd;
                memcpy(s->llr0, s->llr, sizeof(s->llr0));
                if (!ok) c->status = FT8O_ST_STOP_GRID_SD;
                else for (int ap = 0; ap < 5 && !done; ap++) {
                    fc.slot = ap;
                    ft8o_set_ap(s->llr0, ap, s->llr);
                    uint8_t b[91]; for (int k = 0; k < 91; k++) b[k] = s->llr[k] > 0.0f;
                    if (crc_check_bits(b, accept_frame, &fc, 0, &lo, &hi) == 2) { done = 1; c->method = FT8O_M_GOOD91; c->ap = ap; c->msg_lo = lo; c->msg_hi = hi; c->n_its = 0; break; }
                    if (run_ldpc_frame(&fc, s->llr, cfg->bp_nc0_a, cfg->bp_iters_a, c, &has_out)) { done = 1; c->method = FT8O_M_LDPC_A; c->ap = ap; }
                }
            } else if (ip == 1) {
                if (!spec) { spec = (float*)malloc(sizeof(float) * 2 * FT8O_SPEC_BINS); ft8o_cycle_spectrum(audio, cfg, spec); }
                float sd; int32_t snr;
                int r = ft8o_fine(spec, cfg, c->f0_idx, c->h0_idx, &c->ttweak, &c->ftweak, &c->nsync, s->llr, &sd, &snr, NULL);
                if (r == 0) c->status = FT8O_ST_STOP_COSTAS;
                else { c->fine_sd = sd; c->snr_fine = snr; s->sd_key = sd; if (r < 0) c->status = FT8O_ST_STOP_FINE_SD; }
            } else if (ip == 2) {
                memcpy(s->llr0, s->llr, sizeof(s->llr0));
                for (int ap = 0; ap < 2 && !done; ap++) {
                    fc.slot = ap;
                    ft8o_set_ap(s->llr0, ap, s->llr);
                    uint8_t b[91]; for (int k = 0; k < 91; k++) b[k] = s->llr[k] > 0.0f;
                    if (crc_check_bits(b, accept_frame, &fc, 0, &lo, &hi) == 2) { done = 1; c->method = FT8O_M_GOOD91; c->ap = ap; c->msg_lo = lo; c->msg_hi = hi; c->n_its = 0; }
                }
            } else if (ip == 3) {
                for (int ap = 0; ap < 2 && !done; ap++) {
                    fc.slot = ap;
                    ft8o_set_ap(s->llr0, ap, s->llr);
                    if (run_ldpc_frame(&fc, s->llr, cfg->bp_nc0_a, cfg->bp_iters_a, c, &has_out)) { done = 1; c->method = FT8O_M_LDPC_A; c->ap = ap; }
                }
            } else if (ip == 4) {
                for (int ap = 0; ap < 5 && !done; ap++) {
                    fc.slot = ap;
                    ft8o_set_ap(s->llr0, ap, s->llr);
                    if (run_ldpc_frame(&fc, s->llr, cfg->bp_nc0_b, cfg->bp_iters_b, c, &has_out)) { done = 1; c->method = FT8O_M_LDPC_B; c->ap = ap; }
                    else if (has_out) { memcpy(s->saved[s->n_saved], s->llr, sizeof(s->llr)); s->saved_ap[s->n_saved++] = ap; }
                }
            } else if (ip == 5) {
                for (int ap = 0; ap < 5 && !done; ap++) {
                    fc.slot = ap;
                    ft8o_set_ap(s->llr0, ap, s->llr);
                    int32_t trial;
                    if (osd_core(s->llr, cfg->osd_single, cfg->osd_double, cfg->osd_triple, cfg->osd_max_hd, accept_frame, &fc, &lo, &hi, &trial, NULL, NULL)) {
                        done = 1; c->method = FT8O_M_OSD; c->ap = ap; c->msg_lo = lo; c->msg_hi = hi; c->n_its = trial; }
                }
            } else if (ip == 6) {
                for (int k = 0; k < s->n_saved && !done; k++) {
                    fc.slot = 5 + s->saved_ap[k];
                    int32_t trial;
                    if (osd_core(s->saved[k], cfg->osd_single, cfg->osd_double, cfg->osd_triple, cfg->osd_max_hd, accept_frame, &fc, &lo, &hi, &trial, NULL, NULL)) {
                        done = 1; c->method = FT8O_M_LDPC_B_OSD; c->ap = s->saved_ap[k]; c->msg_lo = lo; c->msg_hi = hi; c->n_its = trial; }
                }
            } else {
                c->status = FT8O_ST_EXHAUSTED;
            }
            s->ipass = ip + 1;
            if (done) {
                c->status = FT8O_ST_DECODED; c->ipass = ip;
                char key[48]; snprintf(key, sizeof(key), "%s %s %s", fc.last[0], fc.last[1], fc.last[2]);
                int dup = 0; for (int i = 0; i < nseen; i++) if (!strcmp(seen[i], key)) { dup = 1; break; }
                if (!dup) {
                    strcpy(seen[nseen++], key);
                    if (nm < msg_cap) {
                        ft8o_msg* mo = &msgs[nm]; memset(mo, 0, sizeof(*mo));
                        memcpy(mo->f, fc.last, sizeof(mo->f));
                        mo->cand = ci; mo->fine = (ip >= 2);
                        mo->snr = mo->fine ? c->snr_fine : c->snr_grid;
                        mo->tsec = (double)c->h0_idx / 25.0; mo->fHz = 3.125 * (double)c->f0_idx;
                        if (mo->fine) { mo->tsec = mo->tsec + (double)c->ttweak / 200.0; mo->fHz = mo->fHz + (double)c->ftweak / 16.0; }
                        mo->ipass = ip; mo->ap = c->ap; mo->method = c->method;
                        mo->ttweak = mo->fine ? c->ttweak : 0; mo->ftweak = mo->fine ? c->ftweak : 0;
                    }
                    nm++;
                }
            }
        }
    }
    *n_log = fc.n; *n_msgs = nm;
    ft8o_hash_free(fc.ht); free(grid); free(spec); free(st); free(seen); free(order);
    return 0;
}


/* ------------------------------------------------------------------------------------------------------------------
 * Signal subtraction.  Restates Receiver.subtract_signal (reference tests/pipeline/receiver_sub.py:380-402):
 *   sig   = symbols_to_complex_audio(symbols, f_base = fHz - 0.5)              (PyFT8/transmitter.py:52-70, pulse :41-50)
 *   s0    = int(12000 * tsec); only if s0 > 0 and the 151680 samples fit
 *   y     = audio[s0 : s0+L] * conj(sig)                     (stored as complex64 in a 192000-sample zero-padded array)
 *   A     = fft(y)[0:20]; everything else zeroed; the cos^2 "window" of :393-395 is the single value 1.0
 *   a     = ifft(A)                                           (complex64)
 *   audio[s0 : s0+L] = audio - 2 * real(a[:L] * sig)          (stored as float32)
 * The two 192000-point FFTs are replaced by the 20 DFT bins they keep, evaluated directly in double precision (the
 * reference's complex64 FFT differs from this by its own rounding, ~1e-6 of the signal amplitude).
 * symbols_to_complex_audio: phase = running sum of dphi (= 2 pi / 1920 * sum_sym tone * pulse, pulse = 3-symbol erf shape, BT = 2)
 * + 2 pi f n / 12000 over 81 symbol slots, plus the author's edge terms that add the half pulses of the first / last tone to
 * the PHASE (not to dphi); the first and last slot are cut off; 240-sample raised-cosine ramps with cos(pi i / 239). */
#define SUB_SPS 1920
#define SUB_L (79 * SUB_SPS)
static double g_sub_pulse[3 * SUB_SPS], g_sub_pc[3 * SUB_SPS];
static int g_sub_init = 0;
static void sub_init(void) {
    if (g_sub_init) return;
    const double c = M_PI * sqrt(2.0 / log(2.0)), bt = 2.0;
    double acc = 0.0;
    for (int i = 0; i < 3 * SUB_SPS; i++) {
        const double tt = ((double)i - 1.5 * SUB_SPS) / SUB_SPS;
        g_sub_pulse[i] = 0.5 * (erf(c * bt * (tt + 0.5)) - erf(c * bt * (tt - 0.5)));
        acc += g_sub_pulse[i];
        g_sub_pc[i] = acc;                                   /* inclusive running sum, as np.add.accumulate */
    }
    g_sub_init = 1;
}
/* phase (radians) of output sample m (0 <= m < 151680) of symbols_to_complex_audio */
static double sub_phase(const uint8_t* tones, const double* cum /*[80]: sum of the tones before symbol i*/, double f_base, int m) {
    const int n = m + SUB_SPS;                               /* index in the 81-slot array */
    const double dphi_peak = 2.0 * M_PI / SUB_SPS;
    int ih = n / SUB_SPS; if (ih > 78) ih = 78;
    int il = ih - 2; if (il < 0) il = 0;
    double acc = cum[il] * g_sub_pc[3 * SUB_SPS - 1];        /* symbols whose pulse is fully behind n */
    for (int i = il; i <= ih; i++) {
        int j = n - SUB_SPS * i; if (j > 3 * SUB_SPS - 1) j = 3 * SUB_SPS - 1;
        acc += (double)tones[i] * g_sub_pc[j];
    }
    double phi = dphi_peak * acc + 2.0 * M_PI * f_base * (double)n / 12000.0;
    if (n < 2 * SUB_SPS) phi += dphi_peak * g_sub_pulse[SUB_SPS + n] * (double)tones[0];
    if (n >= 79 * SUB_SPS) phi += dphi_peak * g_sub_pulse[n - 79 * SUB_SPS] * (double)tones[78];
    return phi;
}
int ft8o_subtract(float* audio, const uint8_t* tones, double fHz, double tsec) {
    sub_init();
    const int s0 = (int)(12000.0 * tsec);
    if (!(s0 > 0 && s0 + SUB_L <= FT8O_NSAMP)) return 0;
    const double f_base = fHz - 0.5;
    double cum[80]; cum[0] = 0.0;
    for (int i = 0; i < 79; i++) cum[i + 1] = cum[i] + (double)tones[i];
    double* sr = (double*)malloc(sizeof(double) * SUB_L * 2);
    double* si = sr + SUB_L;
    double Ar[20], Ai[20];
    for (int k = 0; k < 20; k++) { Ar[k] = 0.0; Ai[k] = 0.0; }
    for (int m = 0; m < SUB_L; m++) {
        const double phi = fmod(sub_phase(tones, cum, f_base, m), 2.0 * M_PI);
        double amp = 1.0;
        if (m < 240) amp = (1.0 - cos(M_PI * (double)m / 239.0)) / 2.0;
        else if (m >= SUB_L - 240) amp = (1.0 + cos(M_PI * (double)(m - (SUB_L - 240)) / 239.0)) / 2.0;
        sr[m] = amp * cos(phi); si[m] = amp * sin(phi);
        /* y = x * conj(sig), rounded to complex64 as the reference stores it */
        const double x = (double)audio[s0 + m];
        const double yr = (double)(float)(x * sr[m]), yi = (double)(float)(-x * si[m]);
        for (int k = 0; k < 20; k++) {
            const double ang = -2.0 * M_PI * (double)(((long long)k * m) % 192000) / 192000.0;
            const double c = cos(ang), s = sin(ang);
            Ar[k] += yr * c - yi * s; Ai[k] += yr * s + yi * c;
        }
    }
    for (int m = 0; m < SUB_L; m++) {
        double ar = 0.0, ai = 0.0;
        for (int k = 0; k < 20; k++) {
            const double ang = 2.0 * M_PI * (double)(((long long)k * m) % 192000) / 192000.0;
            const double c = cos(ang), s = sin(ang);
            ar += Ar[k] * c - Ai[k] * s; ai += Ar[k] * s + Ai[k] * c;
        }
        ar = (double)(float)(ar / 192000.0); ai = (double)(float)(ai / 192000.0);          /* ifft output is complex64 */
        audio[s0 + m] = (float)((double)audio[s0 + m] - 2.0 * (ar * sr[m] - ai * si[m]));
    }
    free(sr);
    return 1;
}

/* ------------------------------------------------------------------------------------------------------------------
 * 77-bit word -> the 79 transmitted tones (PyFT8/transmitter.py:181-223 `encode_bits77`): CRC-14 appended, the 83 parity bits of
 * the LDPC(174,91) code from the parity rows FT8_A, three codeword bits per symbol MSB first through the Gray map, Costas blocks
 * at symbols 0, 36, 72.  Pinned by the tone sequences of tests/golden/messages.json (the reference transmitter's own output). */
void ft8o_encode_tones(uint64_t lo, uint64_t hi, uint8_t* tones) {
    static const uint8_t gray[8] = {0, 1, 3, 2, 5, 6, 4, 7};
    hi &= 0x1FFFull;
    const unsigned crc = crc14_of77(lo, hi);
    uint8_t cw[174];
    uint64_t m[2] = {0, 0};                                           /* message bit r (0 = first transmitted) at word r >> 6, bit r & 63 */
    for (int r = 0; r < 91; r++) {
        unsigned b;
        if (r < 77) { const int pos = 76 - r; b = (unsigned)((pos >= 64 ? (hi >> (pos - 64)) : (lo >> pos)) & 1u); }
        else b = (crc >> (13 - (r - 77))) & 1u;
        cw[r] = (uint8_t)b;
        if (b) m[r >> 6] |= 1ull << (r & 63);
    }
    for (int p = 0; p < 83; p++)
        cw[91 + p] = (uint8_t)(__builtin_parityll(FT8_A[p][0] & m[0]) ^ __builtin_parityll(FT8_A[p][1] & m[1]));
    for (int k = 0; k < 7; k++) { tones[k] = (uint8_t)COSTAS[k]; tones[36 + k] = (uint8_t)COSTAS[k]; tones[72 + k] = (uint8_t)COSTAS[k]; }
    for (int i = 0; i < 58; i++)
        tones[PAYLOAD_SYM[i]] = gray[(cw[3 * i] << 2) | (cw[3 * i + 1] << 1) | cw[3 * i + 2]];
}

/* ------------------------------------------------------------------------------------------------------------------
 * Candidate.refine_time_origin of the subtraction experiment (reference tests/pipeline/receiver_sub.py:58-72), which runs on every
 * decode before it is subtracted (:434-436):
 *   fb_0 = int(0.5 + fHz * 16);  tb_0 = int(0.5 + tsec * 200)
 *   for fb in range(fb_0 - 2, fb_0 + 2):  for tb in range(tb_0 - 6, tb_0 + 6):      (the fb loop passes fb_0 every time: four identical sweeps)
 *       _get_signal_grid_fine(fb_0, tb);  if score > best: best = score; tsec = tb / 200; fHz = fb_0 / 16
 * with the EXPERIMENT's _get_signal_grid_fine (:186-211): the 1000-bin slice without edge tapers, 3200-point inverse FFT, 79 symbol
 * DFTs at clip(tb + 32 s, 0, 3168), score = max over the three Costas blocks of the 7x7 correlation; spectrum = rfft of the float32
 * ring zero-padded to 192000 (:273-276), i.e. of the residual after the subtractions made so far.
 * Arithmetic: the build's contract (FFT plans, symbol DFT, fp64 score sums) -- bit-exact against the GPU's refine = 3; against the
 * reference itself the scores agree to ~1e-5 relative and the chosen origins are compared in tests/test_oracle_golden.py. */
void ft8o_refine_time_origin(const float* audio_f32, const ft8o_config* c, double* fHz, double* tsec, float* best_score) {
    float* spec = (float*)malloc(sizeof(float) * 2 * FT8O_SPEC_BINS);
    cycle_spectrum_f32(audio_f32, c, spec);
    const int fb0 = (int)(0.5 + *fHz * 16.0);
    const int tb0 = (int)(0.5 + *tsec * 200.0);
    cpx* z = (cpx*)malloc(sizeof(cpx) * 3200);
    fine_zsig_t(spec, c, fb0, z, 0);
    float best = 0.0f; int have = 0, tbest = tb0;
    for (int tb = tb0 - 6; tb < tb0 + 6; tb++) {
        float s = fine_score_block(z, tb, 0);
        const float s1 = fine_score_block(z, tb, 36), s2 = fine_score_block(z, tb, 72);
        if (s1 > s) s = s1;
        if (s2 > s) s = s2;
        if (!have || s > best) { best = s; tbest = tb; have = 1; }              /* first strict maximum (the reference starts at -1e40) */
    }
    *tsec = (double)tbest / 200.0;
    *fHz = (double)fb0 / 16.0;
    if (best_score) *best_score = best;
    free(z); free(spec);
}

/* ------------------------------------------------------------------------------------------------------------------
 * The build's own origin re-estimation and subtraction on a decimated baseband copy (ft8rx_subtract refine = 2; EXTENSION, no
 * reference counterpart): a plain double-precision statement of what pyft8_amd/csrc/kernels/subtract.hpp computes in float32 with
 * hardware sin/cos -- tolerance stage (tests compare the chosen grid points and the residual, not bits).
 *   fc = (fHz - 0.5) + 21.875;  s00 = int(12000 tsec)
 *   z[m]   = sum_{|j| < 32} (32 - |j|) x[n_c + j] e^{-2 pi i fc (n_c + j) / 12000},  n_c = s00 + 32 (m - 64),  m < 4864      (mix + triangular decimator)
 *   c[m]   = conj(sig[32 m] e^{-2 pi i fc (s0 + 32 m) / 12000}) / (1024 g(tone of sample 32 m)),  m < 4736                    (model for the current origin)
 *   scan:   Y[z][ch] = sum_{m in chunk ch (37 samples)} z[64 + off + m + shift_z / 32] c[m],  off = (s0 - s00) / 32
 *   pick:   argmax over (shift, df) of |sum_ch Y[z][ch] e^{-2 pi i df t_ch}|^2, t_ch = (ch + 1/2) 1184 / 12000; then
 *           s0 += shift, tsec = (s0 + 1/2) / 12000, fHz += df - 1/2
 *   coarse: shifts 32 (-52 + 4 i), i < 15; df = -1 + j / 16, j < 113;   fine: shifts 32 (-4 + i), i < 9; df = 0.4375 + j / 64, j < 9
 *   amplitude: A_k = (32 / 192000) sum_m z[64 + off + m] c[m] e^{-2 pi i k 32 m / 192000}, k < 20;  a[m] = sum_k A_k e^{+2 pi i k 32 m / 192000}
 *   subtract (full rate): x[s0 + n] -= 2 Re(a(n) sig[n]), a(n) interpolated linearly between a[n >> 5] and a[(n >> 5) + 1] */
#define SUBD_D 32
#define SUBD_CH 37
#define SUBD_N (128 * SUBD_CH)
#define SUBD_PAD 64
#define SUBD_NZ (SUBD_PAD + SUBD_N + 64)
typedef struct { double re, im; } dcpx;
static double subd_droop(double f) {
    const double x = M_PI * f / 12000.0;
    if (fabs(x) < 1e-6) return 1.0;
    const double r = sin(32.0 * x) / (32.0 * sin(x));
    return r * r;
}
static void sub_sig(const uint8_t* tones, const double* cum, double f_base, int m, double* sr, double* si) {
    const double phi = fmod(sub_phase(tones, cum, f_base, m), 2.0 * M_PI);
    double amp = 1.0;
    if (m < 240) amp = (1.0 - cos(M_PI * (double)m / 239.0)) / 2.0;
    else if (m >= SUB_L - 240) amp = (1.0 + cos(M_PI * (double)(m - (SUB_L - 240)) / 239.0)) / 2.0;
    *sr = amp * cos(phi); *si = amp * sin(phi);
}
static void subd_model(const uint8_t* tones, const double* cum, double fHz, double tsec, double fc, dcpx* cm) {
    const int s0 = (int)(12000.0 * tsec);
    for (int m = 0; m < SUBD_N; m++) {
        const int n = SUBD_D * m;
        double sr, si; sub_sig(tones, cum, fHz - 0.5, n, &sr, &si);
        const double th = 2.0 * M_PI * fmod(fc * ((double)(s0 + n) / 12000.0), 1.0);
        const double cr = cos(th), ci = sin(th);
        int isym = n / SUB_SPS; if (isym > 78) isym = 78;
        const double g = subd_droop(fHz - 0.5 + 6.25 * (double)tones[isym] - fc);
        const double sc = 1.0 / (1024.0 * g);
        cm[m].re = (sr * cr + si * ci) * sc; cm[m].im = (sr * ci - si * cr) * sc;       /* conj(sig e^{-i th}) */
    }
}
static void subd_scan_pick(const dcpx* zd, const dcpx* cm, int s00, const int* shift, int nshift, double df_lo, double df_step, int ndf,
                           double* fHz, double* tsec) {
    const int s0 = (int)(12000.0 * *tsec);
    const int off = (s0 - s00) / SUBD_D;
    dcpx (*Y)[128] = (dcpx (*)[128])calloc((size_t)nshift * 128, sizeof(dcpx));
    for (int z = 0; z < nshift; z++) {
        const int b0 = s0 + shift[z];
        if (!(b0 > 0 && b0 + SUB_L <= FT8O_NSAMP)) continue;
        for (int ch = 0; ch < 128; ch++) {
            double ar = 0.0, ai = 0.0;
            for (int m = 0; m < SUBD_CH; m++) {
                const int idx = SUBD_PAD + off + ch * SUBD_CH + m + shift[z] / SUBD_D;
                if (idx < 0 || idx >= SUBD_NZ) continue;
                const dcpx zv = zd[idx], wv = cm[ch * SUBD_CH + m];
                ar += zv.re * wv.re - zv.im * wv.im; ai += zv.re * wv.im + zv.im * wv.re;
            }
            Y[z][ch].re = ar; Y[z][ch].im = ai;
        }
    }
    double be = -1.0; int bz = 0, bj = 0;
    for (int z = 0; z < nshift; z++) for (int j = 0; j < ndf; j++) {
        const double df = df_lo + df_step * (double)j;
        double er = 0.0, ei = 0.0;
        for (int ch = 0; ch < 128; ch++) {
            const double ang = -2.0 * M_PI * df * (((double)ch + 0.5) * (double)(SUBD_D * SUBD_CH) / 12000.0);
            const double cr = cos(ang), ci = sin(ang);
            er += Y[z][ch].re * cr - Y[z][ch].im * ci; ei += Y[z][ch].re * ci + Y[z][ch].im * cr;
        }
        const double e = er * er + ei * ei;
        if (e > be) { be = e; bz = z; bj = j; }
    }
    if (be > 0.0) {
        const int s0n = s0 + shift[bz];
        *tsec = ((double)s0n + 0.5) / 12000.0;
        *fHz += (df_lo + df_step * (double)bj) - 0.5;
    }
    free(Y);
}
/* refine = 2 for one signal: re-estimates (fHz, tsec) in place and, if `subtract`, removes the signal from audio (float32, in place).
 * returns 1 if the signal was subtracted */
int ft8o_refine2_subtract(float* audio, const uint8_t* tones, double* fHz, double* tsec, int subtract) {
    sub_init();
    double cum[80]; cum[0] = 0.0;
    for (int i = 0; i < 79; i++) cum[i + 1] = cum[i] + (double)tones[i];
    const double fc = *fHz - 0.5 + 21.875;
    const int s00 = (int)(12000.0 * *tsec);
    /* mixed-down samples over the span the decimator touches, then the triangular filter */
    const int n_lo = s00 + SUBD_D * (0 - SUBD_PAD) - 31, n_hi = s00 + SUBD_D * (SUBD_NZ - 1 - SUBD_PAD) + 31;
    const int span = n_hi - n_lo + 1;
    dcpx* xb = (dcpx*)malloc(sizeof(dcpx) * (size_t)span);
    for (int i = 0; i < span; i++) {
        const int n = n_lo + i;
        if (n < 0 || n >= FT8O_NSAMP) { xb[i].re = 0.0; xb[i].im = 0.0; continue; }
        const double th = -2.0 * M_PI * fmod(fc * ((double)n / 12000.0), 1.0);
        xb[i].re = (double)audio[n] * cos(th); xb[i].im = (double)audio[n] * sin(th);
    }
    dcpx* zd = (dcpx*)malloc(sizeof(dcpx) * SUBD_NZ);
    for (int m = 0; m < SUBD_NZ; m++) {
        const int ic = (s00 + SUBD_D * (m - SUBD_PAD)) - n_lo;
        double ar = 0.0, ai = 0.0;
        for (int j = -31; j <= 31; j++) { const double w = (double)(32 - (j < 0 ? -j : j)); ar += w * xb[ic + j].re; ai += w * xb[ic + j].im; }
        zd[m].re = ar; zd[m].im = ai;
    }
    free(xb);
    dcpx* cm = (dcpx*)malloc(sizeof(dcpx) * SUBD_N);
    int coarse[15], fine[9];
    for (int i = 0; i < 15; i++) coarse[i] = SUBD_D * (-52 + 4 * i);
    for (int i = 0; i < 9; i++) fine[i] = SUBD_D * (-4 + i);
    subd_model(tones, cum, *fHz, *tsec, fc, cm);
    subd_scan_pick(zd, cm, s00, coarse, 15, -1.0, 0.0625, 113, fHz, tsec);
    subd_model(tones, cum, *fHz, *tsec, fc, cm);
    subd_scan_pick(zd, cm, s00, fine, 9, 0.4375, 0.015625, 9, fHz, tsec);
    int done = 0;
    const int s0 = (int)(12000.0 * *tsec);
    if (subtract && s0 > 0 && s0 + SUB_L <= FT8O_NSAMP) {
        subd_model(tones, cum, *fHz, *tsec, fc, cm);
        const int off = (s0 - s00) / SUBD_D;
        double Ar[20], Ai[20];
        for (int k = 0; k < 20; k++) { Ar[k] = 0.0; Ai[k] = 0.0; }
        for (int m = 0; m < SUBD_N; m++) {
            const dcpx zv = zd[SUBD_PAD + off + m], wv = cm[m];
            const double yr = zv.re * wv.re - zv.im * wv.im, yi = zv.re * wv.im + zv.im * wv.re;
            for (int k = 0; k < 20; k++) {
                const double ang = -2.0 * M_PI * (double)(((long long)k * SUBD_D * m) % 192000) / 192000.0;
                const double cr = cos(ang), ci = sin(ang);
                Ar[k] += yr * cr - yi * ci; Ai[k] += yr * ci + yi * cr;
            }
        }
        for (int k = 0; k < 20; k++) { Ar[k] *= (double)SUBD_D / 192000.0; Ai[k] *= (double)SUBD_D / 192000.0; }
        dcpx* ad = (dcpx*)malloc(sizeof(dcpx) * (SUBD_N + 1));
        for (int m = 0; m <= SUBD_N; m++) {
            double er = 0.0, ei = 0.0;
            for (int k = 0; k < 20; k++) {
                const double ang = 2.0 * M_PI * (double)(((long long)k * SUBD_D * m) % 192000) / 192000.0;
                const double cr = cos(ang), ci = sin(ang);
                er += Ar[k] * cr - Ai[k] * ci; ei += Ar[k] * ci + Ai[k] * cr;
            }
            ad[m].re = er; ad[m].im = ei;
        }
        for (int n = 0; n < SUB_L; n++) {
            double sr, si; sub_sig(tones, cum, *fHz - 0.5, n, &sr, &si);
            int md = n >> 5; if (md > SUBD_N - 1) md = SUBD_N - 1;
            double fq = (double)(n - SUBD_D * md) / 32.0; if (fq > 1.0) fq = 1.0;
            const double er = ad[md].re + fq * (ad[md + 1].re - ad[md].re), ei = ad[md].im + fq * (ad[md + 1].im - ad[md].im);
            audio[s0 + n] = (float)((double)audio[s0 + n] - 2.0 * (er * sr - ei * si));
        }
        free(ad);
        done = 1;
    }
    free(cm); free(zd);
    return done;
}
