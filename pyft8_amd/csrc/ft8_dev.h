// ft8_dev.h -- device-side building blocks for the gfx950 FT8 receive kernels.
//
// Arithmetic contract: plain IEEE-754 fp32/fp64, no FMA *contraction* (-ffp-contract=off: an fma happens only where the contract
// names one -- the twiddle multiply cmul below), correctly rounded divide/sqrt (hipcc default), operation order written out explicitly.  DESIGN.md lists the
// contract; the CPU oracle states the same formulas independently, which is what makes the parity
// tests bit-exact.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ft8_tables.h"

typedef float2 cpx;

#define FT8_DEV __device__ __forceinline__

// ------------------------------------------------------------------------------------ elementary math
// a / b, correctly rounded, for operands that need none of the range repairs of the compiler's division: b finite and normal with
// 2^-32 <= |b| <= 2^32, a = 0 or normal with |a| in the same range (or either one NaN).  It IS the compiler's sequence
// (v_rcp_f32, two Newton steps on the reciprocal, the quotient and two residual corrections) without the v_div_scale pair that rescales
// denormal / huge operands, with a plain fma where v_div_fmas would apply that scale, and without v_div_fixup (infinities, zero
// divisors): for these operands all three are identities, so the value is the IEEE quotient the CPU oracle gets from `/` -- 8
// instructions instead of 11 and no VCC hazard.
FT8_DEV float ft8_div_inrange(float a, float b) {
    float r = __builtin_amdgcn_rcpf(b);
    float e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = a * r;
    e = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(e, r, q);
}
FT8_DEV float ft8_log10f(float x) {
    if (!(x > 0.0f)) return (x == 0.0f) ? -__builtin_inff() : __builtin_nanf("");
    if (x > 3.0e38f) return __builtin_inff();
    uint32_t ix = __float_as_uint(x);
    int e = 0;
    if (ix < 0x00800000u) { x = x * 8388608.0f; ix = __float_as_uint(x); e = -23; }
    e += (int)(ix >> 23) - 127;
    float m = __uint_as_float((ix & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float s = ft8_div_inrange(m - 1.0f, m + 1.0f);        // |m - 1| = 0 or in [2^-24, 0.42], m + 1 in [1.7, 2.42]
    float s2 = s * s;
    float p = 0.11111111f;                              // contract: Horner steps and the final combination as named fmas
    p = __builtin_fmaf(p, s2, 0.14285715f);
    p = __builtin_fmaf(p, s2, 0.2f);
    p = __builtin_fmaf(p, s2, 0.33333334f);
    p = __builtin_fmaf(p, s2, 1.0f);
    float lnm = (2.0f * s) * p;
    float fe = (float)e;
    return __builtin_fmaf(fe, 0.301025390625f, __builtin_fmaf(fe, 4.6050390e-6f, lnm * 0.4342945f));
}

// The same function for arguments known to be positive, normal and finite (2^-126 <= x <= 3e38): the three range checks above are
// dead code for them and cost two divergent branches per call.  The spectrogram's argument is |X| + 1e-12 with |X| <= 32768 x 3840.
FT8_DEV float ft8_log10f_normal(float x) {
    const uint32_t ix = __float_as_uint(x);
    int e = (int)(ix >> 23) - 127;
    float m = __uint_as_float((ix & 0x007fffffu) | 0x3f800000u);
    const bool big = m > 1.41421356f;
    m = big ? m * 0.5f : m; e += big ? 1 : 0;
    float s = ft8_div_inrange(m - 1.0f, m + 1.0f);        // |m - 1| = 0 or in [2^-24, 0.42], m + 1 in [1.7, 2.42]
    float s2 = s * s;
    float p = 0.11111111f;                              // contract: Horner steps and the final combination as named fmas
    p = __builtin_fmaf(p, s2, 0.14285715f);
    p = __builtin_fmaf(p, s2, 0.2f);
    p = __builtin_fmaf(p, s2, 0.33333334f);
    p = __builtin_fmaf(p, s2, 1.0f);
    float lnm = (2.0f * s) * p;
    float fe = (float)e;
    return __builtin_fmaf(fe, 0.301025390625f, __builtin_fmaf(fe, 4.6050390e-6f, lnm * 0.4342945f));
}

// single-branch clamped rational (no divergence inside a wavefront): x (P(x^2) / Q(x^2)), one IEEE division.  The quotient is formed
// BEFORE the multiplication by x (contract, round 5): P in [4.9e-3, 0.115], Q in [4.9e-3, 0.91] for every x, so the division never
// needs the range repairs that a numerator P x (tiny for tiny x) would -- see ft8_div_inrange.
FT8_DEV float ft8_tanhf(float x) {
    // clamp = the median of (x, -c, c): one instruction, the sign of a zero survives; a NaN does not (v_med3_f32 returns a number),
    // so the NaN argument is put back at the end (the oracle returns x itself: same NaN-ness) -- three instructions where two
    // compare-and-select pairs took four
    const float xc = __builtin_amdgcn_fmed3f(x, -7.90531111f, 7.90531111f);
    const float x2 = xc * xc;
    float p = -2.76076847742355e-16f;                   // contract: Horner steps as named fmas
    p = __builtin_fmaf(p, x2, 2.00018790482477e-13f);
    p = __builtin_fmaf(p, x2, -8.60467152213735e-11f);
    p = __builtin_fmaf(p, x2, 5.12229709037114e-08f);
    p = __builtin_fmaf(p, x2, 1.48572235717979e-05f);
    p = __builtin_fmaf(p, x2, 6.37261928875436e-04f);
    p = __builtin_fmaf(p, x2, 4.89352455891786e-03f);
    float q = 1.19825839466702e-06f;
    q = __builtin_fmaf(q, x2, 1.18534705686654e-04f);
    q = __builtin_fmaf(q, x2, 2.26843463243900e-03f);
    q = __builtin_fmaf(q, x2, 4.89352518554385e-03f);
    const float t = xc * ft8_div_inrange(p, q);
    return (x != x) ? x : t;
}

// ------------------------------------------------------------------------------------ complex helpers / DFT primitives
// Twiddle multiply of the contract: two roundings per component -- the one product, then the fused multiply-add (C99 fmaf: exactly
// specified, so the CPU oracle reproduces it bit for bit with or without hardware FMA).  6 -> 4 VALU instructions per multiply.
FT8_DEV cpx cmul(cpx a, cpx w) { return make_float2(__builtin_fmaf(a.x, w.x, -(a.y * w.y)), __builtin_fmaf(a.x, w.y, a.y * w.x)); }
FT8_DEV cpx cadd(cpx a, cpx b) { return make_float2(a.x + b.x, a.y + b.y); }
FT8_DEV cpx csub(cpx a, cpx b) { return make_float2(a.x - b.x, a.y - b.y); }

template <int R> FT8_DEV void dft(cpx* a);
template <> FT8_DEV void dft<2>(cpx* a) { cpx t = a[0]; a[0] = cadd(t, a[1]); a[1] = csub(t, a[1]); }
template <> FT8_DEV void dft<3>(cpx* a) {
    cpx t1 = cadd(a[1], a[2]), t2 = csub(a[1], a[2]);
    cpx m = make_float2(__builtin_fmaf(-0.5f, t1.x, a[0].x), __builtin_fmaf(-0.5f, t1.y, a[0].y));       // contract: named fmas
    cpx n = make_float2(0.86602540f * t2.x, 0.86602540f * t2.y);
    a[0] = cadd(a[0], t1);
    a[1] = make_float2(m.x + n.y, m.y - n.x);
    a[2] = make_float2(m.x - n.y, m.y + n.x);
}
template <> FT8_DEV void dft<4>(cpx* a) {
    cpx t0 = cadd(a[0], a[2]), t1 = csub(a[0], a[2]), t2 = cadd(a[1], a[3]), t3 = csub(a[1], a[3]);
    a[0] = cadd(t0, t2); a[2] = csub(t0, t2);
    a[1] = make_float2(t1.x + t3.y, t1.y - t3.x);
    a[3] = make_float2(t1.x - t3.y, t1.y + t3.x);
}
template <> FT8_DEV void dft<5>(cpx* a) {
    const float c1 = 0.30901699f, c2 = -0.80901699f, s1 = 0.95105652f, s2 = 0.58778525f;
    cpx t1 = cadd(a[1], a[4]), t2 = cadd(a[2], a[3]), t3 = csub(a[1], a[4]), t4 = csub(a[2], a[3]);
    // contract: the constant multiplies as named fmas (48 -> 36 instructions per butterfly)
    cpx m1 = make_float2(__builtin_fmaf(c2, t2.x, __builtin_fmaf(c1, t1.x, a[0].x)), __builtin_fmaf(c2, t2.y, __builtin_fmaf(c1, t1.y, a[0].y)));
    cpx m2 = make_float2(__builtin_fmaf(c1, t2.x, __builtin_fmaf(c2, t1.x, a[0].x)), __builtin_fmaf(c1, t2.y, __builtin_fmaf(c2, t1.y, a[0].y)));
    cpx n1 = make_float2(__builtin_fmaf(s1, t3.x, s2 * t4.x), __builtin_fmaf(s1, t3.y, s2 * t4.y));
    cpx n2 = make_float2(__builtin_fmaf(s2, t3.x, -(s1 * t4.x)), __builtin_fmaf(s2, t3.y, -(s1 * t4.y)));
    cpx t5 = cadd(t1, t2);
    a[0] = cadd(a[0], t5);
    a[1] = make_float2(m1.x + n1.y, m1.y - n1.x);
    a[4] = make_float2(m1.x - n1.y, m1.y + n1.x);
    a[2] = make_float2(m2.x + n2.y, m2.y - n2.x);
    a[3] = make_float2(m2.x - n2.y, m2.y + n2.x);
}
template <> FT8_DEV void dft<8>(cpx* a) {
    const float h = 0.70710678f;
    cpx e[4] = {a[0], a[2], a[4], a[6]}, o[4] = {a[1], a[3], a[5], a[7]};
    dft<4>(e); dft<4>(o);
    cpx o1 = make_float2(h * (o[1].x + o[1].y), h * (o[1].y - o[1].x));
    cpx o2 = make_float2(o[2].y, -o[2].x);
    cpx o3 = make_float2(h * (o[3].y - o[3].x), -(h * (o[3].x + o[3].y)));
    a[0] = cadd(e[0], o[0]); a[4] = csub(e[0], o[0]);
    a[1] = cadd(e[1], o1);   a[5] = csub(e[1], o1);
    a[2] = cadd(e[2], o2);   a[6] = csub(e[2], o2);
    a[3] = cadd(e[3], o3);   a[7] = csub(e[3], o3);
}

// ------------------------------------------------------------------------------------ LDS Stockham FFT
// Decimation-in-frequency autosort passes over NSEQ independent length-N sequences laid out
// back to back in LDS.  pass(R): m = n/R; butterfly (p,q): a_j = x[q + s(p + j m)],
// y[q + s(R p + j)] = DFT_R(a)_j * W_N^{j p s}.  Compile-time n, s => index math by constants.
template <int N, int n, int s, int R>
FT8_DEV void fft_pass(const cpx* __restrict__ src, cpx* __restrict__ dst, const cpx* __restrict__ W,
                      int nseq, int tid, int nthreads) {
    constexpr int m = n / R;
    constexpr int nb = N / R;
    const int total = nb * nseq;
    for (int b = tid; b < total; b += nthreads) {
        const int seq = b / nb, bb = b - seq * nb;
        const int p = bb / s, q = bb - p * s;
        const cpx* x = src + seq * N;
        cpx* y = dst + seq * N;
        cpx a[R];
#pragma unroll
        for (int j = 0; j < R; j++) a[j] = x[q + s * (p + j * m)];
        dft<R>(a);
        y[q + s * (R * p)] = a[0];
#pragma unroll
        for (int j = 1; j < R; j++) {
            cpx v = a[j];
            if (m > 1) v = cmul(v, W[j * p * s]);            // unconditional (p = 0: W^0 = (1, -0) is an exact identity): no divergent branch around the twiddle load
            y[q + s * (R * p + j)] = v;
        }
    }
}

template <int N, int n, int s, int... Rs> struct FftPasses;
template <int N, int n, int s> struct FftPasses<N, n, s> {
    static FT8_DEV cpx* run(cpx* src, cpx*, const cpx*, int, int, int) { return src; }
};
template <int N, int n, int s, int R, int... Rest> struct FftPasses<N, n, s, R, Rest...> {
    static FT8_DEV cpx* run(cpx* src, cpx* dst, const cpx* W, int nseq, int tid, int nthreads) {
        fft_pass<N, n, s, R>(src, dst, W, nseq, tid, nthreads);
        __syncthreads();
        return FftPasses<N, n / R, s * R, Rest...>::run(dst, src, W, nseq, tid, nthreads);
    }
};
// Runs the whole plan; returns the buffer that holds the result (data must be visible, i.e. the
// caller has synchronised after filling `a`).
template <int N, int... Rs>
FT8_DEV cpx* lds_fft(cpx* a, cpx* b, const cpx* W, int nseq, int tid, int nthreads) {
    return FftPasses<N, N, 1, Rs...>::run(a, b, W, nseq, tid, nthreads);
}

// ------------------------------------------------------------------------------------ register-fused, in-place stages
// Same arithmetic as the pass-by-pass Stockham schedule above (identical butterflies, identical twiddle
// multiplies and skip rules), but one thread keeps a whole group of R1*R2 points in registers across two
// consecutive passes, so the data makes one LDS round trip per *stage* instead of one per pass, and the
// transform runs in place in a single LDS buffer (load all groups -> barrier -> store all groups).
//   pass A (R1; n, s; m1 = n/R1) followed by pass B (R2; n/R1, s*R1; m2 = m1/R2), group (p', q):
//   in : x[q + s (p' + j' m2 + j m1)]      out: z[q + s j + s R1 (R2 p' + j')]      j < R1, j' < R2
template <int N, int n, int s, int R1, int R2>
struct Fused2 {
    static constexpr int m1 = n / R1, m2 = m1 / R2, groups = N / (R1 * R2);
    // Affine addressing: element (j', j) of the group lives at base + j'*SJP + j*SJ, with compile-time strides, so
    // every LDS access is one base register plus an immediate offset.  The caller picks base/strides to realise
    // whatever physical layout it wants for the stage's input and output images.
    // pass A only (R2 butterflies of radix R1 + twiddles); pass B is done by the caller (output pruning)
    static FT8_DEV void compute_passA(int pp, cpx (&a)[R2][R1], const cpx* __restrict__ W) {
#pragma unroll
        for (int jp = 0; jp < R2; jp++) {
            dft<R1>(a[jp]);
            const int p = pp + jp * m2;
            if (p != 0) {
#pragma unroll
                for (int j = 1; j < R1; j++) a[jp][j] = cmul(a[jp][j], W[j * p * s]);
            }
        }
    }
    template <int SJP, int SJ>
    static FT8_DEV void load_affine(const cpx* __restrict__ x, int base, cpx (&a)[R2][R1]) {
#pragma unroll
        for (int jp = 0; jp < R2; jp++)
#pragma unroll
            for (int j = 0; j < R1; j++) a[jp][j] = x[base + jp * SJP + j * SJ];
    }
    template <int SJP, int SJ>
    static FT8_DEV void store_affine(cpx* __restrict__ z, int base, const cpx (&a)[R2][R1]) {
#pragma unroll
        for (int jp = 0; jp < R2; jp++)
#pragma unroll
            for (int j = 0; j < R1; j++) z[base + jp * SJP + j * SJ] = a[jp][j];
    }
    static FT8_DEV void compute(int g, cpx (&a)[R2][R1], const cpx* __restrict__ W) { compute_pp(g / s, a, W); }
    static FT8_DEV void compute_pp(int pp, cpx (&a)[R2][R1], const cpx* __restrict__ W) {
#pragma unroll
        for (int jp = 0; jp < R2; jp++) {
            dft<R1>(a[jp]);
            const int p = pp + jp * m2;
            if (p != 0) {
#pragma unroll
                for (int j = 1; j < R1; j++) a[jp][j] = cmul(a[jp][j], W[j * p * s]);
            }
        }
#pragma unroll
        for (int j = 0; j < R1; j++) {
            cpx b[R2];
#pragma unroll
            for (int jp = 0; jp < R2; jp++) b[jp] = a[jp][j];
            dft<R2>(b);
            if (m2 > 1 && pp != 0) {
#pragma unroll
                for (int jp = 1; jp < R2; jp++) b[jp] = cmul(b[jp], W[jp * pp * (s * R1)]);
            }
#pragma unroll
            for (int jp = 0; jp < R2; jp++) a[jp][j] = b[jp];
        }
    }
};

// 8 tones of one 32-sample symbol, shared by 4 lanes (n2 = lane & 3):  32 = 4 x 8 decimation in time.
//   u[k] = DFT8_k(x[4 n1 + n2]) on lane n2;  X[k] = ((u0 + u1 W^k) + u2 W^2k) + u3 W^3k on the quad leader.
// `x` holds this lane's 8 samples; returns |X[k]| in mag[0..7] (valid on lanes with n2 == 0).
// The quad sum uses DPP quad_perm broadcasts (lane o of the quad as the second operand of a plain v_add): the same three adds in the
// same order as a shuffle-based reduction, without the LDS crossbar round trips of ds_bpermute.
template <int O> FT8_DEV float quad_lane(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x55 * O, 0xf, 0xf, false));      // quad_perm:[O,O,O,O]
}
// NT = number of tones wanted (7: a Costas score never reads tone 7, receiver.py:203); wq[k] = W32^(n2 k), this lane's twiddles,
// loaded once per thread (sym32_twiddles).
FT8_DEV void sym32_twiddles(const cpx* __restrict__ w32, int n2, cpx* wq) {
#pragma unroll
    for (int k = 1; k < 8; k++) wq[k] = w32[(n2 * k) & 31];
    wq[0] = make_float2(1.0f, -0.0f);
}
template <int NT>
FT8_DEV void sym32_quad(cpx* x, int n2, const cpx* wq, float* mag) {
    dft<8>(x);
#pragma unroll
    for (int k = 0; k < NT; k++) {
        cpx t = x[k];
        if (k != 0) t = cmul(t, wq[k]);                        // the quad leader's factor is W^0 = (1, -0): an exact identity, no select needed
        cpx acc = t;                                   // quad leader: its own term is u0
        acc = cadd(acc, make_float2(quad_lane<1>(t.x), quad_lane<1>(t.y)));
        acc = cadd(acc, make_float2(quad_lane<2>(t.x), quad_lane<2>(t.y)));
        acc = cadd(acc, make_float2(quad_lane<3>(t.x), quad_lane<3>(t.y)));
        mag[k] = sqrtf(acc.x * acc.x + acc.y * acc.y);
    }
}

// Everything below reads tables that ft8rx_create fills at run time in the MAIN translation unit (hipMemcpyToSymbol); the second
// unit (ft8rx_ilp.hip: the FFT kernels, FT8RX_ILP_UNIT) would get zero-filled copies of its own, so it does not see them at all.
#ifndef FT8RX_ILP_UNIT
// ------------------------------------------------------------------------------------ CRC-14 and message validity
// CRC-14 (poly 0x2757, zero init, 77 message bits followed by 19 zero bits; reference decoders.py:123-129) is
// linear over GF(2): crc(m) = XOR of the syndromes of the set bits.  d_CRC_SYN[pos] = crc of the message with
// only bit `pos` (0 = least significant of the 77-bit integer) set; filled by the host at create time with the
// bit-serial definition.
__device__ uint16_t d_CRC_SYN[77];
FT8_DEV unsigned ft8_crc14(uint64_t lo, uint64_t hi) {
    unsigned r = 0;
#pragma unroll 8
    for (int pos = 0; pos < 64; pos++) r ^= ((lo >> pos) & 1ull) ? (unsigned)d_CRC_SYN[pos] : 0u;
#pragma unroll
    for (int pos = 0; pos < 13; pos++) r ^= ((hi >> pos) & 1ull) ? (unsigned)d_CRC_SYN[64 + pos] : 0u;
    return r;
}
// 28-bit standard callsign plausibility (reference decoders.py:95-115); first = first character after strip
FT8_DEV bool ft8_std_call_ok(uint32_t c28, char* first) {
    int64_t nn = (int64_t)c28 - (2063592 + 4194304);
    int ch[6];   // character codes: 0 = space, 1..10 digits, 11..36 letters
    if (nn < 0) { ch[0] = 36; ch[1] = 36; ch[2] = 10; ch[3] = 36; ch[4] = 36; ch[5] = 36; }   // "ZZ9ZZZ" quirk
    else {
        int i0 = (int)(nn / 7085880); nn %= 7085880;
        int i1 = (int)(nn / 196830);  nn %= 196830;
        int i2 = (int)(nn / 19683);   nn %= 19683;
        int i3 = (int)(nn / 729);     nn %= 729;
        int i4 = (int)(nn / 27);      int i5 = (int)(nn % 27);
        ch[0] = i0;                               // ' ' + digits + letters
        ch[1] = i1 + 1;                           // digits + letters
        ch[2] = (i2 < 10) ? i2 + 1 : 0;           // digits + 17 spaces
        ch[3] = i3 ? i3 + 10 : 0; ch[4] = i4 ? i4 + 10 : 0; ch[5] = i5 ? i5 + 10 : 0;   // ' ' + letters
    }
    int a = 0, b = 6;
    while (a < b && ch[a] == 0) a++;
    while (b > a && ch[b - 1] == 0) b--;
    int L = b - a;
    if (L < 3) return false;
    for (int i = a; i < b; i++) if (ch[i] == 0) return false;
    int c0 = ch[a], c1 = ch[a + 1], c2 = ch[a + 2];
    *first = (c0 >= 11) ? (char)('A' + c0 - 11) : (char)('0' + c0 - 1);
    bool d1 = (c1 >= 1 && c1 <= 10), d2 = (c2 >= 1 && c2 <= 10);
    if (c0 >= 11 && ((FT8_PFX1_MASK >> (c0 - 11)) & 1u) && d1)
        if (!(((FT8_PFX1_TRAP >> (c0 - 11)) & 1u) && d2)) return true;
    if (((FT8_PFX2[c0 - 1] >> (c1 - 1)) & 1ULL) && d2) return true;
    return false;
}

FT8_DEV bool ft8_call29_ok(uint32_t c29, int i3) {                            // reference decoders.py:70-93
    uint32_t pr = c29 & 1u, c28 = c29 >> 1;
    if (c28 < 2063592u + 4194303u) return true;          // tokens, CQ nnn, CQ xxxx, hashed calls: always a string
    char first;
    if (!ft8_std_call_ok(c28, &first)) return false;
    if (pr && i3 != 2 && !(first == 'A' || first == 'K' || first == 'N' || first == 'W')) return false;   // '/R' rule
    return true;
}

// unpack() returns a tuple?  (reference decoders.py:16-68)
FT8_DEV bool ft8_valid77(uint64_t lo, uint64_t hi) {
    if (lo == 0 && hi == 0) return false;
    unsigned i3 = (unsigned)(lo & 7u);
    if (i3 == 1 || i3 == 2) {
        uint32_t g16 = (uint32_t)((lo >> 3) & 0xFFFFu);
        uint32_t cb29 = (uint32_t)((lo >> 19) & 0x1FFFFFFFu);
        uint32_t ca29 = (uint32_t)(((lo >> 48) | (hi << 16)) & 0x1FFFFFFFu);
        uint32_t g15 = g16 & 0x7FFFu;
        if (g15 == 0) return false;
        if (g15 == 32400 || g15 == 32401) return false;   // '' grid => tuple contains '' (decoders.py:62,67)
        return ft8_call29_ok(ca29, (int)i3) && ft8_call29_ok(cb29, (int)i3);
    }
    if (i3 == 4) {
        unsigned cq = (unsigned)((lo >> 3) & 1u), rrr = (unsigned)((lo >> 4) & 3u);
        return !((cq && rrr) || (!cq && !rrr));
    }
    return false;
}

// 91 hard bits given as two ballots (b0: codeword bits 0..63, b1: bits 64..90, LSB = lowest index)
// -> 77-bit message (bit 76 = codeword bit 0) and the received CRC field
FT8_DEV void ft8_cw_to_msg(uint64_t b0, uint64_t b1, uint64_t* lo, uint64_t* hi, unsigned* crc) {
    uint64_t r0 = __brevll(b0), r1 = __brevll(b1);
    *hi = r0 >> 51;
    *lo = (r0 << 13) | (r1 >> 51);
    *crc = (unsigned)((r1 >> 37) & 0x3FFFu);
}

// CRC-14 is linear: crc14(message) ^ received field, over the 91-bit word given as codeword bits (b0: bits 0..63, b1: bits 64..90),
// is the XOR of one table entry per byte; the CRC matches iff this syndrome is zero.  d_CRC_T[b][x] = syndrome of byte b (codeword
// bits 8b .. 8b+7, bits >= 91 ignored) holding x; filled by the host at create time from the bit-serial definition (decoders.py:123-129).
__device__ uint16_t d_CRC_T[12][256];
FT8_DEV unsigned ft8_crc_syndrome(uint64_t b0, uint64_t b1) {
    unsigned s = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) s ^= d_CRC_T[b][(b0 >> (8 * b)) & 0xFF];
#pragma unroll
    for (int b = 0; b < 4; b++) s ^= d_CRC_T[8 + b][(b1 >> (8 * b)) & 0xFF];
    return s;
}

// Wave-cooperative forms for wave-uniform words (call with all 64 lanes active).  A uniform-index lookup in a 16-bit table compiles
// to a broadcast *vector* load (gfx950 has no sub-dword scalar loads): 12 of them per word kept the texture-address unit busy
// longer than everything else the callers do.  Here lane b of every 16-lane row fetches byte b's entry -- one load instruction --
// and the rows are XOR-reduced with DPP (quad_perm, quad_perm, row_half_mirror, row_mirror).
FT8_DEV unsigned ft8_xor_row16(unsigned t) {
    t ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0xB1, 0xf, 0xf, false);      // quad_perm:[1,0,3,2]
    t ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x4E, 0xf, 0xf, false);      // quad_perm:[2,3,0,1]
    t ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x141, 0xf, 0xf, false);     // row_half_mirror (quads hold equal values)
    t ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x140, 0xf, 0xf, false);     // row_mirror (halves hold equal values)
    return t;
}
// this lane's table entry for the word (b0, b1): byte (lane & 15) of the 12; 0 for lanes 12..15 of a row
FT8_DEV unsigned ft8_crc_entry(uint64_t b0, uint64_t b1, int lane) {
    const int b = lane & 15;
    const uint64_t w = (b < 8) ? b0 : b1;
    const unsigned byte = (unsigned)(w >> (8 * (b & 7))) & 0xFFu;
    const unsigned t = d_CRC_T[b < 12 ? b : 0][byte];                  // always a valid address; masked below (no branch around the load)
    return t & (unsigned)((b - 12) >> 31);
}
FT8_DEV unsigned ft8_crc_syndrome_wave(uint64_t b0, uint64_t b1, int lane) {
    return (unsigned)__builtin_amdgcn_readfirstlane((int)ft8_xor_row16(ft8_crc_entry(b0, b1 & ((1ull << 27) - 1), lane)));
}

// 0 = no CRC match (or all-zero message), 1 = CRC ok but unpack() -> None, 2 = accepted
FT8_DEV int ft8_crc_check(uint64_t b0, uint64_t b1, uint64_t* lo, uint64_t* hi) {
    if (ft8_crc_syndrome(b0, b1 & ((1ull << 27) - 1)) != 0) return 0;        // 12 table lookups instead of a 77-step bit loop
    unsigned crc;
    ft8_cw_to_msg(b0, b1, lo, hi, &crc);
    if (*lo == 0 && *hi == 0) return 0;
    return ft8_valid77(*lo, *hi) ? 2 : 1;
}
// the same for a wave-uniform word, all 64 lanes active
FT8_DEV int ft8_crc_check_wave(uint64_t b0, uint64_t b1, int lane, uint64_t* lo, uint64_t* hi) {
    if (ft8_crc_syndrome_wave(b0, b1, lane) != 0) return 0;
    unsigned crc;
    ft8_cw_to_msg(b0, b1, lo, hi, &crc);
    if (*lo == 0 && *hi == 0) return 0;
    return ft8_valid77(*lo, *hi) ? 2 : 1;
}
#endif  // FT8RX_ILP_UNIT
