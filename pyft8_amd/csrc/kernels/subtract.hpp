// subtract.hpp -- signal subtraction (reference tests/pipeline/receiver_sub.py:380-402, transmitter.py:41-70)
// Part of libft8rx.so; included by ft8rx.hip (single translation unit: the kernels share __constant__/__device__ tables).
#ifndef FT8RX_SUBTRACT_HPP
#define FT8RX_SUBTRACT_HPP

// ------------------------------------------------------------------------------------ signal subtraction
// Receiver.subtract_signal of the reference's subtraction experiment, per decoded signal:
//   sig = symbols_to_complex_audio(tones, fHz - 0.5);  y = x * conj(sig);  A = the first 20 bins of the 192000-point FFT of y
//   (a 0 .. 1.19 Hz one-sided low-pass: the slowly varying complex amplitude);  a = ifft(A);  x -= 2 Re(a sig).
// The frames' audio lives in a float32 working copy (the reference's ring buffer is float32); signals of one frame are
// subtracted in list order (a later signal sees the residual of the earlier ones), signal s of all frames at once:
//   k_sub_accum  (chunk, frame): partial sums of the 20 DFT bins over 1185 samples  -> part[frame][chunk][20]
//   k_sub_apply  (chunk, frame): bins = fixed-order sum of the 128 partials; subtracts the chunk in place
// Arithmetic: GFSK phase in fp64 (it reaches 1e5 rad), everything else fp32 with fp64 accumulators; results agree with the
// oracle / reference to ~1e-6 of the signal amplitude (tolerance stage, not bit-exact).
#define SUB_L (79 * 1920)
#define SUB_NCH 128
#define SUB_CH (SUB_L / SUB_NCH)          // 1185 samples per chunk

struct SubTables { const double* pulse; const double* pc; };     // [5760] GFSK pulse and its inclusive running sum

// phase of sample m of symbols_to_complex_audio (transmitter.py:52-66), reduced to [0, 2 pi), and the edge ramp (:67-70)
FT8_DEV void sub_signal(const ft8rx_subsig& S, const double* __restrict__ cum /*LDS [80]*/, const SubTables& T, int m, float* sr, float* si) {
    const int n = m + 1920;
    const double dphi_peak = 6.283185307179586 / 1920.0;
    int ih = n / 1920; if (ih > 78) ih = 78;
    const int il = ih - 2 > 0 ? ih - 2 : 0;
    double acc = cum[il] * T.pc[5759];
    for (int i = il; i <= ih; i++) {
        int j = n - 1920 * i; if (j > 5759) j = 5759;
        acc += (double)S.tones[i] * T.pc[j];
    }
    double phi = dphi_peak * acc + 6.283185307179586 * (S.fHz - 0.5) * (double)n / 12000.0;
    if (n < 3840) phi += dphi_peak * T.pulse[1920 + n] * (double)S.tones[0];
    if (n >= 79 * 1920) phi += dphi_peak * T.pulse[n - 79 * 1920] * (double)S.tones[78];
    const double rev = phi * (1.0 / 6.283185307179586);           // revolutions; the hardware sin / cos take the fraction directly
    const float fr = (float)(rev - floor(rev));
    const float s = __builtin_amdgcn_sinf(fr), c = __builtin_amdgcn_cosf(fr);
    float amp = 1.0f;
    if (m < 240) amp = 0.5f * (1.0f - cosf(3.14159265f * (float)m / 239.0f));
    else if (m >= SUB_L - 240) amp = 0.5f * (1.0f + cosf(3.14159265f * (float)(m - (SUB_L - 240)) / 239.0f));
    *sr = amp * c; *si = amp * s;
}

// Work decomposition of the sample kernels: a block is 4 wavefronts; a wavefront owns SUB_CPW chunks and walks each with its 64
// lanes (19 samples per lane), so per-chunk sums need only wave shuffles; the per-signal setup is paid once per 16 chunks.
#define SUB_CPW 4
#define SUB_GRIDX (SUB_NCH / (4 * SUB_CPW))

FT8_DEV bool sub_setup(const ft8rx_subsig* __restrict__ sigs, const int32_t* __restrict__ counts, int max_sigs, int s, int frame,
                       ft8rx_subsig* S, double* cum, int* s0) {
    if (s >= counts[frame]) return false;
    if (threadIdx.x < 24) reinterpret_cast<uint32_t*>(S)[threadIdx.x] = reinterpret_cast<const uint32_t*>(sigs + (size_t)frame * max_sigs + s)[threadIdx.x];
    __syncthreads();
    if (threadIdx.x < 80) { int a = 0; for (int i = 0; i < (int)threadIdx.x; i++) a += S->tones[i]; cum[threadIdx.x] = (double)a; }   // tones before symbol i
    __syncthreads();
    *s0 = (int)(12000.0 * S->tsec);
    return *s0 > 0 && *s0 + SUB_L <= FT8RX_NSAMP;                 // the reference's guard (receiver_sub.py:390) + "fits the buffer"
}

__global__ __launch_bounds__(256) void k_sub_accum(const float* __restrict__ wf, const ft8rx_subsig* __restrict__ sigs,
                                                   const int32_t* __restrict__ counts, int max_sigs, int s, SubTables T,
                                                   double2* __restrict__ part /*[B][SUB_NCH][20]*/) {
    __shared__ ft8rx_subsig S;
    __shared__ double cum[80];
    const int frame = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int s0;
    if (!sub_setup(sigs, counts, max_sigs, s, frame, &S, cum, &s0)) return;
    const float* x = wf + (size_t)frame * FT8RX_NSAMP + s0;
    for (int i = 0; i < SUB_CPW; i++) {
        const int ch = (blockIdx.x * 4 + wave) * SUB_CPW + i;
        float fr_[20], fi_[20];                                       // 19 terms per lane in fp32, the cross-lane sums in fp64
#pragma unroll
        for (int k = 0; k < 20; k++) { fr_[k] = 0.0f; fi_[k] = 0.0f; }
        for (int m = ch * SUB_CH + lane; m < (ch + 1) * SUB_CH; m += 64) {
            float sr, si;
            sub_signal(S, cum, T, m, &sr, &si);
            const float xv = x[m];
            const float yr = xv * sr, yi = -(xv * si);              // y = x conj(sig)  (complex64 in the reference)
            const float rv = (float)m * (1.0f / 192000.0f);        // e^{-2 pi i m / 192000}
            const float wr = __builtin_amdgcn_cosf(rv), wi = -__builtin_amdgcn_sinf(rv);
            float rr = 1.0f, ri = 0.0f;
#pragma unroll
            for (int k = 0; k < 20; k++) {
                fr_[k] += yr * rr - yi * ri; fi_[k] += yr * ri + yi * rr;
                const float t = rr * wr - ri * wi; ri = rr * wi + ri * wr; rr = t;
            }
        }
#pragma unroll
        for (int k = 0; k < 20; k++) {                                // fixed-order butterfly over the 64 lanes
            double vr = (double)fr_[k], vi = (double)fi_[k];
            for (int o = 32; o > 0; o >>= 1) { vr += __shfl_xor(vr, o); vi += __shfl_xor(vi, o); }
            if (lane == k) part[((size_t)frame * SUB_NCH + ch) * 20 + k] = make_double2(vr, vi);
        }
    }
}

__global__ __launch_bounds__(256) void k_sub_apply(float* __restrict__ wf, const ft8rx_subsig* __restrict__ sigs,
                                                   const int32_t* __restrict__ counts, int max_sigs, int s, SubTables T,
                                                   const double2* __restrict__ part) {
    __shared__ ft8rx_subsig S;
    __shared__ double cum[80];
    __shared__ float Ar[20], Ai[20];
    const int frame = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int s0;
    if (!sub_setup(sigs, counts, max_sigs, s, frame, &S, cum, &s0)) return;
    if (tid < 20) {
        double vr = 0.0, vi = 0.0;
        for (int c = 0; c < SUB_NCH; c++) { const double2 v = part[((size_t)frame * SUB_NCH + c) * 20 + tid]; vr += v.x; vi += v.y; }
        Ar[tid] = (float)(vr / 192000.0); Ai[tid] = (float)(vi / 192000.0);          // ifft scale
    }
    __syncthreads();
    float* x = wf + (size_t)frame * FT8RX_NSAMP + s0;
    for (int i = 0; i < SUB_CPW; i++) {
        const int ch = (blockIdx.x * 4 + wave) * SUB_CPW + i;
        for (int m = ch * SUB_CH + lane; m < (ch + 1) * SUB_CH; m += 64) {
            float sr, si;
            sub_signal(S, cum, T, m, &sr, &si);
            const float rv = (float)m * (1.0f / 192000.0f);        // e^{+2 pi i m / 192000}
            const float wr = __builtin_amdgcn_cosf(rv), wi = __builtin_amdgcn_sinf(rv);
            float rr = 1.0f, ri = 0.0f, er = 0.0f, ei = 0.0f;
#pragma unroll
            for (int k = 0; k < 20; k++) {
                er += Ar[k] * rr - Ai[k] * ri; ei += Ar[k] * ri + Ai[k] * rr;
                const float t = rr * wr - ri * wi; ri = rr * wi + ri * wr; rr = t;
            }
            x[m] = x[m] - 2.0f * (er * sr - ei * si);
        }
    }
}

// ---- origin refinement (extension; the reference's refine_time_origin, receiver_sub.py:58-72, scans time only, in 5 ms steps, on
// the 7 Costas symbols).  Subtraction only cancels when the model is aligned to a few ms and a fraction of a Hz, so the scan
// uses the same signal model and all 79 known symbols: for a set of start-sample shifts the despread signal y = x conj(sig) is
// summed over the 128 chunks (k_sub_scan); k_sub_pick then evaluates |sum_c Y_c e^{-2 pi i df t_c}|^2 on a frequency grid and
// moves the signal's (tsec, fHz) to the best (shift, df).
#define SUB_MAXSHIFT 16
#define SUB_MAXSPAN 1800
struct SubShifts { int n; int stride; int shift[SUB_MAXSHIFT]; };   // start-sample shifts relative to int(12000 tsec); stride > 1 would
                                                                  // decimate the scan (tried: it aliases neighbouring signals into the sums)

__global__ __launch_bounds__(256) void k_sub_scan(const float* __restrict__ wf, const ft8rx_subsig* __restrict__ sigs,
                                                  const int32_t* __restrict__ counts, int max_sigs, int s, SubTables T, SubShifts sh,
                                                  double2* __restrict__ scan /*[B][SUB_MAXSHIFT][SUB_NCH]*/) {
    __shared__ ft8rx_subsig S;
    __shared__ double cum[80];
    const int frame = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int s0;
    if (s >= counts[frame]) return;
    sub_setup(sigs, counts, max_sigs, s, frame, &S, cum, &s0);
    const float* xf = wf + (size_t)frame * FT8RX_NSAMP;
    __shared__ float xs[4][SUB_CH + SUB_MAXSPAN + 3];                // per wavefront: the audio its chunk sees under every shift
    const int span = sh.shift[sh.n - 1] - sh.shift[0];               // shifts ascend; span <= SUB_MAXSPAN (host)
    for (int i = 0; i < SUB_CPW; i++) {
        const int ch = (blockIdx.x * 4 + wave) * SUB_CPW + i;
        __syncthreads();                                              // the previous chunk's reads are done
        const int g0 = s0 + sh.shift[0] + ch * SUB_CH;
        for (int j = lane; j < SUB_CH + span; j += 64) { const int g = g0 + j; xs[wave][j] = (g >= 0 && g < FT8RX_NSAMP) ? xf[g] : 0.0f; }
        __syncthreads();
        // the model sample sig[m] does not depend on the shift: evaluate it once and correlate it with all shifted windows
        float ar[SUB_MAXSHIFT], ai[SUB_MAXSHIFT];
#pragma unroll
        for (int z = 0; z < SUB_MAXSHIFT; z++) { ar[z] = 0.0f; ai[z] = 0.0f; }
        for (int m = ch * SUB_CH + lane * sh.stride; m < (ch + 1) * SUB_CH; m += 64 * sh.stride) {
            float sr, si;
            sub_signal(S, cum, T, m, &sr, &si);
            const float* xw = xs[wave] + (m - ch * SUB_CH) - sh.shift[0];
#pragma unroll
            for (int z = 0; z < SUB_MAXSHIFT; z++) {
                const int b0 = s0 + sh.shift[z];                     // wave-uniform
                if (z < sh.n && b0 > 0 && b0 + SUB_L <= FT8RX_NSAMP) {
                    const float xv = xw[sh.shift[z]];
                    ar[z] += xv * sr; ai[z] -= xv * si;
                }
            }
        }
#pragma unroll
        for (int z = 0; z < SUB_MAXSHIFT; z++) {
            double vr = (double)ar[z], vi = (double)ai[z];
            for (int o = 32; o > 0; o >>= 1) { vr += __shfl_xor(vr, o); vi += __shfl_xor(vi, o); }
            if (lane == z && z < sh.n) scan[((size_t)frame * SUB_MAXSHIFT + z) * SUB_NCH + ch] = make_double2(vr, vi);
        }
    }
}

// one block per frame: best (shift, df) of signal s; df on [df_lo, df_lo + ndf * df_step)
__global__ __launch_bounds__(256) void k_sub_pick(ft8rx_subsig* __restrict__ sigs, const int32_t* __restrict__ counts, int max_sigs, int s,
                                                  SubShifts sh, const double2* __restrict__ scan, float df_lo, float df_step, int ndf,
                                                  int chunk_samples) {
    __shared__ float best_e[256];
    __shared__ int best_i[256];
    const int frame = blockIdx.x, tid = threadIdx.x;
    if (s >= counts[frame]) return;
    float be = -1.0f; int bi = 0;
    for (int idx = tid; idx < sh.n * ndf; idx += 256) {
        const int z = idx / ndf, j = idx - z * ndf;
        const float df = df_lo + df_step * (float)j;
        const double2* Y = scan + ((size_t)frame * SUB_MAXSHIFT + z) * SUB_NCH;
        float er = 0.0f, ei = 0.0f;
        // e^{-2 pi i df t_c} at the chunk centres t_c = (c + 1/2) CH / 12000: one rotation step per chunk
        float wr, wi, rr, ri;
        sincosf(-6.28318531f * df * ((float)chunk_samples / 12000.0f), &wi, &wr);
        sincosf(-6.28318531f * df * (0.5f * (float)chunk_samples / 12000.0f), &ri, &rr);
        for (int c = 0; c < SUB_NCH; c++) {
            const float yr = (float)Y[c].x, yi = (float)Y[c].y;
            er += yr * rr - yi * ri; ei += yr * ri + yi * rr;
            const float t = rr * wr - ri * wi; ri = rr * wi + ri * wr; rr = t;
        }
        const float e = er * er + ei * ei;
        if (e > be) { be = e; bi = idx; }
    }
    best_e[tid] = be; best_i[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            const float e2 = best_e[tid + o]; const int i2 = best_i[tid + o];
            if (e2 > best_e[tid] || (e2 == best_e[tid] && i2 < best_i[tid])) { best_e[tid] = e2; best_i[tid] = i2; }
        }
        __syncthreads();
    }
    if (tid == 0 && best_e[0] > 0.0f) {
        const int z = best_i[0] / ndf, j = best_i[0] - z * ndf;
        ft8rx_subsig& S = sigs[(size_t)frame * max_sigs + s];
        const int s0 = (int)(12000.0 * S.tsec) + sh.shift[z];
        S.tsec = ((double)s0 + 0.5) / 12000.0;                        // int(12000 tsec) = s0 exactly
        // the model's tone 0 is at fHz - 0.5 (receiver_sub.py:387), so that the signal sits mid-band of the 0 .. 1.19 Hz low-pass:
        // df is the signal's offset from the model, and the new fHz puts it at +0.5 Hz again
        S.fHz += (double)(df_lo + df_step * (float)j) - 0.5;
    }
}

// ---- fast origin refinement + amplitude estimate on a decimated baseband copy (extension, refine = 2) ---------------------------
// The full-rate scans above evaluate the signal model four times per sample and signal.  Everything they estimate lives within
// +-25 Hz of the signal's centre frequency, so: mix the residual audio down by f_c = (fHz - 0.5) + 21.875 Hz (middle of the eight
// tones) and decimate by 32 with a triangular (two-stage boxcar) filter -- nulls at every multiple of 375 Hz, i.e. exactly where
// the decimation folds other signals onto this one; worst-case alias < -45 dB -- ONCE per signal (k_subd_mix); run the time /
// frequency scans and the 20-bin amplitude estimate on the 375 Hz series (4736 samples instead of 151 680: k_subd_scan,
// k_sub_pick, k_subd_accum) against the model sampled at the same instants with the filter's per-tone droop divided out
// (k_subd_model); and go back to full rate only to subtract (k_subd_apply: exact model, the slowly varying amplitude a(t)
// interpolated linearly between its 375 Hz samples -- it is band-limited to 1.2 Hz).  Time shifts are multiples of 32 samples
// (2.67 ms).  Decimated sample m sits at full-rate index s00 + 32 (m - SUBD_PAD), s00 = the decoder's start sample.
#define SUBD_D 32
#define SUBD_CH 37                              /* decimated samples per chunk: 128 chunks x 37 x 32 = 151 552 of the 151 680 model samples */
#define SUBD_N (SUB_NCH * SUBD_CH)              /* 4736 */
#define SUBD_PAD 64                             /* decimated samples kept before s00 (coarse shifts reach -56) */
#define SUBD_NZ (SUBD_PAD + SUBD_N + 64)        /* 4864 = 19 x 256 */
struct SubdCtx { double fc; int s00; int pad; };

// response of the triangular decimator (boxcar 32 applied twice) at offset f Hz from the mixing frequency, relative to DC
FT8_DEV float subd_droop(float f) {
    const float x = 3.14159265f * f / 12000.0f;
    if (fabsf(x) < 1e-6f) return 1.0f;
    const float r = sinf(32.0f * x) / (32.0f * sinf(x));
    return r * r;
}

// block sums of the mixed-down samples of LDS block lb (32 samples): S = sum xb, R = sum k xb
FT8_DEV void subd_block(const float* xs, int nb, int lb, double fc, float* sr_, float* si_, float* rr_, float* ri_) {
    const int n0 = nb + SUBD_D * lb;
    const double rev = fc * ((double)n0 / 12000.0);
    const float fr = (float)(rev - floor(rev));
    float cr = __builtin_amdgcn_cosf(fr), ci = -__builtin_amdgcn_sinf(fr);              // e^{-2 pi i fc n0 / fs}
    const float st = (float)(fc / 12000.0);
    const float wr = __builtin_amdgcn_cosf(st), wi = -__builtin_amdgcn_sinf(st);        // per-sample rotation
    float sr = 0.0f, si = 0.0f, rr = 0.0f, ri = 0.0f;
    const float* b = xs + 33 * lb;
#pragma unroll 8
    for (int k = 0; k < 32; k++) {
        const float v = b[k];
        const float br = v * cr, bi = v * ci;
        sr += br; si += bi; rr += (float)k * br; ri += (float)k * bi;
        const float t = cr * wr - ci * wi; ci = cr * wi + ci * wr; cr = t;
    }
    *sr_ = sr; *si_ = si; *rr_ = rr; *ri_ = ri;
}
// grid (SUBD_NZ / 256, B): z[m] = sum_{j=-31..31} (32 - |j|) x[n_c + j] e^{-2 pi i fc (n_c + j) / 12000},  n_c = s00 + 32 (m - SUBD_PAD)
//   = R[m-1] + 32 S[m] - R[m]  with the block sums S[m] = sum_{k<32} xb[32 m + k], R[m] = sum_k k xb[32 m + k]
__global__ __launch_bounds__(256) void k_subd_mix(const float* __restrict__ wf, const ft8rx_subsig* __restrict__ sigs,
                                                  const int32_t* __restrict__ counts, int max_sigs, int s,
                                                  float2* __restrict__ zdec /*[B][SUBD_NZ]*/, SubdCtx* __restrict__ ctx /*[B]*/) {
    __shared__ float xs[257 * 33];                                       // 257 blocks of 32 samples, one pad word per block (bank-conflict free)
    __shared__ float2 Rs[257];
    const int frame = blockIdx.y, tid = threadIdx.x;
    if (s >= counts[frame]) return;
    const ft8rx_subsig& S = sigs[(size_t)frame * max_sigs + s];
    const double fc = S.fHz - 0.5 + 21.875;
    const int s00 = (int)(12000.0 * S.tsec);
    if (blockIdx.x == 0 && tid == 0) { SubdCtx c; c.fc = fc; c.s00 = s00; c.pad = 0; ctx[frame] = c; }
    const int m0 = 256 * (int)blockIdx.x;                                // first decimated sample of this block
    const int nb = s00 + SUBD_D * (m0 - 1 - SUBD_PAD);                   // full-rate index of LDS block 0 (= decimated block m0 - 1)
    const float* x = wf + (size_t)frame * FT8RX_NSAMP;
    for (int i = tid; i < 257 * 32; i += 256) { const int g = nb + i; xs[i + (i >> 5)] = (g >= 0 && g < FT8RX_NSAMP) ? x[g] : 0.0f; }
    __syncthreads();
    float sr, si, rr, ri;
    if (tid == 0) { subd_block(xs, nb, 0, fc, &sr, &si, &rr, &ri); Rs[0] = make_float2(rr, ri); }      // the halo block: only its R is needed
    subd_block(xs, nb, tid + 1, fc, &sr, &si, &rr, &ri);
    Rs[tid + 1] = make_float2(rr, ri);
    __syncthreads();
    const float2 rp = Rs[tid];
    zdec[(size_t)frame * SUBD_NZ + m0 + tid] = make_float2(rp.x + 32.0f * sr - rr, rp.y + 32.0f * si - ri);
}

// grid (ceil(SUBD_N / 256), B): c[m] = conj(model_bb[m]) / (1024 g) -- the correlation weight against z -- for the signal's CURRENT
// origin (fHz, tsec as refined so far): model_bb[m] = sig[32 m] e^{-2 pi i fc (s0 + 32 m) / fs}, g = the decimator's droop at the current tone
__global__ __launch_bounds__(256) void k_subd_model(const ft8rx_subsig* __restrict__ sigs, const int32_t* __restrict__ counts, int max_sigs,
                                                    int s, SubTables T, const SubdCtx* __restrict__ ctx, float2* __restrict__ model /*[B][SUBD_N]*/) {
    __shared__ ft8rx_subsig S;
    __shared__ double cum[80];
    const int frame = blockIdx.y, tid = threadIdx.x;
    int s0;
    if (s >= counts[frame]) return;
    sub_setup(sigs, counts, max_sigs, s, frame, &S, cum, &s0);
    const int m = 256 * (int)blockIdx.x + tid;
    if (m >= SUBD_N) return;
    const int n = SUBD_D * m;
    float sr, si;
    sub_signal(S, cum, T, n, &sr, &si);
    const double fc = ctx[frame].fc;
    const double rev = fc * ((double)(s0 + n) / 12000.0);              // k_subd_mix mixes with the ABSOLUTE sample index: so must the model,
    const float fr = (float)(rev - floor(rev));                        // or the amplitude estimate comes out rotated by a constant phase
    const float cr = __builtin_amdgcn_cosf(fr), ci = __builtin_amdgcn_sinf(fr);
    // conj(sig e^{-i th}) = conj(sig) e^{+i th}
    const float mr = sr * cr + si * ci, mi = sr * ci - si * cr;
    int isym = n / 1920; if (isym > 78) isym = 78;                     // the symbol sample n lies in (r02 used the NEXT symbol's tone here: ~1 % amplitude error at the edge tones)
    const float g = subd_droop((float)(S.fHz - 0.5 + 6.25 * (double)S.tones[isym] - fc));
    const float sc = 1.0f / (1024.0f * g);
    model[(size_t)frame * SUBD_N + m] = make_float2(mr * sc, mi * sc);
}

// one block per frame: Y[z][c] = sum_{m in chunk c} zdec[SUBD_PAD + off + m + shift_z / 32] c[m], off = (int(12000 tsec) - s00) / 32.
// Same output layout as k_sub_scan (-> k_sub_pick).  Thread = (chunk, half of the shifts).
__global__ __launch_bounds__(256) void k_subd_scan(const float2* __restrict__ zdec, const float2* __restrict__ model,
                                                   const ft8rx_subsig* __restrict__ sigs, const int32_t* __restrict__ counts, int max_sigs,
                                                   int s, const SubdCtx* __restrict__ ctx, SubShifts sh, double2* __restrict__ scan) {
    const int frame = blockIdx.x, tid = threadIdx.x;
    if (s >= counts[frame]) return;
    const ft8rx_subsig& S = sigs[(size_t)frame * max_sigs + s];
    const int s0 = (int)(12000.0 * S.tsec);
    const int off = (s0 - ctx[frame].s00) / SUBD_D;
    const int c = tid & 127, zh = tid >> 7;
    const float2* z = zdec + (size_t)frame * SUBD_NZ + SUBD_PAD + off + c * SUBD_CH;
    const float2* w = model + (size_t)frame * SUBD_N + c * SUBD_CH;
    float ar[SUB_MAXSHIFT / 2], ai[SUB_MAXSHIFT / 2];
#pragma unroll
    for (int q = 0; q < SUB_MAXSHIFT / 2; q++) { ar[q] = 0.0f; ai[q] = 0.0f; }
    for (int m = 0; m < SUBD_CH; m++) {
        const float2 wv = w[m];
#pragma unroll
        for (int q = 0; q < SUB_MAXSHIFT / 2; q++) {
            const int zi = 2 * q + zh;
            const int b0 = s0 + sh.shift[zi < sh.n ? zi : 0];
            if (zi < sh.n && b0 > 0 && b0 + SUB_L <= FT8RX_NSAMP) {
                const int idx = m + sh.shift[zi] / SUBD_D;
                const int lo = -(SUBD_PAD + off + c * SUBD_CH), hi = SUBD_NZ + lo;       // stay inside this frame's zdec row
                const float2 zv = (idx >= lo && idx < hi) ? z[idx] : make_float2(0.0f, 0.0f);
                ar[q] += zv.x * wv.x - zv.y * wv.y; ai[q] += zv.x * wv.y + zv.y * wv.x;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < SUB_MAXSHIFT / 2; q++) {
        const int zi = 2 * q + zh;
        if (zi < sh.n) scan[((size_t)frame * SUB_MAXSHIFT + zi) * SUB_NCH + c] = make_double2((double)ar[q], (double)ai[q]);
    }
}

// one block per frame: the 20 low bins of y[m] = zdec[..] c[m] (the complex amplitude), then a(t) on the 375 Hz grid:
// adec[m] = (32 / 192000) sum_k A_k e^{+2 pi i k 32 m / 192000},  m = 0 .. SUBD_N  (what Receiver.subtract_signal's ifft of the 20
// kept bins gives at sample 32 m)
__global__ __launch_bounds__(256) void k_subd_accum(const float2* __restrict__ zdec, const float2* __restrict__ model,
                                                    const ft8rx_subsig* __restrict__ sigs, const int32_t* __restrict__ counts, int max_sigs,
                                                    int s, const SubdCtx* __restrict__ ctx, float2* __restrict__ adec /*[B][SUBD_N + 1]*/) {
    __shared__ double pr[4][20], pi_[4][20];
    __shared__ float Ar[20], Ai[20];
    const int frame = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (s >= counts[frame]) return;
    const ft8rx_subsig& S = sigs[(size_t)frame * max_sigs + s];
    const int s0 = (int)(12000.0 * S.tsec);
    const bool ok = s0 > 0 && s0 + SUB_L <= FT8RX_NSAMP;
    const int off = (s0 - ctx[frame].s00) / SUBD_D;
    const float2* z = zdec + (size_t)frame * SUBD_NZ + SUBD_PAD + off;
    const float2* w = model + (size_t)frame * SUBD_N;
    float fr_[20], fi_[20];
#pragma unroll
    for (int k = 0; k < 20; k++) { fr_[k] = 0.0f; fi_[k] = 0.0f; }
    for (int m = tid; m < SUBD_N; m += 256) {
        const float2 zv = z[m], wv = w[m];
        const float yr = zv.x * wv.x - zv.y * wv.y, yi = zv.x * wv.y + zv.y * wv.x;
        const float rv = (float)(SUBD_D * m) * (1.0f / 192000.0f);
        const float wr = __builtin_amdgcn_cosf(rv), wi = -__builtin_amdgcn_sinf(rv);
        float rr = 1.0f, ri = 0.0f;
#pragma unroll
        for (int k = 0; k < 20; k++) {
            fr_[k] += yr * rr - yi * ri; fi_[k] += yr * ri + yi * rr;
            const float t = rr * wr - ri * wi; ri = rr * wi + ri * wr; rr = t;
        }
    }
#pragma unroll
    for (int k = 0; k < 20; k++) {
        double vr = (double)fr_[k], vi = (double)fi_[k];
        for (int o = 32; o > 0; o >>= 1) { vr += __shfl_xor(vr, o); vi += __shfl_xor(vi, o); }
        if (lane == 0) { pr[wave][k] = vr; pi_[wave][k] = vi; }
    }
    __syncthreads();
    if (tid < 20) {
        const double sc = ok ? (double)SUBD_D / 192000.0 : 0.0;           // each decimated sample stands for 32; ifft scale; nothing to subtract if out of range
        Ar[tid] = (float)(((pr[0][tid] + pr[1][tid]) + (pr[2][tid] + pr[3][tid])) * sc);
        Ai[tid] = (float)(((pi_[0][tid] + pi_[1][tid]) + (pi_[2][tid] + pi_[3][tid])) * sc);
    }
    __syncthreads();
    for (int m = tid; m <= SUBD_N; m += 256) {
        const float rv = (float)(SUBD_D * m) * (1.0f / 192000.0f);
        const float wr = __builtin_amdgcn_cosf(rv), wi = __builtin_amdgcn_sinf(rv);
        float rr = 1.0f, ri = 0.0f, er = 0.0f, ei = 0.0f;
#pragma unroll
        for (int k = 0; k < 20; k++) {
            er += Ar[k] * rr - Ai[k] * ri; ei += Ar[k] * ri + Ai[k] * rr;
            const float t = rr * wr - ri * wi; ri = rr * wi + ri * wr; rr = t;
        }
        adec[(size_t)frame * (SUBD_N + 1) + m] = make_float2(er, ei);
    }
}

// full rate: x[n] -= 2 Re(a(n) sig[n]) with a(n) interpolated between adec[n / 32] and adec[n / 32 + 1] (beyond the last decimated
// sample -- the final 128 of the 151 680 -- the last value is held)
__global__ __launch_bounds__(256) void k_subd_apply(float* __restrict__ wf, const ft8rx_subsig* __restrict__ sigs,
                                                    const int32_t* __restrict__ counts, int max_sigs, int s, SubTables T,
                                                    const float2* __restrict__ adec) {
    __shared__ ft8rx_subsig S;
    __shared__ double cum[80];
    const int frame = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int s0;
    if (!sub_setup(sigs, counts, max_sigs, s, frame, &S, cum, &s0)) return;
    float* x = wf + (size_t)frame * FT8RX_NSAMP + s0;
    const float2* a = adec + (size_t)frame * (SUBD_N + 1);
    for (int i = 0; i < SUB_CPW; i++) {
        const int ch = (blockIdx.x * 4 + wave) * SUB_CPW + i;
        for (int m = ch * SUB_CH + lane; m < (ch + 1) * SUB_CH; m += 64) {
            float sr, si;
            sub_signal(S, cum, T, m, &sr, &si);
            int md = m >> 5; if (md > SUBD_N - 1) md = SUBD_N - 1;
            const float2 a0 = a[md], a1 = a[md + 1];
            float fq = (float)(m - SUBD_D * md) * (1.0f / 32.0f); if (fq > 1.0f) fq = 1.0f;
            const float er = a0.x + fq * (a1.x - a0.x), ei = a0.y + fq * (a1.y - a0.y);
            x[m] = x[m] - 2.0f * (er * sr - ei * si);
        }
    }
}

// ---- refine = 3: Candidate.refine_time_origin of the reference's subtraction experiment (tests/pipeline/receiver_sub.py:58-72) ----
// Before a decode is subtracted the experiment re-estimates its START TIME only: tb scans 12 steps of 5 ms from tb_0 - 6, every step
// scored by the experiment's own _get_signal_grid_fine (:186-211) -- the 1000-bin slice around fb_0 WITHOUT edge tapers, 3200-point
// inverse FFT, symbol DFTs, score = max over the three Costas blocks of the 7x7 correlation -- on the spectrum of the float32 ring,
// i.e. of the residual the earlier subtractions left (:273-276; the fb loop of :64 passes fb_0 every time).  First strict maximum;
// tsec = tb / 200, fHz = fb_0 / 16.  Per signal index s and batch: k_cyc_a_f32 + k_cyc_bc (spectrum of the working copy), then
// k_refine3 (one block per frame).  Same FFT plans, symbol DFT and fp64 score sums as k_fine: bit-exact against ft8o_refine_time_origin.
FT8_DEV void cyc_a_load_f32(const float* __restrict__ a, int n2b, int tid, float2 (&raw)[10]) {
#pragma unroll
    for (int q = 0; q < 10; q++) {
        const int i = tid + 256 * q, ic = i < 2400 ? i : 0;
        const int c = ic & 7, n1 = ic >> 3;
        const int m = 320 * n1 + n2b + c;
        const int inb = (2 * m < FT8RX_NSAMP) ? -1 : 0;                     // zero padding by masking a clamped load (cyc_a_load)
        const float2 v = *reinterpret_cast<const float2*>(a + ((2 * m) & inb));
        raw[q] = make_float2(__uint_as_float(__float_as_uint(v.x) & (uint32_t)inb), __uint_as_float(__float_as_uint(v.y) & (uint32_t)inb));
    }
}
__global__ __launch_bounds__(256) void k_cyc_a_f32(const float* __restrict__ audio, cpx* __restrict__ A, Tables T) {
    __shared__ cpx bufA[8 * 300];
    __shared__ cpx bufB[8 * 300];
    const int tile = (blockIdx.x & 7) * 5 + (blockIdx.x >> 3);
    const int f = blockIdx.y, tid = threadIdx.x, n2b = 8 * tile;
    float2 raw[10];
    cyc_a_load_f32(audio + (size_t)f * FT8RX_NSAMP, n2b, tid, raw);
#pragma unroll
    for (int q = 0; q < 10; q++) { const int i = tid + 256 * q; if (i < 2400) bufA[(i & 7) * 300 + (i >> 3)] = raw[q]; }
    cpx w[10];
#pragma unroll
    for (int q = 0; q < 10; q++) { const int i = tid + 256 * q, ic = i < 2400 ? i : 0; w[q] = T.W96000[(n2b + (ic & 7)) * (ic >> 3)]; }
    __syncthreads();
    cpx* r = lds_fft<300, 5, 5, 4, 3>(bufA, bufB, T.W300, 8, tid, 256);
    cpx* out = A + (size_t)f * 96000;
#pragma unroll
    for (int q = 0; q < 10; q++) {
        const int i = tid + 256 * q;
        if (i < 2400) { const int c = i & 7, k1 = i >> 3; out[k1 * 320 + n2b + c] = cmul(r[c * 300 + k1], w[q]); }
    }
}

// one block of FINE_NT threads per frame; T.taper must point at 100 ones (the experiment's slice has no tapers)
__global__ __launch_bounds__(FINE_NT) void k_refine3(const cpx* __restrict__ spec, ft8rx_subsig* __restrict__ sigs, const int32_t* __restrict__ counts,
                                                     int max_sigs, int s, Tables T) {
    __shared__ cpx z[3200];
    __shared__ cpx slice[FINE_SLICE];
    __shared__ cpx w400[400];
    __shared__ cpx w32[32];
    __shared__ __attribute__((aligned(8))) double dsum[12 * 21 * 2];
    const int frame = blockIdx.x, tid = threadIdx.x;
    if (s >= counts[frame]) return;
    ft8rx_subsig& S = sigs[(size_t)frame * max_sigs + s];
    const int fb0 = (int)(0.5 + S.fHz * 16.0);                       // receiver_sub.py:59
    const int tb0 = (int)(0.5 + S.tsec * 200.0);                     // :60
    if (fb0 - 182 < 0 || fb0 - 182 + FINE_SLICE > FT8RX_SPEC_BINS) return;        // outside the kept spectrum: the origin stays as it is
    const cpx* Sg = spec + (size_t)frame * FT8RX_SPEC_BINS + (fb0 - 182);
    for (int i = tid; i < FINE_SLICE; i += FINE_NT) slice[i] = Sg[i];
    for (int i = tid; i < 400; i += FINE_NT) w400[i] = T.W3200[8 * i];
    if (tid < 32) w32[tid] = T.W32[tid];
    __syncthreads();
    cpx wq[8];
    sym32_twiddles(w32, tid & 3, wq);
    FT_DECL                                                           // (timing-only builds: fine_fft takes the mark table)
    fine_fft(slice, 182, z, w400, T, tid, 0, 3200 FT_PASS);
    // 12 time steps x 21 Costas symbols, four lanes each: (on, off) sums per symbol in fp64, as in k_fine's scoring
#pragma unroll 1
    for (int r = 0; r < (12 * 21 * 4 + FINE_NT - 1) / FINE_NT; r++) {
        const int task = tid + FINE_NT * r, qd = task >> 2, n2 = task & 3;
        const bool valid = qd < 12 * 21;
        const int ti = valid ? qd / 21 : 0, sy = valid ? qd - 21 * ti : 0;
        const int blk = sy / 7, a = sy - 7 * blk;
        float mag[8];
        fine_sym_quad<7>(z, tb0 - 6 + ti + 32 * (36 * blk + a), n2, wq, mag);
        if (valid && n2 == 0) {
            const int c = d_COSTAS[a];
            double off = 0.0, on = 0.0;
#pragma unroll
            for (int b = 0; b < 7; b++) { const double m = (double)mag[b]; on = (b == c) ? m : on; off += (b == c) ? 0.0 : m; }
            dsum[(ti * 21 + sy) * 2] = on; dsum[(ti * 21 + sy) * 2 + 1] = off;
        }
    }
    __syncthreads();
    if (tid == 0) {
        float best = 0.0f; int tbest = tb0 - 6;
        for (int ti = 0; ti < 12; ti++) {
            float sc = 0.0f;
            for (int blk = 0; blk < 3; blk++) {
                double s1 = 0.0, s2 = 0.0;
                for (int a = 0; a < 7; a++) { s1 += dsum[(ti * 21 + 7 * blk + a) * 2]; s2 += dsum[(ti * 21 + 7 * blk + a) * 2 + 1]; }
                const float sb = (float)(s1 + W6 * s2);
                if (blk == 0 || sb > sc) sc = sb;                          // np.max over the three blocks (:210)
            }
            if (ti == 0 || sc > best) { best = sc; tbest = tb0 - 6 + ti; }   // first strict maximum (:66)
        }
        S.tsec = (double)tbest / 200.0;                                   // :68
        S.fHz = (double)fb0 / 16.0;                                       // :69
    }
}

__global__ void k_sub_to_f32(const int16_t* __restrict__ a, float* __restrict__ wf, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) wf[i] = (float)a[i];
}
__global__ void k_sub_to_i16(const float* __restrict__ wf, int16_t* __restrict__ a, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float v = rintf(wf[i]); v = v > 32767.0f ? 32767.0f : (v < -32768.0f ? -32768.0f : v); a[i] = (int16_t)v; }
}

#endif
