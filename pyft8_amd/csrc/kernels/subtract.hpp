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
                                                  SubShifts sh, const double2* __restrict__ scan, float df_lo, float df_step, int ndf) {
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
        sincosf(-6.28318531f * df * ((float)SUB_CH / 12000.0f), &wi, &wr);
        sincosf(-6.28318531f * df * (0.5f * (float)SUB_CH / 12000.0f), &ri, &rr);
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

__global__ void k_sub_to_f32(const int16_t* __restrict__ a, float* __restrict__ wf, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) wf[i] = (float)a[i];
}
__global__ void k_sub_to_i16(const float* __restrict__ wf, int16_t* __restrict__ a, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float v = rintf(wf[i]); v = v > 32767.0f ? 32767.0f : (v < -32768.0f ? -32768.0f : v); a[i] = (int16_t)v; }
}

#endif
