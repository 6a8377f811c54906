// synth.hpp -- synthetic frame generator (transmitter.py:41-70)
// Part of libft8rx.so; included by ft8rx.hip (single translation unit: the kernels share __constant__/__device__ tables).
#ifndef FT8RX_SYNTH_HPP
#define FT8RX_SYNTH_HPP

// ------------------------------------------------------------------------------------ synthetic frames
// Device twin of pyft8_amd/synth.py: 79-tone GFSK (BT = 2.0, reference transmitter.py:41-70 as the model) for up to
// 64 signals per frame + unit-variance white noise from a counter-based Philox4x32-10 stream, scaled to sigma = 1000
// counts and clipped to int16.  Workload generator only -- not on the receive path.
struct SynthSig {            // one signal; filled by the host (pyft8_amd/synth.py: device_signal_table)
    double f0;               // Hz
    double cum[82];          // cum[i] = sum_{i'<i} ext[i'] * (Qtot - Qs[i'])   (fully integrated symbols)
    float amp;               // linear amplitude relative to unit-variance noise
    int32_t i0;              // first sample of the 79-symbol waveform inside the frame
    uint8_t ext[84];         // 81 extended tones (first and last repeated), padded
};

FT8_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// one thread = 4 consecutive samples of one frame
__global__ __launch_bounds__(256) void k_synth(int16_t* __restrict__ audio, const SynthSig* __restrict__ sigs, int nsig,
                                               const double* __restrict__ Q /*[5761]*/, uint32_t seed_lo, uint32_t seed_hi, int first_index,
                                               int no_noise) {
    const int f = blockIdx.y;
    const int n0 = 4 * (blockIdx.x * 256 + threadIdx.x);
    if (n0 >= FT8RX_NSAMP) return;
    uint32_t r[4];
    philox4x32_10((uint32_t)(n0 >> 2), (uint32_t)(first_index + f), 0u, 0u, seed_lo, seed_hi, r);
    double x[4];
    {   // Box-Muller: 2 uniform pairs -> 4 normals
        const double u0 = ((double)r[0] + 0.5) * (1.0 / 4294967296.0), u1 = ((double)r[1] + 0.5) * (1.0 / 4294967296.0);
        const double u2 = ((double)r[2] + 0.5) * (1.0 / 4294967296.0), u3 = ((double)r[3] + 0.5) * (1.0 / 4294967296.0);
        const double ra = sqrt(-2.0 * log(u0)), rb = sqrt(-2.0 * log(u2));
        x[0] = ra * cos(6.283185307179586 * u1); x[1] = ra * sin(6.283185307179586 * u1);
        x[2] = rb * cos(6.283185307179586 * u3); x[3] = rb * sin(6.283185307179586 * u3);
    }
    if (no_noise) { x[0] = 0.0; x[1] = 0.0; x[2] = 0.0; x[3] = 0.0; }      // parity tests: the signal part alone
    const SynthSig* S = sigs + (size_t)f * nsig;
    const double Qtot = Q[5760];
    for (int sg = 0; sg < nsig; sg++) {
        const int i0 = S[sg].i0;
        if (n0 + 3 < i0 || n0 >= i0 + 79 * 1920) continue;
        const double f0 = S[sg].f0; const float amp = S[sg].amp;
        for (int k = 0; k < 4; k++) {
            const int m = n0 + k - i0;
            if (m < 0 || m >= 79 * 1920) continue;
            int ih = (m + 3840) / 1920; if (ih > 80) ih = 80;
            const int il = ih - 2 > 0 ? ih - 2 : 0;
            double acc = S[sg].cum[il];
            for (int i = il; i <= ih; i++) {
                int qi = m + 3840 - 1920 * i; if (qi > 5760) qi = 5760;
                const double qs = (i == 0) ? Q[3840] : (i == 1) ? Q[1920] : 0.0;
                acc += (double)S[sg].ext[i] * (Q[qi] - qs);
            }
            (void)Qtot;
            double phi = 6.283185307179586 * (f0 * (double)m + 6.25 * acc) / 12000.0;
            double w = sin(phi);
            if (m < 240) w *= 0.5 * (1.0 - cos(3.141592653589793 * (double)m / 240.0));
            else if (m >= 79 * 1920 - 240) w *= 0.5 * (1.0 - cos(3.141592653589793 * (double)(79 * 1920 - 1 - m) / 240.0));
            x[k] += (double)amp * w;
        }
    }
    short4 o;
    double v;
    v = rint(x[0] * 1000.0); o.x = (short)(v > 32767.0 ? 32767.0 : (v < -32768.0 ? -32768.0 : v));
    v = rint(x[1] * 1000.0); o.y = (short)(v > 32767.0 ? 32767.0 : (v < -32768.0 ? -32768.0 : v));
    v = rint(x[2] * 1000.0); o.z = (short)(v > 32767.0 ? 32767.0 : (v < -32768.0 ? -32768.0 : v));
    v = rint(x[3] * 1000.0); o.w = (short)(v > 32767.0 ? 32767.0 : (v < -32768.0 ? -32768.0 : v));
    *reinterpret_cast<short4*>(audio + (size_t)f * FT8RX_NSAMP + n0) = o;
}

#endif
