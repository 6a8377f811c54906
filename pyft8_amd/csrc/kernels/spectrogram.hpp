// spectrogram.hpp -- K1: 375-hop dB spectrogram (receiver.py:288-306)
// Part of libft8rx.so.  k_spectrogram / k_hop_spectrum are built in the second translation unit only (ft8rx_ilp.hip, ILP scheduling:
// FT8RX_ILP_UNIT) and launched through ilp_launch.hpp; k_fill_row0 belongs to the main unit.
#ifndef FT8RX_SPECTROGRAM_HPP
#define FT8RX_SPECTROGRAM_HPP

// ------------------------------------------------------------------------------------ K1 spectrogram
// one hop: window samples a[base .. base+3840) (zeros before the frame start) -> 976 dB values.
// 128 threads; 1920-point complex FFT (plan [8,4,4,5,3]) in place in one LDS image as three register-fused stages:
//   [8]    240 butterflies straight from global memory (int16 -> f32, Hann window fused in),
//   [4,4]  120 groups of 16 (one per thread), twiddles from the LDS table w240[t] = W1920[8 t],
//   [5,3]  128 groups of 15 (one per thread), compile-time twiddles,
// then the real-FFT split and 20 log10|.|.
#define SPEC_NT 128
#ifdef FT8RX_ILP_UNIT
FT8_DEV void spectrogram_hop(const int16_t* __restrict__ a, int base, float* __restrict__ out, const Tables& T,
                             cpx* z, cpx* w240, int tid) {
    const cpx* __restrict__ W = T.W1920;
    // stage-2 twiddle table: requested now, stored to LDS just before the first barrier (a load -> wait -> store loop here delayed
    // the whole block by a memory round trip before its first sample load)
    const cpx wa = W[8 * tid], wb = W[8 * (tid + SPEC_NT < 240 ? tid + SPEC_NT : 0)];
    {   // pass [8]: n = 1920, s = 1, m = 240: butterfly p reads samples m = p + 240 j.  Straight-line: loads from clamped addresses,
        // zeros by select (hops before the frame start), twiddle multiplies unconditional (W^0 = (1, -0) is an exact identity).
        cpx v[2][8];
        cpx w[2][8];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int p = tid + SPEC_NT * i;
            const int pc = (p < 240) ? p : 239;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int m = pc + 240 * j, i0 = base + 2 * m;
                // samples before the frame start are zeros (the ring starts zeroed, receiver.py:248): masked with integer ALU ops --
                // a select here is turned back into a branch around the load, i.e. one memory round trip per basic block
                uint32_t raw = *reinterpret_cast<const uint32_t*>(a + (i0 & ~(i0 >> 31)));
                raw &= ~(uint32_t)(i0 >> 31);
                const float2 wn = *reinterpret_cast<const float2*>(T.win + 2 * m);
                v[i][j] = make_float2((float)(int16_t)(raw & 0xFFFFu) * wn.x, (float)(int16_t)(raw >> 16) * wn.y);
            }
#pragma unroll
            for (int j = 1; j < 8; j++) w[i][j] = W[j * pc];
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int p = tid + SPEC_NT * i;
            dft<8>(v[i]);
#pragma unroll
            for (int j = 1; j < 8; j++) v[i][j] = cmul(v[i][j], w[i][j]);
            if (p < 240) {
#pragma unroll
                for (int j = 0; j < 8; j++) z[8 * p + j] = v[i][j];
            }
        }
    }
    w240[tid] = wa;
    if (tid + SPEC_NT < 240) w240[tid + SPEC_NT] = wb;
    __syncthreads();
    {   // passes [4,4]: n = 240, s = 8; group g = (pp = g / 8, q = g % 8): in q + 8(pp + 15 j' + 60 j), out q + 8 j + 32(4 pp + j')
        typedef Fused2<1920, 240, 8, 4, 4> F;
        cpx v[4][4];
        const bool on = tid < F::groups;
        if (on) F::load_affine<120, 480>(z, tid, v);
        __syncthreads();
        if (on) {
            const int pp = tid >> 3;
#pragma unroll
            for (int jp = 0; jp < 4; jp++) {
                dft<4>(v[jp]);
                const int pq = pp + 15 * jp;
#pragma unroll
                for (int j = 1; j < 4; j++) v[jp][j] = cmul(v[jp][j], w240[j * pq]);          // W1920[j p 8]
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                cpx u[4];
#pragma unroll
                for (int jp = 0; jp < 4; jp++) u[jp] = v[jp][j];
                dft<4>(u);
#pragma unroll
                for (int jp = 1; jp < 4; jp++) u[jp] = cmul(u[jp], w240[4 * jp * pp]);        // W1920[j' pp 32]
#pragma unroll
                for (int jp = 0; jp < 4; jp++) v[jp][j] = u[jp];
            }
            F::store_affine<32, 8>(z, (tid & 7) + 128 * (tid >> 3), v);
        }
        __syncthreads();
    }
    {   // passes [5,3]: n = 15, s = 128; group q: in q + 128 (j' + 3 j), out q + 128 j + 640 j'
        typedef Fused2<1920, 15, 128, 5, 3> F;
        cpx v[3][5];
        F::load_affine<128, 384>(z, tid, v);
        __syncthreads();
        F::compute_pp(0, v, W);
        F::store_affine<640, 128>(z, tid, v);
        __syncthreads();
    }
    // real-FFT split + dB: the split twiddles of the thread's eight bins are requested together, ahead of the sqrt/log10 chain
    // (with the load inside the loop every iteration waited for its own twiddle: 498 -> 467 us)
    constexpr int NE = (FT8RX_GRID_COLS + SPEC_NT - 1) / SPEC_NT;
    cpx we[NE];
#pragma unroll
    for (int q = 0; q < NE; q++) { const int k = tid + SPEC_NT * q; we[q] = T.WR3840[k < FT8RX_GRID_COLS ? k : 0]; }
#pragma unroll
    for (int q = 0; q < NE; q++) {
        const int k = tid + SPEC_NT * q;
        if (k >= FT8RX_GRID_COLS) break;
        cpx p = z[k], q2 = z[(1920 - k) % 1920];
        float er = 0.5f * (p.x + q2.x), ei = 0.5f * (p.y - q2.y);
        float orr = 0.5f * (p.y + q2.y), oi = 0.5f * (q2.x - p.x);
        cpx w = we[q];
        float xr = er + (w.x * orr - w.y * oi);
        float xi = ei + (w.x * oi + w.y * orr);
        // 20 log10(|X| + 1e-12) (receiver.py:292) as 10 log10(max(|X|^2, 1e-24)) -- the contract's form: the same value to far below a
        // float ulp for any |X| > 1e-6, exactly -240 dB for |X| = 0, and no IEEE square root (a dozen instructions) per bin
        float pw = xr * xr + xi * xi;
        pw = (pw > 1e-24f) ? pw : 1e-24f;
        out[k] = 10.0f * ft8_log10f_normal(pw);                  // 1e-24 <= argument <= 4e15: the general function's range checks are dead here
    }
}

__global__ __launch_bounds__(SPEC_NT) void k_spectrogram(const int16_t* __restrict__ audio, float* __restrict__ grid, Tables T) {
    __shared__ cpx z[1920];
    __shared__ cpx w240[240];
    // XCD-aware hop mapping: workgroup id -> XCD is id % 8 and gridDim.x = 376 = 8 * 47, so the 47 workgroups of a
    // frame that land on one XCD take 47 consecutive hops: each XCD's L2 then sees one eighth of the frame's audio
    // (8x overlapping windows) instead of all of it.
    static_assert(FT8RX_GRID_ROWS == 8 * 47, "the hop map below assumes gridDim.x = 376 = 8 XCDs x 47");
    const int hop = (blockIdx.x & 7) * 47 + (blockIdx.x >> 3) + 1, f = blockIdx.y, tid = threadIdx.x;
    if (hop > 375) {           // the one spare workgroup of a frame: the never-written row 0 (receiver.py:240), which the cycle FFT's scratch overwrites
        float* row0 = grid + (size_t)f * FT8RX_GRID_ROWS * FT8RX_GRID_COLS;
        for (int i = tid; i < FT8RX_GRID_COLS; i += SPEC_NT) row0[i] = 1.0f;
        return;
    }
    spectrogram_hop(audio + (size_t)f * FT8RX_NSAMP, 480 * hop - 3840,
                    grid + ((size_t)f * FT8RX_GRID_ROWS + hop) * FT8RX_GRID_COLS, T, z, w240, tid);
}

// streaming mode: one hop of the live receiver (AudioIn.get_hop_spectrum, receiver.py:288-293)
__global__ __launch_bounds__(SPEC_NT) void k_hop_spectrum(const int16_t* __restrict__ win3840, float* __restrict__ row, Tables T) {
    __shared__ cpx z[1920];
    __shared__ cpx w240[240];
    spectrogram_hop(win3840, 0, row, T, z, w240, threadIdx.x);
}

#else
__global__ void k_fill_row0(float* grid, int B) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * FT8RX_GRID_COLS) grid[(size_t)(i / FT8RX_GRID_COLS) * FT8RX_GRID_ROWS * FT8RX_GRID_COLS + (i % FT8RX_GRID_COLS)] = 1.0f;
}
#endif  // FT8RX_ILP_UNIT

#endif
