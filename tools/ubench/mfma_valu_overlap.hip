// mfma_valu_overlap.hip -- micro-benchmark for VERDICT r3 item 4: is the matrix pipe free while the vector pipe runs k_fine's kind of work?
//
// k_fine is bound by VALU issue (37 % of the issue ceiling at 2 waves per SIMD, register file full).  fp32 MFMA
// (v_mfma_f32_32x32x2_f32 / 16x16x4_f32) computes at the vector FMA rate (64 FLOP/clk/SIMD) but in its own pipe and is bitwise an fmaf
// chain -- so IF the two pipes run side by side, DFT stages expressed as small matrix products could run beside the butterflies.
// This program measures exactly that, per SIMD, on all 256 CUs (one 512-thread block per CU = 2 waves per SIMD; wave w sits on SIMD w % 4,
// so waves w and w + 4 share one):
//   valu2   both waves of a SIMD issue the plain fp32 mix (independent chains of v_add / v_mul / v_fma, no memory)
//   mfma2   both issue dependency-free f32 MFMAs (four accumulators)
//   valu1 / mfma1   one wave per SIMD does the work, the other exits at once
//   mixed   wave w < 4: VALU work, wave w >= 4: MFMA work -- the experiment
//   inter   every wave alternates: one MFMA, then K VALU instructions (what a restructured k_fine would look like)
// additive <=> t(mixed) ~ max(t(valu1), t(mfma1));  serialised <=> t(mixed) ~ t(valu1) + t(mfma1).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

#define VALU8(a0, a1, a2, a3, a4, a5, a6, a7, c) \
    asm volatile("v_add_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_fma_f32 %2, %2, %8, %8\n v_add_f32 %3, %3, %8\n" \
                 "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_fma_f32 %6, %6, %8, %8\n v_add_f32 %7, %7, %8" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c))

// role: 0 = VALU, 1 = MFMA 32x32x2, 2 = MFMA 16x16x4, 3 = interleave (one 32x32x2 MFMA + 8 K VALU), 4 = interleave with 16x16x4, -1 = exit
template <int KV> __device__ __forceinline__ void body(int role, int n_valu8, int n_mfma, float* out) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float c = 1.0000001f;
    f16v acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    f4v q0 = {0}, q1 = {0}, q2 = {0}, q3 = {0};
    const float ma = 1.0f + 1e-7f * threadIdx.x, mb = 0.5f;
    if (role == 0) {
        for (int i = 0; i < n_valu8; i++) VALU8(a0, a1, a2, a3, a4, a5, a6, a7, c);
    } else if (role == 1) {
        for (int i = 0; i < n_mfma; i += 4) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc3, 0, 0, 0);
        }
    } else if (role == 2) {
        for (int i = 0; i < n_mfma; i += 4) {
            q0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, q0, 0, 0, 0);
            q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, q1, 0, 0, 0);
            q2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, q2, 0, 0, 0);
            q3 = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, q3, 0, 0, 0);
        }
    } else if (role == 3) {
        for (int i = 0; i < n_mfma; i += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < KV; k++) VALU8(a0, a1, a2, a3, a4, a5, a6, a7, c);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc1, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < KV; k++) VALU8(a0, a1, a2, a3, a4, a5, a6, a7, c);
        }
    } else if (role == 4) {
        for (int i = 0; i < n_mfma; i += 2) {
            q0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, q0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < KV; k++) VALU8(a0, a1, a2, a3, a4, a5, a6, a7, c);
            q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, q1, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < KV; k++) VALU8(a0, a1, a2, a3, a4, a5, a6, a7, c);
        }
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    for (int i = 0; i < 16; i++) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    for (int i = 0; i < 4; i++) s += q0[i] + q1[i] + q2[i] + q3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KV> __global__ __launch_bounds__(512) void k(int role_lo, int role_hi, int n_valu8, int n_mfma, float* out) {
    const int w = threadIdx.x >> 6;
    const int role = (w < 4) ? role_lo : role_hi;            // waves w and w + 4 share SIMD w % 4
    if (role < 0) return;
    body<KV>(role, n_valu8, n_mfma, out);
}

template <int KV> static float run(int role_lo, int role_hi, int n_valu8, int n_mfma, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KV><<<256, 512>>>(role_lo, role_hi, n_valu8 / 8, n_mfma / 8, d);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(e0);
        k<KV><<<256, 512>>>(role_lo, role_hi, n_valu8, n_mfma, d);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    float* d; hipMalloc(&d, sizeof(float) * 256 * 512);
    const int NV8 = 40000;        // x 8 VALU instructions per wave
    const double nv = 8.0 * NV8;
    for (int big = 1; big >= 0; big--) {
        const int MF = big ? 1 : 2;                    // 32x32x2 (64 cyc/SIMD) or 16x16x4 (32 cyc/SIMD)
        const int NM = big ? 16000 : 32000;            // MFMAs per wave: about the VALU wave's time
        const char* nm = big ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f32_16x16x4_f32";
        const double flop = big ? 4096.0 : 2048.0;
        const float valu2 = run<1>(0, 0, NV8, NM, d), valu1 = run<1>(0, -1, NV8, NM, d);
        const float mfma2 = run<1>(MF, MF, NV8, NM, d), mfma1 = run<1>(MF, -1, NV8, NM, d);
        const float mixed = run<1>(0, MF, NV8, NM, d);
        printf("== %s, per wave: %d VALU instructions (add/mul/fma mix, 8 independent chains) / %d MFMAs (4 accumulators)\n", nm, 8 * NV8, NM);
        printf("valu1 (one VALU wave per SIMD)            %8.3f ms  = %.2f cyc/inst @2.4GHz\n", valu1, valu1 * 2.4e6 / nv);
        printf("valu2 (two VALU waves per SIMD)           %8.3f ms  = %.2f cyc/inst per SIMD\n", valu2, valu2 * 2.4e6 / (2 * nv));
        printf("mfma1 (one MFMA wave per SIMD)            %8.3f ms  = %.1f cyc/MFMA, %.1f TFLOP/s chip\n", mfma1, mfma1 * 2.4e6 / NM, 1024.0 * NM * flop / (mfma1 * 1e-3) / 1e12);
        printf("mfma2 (two MFMA waves per SIMD)           %8.3f ms  = %.1f cyc/MFMA per SIMD, %.1f TFLOP/s chip\n", mfma2, mfma2 * 2.4e6 / (2.0 * NM), 2048.0 * NM * flop / (mfma2 * 1e-3) / 1e12);
        printf("mixed (one VALU wave + one MFMA wave)     %8.3f ms  : max(valu1, mfma1) = %.3f, sum = %.3f  => combined throughput %.2fx of running them one after the other\n",
               mixed, valu1 > mfma1 ? valu1 : mfma1, valu1 + mfma1, (valu1 + mfma1) / mixed);
        // same wave: one MFMA then 8 K VALU instructions, both waves of a SIMD doing it
        {
            const int n = NM / 2;
            const float i1 = run<1>(big ? 3 : 4, big ? 3 : 4, 0, n, d), i2 = run<2>(big ? 3 : 4, big ? 3 : 4, 0, n, d), i4 = run<4>(big ? 3 : 4, big ? 3 : 4, 0, n, d);
            const float m_only = run<1>(MF, MF, 0, n, d);
            printf("inter (2 waves/SIMD, each: MFMA + 8K VALU): K=1 %.3f ms, K=2 %.3f ms, K=4 %.3f ms; the MFMAs alone %.3f ms; the VALU alone would take %.3f / %.3f / %.3f ms\n",
                   i1, i2, i4, m_only, valu2 * (8.0 * n) / nv, valu2 * (16.0 * n) / nv, valu2 * (32.0 * n) / nv);
        }
    }
    hipFree(d);
    return 0;
}
