// valu_rate.hip -- micro-benchmark: issue rate of plain vs packed fp32 VALU ops on gfx950, by waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.  Prints cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP 64
template <int OP> __global__ void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float c = 1.0000001f; const float2 c2 = {c, c};
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
            if (OP == 0) { asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); }
            if (OP == 1) { asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8"
                                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2)); }
            if (OP == 2) { asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); }
            if (OP == 3) { asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8"
                                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2)); }
            if (OP == 4) { asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0"
                                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); }
            if (OP == 5) { asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
                                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2)); }
            if (OP == 7) { asm volatile("v_pk_fma_f32 %0, %0, %8, %8 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n v_pk_fma_f32 %1, %1, %8, %8 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n v_pk_fma_f32 %2, %2, %8, %8 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n v_pk_fma_f32 %3, %3, %8, %8 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n v_pk_fma_f32 %4, %4, %8, %8 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n v_pk_fma_f32 %5, %5, %8, %8 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n v_pk_fma_f32 %6, %6, %8, %8 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n v_pk_fma_f32 %7, %7, %8, %8 op_sel:[0,0,0] op_sel_hi:[0,1,1]"
                                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2)); }
            if (OP == 8) { asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9"
                                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2.y)); }
            if (OP == 9) { asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7"
                                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2.y)); }
            if (OP == 10) { asm volatile("v_fma_f32 %0, %8, %9, %1\n v_fma_f32 %1, %8, %9, %2\n v_fma_f32 %2, %8, %9, %3\n v_fma_f32 %3, %8, %9, %4\n v_fma_f32 %4, %8, %9, %5\n v_fma_f32 %5, %8, %9, %6\n v_fma_f32 %6, %8, %9, %7\n v_fma_f32 %7, %8, %9, %0"
                                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2.y)); }
            if (OP == 6) { asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4"
                                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c2)); }
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (float)(t1 - t0); }
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y;
}

template <int OP> void run(const char* name, int ninst_per_rep) {
    float* d; hipMalloc(&d, sizeof(float) * (1 + 256 * 1024 * 8));
    const int iters = 2000;
    for (int wps = 1; wps <= 8; wps *= 2) {          // waves per SIMD: block of 256*wps threads... use blocks of 256 threads (1 wave/SIMD), wps blocks per CU
        int blocks = 256 * wps;                        // 256 CUs
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<blocks, 256>>>(d, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<OP><<<blocks, 256>>>(d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        float cyc; hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
        double ninst = (double)iters * ninst_per_rep;   // per wave
        // wall-clock based: instr per second per SIMD = ninst * wps / (ms) ; cycles at 2.4 GHz
        printf("%-14s waves/SIMD %d: %.2f ms, counter ticks/inst (wave0) %.2f, wall ns per wave-inst per SIMD %.3f (= %.2f cyc @2.4GHz)\n", name, wps, ms,
               cyc / ninst, ms * 1e6 / (ninst * wps), ms * 1e6 / (ninst * wps) * 2.4);
    }
    hipFree(d);
}

int main() {
    run<0>("v_add_f32", REP);
    run<1>("v_pk_add_f32", REP);
    run<2>("v_fma_f32", REP);
    run<3>("v_pk_fma_f32", REP);
    run<4>("v_mov_b32", REP);
    run<5>("v_pk_mul_f32", REP);
    run<6>("v_add_f64", REP / 2);
    run<7>("v_pk_fma_f32 op_sel", REP);
    run<8>("v_fmac_f32", REP);
    run<9>("v_fma_f32 d=s2", REP);
    run<10>("v_fma_f32 3 src", REP);
    return 0;
}
