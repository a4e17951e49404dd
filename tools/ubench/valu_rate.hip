// Microbenchmark: VALU issue rate of wave64 on one gfx950 SIMD as a function of waves per SIMD and of
// instruction-level parallelism. Decides whether a kernel at "N cycles per VALU instruction" is
// pipe-bound or latency-bound.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ILP, int KIND>
__global__ void k(float* out, int iters, unsigned long long* cyc) {
    float a[ILP];
    for (int i = 0; i < ILP; ++i) a[i] = threadIdx.x * 0.001f + i;
    const float b = 1.000001f, c = 0.5f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (KIND == 0) a[i] = __builtin_fmaf(a[i], b, c);            // v_fma_f32
                if (KIND == 1) a[i] = __builtin_amdgcn_rcpf(a[i]) + c;        // v_rcp_f32 + v_add
                if (KIND == 2) a[i] = __uint_as_float(__float_as_uint(a[i]) * (0x9E3779BBu + it));  // v_mul_lo_u32
                if (KIND == 3) a[i] = a[i] / (b + a[i]);                      // IEEE division
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < ILP; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        cyc[2 * (threadIdx.x >> 6)] = t0;
        cyc[2 * (threadIdx.x >> 6) + 1] = t1;
    }
}

template <int ILP, int KIND>
void run(const char* name, int waves_per_simd) {
    // one workgroup of waves_per_simd*4 waves on ONE CU (grid = 1): each SIMD gets waves_per_simd waves
    int threads = 64 * 4 * waves_per_simd;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, threads * sizeof(float)); hipMalloc(&cyc, 64 * sizeof(unsigned long long));
    int iters = 2000;
    hipLaunchKernelGGL((k<ILP, KIND>), dim3(1), dim3(threads), 0, 0, out, iters, cyc);
    hipLaunchKernelGGL((k<ILP, KIND>), dim3(1), dim3(threads), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    unsigned long long tt[64]; hipMemcpy(tt, cyc, sizeof(tt), hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int w = 0; w < threads / 64; ++w) { if (tt[2*w] < tmin) tmin = tt[2*w]; if (tt[2*w+1] > tmax) tmax = tt[2*w+1]; }
    unsigned long long c = tmax - tmin;  // all waves of the workgroup, first start to last end
    double per_wave_instr = double(c) / (double(iters) * 16 * ILP);
    printf("%-10s ILP=%d waves/SIMD=%d : %.2f cycles per instr per wave -> %.2f cycles per wave-instr per SIMD\n", name, ILP,
           waves_per_simd, per_wave_instr, per_wave_instr / waves_per_simd);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<1, 0>("fma", w); run<4, 0>("fma", w); run<8, 0>("fma", w);
    }
    for (int w : {1, 2, 4}) { run<4, 1>("rcp+add", w); run<4, 2>("mul_lo_u32", w); run<2, 3>("ieee_div", w); }
    return 0;
}
