// Microbenchmark: issue cost of wave64 instructions on one gfx950 SIMD, as a function of resident waves per SIMD
// and of instruction-level parallelism. It decides whether a kernel that runs at "N cycles per VALU instruction" is
// pipe-bound or latency-bound, and gives the per-opcode weights bench.py's `valu_issue` bound is priced with.
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate && tools/ubench/valu_rate
//
// Method: the whole chip is filled evenly -- one workgroup of 4 W waves per CU (W per SIMD; two workgroups of 16 waves
// for W = 8), kept alone on its CU by the LDS it asks for -- and every wave runs a loop whose body is ONE
// inline-assembly block of 64 instructions of one kind on ILP independent registers (one block: the compiler cannot
// fuse, reorder or delete them, nor put s_nop between them). Each wave stamps s_memtime (shader cycles) and s_memrealtime (100 MHz)
// around the loop and records which SIMD it ran on (HW_REG_HW_ID + HW_REG_XCC_ID). Reported per case:
//   cyc/inst per wave        median over waves of  cycles / instructions of the wave
//   SIMD cyc per wave-inst   cyc/inst per wave / W = the reciprocal issue rate of the SIMD with W waves competing
//   overlap                  per SIMD, sum of wave durations / (last end - first start), median over SIMDs: a check that
//                            the W waves really ran side by side
// A value that stops falling as W grows is the pipe's issue cost. Read-out on MI355X (profiles/r03_valu_rate.txt):
//   one wave alone: 4.5 cycles per VALU instruction of any kind (8.5 transcendental) -- a wave cannot issue faster;
//   W >= 4: v_fma/add/mul/sub_f32, v_add_u32, v_xor_b32 reach 2.0 (64 lanes over two passes of the SIMD-32);
//   v_max/min(3)_f32, v_cmp, v_cndmask, v_mul_lo/hi_u32, v_and_or, v_lshl_add, v_alignbit, v_cvt, DPP moves, v_readlane,
//   v_mbcnt and every v_pk_*_f32 stay at ~3.1; v_rcp / v_sqrt 6.1; a scalar instruction in the stream costs ~2.0 (v_fma +
//   s_add pairs: 4.1 per pair): it takes an issue slot of its own; ds_read_b32 6, ds_write_b32 12, ds_bpermute_b32 18.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#define CK(x)                                                                                 \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

// One instruction on register operand n; %8 and %9 are two more VGPR inputs, %10 an SGPR pair output.
#define OP_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n\t"
#define OP_ADD(n) "v_add_f32 %" #n ", %" #n ", %9\n\t"
#define OP_MUL(n) "v_mul_f32 %" #n ", %" #n ", %8\n\t"
#define OP_MAX(n) "v_max_f32 %" #n ", %" #n ", %8\n\t"
#define OP_MAX3(n) "v_max3_f32 %" #n ", %" #n ", %8, %9\n\t"
#define OP_SUB(n) "v_sub_f32 %" #n ", %" #n ", %9\n\t"
#define OP_CMP_CND(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n\tv_cndmask_b32 %" #n ", %" #n ", %9, vcc\n\t"
#define OP_CMP_S(n) "v_cmp_lt_f32 %10, %" #n ", %8\n\t"
#define OP_CMPX(n) "v_cmp_class_f32 vcc, %" #n ", %8\n\t"
#define OP_ADD_U(n) "v_add_u32 %" #n ", %" #n ", %8\n\t"
#define OP_MUL_LO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n\t"
#define OP_MUL_HI(n) "v_mul_hi_u32 %" #n ", %" #n ", %8\n\t"
#define OP_AND_OR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n\t"
#define OP_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n\t"
#define OP_LSHL_ADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 3, %8\n\t"
#define OP_ALIGNBIT(n) "v_alignbit_b32 %" #n ", %" #n ", %" #n ", 5\n\t"
#define OP_CVT(n) "v_cvt_f32_u32 %" #n ", %" #n "\n\t"
#define OP_DPP(n) "v_mov_b32_dpp %" #n ", %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define OP_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n\t"
#define OP_SQRT(n) "v_sqrt_f32 %" #n ", %" #n "\n\t"
#define OP_MBCNT(n) "v_mbcnt_lo_u32_b32 %" #n ", s4, 0\n\tv_mbcnt_hi_u32_b32 %" #n ", s5, %" #n "\n\t"
#define OP_FMA_SALU(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n\ts_add_u32 s4, s4, 7\n\t"
#define OP_FMA_2SALU(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n\ts_add_u32 s4, s4, 7\n\ts_and_b32 s5, s5, s4\n\t"
#define OP_READLANE(n) "v_readlane_b32 s4, %" #n ", 3\n\t"
#define OP_BPERM(n) "ds_bpermute_b32 %" #n ", %8, %" #n "\n\t"
#define OP_DSREAD(n) "ds_read_b32 %" #n ", %8\n\t"
#define OP_DSWRITE(n) "ds_write_b32 %8, %" #n "\n\t"
#define OP_PK_FMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n\t"
#define OP_PK_MUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n\t"
#define OP_PK_ADD(n) "v_pk_add_f32 %" #n ", %" #n ", %9\n\t"

// 16 instructions per group over ILP registers; a loop body is four groups (64 instructions), so that the loop's own
// three scalar instructions and its taken branch (measured below: "loop overhead") are under 10 % of it
#define BODY8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define BODY4(OP) OP(0) OP(1) OP(2) OP(3) OP(0) OP(1) OP(2) OP(3) OP(0) OP(1) OP(2) OP(3) OP(0) OP(1) OP(2) OP(3)
#define BODY2(OP) OP(0) OP(1) OP(0) OP(1) OP(0) OP(1) OP(0) OP(1) OP(0) OP(1) OP(0) OP(1) OP(0) OP(1) OP(0) OP(1)
#define BODY1(OP) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <class T> __device__ __forceinline__ T mkv(float x) { return T(x); }
template <> __device__ __forceinline__ f32x2 mkv<f32x2>(float x) { return f32x2{x, x * 0.5f}; }
template <class T> __device__ __forceinline__ float probe(T v) { return float(v == T(3.0f) ? 1 : 0); }
template <> __device__ __forceinline__ float probe<f32x2>(f32x2 v) { return v.x + v.y; }

struct Stamp {
    unsigned long long cycles, rt0, rt1;
    uint32_t simd_key, pad;
};

// T = float / uint32_t / f32x2: the register class of the 8 independent operands
#define DEFINE_KERNEL(NAME, T, BODY, INIT_B, INIT_C, WAIT)                                                          \
    __global__ __launch_bounds__(1024) void NAME(float* out, int iters, Stamp* st) {                                 \
        extern __shared__ uint32_t lds[];                                                                           \
        T r[8];                                                                                                     \
        for (int i = 0; i < 8; ++i) r[i] = mkv<T>(1.0f + threadIdx.x * 0.001f + i);                                      \
        T b = mkv<T>(INIT_B), c = mkv<T>(INIT_C);                                                                             \
        unsigned long long sg = 0;                                                                                  \
        lds[threadIdx.x & 255] = threadIdx.x;                                                                       \
        __syncthreads();                                                                                            \
        const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();                                            \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                 \
        for (int it = 0; it < iters; ++it) {                                                                        \
            asm volatile(BODY BODY BODY BODY WAIT                                                                                  \
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) \
                         : "v"(b), "v"(c), "s"(sg)                                                                  \
                         : "vcc", "s4", "s5", "scc", "memory");                                                     \
        }                                                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                          \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                 \
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();                                            \
        float s = 0;                                                                                                \
        for (int i = 0; i < 8; ++i) s += probe<T>(r[i]);                                            \
        out[size_t(blockIdx.x) * blockDim.x + threadIdx.x] = s;                                                     \
        if ((threadIdx.x & 63) == 0) {                                                                              \
            uint32_t hw, xcc;                                                                                       \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                        \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));                                \
            Stamp& o = st[size_t(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)];                                             \
            o.cycles = t1 - t0, o.rt0 = rt0, o.rt1 = rt1;                                                           \
            o.simd_key = (xcc << 16) | (hw & 0xFF30u); /* se, sh, cu, simd (wave and pipe ids masked out) */        \
        }                                                                                                           \
    }


#define K(NAME, T, OP, B, C, WAIT)            \
    DEFINE_KERNEL(NAME##_8, T, BODY8(OP), B, C, WAIT) \
    DEFINE_KERNEL(NAME##_4, T, BODY4(OP), B, C, WAIT) \
    DEFINE_KERNEL(NAME##_2, T, BODY2(OP), B, C, WAIT) \
    DEFINE_KERNEL(NAME##_1, T, BODY1(OP), B, C, WAIT)

K(k_fma, float, OP_FMA, 1.000001f, 0.5f, "")
K(k_add, float, OP_ADD, 1.000001f, 0.5f, "")
K(k_mul, float, OP_MUL, 1.000001f, 0.5f, "")
K(k_sub, float, OP_SUB, 1.000001f, 0.5f, "")
K(k_max, float, OP_MAX, 1.000001f, 0.5f, "")
K(k_max3, float, OP_MAX3, 1.000001f, 0.5f, "")
K(k_cmp_cnd, float, OP_CMP_CND, 1.000001f, 0.5f, "")
K(k_cmp_s, float, OP_CMP_S, 1.000001f, 0.5f, "")
K(k_add_u, uint32_t, OP_ADD_U, 3.0f, 5.0f, "")
K(k_mul_lo, uint32_t, OP_MUL_LO, 3.0f, 5.0f, "")
K(k_mul_hi, uint32_t, OP_MUL_HI, 3.0f, 5.0f, "")
K(k_and_or, uint32_t, OP_AND_OR, 3.0f, 5.0f, "")
K(k_xor, uint32_t, OP_XOR, 3.0f, 5.0f, "")
K(k_lshl_add, uint32_t, OP_LSHL_ADD, 3.0f, 5.0f, "")
K(k_alignbit, uint32_t, OP_ALIGNBIT, 3.0f, 5.0f, "")
K(k_cvt, float, OP_CVT, 1.0f, 0.5f, "")
K(k_dpp, uint32_t, OP_DPP, 3.0f, 5.0f, "")
K(k_rcp, float, OP_RCP, 1.0f, 0.5f, "")
K(k_sqrt, float, OP_SQRT, 1.0f, 0.5f, "")
K(k_mbcnt, uint32_t, OP_MBCNT, 3.0f, 5.0f, "")
K(k_fma_salu, float, OP_FMA_SALU, 1.000001f, 0.5f, "")
K(k_fma_2salu, float, OP_FMA_2SALU, 1.000001f, 0.5f, "")
K(k_readlane, uint32_t, OP_READLANE, 3.0f, 5.0f, "")
K(k_bperm, uint32_t, OP_BPERM, 64.0f, 5.0f, "s_waitcnt lgkmcnt(0)\n\t")
K(k_dsread, uint32_t, OP_DSREAD, 64.0f, 5.0f, "s_waitcnt lgkmcnt(0)\n\t")
K(k_dswrite, uint32_t, OP_DSWRITE, 64.0f, 5.0f, "s_waitcnt lgkmcnt(0)\n\t")
K(k_pk_fma, f32x2, OP_PK_FMA, 1.000001f, 0.5f, "")
K(k_pk_mul, f32x2, OP_PK_MUL, 1.000001f, 0.5f, "")
K(k_pk_add, f32x2, OP_PK_ADD, 1.000001f, 0.5f, "")

// Short-body variant (16 FMAs per iteration instead of 64): the difference in cycles per iteration against k_fma_4,
// divided out, prices one loop back-edge (s_add + s_cmp + taken s_cbranch).
__global__ __launch_bounds__(1024) void k_loop16(float* out, int iters, Stamp* st) {
    extern __shared__ uint32_t lds[];
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = 1.0f + threadIdx.x * 0.001f + i;
    float b = 1.000001f, c = 0.5f;
    lds[threadIdx.x & 255] = threadIdx.x;
    __syncthreads();
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters * 4; ++it) {
        asm volatile(BODY4(OP_FMA)
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                     : "v"(b), "v"(c)
                     : "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    out[size_t(blockIdx.x) * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
        Stamp& o = st[size_t(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)];
        o.cycles = t1 - t0, o.rt0 = rt0, o.rt1 = rt1;
        o.simd_key = (xcc << 16) | (hw & 0xFF30u);
    }
}

static int g_cus = 256;
typedef void (*kern_t)(float*, int, Stamp*);

void run(const char* name, kern_t kern, int ilp, int per_op, int waves_per_simd) {
    // W <= 4: one workgroup of 4 W waves per CU (its waves start together, W on each SIMD); W = 8: two such of 16 waves.
    // The LDS request keeps a further workgroup off the CU.
    const int per_cu = waves_per_simd <= 4 ? 1 : waves_per_simd / 4;
    const int blocks = g_cus * per_cu, threads = 256 * (waves_per_simd / per_cu);
    const size_t lds = (160u * 1024u) / size_t(per_cu) - 4096u;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    float* out;
    Stamp* st;
    CK(hipMalloc(&out, size_t(blocks) * threads * sizeof(float)));
    CK(hipMalloc(&st, size_t(blocks) * (threads / 64) * sizeof(Stamp)));
    const int iters = 6000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, out, iters, st);  // warm-up
    CK(hipGetLastError());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, out, iters, st);
    CK(hipGetLastError());
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Stamp> tt(size_t(blocks) * (threads / 64));
    CK(hipMemcpy(tt.data(), st, tt.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<unsigned long long> cyc;
    struct Acc {
        unsigned long long sum = 0, lo = ~0ull, hi = 0;
    };
    std::map<uint32_t, Acc> simds;
    double rt_sum = 0;
    for (const Stamp& s : tt) {
        cyc.push_back(s.cycles);
        Acc& a = simds[s.simd_key];
        a.sum += s.rt1 - s.rt0, a.lo = std::min(a.lo, s.rt0), a.hi = std::max(a.hi, s.rt1);
        rt_sum += double(s.rt1 - s.rt0);
    }
    std::sort(cyc.begin(), cyc.end());
    std::vector<double> conc;
    for (auto& kv : simds) conc.push_back(double(kv.second.sum) / double(kv.second.hi - kv.second.lo));
    std::sort(conc.begin(), conc.end());
    const double resident = conc[conc.size() / 2];
    const double med = double(cyc[cyc.size() / 2]), n_inst = double(iters) * 64 * per_op;
    const double per_wave = med / n_inst;
    const double clock_ghz = med / (rt_sum / double(tt.size()) * 10.0);  // s_memrealtime ticks are 10 ns
    // (the W waves of a SIMD belong to one workgroup -- two for W = 8 -- and are resident together by construction; `overlap`
    // is what their time stamps say, below W when they start a little apart)
    printf("%-26s ILP=%d W=%d : %6.2f cyc/inst per wave -> %5.2f SIMD cyc per wave-inst   (overlap %.2f on %zu SIMDs, %.3f ms, %.2f GHz)\n",
           name, ilp, waves_per_simd, per_wave, per_wave / waves_per_simd, resident, simds.size(), ms, clock_ghz);
    fflush(stdout);
    CK(hipFree(out));
    CK(hipFree(st));
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
}

#define SWEEP(label, NAME, per_op, full)                                   \
    for (int w : {1, 2, 4, 8}) {                                          \
        if (full) run(label, NAME##_1, 1, per_op, w);                     \
        if (full) run(label, NAME##_2, 2, per_op, w);                     \
        run(label, NAME##_4, 4, per_op, w);                               \
        run(label, NAME##_8, 8, per_op, w);                               \
    }

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    g_cus = prop.multiProcessorCount;
    printf("# %s, %d CUs, clock %d kHz; one workgroup of 4 W waves per CU (W per SIMD; W = 8: two of 16 waves)\n", prop.gcnArchName, g_cus,
           prop.clockRate);
    SWEEP("v_fma_f32", k_fma, 1, true)
    for (int w : {1, 2, 4, 8}) run("v_fma_f32, 16 per loop trip", k_loop16, 4, 1, w);
    SWEEP("v_add_f32", k_add, 1, false)
    SWEEP("v_mul_f32", k_mul, 1, false)
    SWEEP("v_sub_f32", k_sub, 1, false)
    SWEEP("v_max_f32", k_max, 1, false)
    SWEEP("v_max3_f32", k_max3, 1, false)
    SWEEP("v_pk_fma_f32", k_pk_fma, 1, true)
    SWEEP("v_pk_mul_f32", k_pk_mul, 1, false)
    SWEEP("v_pk_add_f32", k_pk_add, 1, false)
    SWEEP("v_cmp_lt_f32+v_cndmask", k_cmp_cnd, 2, true)
    SWEEP("v_cmp_lt_f32 -> sgpr", k_cmp_s, 1, false)
    SWEEP("v_add_u32", k_add_u, 1, false)
    SWEEP("v_xor_b32", k_xor, 1, false)
    SWEEP("v_mul_lo_u32", k_mul_lo, 1, false)
    SWEEP("v_mul_hi_u32", k_mul_hi, 1, false)
    SWEEP("v_and_or_b32", k_and_or, 1, false)
    SWEEP("v_lshl_add_u32", k_lshl_add, 1, false)
    SWEEP("v_alignbit_b32", k_alignbit, 1, false)
    SWEEP("v_cvt_f32_u32", k_cvt, 1, false)
    SWEEP("v_mov_b32 dpp", k_dpp, 1, false)
    SWEEP("v_rcp_f32", k_rcp, 1, false)
    SWEEP("v_sqrt_f32", k_sqrt, 1, false)
    SWEEP("v_mbcnt_lo+hi", k_mbcnt, 2, false)
    SWEEP("v_readlane_b32", k_readlane, 1, false)
    SWEEP("v_fma_f32 + s_add_u32", k_fma_salu, 2, false)
    SWEEP("v_fma_f32 + 2 salu", k_fma_2salu, 3, false)
    SWEEP("ds_bpermute_b32", k_bperm, 1, false)
    SWEEP("ds_read_b32", k_dsread, 1, false)
    SWEEP("ds_write_b32", k_dswrite, 1, false)
    return 0;
}
