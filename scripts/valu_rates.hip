// Instruction issue-rate probe for the VALU operations the path kernel is made of (gfx950).
// Each kernel runs 8 independent dependency chains of one instruction per lane so that issue rate, not
// latency, is measured; 1024 workgroups x 256 lanes (4 waves per SIMD).  Output: cycles per wave64
// instruction per SIMD, derived from the measured time at the measured clock.
//   hipcc --offload-arch=gfx950 -O3 scripts/valu_rates.hip -o /tmp/vr && /tmp/vr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 4096
#define CHAINS 8

#define PROBE(NAME, TYPE, INIT, BODY)                                                    \
    __global__ __launch_bounds__(256) void NAME(TYPE *out, TYPE seed)                    \
    {                                                                                    \
        TYPE v[CHAINS];                                                                  \
        _Pragma("unroll") for (int c = 0; c < CHAINS; c++) v[c] = INIT;                  \
        for (int it = 0; it < ITERS; it++) {                                             \
            _Pragma("unroll") for (int c = 0; c < CHAINS; c++) { BODY; }                 \
        }                                                                                \
        TYPE s = v[0];                                                                   \
        _Pragma("unroll") for (int c = 1; c < CHAINS; c++) s += v[c];                    \
        if (s == (TYPE)12345) out[threadIdx.x] = s;                                      \
    }

PROBE(k_fma_f64, double, seed + c + threadIdx.x, asm volatile("v_fma_f64 %0, %0, %0, %1" : "+v"(v[c]) : "v"(seed)))
PROBE(k_mul_f64, double, seed + c + threadIdx.x, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[c]) : "v"(seed)))
PROBE(k_add_f64, double, seed + c + threadIdx.x, asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[c]) : "v"(seed)))
PROBE(k_rcp_f64, double, seed + c + threadIdx.x, asm volatile("v_rcp_f64 %0, %0" : "+v"(v[c])))
PROBE(k_sqrt_f64, double, seed + c + threadIdx.x, asm volatile("v_sqrt_f64 %0, %0" : "+v"(v[c])))
PROBE(k_rsq_f64, double, seed + c + threadIdx.x, asm volatile("v_rsq_f64 %0, %0" : "+v"(v[c])))
PROBE(k_divscale_f64, double, seed + c + threadIdx.x, asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(v[c]) : "v"(seed) : "vcc"))
PROBE(k_divfmas_f64, double, seed + c + threadIdx.x, asm volatile("v_div_fmas_f64 %0, %0, %1, %0" : "+v"(v[c]) : "v"(seed) : "vcc"))
PROBE(k_divfixup_f64, double, seed + c + threadIdx.x, asm volatile("v_div_fixup_f64 %0, %0, %1, %0" : "+v"(v[c]) : "v"(seed)))
PROBE(k_ldexp_f64, double, seed + c + threadIdx.x, asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(v[c])))
PROBE(k_frexp_mant_f64, double, seed + c + threadIdx.x, asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(v[c])))
PROBE(k_mad_u64_u32, uint64_t, seed + c + threadIdx.x,
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(v[c]) : "v"((uint32_t)seed), "v"((uint32_t)threadIdx.x) : "vcc"))
PROBE(k_mul_hi_u32, uint32_t, seed + c + threadIdx.x, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[c]) : "v"(seed)))
PROBE(k_mul_lo_u32, uint32_t, seed + c + threadIdx.x, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[c]) : "v"(seed)))
PROBE(k_xor_b32, uint32_t, seed + c + threadIdx.x, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[c]) : "v"(seed)))
PROBE(k_add_u32, uint32_t, seed + c + threadIdx.x, asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[c]) : "v"(seed)))
PROBE(k_cndmask_b32, uint32_t, seed + c + threadIdx.x, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[c]) : "v"(seed) : "vcc"))
PROBE(k_fma_f32, float, seed + c + threadIdx.x, asm volatile("v_fma_f32 %0, %0, %0, %1" : "+v"(v[c]) : "v"(seed)))
PROBE(k_pk_fma_f32, uint64_t, seed + c + threadIdx.x, asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(v[c])))
PROBE(k_cvt_f64_u32, double, seed + c + threadIdx.x,
      { uint32_t t; asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(t) : "v"(v[c])); asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(v[c]) : "v"(t)); })

template <typename T, typename K>
double run(K kern, T seed, T *buf, double ghz)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(1024), dim3(256), 0, 0, buf, seed);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(1024), dim3(256), 0, 0, buf, seed);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // 1024 WGs x 4 waves = 4096 waves on 1024 SIMDs -> 4 waves per SIMD, each ITERS*CHAINS instructions
    const double instr_per_simd = 4.0 * ITERS * CHAINS;
    return ms * 1e-3 * ghz * 1e9 / instr_per_simd;
}

int main()
{
    void *buf;
    (void)hipMalloc(&buf, 4096);
    int khz = 0;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz * 1e-6;
    printf("clock %.2f GHz (attribute); cycles per wave64 instruction per SIMD (4 = full rate):\n", ghz);
#define R(T, K, S) printf("  %-18s %6.2f\n", #K + 2, run<T>(K, (T)S, (T *)buf, ghz));
    R(double, k_fma_f64, 1.0000001) R(double, k_mul_f64, 1.0000001) R(double, k_add_f64, 1e-9)
    R(double, k_rcp_f64, 1.5) R(double, k_sqrt_f64, 1.5) R(double, k_rsq_f64, 1.5)
    R(double, k_divscale_f64, 1.5) R(double, k_divfmas_f64, 1.5) R(double, k_divfixup_f64, 1.5)
    R(double, k_ldexp_f64, 1.5) R(double, k_frexp_mant_f64, 1.5)
    R(uint64_t, k_mad_u64_u32, 0x9E3779B9u) R(uint32_t, k_mul_hi_u32, 0x9E3779B9u) R(uint32_t, k_mul_lo_u32, 0x9E3779B9u)
    R(uint32_t, k_xor_b32, 0x9E3779B9u) R(uint32_t, k_add_u32, 0x9E3779B9u) R(uint32_t, k_cndmask_b32, 0x9E3779B9u)
    R(float, k_fma_f32, 1.0000001f) R(uint64_t, k_pk_fma_f32, 0x3f8000003f800000ull)
    R(double, k_cvt_f64_u32, 1.5)
    printf("  (cvt_f64_u32 line = v_cvt_u32_f64 + v_cvt_f64_u32 pair)\n");
    return 0;
}
