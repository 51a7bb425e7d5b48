// rt_oprate.hip -- issue cost of the VALU instruction classes the traversal step is made of (one dependent chain of 8 per lane,
// 8 waves per SIMD): tells which instructions are full-rate and which quarter-rate on gfx950.  Standalone, like rt_peaks.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
// 8 independent chains, each step one instruction of the given kind
#define KERNEL(NAME, ASM, ...) \
__global__ __launch_bounds__(256) void NAME(float* out, float a, float b) { \
    float x[8]; \
    for (int k = 0; k < 8; ++k) x[k] = (float)(threadIdx.x + k) * a; \
    for (int it = 0; it < 4096; ++it) { \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile(ASM : "+v"(x[k]) : __VA_ARGS__); \
    } \
    float s = 0.f; for (int k = 0; k < 8; ++k) s += x[k]; \
    if (s == 12345.678f) out[0] = s; }
KERNEL(k_fma,    "v_fma_f32 %0, %0, %1, %2", "v"(a), "v"(b))
KERNEL(k_mul,    "v_mul_f32 %0, %0, %1", "v"(a))
KERNEL(k_add,    "v_add_f32 %0, %0, %1", "v"(a))
KERNEL(k_min,    "v_min_f32 %0, %0, %1", "v"(a))
KERNEL(k_max3,   "v_max3_f32 %0, %0, %1, %2", "v"(a), "v"(b))
KERNEL(k_cvtub,  "v_cvt_f32_ubyte1 %0, %0", "v"(a))
KERNEL(k_cvtu,   "v_cvt_f32_u32 %0, %0", "v"(a))
KERNEL(k_and,    "v_and_b32 %0, %0, %1", "v"(a))
KERNEL(k_addu,   "v_add_u32 %0, %0, %1", "v"(a))
KERNEL(k_minu,   "v_min_u32 %0, %0, %1", "v"(a))
KERNEL(k_lshl,   "v_lshlrev_b32 %0, 1, %0", "v"(a))
KERNEL(k_cndm,   "v_cndmask_b32 %0, %0, %1, vcc", "v"(a))
KERNEL(k_bfe,    "v_bfe_u32 %0, %0, 8, 8", "v"(a))
KERNEL(k_perm,   "v_perm_b32 %0, %0, %1, %2", "v"(a), "v"(b))
KERNEL(k_mov,    "v_mov_b32 %0, %0", "v"(a))
KERNEL(k_rcp,    "v_rcp_f32 %0, %0", "v"(a))
KERNEL(k_cmp,    "v_cmp_lt_f32 vcc, %0, %1", "v"(a))
KERNEL(k_cvtpk,  "v_cvt_pkrtz_f16_f32 %0, %0, %1", "v"(a))
KERNEL(k_sdwa,   "v_cvt_f32_ubyte0_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2", "v"(a))
KERNEL(k_fmamix, "v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]", "v"(a), "v"(b))          // src0 read as f16 (high half)
KERNEL(k_pkminu, "v_pk_min_u16 %0, %0, %1", "v"(a))
KERNEL(k_ldexp,  "v_ldexp_f32 %0, %0, %1", "v"(a))
KERNEL(k_andor,  "v_and_or_b32 %0, %0, %1, %2", "v"(a), "v"(b))
KERNEL(k_lshlor, "v_lshl_or_b32 %0, %0, 2, %1", "v"(a))
KERNEL(k_align,  "v_alignbit_b32 %0, %0, %0, 16")
KERNEL(k_madu24, "v_mad_u32_u24 %0, %0, %1, %2", "v"(a), "v"(b))
KERNEL(k_med3,   "v_med3_f32 %0, %0, %1, %2", "v"(a), "v"(b))
KERNEL(k_cvtf16, "v_cvt_f32_f16 %0, %0")
KERNEL(k_bfi,    "v_bfi_b32 %0, %0, %1, %2", "v"(a), "v"(b))
KERNEL(k_cmpu,   "v_cmp_lt_u32 vcc, %0, %1", "v"(a))
KERNEL(k_cndm2,  "v_cndmask_b32 %0, %1, %2, vcc", "v"(a), "v"(b))                          // no chain through the destination
KERNEL(k_subf,   "v_sub_f32 %0, %0, %1", "v"(a))
KERNEL(k_xor,    "v_xor_b32 %0, %0, %1", "v"(a))
KERNEL(k_or3,    "v_or3_b32 %0, %0, %1, %2", "v"(a), "v"(b))
KERNEL(k_orsdwa, "v_or_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2", "v"(a))   // magic | byte 2: the integer half of a conversion
KERNEL(k_addsdwa,"v_add_f32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1", "v"(a))
KERNEL(k_andsdwa,"v_and_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3", "v"(a))
// same, with a clobber list (instructions that write or read a named SGPR pair) and a per-step instruction count
#define KERNEL_C(NAME, ASM, CLOB, ...) \
__global__ __launch_bounds__(256) void NAME(float* out, float a, float b) { \
    float x[8]; \
    for (int k = 0; k < 8; ++k) x[k] = (float)(threadIdx.x + k) * a; \
    asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20", "s21"); \
    for (int it = 0; it < 4096; ++it) { \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile(ASM : "+v"(x[k]) : __VA_ARGS__ : CLOB); \
    } \
    float s = 0.f; for (int k = 0; k < 8; ++k) s += x[k]; \
    if (s == 12345.678f) out[0] = s; }
KERNEL_C(k_cndm_s,   "v_cndmask_b32_e64 %0, %0, %1, s[20:21]", "memory", "v"(a))                      // mask in an SGPR pair other than vcc
KERNEL_C(k_cndm_fma, "v_cndmask_b32 %0, %0, %1, vcc\n v_fma_f32 %0, %0, %1, %1", "memory", "v"(a))     // 2 instructions per step
KERNEL_C(k_cndm_3fma,"v_cndmask_b32 %0, %0, %1, vcc\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1", "memory", "v"(a))   // 4 per step
KERNEL_C(k_cmp_s,    "v_cmp_lt_f32_e64 s[20:21], %0, %1", "s20", "v"(a))
KERNEL_C(k_cmp_cndm, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc", "vcc", "v"(a))         // the pair the compiler emits for a select
KERNEL_C(k_ashr,     "v_ashrrev_i32 %0, 31, %0", "memory", "v"(a))
KERNEL_C(k_maxi,     "v_max_i32 %0, %0, %1", "memory", "v"(a))
KERNEL_C(k_or,       "v_or_b32 %0, %0, %1", "memory", "v"(a))
KERNEL_C(k_readlane, "v_readfirstlane_b32 s20, %0", "s20", "v"(a))
KERNEL_C(k_cndm_lit, "v_cndmask_b32_e64 %0, %0, -1, vcc", "memory", "v"(a))
typedef void (*kfn)(float*, float, float);
int main() {
    float* out; CHECK(hipMalloc(&out, 64));
    int cus = 0; CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    struct { const char* n; kfn f; } ks[] = { {"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_min_f32", k_min}, {"v_max3_f32", k_max3},
        {"v_cvt_f32_ubyte1", k_cvtub}, {"v_cvt_f32_u32", k_cvtu}, {"v_and_b32", k_and}, {"v_add_u32", k_addu}, {"v_min_u32", k_minu}, {"v_lshlrev_b32", k_lshl},
        {"v_cndmask_b32", k_cndm}, {"v_bfe_u32", k_bfe}, {"v_perm_b32", k_perm}, {"v_mov_b32", k_mov}, {"v_rcp_f32", k_rcp}, {"v_cmp_lt_f32", k_cmp}, {"v_cvt_pkrtz", k_cvtpk}, {"cvt_ubyte_sdwa", k_sdwa},
        {"v_fma_mix_f32", k_fmamix}, {"v_pk_min_u16", k_pkminu}, {"v_ldexp_f32", k_ldexp}, {"v_and_or_b32", k_andor},
        {"v_lshl_or_b32", k_lshlor}, {"v_alignbit_b32", k_align}, {"v_mad_u32_u24", k_madu24}, {"v_med3_f32", k_med3}, {"v_cvt_f32_f16", k_cvtf16},
        {"v_bfi_b32", k_bfi}, {"v_cmp_lt_u32", k_cmpu}, {"v_cndmask(no chain)", k_cndm2}, {"v_sub_f32", k_subf}, {"v_xor_b32", k_xor}, {"v_or3_b32", k_or3}, {"v_or_b32_sdwa byte", k_orsdwa}, {"v_add_f32_sdwa byte", k_addsdwa}, {"v_and_b32_sdwa byte", k_andsdwa},
        {"v_cndmask sgpr pair", k_cndm_s}, {"cndmask+fma (x2)", k_cndm_fma}, {"cndmask+3fma (x4)", k_cndm_3fma}, {"v_cmp -> sgpr pair", k_cmp_s},
        {"v_cmp+v_cndmask (x2)", k_cmp_cndm}, {"v_ashrrev_i32", k_ashr}, {"v_max_i32", k_maxi}, {"v_or_b32", k_or},
        {"v_readfirstlane", k_readlane}, {"v_cndmask -1 literal", k_cndm_lit} };
    const int grid = cus * 32;
    for (auto& k : ks) {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k.f, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k.f, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
        double wi = (double)grid * 4 * 8.0 * 4096;
        std::printf("%-20s %.4f instr/SIMD-cycle  (%.2f cycles per step of the chain; a step is one instruction unless the name says xN)\n", k.n, wi / (ms * 1e-3) / (cus * 4.0 * 2.4e9), (cus * 4.0 * 2.4e9) * (ms * 1e-3) / wi);
    }
    return 0;
}
