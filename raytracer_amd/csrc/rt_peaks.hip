// rt_peaks.hip -- achievable-peak microbenchmarks for the two rooflines of SURVEY.md 8(d), run on the GPU box:
// a float4 stream copy / stream read (HBM) and register-resident FMA chains (fp32 VALU, plain and packed).
// Prints one JSON line.  Standalone: hipcc --offload-arch=gfx950 rt_peaks.hip -o rt_peaks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ a, float* __restrict__ out, size_t n) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;                       // keeps the loads alive, never true for zeroed input
}
// 8 independent chains per lane, ITER steps each: 2 flops per fma per lane
template <int ITER>
__global__ __launch_bounds__(256) void k_fma(float* __restrict__ out, float a, float b) {
    float x[8];
    #pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = (float)(threadIdx.x + k);
    for (int it = 0; it < ITER; ++it) {
        #pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = __builtin_fmaf(x[k], a, b);
    }
    float s = 0.f;
    #pragma unroll
    for (int k = 0; k < 8; ++k) s += x[k];
    if (s == 12345.678f) out[0] = s;
}
template <int ITER>
__global__ __launch_bounds__(256) void k_pk_fma(float* __restrict__ out, float a, float b) {
    v2f x[8];
    const v2f va = { a, a }, vb = { b, b };
    #pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = (v2f){ (float)(threadIdx.x + k), (float)(threadIdx.x - k) };
    for (int it = 0; it < ITER; ++it) {
        #pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = __builtin_elementwise_fma(x[k], va, vb);
    }
    float s = 0.f;
    #pragma unroll
    for (int k = 0; k < 8; ++k) s += x[k].x + x[k].y;
    if (s == 12345.678f) out[0] = s;
}

template <class F>
static int time_ms(F launch, int reps, float* ms) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(ms, e0, e1));
    *ms /= reps;
    CHECK(hipGetLastError());
    return 0;
}

int main() {
    int cus = 0, mhz = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    CHECK(hipDeviceGetAttribute(&mhz, hipDeviceAttributeClockRate, 0));         // kHz
    const size_t bytes = 2ull << 30, n = bytes / sizeof(float4);
    float4 *a = nullptr, *b = nullptr; float* out = nullptr;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, bytes));
    const int grid = cus * 16;
    float ms_copy = 0, ms_read = 0, ms_fma = 0, ms_pk = 0;
    if (time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); }, 10, &ms_copy)) return 1;
    if (time_ms([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, out, n); }, 10, &ms_read)) return 1;
    constexpr int ITER = 4096;
    const int fgrid = cus * 32;                                                  // 8 waves per SIMD
    if (time_ms([&] { hipLaunchKernelGGL(k_fma<ITER>, dim3(fgrid), dim3(256), 0, 0, out, 1.0001f, 0.5f); }, 5, &ms_fma)) return 1;
    if (time_ms([&] { hipLaunchKernelGGL(k_pk_fma<ITER>, dim3(fgrid), dim3(256), 0, 0, out, 1.0001f, 0.5f); }, 5, &ms_pk)) return 1;
    const double lanes = (double)fgrid * 256;
    const double fl = lanes * 8.0 * ITER * 2.0;
    const double wave_instr = lanes / 64.0 * 8.0 * ITER;                         // v_fma_f32 wave-instructions
    std::printf("{\"device_cus\": %d, \"clock_mhz\": %d, \"copy_GBps\": %.1f, \"read_GBps\": %.1f, "
                "\"fma_f32_TFLOPs\": %.2f, \"pk_fma_f32_TFLOPs\": %.2f, \"valu_instr_per_simd_cycle\": %.4f}\n",
                cus, mhz / 1000, 2.0 * bytes / (ms_copy * 1e-3) / 1e9, bytes / (ms_read * 1e-3) / 1e9,
                fl / (ms_fma * 1e-3) / 1e12, 2.0 * fl / (ms_pk * 1e-3) / 1e12,
                wave_instr / (ms_fma * 1e-3) / ((double)cus * 4.0 * (double)mhz * 1e3));
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(out);
    return 0;
}
