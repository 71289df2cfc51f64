// PMC calibration workload (tools/refresh_profiles.sh): kernels that move a KNOWN number of bytes with the access shapes of the
// convolution family, run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`; tools/traffic_summary.py divides the counters by
// the known byte counts and applies the factors to the family's counters (MI355X_MICROARCH.md §HBM: "calibrate on a known byte count
// in your own access pattern before trusting an absolute").
//   calib_copy16_kernel : 16 B per lane global loads  -> 16 B per lane stores   (epilogue stores of the pre-split kernels)
//   calib_copy4_kernel  :  4 B per lane loads          ->  4 B per lane stores   (register-staged kernels' gathers)
//   calib_dma16_kernel  : `buffer_load_dwordx4 ... lds` (LDS-DMA, 16 B per lane) -> ds_read -> 16 B stores  (the pre-split staging)
// Buffers are 1 GiB each (beyond the 256 MiB Infinity Cache), every byte read once and written once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

__global__ __launch_bounds__(256) void calib_copy16_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = x[i];
}
__global__ __launch_bounds__(256) void calib_copy4_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = x[i];
}
__global__ __launch_bounds__(256) void calib_dma16_kernel(const f32x4* x, f32x4* y, int64_t n_units) {      // n_units % (256 * gridDim) == 0
    __shared__ f32x4 buf[256];
    const int lane = threadIdx.x & 63, wn = threadIdx.x >> 6;
    const int64_t per_block = n_units / gridDim.x;                         // < 2^27 units = 2 GiB
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (int64_t)blockIdx.x * per_block), 0, (int)(per_block * 16), 0x00020000);
    f32x4* yo = y + (int64_t)blockIdx.x * per_block;
    for (int64_t i = 0; i < per_block; i += 256) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(buf + wn * 64), 16, (int)((i + wn * 64 + lane) * 16), 0, 0, 0);
        __builtin_amdgcn_s_waitcnt(0x0f70);
        yo[i + threadIdx.x] = buf[threadIdx.x];                           // each wave reads back its own piece
    }
}

int main() {
    const int64_t bytes = 1ll << 30;
    void *x, *y;
    if (hipMalloc(&x, bytes) != hipSuccess || hipMalloc(&y, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(x, 1, bytes); hipMemset(y, 0, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_copy16_kernel, dim3(8192), dim3(256), 0, 0, (const f32x4*)x, (f32x4*)y, bytes / 16);
        hipLaunchKernelGGL(calib_copy4_kernel, dim3(8192), dim3(256), 0, 0, (const float*)x, (float*)y, bytes / 4);
        hipLaunchKernelGGL(calib_dma16_kernel, dim3(4096), dim3(256), 0, 0, (const f32x4*)x, (f32x4*)y, bytes / 16);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("calibration kernels: %s, %lld bytes read + %lld written per launch\n", hipGetErrorString(e), (long long)bytes, (long long)bytes);
    return e == hipSuccess ? 0 : 1;
}
