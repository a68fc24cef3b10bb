#include <hip/hip_runtime.h>
__global__ void k(const unsigned short* a, int nbytes, float* out) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, nbytes, 0x00020000);
    int voff = threadIdx.x * 16 + (threadIdx.x >= 32 ? 1 << 30 : 0);     // upper half out of range -> zeros
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)smem, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = ((float*)smem)[threadIdx.x * 4 + i];
}
int main() {
    float *a, *o; hipMalloc(&a, 4096); hipMalloc(&o, 1024);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i + 1; hipMemcpy(a, h, 4096, hipMemcpyHostToDevice);
    hipMemset(o, 0xff, 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, (const unsigned short*)a, 4096, o);
    float r[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
    printf("%g %g %g %g | %g %g | %g %g\n", r[0], r[1], r[4], r[127], r[128], r[129], r[254], r[255]);
    return 0;
}
