// Probe of ds_read_b64_tr_b16 on gfx950: LDS holds 16-bit values equal to their element index; every lane issues the read at a chosen byte
// address; the four 16-bit results per lane are printed.  Build: hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int mode, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int elem;                                       // element (16-bit) index each lane points at
    if (mode == 0) elem = 4 * l;                    // lane-linear: lane l -> elems 4l..4l+3
    else if (mode == 1) elem = (l & 15) * 64 + (l >> 4) * 4;   // 16 rows of 64 elems, lane group g reads columns 4g..4g+3 of row (l&15)
    else elem = (l & 15) + (l >> 4) * 64;           // the guide's formula base (j*16 stride handled by hardware?)
    unsigned addr = (unsigned)(size_t)(&lds[0]) + 2u * elem;
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = (unsigned short)(v & 0xffff); out[l * 4 + 1] = (unsigned short)((v >> 16) & 0xffff);
    out[l * 4 + 2] = (unsigned short)((v >> 32) & 0xffff); out[l * 4 + 3] = (unsigned short)(v >> 48);
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, mode, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    }
    return 0;
}
