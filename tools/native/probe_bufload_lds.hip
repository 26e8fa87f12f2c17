// Hardware probe (gfx950): what does `buffer_load_dwordx4 ... offen lds` write to LDS for lanes whose offset is out of the
// descriptor's range?  (The ping-pong igemm uses it for the zero padding of 3x3 convolutions.)  Prints the count of lanes that
// received data / zeros / kept the sentinel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const char* a, int n, int koff, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 1024; i += 64) ((unsigned*)smem)[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, n, 0x00020000);
    int voff = threadIdx.x * 16;
    if (threadIdx.x & 1) voff = 0x80000000;          // far out of range
    if ((threadIdx.x & 3) == 2) voff = n - 8;        // straddles the end
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)(smem), 16, voff, koff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = ((unsigned*)smem)[i];
}
int main() {
    const int n = 4096;
    std::vector<unsigned> h(n / 4);
    for (int i = 0; i < n / 4; ++i) h[i] = 0x1000 + i;
    char* d; unsigned* o;
    hipMalloc(&d, n); hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
    for (int koff = 0; koff <= 128; koff += 128) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, n, koff, o);
        std::vector<unsigned> r(256);
        hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
        printf("soffset %d\n", koff);
        for (int l = 0; l < 8; ++l) printf("  lane %d: %08x %08x %08x %08x\n", l, r[4 * l], r[4 * l + 1], r[4 * l + 2], r[4 * l + 3]);
    }
    return 0;
}
