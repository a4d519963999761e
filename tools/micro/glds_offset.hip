// Microbenchmark / semantics probe: where does the instruction offset of global_load_lds_dwordx4 apply -- to the global address
// only, or to the LDS address as well?  One wave copies with offset:1024; the LDS is dumped.
// hipcc --offload-arch=gfx950 -O3 tools/micro/glds_offset.hip -o /tmp/glds_offset && /tmp/glds_offset
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(const unsigned *src, unsigned *dump)
{
    __shared__ __attribute__((aligned(16))) unsigned lds[2048];       // 8 KiB
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned *p = src + threadIdx.x * 4;                          // lane's 16 bytes
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                     (__attribute__((address_space(3))) void *)lds, 16, 1024, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) dump[i] = lds[i];
}

int main()
{
    std::vector<unsigned> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;                           // word i holds i: source word index
    unsigned *src, *dump;
    hipMalloc(&src, 4096 * 4); hipMalloc(&dump, 2048 * 4);
    hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, dump);
    std::vector<unsigned> d(2048);
    hipMemcpy(d.data(), dump, 2048 * 4, hipMemcpyDeviceToHost);
    int first = -1, last = -1;
    for (int i = 0; i < 2048; ++i) if (d[i] != 0xdeadbeefu) { if (first < 0) first = i; last = i; }
    printf("{\"lds_words_written\": [%d, %d], \"first_value_is_source_word\": %u, \"expected_if_offset_applies_to_both\": \"written [256, 511], first value 256\", "
           "\"expected_if_global_only\": \"written [0, 255], first value 256\"}\n", first, last, first >= 0 ? d[first] : 0u);
    return 0;
}
