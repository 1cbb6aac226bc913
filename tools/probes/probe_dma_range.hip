// probe_dma_range.hip -- what does an LDS-DMA buffer load do at the end of its buffer?  (measurement tool, not product)
// A raw buffer resource of N bytes (N = 8 mod 16), every lane asks for 16 bytes at 16 * lane (+ 1024 per piece): the piece that straddles the end
// must deliver its in-range dwords and zeros for the rest, lanes wholly past the end zeros and no memory access -- lec_boxplane.hip
// relies on exactly that (runs of box rows whose length is an odd number of doubles).  Prints what landed in LDS.
// Build: hipcc -O3 --offload-arch=gfx950 probe_dma_range.hip -o probe_dma_range
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

__global__ void k(const double* g, double* out, int nbytes) {
    __shared__ __attribute__((aligned(1024))) double sm[512];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) sm[i] = -1.0;
    __syncthreads();
    const unsigned long long b = (unsigned long long)g;
    u32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) & 0xffffu;
    r.z = __builtin_amdgcn_readfirstlane((unsigned)nbytes);
    r.w = 0x00020000u;
    const unsigned lds0 = (unsigned)(uintptr_t)sm;
    dma16(r, 16u * lane, 0u, lds0);
    dma16(r, 16u * lane, 1024u, lds0 + 1024u);
    dma16(r, 16u * lane, 2048u, lds0 + 2048u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = sm[i];
}

int main() {
    const int n = 600;
    double h[n], *g, *o, ho[512];
    for (int i = 0; i < n; ++i) h[i] = i + 1;
    CK(hipMalloc(&g, sizeof h + 64)); CK(hipMalloc(&o, sizeof ho));
    for (int shift = 0; shift < 2; ++shift) {                 // the run starts on a 16-byte boundary / 8 bytes past one
        CK(hipMemcpy(g + shift, h, sizeof h, hipMemcpyHostToDevice));
        for (int nd : {305, 61, 244, 366, 122}) {             // doubles in the run (odd counts end in the middle of a lane's 16 bytes)
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g + shift, o, nd * 8);
            CK(hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost));
            int ok = 1, first_bad = -1;
            for (int i = 0; i < 384; ++i) {
                const double want = i < nd ? i + 1 : 0.0;
                if (ho[i] != want) { ok = 0; if (first_bad < 0) first_bad = i; }
            }
            printf("start %s, run of %3d doubles: %s", shift ? "8 mod 16" : "16-aligned", nd, ok ? "in-range values, zeros beyond" : "MISMATCH");
            if (!ok) printf(" (first at %d: got %g; around the end: %g %g %g %g)", first_bad, ho[first_bad], ho[nd - 2], ho[nd - 1], ho[nd], ho[nd + 1]);
            printf("\n");
        }
    }
    return 0;
}
