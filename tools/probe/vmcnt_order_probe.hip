// Probe: is `s_waitcnt vmcnt(N)` exact when out-of-range (zero-returning) buffer loads sit between in-range ones?
// The split-bf16 staged weight gradient (csrc/pwsplitw.hip) issues UNCONDITIONAL buffer loads -- slots that have no row get an
// out-of-range offset and read zeros -- and waits with partial counts (vmcnt(7), vmcnt(5), ...), which is only correct if all loads
// retire in issue order.  Symptom that started this (tools/diag_wgrad_race.py): in 1-3 % of backward passes the operand rows staged by
// lanes 48-63 of some waves were stale for one half-stage, in the one template variant where 4 of the 6 loads per thread are out of range.
//
// Each lane issues   A: in-range 16-byte load of COLD memory (walks a 2 GB buffer),  B: out-of-range load,  C: in-range load of a hot line
// with the destination of A preset to a poison value, then waits for `vmcnt(2)` / `vmcnt(1)` (= "A has landed" if retirement is in
// order) and immediately stores A's registers.  A poisoned or partially poisoned store = the wait was released early.
//   hipcc --offload-arch=gfx950 -O3 -o vmcnt_order_probe vmcnt_order_probe.hip && ./vmcnt_order_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int WAITN>
__global__ __launch_bounds__(256) void probe(const unsigned* cold, unsigned cold_bytes, const unsigned* hot, unsigned long long* bad_lane, int iters) {
    const int lane = threadIdx.x & 63;
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)cold, 0, cold_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void*)hot, 0, 4096, 0x00020000);
    unsigned long long bad = 0;
    __shared__ __attribute__((aligned(16))) u4 slot[256];
    const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) u4*)slot + threadIdx.x * 16;
    for (int it = 0; it < iters; ++it) {
        // a pseudo-random 16-byte aligned cold offset per lane group of 8 (one 128-byte line per 8 lanes, as in the kernel)
        unsigned line = (gid >> 3) * 2654435761u + it * 40503u;
        unsigned off = ((line % (cold_bytes / 128)) * 128 + (gid & 7) * 16);
        u4 a = {0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu}, b, c;
        unsigned oob = 0x7ffffff0u, hoff = (gid & 255) * 16;
        if (WAITN == 2) {
            asm volatile("buffer_load_dwordx4 %0, %4, %7, 0 offen\n\t"
                         "buffer_load_dwordx4 %1, %5, %7, 0 offen\n\t"
                         "buffer_load_dwordx4 %2, %6, %8, 0 offen\n\t"
                         "s_waitcnt vmcnt(2)\n\t"
                         "ds_write_b128 %3, %0\n\t"
                         "s_waitcnt vmcnt(0) lgkmcnt(0)"
                         : "+v"(a), "=v"(b), "=v"(c) : "v"(laddr), "v"(off), "v"(oob), "v"(hoff), "s"(rc), "s"(rh) : "memory");
        } else {
            asm volatile("buffer_load_dwordx4 %0, %4, %7, 0 offen\n\t"
                         "buffer_load_dwordx4 %1, %5, %7, 0 offen\n\t"
                         "buffer_load_dwordx4 %2, %6, %8, 0 offen\n\t"
                         "s_waitcnt vmcnt(1)\n\t"
                         "ds_write_b128 %3, %0\n\t"
                         "s_waitcnt vmcnt(0) lgkmcnt(0)"
                         : "+v"(a), "=v"(b), "=v"(c) : "v"(laddr), "v"(off), "v"(oob), "v"(hoff), "s"(rc), "s"(rh) : "memory");
        }
        // the cold buffer holds  word[i] = i * 2654435761 + 12345  -> expected first word of the float4
        const unsigned expect = (off / 4) * 2654435761u + 12345u;
        const u4 snap = slot[threadIdx.x];
        if (snap.x != expect || snap.w != expect + 3u * 2654435761u || a.x != expect) ++bad;
        if (b.x | c.y) bad += 0;     // keep b / c alive
    }
    if (bad) atomicAdd(&bad_lane[lane], bad);
}

__global__ void fill(unsigned* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)i * 2654435761u + 12345u;
}

int main() {
    const size_t cold_bytes = (size_t)2040 << 20;       // < 2 GB: one buffer descriptor
    unsigned *cold, *hot;
    unsigned long long* d;
    if (hipMalloc(&cold, cold_bytes) != hipSuccess || hipMalloc(&hot, 4096) != hipSuccess || hipMalloc(&d, 64 * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, cold, cold_bytes / 4);
    hipLaunchKernelGGL(fill, dim3(1), dim3(256), 0, 0, hot, (size_t)1024);
    for (int w = 2; w >= 1; --w) {
        (void)hipMemset(d, 0, 64 * 8);
        const int iters = 2000, blocks = 4096;
        if (w == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, cold, (unsigned)cold_bytes, hot, d, iters);
        else hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, cold, (unsigned)cold_bytes, hot, d, iters);
        (void)hipDeviceSynchronize();
        unsigned long long h[64], q[4] = {0, 0, 0, 0}, tot = 0;
        (void)hipMemcpy(h, d, 64 * 8, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l) { tot += h[l]; q[l >> 4] += h[l]; }
        printf("wait vmcnt(%d) after [cold in-range, OUT-OF-RANGE, hot in-range]: %llu early releases of %.3g  by lane quarter: %llu %llu %llu %llu%s\n", w, tot,
               (double)blocks * 256 * iters, q[0], q[1], q[2], q[3], w == 1 ? "   (vmcnt(1) is only exact if the hot load cannot overtake the cold one)" : "");
    }
    return 0;
}
