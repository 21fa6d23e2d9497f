// Probe: does a VALU instruction that overwrites the ADDRESS (or DATA) VGPR of a ds_write_b64 right behind it race with the LDS
// store's operand fetch on gfx950?  (tools/diag_wgrad_race.py: in ~1-3 % of backward passes the first pws_wgrad_staged_kernel<2,2,2,..>
// launch produced an fp32-visible error confined to the operand rows staged by lanes 48-63 of the waves that run ahead of the MFMA
// waves; hipcc's code there is  v_add_u32 vA, 0, vB / ds_write_b64 vA, .. / v_add_u32 vA, sX, vB / ds_write_b64 vA, .. / ...)
//
// One "victim" wave per workgroup repeats, with the instruction pair pinned by inline asm:
//      ds_write_b64 vA, vD        (slot of this iteration, tagged data)
//      v_add_u32    vA, vA, vS    (MODE 1: overwrite the address register in the very next issue slot)
//      v_mov_b32    vD, garbage   (MODE 2: overwrite the data register instead)
// while the other waves of the workgroup keep the LDS pipe busy with ds_read_b128 bursts (back-pressure).  After a barrier every slot
// is checked: a lane whose store went to the NEXT slot's address, or carried the garbage, is counted per lane.
//   hipcc --offload-arch=gfx950 -O3 -o lds_war_probe lds_war_probe.hip && ./lds_war_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define SLOTS 64              // 64 slots x 64 lanes x 8 B = 32 KB
#define ROUNDS 200

template <int MODE>
__global__ __launch_bounds__(512) void probe(unsigned long long* bad_lane, int hammer) {
    __shared__ __attribute__((aligned(16))) unsigned sh[SLOTS * 64 * 2 + 16384];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned* noise = sh + SLOTS * 64 * 2;
    for (int i = tid; i < 16384; i += 512) noise[i] = i;
    __syncthreads();
    float sink = 0.f;
    for (int round = 0; round < ROUNDS; ++round) {
        for (int i = tid; i < SLOTS * 64 * 2; i += 512) sh[i] = 0xdeadbeefu;
        __syncthreads();
        if (wave == 0) {
            unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)sh + lane * 8;
            const unsigned step = 64 * 8;
#pragma unroll 1
            for (int s = 0; s < SLOTS; ++s) {
                unsigned d0 = 0x10000u * (round & 0xff) + s * 64 + lane, d1 = ~d0;
                if (MODE == 1) {
                    asm volatile("ds_write_b64 %0, %1\n\tv_add_u32 %0, %0, %2"
                                 : "+v"(addr) : "v"((unsigned long long)d0 | ((unsigned long long)d1 << 32)), "v"(step) : "memory");
                } else if (MODE == 2) {
                    unsigned long long dd = (unsigned long long)d0 | ((unsigned long long)d1 << 32);
                    asm volatile("ds_write_b64 %0, %1\n\tv_mov_b64 %1, 0" : "+v"(addr), "+v"(dd) : : "memory");
                    addr += step;
                } else {
                    asm volatile("ds_write_b64 %0, %1\n\ts_nop 7\n\tv_add_u32 %0, %0, %2"
                                 : "+v"(addr) : "v"((unsigned long long)d0 | ((unsigned long long)d1 << 32)), "v"(step) : "memory");
                }
            }
        } else if (hammer) {
            // back-pressure: 16-byte reads from the other waves
            const uint4* np = reinterpret_cast<const uint4*>(noise);
#pragma unroll 4
            for (int i = 0; i < 256; ++i) {
                const uint4 v = np[(tid * 7 + i * 61) & 4095];
                sink += __uint_as_float((v.x ^ v.y) + (v.z ^ v.w));
            }
        }
        __syncthreads();
        // check
        for (int i = tid; i < SLOTS * 64; i += 512) {
            const int s = i >> 6, l = i & 63;
            const unsigned d0 = 0x10000u * (round & 0xff) + s * 64 + l;
            if (sh[i * 2] != d0 || sh[i * 2 + 1] != ~d0) atomicAdd(&bad_lane[l], 1ULL);
        }
        __syncthreads();
    }
    if (sink == 12345.678f) bad_lane[64] = 1;
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 65 * 8);
    for (int hammer = 0; hammer < 2; ++hammer)
        for (int mode = 0; mode < 3; ++mode) {
            hipMemset(d, 0, 65 * 8);
            for (int rep = 0; rep < 20; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(1024), dim3(512), 0, 0, d, hammer);
                if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(1024), dim3(512), 0, 0, d, hammer);
                if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(1024), dim3(512), 0, 0, d, hammer);
            }
            hipDeviceSynchronize();
            unsigned long long h[65];
            hipMemcpy(h, d, 65 * 8, hipMemcpyDeviceToHost);
            unsigned long long tot = 0, q[4] = {0, 0, 0, 0};
            for (int l = 0; l < 64; ++l) { tot += h[l]; q[l >> 4] += h[l]; }
            printf("hammer %d  mode %d (%s): %llu bad stores of %.3g  by lane quarter: %llu %llu %llu %llu\n", hammer, mode,
                   mode == 0 ? "s_nop 7 between store and overwrite" : mode == 1 ? "address VGPR overwritten in the next slot" : "data VGPRs overwritten in the next slots",
                   tot, 20.0 * 1024 * ROUNDS * SLOTS * 64, q[0], q[1], q[2], q[3]);
        }
    return 0;
}
