// wave_place.hip -- where do the four waves of a 256-thread workgroup land?  Prints, for the workgroups of a few CUs of a grid
// that fills every CU with four workgroups (40 KB of LDS each, like the library's pipeline kernels): SIMD and wave-slot number
// (HW_REG_HW_ID) of waves 0..3.  Basis of wmx::pipeline_role (wmx_internal.h).
//   hipcc --offload-arch=gfx950 -O2 -o wave_place wave_place.hip && ./wave_place
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void k(unsigned *out, int spin) {
    extern __shared__ float lds[];
    if (spin < 0) lds[threadIdx.x] = 0.f;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) {}
    if ((threadIdx.x & 63) == 0) {
        out[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = hw;
        out[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = xcc & 7;
    }
}
int main() {
    const int blocks = 1024;
    unsigned *d;
    (void)hipMalloc(&d, 8 * blocks * 4);
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 39936);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 39936, 0, d, 20000);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(2 * blocks * 4);
    (void)hipMemcpy(h.data(), d, 8 * blocks * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;  // cu key -> workgroups
    int same_slot = 0, distinct_simd = 0;
    for (int b = 0; b < blocks; b++) {
        const unsigned hw = h[2 * (b * 4)], x = h[2 * (b * 4) + 1];
        cu[(x << 16) | (hw & 0xff00)].push_back(b);
        unsigned simds = 0, slot0 = hw & 15;
        bool same = true;
        for (int w = 0; w < 4; w++) {
            const unsigned v = h[2 * (b * 4 + w)];
            simds |= 1u << ((v >> 4) & 3);
            same &= (v & 15) == slot0;
        }
        same_slot += same;
        distinct_simd += simds == 0xf;
    }
    int balanced = 0;
    for (auto &e : cu) {
        unsigned roles = 0;
        for (int b : e.second) {  // SIMD that gets role 3 of workgroup b under role = (simd - slot) & 3
            for (int w = 0; w < 4; w++) {
                const unsigned v = h[2 * (b * 4 + w)];
                if ((((v >> 4) & 3) + 4 - (v & 3)) % 4 == 3) roles |= 1u << ((v >> 4) & 3);
            }
        }
        balanced += e.second.size() == 4 && roles == 0xf;
    }
    printf("{\"workgroups\": %d, \"cus_seen\": %zu, \"workgroups_with_waves_on_4_distinct_simds\": %d, \"workgroups_whose_waves_share_a_slot_number\": %d, "
           "\"cus_where_role_3_lands_on_4_distinct_simds\": %d}\n", blocks, cu.size(), distinct_simd, same_slot, balanced);
    int shown = 0;
    for (auto &e : cu) {
        if (shown++ >= 3) break;
        printf("cu %05x:", e.first);
        for (int b : e.second) {
            printf("  wg %d [", b);
            for (int w = 0; w < 4; w++) printf("%s%u/%u", w ? " " : "", (h[2 * (b * 4 + w)] >> 4) & 3, h[2 * (b * 4 + w)] & 15);
            printf("]");
        }
        printf("   (simd/slot of waves 0..3)\n");
    }
    return 0;
}
