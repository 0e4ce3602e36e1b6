// Where do the waves of small workgroups land?  128-thread workgroups (2 waves, like the role-specialised env kernel), each wave records
// HW_REG_HW_ID (gfx9: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...) and XCC_ID; the host prints, per grid size,
// how many CUs host 1 / 2 / ... workgroups and how often the FIRST waves (the physics waves) of two workgroups of one CU share a SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/hwid_probe.hip -o /tmp/hwid_probe && /tmp/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)
__global__ void probe(unsigned* out, int spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // stay resident for a while so that the whole grid is co-resident like the real kernel's
    long long t0 = clock64();
    while (clock64() - t0 < spin) { }
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
}
__global__ void probe3(unsigned* out, int spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = clock64();
    while (clock64() - t0 < spin) { }
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 3 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 3 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
}
int main() {
    unsigned* d; CK(hipMalloc(&d, 4096 * 2 * 2 * 4));
    for (int grid : {256, 512, 1024, 2048}) {
        hipLaunchKernelGGL(probe, dim3(grid), dim3(128), 8192, 0, d, 200000);
        CK(hipDeviceSynchronize());
        std::vector<unsigned> h(grid * 4);
        CK(hipMemcpy(h.data(), d, grid * 16, hipMemcpyDeviceToHost));
        std::map<unsigned, std::vector<int>> cu_first, cu_second;   // key: xcc | se | sh | cu -> simd ids of wave 0 / wave 1 of its workgroups
        int same_wg_same_simd = 0;
        for (int b = 0; b < grid; ++b) {
            unsigned hw0 = h[(b * 2) * 2], x0 = h[(b * 2) * 2 + 1] & 0xF, hw1 = h[(b * 2 + 1) * 2];
            unsigned key = (x0 << 16) | (((hw0 >> 13) & 7) << 8) | (((hw0 >> 12) & 1) << 4) | ((hw0 >> 8) & 0xF);
            cu_first[key].push_back((hw0 >> 4) & 3);
            cu_second[key].push_back((hw1 >> 4) & 3);
            if (((hw0 >> 4) & 3) == ((hw1 >> 4) & 3)) ++same_wg_same_simd;
        }
        std::map<int, int> per_cu; int share = 0, pairs = 0;
        for (auto& kv : cu_first) {
            per_cu[(int)kv.second.size()]++;
            int cnt[4] = {0, 0, 0, 0};
            for (int s : kv.second) cnt[s]++;
            for (int s = 0; s < 4; ++s) if (cnt[s] > 1) share += cnt[s] - 1;
            pairs += (int)kv.second.size() - 1;
        }
        printf("grid %4d: %zu CUs used;", grid, cu_first.size());
        for (auto& kv : per_cu) printf("  %d CUs with %d workgroups", kv.second, kv.first);
        printf(";  waves 0 and 1 of a workgroup on the same SIMD: %d;  first waves sharing a SIMD with another first wave of their CU: %d of %d extra workgroups\n",
               same_wg_same_simd, share, pairs);
        if (grid == 512) {
            int shown = 0;
            for (auto& kv : cu_first) { if (shown++ >= 6) break; printf("   cu %06x: first-wave simds", kv.first); for (int s : kv.second) printf(" %d", s); printf(" | second-wave simds"); for (int s : cu_second[kv.first]) printf(" %d", s); printf("\n"); }
            printf("   blocks 0..23 -> xcc:cu:simd0/simd1 :");
            for (int b = 0; b < 24; ++b) printf(" %u:%u:%u/%u", h[b * 4 + 1] & 0xF, ((h[b * 4] >> 13) & 7) * 100 + ((h[b * 4] >> 12) & 1) * 16 + ((h[b * 4] >> 8) & 0xF), (h[b * 4] >> 4) & 3, (h[b * 4 + 2] >> 4) & 3);
            printf("\n");
        }
    }
    // 3-wave workgroups (the split kernel: a physics wave + two observation waves per env; c4 shard = 1024 of them): waves per SIMD and
    // first (physics) waves per SIMD, worst CU and average spread
    for (int grid : {512, 1024, 2048}) {
        hipLaunchKernelGGL(probe3, dim3(grid), dim3(192), 8192, 0, d, 200000);
        CK(hipDeviceSynchronize());
        std::vector<unsigned> h(grid * 6);
        CK(hipMemcpy(h.data(), d, grid * 24, hipMemcpyDeviceToHost));
        std::map<unsigned, std::vector<int>> tot, first;
        for (int b = 0; b < grid; ++b)
            for (int w = 0; w < 3; ++w) {
                unsigned hw = h[(b * 3 + w) * 2], x = h[(b * 3 + w) * 2 + 1] & 0xF;
                unsigned key = (x << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF);
                if (tot[key].empty()) { tot[key].assign(4, 0); first[key].assign(4, 0); }
                tot[key][(hw >> 4) & 3]++;
                if (w == 0) first[key][(hw >> 4) & 3]++;
            }
        int worst_tot = 0, worst_first = 0; double spread = 0;
        for (auto& kv : tot) {
            int mx = 0, mn = 99, fx = 0;
            for (int sdx = 0; sdx < 4; ++sdx) { mx = kv.second[sdx] > mx ? kv.second[sdx] : mx; mn = kv.second[sdx] < mn ? kv.second[sdx] : mn; fx = first[kv.first][sdx] > fx ? first[kv.first][sdx] : fx; }
            worst_tot = mx > worst_tot ? mx : worst_tot; worst_first = fx > worst_first ? fx : worst_first; spread += mx - mn;
        }
        auto& ex = *tot.begin();
        printf("3-wave workgroups, grid %4d: %zu CUs; most waves on one SIMD %d, most PHYSICS waves on one SIMD %d, mean (max - min) waves per SIMD %.2f; e.g. CU0 waves per SIMD %d %d %d %d, physics %d %d %d %d\n",
               grid, tot.size(), worst_tot, worst_first, spread / tot.size(), ex.second[0], ex.second[1], ex.second[2], ex.second[3],
               first[ex.first][0], first[ex.first][1], first[ex.first][2], first[ex.first][3]);
    }
    return 0;
}
