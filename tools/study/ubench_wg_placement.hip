// Where do the wavefronts of a workgroup land?  (gfx950; tools only)
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/ubench_wg_placement tools/study/ubench_wg_placement.hip && ./ubench_wg_placement
// Question behind it (profiles/r05/README.md, chain-wavefront form of the exact batch kernel): a workgroup of FIVE wavefronts at
// 96 VGPRs (five per SIMD) and 40 KB of LDS should fit a CU four times (20 wavefronts, 160 KB); the batch sweep says three do.
// Every wavefront records HW_ID (SIMD, CU, SE), XCC_ID and its start time, then holds its slot for `hold` cycles; workgroups
// whose start lies within the first `hold / 2` cycles of the launch are the first generation: count them per CU, and the
// wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

template <int WAVES, int VGPRS>
__global__ __launch_bounds__(64 * WAVES) void probe(unsigned* __restrict__ ids, unsigned long long* __restrict__ t0s, long long hold) {
  extern __shared__ float lds[];
  if (VGPRS == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
  if (VGPRS == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  if (VGPRS == 80) asm volatile("v_mov_b32 v79, 0" ::: "v79");
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));
  lds[threadIdx.x] = (float)hw;
  while ((long long)(__builtin_readcyclecounter() - t0) < hold) __builtin_amdgcn_s_sleep(8);
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * WAVES + (threadIdx.x >> 6);
    ids[2 * w] = hw;
    ids[2 * w + 1] = xcc;
    t0s[w] = t0;
  }
}

template <int WAVES, int VGPRS>
void run(const char* name, int grid, size_t lds_bytes, long long hold) {
  unsigned* d_ids;
  unsigned long long* d_t0;
  const int nw = grid * WAVES;
  (void)hipMalloc(&d_ids, nw * 8);
  (void)hipMalloc(&d_t0, nw * 8);
  (void)hipFuncSetAttribute((const void*)probe<WAVES, VGPRS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((probe<WAVES, VGPRS>), dim3(grid), dim3(64 * WAVES), lds_bytes, 0, d_ids, d_t0, hold);
    (void)hipDeviceSynchronize();
  }
  std::vector<unsigned> ids(2 * nw);
  std::vector<unsigned long long> t0(nw);
  (void)hipMemcpy(ids.data(), d_ids, nw * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(t0.data(), d_t0, nw * 8, hipMemcpyDeviceToHost);
  // cycle counters of different XCDs are not aligned: first generation relative to the earliest start on the same XCD
  std::map<unsigned, unsigned long long> first;
  for (int w = 0; w < nw; ++w) {
    const unsigned x = ids[2 * w + 1] & 15;
    if (!first.count(x) || t0[w] < first[x]) first[x] = t0[w];
  }
  std::map<unsigned, int> wg_per_cu;               // key: xcc, se, sh, cu
  std::map<unsigned, std::vector<int>> simd_waves;  // waves per SIMD of that CU (first generation)
  int gen1 = 0;
  for (int b = 0; b < grid; ++b) {
    const int w0 = b * WAVES;
    const unsigned x = ids[2 * w0 + 1] & 15;
    if (t0[w0] - first[x] > (unsigned long long)hold / 2) continue;
    ++gen1;
    const unsigned hw = ids[2 * w0];
    const unsigned key = (x << 16) | (hw & 0xff00);  // CU_ID[11:8], SH_ID[12], SE_ID[15:13]
    ++wg_per_cu[key];
    auto& v = simd_waves[key];
    v.resize(4);
    for (int k = 0; k < WAVES; ++k) ++v[(ids[2 * (w0 + k)] >> 4) & 3];
  }
  std::map<int, int> hist;
  for (auto& kv : wg_per_cu) ++hist[kv.second];
  std::map<std::vector<int>, int> shapes;
  for (auto& kv : simd_waves) {
    std::vector<int> v = kv.second;
    std::sort(v.begin(), v.end());
    ++shapes[v];
  }
  printf("{\"case\": \"%s\", \"waves_per_wg\": %d, \"vgprs\": %d, \"lds\": %zu, \"grid\": %d, \"first_generation_wgs\": %d, \"cus\": %zu, \"wgs_per_cu_hist\": {", name,
         WAVES, VGPRS, lds_bytes, grid, gen1, wg_per_cu.size());
  bool c = false;
  for (auto& kv : hist) printf("%s\"%d\": %d", c ? ", " : "", kv.first, kv.second), c = true;
  printf("}, \"sorted_waves_per_simd\": {");
  c = false;
  for (auto& kv : shapes) printf("%s\"%d/%d/%d/%d\": %d", c ? ", " : "", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second), c = true;
  printf("}}\n");
  (void)hipFree(d_ids);
  (void)hipFree(d_t0);
}

int main() {
  const long long hold = 200000;  // cycles: ~85 us
  run<4, 128>("4 waves, 128 VGPRs, 40 KB (the shipped exact batch kernel)", 2048, 40224, hold);
  run<5, 96>("5 waves, 96 VGPRs, 40 KB (chain-wavefront form)", 2048, 40224, hold);
  run<5, 96>("5 waves, 96 VGPRs, 36 KB", 2048, 36 * 1024, hold);
  run<5, 96>("5 waves, 96 VGPRs, 32 KB", 2048, 32 * 1024, hold);
  run<5, 96>("5 waves, 96 VGPRs, 1 KB", 2048, 1024, hold);
  run<5, 80>("5 waves, 80 VGPRs, 40 KB", 2048, 40224, hold);
  run<5, 80>("5 waves, 80 VGPRs, 1 KB", 2048, 1024, hold);
  run<6, 80>("6 waves, 80 VGPRs, 1 KB", 2048, 1024, hold);
  run<10, 96>("10 waves, 96 VGPRs, 80 KB", 1024, 80448, hold);
  run<8, 128>("8 waves, 128 VGPRs, 80 KB", 1024, 80448, hold);
  run<5, 128>("5 waves, 128 VGPRs, 74144 B (two rounds per barrier)", 1024, 74144, hold);
  run<5, 128>("5 waves, 128 VGPRs, 65536 B", 1024, 65536, hold);
  run<5, 128>("5 waves, 128 VGPRs, 40224 B", 1024, 40224, hold);
  run<9, 80>("9 waves, 80 VGPRs, 79408 B (paired-chain form)", 1024, 79408, hold);
  run<9, 80>("9 waves, 80 VGPRs, 1 KB", 1024, 1024, hold);
  run<9, 80>("9 waves, 80 VGPRs, 64 KB", 1024, 65536, hold);
  return 0;
}
