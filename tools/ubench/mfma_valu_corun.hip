// Do fp32 MFMA and fp32 VALU overlap on one SIMD?  Round 6 rewrite (round 5's table had a VALU-alone row of 12.8 cycles per
// instruction and a "two VALU waves finish faster than one" row, so its "MFMA || VALU = 1.00 x the sum" stood on a wrong baseline).
// What is different:
//   * every wave reads the shader clock (s_memtime) before and after its stream and its SIMD from HW_REG_HW_ID: the table is in CYCLES PER INSTRUCTION
//     per wave, and the placement the experiment relies on (which waves share a SIMD) is printed, not assumed;
//   * every configuration is run after a warm-up launch of the SAME configuration and timed over 5 launches (median);
//   * VALU streams run with 1, 2 and 4 waves per SIMD explicitly, with 8 and with 16 independent chains, as v_fma_f32 and as v_pk_fma_f32;
//   * the mixed rows state what each kind achieves BESIDE the other (cycles per instruction of the MFMA waves and of the VALU waves separately).
// One workgroup per CU (256 workgroups); a workgroup of 4 * n waves puts n waves on every SIMD.  role[w]: 0 idle, 1 fp32 MFMA 16x16x4, 2 bf16 MFMA
// 16x16x32, 3 v_fma_f32 x 8 chains, 4 v_fma_f32 x 16 chains, 5 v_pk_fma_f32 x 8 chains.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_valu_corun.hip -o mfma_valu_corun
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Roles { int r[16]; };

__device__ inline float stream(int kind, int n, float seed) {
  if (kind == 1) {
    floatx4 a0{0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const float x = seed, y = seed * 0.5f;
    for (int i = 0; i < n; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
    }
    return a0[0] + a1[1] + a2[2] + a3[3];
  }
  if (kind == 2) {
    floatx4 a0{0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    bf16x8 x, y;
    for (int k = 0; k < 8; ++k) { x[k] = (__bf16)seed; y[k] = (__bf16)(seed * 0.5f); }
    for (int i = 0; i < n; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a3, 0, 0, 0);
    }
    return a0[0] + a1[1] + a2[2] + a3[3];
  }
  if (kind == 3) {  // 16 v_fma_f32 per iteration on 8 independent chains
    float v[8];
    for (int k = 0; k < 8; ++k) v[k] = seed + k;
    const float m = 1.0001f, c = 0.5f;
    for (int i = 0; i < n; ++i) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(m), "v"(c));
    }
    float s = 0;
    for (int k = 0; k < 8; ++k) s += v[k];
    return s;
  }
  if (kind == 4) {  // 16 v_fma_f32 per iteration on 16 independent chains
    float v[16];
    for (int k = 0; k < 16; ++k) v[k] = seed + k;
    const float m = 1.0001f, c = 0.5f;
    for (int i = 0; i < n; ++i) {
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(m), "v"(c));
    }
    float s = 0;
    for (int k = 0; k < 16; ++k) s += v[k];
    return s;
  }
  if (kind == 5) {  // 16 v_pk_fma_f32 per iteration on 8 independent chains (two fp32 FMAs per lane per instruction)
    floatx2 v[8];
    for (int k = 0; k < 8; ++k) v[k] = floatx2{seed + k, seed - k};
    const floatx2 m{1.0001f, 0.9999f}, c{0.5f, 0.25f};
    for (int i = 0; i < n; ++i) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(m), "v"(c));
    }
    float s = 0;
    for (int k = 0; k < 8; ++k) s += v[k][0] + v[k][1];
    return s;
  }
  return 0.f;
}

// per wave: out[0] = cycles (s_memtime), out[1] = SIMD id
__global__ void k(const Roles roles, int n, float *sink, unsigned long long *cyc, int *simd) {
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int kind = roles.r[wave];
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const float r = stream(kind, n, 1.f + threadIdx.x * 1e-3f);
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    cyc[blockIdx.x * nw + wave] = t1 - t0;
    simd[blockIdx.x * nw + wave] = (int)((hw >> 4) & 3);
  }
}

static const char *kname(int k) { static const char *n[] = {"-", "f32MFMA", "bf16MFMA", "fma x8", "fma x16", "pk_fma x8"}; return n[k]; }

struct Result { float ms; double cyc_per_instr[6]; int same_simd_pairs, waves; };

static Result run(const std::vector<int> &roles, int n) {
  const int nw = (int)roles.size(), WG = 256;
  static float *sink = nullptr; static unsigned long long *cyc = nullptr; static int *simd = nullptr;
  if (!sink) { hipMalloc(&sink, (size_t)WG * 1024 * 4); hipMalloc(&cyc, WG * 16 * 8); hipMalloc(&simd, WG * 16 * 4); }
  Roles R; memset(&R, 0, sizeof R);
  for (int i = 0; i < nw; ++i) R.r[i] = roles[i];
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) k<<<WG, 64 * nw>>>(R, n, sink, cyc, simd);  // warm-up: the same configuration, the same length
  hipDeviceSynchronize();
  std::vector<float> t;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    k<<<WG, 64 * nw>>>(R, n, sink, cyc, simd);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  std::vector<unsigned long long> hc(WG * nw); std::vector<int> hs(WG * nw);
  hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(hs.data(), simd, hs.size() * 4, hipMemcpyDeviceToHost);
  Result r{}; r.ms = t[2]; r.waves = nw;
  double sum[6] = {0}; int cnt[6] = {0};
  const int per_iter[6] = {1, 4, 4, 16, 16, 16};
  for (int b = 0; b < WG; ++b)
    for (int w = 0; w < nw; ++w) { sum[roles[w]] += (double)hc[b * nw + w] / ((double)n * per_iter[roles[w]]); cnt[roles[w]]++; }
  for (int q = 0; q < 6; ++q) r.cyc_per_instr[q] = cnt[q] ? sum[q] / cnt[q] : 0;
  // placement: in how many workgroups does every SIMD hold exactly nw / 4 waves, and (for mixed rows) one wave of each of the first two distinct kinds
  int ok = 0;
  for (int b = 0; b < WG; ++b) {
    int per[4] = {0, 0, 0, 0}, kinds[4][6] = {{0}};
    for (int w = 0; w < nw; ++w) { per[hs[b * nw + w]]++; kinds[hs[b * nw + w]][roles[w]]++; }
    bool even = true;
    for (int s = 0; s < 4; ++s) {
      even &= per[s] * 4 == nw;
      for (int q = 0; q < 6; ++q) if (cnt[q]) even &= kinds[s][q] * 4 * WG == cnt[q];  // every kind spread evenly over the SIMDs
    }
    ok += even;
  }
  r.same_simd_pairs = ok;
  return r;
}

static void row(const char *label, const std::vector<int> &roles, int n) {
  const Result r = run(roles, n);
  printf("%-44s %7.3f ms |", label, r.ms);
  bool seen[6] = {false};
  for (int k2 : roles) if (k2 && !seen[k2]) { seen[k2] = true; printf(" %s %.2f cyc/instr/wave", kname(k2), r.cyc_per_instr[k2]); }
  printf(" | even placement in %d of 256 workgroups\n", r.same_simd_pairs);
}

int main() {
  const int N = 20000;
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device %s, %d CUs, shader clock %d MHz (s_memtime counts at its own rate: the cyc/instr columns are s_memtime ticks)\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
  // calibrate s_memtime against wall time with a long MFMA stream
  { const Result r = run({1, 1, 1, 1}, N); printf("calibration: 4 MFMA waves per workgroup, %.3f ms for %d x 4 MFMAs = %.1f ns per MFMA; s_memtime says %.2f ticks per MFMA -> %.3f ticks per ns\n", r.ms, N, r.ms * 1e6 / (4.0 * N), r.cyc_per_instr[1], r.cyc_per_instr[1] / (r.ms * 1e6 / (4.0 * N))); }
  printf("---- one kind, n waves per SIMD (workgroup of 4 n waves)\n");
  row("f32 MFMA x1 per SIMD", {1, 1, 1, 1}, N);
  row("f32 MFMA x2 per SIMD", {1, 1, 1, 1, 1, 1, 1, 1}, N);
  row("bf16 MFMA x1 per SIMD", {2, 2, 2, 2}, N);
  row("v_fma_f32 (8 chains) x1 per SIMD", {3, 3, 3, 3}, N);
  row("v_fma_f32 (8 chains) x2 per SIMD", {3, 3, 3, 3, 3, 3, 3, 3}, N);
  row("v_fma_f32 (8 chains) x4 per SIMD", {3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3}, N);
  row("v_fma_f32 (16 chains) x1 per SIMD", {4, 4, 4, 4}, N);
  row("v_fma_f32 (16 chains) x2 per SIMD", {4, 4, 4, 4, 4, 4, 4, 4}, N);
  row("v_pk_fma_f32 (8 chains) x1 per SIMD", {5, 5, 5, 5}, N);
  row("v_pk_fma_f32 (8 chains) x2 per SIMD", {5, 5, 5, 5, 5, 5, 5, 5}, N);
  printf("---- one wave idle, one busy per SIMD (the launch shape of the mixed rows)\n");
  row("f32 MFMA + idle", {1, 1, 1, 1, 0, 0, 0, 0}, N);
  row("v_fma_f32 (8 chains) + idle", {3, 3, 3, 3, 0, 0, 0, 0}, N);
  printf("---- two kinds on EVERY SIMD (waves w and w + 4 of a workgroup)\n");
  row("f32 MFMA + v_fma_f32 (8 chains)", {1, 1, 1, 1, 3, 3, 3, 3}, N);
  row("f32 MFMA + v_fma_f32 (16 chains)", {1, 1, 1, 1, 4, 4, 4, 4}, N);
  row("f32 MFMA + v_pk_fma_f32", {1, 1, 1, 1, 5, 5, 5, 5}, N);
  row("f32 MFMA + 2 x v_fma_f32 (8 chains)", {1, 1, 1, 1, 3, 3, 3, 3, 3, 3, 3, 3}, N);
  row("f32 MFMA + 3 x v_fma_f32 (8 chains)", {1, 1, 1, 1, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3}, N);
  row("bf16 MFMA + v_fma_f32 (8 chains)", {2, 2, 2, 2, 3, 3, 3, 3}, N);
  row("bf16 MFMA + 2 x v_fma_f32 (8 chains)", {2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3}, N);
  row("f32 MFMA + bf16 MFMA", {1, 1, 1, 1, 2, 2, 2, 2}, N);
  printf("---- the two kinds on DIFFERENT SIMDs (waves alternate)\n");
  row("f32 MFMA / v_fma_f32 alternating waves", {1, 3, 1, 3, 1, 3, 1, 3}, N);
  return 0;
}
