// Do fp32 MFMA and fp32 VALU overlap on one SIMD?  (Follow-up of fused_sweep.hip: consumers + producers cost the SUM of their times.)
// One workgroup per CU, 8 waves = 2 per SIMD; wave w runs stream kind[w & 1]:  M = v_mfma_f32_16x16x4_f32 chain (4 independent accumulators),
// B = v_mfma_f32_16x16x16_bf16 chain, V = v_fma_f32 chain (8 independent accumulators), - = idle.  Reported: time of each pairing against the
// two streams alone.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_valu_corun.hip -o mfma_valu_corun
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__device__ inline float stream(int n, float seed) {
  if constexpr (KIND == 0) return 0.f;
  if constexpr (KIND == 1) {  // fp32 MFMA
    floatx4 a0{0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const float x = seed, y = seed * 0.5f;
    for (int i = 0; i < n; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
    }
    return a0[0] + a1[1] + a2[2] + a3[3];
  }
  if constexpr (KIND == 2) {  // bf16 MFMA, same issue interval per instruction class is NOT assumed: counted separately
    floatx4 a0{0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    bf16x8 x, y;
    for (int k = 0; k < 8; ++k) { x[k] = (__bf16)seed; y[k] = (__bf16)(seed * 0.5f); }
    for (int i = 0; i < n; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a3, 0, 0, 0);
    }
    return a0[0] + a1[1] + a2[2] + a3[3];
  }
  if constexpr (KIND == 3) {  // fp32 VALU: 16 FMAs per iteration on 8 independent chains
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
  if constexpr (KIND == 4) {  // integer VALU (no FMA lanes): v_add_u32 / v_xor chains
    unsigned v[8];
    for (int k = 0; k < 8; ++k) v[k] = (unsigned)seed + k;
    for (int i = 0; i < n; ++i) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[k]) : "v"(0x9e3779b9u));
    }
    unsigned s = 0;
    for (int k = 0; k < 8; ++k) s ^= v[k];
    return (float)s;
  }
  return 0.f;
}

template <int KA, int KB>
__global__ __launch_bounds__(512) void k(float *out, int na, int nb, int pair) {
  const int wave = threadIdx.x >> 6;
  float r;
  // a workgroup's waves go to the SIMDs in a cyclic order (MI355X_MICROARCH.md: 0 -> 2 -> 1 -> 3 from a varying start), so waves w and w + 4 share a
  // SIMD: kind A on waves 0-3 and kind B on waves 4-7 puts ONE OF EACH on every SIMD.  (The first version of this file split by wave & 1, which
  // put the two kinds on different SIMDs -- its "0.94 x max" said nothing about sharing; `pair` = 0 reproduces it for comparison.)
  const bool second = pair ? wave >= 4 : (wave & 1);
  if (second) r = stream<KB>(nb, 1.f + threadIdx.x * 1e-3f);
  else r = stream<KA>(na, 1.f + threadIdx.x * 1e-3f);
  out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int KA, int KB>
static float run(int na, int nb, int pair = 1) {
  static float *d = nullptr;
  if (!d) hipMalloc(&d, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<KA, KB><<<256, 512>>>(d, na / 8, nb / 8, pair);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<KA, KB><<<256, 512>>>(d, na, nb, pair);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  // 8 waves per workgroup place TWO waves on each SIMD: with (wave & 1) both kinds land on every SIMD only if the hardware's wave->SIMD order
  // pairs an even with an odd wave; the A|- and -|B rows show what each kind costs alone in the same launch shape.
  const int NM = 20000, NV = 20000;  // iterations: 4 MFMAs resp. 16 VALU each
  const float m = run<1, 0>(NM, 0), b = run<2, 0>(NM, 0), v = run<0, 3>(0, NV), iu = run<0, 4>(0, NV);
  printf("alone:  f32 MFMA %.3f ms (%.1f cyc/MFMA/wave @2.4GHz)   bf16 MFMA %.3f ms (%.1f)   f32 VALU fma %.3f ms (%.2f cyc/instr)   u32 VALU add %.3f ms (%.2f)\n", m, m * 2.4e6 / (4.0 * NM), b,
         b * 2.4e6 / (4.0 * NM), v, v * 2.4e6 / (16.0 * NV), iu, iu * 2.4e6 / (16.0 * NV));
  const float mv = run<1, 3>(NM, NV), bv = run<2, 3>(NM, NV), mi = run<1, 4>(NM, NV), mm = run<1, 1>(NM, NM), vv = run<3, 3>(NV, NV), mb = run<1, 2>(NM, NM);
  printf("f32 MFMA | f32 VALU : %.3f ms = %.2f x max, %.2f x sum\n", mv, mv / fmaxf(m, v), mv / (m + v));
  printf("bf16 MFMA | f32 VALU: %.3f ms = %.2f x max, %.2f x sum\n", bv, bv / fmaxf(b, v), bv / (b + v));
  printf("f32 MFMA | u32 VALU : %.3f ms = %.2f x max, %.2f x sum\n", mi, mi / fmaxf(m, iu), mi / (m + iu));
  printf("f32 MFMA | f32 MFMA : %.3f ms = %.2f x one\n", mm, mm / m);
  printf("f32 VALU | f32 VALU : %.3f ms = %.2f x one\n", vv, vv / v);
  printf("f32 MFMA | bf16 MFMA: %.3f ms = %.2f x max, %.2f x sum\n", mb, mb / fmaxf(m, b), mb / (m + b));
  const float mv0 = run<1, 3>(NM, NV, 0), m0 = run<1, 0>(NM, 0, 0), v0 = run<0, 3>(0, NV, 0);
  printf("(kinds on DIFFERENT SIMDs, wave & 1 split) f32 MFMA %.3f, f32 VALU %.3f, both %.3f ms = %.2f x max\n", m0, v0, mv0, mv0 / fmaxf(m0, v0));
  // the ratio at which a sweep would run beside conv0: ~1 VALU instruction per 1.4 MFMA cycles -- lengthen the VALU stream until both streams alone take the same time
  const int NV2 = (int)(NV * m / v);
  const float v2 = run<0, 3>(0, NV2), mv2 = run<1, 3>(NM, NV2);
  printf("balanced: f32 MFMA %.3f ms, f32 VALU x%.2f %.3f ms, both on every SIMD %.3f ms = %.2f x max, %.2f x sum\n", m, (double)NV2 / NV, v2, mv2, mv2 / fmaxf(m, v2), mv2 / (m + v2));
  return 0;
}
