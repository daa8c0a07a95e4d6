// lab: do PACKED fp32 VALU instructions (v_pk_mul_f32 / v_pk_fma_f32) give wrong results while a second PROCESS uses the
// same GPU?  (LAB_NOTES section 15: the attention walk's `m` rows came out with the LOW half of one packed product
// zeroed in lanes 48..63, only next to a neighbour process, and not at all in a build without packed fp32 ops.)
//   v_pk_*: seven forms of the packed instructions (mul / fma / add, with and without op_sel / op_sel_hi; see k_pk) in a
//           long register-only loop, every result checked against two plain scalar-VALU instructions; mismatches are
//           counted per 16-lane quarter and half, one wrong product is printed with its operands, and per wave whether
//           its HW_ID / XCC_ID changed during the kernel (a wave that moved was context-saved and restored)
//   walk  : compiler-generated code of the attention walk's shape (row loads, fast sigmoid, 8-lane head sums, row times
//           head factor, row store), launched again and again on the same inputs and compared with its first output
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast tools/lab/pk_micro.hip -o tools/lab/pk_micro
//   tools/lab/pk_micro [launches per mode] [twostream]      (twostream: two co-resident kernels of ONE process instead)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
#define GETREG(id) ((31 << 11) | (id))  // whole 32-bit hardware register `id`

struct Stats {
  unsigned long long bad[4][2];   // [16-lane quarter][half]
  unsigned long long waves, moved, bad_in_moved, bad_in_unmoved;
  unsigned sample_taken;
  float sample[6];                // x.lo x.hi a.lo a.hi | packed low result | scalar low result   of one wrong product
};

template <int MODE>
__global__ __launch_bounds__(256) void k_pk(const float* __restrict__ xin, Stats* st, int iters) {
  const int lane = threadIdx.x & 63;
  const unsigned hw0 = __builtin_amdgcn_s_getreg(GETREG(4)), xc0 = __builtin_amdgcn_s_getreg(GETREG(20));
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  v2f x = {xin[2 * gid], xin[2 * gid + 1]};
  v2f a = {0.37f + 1e-3f * (float)lane, 1.25f};
  unsigned bad_lo = 0, bad_hi = 0;
  for (int k = 0; k < iters; ++k) {
    v2f d;
    float lo, hi;
    // MODE 0: mul op_sel:[0,1]     lo = x.lo a.hi   hi = x.hi a.hi      (the form found in the attention walk)
    //      1: mul                  lo = x.lo a.lo   hi = x.hi a.hi
    //      2: fma op_sel:[0,1,0]   lo = x.lo a.hi + x.lo, hi = x.hi a.hi + x.hi
    //      3: mul op_sel:[1,0]     lo = x.hi a.lo   hi = x.hi a.hi
    //      4: mul op_sel_hi:[1,0]  lo = x.lo a.lo   hi = x.hi a.lo
    //      5: add op_sel:[0,1]     lo = x.lo + a.hi hi = x.hi + a.hi
    //      6: mul op_sel:[1,1]     lo = x.hi a.hi   hi = x.hi a.hi
    if (MODE == 0) {
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]\n s_nop 2" : "=v"(d) : "v"(x), "v"(a));
      asm volatile("v_mul_f32 %0, %1, %2\n s_nop 0" : "=v"(lo) : "v"(x.x), "v"(a.y));
      asm volatile("v_mul_f32 %0, %1, %2\n s_nop 0" : "=v"(hi) : "v"(x.y), "v"(a.y));
    } else if (MODE == 1) {
      asm volatile("v_pk_mul_f32 %0, %1, %2\n s_nop 2" : "=v"(d) : "v"(x), "v"(a));
      asm volatile("v_mul_f32 %0, %1, %2\n s_nop 0" : "=v"(lo) : "v"(x.x), "v"(a.x));
      asm volatile("v_mul_f32 %0, %1, %2\n s_nop 0" : "=v"(hi) : "v"(x.y), "v"(a.y));
    } else if (MODE == 2) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,1,0]\n s_nop 2" : "=v"(d) : "v"(x), "v"(a));
      asm volatile("v_fma_f32 %0, %1, %2, %1\n s_nop 0" : "=v"(lo) : "v"(x.x), "v"(a.y));
      asm volatile("v_fma_f32 %0, %1, %2, %1\n s_nop 0" : "=v"(hi) : "v"(x.y), "v"(a.y));
    } else if (MODE == 3) {
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]\n s_nop 2" : "=v"(d) : "v"(x), "v"(a));
      asm volatile("v_mul_f32 %0, %1, %2\n s_nop 0" : "=v"(lo) : "v"(x.y), "v"(a.x));
      asm volatile("v_mul_f32 %0, %1, %2\n s_nop 0" : "=v"(hi) : "v"(x.y), "v"(a.y));
    } else if (MODE == 4) {
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n s_nop 2" : "=v"(d) : "v"(x), "v"(a));
      asm volatile("v_mul_f32 %0, %1, %2\n s_nop 0" : "=v"(lo) : "v"(x.x), "v"(a.x));
      asm volatile("v_mul_f32 %0, %1, %2\n s_nop 0" : "=v"(hi) : "v"(x.y), "v"(a.x));
    } else if (MODE == 5) {
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]\n s_nop 2" : "=v"(d) : "v"(x), "v"(a));
      asm volatile("v_add_f32 %0, %1, %2\n s_nop 0" : "=v"(lo) : "v"(x.x), "v"(a.y));
      asm volatile("v_add_f32 %0, %1, %2\n s_nop 0" : "=v"(hi) : "v"(x.y), "v"(a.y));
    } else {
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1]\n s_nop 2" : "=v"(d) : "v"(x), "v"(a));
      asm volatile("v_mul_f32 %0, %1, %2\n s_nop 0" : "=v"(lo) : "v"(x.y), "v"(a.y));
      asm volatile("v_mul_f32 %0, %1, %2\n s_nop 0" : "=v"(hi) : "v"(x.y), "v"(a.y));
    }
    if (__float_as_uint(d.x) != __float_as_uint(lo) && atomicCAS(&st->sample_taken, 0u, 1u) == 0u) {
      st->sample[0] = x.x; st->sample[1] = x.y; st->sample[2] = a.x; st->sample[3] = a.y; st->sample[4] = d.x; st->sample[5] = lo;
    }
    bad_lo += __float_as_uint(d.x) != __float_as_uint(lo);
    bad_hi += __float_as_uint(d.y) != __float_as_uint(hi);
    x.x = x.x * 0.999f + 1e-3f;  // operands change every pass (all normal numbers)
    x.y = x.y * 1.001f - 1e-3f;
    a.y = a.y * 0.9999f + 1e-4f;
  }
  const unsigned hw1 = __builtin_amdgcn_s_getreg(GETREG(4)), xc1 = __builtin_amdgcn_s_getreg(GETREG(20));
  const bool moved = hw0 != hw1 || xc0 != xc1;
  if (bad_lo) atomicAdd(&st->bad[lane >> 4][0], (unsigned long long)bad_lo);
  if (bad_hi) atomicAdd(&st->bad[lane >> 4][1], (unsigned long long)bad_hi);
  if (bad_lo + bad_hi) atomicAdd(moved ? &st->bad_in_moved : &st->bad_in_unmoved, (unsigned long long)(bad_lo + bad_hi));
  if (lane == 0) {
    atomicAdd(&st->waves, 1ull);
    if (moved) atomicAdd(&st->moved, 1ull);
  }
}

__device__ __forceinline__ float sig(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
// one wave per target row i: for each of its DEG edges e: k, v of the source row, pk, pv of the edge; a = silu(sum over
// the 8 lanes of a head of q k silu(pk)) * C_e; m_e = v silu(pv) a; A_i = sum_e m_e      (H = 256: four floats per lane)
__global__ __launch_bounds__(256) void k_walk(int N, int DEG, const float* __restrict__ qkv, const float* __restrict__ pe,
                                              const float* __restrict__ cs, const int* __restrict__ src,
                                              float* __restrict__ m, float* __restrict__ A) {
  const int lane = threadIdx.x & 63;
  for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < N; i += gridDim.x * 4) {
    const float4 q = *(const float4*)(qkv + (size_t)i * 768 + lane * 4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < DEG; ++t) {
      const int e = __builtin_amdgcn_readfirstlane(i * DEG + t);
      const int j = __builtin_amdgcn_readfirstlane(src[e]);
      const float C = cs[e];
      const float4 k = *(const float4*)(qkv + (size_t)j * 768 + 256 + lane * 4);
      const float4 v = *(const float4*)(qkv + (size_t)j * 768 + 512 + lane * 4);
      const float4 pk = *(const float4*)(pe + (size_t)e * 768 + lane * 4);
      const float4 pv = *(const float4*)(pe + (size_t)e * 768 + 256 + lane * 4);
      float part = q.x * k.x * (pk.x * sig(pk.x)) + q.y * k.y * (pk.y * sig(pk.y)) + q.z * k.z * (pk.z * sig(pk.z)) +
                   q.w * k.w * (pk.w * sig(pk.w));
      part += __shfl_xor(part, 1, 64);
      part += __shfl_xor(part, 2, 64);
      part += __shfl_xor(part, 4, 64);
      const float a = part * sig(part) * C;
      float4 mv;
      mv.x = v.x * (pv.x * sig(pv.x)) * a;
      mv.y = v.y * (pv.y * sig(pv.y)) * a;
      mv.z = v.z * (pv.z * sig(pv.z)) * a;
      mv.w = v.w * (pv.w * sig(pv.w)) * a;
      acc.x += mv.x; acc.y += mv.y; acc.z += mv.z; acc.w += mv.w;
      *(float4*)(m + (size_t)e * 256 + lane * 4) = mv;
    }
    *(float4*)(A + (size_t)i * 256 + lane * 4) = acc;
  }
}

int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 100;
  {  // ---- register-only packed products
    const int blocks = 8192, iters = 20000;
    float* x;
    Stats* st;
    hipMalloc(&x, (size_t)blocks * 256 * 2 * 4);
    hipMalloc(&st, sizeof(Stats));
    std::vector<float> h((size_t)blocks * 256 * 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.5f + 1e-3f * (float)((i * 2654435761u) % 1999);
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const char* names[7] = {"mul op_sel:[0,1]", "mul", "fma op_sel:[0,1,0]", "mul op_sel:[1,0]", "mul op_sel_hi:[1,0]",
                            "add op_sel:[0,1]", "mul op_sel:[1,1]"};
    for (int mode = 0; mode < 7; ++mode) {
      hipMemset(st, 0, sizeof(Stats));
      for (int r = 0; r < R; ++r) {
        switch (mode) {
          case 0: hipLaunchKernelGGL(k_pk<0>, dim3(blocks), dim3(256), 0, 0, x, st, iters); break;
          case 1: hipLaunchKernelGGL(k_pk<1>, dim3(blocks), dim3(256), 0, 0, x, st, iters); break;
          case 2: hipLaunchKernelGGL(k_pk<2>, dim3(blocks), dim3(256), 0, 0, x, st, iters); break;
          case 3: hipLaunchKernelGGL(k_pk<3>, dim3(blocks), dim3(256), 0, 0, x, st, iters); break;
          case 4: hipLaunchKernelGGL(k_pk<4>, dim3(blocks), dim3(256), 0, 0, x, st, iters); break;
          case 5: hipLaunchKernelGGL(k_pk<5>, dim3(blocks), dim3(256), 0, 0, x, st, iters); break;
          default: hipLaunchKernelGGL(k_pk<6>, dim3(blocks), dim3(256), 0, 0, x, st, iters); break;
        }
      }
      Stats s;
      hipMemcpy(&s, st, sizeof(s), hipMemcpyDeviceToHost);
      printf("v_pk_%-20s: %d launches, %llu waves (%llu changed HW_ID/XCC_ID while running); wrong lo/hi per lane quarter:",
             names[mode], R, s.waves, s.moved);
      for (int q = 0; q < 4; ++q) printf(" [%llu %llu]", s.bad[q][0], s.bad[q][1]);
      if (s.sample_taken)
        printf("; one wrong low product: x = (%g, %g) a = (%g, %g): packed %g, scalar %g", s.sample[0], s.sample[1],
               s.sample[2], s.sample[3], s.sample[4], s.sample[5]);
      printf("\n");
      fflush(stdout);
    }
  }
  if (argc > 2 && !strcmp(argv[2], "twostream")) {
    // ---- ONE process, two streams: the failing form on one stream, another packed kernel on the other, co-resident
    const int blocks = 2048, iters = 20000;
    float* x;
    Stats *st, *st2;
    hipMalloc(&x, (size_t)blocks * 256 * 2 * 4); hipMalloc(&st, sizeof(Stats)); hipMalloc(&st2, sizeof(Stats));
    std::vector<float> h((size_t)blocks * 256 * 2, 1.25f);
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(st, 0, sizeof(Stats)); hipMemset(st2, 0, sizeof(Stats));
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    for (int r = 0; r < 4 * R; ++r) {
      hipLaunchKernelGGL(k_pk<0>, dim3(blocks), dim3(256), 0, s1, x, st, iters);
      hipLaunchKernelGGL(k_pk<6>, dim3(blocks), dim3(256), 0, s2, x, st2, iters);
    }
    hipDeviceSynchronize();
    Stats s;
    hipMemcpy(&s, st, sizeof(s), hipMemcpyDeviceToHost);
    printf("two streams of ONE process (op_sel:[0,1] next to op_sel:[1,1], %d launches each, half the chip each): wrong lo/hi per lane quarter:", 4 * R);
    for (int q = 0; q < 4; ++q) printf(" [%llu %llu]", s.bad[q][0], s.bad[q][1]);
    printf("\n");
    return 0;
  }
  {  // ---- the walk
    const int N = 8192, DEG = 16, E = N * DEG;
    float *qkv, *pe, *cs, *m, *A;
    int* src;
    hipMalloc(&qkv, (size_t)N * 768 * 4); hipMalloc(&pe, (size_t)E * 768 * 4); hipMalloc(&cs, (size_t)E * 4);
    hipMalloc(&m, (size_t)E * 256 * 4); hipMalloc(&A, (size_t)N * 256 * 4); hipMalloc(&src, (size_t)E * 4);
    std::vector<float> h((size_t)E * 768);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-3f * (float)((i * 2654435761u) % 4001) - 2.0f;
    hipMemcpy(pe, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(qkv, h.data() + 12345, (size_t)N * 768 * 4, hipMemcpyHostToDevice);
    hipMemcpy(cs, h.data() + 777, (size_t)E * 4, hipMemcpyHostToDevice);
    std::vector<int> hs(E);
    for (int e = 0; e < E; ++e) hs[e] = (int)(((size_t)e * 2654435761u) % N);
    hipMemcpy(src, hs.data(), (size_t)E * 4, hipMemcpyHostToDevice);
    std::vector<float> ref((size_t)E * 256), cur((size_t)E * 256);
    unsigned long long by_c[4] = {0, 0, 0, 0}, by_q[4] = {0, 0, 0, 0}, zeros = 0;
    int bad_launches = 0;
    for (int r = 0; r < R; ++r) {
      hipMemset(m, 0xff, (size_t)E * 256 * 4);
      hipLaunchKernelGGL(k_walk, dim3(N / 4), dim3(256), 0, 0, N, DEG, qkv, pe, cs, src, m, A);
      hipMemcpy(r ? cur.data() : ref.data(), m, (size_t)E * 256 * 4, hipMemcpyDeviceToHost);
      if (!r || !memcmp(cur.data(), ref.data(), cur.size() * 4)) continue;
      ++bad_launches;
      for (size_t i = 0; i < cur.size(); ++i)
        if (memcmp(&cur[i], &ref[i], 4)) {
          const int col = (int)(i % 256);
          ++by_c[col % 4]; ++by_q[col / 64];
          zeros += cur[i] == 0.f || ref[i] == 0.f;
        }
    }
    printf("walk     : %d of %d launches differ from the first; differing elements by channel (col %% 4) [%llu %llu %llu %llu], "
           "by lane quarter (col / 64) [%llu %llu %llu %llu], of which zero in one of the two %llu\n",
           bad_launches, R - 1, by_c[0], by_c[1], by_c[2], by_c[3], by_q[0], by_q[1], by_q[2], by_q[3], zeros);
  }
  return 0;
}
