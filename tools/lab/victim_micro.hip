// lab: which instruction class of a victim kernel changes its results when a neighbour PROCESS runs the split3 GEMM
// (k_gemm3_128: bf16 MFMA + 48 KB of LDS traffic) on the same GPU?  Four victims, each launched R times on the same input
// and compared bit for bit with its first result:
//   trans   : v_exp_f32 / v_rcp_f32 chains on register data
//   shuffle : cross-lane sums (ds_bpermute / DPP) on register data
//   sload   : every wave reads a table entry through the SCALAR cache (uniform address) and scales its row
//   vload   : the same with a vector load of the table entry
//   hipcc --offload-arch=gfx950 -O3 tools/lab/victim_micro.hip -o tools/lab/victim_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void k_trans(const float* x, float* y, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = x[i], acc = 0.f;
  for (int k = 0; k < 64; ++k) {
    float s = __frcp_rn(1.0f + __expf(-v));
    acc += v * s;
    v = v * 0.97f + 0.01f * s;
  }
  y[i] = acc;
}
__global__ void k_shuffle(const float* x, float* y, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = x[i], acc = 0.f;
  for (int k = 0; k < 64; ++k) {
    float p = v;
    for (int o = 4; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
    acc += p;
    v = v * 0.99f + 1e-3f * p;
  }
  y[i] = acc;
}
__global__ void k_sload(const float* x, const float* table, float* y, int rows) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= rows) return;
  const int w = __builtin_amdgcn_readfirstlane(wave);
  float acc = 0.f;
  const int iters = gridDim.y > 1 ? 4096 : 16;  // (grid.y == 2: the long form - milliseconds of scalar loads per launch)
  for (int k = 0; k < iters; ++k) {
    const float c = table[(size_t)((w * 7 + k * 13) % rows) * 8 + 1];  // uniform address: a scalar load
    acc += x[(size_t)w * 64 + lane] * c;
  }
  if (blockIdx.y == 0) y[(size_t)w * 64 + lane] = acc;
}
__global__ void k_vload(const float* x, const float* table, float* y, int rows) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= rows) return;
  float acc = 0.f;
  for (int k = 0; k < 16; ++k) {
    const float c = table[(size_t)((wave * 7 + k * 13 + (lane >> 6)) % rows) * 8 + 1];  // per-lane address: a vector load
    acc += x[(size_t)wave * 64 + lane] * c;
  }
  y[(size_t)wave * 64 + lane] = acc;
}
// the producer of `table` (so that, like geo in the engine, it is WRITTEN by a kernel shortly before it is read)
__global__ void k_fill(float* table, int rows, float seed) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows * 8) table[i] = seed + 1e-3f * (float)(i % 977);
}

int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 200;
  const int n = 1 << 22, rows = n / 64;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = 0.001f * (float)((i * 2654435761u) % 4001) - 2.0f;
  float *x, *y, *table;
  hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&table, (size_t)rows * 8 * 4);
  hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
  std::vector<float> ref(n), cur(n);
  const char* names[5] = {"trans", "shuffle", "sload", "vload", "sload-long"};
  for (int mode = 0; mode < 5; ++mode) {
    int bad = 0;
    for (int r = 0; r < R; ++r) {
      hipMemset(y, 0, n * 4);
      if (mode >= 2) hipLaunchKernelGGL(k_fill, dim3((rows * 8 + 255) / 256), dim3(256), 0, 0, table, rows, 0.5f);
      if (mode == 4) hipLaunchKernelGGL(k_sload, dim3(n / 256, 2), dim3(256), 0, 0, x, table, y, rows);
      if (mode == 0) hipLaunchKernelGGL(k_trans, dim3(n / 256), dim3(256), 0, 0, x, y, n);
      if (mode == 1) hipLaunchKernelGGL(k_shuffle, dim3(n / 256), dim3(256), 0, 0, x, y, n);
      if (mode == 2) hipLaunchKernelGGL(k_sload, dim3(n / 256), dim3(256), 0, 0, x, table, y, rows);
      if (mode == 3) hipLaunchKernelGGL(k_vload, dim3(n / 256), dim3(256), 0, 0, x, table, y, rows);
      hipMemcpy(r ? cur.data() : ref.data(), y, n * 4, hipMemcpyDeviceToHost);
      if (r && memcmp(cur.data(), ref.data(), n * 4)) ++bad;
    }
    printf("victim %-8s: %d of %d repeats differ from the first\n", names[mode], bad, R - 1);
    fflush(stdout);
  }
  return 0;
}
