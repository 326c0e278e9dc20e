// Probe: v_mfma_f64_16x16x4_f64 operand/result layout and issue rate on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void layout(const double *A, const double *B, double *C) {   // A[16][4], B[4][16] row-major -> C[16][16]
  const int l = threadIdx.x;
  const double a = A[(l & 15) * 4 + (l >> 4)];      // A[i=l&15][k=l>>4]
  const double b = B[(l >> 4) * 16 + (l & 15)];     // B[k=l>>4][j=l&15]
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[(l * 4 + r)] = c[r];   // raw: lane-major
}

template <int NACC>
__global__ void rate(double *out, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  std::vector<double> A(64), B(64), C(256);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = (k == (i & 3)) ? 1.0 + i : 0.0;   // picks row k=i&3 of B scaled by 1+i
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 100 * k + j;                     // asymmetric
  double *dA, *dB, *dC;
  hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dC, 256 * 8);
  hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice);
  layout<<<1, 64>>>(dA, dB, dC);
  hipMemcpy(C.data(), dC, 256 * 8, hipMemcpyDeviceToHost);
  // expected C[i][j] = (1+i) * (100*(i&3) + j); find mapping (lane, reg) -> (i, j)
  int ok_guide = 0, ok_alt = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    const double v = C[l * 4 + r];
    { const int j = l & 15, i = (l >> 4) + 4 * r; if (v == (1.0 + i) * (100 * (i & 3) + j)) ok_guide++; }
    { const int j = l & 15, i = (l >> 4) * 4 + r; if (v == (1.0 + i) * (100 * (i & 3) + j)) ok_alt++; }
  }
  printf("layout: guide(row=(l>>4)+4r, col=l&15) matches %d/256 ; alt(row=4(l>>4)+r) matches %d/256\n", ok_guide, ok_alt);
  double *dout; hipMalloc(&dout, 256 * 4 * 1024 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves = 1; waves <= 8; waves *= 2) {
    const int iters = 20000, blocks = 256 * 4;   // 4 blocks per CU
    auto run = [&](auto kern, int nacc) {
      kern<<<blocks, 64 * waves / 4 < 64 ? 64 : 64 * waves / 4>>>(dout, 10);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      kern<<<blocks, 64 * waves / 4 < 64 ? 64 : 64 * waves / 4>>>(dout, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const int wpb = (64 * waves / 4 < 64 ? 64 : 64 * waves / 4) / 64;
      const double flops = (double)blocks * wpb * iters * nacc * 2048.0;
      printf("waves/CU=%d nacc=%d: %.2f TFLOP/s fp64  (%.3f ms)\n", 4 * wpb, nacc, flops / ms / 1e9, ms);
    };
    run(rate<1>, 1); run(rate<4>, 4); run(rate<16>, 16);
  }
  return 0;
}
