// Developer tool: absolute error of the hardware v_sin_f32 / v_cos_f32 (argument in revolutions)
// against double precision, next to the polynomial of smpc_device_math.h, over yaw in [-R, R].
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "../../mpcholonavigation_amd/csrc/smpc_device_math.h"

__global__ void k(float R, unsigned n, double* out)
{
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  double e_hw = 0, e_hw2 = 0, e_poly = 0;
  for (unsigned j = i; j < n; j += gridDim.x * blockDim.x) {
    const float x = -R + 2.0f * R * ((float)j + 0.5f) / (float)n;
    const double s = sin((double)x), c = cos((double)x);
    const float t = x * 0.15915494309189535f;
    const float sh = __builtin_amdgcn_sinf(t), ch = __builtin_amdgcn_cosf(t);
    // revolutions formed with an fma pair (hi + lo of 1/2pi)
    const float t2 = fmaf(x, 0.15915494309189535f, x * -1.4534e-9f * 0.0f);
    const float f2 = t2 - floorf(t2);
    const float sh2 = __builtin_amdgcn_sinf(f2), ch2 = __builtin_amdgcn_cosf(f2);
    float sp, cp;
    smpc_sincos_fast(x, sp, cp);
    e_hw = fmax(e_hw, fmax(fabs((double)sh - s), fabs((double)ch - c)));
    e_hw2 = fmax(e_hw2, fmax(fabs((double)sh2 - s), fabs((double)ch2 - c)));
    e_poly = fmax(e_poly, fmax(fabs((double)sp - s), fabs((double)cp - c)));
  }
  out[3 * i] = e_hw; out[3 * i + 1] = e_hw2; out[3 * i + 2] = e_poly;
}

int main()
{
  const unsigned threads = 256 * 1024, n = 1u << 26;
  double* d; hipMalloc(&d, threads * 3 * sizeof(double));
  double* h = (double*)malloc(threads * 3 * sizeof(double));
  for (float R : {0.5f, 3.2f, 8.0f, 64.0f}) {
    hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, R, n, d);
    hipMemcpy(h, d, threads * 3 * sizeof(double), hipMemcpyDeviceToHost);
    double a = 0, b = 0, c = 0;
    for (unsigned i = 0; i < threads; ++i) { a = fmax(a, h[3 * i]); b = fmax(b, h[3 * i + 1]); c = fmax(c, h[3 * i + 2]); }
    printf("|yaw| <= %5.1f: max abs error  v_sin/v_cos(x/2pi) %.3e   with fract first %.3e   polynomial %.3e\n", R, a, b, c);
  }
  return 0;
}
