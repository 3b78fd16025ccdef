// What do v_cvt_pk_bf16_f32 and v_dot2c_f32_bf16 compute?  (operand split of gemm_bf16x3.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
__global__ void k(const float* x, uint32_t* pk, float* r) {
  const int t = threadIdx.x;
  f32x2 f = {x[2 * t], x[2 * t + 1]};
  bf16x2 h = __builtin_convertvector(f, bf16x2);
  const bf16x2 s0 = {(__bf16)-1.0f, (__bf16)-0.0f};
  const bf16x2 s1 = {(__bf16)-0.0f, (__bf16)-1.0f};
  pk[t] = __builtin_bit_cast(uint32_t, h);
  r[4 * t + 0] = __builtin_amdgcn_fdot2_f32_bf16(h, s0, f[0], false);
  r[4 * t + 1] = __builtin_amdgcn_fdot2_f32_bf16(h, s1, f[1], false);
  r[4 * t + 2] = __builtin_amdgcn_fdot2_f32_bf16(h, s0, 0.0f, false);
  r[4 * t + 3] = __builtin_amdgcn_fdot2_f32_bf16(h, s1, 0.0f, false);
}
int main() {
  float hx[8] = {1.2345678f, -3.1415927f, 1e-3f, 123456.789f, 0.3333333f, -0.6666667f, 1.0f, 2.0f};
  float *dx, *dr; uint32_t* dp;
  hipMalloc(&dx, sizeof(hx)); hipMalloc(&dr, 16 * 4); hipMalloc(&dp, 16);
  hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, dx, dp, dr);
  float hr[16]; uint32_t hp[4];
  hipMemcpy(hr, dr, sizeof(hr), hipMemcpyDeviceToHost); hipMemcpy(hp, dp, sizeof(hp), hipMemcpyDeviceToHost);
  for (int t = 0; t < 4; ++t) {
    uint32_t lo = hp[t] << 16, hi = hp[t] & 0xffff0000u; float flo, fhi;
    memcpy(&flo, &lo, 4); memcpy(&fhi, &hi, 4);
    printf("x=(%.9g,%.9g) pk=%08x -> bf16 (low %.9g, high %.9g)  dot(s0,x0)=%.9g [x0-low=%.9g]  dot(s1,x1)=%.9g [x1-high=%.9g]  dot(s0,0)=%.9g dot(s1,0)=%.9g\n",
           hx[2*t], hx[2*t+1], hp[t], flo, fhi, hr[4*t], hx[2*t]-flo, hr[4*t+1], hx[2*t+1]-fhi, hr[4*t+2], hr[4*t+3]);
  }
  return 0;
}
