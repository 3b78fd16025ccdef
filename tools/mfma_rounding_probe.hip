// How does v_mfma_f32_32x32x16_bf16 round?  (Round-2 question: the split-bf16 GEMM is as accurate as the fp32 MFMA element
// by element, yet a long non-negative-weighted sum over its outputs -- the basis-coefficient gradient -- sits 10x
// further from float64 than the fp32 MFMA's, which is what a small systematic rounding bias would do.)
//
// One MFMA computes D = sum_k A[i][k] B[k][j] + C.  Every row of A and every column of B is the same here, so all
// 1024 outputs are the same number and the test is on scalars:
//     D = p0 + p1 + c,    p0 = a0 * 1, p1 = a1 * 1
// with operands chosen so that the exact sum needs more than 24 bits.  The output is compared with the exact sum
// rounded to nearest-even / toward zero / toward -inf / toward +inf.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_rounding_probe.hip -o /tmp/mfma_rounding_probe
#include <hip/hip_runtime.h>
#include <cfenv>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

struct Case { float a0, a1, b0, b1, c; };

__global__ void k_probe(const Case* cs, float* out, int n) {
  const int lane = threadIdx.x;
  for (int t = 0; t < n; ++t) {
    const Case q = cs[t];
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)0.0f; b[i] = (__bf16)0.0f; }
    if (lane < 32) { a[0] = (__bf16)q.a0; a[1] = (__bf16)q.a1; b[0] = (__bf16)q.b0; b[1] = (__bf16)q.b1; }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = q.c;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[t] = acc[0];
  }
}

static float round_mode(double x, int mode) {
  std::fesetround(mode);
  volatile float f = (float)x;
  std::fesetround(FE_TONEAREST);
  return f;
}
static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

int main() {
  static Case cs[4096];
  static double exact[4096];
  static char label[4096][96];
  int n = 0;
  auto add = [&](float a0, float a1, float b0, float b1, float c, const char* what, int k) {
    cs[n] = Case{a0, a1, b0, b1, c};
    exact[n] = (double)a0 * b0 + (double)a1 * b1 + (double)c;
    std::snprintf(label[n], sizeof label[n], "%s k=%d", what, k);
    ++n;
  };
  const float tail = 1.0f + 1.0f / 128;      // 1.0000001b: 8 significant bits, exact in bf16
  for (int k = 14; k <= 34; ++k) {
    const float small = std::ldexp(tail, -k);
    add(1.0f, small, 1.0f, 1.0f, 0.0f, "p0=1 p1=+s c=0", k);
    add(1.0f, -small, 1.0f, 1.0f, 0.0f, "p0=1 p1=-s c=0", k);
    add(-1.0f, small, 1.0f, 1.0f, 0.0f, "p0=-1 p1=+s c=0", k);
    add(-1.0f, -small, 1.0f, 1.0f, 0.0f, "p0=-1 p1=-s c=0", k);
    add(small, 0.0f, 1.0f, 1.0f, 1.0f, "c=1 p0=+s", k);
    add(-small, 0.0f, 1.0f, 1.0f, 1.0f, "c=1 p0=-s", k);
    add(small, 0.0f, 1.0f, 1.0f, -1.0f, "c=-1 p0=+s", k);
    add(-small, 0.0f, 1.0f, 1.0f, -1.0f, "c=-1 p0=-s", k);
    // a 16-bit product (both factors carry 8 bits) below a large accumulator
    add(std::ldexp(tail, -k / 2), 0.0f, std::ldexp(tail, -(k - k / 2)), 1.0f, 1.0f, "c=1 p0=(s*s)", k);
    add(std::ldexp(tail, -k / 2), 0.0f, -std::ldexp(tail, -(k - k / 2)), 1.0f, 1.0f, "c=1 p0=-(s*s)", k);
  }
  Case* dcs; float* dout;
  hipMalloc(&dcs, sizeof(Case) * n); hipMalloc(&dout, sizeof(float) * n);
  hipMemcpy(dcs, cs, sizeof(Case) * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dcs, dout, n);
  static float out[4096];
  if (hipMemcpy(out, dout, sizeof(float) * n, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
  int agree[4] = {0, 0, 0, 0}, inexact = 0;
  const int modes[4] = {FE_TONEAREST, FE_TOWARDZERO, FE_DOWNWARD, FE_UPWARD};
  const char* names[4] = {"nearest-even", "toward-zero", "toward -inf", "toward +inf"};
  for (int t = 0; t < n; ++t) {
    float r[4];
    for (int m = 0; m < 4; ++m) r[m] = round_mode(exact[t], modes[m]);
    const bool is_inexact = (double)r[0] != exact[t];
    if (!is_inexact) {
      if (out[t] != r[0]) printf("%-28s EXACT case wrong: got %08x want %08x\n", label[t], bits(out[t]), bits(r[0]));
      continue;
    }
    ++inexact;
    char which[64] = "";
    for (int m = 0; m < 4; ++m)
      if (out[t] == r[m]) { ++agree[m]; std::strcat(which, m == 0 ? "N" : m == 1 ? "Z" : m == 2 ? "D" : "U"); }
    printf("%-28s got %08x  rne %08x rtz %08x  -> %s\n", label[t], bits(out[t]), bits(r[0]), bits(r[1]),
           which[0] ? which : "NONE");
  }
  printf("inexact cases: %d\n", inexact);
  for (int m = 0; m < 4; ++m) printf("  agrees with round %-13s in %d\n", names[m], agree[m]);
  return 0;
}
