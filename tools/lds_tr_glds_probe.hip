// Probe: semantics of ds_read_b64_tr_b16 and __builtin_amdgcn_global_load_lds on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

__global__ void k_tr(const uint16_t* in, uint16_t* out, const int* lane_off) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  const int l = threadIdx.x;
  // per-lane byte address supplied by the host table
  uint32_t addr = (uint32_t)(uintptr_t)(lds) + (uint32_t)lane_off[l];
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[l * 4 + 0] = (uint16_t)(v[0] & 0xffff);
  out[l * 4 + 1] = (uint16_t)(v[0] >> 16);
  out[l * 4 + 2] = (uint16_t)(v[1] & 0xffff);
  out[l * 4 + 3] = (uint16_t)(v[1] >> 16);
}

__global__ void k_glds(const uint32_t* src, uint32_t* out) {
  extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
  const int l = threadIdx.x;
  // each lane's source: reversed 16-byte chunks, to show the LDS destination is lane-linear
  const uint32_t* g = src + 4 * (63 - l);
  __builtin_amdgcn_global_load_lds(g, sm + 256 /* dword offset: second KiB */, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = l; i < 512; i += 64) out[i] = sm[i];
}

int main() {
  std::vector<uint16_t> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = (uint16_t)i;
  uint16_t *din, *dout; int* doff;
  hipMalloc(&din, 8192); hipMalloc(&dout, 64 * 8); hipMalloc(&doff, 256);
  hipMemcpy(din, h.data(), 8192, hipMemcpyHostToDevice);
  // pattern A: canonical: lane l -> element index (l>>4)*64 + (l&15)*4   (4x16 row-major block per 16-lane group)
  // pattern B: rows with a free stride: lane i of group g -> row (i>>2) at stride 100 elements, cols 4*(i&3), group base g*1000
  for (int pat = 0; pat < 2; ++pat) {
    std::vector<int> off(64);
    for (int l = 0; l < 64; ++l) {
      int g = l >> 4, i = l & 15;
      int elem = pat == 0 ? g * 64 + i * 4 : g * 1000 + (i >> 2) * 100 + (i & 3) * 4;
      off[l] = elem * 2;
    }
    hipMemcpy(doff, off.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, 0, din, dout, doff);
    std::vector<uint16_t> o(256);
    hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost);
    printf("pattern %d\n", pat);
    for (int l = 0; l < 64; ++l) printf("lane %2d (addr elem %4d): %4d %4d %4d %4d\n", l, off[l] / 2, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
  }
  // glds
  std::vector<uint32_t> s(256); for (int i = 0; i < 256; ++i) s[i] = 1000 + i;
  uint32_t *ds, *dout2; hipMalloc(&ds, 1024); hipMalloc(&dout2, 2048);
  hipMemcpy(ds, s.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_glds, dim3(1), dim3(64), 2048, 0, ds, dout2);
  std::vector<uint32_t> o2(512); hipMemcpy(o2.data(), dout2, 2048, hipMemcpyDeviceToHost);
  printf("glds: sm[256..271] ="); for (int i = 256; i < 272; ++i) printf(" %u", o2[i]); printf("\n");
  printf("glds: sm[508..511] ="); for (int i = 508; i < 512; ++i) printf(" %u", o2[i]); printf("\n");
  hipError_t e = hipDeviceSynchronize(); printf("status %s\n", hipGetErrorString(e));
  return 0;
}
