// L2 -> CU bandwidth of the NT kernels' operand-tile access pattern (hardware probe, not part of the product path).
//
// Every work-group streams "steps" of 32 KB the way conv_nt_bf16_kernel does: 256 row segments of seg bytes at a row
// stride of `stride` bytes, all work-groups on the same column chunk in the same step.  Compared with the same bytes
// laid out contiguously (stride == seg).  Footprint per XCD stays inside its 4 MiB L2, so after the first pass every
// request is an L2 hit.   hipcc -O3 --offload-arch=gfx950 tools/l2_probe.hip -o /tmp/l2_probe && /tmp/l2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// mode 0: plain global_load_dwordx4 into registers; mode 1: global_load_lds_dwordx4 (LDS-DMA)
template <int MODE>
__global__ __launch_bounds__(256) void probe(const char* __restrict__ base, unsigned* __restrict__ sink, int rows_per_wg,
                                             long stride, int seg, int chunks, int steps, int wg_tiles) {
  __shared__ __attribute__((aligned(16))) char lds[2][32768];
  const int tid = threadIdx.x;
  const int per_row = seg / 16;             // 16-byte pieces per row segment
  const int rows_per_pass = 256 / per_row;  // rows covered by one instruction of the work-group
  const int passes = 32768 / 4096;          // 8 x 4 KB = 32 KB per step
  const long tile_row0 = (long)(blockIdx.x % wg_tiles) * rows_per_wg;
  const int r = tid / per_row, piece = tid % per_row;
  uint4 acc = {0, 0, 0, 0};
  for (int s = 0; s < steps; ++s) {
    const int chunk = s % chunks;
    const int shift = (s / chunks) % 7;  // the tap shift: the next "tap" re-reads the tile one row further down
#pragma unroll
    for (int p = 0; p < passes; ++p) {
      const long row = tile_row0 + shift + (long)(p * rows_per_pass + r) % rows_per_wg;
      const char* src = base + row * stride + (long)chunk * seg + piece * 16;
      if (MODE == 0) {
        uint4 v = *(const uint4*)src;
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
      } else {
        // wave-uniform LDS base in M0, each lane lands at base + lane * 16
        char* dst = &lds[s & 1][p * 4096 + (tid / 64) * 1024];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    }
    if (MODE == 1) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // one step in flight
    }
  }
  if (MODE == 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc = *(uint4*)&lds[0][tid * 16];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[blockIdx.x * 256 + tid] = acc.x;
}

// mode 2: what one step of the 4-wave 128x128 NT kernel asks of the LDS, without the MFMAs: per wave 8 LDS-DMA requests
// (32 KB per work-group) and/or 16 ds_read_b128 of the previous tile (64 KB per work-group), 2-slot ring, one barrier.
template <bool DMA, bool READS>
__global__ __launch_bounds__(256) void lds_mix(const char* __restrict__ base, unsigned* __restrict__ sink, int steps) {
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const char* src = base + (long)(blockIdx.x % 64) * 262144 + (tid >> 3) * 512 + (tid & 7) * 16;
  const unsigned rd = (unsigned)(size_t)dyn + wave * 8192 + lane * 16;
  uint4 f[16];
  for (int i = 0; i < 16; ++i) f[i] = make_uint4(0, 0, 0, 0);
  for (int s = 0; s < steps; ++s) {
    char* slot = dyn + (s & 1) * 32768;
    if (DMA) {
#pragma unroll
      for (int p = 0; p < 8; ++p)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long)(s & 3) * 128 + p * 16384),
                                         (__attribute__((address_space(3))) void*)(slot + p * 4096 + wave * 1024), 16, 0, 0);
    }
    if (READS) {
      const unsigned a = rd + ((s + 1) & 1) * 32768;
#pragma unroll
      for (int i = 0; i < 16; ++i)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[i]) : "v"(a), "n"(0) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  unsigned x = 0;
  for (int i = 0; i < 16; ++i) x ^= f[i].x ^ f[i].w;
  if (x == 0x12345u) sink[blockIdx.x * 256 + tid] = x;
}

template <bool DMA, bool READS>
static void run_mix(const char* name, const char* buf, unsigned* sink, hipEvent_t a, hipEvent_t b) {
  const int steps = 2000;
  (void)hipFuncSetAttribute((const void*)lds_mix<DMA, READS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(a));
    lds_mix<DMA, READS><<<256, 256, 65536>>>(buf, sink, steps);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  printf("lds-mix  %-52s %8.1f us  %.3f us/step\n", name, best * 1e3, best * 1e3 / steps);
}

int main() {
  const int wgs = 256, steps = 28 * 20;
  const size_t bytes = 512ull << 20;
  char* buf;
  unsigned* sink;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&sink, wgs * 256 * 4));
  CK(hipMemset(buf, 1, bytes));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  struct Case { const char* name; int rows_per_wg; long stride; int seg; int chunks; int wg_tiles; };
  // 250-channel layers: 128-row tiles (x2 column tiles share them), row stride 512 B, 64-channel chunks of 128 B
  // 2048-channel layers: 256-row tiles, row stride 4096 B
  std::vector<Case> cases = {
      {"rows 128 B @ stride 512 B (inner layers, A tile)", 256, 512, 128, 4, 64},
      {"same bytes, chunk-major (contiguous 128 B rows)", 256, 128, 128, 4, 64},
      {"rows 128 B @ stride 4096 B (2048-ch layers)", 256, 4096, 128, 32, 16},
      {"rows 128 B @ stride 4224 B (padded row)", 256, 4224, 128, 32, 16},
      {"rows 256 B @ stride 512 B (BK = 128)", 128, 512, 256, 2, 64},
  };
  for (int mode = 0; mode < 2; ++mode) {
    for (auto& c : cases) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(a));
        if (mode == 0)
          probe<0><<<wgs, 256>>>(buf, sink, c.rows_per_wg, c.stride, c.seg, c.chunks, steps, c.wg_tiles);
        else
          probe<1><<<wgs, 256>>>(buf, sink, c.rows_per_wg, c.stride, c.seg, c.chunks, steps, c.wg_tiles);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
      }
      const double total = (double)wgs * steps * 32768.0;
      printf("%-8s %-52s %8.1f us  %6.2f TB/s  %6.1f B/clk/CU @2.1GHz  %.3f us/step\n", mode ? "lds-dma" : "vgpr", c.name,
             best * 1e3, total / best / 1e9, total / (best * 1e-3) / 256 / 2.1e9, best * 1e3 / steps);
    }
  }
  run_mix<true, false>("32 KB LDS-DMA per step", buf, sink, a, b);
  run_mix<false, true>("64 KB ds_read_b128 per step", buf, sink, a, b);
  run_mix<true, true>("both (one step of the 128x128 tile, no MFMAs)", buf, sink, a, b);
  return 0;
}
