// Hardware probe (round 6): operand selection of v_pk_mov_b32 on gfx950 -- which op_sel / op_sel_hi bits put the HIGH words of two
// 64-bit sources into one 64-bit destination (the lattice kernel's row store packs the high words of its doubles).
//   hipcc --offload-arch=gfx950 -O2 tools/pk_mov_probe.hip -o tools/_probe/pk_mov_probe && tools/_probe/pk_mov_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned long long* out) {
    const unsigned long long x = 0x1111111122222222ull, y = 0x3333333344444444ull;  // hi:lo
    unsigned long long r0, r1, r2, r3;
    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(r0) : "v"(x), "v"(y));
    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1]" : "=v"(r1) : "v"(x), "v"(y));
    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r2) : "v"(x), "v"(y));
    asm volatile("v_pk_mov_b32 %0, %1, %2" : "=v"(r3) : "v"(x), "v"(y));
    if (threadIdx.x == 0) { out[0] = r0; out[1] = r1; out[2] = r2; out[3] = r3; }
}
int main() {
    unsigned long long* d; unsigned long long h[4];
    (void)hipMalloc(&d, sizeof(h));
    probe<<<1, 64>>>(d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"op_sel:[1,1]", "op_sel:[1,1] op_sel_hi:[1,1]", "op_sel:[1,0] op_sel_hi:[0,1]", "(none)"};
    for (int i = 0; i < 4; ++i) printf("%-32s -> %016llx   (x = 1111111122222222, y = 3333333344444444; want 3333333311111111)\n", names[i], h[i]);
    return 0;
}
