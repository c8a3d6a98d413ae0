#!/usr/bin/env python3
"""mfma_overlap.hip: does a wave's own VALU work overlap with its MFMAs?  Loop of one MFMA + N independent VALU instructions."""
KS = []
def kernel(name, mfma, n, valu="v_fma_f32 v20, v20, v21, v22"):
    b = []
    for rep in range(8):
        b.append(mfma.replace("ACC", "v[40:43]" if rep % 2 == 0 else "v[44:47]"))
        b += [valu] * n
    KS.append((name, len(b)))
    return '''
__global__ __launch_bounds__(64) void %s(float* out, int iters) {
    float r = threadIdx.x * 1e-3f;
    asm volatile(
        "v_mov_b32 v20, %%0\\nv_mov_b32 v21, 0x3f7fff00\\nv_mov_b32 v22, 0x3a000000\\nv_mov_b32 v26, 0\\nv_mov_b32 v27, 0\\nv_mov_b32 v28, 0\\nv_mov_b32 v29, 0\\n"
        "v_mov_b32 v30, 0x3f000000\\nv_mov_b32 v31, 0x3f000000\\nv_mov_b32 v32, 0x3f000000\\nv_mov_b32 v33, 0x3f000000\\n"
        "v_mov_b32 v40, 0\\nv_mov_b32 v41, 0\\nv_mov_b32 v42, 0\\nv_mov_b32 v43, 0\\nv_mov_b32 v44, 0\\nv_mov_b32 v45, 0\\nv_mov_b32 v46, 0\\nv_mov_b32 v47, 0\\n"
        "1:\\n"
        "%s\\n"
        "s_sub_u32 %%1, %%1, 1\\ns_cmp_lg_u32 %%1, 0\\ns_cbranch_scc1 1b\\n"
        "s_nop 7\\ns_nop 7\\ns_nop 7\\nv_add_f32 %%0, v20, v40\\nv_add_f32 %%0, %%0, v44\\nv_add_f32 %%0, %%0, v26\\n"
        : "+v"(r), "+s"(iters) : : "v20","v21","v22","v26","v27","v28","v29","v30","v31","v32","v33",
          "v40","v41","v42","v43","v44","v45","v46","v47","scc","vcc");
    out[blockIdx.x * 64 + threadIdx.x] = r;
}
''' % (name, "\\n".join(b))
src = ['#include <hip/hip_runtime.h>', '#include <cstdio>']
F32 = "v_mfma_f32_16x16x4_f32 ACC, v30, v31, ACC"
BF16 = "v_mfma_f32_16x16x16_bf16 ACC, v[30:31], v[32:33], ACC"      # gfx950: 16x16x32 takes 4 regs; 16x16x16 bf16 2 regs each
F32B = "v_mfma_f32_32x32x2_f32 v[48:63], v30, v31, v[48:63]"
for n in (0, 2, 4, 8, 12, 16):
    src.append(kernel("k_f32_%d" % n, F32, n))
    src.append(kernel("k_bf16_%d" % n, BF16, n))
    src.append(kernel("k_pk_%d" % n, F32, n, "v_pk_fma_f32 v[26:27], v[28:29], v[28:29], v[26:27]"))
src.append('int main() {\n float* d; (void)hipMalloc(&d, 4 * 64 * 4096);\n int clk = 0; (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);\n hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);\n const int iters = 20000;')
src.append(' struct K { const char* n; void (*k)(float*, int); int slots; } ks[] = {')
for name, slots in KS:
    src.append('  {"%s", %s, %d},' % (name, name, slots))
src.append(''' };
 for (int wg : {256, 2048})
 for (auto& k : ks) {
   float ms = 0;
   for (int rep = 0; rep < 2; rep++) {
     (void)hipEventRecord(e0, 0);
     hipLaunchKernelGGL(k.k, dim3(wg), dim3(64), 0, 0, d, iters);
     (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
     (void)hipEventElapsedTime(&ms, e0, e1);
   }
   std::printf("{\\"kernel\\": \\"%s\\", \\"workgroups\\": %d, \\"instrs_per_8_mfma\\": %d, \\"clocks_per_mfma_group\\": %.1f}\\n", k.n, wg, k.slots, ms * 1e-3 * clk * 1e3 / iters / 8);
 }
 return 0;
}''')
open("mfma_overlap.hip", "w").write("\n".join(src))
