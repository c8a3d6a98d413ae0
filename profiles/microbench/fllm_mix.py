#!/usr/bin/env python3
"""Generates fllm_mix.hip: issue-rate microbenchmark for the instruction mix of an FLL wave whose far band-edge taps run as
chained f32 MFMAs in the SAME wave (16 channels per wave, four row groups of two outputs): per sample step ~16 plain VALU,
~12 packed, 4 MFMA (two chains), 2 permlane swaps.  Variants: MFMAs replaced by plain VALU / swaps replaced by v_mov."""
import sys

def body(mfma, swaps, steps=8):
    L = []
    for s in range(steps):
        # dependent scalar-ish chain (NCO + loop): 16 plain VALU on v20
        for i in range(8):
            L.append("v_fma_f32 v20, v20, v21, v22")
        # packed: 3 cmul + 4 near FMAs + 5 error
        for i in range(6):
            L.append("v_pk_fma_f32 v[24:25], v[26:27], v[28:29], v[24:25]")
            L.append("v_fma_f32 v20, v20, v21, v22")
        if mfma:
            L.append("v_mfma_f32_16x16x4_f32 v[40:43], v30, v31, v[40:43]")
        else:
            L.append("v_fma_f32 v40, v30, v31, v40")
        for i in range(3):
            L.append("v_pk_fma_f32 v[32:33], v[26:27], v[28:29], v[32:33]")
            L.append("v_max_f32 v23, v20, v21")
        if mfma:
            L.append("v_mfma_f32_16x16x4_f32 v[44:47], v30, v31, v[44:47]")
        else:
            L.append("v_fma_f32 v44, v30, v31, v44")
        L.append("v_mov_b32 v34, v23")
        L.append("v_mov_b32 v35, v23")
        L.append("v_fma_f32 v20, v20, v21, v22")
        if swaps:
            L.append("v_permlane32_swap_b32 v34, v35")
        else:
            L.append("v_mov_b32 v34, v35")
        if mfma:
            L.append("v_mfma_f32_16x16x4_f32 v[40:43], v30, v31, v[40:43]")
        else:
            L.append("v_fma_f32 v40, v30, v31, v40")
        L.append("v_mov_b32 v36, v34")
        L.append("v_fma_f32 v20, v20, v21, v22")
        L.append("v_fma_f32 v20, v20, v21, v22")
        if swaps:
            L.append("v_permlane16_swap_b32 v36, v34")
        else:
            L.append("v_mov_b32 v36, v34")
        if mfma:
            L.append("v_mfma_f32_16x16x4_f32 v[44:47], v30, v31, v[44:47]")
        else:
            L.append("v_fma_f32 v44, v30, v31, v44")
        for i in range(4):
            L.append("v_fma_f32 v20, v20, v36, v22")
        L.append("v_cndmask_b32 v30, v30, v20, s[10:11]")
        L.append("v_cndmask_b32 v31, v31, v20, s[10:11]")
    return L

def kernel(name, mfma, swaps):
    b = body(mfma, swaps)
    n = len(b)
    asm = "\\n".join(b)
    return n, '''
__global__ __launch_bounds__(64) void %s(float* out, int iters) {
    float r = threadIdx.x * 1e-3f;
    asm volatile(
        "v_mov_b32 v20, %%0\\nv_mov_b32 v21, 0x3f7fff00\\nv_mov_b32 v22, 0x3a000000\\n"
        "v_mov_b32 v24, 0\\nv_mov_b32 v25, 0\\nv_mov_b32 v26, 0x3f000000\\nv_mov_b32 v27, 0x3f000000\\nv_mov_b32 v28, 0x3f000000\\nv_mov_b32 v29, 0x3f000000\\n"
        "v_mov_b32 v30, 0x3f000000\\nv_mov_b32 v31, 0x3f000000\\nv_mov_b32 v32, 0\\nv_mov_b32 v33, 0\\n"
        "v_mov_b32 v40, 0\\nv_mov_b32 v41, 0\\nv_mov_b32 v42, 0\\nv_mov_b32 v43, 0\\nv_mov_b32 v44, 0\\nv_mov_b32 v45, 0\\nv_mov_b32 v46, 0\\nv_mov_b32 v47, 0\\n"
        "v_mov_b32 v34, 0\\nv_mov_b32 v35, 0\\nv_mov_b32 v36, 0\\nv_mov_b32 v23, 0\\n"
        "s_mov_b64 s[10:11], 0x0f0f\\n"
        "1:\\n"
        "%s\\n"
        "s_sub_u32 %%1, %%1, 1\\ns_cmp_lg_u32 %%1, 0\\ns_cbranch_scc1 1b\\n"
        "s_nop 7\\ns_nop 7\\nv_add_f32 %%0, v20, v40\\nv_add_f32 %%0, %%0, v44\\nv_add_f32 %%0, %%0, v24\\nv_add_f32 %%0, %%0, v36\\n"
        : "+v"(r), "+s"(iters) : : "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36",
          "v40","v41","v42","v43","v44","v45","v46","v47","s10","s11","scc","vcc");
    out[blockIdx.x * 64 + threadIdx.x] = r;
}
''' % (name, asm)

src = ['#include <hip/hip_runtime.h>', '#include <cstdio>']
ns = {}
for name, m, s in (("k_valu", 0, 0), ("k_mfma", 1, 0), ("k_swap", 0, 1), ("k_both", 1, 1)):
    n, k = kernel(name, m, s)
    ns[name] = n
    src.append(k)
src.append('''
int main() {
    float* d; hipMalloc(&d, 4 * 64 * 1024);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    struct { const char* n; void (*k)(float*, int); int slots; } ks[] = {
        {"valu", k_valu, %d}, {"mfma", k_mfma, %d}, {"swap", k_swap, %d}, {"both", k_both, %d} };
    for (int wg : {256, 512, 1024})
    for (auto& k : ks) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k.k, dim3(wg), dim3(64), 0, 0, d, iters);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) std::printf("{\\"kernel\\": \\"%%s\\", \\"workgroups\\": %%d, \\"slots_per_8_steps\\": %%d, \\"ms\\": %%.3f, \\"clocks_per_step\\": %%.1f, \\"clocks_per_slot\\": %%.2f}\\n",
                        k.n, wg, k.slots, ms, ms * 1e-3 * clk * 1e3 / iters / 8, ms * 1e-3 * clk * 1e3 / iters / k.slots);
        }
    }
    return 0;
}
''' % (ns["k_valu"] + 3, ns["k_mfma"] + 3, ns["k_swap"] + 3, ns["k_both"] + 3))
open("fllm_mix.hip", "w").write("\n".join(src))
print(ns)
