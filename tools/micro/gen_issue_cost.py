"""Generates tools/micro/issue_cost.hip: cycles per v_mfma_f32_16x16x32_f16 with filler instructions between the MFMAs
(one wave per SIMD, 256 work-groups), for accumulator chains of 1 / 2 / 6 accumulators held in AGPRs or VGPRs.
The loop body is ONE inline-asm block (the compiler cannot reorder or drop anything)."""
import sys
FILL = {
    'none': [],
    'add1': ['v_add_f32 %[f0], %[f0], %[f0]'],
    'add2': ['v_add_f32 %[f0], %[f0], %[f0]', 'v_add_f32 %[f1], %[f1], %[f1]'],
    'add3': ['v_add_f32 %[f0], %[f0], %[f0]', 'v_add_f32 %[f1], %[f1], %[f1]', 'v_add_f32 %[f2], %[f2], %[f2]'],
    'add4': ['v_add_f32 %[f0], %[f0], %[f0]', 'v_add_f32 %[f1], %[f1], %[f1]', 'v_add_f32 %[f2], %[f2], %[f2]', 'v_add_f32 %[f3], %[f3], %[f3]'],
    'mix1': ['v_fma_mix_f32 %[f0], %[f1], %[f2], %[f0] op_sel_hi:[1,0,0]'],
    'mix2': ['v_fma_mix_f32 %[f0], %[f1], %[f2], %[f0] op_sel_hi:[1,0,0]', 'v_fma_mix_f32 %[f3], %[f1], %[f2], %[f3] op_sel_hi:[1,0,0]'],
    'cvt1': ['v_cvt_pk_f16_f32 %[f0], %[f1], %[f2]'],
    'accrd1': ['v_accvgpr_read_b32 %[f0], %[spare]'],
    'accrd2': ['v_accvgpr_read_b32 %[f0], %[spare]', 'v_accvgpr_read_b32 %[f1], %[spare]'],
    'dsr128': ['ds_read_b128 %[ld], %[addr]'],
    'dsr128_add1': ['ds_read_b128 %[ld], %[addr]', 'v_add_f32 %[f0], %[f0], %[f0]'],
    'dsr64': ['ds_read_b64 %[ld2], %[addr]'],
    'dsw128': ['ds_write_b128 %[addr], %[st]'],
    'dsw128_half': None,   # one ds_write_b128 behind every second MFMA
    'gld': ['global_load_dwordx4 %[ld], %[gaddr], off'],
    'gld_quarter': None,   # one global load behind every fourth MFMA
    'dsr128_third': None,  # one ds_read_b128 behind every third MFMA (the sweep's ratio)
    'dsr128_third_add1': None,   # ... plus one v_add_f32 behind every MFMA
    'dsr128_third_add2': None,
    'sweep_mix': None,     # the sweep's main chunk: ds_read_b128 every 3rd MFMA, global_load every 9th, one v_add every 2nd
    'salu2': ['s_add_u32 %[s0], %[s0], 1', 's_and_b32 %[s0], %[s0], 0xffff'],
    'gld_saddr_quarter': None,   # global_load_dwordx4 v, v_off32, s[base:base+1] behind every fourth MFMA
    'bufld_quarter': None,       # buffer_load_dwordx4 v, v_off32, s[rsrc], 0 offen behind every fourth MFMA
    'gld_quarter_u64': None,     # v_lshl_add_u64 + global_load behind every fourth MFMA (what hipcc emits for base + lane offset)
    'bufld_half': None,
    'mixlo2': ['v_fma_mixlo_f16 %[f0], %[f1], %[f2], %[f3] op_sel_hi:[1,0,0]', 'v_fma_mixhi_f16 %[f0], %[f1], %[f2], %[f3] op_sel_hi:[1,0,0]'],
    'dsw128_sixth': None,        # one ds_write_b128 behind every sixth MFMA
    'dsw64_third': None,         # one ds_write_b64 behind every third MFMA
    'wait1': ['s_waitcnt lgkmcnt(15)'],
    'wait2': ['s_waitcnt lgkmcnt(15)', 's_waitcnt vmcnt(63)'],
    'wait1_add2': ['s_waitcnt lgkmcnt(15)', 'v_add_f32 %[f0], %[f0], %[f0]', 'v_add_f32 %[f1], %[f1], %[f1]'],
    'nop1_add2': ['s_nop 0', 'v_add_f32 %[f0], %[f0], %[f0]', 'v_add_f32 %[f1], %[f1], %[f1]'],
    'salu1_add2': ['s_add_u32 %[s0], %[s0], 1', 'v_add_f32 %[f0], %[f0], %[f0]', 'v_add_f32 %[f1], %[f1], %[f1]'],
    'salu2_add2': ['s_add_u32 %[s0], %[s0], 1', 's_and_b32 %[s0], %[s0], 0xffff', 'v_add_f32 %[f0], %[f0], %[f0]', 'v_add_f32 %[f1], %[f1], %[f1]'],
    'dsr_wait_add1': ['ds_read_b128 %[ld], %[addr]', 's_waitcnt lgkmcnt(15)', 'v_add_f32 %[f0], %[f0], %[f0]'],
    'accrd_dep': ['v_accvgpr_read_b32 %[f0], %[spare]', 'v_add_f32 %[f1], %[f0], %[f1]'],
    'add_dep2': ['v_add_f32 %[f0], %[f0], %[f1]', 'v_add_f32 %[f0], %[f0], %[f2]'],
}
NM = 48     # MFMAs per asm block
def body(nacc, chain, fill):
    lines = []
    for k in range(NM):
        a = (k // chain) % nacc
        lines.append(f'v_mfma_f32_16x16x32_f16 %[c{a}], %[a{k % 4}], %[b{(k // 4) % 4}], %[c{a}]')
        if fill == 'dsw128_half':
            if k % 2 == 1: lines.append('ds_write_b128 %[addr], %[st]')
        elif fill == 'gld_quarter':
            if k % 4 == 3: lines.append('global_load_dwordx4 %[ld], %[gaddr], off')
        elif fill == 'gld_saddr_quarter':
            if k % 4 == 3: lines.append('global_load_dwordx4 %[ld], %[voff], %[sbase]')
        elif fill == 'bufld_quarter':
            if k % 4 == 3: lines.append('buffer_load_dwordx4 %[ld], %[voff], %[rsrc], 0 offen')
        elif fill == 'bufld_half':
            if k % 2 == 1: lines.append('buffer_load_dwordx4 %[ld], %[voff], %[rsrc], 0 offen')
        elif fill == 'gld_quarter_u64':
            if k % 4 == 3: lines += ['v_lshl_add_u64 %[ga2], %[voff64], 0, %[sbase]', 'global_load_dwordx4 %[ld], %[ga2], off']
        elif fill == 'dsw128_sixth':
            if k % 6 == 5: lines.append('ds_write_b128 %[addr], %[st]')
        elif fill == 'dsw64_third':
            if k % 3 == 2: lines.append('ds_write_b64 %[addr], %[ld2]')
        elif fill.startswith('dsr128_third'):
            if k % 3 == 2: lines.append('ds_read_b128 %[ld], %[addr]')
            if fill.endswith('add1'): lines.append('v_add_f32 %[f0], %[f0], %[f0]')
            if fill.endswith('add2'): lines += ['v_add_f32 %[f0], %[f0], %[f0]', 'v_add_f32 %[f1], %[f1], %[f1]']
        elif fill == 'sweep_mix':
            if k % 3 == 2: lines.append('ds_read_b128 %[ld], %[addr]')
            if k % 9 == 4: lines.append('global_load_dwordx4 %[ld], %[gaddr], off')
            if k % 2 == 0: lines.append('v_add_f32 %[f0], %[f0], %[f0]')
        else:
            lines += FILL[fill]
    lines.append('s_waitcnt vmcnt(0) lgkmcnt(0)')
    return '\\n\\t'.join(lines)
out = ['// GENERATED by tools/micro/gen_issue_cost.py -- do not edit', '#include <hip/hip_runtime.h>', '#include <stdio.h>',
       'typedef float floatx4 __attribute__((ext_vector_type(4)));', 'typedef _Float16 half8 __attribute__((ext_vector_type(8)));',
       'typedef float floatx2 __attribute__((ext_vector_type(2)));']
kernels = []
only = sys.argv[2].split(',') if len(sys.argv) > 2 else None
for cons in ['a', 'v', 'A', 'W']:
    for nacc, chain in [(1, 1), (2, 1), (2, 3), (6, 1), (6, 3)]:
        for fill in FILL:
            if cons == 'v' and fill not in ('none', 'add1', 'add2', 'mix2', 'dsr128', 'accrd1'): continue
            if cons in ('A', 'W') and fill not in ('none', 'add2', 'dsr128_third_add1', 'sweep_mix'): continue
            if (nacc, chain) not in [(2, 1), (2, 3)] and fill not in ('none', 'add1', 'add2', 'dsr128'): continue
            if only and not ((nacc, chain) == (2, 1) and fill in only and (cons == 'a' or cons in ('A', 'W', 'v'))): continue
            name = f'k_{cons}_{nacc}_{chain}_{fill}'
            kernels.append((name, cons, nacc, chain, fill))
            acons = 'a' if cons in ('a', 'A') else 'v'          # accumulator file
            wcons = 'a' if cons in ('A', 'W') else 'v'          # file of the A operands (weights)
            accs = ', '.join(f'[c{i}] "+{acons}"(c{i})' for i in range(nacc))
            out.append(f'''__global__ __launch_bounds__(256, 1) void {name}(const uint4* __restrict__ w, float* out, int iters, unsigned long long* clk) {{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    half8 a0, a1, a2, a3, b0, b1, b2, b3;
    {{ const half8* h = reinterpret_cast<const half8*>(w) + lane; a0 = h[0]; a1 = h[64]; a2 = h[128]; a3 = h[192]; b0 = h[256]; b1 = h[320]; b2 = h[384]; b3 = h[448]; }}
    floatx4 {', '.join(f'c{i} = {{0.f, 0.f, 0.f, 0.f}}' for i in range(nacc))};
    float spare = 1.5f; floatx4 ld = {{0.f, 0.f, 0.f, 0.f}}, st = {{1.f, 1.f, 1.f, 1.f}};
    floatx2 ld2 = {{0.f, 0.f}};
    float f0 = 1.f, f1 = 0.5f, f2 = 0.25f, f3 = 0.125f;
    unsigned s0 = 0;
    const unsigned addr = (threadIdx.x * 16) & 0xffff;
    const uint4* gaddr = w + (threadIdx.x & 255);
    const unsigned voff = (threadIdx.x & 255) * 16;
    const uint4* sbase = w;
    const unsigned long long voff64 = voff;          // the lane's byte offset as a 64-bit value (base + offset stays inside the buffer)
    typedef int int4v __attribute__((ext_vector_type(4)));
    int4v rsrc;
    rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned long long)w); rsrc[1] = __builtin_amdgcn_readfirstlane((int)((unsigned long long)w >> 32));
    rsrc[2] = -1; rsrc[3] = 0x00020000;
    const uint4* ga2 = gaddr;
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<uint4*>(lds)[i] = w[i & 1023];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {{
        asm volatile("{body(nacc, chain, fill)}"
            : {accs}, [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3), [ld] "+v"(ld), [ld2] "+v"(ld2), [s0] "+s"(s0), [ga2] "+v"(ga2)
            : [a0] "{wcons}"(a0), [a1] "{wcons}"(a1), [a2] "{wcons}"(a2), [a3] "{wcons}"(a3), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3),
              [spare] "a"(spare), [addr] "v"(addr), [st] "v"(st), [gaddr] "v"(gaddr), [voff] "v"(voff), [voff64] "v"(voff64), [sbase] "s"(sbase), [rsrc] "s"(rsrc) : "memory");
    }}
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
    float s = f0 + f1 + f2 + f3 + ld[0] + ld2[0] + (float)s0{''.join(f' + c{i}[0] + c{i}[3]' for i in range(nacc))};
    out[blockIdx.x * 256 + threadIdx.x] = s;
}}''')
out.append('int main() {\n    uint4* w; float* o; unsigned long long* clk;\n    hipMalloc(&w, 1 << 20); hipMalloc(&o, 256 * 256 * 4); hipMalloc(&clk, 256 * 8);\n'
           '    { unsigned short* h = (unsigned short*)malloc(1 << 20); unsigned st = 12345u; for (int i = 0; i < (1 << 19); ++i) { st = st * 1664525u + 1013904223u; unsigned r = st >> 8; h[i] = (unsigned short)(((r & 1) << 15) | ((11 + ((r >> 1) & 3)) << 10) | ((r >> 3) & 0x3ff)); } hipMemcpy(w, h, 1 << 20, hipMemcpyHostToDevice); }\n'
           '    const int iters = 4000; unsigned long long h[256];')
for name, cons, nacc, chain, fill in kernels:
    out.append(f'    fflush(stdout); hipLaunchKernelGGL({name}, dim3(256), dim3(256), 65536, 0, w, o, iters, clk); hipDeviceSynchronize(); hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);\n'
               f'    printf("{{\\"acc_regs\\": \\"{cons}\\", \\"accumulators\\": {nacc}, \\"chain\\": {chain}, \\"filler\\": \\"{fill}\\", \\"cycles_per_mfma\\": %.2f, \\"err\\": %d}}\\n", (double)h[100] / ((double)iters * {NM}), (int)hipGetLastError());')
out.append('    return 0;\n}')
open(sys.argv[1], 'w').write('\n'.join(out) + '\n')
print(len(kernels), 'kernels')
