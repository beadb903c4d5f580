// Round 6 (EXPERIMENTS R6-1): does v_pk_add_f32 with a 64-bit SGPR pair as a source add the pair's HIGH half to the high lane?
// The SLP vectoriser turns `t_max = distance - 1e-4f` and `last.z + 0.0f` into one `fadd <2 x float> v, <-1e-4, 0.0>`; the gfx950
// back end materialises the constant vector with `s_mov_b64 s[a:b], 0xb8d1b717` (low = -1e-4, high = 0) and issues
// `v_pk_add_f32 v[..], v[..], s[a:b]`.  This program runs that very pair of instructions (inline assembly) next to the scalar
// additions and prints both.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/pk_add_sgpr_pair.hip -o tools/microbench/pk_add_sgpr_pair && tools/microbench/pk_add_sgpr_pair
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

typedef float float2v __attribute__((ext_vector_type(2)));

__global__ void pk_add_kernel(const float2v *in, float2v *out_asm, float2v *out_scalar, float2v *out_vector_ir, float2v *out_vgpr)
{
    const float2v v = in[threadIdx.x];
    float2v r;
    // the instruction pair of the miscompiled kernel
    asm volatile("s_mov_b64 s[18:19], 0xb8d1b717\n\tv_pk_add_f32 %0, %1, s[18:19]" : "=v"(r) : "v"(v) : "s18", "s19");
    out_asm[threadIdx.x] = r;
    out_scalar[threadIdx.x] = float2v{v.x + -1e-4f, v.y + 0.0f};
    // the IR form the SLP vectoriser produces: the back end chooses the instructions
    const float2v c = {-1e-4f, 0.0f};
    out_vector_ir[threadIdx.x] = v + c;
    // the same constant pair in VGPRs
    float2v q;
    asm volatile("v_mov_b32 v20, 0xb8d1b717\n\tv_mov_b32 v21, 0\n\tv_pk_add_f32 %0, %1, v[20:21]" : "=v"(q) : "v"(v) : "v20", "v21");
    out_vgpr[threadIdx.x] = q;
}

int main()
{
    float2v host_in[64], *in, *o[4];
    for (int i = 0; i < 64; ++i)
        host_in[i] = float2v{1.0f + i, 100.0f + i};
    hipMalloc(&in, sizeof host_in);
    for (auto &p : o)
        hipMalloc(&p, sizeof host_in);
    hipMemcpy(in, host_in, sizeof host_in, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(pk_add_kernel, dim3(1), dim3(64), 0, 0, in, o[0], o[1], o[2], o[3]);
    float2v got[4][64];
    for (int k = 0; k < 4; ++k)
        hipMemcpy(got[k], o[k], sizeof host_in, hipMemcpyDeviceToHost);
    const char *names[4] = {"asm: v_pk_add_f32 v, v, s[18:19]", "scalar adds", "fadd <2 x float> v, <-1e-4, 0>", "asm: v_pk_add_f32 v, v, v[20:21]"};
    int bad = 0;
    for (int k = 0; k < 4; ++k)
    {
        int wrong = 0;
        for (int i = 0; i < 64; ++i)
            wrong += memcmp(&got[k][i], &got[1][i], sizeof(float2v)) != 0;
        printf("{\"form\": \"%s\", \"lane0\": [%.9g, %.9g], \"expected\": [%.9g, %.9g], \"lanes_differing_from_scalar\": %d}\n", names[k], got[k][0].x, got[k][0].y,
               got[1][0].x, got[1][0].y, wrong);
        bad += wrong;
    }
    return bad ? 1 : 0;
}
