// ubench_valu.hip -- issue-rate microbenchmarks that guided the SOR kernel design (gfx950).
// Build & run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/ubench_valu.hip -o /tmp/ub && /tmp/ub
// Each kernel runs ITER x (K instructions) per wave; we report cycles per wave-instruction per
// SIMD at a given number of waves per SIMD (blockDim 256 = 1 wave per SIMD per block).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 2000

__device__ __forceinline__ float dpp_shr(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_rowshr(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xf, 0xf, false));
}

// 8 independent add chains
__global__ void k_add_indep(float* o, float s) {
    float a0=threadIdx.x,a1=a0+1,a2=a0+2,a3=a0+3,a4=a0+4,a5=a0+5,a6=a0+6,a7=a0+7;
    for (int i=0;i<ITER;++i){
#pragma unroll
        for(int u=0;u<4;++u){a0+=s;a1+=s;a2+=s;a3+=s;a4+=s;a5+=s;a6+=s;a7+=s;}
    }
    o[blockIdx.x*blockDim.x+threadIdx.x]=a0+a1+a2+a3+a4+a5+a6+a7;
}
// 1 dependent chain of add/mul alternating
__global__ void k_dep(float* o, float s) {
    float a=threadIdx.x;
    for (int i=0;i<ITER;++i){
#pragma unroll
        for(int u=0;u<16;++u){a=a+s; a=a*s;}
    }
    o[blockIdx.x*blockDim.x+threadIdx.x]=a;
}
// 2 dependent chains interleaved
__global__ void k_dep2(float* o, float s) {
    float a=threadIdx.x,b=a+1;
    for (int i=0;i<ITER;++i){
#pragma unroll
        for(int u=0;u<8;++u){a=a+s; b=b+s; a=a*s; b=b*s;}
    }
    o[blockIdx.x*blockDim.x+threadIdx.x]=a+b;
}
// independent dpp adds (wave_shr)
__global__ void k_dpp_wave(float* o, float s) {
    float a0=threadIdx.x,a1=a0+1,a2=a0+2,a3=a0+3,a4=a0+4,a5=a0+5,a6=a0+6,a7=a0+7;
    float x=s+threadIdx.x;
    for (int i=0;i<ITER;++i){
#pragma unroll
        for(int u=0;u<4;++u){a0+=dpp_shr(x);a1+=dpp_shr(x);a2+=dpp_shr(x);a3+=dpp_shr(x);a4+=dpp_shr(x);a5+=dpp_shr(x);a6+=dpp_shr(x);a7+=dpp_shr(x);}
    }
    o[blockIdx.x*blockDim.x+threadIdx.x]=a0+a1+a2+a3+a4+a5+a6+a7;
}
__global__ void k_dpp_row(float* o, float s) {
    float a0=threadIdx.x,a1=a0+1,a2=a0+2,a3=a0+3,a4=a0+4,a5=a0+5,a6=a0+6,a7=a0+7;
    float x=s+threadIdx.x;
    for (int i=0;i<ITER;++i){
#pragma unroll
        for(int u=0;u<4;++u){a0+=dpp_rowshr(x);a1+=dpp_rowshr(x);a2+=dpp_rowshr(x);a3+=dpp_rowshr(x);a4+=dpp_rowshr(x);a5+=dpp_rowshr(x);a6+=dpp_rowshr(x);a7+=dpp_rowshr(x);}
    }
    o[blockIdx.x*blockDim.x+threadIdx.x]=a0+a1+a2+a3+a4+a5+a6+a7;
}
// dependent chain where every 8th op is a dpp of the running value (the SOR update shape)
__global__ void k_sor_shape(float* o, float s, float w) {
    float own=threadIdx.x, oc=own+1, sn=own+2, nn=own+3, d=own*0.5f;
    for (int i=0;i<ITER;++i){
#pragma unroll
        for(int u=0;u<4;++u){
            float sum=((dpp_shr(oc)+oc)+sn)+nn;
            float gs=-0.25f*(d-sum);
            float nw=w*own+s*gs;
            nn=sn; sn=oc; oc=nw; own=own+d;   // rotate so the next update depends on this one
        }
    }
    o[blockIdx.x*blockDim.x+threadIdx.x]=own+oc+sn+nn;
}
// LDS: ds_read_b32 lane-linear, independent
__global__ void k_lds(float* o, float s) {
    __shared__ float m[4][64*20];
    float* p=&m[threadIdx.x>>6][threadIdx.x&63];
    for(int k=0;k<20;++k)p[k*64]=s+k;
    float a0=0,a1=0,a2=0,a3=0;
    for (int i=0;i<ITER;++i){
#pragma unroll
        for(int u=0;u<4;++u){a0+=p[(u*4+0)*64];a1+=p[(u*4+1)*64];a2+=p[(u*4+2)*64];a3+=p[(u*4+3)*64];}
        p[(i%20)*64]=a0;
    }
    o[blockIdx.x*blockDim.x+threadIdx.x]=a0+a1+a2+a3;
}

template<class F> double run(F f, int blocks, float* o, const char* name, double instr_per_iter) {
    hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(blocks); hipDeviceSynchronize();
    hipEventRecord(e0); f(blocks); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms,e0,e1);
    // per SIMD: waves per SIMD = blocks / CUs (256 threads = 4 waves = 1 per SIMD)
    double waves_per_simd = blocks/256.0;
    double instr = instr_per_iter*ITER*waves_per_simd;   // wave-instructions issued per SIMD
    double ns_per = ms*1e6/instr;
    printf("%-14s blocks=%5d waves/SIMD=%4.1f  %.3f ms  %.3f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n",
           name, blocks, waves_per_simd, ms, ns_per, ns_per*2.4);
    return ns_per;
}

int main(){
    float* o; hipMalloc(&o, 8192*256*4);
    for (int wps : {1,2,4,8}) {
        int blocks=256*wps;
        run([&](int b){ k_add_indep<<<b,256>>>(o,1.5f); }, blocks,o,"add_indep",32);
        run([&](int b){ k_dep<<<b,256>>>(o,1.0001f); }, blocks,o,"dep_chain",32);
        run([&](int b){ k_dep2<<<b,256>>>(o,1.0001f); }, blocks,o,"dep_chain x2",32);
        run([&](int b){ k_dpp_wave<<<b,256>>>(o,1.5f); }, blocks,o,"dpp_wave_shr",32);
        run([&](int b){ k_dpp_row<<<b,256>>>(o,1.5f); }, blocks,o,"dpp_row_shr",32);
        run([&](int b){ k_sor_shape<<<b,256>>>(o,1.96f,-0.96f); }, blocks,o,"sor_shape",4*9);
        run([&](int b){ k_lds<<<b,256>>>(o,1.5f); }, blocks,o,"lds_read_b32",16);
    }
    return 0;
}
