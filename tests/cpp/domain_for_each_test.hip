// domain_for_each_test.hip -- user-supplied per-cell expressions on the GPU through
// sfl/operations.h's domain_for_each (the device-functor generalisation of the reference's
// domain_iter, operations.h:11-38).  The functors below are a USER's restatement of the
// divergence expressions (finitediff.cpp:9-31) -- the point of the test is that arbitrary
// safe / fast expression pairs run on the device with the reference's cell-pointer convention.
// TEST PROGRAM: argv[1] = input (dim_x, dim_y, velocity), argv[2] = output (float field).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "sfl/operations.h"
#include "sfl/vector.h"

struct DivSafe {
    float two_dx_inv;
    __device__ float operator()(const Vector2<float> *v, int i, int j, int dim_x, int dim_y) const
    {
        float s = 0;
        s += (i > 0) ? -((v - 1)->x) : v->x;
        s += (i < dim_x - 1) ? (v + 1)->x : -(v->x);
        s += (j > 0) ? -((v - dim_x)->y) : v->y;
        s += (j < dim_y - 1) ? (v + dim_x)->y : -(v->y);
        return s * two_dx_inv;
    }
};
struct DivFast {
    float two_dx_inv;
    __device__ float operator()(const Vector2<float> *v, int, int, int dim_x, int) const
    {
        return ((-(v - 1)->x + (v + 1)->x) + (-(v - dim_x)->y + (v + dim_x)->y)) * two_dx_inv;
    }
};

int main(int argc, char **argv)
{
    if (argc != 3) return 2;
    FILE *in = std::fopen(argv[1], "rb");
    int dims[2];
    if (!in || std::fread(dims, sizeof(int), 2, in) != 2) return 3;
    const size_t n = (size_t)dims[0] * dims[1];
    std::vector<Vector2<float>> v(n);
    if (std::fread(v.data(), sizeof(Vector2<float>), n, in) != n) return 3;
    std::fclose(in);
    Vector2<float> *dv;
    float *dd;
    if (hipMalloc(&dv, n * 8) != hipSuccess || hipMalloc(&dd, n * 4) != hipSuccess) return 4;
    hipMemcpy(dv, v.data(), n * 8, hipMemcpyHostToDevice);
    const float k = 1.0f / (2.0f * 1.0f);
    if (domain_for_each<Vector2<float>, float>(DivSafe{k}, DivFast{k}, dd, dv, dims[0], dims[1]) != hipSuccess) return 5;
    std::vector<float> out(n);
    if (hipMemcpy(out.data(), dd, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 6;
    FILE *o = std::fopen(argv[2], "wb");
    std::fwrite(out.data(), 4, n, o);
    std::fclose(o);
    return 0;
}
