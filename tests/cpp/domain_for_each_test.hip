// domain_for_each_test.hip -- user-supplied per-cell expressions on the GPU through sfl/operations.h's
// domain_for_each / domain_for_each_red_black (the device-functor generalisation of the reference's domain_iter,
// operations.h:11-38, and of its colour-split sibling, poisson.cpp:14-61).  The functors below are a USER's
// restatement of the sketch's expression pairs -- the point of the test is that arbitrary safe / fast pairs run on
// the device with the reference's cell-pointer convention, out of place AND in place.
// TEST PROGRAM.  argv[1] = mode:
//   div        argv[2] = input (dim_x, dim_y, velocity), argv[3] = output (float field): out of place, T != U
//   inplace    prints case A of tests/cpp/domain_iter_driver.cpp (order-SENSITIVE expressions, wrt == rd) computed on
//              the device, in the driver's format; then checks a larger grid against the host domain_iter
//   pointwise  argv[2] = input (dim_x, dim_y, velocity, pressure), argv[3] = output (velocity): subtract_gradient's
//              expressions in place, every cell at once (they read only the centre of the field they rewrite)
//   redblack   argv[2] = input (dim_x, dim_y, iters, rhs), argv[3] = output (pressure): the SOR expressions through the
//              colour-split in-place driver
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "sfl/operations.h"
#include "sfl/vector.h"

// ---- calculate_divergence's expressions (finitediff.cpp:9-31) ----------------------------------------------
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

// ---- the order-sensitive pair of tests/cpp/domain_iter_driver.cpp, as functors (host and device) ------------
struct MixFast {
    float scale;
    __host__ __device__ float operator()(const float *c, int i, int j, int dim_x, int) const
    {
        return ((c[-1] * 0.5f - c[1] * 0.25f) + (c[-dim_x] * 0.125f - c[dim_x] * 0.0625f)) * scale + *c +
               float(i) * 0.001f + float(j) * 0.01f;
    }
};
struct MixSafe {
    float scale;
    __host__ __device__ float operator()(const float *c, int i, int j, int dim_x, int dim_y) const
    {
        float acc = *c * 3.0f;
        if (i > 0) acc = acc * 0.5f + c[-1];
        if (i < dim_x - 1) acc = acc * 0.75f - c[1];
        if (j > 0) acc = acc * 1.25f + c[-dim_x];
        if (j < dim_y - 1) acc = acc * 0.875f - c[dim_x];
        return acc * scale;
    }
};
static float mix_fast_host(float *c, int i, int j, int dim_x, int dim_y, void *ctx)
{
    return MixFast{*static_cast<float *>(ctx)}(c, i, j, dim_x, dim_y);
}
static float mix_safe_host(float *c, int i, int j, int dim_x, int dim_y, void *ctx)
{
    return MixSafe{*static_cast<float *>(ctx)}(c, i, j, dim_x, dim_y);
}

// ---- subtract_gradient's expressions (finitediff.cpp:41-73): the pressure is reached through the functor ------
struct GradSub {
    const float *p;
    float two_dx_inv;
    __device__ Vector2<float> operator()(const Vector2<float> *v, int i, int j, int dim_x, int dim_y) const
    {
        const float *pc = p + index(i, j, dim_x);
        const float pw = i > 0 ? pc[-1] : *pc, pe = i < dim_x - 1 ? pc[1] : *pc;
        const float ps = j > 0 ? pc[-dim_x] : *pc, pn = j < dim_y - 1 ? pc[dim_x] : *pc;
        return Vector2<float>(v->x - (pe - pw) * two_dx_inv, v->y - (pn - ps) * two_dx_inv);
    }
};

// ---- the SOR expressions (poisson.cpp:63-112) ------------------------------------------------------------------
struct SorSafe {
    const float *d;
    float dx, omega;
    __device__ float operator()(const float *p, int i, int j, int dim_x, int dim_y) const
    {
        float sum = 0;
        int n = 0;
        if (i > 0) { sum += p[-1]; ++n; }
        if (i < dim_x - 1) { sum += p[1]; ++n; }
        if (j > 0) { sum += p[-dim_x]; ++n; }
        if (j < dim_y - 1) { sum += p[dim_x]; ++n; }
        const float k = n == 2 ? (float)(-1.0 / 2.0) : n == 3 ? (float)(-1.0 / 3.0) : -0.25f;
        const float gs = k * (dx * d[index(i, j, dim_x)] - sum);
        return (1 - omega) * *p + omega * gs;
    }
};
struct SorFast {
    const float *d;
    float dx, omega;
    __device__ float operator()(const float *p, int i, int j, int dim_x, int) const
    {
        const float sum = p[-1] + p[1] + p[-dim_x] + p[dim_x];
        const float gs = -0.25f * (dx * d[index(i, j, dim_x)] - sum);
        return (1 - omega) * *p + omega * gs;
    }
};

#define CHECK(x)                                                                            \
    do {                                                                                    \
        if ((x) != hipSuccess) {                                                            \
            std::fprintf(stderr, "%s failed (line %d)\n", #x, __LINE__);                    \
            return 4;                                                                       \
        }                                                                                   \
    } while (0)

template <class T>
static bool read_all(FILE *in, std::vector<T> &a)
{
    return std::fread(a.data(), sizeof(T), a.size(), in) == a.size();
}

static int run_div(const char *fin, const char *fout)
{
    FILE *in = std::fopen(fin, "rb");
    int dims[2];
    if (!in || std::fread(dims, sizeof(int), 2, in) != 2) return 3;
    const size_t n = (size_t)dims[0] * dims[1];
    std::vector<Vector2<float>> v(n);
    if (!read_all(in, v)) return 3;
    std::fclose(in);
    Vector2<float> *dv;
    float *dd;
    CHECK(hipMalloc(&dv, n * 8));
    CHECK(hipMalloc(&dd, n * 4));
    CHECK(hipMemcpy(dv, v.data(), n * 8, hipMemcpyHostToDevice));
    const float k = 1.0f / (2.0f * 1.0f);
    CHECK((domain_for_each<Vector2<float>, float>(DivSafe{k}, DivFast{k}, dd, dv, dims[0], dims[1])));
    std::vector<float> out(n);
    CHECK(hipMemcpy(out.data(), dd, n * 4, hipMemcpyDeviceToHost));
    FILE *o = std::fopen(fout, "wb");
    std::fwrite(out.data(), 4, n, o);
    std::fclose(o);
    return 0;
}

static int in_place_on_device(std::vector<float> &f, int dim_x, int dim_y, float scale)
{
    float *d;
    CHECK(hipMalloc(&d, f.size() * 4));
    CHECK(hipMemcpy(d, f.data(), f.size() * 4, hipMemcpyHostToDevice));
    CHECK((domain_for_each<float, float>(MixSafe{scale}, MixFast{scale}, d, d, dim_x, dim_y)));   // wrt == rd
    CHECK(hipMemcpy(f.data(), d, f.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipFree(d));
    return 0;
}

static int run_inplace()
{
    const int shapes[][2] = {{2, 2}, {3, 2}, {2, 3}, {5, 4}, {33, 17}, {16, 9}};   // as domain_iter_driver.cpp
    for (const auto &s : shapes) {
        const int dim_x = s[0], dim_y = s[1], n = dim_x * dim_y;
        std::vector<float> f(n);
        uint32_t lcg = 12345u + 977u * dim_x + dim_y;
        for (int k = 0; k < n; ++k) {
            lcg = lcg * 1664525u + 1013904223u;
            f[k] = float(int((lcg >> 8) % 2001) - 1000) / 250.0f;
        }
        if (int rc = in_place_on_device(f, dim_x, dim_y, 0.375f)) return rc;
        std::printf("A %dx%d", dim_x, dim_y);
        for (float x : f) {
            uint32_t u;
            std::memcpy(&u, &x, 4);
            std::printf(" %08x", u);
        }
        std::printf("\n");
    }
    // larger grids (more diagonals than threads, both orientations) against the host driver of the same header
    const int big[][2] = {{1500, 40}, {37, 2100}, {300, 200}};
    for (const auto &s : big) {
        const int dim_x = s[0], dim_y = s[1], n = dim_x * dim_y;
        std::vector<float> f(n), g;
        uint32_t lcg = 99u + dim_x;
        for (int k = 0; k < n; ++k) {
            lcg = lcg * 1664525u + 1013904223u;
            f[k] = float(int((lcg >> 8) % 2001) - 1000) / 500.0f;
        }
        g = f;
        float scale = 0.1f;
        domain_iter<float, float>(mix_safe_host, mix_fast_host, g.data(), g.data(), dim_x, dim_y, &scale);
        if (int rc = in_place_on_device(f, dim_x, dim_y, scale)) return rc;
        std::printf("B %dx%d %s\n", dim_x, dim_y, std::memcmp(f.data(), g.data(), (size_t)n * 4) == 0 ? "same" : "DIFFERENT");
    }
    return 0;
}

static int run_pointwise(const char *fin, const char *fout)
{
    FILE *in = std::fopen(fin, "rb");
    int dims[2];
    if (!in || std::fread(dims, sizeof(int), 2, in) != 2) return 3;
    const size_t n = (size_t)dims[0] * dims[1];
    std::vector<Vector2<float>> v(n);
    std::vector<float> p(n);
    if (!read_all(in, v) || !read_all(in, p)) return 3;
    std::fclose(in);
    Vector2<float> *dv;
    float *dp;
    CHECK(hipMalloc(&dv, n * 8));
    CHECK(hipMalloc(&dp, n * 4));
    CHECK(hipMemcpy(dv, v.data(), n * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dp, p.data(), n * 4, hipMemcpyHostToDevice));
    const GradSub e{dp, 1.0f / (2.0f * 1.0f)};
    CHECK((domain_for_each<Vector2<float>, Vector2<float>>(e, e, dv, dv, dims[0], dims[1], nullptr, sfl_in_place::pointwise)));
    CHECK(hipMemcpy(v.data(), dv, n * 8, hipMemcpyDeviceToHost));
    FILE *o = std::fopen(fout, "wb");
    std::fwrite(v.data(), 8, n, o);
    std::fclose(o);
    return 0;
}

static int run_redblack(const char *fin, const char *fout)
{
    FILE *in = std::fopen(fin, "rb");
    int hdr[3];
    if (!in || std::fread(hdr, sizeof(int), 3, in) != 3) return 3;
    const size_t n = (size_t)hdr[0] * hdr[1];
    std::vector<float> d(n), p(n, 0.0f);   // poisson.cpp:117-119: p starts at zero
    if (!read_all(in, d)) return 3;
    std::fclose(in);
    float *dd, *dp;
    CHECK(hipMalloc(&dd, n * 4));
    CHECK(hipMalloc(&dp, n * 4));
    CHECK(hipMemcpy(dd, d.data(), n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dp, p.data(), n * 4, hipMemcpyHostToDevice));
    for (int it = 0; it < hdr[2]; ++it)   // poisson.cpp:121-124
        CHECK((domain_for_each_red_black<float>(SorSafe{dd, 1.0f, 1.96f}, SorFast{dd, 1.0f, 1.96f}, dp, hdr[0], hdr[1])));
    CHECK(hipMemcpy(p.data(), dp, n * 4, hipMemcpyDeviceToHost));
    FILE *o = std::fopen(fout, "wb");
    std::fwrite(p.data(), 4, n, o);
    std::fclose(o);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    const std::string mode = argv[1];
    if (mode == "inplace") return run_inplace();
    if (argc != 4) return 2;
    if (mode == "div") return run_div(argv[2], argv[3]);
    if (mode == "pointwise") return run_pointwise(argv[2], argv[3]);
    if (mode == "redblack") return run_redblack(argv[2], argv[3]);
    return 2;
}
