// domain_iter_driver.cpp -- TEST HARNESS.  Executes `domain_iter` (operations.h:11-38) from
// whichever "operations.h" the include path offers -- include/sfl (the drop-in header) or the
// reference's directory -- with order-SENSITIVE per-cell expressions, and prints every resulting
// element as hex bits.  tests/test_dropin_headers.py compares the two builds byte for byte.
//
// Case A: in place (wrt == rd), float -> float.  Every expression reads the four neighbours that
//         exist and overwrites the centre, so a cell's result depends on which neighbours were
//         visited before it: any deviation from the reference's visiting order (interior row
//         major; then bottom / top per column, corners included; then left / right per row)
//         changes bits.  The safe and fast expressions differ, so a cell routed to the wrong one
//         shows as well.
// Case B: T != U (int -> float), separate arrays, with a context pointer.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "operations.h"

struct Ctx {
    float scale;
    int calls;
};

static float mix_fast(float *c, int i, int j, int dim_x, int dim_y, void *ctx)
{
    Ctx *k = static_cast<Ctx *>(ctx);
    ++k->calls;
    return ((c[-1] * 0.5f - c[1] * 0.25f) + (c[-dim_x] * 0.125f - c[dim_x] * 0.0625f)) * k->scale + *c +
           float(i) * 0.001f + float(j) * 0.01f;
}

static float mix_safe(float *c, int i, int j, int dim_x, int dim_y, void *ctx)
{
    Ctx *k = static_cast<Ctx *>(ctx);
    ++k->calls;
    float acc = *c * 3.0f;
    if (i > 0) acc = acc * 0.5f + c[-1];
    if (i < dim_x - 1) acc = acc * 0.75f - c[1];
    if (j > 0) acc = acc * 1.25f + c[-dim_x];
    if (j < dim_y - 1) acc = acc * 0.875f - c[dim_x];
    return acc * k->scale;
}

static float count_fast(int *c, int i, int j, int dim_x, int, void *ctx)
{
    return float(*c + c[1] - c[-dim_x]) * static_cast<Ctx *>(ctx)->scale + float(index(i, j, dim_x));
}

static float count_safe(int *c, int i, int j, int dim_x, int, void *ctx)
{
    return -float(*c) * static_cast<Ctx *>(ctx)->scale - float(index(i, j, dim_x));
}

static void dump(const char *tag, int dim_x, int dim_y, const std::vector<float> &a, int calls)
{
    std::printf("%s %dx%d calls=%d", tag, dim_x, dim_y, calls);
    for (float f : a) {
        uint32_t u;
        std::memcpy(&u, &f, 4);
        std::printf(" %08x", u);
    }
    std::printf("\n");
}

int main()
{
    const int shapes[][2] = {{2, 2}, {3, 2}, {2, 3}, {5, 4}, {33, 17}, {16, 9}};
    for (const auto &s : shapes) {
        const int dim_x = s[0], dim_y = s[1], n = dim_x * dim_y;
        std::vector<float> f(n);
        std::vector<int> q(n);
        uint32_t lcg = 12345u + 977u * dim_x + dim_y;
        for (int k = 0; k < n; ++k) {
            lcg = lcg * 1664525u + 1013904223u;
            f[k] = float(int((lcg >> 8) % 2001) - 1000) / 250.0f;
            q[k] = int((lcg >> 20) % 97) - 48;
        }
        Ctx ctx{0.375f, 0};
        domain_iter<float, float>(mix_safe, mix_fast, f.data(), f.data(), dim_x, dim_y, &ctx);  // in place
        dump("A", dim_x, dim_y, f, ctx.calls);
        std::vector<float> g(n, -1.0f);
        Ctx ctx2{1.5f, 0};
        domain_iter<int, float>(count_safe, count_fast, g.data(), q.data(), dim_x, dim_y, &ctx2);
        dump("B", dim_x, dim_y, g, ctx2.calls);
    }
    return 0;
}
