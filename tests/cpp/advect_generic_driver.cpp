// advect_generic_driver.cpp -- TEST HARNESS.  Calls `advect<T, U>` (advect.h:74-85) from whichever "advect.h"
// the include path offers -- the reference's directory (host loops: the fixture, written by
// tests/golden/make_header_goldens.py) or include/sfl (GPU kernels behind libsfl_dropin.so) -- for element
// types OTHER than the sketch's two instantiations, on a small grid whose back-traces hit every branch of
// sample() (interior, each wall, each corner, no-slip on / off), and prints the resulting bits.
//
//   built by g++:    the element types the library holds kernels for (float, UQ32, Vector2<UQ32>, Vector3<float>)
//   built by hipcc:  (-DDRIVER_ANY_TYPE) additionally types nobody has seen before -- Vector2<double> elements, a
//                    Vector2<double> velocity -- for which include/sfl instantiates a kernel from the caller's type
//
// tests/test_dropin_headers.py compares the builds line by line.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "advect.h"
#include "uq32.h"

static uint32_t lcg_state = 4711;
static uint32_t next_u32() { return lcg_state = lcg_state * 1664525u + 1013904223u; }
static float next_unit() { return float(int((next_u32() >> 8) % 2001) - 1000) / 1000.0f; }

template <class T>
static void dump(const char *tag, int no_slip, const std::vector<T> &a)
{
    std::printf("%s no_slip=%d", tag, no_slip);
    const unsigned char *b = reinterpret_cast<const unsigned char *>(a.data());
    for (size_t k = 0; k < a.size() * sizeof(T); k += 4) {
        uint32_t u;
        std::memcpy(&u, b + k, 4);
        std::printf(" %08x", u);
    }
    std::printf("\n");
}

// (every case seeds the generator itself: a build that runs fewer cases prints the same bits for those it runs)
template <class T, class U, class Fill>
static void run(const char *tag, int ordinal, int dim_x, int dim_y, std::vector<Vector2<U>> &vel, Fill fill)
{
    const int n = dim_x * dim_y;
    std::vector<T> p(n), out(n);
    lcg_state = 1000u * ordinal + 31u * dim_x + dim_y;
    for (int k = 0; k < n; ++k) p[k] = fill();
    for (int no_slip = 0; no_slip < 2; ++no_slip) {
        advect(out.data(), p.data(), vel.data(), dim_x, dim_y, 0.3f, no_slip != 0);
        dump(tag, no_slip, out);
    }
}

int main()
{
    const int shapes[][2] = {{9, 7}, {2, 2}, {31, 18}};
    for (const auto &s : shapes) {
        const int dim_x = s[0], dim_y = s[1], n = dim_x * dim_y;
        std::vector<Vector2<float>> vel(n);
        lcg_state = 4711u + 977u * dim_x + dim_y;
        // (draws are sequenced by statements: the order in which a compiler evaluates call arguments is its own)
        for (int k = 0; k < n; ++k) {
            const float vx = next_unit() * 10.0f, vy = next_unit() * 10.0f;   // up to 3 cells per step
            vel[k] = Vector2<float>(vx, vy);
        }
        std::printf("# %d x %d\n", dim_x, dim_y);
        run<float, float>("float", 1, dim_x, dim_y, vel, [] { return next_unit() * 4.0f; });
        run<UQ32, float>("UQ32", 2, dim_x, dim_y, vel, [] { UQ32 c; c.raw = next_u32() >> 1; return c; });
        run<Vector2<UQ32>, float>("Vector2<UQ32>", 3, dim_x, dim_y, vel, [] {
            Vector2<UQ32> c;
            c.x.raw = next_u32() >> 1;
            c.y.raw = next_u32() >> 9;   // small raws too: exact in float
            return c;
        });
        run<Vector3<float>, float>("Vector3<float>", 4, dim_x, dim_y, vel, [] {
            const float a = next_unit(), b = next_unit() * 100.0f, c = next_unit() * 1e-3f;
            return Vector3<float>(a, b, c);
        });
#ifdef DRIVER_ANY_TYPE
        run<Vector2<double>, float>("Vector2<double>", 5, dim_x, dim_y, vel, [] {
            const double a = double(next_unit()) / 3.0, b = double(next_unit()) * 7.0;
            return Vector2<double>(a, b);
        });
        std::vector<Vector2<double>> vel_d(n);
        for (int k = 0; k < n; ++k) vel_d[k] = Vector2<double>(double(vel[k].x) + 1e-9, double(vel[k].y) - 1e-9);
        run<float, double>("float|Vector2<double> velocity", 6, dim_x, dim_y, vel_d, [] { return next_unit() * 4.0f; });
#endif
    }
    return 0;
}
