// sample_driver.cpp -- TEST HARNESS.  Calls the per-point helpers every includer of "advect.h"
// sees (advect.h:10-72: lerp, billinear_interpolate, sample, TPromoted) from whichever "advect.h"
// the include path offers -- include/sfl or the reference's directory -- on a small grid whose
// back-traces hit every branch (interior, each wall, each corner, no-slip on / off; both element
// types of the sketch) and prints the resulting bits.  tests/test_dropin_headers.py compares the
// two builds byte for byte.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <type_traits>
#include <vector>

#include "advect.h"
#include "uq32.h"

static_assert(std::is_same<TPromoted<Vector3<UQ32>>, Vector3<float>>::value, "promotion of the dye element");
static_assert(std::is_same<TPromoted<Vector2<float>>, Vector2<float>>::value, "promotion of the velocity element");
static_assert(std::is_same<TPromoted<float>, float>::value, "promotion of a scalar");

int main()
{
    const int dim_x = 9, dim_y = 7, n = dim_x * dim_y;
    std::vector<Vector2<float>> v(n);
    std::vector<Vector3<UQ32>> c(n);
    uint32_t s = 99;
    auto next = [&] { return s = s * 1664525u + 1013904223u; };
    for (int k = 0; k < n; ++k) {
        v[k].x = float(int((next() >> 8) % 2001) - 1000) / 100.0f;   // up to +-10 cells/s
        v[k].y = float(int((next() >> 8) % 2001) - 1000) / 100.0f;
        c[k].x.raw = next() >> 1;
        c[k].y.raw = next() >> 1;
        c[k].z.raw = next() >> 1;
    }
    for (int no_slip = 0; no_slip < 2; ++no_slip)
        for (int k = 0; k < n; ++k) {
            const int i = k % dim_x, j = k / dim_x;
            const float si = i - v[k].x * 0.3f, sj = j - v[k].y * 0.3f;   // up to 3 cells away
            const Vector2<float> a = sample(v.data(), si, sj, dim_x, dim_y, no_slip != 0);
            const Vector3<UQ32> b = sample(c.data(), si, sj, dim_x, dim_y, no_slip != 0);
            uint32_t u[2];
            std::memcpy(u, &a, 8);
            std::printf("%d %d %d %08x %08x %08x %08x %08x\n", no_slip, i, j, u[0], u[1], b.x.raw, b.y.raw, b.z.raw);
        }
    float f[4] = {1.1f, 2.3f, -3.7f, 4.9f};
    std::printf("%a %a\n", (double)lerp(0.3f, f[0], f[1]),
                (double)billinear_interpolate(0.3f, 0.6f, f[0], f[1], f[2], f[3]));
    return 0;
}
