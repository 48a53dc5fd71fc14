// dropin_loop.cpp -- a caller written the way the reference sketch's loop() is written
// (ESP32-fluid-simulation.ino:249-289: raw new[] / delete[] buffers, the five operator calls in
// order, pointer swaps), compiled against include/sfl/*.h and linked with libsfl_dropin.so.
// It proves the drop-in claim: reference-style caller code builds and runs unchanged on the GPU
// path.  TEST PROGRAM: reads initial fields from argv[1], runs argv[2] steps, writes the final
// velocity / divergence / pressure / colour to argv[3]; pytest compares them with the oracle.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "sfl/advect.h"
#include "sfl/finitediff.h"
#include "sfl/operations.h"
#include "sfl/poisson.h"
#include "sfl/uq32.h"
#include "sfl/vector.h"

#define SWAP(x, y) do { auto temp = (x); (x) = (y); (y) = temp; } while (0)

static int N_ROWS, N_COLS;
static const float DT = 1 / 30.0f;
static Vector2<float> *velocity_field;
static Vector3<UQ32> *color_field;
static float *div_v, *p;

static void sim_step(int iters)
{
    Vector2<float> *v_temp = new Vector2<float>[N_ROWS * N_COLS];
    advect(v_temp, velocity_field, velocity_field, N_ROWS, N_COLS, DT, true);
    SWAP(v_temp, velocity_field);
    delete[] v_temp;

    calculate_divergence(div_v, velocity_field, N_ROWS, N_COLS, 1);
    poisson_solve(p, div_v, N_ROWS, N_COLS, 1, iters, 1.96);
    subtract_gradient(velocity_field, p, N_ROWS, N_COLS, 1);

    Vector3<UQ32> *c_temp = new Vector3<UQ32>[N_ROWS * N_COLS];
    advect(c_temp, color_field, velocity_field, N_ROWS, N_COLS, DT, false);
    SWAP(c_temp, color_field);
    delete[] c_temp;
}

int main(int argc, char **argv)
{
    if (argc != 4) return 2;
    FILE *in = std::fopen(argv[1], "rb");
    if (!in) return 3;
    int hdr[3];
    if (std::fread(hdr, sizeof(int), 3, in) != 3) return 3;
    N_ROWS = hdr[0];  // dim_x, named like the sketch (ino:37-38, :253)
    N_COLS = hdr[1];
    const int iters = hdr[2];
    const size_t n = static_cast<size_t>(N_ROWS) * N_COLS;
    velocity_field = new Vector2<float>[n];
    color_field = new Vector3<UQ32>[n];
    div_v = new float[n];
    p = new float[n];
    if (std::fread(velocity_field, sizeof(Vector2<float>), n, in) != n) return 3;
    if (std::fread(color_field, sizeof(Vector3<UQ32>), n, in) != n) return 3;
    std::fclose(in);

    // a little of the element-type API, as setup() / touch handling use it (ino:199, :266-268)
    Vector2<float> probe(0, 0);
    probe += Vector2<float>(1.5f, -2.0f) * 2.0f;
    probe = probe - 0.5f * Vector2<int>(2, 4);
    if (probe.x != 2.0f || probe.y != -6.0f || index(3, 2, N_ROWS) != 2 * N_ROWS + 3) return 4;

    const int steps = std::atoi(argv[2]);
    for (int s = 0; s < steps; ++s) sim_step(iters);

    FILE *out = std::fopen(argv[3], "wb");
    if (!out) return 5;
    std::fwrite(velocity_field, sizeof(Vector2<float>), n, out);
    std::fwrite(div_v, sizeof(float), n, out);
    std::fwrite(p, sizeof(float), n, out);
    std::fwrite(color_field, sizeof(Vector3<UQ32>), n, out);
    std::fclose(out);
    delete[] velocity_field;
    delete[] color_field;
    delete[] div_v;
    delete[] p;
    return 0;
}
