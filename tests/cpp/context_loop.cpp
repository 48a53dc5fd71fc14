// context_loop.cpp -- the production shape of INTEGRATION.md section 3, straight on the C ABI
// (include/sfl.h, no C++ wrapper): fields uploaded once, resident in HBM, sfl_step per frame with
// queued touch forces, download at the end.  TEST PROGRAM: argv[1] = input file (dim_x, dim_y,
// iters, n_forces, velocity, colour, force cells, force velocities), argv[2] = steps,
// argv[3] = output file (velocity, pressure, colour).  Also exercises the error channel.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "sfl.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != SFL_OK) {                                                     \
            std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, sfl_last_error()); \
            return 10;                                                           \
        }                                                                        \
    } while (0)

int main(int argc, char **argv)
{
    if (argc != 4) return 2;
    FILE *in = std::fopen(argv[1], "rb");
    int hdr[4];
    if (!in || std::fread(hdr, sizeof(int), 4, in) != 4) return 3;
    const int dim_x = hdr[0], dim_y = hdr[1], iters = hdr[2], n_forces = hdr[3];
    const size_t n = (size_t)dim_x * dim_y;
    std::vector<float> vel(2 * n), fvel(2 * n_forces), p(n);
    std::vector<uint32_t> col(3 * n);
    std::vector<int> fcells(2 * n_forces);
    if (std::fread(vel.data(), 8, n, in) != n || std::fread(col.data(), 12, n, in) != n) return 3;
    if (n_forces && (std::fread(fcells.data(), 8, n_forces, in) != (size_t)n_forces ||
                     std::fread(fvel.data(), 8, n_forces, in) != (size_t)n_forces)) return 3;
    std::fclose(in);

    // the error channel: bad arguments come back as codes + text, nothing is thrown or printed
    sfl_context *bad = nullptr;
    if (sfl_create(&bad, 0, 1, 8) != SFL_ERR_INVALID || std::strlen(sfl_last_error()) == 0) return 4;
    if (sfl_abi_version() != SFL_ABI_VERSION) return 4;

    sfl_context *sim = nullptr;
    CHECK(sfl_create(&sim, 0, dim_x, dim_y));
    if (sfl_upload(sim, SFL_FIELD_VELOCITY, vel.data(), 8 * n - 4) != SFL_ERR_INVALID) return 5;  // size check
    CHECK(sfl_upload(sim, SFL_FIELD_VELOCITY, vel.data(), 8 * n));
    CHECK(sfl_upload(sim, SFL_FIELD_COLOR, col.data(), 12 * n));
    const int steps = std::atoi(argv[2]);
    for (int s = 0; s < steps; ++s) {
        if (s == 0 && n_forces) CHECK(sfl_queue_forces(sim, fcells.data(), fvel.data(), n_forces));
        CHECK(sfl_step(sim, 1 / 30.0f, 1.0f, iters, 1.96f));
    }
    CHECK(sfl_synchronize(sim));
    CHECK(sfl_download(sim, SFL_FIELD_VELOCITY, vel.data(), 8 * n));
    CHECK(sfl_download(sim, SFL_FIELD_PRESSURE, p.data(), 4 * n));
    CHECK(sfl_download(sim, SFL_FIELD_COLOR, col.data(), 12 * n));
    // zero-copy view (sfl_field_device_ptr): queried AFTER the last operator, it must show exactly
    // what sfl_download returns -- the operators ping-pong their buffers, so a pointer is only good
    // until the next operator that writes the field (include/sfl.h)
    {
        void *dv = nullptr, *dp = nullptr, *dc = nullptr;
        CHECK(sfl_field_device_ptr(sim, SFL_FIELD_VELOCITY, &dv));
        CHECK(sfl_field_device_ptr(sim, SFL_FIELD_PRESSURE, &dp));
        CHECK(sfl_field_device_ptr(sim, SFL_FIELD_COLOR, &dc));
        std::vector<float> v2(2 * n), p2(n);
        std::vector<uint32_t> c2(3 * n);
        if (hipMemcpy(v2.data(), dv, 8 * n, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(p2.data(), dp, 4 * n, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(c2.data(), dc, 12 * n, hipMemcpyDeviceToHost) != hipSuccess)
            return 7;
        if (std::memcmp(v2.data(), vel.data(), 8 * n) || std::memcmp(p2.data(), p.data(), 4 * n) ||
            std::memcmp(c2.data(), col.data(), 12 * n)) {
            std::fprintf(stderr, "sfl_field_device_ptr does not show the current field contents\n");
            return 8;
        }
    }
    CHECK(sfl_destroy(sim));

    FILE *out = std::fopen(argv[3], "wb");
    if (!out) return 6;
    std::fwrite(vel.data(), 8, n, out);
    std::fwrite(p.data(), 4, n, out);
    std::fwrite(col.data(), 12, n, out);
    std::fclose(out);
    return 0;
}
