// launch_stubs_ok.cpp -- TEST HARNESS ONLY (tests/cpp, `make san_host`): every launcher of csrc/kernels.h as a stub that does NOTHING
// and reports success, so that the HOST side of the library (contexts, options, plans, transports, executors) links without the
// gfx950 kernels and can be RUN THROUGH on the CPU over the fake runtime of fake_hip.cpp (ThreadSanitizer: `make tsan_host`).  Generated from kernels.h by
// tests/cpp/make_launch_stubs.py; never part of the product.
#include "kernels.h"

namespace sfl {
hipError_t launch_advect_vec2f(hipStream_t s, float *next_p, const float *p, const float *vel, Slab g, int g_begin, int g_end, int valid_begin, int valid_end, float dt, bool no_slip, int *halo_flag, const Slab *src, int kernel, int g2_begin, int g2_end) { return hipSuccess; }
hipError_t launch_advect_vec3uq32(hipStream_t s, uint32_t *next_p, const uint32_t *p, const float *vel, Slab g, int g_begin, int g_end, int valid_begin, int valid_end, float dt, bool no_slip, int *halo_flag, const Slab *src, int kernel) { return hipSuccess; }
hipError_t launch_backtrace_reach(hipStream_t s, int *reach, const float *vel, Slab g, int g_begin, int g_end, float dt) { return hipSuccess; }
hipError_t launch_project_advect_vec3uq32(hipStream_t s, uint32_t *next_p, const uint32_t *p, float *vel, const float *pressure, Slab g, int g_begin, int g_end, int valid_begin, int valid_end, float dt, bool no_slip, int *halo_flag, float two_dx_inv, int kernel, bool *reach_measured) { return hipSuccess; }
hipError_t launch_advect_vec2f_tiled(hipStream_t s, float *next_p, const float *p, const float *vel, Slab g, int g_begin, int g_end, int valid_begin, int valid_end, float dt, bool no_slip, int *halo_flag, const Slab *src, int g2_begin, int g2_end) { return hipSuccess; }
hipError_t launch_advect_vec3uq32_tiled(hipStream_t s, uint32_t *next_p, const uint32_t *p, float *vel, const float *pressure, Slab g, int g_begin, int g_end, int valid_begin, int valid_end, float dt, bool no_slip, int *halo_flag, float two_dx_inv, const Slab *src, bool reach) { return hipSuccess; }
hipError_t launch_advect_divergence_tiled(hipStream_t s, float *next_v, float *div, const float *v, Slab g, float dt, bool no_slip, float two_dx_inv) { return hipSuccess; }
hipError_t launch_advect_channels(hipStream_t s, void *next_p, const void *p, const float *vel, int dim_x, int dim_y, float dt, bool no_slip, int channels, int kind) { return hipSuccess; }
hipError_t launch_step_seam_tiled(hipStream_t s, uint32_t *next_col, const uint32_t *col, float *next_v, float *div, const float *v, const float *pressure, Slab g, float dt, float two_dx_inv) { return hipSuccess; }
hipError_t launch_divergence(hipStream_t s, float *div, const float *v, Slab g, int g_begin, int g_end, float two_dx_inv, int kernel) { return hipSuccess; }
hipError_t launch_subtract_gradient(hipStream_t s, float *v, const float *p, Slab g, int g_begin, int g_end, float two_dx_inv, int kernel) { return hipSuccess; }
hipError_t launch_divergence_tiled(hipStream_t s, float *div, const float *v, Slab g, int g_begin, int g_end, float two_dx_inv) { return hipSuccess; }
hipError_t launch_gradient_tiled(hipStream_t s, float *v, const float *p, Slab g, int g_begin, int g_end, float two_dx_inv) { return hipSuccess; }
hipError_t launch_sor_half_sweep(hipStream_t s, float *p, const float *d, Slab g, int g_begin, int g_end, int colour, SorParams prm) { return hipSuccess; }
hipError_t launch_sor_fused(hipStream_t s, float *p_out, const float *p_in, const float *d, Slab g, SorRows rows, int nsweeps, int first_colour, SorParams prm, int rows_per_chunk, int sweep, const HaloWait *wait, int *senders) { if (senders) *senders = 0; return hipSuccess; }
hipError_t launch_signal_arrival(hipStream_t s, int *flag, int value) { return hipSuccess; }
hipError_t launch_wait_count(hipStream_t s, const int *count, int target, int *timed_out, int timeout_us) { return hipSuccess; }
bool small_grid_fits(int dim_x, int dim_y) { return false; }
hipError_t launch_small_solve(hipStream_t s, float *p, const float *d, int dim_x, int dim_y, int iters, SorParams prm) { return hipSuccess; }
hipError_t launch_small_step(hipStream_t s, const SmallStep &a) { return hipSuccess; }
hipError_t launch_zero_rows(hipStream_t s, float *f, Slab g, int g_begin, int g_end) { return hipSuccess; }
hipError_t launch_apply_forces(hipStream_t s, float *v, Slab g, int g_begin, int g_end, const int *cells_ij, const float *vel_xy, int n) { return hipSuccess; }
hipError_t launch_copy_bands(hipStream_t s, void *dst_a, const void *src_a, void *dst_b, const void *src_b, size_t bytes) { return hipSuccess; }
hipError_t launch_spin_us(hipStream_t s, int us) { return hipSuccess; }
hipError_t launch_setup_sketch_fields(hipStream_t s, float *v, uint32_t *colour, int dim_x, int dim_y) { return hipSuccess; }
hipError_t launch_render_rgb565(hipStream_t s, uint16_t *image, const uint32_t *colour, int dim_x, int dim_y, int scaling, bool byteswap) { return hipSuccess; }

}  // namespace sfl
