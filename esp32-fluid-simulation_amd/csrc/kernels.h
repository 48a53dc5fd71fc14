// kernels.h -- launch interface of the gfx950 kernels (internal; the public boundary is
// include/sfl.h).  Every launcher is asynchronous on the given stream and returns the
// hipError_t of the launch.
//
// Geometry.  A context stores each field as a LOCAL array of `lrows` rows of `dim_x`
// elements; local row l holds GLOBAL row grow0 + l of a domain of gdim_y rows (a whole-domain
// context has grow0 = 0, lrows = gdim_y; a slab has ghost rows on both sides, some of which
// may lie outside the domain at the global top / bottom and are then never touched).
// Row ranges passed to launchers are GLOBAL rows [g_begin, g_end), already clipped to the
// domain by the caller.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sfl {

struct Slab {
    int dim_x;   // row length (cells)
    int gdim_y;  // rows of the global domain
    int grow0;   // global row held by local row 0 (may be negative)
    int lrows;   // rows allocated locally
};

// ---- advection (advect.h:24-85) ------------------------------------------------------
// Output rows [g_begin, g_end).  `p` may be read on global rows [valid_begin, valid_end)
// only; a back-trace that needs another row raises *halo_flag (device int, may be null on a
// whole-domain context where every row is present).
// `src` (may be null = g): geometry of the array p when it is not the slab's own local array -- the
// whole domain gathered on this GPU, used when the back-traces outrun the slab's ghost rows.
hipError_t launch_advect_vec2f(hipStream_t s, float *next_p, const float *p, const float *vel,
                               Slab g, int g_begin, int g_end, int valid_begin, int valid_end,
                               float dt, bool no_slip, int *halo_flag, const Slab *src = nullptr,
                               int kernel = 0, int g2_begin = 0, int g2_end = 0);
// (g2_begin < g2_end: a second range of output rows in the same launch -- the two bands of a slab next to its cuts)
hipError_t launch_advect_vec3uq32(hipStream_t s, uint32_t *next_p, const uint32_t *p,
                                  const float *vel, Slab g, int g_begin, int g_end,
                                  int valid_begin, int valid_end, float dt, bool no_slip,
                                  int *halo_flag, const Slab *src = nullptr, int kernel = 0);
// reach[0] / reach[1] (device ints, atomicMax'ed: zero them first) = halo rows the back-traces of
// rows [g_begin, g_end) need below / above that range.
hipError_t launch_backtrace_reach(hipStream_t s, int *reach, const float *vel, Slab g, int g_begin,
                                  int g_end, float dt);

// subtract_gradient (finitediff.cpp:41-82) fused into the dye advection: every cell first projects
// its OWN velocity (in place), then back-traces with it -- ino:276 + ino:282 in one pass over v.
hipError_t launch_project_advect_vec3uq32(hipStream_t s, uint32_t *next_p, const uint32_t *p,
                                          float *vel, const float *pressure, Slab g, int g_begin,
                                          int g_end, int valid_begin, int valid_end, float dt,
                                          bool no_slip, int *halo_flag, float two_dx_inv, int kernel = 0,
                                          bool *reach_measured = nullptr);
// `reach_measured` != nullptr (slabs; halo_flag = word [2] of a reach report, zeroed): when the tile kernel runs it also leaves the
// reach of the projected velocity's back-traces in the report's other words -- what launch_backtrace_reach would measure in three
// launches afterwards -- and sets *reach_measured; the one-thread-per-cell kernel does not.

// `kernel` of the three launchers above: 1 = one thread per cell with a 4-texel gather from memory
// (stencil_kernels.hip), 2 = the source window of a 64 x 32 tile staged in LDS (advect_tiled.hip),
// 0 = automatic (2 from kAdvectTiledMinCells output cells).  Same arithmetic, same bits.
constexpr int kAdvectTiledMinCells = 16384;
hipError_t launch_advect_vec2f_tiled(hipStream_t s, float *next_p, const float *p, const float *vel, Slab g,
                                     int g_begin, int g_end, int valid_begin, int valid_end, float dt,
                                     bool no_slip, int *halo_flag, const Slab *src, int g2_begin = 0, int g2_end = 0);
// pressure != nullptr: the projection fused in (then src must be null)
hipError_t launch_advect_vec3uq32_tiled(hipStream_t s, uint32_t *next_p, const uint32_t *p, float *vel,
                                        const float *pressure, Slab g, int g_begin, int g_end, int valid_begin,
                                        int valid_end, float dt, bool no_slip, int *halo_flag,
                                        float two_dx_inv, const Slab *src, bool reach = false);
// advect(v_next, v, v) (ino:252-256) and calculate_divergence(div, v_next) (ino:274) in one pass of a
// WHOLE-DOMAIN context (grow0 = 0, lrows = gdim_y): the advected tile and the ring around it are
// differenced in LDS.  Same arithmetic as the two operators, same bits.
hipError_t launch_advect_divergence_tiled(hipStream_t s, float *next_v, float *div, const float *v, Slab g,
                                          float dt, bool no_slip, float two_dx_inv);

// advect<T, float> of a WHOLE-DOMAIN field whose element is `channels` (1..3) consecutive 32-bit channels of one
// kind (0 = float, 1 = UQ32 raw): every element type the reference's headers can express (advect_generic.hip).
hipError_t launch_advect_channels(hipStream_t s, void *next_p, const void *p, const float *vel, int dim_x, int dim_y,
                                  float dt, bool no_slip, int channels, int kind);

// The seam between two steps of sfl_step_n on a WHOLE-DOMAIN context: subtract_gradient + dye advection of step k
// (ino:276, :281-287) and velocity advection + divergence of step k + 1 (ino:252-256, :274) in one pass.  Reads v (the
// advected, not yet projected velocity of step k), its pressure and the dye; writes the new dye, the advected velocity
// of step k + 1 and its divergence.  The projected velocity of step k lives in LDS only.  Same bits as the two kernels.
hipError_t launch_step_seam_tiled(hipStream_t s, uint32_t *next_col, const uint32_t *col, float *next_v, float *div,
                                  const float *v, const float *pressure, Slab g, float dt, float two_dx_inv);

// ---- finite differences (finitediff.cpp:9-82) ------------------------------------------
// `kernel` as for the advections: 1 = one thread per cell (stencil_kernels.hip), 2 = the 66 x 34 window of a
// 64 x 32-cell tile staged in LDS (advect_tiled.hip), 0 = automatic (2 from kAdvectTiledMinCells cells).
hipError_t launch_divergence(hipStream_t s, float *div, const float *v, Slab g, int g_begin,
                             int g_end, float two_dx_inv, int kernel = 0);
hipError_t launch_subtract_gradient(hipStream_t s, float *v, const float *p, Slab g, int g_begin,
                                    int g_end, float two_dx_inv, int kernel = 0);
hipError_t launch_divergence_tiled(hipStream_t s, float *div, const float *v, Slab g, int g_begin, int g_end,
                                   float two_dx_inv);
hipError_t launch_gradient_tiled(hipStream_t s, float *v, const float *p, Slab g, int g_begin, int g_end,
                                 float two_dx_inv);

// ---- red-black SOR (poisson.cpp:14-112) ------------------------------------------------
struct SorParams {
    float dx;
    float omega;
    float one_minus_omega;  // (1 - omega) evaluated in float on the host, poisson.cpp:98,111
    float neg_quarter_omega;  // -0.25f * omega, used only when `fold` is set
    int fold;                 // SFL_OPT_SOR_FOLD: 1 = the fused kernel's interior relaxation multiplies ONCE, by -0.25f * omega, where
                              // poisson.cpp:109-111 multiplies by -0.25f and then by omega (sor_stream_core.h relax has the condition under
                              // which that is the same bits); 0 (default) = two products, the reference's bits on every input
};

// Baseline: ONE colour pass, in place, over global rows [g_begin, g_end).
// colour 0 = even (i + j), the first pass of every iteration (poisson.cpp:22,57-60).
hipError_t launch_sor_half_sweep(hipStream_t s, float *p, const float *d, Slab g, int g_begin,
                                 int g_end, int colour, SorParams prm);

// Fused streaming kernel: `nsweeps` (even, 2..SFL_MAX_FUSE) consecutive colour passes,
// starting with `first_colour`, in ONE launch.  Reads p_in (or nothing when p_in == nullptr:
// p is then implicitly zero, the fused zero-fill of poisson.cpp:117-119) and d, writes the
// result for the global rows of `rows` to p_out (p_out must not alias p_in).  Needs p_in
// valid on rows [g_begin - nsweeps, g_end + nsweeps) and d on one row less each side, clipped
// to the domain.  rows_per_chunk = output rows streamed by one wave (0 = auto).  sweep = index of
// the launch within its solve: odd launches stream every tile in the direction opposite to the even
// ones, so that a launch begins on the rows the previous one touched last (still in the Infinity
// Cache); speed only, any value gives the same bits.
// One launch covers output rows [g_begin, g_end) and, optionally, a second disjoint range
// [g2_begin, g2_end) (the two cut-adjacent bands of a slab around a halo exchange in ONE launch).
#define SFL_MAX_FUSE 16
struct SorRows {
    int g_begin, g_end;
    int g2_begin, g2_end;
};
// Device-side halo arrival.  A launch may start BEFORE a halo message it depends on has arrived: its tiles whose
// input rows all lie inside [own_lo, own_hi) -- rows no message writes -- run at once, the others (the tiles next
// to a cut) first wait, inside the launch, until *flag has reached `epoch` (signed distance: the word only
// counts up, launch_signal_arrival), then make the arrived rows visible to their CU (agent-scope acquire).  A wait
// that lasts longer than HaloWait::timeout_us gives up and raises *timed_out (results are then wrong: the host turns
// the word into an error): a lost message must never hang the GPU.  flag == nullptr: nobody waits.
//
// The other direction, in the same struct: `done` != nullptr makes the tiles whose output rows reach below send_lo_end or
// above send_hi_begin -- the rows the NEXT halo message carries -- SENDERS: they run at the top issue priority, and when
// their stores have been written back (agent-scope release) each adds one to *done.  The exchange stream waits for the
// count (launch_wait_count) instead of an event behind the whole launch: the message leaves while the rest of the launch
// is still running, and the compute stream carries no event at all.
struct HaloWait {
    const int *flag;
    int *timed_out;
    int epoch;
    int own_lo, own_hi;
    int *done;
    int send_lo_end, send_hi_begin;
    int timeout_us;     // how long a wait may last before it gives up (0: kHaloWaitDefaultTimeoutUs).  Ranks whose peers are
                        // other processes pass minutes: a peer's host may simply be late (Transport::default_timeout_us)
    int system_scope;   // != 0: the arrived rows were written by ANOTHER GPU (RCCL over xGMI): system-scope acquire
};
constexpr int kHaloWaitDefaultTimeoutUs = 2000000;
hipError_t launch_sor_fused(hipStream_t s, float *p_out, const float *p_in, const float *d,
                            Slab g, SorRows rows, int nsweeps, int first_colour,
                            SorParams prm, int rows_per_chunk, int sweep, const HaloWait *wait = nullptr,
                            int *senders = nullptr);   // *senders = tiles of this launch that will add to *wait->done
// *flag = value, visible to every CU (stream-ordered behind the message / the kernels that relaxed it)
hipError_t launch_signal_arrival(hipStream_t s, int *flag, int value);
// One wave that returns when *count has reached `target` (signed distance; sender tiles of launches on ANOTHER stream
// count it up) or after `timeout_us`, raising *timed_out: what follows on the stream starts then.
hipError_t launch_wait_count(hipStream_t s, const int *count, int target, int *timed_out,
                             int timeout_us = kHaloWaitDefaultTimeoutUs);

// ---- small grids: one workgroup, fields in LDS (small_grid.hip) -------------------------------------
// WHOLE-DOMAIN arrays of dim_x * dim_y <= kSmallGridMaxCells cells (16 B of LDS per cell: 96 KB of the CU's 160;
// beyond ~6 K cells one CU is slower than the general kernels on all of them, measured).
constexpr int kSmallGridMaxCells = 6144;
bool small_grid_fits(int dim_x, int dim_y);   // few enough cells, and few enough of one colour per thread
// poisson_solve (poisson.cpp:114-125): p = iters red-black SOR iterations from zero on rhs d, one launch.
hipError_t launch_small_solve(hipStream_t s, float *p, const float *d, int dim_x, int dim_y, int iters, SorParams prm);
// One whole step (ino:252-287): advect v (no-slip) -> forces -> divergence -> solve -> projection -> advect
// dye (free-slip), one launch.  v_out / col_out must not alias the inputs; div and p receive the step's
// divergence and pressure.  n_forces (cell, velocity) pairs as for launch_apply_forces, applied in order.
struct SmallStep {
    const float *v_in;
    float *v_out;
    const uint32_t *col_in;
    uint32_t *col_out;
    float *div, *p;
    int dim_x, dim_y, iters;
    float dt, two_dx_inv;
    SorParams prm;
    const int *force_cells;
    const float *force_vel;
    int n_forces;
};
hipError_t launch_small_step(hipStream_t s, const SmallStep &a);

// Fill rows [g_begin, g_end) of a float field with zero (poisson.cpp:117-119).
hipError_t launch_zero_rows(hipStream_t s, float *f, Slab g, int g_begin, int g_end);

// Point forces (ino:264-269): velocity[cell] = vel for n (cell, vel) pairs whose row lies in
// [g_begin, g_end).  cells = {i, j} pairs (global), device arrays.
hipError_t launch_apply_forces(hipStream_t s, float *v, Slab g, int g_begin, int g_end,
                               const int *cells_ij, const float *vel_xy, int n);

// Two row bands of a halo exchange between arrays of ONE device (virtual ranks, the emulated rank) in ONE launch:
// dst_a <- src_a and dst_b <- src_b, `bytes` each (a null destination is skipped).  The runtime's own copy takes
// 6 - 7 us per 0.85 MB band and runs its two copies one after the other.
hipError_t launch_copy_bands(hipStream_t s, void *dst_a, const void *src_a, void *dst_b, const void *src_b, size_t bytes);

// Measurement aid: one wave idling for `us` microseconds (the emulated wire of sfl_comm_emulate).
hipError_t launch_spin_us(hipStream_t s, int us);

// Initial condition of the sketch's setup() (ino:196-241): zero velocity, three dye sectors, two
// in-place sequential 1-2-1 blurs; WHOLE-DOMAIN arrays.
hipError_t launch_setup_sketch_fields(hipStream_t s, float *v, uint32_t *colour, int dim_x, int dim_y);

// Dye visualiser (draw task, ino:116-176): scaling x scaling bilinear up-scale + RGB565 pack of a
// WHOLE-DOMAIN colour field; image = scaling*(dim_x-1) rows of scaling*(dim_y-1) pixels (device).
hipError_t launch_render_rgb565(hipStream_t s, uint16_t *image, const uint32_t *colour, int dim_x,
                                int dim_y, int scaling, bool byteswap);

}  // namespace sfl
