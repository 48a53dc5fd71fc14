/*
 * sfl.h -- C ABI of the MI355X-native stable-fluids hot path ("sfl").
 *
 * This is the drop-in boundary for the sim-task arithmetic of
 * colonelwatch/ESP32-fluid-simulation.  Every entry point is `extern "C"`, takes
 * plain pointers / sizes / scalars and returns an int status (0 = SFL_OK, <0 = error,
 * text via sfl_last_error()).  The reference's own functions return void and check
 * nothing; the C++ drop-in headers in include/sfl/ keep those
 * exact signatures on top of this ABI.
 *
 * Citations are file:line under /root/reference/ESP32-fluid-simulation/.
 *
 * Data layout (identical to the reference, operations.h:7-9): element (i, j) of a
 * dim_x * dim_y field at index dim_x*j + i, i fastest.  Velocity = interleaved
 * {x, y} float32 (Vector2<float>, vector.h:4-57, 8 B); dye = interleaved {x, y, z}
 * uint32 raw values (Vector3<UQ32>, vector.h:63-122 + uq32.h:8-16, 12 B);
 * pressure / divergence = float32.
 *
 * Three groups of entry points:
 *   1. host-pointer drop-ins  sfl_host_*      the reference signatures + status; upload,
 *                                             run the HIP kernels, download (parity / porting aid)
 *   2. solver contexts        sfl_create ...  device-resident fields for one GPU's row slab of
 *                                             the domain, operators, RCCL halo exchange
 *   3. utilities              version, errors, device query, slab partition arithmetic
 *
 * There is NO CPU fallback anywhere behind this header: without a usable GPU every
 * compute entry point fails with SFL_ERR_HIP.
 */
#ifndef SFL_H
#define SFL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFL_API __attribute__((visibility("default")))

#define SFL_ABI_VERSION 1

/* ---- status codes ---------------------------------------------------------------- */
#define SFL_OK 0
#define SFL_ERR_INVALID (-1) /* bad argument: NULL, dim < 2, iters < 0, aliasing rule broken   */
#define SFL_ERR_HIP (-2)     /* HIP runtime / no device / kernel launch failure               */
#define SFL_ERR_RCCL (-3)    /* RCCL failure                                                  */
#define SFL_ERR_NOMEM (-4)   /* device or host allocation failed                              */
#define SFL_ERR_STATE (-5)   /* call not valid in this context state (e.g. comm not attached) */
#define SFL_ERR_HALO (-6)    /* a back-trace left the slab's advect halo (multi-GPU only)     */

/* ---- field identifiers of a context ----------------------------------------------- */
#define SFL_FIELD_VELOCITY 0   /* Vector2<float>  velocity_field (ino:54)                    */
#define SFL_FIELD_COLOR 1      /* Vector3<UQ32>   color_field    (ino:55)                    */
#define SFL_FIELD_DIVERGENCE 2 /* float           div_v          (ino:272)                   */
#define SFL_FIELD_PRESSURE 3   /* float           p              (ino:273)                   */

/* ---- tunables (sfl_set_option / sfl_get_option) ------------------------------------------ */
#define SFL_OPT_SOR_KERNEL 0      /* 0 = auto, 1 = one launch per colour pass (baseline kernel),
                                     2 = fused multi-pass streaming kernel.  Both give the reference's
                                     bits on every input (unless SFL_OPT_SOR_FOLD is set)          */
#define SFL_OPT_SOR_FUSE 1        /* colour passes fused per launch by kernel 2: even, 2..16, or
                                     0 = auto (16 on slabs of >= 12 M cells, 10 from 3 M, else 8) */
#define SFL_OPT_ADVECT_HALO 2     /* slabs: rows of the advected field exchanged per side.  0 (default) =
                                     automatic, correct for any velocity: sfl_step sizes the velocity
                                     advection's halo from the reach measured at the end of the previous
                                     step (the same velocity, the same dt: exact), runs the dye advection
                                     on that reach plus a margin and checks it AFTER the step -- a flag and
                                     the true reach travel to the host asynchronously and are examined by
                                     the next call on the context, which repeats the dye advection alone
                                     (from the untouched old buffer) if a back-trace left the guess; no
                                     host round trip inside a step.  The stand-alone sfl_advect_* operators
                                     and a step after the velocity was written from outside measure first
                                     (one host round trip); beyond 64 rows the field is gathered on every
                                     GPU.  1..64 = fixed: a back-trace that leaves them is reported as
                                     SFL_ERR_HALO by sfl_synchronize                                      */
#define SFL_OPT_SOR_ROWS 3        /* output rows per wave tile of kernel 2 (0 = auto)            */
#define SFL_OPT_TRANSPORT 4       /* READ ONLY: 0 = none (whole domain / not attached yet),
                                     1 = RCCL (sfl_comm_attach), 2 = in-process (sfl_group_link),
                                     3 = emulated (sfl_comm_emulate: timing only), 4 = emulated with RCCL
                                     messages to the rank itself (sfl_comm_emulate_rccl: timing only) */
#define SFL_OPT_SOR_LANE_CELLS 5  /* cells per lane of kernel 2: 0 = auto or 2 (the only flavour
                                     left: round 1's packed 4-cell tiles were never faster)        */
#define SFL_OPT_SOR_HALO 6        /* rows of p a superstep's halo makes valid on a slab (kernel 2): 0 =
                                     auto, else fuse..160; larger = fewer, larger exchanges, more recomputed
                                     ghost rows; from 2 x fuse on, exchanges behind events are issued one launch
                                     early (sfl_plan_poisson).  Auto starts from 64 rows on slabs of >= 1024 rows
                                     (32 on thinner ones) and moves to another depth when a model of the solve --
                                     redundant rows against exchanges at their MEASURED cost,
                                     SFL_OPT_MEASURED_WIRE_US -- names another candidate: the first 12 solves of a
                                     kind (same iterations, fuse depth, schedule) then run on the candidates in
                                     turn between pairs of events, the 13th waits for them (one host
                                     hipEventSynchronize) and keeps the fastest (SFL_OPT_LAST_HALO reads what a
                                     solve used).  Same bits at every depth; a caller who times fewer than 13
                                     solves of a kind times the exploration -- or sets a depth                */
#define SFL_OPT_FUSE_PROJECTION 7 /* sfl_step only: 1 (default) = subtract_gradient is applied
                                     inside the dye-advection kernel (one pass over v), 0 = two
                                     kernels                                                     */

#define SFL_OPT_ADVECT_KERNEL 9   /* advection, divergence and gradient kernels: 0 = auto, 1 = one
                                     thread per cell reading its neighbours / texels from memory,
                                     2 = the window of a 64 x 32-cell tile staged in LDS (auto = 2
                                     from 16384 cells per launch); same results                  */
#define SFL_OPT_FUSE_DIVERGENCE 10 /* sfl_step only: 1 (default) = on a whole-domain context with no
                                     queued forces and advection kernel 2, the velocity advection
                                     and calculate_divergence run as one kernel (the advected tile
                                     is differenced in LDS); 0 = two kernels                       */
#define SFL_OPT_SMALL_GRID 11      /* 1 (default) = on a whole-domain context of at most 6144 cells
                                     whose kernel options are all automatic, sfl_poisson_solve and
                                     sfl_step run as ONE launch of one workgroup with the fields in
                                     LDS (the sketch's 61 x 81 grid: one launch instead of six to
                                     ten); 0 = the general kernels                                 */
#define SFL_OPT_EMULATE_WIRE_US 12 /* sfl_comm_emulate / sfl_comm_emulate_rccl only (measurement aid): every emulated halo message is held
                                     back by this many microseconds on the exchange stream before its copy
                                     starts -- the latency of a real xGMI send / receive that a self-copy does
                                     not have; 0 (default) .. 10000                                   */
#define SFL_OPT_STEP_SEAMS 14      /* sfl_step_n on a whole-domain context with the tile kernels: 1 (default) = between two steps
                                     subtract_gradient + dye advection of one and velocity advection + divergence of the
                                     next run as ONE kernel (the projected velocity in between is never written to memory);
                                     0 = n times sfl_step.  Same results either way                              */
#define SFL_OPT_LAST_EARLY_ROWS 17 /* READ ONLY: slabs on the automatic advection halo: L > 0 when the last sfl_step kept the velocity
                                     advection of the rows further than L from both cuts that it had queued BEFORE reading the
                                     previous step's report (they need no halo; the report says whether that held); 0 = the step
                                     advected everything after the report                                            */

#define SFL_OPT_HALO_TIMEOUT_MS 18  /* limit of a wait INSIDE a launch or on the exchange stream (SFL_OPT_EXCHANGE_SCHEDULE = 3),
                                     milliseconds: 0 (default) = the transport's own: 2 s where every party is this process
                                     (virtual ranks, emulated ranks), 300 s on RCCL ranks -- a peer process may simply be late
                                     (I/O, a garbage collection), and what used to be an event wait must not become an error */
#define SFL_OPT_EXCHANGE_SCHEDULE 19 /* slabs, kernel 2: how a solve orders its halo exchanges -- ONE option for what rounds 2 - 5 spread over
                                     SFL_OPT_SOR_OVERLAP (8), SFL_OPT_SOR_ARRIVAL (13) and SFL_OPT_SOR_CHAIN (15; the chained launch was
                                     retired in round 6), whose numbers are no longer accepted.  All schedules give the same bits.
                                     SET: 0 (default) = automatic: 3 on virtual and emulated ranks, 2 on RCCL ranks whose peers are
                                          other processes (a launch that waits inside the kernel is only as safe as its peer is
                                          punctual, and the scheme has not run on more than one GPU yet: tools/first_multi_gpu.sh);
                                          2 wherever the context's compute and exchange stream were found NOT to run side by side
                                          (measured once, at attach / first solve; RCCL ranks agree on it collectively);
                                       1 = in line: every launch whole, every exchange awaited on the compute stream;
                                       2 = one launch early, behind cross-stream events (round 3): a superstep's halo travels on the
                                          exchange stream while the owned rows of that launch are relaxed, its ghost rows are relaxed
                                          behind the message, the launch after waits whole (halo >= 2 x fuse; shallower halos and the
                                          right-hand side's exchange are awaited in line);
                                       3 = in time, counted on the device (sfl_plan_poisson kernel 3): the halo is exchanged after the
                                          launch that produces it; that launch runs the tiles whose rows the message carries at the
                                          top priority, writes through, and counts them; the message leaves on that count while the
                                          rest of the launch is still running; the next launch is queued at once and only its tiles
                                          next to a cut wait, INSIDE the launch, for a count of arrived messages.  A wait that outlasts
                                          SFL_OPT_HALO_TIMEOUT_MS gives up; sfl_synchronize reports it, sfl_download and the next
                                          operator on the context fail instead of handing out / building on an invalid field.
                                     GET: what the next solve will do: 0 = nothing to order (whole domain, no transport, the baseline
                                          kernel), else 1 / 2 / 3 with the automatic choice and the streams' verdict resolved        */
#define SFL_OPT_MEASURED_WIRE_US 20  /* READ ONLY: slabs with a transport: microseconds one halo exchange of this context costs before
                                     its first byte moves (launches, protocol, wire), measured -- not assumed -- with
                                     back-to-back exchanges of p at two depths when the transport was attached (RCCL ranks:
                                     the maximum over the ranks) or in front of the first solve (virtual / emulated ranks: the
                                     copy, and SFL_OPT_EMULATE_WIRE_US if set); -1 = nothing to measure, or not measured yet
                                     (the query itself never measures: that is a collective of the ranks and belongs to a solve).  The automatic halo depth of a
                                     solve (SFL_OPT_SOR_HALO = 0) is chosen from it: deeper halos = fewer exchanges, more rows
                                     relaxed redundantly                                                             */
#define SFL_OPT_LAST_HALO 21         /* READ ONLY: halo depth (rows of p per superstep) of the last solve's plan on this slab  */
#define SFL_OPT_SOR_FOLD 22          /* kernel 2's interior relaxation, poisson.cpp:107-111.  0 (default) = as the reference writes it,
                                     (1 - omega) * p + omega * (-0.25f * t), t = dx * d - sum: every product rounded on its own, the
                                     reference's bits on EVERY input.  1 = (1 - omega) * p + (-0.25f * omega) * t: one product less (7
                                     instead of 8 vector instructions per cell and pass, measured +2.4 .. 3 % cell-iters/s at 8192^2).
                                     The same bits wherever every operand of t (dx * d and the four neighbours' p) is zero or at least
                                     2^-124 = 4.7e-38 in magnitude -- dense fields.  Where one is a nonzero number below that -- the
                                     decaying front of a sparsely forced solution in a quiescent region reaches that range after ~63
                                     iterations: the sketch's own scenario -- the reference rounds -0.25f * t to a denormal first and
                                     the folded product can differ by one unit of 2^-149.  After ONE solve only cells of that front
                                     differ (absolute differences <= ~1e-40).  Over repeated sim steps the difference climbs the scales
                                     of a field that holds every magnitude down to the denormals, and pressure and velocity end up with
                                     ordinary rounding noise against the reference (units in the last place of each value; 1e-8 of the
                                     field's maximum after three steps at 80 iterations) -- inside north_star's 1e-5, but not the
                                     reference's bits.  Opt in only if that is acceptable; the boundary cells keep both products
                                     either way (tests/test_gpu_parity.py test_quiescent_*)                        */

typedef struct sfl_context sfl_context;

/* =====================================================================================
 * 3. utilities
 * ===================================================================================== */
SFL_API int sfl_abi_version(void);
/* Message of the last failing call on this thread ("" if none). */
SFL_API const char *sfl_last_error(void);
/* Number of visible HIP devices (0 and SFL_ERR_HIP when there is none). */
SFL_API int sfl_device_count(int *count);
/* Name / CU count / memory of a device; any out pointer may be NULL. */
SFL_API int sfl_device_info(int device, char *name, size_t name_cap, int *compute_units,
                            size_t *total_mem_bytes);

/* Row-slab partition of dim_y rows over nranks: rank g owns global rows
 * [dim_y*g/nranks, dim_y*(g+1)/nranks).  Pure arithmetic, no GPU needed.                */
SFL_API int sfl_slab_rows(int dim_y, int nranks, int rank, int *row_begin, int *row_end);

/* One step of a rank's program (sfl_plan_poisson): either a halo exchange with both
 * neighbouring slabs or a compute launch.  Row ranges are GLOBAL rows.                        */
typedef struct sfl_plan_step {
    int32_t kind;         /* SFL_STEP_*                                                       */
    int32_t field;        /* EXCHANGE: SFL_FIELD_* whose halo rows are refreshed              */
    int32_t rows;         /* EXCHANGE: rows per side (sent from / received next to the owned
                             block); 0 on compute steps                                       */
    int32_t g_begin;      /* compute: first output row (may extend into the ghost rows).
                             EXCHANGE: depth of the first row exchanged, counted from the cut
                             (0 = the rows next to it): each rank sends its owned rows at depth
                             [g_begin, g_begin + rows) and receives the neighbour's into the ghost
                             rows at the same depth; the ghost rows nearer the cut are still valid */
    int32_t g_end;        /* compute: one past the last output row                            */
    int32_t nsweeps;      /* SOR: colour passes executed by this launch.  Pass j (1-based)
                             covers rows [g_begin-(nsweeps-j), g_end+(nsweeps-j)) clipped to
                             the domain, i.e. halo rows are recomputed redundantly so that
                             the output rows are exact                                        */
    int32_t first_colour; /* SOR: colour of pass 1 (0 = even (i+j), poisson.cpp:22)           */
    int32_t from_zero;    /* SOR: p is implicitly zero on entry (poisson.cpp:117-119)         */
} sfl_plan_step;

#define SFL_STEP_EXCHANGE 1 /* refresh `rows` ghost rows per side of `field`                   */
#define SFL_STEP_SOR 2      /* nsweeps colour passes -> output rows [g_begin, g_end)           */
#define SFL_STEP_ZERO 3     /* zero-fill p on every local row (baseline kernel only)           */

/* Program of one poisson_solve on slab `rank` of `nranks`: kernel = 1 (one colour pass per
 * launch, 1-row exchange before every pass but the first) or 2 (fused: `fuse` passes per
 * launch; launches are grouped into supersteps of at most `halo` passes in total, with ONE
 * exchange of p per superstep but the first -- ghost rows are recomputed redundantly in
 * between -- plus one exchange of the right-hand side up front).
 * halo is clamped to >= fuse; halo == fuse exchanges before every launch.
 * halo >= 2 * fuse: EARLY exchanges.  The exchange of a superstep is issued one launch early,
 * before the LAST launch of the previous superstep, while the ghost rows are still valid as deep
 * as that launch needs for the owned rows (EXCHANGE.g_begin = its nsweeps, rows = halo - nsweeps);
 * that launch's output extends `what the next superstep needs` into the ghost rows, i.e. its passes
 * are repeated on the received rows.  An executor may run the owned rows of that launch while the
 * message travels and the ghost rows behind it (csrc/sor_executor.cpp run_poisson_early does).
 * Writes at most `cap` steps, returns the total in *n_steps.  Pure arithmetic, no GPU needed;
 * the GPU executor walks exactly this program.                                               */
SFL_API int sfl_plan_poisson(int dim_y, int nranks, int rank, int iters, int fuse, int kernel,
                             int halo, sfl_plan_step *steps, int cap, int *n_steps);
/* kernel = 3: kernel 2's launches with IN-TIME exchanges at every halo depth -- never early: the exchange of a superstep
 * follows the launch that produces its rows (SFL_OPT_EXCHANGE_SCHEDULE = 3; with a tail every superstep holds halo - tail passes and its
 * exchange skips the `tail` ghost rows the launch before left exact).                                                    */
/* The same with a TAIL: `tail` ghost rows of p are still exact when the solve ends (every launch of an
 * early-exchange plan then extends that much further into the ghost rows; ignored -- as 0 -- by the other
 * plans).  sfl_step on slabs asks for 1, the row subtract_gradient (finitediff.cpp:41-82) reads beyond a
 * cut, and so needs no exchange of p between the solve and the projection.                          */
SFL_API int sfl_plan_poisson_tail(int dim_y, int nranks, int rank, int iters, int fuse, int kernel,
                                  int halo, int tail, sfl_plan_step *steps, int cap, int *n_steps);

/* Pass plan of one poisson_solve: 2*iters half-sweeps are executed as `*n_passes` launches of
 * at most `fuse` half-sweeps each (the last one may be shorter); passes[k] receives the
 * number of half-sweeps of launch k when passes != NULL (capacity cap).  How these launches
 * interleave with halo exchanges on a slab is sfl_plan_poisson's business.  Pure arithmetic,
 * no GPU needed.                                                                            */
SFL_API int sfl_sor_pass_plan(int iters, int fuse, int *n_passes, int *passes, int cap);

/* =====================================================================================
 * 1. host-pointer drop-ins: the reference's operator signatures + int status.
 *    Pointers are HOST memory; each call uploads, runs the HIP kernels on `device 0`
 *    (or SFL_DEVICE from the environment) and downloads.  Intended for parity tests and as
 *    the first step of a port; production callers keep fields on the device (group 2).
 * ===================================================================================== */

/* advect<Vector2<float>, float>   advect.h:74-85 (sample :24-72).  next_p must not alias p;
 * p may alias vel (self-advection, ino:253).                                               */
SFL_API int sfl_host_advect_vec2f(float *next_p, const float *p, const float *vel, int dim_x,
                                  int dim_y, float dt, int no_slip);
/* advect<Vector3<UQ32>, float>    advect.h:74-85 with uq32.h:13,15 (ino:282)               */
SFL_API int sfl_host_advect_vec3uq32(uint32_t *next_p, const uint32_t *p, const float *vel,
                                     int dim_x, int dim_y, float dt, int no_slip);
/* advect<T, float>                advect.h:74-85 for EVERY element type the reference's headers can express:
 * T = `channels` (1..3) consecutive 32-bit channels of one `kind` -- float, Vector2<float>, Vector3<float>
 * (SFL_CHANNEL_F32) or UQ32, Vector2<UQ32>, Vector3<UQ32> (SFL_CHANNEL_UQ32; vector.h:4-126, uq32.h:8-16).
 * sample() (advect.h:24-72) acts channel by channel on all of them; the sketch's two instantiations
 * (2 x f32, 3 x uq32) take the same kernels as the two entry points above.                     */
#define SFL_CHANNEL_F32 0
#define SFL_CHANNEL_UQ32 1
SFL_API int sfl_host_advect_channels(void *next_p, const void *p, const float *vel, int dim_x, int dim_y,
                                     float dt, int no_slip, int channels, int kind);
/* calculate_divergence            finitediff.h:6-7, finitediff.cpp:9-39                     */
SFL_API int sfl_host_calculate_divergence(float *div, const float *v, int dim_x, int dim_y,
                                          float dx);
/* subtract_gradient (in place)    finitediff.h:9-10, finitediff.cpp:41-82                   */
SFL_API int sfl_host_subtract_gradient(float *v, const float *p, int dim_x, int dim_y, float dx);
/* poisson_solve                   poisson.h:4-5, poisson.cpp:114-125                        */
SFL_API int sfl_host_poisson_solve(float *p, const float *div, int dim_x, int dim_y, float dx,
                                   int iters, float omega);
/* The operators above never retain caller pointers (as the reference: finitediff.cpp:36,78-79,
 * poisson.cpp:120 keep their contexts on the stack).  For grids of up to 2^26 cells (8192^2) they do keep
 * their device-side working context with the calling thread between calls, so that a loop() of
 * five operators per frame (ino:252-287) does not set up streams and buffers five times per
 * frame; this releases it (optional; also replaced whenever the grid shape changes).          */
SFL_API int sfl_host_release(void);

/* =====================================================================================
 * 2. solver contexts: one context = one GPU's row slab, fields resident in HBM.
 * ===================================================================================== */

/* Whole domain on one device (rank 0 of 1). */
SFL_API int sfl_create(sfl_context **out, int device, int dim_x, int dim_y);
/* Limits: dim_x, dim_y >= 2; at most 2^30 cells per domain and 2^28 cells (owned + 128 ghost rows
 * on a slab) per context -- the kernels address a context's arrays with 32-bit byte offsets;
 * larger domains need more slabs.  Violations return SFL_ERR_INVALID before any GPU is touched.  */
/* Row slab `rank` of `nranks` (sfl_slab_rows) of a dim_x * dim_y domain on `device`.
 * Neighbouring slabs exchange halos through RCCL once sfl_comm_attach() has run, or through
 * in-process copies when the contexts were joined with sfl_group_link().                    */
SFL_API int sfl_create_slab(sfl_context **out, int device, int dim_x, int dim_y, int rank,
                            int nranks);
SFL_API int sfl_destroy(sfl_context *ctx);

/* Options of contexts joined by sfl_group_link are group-wide (setting one member sets all; linking
 * aligns the members with slab 0).  RCCL ranks are separate processes: set the same values on each. */
SFL_API int sfl_set_option(sfl_context *ctx, int option, int value);
SFL_API int sfl_get_option(sfl_context *ctx, int option, int *value);

/* Geometry of this context's slab: global rows [row_begin, row_end). */
SFL_API int sfl_slab_of(sfl_context *ctx, int *row_begin, int *row_end, int *rank, int *nranks);

/* --- RCCL bootstrap: rank 0 creates the id, the launcher distributes the bytes (e.g.
 *     torch.distributed broadcast), every rank attaches.  id_bytes = 128.                  */
SFL_API int sfl_comm_unique_id(void *id_out, size_t id_bytes);
SFL_API int sfl_comm_attach(sfl_context *ctx, const void *id, size_t id_bytes);
/* Collective over the attached communicator (called by sfl_comm_attach itself; call it again after changing
 * options): all-gathers domain, group size and every option the solve's program depends on, and fails with
 * SFL_ERR_STATE on the ranks that differ from any other -- mismatched programs would otherwise hang in a
 * send / receive or exchange the wrong rows.  Synchronises the context's streams.                       */
SFL_API int sfl_comm_check_options(sfl_context *ctx);
/* RCCL bring-up check for boxes with a single GPU (a communicator cannot hold two ranks of one
 * device): inside one ncclGroup, send the first `rows` owned rows of the divergence field to
 * this rank itself and receive them into the first `rows` owned rows of the pressure field --
 * the same pointer / count / stream arithmetic as a neighbour halo exchange.                  */
SFL_API int sfl_comm_loopback(sfl_context *ctx, int rows);
/* Measurement aid for boxes with fewer GPUs than ranks: this slab context runs ITS rank's program alone.
 * Every halo message it would send is copied -- same size, same stream, same ordering events -- into the
 * ghost rows it would receive into, so launches, copies and their overlap are exactly the rank's own while
 * the values next to the cuts are meaningless (never use the results).  SFL_OPT_TRANSPORT reads 3.
 * bench.py --emulate-rank R --of N reports the time of one solve on such a context.                 */
SFL_API int sfl_comm_emulate(sfl_context *ctx);
/* The same rank program with RCCL ITSELF as the transport: a one-rank communicator is created for the context and every halo
 * message becomes a real ncclSend / ncclRecv of this rank to itself, issued through the very code path a rank of a real
 * communicator takes (one ncclGroup per exchange on the exchange stream, the sender count in front of it and the arrival
 * count behind it when exchanges are counted on the device); the step's reductions over the ranks run as ncclAllReduce on
 * that communicator.  What one GPU can show of the multi-GPU path: RCCL's own kernels beside the solve's launches, their
 * launch latency, their stream semantics.  SFL_OPT_TRANSPORT reads 4; values next to the cuts are meaningless.
 * bench.py --emulate-rank R --of N --via-rccl.                                                                     */
SFL_API int sfl_comm_emulate_rccl(sfl_context *ctx);
/* In-process transport between virtual ranks living on ONE device (bring-up / tests):
 * ctxs[r] must be slab r of nranks == n, all created on the same device.                   */
SFL_API int sfl_group_link(sfl_context **ctxs, int n);

/* --- field I/O: the OWNED rows of this slab, host <-> device, synchronous.
 *     `host` holds (row_end-row_begin) * dim_x elements of the field's element type.       */
SFL_API int sfl_upload(sfl_context *ctx, int field, const void *host, size_t bytes);
SFL_API int sfl_download(sfl_context *ctx, int field, void *host, size_t bytes);
/* Device pointer of the first OWNED row of the field's CURRENT buffer (zero-copy interop; the
 * context keeps ownership).  The pointer is writable, so the query counts as a write from outside (as
 * sfl_upload does): on slabs the next operator exchanges / measures again instead of trusting ghost rows
 * and back-trace reaches it knew before.  Velocity, colour and pressure are ping-ponged between two buffers by
 * the operators that rewrite them (advect: ino:255,286; the fused SOR launches), so the pointer
 * is valid only until the next operator that writes that field -- sfl_step writes all of them:
 * query again after every such call (the query costs nothing).  Asynchronous work may still be
 * pending: sfl_synchronize() before reading through the pointer from another stream.           */
SFL_API int sfl_field_device_ptr(sfl_context *ctx, int field, void **dev_ptr);

/* --- operators on the resident fields, asynchronous on the context's stream.
 *     On a slab each call performs the halo exchanges it needs.  All ranks of a group must
 *     issue the same calls in the same order (they contain matched send/recv pairs).        */
/* velocity <- advect(velocity, velocity, dt, no_slip)       ino:252-256 */
SFL_API int sfl_advect_velocity(sfl_context *ctx, float dt, int no_slip);
/* colour   <- advect(colour, velocity, dt, no_slip)         ino:281-287 */
SFL_API int sfl_advect_color(sfl_context *ctx, float dt, int no_slip);
/* divergence <- calculate_divergence(velocity, dx)          ino:274     */
SFL_API int sfl_calculate_divergence(sfl_context *ctx, float dx);
/* pressure <- poisson_solve(divergence, dx, iters, omega)   ino:275     */
SFL_API int sfl_poisson_solve(sfl_context *ctx, float dx, int iters, float omega);
/* velocity <- subtract_gradient(velocity, pressure, dx)     ino:276     */
SFL_API int sfl_subtract_gradient(sfl_context *ctx, float dx);
/* next_p <- advect(p, velocity, dt, no_slip) for a field of the CALLER's, resident on the context's device
 * (advect.h:74-85; element = `channels` x `kind` as for sfl_host_advect_channels): further quantities carried by
 * the flow -- a temperature, a second dye -- without a round trip through the host.  Whole-domain contexts only;
 * next_p_dev must not alias p_dev; asynchronous on the context's stream (sfl_synchronize before reading).   */
SFL_API int sfl_advect_external(sfl_context *ctx, void *next_p_dev, const void *p_dev, int channels, int kind,
                                float dt, int no_slip);
/* One sim step in the order of ino:252-287: advect velocity (no-slip), [apply queued
 * forces], divergence, poisson_solve, subtract_gradient, advect colour (free-slip).         */
SFL_API int sfl_step(sfl_context *ctx, float dt, float dx, int iters, float omega);
/* n sim steps, exactly as n calls of sfl_step (the sim task's loop, ino:249-289, calls the step back to back); forces
 * queued before the call go into the FIRST step.  Knowing the next step lets the library fuse across the step boundary
 * (SFL_OPT_STEP_SEAMS).  n == 0 does nothing.                                                                       */
SFL_API int sfl_step_n(sfl_context *ctx, int n, float dt, float dx, int iters, float omega);
/* Queue point forces applied by the next sfl_step between the velocity advection and the
 * divergence (ino:264-269): velocity[index(cells[2k], cells[2k+1])] = (vel[2k], vel[2k+1])
 * in SIMULATION coordinates (the sketch's x/y swap is the caller's business), GLOBAL cell
 * indices.  On slabs EVERY rank queues the SAME list: a rank applies the cells that fall into
 * its rows and into the ghost row next to each cut, which sfl_step keeps exact instead of
 * exchanging it again (a list that differs between ranks gives a wrong divergence at the cuts).   */
SFL_API int sfl_queue_forces(sfl_context *ctx, const int *cells_ij, const float *vel_xy, int n);
/* The same in the sketch's own terms: `struct drag` as the touch task queues it (ino:45-48: Vector2<uint16_t>
 * coords, Vector2<float> velocity, GRAPHICS coordinates) with the transform loop() applies to it (ino:264-269):
 * cell = index(coords.y, coords.x), velocity = (velocity.y, velocity.x).  sizeof(sfl_drag) == sizeof(struct drag)
 * == 12, same member order: a drag_queue message can be passed as it is.  Coordinates outside the domain are
 * refused (SFL_ERR_INVALID, nothing queued); the sketch would write out of bounds.                        */
typedef struct sfl_drag {
    uint16_t coord_x, coord_y; /* msg.coords.x, msg.coords.y     */
    float vel_x, vel_y;        /* msg.velocity.x, msg.velocity.y */
} sfl_drag;
SFL_API int sfl_queue_drags(sfl_context *ctx, const sfl_drag *msgs, int n);

/* --- initial condition of the sketch (setup(), ino:196-241): velocity = 0; dye = three
 *     120-degree sectors around the centre chosen by atan2f, then the sketch's two in-place
 *     sequential 1-2-1 blur passes in UQ32.  The sketch pushes UINT32_MAX through float -> uint32
 *     conversions that are undefined in C++; they SATURATE here (as on the ESP32).  Whole-domain
 *     contexts only; asynchronous on the context's stream.                                   */
SFL_API int sfl_setup_sketch_fields(sfl_context *ctx);

/* --- dye visualiser (the arithmetic of the sketch's draw task, ino:116-176): every cell block
 *     is up-scaled `scaling` x `scaling` by the sketch's incremental lerps, narrowed to UQ32 and
 *     packed to RGB565 (byte-swapped like ino:173 when byteswap != 0).  `host_image` receives
 *     scaling*(dim_x-1) rows of scaling*(dim_y-1) uint16 pixels: the sim's i axis runs down the
 *     screen, j across (ino:164,180).  Whole-domain contexts only; synchronous.               */
SFL_API int sfl_render_rgb565(sfl_context *ctx, int scaling, int byteswap, uint16_t *host_image,
                              size_t bytes);

/* --- synchronisation / timing on the context's stream ---------------------------------- */
SFL_API int sfl_synchronize(sfl_context *ctx);
/* HIP-event stopwatch on the compute stream: start, run work, stop -> elapsed ms (blocks
 * until the stop event has completed).                                                     */
SFL_API int sfl_timer_start(sfl_context *ctx);
SFL_API int sfl_timer_stop(sfl_context *ctx, float *elapsed_ms);
/* Launch statistics of the last sfl_poisson_solve on this context: kernel launches, halo
 * exchanges, half-sweeps fused per launch.  Any out pointer may be NULL.                    */
SFL_API int sfl_last_solve_info(sfl_context *ctx, int *launches, int *exchanges, int *fuse);

#ifdef __cplusplus
}
#endif
#endif /* SFL_H */
