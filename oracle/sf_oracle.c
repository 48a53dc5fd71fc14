/*
 * sf_oracle.c -- CPU ORACLE for the stable-fluids sim-task hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke
 * check in __graft_entry__.py and bench.py's cpu_baseline leg may load it.
 * The product path (libsfl_hip.so) never links, loads or calls anything here.
 *
 * It restates, in plain C99 on flat float / uint32 arrays, the arithmetic of
 * the reference sketch's sim task (citations are file:line under
 * /root/reference/ESP32-fluid-simulation/):
 *
 *   orc_advect_vec2f / orc_advect_vec3uq32   advect.h:74-85, sample advect.h:24-72,
 *                                            lerp/bilinear advect.h:13-22,
 *                                            element types vector.h:4-126, uq32.h:8-16
 *   orc_advect_channels                      advect<T,float> for T = float, UQ32, Vector2 / Vector3 of either
 *   orc_divergence                           finitediff.cpp:9-39 via operations.h:11-38
 *   orc_subtract_gradient                    finitediff.cpp:41-82
 *   orc_poisson_solve                        poisson.cpp:14-125
 *   orc_sor_half_sweep_rows                  one colour pass of poisson.cpp:14-61 restricted
 *                                            to a row range (used by the slab-sharding tests)
 *   orc_step                                 call order of ESP32-fluid-simulation.ino:252-287
 *                                            (no force injection, no RTOS hand-off)
 *   orc_setup_fields                         initial condition of setup(), ino:196-241 (PARITY
 *                                            UNPINNED; saturating float -> uint32 by definition)
 *   orc_render_rgb565                        draw-task arithmetic, ino:116-176 (PARITY UNPINNED:
 *                                            the .ino does not compile outside the Arduino core)
 *
 * Parity status: PINNED.  Every function is compared bit-for-bit against the
 * unmodified reference sources compiled in place (oracle/Makefile -> oracle/_ref,
 * tests/test_oracle_vs_reference.py) and against the known-answer hashes of
 * SURVEY.md 8(c) (tests/test_oracle_kat.py); committed fixtures in tests/golden/
 * were produced by the compiled reference (tests/golden/make_golden.py).
 *
 * Build: gcc -std=c99 -O2 -ffp-contract=off -fPIC -shared   (NO fused multiply-add:
 * every product and sum below is individually rounded, as in the reference
 * when it is built without contraction.)
 *
 * Layout: element (i, j) of a dim_x * dim_y field lives at dim_x*j + i
 * (operations.h:7-9); velocity is interleaved {x,y} floats (Vector2<float>, 8 B),
 * dye is interleaved {x,y,z} uint32 raw values (Vector3<UQ32>, 12 B).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* UQ32 (uq32.h:13,15): float -> raw is "+0.5f then truncate", raw -> float is RNE */
static inline uint32_t uq32_from_float(float x) { return (uint32_t)(x + 0.5f); }
static inline float uq32_to_float(uint32_t raw) { return (float)raw; }

/* lerp(t, a, b) = a*(1-t) + b*t   (advect.h:13-16): two products, one sum */
static inline float mix1(float t, float a, float b)
{
    float wa = 1.0f - t;
    float pa = a * wa;
    float pb = b * t;
    return pa + pb;
}

/* ------------------------------------------------------------------ */
/* Classification + fractional parts shared by both sample() flavours
 * (advect.h:26-35).                                                      */
typedef struct {
    int x_under, y_under, x_oob, y_oob;
    int ci, cj;       /* truncated floorf() of the source position       */
    float di, dj;     /* fractional parts                                */
} src_pos;

static inline src_pos classify(float si, float sj, int dim_x, int dim_y)
{
    src_pos s;
    int x_over = si >= (float)(dim_x - 1);
    int y_over = sj >= (float)(dim_y - 1);
    float fi = floorf(si), fj = floorf(sj);
    s.x_under = si < 0.0f;
    s.y_under = sj < 0.0f;
    s.x_oob = s.x_under || x_over;
    s.y_oob = s.y_under || y_over;
    s.di = si - fi;
    s.dj = sj - fj;
    /* only meaningful (and only used) on an in-range axis */
    s.ci = s.x_oob ? 0 : (int)fi;
    s.cj = s.y_oob ? 0 : (int)fj;
    return s;
}

/* no-slip discount (advect.h:62-70): product over out-of-range axes of
 * (o < 0.5 ? 1 - 2o : 0), o = overshoot distance                         */
static inline float wall_discount(const src_pos *s, float si, float sj, int dim_x, int dim_y)
{
    float factor = 1.0f;
    if (s->x_oob) {
        float over = s->x_under ? -si : si - (float)(dim_x - 1);
        factor *= (over < 0.5f) ? (1.0f - 2.0f * over) : 0.0f;
    }
    if (s->y_oob) {
        float over = s->y_under ? -sj : sj - (float)(dim_y - 1);
        factor *= (over < 0.5f) ? (1.0f - 2.0f * over) : 0.0f;
    }
    return factor;
}

/* sample<Vector2<float>> (advect.h:24-72): storage type == promoted type */
static void sample_vec2f(const float *p, float si, float sj, int dim_x, int dim_y,
                         int no_slip, float *out)
{
    src_pos s = classify(si, sj, dim_x, dim_y);
    int k;
    if (!s.x_oob && !s.y_oob) {
        const float *t = p + 2 * ((long)dim_x * s.cj + s.ci);
        const float *u = t + 2 * (long)dim_x;
        for (k = 0; k < 2; ++k) {
            float lo = mix1(s.dj, t[k], u[k]);          /* p11, p12 */
            float hi = mix1(s.dj, t[2 + k], u[2 + k]);  /* p21, p22 */
            out[k] = mix1(s.di, lo, hi);
        }
        return;
    }
    if (s.x_oob && s.y_oob) {
        const float *t = p + 2 * ((long)dim_x * (s.y_under ? 0 : dim_y - 1) +
                                  (s.x_under ? 0 : dim_x - 1));
        out[0] = t[0];
        out[1] = t[1];
    } else if (s.x_oob) {
        const float *t = p + 2 * ((long)dim_x * s.cj + (s.x_under ? 0 : dim_x - 1));
        const float *u = t + 2 * (long)dim_x;
        for (k = 0; k < 2; ++k) out[k] = mix1(s.dj, t[k], u[k]);
    } else {
        const float *t = p + 2 * ((long)dim_x * (s.y_under ? 0 : dim_y - 1) + s.ci);
        for (k = 0; k < 2; ++k) out[k] = mix1(s.di, t[k], t[2 + k]);
    }
    if (no_slip) {
        float f = wall_discount(&s, si, sj, dim_x, dim_y);
        out[0] = out[0] * f;
        out[1] = out[1] * f;
    }
}

/* sample<Vector3<UQ32>> (advect.h:24-72 with uq32.h:13,15).  The result is
 * returned as RAW storage values: the interior value is narrowed once
 * (return conversion, advect.h:40); an edge value is narrowed when assigned
 * to "T p_edge" (advect.h:45-54) and returned untouched when !no_slip
 * (:57-59) -- it must NOT be narrowed a second time, float(raw)+0.5f is not
 * the identity for raw >= 2^23 -- or widened, scaled and narrowed again when
 * no_slip (:71).                                            */
static void sample_vec3uq(const uint32_t *p, float si, float sj, int dim_x, int dim_y,
                          int no_slip, uint32_t *out)
{
    src_pos s = classify(si, sj, dim_x, dim_y);
    int k;
    if (!s.x_oob && !s.y_oob) {
        const uint32_t *t = p + 3 * ((long)dim_x * s.cj + s.ci);
        const uint32_t *u = t + 3 * (long)dim_x;
        for (k = 0; k < 3; ++k) {
            float lo = mix1(s.dj, uq32_to_float(t[k]), uq32_to_float(u[k]));
            float hi = mix1(s.dj, uq32_to_float(t[3 + k]), uq32_to_float(u[3 + k]));
            out[k] = uq32_from_float(mix1(s.di, lo, hi));
        }
        return;
    }
    if (s.x_oob && s.y_oob) {
        const uint32_t *t = p + 3 * ((long)dim_x * (s.y_under ? 0 : dim_y - 1) +
                                     (s.x_under ? 0 : dim_x - 1));
        for (k = 0; k < 3; ++k) out[k] = t[k];
    } else if (s.x_oob) {
        const uint32_t *t = p + 3 * ((long)dim_x * s.cj + (s.x_under ? 0 : dim_x - 1));
        const uint32_t *u = t + 3 * (long)dim_x;
        for (k = 0; k < 3; ++k)
            out[k] = uq32_from_float(mix1(s.dj, uq32_to_float(t[k]), uq32_to_float(u[k])));
    } else {
        const uint32_t *t = p + 3 * ((long)dim_x * (s.y_under ? 0 : dim_y - 1) + s.ci);
        for (k = 0; k < 3; ++k)
            out[k] = uq32_from_float(mix1(s.di, uq32_to_float(t[k]), uq32_to_float(t[3 + k])));
    }
    if (no_slip) {
        float f = wall_discount(&s, si, sj, dim_x, dim_y);
        for (k = 0; k < 3; ++k) out[k] = uq32_from_float(uq32_to_float(out[k]) * f);
    }
}

/* advect<Vector2<float>, float>  (advect.h:74-85) */
ORC_API void orc_advect_vec2f(float *next_p, const float *p, const float *vel,
                              int dim_x, int dim_y, float dt, int no_slip)
{
    int i, j;
    for (j = 0; j < dim_y; ++j) {
        for (i = 0; i < dim_x; ++i) {
            long cell = (long)dim_x * j + i;
            float sx = (float)i - vel[2 * cell] * dt;
            float sy = (float)j - vel[2 * cell + 1] * dt;
            sample_vec2f(p, sx, sy, dim_x, dim_y, no_slip, next_p + 2 * cell);
        }
    }
}

/* advect<Vector3<UQ32>, float>  (advect.h:74-85, uq32.h) */
ORC_API void orc_advect_vec3uq32(uint32_t *next_p, const uint32_t *p, const float *vel,
                                 int dim_x, int dim_y, float dt, int no_slip)
{
    int i, j;
    for (j = 0; j < dim_y; ++j) {
        for (i = 0; i < dim_x; ++i) {
            long cell = (long)dim_x * j + i;
            float sx = (float)i - vel[2 * cell] * dt;
            float sy = (float)j - vel[2 * cell + 1] * dt;
            sample_vec3uq(p, sx, sy, dim_x, dim_y, no_slip, next_p + 3 * cell);
        }
    }
}

/* advect<T, float> (advect.h:74-85) for every element type the reference's headers can express:
 * `channels` (1..3) consecutive 32-bit channels, all float (kind 0: float, Vector2<float>,
 * Vector3<float>) or all UQ32 raw (kind 1: UQ32, Vector2<UQ32>, Vector3<UQ32>).  sample()
 * (advect.h:24-72) is written against T's operators, which act channel by channel
 * (vector.h:23-61, :83-126); TPromoted<T> has float channels (advect.h:10-11) and converting
 * back narrows each channel on its own (uq32.h:13).  Same narrowing points as sample_vec3uq.   */
static float chan_widen(uint32_t bits, int uq)
{
    float f;
    if (uq) return uq32_to_float(bits);
    memcpy(&f, &bits, 4);
    return f;
}
static uint32_t chan_narrow(float x, int uq)
{
    uint32_t bits;
    if (uq) return uq32_from_float(x);
    memcpy(&bits, &x, 4);
    return bits;
}
static void sample_channels(const uint32_t *p, float si, float sj, int dim_x, int dim_y, int no_slip,
                            int nc, int uq, uint32_t *out)
{
    src_pos s = classify(si, sj, dim_x, dim_y);
    int k;
    if (!s.x_oob && !s.y_oob) {                      /* advect.h:37-42 */
        const uint32_t *t = p + nc * ((long)dim_x * s.cj + s.ci);
        const uint32_t *u = t + nc * (long)dim_x;
        for (k = 0; k < nc; ++k) {
            float lo = mix1(s.dj, chan_widen(t[k], uq), chan_widen(u[k], uq));
            float hi = mix1(s.dj, chan_widen(t[nc + k], uq), chan_widen(u[nc + k], uq));
            out[k] = chan_narrow(mix1(s.di, lo, hi), uq);
        }
        return;
    }
    if (s.x_oob && s.y_oob) {                        /* advect.h:46-48: the corner texel as stored */
        const uint32_t *t = p + nc * ((long)dim_x * (s.y_under ? 0 : dim_y - 1) + (s.x_under ? 0 : dim_x - 1));
        for (k = 0; k < nc; ++k) out[k] = t[k];
    } else if (s.x_oob) {                            /* advect.h:49-51 */
        const uint32_t *t = p + nc * ((long)dim_x * s.cj + (s.x_under ? 0 : dim_x - 1));
        const uint32_t *u = t + nc * (long)dim_x;
        for (k = 0; k < nc; ++k) out[k] = chan_narrow(mix1(s.dj, chan_widen(t[k], uq), chan_widen(u[k], uq)), uq);
    } else {                                         /* advect.h:52-55 */
        const uint32_t *t = p + nc * ((long)dim_x * (s.y_under ? 0 : dim_y - 1) + s.ci);
        for (k = 0; k < nc; ++k) out[k] = chan_narrow(mix1(s.di, chan_widen(t[k], uq), chan_widen(t[nc + k], uq)), uq);
    }
    if (no_slip) {                                   /* advect.h:61-71 */
        float f = wall_discount(&s, si, sj, dim_x, dim_y);
        for (k = 0; k < nc; ++k) out[k] = chan_narrow(f * chan_widen(out[k], uq), uq);
    }
}

ORC_API void orc_advect_channels(void *next_p, const void *p, const float *vel, int dim_x, int dim_y,
                                 float dt, int no_slip, int channels, int kind)
{
    int i, j;
    for (j = 0; j < dim_y; ++j) {
        for (i = 0; i < dim_x; ++i) {
            long cell = (long)dim_x * j + i;
            float sx = (float)i - vel[2 * cell] * dt;
            float sy = (float)j - vel[2 * cell + 1] * dt;
            sample_channels((const uint32_t *)p, sx, sy, dim_x, dim_y, no_slip, channels, kind,
                            (uint32_t *)next_p + channels * cell);
        }
    }
}

/* ------------------------------------------------------------------ */
/* calculate_divergence (finitediff.cpp:9-39).  Interior cells use the
 * pairwise association of :29, perimeter cells the running sum of :16-20
 * with ghost velocity = -own.                                            */
ORC_API void orc_divergence(float *div, const float *v, int dim_x, int dim_y, float dx)
{
    float two_dx_inv = 1.0f / (2.0f * dx);
    int i_max = dim_x - 1, j_max = dim_y - 1;
    int i, j;
    for (j = 0; j <= j_max; ++j) {
        for (i = 0; i <= i_max; ++i) {
            long c = (long)dim_x * j + i;
            const float *vc = v + 2 * c;
            float s;
            if (i > 0 && i < i_max && j > 0 && j < j_max) {
                float hx = -vc[-2] + vc[2];
                float hy = -vc[-2 * (long)dim_x + 1] + vc[2 * (long)dim_x + 1];
                s = hx + hy;
            } else {
                s = 0.0f;
                s += (i > 0) ? -vc[-2] : vc[0];
                s += (i < i_max) ? vc[2] : -vc[0];
                s += (j > 0) ? -vc[-2 * (long)dim_x + 1] : vc[1];
                s += (j < j_max) ? vc[2 * (long)dim_x + 1] : -vc[1];
            }
            div[c] = s * two_dx_inv;
        }
    }
}

/* subtract_gradient (finitediff.cpp:41-82): in place on v; a missing
 * neighbour's pressure is the cell's own.                                */
ORC_API void orc_subtract_gradient(float *v, const float *p, int dim_x, int dim_y, float dx)
{
    float two_dx_inv = 1.0f / (2.0f * dx);
    int i_max = dim_x - 1, j_max = dim_y - 1;
    int i, j;
    for (j = 0; j <= j_max; ++j) {
        for (i = 0; i <= i_max; ++i) {
            long c = (long)dim_x * j + i;
            float pw = (i > 0) ? p[c - 1] : p[c];
            float pe = (i < i_max) ? p[c + 1] : p[c];
            float ps = (j > 0) ? p[c - dim_x] : p[c];
            float pn = (j < j_max) ? p[c + dim_x] : p[c];
            float gx = (pe - pw) * two_dx_inv;
            float gy = (pn - ps) * two_dx_inv;
            v[2 * c] = v[2 * c] - gx;
            v[2 * c + 1] = v[2 * c + 1] - gy;
        }
    }
}

/* ------------------------------------------------------------------ */
/* One SOR update of cell (i, j) (poisson.cpp:63-112).  gj / gdim_y are the
 * row index and height in the GLOBAL domain (they equal j / dim_y when the
 * array is the whole domain); `p` and `d` point at the cell.               */
static inline float sor_cell(const float *p, const float *d, int i, int gj, int dim_x,
                             int gdim_y, float dx, float omega)
{
    float p_gs;
    if (i > 0 && i < dim_x - 1 && gj > 0 && gj < gdim_y - 1) {
        float sum = p[-1] + p[1] + p[-dim_x] + p[dim_x];
        p_gs = -0.25f * (dx * d[0] - sum);
    } else {
        /* {0, 0, -1/2, -1/3, -1/4} evaluated in double, narrowed (poisson.cpp:67) */
        static const float neg_inv[5] = {0.0f, 0.0f, (float)(-1.0 / 2.0), (float)(-1.0 / 3.0),
                                         (float)(-1.0 / 4.0)};
        float sum = 0.0f;
        int n = 0;
        if (i > 0) { sum += p[-1]; ++n; }
        if (i < dim_x - 1) { sum += p[1]; ++n; }
        if (gj > 0) { sum += p[-dim_x]; ++n; }
        if (gj < gdim_y - 1) { sum += p[dim_x]; ++n; }
        p_gs = neg_inv[n] * (dx * d[0] - sum);
    }
    return (1.0f - omega) * p[0] + omega * p_gs;
}

/* One colour pass over local rows [row_begin, row_end) of an array that
 * holds global rows [grow0, grow0 + local rows).  colour 0 = even (i+gj)
 * (the FIRST pass, poisson.cpp:22 on_red=false), colour 1 = odd.          */
ORC_API void orc_sor_half_sweep_rows(float *p, const float *d, int dim_x, int gdim_y,
                                     float dx, float omega, int colour, int row_begin,
                                     int row_end, int grow0)
{
    int i, j;
    for (j = row_begin; j < row_end; ++j) {
        int gj = grow0 + j;
        for (i = (gj + colour) & 1; i < dim_x; i += 2) {
            long c = (long)dim_x * j + i;
            p[c] = sor_cell(p + c, d + c, i, gj, dim_x, gdim_y, dx, omega);
        }
    }
}

/* poisson_solve (poisson.cpp:114-125) */
ORC_API void orc_poisson_solve(float *p, const float *div, int dim_x, int dim_y, float dx,
                               int iters, float omega)
{
    long n = (long)dim_x * dim_y;
    int k;
    for (long c = 0; c < n; ++c) p[c] = 0.0f;
    for (k = 0; k < iters; ++k) {
        orc_sor_half_sweep_rows(p, div, dim_x, dim_y, dx, omega, 0, 0, dim_y, 0);
        orc_sor_half_sweep_rows(p, div, dim_x, dim_y, dx, omega, 1, 0, dim_y, 0);
    }
}

/* SOR continuing from the given p (no zero fill): used by property tests */
ORC_API void orc_sor_iterate(float *p, const float *div, int dim_x, int dim_y, float dx,
                             int iters, float omega)
{
    int k;
    for (k = 0; k < iters; ++k) {
        orc_sor_half_sweep_rows(p, div, dim_x, dim_y, dx, omega, 0, 0, dim_y, 0);
        orc_sor_half_sweep_rows(p, div, dim_x, dim_y, dx, omega, 1, 0, dim_y, 0);
    }
}

/* ------------------------------------------------------------------ */
/* One sim step in the order of ESP32-fluid-simulation.ino:252-287.
 * v and colour are updated in place (the sketch swaps pointers); div and p
 * are caller-provided scratch of dim_x*dim_y floats and hold the step's
 * divergence / pressure on return.  Returns 0, or -1 on allocation failure. */
ORC_API int orc_step(float *v, uint32_t *colour, float *div, float *p, int dim_x, int dim_y,
                     float dt, float dx, int iters, float omega)
{
    size_t n = (size_t)dim_x * dim_y;
    float *vt = (float *)malloc(n * 2 * sizeof(float));
    uint32_t *ct = (uint32_t *)malloc(n * 3 * sizeof(uint32_t));
    if (!vt || !ct) { free(vt); free(ct); return -1; }
    orc_advect_vec2f(vt, v, v, dim_x, dim_y, dt, 1);
    memcpy(v, vt, n * 2 * sizeof(float));
    orc_divergence(div, v, dim_x, dim_y, dx);
    orc_poisson_solve(p, div, dim_x, dim_y, dx, iters, omega);
    orc_subtract_gradient(v, p, dim_x, dim_y, dx);
    orc_advect_vec3uq32(ct, colour, v, dim_x, dim_y, dt, 0);
    memcpy(colour, ct, n * 3 * sizeof(uint32_t));
    free(vt);
    free(ct);
    return 0;
}

/* ------------------------------------------------------------------ */
/* Dye visualiser: restates the arithmetic of the sketch's draw task,
 * ESP32-fluid-simulation.ino:116-176 -- SCALING x SCALING bilinear up-scale of every cell block
 * by strength-reduced lerps (c += dc, :134-161), float -> UQ32 narrowing (:168), RGB565 pack
 * (:170-172) and optional byte swap (:173).  PARITY UNPINNED: the .ino cannot be compiled here
 * (Arduino core / TFT_eSPI / FreeRTOS), so this restatement is checked by reading only.
 *
 * Screen geometry (ino:116-117,164,180): sim index i (fast axis, dim_x) runs down the screen,
 * sim index j (dim_y) runs across; image = scaling*(dim_x-1) rows of scaling*(dim_y-1) pixels.
 * The sketch re-uses a block's right edge as the next block's left edge (:139-143); recomputing
 * the left edge from its two corner texels performs the identical operations.               */
ORC_API void orc_render_rgb565(uint16_t *image, const uint32_t *colour, int dim_x, int dim_y,
                               int scaling, int byteswap)
{
    const float inv = 1.0f / (float)scaling;
    const int width = scaling * (dim_y - 1);
    int i, j, ii, jj, k;
    for (i = 0; i < dim_x - 1; ++i) {
        for (j = 0; j < dim_y - 1; ++j) {
            const uint32_t *t1 = colour + 3 * ((long)dim_x * j + i);           /* (i,   j)   */
            const uint32_t *t2 = colour + 3 * ((long)dim_x * (j + 1) + i);     /* (i,   j+1) */
            const uint32_t *t3 = t1 + 3;                                       /* (i+1, j)   */
            const uint32_t *t4 = t2 + 3;                                       /* (i+1, j+1) */
            for (ii = 0; ii < scaling; ++ii) {
                for (jj = 0; jj < scaling; ++jj) {
                    uint32_t raw[3];
                    uint16_t px;
                    for (k = 0; k < 3; ++k) {
                        float l = uq32_to_float(t1[k]), r = uq32_to_float(t2[k]), c;
                        const float dl = (uq32_to_float(t3[k]) - l) * inv;
                        const float dr = (uq32_to_float(t4[k]) - r) * inv;
                        float dc;
                        int n;
                        for (n = 0; n < ii; ++n) { l += dl; r += dr; }
                        c = l;
                        dc = (r - l) * inv;
                        for (n = 0; n < jj; ++n) c += dc;
                        raw[k] = uq32_from_float(c);
                    }
                    px = (uint16_t)(((raw[0] & 0xF8000000u) >> 16) | ((raw[1] & 0xFC000000u) >> 21) |
                                    ((raw[2] & 0xF8000000u) >> 27));
                    if (byteswap) px = (uint16_t)((px >> 8) | (px << 8));
                    image[(long)(i * scaling + ii) * width + (j * scaling + jj)] = px;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* Initial condition of the sketch: restates setup(), ino:196-241 -- zero
 * velocity (:197-201); dye = three 120-degree sectors chosen by atan2f (:204-218); two in-place,
 * sequential 1-2-1 blur passes in UQ32, first along j (:219-229: the left neighbour is already
 * blurred, the right one is not), then along i (:230-241).  PARITY UNPINNED (the .ino does not
 * compile here).  One deliberate definition: the sketch feeds UINT32_MAX through float -> uint32
 * conversions that are undefined behaviour in C++ (4294967295 rounds to 2^32 as a float,
 * SURVEY 5.1-6,11); like the ESP32's and the GPU's conversion instructions this restatement
 * SATURATES (x86 would wrap to 0).                                                          */
static inline uint32_t uq32_from_float_sat(float x)
{
    const float y = x + 0.5f;
    return y >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)y;
}

ORC_API void orc_setup_fields(float *v, uint32_t *colour, int dim_x, int dim_y)
{
    const double third_pi = 3.1415926535897932384626433832795 / 3;  /* Arduino's PI / 3 */
    const int ci = dim_x / 2, cj = dim_y / 2;
    int i, j, k;
    for (long n = 0; n < 2L * dim_x * dim_y; ++n) v[n] = 0.0f;
    for (i = 0; i < dim_x; ++i) {
        for (j = 0; j < dim_y; ++j) {
            const float angle = atan2f((float)(-(i - ci)), (float)(j - cj));
            uint32_t *c = colour + 3 * ((long)dim_x * j + i);
            const int sector = ((double)angle < -third_pi) ? 0 : ((double)angle < third_pi) ? 1 : 2;
            for (k = 0; k < 3; ++k) c[k] = uq32_from_float_sat(k == sector ? (float)4294967295u : 0.0f);
        }
    }
    for (i = 0; i < dim_x; ++i) {           /* blur along j, in place, increasing j */
        for (j = 0; j < dim_y; ++j) {
            uint32_t *c = colour + 3 * ((long)dim_x * j + i);
            const uint32_t *l = (j == 0) ? c : c - 3 * (long)dim_x;
            const uint32_t *r = (j == dim_y - 1) ? c : c + 3 * (long)dim_x;
            for (k = 0; k < 3; ++k) {
                const float s = (0.25f * uq32_to_float(l[k]) + 0.5f * uq32_to_float(c[k])) +
                                0.25f * uq32_to_float(r[k]);
                c[k] = uq32_from_float_sat(s);
            }
        }
    }
    for (i = 0; i < dim_x; ++i) {           /* blur along i, in place, increasing i */
        for (j = 0; j < dim_y; ++j) {
            uint32_t *c = colour + 3 * ((long)dim_x * j + i);
            const uint32_t *t = (i == 0) ? c : c - 3;
            const uint32_t *b = (i == dim_x - 1) ? c : c + 3;
            for (k = 0; k < 3; ++k) {
                const float s = (0.25f * uq32_to_float(t[k]) + 0.5f * uq32_to_float(c[k])) +
                                0.25f * uq32_to_float(b[k]);
                c[k] = uq32_from_float_sat(s);
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* Seeded inputs of SURVEY.md 8(c): LCG S <- S*1664525 + 1013904223 (mod 2^32) */
ORC_API void orc_lcg_fields(float *v, uint32_t *colour, int dim_x, int dim_y, uint32_t seed,
                            float vamp)
{
    uint32_t s = seed;
    long n = (long)dim_x * dim_y, k;
    for (k = 0; k < n; ++k) {
        s = s * 1664525u + 1013904223u;
        v[2 * k] = (float)((int)((s >> 8) % 2001u) - 1000) / 1000.0f * vamp;
        s = s * 1664525u + 1013904223u;
        v[2 * k + 1] = (float)((int)((s >> 8) % 2001u) - 1000) / 1000.0f * vamp;
        s = s * 1664525u + 1013904223u;
        colour[3 * k] = s >> 1;
        s = s * 1664525u + 1013904223u;
        colour[3 * k + 1] = s >> 1;
        s = s * 1664525u + 1013904223u;
        colour[3 * k + 2] = s >> 1;
    }
}

/* FNV-1a-64 over raw bytes */
ORC_API uint64_t orc_fnv1a64(const void *data, size_t nbytes)
{
    const unsigned char *b = (const unsigned char *)data;
    uint64_t h = 1469598103934665603ull;
    size_t k;
    for (k = 0; k < nbytes; ++k) {
        h ^= b[k];
        h *= 1099511628211ull;
    }
    return h;
}
