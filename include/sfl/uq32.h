// sfl/uq32.h -- 32-bit unsigned fixed-point colour channel, source compatible with the reference
// (ESP32-fluid-simulation/uq32.h:8-16): public `raw`, construction from float rounds half up by
// adding 0.5f and truncating (:13), conversion to float is the round-to-nearest-even widening of
// `raw` (:15).  Values >= 2^32 are undefined behaviour in the reference; the GPU path saturates
// (v_cvt_u32_f32), see SURVEY.md 5.1-6 -- keep channels below ~0xFF000000.
#ifndef SFL_UQ32_H
#define SFL_UQ32_H

#include <cstdint>

#include "vector.h"

struct UQ32 {
    uint32_t raw;

    SFL_XPU UQ32() {}
    SFL_XPU UQ32(float value) : raw(static_cast<uint32_t>(value + 0.5f)) {}
    SFL_XPU operator float() const { return static_cast<float>(raw); }

    // extension: wrap an already-narrowed channel without touching it
    SFL_XPU static UQ32 from_raw(uint32_t bits) { UQ32 c; c.raw = bits; return c; }
};

static_assert(sizeof(UQ32) == 4, "UQ32 is one 32-bit word");
static_assert(sizeof(Vector2<float>) == 8, "Vector2<float> is two packed floats");
static_assert(sizeof(Vector3<UQ32>) == 12, "Vector3<UQ32> is three packed words");

#endif  // SFL_UQ32_H
