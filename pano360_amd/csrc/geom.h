// Device-side geometry shared by the warp and ownership kernels: the inverse
// map of stitcher.py:300-312, cv2.remap's fixed-point tap selection, and the
// analytic alpha of _add_weights (stitcher.py:251-263).
#pragma once
#include "common.h"

struct Taps {
    int x0, x1, y0, y1;
    float w00, w01, w10, w11;
};

// cv2.remap's coordinate handling (INTER_BITS = 5), see include/pano360.h.
// tap_base: the sample position in fixed point before any border handling - x0 / y0 the
// integer parts (saturated like remap's short coordinates), x1 / y1 their successors - and
// the four table weights.
__device__ __forceinline__ Taps tap_base(float px, float py) {
    int sx = cv_round(px * 32.0f), sy = cv_round(py * 32.0f);
    int fx = sx & 31, fy = sy & 31;
    Taps t;
    t.x0 = sat16(sx >> 5);
    t.y0 = sat16(sy >> 5);
    t.x1 = t.x0 + 1;
    t.y1 = t.y0 + 1;
    float ax = (float)fx * (1.0f / 32.0f), ay = (float)fy * (1.0f / 32.0f);
    t.w00 = (1.0f - ay) * (1.0f - ax);
    t.w01 = (1.0f - ay) * ax;
    t.w10 = ay * (1.0f - ax);
    t.w11 = ay * ax;
    return t;
}

// BORDER_REFLECT on the four coordinates of tap_base.
__device__ __forceinline__ void reflect_taps(Taps &t, int sw, int sh) {
    t.x1 = reflect_edge(t.x0 + 1, sw);
    t.x0 = reflect_edge(t.x0, sw);
    t.y1 = reflect_edge(t.y0 + 1, sh);
    t.y0 = reflect_edge(t.y0, sh);
}

__device__ __forceinline__ Taps make_taps(float px, float py, int sw, int sh) {
    Taps t = tap_base(px, py);
    reflect_taps(t, sw, sh);
    return t;
}

// All four taps of tap_base lie in the frame: no border handling, the two taps of a row are
// neighbours, the rows follow each other.
__device__ __forceinline__ bool taps_interior(const Taps &t, int sw, int sh) {
    return (unsigned)t.x0 < (unsigned)(sw - 1) && (unsigned)t.y0 < (unsigned)(sh - 1);
}

// Taps of a pixel map_pixel did NOT mask: 0 <= px <= sw - 1 and 0 <= py <= sh - 1, so
// x0 = (round(32 px)) >> 5 lies in [0, sw - 1] and only its successor can leave the
// frame, by one; BORDER_REFLECT maps sw to sw - 1.  Same taps as make_taps without its
// four range tests (each hides a modulo the compiler has to branch around).
__device__ __forceinline__ Taps make_taps_unmasked(float px, float py, int sw, int sh) {
    Taps t = tap_base(px, py);
    t.x1 = min(t.x1, sw - 1);
    t.y1 = min(t.y1, sh - 1);
    return t;
}

// v00*w00 + v01*w01 + v10*w10 + v11*w11, left to right, one rounding per
// operation (the library is built with -ffp-contract=off).
__device__ __forceinline__ float lerp4(float v00, float v01, float v10, float v11,
                                       const Taps &t) {
    float a = v00 * t.w00;
    a = a + v01 * t.w01;
    a = a + v10 * t.w10;
    a = a + v11 * t.w11;
    return a;
}

// The four taps' uint8 RGB triples, as OFFSETS INTO A FLOAT TABLE (4 x the byte's value: what
// every user does with a tap's byte is look its colour up in a 256-entry float table).  Away
// from the frame border the two taps of a row are neighbours (x1 == x0 + 1): their six bytes
// are fetched with one 4-byte and one 2-byte load (any alignment) instead of six byte loads -
// the sampling kernels are bound by the number of their instructions, not by bytes - and a
// byte leaves its dword already multiplied by four, in ONE instruction: `v_lshlrev_b32` with
// an SDWA byte select on its operand (mask, shift and scale were three: 26 of the warp's 139
// vector instructions per pixel, profiles/r05/warp_isa_breakdown.txt).
struct TapBytes {
    uint32_t v[4][3];               // [tap 00, 01, 10, 11][channel]: 4 * byte
};
typedef uint32_t u32_any __attribute__((aligned(1)));
typedef uint16_t u16_any __attribute__((aligned(1)));
// a frame pointer taken out of a camera record is generic to the compiler (flat_load);
// frames live in global memory
typedef const __attribute__((address_space(1))) uint8_t *frame_ptr;
typedef const __attribute__((address_space(1))) u32_any *frame_ptr32;
typedef const __attribute__((address_space(1))) u16_any *frame_ptr16;

#define PANO_BYTE_X4(B)                                                                        \
    __device__ __forceinline__ uint32_t byte##B##_x4(uint32_t x) {                             \
        uint32_t r;                                                                            \
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD "               \
            "src0_sel:DWORD src1_sel:BYTE_" #B                                                 \
            : "=v"(r)                                                                          \
            : "v"(2u), "v"(x));                                                                \
        return r;                                                                              \
    }
PANO_BYTE_X4(0)
PANO_BYTE_X4(1)
PANO_BYTE_X4(2)
PANO_BYTE_X4(3)
#undef PANO_BYTE_X4

// entry of a 256-float table at a TapBytes offset
__device__ __forceinline__ float lut_at(const float *__restrict__ table, uint32_t off) {
    return *(const float *)((const char *)table + off);
}

// a row's two neighbouring taps out of its six bytes: lo = bytes 0..3, hi = bytes 4..5
__device__ __forceinline__ void unpack_row(TapBytes &t, int r, uint32_t lo, uint32_t hi) {
    t.v[2 * r][0] = byte0_x4(lo);
    t.v[2 * r][1] = byte1_x4(lo);
    t.v[2 * r][2] = byte2_x4(lo);
    t.v[2 * r + 1][0] = byte3_x4(lo);
    t.v[2 * r + 1][1] = byte0_x4(hi);
    t.v[2 * r + 1][2] = byte1_x4(hi);
}

__device__ __forceinline__ TapBytes load_taps(const uint8_t *__restrict__ frame, int sw,
                                              const Taps &tp) {
    TapBytes t;
    const frame_ptr base = (frame_ptr)frame;
    // (frames are smaller than 2^32 bytes: both sides are below 32768 pixels)
    const uint32_t pitch = (uint32_t)sw * 3u;
    const frame_ptr rows[2] = {base + (uint32_t)tp.y0 * pitch, base + (uint32_t)tp.y1 * pitch};
    const bool pair = tp.x1 == tp.x0 + 1;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const frame_ptr a = rows[r] + (uint32_t)tp.x0 * 3u;
        if (pair) {
            unpack_row(t, r, *(frame_ptr32)a, *(frame_ptr16)(a + 4));
        } else {
            const frame_ptr b = rows[r] + (uint32_t)tp.x1 * 3u;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                t.v[2 * r][k] = (uint32_t)a[k] << 2;
                t.v[2 * r + 1][k] = (uint32_t)b[k] << 2;
            }
        }
    }
    return t;
}

// The taps of a pixel whose four taps lie in the frame (taps_interior): one 4-byte and one
// 2-byte load per row at a 32-bit offset from the frame's (wave-uniform) base.
__device__ __forceinline__ TapBytes load_taps_interior(const uint8_t *__restrict__ frame, int sw,
                                                       const Taps &tp) {
    TapBytes t;
    const frame_ptr base = (frame_ptr)frame;
    const uint32_t pitch = (uint32_t)sw * 3u;
    const uint32_t o0 = (uint32_t)tp.y0 * pitch + (uint32_t)tp.x0 * 3u;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const frame_ptr a = base + (r ? o0 + pitch : o0);
        unpack_row(t, r, *(frame_ptr32)a, *(frame_ptr16)(a + 4));
    }
    return t;
}

// table[i] of a double table in global memory whose base is wave-uniform (the trig tables of a
// mosaic: i < 2^28): the index goes in as a 32-bit byte offset beside the scalar base instead of
// a sign-extended 64-bit address per load (three vector instructions each).
__device__ __forceinline__ double table_f64(const double *__restrict__ table, int i) {
    typedef const __attribute__((address_space(1))) char *gbyte;
    return *(const __attribute__((address_space(1))) double *)((gbyte)table + (uint32_t)i * 8u);
}

// ray = (sin theta, tan phi, cos theta); pixel = K R ray in double as an FMA
// chain over k, rounded to float32, divided and centred in float32 (:303-310);
// mask = behind the camera or outside [0, w-1] x [0, h-1] (:308, :311-312).
__device__ __forceinline__ bool map_pixel(const double *K, double s, double c, double t,
                                          int sw, int sh, float &px, float &py) {
    const double vx = fma(K[2], c, fma(K[1], t, K[0] * s));
    const double vy = fma(K[5], c, fma(K[4], t, K[3] * s));
    const double vz = fma(K[8], c, fma(K[7], t, K[6] * s));
    const float fx = (float)vx, fy = (float)vy, fz = (float)vz;
    const float cx = (float)((double)sw / 2.0), cy = (float)((double)sh / 2.0);
    px = __fdiv_rn(fx, fz) + cx;
    py = __fdiv_rn(fy, fz) + cy;
    bool m = fz < 0.0f;
    m |= (px < 0.0f) | (px > (float)(sw - 1)) | (py < 0.0f) | (py > (float)(sh - 1));
    return m;
}

// Bilinear sample of the alpha plane _add_weights would have stored:
// float32(hat_y[y] * hat_x[x]) with the product taken in double.
__device__ __forceinline__ float alpha_at(const double *__restrict__ hat_x_,
                                          const double *__restrict__ hat_y_,
                                          const Taps &tp) {
    // (the hat tables live in global memory; a pointer read out of a camera record that was
    // staged in LDS is generic to the compiler and would load through the flat path)
    typedef const __attribute__((address_space(1))) double *hat_ptr;
    const hat_ptr hat_x = (hat_ptr)hat_x_, hat_y = (hat_ptr)hat_y_;
    const double hy0 = hat_y[tp.y0], hy1 = hat_y[tp.y1];
    const double hx0 = hat_x[tp.x0], hx1 = hat_x[tp.x1];
    return lerp4((float)(hy0 * hx0), (float)(hy0 * hx1), (float)(hy1 * hx0),
                 (float)(hy1 * hx1), tp);
}
