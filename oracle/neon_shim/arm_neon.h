/* arm_neon.h — TEST INFRASTRUCTURE ONLY.
 *
 * A portable-C stand-in for the five NEON intrinsics that the reference's neon_search() uses
 * (krep.c:4528-4553: vdupq_n_u8, vld1q_u8, vceqq_u8, vmaxvq_u8, vst1q_u8).  oracle/Makefile puts this
 * directory on the include path and defines __ARM_NEON, so the UNMODIFIED reference sources compile
 * their arm64 code path (krep.c:69-74, :4505-4694) on this x86-64 box into oracle/_ref/libkrep_ref_neon.so.
 * That library pins the restatement ko_neon_search() (oracle/krep_oracle.c) and, through it, the
 * KREP_REF_NEON arm of the GPU backend.  Semantics follow the Arm C Language Extensions: lane-wise,
 * unsigned 8-bit, 16 lanes; compare yields 0xFF / 0x00 per lane.
 */
#ifndef KREP_ORACLE_NEON_SHIM_H
#define KREP_ORACLE_NEON_SHIM_H
#include <stdint.h>
#include <string.h>

typedef struct
{
    uint8_t lane[16];
} uint8x16_t;

static inline uint8x16_t vdupq_n_u8(uint8_t value)
{
    uint8x16_t r;
    memset(r.lane, value, sizeof r.lane);
    return r;
}
static inline uint8x16_t vld1q_u8(const uint8_t *ptr)
{
    uint8x16_t r;
    memcpy(r.lane, ptr, sizeof r.lane);
    return r;
}
static inline void vst1q_u8(uint8_t *ptr, uint8x16_t v) { memcpy(ptr, v.lane, sizeof v.lane); }
static inline uint8x16_t vceqq_u8(uint8x16_t a, uint8x16_t b)
{
    uint8x16_t r;
    for (int i = 0; i < 16; ++i)
        r.lane[i] = a.lane[i] == b.lane[i] ? 0xFF : 0x00;
    return r;
}
static inline uint8_t vmaxvq_u8(uint8x16_t v)
{
    uint8_t m = 0;
    for (int i = 0; i < 16; ++i)
        if (v.lane[i] > m)
            m = v.lane[i];
    return m;
}
#endif
