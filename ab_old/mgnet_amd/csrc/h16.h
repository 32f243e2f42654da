// h16.h -- the 16-bit floating-point activation format of a translation unit.
//
// The kernels move activations as opaque 16-bit words; the format only matters where a word meets arithmetic: the MFMA
// instruction, the fp32 <-> 16-bit conversions of the epilogues and the element-wise kernels.  Every source that touches
// activations is therefore compiled twice -- `x.hip` for bf16 (the default of SOLVER.AMP in this stack) and `x_f16.hip`
// (= `#define MGN_F16` + `#include "x.hip"`) for IEEE fp16, the reference's AMP format (configs/MGNet-*.yaml: AMP ENABLED,
// torch.cuda.amp = fp16 + GradScaler) -- and exports its entry points as `mgn_<name>` and `mgn_<name>_f16`.
#pragma once
#include <stdint.h>

#ifdef MGN_F16
#define MGN_SYM(name) name##_f16
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
#define MGN_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define MGN_H16_LOWEST 0xfbffu   /* -65504 */
__device__ __forceinline__ float mgn_lo2f(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)); }
__device__ __forceinline__ float mgn_hi2f(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }
__device__ __forceinline__ float mgn_h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint32_t mgn_f2h(float f) { return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)f); }   // v_cvt_f16_f32: RNE, overflow -> inf
typedef float mgn_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 mgn_f16x2 __attribute__((ext_vector_type(2)));
#define MGN_HAVE_PACK2 1
__device__ __forceinline__ uint32_t mgn_pack2(float a, float b) {   // v_cvt_pk_f16_f32 (gfx950): both values in one instruction
    const mgn_f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, mgn_f16x2));
}
#else
#define MGN_SYM(name) name
typedef __bf16 h16x8 __attribute__((ext_vector_type(8)));
#define MGN_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define MGN_H16_LOWEST 0xff7fu   /* -3.39e38 */
__device__ __forceinline__ float mgn_lo2f(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float mgn_hi2f(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float mgn_h2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
// gfx950 converts in hardware (v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN, two values per instruction); the integer
// emulation `(u + 0x7fff + ((u >> 16) & 1)) >> 16` it replaces cost ~6 VALU instructions per value in every epilogue
typedef float mgn_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 mgn_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t mgn_f2h(float f) { return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)f); }
#define MGN_HAVE_PACK2 1
__device__ __forceinline__ uint32_t mgn_pack2(float a, float b) {
    const mgn_f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, mgn_bf16x2));
}
#endif
#ifndef MGN_HAVE_PACK2
__device__ __forceinline__ uint32_t mgn_pack2(float a, float b) { return mgn_f2h(a) | (mgn_f2h(b) << 16); }
#endif
