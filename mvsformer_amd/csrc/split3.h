// The three-term bf16 split every "x3" kernel is built on (conv3d_x3.hip, vis_net_x3.hip, tail_x3.hip): an fp32 value v is EXACTLY
// h + m + l with h = bf16(v), m = bf16(v - h), l = bf16(v - h - m) - three 8-bit significands cover fp32's 24 bits, every difference is
// exact - so a product x*w = xh*wh + (xh*wm + xm*wh) + (xh*wl + xl*wh + xm*wm) + terms <= 2^-24 |x*w| is six v_mfma_f32_16x16x32_bf16
// with fp32 accumulation.  Edges (include/mvs_hip.h "Arithmetic", tests/test_hip_x3.py):
//   * |v| above the largest finite bf16 (0x7F7F0000 = 3.3895e38) would round to +-Inf: h is clamped to that value (one v_med3_f32), the
//     remainder (< 2^120) fits m and l exactly, so finite inputs up to FLT_MAX keep the exact three-term form;
//   * +-Inf / NaN: the remainder v - h is Inf / NaN, so at least one term is non-finite and every output the value reaches is non-finite;
//   * subnormal v: h, m, l are subnormal bf16 values; the matrix cores may flush them (absolute error <= 2^-126 per operand).
#pragma once

namespace mvsx3 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr float BF16_MAX = 3.3895313892515355e38f;         // 0x7F7F0000

__device__ __forceinline__ void split3(float v, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)__builtin_amdgcn_fmed3f(v, -BF16_MAX, BF16_MAX);
    const float r = v - (float)h;                          // exact
    m = (__bf16)r;
    l = (__bf16)(r - (float)m);                            // exact difference, exact conversion (8 bits left)
}

// the same split without the clamp, for values that cannot leave the finite bf16 range by construction (the visibility CNN's ReLU
// activations: BatchNorm'd sums of <= 144 products of bounded entropies) - one vector instruction less per value
__device__ __forceinline__ void split3_bounded(float v, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)v;
    const float r = v - (float)h;
    m = (__bf16)r;
    l = (__bf16)(r - (float)m);
}

// Two values at once, the form the kernels use: v_cvt_pk_bf16_f32 converts a PAIR per instruction, so the split of (a, b) is 3 conversions +
// 4 unpacks (shift / mask of the packed word: bf16 -> fp32 is exact) + 4 subtractions = 11 vector instructions (+2 clamps) against the 19 the
// compiler makes of two scalar splits (it converts every value alone, then converts again to pack).  Bit-identical to split3 per value.
// Each result word = {bf16(a) in the low half, bf16(b) in the high half}: consecutive elements of a bf16 vector.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
template <bool CLAMP>
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = CLAMP ? cvt_pk_bf16(__builtin_amdgcn_fmed3f(a, -BF16_MAX, BF16_MAX), __builtin_amdgcn_fmed3f(b, -BF16_MAX, BF16_MAX)) : cvt_pk_bf16(a, b);
    const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);      // exact
    m = cvt_pk_bf16(ra, rb);
    l = cvt_pk_bf16(ra - __builtin_bit_cast(float, m << 16), rb - __builtin_bit_cast(float, m & 0xffff0000u));           // exact, 8 bits left
}

struct Split3 { bf16x8 h, m, l; };
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Split3 split3(const float (&v)[8]) {
    u32x4_t h, m, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned a, b, c;
        split3_pair<true>(v[2 * e], v[2 * e + 1], a, b, c);
        h[e] = a;
        m[e] = b;
        l[e] = c;
    }
    Split3 s;
    s.h = __builtin_bit_cast(bf16x8, h);
    s.m = __builtin_bit_cast(bf16x8, m);
    s.l = __builtin_bit_cast(bf16x8, l);
    return s;
}

// one term (0 = h, 1 = m, 2 = l) of the split of f: the weight-packing kernels
__device__ __forceinline__ __bf16 split3_term(float f, int term) {
    __bf16 h, m, l;
    split3(f, h, m, l);
    return term == 0 ? h : (term == 1 ? m : l);
}

// six MFMAs of one fp32-equivalent K = 32 step, smallest products first; a = first MFMA operand's terms, b = second's
__device__ __forceinline__ __attribute__((ext_vector_type(4))) float mfma6(const bf16x8& ah, const bf16x8& am, const bf16x8& al, const bf16x8& bh,
                                                                          const bf16x8& bm, const bf16x8& bl,
                                                                          __attribute__((ext_vector_type(4))) float c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
    return c;
}

}  // namespace mvsx3
