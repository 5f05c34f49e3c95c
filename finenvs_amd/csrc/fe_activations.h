// fe_activations.h -- part of fe_env.hip (one translation unit; see the overview there): the exact-operation sigmoid / tanh
// shared by the LSTM head (gates, actor output) and the MLP head (tanh hidden activation).
#pragma once
#include "fe_device_common.h"

namespace {

// ---- the LSTM head's sigmoid / tanh: exactly-rounded operations only, two activations per instruction ----
//   e = exp(-s |x|) (s = 1 sigmoid, 2 tanh; argument clamped at -60), Cephes expf's reduction and polynomial;
//   sigmoid = (x >= 0 ? 1 : e) / (1 + e),   tanh = copysign((1 - e) / (1 + e), x).
// Every step is an IEEE-exact f32 operation (mul, fma, rint, ldexp, and a division), so the same sequence on the CPU
// (fo_lstm_sigmoid / fo_lstm_tanh of the tests' restatement) gives the same bits.  The f32 MFMA shares the vector
// ALUs with these (SQ_VALU_MFMA_COEXEC_CYCLES = 0), so their instruction count is kernel time: the chains run on
// pairs of activations with packed-f32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32), and the division
// is the correctly-rounded rcp + fma sequence the compiler itself emits for `/`, minus its v_div_scale / v_div_fixup
// range handling -- the denominator is in [1, 2] and the numerator in {0} U [2^-87, 1], where that handling is the
// identity (this is why the argument clamp is -60: a smaller numerator would need the scaling).
// NaN is not propagated (a NaN pre-activation acts like -60); the host refuses non-finite weights.
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f pk_splat(float v) { return (v2f){v, v}; }

__device__ __forceinline__ v2f lstm_exp_nonpos2(v2f y0) {
    v2f y = {fmaxf(y0.x, -60.0f), fmaxf(y0.y, -60.0f)};
    v2f n = y * pk_splat(1.44269504f);
    n = (v2f){rintf(n.x), rintf(n.y)};
    v2f r = pk_fma(n, pk_splat(-0.693359375f), y);
    r = pk_fma(n, pk_splat(2.12194440e-4f), r);
    v2f q = pk_splat(1.9875691500e-4f);
    q = pk_fma(q, r, pk_splat(1.3981999507e-3f));
    q = pk_fma(q, r, pk_splat(8.3334519073e-3f));
    q = pk_fma(q, r, pk_splat(4.1665795894e-2f));
    q = pk_fma(q, r, pk_splat(1.6666665459e-1f));
    q = pk_fma(q, r, pk_splat(5.0000001201e-1f));
    const v2f r2 = r * r;
    q = pk_fma(q, r2, r);
    q = q + pk_splat(1.0f);
    return (v2f){ldexpf(q.x, (int)n.x), ldexpf(q.y, (int)n.y)};
}

// num / den, correctly rounded, for den in [1, 2] and num in {0} U [2^-87, 1] (see above)
__device__ __forceinline__ v2f lstm_div2(v2f num, v2f den) {
    v2f r = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    const v2f e0 = pk_fma(-den, r, pk_splat(1.0f));
    r = pk_fma(e0, r, r);
    v2f q = num * r;
    v2f rem = pk_fma(-den, q, num);
    q = pk_fma(rem, r, q);
    rem = pk_fma(-den, q, num);
    return pk_fma(rem, r, q);
}

// two activations at once; T0 / T1: the element is a tanh (else a sigmoid)
template <bool T0, bool T1>
__device__ __forceinline__ v2f lstm_act2(v2f x) {
    const v2f ax = {fabsf(x.x), fabsf(x.y)};
    const v2f e = lstm_exp_nonpos2(ax * (v2f){T0 ? -2.0f : -1.0f, T1 ? -2.0f : -1.0f});
    const v2f den = pk_splat(1.0f) + e;
    v2f num;
    num.x = T0 ? 1.0f - e.x : (x.x >= 0.0f ? 1.0f : e.x);
    num.y = T1 ? 1.0f - e.y : (x.y >= 0.0f ? 1.0f : e.y);
    v2f v = lstm_div2(num, den);
    if (T0) v.x = copysignf(v.x, x.x);
    if (T1) v.y = copysignf(v.y, x.y);
    return v;
}

__device__ __forceinline__ float lstm_tanh(float x) { return lstm_act2<true, true>((v2f){x, x}).x; }

}  // namespace
