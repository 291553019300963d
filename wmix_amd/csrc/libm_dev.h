// libm_dev.h -- the reference's double-precision libm calls, rounded to float.
//
// The float NS / AEC call log / exp / pow / tanh on a float promoted to double and round the result back
// (ns_core.c:228,233,...; aec_core.c:278,1049).  Evaluating in double on the device and rounding to float
// reproduces glibc's correctly-rounded-in-practice results.  The bodies are kept out of
// line on purpose: inlined, the compiler hoists their ~40 registers of polynomial coefficients out of the
// kernels' packet loops and keeps them live for the whole kernel, which costs a wave per SIMD.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>

namespace wmx {

__device__ __noinline__ static float log_d(float x) { return (float)log((double)x); }
__device__ __noinline__ static float exp_d(float x) { return (float)exp((double)x); }
__device__ __noinline__ static float tanh_d(float x) { return (float)tanh((double)x); }
__device__ __noinline__ static float pow_d(float x, float y) { return (float)pow((double)x, (double)y); }

}  // namespace wmx
