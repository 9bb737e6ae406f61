// device_math.hpp -- scalar device helpers shared by the EKF and the particle-filter kernels.
#pragma once

#include <hip/hip_runtime.h>

namespace cslam
{

constexpr double kPi = 3.14159265358979323846264338327950288; // the reference's std::_Pi_val

// ------------------------------------------------------------------------------------------------
// scalar helpers
// ------------------------------------------------------------------------------------------------
__device__ inline float  dsqrt(float x) { return sqrtf(x); }
__device__ inline double dsqrt(double x) { return sqrt(x); }
__device__ inline float  dabs(float x) { return fabsf(x); }
__device__ inline double dabs(double x) { return fabs(x); }
__device__ inline float  dlog(float x) { return logf(x); }
__device__ inline double dlog(double x) { return log(x); }
__device__ inline float  datan2(float y, float x) { return atan2f(y, x); }
__device__ inline double datan2(double y, double x) { return atan2(y, x); }
__device__ inline float  dsin(float x) { return sinf(x); }
__device__ inline double dsin(double x) { return sin(x); }
__device__ inline float  dcos(float x) { return cosf(x); }
__device__ inline double dcos(double x) { return cos(x); }
__device__ inline float  dfmod(float x, float y) { return fmodf(x, y); }
__device__ inline double dfmod(double x, double y) { return fmod(x, y); }
__device__ inline bool   dfinite(float x) { return isfinite(x); }
__device__ inline bool   dfinite(double x) { return isfinite(x); }

// slam.h:816-829 -- fmod at the scalar's precision, +-2pi corrections compared and added in double
template <typename T>
__device__ inline T pi2pi(T angle)
{
    angle = dfmod(angle, (T)(2.0 * kPi));
    if ((double)angle > kPi)
    {
        angle = (T)((double)angle - (2.0 * kPi));
    }
    if ((double)angle < -kPi)
    {
        angle = (T)((double)angle + (2.0 * kPi));
    }
    return angle;
}

} // namespace cslam
