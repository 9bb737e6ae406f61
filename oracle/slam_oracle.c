/*
 * slam_oracle.c -- instantiates the CPU oracle for float and double.
 * TEST INFRASTRUCTURE (see slam_oracle.h): never linked into or called by the HIP engine.
 * PARITY UNPINNED (reference not buildable here, ships no golden vectors) -- see slam_oracle.h.
 */
#include "slam_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* MSVC's std::_Pi_val, the constant the reference uses everywhere (slam.h:66-81, 818-825) */
#define ORC_PI 3.14159265358979323846264338327950288

/* ------------------------------------------------ float */
#define T float
#define FN(x) x##_f32
#define M_SIN sinf
#define M_COS cosf
#define M_SQRT sqrtf
#define M_ATAN2 atan2f
#define M_FMOD fmodf
#define M_FABS fabsf
#define M_EXP expf
#define M_LOG logf
#include "slam_oracle_impl.inc"
#include "slam_oracle_pf.inc"
#undef T
#undef FN
#undef M_SIN
#undef M_COS
#undef M_SQRT
#undef M_ATAN2
#undef M_FMOD
#undef M_FABS
#undef M_EXP
#undef M_LOG

/* ------------------------------------------------ double */
#define T double
#define FN(x) x##_f64
#define M_SIN sin
#define M_COS cos
#define M_SQRT sqrt
#define M_ATAN2 atan2
#define M_FMOD fmod
#define M_FABS fabs
#define M_EXP exp
#define M_LOG log
#include "slam_oracle_impl.inc"
#include "slam_oracle_pf.inc"
#undef T
#undef FN
#undef M_SIN
#undef M_COS
#undef M_SQRT
#undef M_ATAN2
#undef M_FMOD
#undef M_FABS
#undef M_EXP
#undef M_LOG

/* ---------------------------------------------------------------- EKF.cpp:146-233 */
int orc_data_associate_table(const void* Z, const int* tags, int m, int* table, int nf, void* ZF, int* idf,
                             void* ZN, int* mn, int elem_size)
{
    const char* z  = (const char*)Z;
    char*       zf = (char*)ZF;
    char*       zn = (char*)ZN;
    int         cf = 0, cn = 0;
    size_t      col = 2 * (size_t)elem_size;
    /* first pass (EKF.cpp:169-182): split by the table as it stands BEFORE this scan is added */
    int* newtags = (int*)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
    for (int i = 0; i < m; i++)
    {
        int id = tags[i];
        if (table[id - 1] == 0)
        {
            memcpy(zn + col * (size_t)cn, z + col * (size_t)i, col);
            newtags[cn] = id;
            cn++;
        }
        else
        {
            memcpy(zf + col * (size_t)cf, z + col * (size_t)i, col);
            idf[cf] = table[id - 1];
            cf++;
        }
    }
    /* EKF.cpp:213-226: new features get state positions nf+1, nf+2, ... */
    for (int i = 0; i < cn; i++)
    {
        table[newtags[i] - 1] = nf + i + 1;
    }
    free(newtags);
    *mn = cn;
    return cf;
}
