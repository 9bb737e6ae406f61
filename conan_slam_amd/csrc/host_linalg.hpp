// host_linalg.hpp -- small dense host routines of the engine's exceptional path.
//
// When the LLT of the innovation covariance fails, the reference falls back to an eigen-decomposition
// "square root" (slam.h:425-429) and a general inverse (slam.h:251).  That path is rare and k x k small,
// so the engine runs it on the host (sync mode only, see cslam_ekf_set_sync_mode).  This is product
// code: it does not use anything under oracle/.
#pragma once

#include <algorithm>
#include <cmath>
#include <vector>

namespace cslam
{

// Symmetric eigen-decomposition (cyclic Jacobi), eigenvalues ascending as Eigen's
// SelfAdjointEigenSolver returns them; lower triangle of M (k x k, column-major) is authoritative.
template <typename T>
void host_eigh(const T* M, int k, std::vector<T>& evals, std::vector<T>& V)
{
    std::vector<T> A((size_t)k * k);
    V.assign((size_t)k * k, (T)0);
    evals.assign((size_t)k, (T)0);
    auto at = [k](int r, int c) { return (size_t)c * k + r; };
    for (int c = 0; c < k; c++)
    {
        for (int r = 0; r < k; r++)
        {
            A[at(r, c)] = (r >= c) ? M[at(r, c)] : M[at(c, r)];
        }
        V[at(c, c)] = (T)1;
    }
    for (int sweep = 0; sweep < 64; sweep++)
    {
        double off = 0.0;
        for (int c = 0; c < k; c++)
        {
            for (int r = c + 1; r < k; r++)
            {
                off += (double)A[at(r, c)] * (double)A[at(r, c)];
            }
        }
        if (!(off > 0.0))
        {
            break;
        }
        for (int p = 0; p < k - 1; p++)
        {
            for (int q = p + 1; q < k; q++)
            {
                T apq = A[at(p, q)];
                if (apq == (T)0)
                {
                    continue;
                }
                T theta = (A[at(q, q)] - A[at(p, p)]) / ((T)2 * apq);
                T t     = (theta >= (T)0 ? (T)1 : (T)-1) / (std::fabs(theta) + std::sqrt(theta * theta + (T)1));
                T cs = (T)1 / std::sqrt(t * t + (T)1), sn = t * cs;
                for (int r = 0; r < k; r++)
                {
                    T arp = A[at(r, p)], arq = A[at(r, q)];
                    A[at(r, p)] = cs * arp - sn * arq;
                    A[at(r, q)] = sn * arp + cs * arq;
                }
                for (int c = 0; c < k; c++)
                {
                    T apc = A[at(p, c)], aqc = A[at(q, c)];
                    A[at(p, c)] = cs * apc - sn * aqc;
                    A[at(q, c)] = sn * apc + cs * aqc;
                }
                for (int r = 0; r < k; r++)
                {
                    T vrp = V[at(r, p)], vrq = V[at(r, q)];
                    V[at(r, p)] = cs * vrp - sn * vrq;
                    V[at(r, q)] = sn * vrp + cs * vrq;
                }
            }
        }
    }
    std::vector<int> order(k);
    for (int i = 0; i < k; i++)
    {
        order[i] = i;
        evals[i] = A[at(i, i)];
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return evals[a] < evals[b]; });
    std::vector<T> ev2(k), V2((size_t)k * k);
    for (int i = 0; i < k; i++)
    {
        ev2[i] = evals[order[i]];
        for (int r = 0; r < k; r++)
        {
            V2[at(r, i)] = V[at(r, order[i])];
        }
    }
    evals.swap(ev2);
    V.swap(V2);
}

// General inverse by LU with partial pivoting (what MatrixXf::inverse() does for dynamic sizes).
template <typename T>
void host_inverse(const T* Ain, int k, std::vector<T>& Ainv)
{
    std::vector<T>   LU(Ain, Ain + (size_t)k * k);
    std::vector<int> perm(k);
    auto             at = [k](int r, int c) { return (size_t)c * k + r; };
    for (int i = 0; i < k; i++)
    {
        perm[i] = i;
    }
    for (int c = 0; c < k; c++)
    {
        int piv  = c;
        T   best = std::fabs(LU[at(c, c)]);
        for (int r = c + 1; r < k; r++)
        {
            T a = std::fabs(LU[at(r, c)]);
            if (a > best)
            {
                best = a;
                piv  = r;
            }
        }
        if (best != (T)0)
        {
            if (piv != c)
            {
                for (int cc = 0; cc < k; cc++)
                {
                    std::swap(LU[at(c, cc)], LU[at(piv, cc)]);
                }
                std::swap(perm[c], perm[piv]);
            }
            T d = LU[at(c, c)];
            for (int r = c + 1; r < k; r++)
            {
                LU[at(r, c)] /= d;
            }
        }
        for (int cc = c + 1; cc < k; cc++)
        {
            T u = LU[at(c, cc)];
            for (int r = c + 1; r < k; r++)
            {
                LU[at(r, cc)] -= LU[at(r, c)] * u;
            }
        }
    }
    Ainv.assign((size_t)k * k, (T)0);
    for (int c = 0; c < k; c++)
    {
        T* x = &Ainv[at(0, c)];
        for (int r = 0; r < k; r++)
        {
            x[r] = (perm[r] == c) ? (T)1 : (T)0;
        }
        for (int r = 0; r < k; r++)
        {
            T s = x[r];
            for (int q = 0; q < r; q++)
            {
                s -= LU[at(r, q)] * x[q];
            }
            x[r] = s;
        }
        for (int r = k - 1; r >= 0; r--)
        {
            T s = x[r];
            for (int q = r + 1; q < k; q++)
            {
                s -= LU[at(r, q)] * x[q];
            }
            x[r] = s / LU[at(r, r)];
        }
    }
}

template <typename T>
bool host_all_finite(const std::vector<T>& v)
{
    for (T x : v)
    {
        if (!std::isfinite(x))
        {
            return false;
        }
    }
    return true;
}

// slam.h:425-434 + 251-255 for a matrix whose LLT failed: returns true and fills G (k x k, final
// orientation) when the eigen "square root" and its inverse are finite; false means "zeros" (no-op).
template <typename T>
bool host_eigen_fallback_gain(const T* S, int k, bool textbook, std::vector<T>& G)
{
    std::vector<T> ev, V;
    host_eigh(S, k, ev, V);
    std::vector<T> M((size_t)k * k);
    for (int c = 0; c < k; c++)
    {
        T s = std::sqrt(ev[c]); // NaN for a negative eigenvalue, as cwiseSqrt() gives
        for (int r = 0; r < k; r++)
        {
            M[(size_t)c * k + r] = V[(size_t)c * k + r] * s;
        }
    }
    if (!host_all_finite(M))
    {
        return false; // slam.h:431-434 -> zero factor -> inverse non-finite -> zeros
    }
    host_inverse(M.data(), k, G);
    if (textbook)
    {
        for (int c = 0; c < k; c++)
        {
            for (int r = c + 1; r < k; r++)
            {
                std::swap(G[(size_t)c * k + r], G[(size_t)r * k + c]);
            }
        }
    }
    return host_all_finite(G);
}

} // namespace cslam
