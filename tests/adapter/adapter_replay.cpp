// Drives the reference-side binding (include/cslam_adapter.hpp: HipEKF / HipPF) the way the reference's driver drives
// its back-ends -- through std::shared_ptr<Slam> and the virtuals of `class Slam` (test/main.cpp:89, 165-189, 279-311) --
// on a recorded call stream, and prints what the filter ended with.  Built with g++ against the Eigen-free stand-in
// (adapter_standin.hpp) and RUN on the GPU box by tests/test_adapter_gpu.py.  TEST INFRASTRUCTURE.
//
// Stream format (text, one call per line; floats printed with 9 significant digits, i.e. exact for float):
//   P v swa wb dt q00 q10 q01 q11                      Slam::predict          (main.cpp:165)
//   H phi use                                          Slam::observeHeading   (main.cpp:168)
//   U batch m  z[2m] idf[m]  r00 r10 r01 r11           Slam::update           (main.cpp:188)
//   A q  z[2q]  r00 r10 r01 r11                        Slam::augment          (main.cpp:189)
//   F np nf                                            start of the particle-filter section: np particles follow,
//     each `w x y phi p[9] xf[2nf] pf[4nf]`, then
//   p v swa wb dt q00 q10 q01 q11                      predict of every particle     (main.cpp:279-286)
//   s m z[2m] idf[m] r[4] normals[3np]                 sampleProposal + featureUpdate (main.cpp:305-309)
//   r numEffective select[np]                          resampleParticles             (main.cpp:310)
#define CSLAM_ADAPTER_STANDIN "adapter_standin.hpp"
#include "cslam_adapter.hpp"

#include <cstdio>
#include <fstream>
#include <memory>
#include <string>

static void read_mat(std::istream& in, Eigen::MatrixXf& M, long r, long c)
{
    M.resize(r, c);
    for (long j = 0; j < c; j++)
    {
        for (long i = 0; i < r; i++)
        {
            in >> M(i, j);
        }
    }
}

int main(int argc, char** argv)
{
    if (argc < 3)
    {
        std::fprintf(stderr, "usage: adapter_replay <stream> <max_landmarks>\n");
        return 2;
    }
    std::ifstream in(argv[1]);
    if (!in)
    {
        std::fprintf(stderr, "cannot open %s\n", argv[1]);
        return 2;
    }
    const int       max_lm = std::atoi(argv[2]);
    Eigen::MatrixXf LM(2, 1), WP(2, 1);
    // test/main.cpp:89 -- the driver only ever holds the base pointer
    std::shared_ptr<Slam> slam = std::make_shared<HipEKF>(LM, WP, max_lm);
    Eigen::VectorXf       X(3);    // main.cpp:107-108: X = 0_3, P = 0_3x3
    Eigen::MatrixXf       P(3, 3);
    Eigen::MatrixXf       Q, R, Z;
    Eigen::VectorXi       idf;
    long                  calls = 0;
    std::string           op;
    std::shared_ptr<HipPF> pf;
    std::vector<Slam::Particle_t> parts;
    int                    np = 0, nf = 0;
    while (in >> op)
    {
        calls++;
        if (op == "P")
        {
            float v, swa, wb, dt;
            in >> v >> swa >> wb >> dt;
            read_mat(in, Q, 2, 2);
            slam->predict(X, P, v, swa, Q, wb, dt);
        }
        else if (op == "H")
        {
            float phi;
            int   use;
            in >> phi >> use;
            slam->observeHeading(X, P, phi, use != 0);
        }
        else if (op == "U")
        {
            int batch, m;
            in >> batch >> m;
            read_mat(in, Z, 2, m);
            idf.resize(m);
            for (int i = 0; i < m; i++)
            {
                in >> idf(i);
            }
            read_mat(in, R, 2, 2);
            slam->update(X, P, Z, R, idf, batch != 0);
        }
        else if (op == "A")
        {
            int q;
            in >> q;
            read_mat(in, Z, 2, q);
            read_mat(in, R, 2, 2);
            slam->augment(X, P, Z, R);
        }
        else if (op == "F")
        {
            in >> np >> nf;
            parts.resize(static_cast<size_t>(np));
            for (auto& p : parts)
            {
                in >> p.w;
                p.X.resize(3);
                in >> p.X(0) >> p.X(1) >> p.X(2);
                read_mat(in, p.P, 3, 3);
                read_mat(in, p.XF, 2, nf);
                p.PF.resize(static_cast<size_t>(nf));
                for (auto& b : p.PF)
                {
                    read_mat(in, b, 2, 2);
                }
            }
            pf = std::make_shared<HipPF>(LM, WP, np, nf);
            pf->upload(parts);
        }
        else if (op == "p")
        {
            float v, swa, wb, dt;
            in >> v >> swa >> wb >> dt;
            read_mat(in, Q, 2, 2);
            pf->predictAll(v, swa, Q, wb, dt);
        }
        else if (op == "s")
        {
            int m;
            in >> m;
            read_mat(in, Z, 2, m);
            idf.resize(m);
            for (int i = 0; i < m; i++)
            {
                in >> idf(i);
            }
            read_mat(in, R, 2, 2);
            Eigen::MatrixXf normals;
            read_mat(in, normals, 3, np);
            pf->sampleProposalAll(Z, idf, R, normals);
            pf->featureUpdateAll(Z, idf, R);
        }
        else if (op == "r")
        {
            int neff_min;
            in >> neff_min;
            Eigen::VectorXf select(np);
            for (int i = 0; i < np; i++)
            {
                in >> select(i);
            }
            pf->setStrata(select);
            std::shared_ptr<Slam> base = pf; // main.cpp:310 calls it through the base pointer
            base->resampleParticles(parts, neff_min, true);
        }
        else
        {
            std::fprintf(stderr, "bad op '%s' at call %ld\n", op.c_str(), calls);
            return 2;
        }
    }
    // what the filter ended with: n, X (the adapter refreshed it after every call), trace(P) after syncP
    static_cast<HipEKF*>(slam.get())->syncP(X, P);
    const long n  = X.rows();
    double     tr = 0.0;
    for (long i = 0; i < n; i++)
    {
        tr += static_cast<double>(P(i, i));
    }
    int flags = 0;
    cslam_ekf_factor_status(static_cast<HipEKF*>(slam.get())->handle(), &flags, 0);
    std::printf("{\"calls\": %ld, \"n\": %ld, \"trace_P\": %.17g, \"factor_flags\": %d, \"X\": [", calls, n, tr, flags);
    for (long i = 0; i < n; i++)
    {
        std::printf("%s%.9g", i ? ", " : "", static_cast<double>(X(i)));
    }
    std::printf("]");
    if (pf)
    {
        pf->download(parts);
        std::printf(", \"pf\": {\"neff\": %.9g, \"resampled\": %d, \"particles\": [", static_cast<double>(pf->lastNeff()),
                    pf->lastResampled() ? 1 : 0);
        for (size_t i = 0; i < parts.size(); i++)
        {
            const auto& p = parts[i];
            std::printf("%s[%.9g, %.9g, %.9g, %.9g", i ? ", " : "", static_cast<double>(p.w), static_cast<double>(p.X(0)),
                        static_cast<double>(p.X(1)), static_cast<double>(p.X(2)));
            for (long f = 0; f < p.XF.cols(); f++)
            {
                std::printf(", %.9g, %.9g", static_cast<double>(p.XF(0, f)), static_cast<double>(p.XF(1, f)));
            }
            std::printf("]");
        }
        std::printf("]}");
    }
    std::printf("}\n");
    return 0;
}
