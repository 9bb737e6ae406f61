// cslam_adapter.hpp -- the reference-side binding of libcslam_hip.so: `HipEKF` and `HipPF`, subclasses of the
// reference's own back-ends that forward the hot-path virtuals of `class Slam` (slam/include/slam.h) to the C ABI of
// include/cslam.h.  Header-only; it is compiled BY the reference (which owns Eigen) -- a maintainer adds
//
//     #include "cslam_adapter.hpp"
//     std::shared_ptr<Slam> pSLAM = std::make_shared<HipEKF>(LM, WP, /*maxLandmarks*/ LM.cols());   // test/main.cpp:89
//
// and links -lcslam_hip.  Everything that is NOT on the hot path (dataAssociateTable, the simulator helpers, ...) stays
// the reference's own code, inherited from EKF / PF.
//
// Interfaces replaced (reference file:line):
//   HipEKF::predict          slam.h:841-847   EKF.cpp:406-455        -> cslam_ekf_predict
//   HipEKF::update           slam.h:938-943   EKF.cpp:481-496        -> cslam_ekf_update   (batchUpdate / singleUpdate too)
//   HipEKF::augment          slam.h:190-191   EKF.cpp:9-26           -> cslam_ekf_augment  (addOneNewFeature too)
//   HipEKF::observeHeading   slam.h:788       EKF.cpp:328-352        -> cslam_ekf_observe_heading
//   HipEKF::dataAssociate    slam.h:482-487   EKF.cpp:235-326        -> cslam_ekf_associate
//   HipPF::resampleParticles slam.h:871-872   PF.cpp:473-500         -> cslam_pf_resample_local / _sharded
//   HipPF (particle-set forms of) predict / observeHeading / sampleProposal / featureUpdate / addOneNewFeature
//                            slam.h:858-863, 796, 881-884, 549-552, 134; PF.cpp:419-471, 382-417, 502-544, 222-277, 9-60
//
// Ownership: the reference passes X and P by reference into every call; here the handle owns them in HBM.  X is
// refreshed after every call (the driver reads it every iteration, test/main.cpp:136); P is refreshed on demand
// (`syncP`), because the driver never reads it between calls (test/main.cpp:165-196).  If the caller edits X or P itself
// it calls `invalidate()` and the next call uploads them again.
//
// The reference's error convention is print-and-continue (e.g. EKF.cpp:125-128): `report` does the same with
// cslam_last_error().
//
// Build variants:
//   * inside the reference tree (Eigen + EKF.h + PF.h on the include path): nothing to define;
//   * CSLAM_ADAPTER_STANDIN="file.hpp": a header that declares stand-ins for the Eigen types and for EKF / PF with the
//     same virtual signatures -- used by this repository's CPU tests (tests/adapter_standin.hpp) to compile the
//     forwarding without Eigen.
#pragma once

#if defined(CSLAM_ADAPTER_STANDIN)
#include CSLAM_ADAPTER_STANDIN
#define CSLAM_ADAPTER_AVAILABLE 1
#elif defined(__has_include)
#if __has_include(<Eigen/Dense>) && __has_include("EKF.h") && __has_include("PF.h")
#include "EKF.h"
#include "PF.h"
#define CSLAM_ADAPTER_AVAILABLE 1
#endif
#endif

#if defined(CSLAM_ADAPTER_AVAILABLE)

#include <iostream>
#include <vector>

#include "cslam.h"

class HipEKF : public EKF
{
  public:
    /// maxLandmarks bounds the state (P is preallocated: augment is O(n), EKF.cpp:67-71's copy-resize disappears)
    HipEKF(const Eigen::MatrixXf& landMarks, const Eigen::MatrixXf& wayPoints, int maxLandmarks, int device = -1,
           int quirks = CSLAM_Q_REF_EXACT)
        : EKF(landMarks, wayPoints)
    {
        report(cslam_ekf_create(maxLandmarks, CSLAM_F32, device, quirks, &h_), "HipEKF::create");
    }
    ~HipEKF() { cslam_ekf_destroy(h_); }
    HipEKF(const HipEKF&)            = delete;
    HipEKF& operator=(const HipEKF&) = delete;

    cslam_ekf_t handle() const { return h_; }
    /// the caller changed X or P behind the engine's back: upload them again at the next call
    void invalidate() { uploaded_ = false; }
    /// refresh the caller's P (and X) from HBM -- the engine's P is the authoritative one between calls
    void syncP(Eigen::VectorXf& X, Eigen::MatrixXf& P)
    {
        int n = 0;
        cslam_ekf_get_n(h_, &n);
        X.resize(n);
        P.resize(n, n);
        report(cslam_ekf_get_state(h_, X.data(), P.data(), static_cast<int>(P.outerStride())), "HipEKF::syncP");
    }

    void predict(Eigen::VectorXf& X, Eigen::MatrixXf& P, const float& v, const float& swa, const Eigen::MatrixXf& Q,
                 const float& wb, const float& dt) override
    {
        push(X, P);
        report(cslam_ekf_predict(h_, v, swa, Q.data(), wb, dt), "HipEKF::predict");
        pullX(X);
    }

    void update(Eigen::VectorXf& X, Eigen::MatrixXf& P, const Eigen::MatrixXf& Z, const Eigen::MatrixXf& R,
                const Eigen::VectorXi& idf, bool batch = false) override
    {
        push(X, P);
        report(cslam_ekf_update(h_, Z.data(), static_cast<int>(Z.cols()), R.data(), idf.data(), batch ? 1 : 0),
               "HipEKF::update");
        pullX(X);
    }
    void batchUpdate(Eigen::VectorXf& X, Eigen::MatrixXf& P, const Eigen::MatrixXf& Z, const Eigen::MatrixXf& R,
                     const Eigen::VectorXi& idf) override
    {
        update(X, P, Z, R, idf, true);
    }
    void singleUpdate(Eigen::VectorXf& X, Eigen::MatrixXf& P, const Eigen::MatrixXf& Z, const Eigen::MatrixXf& R,
                      const Eigen::VectorXi& idf) override
    {
        update(X, P, Z, R, idf, false);
    }

    void augment(Eigen::VectorXf& X, Eigen::MatrixXf& P, const Eigen::MatrixXf& Z, const Eigen::MatrixXf& R) override
    {
        push(X, P);
        report(cslam_ekf_augment(h_, Z.data(), static_cast<int>(Z.cols()), R.data()), "HipEKF::augment");
        pullX(X); // X grows by two per new feature (EKF.cpp:40-49); P keeps growing on the device only
    }
    void addOneNewFeature(Eigen::VectorXf& X, Eigen::MatrixXf& P, const Eigen::MatrixXf& Z,
                          const Eigen::MatrixXf& R) override
    {
        augment(X, P, Z, R);
    }
    using EKF::addOneNewFeature; // (the particle overload stays the reference's empty stub, EKF.h:22-24)

    void observeHeading(Eigen::VectorXf& X, Eigen::MatrixXf& P, const float& phi, bool useHeading = false) override
    {
        push(X, P);
        report(cslam_ekf_observe_heading(h_, phi, useHeading ? 1 : 0), "HipEKF::observeHeading");
        pullX(X);
    }
    using EKF::observeHeading;

    Association_t dataAssociate(const Eigen::VectorXf& X, const Eigen::MatrixXf& P, const Eigen::MatrixXf& Z,
                                const Eigen::MatrixXf& R, const float& gate1, const float& gate2) override
    {
        push(X, P);
        const int        m = static_cast<int>(Z.cols());
        std::vector<int> idf(static_cast<size_t>(m > 0 ? m : 1)), kind(static_cast<size_t>(m > 0 ? m : 1));
        report(cslam_ekf_associate(h_, Z.data(), m, R.data(), gate1, gate2, idf.data(), kind.data()),
               "HipEKF::dataAssociate");
        int nf = 0;
        for (int i = 0; i < m; i++)
        {
            nf += (kind[static_cast<size_t>(i)] == 1) ? 1 : 0;
        }
        Association_t out;
        out.ZF.resize(2, nf);
        out.idf.resize(nf);
        out.ZN.resize(0, 0); // EKF.cpp:307 re-declares ZN: the reference returns an EMPTY new-feature list
        for (int i = 0, c = 0; i < m; i++)
        {
            if (kind[static_cast<size_t>(i)] == 1)
            {
                out.ZF(0, c)  = Z(0, i);
                out.ZF(1, c)  = Z(1, i);
                out.idf(c++) = idf[static_cast<size_t>(i)];
            }
        }
        return out;
    }

  private:
    cslam_ekf_t h_        = nullptr;
    bool        uploaded_ = false;

    static void report(int rc, const char* who)
    {
        if (rc != CSLAM_OK)
        {
            std::cout << cslam_last_error() << "\t" << who << std::endl; // cf. EKF.cpp:125-128
        }
    }
    void push(const Eigen::VectorXf& X, const Eigen::MatrixXf& P)
    {
        if (!uploaded_)
        {
            report(cslam_ekf_set_state(h_, X.data(), static_cast<int>(X.rows()), P.data(), static_cast<int>(P.outerStride())),
                   "HipEKF::set_state");
            uploaded_ = true;
        }
    }
    void pullX(Eigen::VectorXf& X)
    {
        int n = 0;
        cslam_ekf_get_n(h_, &n);
        if (X.rows() != n)
        {
            X.resize(n);
        }
        report(cslam_ekf_get_x(h_, X.data(), n), "HipEKF::get_x");
    }
};

// FastSLAM-2: the reference calls the per-particle virtuals in a loop over std::vector<Particle_t>
// (test/main.cpp:279-286, 303-311); the GPU wants the whole set at once, so HipPF adds set-level overloads with the
// same names and argument meaning, and overrides the one virtual that is set-level in the reference already:
// resampleParticles.  The particle set lives in HBM (structure of arrays); `upload` / `download` move it.
class HipPF : public PF
{
  public:
    HipPF(const Eigen::MatrixXf& landMarks, const Eigen::MatrixXf& wayPoints, int numParticles, int maxFeatures,
          int device = -1, int quirks = CSLAM_Q_REF_EXACT)
        : PF(landMarks, wayPoints)
        , np_(numParticles)
    {
        report(cslam_pf_create(numParticles, maxFeatures, CSLAM_F32, device, quirks, &h_), "HipPF::create");
        report(cslam_pf_set_uniform_weight(h_, 1.0 / numParticles), "HipPF::init"); // PF.cpp:327
    }
    ~HipPF() { cslam_pf_destroy(h_); }
    HipPF(const HipPF&)            = delete;
    HipPF& operator=(const HipPF&) = delete;

    cslam_pf_t handle() const { return h_; }

    void upload(const std::vector<Particle_t>& particles)
    {
        std::vector<float> pf;
        for (int i = 0; i < np_ && i < static_cast<int>(particles.size()); i++)
        {
            const Particle_t& p  = particles[static_cast<size_t>(i)];
            const int         nf = static_cast<int>(p.XF.cols());
            pf.resize(static_cast<size_t>(4 * nf));
            for (int f = 0; f < nf; f++)
            {
                for (int e = 0; e < 4; e++)
                {
                    pf[static_cast<size_t>(4 * f + e)] = p.PF[static_cast<size_t>(f)].data()[e];
                }
            }
            report(cslam_pf_set_particle(h_, i, &p.w, p.X.data(), p.P.data(), nf ? p.XF.data() : nullptr,
                                         nf ? pf.data() : nullptr, nf),
                   "HipPF::upload");
        }
    }
    void download(std::vector<Particle_t>& particles)
    {
        int np = 0, nf = 0;
        cslam_pf_get_counts(h_, &np, &nf);
        particles.resize(static_cast<size_t>(np));
        std::vector<float> pf(static_cast<size_t>(4 * (nf > 0 ? nf : 1)));
        for (int i = 0; i < np; i++)
        {
            Particle_t& p = particles[static_cast<size_t>(i)];
            p.X.resize(3);
            p.P.resize(3, 3);
            p.XF.resize(2, nf);
            p.PF.resize(static_cast<size_t>(nf));
            report(cslam_pf_get_particle(h_, i, &p.w, p.X.data(), p.P.data(), nf ? p.XF.data() : nullptr,
                                         nf ? pf.data() : nullptr),
                   "HipPF::download");
            for (int f = 0; f < nf; f++)
            {
                p.PF[static_cast<size_t>(f)].resize(2, 2);
                for (int e = 0; e < 4; e++)
                {
                    p.PF[static_cast<size_t>(f)].data()[e] = pf[static_cast<size_t>(4 * f + e)];
                }
            }
        }
    }

    // ---- set-level forms of the per-particle virtuals (every owned particle in one call)
    void predictAll(const float& v, const float& swa, const Eigen::MatrixXf& Q, const float& wb, const float& dt)
    {
        report(cslam_pf_predict(h_, v, swa, Q.data(), wb, dt), "HipPF::predict"); // PF.cpp:419-471
    }
    void observeHeadingAll(const float& phi, bool useHeading = false)
    {
        report(cslam_pf_observe_heading(h_, phi, useHeading ? 1 : 0), "HipPF::observeHeading"); // PF.cpp:382-417
    }
    /// normals: 3 x numParticles standard-normal draws, the ones slam.h:753-764 would make (column i for particle i)
    void sampleProposalAll(const Eigen::MatrixXf& Z, const Eigen::VectorXi& idf, const Eigen::MatrixXf& R,
                           const Eigen::MatrixXf& normals)
    {
        // the engine wants the draws component-major (normals[e * numParticles + p], one coalesced load per component);
        // an Eigen 3 x numParticles matrix is particle-major
        if (normals.rows() != 3 || normals.cols() != np_)
        {
            std::cout << "HipPF::sampleProposal: normals must be 3 x " << np_ << "\t" << "sampleProposal" << std::endl;
            return;
        }
        std::vector<float> nrm(static_cast<size_t>(3) * static_cast<size_t>(np_));
        for (int p = 0; p < np_; p++)
        {
            for (int e = 0; e < 3; e++)
            {
                nrm[static_cast<size_t>(e) * static_cast<size_t>(np_) + static_cast<size_t>(p)] = normals(e, p);
            }
        }
        report(cslam_pf_sample_proposal(h_, Z.data(), static_cast<int>(Z.cols()), idf.data(), R.data(), nrm.data()),
               "HipPF::sampleProposal"); // PF.cpp:502-544
    }
    void featureUpdateAll(const Eigen::MatrixXf& Z, const Eigen::VectorXi& idf, const Eigen::MatrixXf& R)
    {
        report(cslam_pf_feature_update(h_, Z.data(), static_cast<int>(Z.cols()), idf.data(), R.data()),
               "HipPF::featureUpdate"); // PF.cpp:222-277
    }
    void addNewFeaturesAll(const Eigen::MatrixXf& Z, const Eigen::MatrixXf& R)
    {
        report(cslam_pf_add_features(h_, Z.data(), static_cast<int>(Z.cols()), R.data()), "HipPF::addOneNewFeature");
    }

    /// the strata positions of PF.cpp:557 (stratifiedRandom): supplied by the caller so that runs are reproducible
    /// (the reference seeds from the clock, slam.h:587-594); when a communicator is set, `select` has one entry per
    /// particle of the WHOLE set (numParticles x ranks) and must be identical on every rank
    /// The reference draws fresh strata on every call (stratifiedRandom, PF.cpp:557): so must the caller -- the strata
    /// are consumed by the next resampleParticles and have to be set again before the one after.
    void setStrata(const Eigen::VectorXf& select) { select_ = select; }
    /// shard the particle set over ranks: resampleParticles then runs the three collectives of SURVEY 8e over RCCL
    void setCommunicator(cslam_comm_t comm)
    {
        comm_  = comm;
        world_ = 1;
        if (comm != nullptr)
        {
            int rank = 0;
            report(cslam_comm_info(comm, &rank, &world_), "HipPF::setCommunicator");
        }
    }

    /// slam.h:871-872, PF.cpp:473-500 -- on the particle set held in HBM.  `particles` is not touched: call download()
    /// when the host copy is needed.
    void resampleParticles(std::vector<Particle_t>& /*particles*/, int numEffective, bool resampleStatus = false) override
    {
        double neff      = 0.0;
        int    resampled = 0;
        // the engine reads numParticles x ranks strata positions from this pointer: refuse anything else, as loudly as
        // the reference's catch blocks do (PF.cpp:215-218), instead of reading past the caller's vector
        if (select_.rows() != static_cast<long>(np_) * world_)
        {
            std::cout << "HipPF::resampleParticles: setStrata() must supply " << static_cast<long>(np_) * world_
                      << " strata positions before every resample (got " << select_.rows() << ")"
                      << "\t" << "resampleParticles" << std::endl;
            return;
        }
        if (comm_ != nullptr)
        {
            report(cslam_pf_resample_sharded(h_, comm_, select_.data(), numEffective, resampleStatus ? 1 : 0, &neff,
                                             &resampled),
                   "HipPF::resampleParticles");
        }
        else
        {
            report(cslam_pf_resample_local(h_, select_.data(), numEffective, resampleStatus ? 1 : 0, &neff, &resampled),
                   "HipPF::resampleParticles");
        }
        lastNeff_      = static_cast<float>(neff);
        lastResampled_ = resampled != 0;
        select_.resize(0); // consumed: the next resample needs its own strata
    }
    float lastNeff() const { return lastNeff_; }
    bool  lastResampled() const { return lastResampled_; }

  private:
    cslam_pf_t      h_    = nullptr;
    cslam_comm_t    comm_ = nullptr;
    int             world_ = 1;
    int             np_   = 0;
    Eigen::VectorXf select_;
    float           lastNeff_      = 0.f;
    bool            lastResampled_ = false;

    static void report(int rc, const char* who)
    {
        if (rc != CSLAM_OK)
        {
            std::cout << cslam_last_error() << "\t" << who << std::endl; // cf. PF.cpp:215-218
        }
    }
};

#endif // CSLAM_ADAPTER_AVAILABLE
