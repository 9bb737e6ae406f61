// Stand-ins for the types the reference's headers provide, so that include/cslam_adapter.hpp can be COMPILED in this
// repository's CPU tests without Eigen (which is not in the image) and without the reference tree.
//
// This is not the reference's code: only the shapes the adapter touches are declared -- a few members of
// Eigen::VectorXf / MatrixXf / VectorXi (data(), rows(), cols(), outerStride(), resize(), operator()) and the virtual
// signatures of the hot-path members of `class Slam` (slam/include/slam.h:134, 190-191, 201-204, 213-217, 482-487,
// 549-552, 788, 796, 841-847, 858-863, 871-872, 881-884, 912-916, 938-943) with `EKF` / `PF` as concrete subclasses,
// as in slam/include/EKF.h:5-13 and PF.h:5-14.  TEST INFRASTRUCTURE.
#pragma once
#include <cstddef>
#include <vector>

namespace Eigen
{
template <typename S>
struct Mat
{
    std::vector<S> v;
    long           r = 0, c = 0;
    Mat() {}
    explicit Mat(long rr) { resize(rr, 1); }
    Mat(long rr, long cc) { resize(rr, cc); }
    S*       data() { return v.data(); }
    const S* data() const { return v.data(); }
    long     rows() const { return r; }
    long     cols() const { return c; }
    long     outerStride() const { return r; }
    void     resize(long rr, long cc = 1)
    {
        r = rr;
        c = cc;
        v.assign(static_cast<size_t>(rr * cc), S());
    }
    S&       operator()(long i, long j) { return v[static_cast<size_t>(j * r + i)]; }
    const S& operator()(long i, long j) const { return v[static_cast<size_t>(j * r + i)]; }
    S&       operator()(long i) { return v[static_cast<size_t>(i)]; }
    const S& operator()(long i) const { return v[static_cast<size_t>(i)]; }
};
typedef Mat<float> MatrixXf;
typedef Mat<float> VectorXf;
typedef Mat<int>   VectorXi;
} // namespace Eigen

class Slam
{
  public:
    Slam(const Eigen::MatrixXf&, const Eigen::MatrixXf&) {}
    virtual ~Slam() = default;
    struct Particle_t
    {
        float                        w;
        Eigen::VectorXf              X;
        Eigen::MatrixXf              P;
        Eigen::MatrixXf              XF;
        std::vector<Eigen::MatrixXf> PF;
    };
    struct Association_t
    {
        Eigen::MatrixXf ZF;
        Eigen::MatrixXf ZN;
        Eigen::VectorXi idf;
    };
    virtual void addOneNewFeature(Particle_t& particle, const Eigen::MatrixXf& Z, const Eigen::MatrixXf& R) = 0;
    virtual void augment(Eigen::VectorXf& X, Eigen::MatrixXf& P, const Eigen::MatrixXf& Z, const Eigen::MatrixXf& R) = 0;
    virtual void addOneNewFeature(Eigen::VectorXf& X, Eigen::MatrixXf& P, const Eigen::MatrixXf& Z,
                                  const Eigen::MatrixXf& R) = 0;
    virtual void batchUpdate(Eigen::VectorXf& X, Eigen::MatrixXf& P, const Eigen::MatrixXf& Z, const Eigen::MatrixXf& R,
                             const Eigen::VectorXi& idf) = 0;
    virtual Association_t dataAssociate(const Eigen::VectorXf& X, const Eigen::MatrixXf& P, const Eigen::MatrixXf& Z,
                                        const Eigen::MatrixXf& R, const float& gate1, const float& gate2) = 0;
    virtual void featureUpdate(Particle_t& particle, const Eigen::MatrixXf& Z, const Eigen::VectorXi& idf,
                               const Eigen::MatrixXf& R) = 0;
    virtual void observeHeading(Eigen::VectorXf& X, Eigen::MatrixXf& P, const float& phi, bool useHeading = false) = 0;
    virtual void observeHeading(Particle_t& particle, const float& phi, bool useHeading = false) = 0;
    virtual void predict(Eigen::VectorXf& X, Eigen::MatrixXf& P, const float& v, const float& swa, const Eigen::MatrixXf& Q,
                         const float& wb, const float& dt) = 0;
    virtual void predict(Particle_t& particle, const float& v, const float& swa, const Eigen::MatrixXf& Q, const float& wb,
                         const float& dt) = 0;
    virtual void resampleParticles(std::vector<Particle_t>& particles, int numEffective, bool resampleStatus = false) = 0;
    virtual void sampleProposal(Particle_t& particle, const Eigen::MatrixXf& Z, const Eigen::VectorXi& idf,
                                const Eigen::MatrixXf& R) = 0;
    virtual void singleUpdate(Eigen::VectorXf& X, Eigen::MatrixXf& P, const Eigen::MatrixXf& Z, const Eigen::MatrixXf& R,
                              const Eigen::VectorXi& idf) = 0;
    virtual void update(Eigen::VectorXf& X, Eigen::MatrixXf& P, const Eigen::MatrixXf& Z, const Eigen::MatrixXf& R,
                        const Eigen::VectorXi& idf, bool batch = false) = 0;
};

// concrete back-ends with empty bodies (the real ones are the reference's EKF.cpp / PF.cpp)
#define STANDIN_BODIES                                                                                                     \
    void addOneNewFeature(Particle_t&, const Eigen::MatrixXf&, const Eigen::MatrixXf&) override {}                         \
    void augment(Eigen::VectorXf&, Eigen::MatrixXf&, const Eigen::MatrixXf&, const Eigen::MatrixXf&) override {}           \
    void addOneNewFeature(Eigen::VectorXf&, Eigen::MatrixXf&, const Eigen::MatrixXf&, const Eigen::MatrixXf&) override {}  \
    void batchUpdate(Eigen::VectorXf&, Eigen::MatrixXf&, const Eigen::MatrixXf&, const Eigen::MatrixXf&,                   \
                     const Eigen::VectorXi&) override {}                                                                   \
    Association_t dataAssociate(const Eigen::VectorXf&, const Eigen::MatrixXf&, const Eigen::MatrixXf&,                    \
                                const Eigen::MatrixXf&, const float&, const float&) override { return Association_t(); }   \
    void featureUpdate(Particle_t&, const Eigen::MatrixXf&, const Eigen::VectorXi&, const Eigen::MatrixXf&) override {}    \
    void observeHeading(Eigen::VectorXf&, Eigen::MatrixXf&, const float&, bool = false) override {}                        \
    void observeHeading(Particle_t&, const float&, bool = false) override {}                                               \
    void predict(Eigen::VectorXf&, Eigen::MatrixXf&, const float&, const float&, const Eigen::MatrixXf&, const float&,     \
                 const float&) override {}                                                                                 \
    void predict(Particle_t&, const float&, const float&, const Eigen::MatrixXf&, const float&, const float&) override {}  \
    void resampleParticles(std::vector<Particle_t>&, int, bool = false) override {}                                        \
    void sampleProposal(Particle_t&, const Eigen::MatrixXf&, const Eigen::VectorXi&, const Eigen::MatrixXf&) override {}   \
    void singleUpdate(Eigen::VectorXf&, Eigen::MatrixXf&, const Eigen::MatrixXf&, const Eigen::MatrixXf&,                  \
                      const Eigen::VectorXi&) override {}                                                                  \
    void update(Eigen::VectorXf&, Eigen::MatrixXf&, const Eigen::MatrixXf&, const Eigen::MatrixXf&, const Eigen::VectorXi&, \
                bool = false) override {}

class EKF : public Slam
{
  public:
    EKF(const Eigen::MatrixXf& lm, const Eigen::MatrixXf& wp) : Slam(lm, wp) {}
    STANDIN_BODIES
};
class PF : public Slam
{
  public:
    PF(const Eigen::MatrixXf& lm, const Eigen::MatrixXf& wp) : Slam(lm, wp) {}
    STANDIN_BODIES
};
