// Compiles include/cslam_adapter.hpp against the Eigen-free stand-in (tests/adapter/adapter_standin.hpp) and drives
// every forwarding member once through the `Slam` base pointer, exactly as the reference's driver does
// (test/main.cpp:89, 165-189, 279-311).  Built (and linked against libcslam_hip.so) by tests/test_capi_cpu.py; never run
// on the CPU box.  TEST INFRASTRUCTURE.
#define CSLAM_ADAPTER_STANDIN "adapter_standin.hpp"
#include "cslam_adapter.hpp"

#include <memory>

#ifndef CSLAM_ADAPTER_AVAILABLE
#error "the adapter did not enable itself"
#endif

int drive()
{
    Eigen::MatrixXf LM(2, 4), WP(2, 2), Q(2, 2), R(2, 2), Z(2, 1), P(3, 3);
    Eigen::VectorXf X(3);
    Eigen::VectorXi idf(1);
    std::shared_ptr<Slam> s = std::make_shared<HipEKF>(LM, WP, 16);
    s->predict(X, P, 83.33f, 0.1f, Q, 73.f, 0.01f);
    s->observeHeading(X, P, 0.1f, true);
    s->augment(X, P, Z, R);
    s->update(X, P, Z, R, idf, true);
    s->batchUpdate(X, P, Z, R, idf);
    s->singleUpdate(X, P, Z, R, idf);
    Slam::Association_t a = s->dataAssociate(X, P, Z, R, 4.f, 25.f);
    static_cast<HipEKF*>(s.get())->syncP(X, P);

    std::vector<Slam::Particle_t> parts(8);
    auto                          pf = std::make_shared<HipPF>(LM, WP, 8, 4);
    Eigen::MatrixXf               normals(3, 8);
    Eigen::VectorXf               select(8);
    pf->upload(parts);
    pf->predictAll(83.33f, 0.1f, Q, 73.f, 0.01f);
    pf->observeHeadingAll(0.1f, true);
    pf->addNewFeaturesAll(Z, R);
    pf->sampleProposalAll(Z, idf, R, normals);
    pf->featureUpdateAll(Z, idf, R);
    pf->setStrata(select);
    std::shared_ptr<Slam> sp = pf;
    sp->resampleParticles(parts, 6, true);
    pf->download(parts);
    return static_cast<int>(a.idf.rows()) + (pf->lastResampled() ? 1 : 0);
}

int main()
{
    return drive() >= 0 ? 0 : 1;
}
