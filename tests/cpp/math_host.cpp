// Host build of dbot_ros_amd/csrc/rbs_math.h (the F64 likelihood's exp / erfc / log): the same
// source the kernels compile, exposed to tests/test_math_cpu.py through a C-ABI so that its accuracy
// is checked against libm / mpmath without a GPU.  Test infrastructure; not part of the product.
#include "../../dbot_ros_amd/csrc/rbs_math.h"

extern "C" {
void rbsm_exp_nonpos(const double* x, double* out, long n)
{
    for (long i = 0; i < n; ++i) out[i] = rbsm::exp_nonpos(x[i], rbsm::kExpPoly);
}
void rbsm_erfc_pos(const double* x, double* out, long n)
{
    for (long i = 0; i < n; ++i) out[i] = rbsm::erfc_pos(x[i], rbsm::kErfcTab);
}
void rbsm_log_f32(const float* x, double* out, long n)
{
    for (long i = 0; i < n; ++i) out[i] = rbsm::log_f32(x[i], rbsm::kLogTab);
}
}

// The whole pixel likelihood as the F64 raster kernel evaluates it (rbsm::frame_terms +
// rbsm::depth_term + rbsm::pixel_loglik_f64), for n (observation, rendered depth, prior) triples.
extern "C" void rbsm_pixel_loglik(const float* obs, const float* depth, const float* prior, long n, double tw, double ms,
                                  double sf, double lam, double max_depth, double* ll, float* post)
{
    const rbsm::PixelConsts C = {lam, tw / max_depth, (1.0 - tw) / sqrt(M_PI)};
    for (long i = 0; i < n; ++i) {
        double t4[4];
        rbsm::frame_terms((double)obs[i], tw, ms, sf, lam, t4);
        const double g = rbsm::depth_term(C, (double)depth[i]);
        ll[i] = rbsm::pixel_loglik_f64(C, g, t4[0], t4[1], t4[2], t4[3], (double)depth[i], prior[i], rbsm::kErfcTab,
                                       rbsm::kLogTab, post[i]);
    }
}
