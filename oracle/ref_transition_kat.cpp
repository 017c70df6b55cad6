// ORACLE -- TEST INFRASTRUCTURE ONLY. Cross-check driver for the ONE file of the reference
// that compiles stand-alone: src/integrators/drmlt/tools/transition.h. The
// header is included where it lies under /root/reference (never copied); the build output
// goes to oracle/_ref/ only. Everything else of the reference needs Boost/Xerces/OpenEXR and
// is unbuildable in this image.
//
// The header expects to be included after Mitsuba's core headers; the few names it uses are
// declared here with the meaning Mitsuba gives them:
//   Float              platform.h:180 (double in the CMake build, float with -DREF_SINGLE)
//   Random::nextFloat  random.h (uniform in [0,1)); here it replays a caller-supplied list
//   math::fastexp/fastlog = exp/log (math.h:201-215), math::safe_acos = clamped acos
//
// usage: transition_kat <kind 0..3> <p0> <p1> <n_samples> <n_pdf>  < uniforms+du (binary doubles)
// prints: n_samples samples, then n_pdf (pdf, logpdf) pairs, one number per line (%.17g)
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifdef REF_SINGLE
typedef float Float;
#else
typedef double Float;
#endif
#define MTS_NAMESPACE_BEGIN namespace mitsuba {
#define MTS_NAMESPACE_END }
namespace mitsuba {
class Random {
public:
    explicit Random(const std::vector<double> &u) : m_u(u), m_i(0) {}
    Float nextFloat() { return (Float) m_u.at(m_i++); }
private:
    const std::vector<double> &m_u;
    size_t m_i;
};
namespace math {
inline Float fastexp(Float v) { return std::exp(v); }
inline Float fastlog(Float v) { return std::log(v); }
inline Float safe_acos(Float v) { return std::acos(std::min((Float) 1, std::max((Float) -1, v))); }
}
}
#include "transition.h"

int main(int argc, char **argv) {
    if (argc != 6) return 2;
    int kind = atoi(argv[1]);
    double p0 = atof(argv[2]), p1 = atof(argv[3]);
    size_t ns = (size_t) atol(argv[4]), np = (size_t) atol(argv[5]);
    std::vector<double> in;
    double v;
    while (fread(&v, sizeof v, 1, stdin) == 1) in.push_back(v);
    if (in.size() < np) return 3;
    std::vector<double> uni(in.begin(), in.end() - np), du(in.end() - np, in.end());
    mitsuba::Random rnd(uni);
    mitsuba::TransitionKernel *k = nullptr;
    switch (kind) {
        case 0: k = new mitsuba::GaussianKernel((Float) p0); break;
        case 1: k = new mitsuba::KelemenKernel((Float) p0, (Float) p1); break;
        case 2: k = new mitsuba::IdentityKernel(); break;
        default: k = new mitsuba::WrappedCauchyKernel((Float) p0); break;
    }
    for (size_t i = 0; i < ns; ++i) printf("%.17g\n", (double) k->sample(&rnd));
    for (size_t i = 0; i < np; ++i) printf("%.17g\n%.17g\n", (double) k->pdf((Float) du[i]), (double) k->logPdf((Float) du[i]));
    delete k;
    return 0;
}
