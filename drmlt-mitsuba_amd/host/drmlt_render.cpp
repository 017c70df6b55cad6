// drmlt_render: `mitsuba scene.xml -D key=value -o out` in miniature for the MI355X backend.
//   drmlt_render scene.bin -D technique=path -D type=orbital -D maxDepth=8 -D sampleCount=64 -o out.pfm
//   drmlt_render --check -D ...      parse the parameters only (no GPU needed), print the configuration
// -D pairs are the plugin parameters of the reference (README.md:95-104, drmlt.cpp:193-349).
#include "drmlt_integrator.hpp"

#include <cstdlib>

using namespace drmlt_host;

int main(int argc, char **argv) {
    Properties props;
    std::string scene, out = "out.pfm";
    bool check = false;
    try {
        for (int i = 1; i < argc; ++i) {
            std::string a = argv[i];
            if (a == "-D" && i + 1 < argc) {
                std::string kv = argv[++i];
                size_t eq = kv.find('=');
                if (eq == std::string::npos) throw std::runtime_error("-D expects key=value, got \"" + kv + "\"");
                props.set(kv.substr(0, eq), kv.substr(eq + 1));
            } else if (a == "-o" && i + 1 < argc) out = argv[++i];
            else if (a == "--check") check = true;
            else if (a.size() && a[0] != '-') scene = a;
            else throw std::runtime_error("unknown argument " + a);
        }
        DRMLTIntegrator integrator(props);
        const drmlt_config &c = integrator.config();
        if (check) {
            printf("technique=%d type=%d maxDepth=%d rrDepth=%d directSamples=%d luminanceSamples=%d pLarge=%g workUnits=%d "
                   "sigma=%g scaleSecond=%g acceptanceMap=%d timidAfterLarge=%d useMixture=%d sampleCount=%d\n",
                   c.technique, c.type, c.max_depth, c.rr_depth, c.direct_samples, c.luminance_samples, c.p_large, c.work_units,
                   c.sigma, c.scale_second, c.acceptance_map, c.timid_after_large, c.use_mixture, c.sample_count);
            return 0;
        }
        if (scene.empty()) throw std::runtime_error("no scene file given");
        SceneFile sf = SceneFile::load(scene);
        std::vector<float> img;
        drmlt_stats st;
        double b = 0;
        bool ok = integrator.render(sf, img, &st, &b);
        if (!ok) { fprintf(stderr, "render cancelled\n"); return 2; }
        write_pfm(out, sf.camera.width, sf.camera.height, img);
        printf("b=%.9g mutations=%llu accepted=%llu kernel_ms=%.3f mutations_per_s=%.4e\n", b, (unsigned long long) st.mutations,
               (unsigned long long) st.accepted, st.kernel_ms, st.kernel_ms > 0 ? 1e3 * (double) st.mutations / st.kernel_ms : 0.0);
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "Error: %s\n", e.what()); // Log(EError) equivalent
        return 1;
    }
}
