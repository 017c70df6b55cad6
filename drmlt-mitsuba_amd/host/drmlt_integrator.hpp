// Stand-alone C++ host side above the C-ABI: the reference's `DRMLT` integrator surface
// (src/integrators/drmlt/drmlt.cpp:176-618) without Mitsuba. Same parameter names, defaults
// and error behaviour: a bad or missing parameter throws std::runtime_error, the way
// Log(EError, ...) does in the reference (logger.cpp:100-147); render() returns false on
// cancellation. Used by drmlt_render.cpp and by tests/test_host_cli.py.
#pragma once
#include "../../include/drmlt_abi.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace drmlt_host {

// mitsuba::Properties in miniature: typed getters with and without defaults
class Properties {
public:
    void set(const std::string &k, const std::string &v) { m_[k] = v; }
    bool has(const std::string &k) const { return m_.count(k) != 0; }
    std::string getString(const std::string &k) const {
        auto it = m_.find(k);
        if (it == m_.end()) throw std::runtime_error("Property \"" + k + "\" has not been specified!");
        return it->second;
    }
    std::string getString(const std::string &k, const std::string &d) const { return has(k) ? m_.at(k) : d; }
    int getInteger(const std::string &k, int d) const {
        if (!has(k)) return d;
        char *end = nullptr;
        long v = strtol(m_.at(k).c_str(), &end, 10);
        if (*end) throw std::runtime_error("Property \"" + k + "\" has the wrong type (expected <integer>)");
        return (int) v;
    }
    double getFloat(const std::string &k, double d) const {
        if (!has(k)) return d;
        char *end = nullptr;
        double v = strtod(m_.at(k).c_str(), &end);
        if (*end) throw std::runtime_error("Property \"" + k + "\" has the wrong type (expected <float>)");
        return v;
    }
    bool getBoolean(const std::string &k, bool d) const {
        if (!has(k)) return d;
        const std::string &v = m_.at(k);
        if (v == "true") return true;
        if (v == "false") return false;
        throw std::runtime_error("Property \"" + k + "\" has the wrong type (expected <boolean>)");
    }
private:
    std::map<std::string, std::string> m_;
};

// Flat scene as written by SceneData.save() (drmlt-mitsuba_amd/scenes.py): the arrays of drmlt_scene
struct SceneFile {
    std::vector<drmlt_shape> shapes;
    std::vector<drmlt_bsdf> bsdfs;
    std::vector<drmlt_emitter> emitters;
    drmlt_camera camera{};
    drmlt_scene view() const {
        drmlt_scene s;
        memset(&s, 0, sizeof s);
        s.struct_size = sizeof s;
        s.n_shapes = (int32_t) shapes.size(); s.shapes = shapes.data();
        s.n_bsdfs = (int32_t) bsdfs.size(); s.bsdfs = bsdfs.data();
        s.n_emitters = (int32_t) emitters.size(); s.emitters = emitters.data();
        s.camera = camera;
        return s;
    }
    static SceneFile load(const std::string &path) {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("cannot open scene file " + path);
        SceneFile sf;
        uint32_t hdr[8];
        bool ok = fread(hdr, sizeof hdr, 1, f) == 1 && hdr[0] == 0x4C4D5244u /* "DRML" */ && hdr[1] == DRMLT_ABI_VERSION &&
                  hdr[5] == sizeof(drmlt_shape) && hdr[6] == sizeof(drmlt_bsdf) && hdr[7] == sizeof(drmlt_emitter);
        if (ok) {
            sf.shapes.resize(hdr[2]); sf.bsdfs.resize(hdr[3]); sf.emitters.resize(hdr[4]);
            ok = fread(sf.shapes.data(), sizeof(drmlt_shape), hdr[2], f) == hdr[2] &&
                 fread(sf.bsdfs.data(), sizeof(drmlt_bsdf), hdr[3], f) == hdr[3] &&
                 fread(sf.emitters.data(), sizeof(drmlt_emitter), hdr[4], f) == hdr[4] &&
                 fread(&sf.camera, sizeof sf.camera, 1, f) == 1;
        }
        fclose(f);
        if (!ok) throw std::runtime_error("malformed scene file " + path);
        return sf;
    }
};

class DRMLTIntegrator {
public:
    // parameter surface of DRMLT::DRMLT(const Properties&), drmlt.cpp:193-349
    explicit DRMLTIntegrator(const Properties &props) {
        memset(&m_cfg, 0, sizeof m_cfg);
        m_cfg.struct_size = sizeof m_cfg;
        m_cfg.algo = DRMLT_ALGO_DRMLT;
        std::string technique = props.getString("technique");
        if (technique == "path") m_cfg.technique = DRMLT_TECH_PATH;
        else if (technique == "bdpt") m_cfg.technique = DRMLT_TECH_BDPT;
        else if (technique == "mmlt") m_cfg.technique = DRMLT_TECH_MMLT;
        else throw std::runtime_error("Unknown technique type");
        m_cfg.max_depth = props.getInteger("maxDepth", -1);
        if (m_cfg.technique == DRMLT_TECH_MMLT && m_cfg.max_depth == -1)
            throw std::runtime_error("Impossible to use MMLT with no max depth");
        m_cfg.rr_depth = props.getInteger("rrDepth", 5);
        m_cfg.direct_samples = props.getInteger("directSamples", 16);
        m_cfg.luminance_samples = props.getInteger("luminanceSamples", 100000);
        m_cfg.p_large = (float) props.getFloat("pLarge", 0.3);
        m_cfg.work_units = props.getInteger("workUnits", -1);
        m_cfg.kelemen_style_weights = props.getBoolean("kelemenStyleWeights", true);
        m_twoStage = props.getBoolean("twoStage", false);                             // drmlt.cpp:278
        m_firstStageSizeReduction = props.getInteger("firstStageSizeReduction", 16);  // :292-293
        if (m_firstStageSizeReduction <= 0) throw std::runtime_error("firstStageSizeReduction must be positive");
        m_cfg.timeout_s = props.getInteger("timeout", 0);                             // :296
        m_cfg.no_light_image = props.getBoolean("lightImage", true) ? 0 : 1;          // :301
        // :228-231 (forced off for mmlt). bdpt: the device builds the directSampling=false variant only (see drmlt_create)
        m_cfg.no_direct_sampling = (props.getBoolean("directSampling", true) && m_cfg.technique != DRMLT_TECH_MMLT) ? 0 : 1;
        m_cfg.average_luminance = (float) props.getFloat("averageLuminance", -1.0);
        std::string type = props.getString("type");
        if (type == "green") m_cfg.type = DRMLT_TYPE_GREEN;
        else if (type == "mira") m_cfg.type = DRMLT_TYPE_MIRA;
        else if (type == "mirasym" || type == "orbital") m_cfg.type = DRMLT_TYPE_ORBITAL;
        else throw std::runtime_error("Unknown implementation type");
        m_cfg.acceptance_map = props.getBoolean("acceptanceMap", false);
        m_cfg.timid_after_large = props.getBoolean("timidAfterLarge", false);
        m_cfg.fix_emitter_path = props.getBoolean("fixEmitterPath", false);
        if (m_cfg.technique != DRMLT_TECH_MMLT && m_cfg.fix_emitter_path)
            throw std::runtime_error("Impossible to use fixEmitterPath without MMLT");
        m_cfg.use_mixture = props.getBoolean("useMixture", false);
        m_cfg.sigma = (float) props.getFloat("sigma", 1.0 / 64.0);
        m_cfg.scale_second = (float) props.getFloat("scaleSecond", 0.1);
        if (m_cfg.scale_second > 1.0f) throw std::runtime_error("scaleSecond is bigger than the first stage");
        m_cfg.kelemen_style_mutation = 1;
        m_cfg.sample_count = props.getInteger("sampleCount", 4); // the independent sampler's sampleCount (default 4)
        m_device = props.getInteger("device", 0);
        m_seed = (uint64_t) props.getInteger("seed", 0x5EED);
    }

    const drmlt_config &config() const { return m_cfg; }
    void cancel() { m_stop = 1; } // asynchronous, like Integrator::cancel (integrator.h:77-84)

    // DRMLT::render: seeding -> chain loop -> develop. `out` receives W*H*3 floats.
    bool render(const SceneFile &scene, std::vector<float> &out, drmlt_stats *stats = nullptr, double *b_out = nullptr) {
        drmlt_scene sc = scene.view();
        char err[512] = {0};
        m_stop = 0;
        // two-stage MLT: nested first stage on a reduced film (BidirectionalUtils::mltLuminancePass, util.cpp:96-199)
        std::vector<float> importance;
        if (m_twoStage) {
            drmlt_scene small = sc;
            small.camera.width = std::max(1, sc.camera.width / m_firstStageSizeReduction);
            small.camera.height = std::max(1, sc.camera.height / m_firstStageSizeReduction);
            small.camera.filter = DRMLT_FILTER_GAUSSIAN; small.camera.filter_param = 0.5f; // the nested hdrfilm's default
            drmlt_config c1 = m_cfg;
            c1.sample_count = m_cfg.sample_count * m_firstStageSizeReduction;              // util.cpp:130-132
            c1.acceptance_map = 0;
            drmlt_ctx *first = drmlt_create(&c1, &small, m_device, err, sizeof err);
            if (!first) throw std::runtime_error(err);
            std::vector<float> img((size_t) small.camera.width * small.camera.height * 3);
            int rc1 = drmlt_seed(first, m_seed, 0, nullptr);
            if (rc1 == DRMLT_OK) rc1 = drmlt_run(first, (uint64_t) small.camera.width * small.camera.height * (uint64_t) c1.sample_count, &m_stop, nullptr, nullptr);
            if (rc1 == DRMLT_OK) rc1 = drmlt_develop(first, nullptr, img.data());
            std::string msg1 = rc1 == DRMLT_OK ? "" : drmlt_last_error(first);
            drmlt_destroy(first);
            if (rc1 == DRMLT_E_CANCELLED) return false;
            if (rc1 != DRMLT_OK) throw std::runtime_error("First-stage MLT process failed! " + msg1);
            importance.resize((size_t) sc.camera.width * sc.camera.height);
            drmlt_luminance_map(img.data(), small.camera.width, small.camera.height, sc.camera.width, sc.camera.height, importance.data());
        }
        drmlt_ctx *ctx = drmlt_create(&m_cfg, &sc, m_device, err, sizeof err);
        if (!ctx) throw std::runtime_error(err);
        double b = 0;
        int rc = importance.empty() ? DRMLT_OK : drmlt_set_importance_map(ctx, importance.data());
        if (rc == DRMLT_OK) rc = drmlt_seed(ctx, m_seed, 0, &b);
        if (rc == DRMLT_OK) {
            uint64_t total = (uint64_t) sc.camera.width * sc.camera.height * (uint64_t) m_cfg.sample_count;
            rc = drmlt_run(ctx, total, &m_stop, nullptr, nullptr);
        }
        bool ok = rc == DRMLT_OK;
        std::string msg;
        if (ok) {
            out.assign((size_t) sc.camera.width * sc.camera.height * 3, 0.f);
            rc = drmlt_develop(ctx, nullptr, out.data());
            ok = rc == DRMLT_OK;
        }
        if (!ok && rc != DRMLT_E_CANCELLED) msg = drmlt_last_error(ctx);
        if (stats) drmlt_stats_get(ctx, stats);
        if (b_out) *b_out = b;
        drmlt_destroy(ctx);
        if (!msg.empty()) throw std::runtime_error(msg);
        return ok;
    }

private:
    drmlt_config m_cfg;
    int m_device = 0, m_firstStageSizeReduction = 16;
    bool m_twoStage = false;
    uint64_t m_seed = 0x5EED;
    volatile int m_stop = 0;
};

inline void write_pfm(const std::string &path, int w, int h, const std::vector<float> &rgb) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    fprintf(f, "PF\n%d %d\n-1.0\n", w, h);
    for (int y = h - 1; y >= 0; --y) fwrite(&rgb[(size_t) y * w * 3], sizeof(float), (size_t) w * 3, f); // bottom-up
    fclose(f);
}

} // namespace drmlt_host
