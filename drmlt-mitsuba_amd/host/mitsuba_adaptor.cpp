// Mitsuba 0.6 plugin `drmlt.so` backed by libdrmlt_amd.so: the thin adaptor of SURVEY.md 8(b).
//
// Build inside a Mitsuba build tree of the fork (its headers + Boost/Xerces/OpenEXR):
//   add to src/integrators/CMakeLists.txt
//     add_integrator(drmlt <path>/mitsuba_adaptor.cpp)      # replaces the four drmlt/*.cpp files
//     target_link_libraries(drmlt drmlt_amd)                # libdrmlt_amd.so + include/drmlt_abi.h
// None of those dependencies exist in the development image; there this file is compiled by build() against
// tests/native/fake_mitsuba/ (same class / method names and signatures, each checked against the reference headers)
// and driven by tests/native/adaptor_harness.cpp, so that everything below except the real headers is exercised.
//
// It keeps the reference's plugin surface (class name, parameters, RTTI, CreateInstance symbol:
// include/mitsuba/core/cobject.h:99-107, src/integrators/drmlt/drmlt.cpp:176-621), so
//   mitsuba scene.xml -D integrator=drmlt -D technique=path -D type=orbital
// runs against unmodified scene XML. Everything inside render() that the reference does on CPU threads (seeding,
// chain loop, film merge, develop) happens behind the C-ABI on the GPUs of `devices`.
#include <mitsuba/bidir/util.h>
#include <mitsuba/core/fresolver.h>
#include <mitsuba/core/plugin.h>
#include <mitsuba/core/statistics.h>
#include <mitsuba/render/renderjob.h>
#include <mitsuba/render/scene.h>
#include <mitsuba/render/trimesh.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <ctime>
#include <vector>

#include "drmlt_abi.h"

MTS_NAMESPACE_BEGIN

class DRMLT : public Integrator {
public:
    DRMLT(const Properties &props) : Integrator(props), m_stop(0) {
        memset(&m_cfg, 0, sizeof m_cfg);
        m_cfg.struct_size = sizeof m_cfg;
        m_cfg.algo = DRMLT_ALGO_DRMLT;
        // parameter names and defaults: drmlt.cpp:193-349
        std::string technique = props.getString("technique");
        if (technique == "path") m_cfg.technique = DRMLT_TECH_PATH;
        else if (technique == "bdpt") m_cfg.technique = DRMLT_TECH_BDPT;
        else if (technique == "mmlt") m_cfg.technique = DRMLT_TECH_MMLT;
        else Log(EError, "Unknown technique type");
        std::string type = props.getString("type");
        if (type == "green") m_cfg.type = DRMLT_TYPE_GREEN;
        else if (type == "mira") m_cfg.type = DRMLT_TYPE_MIRA;
        else if (type == "mirasym" || type == "orbital") m_cfg.type = DRMLT_TYPE_ORBITAL;
        else Log(EError, "Unknown implementation type");
        m_cfg.max_depth = props.getInteger("maxDepth", -1);
        m_cfg.rr_depth = props.getInteger("rrDepth", 5);
        m_cfg.direct_samples = props.getInteger("directSamples", 16);
        m_cfg.luminance_samples = props.getInteger("luminanceSamples", 100000);
        m_cfg.p_large = (float) props.getFloat("pLarge", 0.3f);
        m_cfg.work_units = props.getInteger("workUnits", -1);
        m_cfg.average_luminance = (float) props.getFloat("averageLuminance", -1.0f);
        m_cfg.acceptance_map = props.getBoolean("acceptanceMap", false);
        m_cfg.timid_after_large = props.getBoolean("timidAfterLarge", false);
        m_cfg.fix_emitter_path = props.getBoolean("fixEmitterPath", false);
        m_cfg.use_mixture = props.getBoolean("useMixture", false);
        m_cfg.sigma = (float) props.getFloat("sigma", 1.0f / 64.0f);
        m_cfg.scale_second = (float) props.getFloat("scaleSecond", 0.1);
        m_cfg.kelemen_style_weights = props.getBoolean("kelemenStyleWeights", true);
        m_cfg.kelemen_style_mutation = 1;
        m_twoStage = props.getBoolean("twoStage", false);
        // "Used internally to let the nested rendering process of a two-stage MLT approach know that it is running the first
        // stage" (drmlt.cpp:282-284): such an instance renders no first stage of its own and no direct image (:406,420,479)
        m_firstStage = props.getBoolean("firstStage", false);
        m_firstStageSizeReduction = props.getInteger("firstStageSizeReduction", 16);
        m_cfg.timeout_s = props.getInteger("timeout", 0);
        m_cfg.no_light_image = props.getBoolean("lightImage", true) ? 0 : 1;
        m_cfg.no_direct_sampling = (props.getBoolean("directSampling", true) && m_cfg.technique != DRMLT_TECH_MMLT) ? 0 : 1;
        // backend parameters (not in the reference):
        //  * firstStageSeeding = target (default) | reference: what two-stage chain seeds are resampled in proportion to -- the
        //    luminance under the importance map (the chains' own target) or this fork's pathsampler.cpp:901-905 (the plain
        //    luminance); include/drmlt_abi.h, DRMLT_SEED_*. Scene XML selects the upstream behaviour with "reference".
        //  * workUnitsRule = device (default) | reference: what workUnits = -1 derives (drmlt.cpp:434-444 under "reference")
        std::string seeding = props.getString("firstStageSeeding", "target");
        if (seeding == "target") m_cfg.seed_rule = DRMLT_SEED_TARGET;
        else if (seeding == "reference") m_cfg.seed_rule = DRMLT_SEED_REFERENCE;
        else Log(EError, "Unknown firstStageSeeding (target | reference)");
        std::string wuRule = props.getString("workUnitsRule", "device");
        if (wuRule == "device") m_cfg.work_units_rule = DRMLT_WORK_UNITS_DEVICE;
        else if (wuRule == "reference") m_cfg.work_units_rule = DRMLT_WORK_UNITS_REFERENCE;
        else Log(EError, "Unknown workUnitsRule (device | reference)");
        //  * which GPUs of the node render (bit d = HIP device d; `device` is the
        // single-GPU shorthand) and a fixed seed for reproducible renders (the reference seeds from /dev/urandom)
        m_deviceMask = props.hasProperty("devices") ? (uint32_t) props.getInteger("devices") : (1u << props.getInteger("device", 0));
        m_hasSeed = props.hasProperty("seed");
        m_seed = m_hasSeed ? (uint64_t) props.getInteger("seed") : 0;
    }

    DRMLT(Stream *stream, InstanceManager *manager) : Integrator(stream, manager), m_stop(0) {
        stream->read(&m_cfg, sizeof m_cfg);
        m_deviceMask = (uint32_t) stream->readInt();
        m_twoStage = stream->readBool();
        m_firstStage = stream->readBool();
        m_firstStageSizeReduction = stream->readInt();
        m_hasSeed = stream->readBool();
        m_seed = (uint64_t) stream->readInt();
    }

    void serialize(Stream *stream, InstanceManager *manager) const {
        Integrator::serialize(stream, manager);
        stream->write(&m_cfg, sizeof m_cfg);
        stream->writeInt((int) m_deviceMask);
        stream->writeBool(m_twoStage);
        stream->writeBool(m_firstStage);
        stream->writeInt(m_firstStageSizeReduction);
        stream->writeBool(m_hasSeed);
        stream->writeInt((int) m_seed);
    }

    bool preprocess(const Scene *scene, RenderQueue *queue, const RenderJob *job, int sceneResID, int sensorResID,
                    int samplerResID) {
        Integrator::preprocess(scene, queue, job, sceneResID, sensorResID, samplerResID);
        if (scene->getSubsurfaceIntegrators().size() > 0)
            Log(EError, "Subsurface integrators are not supported by MLT!");
        if (scene->getSensor()->getSampler()->getClass()->getName() != "IndependentSampler")
            Log(EError, "Metropolis light transport requires the independent sampler");
        return true;
    }

    void cancel() { m_stop = 1; } // asynchronous (integrator.h:77-84); polled between kernel launches by drmlt_run

    bool render(Scene *scene, RenderQueue *queue, const RenderJob *job, int sceneResID, int sensorResID,
                int samplerResID) {
        ref<Sensor> sensor = scene->getSensor();
        Film *film = sensor->getFilm();
        const Vector2i crop = film->getCropSize();
        m_cfg.sample_count = (int32_t) sensor->getSampler()->getSampleCount(); // mutations per pixel, drmlt.cpp:400
        m_stop = 0;

        // ---- flatten the scene through public accessors only
        std::vector<drmlt_shape> shapes;
        std::vector<drmlt_bsdf> bsdfs;
        std::vector<drmlt_emitter> emitters;
        const ref_vector<Shape> &mtsShapes = scene->getShapes();
        for (size_t i = 0; i < mtsShapes.size(); ++i)
            appendShape(mtsShapes[i].get(), shapes, bsdfs, emitters);

        drmlt_scene sc;
        memset(&sc, 0, sizeof sc);
        sc.struct_size = sizeof sc;
        sc.n_shapes = (int32_t) shapes.size(); sc.shapes = shapes.data();
        sc.n_bsdfs = (int32_t) bsdfs.size(); sc.bsdfs = bsdfs.data();
        sc.n_emitters = (int32_t) emitters.size(); sc.emitters = emitters.data();
        const PerspectiveCamera *cam = dynamic_cast<const PerspectiveCamera *>(sensor.get());
        if (!cam || cam->needsApertureSample())
            Log(EError, "The MI355X drmlt backend supports the `perspective` sensor only");
        const Transform toWorld = cam->getWorldTransform(0);
        const Matrix4x4 &m = toWorld.getMatrix();
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) sc.camera.to_world[r * 4 + c] = (float) m(r, c);
        sc.camera.fov_x_deg = (float) cam->getXFov();
        sc.camera.near_clip = (float) cam->getNearClip();
        sc.camera.far_clip = (float) cam->getFarClip();
        sc.camera.width = crop.x; sc.camera.height = crop.y;
        // the filter's own parameter, not getRadius(): BoxFilter stores radius + 1e-5 (box.cpp:38), GaussianFilter 4 stddev
        const ReconstructionFilter *rf = film->getReconstructionFilter();
        std::string rfName = rf->getClass()->getName();
        if (rfName == "BoxFilter") { sc.camera.filter = DRMLT_FILTER_BOX; sc.camera.filter_param = (float) rf->getProperties().getFloat("radius", 0.5f); }
        else if (rfName == "GaussianFilter") { sc.camera.filter = DRMLT_FILTER_GAUSSIAN; sc.camera.filter_param = (float) rf->getProperties().getFloat("stddev", 0.5f); }
        else Log(EError, "Unsupported reconstruction filter for the MI355X drmlt backend: %s", rfName.c_str());

        char err[512];
        uint64_t seed = m_hasSeed ? m_seed : (((uint64_t) (uintptr_t) this << 16) ^ (uint64_t) (uintptr_t) job ^ (uint64_t) time(NULL));

        // two-stage MLT (drmlt.cpp:406-418): the nested first stage is a second context on a reduced film; its
        // developed image becomes the luminance image of the full render (util.cpp:96-199)
        std::vector<float> importance;
        const bool nested = m_twoStage && m_firstStage; // drmlt.cpp:420
        if (m_twoStage && !m_firstStage) {
            Log(EInfo, "Executing first MLT stage");
            drmlt_scene small = sc;
            small.camera.width = std::max(1, crop.x / m_firstStageSizeReduction);
            small.camera.height = std::max(1, crop.y / m_firstStageSizeReduction);
            small.camera.filter = DRMLT_FILTER_GAUSSIAN; small.camera.filter_param = 0.5f;
            drmlt_config c1 = m_cfg;
            c1.sample_count = m_cfg.sample_count * m_firstStageSizeReduction;
            c1.acceptance_map = 0;
            int firstDevice = 0;
            while (firstDevice < 31 && !(m_deviceMask & (1u << firstDevice))) ++firstDevice;
            drmlt_ctx *first = drmlt_create(&c1, &small, firstDevice, err, sizeof err);
            if (!first) Log(EError, "%s", err);
            std::vector<float> img((size_t) small.camera.width * small.camera.height * 3);
            int rc1 = drmlt_seed(first, seed ^ 0x1571, 0, NULL);
            if (rc1 == DRMLT_OK)
                rc1 = drmlt_run(first, (uint64_t) small.camera.width * small.camera.height * (uint64_t) c1.sample_count, &m_stop, NULL, NULL);
            if (rc1 == DRMLT_OK) rc1 = drmlt_develop(first, NULL, img.data());
            drmlt_destroy(first);
            if (rc1 != DRMLT_OK) { Log(EWarn, "First-stage MLT process failed!"); return false; }
            importance.resize((size_t) crop.x * crop.y);
            drmlt_luminance_map(img.data(), small.camera.width, small.camera.height, crop.x, crop.y, importance.data());
        }

        // one node context drives every GPU of the mask: chains partitioned, one seed pool, film reduce over RCCL (C++)
        drmlt_node *node = drmlt_node_create(&m_cfg, &sc, m_deviceMask, err, sizeof err);
        if (!node) Log(EError, "%s", err); // throws, as the reference's parameter checks do
        if (!importance.empty() && drmlt_node_set_importance_map(node, importance.data()) != DRMLT_OK) {
            std::string msg = drmlt_node_last_error(node);
            drmlt_node_destroy(node);
            Log(EError, "%s", msg.c_str());
        }

        // separate direct pass stays on the host integrator (util.cpp:30-92), exactly as in the reference
        ref<Bitmap> directImage;
        if (m_cfg.direct_samples > 0 && !nested) { // drmlt.cpp:479
            directImage = BidirectionalUtils::renderDirectComponent(scene, sceneResID, sensorResID, queue, job,
                                                                    m_cfg.direct_samples);
            if (directImage == NULL) { drmlt_node_destroy(node); return false; }
        }

        double b = 0;
        int rc = drmlt_node_seed(node, seed, &b);
        if (rc == DRMLT_OK) {
            Log(EInfo, "Normalization factor computed: %lf", b);
            uint64_t total = (uint64_t) crop.x * crop.y * (uint64_t) m_cfg.sample_count;
            ProgressReporter progress("Rendering", (long long) total, job); // drmlt_proc.cpp:891
            rc = drmlt_node_run(node, total, &m_stop, &DRMLT::onProgress, &progress);
            if (rc == DRMLT_OK) progress.finish();
        }
        bool ok = rc == DRMLT_OK;
        if (ok) {
            ref<Bitmap> out = new Bitmap(Bitmap::ESpectrum, Bitmap::EFloat32, crop);
            std::vector<float> direct;
            if (directImage) {
                ref<Bitmap> d32 = directImage->convert(Bitmap::ESpectrum, Bitmap::EFloat32);
                direct.assign(d32->getFloat32Data(), d32->getFloat32Data() + (size_t) crop.x * crop.y * 3);
            }
            rc = drmlt_node_develop(node, direct.empty() ? NULL : direct.data(), out->getFloat32Data());
            ok = rc == DRMLT_OK;
            if (ok) { film->setBitmap(out); queue->signalRefresh(job); } // drmlt_proc.cpp:850-853
            drmlt_stats st;
            if (drmlt_node_stats_get(node, &st) == DRMLT_OK) logStats(st);
        }
        std::string msg = ok || rc == DRMLT_E_CANCELLED ? "" : drmlt_node_last_error(node);
        drmlt_node_destroy(node);
        if (!msg.empty()) Log(EError, "%s", msg.c_str());
        return ok; // false after cancel(), like the reference's m_process cancellation (drmlt.cpp:386-391,604-610)
    }

    MTS_DECLARE_CLASS()
private:
    static void onProgress(uint64_t done, uint64_t total, void *user) {
        (void) total;
        static_cast<ProgressReporter *>(user)->update((long long) done);
    }

    static void logStats(const drmlt_stats &s) { // the seven StatsCounters of drmlt_proc.cpp:34-49, in their order
        #define PCT(a, b) ((b) ? 100.0 * (double) (a) / (double) (b) : 0.0)
        SLog(EInfo, "Accepted 1st-stage mutations : %.2f %%", PCT(s.first_acc, s.first_base));
        SLog(EInfo, "Accepted large mutations in the 1st stage : %.2f %%", PCT(s.large_acc, s.large_base));
        SLog(EInfo, "Accepted bold mutation in the 1st stage : %.2f %%", PCT(s.bold_acc, s.bold_base));
        SLog(EInfo, "Accepted 2nd-stage mutations : %.2f %%", PCT(s.second_acc, s.second_base));
        SLog(EInfo, "Accepted 2nd-stage mutations after large mutation : %.2f %%", PCT(s.second_large_acc, s.second_large_base));
        SLog(EInfo, "Accepted 2nd-stage mutations after bold mutation : %.2f %%", PCT(s.second_bold_acc, s.second_bold_base));
        SLog(EInfo, "Overall acceptance rate : %.2f %%", PCT(s.overall_acc, s.overall_base));
        SLog(EInfo, "%.3e mutations/s on the device(s)", 1e3 * (double) s.mutations / s.kernel_ms);
        #undef PCT
    }

    // Index of refraction given as a number or as a material name (the BSDF plugins' private table, src/bsdfs/ior.h, is
    // not part of the public headers; these are the physical constants at ~589 nm)
    static Float lookupIOR(const Properties &props, const std::string &name, const std::string &def) {
        if (props.hasProperty(name) && props.getType(name) != Properties::EString) return props.getFloat(name);
        std::string v = props.getString(name, def);
        std::transform(v.begin(), v.end(), v.begin(), [](unsigned char c) { return (char) std::tolower(c); });
        static const struct { const char *name; double ior; } table[] = {
            {"vacuum", 1.0}, {"helium", 1.000036}, {"hydrogen", 1.000132}, {"air", 1.000277}, {"carbon dioxide", 1.00045},
            {"water", 1.3330}, {"acetone", 1.36}, {"ethanol", 1.361}, {"carbon tetrachloride", 1.461}, {"glycerol", 1.4729},
            {"benzene", 1.501}, {"silicone oil", 1.52045}, {"bromine", 1.661}, {"water ice", 1.31}, {"fused quartz", 1.458},
            {"pyrex", 1.470}, {"acrylic glass", 1.49}, {"polypropylene", 1.49}, {"bk7", 1.5046}, {"sodium chloride", 1.544},
            {"amber", 1.55}, {"pet", 1.5750}, {"diamond", 2.419}};
        for (const auto &e : table) if (v == e.name) return (Float) e.ior;
        Log(EError, "Unable to find an IOR value for \"%s\"", v.c_str());
        return 0;
    }

    int bsdfIndex(const BSDF *bsdf, std::vector<drmlt_bsdf> &bsdfs) {
        const Properties &p = bsdf->getProperties();
        std::string name = bsdf->getClass()->getName();
        drmlt_bsdf b;
        memset(&b, 0, sizeof b);
        Float r, g, bl;
        if (name == "SmoothDiffuse") {
            b.type = DRMLT_BSDF_DIFFUSE;
            Intersection its;
            bsdf->getDiffuseReflectance(its).toLinearRGB(r, g, bl); // constant textures only
            b.rgb[0] = (float) r; b.rgb[1] = (float) g; b.rgb[2] = (float) bl;
        } else if (name == "SmoothDielectric") {
            b.type = DRMLT_BSDF_DIELECTRIC;
            b.p[0] = (float) lookupIOR(p, "intIOR", "bk7");
            b.p[1] = (float) lookupIOR(p, "extIOR", "air");
        } else if (name == "RoughConductor") { // roughconductor.cpp:167-199 + microfacet.h:99-139
            b.type = DRMLT_BSDF_ROUGHCONDUCTOR;
            p.getSpectrum("specularReflectance", Spectrum(1.0f)).toLinearRGB(r, g, bl);
            b.rgb[0] = (float) r; b.rgb[1] = (float) g; b.rgb[2] = (float) bl;
            std::string distr = p.getString("distribution", "beckmann");
            std::transform(distr.begin(), distr.end(), distr.begin(), [](unsigned char c) { return (char) std::tolower(c); });
            if (distr == "beckmann") b.p[7] = 0.f;
            else if (distr == "ggx") b.p[7] = 1.f;
            else Log(EError, "roughconductor: the `%s` microfacet distribution has no MI355X drmlt implementation", distr.c_str());
            if (p.hasProperty("alphaU") || p.hasProperty("alphaV")) {
                if (!(p.hasProperty("alphaU") && p.hasProperty("alphaV")) || p.getFloat("alphaU") != p.getFloat("alphaV"))
                    Log(EError, "roughconductor: anisotropic roughness has no MI355X drmlt implementation");
                b.p[0] = (float) p.getFloat("alphaU");
            } else b.p[0] = (float) p.getFloat("alpha", 0.1f);
            b.p[0] = std::max(b.p[0], 1e-4f);
            if (!p.getBoolean("sampleVisible", true))
                Log(EError, "roughconductor: sampleVisible=false has no MI355X drmlt implementation");
            // complex IOR: the material's measured spectra unless eta / k are given; "none" = perfect mirror
            std::string material = p.getString("material", "Cu");
            std::transform(material.begin(), material.end(), material.begin(), [](unsigned char c) { return (char) std::tolower(c); });
            Spectrum intEta, intK;
            if (material == "none") { intEta = Spectrum(0.0f); intK = Spectrum(1.0f); }
            else if (!(p.hasProperty("eta") && p.hasProperty("k"))) {
                ref<FileResolver> fResolver = Thread::getThread()->getFileResolver();
                intEta.fromContinuousSpectrum(InterpolatedSpectrum(fResolver->resolve("data/ior/" + p.getString("material", "Cu") + ".eta.spd")));
                intK.fromContinuousSpectrum(InterpolatedSpectrum(fResolver->resolve("data/ior/" + p.getString("material", "Cu") + ".k.spd")));
            }
            const Float extEta = lookupIOR(p, "extEta", "air");
            (p.getSpectrum("eta", intEta) / extEta).toLinearRGB(r, g, bl);
            b.p[1] = (float) r; b.p[2] = (float) g; b.p[3] = (float) bl;
            (p.getSpectrum("k", intK) / extEta).toLinearRGB(r, g, bl);
            b.p[4] = (float) r; b.p[5] = (float) g; b.p[6] = (float) bl;
        } else {
            Log(EError, "BSDF type %s has no MI355X drmlt implementation (refusing rather than approximating)", name.c_str());
        }
        bsdfs.push_back(b);
        return (int) bsdfs.size() - 1;
    }

    void appendShape(const Shape *shape, std::vector<drmlt_shape> &shapes, std::vector<drmlt_bsdf> &bsdfs,
                     std::vector<drmlt_emitter> &emitters) {
        std::string name = shape->getClass()->getName();
        int bsdf = bsdfIndex(shape->getBSDF(), bsdfs);
        size_t first = shapes.size();
        drmlt_shape s;
        memset(&s, 0, sizeof s);
        s.bsdf = bsdf; s.emitter = -1;
        std::vector<double> areas; // per record, for area lights on meshes
        if (name == "Rectangle") {
            s.type = DRMLT_SHAPE_RECTANGLE;
            Transform t = shape->getProperties().getTransform("toWorld", Transform());
            if (shape->getProperties().getBoolean("flipNormals", false)) t = t * Transform::scale(Vector(1, 1, -1));
            const Matrix4x4 &m = t.getMatrix();
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) s.data[r * 4 + c] = (float) m(r, c);
            shapes.push_back(s);
        } else if (name == "Sphere") {
            s.type = DRMLT_SHAPE_SPHERE;
            AABB box = shape->getAABB();
            Point c = box.getCenter();
            s.data[0] = (float) c.x; s.data[1] = (float) c.y; s.data[2] = (float) c.z;
            s.data[3] = (float) (0.5f * (box.max.x - box.min.x));
            shapes.push_back(s);
        } else if (shape->getClass()->derivesFrom(MTS_CLASS(TriMesh))) {
            const TriMesh *mesh = static_cast<const TriMesh *>(shape);
            if (mesh->getVertexNormals() != NULL && !shape->getProperties().getBoolean("faceNormals", false))
                Log(EWarn, "Mesh \"%s\": smooth vertex normals are ignored (face normals are used)", mesh->getName().c_str());
            const Triangle *tri = mesh->getTriangles();
            const Point *pos = mesh->getVertexPositions();
            s.type = DRMLT_SHAPE_TRIANGLE;
            for (size_t i = 0; i < mesh->getTriangleCount(); ++i) {
                double q[3][3];
                for (int v = 0; v < 3; ++v) {
                    const Point &p = pos[tri[i].idx[v]];
                    q[v][0] = p.x; q[v][1] = p.y; q[v][2] = p.z;
                    s.data[3 * v] = (float) p.x; s.data[3 * v + 1] = (float) p.y; s.data[3 * v + 2] = (float) p.z;
                }
                const double e1[3] = {q[1][0] - q[0][0], q[1][1] - q[0][1], q[1][2] - q[0][2]}, e2[3] = {q[2][0] - q[0][0], q[2][1] - q[0][1], q[2][2] - q[0][2]};
                const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
                areas.push_back(0.5 * std::sqrt(cx * cx + cy * cy + cz * cz));
                shapes.push_back(s);
            }
        } else {
            Log(EError, "Shape type %s has no MI355X drmlt implementation", name.c_str());
        }
        if (shape->isEmitter()) {
            const Emitter *em = shape->getEmitter();
            if (em->getClass()->getName() != "AreaLight") Log(EError, "Only area emitters are supported");
            Float r, g, b;
            em->getProperties().getSpectrum("radiance").toLinearRGB(r, g, b);
            // An area light on a mesh: Mitsuba picks the emitter by its sampling weight and then a triangle in proportion to
            // its area (trimesh.cpp samplePosition). One emitter per triangle with weight x area share is the same density.
            const size_t n = shapes.size() - first;
            double total = 0;
            for (double a : areas) total += a;
            for (size_t i = 0; i < n; ++i) {
                drmlt_emitter e;
                memset(&e, 0, sizeof e);
                e.type = DRMLT_EMITTER_AREA;
                e.shape = (int32_t) (first + i);
                e.radiance[0] = (float) r; e.radiance[1] = (float) g; e.radiance[2] = (float) b;
                e.sampling_weight = (float) (em->getSamplingWeight() * (n > 1 && total > 0 ? areas[i] / total : 1.0));
                shapes[first + i].emitter = (int32_t) emitters.size();
                emitters.push_back(e);
            }
        }
    }

    drmlt_config m_cfg;
    uint32_t m_deviceMask;
    bool m_twoStage = false, m_firstStage = false;
    int m_firstStageSizeReduction = 16;
    bool m_hasSeed = false;
    uint64_t m_seed = 0;
    volatile int m_stop;
};

MTS_IMPLEMENT_CLASS_S(DRMLT, false, Integrator)
MTS_EXPORT_PLUGIN(DRMLT, "Delayed Rejection MLT (MI355X backend)");
MTS_NAMESPACE_END
