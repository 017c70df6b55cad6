// Drives drmlt-mitsuba_amd/host/mitsuba_adaptor.cpp (compiled against tests/native/fake_mitsuba/) the way Mitsuba would:
// CreateInstance(props) -> preprocess -> render [-> cancel from a second thread], on a scene rebuilt from a flat scene file
// (SceneData.save of scenes.py) as Mitsuba objects (Rectangle / Sphere / TriMesh shapes, BSDF and emitter plugins with their
// Properties, perspective sensor, film, reconstruction filter, independent sampler).
//
//   adaptor_harness <scene.bin> <out prefix> [--meshlight] [--cancel-after-ms N] [-D key=value ...]
//
// Built twice by tests/test_adaptor.py:
//   -DMOCK_ABI  the drmlt_* entry points are defined HERE: they record what the adaptor hands over (<prefix>.scene in the
//               scene-file format, <prefix>.cfg = the drmlt_config bytes) and play a short render -- runs without a GPU;
//   (default)   linked with libdrmlt_amd.so: a real render through the plugin surface (<prefix>.img = W*H*3 floats).
// Log lines go to <prefix>.log as "<level>\t<text>".
#include <mitsuba/render/scene.h>

#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <thread>

#include "drmlt_abi.h"

using namespace mitsuba;

extern "C" void *CreateInstance(const Properties &props);

static std::string g_prefix;

#ifdef MOCK_ABI
// ------------------------------------------------------------------ a recording stand-in for libdrmlt_amd.so
struct drmlt_ctx { int w, h; };
struct drmlt_node { int w, h; std::string err; };
static void dump_scene(const drmlt_config *cfg, const drmlt_scene *sc, uint32_t mask) {
    FILE *f = fopen((g_prefix + ".scene").c_str(), "wb");
    uint32_t hdr[8] = {0x4C4D5244u, DRMLT_ABI_VERSION, (uint32_t) sc->n_shapes, (uint32_t) sc->n_bsdfs, (uint32_t) sc->n_emitters,
                       (uint32_t) sizeof(drmlt_shape), (uint32_t) sizeof(drmlt_bsdf), (uint32_t) sizeof(drmlt_emitter)};
    fwrite(hdr, sizeof hdr, 1, f);
    fwrite(sc->shapes, sizeof(drmlt_shape), sc->n_shapes, f);
    fwrite(sc->bsdfs, sizeof(drmlt_bsdf), sc->n_bsdfs, f);
    fwrite(sc->emitters, sizeof(drmlt_emitter), sc->n_emitters, f);
    fwrite(&sc->camera, sizeof sc->camera, 1, f);
    fclose(f);
    f = fopen((g_prefix + ".cfg").c_str(), "wb");
    fwrite(cfg, sizeof *cfg, 1, f);
    fwrite(&mask, sizeof mask, 1, f);
    fclose(f);
}
extern "C" {
drmlt_ctx *drmlt_create(const drmlt_config *, const drmlt_scene *sc, int, char *, size_t) { return new drmlt_ctx{sc->camera.width, sc->camera.height}; }
int drmlt_seed(drmlt_ctx *, uint64_t, uint32_t, double *b) { if (b) *b = 0.5; return DRMLT_OK; }
int drmlt_run(drmlt_ctx *, uint64_t, volatile int *, drmlt_progress_cb, void *) { return DRMLT_OK; }
int drmlt_develop(drmlt_ctx *c, const float *, float *out) { for (int i = 0; i < c->w * c->h * 3; ++i) out[i] = 0.5f; return DRMLT_OK; }
void drmlt_destroy(drmlt_ctx *c) { delete c; }
int drmlt_luminance_map(const float *, int, int, int W, int H, float *out) { for (int i = 0; i < W * H; ++i) out[i] = 2.0f; return DRMLT_OK; }
drmlt_node *drmlt_node_create(const drmlt_config *cfg, const drmlt_scene *sc, uint32_t mask, char *err, size_t errlen) {
    dump_scene(cfg, sc, mask);
    if (cfg->max_depth <= 0) { snprintf(err, errlen, "technique=path needs a finite maxDepth (pssmlt_utils.h:63)"); return nullptr; }
    return new drmlt_node{sc->camera.width, sc->camera.height, ""};
}
int drmlt_node_set_importance_map(drmlt_node *, const float *m) { FILE *f = fopen((g_prefix + ".imp").c_str(), "wb"); fwrite(m, 4, 1, f); fclose(f); return DRMLT_OK; }
int drmlt_node_seed(drmlt_node *, uint64_t seed, double *b) { FILE *f = fopen((g_prefix + ".seed").c_str(), "w"); fprintf(f, "%llu\n", (unsigned long long) seed); fclose(f); *b = 0.125; return DRMLT_OK; }
int drmlt_node_run(drmlt_node *n, uint64_t total, volatile int *stop, drmlt_progress_cb cb, void *user) {
    for (int i = 1; i <= 20; ++i) { // 20 "launches" of 10 ms
        if (stop && *stop) { n->err = "cancelled"; return DRMLT_E_CANCELLED; }
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
        if (cb) cb(total * i / 20, total, user);
    }
    return DRMLT_OK;
}
int drmlt_node_develop(drmlt_node *n, const float *direct, float *out) {
    for (int i = 0; i < n->w * n->h * 3; ++i) out[i] = 1.0f + (direct ? direct[i] : 0.f);
    return DRMLT_OK;
}
int drmlt_node_stats_get(drmlt_node *, drmlt_stats *s) {
    memset(s, 0, sizeof *s);
    s->first_acc = 10; s->first_base = 100; s->large_acc = 20; s->large_base = 100; s->bold_acc = 30; s->bold_base = 100;
    s->second_acc = 40; s->second_base = 100; s->second_large_acc = 50; s->second_large_base = 100;
    s->second_bold_acc = 60; s->second_bold_base = 100; s->overall_acc = 70; s->overall_base = 100;
    s->mutations = 1000; s->kernel_ms = 1.0;
    return DRMLT_OK;
}
const char *drmlt_node_last_error(drmlt_node *n) { return n->err.c_str(); }
void drmlt_node_destroy(drmlt_node *n) { delete n; }
}
#endif

// ------------------------------------------------------------------ flat scene file -> Mitsuba objects
static Class *named(const char *name, const Class *super) { return new Class(name, super); }

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    g_prefix = argv[2];
    bool meshlight = false;
    std::string rc_material; // rough conductors name a material instead of giving eta / k
    int cancel_ms = -1;
    Properties iprops;
    for (int i = 3; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--meshlight") meshlight = true;
        else if (a == "--rc-material" && i + 1 < argc) rc_material = argv[++i];
        else if (a == "--resolver-prefix" && i + 1 < argc) FileResolver::prefix() = argv[++i];
        else if (a == "--cancel-after-ms" && i + 1 < argc) cancel_ms = atoi(argv[++i]);
        else if (a == "-D" && i + 1 < argc) {
            std::string kv = argv[++i], k = kv.substr(0, kv.find('=')), v = kv.substr(kv.find('=') + 1);
            char *end = nullptr;
            long iv = strtol(v.c_str(), &end, 10);
            if (v == "true" || v == "false") iprops.setBoolean(k, v == "true");
            else if (*end == 0 && !v.empty()) iprops.setInteger(k, (int) iv);
            else { double dv = strtod(v.c_str(), &end); if (*end == 0 && !v.empty()) iprops.setFloat(k, dv); else iprops.setString(k, v); }
        }
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
    uint32_t hdr[8];
    if (fread(hdr, sizeof hdr, 1, f) != 1 || hdr[0] != 0x4C4D5244u) return 2;
    std::vector<drmlt_shape> shapes(hdr[2]);
    std::vector<drmlt_bsdf> bsdfs(hdr[3]);
    std::vector<drmlt_emitter> emitters(hdr[4]);
    drmlt_camera cam;
    bool ok = fread(shapes.data(), sizeof(drmlt_shape), shapes.size(), f) == shapes.size() && fread(bsdfs.data(), sizeof(drmlt_bsdf), bsdfs.size(), f) == bsdfs.size() &&
              fread(emitters.data(), sizeof(drmlt_emitter), emitters.size(), f) == emitters.size() && fread(&cam, sizeof cam, 1, f) == 1;
    fclose(f);
    if (!ok) return 2;

    ref<Scene> scene = new Scene();
    auto make_bsdf = [&](const drmlt_bsdf &b) -> BSDF * {
        Properties p;
        const char *cls = "?";
        if (b.type == DRMLT_BSDF_DIFFUSE) { cls = "SmoothDiffuse"; p.setSpectrum("reflectance", Spectrum(b.rgb[0], b.rgb[1], b.rgb[2])); }
        else if (b.type == DRMLT_BSDF_DIELECTRIC) { cls = "SmoothDielectric"; p.setFloat("intIOR", b.p[0]); p.setFloat("extIOR", b.p[1]); }
        else if (b.type == DRMLT_BSDF_ROUGHCONDUCTOR) {
            cls = "RoughConductor";
            p.setSpectrum("specularReflectance", Spectrum(b.rgb[0], b.rgb[1], b.rgb[2]));
            p.setFloat("alpha", b.p[0]);
            p.setString("distribution", b.p[7] != 0.f ? "GGX" : "beckmann");
            if (rc_material.empty()) { p.setSpectrum("eta", Spectrum(b.p[1], b.p[2], b.p[3])); p.setSpectrum("k", Spectrum(b.p[4], b.p[5], b.p[6])); }
            else p.setString("material", rc_material);
            p.setFloat("extEta", 1.0); // the file holds eta / extEta already
        }
        BSDF *o = new BSDF(p);
        o->m_class = named(cls, BSDF::m_theClass);
        return o;
    };
    auto make_emitter = [&](const drmlt_emitter &e) -> Emitter * {
        Properties p;
        p.setSpectrum("radiance", Spectrum(e.radiance[0], e.radiance[1], e.radiance[2]));
        p.setFloat("samplingWeight", e.sampling_weight);
        Emitter *o = new Emitter(p);
        o->m_class = named("AreaLight", ConfigurableObject::m_theClass);
        return o;
    };
    for (const drmlt_shape &s : shapes) {
        ref<Shape> sh;
        if (s.type == DRMLT_SHAPE_RECTANGLE && !(meshlight && s.emitter >= 0)) {
            Properties p;
            Matrix4x4 m;
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) m(r, c) = s.data[r * 4 + c];
            p.setTransform("toWorld", Transform(m));
            sh = new Shape(p);
            sh->m_class = named("Rectangle", Shape::m_theClass);
        } else if (s.type == DRMLT_SHAPE_RECTANGLE) { // --meshlight: the emitting rectangle as a two-triangle mesh (rectangle.cpp:170-203)
            TriMesh *mesh = new TriMesh();
            auto corner = [&](double u, double v) { return Point(s.data[0] * u + s.data[1] * v + s.data[3], s.data[4] * u + s.data[5] * v + s.data[7], s.data[8] * u + s.data[9] * v + s.data[11]); };
            mesh->m_pos = {corner(-1, -1), corner(1, -1), corner(1, 1), corner(-1, 1)};
            mesh->m_tris = {Triangle{{0, 1, 2}}, Triangle{{2, 3, 0}}};
            mesh->m_class = named("ObjMesh", TriMesh::m_theClass); // an emitting quad loaded as a mesh
            sh = mesh;
        } else if (s.type == DRMLT_SHAPE_SPHERE) {
            sh = new Shape(Properties());
            sh->m_class = named("Sphere", Shape::m_theClass);
            sh->m_aabb.min = Point(s.data[0] - s.data[3], s.data[1] - s.data[3], s.data[2] - s.data[3]);
            sh->m_aabb.max = Point(s.data[0] + s.data[3], s.data[1] + s.data[3], s.data[2] + s.data[3]);
        } else {
            TriMesh *mesh = new TriMesh();
            for (int v = 0; v < 3; ++v) mesh->m_pos.push_back(Point(s.data[3 * v], s.data[3 * v + 1], s.data[3 * v + 2]));
            mesh->m_tris.push_back(Triangle{{0, 1, 2}});
            mesh->m_class = named("ObjMesh", TriMesh::m_theClass);
            sh = mesh;
        }
        sh->m_bsdf = make_bsdf(bsdfs[s.bsdf]);
        if (s.emitter >= 0) sh->m_emitter = make_emitter(emitters[s.emitter]);
        scene->m_shapes.push_back(sh);
    }
    ref<PerspectiveCamera> camera = new PerspectiveCamera();
    Matrix4x4 m;
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) m(r, c) = cam.to_world[r * 4 + c];
    camera->m_toWorld = Transform(m);
    camera->m_xfov = cam.fov_x_deg; camera->m_nearClip = cam.near_clip; camera->m_farClip = cam.far_clip;
    ref<Film> film = new Film();
    film->m_cropSize = Vector2i(cam.width, cam.height);
    Properties fp;
    ref<ReconstructionFilter> rf;
    if (cam.filter == DRMLT_FILTER_BOX) { fp.setFloat("radius", cam.filter_param); rf = new ReconstructionFilter(); rf->m_radius = cam.filter_param + 1e-5f; rf->m_class = named("BoxFilter", ConfigurableObject::m_theClass); }
    else { fp.setFloat("stddev", cam.filter_param); rf = new ReconstructionFilter(); rf->m_radius = 4 * cam.filter_param; rf->m_class = named("GaussianFilter", ConfigurableObject::m_theClass); }
    *const_cast<Properties *>(&rf->getProperties()) = fp;
    film->m_filter = rf;
    camera->m_film = film;
    ref<Sampler> sampler = new Sampler();
    sampler->m_sampleCount = (size_t) iprops.getInteger("sampleCount", 4);
    if (iprops.getString("sampler", "independent") != "independent") sampler->m_class = named("LowDiscrepancySampler", ConfigurableObject::m_theClass);
    camera->m_sampler = sampler;
    scene->m_sensor = camera.get();

    int rc = 0;
    ref<RenderQueue> queue = new RenderQueue();
    ref<RenderJob> job = new RenderJob();
    try {
        ref<Integrator> integrator = static_cast<Integrator *>(CreateInstance(iprops));
        if (iprops.getBoolean("roundtrip", false)) { // Integrator(Stream*) / serialize: what the scheduler does with remote workers
            Stream st;
            integrator->serialize(&st, nullptr);
            FILE *g = fopen((g_prefix + ".ser").c_str(), "wb"); fwrite(st.buf.data(), 1, st.buf.size(), g); fclose(g);
        }
        integrator->preprocess(scene, queue, job, 0, 1, 2);
        std::thread canceller;
        if (cancel_ms >= 0) canceller = std::thread([&] { std::this_thread::sleep_for(std::chrono::milliseconds(cancel_ms)); integrator->cancel(); });
        const bool done = integrator->render(scene, queue, job, 0, 1, 2);
        if (canceller.joinable()) canceller.join();
        FakeLog::log(EInfo, "render returned %s, %d refresh signal(s), %d direct pass sample(s)", done ? "true" : "false", queue->refreshes, BidirectionalUtils::calls());
        if (done) {
            FILE *g = fopen((g_prefix + ".img").c_str(), "wb");
            fwrite(film->m_result.data(), sizeof(float), film->m_result.size(), g);
            fclose(g);
        }
    } catch (const std::exception &e) {
        rc = 1; // Log(EError) of the plugin
    }
    f = fopen((g_prefix + ".log").c_str(), "w");
    for (auto &l : FakeLog::lines()) fprintf(f, "%d\t%s\n", l.first, l.second.c_str());
    fprintf(f, "%d\tprogress updates: %zu\n", (int) EInfo, ProgressReporter::updates().size());
    fclose(f);
    return rc;
}
