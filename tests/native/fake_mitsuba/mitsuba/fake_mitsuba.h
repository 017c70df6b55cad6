// A FAKE of the part of Mitsuba 0.6's public API that drmlt-mitsuba_amd/host/mitsuba_adaptor.cpp touches: same class
// and method names and signatures (checked against the reference's include/mitsuba/**: sensor.h:403-499, film.h:39-115,
// rfilter.h:67, sampler.h:156, scene.h:1011-1115, trimesh.h:122-139, shape.h:223,458-478, bsdf.h:337, emitter.h:536,
// renderqueue.h:105, bidir/util.h:40, cobject.h:77-107, properties.h, bitmap.h:685,1216, spectrum.h:796, statistics.h:287-311),
// with just enough behaviour behind them for the adaptor to be compiled and driven on a machine that has neither Mitsuba
// nor its dependencies (Boost, Xerces, OpenEXR). Test infrastructure: nothing here is Mitsuba code and
// nothing here ships; a real build uses the real headers.
#pragma once
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#define MTS_NAMESPACE_BEGIN namespace mitsuba {
#define MTS_NAMESPACE_END }

namespace mitsuba {

typedef double Float; // the reference's CMake default (DOUBLE_PRECISION)

// ---- logging: EError throws (logger.cpp:100-147), everything is recorded for the tests
enum ELogLevel { ETrace = 0, EDebug = 100, EInfo = 200, EWarn = 300, EError = 400 };
struct FakeLog {
    static std::vector<std::pair<int, std::string>> &lines() { static std::vector<std::pair<int, std::string>> l; return l; }
    static void log(ELogLevel level, const char *fmt, ...) {
        char buf[1024];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        lines().push_back({(int) level, buf});
        if (level >= EError) throw std::runtime_error(buf);
    }
};
#define Log(level, ...) ::mitsuba::FakeLog::log(level, __VA_ARGS__)
#define SLog(level, ...) ::mitsuba::FakeLog::log(level, __VA_ARGS__)

// ---- RTTI
class Class {
public:
    Class(const std::string &name, const Class *super) : m_name(name), m_super(super) {}
    const std::string &getName() const { return m_name; }
    bool derivesFrom(const Class *c) const { for (const Class *k = this; k; k = k->m_super) if (k == c) return true; return false; }
private:
    std::string m_name;
    const Class *m_super;
};
#define MTS_CLASS(x) x::m_theClass
#define MTS_DECLARE_CLASS() \
    virtual const ::mitsuba::Class *getClass() const; \
    static ::mitsuba::Class *m_theClass;
#define MTS_IMPLEMENT_CLASS_S(name, abstract, super) \
    ::mitsuba::Class *name::m_theClass = new ::mitsuba::Class(#name, super::m_theClass); \
    const ::mitsuba::Class *name::getClass() const { return m_theClass; }
#define MTS_IMPLEMENT_CLASS(name, abstract, super) MTS_IMPLEMENT_CLASS_S(name, abstract, super)
// plugin.h / cobject.h:99-107
#define MTS_EXPORT_PLUGIN(name, descr) \
    extern "C" { \
        void *CreateInstance(const ::mitsuba::Properties &props) { return new name(props); } \
        const char *GetDescription() { return descr; } \
    }

class Object {
public:
    Object() : m_refs(0) {}
    virtual ~Object() {}
    void incRef() const { ++m_refs; }
    void decRef() const { if (--m_refs <= 0) delete this; }
    virtual const Class *getClass() const { return m_theClass; }
    static inline Class *m_theClass = new Class("Object", nullptr);
private:
    mutable int m_refs;
};

template <class T> class ref {
public:
    ref() : m_p(nullptr) {}
    ref(T *p) : m_p(p) { if (m_p) m_p->incRef(); }
    ref(const ref &o) : m_p(o.m_p) { if (m_p) m_p->incRef(); }
    ~ref() { if (m_p) m_p->decRef(); }
    ref &operator=(const ref &o) { if (o.m_p) o.m_p->incRef(); if (m_p) m_p->decRef(); m_p = o.m_p; return *this; }
    ref &operator=(T *p) { if (p) p->incRef(); if (m_p) m_p->decRef(); m_p = p; return *this; }
    T *operator->() const { return m_p; }
    T &operator*() const { return *m_p; }
    operator T *() const { return m_p; }
    T *get() const { return m_p; }
    bool operator==(const T *p) const { return m_p == p; }
private:
    T *m_p;
};
template <class T> class ref_vector : public std::vector<ref<T>> {};

// ---- small math
struct Vector { Float x, y, z; Vector() : x(0), y(0), z(0) {} Vector(Float a, Float b, Float c) : x(a), y(b), z(c) {} };
struct Point { Float x, y, z; Point() : x(0), y(0), z(0) {} Point(Float a, Float b, Float c) : x(a), y(b), z(c) {} };
typedef Vector Normal;
struct Vector2i { int x, y; Vector2i() : x(0), y(0) {} Vector2i(int a, int b) : x(a), y(b) {} };
struct AABB {              // aabb.h: TAABB<Point>
    typedef Point PointType;
    Point min, max;
    PointType getCenter() const { return Point(0.5 * (min.x + max.x), 0.5 * (min.y + max.y), 0.5 * (min.z + max.z)); }
};
struct Matrix4x4 {
    Float m[4][4];
    Matrix4x4() { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) m[r][c] = r == c; }
    Float &operator()(int r, int c) { return m[r][c]; }
    const Float &operator()(int r, int c) const { return m[r][c]; }
};
class Transform {
public:
    Transform() {}
    explicit Transform(const Matrix4x4 &m) : m_m(m) {}
    const Matrix4x4 &getMatrix() const { return m_m; }
    Transform operator*(const Transform &o) const {
        Matrix4x4 r;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { r(i, j) = 0; for (int k = 0; k < 4; ++k) r(i, j) += m_m(i, k) * o.m_m(k, j); }
        return Transform(r);
    }
    static Transform scale(const Vector &v) { Matrix4x4 r; r(0, 0) = v.x; r(1, 1) = v.y; r(2, 2) = v.z; return Transform(r); }
private:
    Matrix4x4 m_m;
};

namespace fs { // the reference's fs = boost::filesystem (fresolver.h:24): what the adaptor needs of a path -- made from a string, read back as one
class path {
public:
    path() {}
    path(const std::string &s) : m_s(s) {}
    path(const char *s) : m_s(s) {}
    const std::string &string() const { return m_s; }
private:
    std::string m_s;
};
} // namespace fs
class ContinuousSpectrum { // spectrum.h:45-90: what Spectrum::fromContinuousSpectrum takes
public:
    virtual ~ContinuousSpectrum() {}
};
class InterpolatedSpectrum : public ContinuousSpectrum { // spectrum.h: a tabulated spectrum read from an .spd file
public:
    explicit InterpolatedSpectrum(const fs::path &path) : m_path(path.string()) {}
    const std::string &path() const { return m_path; }
private:
    std::string m_path;
};
class Spectrum {
public:
    Spectrum() { s[0] = s[1] = s[2] = 0; }
    explicit Spectrum(Float v) { s[0] = s[1] = s[2] = v; }
    Spectrum(Float r, Float g, Float b) { s[0] = r; s[1] = g; s[2] = b; }
    void toLinearRGB(Float &r, Float &g, Float &b) const { r = s[0]; g = s[1]; b = s[2]; }
    Spectrum operator/(Float f) const { return Spectrum(s[0] / f, s[1] / f, s[2] / f); }
    // the fake knows no measured spectra: a file name carries "r_g_b" so that tests can see the lookup happen
    enum EConversionIntent { EReflectance, EIlluminant }; // spectrum.h:343-356
    void fromContinuousSpectrum(const ContinuousSpectrum &smooth) {
        const InterpolatedSpectrum &sp = dynamic_cast<const InterpolatedSpectrum &>(smooth);
        double r = 0, g = 0, b = 0;
        const size_t p = sp.path().find("fake:");
        if (p == std::string::npos || sscanf(sp.path().c_str() + p + 5, "%lf_%lf_%lf", &r, &g, &b) != 3) Log(EError, "cannot read %s", sp.path().c_str());
        s[0] = r; s[1] = g; s[2] = b;
    }
    Float s[3];
};

// ---- Properties (properties.h)
class Properties {
public:
    enum EPropertyType { EBoolean = 0, EInteger, EFloat, EPoint, ETransform, EAnimatedTransform, ESpectrum, EString, EData, EVector };
    void setString(const std::string &k, const std::string &v) { m_type[k] = EString; m_s[k] = v; }
    void setInteger(const std::string &k, int v) { m_type[k] = EInteger; m_f[k] = v; }
    void setFloat(const std::string &k, Float v) { m_type[k] = EFloat; m_f[k] = v; }
    void setBoolean(const std::string &k, bool v) { m_type[k] = EBoolean; m_f[k] = v; }
    void setTransform(const std::string &k, const Transform &v) { m_type[k] = ETransform; m_t[k] = v; }
    void setSpectrum(const std::string &k, const Spectrum &v) { m_type[k] = ESpectrum; m_sp[k] = v; }
    bool hasProperty(const std::string &k) const { return m_type.count(k) != 0; }
    EPropertyType getType(const std::string &k) const { need(k); return m_type.at(k); }
    std::string getString(const std::string &k) const { need(k, EString); return m_s.at(k); }
    std::string getString(const std::string &k, const std::string &d) const { return hasProperty(k) ? getString(k) : d; }
    int getInteger(const std::string &k) const { need(k, EInteger); return (int) m_f.at(k); }
    int getInteger(const std::string &k, const int &d) const { return hasProperty(k) ? getInteger(k) : d; }
    Float getFloat(const std::string &k) const { need(k, EFloat); return m_f.at(k); }
    Float getFloat(const std::string &k, const Float &d) const { return hasProperty(k) ? getFloat(k) : d; }
    bool getBoolean(const std::string &k) const { need(k, EBoolean); return m_f.at(k) != 0; }
    bool getBoolean(const std::string &k, const bool &d) const { return hasProperty(k) ? getBoolean(k) : d; }
    Transform getTransform(const std::string &k) const { need(k, ETransform); return m_t.at(k); }
    Transform getTransform(const std::string &k, const Transform &d) const { return hasProperty(k) ? getTransform(k) : d; }
    Spectrum getSpectrum(const std::string &k) const { need(k, ESpectrum); return m_sp.at(k); }
    Spectrum getSpectrum(const std::string &k, const Spectrum &d) const { return hasProperty(k) ? getSpectrum(k) : d; }
private:
    void need(const std::string &k) const { if (!hasProperty(k)) Log(EError, "Property \"%s\" has not been specified!", k.c_str()); }
    void need(const std::string &k, EPropertyType t) const {
        need(k);
        // an integer literal is accepted where a float is expected, as the XML loader does
        if (m_type.at(k) != t && !(t == EFloat && m_type.at(k) == EInteger)) Log(EError, "Property \"%s\" has the wrong type", k.c_str());
    }
    std::map<std::string, EPropertyType> m_type;
    std::map<std::string, std::string> m_s;
    std::map<std::string, Float> m_f;
    std::map<std::string, Transform> m_t;
    std::map<std::string, Spectrum> m_sp;
};

class Stream {
public:
    void write(const void *p, size_t n) { const char *c = (const char *) p; buf.insert(buf.end(), c, c + n); }
    void read(void *p, size_t n) { memcpy(p, buf.data() + pos, n); pos += n; }
    void writeInt(int v) { write(&v, sizeof v); }
    int readInt() { int v; read(&v, sizeof v); return v; }
    void writeBool(bool v) { char c = v; write(&c, 1); }
    bool readBool() { char c; read(&c, 1); return c != 0; }
    std::vector<char> buf;
    size_t pos = 0;
};
class InstanceManager {};

class ConfigurableObject : public Object {
public:
    ConfigurableObject() {}
    explicit ConfigurableObject(const Properties &p) : m_properties(p) {}
    const Properties &getProperties() const { return m_properties; } // cobject.h:77
    static inline Class *m_theClass = new Class("ConfigurableObject", Object::m_theClass);
protected:
    Properties m_properties;
};

class Sampler : public ConfigurableObject {
public:
    size_t getSampleCount() const { return m_sampleCount; } // sampler.h:156
    const Class *getClass() const { return m_class; }
    size_t m_sampleCount = 4;
    const Class *m_class = new Class("IndependentSampler", ConfigurableObject::m_theClass);
};
class ReconstructionFilter : public ConfigurableObject {
public:
    Float getRadius() const { return m_radius; } // rfilter.h:67
    const Class *getClass() const { return m_class; }
    Float m_radius = 0.5;
    const Class *m_class = nullptr;
};
class Bitmap : public Object {
public:
    enum EPixelFormat { ELuminance = 0, ERGB = 2, ESpectrum = 8 };
    enum EComponentFormat { EFloat16 = 5, EFloat32 = 6, EFloat64 = 7 };
    Bitmap(EPixelFormat pf, EComponentFormat cf, const Vector2i &size) : m_pf(pf), m_cf(cf), m_size(size), m_data((size_t) size.x * size.y * 3, 0.f) {}
    float *getFloat32Data() { return m_data.data(); }             // bitmap.h:1216
    const float *getFloat32Data() const { return m_data.data(); }
    // bitmap.h:685-687 (not a const member in the reference either)
    ref<Bitmap> convert(EPixelFormat, EComponentFormat, Float gamma = 1.0f, Float multiplier = 1.0f, Spectrum::EConversionIntent intent = Spectrum::EReflectance) {
        (void) gamma; (void) multiplier; (void) intent;
        Bitmap *b = new Bitmap(m_pf, m_cf, m_size); b->m_data = m_data; return b;
    }
    const Vector2i &getSize() const { return m_size; }
private:
    EPixelFormat m_pf; EComponentFormat m_cf; Vector2i m_size;
    std::vector<float> m_data;
};
class Film : public ConfigurableObject {
public:
    const Vector2i &getCropSize() const { return m_cropSize; }                                   // film.h:39
    const ReconstructionFilter *getReconstructionFilter() const { return m_filter.get(); }      // film.h:115
    virtual void setBitmap(const Bitmap *bitmap, Float multiplier = 1.0f) {                      // film.h:51
        (void) multiplier;
        m_result.assign(bitmap->getFloat32Data(), bitmap->getFloat32Data() + (size_t) m_cropSize.x * m_cropSize.y * 3);
    }
    Vector2i m_cropSize;
    ref<ReconstructionFilter> m_filter;
    std::vector<float> m_result;
};
class Sensor : public ConfigurableObject {
public:
    enum { ENeedsApertureSample = 0x10 };
    Film *getFilm() { return m_film; }
    const Film *getFilm() const { return m_film.get(); }
    Sampler *getSampler() { return m_sampler; }
    const Sampler *getSampler() const { return m_sampler.get(); }
    bool needsApertureSample() const { return m_type & ENeedsApertureSample; } // sensor.h:303
    static inline Class *m_theClass = new Class("Sensor", ConfigurableObject::m_theClass);
    ref<Film> m_film;
    ref<Sampler> m_sampler;
    int m_type = 0;
};
class PerspectiveCamera : public Sensor {
public:
    const Transform getWorldTransform(Float t) const { (void) t; return m_toWorld; } // sensor.h:403
    Float getNearClip() const { return m_nearClip; }                                  // sensor.h:443
    Float getFarClip() const { return m_farClip; }
    Float getXFov() const { return m_xfov; }                                          // sensor.h:499
    const Class *getClass() const { return m_theClass; }
    static inline Class *m_theClass = new Class("PerspectiveCamera", Sensor::m_theClass);
    Transform m_toWorld;
    Float m_nearClip = 1e-2, m_farClip = 1e4, m_xfov = 90;
};

struct Intersection {};
class BSDF : public ConfigurableObject {
public:
    BSDF() {}
    explicit BSDF(const Properties &p) : ConfigurableObject(p) {}
    virtual Spectrum getDiffuseReflectance(const Intersection &) const { return m_properties.getSpectrum("reflectance", Spectrum(0.5)); } // bsdf.h:337
    const Class *getClass() const { return m_class; }
    static inline Class *m_theClass = new Class("BSDF", ConfigurableObject::m_theClass);
    const Class *m_class = nullptr;
};
class Emitter : public ConfigurableObject {
public:
    explicit Emitter(const Properties &p) : ConfigurableObject(p), m_samplingWeight(p.getFloat("samplingWeight", 1.0)) {}
    Float getSamplingWeight() const { return m_samplingWeight; } // emitter.h:536
    const Class *getClass() const { return m_class; }
    Float m_samplingWeight;
    const Class *m_class = nullptr;
};
class Shape : public ConfigurableObject {
public:
    Shape() {}
    explicit Shape(const Properties &p) : ConfigurableObject(p) {}
    bool isEmitter() const { return m_emitter.get() != NULL; }   // shape.h:458
    const Emitter *getEmitter() const { return m_emitter.get(); } // shape.h:462
    const BSDF *getBSDF() const { return m_bsdf.get(); }          // shape.h:476
    virtual AABB getAABB() const { return m_aabb; }               // shape.h:223
    virtual std::string getName() const { return m_name; }        // shape.h:198
    const Class *getClass() const { return m_class; }
    static inline Class *m_theClass = new Class("Shape", ConfigurableObject::m_theClass);
    ref<Emitter> m_emitter;
    ref<BSDF> m_bsdf;
    AABB m_aabb;
    std::string m_name = "shape";
    const Class *m_class = nullptr;
};
struct Triangle { uint32_t idx[3]; };
class TriMesh : public Shape {
public:
    size_t getTriangleCount() const { return m_tris.size(); }                                   // trimesh.h:122
    const Triangle *getTriangles() const { return m_tris.data(); }                               // trimesh.h:127
    const Point *getVertexPositions() const { return m_pos.data(); }                             // trimesh.h:132
    const Normal *getVertexNormals() const { return m_normals.empty() ? NULL : m_normals.data(); } // trimesh.h:137
    static inline Class *m_theClass = new Class("TriMesh", Shape::m_theClass);
    std::vector<Triangle> m_tris;
    std::vector<Point> m_pos;
    std::vector<Normal> m_normals;
};
class Subsurface : public ConfigurableObject {};

class RenderJob : public Object {
public:
    static int getID() { return 7; } // thread.h:95 (RenderJob is a Thread)
};
class RenderQueue : public Object {
public:
    void signalRefresh(const RenderJob *) { ++refreshes; } // renderqueue.h:105
    int refreshes = 0;
};
class Scene : public ConfigurableObject {
public:
    Sensor *getSensor() { return m_sensor; }                                                    // scene.h:1011
    const Sensor *getSensor() const { return m_sensor.get(); }
    ref_vector<Shape> &getShapes() { return m_shapes; }                                          // scene.h:1103
    const ref_vector<Shape> &getShapes() const { return m_shapes; }
    const ref_vector<Subsurface> &getSubsurfaceIntegrators() const { return m_ss; }             // scene.h:1096
    ref<Sensor> m_sensor;
    ref_vector<Shape> m_shapes;
    ref_vector<Subsurface> m_ss;
};

class Integrator : public ConfigurableObject {
public:
    explicit Integrator(const Properties &p) : ConfigurableObject(p) {}
    Integrator(Stream *, InstanceManager *) {}
    virtual void serialize(Stream *, InstanceManager *) const {}
    virtual bool preprocess(const Scene *, RenderQueue *, const RenderJob *, int, int, int) { return true; } // integrator.h:61
    virtual bool render(Scene *, RenderQueue *, const RenderJob *, int, int, int) = 0;                        // integrator.h:77
    virtual void cancel() = 0;
    static inline Class *m_theClass = new Class("Integrator", ConfigurableObject::m_theClass);
};

class ProgressReporter { // statistics.h:287-311
public:
    ProgressReporter(const std::string &title, long long total, const void *ptr) : m_total(total) { (void) title; (void) ptr; }
    void update(long long value) { updates().push_back(value); (void) m_total; }
    void finish() { updates().push_back(-1); }
    static std::vector<long long> &updates() { static std::vector<long long> u; return u; }
private:
    long long m_total;
};

class FileResolver : public Object {
public:
    fs::path resolve(const fs::path &path) const { return fs::path(prefix() + path.string()); }
    static std::string &prefix() { static std::string s; return s; }
};
class Thread {
public:
    static Thread *getThread() { static Thread t; return &t; }
    FileResolver *getFileResolver() { static ref<FileResolver> r = new FileResolver(); return r; }
};

struct BidirectionalUtils { // bidir/util.h:40: the separate direct-illumination pass stays on the host integrator
    static ref<Bitmap> renderDirectComponent(Scene *scene, int, int, RenderQueue *, const RenderJob *, size_t directSamples) {
        const Vector2i size = scene->getSensor()->getFilm()->getCropSize();
        ref<Bitmap> b = new Bitmap(Bitmap::ESpectrum, Bitmap::EFloat32, size);
        for (size_t i = 0; i < (size_t) size.x * size.y * 3; ++i) b->getFloat32Data()[i] = 0.25f; // recognisable constant
        calls() += (int) directSamples;
        return b;
    }
    static int &calls() { static int c = 0; return c; }
};

} // namespace mitsuba
