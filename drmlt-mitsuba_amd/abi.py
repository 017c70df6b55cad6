"""ctypes mirror of include/drmlt_abi.h (struct layouts and enum values are ABI)."""
import ctypes as C

TECH_PATH, TECH_BDPT, TECH_MMLT = 0, 1, 2
TYPE_GREEN, TYPE_MIRA, TYPE_ORBITAL = 0, 1, 2
ALGO_DRMLT, ALGO_PSSMLT = 0, 1
SHAPE_TRIANGLE, SHAPE_RECTANGLE, SHAPE_SPHERE = 0, 1, 2
BSDF_DIFFUSE, BSDF_DIELECTRIC, BSDF_ROUGHCONDUCTOR, BSDF_CONDUCTOR = 0, 1, 2, 3
EMITTER_AREA = 0
FILTER_BOX, FILTER_GAUSSIAN = 0, 1
SEED_TARGET, SEED_REFERENCE = 0, 1            # drmlt_config.seed_rule (two-stage MLT seeding)
WORK_UNITS_DEVICE, WORK_UNITS_REFERENCE = 0, 1  # drmlt_config.work_units_rule (what workUnits = -1 derives)

ABI_VERSION = 4  # include/drmlt_abi.h: DRMLT_ABI_VERSION
OK, E_INVALID, E_DEVICE, E_STATE, E_ZERO_LUM, E_REPLAY, E_CANCELLED = 0, -1, -2, -3, -4, -5, -6

TYPE_NAMES = {"green": TYPE_GREEN, "mira": TYPE_MIRA, "orbital": TYPE_ORBITAL, "mirasym": TYPE_ORBITAL}
TECH_NAMES = {"path": TECH_PATH, "bdpt": TECH_BDPT, "mmlt": TECH_MMLT}
SEED_RULE_NAMES = {"target": SEED_TARGET, "reference": SEED_REFERENCE}
WORK_UNITS_RULE_NAMES = {"device": WORK_UNITS_DEVICE, "reference": WORK_UNITS_REFERENCE}


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("algo", C.c_int32), ("technique", C.c_int32), ("type", C.c_int32),
        ("max_depth", C.c_int32), ("rr_depth", C.c_int32), ("direct_samples", C.c_int32),
        ("luminance_samples", C.c_int32), ("work_units", C.c_int32), ("sample_count", C.c_int32),
        ("p_large", C.c_float), ("sigma", C.c_float), ("scale_second", C.c_float),
        ("average_luminance", C.c_float),
        ("acceptance_map", C.c_int32), ("timid_after_large", C.c_int32), ("fix_emitter_path", C.c_int32),
        ("use_mixture", C.c_int32), ("kelemen_style_weights", C.c_int32), ("kelemen_style_mutation", C.c_int32),
        ("no_light_image", C.c_int32), ("timeout_s", C.c_int32), ("no_direct_sampling", C.c_int32),
        ("seed_rule", C.c_int32), ("work_units_rule", C.c_int32), ("reserved", C.c_int32 * 3),
    ]


class Shape(C.Structure):
    _fields_ = [("type", C.c_int32), ("bsdf", C.c_int32), ("emitter", C.c_int32), ("reserved", C.c_int32),
                ("data", C.c_float * 12)]


class Bsdf(C.Structure):
    _fields_ = [("type", C.c_int32), ("rgb", C.c_float * 3), ("p", C.c_float * 8)]


class Emitter(C.Structure):
    _fields_ = [("type", C.c_int32), ("shape", C.c_int32), ("radiance", C.c_float * 3),
                ("sampling_weight", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("to_world", C.c_float * 16), ("fov_x_deg", C.c_float), ("near_clip", C.c_float),
                ("far_clip", C.c_float), ("width", C.c_int32), ("height", C.c_int32), ("filter", C.c_int32),
                ("filter_param", C.c_float)]


class Scene(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_shapes", C.c_int32), ("n_bsdfs", C.c_int32),
                ("n_emitters", C.c_int32), ("shapes", C.POINTER(Shape)), ("bsdfs", C.POINTER(Bsdf)),
                ("emitters", C.POINTER(Emitter)), ("camera", Camera)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "first_acc", "first_base", "large_acc", "large_base", "bold_acc", "bold_base",
        "second_acc", "second_base", "second_large_acc", "second_large_base",
        "second_bold_acc", "second_bold_base", "overall_acc", "overall_base",
        "mutations", "path_evals", "rays", "accepted")] + [
        ("kernel_ms", C.c_double), ("seed_ms", C.c_double), ("n_chains", C.c_uint32), ("max_dim", C.c_uint32),
        ("launches", C.c_uint64), ("bvh_node_visits", C.c_uint64), ("bvh_prim_tests", C.c_uint64),
        ("bvh_node_iterations", C.c_uint64), ("bvh_leaf_iterations", C.c_uint64)]

    RATIOS = ("first", "large", "bold", "second", "second_large", "second_bold", "overall")

    def ratios(self):
        out = {}
        for k in self.RATIOS:
            base = getattr(self, k + "_base")
            out[k] = (getattr(self, k + "_acc") / base) if base else float("nan")
        return out

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Splat(C.Structure):
    _fields_ = [("luminance", C.c_float), ("x", C.c_float), ("y", C.c_float), ("rgb", C.c_float * 3),
                ("n_dims", C.c_int32), ("n_rays", C.c_int32)]


PROGRESS_CB = C.CFUNCTYPE(None, C.c_uint64, C.c_uint64, C.c_void_p)


def make_config(**kw):
    """Defaults follow DRMLT's ctor (reference src/integrators/drmlt/drmlt.cpp:193-349)."""
    c = Config()
    c.struct_size = C.sizeof(Config)
    c.algo = ALGO_DRMLT
    c.technique = TECH_PATH
    c.type = TYPE_ORBITAL
    c.max_depth = -1
    c.rr_depth = 5
    c.direct_samples = 16
    c.luminance_samples = 100000
    c.work_units = -1
    c.sample_count = 1
    c.p_large = 0.3
    c.sigma = 1.0 / 64.0
    c.scale_second = 0.1
    c.average_luminance = -1.0
    c.kelemen_style_weights = 1
    c.kelemen_style_mutation = 1
    for k, v in kw.items():
        if k == "type" and isinstance(v, str):
            v = TYPE_NAMES[v]
        if k == "technique" and isinstance(v, str):
            v = TECH_NAMES[v]
        if k == "seed_rule" and isinstance(v, str):
            v = SEED_RULE_NAMES[v]
        if k == "work_units_rule" and isinstance(v, str):
            v = WORK_UNITS_RULE_NAMES[v]
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c
