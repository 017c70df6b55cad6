"""placeholder"""
class DrmltError(RuntimeError):
    pass
class Context:
    pass
def build_native(force=False):
    pass
def library_path():
    return None
