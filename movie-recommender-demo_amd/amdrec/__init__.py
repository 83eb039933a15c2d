"""amdrec - MI355X-native retrieval + ranking hot path (host-side mirror of the reference's
Python surfaces over the libamdrec C ABI).  See DESIGN.md / INTEGRATION.md."""
from . import synth  # noqa: F401  (numpy only)

__all__ = ["synth"]


def __getattr__(name):
    # torch-dependent modules are imported lazily so that `import amdrec.synth` stays light
    import importlib
    if name in ("index", "towers", "ranker", "pipeline", "ivf", "sharded", "_lib", "weights"):
        return importlib.import_module(f"{__name__}.{name}")
    if name == "FAISSIndex":
        return importlib.import_module(f"{__name__}.index").FAISSIndex
    raise AttributeError(name)
