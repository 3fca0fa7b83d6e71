"""Import the upstream reference (read-only at /root/reference) in the BUILD container only.

The reference needs torchvision / crp / zennit at import time for code that is
outside the concept-DB hot path.  We register empty
placeholder modules for those names so that the *unmodified* hot-path modules
(lens, scores, activation_based, activation_caching, aggregators,
foundation_models.base) import and run.  Nothing here is shipped or used on the
GPU box: it is only used by make_golden.py to generate fixtures.
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def import_reference():
    if "semanticlens" in sys.modules:
        return sys.modules["semanticlens"]
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional"):
        if name not in sys.modules:
            _stub(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    sys.modules["torchvision.transforms.functional"].gaussian_blur = None

    class _Any:  # base class placeholder for crp.visualization.FeatureVisualization
        pass

    _stub("crp")
    _stub("crp.concepts", ChannelConcept=_Any)
    _stub("crp.helper", load_maximization=None)
    _stub("crp.visualization", FeatureVisualization=_Any)
    _stub("crp.image", get_crop_range=None, imgify=None)
    _stub("crp.maximization")
    _stub("crp.statistics")
    _stub("zennit")
    _stub("zennit.composites", EpsilonPlusFlat=_Any)
    _stub("zennit.core", stabilize=None)
    sys.path.insert(0, REFERENCE_ROOT)
    import semanticlens  # noqa: E402

    return semanticlens
