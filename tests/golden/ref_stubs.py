"""sys.modules stand-ins that let the reference's torso-model package import in this container: cv2, imageio,
torchvision, timm, kornia, mmcv, pretrainedmodels, torchshow are not installed; nothing on the path under test calls into them."""
import importlib.machinery
import sys
import types

NAMES = ["cv2", "imageio", "torchvision", "torchvision.models", "torchvision.transforms", "torchvision.models.resnet", "timm",
         "timm.models", "timm.models.layers", "timm.models.vision_transformer", "timm.models.registry", "kornia", "mmcv",
         "pretrainedmodels", "torchshow"]


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        m = _Stub(self.__name__ + "." + name)
        setattr(self, name, m)
        return m

    def __call__(self, *a, **k):
        return None


def install():
    for name in NAMES:
        if name not in sys.modules:
            m = _Stub(name)
            m.__spec__ = importlib.machinery.ModuleSpec(name, None)
            m.__path__ = []
            sys.modules[name] = m
