"""String -> constructor registries = the reference's plugin API (SURVEY §8b Tier 1):
META_ARCH_REGISTRY (uwsod/detectron2/modeling/meta_arch/build.py:6-23), BACKBONE_REGISTRY
(modeling/backbone/build.py:20), ROI_HEADS_REGISTRY (modeling/roi_heads/roi_heads.py:38-43),
ROI_BOX_HEAD_REGISTRY (modeling/roi_heads/box_head.py:112-117).  Entries are selected by YAML strings."""


class Registry:
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._do_register(o.__name__, o)
                return o
            return deco
        self._do_register(obj.__name__, obj)
        return obj

    def _do_register(self, name, obj):
        assert name not in self._obj_map, "An object named '{}' was already registered in '{}' registry!".format(
            name, self._name)
        self._obj_map[name] = obj

    def get(self, name):
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError("No object named '{}' found in '{}' registry!".format(name, self._name))
        return ret

    def __contains__(self, name):
        return name in self._obj_map


META_ARCH_REGISTRY = Registry("META_ARCH")
BACKBONE_REGISTRY = Registry("BACKBONE")
ROI_HEADS_REGISTRY = Registry("ROI_HEADS")
ROI_BOX_HEAD_REGISTRY = Registry("ROI_BOX_HEAD")
