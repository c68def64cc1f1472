"""Name -> class registries with the lookup contract of basicsr/utils/registry.py:4-88:
``@REG.register()`` registers under ``__name__``, duplicate names are an error, ``get`` raises
KeyError for unknown names, registries are iterable / support ``in`` / ``keys()``."""


class Registry:

    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    @property
    def name(self):
        return self._name

    def _add(self, obj, suffix=None):
        key = obj.__name__ if suffix is None else f'{obj.__name__}_{suffix}'
        if key in self._obj_map:
            raise AssertionError(f"An object named '{key}' was already registered in '{self._name}' registry!")
        self._obj_map[key] = obj
        return obj

    def register(self, obj=None, suffix=None):
        if obj is not None:          # REG.register(cls)
            return self._add(obj, suffix)
        return lambda o: self._add(o, suffix)   # @REG.register()

    def get(self, name, suffix='basicsr'):
        found = self._obj_map.get(name)
        if found is None:
            found = self._obj_map.get(f'{name}_{suffix}')
        if found is None:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return found

    def __contains__(self, name):
        return name in self._obj_map

    def __iter__(self):
        return iter(self._obj_map.items())

    def keys(self):
        return self._obj_map.keys()


ARCH_REGISTRY = Registry('arch')
MODEL_REGISTRY = Registry('model')
