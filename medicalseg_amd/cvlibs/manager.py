"""Component registries: the reference's plug-in mechanism (medicalseg/cvlibs/manager.py:23-149).

YAML ``type: Name`` is resolved against MODELS, BACKBONES, DATASETS, TRANSFORMS, LOSSES in
that order (cvlibs/config.py:371-382).  Registration is by ``__name__``; re-registration
warns and replaces, non class/function objects raise TypeError -- same observable behaviour
as the reference."""
import inspect
import warnings
from collections.abc import Sequence


class ComponentManager:
    def __init__(self, name=None):
        self._components_dict = {}
        self._name = name

    def __len__(self):
        return len(self._components_dict)

    def __repr__(self):
        label = self._name if self._name else type(self).__name__
        return "{}:{}".format(label, list(self._components_dict))

    def __getitem__(self, item):
        try:
            return self._components_dict[item]
        except KeyError:
            raise KeyError("{} does not exist in availabel {}".format(item, self)) from None

    def __contains__(self, item):
        return item in self._components_dict

    @property
    def components_dict(self):
        return self._components_dict

    @property
    def name(self):
        return self._name

    def _add_single_component(self, component):
        if not (inspect.isclass(component) or inspect.isfunction(component)):
            raise TypeError("Expect class/function type, but received {}".format(type(component)))
        key = component.__name__
        if key in self._components_dict:
            warnings.warn("{} exists already! It is now updated to {} !!!".format(key, component))
        self._components_dict[key] = component

    def add_component(self, components):
        """Usable as a decorator or with a class/function or a sequence of them."""
        if isinstance(components, Sequence):
            for c in components:
                self._add_single_component(c)
        else:
            self._add_single_component(components)
        return components


MODELS = ComponentManager("models")
BACKBONES = ComponentManager("backbones")
DATASETS = ComponentManager("datasets")
TRANSFORMS = ComponentManager("transforms")
LOSSES = ComponentManager("losses")
