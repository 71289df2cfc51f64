"""Minimal stand-in for the reference's `dnnlib` namespace: only what the generator-forward path touches
(reference dnnlib/util.py:44 EasyDict, :305 construct_class_by_name)."""
import importlib


class EasyDict(dict):
    """dict with attribute access."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        del self[name]


def get_obj_by_name(name):
    module, _, attr = name.rpartition('.')
    return getattr(importlib.import_module(module), attr)


def construct_class_by_name(*args, class_name=None, **kwargs):
    return get_obj_by_name(class_name)(*args, **kwargs)


class _Util:
    EasyDict = EasyDict
    construct_class_by_name = staticmethod(construct_class_by_name)
    get_obj_by_name = staticmethod(get_obj_by_name)


util = _Util
