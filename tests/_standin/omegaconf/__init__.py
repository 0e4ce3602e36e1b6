"""TEST-ONLY stand-in for the `omegaconf` package (absent from this image; no network).

It carries no arithmetic: just enough of the container API that the reference's entry point uses
(uav_dcc_control/train.py:12-26: `OmegaConf.load`, `OmegaConf.merge`, attribute get/set on the merged
`DictConfig`) plus what this package's Learner calls on such an object (`OmegaConf.is_config`,
`OmegaConf.to_container`).  tests/test_train_entry.py puts this directory on PYTHONPATH to drive
`Learner(cfg)` with a DictConfig-shaped object the way an unchanged train.py does; product code never imports it
(on a box with the real omegaconf the real one is used).
"""
import re

import yaml


class _Loader(yaml.SafeLoader):
    pass


# OmegaConf resolves `5e-4` as a float (YAML 1.2 core schema); PyYAML's 1.1 resolver needs the dot
_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"""^(?:[-+]?(?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?
                    |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
                    |\.[0-9_]+(?:[eE][-+][0-9]+)?
                    |[-+]?\.(?:inf|Inf|INF)|\.(?:nan|NaN|NAN))$""", re.X),
    list("-+0123456789."))


class DictConfig(object):
    """Attribute- and item-addressable mapping; deliberately NOT a dict subclass (like the real one)."""

    def __init__(self, content=None):
        object.__setattr__(self, "_content", dict(content or {}))

    def __getattr__(self, k):
        try:
            return object.__getattribute__(self, "_content")[k]
        except KeyError:
            raise AttributeError("Missing key %s" % k)

    def __setattr__(self, k, v):
        self._content[k] = v

    def __getitem__(self, k):
        return self._content[k]

    def __setitem__(self, k, v):
        self._content[k] = v

    def __contains__(self, k):
        return k in self._content

    def __iter__(self):
        return iter(self._content)

    def keys(self):
        return self._content.keys()

    def items(self):
        return self._content.items()

    def __len__(self):
        return len(self._content)


class OmegaConf(object):
    @staticmethod
    def load(path):
        with open(path) as f:
            return DictConfig(yaml.load(f, Loader=_Loader))

    @staticmethod
    def merge(*cfgs):
        out = {}
        for c in cfgs:          # later configs win
            out.update(c._content if isinstance(c, DictConfig) else dict(c))
        return DictConfig(out)

    @staticmethod
    def create(d=None):
        return DictConfig(d)

    @staticmethod
    def is_config(obj):
        return isinstance(obj, DictConfig)

    @staticmethod
    def to_container(cfg, resolve=False):
        return dict(cfg._content)
