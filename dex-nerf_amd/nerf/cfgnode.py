"""Minimal attribute-dict config node (the hot path only does attribute reads; reference nerf/cfgnode.py is a
YACS-style class).  Supports nested dicts, attribute and item access, YAML loading and dumping."""
import copy

import yaml


class CfgNode(dict):
    def __init__(self, init_dict=None, key_list=None, new_allowed=False):
        super().__init__()
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as exc:
            raise AttributeError(name) from exc

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return CfgNode(copy.deepcopy(dict(self), memo))

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, CfgNode) else v) for k, v in self.items()}

    def dump(self, **kwargs):
        return yaml.safe_dump(self.to_dict(), **kwargs)

    @classmethod
    def load_yaml_with_base(cls, filename):
        with open(filename, "r") as f:
            return cls(yaml.safe_load(f))

    def merge_from_file(self, filename):
        with open(filename, "r") as f:
            other = yaml.safe_load(f)

        def merge(dst, src):
            for k, v in src.items():
                if isinstance(v, dict) and isinstance(dst.get(k), dict):
                    merge(dst[k], v)
                else:
                    dst[k] = CfgNode(v) if isinstance(v, dict) else v
        merge(self, other)
